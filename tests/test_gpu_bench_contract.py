"""bench.py's line as the driver reads it: one JSON object on stdout and nothing else, the contract's keys with the values they
must have at N = 1, `roofline` consistent with the kernel time it quotes.  (The CPU baselines are left to the default run.)"""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_line_at_one_gpu():
    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    env = dict(os.environ, MPPI_BENCH_SETUP_SOLVES="20")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "10", "--warmup", "3",
                        "--no-cpu-baseline", "--no-extras"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1, r.stdout[-1000:]
    d = json.loads(lines[0])
    assert d["metric"] == "sample_steps_per_sec" and d["unit"] == "sample-steps/s" and d["higher_is_better"] is True
    assert (d["n_gpus"], d["steps"], d["warmup"]) == (1, 10, 3)
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "racing" in d["config"]["workload"] and "model" not in d["config"]
    N, T = 1 << 20, 50
    assert d["value"] == pytest.approx(N * T / (d["ms_per_step"] * 1e-3), rel=1e-9)
    assert 0.05 < d["ms_per_step"] < 1.0
    rf = d["roofline"]
    assert rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert rf["frac"] == pytest.approx(rf["achieved"] / rf["peak"], rel=1e-9)
    # algorithmic bytes of one launch (8.08 B per sample-step) over the kernel's own time, which one solve contains
    assert rf["algorithmic_bytes_per_launch"] == 423624704
    assert rf["achieved"] == pytest.approx(rf["algorithmic_bytes_per_launch"] / (rf["kernel_ms"] * 1e-3) / 1e9, rel=1e-6)
    assert rf["kernel_ms"] <= d["ms_per_step"]
    assert rf["traffic"] is None or rf["traffic"] > 0
    assert isinstance(rf["traffic_stale"], bool)
    # the headline is the product out of the box: no extension keyword, state_seq completed inside the solve
    assert d["config"]["solver_kwargs"] == {} and d["config"]["state_seq"].startswith("default")
    # ... and a longer region on the same solver does not contradict it: a timed region only ever ADDS a fixed cost, so the
    # best of three 200-step regions must not be slower per step than the contract's K-step one (1 % for the box's jitter)
    lr = d["long_run"]
    assert lr["same_solver_and_timing_as_headline"] and len(lr["repetitions"]) == 3
    assert lr["ms_per_step"] <= d["ms_per_step"] * 1.01, (lr, d["ms_per_step"])
