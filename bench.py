#!/usr/bin/env python3
"""bench.py — MPPI hot-path throughput on MI355X (BASELINE.json metric).

Workload (config.workload): racing kinematic-bicycle, horizon T=50, num_samples N=1,048,576 per GPU,
lambda=1.0 (BASELINE configs[2] = C3; with --gpus G it is C4: G*N samples sharded over G ranks, weak
scaling, ONE exchange of 4+T*dc floats per solve).  A "step" is one MPPI solve = one pass of the hot
path: sample -> rollout+cost -> weights+reduce -> finalize, with every input resident in HBM.

    python bench.py                       # 1 GPU, 200 steps, 20 warm-up
    python bench.py --gpus 8              # launches its own 8 ranks (one per GPU, RCCL; every exchange transport is timed with
                                          # the full K steps, the best complete run is `value`); or, equivalently,
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
        --master-port 29500 bench.py --gpus 8 --steps 50 --warmup 10
    MPPI_BENCH_ONE_DEVICE=1 MPPI_BENCH_BACKEND=gloo python bench.py --gpus 2   # dry run of the N>1 path on ONE GPU

Prints ONE JSON line on rank 0.  `value` = sample-steps/s (N_total * T * solves/s), whole job.
`roofline` is for the dominant kernel (rollout_cost_kernel), measured with HIP events on the launch
stream inside the timed region.  At N=1, rank 0 also reports, in the same line:
  cpu_baseline        the oracle (C restatement of the reference algorithm, OpenMP over samples) on this host
  cpu_baseline_torch  the reference's own op structure (per-timestep Python loops of batched torch ops) over the same
                      plugins on this host: best of a thread-count scan + the all-cores figure (SURVEY 8d / BASELINE.md 3)
  closed_loop         >= 100 racing control ticks: reference window recomputed, solve, a[0] applied (example/racing.py:221-266)
  other_configs       solve times of BASELINE configs C1 / C2 / C5
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6290 GB/s measured copy
SETUP_SOLVES = int(os.environ.get("MPPI_BENCH_SETUP_SOLVES", "200"))  # un-timed solves before the contract's warm-up: leave the idle power state (see main)


_LINE_FD = None


def own_stdout():
    """stdout carries ONE thing, the JSON line.  Libraries that print from C get stderr instead: RCCL writes a version
    banner at the first communicator of a process, into a stdio buffer that is flushed at exit — i.e. AFTER the line."""
    global _LINE_FD
    if _LINE_FD is None:
        sys.stdout.flush()
        _LINE_FD = os.dup(1)
        os.dup2(2, 1)


def print_line(text: str) -> None:
    data = (text + "\n").encode()
    fd = 1 if _LINE_FD is None else _LINE_FD
    while data:
        data = data[os.write(fd, data):]


def null_line(args, world, reason, transports=()):
    """A contract-shaped line for a multi-GPU run that produced NO complete measurement (`value` null): the keys the driver
    reads, the reason, and what every transport that was tried reported — so that a hang in a communicator's set-up or first
    collective costs one run, not the evidence of what happened."""
    return {"metric": "sample_steps_per_sec", "value": None, "unit": "sample-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "racing kinematic-bicycle MPPI solve (BASELINE configs[2]; configs[3] when n_gpus = 8)",
                       "num_samples_per_gpu": args.samples, "horizon": args.horizon, "lambda": 1.0},
            "error": reason,
            "transports": [{"exchange": r.get("exchange"), "requested": r.get("requested", r.get("exchange")),
                            "error": r.get("error", None if r.get("finite", True) else "non-finite outputs"),
                            "ms_per_step": None if "dt" not in r else r["dt"] / args.steps * 1e3} for r in transports]}


class Watchdog:
    """Wall-clock guard of the multi-GPU legs.  arm(seconds, phase) (re)starts the clock for one phase — process-group
    set-up, each transport's set-up + self-test + timed run, the strong-scaling leg; when a phase outlives its budget
    `on_expiry(phase)` runs on the timer thread (rank 0 prints the best complete run so far, or a null line) and the
    process ends through `exit_fn` — a rank blocked inside a collective cannot be unwound any other way."""

    def __init__(self, on_expiry, exit_fn=os._exit):
        import threading

        self._threading, self._on_expiry, self._exit, self._timer, self.phase = threading, on_expiry, exit_fn, None, None

    def arm(self, seconds, phase):
        self.cancel()
        self.phase = phase
        self._timer = self._threading.Timer(seconds, self._fire, args=(phase, seconds))
        self._timer.daemon = True
        self._timer.start()

    def _fire(self, phase, seconds):
        code = 1
        try:
            code = self._on_expiry(phase, seconds)
        finally:
            self._exit(code)

    def cancel(self):
        if self._timer is not None:
            self._timer.cancel()
            self._timer = None


# ---- the predicted 1 -> 8 GPU curve (no 8-GPU node has been available to any round: SCALE_r01..r05 are "skipped") ----------
# Every input is either MEASURED on one MI355X (source named) or an ASSUMPTION about a path that has never crossed a device
# boundary (marked, with a range).  DESIGN.md section 5 carries the same table; the first multi-GPU line prints
# `predicted` next to what it measured so that the record judges itself.
SCALING_INPUTS = {
    # one unsharded solve of 2^20 samples on the product's defaults (BENCH_r05.json `eager_state_seq`; replaced by the live
    # value whenever this run measured one)
    "solve_ms_1gpu": {"value": 0.14518, "source": "measured: BENCH_r05.json eager_state_seq.ms_per_step"},
    # what the sharded code path adds with ONE rank on the real RCCL backend (summary -> exchange -> combine of 1 shard)
    "exchange_1rank_us": {"rccl": 3.2, "nccl": 11.6, "p2p": 3.0,
                          "source": "measured: profiles/r05_visitF_exchange_single_rank.txt (rccl = in-library ncclAllGather on "
                                    "the solve's stream, nccl = torch.distributed all_gather); p2p = assumed equal to rccl's "
                                    "(same summarize/finalize kernels, a store + a poll instead of the collective)"},
    # latency of a 416-B all_gather among W ranks beyond what one rank already pays (RCCL LL protocol over xGMI: a ring of
    # W - 1 steps, each one xGMI hop ~ 2-3 us end to end for an 8-byte-flagged line) - ASSUMED, never measured here
    "collective_extra_us": {"rccl": {2: 4.0, 4: 9.0, 8: 18.0}, "nccl": {2: 4.0, 4: 9.0, 8: 18.0}, "p2p": {2: 2.5, 4: 3.0, 8: 4.0},
                            "range_factor": [0.5, 2.0],
                            "source": "ASSUMED: ring all_gather = (W-1) hops x ~2.5 us (xGMI store + poll of an LL line); p2p = every "
                                      "rank stores its 52 cells into all peers at once (one xGMI round, growing with fan-out) and "
                                      "polls its own buffer"},
    # the slowest of W ranks finishes later than the average one (kernel-time jitter of the 121 us rollout: ~1 %)
    "straggler_us": {2: 0.6, 4: 1.0, 8: 1.3, "source": "measured spread of rollout_cost_kernel over 1 300 launches (profiles/"
                                                         "r05_visitF_c3_kernel_stats_pmc.md: 121.15 us average, ~1 % sigma) x the "
                                                         "expected maximum of W normal draws"},
    # strong scaling: the same 2^20 samples split W ways - one solve of 2^20 / W samples on one GPU
    "solve_ms_by_local_samples": {1048576: 0.14518, 524288: 0.0790, 262144: 0.0495, 131072: 0.0357,
                                  "source": "measured: profiles/r05_experiments.md (racing N = 262 144 / 131 072: 49.5 / 35.7 us per "
                                            "solve, rollout 39.3 / 25.6 us - the lone-wave regime); 524 288 interpolated (rollout "
                                            "~ 123.5 / 2 + 2 us)"},
}


def predict_scaling(solve_ms_1gpu=None, exchange_1rank_us=None):
    """Predicted weak- and strong-scaling curve of the metric's workload (racing, 2^20 samples per GPU, T = 50, lambda = 1) at
    W = 1, 2, 4, 8 per transport, from SCALING_INPUTS (live one-GPU measurements override the committed ones).  Weak:
    t(W) = t_solve + exchange(1 rank) + collective_extra(W) + straggler(W); value = W * 2^20 * 50 / t.  Strong: the same with
    t_solve of 2^20 / W samples.  `low` / `high` apply the assumed collective latency's range."""
    I = SCALING_INPUTS
    t1 = float(solve_ms_1gpu if solve_ms_1gpu is not None else I["solve_ms_1gpu"]["value"])
    ex1 = dict(I["exchange_1rank_us"])
    if exchange_1rank_us:
        ex1.update({k: v for k, v in exchange_1rank_us.items() if v is not None})
    n, T = 1 << 20, 50
    lo_f, hi_f = I["collective_extra_us"]["range_factor"]
    scale = t1 / I["solve_ms_1gpu"]["value"]  # (a faster / slower box moves the strong-scaling solves with it)
    out = {"workload": "racing T=50 lambda=1, 2^20 samples per GPU (weak) / in total (strong)", "transports": {},
           "inputs": {"solve_ms_1gpu": t1, "exchange_1rank_us": {k: ex1[k] for k in ("rccl", "nccl", "p2p")},
                      "assumed": "collective_extra_us (x%.1f .. x%.1f), p2p's one-rank cost" % (lo_f, hi_f)}}
    for tr in ("rccl", "nccl", "p2p"):
        rows = {}
        for W in (1, 2, 4, 8):
            if W == 1:
                rows["1"] = {"ms_per_step": t1, "value": n * T / (t1 * 1e-3), "efficiency": 1.0,
                             "strong_ms_per_step": t1, "strong_value": n * T / (t1 * 1e-3), "strong_efficiency": 1.0}
                continue
            extra = I["collective_extra_us"][tr][W]
            fixed = ex1[tr] + I["straggler_us"][W]
            r = {}
            for tag, f in (("", 1.0), ("_low", hi_f), ("_high", lo_f)):  # (low value = high latency)
                t = t1 + (fixed + extra * f) * 1e-3
                ts = I["solve_ms_by_local_samples"][n // W] * scale + (fixed + extra * f) * 1e-3
                r["ms_per_step" + tag] = t
                r["value" + tag] = W * n * T / (t * 1e-3)
                r["efficiency" + tag] = t1 / t
                r["strong_ms_per_step" + tag] = ts
                r["strong_value" + tag] = n * T / (ts * 1e-3)
                r["strong_efficiency" + tag] = t1 / (W * ts)
            rows[str(W)] = r
        out["transports"][tr] = rows
    return out


def best_of(runs):
    ok = [r for r in runs if "error" not in r and r["finite"]]
    return min(ok, key=lambda r: r["dt"]) if ok else None


def run_transports(order, timed_run, dog, first_budget_s, alt_budget_s, runs):
    """Time every transport in `order` (appending to `runs`), each under its own watchdog phase: the first one — nothing has
    completed yet — with `first_budget_s`, the later ones with `alt_budget_s`."""
    for i, mode in enumerate(order):
        dog.arm(first_budget_s if i == 0 else alt_budget_s, f"transport {mode}")
        runs.append(timed_run(mode))
    dog.cancel()
    return runs


def run_with_deadline(cmd, env, timeout_s, on_timeout_line):
    """Run the launcher with a wall-clock limit; the ranks' stdout (rank 0's JSON line) passes through.  On expiry the whole
    process group is killed and, if no line came out, `on_timeout_line()` is printed instead — the driver's own limit would
    otherwise end the run with nothing on stdout."""
    import signal
    import threading

    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, start_new_session=True)
    seen = {"line": False}

    def pump():
        for raw in proc.stdout:
            text = raw.decode(errors="replace").rstrip("\n")
            if text.startswith("{"):
                seen["line"] = True
            print_line(text)

    t = threading.Thread(target=pump, daemon=True)
    t.start()
    try:
        rc = proc.wait(timeout=timeout_s)
        t.join(timeout=10)
        return rc
    except subprocess.TimeoutExpired:
        try:
            os.killpg(proc.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
        proc.wait()
        t.join(timeout=10)
        if not seen["line"]:
            print_line(json.dumps(on_timeout_line()))
        return 124


def run_child_transport(mode, index, budget_s, script=None):
    """One transport in its OWN process (this rank's child; the children of all ranks rendezvous on a port of their own): the
    same script with `--exchange <mode>`, i.e. a complete single-transport run that prints its own contract line on rank 0.
    Returns (exit code, stdout text, seconds); 124 = killed at the budget."""
    import signal

    env = dict(os.environ)
    env["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29500")) + 17 * (index + 1))
    env["MPPI_BENCH_CHILD"] = mode
    # a launcher's TORCHELASTIC_USE_AGENT_STORE makes env:// rendezvous a CLIENT of the launcher's store on the original port:
    # the children host their own store on the shifted port (rank 0), so the launcher's variables must not reach them
    for k in [k for k in env if k.startswith("TORCHELASTIC_")]:
        env.pop(k)
    cmd = [sys.executable, script or os.path.abspath(__file__), *sys.argv[1:], "--exchange", mode]
    t0 = time.perf_counter()
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, start_new_session=True)
    try:
        out, _ = proc.communicate(timeout=budget_s)
        rc = proc.returncode
    except subprocess.TimeoutExpired:
        try:
            os.killpg(proc.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
        out, _ = proc.communicate()
        rc = 124
    return rc, (out or b"").decode(errors="replace"), time.perf_counter() - t0


def orchestrate_transports(args, world, rank, order, run_child=run_child_transport):
    """`--exchange all` at N > 1 (round 5): every transport runs in a process of its own, one after the other, on every rank.
    A transport that hangs — in its communicator's set-up, its self-test or a collective — is killed at its budget and costs
    only itself: the parents (this function, one per rank; they never touch the GPU or a process group) go on to the next
    transport.  Rank 0 merges the children's lines: the best complete run is the line, all of them are listed under
    `transports`; no complete run -> a null line with every transport's error.  Returns the exit code."""
    entries, lines = [], {}
    for i, mode in enumerate(order):
        rc, text, secs = run_child(mode, i, args.first_budget_s)
        found = [ln for ln in text.splitlines() if ln.startswith("{")]
        line = None
        if found:
            try:
                line = json.loads(found[-1])
            except ValueError:
                line = None
        e = {"exchange": mode, "requested": mode, "isolated_process": True, "exit_code": rc, "seconds": round(secs, 1)}
        if line is not None and line.get("value"):
            lines[mode] = line
            e.update({"exchange": line.get("config", {}).get("exchange_used", mode), "value": line["value"],
                      "ms_per_step": line["ms_per_step"], "exchange_us": line.get("exchange_us"),
                      "rccl_ranks": line.get("rccl_ranks"), "strong": line.get("strong"),
                      "per_rank_stages_ms": (line.get("transports") or [{}])[0].get("per_rank_stages_ms")})
        else:
            why = (line or {}).get("error") if line else None
            inner = [t.get("error") for t in (line or {}).get("transports") or [] if t.get("error")]
            if inner:  # (the child's own transport error says more than its "no transport completed")
                why = inner[0]
            e["error"] = why or ("killed at its %.0f s budget" % args.first_budget_s if rc == 124 else
                                 "exit code %d, no line" % rc if rank == 0 else "exit code %d" % rc)
            if line is not None and line.get("transports"):
                e["detail"] = line["transports"]
        entries.append(e)
    if rank != 0:
        return 0
    if not lines:
        print_line(json.dumps(null_line(args, world, "no transport completed a run (each ran in its own process)", entries)))
        return 1
    best = max(lines, key=lambda m: lines[m]["value"])
    out = dict(lines[best])
    out["transports"] = entries
    out.setdefault("config", {})["exchange_requested"] = "all (one process per transport)"
    print_line(json.dumps(out))
    return 0


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` without a launcher: re-run this script under torch.distributed.run, one rank per
    GPU of this node, rendezvous on 127.0.0.1; the ranks' output (rank 0's JSON line) passes through.  The launcher gets a
    wall-clock limit (--launch-timeout-s) below the driver's."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    return run_with_deadline(cmd, env, args.launch_timeout_s,
                             lambda: null_line(args, args.gpus, f"the {args.gpus} ranks did not finish within "
                                               f"--launch-timeout-s = {args.launch_timeout_s:.0f} s (killed)"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--samples", type=int, default=1 << 20, help="samples per GPU")
    ap.add_argument("--horizon", type=int, default=50)
    ap.add_argument("--exchange", choices=("all", "nccl", "rccl", "p2p", "auto"), default="all",
                    help="per-solve exchange of the shard summaries at N > 1: all (default) = time EVERY transport with the "
                         "full K steps — the library's own RCCL communicator first (rccl: ncclAllGather on the solve's stream, "
                         "the cheapest at one rank: +3 us), then one all_gather through torch.distributed (nccl), then the "
                         "peer-to-peer buffers (p2p) — and report the best complete run as `value` (all of them under "
                         "`transports`); nccl / rccl / p2p / auto (rccl when its self-test passes on every rank, else nccl) "
                         "time only that one")
    ap.add_argument("--alt-budget-s", type=float, default=150.0,
                    help="wall-clock guard of every transport after the first (a watchdog prints the best run so far — or a "
                         "line with value null and every transport's error — and ends every rank if one hangs)")
    ap.add_argument("--first-budget-s", type=float, default=300.0,
                    help="wall-clock guard of the process-group set-up and of the FIRST transport (armed before either starts)")
    ap.add_argument("--launch-timeout-s", type=float, default=1500.0,
                    help="wall-clock limit of the ranks `python bench.py --gpus N` launches itself (below the driver's)")
    ap.add_argument("--isolate", type=int, default=1,
                    help="N > 1 with --exchange all: 1 (default) = every transport in a process of its own (a hang costs that "
                         "transport only), 0 = all transports in this process under the watchdog")
    ap.add_argument("--preflight", action="store_true",
                    help="N > 1: process-group set-up, every transport's own self-test and ONE sharded solve per transport on a "
                         "small problem, under the watchdog; prints one JSON line of statuses (< 60 s) instead of the bench line")
    ap.add_argument("--math", type=int, default=2, help="2 = fast-path math with hardware sin/cos of the wrapped heading "
                    "(default), 1 = fast-path math with polynomial sin/cos, 0 = library math")
    ap.add_argument("--noise-regen", type=int, default=1,
                    help="1 = regenerate the Philox noise in registers (default), 0 = materialise the noise tiles")
    ap.add_argument("--mapping", type=int, default=0, help="0 = lane per trajectory (default), 1 = the literal "
                    "wavefront-per-trajectory rollout (comparison only)")
    ap.add_argument("--lazy-state-seq", type=int, default=0, help="0 (default) = the solver's own default: `MPPI(...)` with no "
                    "extension keyword, state_seq rolled out inside the solve's last kernel; 1 = the opt-in lazily completed "
                    "state sequence (it rides in an extra block of the next solve's rollout launch; every one is still "
                    "completed inside the timed region).  The line reports the other setting as a labelled extra "
                    "(`lazy_state_seq_opt_in`)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip both CPU baselines")
    ap.add_argument("--no-extras", action="store_true", help="skip closed_loop and other_configs")
    ap.add_argument("--workload", choices=("c3", "c2", "c2_lbps", "c5", "c3_dense"), default="c3",
                    help="c3 (default) = the metric's workload; c2 / c5 = time another BASELINE config's solve loop; c3_dense = "
                         "the metric's workload with a dense softmax (lambda = 5000) (profiling aids: print a short line, not "
                         "the contract's)")
    ap.add_argument("--timing", type=int, default=2, help="HIP-event instrumentation inside the timed region: "
                    "1 = every stage, 2 = dominant kernel only, 0 = none (stage times from a second pass)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    own_stdout()
    if (args.gpus > 1 and int(os.environ.get("WORLD_SIZE", "1")) == args.gpus and args.exchange == "all" and args.isolate
            and not args.preflight and args.workload == "c3"):
        sys.exit(orchestrate_transports(args, args.gpus, int(os.environ.get("RANK", "0")), ["rccl", "nccl", "p2p"]))

    import numpy as np
    import torch
    import torch.distributed as dist

    import mppi_playground_amd  # noqa: F401
    from envs.racing_controller import racing_controller
    from envs.racing_env import RacingEnv

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    one_device = bool(os.environ.get("MPPI_BENCH_ONE_DEVICE"))
    backend = "none"
    printed = {"done": False}

    def emit(out):
        if not printed["done"]:
            printed["done"] = True
            print_line(json.dumps(out))

    # what the watchdog can report when a phase of the multi-GPU run outlives its budget: the runs completed so far and,
    # once it exists, the function that turns the best of them into the contract's line
    mg = {"runs": [], "compose": None}
    dog = None
    if world > 1 or args.gpus > 1:
        if world != args.gpus:
            sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                     f"(python bench.py --gpus {args.gpus} does that itself)")
        ndev = torch.cuda.device_count()
        if ndev < world and not one_device:
            sys.exit(f"bench.py: {world} ranks but {ndev} visible GPU(s).  For a dry run of the multi-rank path on one "
                     "GPU set MPPI_BENCH_ONE_DEVICE=1 MPPI_BENCH_BACKEND=gloo (RCCL needs one device per rank)")

        def on_expiry(phase, seconds):
            why = f"{phase} exceeded its {seconds:.0f} s budget"
            runs = mg["runs"] + [{"exchange": phase, "error": why}]
            best = best_of(mg["runs"])
            if rank == 0:
                if best is not None and mg["compose"] is not None:
                    emit(mg["compose"](best, runs))
                else:
                    emit(null_line(args, world, why + "; no transport had completed", runs))
            return 0 if best is not None else 1

        # armed BEFORE the process group exists: a hang in its set-up or in the first transport's first collective must
        # still leave a line on stdout
        dog = Watchdog(on_expiry)
        dog.arm(args.first_budget_s, "process-group set-up")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.exchange != "all":
            os.environ["MPPI_EXCHANGE"] = args.exchange
        # MPPI_BENCH_BACKEND=gloo + MPPI_BENCH_ONE_DEVICE=1: dry run of the multi-rank path on a 1-GPU box
        backend = os.environ.get("MPPI_BENCH_BACKEND", "nccl")
        dev = 0 if one_device else local_rank
        torch.cuda.set_device(dev)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend)
        dog.cancel()
    else:
        torch.cuda.set_device(0)

    if args.workload != "c3":
        return other_workload(args, torch, np)

    N_local, T = args.samples, args.horizon
    N_total = N_local * world
    env = RacingEnv()
    state = env.reset()
    x0 = state.clone()
    ref_holder = {}

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_run(mode, n_total=None, force_exchange=False, lazy=None):
        """One complete measurement: a solver on transport `mode` (None at one GPU), SETUP_SOLVES un-timed solves to leave
        the idle power state, the contract's W warm-up steps, then EXACTLY K timed steps between barrier + synchronise on
        both sides, MAX over ranks.  `n_total`: the global sample count (default: weak scaling, N_local per rank).
        `force_exchange`: one rank, but through the sharded code path (summary -> exchange -> combine).
        Returns {exchange, dt, stages, exchange_ms, ctrl, ...} or {exchange, error}."""
        from mppi_playground_amd import _capi

        n_total = N_total if n_total is None else n_total
        if mode is not None:
            os.environ["MPPI_EXCHANGE"] = mode
        try:
            lz = bool(args.lazy_state_seq) if lazy is None else lazy
            dkw = {"lazy_state_seq": True} if lz else {}  # (nothing passed = the product's defaults)
            if force_exchange:
                dkw["_force_exchange"] = True
            ctrl = racing_controller(env, horizon=T, num_samples=n_total, lambda_=1.0,
                                     shard_samples=world > 1 or force_exchange, **dkw)
        except (_capi.MppiError, RuntimeError) as e:  # the transport's set-up / self-test failed on every rank alike
            return {"exchange": mode, "error": str(e)[:300]}
        ctrl.set_cost_map(env._obstacle_map, env._lane_map)
        solver = ctrl.solver
        solver.set_option("math", args.math)
        solver.set_option("noise_regen", args.noise_regen)
        solver.set_option("mapping", args.mapping)
        if "ref" not in ref_holder:
            ref_holder["ref"], _ = ctrl.calc_ref_trajectory(state, env.racing_center_path, 0, T, DL=0.1, lookahead_distance=3,
                                                            reference_path_interval=0.85)
        ctrl.set_reference(ref_holder["ref"])
        # set-up, before the contract's W warm-up steps: bring the device out of its idle power state (the first
        # ~20 ms of load run at lower clocks) so that short --warmup values do not time the clock ramp.  Reported as
        # `setup_solves` in the JSON line; never inside the timed region.
        # (the HIP events of --timing are created on demand, ~3 us each: the set-up solves run instrumented so that the pool exists
        # before the timed region — 40 creations inside a 20-step region were 6 us per step of round 5's headline)
        solver.set_option("timing", args.timing)
        for _ in range(max(SETUP_SOLVES, args.steps)):
            solver.forward(x0)
        sync()
        solver.stage_times_ms()  # drain (the events stay in the pool)
        for _ in range(args.warmup):
            solver.forward(x0)
        sync()
        solver.stage_times_ms()  # drain
        t0 = time.perf_counter()
        for _ in range(args.steps):
            a, s = solver.forward(x0)
        solver.join_state_seq()  # (lazy state sequences: all K of them are completed INSIDE the timed region)
        sync()
        dt = time.perf_counter() - t0
        stages = solver.stage_times_ms()
        t_exchange_ms = None
        if args.timing != 1:  # complete the per-stage picture with a separate instrumented pass
            solver.set_option("timing", 1)
            n2 = min(args.steps, 50)
            t1 = time.perf_counter()
            for _ in range(n2):
                solver.forward(x0)
            sync()
            wall2 = (time.perf_counter() - t1) / n2 * 1e3
            extra = solver.stage_times_ms()
            if args.timing == 2:
                extra["rollout_cost"] = stages["rollout_cost"]
            stages = extra
            if world > 1:  # what is left of a solve's wall time after the device stages: the exchange + its stream hand-offs
                t_exchange_ms = max(wall2 - sum(stages[k] for k in ("sample", "rollout_cost", "weights_reduce", "finalize")), 0.0)
        solver.set_option("timing", 0)
        if world > 1:
            t = torch.tensor([dt], device="cuda", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        finite = bool(torch.isfinite(a).all() and torch.isfinite(s).all())
        sharded = world > 1 or force_exchange
        used = "none" if not sharded else ("p2p" if solver._p2p else "rccl" if solver._comm else "nccl")
        # every rank's own stage times (a straggler shows here, not in the MAX-over-ranks wall clock) and what RCCL itself
        # reports for the communicator that carried the exchange
        rank_stages, rccl_ranks = [stages], None
        if world > 1:
            rank_stages = [None] * world
            dist.all_gather_object(rank_stages, {k: round(v, 6) for k, v in stages.items()})
        if sharded:
            if solver._comm:
                import ctypes as C

                cnt, rk = C.c_int(-1), C.c_int(-1)
                try:
                    solver._h.call("mppi_comm_info", C.byref(cnt), C.byref(rk))
                    rccl_ranks = {"ncclCommCount": cnt.value, "ncclCommUserRank_of_rank0": rk.value, "communicator": "library"}
                except _capi.MppiError as e:
                    rccl_ranks = {"error": str(e)[:200]}
            else:
                rccl_ranks = {"world_size": dist.get_world_size(), "backend": dist.get_backend(),
                              "communicator": "torch.distributed process group"}
        return {"exchange": used, "requested": mode, "dt": dt, "stages": stages, "exchange_ms": t_exchange_ms, "finite": finite,
                "ctrl": ctrl, "n_total": n_total, "rank_stages": rank_stages, "rccl_ranks": rccl_ranks,
                "solver_kwargs": {k: v for k, v in dkw.items() if not k.startswith("_")}}

    def compose(best, runs):
        """The contract's JSON line from the best complete run (rank 0)."""
        dt, stages, solver = best["dt"], best["stages"], best["ctrl"].solver
        ms_per_step = dt / args.steps * 1e3
        solves_per_s = args.steps / dt
        value = N_total * T * solves_per_s
        dc = 2
        # algorithmic bytes: per sample-step 4*dc B noise written by the sampler, read by the
        # rollout, read again by the weighted reduction, + 8 B/sample of costs -> per solve and per GPU:
        b_alg_solve = 3 * 4 * dc * N_local * T + 8 * N_local
        # dominant kernel = rollout_cost_kernel: reads the noise once, writes costs once
        b_alg_rollout = 4 * dc * N_local * T + 4 * N_local
        t_roll = stages["rollout_cost"] * 1e-3
        achieved = b_alg_rollout / t_roll / 1e9
        dev_solve_ms = sum(stages[k] for k in ("sample", "rollout_cost", "weights_reduce", "finalize"))
        # PMC counters of the dominant kernel for THIS configuration, per launch: collected with rocprofv3 --pmc in
        # separate passes over this same command and committed (profiles/pmc_constants.json names the CSV summaries
        # they were generated from by scripts/pmc_constants.py); a bench run cannot collect counters on itself.
        traffic, valu, traffic_src, traffic_stale, solve_pmc = None, None, None, None, None
        try:
            if (N_local, T, args.math) == (1 << 20, 50, 2):
                from mppi_playground_amd import _build

                pc = json.load(open(os.path.join(ROOT, "profiles", "pmc_constants.json")))
                # counters are per BUILD of the kernels: the file carries the sha256 of the sources it was measured on
                traffic_stale = pc.get("csrc_sha256") != _build.source_digest()
                k = pc["rollout_regen" if args.noise_regen else "rollout_tiles"]
                traffic = int((2 * k["fetch_kb"] + k["write_kb"]) * 1024)  # FETCH_SIZE doubled: gfx950 wide-read correction
                traffic_src = pc.get("source")
                peak = 1024 * 2.4e9 / 2  # wave64 VALU instructions/s: 1024 SIMD32s, 2 cycles each, 2.4 GHz
                valu = {"kernel": "rollout_cost_kernel<racing>", "valu_insts_per_launch": k["valu_insts"],
                        "achieved_Ginst_per_s": k["valu_insts"] / t_roll / 1e9, "peak_Ginst_per_s": peak / 1e9,
                        "frac": k["valu_insts"] / t_roll / peak,
                        "measured_peak_Ginst_per_s": pc.get("valu_issue_ubench", {}).get("mul_add_Ginst_per_s"),
                        "cycles_per_inst_per_simd_at_2p4GHz": 1024 * 2.4e9 * t_roll / k["valu_insts"],
                        "counters_stale": traffic_stale,
                        "note": "wave64 VALU instructions (SQ_INSTS_VALU, rocprofv3) / live kernel time"}
                ks = pc.get("solve_regen" if args.noise_regen else "solve_tiles")
                if ks and ks.get("fetch_kb") is not None:
                    solve_pmc = {"bytes_moved": int((2 * ks["fetch_kb"] + ks["write_kb"]) * 1024),
                                 "valu_insts": ks.get("valu_insts"),
                                 "valu_frac": None if not ks.get("valu_insts") else ks["valu_insts"] / (dev_solve_ms * 1e-3) / peak}
        except Exception:
            pass
        used = best["exchange"]
        out = {
            "metric": "sample_steps_per_sec", "value": value, "unit": "sample-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "setup_solves": SETUP_SOLVES,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "racing kinematic-bicycle MPPI solve (BASELINE configs[2]; configs[3] when n_gpus = 8)",
                       "num_samples_per_gpu": N_local, "num_samples_total": N_total, "horizon": T,
                       "lambda": 1.0, "noise": "device philox4x32-10 (" + ("regenerated in registers" if args.noise_regen else "materialised tiles") + ")", "math": {0: "library", 1: "fast (polynomial sin/cos)", 2: "fast (hardware sin/cos of the wrapped heading)"}[args.math],
                       "mapping": "lane-per-trajectory" if not args.mapping else "wavefront-per-trajectory",
                       "sharding": f"num_samples x{world}" if world > 1 else "none",
                       "exchange": {"p2p": "peer-to-peer buffers (xGMI stores, polled)",
                                    "rccl": "ncclAllGather of 4+T*dc floats issued by the library on the solve's stream",
                                    "nccl": "torch.distributed all_gather of 4+T*dc floats", "none": "none"}[used],
                       "exchange_used": used, "exchange_requested": args.exchange if world > 1 else None,
                       "backend": backend, "ranks_share_one_device": one_device and world > 1},
            "solves_per_sec": solves_per_s,
            # the kernel is VALU-issue bound (valu_roofline), not HBM bound: `achieved`/`frac` price its ALGORITHMIC
            # bytes (the noise it consumes is regenerated in registers, so `traffic` is ~1 % of them) against the
            # HBM peak, as the metric asks; they cannot exceed ~0.45 at this instruction count
            "roofline": {"bound": "valu", "kernel": "rollout_cost_kernel<racing>", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src, "traffic_stale": traffic_stale,
                         "algorithmic_bytes_per_launch": b_alg_rollout, "kernel_ms": stages["rollout_cost"],
                         # what the kernel really moves (PMC bytes / its time) and how much of the VALU issue peak it uses:
                         # read `frac` next to these two
                         "hbm_measured_GBps": None if traffic is None else traffic / t_roll / 1e9,
                         "valu_frac": None if valu is None else valu["frac"],
                         "note": "`achieved` / `frac` price the kernel's ALGORITHMIC bytes (SURVEY 8d: the noise it consumes + the "
                                 "costs it writes) against the HBM peak as the metric asks; the kernel regenerates that noise in "
                                 "registers, moves ~1 %% of those bytes (`hbm_measured_GBps`) and is bound by VALU issue "
                                 "(`valu_frac`; 0.69 is what its instruction mix allows).  Whole solve: SURVEY 8d's B_alg / "
                                 "ms_per_step = %.2f x the HBM peak — not a bandwidth: two of B_alg's three noise-sized terms never "
                                 "exist (no sampler pass, no second read) and at lambda = 1 the weighted reduction is an arg-min "
                                 "over one or two of 16 384 tiles.  `other_configs.c3_dense` is the same workload with a dense "
                                 "softmax (every tile regenerated a second time), with its own per-kernel roofline."
                                 % (b_alg_solve / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS)},
            # the whole solve (every kernel of it): what it MOVES through HBM (PMC, all mppi:: kernels of a solve) and how
            # much of the VALU issue peak it uses over its device time.  SURVEY 8d's B_alg (three noise-sized terms + the
            # costs) is kept for reference only: two of its terms never touch HBM in this design (the noise is
            # regenerated in registers) and at lambda = 1 the third is an arg-min over one or two tiles, so
            # B_alg / time is not a bandwidth (it exceeded the 8 TB/s peak in round 3); `other_configs.c3_dense`
            # times the solve whose weighted reduction does touch every tile.
            "solve_roofline": {"device_ms_per_solve": dev_solve_ms,
                               "bytes_moved": None if solve_pmc is None else solve_pmc["bytes_moved"],
                               "moved_GBps": None if solve_pmc is None else solve_pmc["bytes_moved"] / (dev_solve_ms * 1e-3) / 1e9,
                               "valu_insts": None if solve_pmc is None else solve_pmc["valu_insts"],
                               "valu_frac": None if solve_pmc is None else solve_pmc["valu_frac"],
                               "counters_stale": traffic_stale,
                               "algorithmic_bytes_per_solve_survey_8d": b_alg_solve,
                               "note": "bytes_moved / valu_insts: rocprofv3 PMC sums over every kernel of one solve "
                                       "(profiles/pmc_constants.json); the solve is VALU-issue bound, not HBM bound"},
            "stages_ms": {k: stages[k] for k in ("sample", "rollout_cost", "weights_reduce", "finalize")},
        }
        out["config"]["state_seq"] = ("opt-in lazy_state_seq=True: solve k's batch-1 rollout rides in an extra block of solve k+1's "
                                      "rollout launch; the last one is completed inside the timed region" if solver._lazy_state else
                                      "default: rolled out inside the solve's last kernel")
        out["config"]["solver_kwargs"] = best.get("solver_kwargs", {})  # extension keywords the timed solver was built with ({} = defaults)
        if best["exchange_ms"] is not None:
            out["stages_ms"]["exchange_and_handoffs"] = best["exchange_ms"]
            out["exchange_us"] = best["exchange_ms"] * 1e3  # per solve: wall time minus the device stages (instrumented pass)
        if world > 1:  # every transport that was timed with the full K steps; `value` is the best complete one
            out["transports"] = [
                {"exchange": r["exchange"], "requested": r.get("requested"), "error": r["error"]} if "error" in r else
                {"exchange": r["exchange"], "requested": r.get("requested"), "ms_per_step": r["dt"] / args.steps * 1e3,
                 "value": r["n_total"] * T * args.steps / r["dt"], "finite": r["finite"],
                 "exchange_us": None if r["exchange_ms"] is None else r["exchange_ms"] * 1e3,
                 "rccl_ranks": r["rccl_ranks"], "per_rank_stages_ms": r["rank_stages"]} for r in runs]
            out["rccl_ranks"] = best["rccl_ranks"]
        if valu is not None:
            out["valu_roofline"] = valu
        # the predicted 1 -> 8 curve next to whatever this run measured (live one-GPU solve time when this IS the one-GPU
        # run; at N > 1 the committed inputs, and this run's own number held against the prediction for its N)
        pred = predict_scaling(solve_ms_1gpu=ms_per_step if world == 1 else None)
        if world > 1:
            tr = {"rccl": "rccl", "nccl": "nccl", "p2p": "p2p"}.get(best["exchange"], "nccl")
            row = pred["transports"][tr].get(str(world))
            if row is not None:
                pred["vs_measured"] = {"n_gpus": world, "transport": tr, "predicted_ms_per_step": row["ms_per_step"],
                                       "predicted_range_ms": [row["ms_per_step_high"], row["ms_per_step_low"]],
                                       "measured_ms_per_step": ms_per_step, "measured_over_predicted": ms_per_step / row["ms_per_step"],
                                       "inside_predicted_range": bool(row["ms_per_step_high"] <= ms_per_step <= row["ms_per_step_low"]),
                                       # (a dry run — all ranks time-sharing ONE device — measures the time slicing, not the exchange)
                                       "ranks_share_one_device": bool(os.environ.get("MPPI_BENCH_ONE_DEVICE"))}
        out["predicted"] = pred
        out.update(extras)
        return out

    extras = {}  # entries added to the line once they exist (strong-scaling leg)
    mg["compose"] = compose

    if world == 1:
        runs = [timed_run(None)]
        assert "error" not in runs[0] and runs[0]["finite"]
        out = compose(runs[0], runs)
        ctrl = runs[0]["ctrl"]
        if args.steps < 200:
            # the same solver, the same instrumentation (--timing), over 200 steps, three times: a timed region has a fixed cost
            # (pipeline fill after the synchronise, the final wake-up), so the per-step time of a longer region must not
            # be ABOVE the contract's K-step one.  Round 5's driver line had 0.1450 against 0.1403; reproduced in round 6: a
            # region that needs more HIP events than the solver's pool holds creates them inside the region (hipEventCreate,
            # ~3 us each, two per instrumented stage per step) — the first 200-step repetition measured 0.1500 against 0.1436
            # for the second and third.  The pool is now filled before any timed region (timed_run) and here (one un-timed
            # pass); tests/test_gpu_bench_contract.py holds the best repetition to 1.01x the headline.
            sv = ctrl.solver
            reps = []
            sv.set_option("timing", args.timing)
            for _ in range(200):  # (un-timed: fills the event pool)
                sv.forward(x0)
            sync()
            sv.stage_times_ms()
            for _ in range(3):
                sv.set_option("timing", args.timing)
                sv.stage_times_ms()  # drain
                sync()
                t0 = time.perf_counter()
                for _ in range(200):
                    sv.forward(x0)
                sv.join_state_seq()
                sync()
                dt200 = time.perf_counter() - t0
                st200 = sv.stage_times_ms()
                sv.set_option("timing", 0)
                reps.append({"ms_per_step": dt200 / 200 * 1e3, "rollout_cost_ms": st200.get("rollout_cost"),
                             "state_seq_standalone_launches": st200.get("state_seq_standalone_launches")})
            b = min(reps, key=lambda r: r["ms_per_step"])
            out["long_run"] = {"steps": 200, "ms_per_step": b["ms_per_step"], "value": N_total * T / (b["ms_per_step"] * 1e-3),
                               "repetitions": reps, "same_solver_and_timing_as_headline": True,
                               "ratio_to_headline": b["ms_per_step"] / (runs[0]["dt"] / args.steps * 1e3),
                               "note": "best of three 200-step regions on the headline's solver (every repetition listed); the "
                                       "contract's line above times ONE region of --steps %d" % args.steps}
        if not args.no_extras:
            # the opt-in lazily completed state sequence (the headline is the solver's default: completed inside finalize_kernel)
            er = timed_run(None, lazy=not bool(args.lazy_state_seq))
            out["lazy_state_seq_opt_in" if not args.lazy_state_seq else "default_state_seq"] = {
                "ms_per_step": er["dt"] / args.steps * 1e3, "value": N_total * T * args.steps / er["dt"],
                "stages_ms": {k: er["stages"][k] for k in ("rollout_cost", "weights_reduce", "finalize")},
                "solver_kwargs": er["solver_kwargs"],
                "note": "the same run with the other state_seq setting"}
            er["ctrl"] = None
            out["sharded_one_rank"] = sharded_one_rank(torch, dist, timed_run, runs[0], args, N_total, T)
            out["closed_loop"] = closed_loop(torch, env, ctrl, T, N_total)
            out["example_loop"] = example_loop(torch)
            out["other_configs"] = other_configs(torch, np)
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(np, T, ref_holder["ref"].numpy(), x0.cpu().numpy())
            runs[0]["ctrl"] = None
            del ctrl
            torch.cuda.empty_cache()
            out["cpu_baseline_torch"] = cpu_baseline_torch(torch, np, T)
        emit(out)
        return

    # N > 1: time EVERY transport of the per-solve exchange with the full K steps and report the best complete run as
    # `value`, all of them under `transports`.  No transport had crossed a device boundary before the first multi-GPU run,
    # so EVERY phase runs under the watchdog (armed before the process group was set up): the cheapest transport first
    # (the in-library communicator: +3 us per solve at one rank), so that a good number exists before the riskier ones; a
    # phase that outlives its budget ends every rank after rank 0 printed the best run completed so far — or, if none, a
    # contract-shaped line with `value: null` and every transport's error.
    order = {"all": ["rccl", "nccl", "p2p"]}.get(args.exchange, [args.exchange])
    if args.preflight:
        return preflight(args, torch, dist, env, order, dog, emit, world, rank)
    runs = run_transports(order, timed_run, dog, args.first_budget_s, args.alt_budget_s, mg["runs"])
    best = best_of(runs)
    if best is None:
        if rank == 0:
            emit(null_line(args, world, "no transport completed a finite run", runs))
        dist.barrier()
        dist.destroy_process_group()
        sys.exit(1)
    # strong-scaling leg (BASELINE's metric also reads "racing N = 1M ... at 1/2/4/8 GPUs"): the SAME 2^20 samples split
    # over the ranks, on the transport that won the weak leg, behind the same watchdog.  Reported next to the weak-scaling
    # `value` (which stays the contract's line: fixed work per GPU), never instead of it.
    dog.arm(args.alt_budget_s, "strong-scaling leg")
    sr = timed_run(best["requested"], n_total=N_local)
    dog.cancel()
    if "error" in sr or not sr["finite"]:
        extras["strong"] = {"error": sr.get("error", "non-finite outputs")}
    else:
        extras["strong"] = {"scaling": "strong", "num_samples_total": N_local, "num_samples_per_gpu": N_local // world,
                            "exchange": sr["exchange"], "ms_per_step": sr["dt"] / args.steps * 1e3,
                            "solves_per_sec": args.steps / sr["dt"], "value": N_local * T * args.steps / sr["dt"],
                            "unit": "sample-steps/s", "exchange_us": None if sr["exchange_ms"] is None else sr["exchange_ms"] * 1e3,
                            "per_rank_stages_ms": sr["rank_stages"]}
    sr["ctrl"] = None
    if rank == 0:
        emit(compose(best, runs))
    dog.arm(60.0, "process-group tear-down")
    dist.barrier()
    dist.destroy_process_group()
    dog.cancel()


def preflight(args, torch, dist, env, order, dog, emit, world, rank):
    """--preflight: what a multi-GPU run needs before its first timed step, on a small problem and in under a minute — the
    process group (already up when this runs), every transport's own set-up and self-test (three pattern exchanges checked
    on every rank: `MPPI(shard_samples=True)`), and ONE sharded solve per transport whose outputs every rank checks to be
    finite and identical to rank 0's.  One JSON line of statuses; exit code 0 only if at least one transport passed."""
    from envs.racing_controller import racing_controller
    from mppi_playground_amd import _capi

    t_all = time.perf_counter()
    status = []
    N, T = 65536 * world, args.horizon
    state = env.reset()
    for mode in order:
        dog.arm(60.0, f"preflight {mode}")
        t0 = time.perf_counter()
        os.environ["MPPI_EXCHANGE"] = mode
        try:
            ctrl = racing_controller(env, horizon=T, num_samples=N, lambda_=1.0, shard_samples=True)
            ctrl.set_cost_map(env._obstacle_map, env._lane_map)
            ref, _ = ctrl.calc_ref_trajectory(state, env.racing_center_path, 0, T, DL=0.1, lookahead_distance=3,
                                              reference_path_interval=0.85)
            ctrl.set_reference(ref)
            a, s = ctrl.solver.forward(state)
            torch.cuda.synchronize()
            a0 = a.clone()
            dist.broadcast(a0, src=0)
            ok = bool(torch.isfinite(a).all() and torch.isfinite(s).all() and torch.equal(a0, a))
            flag = torch.tensor([1.0 if ok else 0.0], device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            used = "p2p" if ctrl.solver._p2p else "rccl" if ctrl.solver._comm else "nccl"
            status.append({"requested": mode, "exchange": used, "ok": bool(flag.item() == 1.0),
                           "seconds": round(time.perf_counter() - t0, 3)})
            del ctrl
        except (_capi.MppiError, RuntimeError) as e:
            status.append({"requested": mode, "ok": False, "error": str(e)[:300], "seconds": round(time.perf_counter() - t0, 3)})
    dog.cancel()
    ok_any = any(x["ok"] and x.get("exchange") == x["requested"] for x in status)
    if rank == 0:
        emit({"preflight": status, "n_gpus": world, "backend": dist.get_backend(), "num_samples_total": N, "horizon": T,
              "seconds": round(time.perf_counter() - t_all, 3), "ok": ok_any})
    dog.arm(60.0, "process-group tear-down")
    dist.barrier()
    dist.destroy_process_group()
    dog.cancel()
    sys.exit(0 if ok_any else 1)


def sharded_one_rank(torch, dist, timed_run, plain, args, N, T):
    """The N-GPU code path with ONE rank (shard_samples=True + the solver's private one-rank exchange hook: summary ->
    exchange over the real RCCL backend -> combine), timed like the headline: a multi-GPU run at N = 1 and this bench's
    single-GPU line must be the same number, so the first point of a scaling curve is not an artefact of the sharded
    path's fixed cost.  Reports both transports and whether each is within 3 % of the unsharded `value`."""
    import tempfile

    out = {}
    try:
        store = dist.FileStore(os.path.join(tempfile.mkdtemp(prefix="mppi_bench_"), "store"), 1)
        dist.init_process_group("nccl", store=store, rank=0, world_size=1, device_id=torch.device("cuda", torch.cuda.current_device()))
    except Exception as e:  # noqa: BLE001
        return {"error": f"one-rank nccl process group: {type(e).__name__}: {str(e)[:200]}"}
    try:
        for mode in ("rccl", "nccl"):
            r = timed_run(mode, n_total=N, force_exchange=True)
            if "error" in r:
                out[mode] = {"error": r["error"]}
                continue
            ratio = plain["dt"] / r["dt"]
            out[mode] = {"ms_per_step": r["dt"] / args.steps * 1e3, "value": N * T * args.steps / r["dt"],
                         "value_over_unsharded": ratio, "within_3pct": bool(abs(ratio - 1.0) <= 0.03), "finite": r["finite"],
                         "exchange": r["exchange"], "rccl_ranks": r["rccl_ranks"]}
            r["ctrl"] = None
    finally:
        os.environ.pop("MPPI_EXCHANGE", None)
        dist.destroy_process_group()
    return out


def closed_loop(torch, env, ctrl, T, N):
    """The reference's racing control loop minus rendering (example/racing.py:221-266), resident on the device: every tick
    rebuilds the reference window from the state in HBM (mppi_ref_window: nearest centre-line point, monotone path index
    kept on the device), solves with the warm start of the previous tick, and applies a[0] through env.step (one launch
    of the library's racing functor, mppi_model_step).  No host synchronisation per tick."""
    state = env.reset()
    ctrl.current_path_index = 0
    ctrl.solver.reset()
    ticks, warm = 100, 20
    t_upd = t_step = 0.0
    for tick in range(warm + ticks):
        if tick == warm:
            torch.cuda.synchronize()
            t_upd = t_step = 0.0
            t0 = time.perf_counter()
        ta = time.perf_counter()
        a, s = ctrl.update(state, env.racing_center_path)
        tb = time.perf_counter()
        state, _ = env.step(a[0, :])
        tc = time.perf_counter()
        t_upd += tb - ta
        t_step += tc - tb
    ctrl.solver.join_state_seq()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"ticks": ticks, "warmup_ticks": warm, "ms_per_tick": dt / ticks * 1e3, "solves_per_sec": ticks / dt,
            "sample_steps_per_sec": N * T * ticks / dt,
            "device_resident_tick": bool(ctrl._window_on_device and env._native_step),
            "host_enqueue_ms_per_tick": {"controller.update (reference-window kernel + solve)": t_upd / ticks * 1e3,
                                         "env.step (one native launch)": t_step / ticks * 1e3},
            "path_index_after": int(ctrl.current_path_index),
            "final_speed_mps": float(state[3]),
            "note": "open-loop `value` above times solves from a fixed state; here the state, the reference window and "
                    "the warm start change every tick"}


def example_loop(torch):
    """The loop of the reference's racing example at ITS sizes (example/racing.py:25-26,221-266: T = 25, N = 4000) minus
    rendering: controller.update (reference window + solve), env.step, env.collision_check(state_seq) and
    get_top_samples(300) — the last two are what the reference draws every tick.  No host synchronisation per tick."""
    from envs.racing_controller import racing_controller
    from envs.racing_env import RacingEnv

    env = RacingEnv()
    ctrl = racing_controller(env, horizon=25, num_samples=4000, lambda_=1.0)
    ctrl.set_cost_map(env._obstacle_map, env._lane_map)
    state = env.reset()
    ticks, warm = 200, 20
    runs = []
    for rep in range(3):  # (min of three 200-tick loops: one loop is 12 ms of wall clock, a single host hiccup is 10 % of it)
        for tick in range((warm if rep == 0 else 0) + ticks):
            if tick == (warm if rep == 0 else 0):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            a, s = ctrl.update(state, env.racing_center_path)
            state, _ = env.step(a[0, :])
            env.collision_check(state=s)
            ctrl.get_top_samples(num_samples=300)
        torch.cuda.synchronize()
        runs.append(time.perf_counter() - t0)
    dt = min(runs)
    return {"config": "racing T=25 N=4000 lambda=1 (the reference example's own size): update + env.step + collision_check + "
                      "get_top_samples(300) per tick", "ticks": ticks, "ms_per_tick": dt / ticks * 1e3, "ticks_per_sec": ticks / dt,
            "timing": "min of three 200-tick loops", "ms_per_tick_all": [r / ticks * 1e3 for r in runs]}


def _time_solver(torch, solver, x0, n=50, warm=10, repeats=3):
    """Seconds per solve: un-timed solves for at least `warm` solves AND 50 ms of wall clock (these secondary entries are
    2-12 ms of GPU time each: a box that has dropped to its idle power state during the host-heavy parts of the bench runs the
    first tens of milliseconds of tiny kernels at a fraction of its clock — 174 instead of 28 us per solve was measured once),
    then the best of `repeats` timed loops of `n` solves."""
    t_warm = time.perf_counter()
    done = 0
    while done < warm or time.perf_counter() - t_warm < 0.05:
        solver.forward(x0)
        done += 1
        if done % 10 == 0:
            torch.cuda.synchronize()
    best = float("inf")
    for _ in range(repeats):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            solver.forward(x0)
        solver.join_state_seq()  # (a lazily completed state sequence of the last solve belongs inside the timed region)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n)
    return best


def _other_solvers(torch, np, which=None):
    """The other BASELINE configs: (key, label, N*T, B_alg per solve, solver factory, x0)."""
    from envs import classic_control as cc
    from envs.navigation_2d import Navigation2DEnv
    from pi_mpc.mppi import MPPI

    nav = Navigation2DEnv()
    t = torch.tensor
    # (every solver below is built with the product's defaults unless its label names an option)
    rows = [
        ("c1", "C1 pendulum T=50 N=1000 ESSPS", 1000 * 50, 3 * 4 * 1 * 1000 * 50 + 8 * 1000,
         lambda: MPPI(50, 1000, 2, 1, cc.pendulum_dynamics, cc.pendulum_cost, t([-2.0]), t([2.0]), t([1.0]), "ESSPS"),
         t([np.pi, 0.0], device="cuda", dtype=torch.float32)),
        ("c2", "C2 nav2d T=50 N=65536 lambda=1", 65536 * 50, 3 * 4 * 2 * 65536 * 50 + 8 * 65536,
         lambda: MPPI(50, 65536, 3, 2, nav.dynamics, nav.cost_function, nav.u_min, nav.u_max, t([0.5, 0.5]), 1.0),
         nav.reset().clone()),
        ("c2_essps", "C2 nav2d T=50 N=65536 ESSPS", 65536 * 50, 3 * 4 * 2 * 65536 * 50 + 8 * 65536,
         lambda: MPPI(50, 65536, 3, 2, nav.dynamics, nav.cost_function, nav.u_min, nav.u_max, t([0.5, 0.5]), "ESSPS"),
         nav.reset().clone()),
        ("c2_lbps_brent", "C2 nav2d T=50 N=65536 LBPS (the default: scipy's bounded Brent as ONE kernel on the device, round 6 — "
         "the reference's own search, no host wait)", 65536 * 50, 3 * 4 * 2 * 65536 * 50 + 8 * 65536,
         lambda: MPPI(50, 65536, 3, 2, nav.dynamics, nav.cost_function, nav.u_min, nav.u_max, t([0.5, 0.5]), "LBPS"),
         nav.reset().clone()),
        ("c2_lbps_brent_host", "C2 nav2d T=50 N=65536 LBPS (lbps_search='brent_host': the same search as a host loop inside the "
         "library, one read-back of the device statistics per probe — round 5's default; same temperature to the bit)",
         65536 * 50, 3 * 4 * 2 * 65536 * 50 + 8 * 65536,
         lambda: MPPI(50, 65536, 3, 2, nav.dynamics, nav.cost_function, nav.u_min, nav.u_max, t([0.5, 0.5]), "LBPS",
                      lbps_search="brent_host"),
         nav.reset().clone()),
        ("c2_lbps_grid", "C2 nav2d T=50 N=65536 LBPS (lbps_search='grid': two 32-temperature grids + a quartic as kernels; not the "
         "reference's search)", 65536 * 50, 3 * 4 * 2 * 65536 * 50 + 8 * 65536,
         lambda: MPPI(50, 65536, 3, 2, nav.dynamics, nav.cost_function, nav.u_min, nav.u_max, t([0.5, 0.5]), "LBPS",
                      lbps_search="grid"),
         nav.reset().clone()),
        ("c2_mpo", "C2 nav2d T=50 N=65536 MPO (dual on the device)", 65536 * 50, 3 * 4 * 2 * 65536 * 50 + 8 * 65536,
         lambda: MPPI(50, 65536, 3, 2, nav.dynamics, nav.cost_function, nav.u_min, nav.u_max, t([0.5, 0.5]), "MPO"),
         nav.reset().clone()),
        ("c5", "C5 cartpole T=64 N=262144 ESSPS + Savitzky-Golay", 262144 * 64, 3 * 4 * 1 * 262144 * 64 + 8 * 262144,
         lambda: MPPI(64, 262144, 4, 1, cc.cartpole_dynamics, cc.cartpole_cost, t([-3.0]), t([3.0]), t([1.0]), "ESSPS",
                      use_sg_filter=True),
         t([0.01, 0.0, 0.02, 0.0], device="cuda")),
    ]
    return [r for r in rows if which is None or r[0] in which]


def _stage_times(torch, solver, x0, n=30):
    """Per-stage device times (HIP events around each stage's launches) from a separate instrumented pass."""
    solver.set_option("timing", 1)
    solver.stage_times_ms()  # drain
    for _ in range(n):
        solver.forward(x0)
    torch.cuda.synchronize()
    st = solver.stage_times_ms()
    solver.set_option("timing", 0)
    return {k: st[k] for k in ("sample", "rollout_cost", "weights_reduce", "finalize")}


def other_configs(torch, np):
    """Solve times of the other BASELINE configs (open loop, 10 warm-up + 50 timed solves each), and the metric's own
    workload with a DENSE softmax: at lambda = 1 (configs[2], the headline) the racing softmax is an arg-min and the
    weighted reduction touches one or two of the 16 384 tiles; `c3_dense` (lambda = 5000: ESS of a few thousand) and
    `c3_essps` (ESS = N / 10) are what a non-degenerate 1 M-sample solve costs — every tile's noise is regenerated a
    second time and accumulated."""
    out = {}
    for key, label, work, b_alg, make, x0 in _other_solvers(torch, np):
        s = make()
        dt = _time_solver(torch, s, x0)
        out[key] = {"config": label, "ms_per_solve": dt * 1e3, "solves_per_sec": 1 / dt,
                    "sample_steps_per_sec": work / dt, "algorithmic_bytes_per_solve": b_alg,
                    "frac_of_8TBps": b_alg / dt / 1e9 / HBM_PEAK_GBS, "lambda": s._last_lambda,
                    "timing": "best of three 50-solve loops (min-of-3)"}
        del s
    n, T = 1 << 20, 50
    for key, lam, label in (("c3_dense", 5000.0, "C3 racing T=50 N=1048576 lambda=5000 (dense softmax)"),
                            ("c3_essps", "ESSPS", "C3 racing T=50 N=1048576 ESSPS (target ESS = N/10; device-resident search)")):
        ctrl, x0 = _racing_c3(torch, lam, **({"lambda_max": 1.0e5} if lam == "ESSPS" else {}))
        s = ctrl.solver
        dt = _time_solver(torch, s, x0, n=50, warm=20)
        st = s.last_stats()
        stages = _stage_times(torch, s, x0)
        out[key] = {"config": label, "ms_per_solve": dt * 1e3, "solves_per_sec": 1 / dt, "sample_steps_per_sec": n * T / dt,
                    "stages_ms": stages, "other_launches_ms": max(dt * 1e3 - sum(stages.values()), 0.0),
                    "lambda": s._last_lambda, "ess": st["ess"], "timing": "best of three 50-solve loops (min-of-3)"}
        if key == "c3_dense":
            out[key]["roofline"] = _dense_roofline(stages)
        del s, ctrl
        torch.cuda.empty_cache()
    return out


def _dense_roofline(stages):
    """Per-kernel roofline of the dense C3 solve: the two kernels that each consume all N*T*dc noise values (algorithmic
    423.6 MB per launch) against the HBM peak, and — from profiles/pmc_constants.json (`dense_*`: rocprofv3 PMC of
    `bench.py --workload c3_dense`, keyed on the source hash) — their share of the VALU issue peak and what they really move.
    Stage times are live HIP events; the weights_reduce STAGE is the reduction kernel plus the fold of its partial rows
    (summarize_kernel, ~6 us)."""
    n, T, dc = 1 << 20, 50, 2
    b_alg = 4 * dc * n * T + 4 * n
    peak = 1024 * 2.4e9 / 2
    pc, stale = {}, None
    try:
        from mppi_playground_amd import _build

        pc = json.load(open(os.path.join(ROOT, "profiles", "pmc_constants.json")))
        stale = pc.get("csrc_sha256") != _build.source_digest()
    except Exception:
        pass
    out = {"bound": "valu", "unit": "GB/s", "peak": HBM_PEAK_GBS, "counters_stale": stale}
    for stage, kname, key in (("rollout_cost", "rollout_cost_kernel<racing>", "dense_rollout"),
                              ("weights_reduce", "weights_reduce_kernel (+ summarize_kernel)", "dense_reduce")):
        t = stages[stage] * 1e-3
        k = pc.get(key) or {}
        moved = None if k.get("fetch_kb") is None else int((2 * k["fetch_kb"] + k["write_kb"]) * 1024)
        out[stage] = {"kernel": kname, "kernel_ms": stages[stage], "algorithmic_bytes_per_launch": b_alg,
                      "achieved": b_alg / t / 1e9, "frac": b_alg / t / 1e9 / HBM_PEAK_GBS,
                      "valu_insts_per_launch": k.get("valu_insts"),
                      "valu_frac": None if not k.get("valu_insts") else k["valu_insts"] / t / peak,
                      "traffic": moved, "hbm_measured_GBps": None if moved is None else moved / t / 1e9}
    return out


def _racing_c3(torch, lam, **kw):
    """(controller, x0) of the metric's workload (racing N = 2^20, T = 50) at temperature `lam`."""
    from envs.racing_controller import racing_controller
    from envs.racing_env import RacingEnv

    env = RacingEnv()
    x0 = env.reset().clone()
    ctrl = racing_controller(env, horizon=50, num_samples=1 << 20, lambda_=lam, **kw)
    ctrl.set_cost_map(env._obstacle_map, env._lane_map)
    ref, _ = ctrl.calc_ref_trajectory(x0, env.racing_center_path, 0, 50, DL=0.1, lookahead_distance=3, reference_path_interval=0.85)
    ctrl.set_reference(ref)
    return ctrl, x0


def other_workload(args, torch, np):
    """--workload c2|c5|c3_dense: the solve loop of another configuration, for rocprofv3 runs."""
    if args.workload == "c3_dense":
        ctrl, x0 = _racing_c3(torch, 5000.0)
        dt = _time_solver(torch, ctrl.solver, x0, n=args.steps, warm=args.warmup)
        print_line(json.dumps({"workload": "C3 racing T=50 N=1048576 lambda=5000 (dense softmax)", "ms_per_solve": dt * 1e3,
                          "steps": args.steps, "lambda": ctrl.solver._last_lambda, "ess": ctrl.solver.last_stats()["ess"]}))
        return
    which = {"c2": ("c2_essps",), "c2_lbps": ("c2_lbps_brent",), "c5": ("c5",)}[args.workload]
    for key, label, work, b_alg, make, x0 in _other_solvers(torch, np, which):
        s = make()
        dt = _time_solver(torch, s, x0, n=args.steps, warm=args.warmup)
        print_line(json.dumps({"workload": label, "ms_per_solve": dt * 1e3, "steps": args.steps,
                          "algorithmic_bytes_per_solve": b_alg, "lambda": s._last_lambda}))


def cpu_baseline(np, T, ref, x0):
    """The oracle (oracle/mppi_oracle.c: C restatement of the reference algorithm, OpenMP over the
    samples) on this host: full racing solves (N=1,048,576, T=50: clamp, rollout, costs, softmax,
    weighted mean) repeated for ~10 s; reported in the metric's unit.  Noise generation is excluded
    (the oracle's Philox restatement is single-threaded test code)."""
    from helpers import oracle_problem, orc

    n = 1 << 20
    P = oracle_problem("racing", n, T, ref_path=ref)
    eps = orc.philox_normal(42, 1, 0, n, T, 2, [0.5, 0.1])
    mean = np.zeros((T, 2), np.float32)
    P.rollout_cost(x0, mean, eps)  # page in
    t0 = time.perf_counter()
    reps = 0
    while reps < 2 or time.perf_counter() - t0 < 10.0:
        r = P.rollout_cost(x0, mean, eps)
        w, _ = orc.softmax_weights(r["costs"], 1.0)
        P.weighted_actions(w, mean, eps)
        reps += 1
    dt = time.perf_counter() - t0
    return {"value": reps * n * T / dt, "unit": "sample-steps/s", "cores": orc.num_threads(), "kind": "port",
            "sample": f"{reps} full solves of racing N={n} T={T} in {dt:.1f} s (oracle C port, OpenMP over samples; "
                      "noise generation excluded)", "solves_per_sec": reps / dt}


def cpu_baseline_torch(torch, np, T):
    """The reference's own op structure on this host's cores: oracle/
    torch_reference_loop.py — one [N,T,dc] torch.randn draw, two Python loops of T batched torch ops over the
    racing plugins (strided [N,ds] views of S[N,T+1,ds]), the dead action-cost product, softmax, weighted sum, batch-1
    rollout — at the metric's size (C3: N = 1,048,576, T = 50, ~4.3 GB resident).  Pinned against the reference
    fixtures by tests/test_oracle_vs_golden.py.

    These strided batch ops do not scale with cores (128 intra-op threads are ~7x SLOWER than 16 on a 2 x 64-core
    host), so `value` is the BEST of a short scan over thread counts (1 warm-up + 2 timed solves each) — the baseline
    is not handicapped by its default — and the all-cores figure the survey asked for is reported next to it."""
    from envs.racing_controller import racing_controller
    from envs.racing_env import RacingEnv
    from oracle.torch_reference_loop import TorchReferenceLoop

    n = 1 << 20
    cpu = torch.device("cpu")
    env = RacingEnv(device=cpu)
    ctrl = racing_controller(env, device=cpu, horizon=T, num_samples=n, lambda_=1.0, mppi_cls=TorchReferenceLoop)
    ctrl.set_cost_map(env._obstacle_map, env._lane_map)
    state = env.reset()
    ref, _ = ctrl.calc_ref_trajectory(state, env.racing_center_path, 0, T, DL=0.1, lookahead_distance=3,
                                      reference_path_interval=0.85)
    ctrl.set_reference(ref)
    default_threads = torch.get_num_threads()

    def timed(threads, reps):
        torch.set_num_threads(threads)
        ctrl.solver.forward(state.clone())  # warm-up (first call also pages in ~4 GB)
        out = []
        for _ in range(reps):
            t0 = time.perf_counter()
            ctrl.solver.forward(state.clone())
            out.append(time.perf_counter() - t0)
        return out

    scan = {}
    for threads in sorted({min(16, default_threads), min(8, default_threads), min(32, default_threads)}):
        scan[threads] = float(np.median(timed(threads, 2)))
    best = min(scan, key=scan.get)
    all_cores = timed(default_threads, 1)[0] if default_threads not in scan else scan[default_threads]
    torch.set_num_threads(default_threads)
    cpu_model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": n * T / scan[best], "unit": "sample-steps/s", "cores": best, "kind": "port",
            "sample": f"racing N={n} T={T}, torch-CPU restatement of the reference loop (noise draw included): median of 2 "
                      f"solves after 1 warm-up per thread count; best = {best} threads, {scan[best]:.2f} s per solve",
            "solves_per_sec": 1 / scan[best], "s_per_solve_by_threads": {str(k): v for k, v in scan.items()},
            "all_cores": {"torch_threads": default_threads, "s_per_solve": all_cores, "solves_per_sec": 1 / all_cores},
            "nproc": os.cpu_count(), "cpu_model": cpu_model}


if __name__ == "__main__":
    main()
