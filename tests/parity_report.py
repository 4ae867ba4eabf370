"""Collector of MEASURED parity errors during a GPU test session (test infrastructure).

The parity tests are pass/fail; this module keeps what they measured — per test: the quantity, its value, the limit it
was checked against — so that the session ends with a tracked report (gpurun_out/parity_report.json on the GPU box, copied
to profiles/rNN_parity_report.json) that says how much of each tolerance is actually used.  tests/test_gpu_zz_report.py
writes the file and asserts the headroom."""
import json
import os

current_test = "?"
entries = []


def record(quantity: str, value: float, limit: float, **extra) -> None:
    """`value` was checked against `limit` (value <= limit) in the test that is running."""
    entries.append(dict(test=current_test, quantity=quantity, value=float(value), limit=float(limit), **extra))


def summary() -> dict:
    out = {}
    for e in entries:
        q = out.setdefault(e["quantity"], dict(count=0, worst_fraction_of_limit=0.0, worst=None, limit_reached=0))
        q["count"] += 1
        frac = e["value"] / e["limit"] if e["limit"] > 0 else (0.0 if e["value"] == 0 else float("inf"))
        if frac >= q["worst_fraction_of_limit"]:
            q["worst_fraction_of_limit"], q["worst"] = frac, e
        if frac > 0.5:
            q["limit_reached"] += 1
    return out


def write(path: str, **extra) -> dict:
    os.makedirs(os.path.dirname(path), exist_ok=True)
    rep = dict(summary=summary(), **extra, entries=entries)
    with open(path, "w") as f:
        json.dump(rep, f, indent=1)
    return rep
