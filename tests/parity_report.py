"""Collector of MEASURED parity errors during a GPU test session (test infrastructure).

The parity tests are pass/fail; this module keeps what they measured — per test: the quantity, its value, the limit it
was checked against — so that the session ends with a tracked report (gpurun_out/parity_report.json on the GPU box, copied
to profiles/rNN_parity_report.json) that says how much of each tolerance is actually used.  tests/test_gpu_zz_report.py
writes the file and asserts the headroom."""
import json
import os

current_test = "?"
entries = []
strict = False  # pytest --strict-parity (tests/conftest.py): full-size configurations are held to the plain 1e-5


def record(quantity: str, value: float, limit: float, **extra) -> None:
    """`value` was checked against `limit` (value <= limit) in the test that is running.  `ordinal` numbers the checks of one
    quantity inside one test: (test, quantity, ordinal) identifies a check from one session to the next."""
    ordinal = sum(1 for e in entries if e["test"] == current_test and e["quantity"] == quantity)
    entries.append(dict(test=current_test, quantity=quantity, ordinal=ordinal, value=float(value), limit=float(limit), **extra))


def summary() -> dict:
    out = {}
    for e in entries:
        q = out.setdefault(e["quantity"], dict(count=0, worst_fraction_of_limit=0.0, worst=None, limit_reached=0, on_the_band=0,
                                               on_the_band_explained_by_a_probe=0))
        q["count"] += 1
        frac = e["value"] / e["limit"] if e["limit"] > 0 else (0.0 if e["value"] == 0 else float("inf"))
        if frac >= q["worst_fraction_of_limit"]:
            q["worst_fraction_of_limit"], q["worst"] = frac, e
        if frac > 0.5:
            q["limit_reached"] += 1
        if "reference_band" in e and frac >= 0.999 and e["value"] > 1e-5:  # sits ON the reference's sample maximum
            q["on_the_band"] += 1
            q["on_the_band_explained_by_a_probe"] += 1 if e.get("coincides_with") else 0
    return out


def write(path: str, **extra) -> dict:
    os.makedirs(os.path.dirname(path), exist_ok=True)
    rep = dict(summary=summary(), **extra, entries=entries)
    with open(path, "w") as f:
        json.dump(rep, f, indent=1)
    return rep
