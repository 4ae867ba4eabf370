"""Last module of the GPU session (file order): writes what the parity tests MEASURED to gpurun_out/parity_report.json
and asserts the headroom — every FIXED tolerance is used to at most half, except the quantities named in EXCEPTIONS.
Quantities whose limit is max(1e-5, the reference's own measured spread) (entries carrying `reference_band`, see
tests/golden/make_golden.py) are held to that limit as it is: a band is a measurement of the reference, not a tolerance
of ours, and the report counts how many of those checks hold the plain 1e-5."""
import glob
import json
import os
import re

import pytest

import parity_report
from helpers import ROOT

pytestmark = pytest.mark.gpu

# fixed tolerances that may come closer than 2x to their limit, and why
EXCEPTIONS = {
    # randomised parameter sets deliberately include maps where many samples sit on cell boundaries
    "map_cell_flips",
}
# A banded check may move inside the reference's own spread from one build to the next (another equally valid rounding), but
# not by more than this share of the probes relative to the committed report (profiles/rNN_parity_report.json, the newest one
# that carries ranks) — a quarter of the distribution is a regression even when the sample maximum still holds.
RANK_JUMP = 0.25


def committed_ranks():
    """{(test, quantity, ordinal): (rank share, value)} of the newest committed report that carries ranks, and its name."""
    best = None
    for path in glob.glob(os.path.join(ROOT, "profiles", "r*_parity_report.json")):
        m = re.search(r"r(\d+)_parity_report", os.path.basename(path))
        if m and (best is None or int(m.group(1)) > best[0]):
            best = (int(m.group(1)), path)
    if best is None:
        return {}, None
    try:
        rep = json.load(open(best[1]))
    except (OSError, ValueError):
        return {}, None
    out = {}
    for e in rep.get("entries", ()):
        if e.get("rank") is not None and e.get("band_probes"):
            out[(e["test"], e["quantity"], e.get("ordinal", 0))] = (e["rank"] / e["band_probes"], e["value"])
    return out, os.path.basename(best[1])


def test_parity_report_and_headroom():
    if not parity_report.entries:
        pytest.skip("no parity measurements in this session (run the whole GPU suite)")
    banded = {e["quantity"] for e in parity_report.entries if "reference_band" in e}
    # the end-to-end quantities held to max(1e-5, 1.0 x the reference's measured band of 256 probes): how many checks sit
    # above the plain 1e-5, and how many above the band ITSELF (target: 0 — anything above passed only through the floor)
    fast = [e for e in parity_report.entries if "reference_band" in e and e["quantity"].endswith("_grid")]
    be = [e for e in parity_report.entries if "reference_band" in e and not e["quantity"].endswith("_grid")]
    totals = {"checks": len(be), "above_1e-5": sum(1 for e in be if not e.get("within_1e5", e["value"] <= 1e-5)),
              "above_limit": sum(1 for e in be if e["value"] > e["limit"]),
              "above_1e-5_and_above_the_band": sum(1 for e in be if not e.get("within_band", True)),
              "band_margin": 1.0, "probes_per_band": sorted({e.get("band_probes") for e in be if e.get("band_probes")}),
              # the opt-in grid search of LBPS (lbps_search="grid") is held to 1.5x the band end to end
              "opt_in_device_lbps": {"checks": len(fast), "above_limit": sum(1 for e in fast if e["value"] > e["limit"]),
                                     "above_1e-5_and_above_the_band": sum(1 for e in fast if not e.get("within_band", True))}}
    # every check that sits ON its band (value = the reference's sample maximum to within 1e-3) must BE one of the reference's
    # probes bit for bit — the device then computes exactly that equally valid evaluation of the costs, which is an explanation;
    # anything else on the edge is luck that a compiler bump would turn red
    edge = [e for e in be if e["limit"] > 0 and e["value"] / e["limit"] >= 0.999 and e["value"] > 1e-5]
    unexplained = [(e["test"], e["quantity"], e["value"]) for e in edge if not e.get("coincides_with")]
    totals["on_the_band"] = len(edge)
    totals["on_the_band_explained_by_a_probe"] = len(edge) - len(unexplained)
    totals["probes_coincided_with"] = sorted({n.split(":")[-1].rstrip("0123456789").rstrip("_") for e in be for n in e.get("coincides_with") or ()})
    # ... and no check may have moved up the reference's distribution by more than RANK_JUMP since the committed report
    base, base_name = committed_ranks()
    jumps = []
    for e in be:
        key = (e["test"], e["quantity"], e.get("ordinal", 0))
        if key in base and e.get("rank") is not None and e.get("band_probes") and e["value"] > 1e-5:
            share = e["rank"] / e["band_probes"]
            if share - base[key][0] > RANK_JUMP:
                jumps.append((e["test"], e["quantity"], f"rank share {base[key][0]:.2f} -> {share:.2f}", f"value {base[key][1]:.2e} -> {e['value']:.2e}"))
    totals["rank_baseline"] = base_name
    totals["rank_baseline_checks_matched"] = sum(1 for e in be if (e["test"], e["quantity"], e.get("ordinal", 0)) in base)
    totals["rank_jumps_beyond_%.2f" % RANK_JUMP] = len(jumps)
    totals["strict_parity"] = bool(parity_report.strict)
    rep = parity_report.write(os.path.join(ROOT, "gpurun_out", "parity_report.json"), banded_totals=totals)
    print("banded end-to-end checks:", totals)
    tight = {q: v["worst_fraction_of_limit"] for q, v in rep["summary"].items()
             if v["worst_fraction_of_limit"] > (1.0 if q in banded else 0.5) and q not in EXCEPTIONS}
    for q, v in sorted(rep["summary"].items()):
        extra = ""
        if q in banded:
            es = [e for e in parity_report.entries if e["quantity"] == q]
            n5 = sum(1 for e in es if e.get("within_1e5", e["value"] <= 1e-5))
            nb = sum(1 for e in es if e.get("within_band", True))
            extra = f"  [{n5}/{len(es)} within 1e-5; {nb}/{len(es)} within max(floor, 1.0x the reference's own measured band)]"
        print(f"{q:52s} n={v['count']:4d}  worst {v['worst']['value']:.3e} of {v['worst']['limit']:.3e} "
              f"({100 * v['worst_fraction_of_limit']:.1f} %) in {v['worst']['test']}{extra}")
    assert not tight, f"beyond the limit / less than 2x headroom on a fixed tolerance: {tight}"
    assert totals["above_limit"] == 0 and totals["above_1e-5_and_above_the_band"] == 0, totals
    assert not unexplained, f"checks sitting on the reference's sample maximum without being one of its probes: {unexplained}"
    assert not jumps, f"checks that moved up the reference's own distribution by more than {RANK_JUMP} since {base_name}: {jumps}"
    assert totals["opt_in_device_lbps"]["above_limit"] == 0, totals
    # the headline config: how many of the allowed boundary flips C3 really uses
    c3 = [e for e in parity_report.entries if e["quantity"] == "map_cell_flips" and e.get("n") == 1 << 20]
    assert c3 and max(e["value"] for e in c3) <= 20, c3
