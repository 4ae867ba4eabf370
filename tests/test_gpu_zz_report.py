"""Last module of the GPU session (file order): writes what the parity tests MEASURED to gpurun_out/parity_report.json
and asserts the headroom — every FIXED tolerance is used to at most half, except the quantities named in EXCEPTIONS.
Quantities whose limit is max(1e-5, the reference's own measured spread) (entries carrying `reference_band`, see
tests/golden/make_golden.py) are held to that limit as it is: a band is a measurement of the reference, not a tolerance
of ours, and the report counts how many of those checks hold the plain 1e-5."""
import os

import pytest

import parity_report
from helpers import ROOT

pytestmark = pytest.mark.gpu

# fixed tolerances that may come closer than 2x to their limit, and why
EXCEPTIONS = {
    # randomised parameter sets deliberately include maps where many samples sit on cell boundaries
    "map_cell_flips",
}


def test_parity_report_and_headroom():
    if not parity_report.entries:
        pytest.skip("no parity measurements in this session (run the whole GPU suite)")
    banded = {e["quantity"] for e in parity_report.entries if "reference_band" in e}
    # the end-to-end quantities held to max(1e-5, 1.0 x the reference's measured band of 256 probes): how many checks sit
    # above the plain 1e-5, and how many above the band ITSELF (target: 0 — anything above passed only through the floor)
    fast = [e for e in parity_report.entries if "reference_band" in e and e["quantity"].endswith("_device")]
    be = [e for e in parity_report.entries if "reference_band" in e and not e["quantity"].endswith("_device")]
    totals = {"checks": len(be), "above_1e-5": sum(1 for e in be if not e.get("within_1e5", e["value"] <= 1e-5)),
              "above_limit": sum(1 for e in be if e["value"] > e["limit"]),
              "above_1e-5_and_above_the_band": sum(1 for e in be if not e.get("within_band", True)),
              "band_margin": 1.0, "probes_per_band": sorted({e.get("band_probes") for e in be if e.get("band_probes")}),
              # the opt-in device-resident LBPS search (lbps_search="device") is held to 1.5x the band end to end
              "opt_in_device_lbps": {"checks": len(fast), "above_limit": sum(1 for e in fast if e["value"] > e["limit"]),
                                     "above_1e-5_and_above_the_band": sum(1 for e in fast if not e.get("within_band", True))}}
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
    assert totals["opt_in_device_lbps"]["above_limit"] == 0, totals
    # the headline config: how many of the allowed boundary flips C3 really uses
    c3 = [e for e in parity_report.entries if e["quantity"] == "map_cell_flips" and e.get("n") == 1 << 20]
    assert c3 and max(e["value"] for e in c3) <= 20, c3
