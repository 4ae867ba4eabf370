#!/usr/bin/env python3
"""Short digest of a bench.py JSON line (file argument): headline, stages, the secondary configs."""
import json
import sys

line = [ln for ln in open(sys.argv[1]).read().splitlines() if ln.startswith("{")]
if not line:
    sys.exit("no JSON line in " + sys.argv[1])
d = json.loads(line[-1])
print(f"ms_per_step {d['ms_per_step']}  value {d['value']:.4g}  steps {d['steps']}  roofline.frac {d.get('roofline', {}).get('frac')}")
print("stages_ms", d.get("stages_ms"))
for k in ("eager_state_seq", "lazy_state_seq"):
    if k in d:
        print(k, d[k]["ms_per_step"], d[k]["stages_ms"])
if "sharded_one_rank" in d:
    print("sharded_one_rank", {k: (v.get("value_over_unsharded") if isinstance(v, dict) else v) for k, v in d["sharded_one_rank"].items()})
if "closed_loop" in d:
    print("closed_loop ms/tick", d["closed_loop"]["ms_per_tick"])
if "example_loop" in d:
    print("example_loop ms/tick", d["example_loop"]["ms_per_tick"])
for k, v in (d.get("other_configs") or {}).items():
    st = v.get("stages_ms")
    print(f"  {k:14s} {v['ms_per_solve'] * 1e3:8.1f} us/solve", "" if not st else {a: round(b * 1e3, 1) for a, b in st.items()})
for k in ("cpu_baseline", "cpu_baseline_torch"):
    if k in d:
        print(k, d[k].get("value"), d[k].get("cores"))
