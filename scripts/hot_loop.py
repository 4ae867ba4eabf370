#!/usr/bin/env python3
"""Instruction census of the hot loop of rollout_cost_kernel<racing, fast, regen> from a hipcc -save-temps listing:
    python scripts/hot_loop.py /tmp/v/<name>/mppi_capi-hip-amdgcn-amd-amdhsa-gfx950.s [--dump]
(the longest single-block loop of the kernel = two racing steps + one float4 of noise)."""
import collections
import re
import sys

KERNEL = "_ZN4mppi19rollout_cost_kernelILi4ELi2ELb1ELb1EEE"
TRANS = r"v_(log|sqrt|sin|cos|rcp|rsq|exp)_"


def main():
    s = open(sys.argv[1]).read()
    m = re.search(r"^(" + KERNEL + r"\w*):", s, re.M)
    body = s[m.start():s.index(".Lfunc_end", m.start())].split("\n")
    labels = {l.split(":")[0]: i for i, l in enumerate(body) if re.match(r"\.LBB\d+_\d+:", l)}
    best = None
    for i, l in enumerate(body):
        mm = re.match(r"\s+s_cbranch_\w+ (\.LBB\d+_\d+)", l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
            j = labels[mm.group(1)]
            if not any(re.match(r"\.LBB", x) for x in body[j + 1:i]) and (best is None or i - j > best[1] - best[0]):
                best = (j, i)
    loop = [l.strip() for l in body[best[0] + 1:best[1] + 1] if l.strip() and not l.strip().startswith(";")]
    ops = collections.Counter(l.split()[0] for l in loop)
    trans = sum(v for k, v in ops.items() if re.match(TRANS, k))
    valu = sum(v for k, v in ops.items() if k.startswith("v_"))
    salu = sum(v for k, v in ops.items() if k.startswith("s_") and not k.startswith("s_waitcnt"))
    print(f"loop {len(loop)} instructions: VALU {valu} (transcendental {trans}, v_mad_u64_u32 {ops.get('v_mad_u64_u32', 0)}, "
          f"v_mov {ops.get('v_mov_b32_e32', 0)}), SALU {salu}, waitcnt {ops.get('s_waitcnt', 0)}, "
          f"LDS {sum(v for k, v in ops.items() if k.startswith('ds_'))}, "
          f"VMEM {sum(v for k, v in ops.items() if k.startswith(('global_', 'buffer_')))}")
    print("transcendental positions:", [i for i, l in enumerate(loop) if re.match(TRANS, l)])
    if "--dump" in sys.argv:
        print("\n".join(loop))


if __name__ == "__main__":
    main()
