#!/usr/bin/env python
"""Kernel-time split by family from a rocprofv3 --stats CSV (p_kernel_stats.csv): ms per step.   python tools/stats_families.py <csv> <steps>"""
import csv, re, sys
FAM = [("weight gradients (sliced + direct + rows + 8-wave)", ("wgrad",)), ("slab sums (unpack_*)", ("unpack_",)),
       ("direct 64-channel fwd + dgrad", ("dconv64",)), ("8-wave implicit GEMM fwd + dgrad", ("igemm8",)),
       ("4-wave implicit GEMM fwd + dgrad", ("igemm_kernel",)), ("norm / activation streaming passes", ("affine_act", "group_stats")),
       ("BatchNorm + SE tails", ("bnse",)), ("small glue (statistic folds, packs, optimizers, losses, layout)", ("",))]
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
tot = {f: 0.0 for f, _ in FAM}
for r in rows:
    n = r["Name"]
    for f, keys in FAM:
        if any(k in n for k in keys):
            tot[f] += float(r["TotalDurationNs"]) / 1e6 / steps
            break
for f, v in tot.items():
    print(f"{v:8.2f} ms/step  {f}")
print(f"{sum(tot.values()):8.2f} ms/step  sum of kernel durations (streams overlap)")
top = sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:int(sys.argv[3]) if len(sys.argv) > 3 else 12]
for r in top:
    n = re.sub(r"\(anonymous namespace\)::|void ", "", r["Name"]).split("(")[0][:70]
    print(f"   {float(r['TotalDurationNs']) / 1e6 / steps:7.3f} ms/step {int(r['Calls']) / steps:6.1f} x {float(r['AverageNs']) / 1e3:8.1f} us  {n}")
