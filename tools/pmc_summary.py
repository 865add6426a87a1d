#!/usr/bin/env python
"""Per-kernel averages of the counters collected by tools/pmc_counters.sh."""
import collections, csv, glob, sys
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(sys.argv[1] + "/g*/p_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "igemm" in k or "wgrad_kernel" in k:
            short = k.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0][:70]
            vals[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in vals.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    cyc = m.get("GRBM_GUI_ACTIVE", 0) / 8
    print(k)
    print("   " + "  ".join(f"{c}={v:.3g}" for c, v in sorted(m.items())))
    if cyc and "SQ_VALU_MFMA_BUSY_CYCLES" in m:
        print(f"   kernel cycles {cyc:.0f}; MFMA busy / (cycles x 1024 SIMDs) = {m['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024):.3f}; "
              f"LDS active / (cycles x 256 CUs) = {m.get('SQ_LDS_IDX_ACTIVE', 0) / (cyc * 256):.3f}; "
              f"wave-cycles x4 / waves = {4 * m.get('SQ_WAVE_CYCLES', 0) / max(m.get('SQ_WAVES', 1), 1):.0f}")
