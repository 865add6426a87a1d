#!/usr/bin/env python
"""Fold the per-dispatch counter CSVs of tools/pmc_kernels.sh into profiles/rNN_counters.json: per kernel (launch order inside the
target scripts identifies the site) the averages over launches of every counter, MFMA-busy fraction, VALU per MFMA, and HBM bytes
per launch = 2 x FETCH_SIZE (gfx950 under-counts wide coalesced reads by 2x, MI355X_MICROARCH.md HBM section) + WRITE_SIZE, both
reported by rocprofv3 in KiB."""
import collections, csv, glob, json, re, sys

root = sys.argv[1]
# launch order per repetition in tools/one_conv.py (SHAPE=256,256,256,14,1) and tools/one_conv64.py (SHAPE=128,112)
ORDER = {"a": ["fwd conv3x3 256->256 @14x14 s1", "dgrad conv3x3 256->256 @14x14 s1", "wgrad conv3x3 256->256 @14x14 s1"],
         "b": ["fwd conv3x3 64->64 @112x112 s1 (+stats)", "fwd conv3x3 64->64 @112x112 s1 (+IN/PReLU on load, +stats)",
               "dgrad conv3x3 64->64 @112x112 s1", "dgrad conv3x3 64->64 @112x112 s1 (+residual add)",
               "dgrad conv3x3 64->64 @112x112 s1 (+InstanceNorm-backward sums)",
               "dgrad conv3x3 64->64 @112x112 s1 (+residual add, chained tail: prelu' and sums of the previous block)",
               "wgrad conv3x3 64->64 @112x112 s1", "wgrad conv3x3 64->64 @112x112 s1 (+IN/PReLU on load)"]}
FLOP = {"a": 2.0 * 256 * 14 * 14 * 256 * 256 * 9, "b": 2.0 * 128 * 112 * 112 * 64 * 64 * 9}
ALG_BYTES = {"a": {"fwd": 2 * 256 * 196 * 256 * 2 + 256 * 2304 * 2, "dgrad": 2 * 256 * 196 * 256 * 2 + 256 * 2304 * 2,
                   "wgrad": 2 * 256 * 196 * 256 * 2},
             "b": {"fwd": 2 * 128 * 12544 * 64 * 2, "dgrad": 2 * 128 * 12544 * 64 * 2, "wgrad": 2 * 128 * 12544 * 64 * 2}}
sites = collections.OrderedDict()
for f in sorted(glob.glob(root + "/[ab]*/p_counter_collection.csv")):
    grp = re.search(r"/([ab])\d+/", f).group(1)
    rows = [r for r in csv.DictReader(open(f))]
    # conv kernels only, in dispatch order; one row per (dispatch, counter)
    disp = collections.OrderedDict()
    for r in rows:
        k = r["Kernel_Name"]
        if not any(s in k for s in ("igemm", "wgrad_kernel", "wgrad8_kernel", "wgrad64", "dconv64")):
            continue
        disp.setdefault(int(r["Dispatch_Id"]), {"kernel": k})[r["Counter_Name"]] = float(r["Counter_Value"])
    seq = [disp[d] for d in sorted(disp)]
    n = len(ORDER[grp])
    for i, d in enumerate(seq):
        site = ORDER[grp][i % n]
        e = sites.setdefault(site, {"kernel": re.sub(r"\(anonymous namespace\)::|void ", "", d["kernel"]).split("(")[0], "_n": collections.Counter(),
                                    "_sum": collections.Counter(), "group": grp})
        for c, v in d.items():
            if c != "kernel":
                e["_sum"][c] += v
                e["_n"][c] += 1
out = {"source": "rocprofv3 --pmc <group> --kernel-trace, one pass per group (tools/pmc_kernels.sh), averages over the launches of "
                 "tools/one_conv.py (256->256 @14x14, batch 256) and tools/one_conv64.py (64->64 @112x112, batch 128); isolated "
                 "launches, cache-warm operands", "gfx950_fetch_correction": 2.0, "sites": {}}
for site, e in sites.items():
    m = {c: e["_sum"][c] / e["_n"][c] for c in e["_sum"]}
    r = {"kernel": e["kernel"], "counters": {c: round(v, 1) for c, v in sorted(m.items())}}
    cyc = m.get("GRBM_GUI_ACTIVE", 0.0) / 8.0       # rocprofv3 sums the 8 XCDs
    if cyc and "SQ_VALU_MFMA_BUSY_CYCLES" in m:
        r["kernel_cycles"] = round(cyc)
        r["mfma_busy_frac_of_simd_cycles"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024), 4)
    if m.get("SQ_INSTS_MFMA"):
        r["valu_per_mfma"] = round(m.get("SQ_INSTS_VALU", 0.0) / m["SQ_INSTS_MFMA"], 2)
        r["lds_inst_per_mfma"] = round(m.get("SQ_INSTS_LDS", 0.0) / m["SQ_INSTS_MFMA"], 2)
    if m.get("SQ_WAVE_CYCLES"):
        r["wait_any_frac_of_wave_cycles"] = round(m.get("SQ_WAIT_ANY", 0.0) / m["SQ_WAVE_CYCLES"], 4)
    if m.get("SQ_LDS_IDX_ACTIVE"):
        r["lds_bank_conflict_frac"] = round(m.get("SQ_LDS_BANK_CONFLICT", 0.0) / m["SQ_LDS_IDX_ACTIVE"], 4)
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        r["hbm_bytes_per_launch"] = int(2.0 * m["FETCH_SIZE"] * 1024 + m["WRITE_SIZE"] * 1024)
        kind = site.split()[0]
        r["algorithmic_bytes_per_launch"] = ALG_BYTES[e["group"]][kind]
    r["algorithmic_gflop_per_launch"] = round(FLOP[e["group"]] / 1e9, 2)
    out["sites"][site] = r
json.dump(out, sys.stdout, indent=1)
