#!/usr/bin/env python
"""Fold the per-dispatch FETCH_SIZE / WRITE_SIZE CSVs of tools/pmc_step.sh into profiles/rNN_step_traffic.json: HBM bytes per training
step = (2 x FETCH_SIZE + WRITE_SIZE) KiB summed over every kernel dispatch of the profiled run / number of steps (the factor 2 is the
gfx950 correction for wide coalesced reads, MI355X_MICROARCH.md HBM section), with the split by kernel family."""
import collections, csv, glob, json, os, re, sys

root, works = sys.argv[1], sys.argv[2:]
FAM = [("weight gradients", ("wgrad",)), ("slab sums", ("unpack_",)), ("direct 64-channel conv", ("dconv64",)),
       ("8-wave implicit GEMM", ("igemm8",)), ("4-wave implicit GEMM", ("igemm_kernel",)),
       ("norm / activation passes", ("affine_act", "group_stats")), ("BatchNorm + SE tails / glue", ("bnse", "norm_", "reduce_groups", "small_atb", "se_excite")),
       ("packs / optimizers / losses / layout", ("pack_", "rmsprop", "sgd_", "adam", "loss_", "nchw", "nhwc", "copy", "subsample", "maxpool", "upadd", "dropout"))]
out = {"source": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE --kernel-trace (separate passes, tools/pmc_step.sh) over tools/prof_work.py; "
                 "bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024", "gfx950_fetch_correction": 2.0}
for w in works:
    tot = {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0}
    fam = collections.OrderedDict((f, 0.0) for f, _ in FAM)
    fam["other"] = 0.0
    steps = None
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        files = glob.glob(os.path.join(root, f"{w}_{c}", "**", "*counter_collection.csv"), recursive=True)
        log = open(os.path.join(root, f"{w}_{c}.log")).read()
        steps = len(re.findall(r"^step \d+:", log, flags=re.M)) or steps
        for f in files:
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] != c:
                    continue
                v = float(r["Counter_Value"]) * 1024.0 * (2.0 if c == "FETCH_SIZE" else 1.0)
                tot[c] += v
                k = r["Kernel_Name"]
                for name, keys in FAM:
                    if any(s in k for s in keys):
                        fam[name] += v
                        break
                else:
                    fam["other"] += v
    n = 128 if w == "c3" else 256
    out[w] = {"batch": n, "steps_profiled": steps, "hbm_bytes_per_step": int((tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) / max(steps or 1, 1)),
              "read_bytes_per_step": int(tot["FETCH_SIZE"] / max(steps or 1, 1)), "write_bytes_per_step": int(tot["WRITE_SIZE"] / max(steps or 1, 1)),
              "by_family_bytes_per_step": {k: int(v / max(steps or 1, 1)) for k, v in fam.items()}}
json.dump(out, sys.stdout, indent=1)
