#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (kernel trace of bench.py): per-kernel time per step, launches per step, average."""
import collections, re, sqlite3, sys

path, steps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 6
db = sqlite3.connect(path)
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = db.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id").fetchall()
agg = collections.defaultdict(lambda: [0, 0.0])
for n, s, e in rows:
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n).split("(")[0]
    agg[n][0] += 1
    agg[n][1] += (e - s) / 1e3
tot = sum(v[1] for v in agg.values())
print(f"total kernel time per step: {tot / steps:.1f} us over {steps} steps (warm-up included)")
for n, v in sorted(agg.items(), key=lambda x: -x[1][1])[: int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print(f"{v[1] / steps:9.1f} us/step {v[0] / steps:6.1f} launches {v[1] / v[0]:8.1f} us avg  {n[:110]}")
