#!/usr/bin/env python
"""Per-queue timeline of ONE training step from a rocprofv3 kernel trace (p_kernel_trace.csv of tools/prof_bench.sh / prof_run.sh):
busy time and first / last activity of every HIP queue inside the step, the union busy time, idle gaps, and the chain of kernels
that ends the step -- which stream is the critical path of the backward pass?   python tools/trace_timeline.py <csv> [step_index]"""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
which = int(sys.argv[2]) if len(sys.argv) > 2 else -2
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), r["Kernel_Name"]) for r in rows]
ev.sort()
# step boundaries: the optimizer kernel (sgd_kernel / rmsprop_kernel / adam_kernel) ends a step
ends = [i for i, e in enumerate(ev) if re.search(r"sgd_kernel|rmsprop_kernel|adam_kernel", e[3])]
# group consecutive optimizer launches (several per step in C3 / C4)
bounds = []
for i in ends:
    if bounds and ev[i][0] - ev[bounds[-1]][1] < 2_000_000:
        bounds[-1] = i
    else:
        bounds.append(i)
lo = bounds[which - 1] + 1
hi = bounds[which]
step = ev[lo:hi + 1]
t0, t1 = step[0][0], max(e[1] for e in step)
print(f"step {which}: {len(step)} kernels, wall {(t1 - t0) / 1e6:.3f} ms")
byq = collections.defaultdict(list)
for s, e, q, n in step:
    byq[q].append((s, e, n))
for q, lst in sorted(byq.items()):
    busy = sum(e - s for s, e, _ in lst)
    print(f"  queue {q}: {len(lst):5d} kernels, busy {busy / 1e6:7.3f} ms, active {(lst[0][0] - t0) / 1e6:7.3f} .. {(max(e for _, e, _ in lst) - t0) / 1e6:7.3f} ms")
# union busy and gaps
iv = sorted((s, e) for s, e, _, _ in step)
cur_s, cur_e = iv[0]
union = 0
gaps = []
for s, e in iv[1:]:
    if s > cur_e:
        union += cur_e - cur_s
        gaps.append((s - cur_e, cur_e - t0))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
print(f"  some kernel running: {union / 1e6:.3f} ms; idle gaps: {sum(g for g, _ in gaps) / 1e6:.3f} ms in {len(gaps)} gaps "
      f"(largest {max(gaps)[0] / 1e3:.1f} us at {max(gaps)[1] / 1e6:.2f} ms)" if gaps else "  no gaps")
# time with >= 2 queues busy
pts = []
for s, e, q, _ in step:
    pts.append((s, 1)); pts.append((e, -1))
pts.sort()
depth = 0; last = pts[0][0]; hist = collections.Counter()
for t, d in pts:
    hist[depth] += t - last
    last = t
    depth += d
print("  concurrency histogram (ms): " + ", ".join(f"{k}: {v / 1e6:.3f}" for k, v in sorted(hist.items())))
print("  last 12 kernels of the step:")
for s, e, q, n in sorted(step, key=lambda x: x[1])[-12:]:
    n = re.sub(r"\(anonymous namespace\)::|void ", "", n).split("(")[0][:60]
    print(f"    q{q} {(s - t0) / 1e6:7.3f} -> {(e - t0) / 1e6:7.3f} ms  {(e - s) / 1e3:7.1f} us  {n}")
