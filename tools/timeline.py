"""Per-step timeline from a rocprofv3 kernel trace (gpurun_out/ev_stats/a_kernel_trace.csv): which hardware queue runs
what when, how long no kernel at all is running, and how the step splits into forward / backward / optimizer.
usage: timeline.py [trace.csv] [step index from the end, default 3]"""
import csv, sys
from collections import defaultdict

path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/ev_stats/a_kernel_trace.csv"
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), r["Kernel_Name"]))
rows.sort()
# steps are delimited by the optimizer kernel
adam = [i for i, r in enumerate(rows) if "adamw_kernel" in r[3]]
lo, hi = adam[-back - 1], adam[-back]
step = rows[lo + 1:hi + 1]
t0, t1 = step[0][0], step[-1][1]
print(f"step: {len(step)} kernels, {(t1 - t0) / 1e3:.1f} us from the first kernel start to the end of adamw")
# union of busy intervals
iv = sorted((s, e) for s, e, _, _ in step)
busy, cur_s, cur_e, gaps = 0, iv[0][0], iv[0][1], []
for s, e in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, cur_e - t0))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"some kernel running: {busy / 1e3:.1f} us; nothing running: {(t1 - t0 - busy) / 1e3:.1f} us in {len(gaps)} gaps")
gaps.sort(reverse=True)
print("largest gaps (us, at us):", [(round(g / 1e3, 1), round(at / 1e3)) for g, at in gaps[:8]])
perq = defaultdict(float)
for s, e, q, _ in step:
    perq[q] += e - s
print("kernel time per hardware queue (us):", {q: round(v / 1e3) for q, v in sorted(perq.items())})
# concurrency histogram: time with k kernels running
ev = sorted([(s, 1) for s, e, _, _ in step] + [(e, -1) for s, e, _, _ in step])
lvl, last, hist = 0, ev[0][0], defaultdict(float)
for t, d in ev:
    hist[lvl] += t - last
    lvl += d
    last = t
print("time with k kernels in flight (us):", {k: round(v / 1e3) for k, v in sorted(hist.items())})
# phase boundaries: first backward kernel = first kernel whose name contains 'bwd' or tn_kernel
firstb = next(i for i, r in enumerate(step) if "bwd" in r[3] or "tn_kernel" in r[3] or "EPI" in r[3] and False)
print(f"forward (+losses) ends ~{(step[firstb][0] - t0) / 1e3:.0f} us; backward+optimizer {(t1 - step[firstb][0]) / 1e3:.0f} us")
if "-v" in sys.argv:
    for s, e, q, n in step:
        print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} q{q} {n[:90]}")
