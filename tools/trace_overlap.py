#!/usr/bin/env python3
"""Two (or more) streams in flight: how busy is the GPU?  From a rocprofv3 kernel_trace.csv: over the middle 60 % of the trace's time span — wall time, the
union of all kernel intervals (GPU busy), the sum of kernel durations (overlap = sum - union), idle time and its largest holes.
    python tools/trace_overlap.py <kernel_trace.csv> [t_lo_frac] [t_hi_frac]"""
import csv
import sys

path = sys.argv[1]
lo_f = float(sys.argv[2]) if len(sys.argv) > 2 and len(sys.argv) <= 4 else 0.35
hi_f = float(sys.argv[3]) if len(sys.argv) > 3 and len(sys.argv) <= 4 else 0.60
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", ""))))
rows.sort()
t0, t1 = rows[0][0], rows[-1][1]
lo, hi = t0 + (t1 - t0) * lo_f, t0 + (t1 - t0) * hi_f
if len(sys.argv) > 4:   # window by step markers: from the a-th to the b-th last launch of the marker kernel (one launch per step), e.g. ce_rows -35 -20
    marks = [r[0] for r in rows if sys.argv[4] in r[2]]
    lo, hi = marks[int(sys.argv[2])], marks[int(sys.argv[3])]
    print(f"{len(marks)} launches of {sys.argv[4]}; window = launches {sys.argv[2]} .. {sys.argv[3]}: {int(sys.argv[3]) - int(sys.argv[2])} steps")
win = [r for r in rows if r[0] >= lo and r[1] <= hi]
wall = (win[-1][1] - win[0][0]) / 1e6
tot = sum(e - s for s, e, _, _ in win) / 1e6
busy, cur_s, cur_e, holes = 0, win[0][0], win[0][1], []
for s, e, n, q in win[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        holes.append(((s - cur_e) / 1e3, n))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
busy /= 1e6
streams = sorted(set(q for _, _, _, q in win))
print(f"window {wall:.2f} ms, {len(win)} kernels on streams/queues {streams}: GPU busy (union) {busy:.2f} ms = {busy / wall * 100:.1f} %, sum of kernel durations {tot:.2f} ms "
      f"(overlapped {tot - busy:.2f} ms = {(tot - busy) / wall * 100:.1f} % of the wall), idle {wall - busy:.3f} ms")
holes.sort(reverse=True)
print("largest idle holes (us, next kernel):", [(round(h, 1), n[:50]) for h, n in holes[:8]])
# concurrency histogram: time with 0 / 1 / 2+ kernels resident
ev = sorted([(s, 1) for s, e, _, _ in win] + [(e, -1) for s, e, _, _ in win])
lvl, prev, hist = 0, ev[0][0], {}
for t, d in ev:
    hist[lvl] = hist.get(lvl, 0) + (t - prev)
    lvl += d
    prev = t
print("time with k kernels in flight:", {k: f"{v / 1e6:.2f} ms" for k, v in sorted(hist.items())})
