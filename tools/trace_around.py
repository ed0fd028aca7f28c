#!/usr/bin/env python3
"""The kernels in front of and behind the LAST launch of <symbol substring> in a rocprofv3 kernel_trace.csv: start (us, relative), duration, gap.
    python tools/trace_around.py <kernel_trace.csv> gather_rows [before] [after]"""
import csv
import sys

path, sym = sys.argv[1], sys.argv[2]
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 12
na = int(sys.argv[4]) if len(sys.argv) > 4 else 6
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", ""), r.get("Stream_Id", "")))
rows.sort()
idx = max(i for i, r in enumerate(rows) if sym in r[2])
t0 = rows[idx][0]
prev_end = None
for i in range(max(0, idx - nb), min(len(rows), idx + na + 1)):
    s, e, n, q, st = rows[i]
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print(f"{'>>' if i == idx else '  '} start {(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f} us  gap {gap:7.1f} us  q {q} s {st}  {n[:90]}")
    prev_end = e
