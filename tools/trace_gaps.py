#!/usr/bin/env python3
"""Where the step time goes BETWEEN kernels: reads a `rocprofv3 --kernel-trace --output-format csv` kernel trace of bench.py, cuts it into
steps at the `ce_rows_kernel` launches (one per step) and reports, for the last N steps: wall span, sum of kernel durations, sum of the
gaps between consecutive kernels, the gap distribution, and the gaps grouped by the kernel that FOLLOWS them (a host-bound launch shows
up as a long gap in front of it; a dependent-launch boundary as ~1.5-2 us).

    python tools/trace_gaps.py <kernel_trace.csv> [--steps 3]"""
import argparse
import csv
import re
from collections import defaultdict


def fam(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z_0-9:]+)(<[^(]*>)?", name)
    base = m.group(1) if m else name
    if base.startswith("at::native"):
        return "torch:" + name[:60]
    return base + (m.group(2) or "" if m else "")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--steps", type=int, default=3)
    a = ap.parse_args()
    rows = []
    for r in csv.DictReader(open(a.csv)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    cuts = [i for i, r in enumerate(rows) if "ce_rows_kernel" in r[2]]
    if len(cuts) < a.steps + 1:
        raise SystemExit(f"only {len(cuts)} steps in the trace")
    lo, hi = cuts[-a.steps - 1] + 1, cuts[-1] + 1
    seg = rows[lo:hi]
    span = (seg[-1][1] - rows[lo - 1][1]) / 1e3
    busy = sum(e - s for s, e, _ in seg) / 1e3
    gaps = []
    by_next = defaultdict(list)
    prev_end = rows[lo - 1][1]
    for s, e, n in seg:
        g = (s - prev_end) / 1e3
        gaps.append(g)
        by_next[fam(n)].append(g)
        prev_end = max(prev_end, e)
    n = a.steps
    print(f"# gaps between kernels, last {n} steps of {a.csv.split('/')[-1]}\n")
    print(f"per step: wall {span / n / 1e3:.3f} ms, kernels {busy / n / 1e3:.3f} ms ({len(seg) / n:.0f} launches), gaps {sum(gaps) / n / 1e3:.3f} ms\n")
    gs = sorted(gaps)
    q = lambda p: gs[min(len(gs) - 1, int(p * len(gs)))]
    print(f"gap quantiles (us): p10 {q(.1):.2f}  p50 {q(.5):.2f}  p90 {q(.9):.2f}  p99 {q(.99):.2f}  max {gs[-1]:.1f}")
    for lim in (3, 5, 10, 50):
        big = [g for g in gaps if g > lim]
        print(f"  gaps > {lim:>2} us: {len(big) / n:6.1f} per step, {sum(big) / n / 1e3:.3f} ms per step")
    order = sorted(range(len(seg)), key=lambda i: -gaps[i])[:12]
    print("\nlargest gaps (us): previous kernel -> next kernel")
    for i in order:
        prev = rows[lo + i - 1][2]
        print(f"  {gaps[i]:8.1f}  {fam(prev)[:70]} -> {fam(seg[i][2])[:70]}  (launch {i} of the window)")
    print("\n| kernel that follows the gap | launches / step | mean gap us | total gap ms / step | kernel ms / step |")
    print("|---|---|---|---|---|")
    dur = defaultdict(float)
    for s, e, nme in seg:
        dur[fam(nme)] += (e - s) / 1e3
    for k, v in sorted(by_next.items(), key=lambda kv: -sum(kv[1])):
        print(f"| `{k[:90]}` | {len(v) / n:.1f} | {sum(v) / len(v):.2f} | {sum(v) / n / 1e3:.3f} | {dur[k] / n / 1e3:.3f} |")


if __name__ == "__main__":
    main()
