#!/usr/bin/env python3
"""Markdown summary of a `rocprofv3 --kernel-trace --stats --output-format csv` run: per-kernel total / calls / average,
with the torch integer-hash kernels of the synthetic weight generator (stllm_amd.synth fills the parameters on the GPU once at
start-up) filtered out.

    python tools/prof_summary.py <kernel_stats.csv> [--div N] [--top K] [--title "..."]
`--div N` divides totals and call counts by N (steps / tokens in the trace)."""
import argparse
import csv
import re


def is_generator(name):
    return (("long" in name and "at::native" in name) or "arange_cuda" in name or "philox" in name.lower()
            or "distribution_" in name)


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = name.split("(")[0] if name.startswith(("gemm", "gemv", "attn", "norm", "adamw", "transpose", "ce_", "sumsq")) else name
    return name[:110]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--div", type=float, default=1.0)
    ap.add_argument("--top", type=int, default=30)
    ap.add_argument("--title", default="")
    a = ap.parse_args()
    rows = [r for r in csv.DictReader(open(a.csv)) if not is_generator(r["Name"])]
    tot = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6 / a.div
    if a.title:
        print(f"# {a.title}\n")
    print(f"Kernel time per unit: **{tot:.3f} ms** over {len(rows)} kernel symbols (weight-generator kernels removed; totals divided by {a.div:g}).\n")
    print("| kernel | ms / unit | launches / unit | avg us | % |")
    print("|---|---|---|---|---|")
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[: a.top]:
        ms = float(r["TotalDurationNs"]) / 1e6 / a.div
        print(f"| `{short(r['Name'])}` | {ms:.3f} | {float(r['Calls']) / a.div:.1f} | {float(r['AverageNs']) / 1e3:.1f} | {100 * ms / tot:.1f} |")


if __name__ == "__main__":
    main()
