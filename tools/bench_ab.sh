#!/bin/bash
# Same-box A/B of bench.py under environment switches, alternating:  tools/bench_ab.sh <out.log> <reps> "VAR=a" "VAR=b" ...
# Each arm: the headline bench without the extra legs (50 timed steps); prints ms_per_step, the block medians, the clock and the per-shape GEMM table.
out=$1; reps=$2; shift 2
: > $out
for r in $(seq 1 $reps); do
  for arm in "$@"; do
    echo "== $arm" >> $out
    env $arm python bench.py --steps 50 --warmup 5 --no-extra-legs --no-cpu-baseline --no-projection 2>/dev/null | python -c '
import json, sys
d = json.loads(sys.stdin.readlines()[-1])
print("ms_per_step", d["ms_per_step"], "blocks", d.get("ms_per_step_blocks", {}).get("ms"), "sclk", d.get("telemetry", {}).get("sclk_mhz", {}).get("mean"))
for k, v in ((d.get("roofline") or {}).get("per_shape_all_ge_1ms") or {}).items():
    print("   ", k, v["avg_launch_us"], "us x", v["launches"])
' >> $out 2>&1
  done
done
cat $out
