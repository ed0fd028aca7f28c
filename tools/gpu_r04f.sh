#!/bin/bash
# round-4 GPU call F: where do the MVM configs (c4 / c5) spend the time that is not GEMM?  kernel trace + gaps per step
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04f
mkdir -p $O
export TMPDIR=/tmp
for c in c4 c5; do
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$c -- python $R/bench.py --config $c --steps 6 --warmup 2 --no-extra-legs --no-cpu-baseline --no-roofline > $O/prof_$c.json 2> $O/prof_$c.err
  cd $R
  find $O/prof_$c -name "*kernel_trace.csv" | head -1 | xargs -I{} python tools/trace_gaps.py {} --steps 3 > $O/${c}_gaps.md
  find $O/prof_$c -type f -size +3M -delete
  head -24 $O/${c}_gaps.md; sed -n 24,60p $O/${c}_gaps.md | sort -t'|' -k6 -nr | head -14
done
