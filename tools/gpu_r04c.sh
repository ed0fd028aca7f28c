#!/bin/bash
# round-4 GPU call C: the whole GPU suite (ABI 5, FOLD removed, split mode, fp32 matrix-core attention, c3/c4/c5 full-size), default bench, split-step profile
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04c
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x -s > $O/gpu_tests_full.log 2>&1; echo "suite rc $?" >> $O/gpu_tests_full.log
grep -h "^\[\|passed\|failed\|^FAILED\|^E  " $O/gpu_tests_full.log | cut -c1-240 > $O/gpu_tests.log
timeout 900 python bench.py --steps 30 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?" >> $O/bench.err
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_x3 -- python $R/bench.py --dtype bf16x3 --steps 5 --warmup 2 --no-extra-legs --no-cpu-baseline --no-roofline > $O/prof_x3.json 2> $O/prof_x3.err
cd $R
find $O/prof_x3 -name "*kernel_trace.csv" | head -1 | xargs -I{} python tools/trace_gaps.py {} --steps 3 > $O/x3_gaps.md
find $O/prof_x3 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/x3_kernel_stats.csv
find $O/prof_x3 -type f -size +5M -delete
tail -n 25 $O/gpu_tests.log
sed -n 20,50p $O/x3_gaps.md
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04c/bench.json"))
print(d["ms_per_step"], d["parity"]["logits_max_abs_err"], d["parity"]["fp32_verify"]["ms_per_step"], d["parity"]["split_verify"], d["telemetry"])
for n,b in d["frame_parallel_projection"]["n"].items(): print(n, b["frames_per_rank"], [s["ms"] for s in b["shares"]], b["projected_ms"], b["projected_speedup"])
PY
