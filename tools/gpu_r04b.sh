#!/bin/bash
# round-4 GPU call B: bf16x3 kernel tests, full-size c3/c4/c5 + split-verify model tests, per-kernel profile of the split step, default bench
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04b
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -s -k "bf16x3" > $O/t_kernels.log 2>&1; echo "kernels rc $?" >> $O/t_kernels.log
timeout 1800 python -m pytest tests/test_model_gpu.py -q -x -s -k "split_verify or full_size_configs" > $O/t_model.log 2>&1; echo "model rc $?" >> $O/t_model.log
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_x3 -- python $R/bench.py --dtype bf16x3 --steps 5 --warmup 2 --no-extra-legs --no-cpu-baseline --no-roofline > $O/prof_x3.json 2> $O/prof_x3.err
cd $R
find $O/prof_x3 -name "*kernel_trace.csv" | head -1 | xargs -I{} python tools/trace_gaps.py {} --steps 3 > $O/x3_gaps.md
find $O/prof_x3 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/x3_kernel_stats.csv
find $O/prof_x3 -name "*.csv" -size +5M -delete; find $O/prof_x3 -type f -size +5M -delete
timeout 900 python bench.py --steps 30 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?" >> $O/bench.err
grep -h "^\[" $O/t_kernels.log $O/t_model.log | cut -c1-260
tail -n 3 $O/t_kernels.log $O/t_model.log $O/bench.err
head -30 $O/x3_gaps.md
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04b/bench.json"))
print(d["ms_per_step"], d["parity"]["logits_max_abs_err"], d["parity"]["split_verify"], d["telemetry"])
for n,b in d["frame_parallel_projection"]["n"].items(): print(n, b["frames_per_rank"], [s["ms"] for s in b["shares"]], b["projected_ms"], b["projected_speedup"])
PY
