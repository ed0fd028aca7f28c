#!/bin/bash
# round-4 GPU call N: split verify mode with the chained split GEMMs (norms / GELU / SwiGLU write the split image): tests + timing + kernel table
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04n
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "bf16x3 or norm" > $O/t_kernels.log 2>&1; tail -2 $O/t_kernels.log
timeout 1500 python -m pytest tests/test_model_gpu.py -q -x -s -k "split_verify or stack_entry" > $O/t_model.log 2>&1; grep "^\[\|passed\|failed\|^E " $O/t_model.log | cut -c1-220
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_x3 -- python $R/bench.py --dtype bf16x3 --steps 5 --warmup 2 --no-extra-legs --no-cpu-baseline --no-roofline --no-projection > $O/prof_x3.json 2> $O/prof_x3.err
cd $R
find $O/prof_x3 -name "*kernel_trace.csv" | head -1 | xargs -I{} python tools/trace_gaps.py {} --steps 3 > $O/x3_gaps.md
find $O/prof_x3 -type f -size +3M -delete
sed -n 1,4p $O/x3_gaps.md; sed -n 24,52p $O/x3_gaps.md | sort -t'|' -k6 -nr | head -16
timeout 600 python bench.py --dtype bf16x3 --steps 10 --warmup 3 --no-extra-legs --no-cpu-baseline --no-roofline --no-projection > $O/bench_x3.json 2>/dev/null
python -c "import json; d=json.load(open('$O/bench_x3.json')); print('bf16x3 timed', d['ms_per_step'], d['ms_per_step_blocks']['ms'], d['parity']['logits_max_abs_err'], d['parity']['top1_agreement'])"
