# Round-end validation on the GPU box: full GPU test suite, smoke(), the default bench line and a rocprofv3 kernel trace of a short bench run.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- "bash tools/gpu_validate.sh"   (outputs under gpurun_out/r4t/)
mkdir -p gpurun_out/r4t
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6) > gpurun_out/r4t/gpu_tests.log; tail -2 gpurun_out/r4t/gpu_tests.log
(timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2) > gpurun_out/r4t/smoke.log; tail -1 gpurun_out/r4t/smoke.log
timeout 600 python bench.py > gpurun_out/r4t/bench.json 2> gpurun_out/r4t/bench.err; python -c "
import json; d=json.load(open('gpurun_out/r4t/bench.json')); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d.get('fp16',{}).get('ms_per_step'), d['cpu_baseline']['value'])"
export TMPDIR=/tmp; cd /tmp; (timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4t/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra-legs --no-projection > $GRAFT_REPO_ROOT/gpurun_out/r4t/prof_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/r4t/prof.err); cd $GRAFT_REPO_ROOT
find gpurun_out/r4t/prof -name "*kernel_trace.csv" | head -1 | xargs -I{} python tools/trace_gaps.py {} --steps 3 > gpurun_out/r4t/gaps.md; head -8 gpurun_out/r4t/gaps.md
find gpurun_out/r4t/prof -name "*.csv" -size +20M -delete
cat gpurun_out/r4t/prof_bench.json | head -c 400
