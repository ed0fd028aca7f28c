# Round-end validation on the GPU box: full GPU test suite, smoke(), the default bench line and a rocprofv3 kernel trace of a short bench run.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- "bash tools/gpu_validate.sh [outdir]"   (outputs under gpurun_out/<outdir>, default "validate")
O=gpurun_out/${1:-validate}; mkdir -p $O
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6) > $O/gpu_tests.log; tail -2 $O/gpu_tests.log
(timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2) > $O/smoke.log; tail -1 $O/smoke.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d.get('fp16',{}).get('ms_per_step'), d['cpu_baseline']['value'])"
export TMPDIR=/tmp; cd /tmp; (timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra-legs --no-projection > $GRAFT_REPO_ROOT/$O/prof_bench.json 2> $GRAFT_REPO_ROOT/$O/prof.err); cd $GRAFT_REPO_ROOT
find $O/prof -name "*kernel_trace.csv" | head -1 | xargs -I{} python tools/trace_gaps.py {} --steps 3 > $O/gaps.md; head -8 $O/gaps.md
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c "python tools/prof_summary.py {} --div 24 --top 40 > $O/kernel_stats.md; cp {} $O/kernel_stats.csv"
rm -rf $O/prof
head -c 400 $O/prof_bench.json
