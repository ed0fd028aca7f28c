# Round-end validation on the GPU box: full GPU test suite, smoke(), the default bench line (two clips in flight + the single-stream bracket), the other
# configs, and rocprofv3 kernel traces of a short bench run with one stream (clean per-kernel durations, gaps) and with the default two.
#   /usr/local/graft/bin/gpurun --timeout 2400 -- "bash tools/gpu_validate.sh [outdir]"   (outputs under gpurun_out/<outdir>, default "validate")
O=gpurun_out/${1:-validate}; mkdir -p $O
(timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -6) > $O/gpu_tests.log; tail -2 $O/gpu_tests.log
(timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2) > $O/smoke.log; tail -1 $O/smoke.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['ms_per_step'], d['value'], d.get('single_stream'), d['roofline']['frac'], d.get('fp16',{}).get('ms_per_step'), d['cpu_baseline']['value'])"
for c in c3 c4 c5; do timeout 600 python bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline --no-projection > $O/bench_$c.json 2> $O/bench_$c.err; python -c "
import json; d=json.load(open('$O/bench_$c.json')); print('$c', d['ms_per_step'], d.get('single_stream'), d.get('parity', {}).get('logits_max_abs_err'))"; done
bash tools/gpu_profile.sh ${1:-validate}/prof_1stream 1 > /dev/null 2>&1; head -5 $O/prof_1stream/gaps.md
bash tools/gpu_profile.sh ${1:-validate}/prof_2streams 2 > /dev/null 2>&1; head -3 $O/prof_2streams/kernel_stats.md
