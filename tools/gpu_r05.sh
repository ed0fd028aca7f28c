mkdir -p gpurun_out/r05a
python -m pytest tests/test_model_gpu.py -x -q -k "stack_entry or qformer or stllm_forward or generate" > gpurun_out/r05a/tests.log 2>&1
echo "tests rc $?" >> gpurun_out/r05a/tests.log
python bench.py --steps 40 --warmup 5 --no-extra-legs --no-projection --no-cpu-baseline > gpurun_out/r05a/bench_new.json 2> gpurun_out/r05a/bench_new.err
STLLM_STACK_ENTRY=0 python bench.py --steps 40 --warmup 5 --no-extra-legs --no-projection --no-cpu-baseline > gpurun_out/r05a/bench_perop.json 2> gpurun_out/r05a/bench_perop.err
python bench.py --steps 40 --warmup 5 --no-extra-legs --no-projection --no-cpu-baseline > gpurun_out/r05a/bench_new2.json 2> gpurun_out/r05a/bench_new2.err
python tools/host_timeline.py --steps 2 > gpurun_out/r05a/host_timeline.log 2>&1
tail -3 gpurun_out/r05a/tests.log
grep -o '"ms_per_step": [0-9.]*' gpurun_out/r05a/bench_*.json
cat gpurun_out/r05a/host_timeline.log | cut -c1-600
