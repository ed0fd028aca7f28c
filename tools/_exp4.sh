mkdir -p gpurun_out/r3w
(timeout 500 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "w4 or gemm" 2>&1 | tail -4) > gpurun_out/r3w/ktests.log; tail -2 gpurun_out/r3w/ktests.log
for cs in "4 43" "2 33" "3 32"; do
  set -- $cs
  echo "== case $1 tile $2" >> gpurun_out/r3w/epi.log
  LD_LIBRARY_PATH=st-llm_amd/trace timeout 60 tools/gemm_harness 5 $1 1 1 $2 1 0 2>&1 | grep -E "tag +[0-9]+ -> +[0-9]+|^vit" | sed 's/maxabs.*//' >> gpurun_out/r3w/epi.log
done
grep -v "^  tag  [1235] ->" gpurun_out/r3w/epi.log
for i in 1 2; do for lib in st-llm_amd/prev/libstllm_hip.so st-llm_amd/libstllm_hip.so; do STLLM_LIB=$lib timeout 200 python bench.py --no-cpu-baseline --no-extra-legs --steps 60 --warmup 5 2>gpurun_out/r3w/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"$lib\", d[\"ms_per_step\"], d[\"roofline\"][\"frac\"], d[\"parity\"][\"logits_max_abs_err\"], {k: v[\"ms\"] for k, v in d[\"roofline\"][\"all_gemm_kernels_one_step\"].items() if \"w4\" in k})" | tee -a gpurun_out/r3w/ab.log; done; done
