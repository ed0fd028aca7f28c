#!/bin/bash
# re-calibrated estimate of the 64 x 64 kernel: which shapes change kernel, per-shape times, bench A/B against the previous library (st-llm_amd/prev = HEAD~ build), tests
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04ab; mkdir -p $O; cd $R
for lib in prev new; do
  echo "== $lib"; if [ $lib = prev ]; then export STLLM_LIB=$R/st-llm_amd/prev/libstllm_hip.so; else unset STLLM_LIB; fi
  timeout 200 python tools/gemm_bench.py --iters 30 2>&1 | grep -v amdgpu | cut -c1-150
done > $O/gemm_bench_prev_new.log 2>&1
cat $O/gemm_bench_prev_new.log
for i in 1 2 3; do for lib in prev new; do
  if [ $lib = prev ]; then export STLLM_LIB=$R/st-llm_amd/prev/libstllm_hip.so; else unset STLLM_LIB; fi
  timeout 600 python bench.py --steps 60 --warmup 5 --no-extra-legs --no-cpu-baseline --no-projection > $O/b.json 2>/dev/null
  python -c "import json; d=json.load(open('$O/b.json')); r=d['roofline']['all_gemm_kernels_one_step']; print('$lib', d['ms_per_step'], d['ms_per_step_blocks']['ms'], d['parity']['logits_max_abs_err'], d['telemetry']['sclk_mhz']['mean'], {k: v['ms'] for k, v in r.items() if 'RESID' in k})"
done; done | tee $O/bench_ab.log
unset STLLM_LIB
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "bench_shapes or auto_dispatch or gemm_small or resid" > $O/t_k.log 2>&1; tail -2 $O/t_k.log
timeout 900 python -m pytest tests/test_model_gpu.py -q -x > $O/t_model.log 2>&1; tail -2 $O/t_model.log
