#!/bin/bash
# 128 x 256 one-wave tile for the Llama prefill qkv GEMM in the model: bench A/B (STLLM_GEMM_W4_WIDE = 0 / 1 alternating on one box) + the GEMM / model tests on the new default
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04x
mkdir -p $O
cd $R
for i in 1 2 3; do
  for w in 0 1; do
    STLLM_GEMM_W4_WIDE=$w timeout 600 python bench.py --steps 60 --warmup 5 --no-extra-legs --no-cpu-baseline --no-projection > $O/b.json 2>/dev/null
    python -c "import json; d=json.load(open('$O/b.json')); r=d['roofline']['all_gemm_kernels_one_step']; print('wide=$w', d['ms_per_step'], d['ms_per_step_blocks']['ms'], d['parity']['logits_max_abs_err'], d['telemetry']['sclk_mhz']['mean'], {k: v['ms'] for k, v in r.items() if 'ROPE' in k})"
  done
done | tee $O/bench_ab_wide.log
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "bench_shapes or rope or gemm" > $O/t_k.log 2>&1; tail -2 $O/t_k.log
timeout 900 python -m pytest tests/test_model_gpu.py -q -x -k "stack_entry or c2_full_size_vs or llama or kv_cache" > $O/t_model.log 2>&1; tail -2 $O/t_model.log
