#!/bin/bash
# round-4 GPU call P: o_proj on the one-wave kernel with the pairwise exchange, in the model: bench A/B (STLLM_GEMM_W4_PAIR = 0 / 1 alternating) + model tests
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04p
mkdir -p $O
cd $R
for i in 1 2 3; do
  for pair in 0 1; do
    STLLM_GEMM_W4_PAIR=$pair timeout 600 python bench.py --steps 60 --warmup 5 --no-extra-legs --no-cpu-baseline --no-projection > $O/b.json 2>/dev/null
    python -c "import json; d=json.load(open('$O/b.json')); r=d['roofline']['all_gemm_kernels_one_step']; print('pair=$pair', d['ms_per_step'], d['ms_per_step_blocks']['ms'], d['parity']['logits_max_abs_err'], d['telemetry']['sclk_mhz']['mean'], {k: v['ms'] for k, v in r.items() if 'RESID' in k})"
  done
done | tee $O/bench_ab_pair.log
timeout 900 python -m pytest tests/test_model_gpu.py -q -x -k "stack_entry or c2_full_size_vs or llama or kv_cache" > $O/t_model.log 2>&1; tail -2 $O/t_model.log
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "bench_shapes or resid" > $O/t_k.log 2>&1; tail -2 $O/t_k.log
