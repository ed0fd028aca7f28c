#!/bin/bash
# RoPE table ring in the 128 x 256 epilogue: timeline (trace build), kernel tests, bench
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04y; mkdir -p $O; cd $R
LD_LIBRARY_PATH=$R/st-llm_amd/trace timeout 120 tools/gemm_harness 30 6 1 1 24 0 0 | grep -v "^  wg\|max LDS" | cut -c1-220 > $O/timeline_24_rope.log; cat $O/timeline_24_rope.log
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "bench_shapes or rope or gemm" > $O/t_k.log 2>&1; tail -2 $O/t_k.log
for i in 1 2; do
timeout 600 python bench.py --steps 60 --warmup 5 --no-extra-legs --no-cpu-baseline --no-projection > $O/b.json 2>/dev/null
python -c "import json; d=json.load(open('$O/b.json')); r=d['roofline']['all_gemm_kernels_one_step']; print(d['ms_per_step'], d['ms_per_step_blocks']['ms'], d['parity']['logits_max_abs_err'], d['telemetry']['sclk_mhz']['mean'], {k: v['ms'] for k, v in r.items() if 'ROPE' in k})"
done | tee $O/bench.log
timeout 900 python -m pytest tests/test_model_gpu.py -q -x -k "stack_entry or c2_full_size_vs or llama or kv_cache" > $O/t_model.log 2>&1; tail -2 $O/t_model.log
