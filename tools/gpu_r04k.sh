#!/bin/bash
# round-4 GPU call K: 4-stage LDS ring of the 64 x 64 tile (Q-Former-sized GEMMs): harness old (2 stages, st-llm_amd/prev) vs new, GEMM tests, bench A/B
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04k
mkdir -p $O
cd $R
for lib in st-llm_amd/prev st-llm_amd; do
  export LD_LIBRARY_PATH=$R/$lib:/opt/rocm/lib
  echo "=== library $lib"
  for c in 0 1 21 22 23 24; do timeout 60 tools/gemm_harness 200 $c 0 0 0 0 0 | grep -v "max LDS\|HARNESS"; done
done > $O/harness_ring.log 2>&1
cut -c1-150 $O/harness_ring.log
unset LD_LIBRARY_PATH
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "gemm" > $O/t_gemm.log 2>&1; tail -2 $O/t_gemm.log
for i in 1 2; do
  for lib in st-llm_amd/prev/libstllm_hip.so st-llm_amd/libstllm_hip.so; do
    STLLM_LIB=$R/$lib timeout 600 python bench.py --steps 40 --warmup 5 --no-extra-legs --no-cpu-baseline --no-projection > $O/b.json 2>/dev/null
    python -c "import json; d=json.load(open('$O/b.json')); print('$lib', d['ms_per_step'], d['ms_per_step_blocks']['ms'], d['parity']['logits_max_abs_err'], d['telemetry']['sclk_mhz']['mean'])"
  done
done | tee $O/bench_ab_ring.log
