#!/bin/bash
# round-4 GPU call O: pairwise in-place K-split exchange of the one-wave kernel (RESID, s = 2): Llama o_proj / down, correctness (vs the 64x64 kernel, bit-identical repeats) + timeline
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04o
mkdir -p $O
cd $R
export LD_LIBRARY_PATH=$R/st-llm_amd:/opt/rocm/lib
for pair in 1 0; do
  echo "=== STLLM_GEMM_W4_PAIR=$pair (cold weights)"
  for c in 7 9; do STLLM_GEMM_W4_PAIR=$pair timeout 60 tools/gemm_harness 50 $c 0 1 32 1 700 | grep -v "max LDS\|HARNESS OK"; done
done > $O/harness_pair.log 2>&1
export LD_LIBRARY_PATH=$R/st-llm_amd/trace:/opt/rocm/lib
echo "=== timeline, pairwise" >> $O/harness_pair.log
timeout 60 tools/gemm_harness 30 7 1 1 32 1 700 | grep -v "^  wg\|max LDS" >> $O/harness_pair.log 2>&1
cut -c1-210 $O/harness_pair.log
unset LD_LIBRARY_PATH
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "test_gemm_w4 or resid" > $O/t.log 2>&1; tail -2 $O/t.log
