#!/bin/bash
# round-4 GPU call M: in-kernel timeline of the Llama o_proj / down GEMMs on the one-wave kernel (192 x 128 tile, K split in two) and on the phased kernel
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04m
mkdir -p $O
cd $R
export LD_LIBRARY_PATH=$R/st-llm_amd/trace:/opt/rocm/lib
for c in 7 9; do
  echo "=== case $c w4=32"; timeout 60 tools/gemm_harness 30 $c 1 1 32 1 700 | grep -v "^  wg\|max LDS"
  echo "=== case $c p8";    timeout 60 tools/gemm_harness 30 $c 1 1 0 1 700 | grep -v "^  wg\|max LDS"
done > $O/timeline_llm_o_down.log 2>&1
cat $O/timeline_llm_o_down.log | cut -c1-200
