#!/bin/bash
# round-4 GPU call D: is the weight stream what paces the M = 576 GEMMs?  w4 tiles with and without it (trace_x4 build: every W piece reads the same 8 rows)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04d
mkdir -p $O
cd $R
for lib in st-llm_amd st-llm_amd/trace_x4; do
  export LD_LIBRARY_PATH=$R/$lib:/opt/rocm/lib
  echo "=== library $lib (cold weights: 700 MiB of W copies rotating)"
  for c in 6 7 8 9; do for w in 32 34; do timeout 120 tools/gemm_harness 40 $c 0 1 $w 0 700 | grep -v "max LDS\|HARNESS"; done; done
  echo "=== library $lib (warm weights)"
  for c in 6 8; do for w in 32 34; do timeout 120 tools/gemm_harness 40 $c 0 1 $w 0 0 | grep -v "max LDS\|HARNESS"; done; done
done > $O/harness_nostream.log 2>&1
cat $O/harness_nostream.log | cut -c1-230
