# the 128 x 256 one-wave tile (shape 24) on the Llama prefill shapes: harness, cold (700 MiB of rotating W copies) and warm, next to the current default kernels
mkdir -p gpurun_out/r4t2
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/st-llm_amd:$LD_LIBRARY_PATH
for c in 6 7 8 9 10 2 4; do
  timeout 120 tools/gemm_harness 30 $c 0 1 24 1 700 | grep -v "max LDS\|HARNESS"
done > gpurun_out/r4t2/w4_24_cold.log 2>&1
for c in 6 7 8 9 10; do
  timeout 120 tools/gemm_harness 30 $c 0 1 24 0 0 | grep -v "max LDS\|HARNESS"
done > gpurun_out/r4t2/w4_24_warm.log 2>&1
cat gpurun_out/r4t2/w4_24_cold.log gpurun_out/r4t2/w4_24_warm.log | cut -c1-250
