# in-kernel timeline (trace build) of the 128 x 256 tile on the Llama qkv shape, next to 256 x 128 on the ViT fc1 shape (same per-wave MFMA count)
mkdir -p gpurun_out/r4v
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/st-llm_amd/trace:$LD_LIBRARY_PATH
for cw in "6 24" "20 24" "8 24" "4 42" "6 32"; do set -- $cw
  timeout 120 tools/gemm_harness 30 $1 1 1 $2 0 0 | grep -v "max LDS\|HARNESS"
done > gpurun_out/r4v/timeline.log 2>&1
cut -c1-300 gpurun_out/r4v/timeline.log
