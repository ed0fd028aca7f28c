# K-loop ablations of the one-wave kernel (trace-x builds; results garbage, timing only): 8 = no barrier, 16 = no wait for the next unit's pieces, 32 = no LDS-DMA issue, 24 = 8 + 16, 56 = all
mkdir -p gpurun_out/r4w
for x in "" _x8 _x16 _x32 _x24 _x56; do
  [ -d st-llm_amd/trace$x ] || continue
  echo "== trace$x"
  for cw in "6 24" "5 32" "4 43"; do set -- $cw
    LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/st-llm_amd/trace$x timeout 120 tools/gemm_harness 30 $1 1 1 $2 0 0 | grep "tag  2 ->  4\|tag  8 ->  4\|^[a-z]" | cut -c1-200
  done
done > gpurun_out/r4w/kloop_ablation.log 2>&1
cat gpurun_out/r4w/kloop_ablation.log
