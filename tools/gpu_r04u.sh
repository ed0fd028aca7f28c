# does the 8 KiB row stride of the K = 4096 operands cost bandwidth (channel aliasing)?  Same GEMMs with the rows of A and W 64 / 32 elements further apart.
mkdir -p gpurun_out/r4u
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/st-llm_amd:$LD_LIBRARY_PATH
for pad in 0 64 32 0 64; do
  echo "== row stride K + $pad"
  for cw in "6 24" "6 0" "7 0" "8 0" "9 0" "8 34" "6 32"; do set -- $cw
    HARNESS_LDPAD=$pad timeout 120 tools/gemm_harness 40 $1 0 1 $2 0 700 | grep -v "max LDS\|HARNESS"
  done
done > gpurun_out/r4u/ldpad.log 2>&1
cut -c1-250 gpurun_out/r4u/ldpad.log
