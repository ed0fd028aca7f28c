# calibration: the vendor library's GEMMs (hipBLASLt / rocBLAS through torch.matmul) on the hot path's shapes next to ours, then the kernel names under rocprofv3
mkdir -p gpurun_out/r4r
(timeout 300 python tools/gemm_bench.py --vendor --iters 30 2>&1 | grep -v amdgpu.ids) > gpurun_out/r4r/gemm_vs_vendor.log; cat gpurun_out/r4r/gemm_vs_vendor.log
export TMPDIR=/tmp; cd /tmp; (timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4r/prof -- python $GRAFT_REPO_ROOT/tools/gemm_bench.py --vendor --iters 10 > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/r4r/prof.err); cd $GRAFT_REPO_ROOT
find gpurun_out/r4r/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r4r/kernel_stats.csv
find gpurun_out/r4r/prof -name "*kernel_trace.csv" -delete
cut -c1-260 gpurun_out/r4r/kernel_stats.csv | head -60
