set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for s in "4112 1408 1408 resid" "4112 1408 6144 resid" "4112 6144 1408 gelu" "4112 4224 1408 store" "576 12288 4096 rope" "576 4096 4096 resid"; do bash tools/pmc_traffic.sh 6 $s 2>&1 | tail -1; done
cp profiles/traffic_r06.json gpurun_out/traffic_r06.json
# group-M = 4 variant (experiment library): fc1 + qkv traffic into a scratch file, times of the four ViT GEMMs from both libraries
for s in "4112 6144 1408 gelu" "4112 4224 1408 store" "4112 1408 6144 resid"; do STLLM_LIB=st-llm_amd/gm4/libstllm_hip.so bash tools/pmc_traffic.sh 96 $s 2>&1 | tail -1; done
cp profiles/traffic_r96.json gpurun_out/traffic_r06_group_m4.json
for lib in st-llm_amd/libstllm_hip.so st-llm_amd/gm4/libstllm_hip.so st-llm_amd/libstllm_hip.so st-llm_amd/gm4/libstllm_hip.so; do echo "== $lib"; STLLM_LIB=$lib python tools/gemm_bench.py 2>&1 | grep "^vit_"; done
