#!/bin/bash
# Trace build of the attention kernels (STLLM_ATTN_TRACE: s_memtime stamps per wave): st-llm_amd/attn_trace/libstllm_hip.so = the shipped objects of
# st-llm_amd/build/ with attention.hip recompiled.  Use: STLLM_LIB=st-llm_amd/attn_trace/libstllm_hip.so python tools/attn_trace.py vit|llama
set -e
cd "$(dirname "$0")/.."
d=st-llm_amd/attn_trace
mkdir -p $d
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DSTLLM_ATTN_TRACE "$@" -x hip -c st-llm_amd/csrc/attention.hip -o $d/attention.o
objs=$(ls st-llm_amd/build/*.o | grep -v "/attention.hip.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/libstllm_hip.so $objs $d/attention.o
echo built $d
