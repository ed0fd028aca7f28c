#!/bin/bash
# Ablation builds of the W-direct GEMM (gemm_wd.inc, STLLM_WD_EXPERIMENT bits): st-llm_amd/wdx<bits>/libstllm_hip.so = the shipped objects of
# st-llm_amd/build/ with gemm_wd_bf16.hip recompiled.  Results of those libraries are garbage: timing only (STLLM_LIB=... python tools/wd_check.py --time-only).
set -e
cd "$(dirname "$0")/.."
# argument: <experiment bits>[,RA,RW]
for a in "$@"; do
  x=${a%%,*}; extra=""; d=st-llm_amd/wdx$x
  if [ "$a" != "$x" ]; then rest=${a#*,}; ra=${rest%%,*}; rw=${rest#*,}; extra="-DSTLLM_T1_RA=$ra -DSTLLM_T1_RW=$rw"; d=st-llm_amd/wdx${x}_${ra}_${rw}; fi
  mkdir -p $d
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DSTLLM_WD_EXPERIMENT=$x $extra -x hip -c st-llm_amd/csrc/gemm_wd_bf16.hip -o $d/gemm_wd_bf16.o
  objs=$(ls st-llm_amd/build/*.o | grep -v gemm_wd_bf16)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/libstllm_hip.so $objs $d/gemm_wd_bf16.o
  echo built $d
done
