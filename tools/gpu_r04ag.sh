# dispatch audit over every configuration's row counts + the training step (after a cost-model change): lines where something beats the automatic choice by > 3 %
mkdir -p gpurun_out/r4ag
( echo "== c2"; timeout 280 python tools/gemm_bench.py --audit --iters 20 2>&1 | grep -v amdgpu | cut -c1-200
  for r in 8224,528 8224,1088 16448,1088 32896,2304 65792,2304; do echo "== rows $r"; timeout 280 python tools/gemm_bench.py --audit --iters 10 --rows $r --only vit_qkv,vit_proj,vit_fc1,vit_fc2,llm_qkv,llm_o,llm_gu,llm_down,lm_head 2>&1 | grep -v amdgpu | cut -c1-200; done
  echo "== training step"; timeout 280 python tools/gemm_bench.py --audit --iters 8 --train 2>&1 | grep -v amdgpu | cut -c1-200 ) > gpurun_out/r4ag/audit_all.log 2>&1
grep "^[a-z=]\|faster\|again" gpurun_out/r4ag/audit_all.log | cut -c1-230
