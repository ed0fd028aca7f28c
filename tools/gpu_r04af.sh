# dispatch audits at the row counts of the other configurations and of the training step
mkdir -p gpurun_out/r4af
for r in 16448,1088 32896,2304; do echo "== rows $r"; timeout 280 python tools/gemm_bench.py --audit --iters 12 --rows $r --only vit_qkv,vit_proj,vit_fc1,vit_fc2 2>&1 | grep -v amdgpu | cut -c1-200; done > gpurun_out/r4af/audit_vit_rows.log 2>&1
(echo "== training step"; timeout 280 python tools/gemm_bench.py --audit --iters 8 --train 2>&1 | grep -v amdgpu | cut -c1-200) > gpurun_out/r4af/audit_train.log 2>&1
grep "^[a-z=]\|faster\|again" gpurun_out/r4af/audit_vit_rows.log gpurun_out/r4af/audit_train.log | cut -c1-230
