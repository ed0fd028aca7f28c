#!/usr/bin/env python3
"""Run ONE GEMM shape a few times (for rocprofv3 --pmc passes).  python tools/gemm_one.py M N K [iters] [store|resid|gelu|rope|swiglu]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stllm_amd import hip
M, N, K = map(int, sys.argv[1:4])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 3
epi = sys.argv[5] if len(sys.argv) > 5 else "store"
A = (torch.rand(M, K, device="cuda") * 2 - 1).to(torch.bfloat16)
W = ((torch.rand(N, K, device="cuda") * 2 - 1) * 0.05).to(torch.bfloat16)
x = torch.rand(M, N, device="cuda")
b = torch.rand(N, device="cuda")
for _ in range(iters):
    if epi == "resid":
        hip.gemm(A, W, dtype="bf16", epilogue=hip.EPI_RESID, bias=b, resid=x)
    elif epi == "gelu":
        hip.gemm(A, W, dtype="bf16", bias=b, act=hip.ACT_GELU)
    elif epi == "swiglu":
        hip.gemm(A, W, dtype="bf16", epilogue=hip.EPI_SWIGLU)
    elif epi == "rope":
        from stllm_amd import pack
        cos, sin = pack.rope_tables(M, device="cuda")
        Wf = pack.frag32_or_none(W)   # the model's packed layer carries the fragment-major copy: the W-direct kernel where it applies
        hip.gemm(A, W, dtype="bf16", epilogue=hip.EPI_ROPE, rope=(cos, sin), rope_seq=M, rope_cols=(N // 3) * 2 // 128 * 128, **({"w_frag": Wf} if Wf is not None else {}))
    else:
        hip.gemm(A, W, dtype="bf16")
torch.cuda.synchronize()
print("done", hip.lib().stllm_last_kernel().decode())
