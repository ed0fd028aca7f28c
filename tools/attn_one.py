#!/usr/bin/env python3
"""Run ONE attention shape a few times (for rocprofv3 --pmc passes).  python tools/attn_one.py vit|llama [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stllm_amd import hip
which = sys.argv[1] if len(sys.argv) > 1 else "vit"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 4
B, H, S, D, causal = (16, 16, 257, 88, False) if which == "vit" else (1, 32, 576, 128, True)
buf = torch.randn(B * S, 3 * H * D, device="cuda").to(torch.bfloat16)
q, k, v = buf[:, :H * D], buf[:, H * D:2 * H * D], buf[:, 2 * H * D:]
for _ in range(iters):
    out = hip.attention(q, k, v, B=B, H=H, Sq=S, Skv=S, D=D, scale=D ** -0.5, causal=causal)
torch.cuda.synchronize()
print("done")
