#!/usr/bin/env python3
"""Micro-benchmark of the decode-regime GEMMs at Vicuna-7B size (weights rotate over enough copies to come from HBM every launch).
   python tools/gemv_bench.py [rows ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stllm_amd import hip, pack

rows = [int(a) for a in sys.argv[1:]] or [1, 5]
SHAPES = [("qkv rope", 12288, 4096, "rope"), ("o resid", 4096, 4096, "resid"), ("gate/up swiglu", 22016, 4096, "swiglu"), ("down resid", 4096, 11008, "resid")]
dt = "bf16"
for M in rows:
    tot = 0.0
    for name, N, K, kind in SHAPES:
        ncopy = max(2, int(600e6 // (N * K * 2)) + 1)
        ws = [(torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16) for _ in range(ncopy)]
        a = (torch.randn(M, K, device="cuda") * 0.5).to(torch.bfloat16)
        x = torch.randn(M, N, device="cuda")
        cos = torch.ones(1, 64, device="cuda"); sin = torch.zeros(1, 64, device="cuda")
        def run(w):
            if kind == "resid":
                hip.gemm(a, w, dtype=dt, epilogue=hip.EPI_RESID, resid=x)
            elif kind == "swiglu":
                hip.gemm(a, w, dtype=dt, epilogue=hip.EPI_SWIGLU)
            else:
                hip.gemm(a, w, dtype=dt, epilogue=hip.EPI_ROPE, rope=(cos, sin), rope_seq=1, rope_cols=N // 3 * 2)
        for w in ws[:2]:
            run(w)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 3 * ncopy
        s.record()
        for i in range(reps):
            run(ws[i % ncopy])
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) * 1e3 / reps
        tot += us
        print(f"M={M:2d} {name:15s} N={N:5d} K={K:5d} {us:7.1f} us  {N * K * 2 / us / 1e6:5.2f} TB/s  [{hip.lib().stllm_last_kernel().decode()}]")
        del ws
    print(f"M={M:2d} per layer {tot:.1f} us -> x32 = {tot * 32 / 1e3:.2f} ms")
