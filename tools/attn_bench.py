#!/usr/bin/env python3
"""Micro-benchmark of the attention shapes on the hot path (config 2).  GPU only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stllm_amd import hip

CASES = [("vit", 16, 16, 257, 257, 88, False, 39), ("llama", 1, 32, 576, 576, 128, True, 32), ("llama_b4", 4, 32, 576, 576, 128, True, 32),
         ("qf_self", 16, 12, 32, 32, 64, False, 12), ("qf_cross", 16, 12, 32, 257, 64, False, 6)]
if "--probe" in sys.argv:   # how does the Llama prefill kernel's time scale with the keys a workgroup walks / the workgroups in flight?  (non-causal: every workgroup alike)
    CASES += [("nc_k128", 1, 32, 576, 128, 128, False, 0), ("nc_k256", 1, 32, 576, 256, 128, False, 0), ("nc_k288", 1, 32, 576, 288, 128, False, 0),
              ("nc_k576", 1, 32, 576, 576, 128, False, 0), ("nc_k288_h51", 1, 51, 576, 288, 128, False, 0), ("nc_k288_h64", 1, 64, 576, 288, 128, False, 0),
              ("c_s288", 1, 32, 288, 288, 128, True, 0), ("c_s288_h64", 1, 64, 288, 288, 128, True, 0)]
AUDIT = "--audit" in sys.argv   # the Llama prefill kernel's work split (query tiles per workgroup x waves per query tile) forced through option attn_dma = 10 nw + ks
if AUDIT:
    CASES = [(f"llama_S{S}_B{B}", B, 32, S, S, 128, True, 32) for B, S in ((1, 178), (1, 296), (1, 400), (1, 528), (1, 576), (1, 700), (1, 1088), (4, 576), (2, 576), (16, 576))]
for name, B, H, Sq, Skv, D, causal, per_clip in CASES:
    dt = torch.bfloat16
    if Sq == Skv:
        buf = torch.randn(B * Sq, 3 * H * D, device="cuda").to(dt)
        q, k, v = buf[:, :H * D], buf[:, H * D:2 * H * D], buf[:, 2 * H * D:]
    else:
        q = torch.randn(B * Sq, H * D, device="cuda").to(dt)
        kv = torch.randn(B * Skv, 2 * H * D, device="cuda").to(dt)
        k, v = kv[:, :H * D], kv[:, H * D:]
    out = hip.attention(q, k, v, B=B, H=H, Sq=Sq, Skv=Skv, D=D, scale=D ** -0.5, causal=causal)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        hip.attention(q, k, v, B=B, H=H, Sq=Sq, Skv=Skv, D=D, scale=D ** -0.5, causal=causal, out=out)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 20
    fl = 4.0 * B * H * Sq * Skv * D * (0.5 if causal else 1.0)
    print(f"{name:9s} B={B:3d} H={H:3d} Sq={Sq:4d} Skv={Skv:4d} D={D:4d} {ms * 1e3:8.1f} us  {fl / ms / 1e9:7.1f} TF   x{per_clip} = {ms * per_clip:.3f} ms")
    if AUDIT:
        rows = []
        for cfg in (121, 81, 61, 41, 62, 42, 32, 22, 34, 24, 14):
            hip.set_option("attn_dma", cfg)
            try:
                for _ in range(3): hip.attention(q, k, v, B=B, H=H, Sq=Sq, Skv=Skv, D=D, scale=D ** -0.5, causal=causal, out=out)
                s.record()
                for _ in range(20): hip.attention(q, k, v, B=B, H=H, Sq=Sq, Skv=Skv, D=D, scale=D ** -0.5, causal=causal, out=out)
                e.record(); torch.cuda.synchronize()
                rows.append((s.elapsed_time(e) / 20 * 1e3, cfg))
            except Exception as ex:   # noqa: BLE001
                rows.append((float("inf"), cfg))
            finally:
                hip.set_option("attn_dma", 1)
        rows.sort()
        print("      " + "  ".join(f"{c // 10}x{c % 10}: {us:.1f}" for us, c in rows[:6]) + (f"   <-- {ms * 1e3 - rows[0][0]:.1f} us faster than the automatic choice" if rows[0][0] < 0.97 * ms * 1e3 else ""))
