#!/usr/bin/env python3
"""In-kernel timeline of the folded-LayerNorm GEMMs (trace build: python st-llm_amd/build.py --trace; STLLM_LIB=st-llm_amd/trace/libstllm_hip.so).
Runs the producer (proj shape) and the two consumers (qkv, fc1) with and without the fold and prints the mean ticks between stamps."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stllm_amd import hip, pack  # noqa: E402


def stamps(dbg):
    h = dbg.cpu().view(256, 64)
    trans = {}
    for g in range(256):
        n = int(h[g, 0])
        prev = None
        for i in range(1, min(n, 64)):
            v = int(h[g, i])
            if v == 0:
                continue
            tag, t = (v >> 56) & 0xff, v & ((1 << 56) - 1)
            if prev is not None:
                trans.setdefault((prev[0], tag), []).append(t - prev[1])
            prev = (tag, t)
    return {k: (len(v), sum(v) / len(v)) for k, v in sorted(trans.items())}


def run(name, fn):
    dbg = torch.zeros(256 * 64, dtype=torch.int64, device="cuda")
    hip.set_option("gemm_debug", 16)
    hip._trace_frames = dbg
    for _ in range(3):
        fn(dbg)
    torch.cuda.synchronize()
    dbg.zero_()
    fn(dbg)
    torch.cuda.synchronize()
    hip.set_option("gemm_debug", 0)
    print(name, hip.lib().stllm_last_kernel().decode())
    for (a, b), (n, m) in stamps(dbg).items():
        print(f"   tag {a} -> {b}: n={n} mean {m:9.0f} ticks")


def main():
    M, D = 4112, 1408
    dt = "bf16"
    td = torch.bfloat16
    x = torch.randn(M, D, device="cuda")
    g, b = torch.rand(D, device="cuda") + 0.5, torch.randn(D, device="cuda") * 0.1
    xb, st = hip.row_stats(x, dt)
    h = hip.layernorm(x, g, b, 1e-6, dtype=dt)[0]
    orig = hip.gemm

    def gemm_dbg(dbg, *a, **k):   # hip.gemm with the debug buffer in `frames`
        args_hook["dbg"] = dbg
        return orig(*a, **k)
    args_hook = {}
    real = hip.lib().stllm_gemm

    class Wrap:
        def __call__(self, argp, stream):
            argp._obj.frames = args_hook["dbg"].data_ptr()
            return real(argp, stream)
    hip.lib().stllm_gemm = Wrap()
    for N, gelu in ((4224, False), (6144, True)):
        w = torch.randn(N, D, device="cuda") * 0.05
        bias = torch.randn(N, device="cuda") * 0.1
        wf, bf, cs = pack.fold_layernorm(w, bias, g, b, dt)
        w16 = w.to(td)
        act = hip.ACT_GELU if gelu else hip.ACT_NONE
        run(f"plain   N={N}", lambda dbg: gemm_dbg(dbg, h, w16, dtype=dt, bias=bias, act=act))
        run(f"folded  N={N}", lambda dbg: gemm_dbg(dbg, xb, wf, dtype=dt, bias=bf, act=act, fold_in=(st, 1e-6, cs)))
    a = (torch.randn(M, D, device="cuda") * 0.5).to(td)
    wp = (torch.randn(D, D, device="cuda") * 0.05).to(td)
    bp = torch.randn(D, device="cuda")
    run("producer plain ", lambda dbg: gemm_dbg(dbg, a, wp, dtype=dt, epilogue=hip.EPI_RESID, bias=bp, resid=x))
    run("producer folded", lambda dbg: gemm_dbg(dbg, a, wp, dtype=dt, epilogue=hip.EPI_RESID, bias=bp, resid=x, fold_out=(xb, st)))


if __name__ == "__main__":
    main()
