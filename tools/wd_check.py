#!/usr/bin/env python3
"""GPU: the W-direct GEMM (gemm_wd.inc) against a float64 reference / the other kernels' epilogues + timing next to the automatic choice.
    python tools/wd_check.py [--iters 30]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from stllm_amd import hip, pack  # noqa: E402


def timeit(fn, iters):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def problem(M, N, K, epi, td):
    A = (torch.rand(M, K, device="cuda") * 2 - 1).to(td)
    W = ((torch.rand(N, K, device="cuda") * 2 - 1) * 0.05).to(td)
    bias = torch.rand(N, device="cuda")
    kw = dict(dtype=td, bias=bias)
    if epi == "store32": kw.update(out_f32=True)
    elif epi == "swiglu": kw.update(epilogue=hip.EPI_SWIGLU)
    elif epi == "rope":
        cos, sin = pack.rope_tables(M, 128, device="cuda")
        kw.update(epilogue=hip.EPI_ROPE, rope=(cos, sin), rope_seq=M, rope_cols=2 * N // 3)
    return A, W, kw


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--time-only", action="store_true")
    a = ap.parse_args()
    td = hip.torch_dtype(a.dtype)
    torch.manual_seed(0)
    ok = True
    cases = [(576, 1024, 1024, "store32", 4), (576, 1024, 1024, "store32", 6), (130, 256, 256, "store32", 4), (300, 512, 4096, "store", 4), (576, 1536, 1024, "rope", 4),
             (576, 1536, 1024, "rope", 6), (580, 1024, 2048, "swiglu", 4), (580, 1024, 2048, "swiglu", 6), (576, 12288, 4096, "rope", 4), (576, 22016, 4096, "swiglu", 6)]
    for (M, N, K, epi, shape) in cases * (0 if a.time_only else 1):
        A, W, kw = problem(M, N, K, epi, td)
        Wf = pack.frag32(W)
        hip.set_option("gemm_wd", 0)
        ref = hip.gemm(A, W, **kw).double()   # the other kernels (tested against fp64 / the oracle elsewhere)
        ref_name = hip.lib().stllm_last_kernel().decode()
        if epi == "store32":
            ref = A.double() @ W.double().t() + kw["bias"].double()
        hip.set_option("gemm_wd", shape)
        try:
            out = hip.gemm(A, W, w_frag=Wf, **kw)
            name = hip.lib().stllm_last_kernel().decode()
        finally:
            hip.set_option("gemm_wd", -1)
        err = (out.double() - ref).abs().max().item()
        scale = ref.abs().max().item()
        tol = (1e-4 if epi == "store32" else 2.0 ** -7) * scale   # 16-bit outputs: both kernels round the same fp32 sums, up to accumulation order
        nbad = ((out.double() - ref).abs() > 2.0 ** -9 * scale).float().mean().item() if epi != "store32" else 0.0
        good = err <= tol and name.startswith("gemm_wd") and nbad < 0.02
        ok &= good
        print(f"M={M} N={N} K={K} {epi:8s} shape {shape}: max err {err:.3e} (tol {tol:.3e}, {nbad * 100:.2f} % beyond half an ulp of the scale) [{name}] vs [{ref_name}] {'ok' if good else 'FAIL'}", flush=True)
    for (name_, M, N, K, epi, shapes) in [("llm_qkv", 576, 12288, 4096, "rope", (4, 6)), ("llm_gu", 576, 22016, 4096, "swiglu", (4, 6)), ("lm_head", 576, 32000, 4096, "store32", (4, 6)),
                                           ("sp_qkv", 288, 12288, 4096, "rope", (4, 6)), ("sp_gu", 288, 22016, 4096, "swiglu", (4, 6))]:
        A, W, kw = problem(M, N, K, epi, td)
        Wf = pack.frag32(W)
        res = []
        for label, wd in [("off", 0)] + [(f"wd {s}", s) for s in shapes] + [("off", 0)] + [(f"wd {s}", s) for s in shapes]:
            hip.set_option("gemm_wd", wd)
            try:
                us = timeit(lambda: hip.gemm(A, W, w_frag=Wf, **kw), a.iters)
                res.append(f"{label} {us:6.1f} us [{hip.lib().stllm_last_kernel().decode()[:34]}]")
            finally:
                hip.set_option("gemm_wd", -1)
        print(f"{name_:8s} M={M} N={N} K={K}: " + " | ".join(res), flush=True)
    print("ALL OK" if ok else "FAILURES")


if __name__ == "__main__":
    main()
