#!/usr/bin/env python3
"""GPU: the one-wave GEMM reading W from its LDS-image copy (pack.lds_image, stllm_gemm_args.w_lds) — bit-identity with the row-major path + timing.
    python tools/wlds_check.py [--iters 30]"""
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=30)
    a = ap.parse_args()
    td = torch.bfloat16
    torch.manual_seed(0)
    ok = True
    for (name, M, N, K, epi, bn) in [("vit_qkv", 4112, 4224, 1408, "store", 192), ("vit_proj", 4112, 1408, 1408, "resid", 128), ("vit_fc1", 4112, 6144, 1408, "gelu", 192),
                                     ("vit_fc2", 4112, 1408, 6144, "resid", 128), ("small", 300, 384, 256, "store", 128), ("qf_ckv", 4112, 1536, 1408, "store", 192)]:
        A = (torch.rand(M, K, device="cuda") * 2 - 1).to(td)
        W = ((torch.rand(N, K, device="cuda") * 2 - 1) * 0.05).to(td)
        bias = torch.rand(N, device="cuda")
        x0 = torch.rand(M, N, device="cuda")
        img = pack.lds_image(W, bn)

        def run(use):
            kw = dict(dtype=td, bias=bias)
            if use: kw["w_lds"] = (img, bn)
            if epi == "resid":
                x = x0.clone()
                hip.gemm(A, W, epilogue=hip.EPI_RESID, resid=x, **kw)
                return x
            if epi == "gelu": return hip.gemm(A, W, act=hip.ACT_GELU, **kw)
            return hip.gemm(A, W, **kw)
        r0 = run(False); k0 = hip.lib().stllm_last_kernel().decode()
        r1 = run(True); k1 = hip.lib().stllm_last_kernel().decode()
        same = torch.equal(r0, r1)
        ok &= same
        xr = x0.clone()
        kwt = dict(dtype=td, bias=bias)
        res = []
        for use in (False, True, False, True):
            kw = dict(kwt, **({"w_lds": (img, bn)} if use else {}))
            if epi == "resid": fn = lambda: hip.gemm(A, W, epilogue=hip.EPI_RESID, resid=xr, **kw)
            elif epi == "gelu": fn = lambda: hip.gemm(A, W, act=hip.ACT_GELU, **kw)
            else: fn = lambda: hip.gemm(A, W, **kw)
            res.append(f"{'image' if use else 'rows '} {timeit(fn, a.iters):6.1f} us")
        print(f"{name:9s} M={M} N={N} K={K} {epi:6s} bn {bn}: bit-identical {same} [{k0}] | " + " | ".join(res), flush=True)
    print("ALL OK" if ok else "FAILURES")


if __name__ == "__main__":
    main()
