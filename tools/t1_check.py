#!/usr/bin/env python3
"""GPU: the tall-tile one-round GEMM (gemm_t1.inc) against a float64 reference + timing next to the automatic choice.
    python tools/t1_check.py [--iters 30]"""
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
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--time-only", action="store_true")
    a = ap.parse_args()
    td = hip.torch_dtype(a.dtype)
    torch.manual_seed(0)
    ok = True
    for (M, N, K, epi, shape) in [(576, 4096, 4096, "resid", 2), (576, 4096, 11008, "resid", 2), (300, 512, 1408, "resid", 2), (577, 256, 768, "store32", 2),
                                  (144, 128, 704, "store", 2), (290, 1024, 4096, "store", 2)] * (0 if a.time_only else 1):
        A = (torch.rand(M, K, device="cuda") * 2 - 1).to(td)
        W = ((torch.rand(N, K, device="cuda") * 2 - 1) * 0.05).to(td)
        bias = torch.rand(N, device="cuda")
        x = torch.rand(M, N, device="cuda")
        ref = A.double() @ W.double().t() + bias.double()
        kw = dict(dtype=td, bias=bias)
        if epi == "resid": kw.update(epilogue=hip.EPI_RESID, resid=x.clone()); ref = ref + x.double()
        elif epi == "store32": kw.update(out_f32=True)
        hip.set_option("gemm_t1", shape)
        try:
            out = hip.gemm(A, W, **kw)
            if epi == "resid": out = kw["resid"]
            name = hip.lib().stllm_last_kernel().decode()
        finally:
            hip.set_option("gemm_t1", -1)
        err = (out.double() - ref).abs().max().item()
        scale = ref.abs().max().item()
        tol = (1e-4 if epi != "store" else 2.0 ** -8) * scale
        good = err <= tol and name.startswith("gemm_t1")
        ok &= good
        print(f"M={M} N={N} K={K} {epi:8s} shape {shape}: max err {err:.3e} (tol {tol:.3e}) [{name}] {'ok' if good else 'FAIL'}")
    for (name_, M, N, K, epi, shape) in [("llm_o", 576, 4096, 4096, "resid", 2), ("llm_down", 576, 4096, 11008, "resid", 2), 
                                          ("sp_o", 288, 4096, 4096, "resid", 2), ("sp_down", 288, 4096, 11008, "resid", 2)]:
        A = (torch.rand(M, K, device="cuda") * 2 - 1).to(td)
        W = ((torch.rand(N, K, device="cuda") * 2 - 1) * 0.05).to(td)
        bias = torch.rand(N, device="cuda")
        x = torch.rand(M, N, device="cuda")
        kw = dict(dtype=td, bias=bias, epilogue=hip.EPI_RESID, resid=x)
        res = []
        for label, t1 in (("auto/off", 0), (f"t1 {shape}", shape), ("auto/off", 0), (f"t1 {shape}", shape)):
            hip.set_option("gemm_t1", t1)
            try:
                us = timeit(lambda: hip.gemm(A, W, **kw), a.iters)
                res.append(f"{label} {us:6.1f} us [{hip.lib().stllm_last_kernel().decode()}]")
            finally:
                hip.set_option("gemm_t1", -1)
        print(f"{name_:8s} M={M} N={N} K={K}: " + " | ".join(res))
    # channel camping?  the same GEMMs with the row stride of A and W padded by 64 elements (128 bytes): K = 4096 bf16 rows are 8 KiB apart,
    # every row's k-th chunk lands on the same L2 / memory channel
    for (name_, M, N, K) in [("llm_o", 576, 4096, 4096), ("llm_down", 576, 4096, 11008), ("llm_qkv(store)", 576, 12288, 4096)]:
        for pad in (0, 64, 0, 64):
            Ab = (torch.rand(M, K + pad, device="cuda") * 2 - 1).to(td)
            Wb = ((torch.rand(N, K + pad, device="cuda") * 2 - 1) * 0.05).to(td)
            A, W = Ab[:, :K], Wb[:, :K]
            bias = torch.rand(N, device="cuda")
            x = torch.rand(M, N, device="cuda")
            kw = dict(dtype=td, bias=bias, epilogue=hip.EPI_RESID, resid=x) if N == 4096 else dict(dtype=td, bias=bias)
            res = []
            for label, t1 in (("other", 0), ("t1", 2)):
                hip.set_option("gemm_t1", t1)
                try:
                    us = timeit(lambda: hip.gemm(A, W, **kw), a.iters)
                    res.append(f"{label} {us:6.1f} us [{hip.lib().stllm_last_kernel().decode()[:28]}]")
                finally:
                    hip.set_option("gemm_t1", -1)
            print(f"{name_:8s} row pad {pad:3d}: " + " | ".join(res))
    print("ALL OK" if ok else "FAILURES")


if __name__ == "__main__":
    main()
