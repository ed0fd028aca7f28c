#!/usr/bin/env python3
"""Micro-benchmark of the GEMM shapes on the hot path (config 2: T=16, S=576).  GPU only.
    python tools/gemm_bench.py [--dtype bf16] [--iters 20] [--only name,...]
Random [-1,1)-ish operands (never zero-filled: guide rule 25)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from stllm_amd import hip, pack  # noqa: E402

SHAPES = [  # name, M, N, K, epilogue, launches per clip
    ("vit_qkv", 4112, 4224, 1408, "store", 39), ("vit_proj", 4112, 1408, 1408, "resid", 39),
    ("vit_fc1", 4112, 6144, 1408, "gelu", 39), ("vit_fc2", 4112, 1408, 6144, "resid", 39),
    ("qf_ckv", 4112, 1536, 1408, "store", 6), ("qf_qkv", 512, 2304, 768, "store", 12),
    ("qf_ffn1", 512, 3072, 768, "gelu", 12), ("qf_ffn2", 512, 768, 3072, "resid", 12),
    ("llm_qkv", 576, 12288, 4096, "rope", 32), ("llm_o", 576, 4096, 4096, "resid", 32),
    ("llm_gu", 576, 22016, 4096, "swiglu", 32), ("llm_down", 576, 4096, 11008, "resid", 32),
    ("lm_head", 576, 32000, 4096, "store32", 1),
    ("qf_out", 512, 768, 768, "resid", 18), ("qf_xq", 512, 768, 768, "store", 6), ("proj4096", 512, 4096, 768, "store32", 1)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default="")
    ap.add_argument("--vendor", action="store_true", help="also time torch's library GEMM (hipBLASLt, then rocBLAS) on the same operands: plain "
                    "A @ W^T -> 16-bit, no epilogue — a calibration of what the vendor kernels reach on these shapes, never a product path")
    ap.add_argument("--rows", default="", help="vit_rows,llm_rows: the same layer shapes at other row counts (c4: 8224,528 and 8224,1088; c3 on one GPU: 65792,2304; per GPU at N = 8: 8224,580)")
    ap.add_argument("--x3", action="store_true", help="the inner bf16 GEMMs of the split verify mode (bf16x3): K' = 3 K, fp32 output (plain store or residual)")
    ap.add_argument("--train", action="store_true", help="the training step's Llama GEMM shapes instead of the inference ones")
    ap.add_argument("--audit", action="store_true", help="time every kernel family / forced tile on every shape next to the automatic choice: does the dispatcher pick the fastest?")
    a = ap.parse_args()
    td = hip.torch_dtype(a.dtype)
    only = set(a.only.split(",")) if a.only else None
    shapes = SHAPES
    if a.train:   # the Llama GEMMs of the training step, 16 clips x 576 tokens = 9216 rows: forward, dgrad = gemm(dY, W^T), wgrad = gemm(dY^T, X^T) with fp32 output
        R = 9216
        shapes = [("f_qkv", R, 12288, 4096, "rope", 32), ("f_o", R, 4096, 4096, "resid", 32), ("f_gu", R, 22016, 4096, "store", 32), ("f_down", R, 4096, 11008, "resid", 32),
                  ("f_lm", R, 32000, 4096, "store32", 1), ("d_qkv", R, 4096, 12288, "store", 32), ("d_o", R, 4096, 4096, "store", 32), ("d_gu", R, 4096, 22016, "store", 32),
                  ("d_down", R, 11008, 4096, "store", 32), ("d_lm", R, 4096, 32000, "store", 1), ("w_qkv", 12288, 4096, R, "store32", 32), ("w_o", 4096, 4096, R, "store32", 32),
                  ("w_gu", 22016, 4096, R, "store32", 32), ("w_down", 4096, 11008, R, "store32", 32), ("w_lm", 32000, 4096, R, "store32", 1)]
    if a.x3:
        shapes = [(n, M, N, 3 * K, "resid" if e == "resid" else "store32", c) for n, M, N, K, e, c in shapes]
    if a.rows:
        vr, lr = (int(x) for x in a.rows.split(","))
        shapes = [(n, vr if n.startswith(("vit_", "qf_ckv")) else lr if n.startswith(("llm_", "lm_head")) else M, N, K, e, c) for n, M, N, K, e, c in SHAPES]
    total_ms = 0.0
    total_fl = 0.0
    vendor_ms = {}
    for name, M, N, K, epi, per_clip in shapes:
        if only and name not in only:
            continue
        A = (torch.rand(M, K, device="cuda") * 2 - 1).to(td)
        W = ((torch.rand(N, K, device="cuda") * 2 - 1) * 0.05).to(td)
        bias = torch.rand(N, device="cuda")
        x = torch.rand(M, N, device="cuda")
        cos, sin = pack.rope_tables(576, device="cuda")
        kw = dict(dtype=td)
        if epi == "store": kw.update(bias=bias)
        elif epi == "store32": kw.update(out_f32=True)
        elif epi == "gelu": kw.update(bias=bias, act=hip.ACT_GELU)
        elif epi == "resid": kw.update(bias=bias, epilogue=hip.EPI_RESID, resid=x)
        elif epi == "swiglu": kw.update(epilogue=hip.EPI_SWIGLU)
        elif epi == "rope": kw.update(epilogue=hip.EPI_ROPE, rope=(cos, sin), rope_seq=576, rope_cols=8192)
        out = None if epi == "resid" else hip.gemm(A, W, **kw)
        if out is not None: kw["out"] = out
        for _ in range(3): hip.gemm(A, W, **kw)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(a.iters): hip.gemm(A, W, **kw)
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / a.iters
        fl = 2.0 * M * N * K
        total_ms += ms * per_clip; total_fl += fl * per_clip
        print(f"{name:9s} M={M:5d} N={N:6d} K={K:6d} {epi:8s} {ms * 1e3:8.1f} us  {fl / ms / 1e9:7.1f} TF   x{per_clip:2d} = {ms * per_clip:6.3f} ms  [{hip.lib().stllm_last_kernel().decode()}]")
        if a.audit:
            auto_us, auto_k = ms * 1e3, hip.lib().stllm_last_kernel().decode()
            rows = []
            for label, opts in [("128/64 kernels", dict(gemm_p8=0, gemm_w4=0, gemm_sk=0)), ("phased auto", dict(gemm_p8=1, gemm_w4=0)), ("phased 192", dict(gemm_p8=3, gemm_w4=0)),
                                ("phased 256", dict(gemm_p8=4, gemm_w4=0))] + [(f"w4 {t}", dict(gemm_w4=t)) for t in (32, 42, 34, 24, 33, 43)] + [("automatic, again", dict())]:
                for k_, v_ in opts.items(): hip.set_option(k_, v_)
                try:
                    if out is not None: kw["out"] = out
                    for _ in range(3): hip.gemm(A, W, **kw)
                    kn = hip.lib().stllm_last_kernel().decode()
                    s.record()
                    for _ in range(a.iters): hip.gemm(A, W, **kw)
                    e.record(); torch.cuda.synchronize()
                    rows.append((s.elapsed_time(e) / a.iters * 1e3, label, kn))
                except Exception as ex:   # noqa: BLE001
                    rows.append((float("inf"), label, f"{type(ex).__name__}"))
                finally:
                    for k_ in ("gemm_p8", "gemm_w4", "gemm_sk"): hip.set_option(k_, -1)
            rows.sort()
            best = rows[0]
            flag = "" if best[0] > 0.97 * auto_us else f"   <-- {auto_us - best[0]:.1f} us ({(auto_us - best[0]) * per_clip / 1e3:.3f} ms per clip) faster than the automatic choice"
            seen = set()
            for us, label, kn in rows:
                if kn in seen: continue
                seen.add(kn)
                print(f"      {label:15s} {us:8.1f} us  [{kn}]" + (flag if (us, label, kn) == best else ""))
            assert hip.gemm_workspace_ok()
        if a.vendor:
            Wv = W[:, :K].contiguous()
            for libname in ("hipblaslt", "cublas"):     # torch's name for rocBLAS on ROCm
                try:
                    torch.backends.cuda.preferred_blas_library(libname)
                    o = torch.empty(M, N, device="cuda", dtype=td)
                    for _ in range(3): torch.matmul(A, Wv.t(), out=o)
                    s.record()
                    for _ in range(a.iters): torch.matmul(A, Wv.t(), out=o)
                    e.record(); torch.cuda.synchronize()
                    vms = s.elapsed_time(e) / a.iters
                    vendor_ms[libname] = vendor_ms.get(libname, 0.0) + vms * per_clip
                    print(f"    {'rocblas' if libname == 'cublas' else libname:9s} plain store {vms * 1e3:8.1f} us  {fl / vms / 1e9:7.1f} TF   ours / vendor time = {ms / vms:5.2f}")
                except Exception as ex:   # noqa: BLE001
                    print(f"    {libname}: {type(ex).__name__}: {ex}")
    print(f"total GEMM time per clip: {total_ms:.3f} ms  ({total_fl / total_ms / 1e9:.1f} TF/s average)")
    for libname, v in vendor_ms.items():
        print(f"   {'rocblas' if libname == 'cublas' else libname} (plain store, no epilogue): {v:.3f} ms  ({total_fl / v / 1e9:.1f} TF/s average)")


if __name__ == "__main__":
    main()
