"""GPU: every HIP kernel behind the C ABI against a plain fp32/fp64 torch reference of the same op.

Inputs are rounded to the compute dtype first, so the only differences are fp32 accumulation order
and the final rounding of the output dtype:
    out_f32 epilogues:  |err| <= 1e-4 * max|ref|     (catches any MFMA-layout / swizzle / tiling bug)
    bf16 outputs:       |err| <= 2^-8  * max|ref|     (one bf16 rounding, 2^-9 relative, + slack)
    fp16 outputs:       |err| <= 2^-10 * max|ref|
    fp32 ("verify") dtype: |err| <= 2e-5 * max|ref|
Data are asymmetric random (guide rule 16: transpose-detecting)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import stllm_oracle as O
from _util import T

pytestmark = pytest.mark.gpu
DT = ["bf16", "fp16", "fp32"]
OUT_TOL = {"bf16": 2.0 ** -8, "fp16": 2.0 ** -10, "fp32": 2e-5}
ACC_TOL = {"bf16": 1e-4, "fp16": 1e-4, "fp32": 2e-5}


@pytest.fixture(scope="module")
def hip():
    from stllm_amd import hip as h
    h.lib()
    return h


def dev(t, dtype=None):
    t = t.cuda()
    return t.to(dtype) if dtype is not None else t


def rnd(name, shape, dtype, std=1.0):
    """random tensor rounded to compute dtype; returns (device tensor in dtype, fp64 CPU copy of the rounded values)"""
    from stllm_amd.hip import torch_dtype
    x = T(name, shape, std).to(torch_dtype(dtype))
    return x.cuda(), x.double()


def check(got, ref, tol, what):
    got = got.detach().double().cpu()
    ref = ref.double()
    assert got.shape == ref.shape, f"{what}: {got.shape} vs {ref.shape}"
    assert torch.isfinite(got).all(), f"{what}: non-finite output"
    scale = max(ref.abs().max().item(), 1e-6)
    err = (got - ref).abs().max().item()
    assert err <= tol * scale, f"{what}: max err {err:.3e} > {tol:.1e} * {scale:.3e}"


GEMM_SHAPES = [  # (M, N, K): ragged M, every tile path (64x64 small-problem path and 128x128)
    (307, 384, 1408), (64, 128, 768), (33, 256, 128), (1100, 2304, 768), (4112 // 4, 4224, 1408)]


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
def test_gemm_store(hip, dtype, M, N, K):
    a, a64 = rnd("a", (M, K), dtype)
    w, w64 = rnd("w", (N, K), dtype, 0.05)
    b = T("b", (N,), 0.5)
    ref = a64 @ w64.t() + b.double()
    out32 = hip.gemm(a, w, dtype=dtype, bias=b.cuda(), out_f32=True)
    check(out32, ref, ACC_TOL[dtype], "gemm f32 out")
    out = hip.gemm(a, w, dtype=dtype, bias=b.cuda())
    check(out, ref, OUT_TOL[dtype], "gemm T out")
    nob = hip.gemm(a, w, dtype=dtype, out_f32=True)
    check(nob, a64 @ w64.t(), ACC_TOL[dtype], "gemm no bias")


@pytest.mark.parametrize("dtype", DT)
def test_gemm_identity_asymmetric(hip, dtype):
    """A = I picks out W^T exactly: any row/col swap or lane mis-map shows up bit-exactly."""
    K = N = 256
    from stllm_amd.hip import torch_dtype
    a = torch.eye(K, dtype=torch_dtype(dtype)).cuda()
    w, w64 = rnd("w_asym", (N, K), dtype)
    out = hip.gemm(a, w, dtype=dtype, out_f32=True)
    assert torch.equal(out.cpu().double(), w64.t())


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("act", ["gelu", "relu"])
def test_gemm_act(hip, dtype, act):
    M, N, K = 300, 6144 // 4, 1408
    a, a64 = rnd("a", (M, K), dtype)
    w, w64 = rnd("w", (N, K), dtype, 0.03)
    b = T("b", (N,), 0.2)
    pre = a64 @ w64.t() + b.double()
    ref = O.gelu(pre) if act == "gelu" else torch.relu(pre)
    out = hip.gemm(a, w, dtype=dtype, bias=b.cuda(), act=hip.ACT_GELU if act == "gelu" else hip.ACT_RELU, out_f32=True)
    check(out, ref, max(ACC_TOL[dtype], 3e-6), f"gemm {act}")  # erf by A&S 7.1.26, 1.5e-7 abs


@pytest.mark.parametrize("dtype", DT)
def test_gemm_resid(hip, dtype):
    M, N, K = 515, 1408, 6144 // 2
    a, a64 = rnd("a", (M, K), dtype)
    w, w64 = rnd("w", (N, K), dtype, 0.02)
    b = T("b", (N,), 0.2)
    x = T("x", (M, N), 2.0)
    ref = x.double() + a64 @ w64.t() + b.double()
    xd = x.cuda()
    out = torch.empty_like(xd)
    hip.gemm(a, w, dtype=dtype, epilogue=hip.EPI_RESID, bias=b.cuda(), resid=xd, out=out)
    check(out, ref, ACC_TOL[dtype], "resid out-of-place")
    hip.gemm(a, w, dtype=dtype, epilogue=hip.EPI_RESID, bias=b.cuda(), resid=xd)  # in place
    check(xd, ref, ACC_TOL[dtype], "resid in-place")


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("M", [50, 333])
def test_gemm_swiglu(hip, dtype, M):
    from stllm_amd import pack
    K, I = 512, 1024 + 128 * 3
    a, a64 = rnd("a", (M, K), dtype)
    wg, wg64 = rnd("wg", (I, K), dtype, 0.05)
    wu, wu64 = rnd("wu", (I, K), dtype, 0.05)
    ref = F.silu(a64 @ wg64.t()) * (a64 @ wu64.t())
    out = hip.gemm(a, pack.llama_gate_up(wg, wu, dtype), dtype=dtype, epilogue=hip.EPI_SWIGLU)
    check(out, ref, OUT_TOL[dtype], "swiglu")


@pytest.mark.parametrize("dtype", DT)
def test_gemm_rope_qkv(hip, dtype):
    """fused QKV + rotate-half RoPE in the packed head layout: q.k^T per head must equal the
    reference's (the packing permutes q and k identically), v must be untouched."""
    from stllm_amd import pack
    B, S, H, D = 2, 37, 4, 128
    K = 512
    a, a64 = rnd("a", (B * S, K), dtype)
    wq, wq64 = rnd("wq", (H * D, K), dtype, 0.05)
    wk, wk64 = rnd("wk", (H * D, K), dtype, 0.05)
    wv, wv64 = rnd("wv", (H * D, K), dtype, 0.05)
    cos, sin = pack.rope_tables(S)
    qkv = hip.gemm(a, pack.llama_qkv(wq, wk, wv, dtype, n_heads=H), dtype=dtype, epilogue=hip.EPI_ROPE,
                   rope=(cos.cuda(), sin.cuda()), rope_seq=S, rope_cols=2 * H * D)
    qkv = qkv.double().cpu().view(B, S, 3, H, D)
    c, s = O.rope_tables(S, D)
    q = (a64 @ wq64.t()).view(B, S, H, D).transpose(1, 2)
    k = (a64 @ wk64.t()).view(B, S, H, D).transpose(1, 2)
    v = (a64 @ wv64.t()).view(B, S, H, D)
    q = q * c.double() + O._rotate_half(q) * s.double()
    k = k * c.double() + O._rotate_half(k) * s.double()
    tol = OUT_TOL[dtype]
    check(qkv[:, :, 2], v, tol, "v passthrough")
    perm = pack.rope_head_perm(1)
    check(qkv[:, :, 0].transpose(1, 2), q[..., perm], tol, "q rope (packed order)")
    check(qkv[:, :, 1].transpose(1, 2), k[..., perm], tol, "k rope (packed order)")
    got_s = qkv[:, :, 0].transpose(1, 2) @ qkv[:, :, 1].transpose(1, 2).transpose(-1, -2)
    check(got_s, q @ k.transpose(-1, -2), 4 * tol, "q.k^T invariance")


@pytest.mark.parametrize("dtype", DT)
def test_patch_embed(hip, dtype):
    from stllm_amd import pack
    from stllm_amd.hip import torch_dtype
    n = 3
    frames = T("input.frames", (n, 3, 224, 224))
    w = T("pw", (1408, 3, 14, 14), 0.02)
    b = T("pb", (1408,), 0.1)
    pos = T("pos", (1, 257, 1408), 0.1)
    cls = T("cls", (1, 1, 1408), 0.1)
    td = torch_dtype(dtype)
    sd = {"patch_embed.proj.weight": w.to(td).float(), "patch_embed.proj.bias": b, "cls_token": cls, "pos_embed": pos}
    ref = O.vit_embed(frames.to(td).float(), sd, "")
    x = torch.empty((n * 257, 1408), device="cuda")
    hip.gemm(None, pack.patch_weight(w.cuda(), dtype), dtype=dtype, epilogue=hip.EPI_PATCH, bias=b.cuda(), out=x,
             frames=frames.cuda(), pos_embed=pos.cuda().view(257, 1408), n_frames=n)
    hip.vit_cls_rows(cls.cuda().view(-1), pos.cuda().view(257, 1408), x, n)
    check(x.view(n, 257, 1408), ref, ACC_TOL[dtype], "patch embed + cls + pos")


@pytest.mark.parametrize("dtype", DT)
def test_gemm_two_level_rows(hip, dtype):
    """Q-Former row groups: A rows read from / out rows written into a [N, S, C] buffer."""
    N_, S, Q, C, Nout = 5, 44, 32, 768, 256
    buf, buf64 = rnd("buf", (N_ * S, C), dtype)
    w, w64 = rnd("w", (Nout, C), dtype, 0.05)
    ref_q = buf64.view(N_, S, C)[:, :Q].reshape(-1, C) @ w64.t()
    ref_t = buf64.view(N_, S, C)[:, Q:].reshape(-1, C) @ w64.t()
    out = torch.zeros((N_ * S, Nout), device="cuda", dtype=torch.float32)
    hip.gemm(buf, w, dtype=dtype, out=out, out_f32=True, M=N_ * Q, a_rows=(Q, S * C), o_rows=(Q, S * Nout))
    hip.gemm(buf[Q:], w, dtype=dtype, out=out[Q:], out_f32=True, M=N_ * (S - Q), a_rows=(S - Q, S * C),
             o_rows=(S - Q, S * Nout))
    o = out.view(N_, S, Nout)
    check(o[:, :Q].reshape(-1, Nout), ref_q, ACC_TOL[dtype], "query rows")
    check(o[:, Q:].reshape(-1, Nout), ref_t, ACC_TOL[dtype], "text rows")


SK_SHAPES = [(4112 // 2, 4224, 1408), (576, 4096, 4096), (576, 1536, 11008 // 2), (300, 768, 3072), (97, 256, 6144), (1, 128, 128 * 7)]


@pytest.mark.parametrize("bm", [1, 2, 3])
@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("M,N,K", SK_SHAPES)
def test_gemm_stream_k(hip, dtype, bm, M, N, K):
    """stream-K kernel forced on (both tile heights): split tiles, partial slabs, flags, N/M tails, every epilogue."""
    hip.set_option("gemm_sk", bm)
    try:
        a, a64 = rnd("a", (M, K), dtype, 0.5)
        w, w64 = rnd("w", (N, K), dtype, 0.05)
        b = T("b", (N,), 0.5)
        ref = a64 @ w64.t() + b.double()
        out = hip.gemm(a, w, dtype=dtype, bias=b.cuda(), out_f32=True)
        assert hip.lib().stllm_last_kernel().decode().startswith("gemm_sk_kernel<")
        check(out, ref, ACC_TOL[dtype], "sk store f32")
        check(hip.gemm(a, w, dtype=dtype, bias=b.cuda(), act=hip.ACT_GELU), O.gelu(ref), OUT_TOL[dtype], "sk gelu T")
        x = T("x", (M, N), 2.0)
        xd = x.cuda()
        for rep in range(2):  # run twice: flags from the previous launch (older epoch) must not satisfy the next one
            hip.gemm(a, w, dtype=dtype, epilogue=hip.EPI_RESID, bias=b.cuda(), resid=xd)
        check(xd, x.double() + 2 * ref, ACC_TOL[dtype], "sk resid x2")
        # determinism: bit-identical across launches
        o1 = hip.gemm(a, w, dtype=dtype, bias=b.cuda(), out_f32=True)
        assert torch.equal(o1, out)
    finally:
        hip.set_option("gemm_sk", -1)


@pytest.mark.parametrize("bm", [1, 2, 3])
@pytest.mark.parametrize("dtype", ["bf16", "fp32"])
def test_gemm_stream_k_swiglu_rope(hip, dtype, bm):
    from stllm_amd import pack
    hip.set_option("gemm_sk", bm)
    try:
        M, K, I = 333, 1024, 1024 + 128 * 3
        a, a64 = rnd("a", (M, K), dtype)
        wg, wg64 = rnd("wg", (I, K), dtype, 0.05)
        wu, wu64 = rnd("wu", (I, K), dtype, 0.05)
        out = hip.gemm(a, pack.llama_gate_up(wg, wu, dtype), dtype=dtype, epilogue=hip.EPI_SWIGLU)
        assert "gemm_sk_kernel" in hip.lib().stllm_last_kernel().decode()
        check(out, F.silu(a64 @ wg64.t()) * (a64 @ wu64.t()), OUT_TOL[dtype], "sk swiglu")
        B, S, H, D = 2, 150, 4, 128
        a, a64 = rnd("a2", (B * S, K), dtype)
        wq, wq64 = rnd("wq", (H * D, K), dtype, 0.05)
        wk, wk64 = rnd("wk", (H * D, K), dtype, 0.05)
        wv, wv64 = rnd("wv", (H * D, K), dtype, 0.05)
        cos, sin = pack.rope_tables(S)
        qkv = hip.gemm(a, pack.llama_qkv(wq, wk, wv, dtype, n_heads=H), dtype=dtype, epilogue=hip.EPI_ROPE,
                       rope=(cos.cuda(), sin.cuda()), rope_seq=S, rope_cols=2 * H * D).double().cpu().view(B, S, 3, H, D)
        c, s = O.rope_tables(S, D)
        q = (a64 @ wq64.t()).view(B, S, H, D).transpose(1, 2)
        k = (a64 @ wk64.t()).view(B, S, H, D).transpose(1, 2)
        q = q * c.double() + O._rotate_half(q) * s.double()
        k = k * c.double() + O._rotate_half(k) * s.double()
        perm = pack.rope_head_perm(1)
        check(qkv[:, :, 0].transpose(1, 2), q[..., perm], OUT_TOL[dtype], "sk q rope")
        check(qkv[:, :, 1].transpose(1, 2), k[..., perm], OUT_TOL[dtype], "sk k rope")
        check(qkv[:, :, 2], (a64 @ wv64.t()).view(B, S, H, D), OUT_TOL[dtype], "sk v")
    finally:
        hip.set_option("gemm_sk", -1)


P8_SHAPES = [(4112 // 2, 4224, 1408), (576, 4096, 4096), (576, 1536, 11008 // 2), (300, 768, 3072), (97, 256, 6144), (1, 128, 128 * 7),
             (2100, 2944, 256)]


@pytest.mark.parametrize("miw", [3, 4])
@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("M,N,K", P8_SHAPES)
def test_gemm_phased(hip, dtype, miw, M, N, K):
    """phased kernel forced on (192- and 256-row tiles): data-parallel rounds, K-split remainder groups with the
    reduce-scatter exchange, M / N tails, fp32 / GELU / residual epilogues, epoch flags, determinism."""
    hip.set_option("gemm_p8", miw)
    try:
        a, a64 = rnd("a", (M, K), dtype, 0.5)
        a2, a264 = rnd("a_other", (M, K), dtype, 0.5)
        w, w64 = rnd("w", (N, K), dtype, 0.05)
        b = T("b", (N,), 0.5)
        ref = a64 @ w64.t() + b.double()
        ref2 = a264 @ w64.t() + b.double()
        out = hip.gemm(a, w, dtype=dtype, bias=b.cuda(), out_f32=True)
        assert hip.lib().stllm_last_kernel().decode().startswith(f"gemm_p8_kernel<{'bf16_t' if dtype == 'bf16' else 'f16_t'},{miw},")
        check(out, ref, ACC_TOL[dtype], "p8 store f32")
        # alternate operands: a stale partial slab from the previous launch would show up here
        for rep in range(3):
            check(hip.gemm(a2, w, dtype=dtype, bias=b.cuda(), out_f32=True), ref2, ACC_TOL[dtype], f"p8 store f32 (other operand, rep {rep})")
            o1 = hip.gemm(a, w, dtype=dtype, bias=b.cuda(), out_f32=True)
            assert torch.equal(o1, out), "p8: not bit-identical across launches"
        check(hip.gemm(a, w, dtype=dtype, bias=b.cuda(), act=hip.ACT_GELU), O.gelu(ref), OUT_TOL[dtype], "p8 gelu T")
        check(hip.gemm(a, w, dtype=dtype), a64 @ w64.t(), OUT_TOL[dtype], "p8 store T, no bias")
        x = T("x", (M, N), 2.0)
        xd = x.cuda()
        for rep in range(2):
            hip.gemm(a, w, dtype=dtype, epilogue=hip.EPI_RESID, bias=b.cuda(), resid=xd)
        check(xd, x.double() + 2 * ref, ACC_TOL[dtype], "p8 resid x2")
    finally:
        hip.set_option("gemm_p8", -1)


@pytest.mark.parametrize("miw", [3, 4])
def test_gemm_phased_swiglu_rope_rows(hip, miw):
    from stllm_amd import pack
    dtype = "bf16"
    hip.set_option("gemm_p8", miw)
    try:
        M, K, I = 333, 1024, 1024 + 128 * 3
        a, a64 = rnd("a", (M, K), dtype)
        wg, wg64 = rnd("wg", (I, K), dtype, 0.05)
        wu, wu64 = rnd("wu", (I, K), dtype, 0.05)
        out = hip.gemm(a, pack.llama_gate_up(wg, wu, dtype), dtype=dtype, epilogue=hip.EPI_SWIGLU)
        assert "gemm_p8_kernel" in hip.lib().stllm_last_kernel().decode()
        check(out, F.silu(a64 @ wg64.t()) * (a64 @ wu64.t()), OUT_TOL[dtype], "p8 swiglu")
        B, S, H, D = 2, 150, 4, 128
        a, a64 = rnd("a2", (B * S, K), dtype)
        wq, wq64 = rnd("wq", (H * D, K), dtype, 0.05)
        wk, wk64 = rnd("wk", (H * D, K), dtype, 0.05)
        wv, wv64 = rnd("wv", (H * D, K), dtype, 0.05)
        cos, sin = pack.rope_tables(S)
        qkv = hip.gemm(a, pack.llama_qkv(wq, wk, wv, dtype, n_heads=H), dtype=dtype, epilogue=hip.EPI_ROPE,
                       rope=(cos.cuda(), sin.cuda()), rope_seq=S, rope_cols=2 * H * D).double().cpu().view(B, S, 3, H, D)
        assert "gemm_p8_kernel" in hip.lib().stllm_last_kernel().decode()
        c, s = O.rope_tables(S, D)
        q = (a64 @ wq64.t()).view(B, S, H, D).transpose(1, 2)
        k = (a64 @ wk64.t()).view(B, S, H, D).transpose(1, 2)
        q = q * c.double() + O._rotate_half(q) * s.double()
        k = k * c.double() + O._rotate_half(k) * s.double()
        perm = pack.rope_head_perm(1)
        check(qkv[:, :, 0].transpose(1, 2), q[..., perm], OUT_TOL[dtype], "p8 q rope")
        check(qkv[:, :, 1].transpose(1, 2), k[..., perm], OUT_TOL[dtype], "p8 k rope")
        check(qkv[:, :, 2], (a64 @ wv64.t()).view(B, S, H, D), OUT_TOL[dtype], "p8 v")
        # 2-level row indexing (Q-Former style row groups) through the phased kernel
        N_, S2, Q, C, Nout = 5, 44, 32, 768, 256
        buf, buf64 = rnd("buf", (N_ * S2, C), dtype)
        w, w64 = rnd("w2", (Nout, C), dtype, 0.05)
        outb = torch.zeros((N_ * S2, Nout), device="cuda", dtype=torch.float32)
        hip.gemm(buf, w, dtype=dtype, out=outb, out_f32=True, M=N_ * Q, a_rows=(Q, S2 * C), o_rows=(Q, S2 * Nout))
        assert "gemm_p8_kernel" in hip.lib().stllm_last_kernel().decode()
        check(outb.view(N_, S2, Nout)[:, :Q].reshape(-1, Nout), buf64.view(N_, S2, C)[:, :Q].reshape(-1, C) @ w64.t(), ACC_TOL[dtype], "p8 query rows")
        assert float(outb.view(N_, S2, Nout)[:, Q:].abs().max()) == 0.0
    finally:
        hip.set_option("gemm_p8", -1)


BENCH_SHAPES = [(4112, 6144, 1408, "gelu"), (4112, 1408, 6144, "resid"), (4112, 4224, 1408, "store"), (4112, 1408, 1408, "resid"),
                (576, 12288, 4096, "store"), (576, 4096, 11008, "resid"), (576, 4096, 4096, "resid"), (576, 32000, 4096, "store")]


def test_gemm_profile_modes(hip):
    """stllm_gemm_profile (what bench.py's roofline leg reads): mode 1 records every stllm_gemm launch with its kernel symbol, shape and
    algorithmic FLOPs; mode 2 only launches of the target symbol; mode 3 every 7th of those (the timed region: the event records cost the
    step they measure); a new session starts empty and mode 0 records nothing."""
    dtype = "bf16"
    a, _ = rnd("pa", (512, 256), dtype, 0.5)
    w1, _ = rnd("pw1", (256, 256), dtype, 0.05)
    w2, _ = rnd("pw2", (384, 256), dtype, 0.05)
    prof = hip.GemmProfiler()
    try:
        prof.start_all()
        for _ in range(3):
            hip.gemm(a, w1, dtype=dtype)
            hip.gemm(a, w2, dtype=dtype)
        s = prof.summary()
        assert sum(v["launches"] for v in s.values()) == 6
        shapes = {k: v["launches"] for e in s.values() for k, v in e["shapes"].items()}
        assert shapes == {"512x256x256": 3, "512x384x256": 3}, shapes
        for e in s.values():
            for k, v in e["shapes"].items():
                M, N, K = map(int, k.split("x"))
                assert v["flops"] == pytest.approx(2.0 * M * N * K * v["launches"]) and v["total_ms"] > 0
        # the symbol of the first shape (both may share one symbol: count what the target mode must see)
        sym = next(k for k, e in s.items() if "512x256x256" in e["shapes"])
        per_round = sum(1 for k, e in s.items() if k == sym for _ in e["shapes"])
        prof.start_target(sym)
        for _ in range(14):
            hip.gemm(a, w1, dtype=dtype)
            hip.gemm(a, w2, dtype=dtype)
        t = prof.summary()
        assert set(t) == {sym} and t[sym]["launches"] == 14 * per_round, t
        prof.start_target(sym, sampled=True)
        for _ in range(14):
            hip.gemm(a, w1, dtype=dtype)
            hip.gemm(a, w2, dtype=dtype)
        t = prof.summary()
        n = 14 * per_round
        assert t[sym]["launches"] == (n + hip.GemmProfiler.SAMPLE_EVERY - 1) // hip.GemmProfiler.SAMPLE_EVERY, t
        prof.stop()
        hip.gemm(a, w1, dtype=dtype)
        assert prof.summary() == {}
    finally:
        prof.stop()


@pytest.mark.parametrize("M,N,K,kind", BENCH_SHAPES)
def test_gemm_bench_shapes_auto_dispatch(hip, M, N, K, kind):
    """the exact GEMM shapes bench.py times (T = 16 ViT rows 16 x 257 = 4112, Llama prefill S = 576), through the automatic
    dispatch (whatever kernel the cost model picks: phased q = 2 rounds + K-split remainder for fc1), against fp64"""
    dtype = "bf16"
    a, a64 = rnd("a", (M, K), dtype, 0.5)
    w, w64 = rnd("w", (N, K), dtype, 0.05)
    b = T("b", (N,), 0.5)
    ref = a64 @ w64.t() + b.double()
    if kind == "gelu":
        out = hip.gemm(a, w, dtype=dtype, bias=b.cuda(), act=hip.ACT_GELU)
        name = hip.lib().stllm_last_kernel().decode()
        check(out, O.gelu(ref), OUT_TOL[dtype], f"bench shape gelu [{name}]")
    elif kind == "store":
        out = hip.gemm(a, w, dtype=dtype, bias=b.cuda())
        name = hip.lib().stllm_last_kernel().decode()
        check(out, ref, OUT_TOL[dtype], f"bench shape store [{name}]")
    else:
        x = T("x", (M, N), 2.0)
        xd = x.cuda()
        hip.gemm(a, w, dtype=dtype, epilogue=hip.EPI_RESID, bias=b.cuda(), resid=xd)
        name = hip.lib().stllm_last_kernel().decode()
        check(xd, x.double() + ref, ACC_TOL[dtype], f"bench shape resid [{name}]")
    assert hip.gemm_workspace_ok(), hip.lib().stllm_last_error().decode()


W4_SHAPES = [(4112 // 2, 4224, 1408), (576, 4096, 4096), (300, 768, 3072), (97, 256, 6144), (1, 128, 128 * 7), (2100, 2944, 256), (3072, 2048, 704),
             (528, 768, 1408), (596, 384, 256)]   # the last two: a tail of 16 rows past 256-row tiles / 20 rows past 192-row tiles (thin-tail path)


@pytest.mark.parametrize("shape", [32, 34, 42, 24, 22])
@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("M,N,K", W4_SHAPES)
def test_gemm_w4(hip, dtype, shape, M, N, K):
    """one-wave-per-SIMD kernel forced on (192 x 128, 192 x 256, 256 x 256, 256 x 128, — round 4 — 128 x 256 tiles and — round 5, code 22 — the 128 x 128 tile
    as TWO workgroups per CU, 512 persistent workgroups, exchange-free): whole rounds, remainder-first K-split
    with the end-of-launch reduction, M / N tails (256 x 256 retired in round 3: a forced 44 falls back to the other kernels), (incl. the thin-tail rows computed outside the tile grid), fp32 / GELU / residual
    epilogues, epoch flags, determinism."""
    hip.set_option("gemm_w4", shape)
    try:
        a, a64 = rnd("a", (M, K), dtype, 0.5)
        a2, a264 = rnd("a_other", (M, K), dtype, 0.5)
        w, w64 = rnd("w", (N, K), dtype, 0.05)
        b = T("b", (N,), 0.5)
        ref = a64 @ w64.t() + b.double()
        ref2 = a264 @ w64.t() + b.double()
        out = hip.gemm(a, w, dtype=dtype, bias=b.cuda(), out_f32=True)
        assert hip.lib().stllm_last_kernel().decode().startswith(f"gemm_w4_kernel<{'bf16_t' if dtype == 'bf16' else 'f16_t'},{shape // 10},{shape % 10},")
        check(out, ref, ACC_TOL[dtype], "w4 store f32")
        for rep in range(3):   # alternate operands: a stale partial slab from the previous launch would show up here
            check(hip.gemm(a2, w, dtype=dtype, bias=b.cuda(), out_f32=True), ref2, ACC_TOL[dtype], f"w4 store f32 (other operand, rep {rep})")
            o1 = hip.gemm(a, w, dtype=dtype, bias=b.cuda(), out_f32=True)
            assert torch.equal(o1, out), "w4: not bit-identical across launches"
        check(hip.gemm(a, w, dtype=dtype, bias=b.cuda(), act=hip.ACT_GELU), O.gelu(ref), OUT_TOL[dtype], "w4 gelu T")
        check(hip.gemm(a, w, dtype=dtype), a64 @ w64.t(), OUT_TOL[dtype], "w4 store T, no bias")
        x = T("x", (M, N), 2.0)
        xd = x.cuda()
        for rep in range(2):
            hip.gemm(a, w, dtype=dtype, epilogue=hip.EPI_RESID, bias=b.cuda(), resid=xd)
        check(xd, x.double() + 2 * ref, ACC_TOL[dtype], "w4 resid x2")
        assert hip.gemm_workspace_ok(), hip.lib().stllm_last_error().decode()
    finally:
        hip.set_option("gemm_w4", -1)


T1_SHAPES = [(576, 4096, 4096), (288, 1024, 1408), (300, 512, 2816), (577, 256, 1536), (144, 128, 1408), (17, 128, 1408), (1000, 384, 3072)]


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("M,N,K", T1_SHAPES)
def test_gemm_t1_tall_tile(hip, dtype, M, N, K):
    """round 6: the tall-tile one-round kernel (gemm_t1.inc: 144 x 64 tiles, whole K per workgroup, 4 compute + 4 loader waves, asm MFMAs
    with tied accumulators) forced on: fp32 / 16-bit stores and the residual epilogue, with and without bias, row tails (M % 144), determinism
    (nothing is exchanged between workgroups: bit-identical across launches by construction, asserted all the same)."""
    hip.set_option("gemm_t1", 2)
    try:
        a, a64 = rnd("a", (M, K), dtype, 0.5)
        a2, a264 = rnd("a_other", (M, K), dtype, 0.5)
        w, w64 = rnd("w", (N, K), dtype, 0.05)
        b = T("b", (N,), 0.5)
        ref = a64 @ w64.t() + b.double()
        out = hip.gemm(a, w, dtype=dtype, bias=b.cuda(), out_f32=True)
        assert hip.lib().stllm_last_kernel().decode().startswith(f"gemm_t1_kernel<{'bf16_t' if dtype == 'bf16' else 'f16_t'},2,STORE,0,1>")
        check(out, ref, ACC_TOL[dtype], "t1 store f32")
        for rep in range(2):
            check(hip.gemm(a2, w, dtype=dtype, bias=b.cuda(), out_f32=True), a264 @ w64.t() + b.double(), ACC_TOL[dtype], f"t1 store f32 (other operand, rep {rep})")
            assert torch.equal(hip.gemm(a, w, dtype=dtype, bias=b.cuda(), out_f32=True), out), "t1: not bit-identical across launches"
        check(hip.gemm(a, w, dtype=dtype, bias=b.cuda()), ref, OUT_TOL[dtype], "t1 store T")
        check(hip.gemm(a, w, dtype=dtype), a64 @ w64.t(), OUT_TOL[dtype], "t1 store T, no bias")
        x = T("x", (M, N), 2.0)
        xd = x.cuda()
        for rep in range(2):
            hip.gemm(a, w, dtype=dtype, epilogue=hip.EPI_RESID, bias=b.cuda(), resid=xd)
            assert hip.lib().stllm_last_kernel().decode().startswith("gemm_t1_kernel<")
        check(xd, x.double() + 2 * ref, ACC_TOL[dtype], "t1 resid x2 (in place)")
        o2 = torch.empty_like(xd)
        hip.gemm(a, w, dtype=dtype, epilogue=hip.EPI_RESID, resid=xd, out=o2)
        check(o2, x.double() + 2 * ref + a64 @ w64.t(), ACC_TOL[dtype], "t1 resid out of place, no bias")
        # shapes / epilogues the kernel does not have fall through to the other kernels and stay correct
        check(hip.gemm(a, w, dtype=dtype, bias=b.cuda(), act=hip.ACT_GELU), O.gelu(ref), OUT_TOL[dtype], "forced t1, GELU -> fallback")
        assert not hip.lib().stllm_last_kernel().decode().startswith("gemm_t1")
    finally:
        hip.set_option("gemm_t1", -1)


def test_gemm_t1_two_level_rows_and_fallback_k(hip):
    """the tall-tile kernel with 2-level A rows (frames inside a larger buffer) and 2-level output rows; K % 128 != 0 or fewer than 11 K units
    are refused (the dispatcher falls back)."""
    hip.set_option("gemm_t1", 2)
    try:
        dtype = "bf16"
        nb, rpb, K, N = 5, 61, 1408, 256
        buf, buf64 = rnd("a_buf", (nb, rpb + 3, K), dtype, 0.5)   # batches of rpb rows, 3 unused rows behind each
        w, w64 = rnd("w", (N, K), dtype, 0.05)
        M = nb * rpb
        ref = (buf64[:, :rpb].reshape(M, K) @ w64.t())   # (the kernel addresses A by (rows per batch, batch stride) inside the larger buffer)
        out = torch.zeros(nb, rpb + 2, N, device="cuda")    # 2-level output rows too: 2 unused rows behind every batch stay zero
        hip.gemm(buf, w, dtype=dtype, out=out, out_f32=True, M=M, a_rows=(rpb, (rpb + 3) * K), o_rows=(rpb, (rpb + 2) * N))
        assert hip.lib().stllm_last_kernel().decode().startswith("gemm_t1_kernel<")
        check(out[:, :rpb].reshape(M, N), ref, ACC_TOL[dtype], "t1 two-level A / output rows")
        assert float(out[:, rpb:].abs().max()) == 0.0
        for Kbad in (704, 1344):   # 11 units (odd), 21 units
            a, a64 = rnd("a", (200, Kbad), dtype, 0.5)
            w2, w264 = rnd("w2", (N, Kbad), dtype, 0.05)
            check(hip.gemm(a, w2, dtype=dtype, out_f32=True), a64 @ w264.t(), ACC_TOL[dtype], f"K = {Kbad}: fallback")
            assert not hip.lib().stllm_last_kernel().decode().startswith("gemm_t1")
    finally:
        hip.set_option("gemm_t1", -1)


def test_llama_o_proj_runs_on_the_tall_tile_kernel(hip):
    """automatic dispatch (round 6): 576 x 4096 x 4096 with the residual epilogue = 4 x 64 tiles of 144 x 64 on gemm_t1 (38 vs 44 us inside the model);
    the down projection (K = 11008) stays on the phased kernel (88 vs 71 us inside the model, profiles/r06_bench_ab_t1.log)."""
    dtype = "bf16"
    for K, want_t1 in ((4096, True), (11008, False)):
        a, a64 = rnd("a", (576, K), dtype, 0.5)
        w, w64 = rnd("w", (4096, K), dtype, 0.02)
        x = T("x", (576, 4096), 2.0)
        xd = x.cuda()
        hip.gemm(a, w, dtype=dtype, epilogue=hip.EPI_RESID, resid=xd)
        name = hip.lib().stllm_last_kernel().decode()
        assert name.startswith("gemm_t1_kernel<bf16_t,2,RESID") == want_t1, name
        check(xd, x.double() + a64 @ w64.t(), ACC_TOL[dtype], f"o / down at K = {K} [{name}]")
    assert hip.gemm_workspace_ok()


WD_SHAPES = [(576, 1024, 1024), (130, 256, 256), (300, 512, 4096), (17, 256, 512), (700, 768, 1536)]


@pytest.mark.parametrize("wm", [4, 6])
@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("M,N,K", WD_SHAPES)
def test_gemm_wd_w_direct(hip, dtype, wm, M, N, K):
    """round 6: the W-direct kernel (gemm_wd.inc: (32 wm) x 256 tiles, W fragments straight into registers from the fragment-major copy
    pack.frag32, A through the LDS ring) forced on: fp32 / 16-bit stores with and without bias, row tails, determinism; without w_frag, or
    with an epilogue it does not have, the dispatcher falls back."""
    from stllm_amd import pack
    hip.set_option("gemm_wd", wm)
    try:
        a, a64 = rnd("a", (M, K), dtype, 0.5)
        a2, a264 = rnd("a_other", (M, K), dtype, 0.5)
        w, w64 = rnd("w", (N, K), dtype, 0.05)
        wf = pack.frag32(w)
        b = T("b", (N,), 0.5)
        ref = a64 @ w64.t() + b.double()
        out = hip.gemm(a, w, dtype=dtype, bias=b.cuda(), out_f32=True, w_frag=wf)
        assert hip.lib().stllm_last_kernel().decode().startswith(f"gemm_wd_kernel<{'bf16_t' if dtype == 'bf16' else 'f16_t'},{wm},STORE,0,1>")
        check(out, ref, ACC_TOL[dtype], "wd store f32")
        for rep in range(2):
            check(hip.gemm(a2, w, dtype=dtype, bias=b.cuda(), out_f32=True, w_frag=wf), a264 @ w64.t() + b.double(), ACC_TOL[dtype], f"wd store f32 (other operand, rep {rep})")
            assert torch.equal(hip.gemm(a, w, dtype=dtype, bias=b.cuda(), out_f32=True, w_frag=wf), out), "wd: not bit-identical across launches"
        check(hip.gemm(a, w, dtype=dtype, bias=b.cuda(), w_frag=wf), ref, OUT_TOL[dtype], "wd store T")
        assert hip.lib().stllm_last_kernel().decode().startswith("gemm_wd_kernel<")
        check(hip.gemm(a, w, dtype=dtype, w_frag=wf), a64 @ w64.t(), OUT_TOL[dtype], "wd store T, no bias")
        # no fragment copy / an epilogue the kernel does not have: the other kernels, same numbers
        check(hip.gemm(a, w, dtype=dtype, bias=b.cuda(), out_f32=True), ref, ACC_TOL[dtype], "forced wd without w_frag -> fallback")
        assert not hip.lib().stllm_last_kernel().decode().startswith("gemm_wd")
        check(hip.gemm(a, w, dtype=dtype, bias=b.cuda(), act=hip.ACT_GELU, w_frag=wf), O.gelu(ref), OUT_TOL[dtype], "forced wd, GELU -> fallback")
        assert not hip.lib().stllm_last_kernel().decode().startswith("gemm_wd")
    finally:
        hip.set_option("gemm_wd", -1)


@pytest.mark.parametrize("wm", [4, 6])
def test_gemm_wd_swiglu_rope_rows(hip, wm):
    """the W-direct kernel's SwiGLU and RoPE epilogues against float64 (the packed layouts of pack.llama_gate_up / pack.llama_qkv), 2-level A and output rows"""
    from stllm_amd import pack
    dtype = "bf16"
    hip.set_option("gemm_wd", wm)
    try:
        M, K, I = 333, 1024, 1024 + 128 * 2
        a, a64 = rnd("a", (M, K), dtype)
        wg, wg64 = rnd("wg", (I, K), dtype, 0.05)
        wu, wu64 = rnd("wu", (I, K), dtype, 0.05)
        wgu = pack.llama_gate_up(wg, wu, dtype)
        out = hip.gemm(a, wgu, dtype=dtype, epilogue=hip.EPI_SWIGLU, w_frag=pack.frag32(wgu))
        assert hip.lib().stllm_last_kernel().decode().startswith(f"gemm_wd_kernel<bf16_t,{wm},SWIGLU")
        check(out, F.silu(a64 @ wg64.t()) * (a64 @ wu64.t()), OUT_TOL[dtype], "wd swiglu")
        B, S, H, D = 2, 150, 4, 128
        a, a64 = rnd("a2", (B * S, K), dtype)
        wq, wq64 = rnd("wq", (H * D, K), dtype, 0.05)
        wk, wk64 = rnd("wk", (H * D, K), dtype, 0.05)
        wv, wv64 = rnd("wv", (H * D, K), dtype, 0.05)
        cos, sin = pack.rope_tables(S)
        wqkv = pack.llama_qkv(wq, wk, wv, dtype, n_heads=H)
        cache = torch.zeros((B, S + 5, 3 * H * D), device="cuda", dtype=torch.bfloat16)   # the KV-cache form: 2-level output rows
        hip.gemm(a, wqkv, dtype=dtype, epilogue=hip.EPI_ROPE, rope=(cos.cuda(), sin.cuda()), rope_seq=S, rope_cols=2 * H * D, w_frag=pack.frag32(wqkv),
                 out=cache.view(B * (S + 5), 3 * H * D), M=B * S, o_rows=(S, (S + 5) * 3 * H * D))
        assert hip.lib().stllm_last_kernel().decode().startswith(f"gemm_wd_kernel<bf16_t,{wm},ROPE")
        assert float(cache[:, S:].abs().max()) == 0.0
        qkv = cache[:, :S].double().cpu().view(B, S, 3, H, D)
        c, s_ = O.rope_tables(S, D)
        q = (a64 @ wq64.t()).view(B, S, H, D).transpose(1, 2)
        k = (a64 @ wk64.t()).view(B, S, H, D).transpose(1, 2)
        q = q * c.double() + O._rotate_half(q) * s_.double()
        k = k * c.double() + O._rotate_half(k) * s_.double()
        perm = pack.rope_head_perm(1)
        check(qkv[:, :, 0].transpose(1, 2), q[..., perm], OUT_TOL[dtype], "wd q rope")
        check(qkv[:, :, 1].transpose(1, 2), k[..., perm], OUT_TOL[dtype], "wd k rope")
        check(qkv[:, :, 2], (a64 @ wv64.t()).view(B, S, H, D), OUT_TOL[dtype], "wd v")
        # 2-level A rows (row groups inside a larger buffer)
        N_, S2, Q, C, Nout = 5, 44, 32, 768, 256
        buf, buf64 = rnd("buf", (N_ * S2, C), dtype)
        w, w64 = rnd("w2", (Nout, C), dtype, 0.05)
        outb = torch.zeros((N_ * S2, Nout), device="cuda", dtype=torch.float32)
        hip.gemm(buf, w, dtype=dtype, out=outb, out_f32=True, M=N_ * Q, a_rows=(Q, S2 * C), o_rows=(Q, S2 * Nout), w_frag=pack.frag32(w))
        assert hip.lib().stllm_last_kernel().decode().startswith("gemm_wd_kernel<")
        check(outb.view(N_, S2, Nout)[:, :Q].reshape(-1, Nout), buf64.view(N_, S2, C)[:, :Q].reshape(-1, C) @ w64.t(), ACC_TOL[dtype], "wd query rows")
        assert float(outb.view(N_, S2, Nout)[:, Q:].abs().max()) == 0.0
    finally:
        hip.set_option("gemm_wd", -1)


def test_llama_prefill_qkv_runs_on_the_w_direct_kernel(hip):
    """automatic dispatch (round 6): with the fragment-major copy at hand the Llama qkv GEMM at 449..640 rows (5 x 48 = 240 tiles of 128 x 256 = one round)
    runs on gemm_wd (66.5 vs 74.7 us inside the model, profiles/r06_bench_ab_wd.log); other row counts, and callers without the copy, keep the other kernels.
    Same numbers either way (1 ulp of the 16-bit output at most)."""
    from stllm_amd import pack
    dtype = "bf16"
    H, D, K = 32, 128, 4096
    wq, _ = rnd("q24.wq", (H * D, K), dtype, 0.02)
    wk, _ = rnd("q24.wk", (H * D, K), dtype, 0.02)
    wv, _ = rnd("q24.wv", (H * D, K), dtype, 0.02)
    w = pack.llama_qkv(wq, wk, wv, dtype, n_heads=H)
    wf = pack.frag32_or_none(w)
    assert wf is not None
    for S, want in ((576, True), (640, True), (580, True), (448, False), (384, False), (150, False)):
        a, _ = rnd(f"q24.a{S}", (S, K), dtype)
        cos, sin = pack.rope_tables(S)
        kw = dict(dtype=dtype, epilogue=hip.EPI_ROPE, rope=(cos.cuda(), sin.cuda()), rope_seq=S, rope_cols=2 * H * D)
        out = hip.gemm(a, w, w_frag=wf, **kw)
        name = hip.lib().stllm_last_kernel().decode()
        assert name.startswith("gemm_wd_kernel<bf16_t,4,ROPE") == want, (S, name)
        ref = hip.gemm(a, w, **kw)
        assert not hip.lib().stllm_last_kernel().decode().startswith("gemm_wd")
        d = (out.float() - ref.float()).abs()
        assert float((d / ref.float().abs().clamp(min=1.0)).max()) <= 2 ** -7, (S, float(d.max()))
    assert hip.gemm_workspace_ok(), hip.lib().stllm_last_error().decode()


W4_ODD_SHAPES = [(4112 // 2, 4224, 1408), (576, 1536, 4096), (300, 768, 3072), (97, 384, 6144), (1, 384, 128 * 7), (3072, 2304, 704),
                 (528, 768, 1408), (596, 384, 256)]   # N % 384 == 0 (stllm_gemm wants N % 128 == 0, the tiles N % 192 == 0); the last two: thin tails past 256-row / 192-row tiles


@pytest.mark.parametrize("shape", [43, 33])
@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("M,N,K", W4_ODD_SHAPES)
def test_gemm_w4_192_column_tiles(hip, dtype, shape, M, N, K):
    """round 3: the 256 x 192 and 192 x 192 tiles of the one-wave-per-SIMD kernel (three 32-column fragments per wave: the odd one has
    its own epilogue path): 16-bit STORE outputs with / without bias, GELU, M tails incl. thin rows, determinism; plans that would
    need a K-split are refused (the dispatcher then falls back), which the kernel-name assert makes visible."""
    plan = hip.gemm_w4_plan(M, N, K, 8, shape)
    hip.set_option("gemm_w4", shape)
    try:
        a, a64 = rnd("a", (M, K), dtype, 0.5)
        w, w64 = rnd("w", (N, K), dtype, 0.05)
        b = T("b", (N,), 0.5)
        ref = a64 @ w64.t() + b.double()
        out = hip.gemm(a, w, dtype=dtype, bias=b.cuda())
        name = hip.lib().stllm_last_kernel().decode()
        if plan[2] == 1:
            assert name.startswith(f"gemm_w4_kernel<{'bf16_t' if dtype == 'bf16' else 'f16_t'},{shape // 10},{shape % 10},STORE,0,"), name
        check(out, ref, OUT_TOL[dtype], f"w4 {shape} store [{name}]")
        assert torch.equal(out, hip.gemm(a, w, dtype=dtype, bias=b.cuda())), "not bit-identical across launches"
        check(hip.gemm(a, w, dtype=dtype, bias=b.cuda(), act=hip.ACT_GELU), O.gelu(ref), OUT_TOL[dtype], f"w4 {shape} gelu")
        check(hip.gemm(a, w, dtype=dtype), a64 @ w64.t(), OUT_TOL[dtype], f"w4 {shape} store, no bias")
        # epilogues the 192-column tiles do not have fall through to the other kernels and stay correct
        check(hip.gemm(a, w, dtype=dtype, bias=b.cuda(), out_f32=True), ref, ACC_TOL[dtype], "forced odd tile, fp32 output -> fallback")
        assert hip.gemm_workspace_ok(), hip.lib().stllm_last_error().decode()
    finally:
        hip.set_option("gemm_w4", -1)


@pytest.mark.parametrize("M,D", [(4112, 1408), (576, 4096), (37, 768)])
def test_norms_write_the_split_image(hip, M, D):
    """stllm_layernorm / stllm_rmsnorm with dtype STLLM_BF16X3: out_t = bf16 [M, 3 D] = (hi | hi | lo) of the fp32 result, bit-identical to
    splitting the fp32 output afterwards (both norm kernels: one wave per row and one workgroup per row)."""
    x = T("x3.norm.x", (M, D), 1.5).cuda()
    gam, bet = (T("x3.norm.g", (D,), 0.2) + 1.0).cuda(), T("x3.norm.b", (D,), 0.1).cuda()
    ln32 = hip.layernorm(x, gam, bet, 1e-6, dtype="fp32")[0]
    assert torch.equal(hip.layernorm(x, gam, bet, 1e-6, dtype="bf16x3")[0], hip.split3(ln32))
    rms32 = hip.rmsnorm(x, gam, 1e-6, dtype="fp32")[0]
    assert torch.equal(hip.rmsnorm(x, gam, 1e-6, dtype="bf16x3")[0], hip.split3(rms32))


BF16X3_CASES = [  # (name, M, N, K, epilogue): the benchmarked GEMM shapes of the ViT and of the Llama prefill + a ragged small one
    ("vit_qkv", 4112, 4224, 1408, "store"), ("vit_proj", 4112, 1408, 1408, "resid"), ("vit_fc1", 4112, 6144, 1408, "gelu"),
    ("vit_fc2", 4112, 1408, 6144, "resid"), ("llm_qkv", 576, 12288, 4096, "rope"), ("llm_gu", 576, 22016, 4096, "swiglu"),
    ("llm_down", 576, 4096, 11008, "resid"), ("small", 70, 256, 128, "relu")]


@pytest.mark.parametrize("name,M,N,K,epi", BF16X3_CASES, ids=[c[0] for c in BF16X3_CASES])
def test_gemm_bf16x3_split_mode(hip, name, M, N, K, epi):
    """STLLM_BF16X3 (round 4, the split verify mode): fp32 A, weight packed (hi | lo | hi) by pack.split3_weight, ONE bf16 GEMM with
    K' = 3 K on the 16-bit kernels + fp32 post-epilogue, against fp64 of the UNROUNDED fp32 operands.  Bound: 2e-5 of the abs-max — the
    exact-fp32 MFMA mode's bound — where the plain bf16 GEMM of the same operands is off by ~4e-3."""
    from stllm_amd import pack
    a = T(f"x3.a.{name}", (M, K), 1.0)
    w = T(f"x3.w.{name}", (N, K), 0.03)
    b = T(f"x3.b.{name}", (N,), 0.2)
    a64, w64 = a.double(), w.double()
    w3 = pack.split3_weight(w.cuda())
    assert w3.shape == (N, 3 * K) and w3.dtype == torch.bfloat16
    acc = a64 @ w64.t()
    ad, bd = a.cuda(), b.cuda()
    if epi == "store":
        got, ref = hip.gemm(ad, w3, dtype="fp32", bias=bd), acc + b.double()
    elif epi in ("gelu", "relu"):
        got = hip.gemm(ad, w3, dtype="fp32", bias=bd, act=hip.ACT_GELU if epi == "gelu" else hip.ACT_RELU)
        ref = O.gelu(acc + b.double()) if epi == "gelu" else torch.relu(acc + b.double())
    elif epi == "resid":
        x = T(f"x3.x.{name}", (M, N), 2.0)
        xd = x.cuda()
        got, ref = hip.gemm(ad, w3, dtype="fp32", epilogue=hip.EPI_RESID, bias=bd, resid=xd), x.double() + acc + b.double()
        assert got.data_ptr() == xd.data_ptr()
    elif epi == "swiglu":   # packed [32 gate | 32 up] rows (pack.llama_gate_up)
        g_, u_ = acc.view(M, N // 64, 2, 32)[:, :, 0], acc.view(M, N // 64, 2, 32)[:, :, 1]
        got, ref = hip.gemm(ad, w3, dtype="fp32", epilogue=hip.EPI_SWIGLU), (F.silu(g_) * u_).reshape(M, N // 2)
    else:                   # rope on the packed [x_lo | x_hi] groups of the first 2/3 of the columns, the rest stored as is
        S = 96
        cos, sin = pack.rope_tables(S, 128)
        got = hip.gemm(ad, w3, dtype="fp32", epilogue=hip.EPI_ROPE, rope=(cos.cuda(), sin.cuda()), rope_seq=S, rope_cols=2 * N // 3)
        v = acc.clone().view(M, N // 64, 2, 32)
        pos = torch.arange(M) % S
        for g in range((2 * N // 3) // 64):
            c_, s_ = cos.double()[pos][:, (g & 1) * 32:(g & 1) * 32 + 32], sin.double()[pos][:, (g & 1) * 32:(g & 1) * 32 + 32]
            lo, hi_ = acc.view(M, N // 64, 2, 32)[:, g, 0], acc.view(M, N // 64, 2, 32)[:, g, 1]
            v[:, g, 0], v[:, g, 1] = lo * c_ - hi_ * s_, hi_ * c_ + lo * s_
        ref = v.reshape(M, N)
    assert got.dtype == torch.float32
    check(got, ref, 2e-5, f"bf16x3 {name}")
    # chained form (what the stack entry points issue): A already split, GELU / SwiGLU leave as the split image — the same bits as the unfused calls
    a3 = hip.split3(ad)
    assert torch.equal(a3[:, :K], ad.to(torch.bfloat16)) and torch.equal(a3[:, 2 * K:], (ad - ad.to(torch.bfloat16).float()).to(torch.bfloat16))
    if epi in ("gelu", "relu"):
        act = hip.ACT_GELU if epi == "gelu" else hip.ACT_RELU
        assert torch.equal(hip.gemm(a3, w3, dtype="fp32", bias=bd, act=act, a_presplit=True, out_split=True), hip.split3(got))
    elif epi == "swiglu":
        assert torch.equal(hip.gemm(a3, w3, dtype="fp32", epilogue=hip.EPI_SWIGLU, a_presplit=True, out_split=True), hip.split3(got))
    elif epi == "store":
        assert torch.equal(hip.gemm(a3, w3, dtype="fp32", bias=bd, a_presplit=True), got)
    plain = hip.gemm(ad.to(torch.bfloat16), w.cuda().to(torch.bfloat16), dtype="bf16", out_f32=True)
    e3 = (hip.gemm(ad, w3, dtype="fp32").double().cpu() - acc).abs().max().item()
    e1 = (plain.double().cpu() - acc).abs().max().item()
    print(f"\n[bf16x3 {name}] max-abs err {e3:.3e} vs plain bf16 {e1:.3e} (abs-max {acc.abs().max().item():.2f}); kernel {hip.lib().stllm_last_kernel().decode()}")
    assert e3 * 50 < e1
    assert hip.gemm_workspace_ok()


def test_gemm_w4_thin_tail_at_vit_fc1_size(hip):
    """ViT fc1 at the benchmarked size on the 256 x 128 tile: 4112 rows = 16 tile rows (768 tiles = three whole rounds) + 16 thin
    rows; GELU 16-bit output; fc2 on the same tile (176 tiles + thin rows on idle workgroups too); the thin rows are checked on
    their own as well."""
    M, N, K = 4112, 6144, 1408
    assert hip.gemm_w4_plan(M, N, K, 2 | 8, 42)[:3] == (3, 0, 1)
    assert hip.gemm_w4_plan(M, N, K, 2 | 8, 43)[:3] == (2, 0, 1)
    for dtype in ("bf16", "fp16"):       # round 3: the automatic choice for fc1 is the 256 x 192 tile (two whole rounds + thin rows)
        a, a64 = rnd("a", (M, K), dtype, 0.5)
        w, w64 = rnd("w", (N, K), dtype, 0.05)
        b = T("b", (N,), 0.5)
        ref = (a64.cuda() @ w64.cuda().t() + b.double().cuda()).cpu()
        out = hip.gemm(a, w, dtype=dtype, bias=b.cuda(), act=hip.ACT_GELU)
        assert hip.lib().stllm_last_kernel().decode().startswith(f"gemm_w4_kernel<{'bf16_t' if dtype == 'bf16' else 'f16_t'},4,3,STORE,1,"), hip.lib().stllm_last_kernel()
        check(out, O.gelu(ref), OUT_TOL[dtype], "fc1 gelu (auto: 256 x 192)")
        check(out[4096:], O.gelu(ref[4096:]), OUT_TOL[dtype], "fc1 gelu, thin rows (auto)")
    hip.set_option("gemm_w4", 42)
    try:
        for dtype in ("bf16", "fp16"):
            a, a64 = rnd("a", (M, K), dtype, 0.5)
            w, w64 = rnd("w", (N, K), dtype, 0.05)
            b = T("b", (N,), 0.5)
            ref = (a64.cuda() @ w64.cuda().t() + b.double().cuda()).cpu()
            out = hip.gemm(a, w, dtype=dtype, bias=b.cuda(), act=hip.ACT_GELU)
            assert hip.lib().stllm_last_kernel().decode().startswith(f"gemm_w4_kernel<{'bf16_t' if dtype == 'bf16' else 'f16_t'},4,2,STORE,1,")
            check(out, O.gelu(ref), OUT_TOL[dtype], "fc1 gelu")
            check(out[4096:], O.gelu(ref[4096:]), OUT_TOL[dtype], "fc1 gelu, thin rows")
            assert torch.equal(out, hip.gemm(a, w, dtype=dtype, bias=b.cuda(), act=hip.ACT_GELU)), "not bit-identical across launches"
            wt, wt64 = rnd("w2", (1408, 6144), dtype, 0.05)
            a2, a264 = rnd("a2", (M, 6144), dtype, 0.5)
            x = T("x", (M, 1408), 2.0)
            xd = x.cuda()
            hip.gemm(a2, wt, dtype=dtype, epilogue=hip.EPI_RESID, resid=xd)
            check(xd, (x.double().cuda() + a264.cuda() @ wt64.cuda().t()).cpu(), ACC_TOL[dtype], "fc2 resid")
            check(xd[4096:], (x.double().cuda() + a264.cuda() @ wt64.cuda().t()).cpu()[4096:], ACC_TOL[dtype], "fc2 resid, thin rows")
        assert hip.gemm_workspace_ok(), hip.lib().stllm_last_error().decode()
    finally:
        hip.set_option("gemm_w4", -1)


@pytest.mark.parametrize("shape,n_seq", [(42, 9), (32, 7), (34, 7)])
def test_gemm_w4_thin_tail_with_two_level_rows(hip, shape, n_seq):
    """the thin tail rows go through the same 2-level row addressing as the tiles (Q-Former row groups: A rows read from / out rows
    written into a [n, S, C] buffer): n x 32 query rows = one full tile + 32 thin rows, fp32 output and the residual epilogue"""
    S, Q, C, Nout = 44, 32, 768, 384
    hip.set_option("gemm_w4", shape)
    try:
        for dtype in ("bf16", "fp16"):
            buf, buf64 = rnd("buf", (n_seq * S, C), dtype)
            w, w64 = rnd("w", (Nout, C), dtype, 0.05)
            ref = buf64.view(n_seq, S, C)[:, :Q].reshape(-1, C) @ w64.t()
            out = torch.zeros((n_seq * S, Nout), device="cuda", dtype=torch.float32)
            hip.gemm(buf, w, dtype=dtype, out=out, out_f32=True, M=n_seq * Q, a_rows=(Q, S * C), o_rows=(Q, S * Nout))
            assert "gemm_w4_kernel" in hip.lib().stllm_last_kernel().decode()
            o = out.view(n_seq, S, Nout)
            check(o[:, :Q].reshape(-1, Nout), ref, ACC_TOL[dtype], "query rows")
            assert float(o[:, Q:].abs().max()) == 0.0, "rows outside the groups were written"
            x = T("x", (n_seq * Q, Nout), 2.0)
            xd = x.cuda()
            hip.gemm(buf, w, dtype=dtype, epilogue=hip.EPI_RESID, resid=xd, M=n_seq * Q, a_rows=(Q, S * C))
            check(xd, x.double() + ref, ACC_TOL[dtype], "query rows, residual")
        assert hip.gemm_workspace_ok(), hip.lib().stllm_last_error().decode()
    finally:
        hip.set_option("gemm_w4", -1)


@pytest.mark.parametrize("shape", [32, 34, 42, 24])
def test_gemm_w4_swiglu_rope_rows(hip, shape):
    from stllm_amd import pack
    dtype = "bf16"
    hip.set_option("gemm_w4", shape)
    try:
        M, K, I = 333, 1024, 1024 + 128 * 3
        a, a64 = rnd("a", (M, K), dtype)
        wg, wg64 = rnd("wg", (I, K), dtype, 0.05)
        wu, wu64 = rnd("wu", (I, K), dtype, 0.05)
        out = hip.gemm(a, pack.llama_gate_up(wg, wu, dtype), dtype=dtype, epilogue=hip.EPI_SWIGLU)
        assert "gemm_w4_kernel" in hip.lib().stllm_last_kernel().decode()
        check(out, F.silu(a64 @ wg64.t()) * (a64 @ wu64.t()), OUT_TOL[dtype], "w4 swiglu")
        B, S, H, D = 2, 150, 4, 128
        a, a64 = rnd("a2", (B * S, K), dtype)
        wq, wq64 = rnd("wq", (H * D, K), dtype, 0.05)
        wk, wk64 = rnd("wk", (H * D, K), dtype, 0.05)
        wv, wv64 = rnd("wv", (H * D, K), dtype, 0.05)
        cos, sin = pack.rope_tables(S)
        qkv = hip.gemm(a, pack.llama_qkv(wq, wk, wv, dtype, n_heads=H), dtype=dtype, epilogue=hip.EPI_ROPE,
                       rope=(cos.cuda(), sin.cuda()), rope_seq=S, rope_cols=2 * H * D).double().cpu().view(B, S, 3, H, D)
        assert "gemm_w4_kernel" in hip.lib().stllm_last_kernel().decode()
        c, s = O.rope_tables(S, D)
        q = (a64 @ wq64.t()).view(B, S, H, D).transpose(1, 2)
        k = (a64 @ wk64.t()).view(B, S, H, D).transpose(1, 2)
        q = q * c.double() + O._rotate_half(q) * s.double()
        k = k * c.double() + O._rotate_half(k) * s.double()
        perm = pack.rope_head_perm(1)
        check(qkv[:, :, 0].transpose(1, 2), q[..., perm], OUT_TOL[dtype], "w4 q rope")
        check(qkv[:, :, 1].transpose(1, 2), k[..., perm], OUT_TOL[dtype], "w4 k rope")
        check(qkv[:, :, 2], (a64 @ wv64.t()).view(B, S, H, D), OUT_TOL[dtype], "w4 v")
        # 2-level row indexing (Q-Former style row groups)
        N_, S2, Q, C, Nout = 5, 44, 32, 768, 256
        buf, buf64 = rnd("buf", (N_ * S2, C), dtype)
        w, w64 = rnd("w2", (Nout, C), dtype, 0.05)
        outb = torch.zeros((N_ * S2, Nout), device="cuda", dtype=torch.float32)
        hip.gemm(buf, w, dtype=dtype, out=outb, out_f32=True, M=N_ * Q, a_rows=(Q, S2 * C), o_rows=(Q, S2 * Nout))
        assert "gemm_w4_kernel" in hip.lib().stllm_last_kernel().decode()
        check(outb.view(N_, S2, Nout)[:, :Q].reshape(-1, Nout), buf64.view(N_, S2, C)[:, :Q].reshape(-1, C) @ w64.t(), ACC_TOL[dtype], "w4 query rows")
        assert float(outb.view(N_, S2, Nout)[:, Q:].abs().max()) == 0.0
    finally:
        hip.set_option("gemm_w4", -1)


def test_llama_prefill_qkv_runs_on_the_128x256_tile(hip):
    """round 4: a ROPE-epilogue GEMM whose 128 x 256 tiles make one round of 192..256 tiles (Llama qkv at 385..640 rows) is dispatched to the
    one-wave kernel's 128 x 256 tile (option gemm_w4_wide, default on); other row counts and the switched-off option keep the 128 x 128 kernel.
    Same numbers either way (the epilogue arithmetic is the same; 1 ulp of the 16-bit output at most)."""
    from stllm_amd import pack
    dtype = "bf16"
    H, D, K = 32, 128, 4096
    wq, _ = rnd("q24.wq", (H * D, K), dtype, 0.02)
    wk, _ = rnd("q24.wk", (H * D, K), dtype, 0.02)
    wv, _ = rnd("q24.wv", (H * D, K), dtype, 0.02)
    w = pack.llama_qkv(wq, wk, wv, dtype, n_heads=H)
    for S, want in ((576, True), (400, True), (640, True), (641, False), (384, False), (150, False)):
        a, _ = rnd(f"q24.a{S}", (S, K), dtype)
        cos, sin = pack.rope_tables(S)
        kw = dict(dtype=dtype, epilogue=hip.EPI_ROPE, rope=(cos.cuda(), sin.cuda()), rope_seq=S, rope_cols=2 * H * D)
        out = hip.gemm(a, w, **kw)
        name = hip.lib().stllm_last_kernel().decode()
        assert name.startswith("gemm_w4_kernel<bf16_t,2,4,ROPE") == want, (S, name)
        hip.set_option("gemm_w4_wide", 0)
        try:
            ref = hip.gemm(a, w, **kw)
            assert "gemm_w4_kernel<bf16_t,2,4" not in hip.lib().stllm_last_kernel().decode()
        finally:
            hip.set_option("gemm_w4_wide", 1)
        d = (out.float() - ref.float()).abs()
        assert float((d / ref.float().abs().clamp(min=1.0)).max()) <= 2 ** -7, (S, float(d.max()))
    assert hip.gemm_workspace_ok(), hip.lib().stllm_last_error().decode()


@pytest.mark.parametrize("M,N,K,epi,want", [
    (9216, 12288, 4096, "rope", "gemm_w4_kernel<bf16_t,3,4,ROPE"),      # training forward qkv: 9 whole rounds of 192 x 256 (was the phased kernel: 1040-1060 vs 770-790 us)
    (1088, 12288, 4096, "rope", "gemm_w4_kernel<bf16_t,"),               # c4's unmasked prefill: a one-wave plan, not the phased kernel's ROPE epilogue (166 vs 121 us)
    (2304, 4096, 11008, "resid", "gemm_w4_kernel<bf16_t,"),              # c3's batched prefill, down_proj: one whole round + a hidden K-split (228 vs 189-197 us)
    (32896, 1408, 6144, "resid", "gemm_p8_kernel<bf16_t,4,RESID"),       # ViT fc2 at 128 frames: the 256-row phased tile (582 vs 504 us)
    (296, 12288, 4096, "rope", "gemm_w4_kernel<bf16_t,3,2,ROPE"),         # c5's masked prefill: ONE partial round of 192 x 128 tiles, nothing exchanged (68 vs 49 us; c5 34.1 -> 33.0 ms)
])
def test_large_m_dispatch_follows_the_round4_audit(hip, M, N, K, epi, want):
    """round 4: the cost models were re-calibrated on dispatch audits at 4 k - 66 k rows (tools/gemm_bench.py --audit, profiles/r04_gemm_dispatch_audit_*.log).
    The choices that moved c3 on one GPU by 5.6 % and the training step's forward qkv by 25 % are pinned here, each checked against the 128 x 128 kernels."""
    from stllm_amd import pack
    dtype = "bf16"
    a, _ = rnd(f"lm.a.{M}.{K}", (M, K), dtype, 0.5)
    w, _ = rnd(f"lm.w.{N}.{K}", (N, K), dtype, 0.02)
    kw = dict(dtype=dtype)
    x = None
    if epi == "rope":
        cos, sin = pack.rope_tables(576)
        kw.update(epilogue=hip.EPI_ROPE, rope=(cos.cuda(), sin.cuda()), rope_seq=576 if M % 576 == 0 else M // 2, rope_cols=8192)
        if M % 576:
            cos, sin = pack.rope_tables(M // 2)
            kw["rope"] = (cos.cuda(), sin.cuda())
    else:
        x = T(f"lm.x.{M}.{N}", (M, N), 1.0).cuda()
        kw.update(epilogue=hip.EPI_RESID)
    run = lambda: hip.gemm(a, w, **(dict(kw, resid=x.clone()) if x is not None else kw))
    out = run()
    name = hip.lib().stllm_last_kernel().decode()
    assert name.startswith(want), name
    hip.set_option("gemm_w4", 0); hip.set_option("gemm_p8", 0)
    try:
        ref = run()
        assert hip.lib().stllm_last_kernel().decode().startswith("gemm_kernel<")
    finally:
        hip.set_option("gemm_w4", -1); hip.set_option("gemm_p8", -1)
    d = (out.float() - ref.float()).abs()
    tol = 1e-3 if x is not None else 2 ** -7     # fp32 residual stream / one bf16 ulp of the 16-bit output
    assert float((d / ref.float().abs().clamp(min=1.0)).max()) <= tol, float(d.max())
    assert hip.gemm_workspace_ok(), hip.lib().stllm_last_error().decode()


def test_gemm_phased_auto_dispatch(hip):
    """the cost model sends the long-K / few-row-tile prefill shapes to the phased kernel and keeps small problems on
    the 128x128 kernels"""
    dtype = "bf16"
    a, a64 = rnd("a", (576, 11008), dtype, 0.5)
    w, w64 = rnd("w", (4096, 11008), dtype, 0.05)
    x = torch.zeros((576, 4096), device="cuda")
    hip.gemm(a, w, dtype=dtype, epilogue=hip.EPI_RESID, resid=x)
    assert hip.lib().stllm_last_kernel().decode().startswith("gemm_p8_kernel<bf16_t,3,RESID")
    check(x, a64 @ w64.t(), ACC_TOL[dtype], "auto p8 resid")
    a, a64 = rnd("a3", (512, 768), dtype, 0.5)
    w, w64 = rnd("w3", (768, 768), dtype, 0.05)
    hip.gemm(a, w, dtype=dtype)
    assert hip.lib().stllm_last_kernel().decode().startswith("gemm_kernel<")
    assert hip.gemm_workspace_ok(), hip.lib().stllm_last_error().decode()   # no split-K exchange of this process ever timed out


def _gemv_kernel_expected(M, K, mfma):
    """which kernel family stllm_gemm picks in the decode regime (gemv.hip: stllm_gemv_launch)"""
    if mfma != 0 and M >= (1 if mfma == 1 else 3) and K % 64 == 0:
        return "gemv_mfma_kernel<"
    mr = M if M <= 2 else (M + 1) // 2 * 2
    if M <= 8 and mr * K * 2 <= 150 * 1024:
        return "gemv_kernel<"
    return "gemm"       # the staged rows of A do not fit the LDS: the tile kernels


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("M,mfma", [(1, -1), (2, -1), (3, -1), (4, -1), (5, -1), (8, -1), (12, -1), (16, -1), (3, 0), (5, 0), (8, 0), (1, 1), (2, 1)])
@pytest.mark.parametrize("N,K", [(4096, 4096), (256, 11008), (1536, 704)])
def test_gemv_decode_regime(hip, dtype, M, N, K, mfma):
    """skinny kernels of the decode regime (gemv.hip; M <= 16 since round 2: the 5 beams of demo.py, small serving batches): the VALU
    kernel (M <= 2 by default) and the matrix-core kernel (M >= 3): every epilogue against fp64, K tails (K % 512 != 0), strided
    output rows, the LDS-fit fallbacks"""
    from stllm_amd import pack
    hip.set_option("gemv_mfma", mfma)
    try:
        _gemv_decode_regime(hip, dtype, M, N, K, _gemv_kernel_expected(M, K, mfma))
    finally:
        hip.set_option("gemv_mfma", -1)


def _gemv_decode_regime(hip, dtype, M, N, K, want):
    from stllm_amd import pack
    a, a64 = rnd("a", (M, K), dtype, 0.5)
    w, w64 = rnd("w", (N, K), dtype, 0.05)
    b = T("b", (N,), 0.5)
    ref = a64 @ w64.t() + b.double()
    out = hip.gemm(a, w, dtype=dtype, bias=b.cuda(), out_f32=True)
    assert hip.lib().stllm_last_kernel().decode().startswith(want), (hip.lib().stllm_last_kernel().decode(), want)
    check(out, ref, ACC_TOL[dtype], "gemv store f32")
    check(hip.gemm(a, w, dtype=dtype, bias=b.cuda(), act=hip.ACT_GELU), O.gelu(ref), OUT_TOL[dtype], "gemv gelu T")
    x = T("x", (M, N), 2.0)
    xd = x.cuda()
    hip.gemm(a, w, dtype=dtype, epilogue=hip.EPI_RESID, bias=b.cuda(), resid=xd)
    check(xd, x.double() + ref, ACC_TOL[dtype], "gemv resid")
    # output rows with a stride (the decode step writes one row per sequence into the KV cache buffer)
    buf = torch.zeros((M, 3, N), device="cuda", dtype=hip.torch_dtype(dtype))
    hip.gemm(a, w, dtype=dtype, out=buf[:, 1])
    check(buf[:, 1], a64 @ w64.t(), OUT_TOL[dtype], "gemv strided rows")
    assert float(buf[:, 0].abs().max()) == 0.0 and float(buf[:, 2].abs().max()) == 0.0
    if N % 128 == 0 and K % 64 == 0 and N >= 1024:
        I = N // 2
        wg, wg64 = rnd("wg", (I, K), dtype, 0.05)
        wu, wu64 = rnd("wu", (I, K), dtype, 0.05)
        o = hip.gemm(a, pack.llama_gate_up(wg, wu, dtype), dtype=dtype, epilogue=hip.EPI_SWIGLU)
        assert hip.lib().stllm_last_kernel().decode().startswith(want)
        check(o, F.silu(a64 @ wg64.t()) * (a64 @ wu64.t()), OUT_TOL[dtype], "gemv swiglu")
        H, D = N // 3 // 128, 128
        if H >= 1 and H * 3 * 128 == N:
            wq, wq64 = rnd("wq", (H * D, K), dtype, 0.05)
            wk, wk64 = rnd("wk", (H * D, K), dtype, 0.05)
            wv, wv64 = rnd("wv", (H * D, K), dtype, 0.05)
            S = 7
            cos, sin = pack.rope_tables(S)
            pos = 5   # decode: every row sits at position `pos` (rope_seq = 1 with a one-row table slice)
            qkv = hip.gemm(a, pack.llama_qkv(wq, wk, wv, dtype, n_heads=H), dtype=dtype, epilogue=hip.EPI_ROPE,
                           rope=(cos[pos:pos + 1].cuda(), sin[pos:pos + 1].cuda()), rope_seq=1, rope_cols=2 * H * D).double().cpu().view(M, 3, H, D)
            c, s_ = O.rope_tables(S, D)
            q = (a64 @ wq64.t()).view(M, H, D)
            k = (a64 @ wk64.t()).view(M, H, D)
            q = q * c[pos].double() + O._rotate_half(q) * s_[pos].double()
            k = k * c[pos].double() + O._rotate_half(k) * s_[pos].double()
            perm = pack.rope_head_perm(1)
            check(qkv[:, 0], q[..., perm], OUT_TOL[dtype], "gemv q rope")
            check(qkv[:, 1], k[..., perm], OUT_TOL[dtype], "gemv k rope")
            check(qkv[:, 2], (a64 @ wv64.t()).view(M, H, D), OUT_TOL[dtype], "gemv v")


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("single", [1, 0])
@pytest.mark.parametrize("B,H,Skv", [(1, 32, 580), (2, 4, 1), (3, 2, 47), (1, 8, 2048), (2, 3, 97), (5, 32, 601), (1, 4, 1536), (1, 4, 1537)])
def test_attention_decode_split_kv(hip, dtype, B, H, Skv, single):
    """one-token decode against a KV cache laid out like LlamaModel's (fused [B, max_len, 3*H*128] buffer): the single-pass
    kernel (one workgroup per head, Skv <= 1536) and the split-KV pair vs fp64 softmax(q K^T / sqrt(d)) V and vs the tile kernel"""
    hip.set_option("attn_decode_single", single)
    D, max_len = 128, Skv + 5
    td = hip.torch_dtype(dtype)
    cache, c64 = rnd("kvcache", (B * max_len, 3 * H * D), dtype)
    full = cache.view(B * max_len, 3 * H * D)
    row = cache.view(B, max_len, 3 * H * D)[:, Skv - 1]                       # the newest token's fused QKV row: [B, 3HD] strided view
    ML3 = max_len * 3 * H * D
    kw = dict(B=B, H=H, Sq=1, Skv=Skv, D=D, scale=D ** -0.5, causal=False, q_strides=(ML3, 3 * H * D), k_strides=(ML3, 3 * H * D),
              v_strides=(ML3, 3 * H * D))
    got = hip.attention(row[:, :H * D], full[:, H * D:2 * H * D], full[:, 2 * H * D:], **kw)
    c = c64.view(B, max_len, 3, H, D)
    q = c[:, Skv - 1, 0]                                                        # [B, H, D]
    k, v = c[:, :Skv, 1].transpose(1, 2), c[:, :Skv, 2].transpose(1, 2)        # [B, H, Skv, D]
    ref = (torch.softmax((q.unsqueeze(2) @ k.transpose(-1, -2)) * D ** -0.5, dim=-1) @ v).reshape(B, H * D)
    check(got, ref, 2 * OUT_TOL[dtype], "decode attention vs fp64")
    hip._decode_attn = False
    try:
        tile = hip.attention(row[:, :H * D], full[:, H * D:2 * H * D], full[:, 2 * H * D:], **kw)
    finally:
        hip._decode_attn = True
    check(got, tile.double().cpu(), 2 * OUT_TOL[dtype], "decode attention vs tile kernel")
    hip.set_option("attn_decode_single", 1)


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("M", [1, 5])
def test_gemv_fused_rmsnorm_operand(hip, dtype, M):
    """decode step: Llama's RMSNorm fused into the GEMV that consumes it (stllm_gemm_args.a_norm_*): same numbers as the
    rmsnorm kernel followed by the GEMV, and fp64-close; rejected outside the decode regime"""
    from stllm_amd import pack
    K, I = 4096, 1024
    x = T("nx", (M, K), 1.7).cuda()
    gamma = (T("ngamma", (K,), 0.2) + 1.0).cuda()
    wg, wg64 = rnd("wg", (I, K), dtype, 0.05)
    wu, wu64 = rnd("wu", (I, K), dtype, 0.05)
    wgu = pack.llama_gate_up(wg, wu, dtype)
    h16, _ = hip.rmsnorm(x, gamma, 1e-6, dtype=dtype)
    two = hip.gemm(h16, wgu, dtype=dtype, epilogue=hip.EPI_SWIGLU)
    one = hip.gemm(None, wgu, dtype=dtype, epilogue=hip.EPI_SWIGLU, a_norm=(x, gamma, 1e-6))
    assert "gemv_kernel" in hip.lib().stllm_last_kernel().decode()
    assert (one.float() - two.float()).abs().max().item() <= 2e-2 * two.float().abs().max().item()
    x64 = x.double().cpu()
    hn = (gamma.double().cpu() * x64 * torch.rsqrt((x64 ** 2).mean(-1, keepdim=True) + 1e-6))
    hn = hn.to(hip.torch_dtype(dtype)).double()
    check(one, F.silu(hn @ wg64.t()) * (hn @ wu64.t()), OUT_TOL[dtype], "fused rmsnorm gemv swiglu vs fp64")
    with pytest.raises(RuntimeError, match="a_norm"):
        hip.gemm(None, wgu, dtype=dtype, epilogue=hip.EPI_SWIGLU, a_norm=(T("nx9", (9, K), 1.0).cuda(), gamma, 1e-6))


def test_gemm_rejects_bad_shapes(hip):
    a = torch.zeros((8, 100), device="cuda", dtype=torch.bfloat16)
    w = torch.zeros((128, 100), device="cuda", dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="multiple"):
        hip.gemm(a, w, dtype="bf16")
    with pytest.raises(RuntimeError):
        hip.gemm(a.cpu(), w, dtype="bf16")


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("D", [768, 1408, 4096])
def test_layernorm_rmsnorm(hip, dtype, D):
    M = 131
    x = T("x", (M, D), 3.0) + 0.7
    g = T("g", (D,), 0.2) + 1.0
    b = T("b", (D,), 0.1)
    ref = F.layer_norm(x.double(), (D,), g.double(), b.double(), 1e-6)
    ot, of = hip.layernorm(x.cuda(), g.cuda(), b.cuda(), 1e-6, dtype=dtype, want_f32=True)
    check(of, ref, 2e-6, "layernorm f32")
    check(ot, ref, OUT_TOL[dtype], "layernorm T")
    ref = O.rms_norm(x.double(), g.double(), 1e-6)
    ot, of = hip.rmsnorm(x.cuda(), g.cuda(), 1e-6, dtype=dtype, want_f32=True)
    check(of, ref, 2e-6, "rmsnorm f32")
    check(ot, ref, OUT_TOL[dtype], "rmsnorm T")


def _attn_ref(q, k, v, scale, causal, kv_len):
    B, H, Sq, D = q.shape
    Skv = k.shape[2]
    s = (q @ k.transpose(-1, -2)) * scale
    if causal:
        s = s.masked_fill(torch.ones(Sq, Skv).triu(1).bool(), float("-inf"))
    if kv_len is not None:
        dead = torch.arange(Skv)[None, :] >= torch.tensor(kv_len)[:, None]
        s = s.masked_fill(dead[:, None, None, :], float("-inf"))
    return s.softmax(-1) @ v


ATTN_CASES = [  # name, B, H, Sq, Skv, D, causal, kv_len
    ("vit", 3, 16, 257, 257, 88, False, None),
    ("qf_self", 4, 12, 32, 32, 64, False, None),
    ("qf_self_text", 3, 12, 44, 44, 64, False, [44, 41, 33]),
    ("qf_cross", 4, 12, 32, 257, 64, False, None),
    ("llama_causal", 2, 8, 200, 200, 128, True, None),
    ("llama_causal_pad", 2, 8, 131, 131, 128, True, [131, 97]),
    ("llama_short", 1, 4, 7, 7, 128, True, None),
    ("llama_576", 1, 32, 576, 576, 128, True, None),              # the benchmarked prefill: 4 query tiles x 2 key-split waves per workgroup
    ("llama_576_b2", 2, 32, 576, 576, 128, True, [576, 400]),     # >= 128 workgroups already with 12 query tiles x 1 wave
    ("llama_noncausal", 2, 4, 300, 300, 128, False, [300, 170]),
    ("llama_1100", 1, 32, 1100, 1100, 128, True, None),           # BASELINE configs[3]: the un-masked pass of the MVM forward at T = 32 (S ~ 1100)
    ("llama_1100_ragged", 2, 32, 1100, 1100, 128, True, [1100, 640]),
]


@pytest.mark.parametrize("dma", [1, 0])
@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("case", ATTN_CASES, ids=[c[0] for c in ATTN_CASES])
def test_attention(hip, dtype, case, dma):
    """every attention shape of the path against fp64; `attn_dma`: 1 = default (head_dim 128 on the LDS-DMA kernel with the hardware
    transposing V reads, head_dim 88 likewise), 0 = the register-staged kernels everywhere"""
    hip.set_option("attn_dma", dma)
    hip.set_option("attn_f32_mfma", dma)   # fp32: 1 = the exact-fp32 matrix-core kernel (round 4: attn_mfma_f32_kernel), 0 = the one-wave-per-row vector kernel
    try:
        _attention_case(hip, dtype, case)
    finally:
        hip.set_option("attn_dma", 1)
        hip.set_option("attn_f32_mfma", 1)


def _attention_case(hip, dtype, case):
    _, B, H, Sq, Skv, D, causal, kv_len = case
    # q/k/v as column slices of one fused buffer, exactly how the model calls it
    C = 3 * H * D
    if Sq == Skv:
        buf, buf64 = rnd("qkv", (B * Sq, C), dtype)
        q, k, v = buf[:, :H * D], buf[:, H * D:2 * H * D], buf[:, 2 * H * D:]
        q64, k64, v64 = [buf64[:, i * H * D:(i + 1) * H * D].reshape(B, Sq, H, D).transpose(1, 2) for i in range(3)]
    else:
        q, q64 = rnd("q", (B * Sq, H * D), dtype)
        kv, kv64 = rnd("kv", (B * Skv, 2 * H * D), dtype)
        k, v = kv[:, :H * D], kv[:, H * D:]
        q64 = q64.view(B, Sq, H, D).transpose(1, 2)
        k64 = kv64[:, :H * D].reshape(B, Skv, H, D).transpose(1, 2)
        v64 = kv64[:, H * D:].reshape(B, Skv, H, D).transpose(1, 2)
    scale = D ** -0.5
    ref = _attn_ref(q64, k64, v64, scale, causal, kv_len).transpose(1, 2).reshape(B * Sq, H * D)
    kl = None if kv_len is None else torch.tensor(kv_len, dtype=torch.int32).cuda()
    out = hip.attention(q, k, v, B=B, H=H, Sq=Sq, Skv=Skv, D=D, scale=scale, causal=causal, kv_len=kl)
    if kv_len is not None and causal:
        # padded query rows are defined (attend to the valid keys) — compared as well
        pass
    # P is rounded to the compute dtype before P.V: allow 2 roundings
    check(out, ref, 2 * OUT_TOL[dtype] if dtype != "fp32" else 2e-5, f"attention {case[0]}")


@pytest.mark.parametrize("dtype", ["bf16", "fp16", "fp32"])
def test_attention_online_softmax_rescale(hip, dtype):
    """Force the running max to jump late (rule 26): one key far down the sequence dominates."""
    B, H, S, D = 1, 2, 160, 128
    q, q64 = rnd("q", (S, H * D), dtype, 0.3)
    k, k64 = rnd("k", (S, H * D), dtype, 0.3)
    v, v64 = rnd("v", (S, H * D), dtype)
    k[150] = (q[5] * 6).to(k.dtype)
    k64[150] = k[150].double().cpu()
    r = lambda t: t.view(1, S, H, D).transpose(1, 2)
    ref = _attn_ref(r(q64), r(k64), r(v64), D ** -0.5, False, None).transpose(1, 2).reshape(S, H * D)
    out = hip.attention(q, k, v, B=B, H=H, Sq=S, Skv=S, D=D, scale=D ** -0.5)
    check(out, ref, 2 * OUT_TOL[dtype] if dtype != "fp32" else 2e-5, "late-max rescale")


# ---------------------------------------------------------------------------------------------
def test_gather_mean_cls_cosine_ce(hip):
    D = 4096
    a = T("ga", (50, D))
    b = T("gb", (70, D))
    add = T("gadd", (9, D))
    idx = torch.tensor([3, -1, 49, -70, 0, 7, -5], dtype=torch.int32)
    ia = torch.tensor([0, 8, 3, 3, 1, 2, 5], dtype=torch.int32)
    ref = torch.stack([a[i] if i >= 0 else b[-i - 1] for i in idx.tolist()])
    out = hip.gather_rows(a.cuda(), idx.cuda(), src_b=b.cuda())
    assert torch.equal(out.cpu(), ref)
    out = hip.gather_rows(a.cuda(), idx.cuda(), src_b=b.cuda(), add=add.cuda(), idx_add=ia.cuda())
    assert torch.equal(out.cpu(), ref + add[ia.long()])
    x = T("mx", (2, 8, 32, D))
    check(hip.mean_t(x.cuda()), x.double().mean(1), 1e-6, "mean_t")
    u, w = T("cu", (40, D)), T("cw", (64, D))
    iu = torch.arange(40, dtype=torch.int32)
    iw = torch.randperm(64, generator=torch.Generator().manual_seed(0))[:40].to(torch.int32)
    ref = 2 - 2 * (F.normalize(u.double(), dim=-1) * F.normalize(w.double()[iw.long()], dim=-1)).sum(-1)
    check(hip.cosine_rows(u.cuda(), w.cuda(), iu.cuda(), iw.cuda()), ref, 1e-5, "cosine rows")
    V = 32000
    lg = T("lg", (33, V), 2.0)
    lab = torch.randint(0, V, (33,), generator=torch.Generator().manual_seed(1)).to(torch.int32)
    lab[5] = -100
    ref = F.cross_entropy(lg.double(), lab.long(), ignore_index=-100, reduction="none")
    check(hip.cross_entropy_rows(lg.cuda(), lab.cuda()), ref, 1e-5, "cross entropy rows")
