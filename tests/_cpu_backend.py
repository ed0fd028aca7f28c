"""TEST-ONLY stand-in for ``stllm_amd.hip``: the same call surface implemented with plain fp32 torch ops on the CPU.

Used by ``tests/test_host_orchestration_cpu.py`` to run the *product's host code* (module graph, weight packers,
index tables, 2-level row indexing, frame-parallel + clip-parallel logic under gloo) without a GPU and compare it
with the oracle.  It is NOT a fallback: nothing in ``st-llm_amd/`` imports it, and it lives under ``tests/``.
Every function restates the contract documented in include/stllm_hip.h (incl. the packed weight layouts)."""
import contextlib
import math

import torch
import torch.nn.functional as F

BF16, F16, F32 = 0, 1, 2
EPI_STORE, EPI_RESID, EPI_SWIGLU, EPI_ROPE, EPI_PATCH = 0, 1, 2, 3, 4
ACT_NONE, ACT_GELU, ACT_RELU = 0, 1, 2


def torch_dtype(d):
    return {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}.get(d, d) if isinstance(d, str) else d


def _rows(t, n, rows):
    """row indices of the logical rows 0..n-1 inside the 2-D buffer `t` under (rows_per_batch, batch_stride)"""
    if rows is None:
        return torch.arange(n)
    rpb, bs = rows
    m = torch.arange(n)
    return (m // rpb) * (bs // t.stride(-2)) + (m % rpb)


def gemm(a, w, *, dtype, epilogue=EPI_STORE, bias=None, out=None, out_f32=False, act=ACT_NONE, resid=None,
         rope=None, rope_seq=0, rope_cols=0, frames=None, pos_embed=None, n_frames=0, M=None, a_rows=None, o_rows=None, a_norm=None):
    td = torch_dtype(dtype)
    if a_norm is not None:   # fused RMSNorm operand of the decode regime (stllm_hip.h: a_norm_*)
        xn, gamma, eps = a_norm
        assert a is None and xn.shape[0] <= 8 and td != torch.float32
        a = rmsnorm(xn, gamma, eps, dtype=dtype)[0]
    wf = w.float()
    N = w.shape[0]
    if epilogue == EPI_PATCH:
        n = n_frames
        cols = frames.float().reshape(n, 3, 16, 14, 16, 14).permute(0, 2, 4, 1, 3, 5).reshape(n * 256, 588)
        acc = cols.to(td).float() @ wf[:, :588].t()
        acc = acc + (bias if bias is not None else 0)
        o = out.view(n, 257, -1)
        o[:, 1:] = acc.view(n, 256, N) + pos_embed[1:].view(1, 256, N)
        return out
    if M is None:
        M = a.shape[0]
    if a_rows is None:
        A = a[:M].float()
    else:   # (rows_per_batch, batch_stride in elements): the stride need not be a whole number of rows (the no-Q-Former projector reads 64 rows of
        # 5632 behind the CLS row of every 257 x 1408 frame block) -> a strided view of the storage, as the kernel's byte addressing sees it
        rpb, bs = a_rows
        nb = (M + rpb - 1) // rpb
        A = torch.as_strided(a, (nb, rpb, a.shape[-1]), (bs, a.stride(-2), 1), a.storage_offset()).reshape(nb * rpb, a.shape[-1])[:M].float()
    acc = A @ wf.t()
    if bias is not None:
        acc = acc + bias
    if epilogue == EPI_RESID:
        res = resid[:M].float() + acc
        dst = resid if out is None else out
        dst[_rows(dst, M, o_rows)] = res
        return dst
    if epilogue == EPI_SWIGLU:
        g = acc.view(M, N // 64, 2, 32)
        val = (F.silu(g[:, :, 0]) * g[:, :, 1]).reshape(M, N // 2)
    elif epilogue == EPI_ROPE:
        cos, sin = rope
        x = acc.view(M, N // 64, 2, 32).clone()
        grp = torch.arange(N // 64)
        pos = torch.arange(M) % rope_seq
        fi = (grp % 2)[:, None] * 32 + torch.arange(32)[None, :]             # [groups, 32]
        c, s = cos[pos][:, fi], sin[pos][:, fi]                                # [M, groups, 32]
        live = (grp * 64 < rope_cols)[None, :, None]
        x1, x2 = x[:, :, 0], x[:, :, 1]
        y1 = torch.where(live, x1 * c - x2 * s, x1)
        y2 = torch.where(live, x2 * c + x1 * s, x2)
        val = torch.stack((y1, y2), dim=2).reshape(M, N)
    else:
        val = acc
        if act == ACT_GELU:
            val = 0.5 * val * (1.0 + torch.erf(val / math.sqrt(2.0)))
        elif act == ACT_RELU:
            val = torch.relu(val)
    odt = torch.float32 if (out_f32 and epilogue == EPI_STORE) else td
    if out is None:
        return val.to(odt)
    out[_rows(out, M, o_rows)] = val.to(out.dtype)
    return out


def layernorm(x, gamma, beta, eps, *, dtype, out_t=None, out_f32=None, want_t=True, want_f32=False):
    y = F.layer_norm(x.float(), (x.shape[-1],), gamma.float(), beta.float(), eps)
    return (y.to(torch_dtype(dtype)) if want_t or out_t is not None else None), (y if want_f32 or out_f32 is not None else None)


def rmsnorm(x, gamma, eps, *, dtype, out_t=None, out_f32=None, want_t=True, want_f32=False):
    xf = x.float()
    y = gamma.float() * (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps))
    return (y.to(torch_dtype(dtype)) if want_t else None), (y if want_f32 else None)


def attention(q, k, v, *, B, H, Sq, Skv, D, scale, causal=False, kv_len=None, out=None, q_strides=None, k_strides=None,
              v_strides=None):
    def heads(t, S, strides):
        """element (b, s, h, d) lives at t.flat[b * batch_stride + s * row_stride + h * D + d] (include/stllm_hip.h)"""
        bs, rs = strides if strides is not None else (S * t.stride(0), t.stride(0))
        return torch.as_strided(t, (B, S, H, D), (bs, rs, D, 1), t.storage_offset()).float().transpose(1, 2)
    qh, kh, vh = heads(q, Sq, q_strides), heads(k, Skv, k_strides), heads(v, Skv, v_strides)
    s = (qh @ kh.transpose(-1, -2)) * scale
    if causal:
        s = s.masked_fill(torch.ones(Sq, Skv).triu(1).bool(), float("-inf"))
    if kv_len is not None:
        dead = torch.arange(Skv)[None, :] >= kv_len.long()[:, None]
        s = s.masked_fill(dead[:, None, None, :], float("-inf"))
    o = (s.softmax(-1) @ vh).transpose(1, 2).reshape(B * Sq, H * D).to(q.dtype)
    if out is not None:
        out.copy_(o)
        return out
    return o


def gather_rows(src_a, idx_a, *, src_b=None, add=None, idx_add=None, out=None, scale=1.0):
    idx = idx_a.long()
    res = torch.empty((idx.numel(), src_a.shape[-1]))
    pos = idx >= 0
    res[pos] = src_a[idx[pos]].float()
    if (~pos).any():
        res[~pos] = src_b[-idx[~pos] - 1].float()
    if add is not None:
        res = res + add[idx_add.long()].float()
    if scale != 1.0:
        res = res * scale
    if out is not None:
        out[: res.shape[0]] = res
        return out
    return res


def mean_t(x):
    assert x.shape[0] > 0 and x.shape[1] > 0, "stllm_mean_t: empty batch (the device binding indexes x[0, 0])"
    return x.float().mean(dim=1)


def vit_cls_rows(cls, pos, x, n_frames):
    x.view(n_frames, 257, -1)[:, 0] = cls + pos[0]


def cosine_rows(a, b, idx_a=None, idx_b=None, n_rows=None):
    aa = a if idx_a is None else a[idx_a.long()]
    bb = b if idx_b is None else b[idx_b.long()]
    n = n_rows if n_rows is not None else aa.shape[0]
    aa, bb = aa[:n].float(), bb[:n].float()
    return 2 - 2 * (F.normalize(aa, dim=-1) * F.normalize(bb, dim=-1)).sum(-1)


def cross_entropy_rows(logits, labels):
    return F.cross_entropy(logits.float(), labels.long(), ignore_index=-100, reduction="none")


def cast_rows(x, dtype, out=None):
    return x.to(torch_dtype(dtype))


def set_profiler(p):
    pass


def preprocess_frames(frames_u8, out=None):
    """stllm_preprocess_frames via the oracle (tests only): uint8 [T,H,W,3] -> f32 [T,3,224,224]"""
    import preprocess_oracle as P
    r = torch.from_numpy(P.video_transform(frames_u8.cpu().numpy())).view(-1, 3, 224, 224)
    if out is not None:
        out.copy_(r)
        return out
    return r


# ---- training entry points (contracts: include/stllm_hip.h "backward / optimizer") -----------------------------------
def transpose(x, *, pad=64, out=None):
    R, C = x.shape
    Rp = (R + pad - 1) // pad * pad
    o = torch.zeros((C, Rp), dtype=x.dtype)
    o[:, :R] = x.t()
    return o


def rmsnorm_bwd(x, gamma, eps, dy, dx, *, accumulate=True):
    xf, g = x.float(), gamma.float() * dy.float()
    r = torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)
    xh = xf * r
    d = r * (g - xh * (g * xh).mean(-1, keepdim=True))
    if accumulate:
        dx += d
    else:
        dx.copy_(d)
    return (dy.float() * xh).sum(0)


def layernorm_bwd(x, gamma, eps, dy, dx=None, *, accumulate=False):
    xf = x.float()
    mu = xf.mean(-1, keepdim=True)
    r = torch.rsqrt(xf.var(-1, unbiased=False, keepdim=True) + eps)
    xh = (xf - mu) * r
    g = gamma.float() * dy.float()
    d = r * (g - g.mean(-1, keepdim=True) - xh * (g * xh).mean(-1, keepdim=True))
    if dx is None:
        dx = d
    elif accumulate:
        dx += d
    else:
        dx.copy_(d)
    return dx, (dy.float() * xh).sum(0), dy.float().sum(0)


def swiglu(gu, out=None):
    M, N2 = gu.shape
    g = gu.float().view(M, N2 // 64, 2, 32)
    return (F.silu(g[:, :, 0]) * g[:, :, 1]).reshape(M, N2 // 2).to(gu.dtype)


def swiglu_bwd(gu, dg, out=None):
    M, N2 = gu.shape
    g = gu.float().view(M, N2 // 64, 2, 32)
    gate, up = g[:, :, 0], g[:, :, 1]
    d = dg.float().view(M, N2 // 64, 32)
    sg = torch.sigmoid(gate)
    dgate = d * up * sg * (1 + gate * (1 - sg))
    dup = d * gate * sg
    return torch.stack((dgate, dup), dim=2).reshape(M, N2).to(gu.dtype)


def rope_bwd(dqkv, cos, sin, *, rope_seq, rope_cols):
    M, N = dqkv.shape
    x = dqkv.float().view(M, N // 64, 2, 32)
    grp = torch.arange(N // 64)
    pos = torch.arange(M) % rope_seq
    fi = (grp % 2)[:, None] * 32 + torch.arange(32)[None, :]
    c, s = cos[pos][:, fi], sin[pos][:, fi]
    live = (grp * 64 < rope_cols)[None, :, None]
    d1, d2 = x[:, :, 0], x[:, :, 1]
    x1 = torch.where(live, d1 * c + d2 * s, d1)
    x2 = torch.where(live, d2 * c - d1 * s, d2)
    dqkv.copy_(torch.stack((x1, x2), dim=2).reshape(M, N).to(dqkv.dtype))
    return dqkv


def attention_bwd(q, k, v, o, do, dq, dk, dv, *, B, H, S=None, D, scale, causal=True, kv_len=None, Sq=None, Skv=None, strides=None,
                  kv_strides=None, do_strides=None, d_strides=None, dkv_strides=None):
    Sq = S if Sq is None else Sq
    Skv = S if Skv is None else Skv

    def heads(t, n, given):
        bs, rs = given if given is not None else (n * t.stride(0), t.stride(0))
        return torch.as_strided(t, (B, n, H, D), (bs, rs, D, 1), t.storage_offset())
    same = Sq == Skv
    ks = kv_strides if kv_strides is not None else (strides if same else None)
    dks = dkv_strides if dkv_strides is not None else (d_strides if same else None)
    qh = heads(q, Sq, strides).float().transpose(1, 2)
    kh, vh = (heads(t, Skv, ks).float().transpose(1, 2) for t in (k, v))
    doh = heads(do, Sq, do_strides).float().transpose(1, 2)
    s = (qh @ kh.transpose(-1, -2)) * scale
    if causal:
        s = s.masked_fill(torch.ones(Sq, Skv).triu(1).bool(), float("-inf"))
    if kv_len is not None:
        dead = torch.arange(Skv)[None, :] >= kv_len.long()[:, None]
        s = s.masked_fill(dead[:, None, None, :], float("-inf"))
    p = s.softmax(-1)
    gv = p.transpose(-1, -2) @ doh
    dp = doh @ vh.transpose(-1, -2)
    ds = p * (dp - (dp * p).sum(-1, keepdim=True))
    gq = (ds @ kh) * scale
    gk = (ds.transpose(-1, -2) @ qh) * scale
    heads(dq, Sq, d_strides).copy_(gq.transpose(1, 2).to(dq.dtype))
    heads(dk, Skv, dks).copy_(gk.transpose(1, 2).to(dk.dtype))
    heads(dv, Skv, dks).copy_(gv.transpose(1, 2).to(dv.dtype))
    return dq, dk, dv


def cross_entropy_bwd(logits, labels, scale, *, dtype, vocab=None):
    n, Vp = logits.shape
    V = Vp if vocab is None else vocab
    lab = labels.long()
    pr = logits[:, :V].float().softmax(-1)
    ok = lab >= 0
    pr[ok, lab[ok]] -= 1.0
    pr[~ok] = 0.0
    out = torch.zeros((n, Vp), dtype=torch_dtype(dtype))
    out[:, :V] = (pr * scale).to(out.dtype)
    return out


def scatter_add_rows(src, idx, dst_a, dst_b=None, scale=1.0):
    i = idx.long()
    pos = i >= 0
    dst_a.index_add_(0, i[pos], src[: i.numel()][pos].float() * scale)
    if (~pos).any():
        dst_b.index_add_(0, -i[~pos] - 1, src[: i.numel()][~pos].float() * scale)


def cosine_rows_bwd(a, b, idx_a=None, idx_b=None, n_rows=None, scale=1.0):
    aa = a if idx_a is None else a[idx_a.long()]
    bb = b if idx_b is None else b[idx_b.long()]
    n = n_rows if n_rows is not None else aa.shape[0]
    aa, bb = aa[:n].float(), bb[:n].float()
    na = aa.norm(dim=-1, keepdim=True)
    ah, bh = aa / na, F.normalize(bb, dim=-1)
    return -2.0 * scale * (bh - (ah * bh).sum(-1, keepdim=True) * ah) / na


def colsum(x):
    return x.float().sum(0)


def relu_bwd(dy, y):
    return dy * (y > 0).to(dy.dtype)


def gelu(x):
    xf = x.float()
    return (0.5 * xf * (1.0 + torch.erf(xf / math.sqrt(2.0)))).to(x.dtype)


def gelu_bwd(x, dy):
    xf = x.float()
    cdf = 0.5 * (1.0 + torch.erf(xf / math.sqrt(2.0)))
    return (dy.float() * (cdf + xf * torch.exp(-0.5 * xf * xf) / math.sqrt(2.0 * math.pi))).to(x.dtype)


def scale_rows(x, scale, *, rows_per_group=0, idx=None):
    g = idx.long() if idx is not None else torch.arange(x.shape[0]) // rows_per_group
    x.copy_((x.float() * scale.float()[g][:, None]).to(x.dtype))
    return x


def bcast_add_t(dst, src, scale):
    dst += scale * src.unsqueeze(1)


def adamw(p, g, m, v, *, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0, p16=None):
    gg = g * grad_scale
    p.mul_(1 - lr * weight_decay)
    m.mul_(beta1).add_(gg, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(gg, gg, value=1 - beta2)
    denom = (v.sqrt() / math.sqrt(1 - beta2 ** step)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / (1 - beta1 ** step))
    if p16 is not None:
        p16.copy_(p)


def sumsq(x, out=None):
    if out is None:
        out = torch.zeros(1)
    out += x.float().pow(2).sum()
    return out


def vit_block_array(blocks):
    return None


def vit_blocks(x, blocks, carr, *, n_seq, seq_len, num_heads, dtype):
    """stllm_vit_blocks: the block loop, one contract call per op (what the C entry point issues)"""
    from stllm_amd.models.eva_vit import block_forward
    for bp in blocks:
        block_forward(x, bp, n_seq, seq_len, num_heads, dtype)
    return x


def llama_layer_array(layers, cache=None):
    return None


class _LM:   # the bits of LlamaModel that prefill_layers_per_op reads
    def __init__(self, hidden, n_heads, eps):
        import types
        self.config = types.SimpleNamespace(hidden_size=hidden, num_attention_heads=n_heads, rms_norm_eps=eps)


def llama_layers(x, layers, carr, *, B, S, n_heads, eps, rope, dtype, kv_len=None, cache=None):
    from stllm_amd.models.llama import LlamaModel
    return LlamaModel.prefill_layers_per_op(_LM(x.shape[1], n_heads, eps), x, layers, B, S, rope[0], rope[1], kv_len, cache, dtype)


def qformer_layer_array(layers):
    return None


def qformer_layers(hq32, hq16, ht32, ht16, enc16, layers, carr, *, n_seq, n_query, n_text, n_heads, dtype, kv_len=None):
    """stllm_qformer_layers: the BertLayer loop, one contract call per op (what the C entry point issues)"""
    import types
    from stllm_amd.models.Qformer import BertModel
    shim = types.SimpleNamespace(config=types.SimpleNamespace(hidden_size=hq32.shape[1], num_attention_heads=n_heads))
    return BertModel.encode_layers_per_op(shim, layers, hq32, hq16, ht32, ht16, enc16, n_seq, n_query, n_text, kv_len, dtype)


@contextlib.contextmanager
def installed():
    """Monkey-patch stllm_amd.hip's compute entry points with the functions above (tests only)."""
    from stllm_amd import hip
    names = ["gemm", "layernorm", "rmsnorm", "attention", "gather_rows", "mean_t", "vit_cls_rows", "cosine_rows",
             "cross_entropy_rows", "cast_rows", "preprocess_frames", "transpose", "rmsnorm_bwd", "layernorm_bwd", "swiglu",
             "swiglu_bwd", "rope_bwd", "attention_bwd", "cross_entropy_bwd", "scatter_add_rows", "cosine_rows_bwd", "colsum",
             "relu_bwd", "gelu", "gelu_bwd", "scale_rows", "bcast_add_t", "adamw", "sumsq", "vit_block_array", "vit_blocks",
             "llama_layer_array", "llama_layers", "qformer_layer_array", "qformer_layers"]
    saved = {n: getattr(hip, n) for n in names}
    try:
        for n in names:
            setattr(hip, n, globals()[n])
        yield
    finally:
        for n, f in saved.items():
            setattr(hip, n, f)
