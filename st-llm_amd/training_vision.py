"""Backward THROUGH the frozen vision stack, for the BT-Adapter's trainable parameters.

The reference trains `visual_encoder.BTAdapter*` on 4 of its 5 shipped configs (st_llm.py:257-261) while ViT, ln_vision and the
Q-Former stay frozen: autograd still has to carry the gradient from the projected tokens back through the Q-Former
(Qformer.py:402-484: self-attention over [queries | text], cross-attention to the image tokens on even layers, two FFN streams)
and ln_vision (blip2.py:103-109) to the adapter's output.  Frozen weights get no gradient, so this is a dgrad-only sweep: per
layer the LayerNorm inputs, the fused QKV / attention outputs and the raw FFN pre-activations are kept, and every linear layer's
backward is one `stllm_gemm` on the transposed weight, accumulated into the fp32 stream gradient by the RESID epilogue.

  qformer_forward_taped / qformer_backward : d(loss)/d(query-row output) -> d(loss)/d(image tokens entering cross-attention)
"""
import math

import torch

from . import hip, runtime


class QFormerTape:
    def __init__(self):
        self.layers = []
        self.n = self.Q = self.Lt = self.P = 0
        self.kv_len = None


def _post_ln_taped(ctx, dense, resid32, dt, rec, key, **rows):
    """models/Qformer.py:_post_ln, keeping the LayerNorm input"""
    tmp = torch.empty_like(resid32)
    hip.gemm(ctx, dense["w"], dtype=dt, epilogue=hip.EPI_RESID, bias=dense["b"], resid=resid32, out=tmp, **rows)
    rec[key] = tmp
    h16, h32 = hip.layernorm(tmp, dense["g"], dense["beta"], dense["eps"], dtype=dt, want_f32=True)
    return h32, h16


def qformer_forward_taped(bert, query_tokens, enc16, n, input_ids=None, text_mask=None):
    """BertModel.encode (models/Qformer.py) with the activations its backward needs.  Returns (hq32 [n*Q, C], hq16, tape).
    The FFN GELU runs as its own kernel on stored pre-activations (the forward's fused epilogue does not keep them)."""
    cfg = bert.config
    dt = runtime.compute_dtype()
    layers = bert.pack(dt)
    dev = enc16.device
    C, H, Q = cfg.hidden_size, cfg.num_attention_heads, query_tokens.shape[0]
    P = enc16.shape[0] // n
    emb = bert.embeddings
    t = QFormerTape()
    q_idx = torch.arange(Q, dtype=torch.int32).repeat(n).to(dev)
    q_emb = hip.gather_rows(query_tokens.float().contiguous(), q_idx)
    hq16, hq32 = hip.layernorm(q_emb, emb.LayerNorm.weight, emb.LayerNorm.bias, emb.LayerNorm.eps, dtype=dt, want_f32=True)
    Lt, ht32, ht16, kv_len = 0, None, None, None
    if input_ids is not None:
        Lt = input_ids.shape[1]
        ids = input_ids.reshape(-1).to(torch.int64)
        w_idx = (-(ids) - 1).to(torch.int32).to(dev)
        p_idx = torch.arange(Lt, dtype=torch.int32).repeat(n).to(dev)
        t_emb = hip.gather_rows(emb.word_embeddings.weight, w_idx, src_b=emb.word_embeddings.weight,
                                add=emb.position_embeddings.weight, idx_add=p_idx)
        ht16, ht32 = hip.layernorm(t_emb, emb.LayerNorm.weight, emb.LayerNorm.bias, emb.LayerNorm.eps, dtype=dt, want_f32=True)
        kv_len = (Q + text_mask.long().sum(dim=1)).to(torch.int32).to(dev)
    t.n, t.Q, t.Lt, t.P, t.kv_len = n, Q, Lt, P, kv_len
    S = Q + Lt
    rows_q = dict(M=n * Q, o_rows=(Q, S * 3 * C)) if Lt else {}
    rows_t = dict(M=n * Lt, o_rows=(Lt, S * 3 * C)) if Lt else {}
    hd = C // H
    for pk in layers:
        rec = {}
        qkv = torch.empty((n * S, 3 * C), device=dev, dtype=dt)
        hip.gemm(hq16, pk["wqkv"], dtype=dt, bias=pk["bqkv"], out=qkv, **rows_q)
        if Lt:
            hip.gemm(ht16, pk["wqkv"], dtype=dt, bias=pk["bqkv"], out=qkv[Q:], **rows_t)
        ctx = hip.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], B=n, H=H, Sq=S, Skv=S, D=hd, scale=1.0 / math.sqrt(hd),
                            kv_len=kv_len)
        rec.update(qkv=qkv, ctx=ctx)
        a_q = dict(M=n * Q, a_rows=(Q, S * C)) if Lt else {}
        hq32, hq16 = _post_ln_taped(ctx, pk["attn_out"], hq32, dt, rec, "tmp_attn_q", **a_q)
        if Lt:
            ht32, ht16 = _post_ln_taped(ctx[Q:], pk["attn_out"], ht32, dt, rec, "tmp_attn_t", M=n * Lt, a_rows=(Lt, S * C))
        if "cq_w" in pk:
            cq = hip.gemm(hq16, pk["cq_w"], dtype=dt, bias=pk["cq_b"])
            ckv = hip.gemm(enc16, pk["ckv_w"], dtype=dt, bias=pk["ckv_b"])
            cctx = hip.attention(cq, ckv[:, :C], ckv[:, C:], B=n, H=H, Sq=Q, Skv=P, D=hd, scale=1.0 / math.sqrt(hd))
            rec.update(cq=cq, ckv=ckv, cctx=cctx)
            hq32, hq16 = _post_ln_taped(cctx, pk["cross_out"], hq32, dt, rec, "tmp_cross")
        f = pk["ffn_q"]
        rec["raw_q"] = hip.gemm(hq16, f["w1"], dtype=dt, bias=f["b1"])
        hq32, hq16 = _post_ln_taped(hip.gelu(rec["raw_q"]), f["out"], hq32, dt, rec, "tmp_ffn_q")
        if Lt:
            f = pk["ffn_t"]
            rec["raw_t"] = hip.gemm(ht16, f["w1"], dtype=dt, bias=f["b1"])
            ht32, ht16 = _post_ln_taped(hip.gelu(rec["raw_t"]), f["out"], ht32, dt, rec, "tmp_ffn_t")
        t.layers.append(rec)
    return hq32, hq16, t


def _kgran(dt):
    return 32 if dt == torch.float32 else 64


def _dgrad(dy, w, dt, **kw):
    """dy @ w for a frozen y = x @ w^T (no weight gradient): one GEMM on the transposed weight"""
    return hip.gemm(dy, hip.transpose(w, pad=_kgran(dt)), dtype=dt, **kw)


def _post_ln_bwd(d_h, tmp, dense, dt):
    """h = LN(tmp), tmp = ctx @ W^T + b + h_in  ->  (d_tmp f32 = d_h_in, d_tmp in the compute dtype for the dgrad GEMM)"""
    d_tmp, _, _ = hip.layernorm_bwd(tmp, dense["g"], dense["eps"], d_h)
    return d_tmp, hip.cast_rows(d_tmp, dt)


def qformer_backward(bert, tape, d_hq32):
    """d_hq32: gradient w.r.t. the query rows of last_hidden_state, f32 [n*Q, C].  Returns d(enc) f32 [n*P, encoder_width]: the
    gradient w.r.t. the (ln_vision'd) image tokens, which enter only through the cross-attention K/V projections."""
    cfg = bert.config
    dt = runtime.compute_dtype()
    layers = bert.pack(dt)
    n, Q, Lt, P = tape.n, tape.Q, tape.Lt, tape.P
    C, H = cfg.hidden_size, cfg.num_attention_heads
    hd = C // H
    S = Q + Lt
    dev = d_hq32.device
    d_hq = d_hq32.float().clone()
    d_ht = torch.zeros((n * Lt, C), device=dev, dtype=torch.float32) if Lt else None
    d_enc = torch.zeros((n * P, cfg.encoder_width), device=dev, dtype=torch.float32)
    for li in range(len(layers) - 1, -1, -1):
        pk, rec = layers[li], tape.layers[li]
        # ---- FFN streams ---------------------------------------------------------------------------------------------
        for d_h, f, raw, tmp in ((d_hq, pk["ffn_q"], rec["raw_q"], rec["tmp_ffn_q"]),) + \
                (((d_ht, pk["ffn_t"], rec["raw_t"], rec["tmp_ffn_t"]),) if Lt else ()):
            d_tmp, d_tmp16 = _post_ln_bwd(d_h, tmp, f["out"], dt)
            d_raw = hip.gelu_bwd(raw, _dgrad(d_tmp16, f["out"]["w"], dt))
            d_h.copy_(d_tmp)
            _dgrad(d_raw, f["w1"], dt, epilogue=hip.EPI_RESID, resid=d_h)
        # ---- cross-attention (query rows, even layers) -----------------------------------------------------------------------
        if "cq_w" in pk:
            d_tmp, d_tmp16 = _post_ln_bwd(d_hq, rec["tmp_cross"], pk["cross_out"], dt)
            d_cctx = _dgrad(d_tmp16, pk["cross_out"]["w"], dt)
            ckv = rec["ckv"]
            d_cq = torch.empty_like(rec["cq"])
            d_ckv = torch.empty_like(ckv)
            hip.attention_bwd(rec["cq"], ckv[:, :C], ckv[:, C:], rec["cctx"], d_cctx, d_cq, d_ckv[:, :C], d_ckv[:, C:], B=n, H=H, Sq=Q,
                              Skv=P, D=hd, scale=1.0 / math.sqrt(hd), causal=False)
            d_hq.copy_(d_tmp)
            _dgrad(d_cq, pk["cq_w"], dt, epilogue=hip.EPI_RESID, resid=d_hq)
            _dgrad(d_ckv, pk["ckv_w"], dt, epilogue=hip.EPI_RESID, resid=d_enc)
        # ---- self-attention over [queries | text] -------------------------------------------------------------------------------
        d_ctx = torch.empty((n * S, C), device=dev, dtype=dt)
        d_tmp, d_tmp16 = _post_ln_bwd(d_hq, rec["tmp_attn_q"], pk["attn_out"], dt)
        d_hq.copy_(d_tmp)
        _dgrad(d_tmp16, pk["attn_out"]["w"], dt, out=d_ctx, **(dict(M=n * Q, o_rows=(Q, S * C)) if Lt else {}))
        if Lt:
            d_tmp, d_tmp16 = _post_ln_bwd(d_ht, rec["tmp_attn_t"], pk["attn_out"], dt)
            d_ht.copy_(d_tmp)
            _dgrad(d_tmp16, pk["attn_out"]["w"], dt, out=d_ctx[Q:], M=n * Lt, o_rows=(Lt, S * C))
        qkv = rec["qkv"]
        d_qkv = torch.empty_like(qkv)
        hip.attention_bwd(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], rec["ctx"], d_ctx, d_qkv[:, :C], d_qkv[:, C:2 * C], d_qkv[:, 2 * C:],
                          B=n, H=H, S=S, D=hd, scale=1.0 / math.sqrt(hd), causal=False, kv_len=tape.kv_len)
        _dgrad(d_qkv, pk["wqkv"], dt, epilogue=hip.EPI_RESID, resid=d_hq, **(dict(M=n * Q, a_rows=(Q, S * 3 * C)) if Lt else {}))
        if Lt:
            _dgrad(d_qkv[Q:], pk["wqkv"], dt, epilogue=hip.EPI_RESID, resid=d_ht, M=n * Lt, a_rows=(Lt, S * 3 * C))
    return d_enc
