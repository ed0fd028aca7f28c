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


# =========================================================================================================================
# BT-Adapter (eva_btadapter.py:147-310): forward that keeps the branch's activations, and the branch's backward.
# The frozen ViT stream `h` is a constant input of the branch at every adapter layer (the backbone never reads the branch), so
# only the side branch is differentiated: 3 x (temporal block over T per patch, spatial block over the 257 tokens per frame).
# =========================================================================================================================
from .training import linear_bwd   # noqa: E402  (dgrad + wgrad of y = x @ w^T through stllm_gemm on transposed operands)


def btadapter_forward_taped(vit, x, drop=None):
    """EVAVisionTransformer_BTAdapter.forward_flat (models/eva_btadapter.py) keeping what the branch's backward needs.
    Returns (flat fp32 stream [(B*T)*257, 1408], tape).
    drop: train-mode stochastic depth of the adapter blocks (DropPath 0.1, eva_btadapter.py:259, 274, 280, 303), one dict per adapter
    layer {"t": f32 [B*256], "s": f32 [B*T], "o": f32 [B]} of per-sample factors keep / keep_prob (drop_path_factors draws them);
    None = the deterministic (eval-mode) step."""
    from .models.eva_vit import block_forward
    if x.ndim == 5:
        if x.shape[1] == 3:
            x = x.permute(0, 2, 1, 3, 4)
        B, T = x.shape[0], x.shape[1]
        x = x.reshape((-1,) + tuple(x.shape[2:]))
    elif x.ndim == 4:
        T, B = x.shape[0], 1
    else:
        raise ValueError("expected 4-D or 5-D input")
    vit.T = T
    dt = runtime.compute_dtype()
    pk, bt = vit.pack(dt), vit.pack_bt(dt)
    dev = x.device
    tb = vit._tables(B, T, dev)
    P, L, D, H = tb["P"], tb["L"], vit.embed_dim, vit.num_heads
    N, nbr = B * T, B * tb["P"] * T
    hd = D // H
    h = vit.embed_flat(x, pk, dt)
    br = None
    o_idx = torch.cat([torch.arange(nbr) // (P * T), torch.arange(B)]).to(torch.int32).to(dev) if drop is not None else None
    tape = dict(B=B, T=T, layers=[], tb=tb, drop=drop, o_idx=o_idx)
    for i, bp_ in enumerate(pk["blocks"]):
        block_forward(h, bp_, N, L, H, dt)
        if i < vit.num_layers - vit.depth:
            continue
        j = i + vit.depth - vit.num_layers
        rec = {}
        new = torch.empty((nbr + B, D), device=dev, dtype=torch.float32)
        cls_mean = vit._cls_mean(h, tb["cls_main"], B, T)
        if br is None:
            pt = hip.gather_rows(pk["pos"], (1 + torch.arange(P).view(P, 1).expand(P, T)).reshape(-1).to(torch.int32).to(dev),
                                 add=vit.BTAdapter_position.weight,
                                 idx_add=torch.arange(T).view(1, T).expand(P, T).reshape(-1).to(torch.int32).to(dev))
            hip.gather_rows(h, tb["main_of_bpt"], add=pt, idx_add=tb["pt_of_bpt"], out=new[:nbr])
            cls_br = hip.gather_rows(vit.BTAdapter_cls.view(1, D).float().contiguous(), tb["zeros_b"][:1], add=pk["pos"],
                                     idx_add=tb["zeros_b"][:1])
            hip.gather_rows(cls_mean, tb["arange_b"], add=cls_br, idx_add=tb["zeros_b"], out=new[nbr:], scale=0.5)
        else:
            hip.gather_rows(h, tb["main_of_bpt"], add=br, idx_add=tb["arange_bpt"], out=new[:nbr])
            hip.gather_rows(cls_mean, tb["arange_b"], add=br[nbr:], idx_add=tb["arange_b"], out=new[nbr:])
        br = new
        # ---- temporal block ----
        t_ = bt["T"][j]
        patches = br[:nbr]
        rec["t_in"] = patches.clone()
        rec["t_hn"], _ = hip.layernorm(patches, t_["n1w"], t_["n1b"], t_["e1"], dtype=dt)
        rec["t_qkv"] = hip.gemm(rec["t_hn"], t_["wqkv"], dtype=dt, bias=t_["bqkv"])
        q = rec["t_qkv"]
        rec["t_a"] = hip.attention(q[:, :D], q[:, D:2 * D], q[:, 2 * D:], B=B * P, H=H, Sq=T, Skv=T, D=hd, scale=hd ** -0.5)
        rec["t_pr"] = hip.gemm(rec["t_a"], t_["wproj"], dtype=dt, bias=t_["bproj"])
        if drop is not None:
            hip.scale_rows(rec["t_pr"], drop[j]["t"], rows_per_group=T)              # res_temporal = drop_path(attn(...)), per (b p)
        hip.gemm(rec["t_pr"], t_["wfc"], dtype=dt, epilogue=hip.EPI_RESID, bias=t_["bfc"], resid=patches)
        # ---- spatial block ----
        s_ = bt["S"][j]
        rec["s_x"] = hip.gather_rows(br, tb["sp_src"])
        rec["s_hn1"], _ = hip.layernorm(rec["s_x"], s_["n1w"], s_["n1b"], s_["e1"], dtype=dt)
        rec["s_qkv"] = hip.gemm(rec["s_hn1"], s_["wqkv"], dtype=dt, bias=s_["bqkv"])
        q = rec["s_qkv"]
        rec["s_a"] = hip.attention(q[:, :D], q[:, D:2 * D], q[:, 2 * D:], B=N, H=H, Sq=L, Skv=L, D=hd, scale=hd ** -0.5)
        res = hip.gemm(rec["s_a"], s_["wproj"], dtype=dt, bias=s_["bproj"], out_f32=True)
        if drop is not None:
            hip.scale_rows(res, drop[j]["s"], rows_per_group=L)                       # res_spatial = drop_path(attn(...)), per (b t)
        nxt = torch.empty_like(br)
        hip.gather_rows(res, tb["sp_patch_rows"], add=br, idx_add=tb["arange_bpt"], out=nxt[:nbr])
        cls_res = vit._cls_mean(res, tb["sp_cls_rows"], B, T)
        hip.gather_rows(cls_res, tb["arange_b"], add=br[nbr:], idx_add=tb["arange_b"], out=nxt[nbr:])
        br = nxt
        rec["m_in"] = br.clone()
        rec["m_hn"], _ = hip.layernorm(br, s_["n2w"], s_["n2b"], s_["e2"], dtype=dt)
        rec["m_raw"] = hip.gemm(rec["m_hn"], s_["wfc1"], dtype=dt, bias=s_["bfc1"])
        rec["m_g"] = hip.gelu(rec["m_raw"])
        hip.gemm(rec["m_g"], s_["wfc2"], dtype=dt, epilogue=hip.EPI_RESID, bias=s_["bfc2"], resid=br)
        if drop is not None:
            hip.scale_rows(br, drop[j]["o"], idx=o_idx)                               # x = drop_path(x): the whole block output, per b
        tape["layers"].append(rec)
    out = hip.gather_rows(br, tb["out_src"], add=h, idx_add=tb["arange_main"], scale=0.5)
    return out, tape


def _attn_block_bwd(grads, name, d_out16, a, qkv, hn, pk_w, pk_proj, Bn, H, S, D, dt):
    """y = proj(attention(hn @ Wqkv^T + [q_bias, 0, v_bias])): weight / bias gradients into `grads`, returns d(hn) (compute dtype)"""
    hd = D // H
    d_a, dw = linear_bwd(d_out16, a, pk_proj, dt)
    grads[name + "attn.proj.weight"], grads[name + "attn.proj.bias"] = dw, hip.colsum(d_out16)
    d_qkv = torch.empty_like(qkv)
    hip.attention_bwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], a, d_a, d_qkv[:, :D], d_qkv[:, D:2 * D], d_qkv[:, 2 * D:], B=Bn, H=H, S=S,
                      D=hd, scale=hd ** -0.5, causal=False)
    d_hn, dw = linear_bwd(d_qkv, hn, pk_w, dt)
    bsum = hip.colsum(d_qkv)
    grads[name + "attn.qkv.weight"] = dw
    grads[name + "attn.q_bias"], grads[name + "attn.v_bias"] = bsum[:D].clone(), bsum[2 * D:].clone()
    return d_hn


def btadapter_backward(vit, tape, d_out, prefix="model.stllm_model.visual_encoder."):
    """d_out: gradient w.r.t. forward_flat's output, f32 [(B*T)*257, D].  Returns {reference name: fp32 gradient} of every
    `BTAdapter*` parameter."""
    dt = runtime.compute_dtype()
    bt = vit.pack_bt(dt)
    tb, B, T = tape["tb"], tape["B"], tape["T"]
    P, L, D, H = tb["P"], tb["L"], vit.embed_dim, vit.num_heads
    N, nbr = B * T, B * P * T
    dev = d_out.device
    grads = {}
    d_br = torch.zeros((nbr + B, D), device=dev, dtype=torch.float32)
    hip.scatter_add_rows(d_out, tb["out_src"], d_br, scale=0.5)                    # out = (br[out_src] + h) / 2
    for j in range(vit.depth - 1, -1, -1):
        rec, s_, t_ = tape["layers"][j], bt["S"][j], bt["T"][j]
        sp, tp = f"{prefix}BTAdapter_S.{j}.", f"{prefix}BTAdapter_T.{j}."
        dj = tape["drop"][j] if tape["drop"] is not None else None
        if dj is not None:
            hip.scale_rows(d_br, dj["o"], idx=tape["o_idx"])
        # ---- MLP: br += fc2(gelu(fc1(LN2(br)))) -----------------------------------------------------------------------------
        d16 = hip.cast_rows(d_br, dt)
        d_g, dw = linear_bwd(d16, rec["m_g"], s_["wfc2"], dt)
        grads[sp + "mlp.fc2.weight"], grads[sp + "mlp.fc2.bias"] = dw, hip.colsum(d16)
        d_raw = hip.gelu_bwd(rec["m_raw"], d_g)
        d_hn, dw = linear_bwd(d_raw, rec["m_hn"], s_["wfc1"], dt)
        grads[sp + "mlp.fc1.weight"], grads[sp + "mlp.fc1.bias"] = dw, hip.colsum(d_raw)
        _, grads[sp + "norm2.weight"], grads[sp + "norm2.bias"] = hip.layernorm_bwd(rec["m_in"], s_["n2w"], s_["e2"], d_hn, d_br, accumulate=True)
        # ---- spatial attention: nxt = res[patch rows] + br ; CLS: mean_t(res[cls rows]) + br --------------------------------------
        d_res = torch.zeros((N * L, D), device=dev, dtype=torch.float32)
        hip.scatter_add_rows(d_br, tb["sp_patch_rows"], d_res)                     # rows [:nbr] of d_br
        d_cls = torch.zeros((B, T, D), device=dev, dtype=torch.float32)
        hip.bcast_add_t(d_cls, d_br[nbr:].contiguous(), 1.0 / T)
        hip.scatter_add_rows(d_cls.view(B * T, D), tb["sp_cls_rows"], d_res)
        if dj is not None:
            hip.scale_rows(d_res, dj["s"], rows_per_group=L)
        d_res16 = hip.cast_rows(d_res, dt)
        d_hn = _attn_block_bwd(grads, sp, d_res16, rec["s_a"], rec["s_qkv"], rec["s_hn1"], s_["wqkv"], s_["wproj"], N, H, L, D, dt)
        d_sx, grads[sp + "norm1.weight"], grads[sp + "norm1.bias"] = hip.layernorm_bwd(rec["s_x"], s_["n1w"], s_["e1"], d_hn)
        hip.scatter_add_rows(d_sx, tb["sp_src"], d_br)                             # sx = br[sp_src]
        # ---- temporal block on the patch rows: p += fc(proj(attention(LN(p)))) --------------------------------------------------
        d_p = d_br[:nbr]
        d_p16 = hip.cast_rows(d_p, dt)
        d_pr, dw = linear_bwd(d_p16, rec["t_pr"], t_["wfc"], dt)
        grads[tp + "temporal_fc.weight"], grads[tp + "temporal_fc.bias"] = dw, hip.colsum(d_p16)
        if dj is not None:
            hip.scale_rows(d_pr, dj["t"], rows_per_group=T)
        d_hn = _attn_block_bwd(grads, tp, d_pr, rec["t_a"], rec["t_qkv"], rec["t_hn"], t_["wqkv"], t_["wproj"], B * P, H, T, D, dt)
        _, grads[tp + "norm1.weight"], grads[tp + "norm1.bias"] = hip.layernorm_bwd(rec["t_in"], t_["n1w"], t_["e1"], d_hn, d_p, accumulate=True)
        # ---- layer input: j > 0: new = gather(h) + previous branch (identity);  j == 0: init_input ------------------------------------
    d_pt = torch.zeros((P * T, D), device=dev, dtype=torch.float32)
    hip.scatter_add_rows(d_br, tb["pt_of_bpt"], d_pt)                              # new[:nbr] = h[...] + (pos[1+p] + position[t])[p*T+t]
    d_pos = torch.zeros_like(vit.BTAdapter_position.weight, dtype=torch.float32)
    hip.scatter_add_rows(d_pt, torch.arange(T).view(1, T).expand(P, T).reshape(-1).to(torch.int32).to(dev), d_pos)
    grads[prefix + "BTAdapter_position.weight"] = d_pos
    grads[prefix + "BTAdapter_cls"] = (hip.colsum(d_br[nbr:].contiguous()) * 0.5).view(1, 1, D)   # new[nbr:] = (cls_mean + cls + pos[0]) / 2
    return grads


def drop_path_factors(B, T, depth=3, drop_prob=0.1, generator=None, device="cpu", patches=256):
    """timm.layers.drop_path's per-sample factors for one training step of the adapter (bernoulli(keep) / keep, scale_by_keep) at its
    three call sites per layer: temporal residual per (b, patch), spatial residual per (b, t), block output per b."""
    keep = 1.0 - drop_prob

    def draw(n):
        return (torch.bernoulli(torch.full((n,), keep), generator=generator) / keep).to(torch.float32).to(device)
    return [dict(t=draw(B * patches), s=draw(B * T), o=draw(B)) for _ in range(depth)]
