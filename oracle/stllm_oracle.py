"""CPU oracle for the ST-LLM video-token hot path — TEST INFRASTRUCTURE ONLY.

This file is the checker, never the product: only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import it.  The product path
(``stllm_amd.*``) never imports anything from ``oracle/`` and fails loudly when the HIP
library is missing.

It is a plain-PyTorch **fp32, CPU** restatement (own code, functional style, no nn.Module)
of the arithmetic of the reference path named by BASELINE.json:north_star.  Every function
cites the reference lines it follows (paths relative to /root/reference).  All tensors are
keyed by the *reference's parameter names* (SURVEY.md Appendix D), so a reference
``state_dict()`` can be fed in unchanged.

Parity pin: the reference has no tests / golden vectors of its own for this path
 — "parity unpinned" by the reference's own tests.  The pin used instead:
``tests/golden/make_fixtures.py`` imports the reference's model code in the build container,
runs it on seeded inputs and commits input/output vectors under ``tests/golden/*.npz``;
``tests/test_oracle_vs_golden.py`` checks every function below against those vectors
(fp32, ≤1e-5 relative).  The Llama arithmetic is third-party (HF ``transformers``, pinned
4.28.0 by the reference ``requirement.txt:32``; 5.15.0 is what is installed and what the
fixtures were generated with — identical math in fp32); the in-tree statement
of the same math is ``stllm/models/modeling_llama_mem.py:61-144``.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


# =====================================================================================
# small helpers
# =====================================================================================
def _lin(x: Tensor, sd: SD, name: str, bias: bool = True) -> Tensor:
    b = sd.get(name + ".bias") if bias else None
    return F.linear(x, sd[name + ".weight"], b)


def layer_norm(x: Tensor, sd: SD, name: str, eps: float) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], eps)


def gelu(x: Tensor) -> Tensor:
    """nn.GELU() / ACT2FN['gelu'] — exact erf form (eva_vit.py:45,154; Qformer.py:353-356)."""
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


# =====================================================================================
# EVA-CLIP-g ViT  (stllm/models/eva_vit.py)
# =====================================================================================
VIT_DIM, VIT_HEADS, VIT_HEAD_DIM, VIT_MLP, VIT_PATCH, VIT_TOKENS = 1408, 16, 88, 6144, 14, 257


def vit_patch_embed(x: Tensor, sd: SD, p: str) -> Tensor:
    """PatchEmbed.forward eva_vit.py:198-204: Conv2d(3,1408,k14,s14) == a [N*256,588]x[588,1408]
    GEMM over (c,dy,dx)-ordered patches; output patches row-major over (h,w)."""
    N, C, H, W = x.shape
    assert H == 224 and W == 224, "Input image size doesn't match model (eva_vit.py:201)"
    ph = H // VIT_PATCH
    w = sd[p + "patch_embed.proj.weight"]  # [1408,3,14,14]
    cols = x.reshape(N, C, ph, VIT_PATCH, ph, VIT_PATCH).permute(0, 2, 4, 1, 3, 5)  # N,h,w,c,dy,dx
    cols = cols.reshape(N, ph * ph, C * VIT_PATCH * VIT_PATCH)
    return cols @ w.reshape(w.shape[0], -1).t() + sd[p + "patch_embed.proj.bias"]


def vit_attention(x: Tensor, sd: SD, p: str, num_heads: int = VIT_HEADS) -> Tensor:
    """Attention.forward eva_vit.py:118-148.  qkv bias = cat(q_bias, 0, v_bias) (:120-124);
    q scaled by head_dim^-0.5 BEFORE q@k^T (:128); no rel-pos bias for eva_clip_g (:270-273)."""
    B, N, C = x.shape
    qb, vb = sd[p + "q_bias"], sd[p + "v_bias"]
    qkv = F.linear(x, sd[p + "qkv.weight"], torch.cat((qb, torch.zeros_like(vb), vb)))
    qkv = qkv.reshape(B, N, 3, num_heads, -1).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    q = q * (q.shape[-1] ** -0.5)
    attn = (q @ k.transpose(-2, -1)).softmax(dim=-1)
    out = (attn @ v).transpose(1, 2).reshape(B, N, -1)
    return _lin(out, sd, p + "proj")


def vit_mlp(x: Tensor, sd: SD, p: str) -> Tensor:
    """Mlp.forward eva_vit.py:54-61."""
    return _lin(gelu(_lin(x, sd, p + "fc1")), sd, p + "fc2")


def vit_block(x: Tensor, sd: SD, p: str, eps: float = 1e-6) -> Tensor:
    """Block.forward eva_vit.py:173-180 with gamma_1/2 None (init_values None, :416-428)."""
    x = x + vit_attention(layer_norm(x, sd, p + "norm1", eps), sd, p + "attn.")
    x = x + vit_mlp(layer_norm(x, sd, p + "norm2", eps), sd, p + "mlp.")
    return x


def vit_embed(x: Tensor, sd: SD, p: str) -> Tensor:
    """forward_features eva_vit.py:325-332: patch embed, CLS concat, + pos_embed."""
    t = vit_patch_embed(x, sd, p)
    cls = sd[p + "cls_token"].expand(t.shape[0], -1, -1)
    return torch.cat((cls, t), dim=1) + sd[p + "pos_embed"]


def vit_depth(sd: SD, p: str) -> int:
    d = 0
    while f"{p}blocks.{d}.norm1.weight" in sd:
        d += 1
    return d


def vit_forward(x: Tensor, sd: SD, p: str = "") -> Tensor:
    """VisionTransformer.forward_features eva_vit.py:324-340 (no final norm, :284,341)."""
    h = vit_embed(x, sd, p)
    for i in range(vit_depth(sd, p)):
        h = vit_block(h, sd, f"{p}blocks.{i}.")
    return h


def ln_vision(x: Tensor, sd: SD, p: str = "ln_vision") -> Tensor:
    """blip2.LayerNorm blip2.py:103-109: fp32 LayerNorm(1408), default eps 1e-5 (built :73)."""
    return layer_norm(x.float(), sd, p, 1e-5).to(x.dtype)


# =====================================================================================
# BT-Adapter visual backbone  (stllm/models/eva_btadapter.py)
# =====================================================================================
def _drop(x: Tensor, m: Optional[Tensor]) -> Tensor:
    """DropPath (eva_vit.py:30-38 -> timm.layers.drop_path, not in the reference tree; timm's published algorithm): per-sample
    x * bernoulli(keep) / keep.  Here the per-sample factor m (0 or 1 / keep_prob) is INJECTED, like the dynamic mask; None = eval."""
    return x if m is None else x * m.view((-1,) + (1,) * (x.ndim - 1))


def _bt_temp(x: Tensor, T: int, sd: SD, p: str, drop: Optional[Tensor] = None) -> Tensor:
    """BTAdapter_Temp.forward eva_btadapter.py:295-310 (norm eps 1e-6, :284).  drop: [b * patches] factors (:303)."""
    residual = x[:, 1:, :]
    cls = x[:, :1, :]
    b, pt, m = residual.shape
    pch = pt // T
    h = residual.reshape(b * pch, T, m)
    h = _drop(vit_attention(layer_norm(h, sd, p + "norm1", 1e-6), sd, p + "attn."), drop)
    h = _lin(h, sd, p + "temporal_fc")
    h = h.reshape(b, pch * T, m) + residual
    return torch.cat((cls, h), 1)


def _bt_spatial(x: Tensor, T: int, sd: SD, p: str, drop_s: Optional[Tensor] = None, drop_o: Optional[Tensor] = None) -> Tensor:
    """BTAdapter_Spatial.forward eva_btadapter.py:261-281.  NOTE: built through Block's default
    norm_layer=nn.LayerNorm (eva_btadapter.py:259, eva_vit.py:154) => eps **1e-5**, unlike the
    ViT blocks (1e-6) whose weights it clones (:89-99)."""
    residual = x
    cls0 = x[:, :1, :]
    q = x[:, 1:, :]
    b, pt, m = q.shape
    pch = pt // T
    cls = cls0.unsqueeze(1).repeat(1, T, 1, 1).reshape(b * T, 1, m)
    q = q.reshape(b, pch, T, m).permute(0, 2, 1, 3).reshape(b * T, pch, m)  # 'b (p t) m -> (b t) p m'
    h = torch.cat((cls, q), 1)
    h = _drop(vit_attention(layer_norm(h, sd, p + "norm1", 1e-5), sd, p + "attn."), drop_s)       # :274, per (b t) sample
    cls = h[:, :1, :].reshape(b, T, 1, m).mean(1)
    rs = h[:, 1:, :].reshape(b, T, pch, m).permute(0, 2, 1, 3).reshape(b, pch * T, m)  # '(b t) p m -> b (p t) m'
    x = residual + torch.cat((cls, rs), 1)
    x = x + vit_mlp(layer_norm(x, sd, p + "norm2", 1e-5), sd, p + "mlp.")
    return _drop(x, drop_o)                                                                           # :280, the whole block output per b


def btadapter_forward(x: Tensor, sd: SD, p: str = "", adapter_depth: int = 3,
                      return_branches: bool = False, drop=None):
    """EVAVisionTransformer_BTAdapter.forward/forward_features/forward_branch/init_input
    eva_btadapter.py:147-255.  x: [B,T,3,224,224] or 4-D [T,3,224,224] (B=1).
    drop: train-mode stochastic depth, one dict per adapter layer {"t": [B*256], "s": [B*T], "o": [B]} of injected factors."""
    if x.ndim == 5:
        if x.shape[1] == 3:  # reference quirk (:235-237): dim-1 == 3 is read as B,C,T,H,W
            x = x.permute(0, 2, 1, 3, 4)
        B, T = x.shape[0], x.shape[1]
        x = x.reshape((-1,) + tuple(x.shape[2:]))
    else:
        T, B = x.shape[0], 1
    depth = vit_depth(sd, p)
    h = vit_embed(x, sd, p)
    branch = None
    branches = []
    for i in range(depth):
        h = vit_block(h, sd, f"{p}blocks.{i}.")
        if i >= depth - adapter_depth:
            j = i + adapter_depth - depth
            xb = h.reshape(B, T, h.shape[1], h.shape[2])  # '(b t) l d -> b t l d'
            if branch is not None:  # forward_branch :188-196
                cls_b = xb[:, :, 0].mean(dim=1).unsqueeze(1)
                xp = xb[:, :, 1:].permute(0, 2, 1, 3).reshape(B, -1, h.shape[2])  # 'b t l d -> b (l t) d'
                xb = torch.cat((cls_b, xp), dim=1) + branch
            if j == 0:  # init_input :209-231
                cls_x = xb[:, :, 0].mean(dim=1).unsqueeze(1)
                xp = xb[:, :, 1:, :]
                b, t, l, d = xp.shape
                xp = xp.reshape(b * t, l, d)
                cls_br = sd[p + "BTAdapter_cls"].expand(xp.shape[0], 1, -1)
                xp = torch.cat((cls_br, xp), dim=1) + sd[p + "pos_embed"]
                cls_br = xp[:b, 0, :].unsqueeze(1)
                xp = xp[:, 1:, :].reshape(b, t, l, d).permute(0, 2, 1, 3).reshape(b * l, t, d)  # '(b t) l d -> (b l) t d'
                xp = xp + sd[p + "BTAdapter_position.weight"][:t]
                xp = xp.reshape(b, l * t, d)  # '(b l) t d -> b (l t) d'
                xb = torch.cat(((cls_x + cls_br) / 2, xp), dim=1)
            dj = drop[j] if drop is not None else {}
            xb = _bt_temp(xb, T, sd, f"{p}BTAdapter_T.{j}.", dj.get("t"))
            xb = _bt_spatial(xb, T, sd, f"{p}BTAdapter_S.{j}.", dj.get("s"), dj.get("o"))
            branch = xb
            branches.append(xb)
    pch = h.shape[1] - 1
    br_cls, br_patch = branch[:, 0], branch[:, 1:]
    br_patch = br_patch.reshape(B, pch, T, -1).permute(0, 2, 1, 3).reshape(B * T, pch, -1)  # 'b (p t) m -> (b t) p m'
    br_cls = br_cls.repeat(1, T).view(br_cls.shape[0] * T, -1).unsqueeze(1)
    out = (h + torch.cat((br_cls, br_patch), dim=1)) / 2
    return (out, branches) if return_branches else out


# =====================================================================================
# Q-Former  (stllm/models/Qformer.py), encoder forward only
# =====================================================================================
QF_DIM, QF_HEADS, QF_EPS = 768, 12, 1e-12


def _bert_attn(hq: Tensor, hkv: Tensor, add_mask: Optional[Tensor], sd: SD, p: str) -> Tensor:
    """BertSelfAttention.forward Qformer.py:169-275 (+BertSelfOutput :285-289 by the caller).
    scores / sqrt(64) AFTER q@k^T (:244), additive mask (:247)."""
    B, Sq, _ = hq.shape
    H = QF_HEADS
    q = _lin(hq, sd, p + "query").view(B, Sq, H, -1).permute(0, 2, 1, 3)
    k = _lin(hkv, sd, p + "key").view(B, hkv.shape[1], H, -1).permute(0, 2, 1, 3)
    v = _lin(hkv, sd, p + "value").view(B, hkv.shape[1], H, -1).permute(0, 2, 1, 3)
    s = (q @ k.transpose(-1, -2)) / math.sqrt(q.shape[-1])
    if add_mask is not None:
        s = s + add_mask
    ctx = s.softmax(dim=-1) @ v
    return ctx.permute(0, 2, 1, 3).reshape(B, Sq, -1)


def _bert_out(ctx: Tensor, inp: Tensor, sd: SD, p: str) -> Tensor:
    """BertSelfOutput / BertOutput: LayerNorm(dense(x) + input) (Qformer.py:285-289, 371-375)."""
    return layer_norm(_lin(ctx, sd, p + "dense") + inp, sd, p + "LayerNorm", QF_EPS)


def bert_layer(h: Tensor, add_mask: Optional[Tensor], enc: Tensor, i: int, sd: SD, p: str,
               query_length: int) -> Tensor:
    """BertLayer.forward Qformer.py:402-484; cross-attn on even layers (:386-395), only for the
    first `query_length` rows (:430-444); query rows use intermediate_query/output_query,
    text rows intermediate/output (:449-462)."""
    lp = f"{p}encoder.layer.{i}."
    a = _bert_out(_bert_attn(h, h, add_mask, sd, lp + "attention.self."), h, sd, lp + "attention.output.")
    qa = a[:, :query_length]
    if i % 2 == 0:
        # encoder mask is all-ones => additive 0 (Qformer.py:918-926)
        qa = _bert_out(_bert_attn(qa, enc, None, sd, lp + "crossattention.self."), qa, sd,
                       lp + "crossattention.output.")
    out = _bert_out(gelu(_lin(qa, sd, lp + "intermediate_query.dense")), qa, sd, lp + "output_query.")
    if a.shape[1] > query_length:
        ta = a[:, query_length:]
        out_t = _bert_out(gelu(_lin(ta, sd, lp + "intermediate.dense")), ta, sd, lp + "output.")
        out = torch.cat([out, out_t], dim=1)
    return out


def qformer_num_layers(sd: SD, p: str) -> int:
    n = 0
    while f"{p}encoder.layer.{n}.attention.self.query.weight" in sd:
        n += 1
    return n


def qformer_forward(query_embeds: Tensor, enc: Tensor, sd: SD, p: str = "Qformer.bert.",
                    input_ids: Optional[Tensor] = None, attention_mask: Optional[Tensor] = None) -> Tensor:
    """BertModel.forward Qformer.py:804-965 as called by st_llm.py:353-367.
    query_embeds [N,32,768]; enc = ln_vision'd image tokens [N,257,1408];
    input_ids [N,Lt] (optional); attention_mask [N,32+Lt] (1 = attend).  Returns last_hidden_state."""
    ql = query_embeds.shape[1]
    if input_ids is not None:  # BertEmbeddings.forward :95-102 (queries get no position emb)
        Lt = input_ids.shape[1]
        e = sd[p + "embeddings.word_embeddings.weight"][input_ids] + \
            sd[p + "embeddings.position_embeddings.weight"][:Lt]
        emb = torch.cat((query_embeds, e), dim=1)
    else:
        emb = query_embeds
    h = layer_norm(emb, sd, p + "embeddings.LayerNorm", QF_EPS)
    add_mask = None
    if attention_mask is not None:  # get_extended_attention_mask :785,801
        add_mask = (1.0 - attention_mask[:, None, None, :].to(h.dtype)) * -10000.0
    for i in range(qformer_num_layers(sd, p)):
        h = bert_layer(h, add_mask, enc, i, sd, p, ql)
    return h


# =====================================================================================
# STLLMModel: encode_img, pooling, masking, token-block assembly  (stllm/models/st_llm.py)
# =====================================================================================
def encode_img(image: Tensor, sd: SD, p: str = "", vit_model: str = "eva_clip_g",
               text_ids: Optional[Tensor] = None, text_mask: Optional[Tensor] = None,
               adapter_depth: int = 3, has_qformer: bool = True) -> Tensor:
    """STLLMModel.encode_img st_llm.py:321-377.
    image: 5-D [B,T,3,224,224] -> [B,T,32,4096]; 4-D [T,3,224,224] (inference) -> [T,32,4096].
    text_ids/text_mask: BERT ids already repeated per frame, [N,Lt] (st_llm.py:337-350).
    has_qformer=False (st_llm.py:369-373): no Q-Former — the 256 patch tokens (CLS dropped) are viewed as 64 rows of 4 concatenated
    tokens, [N, 64, 5632], and projected by llama_proj(5632 -> 4096): 64 tokens per frame."""
    five = image.ndim == 5
    T = image.shape[1]
    if vit_model == "eva_clip_g":
        x = image.reshape((-1,) + tuple(image.shape[2:])) if five else image  # :327-328
        emb = vit_forward(x, sd, p + "visual_encoder.")
    else:
        emb = btadapter_forward(image, sd, p + "visual_encoder.", adapter_depth)
    emb = ln_vision(emb, sd, p + "ln_vision")
    N = emb.shape[0]
    if not has_qformer:
        emb = emb[:, 1:, :]                                            # :370
        bs, pn, hs = emb.shape
        out = _lin(emb.reshape(bs, pn // 4, hs * 4), sd, p + "llama_proj")   # :371-373
        if five:
            out = out.reshape(-1, T, out.shape[1], out.shape[2])
        return out
    q = sd[p + "query_tokens"].expand(N, -1, -1)
    att = None
    if text_ids is not None:
        att = torch.cat([torch.ones(N, q.shape[1], dtype=torch.long), text_mask.long()], dim=1)
    h = qformer_forward(q, emb, sd, p + "Qformer.bert.", text_ids, att)
    out = _lin(h[:, : q.shape[1], :], sd, p + "llama_proj")  # :368
    if five:
        out = out.reshape(-1, T, out.shape[1], out.shape[2])  # :375
    return out


def get_residual_index(sample_segments: int, total_segments: int) -> np.ndarray:
    """st_llm.py:434-445 / conversation.py:118-125: idx_i = int(seg/2 + round(seg*i)),
    numpy round = half-to-even."""
    seg = float(total_segments) / sample_segments
    return np.array([int((seg / 2) + np.round(seg * i)) for i in range(sample_segments)])


def video_pool(img_embeds: Tensor, video_input: Optional[str], sd: SD, p: str = "",
               residual_size: int = 4) -> Tensor:
    """st_llm.py:463-478.  img_embeds [B,T,32,4096] -> [B,1,L,4096]."""
    B, T = img_embeds.shape[0], img_embeds.shape[1]
    D = img_embeds.shape[-1]
    if video_input == "all":
        return img_embeds.reshape(B, 1, -1, D)
    if video_input == "mean":
        return img_embeds.mean(dim=1, keepdim=True)
    if video_input == "residual":
        idx = torch.from_numpy(get_residual_index(residual_size, T))
        g = img_embeds.mean(dim=1, keepdim=True)
        local = img_embeds[:, idx]
        g = g.expand((-1, residual_size, -1, -1))
        g = _lin(torch.relu(_lin(g, sd, p + "down_proj")), sd, p + "up_proj")
        return (local + g).reshape(B, 1, -1, D)
    return img_embeds


def video_pool_infer(video_emb: Tensor, video_input: str, sd: SD, p: str = "", residual_size: int = 4) -> Tensor:
    """Chat.upload_video conversation.py:280-293: [T,32,4096] -> [1,L,4096]."""
    return video_pool(video_emb.unsqueeze(0), video_input, sd, p, residual_size)[:, 0]


def random_masking_generator(num_patches: int, mask_ratio: float, batch: int) -> np.ndarray:
    """models/utils.py:4-16 — numpy global RNG, True = dropped."""
    num_mask = int(mask_ratio * num_patches)
    rows = []
    for _ in range(batch):
        m = np.hstack([np.zeros(num_patches - num_mask), np.ones(num_mask)])
        np.random.shuffle(m)
        rows.append(m)
    return np.array(rows).astype(bool)


def apply_mask(img_embeds: Tensor, mask: Tensor) -> Tensor:
    """st_llm.py:486-491: img_embeds [B,1,L,D], mask [B,L] bool (True = dropped) -> [B,1,L-k,D]."""
    B, _, L, D = img_embeds.shape
    return img_embeds[~mask.unsqueeze(1)].reshape(B, 1, -1, D)


def prompt_wrap(img_embeds: Tensor, before_ids: Sequence[Sequence[int]], after_ids: Sequence[Sequence[int]],
                embed: Tensor, pad_id: int):
    """st_llm.py:379-407 with token ids instead of strings: cat[before | video | after],
    right-pad with the pad-token embedding.  img_embeds [B,1,L,D]."""
    embs = []
    for i in range(img_embeds.shape[0]):
        parts = []
        if len(before_ids[i]) > 0:
            parts.append(embed[torch.tensor(before_ids[i], dtype=torch.long)])
        parts.append(img_embeds[i].reshape(-1, img_embeds.shape[-1]))
        parts.append(embed[torch.tensor(after_ids[i], dtype=torch.long)])
        embs.append(torch.cat(parts, dim=0))
    lens = [e.shape[0] for e in embs]
    out = embed[pad_id].expand(len(lens), max(lens), -1).clone()
    att = torch.zeros(len(lens), max(lens), dtype=torch.int)
    for i, e in enumerate(embs):
        out[i, : lens[i]] = e
        att[i, : lens[i]] = 1
    return out, att


def concat_emb_input_output(in_embs: Tensor, in_atts: Tensor, out_embs: Tensor, out_atts: Tensor):
    """st_llm.py:409-432."""
    lens, ce, ca = [], [], []
    for i in range(in_embs.shape[0]):
        n = int(in_atts[i].sum())
        lens.append(n)
        ce.append(torch.cat([in_embs[i][:n], out_embs[i], in_embs[i][n:]]))
        ca.append(torch.cat([in_atts[i][:n], out_atts[i], in_atts[i][n:]]))
    return torch.stack(ce), torch.stack(ca), lens


def assemble(img_embeds: Tensor, before_ids, after_ids, answer_ids: Sequence[Sequence[int]], embed: Tensor,
             pad_id: int, bos_id: int, prepend_bos: bool, unmask_img_embeds: Optional[Tensor] = None):
    """STLLMModel.forward st_llm.py:496-546 (token-block assembly + targets), token ids in.
    answer_ids are already end_sym/eos-terminated and truncated to max_txt_len."""
    B = img_embeds.shape[0]
    wrapped, atts = prompt_wrap(img_embeds, before_ids, after_ids, embed, pad_id)
    La = max(len(a) for a in answer_ids)
    ans = torch.full((B, La), pad_id, dtype=torch.long)
    ans_att = torch.zeros(B, La, dtype=torch.long)
    for i, a in enumerate(answer_ids):
        ans[i, : len(a)] = torch.tensor(a, dtype=torch.long)
        ans_att[i, : len(a)] = 1
    ans_emb = embed[ans]
    inputs_embeds, attention_mask, input_lens = concat_emb_input_output(wrapped, atts, ans_emb, ans_att)
    un_e = un_a = None
    if unmask_img_embeds is not None:
        uw, ua = prompt_wrap(unmask_img_embeds, before_ids, after_ids, embed, pad_id)
        un_e, un_a, _ = concat_emb_input_output(uw, ua, ans_emb, ans_att)
    if prepend_bos:  # not qformer_text_input (:519-530)
        bos = embed[torch.full((B, 1), bos_id, dtype=torch.long)]
        one = torch.ones(B, 1, dtype=attention_mask.dtype)
        inputs_embeds = torch.cat([bos, inputs_embeds], dim=1)
        attention_mask = torch.cat([one, attention_mask], dim=1)
        if un_e is not None:
            un_e = torch.cat([bos, un_e], dim=1)
            un_a = torch.cat([one, un_a], dim=1)
    part = ans.masked_fill(ans == pad_id, -100)
    targets = torch.full((B, inputs_embeds.shape[1]), -100, dtype=torch.long)
    off = 1 if prepend_bos else 0
    for i in range(B):
        targets[i, input_lens[i] + off: input_lens[i] + La + off] = part[i]
    return inputs_embeds, attention_mask, un_e, un_a, targets


# =====================================================================================
# Vicuna-7B / Llama prefill  (HF transformers LlamaModel; in-tree spec modeling_llama_mem.py)
# =====================================================================================
def rms_norm(x: Tensor, w: Tensor, eps: float = 1e-6) -> Tensor:
    """LlamaRMSNorm modeling_llama_mem.py:61-78 (variance in fp32)."""
    v = x.float().pow(2).mean(-1, keepdim=True)
    return w * (x.float() * torch.rsqrt(v + eps)).to(x.dtype)


def rope_tables(S: int, head_dim: int = 128, base: float = 10000.0):
    """LlamaRotaryEmbedding modeling_llama_mem.py:81-110: emb = cat(freqs, freqs)."""
    inv = 1.0 / (base ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    f = torch.outer(torch.arange(S, dtype=torch.float32), inv)
    emb = torch.cat((f, f), dim=-1)
    return emb.cos(), emb.sin()


def _rotate_half(x: Tensor) -> Tensor:
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def llama_num_layers(sd: SD, p: str = "model.") -> int:
    n = 0
    while f"{p}layers.{n}.input_layernorm.weight" in sd:
        n += 1
    return n


def llama_layer(h: Tensor, add_mask: Tensor, cos: Tensor, sin: Tensor, sd: SD, lp: str, n_heads: int,
                eps: float) -> Tensor:
    """LlamaDecoderLayer: modeling_llama_mem.py:113-144 (RoPE rotate-half, SwiGLU), :172-248
    (attention; here the eager softmax(QK^T/sqrt(d)+mask)V form HF uses)."""
    B, S, D = h.shape
    x = rms_norm(h, sd[lp + "input_layernorm.weight"], eps)
    q = F.linear(x, sd[lp + "self_attn.q_proj.weight"]).view(B, S, n_heads, -1).transpose(1, 2)
    k = F.linear(x, sd[lp + "self_attn.k_proj.weight"]).view(B, S, n_heads, -1).transpose(1, 2)
    v = F.linear(x, sd[lp + "self_attn.v_proj.weight"]).view(B, S, n_heads, -1).transpose(1, 2)
    q = q * cos + _rotate_half(q) * sin
    k = k * cos + _rotate_half(k) * sin
    s = (q @ k.transpose(-1, -2)) / math.sqrt(q.shape[-1]) + add_mask
    a = (s.softmax(dim=-1) @ v).transpose(1, 2).reshape(B, S, D)
    h = h + F.linear(a, sd[lp + "self_attn.o_proj.weight"])
    x = rms_norm(h, sd[lp + "post_attention_layernorm.weight"], eps)
    m = F.linear(F.silu(F.linear(x, sd[lp + "mlp.gate_proj.weight"])) * F.linear(x, sd[lp + "mlp.up_proj.weight"]),
                 sd[lp + "mlp.down_proj.weight"])
    return h + m


def llama_forward(inputs_embeds: Tensor, attention_mask: Optional[Tensor], sd: SD, p: str = "model.",
                  n_heads: int = 32, eps: float = 1e-6, final_norm: bool = True) -> Tensor:
    """LlamaModel.forward, prefill (use_cache=False), positions 0..S-1, causal + key-padding mask
    (st_llm.py:62-67).  Returns last hidden state AFTER model.norm (== hidden_states[-1])."""
    B, S, D = inputs_embeds.shape
    cos, sin = rope_tables(S, D // n_heads)
    neg = torch.finfo(torch.float32).min
    causal = torch.full((S, S), neg).triu(1)[None, None]
    add_mask = causal
    if attention_mask is not None:
        pad = (attention_mask[:, None, None, :] == 0)
        add_mask = causal.expand(B, 1, S, S).masked_fill(pad, neg)
    h = inputs_embeds
    for i in range(llama_num_layers(sd, p)):
        h = llama_layer(h, add_mask, cos, sin, sd, f"{p}layers.{i}.", n_heads, eps)
    return rms_norm(h, sd[p + "norm.weight"], eps) if final_norm else h


def lm_logits(hidden: Tensor, sd: SD) -> Tensor:
    """st_llm.py:122 — lm_head on ALL positions."""
    return F.linear(hidden, sd["lm_head.weight"])


def causal_lm_loss(logits: Tensor, labels: Tensor) -> Tensor:
    """st_llm.py:125-135 — shifted cross-entropy, ignore_index -100."""
    V = logits.shape[-1]
    return F.cross_entropy(logits[..., :-1, :].reshape(-1, V), labels[..., 1:].reshape(-1), ignore_index=-100)


def mvm_loss(mask_hidden: Tensor, unmask_hidden: Tensor, mask: Tensor, img_start: int, img_len: int,
             mask_img_len: int, sd: SD, p: str = "model.stllm_model.") -> Tensor:
    """STLLMLlamaModel.forward MVM branch st_llm.py:71-91.  mask [B,L] bool, True = dropped.
    Linear_Decoder = LayerNorm(Linear(4096,4096)), eps 1e-5 (:35-43)."""
    B, _, D = mask_hidden.shape
    a = mask_hidden[:, img_start: img_start + mask_img_len]
    if (p + "mvm_decoder.head.weight") in sd:
        a = layer_norm(_lin(a, sd, p + "mvm_decoder.head"), sd, p + "mvm_decoder.norm", 1e-5)
    b = unmask_hidden[:, img_start: img_start + img_len]
    b = b[~mask].reshape(B, -1, D)
    a = a / a.norm(dim=-1, keepdim=True)
    b = b / b.norm(dim=-1, keepdim=True)
    return (2 - 2 * (a * b).sum(dim=-1)).mean()


# =====================================================================================
# End-to-end: STLLMForCausalLM.forward(samples)  (st_llm.py:116-146) on token ids
# =====================================================================================
def stllm_forward(samples: dict, sd: SD, cfg: dict):
    """samples: {"image": [B,T,3,224,224], "before_ids","after_ids","answer_ids": list[list[int]],
    optional "qformer_ids"/"qformer_mask": [B,Lt], optional "mask": [B,L] bool (injected, True = dropped)}.
    cfg: vit_model, video_input, residual_size, use_mask, mvm_decode, qformer_text_input, has_qformer, pre_encoding, pad_id, bos_id, n_heads.
    Returns dict(logits, loss, loss_mvm, inputs_embeds, attention_mask, targets)."""
    p = "model.stllm_model."
    image = samples["image"]
    B, T = image.shape[0], image.shape[1]
    tids = tmask = None
    if cfg.get("qformer_text_input", False):
        tids = samples["qformer_ids"].repeat_interleave(T, dim=0)
        tmask = samples["qformer_mask"].repeat_interleave(T, dim=0)
    if cfg.get("pre_encoding", False):
        # st_llm.py:452-455: `image` holds pre-extracted features [B, T, L, C]; only llama_proj runs (use_image stays False whatever T is)
        emb = F.linear(image.float(), sd[p + "llama_proj.weight"], sd[p + "llama_proj.bias"])
    else:
        emb = encode_img(image, sd, p, cfg.get("vit_model", "eva_clip_g"), tids, tmask, has_qformer=cfg.get("has_qformer", True))
    emb = video_pool(emb, cfg.get("video_input"), sd, p, cfg.get("residual_size", 4))
    un = None
    mask = None
    if cfg.get("use_mask", False):
        mask = samples["mask"]
        un = emb
        emb = apply_mask(emb, mask)
    embed = sd["model.embed_tokens.weight"]
    prepend_bos = not cfg.get("qformer_text_input", False)
    ie, am, ue, ua, targets = assemble(emb, samples["before_ids"], samples["after_ids"], samples["answer_ids"],
                                       embed, cfg["pad_id"], cfg["bos_id"], prepend_bos, un)
    nh = cfg.get("n_heads", 32)
    hid = llama_forward(ie, am, sd, "model.", nh)
    out = {"inputs_embeds": ie, "attention_mask": am, "targets": targets, "loss_mvm": None}
    if un is not None:
        with torch.no_grad():                      # st_llm.py:77-83: the un-masked pass is the (detached) target
            uh = llama_forward(ue, ua, sd, "model.", nh)
        img_start = 0 if cfg.get("qformer_text_input", False) else 8
        out["loss_mvm"] = mvm_loss(hid, uh, mask, img_start, un.shape[2], emb.shape[2], sd, p)
    logits = lm_logits(hid, sd)
    loss = causal_lm_loss(logits, targets)
    if out["loss_mvm"] is not None:
        loss = loss + out["loss_mvm"]
    out.update(logits=logits, loss=loss, hidden=hid)
    return out
