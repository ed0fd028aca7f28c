"""BLIP-2 Q-Former encoder behind the reference's surface (stllm/models/Qformer.py), on the HIP C ABI.

Only the encoder forward used by ST-LLM is implemented: LM/MLM heads, pruning,
kv-cache and the decoder (is_decoder) mask are out of scope.  Parameter names follow the reference
(Qformer.py:127-133, 281-282, 352, 367-368, 384-400) so BLIP-2 / InstructBLIP checkpoints load unchanged.

Data layout: the query rows [N*32, 768] and the text rows [N*Lt, 768] live in two compact fp32 streams
(post-LN residual, Qformer.py:285-289) — they only meet inside self-attention, whose fused QKV buffer
[N, 32+Lt, 2304] is written / read through the GEMM's 2-level row indexing.  Cross-attention (even
layers, query rows only, Qformer.py:430-444) reads K/V = Linear(1408->768) of the ln_vision'd image tokens.
"""
import math

import torch
import torch.nn as nn

from .. import hip, pack, runtime
from .layers import Embedding, LayerNorm, Linear, Output, ParamList, params_fingerprint


class BertConfig:
    """bert-base-uncased defaults + the Q-Former extras set at blip2.py:48-53."""

    def __init__(self, **kw):
        self.vocab_size = 30522
        self.hidden_size = 768
        self.num_hidden_layers = 12
        self.num_attention_heads = 12
        self.intermediate_size = 3072
        self.max_position_embeddings = 512
        self.layer_norm_eps = 1e-12
        self.encoder_width = 1408
        self.add_cross_attention = True
        self.cross_attention_freq = 2
        self.query_length = 32
        self.__dict__.update(kw)


class BertSelfAttention(nn.Module):
    def __init__(self, cfg, is_cross, device):
        super().__init__()
        kv = cfg.encoder_width if is_cross else cfg.hidden_size
        self.query = Linear(cfg.hidden_size, cfg.hidden_size, device=device)
        self.key = Linear(kv, cfg.hidden_size, device=device)
        self.value = Linear(kv, cfg.hidden_size, device=device)


class BertSelfOutput(nn.Module):
    def __init__(self, cfg, device, in_dim=None):
        super().__init__()
        self.dense = Linear(in_dim or cfg.hidden_size, cfg.hidden_size, device=device)
        self.LayerNorm = LayerNorm(cfg.hidden_size, cfg.layer_norm_eps, device)


class BertAttention(nn.Module):
    def __init__(self, cfg, is_cross=False, device=None):
        super().__init__()
        self.self = BertSelfAttention(cfg, is_cross, device)
        self.output = BertSelfOutput(cfg, device)


class BertIntermediate(nn.Module):
    def __init__(self, cfg, device):
        super().__init__()
        self.dense = Linear(cfg.hidden_size, cfg.intermediate_size, device=device)


class BertOutput(BertSelfOutput):
    def __init__(self, cfg, device):
        super().__init__(cfg, device, in_dim=cfg.intermediate_size)


class BertLayer(nn.Module):
    def __init__(self, cfg, layer_num, device=None):
        super().__init__()
        self.layer_num = layer_num
        self.attention = BertAttention(cfg, device=device)
        self.has_cross_attention = cfg.add_cross_attention and layer_num % cfg.cross_attention_freq == 0
        if self.has_cross_attention:
            self.crossattention = BertAttention(cfg, is_cross=True, device=device)
        self.intermediate = BertIntermediate(cfg, device)
        self.output = BertOutput(cfg, device)
        self.intermediate_query = BertIntermediate(cfg, device)
        self.output_query = BertOutput(cfg, device)

    def pack(self, dt):
        def out(o):
            return dict(w=pack.linear(o.dense.weight, dt), b=pack.f32(o.dense.bias), g=o.LayerNorm.weight,
                        beta=o.LayerNorm.bias, eps=o.LayerNorm.eps)

        def ffn(i, o):
            return dict(w1=pack.linear(i.dense.weight, dt), b1=pack.f32(i.dense.bias), out=out(o))
        s = self.attention.self
        wqkv, bqkv = pack.bert_qkv(s.query, s.key, s.value, dt)
        pk = dict(wqkv=wqkv, bqkv=bqkv, attn_out=out(self.attention.output), ffn_q=ffn(self.intermediate_query, self.output_query))
        if self.has_cross_attention:
            c = self.crossattention.self
            wkv, bkv = pack.bert_kv(c.key, c.value, dt)
            pk.update(cq_w=pack.linear(c.query.weight, dt), cq_b=pack.f32(c.query.bias), ckv_w=wkv, ckv_b=bkv,
                      cross_out=out(self.crossattention.output))
        if self.intermediate is not None and self.output is not None:
            pk["ffn_t"] = ffn(self.intermediate, self.output)
        return pk


class BertEmbeddings(nn.Module):
    def __init__(self, cfg, device=None):
        super().__init__()
        self.word_embeddings = Embedding(cfg.vocab_size, cfg.hidden_size, device)
        self.position_embeddings = Embedding(cfg.max_position_embeddings, cfg.hidden_size, device)
        self.LayerNorm = LayerNorm(cfg.hidden_size, cfg.layer_norm_eps, device)
        self.register_buffer("position_ids", torch.arange(cfg.max_position_embeddings).expand((1, -1)), persistent=True)


class BertEncoder(nn.Module):
    def __init__(self, cfg, device=None):
        super().__init__()
        self.layer = nn.ModuleList([BertLayer(cfg, i, device) for i in range(cfg.num_hidden_layers)])


def _post_ln(ctx, dense, resid32, dt, **rows):
    """BertSelfOutput / BertOutput: LayerNorm(dense(ctx) + input) -> (fp32 stream, compute-dtype copy)."""
    tmp = torch.empty_like(resid32)
    hip.gemm(ctx, dense["w"], dtype=dt, epilogue=hip.EPI_RESID, bias=dense["b"], resid=resid32, out=tmp, **rows)
    h16, h32 = hip.layernorm(tmp, dense["g"], dense["beta"], dense["eps"], dtype=dt, want_f32=True)
    return h32, h16


class BertModel(nn.Module):
    def __init__(self, config, add_pooling_layer=False, device=None):
        super().__init__()
        self.config = config
        self.embeddings = BertEmbeddings(config, device)
        self.encoder = BertEncoder(config, device)
        self._packed = {}
        self._carr = {}   # C-side table of the packed layers: rebuilt when they are re-packed
        self._plist = ParamList(lambda: self.encoder.layer.parameters())

    def pack(self, dtype=None):
        dt = hip.torch_dtype(dtype) if dtype is not None else runtime.compute_dtype()
        fp = params_fingerprint(self._plist.get())
        hit = self._packed.get(dt)
        if hit is None or hit[0] != fp:
            layers = [l.pack(dt) for l in self.encoder.layer]
            # the cross-attention K/V projections of ALL layers read the same image tokens: one [n_cross * 2C, 1408] weight, one GEMM
            # per forward instead of one per cross layer (6 launches of 4112 x 1536 x 1408 -> 1 of 4112 x 9216 x 1408 at config 2)
            cross = [pk for pk in layers if "ckv_w" in pk]
            if cross:
                w_all = torch.cat([pk["ckv_w"] for pk in cross], dim=0).contiguous()
                b_all = torch.cat([pk["ckv_b"] for pk in cross], dim=0).contiguous()
                for j, pk in enumerate(cross):
                    pk["ckv_all"] = (w_all, b_all, j, len(cross))
            hit = (fp, layers)
            self._packed = {dt: hit}
            self._carr = {}
        return hit[1]

    def repack(self):
        self._packed = {}
        self._carr = {}
        self._plist.reset()

    def _load_from_state_dict(self, *a, **k):
        self._packed = {}
        self._carr = {}
        self._plist.reset()
        return super()._load_from_state_dict(*a, **k)

    # ------------------------------------------------------------------------------------------
    def encode(self, query_tokens, enc16, n, input_ids=None, text_mask=None):
        """query_tokens f32 [32,768] (shared by all n sequences); enc16 compute-dtype [n*P,1408] = ln_vision'd
        image tokens; input_ids/text_mask host LongTensors [n,Lt] or None.
        Returns (hq32 [n*32,768], hq16, ht32 [n*Lt,768] | None)."""
        cfg = self.config
        dt = runtime.compute_dtype()
        layers = self.pack(dt)
        dev = enc16.device
        C, H, Q = cfg.hidden_size, cfg.num_attention_heads, query_tokens.shape[0]
        P = enc16.shape[0] // n
        emb = self.embeddings
        # ---- embeddings (Qformer.py:78-108): queries get no position embedding --------------------
        q_idx = hip.arange_repeat(Q, n, dev)
        q_emb = hip.gather_rows(query_tokens.float().contiguous(), q_idx)
        hq16, hq32 = hip.layernorm(q_emb, emb.LayerNorm.weight, emb.LayerNorm.bias, emb.LayerNorm.eps, dtype=dt, want_f32=True)
        Lt, ht32, ht16, kv_len = 0, None, None, None
        if input_ids is not None:
            Lt = input_ids.shape[1]
            ids = input_ids.reshape(-1).to(torch.int64)
            w_idx = hip.h2d((-(ids) - 1).to(torch.int32), dev)
            p_idx = hip.arange_repeat(Lt, n, dev)
            t_emb = hip.gather_rows(emb.word_embeddings.weight, w_idx, src_b=emb.word_embeddings.weight,
                                    add=emb.position_embeddings.weight, idx_add=p_idx)
            ht16, ht32 = hip.layernorm(t_emb, emb.LayerNorm.weight, emb.LayerNorm.bias, emb.LayerNorm.eps, dtype=dt, want_f32=True)
            kv_len = hip.h2d((Q + text_mask.long().sum(dim=1)).to(torch.int32), dev)
            m = text_mask.long()
            assert bool((m[:, 1:] <= m[:, :-1]).all()), "Q-Former text mask must be right-padded (padding='longest')"
        from . import llama as _llama
        if _llama.STACK_ENTRY:   # Qformer.py:495-589: the whole layer loop is ONE call into the C ABI (stllm_qformer_layers; == encode_layers_per_op, bit for bit)
            if self._carr.get("layers") is not layers:
                self._carr = {"layers": layers, "carr": hip.qformer_layer_array(layers)}
            return hip.qformer_layers(hq32, hq16, ht32, ht16, enc16, layers, self._carr["carr"], n_seq=n, n_query=Q, n_text=Lt, n_heads=H,
                                      dtype=dt, kv_len=kv_len)
        return self.encode_layers_per_op(layers, hq32, hq16, ht32, ht16, enc16, n, Q, Lt, kv_len, dt)

    def encode_layers_per_op(self, layers, hq32, hq16, ht32, ht16, enc16, n, Q, Lt, kv_len, dt):
        """The BertLayer loop as one C-ABI call per op — what stllm_qformer_layers issues from C.  Kept as the reference the stack entry
        point is tested against (-m gpu: bit-identical) and as the body the test-only CPU contract backend runs."""
        cfg = self.config
        C, H = cfg.hidden_size, cfg.num_attention_heads
        dev = hq32.device
        P = enc16.shape[0] // n
        S = Q + Lt
        qkv = torch.empty((n * S, 3 * C), device=dev, dtype=dt)
        rows_q = dict(M=n * Q, o_rows=(Q, S * 3 * C)) if Lt else {}
        rows_t = dict(M=n * Lt, o_rows=(Lt, S * 3 * C)) if Lt else {}
        hd = C // H
        ckv_all = None
        for i, pk in enumerate(layers):
            # ---- self-attention over [queries | text] (Qformer.py:417-424) -------------------------
            hip.gemm(hq16, pk["wqkv"], dtype=dt, bias=pk["bqkv"], out=qkv, **rows_q)
            if Lt:
                hip.gemm(ht16, pk["wqkv"], dtype=dt, bias=pk["bqkv"], out=qkv[Q:], **rows_t)
            ctx = hip.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], B=n, H=H, Sq=S, Skv=S, D=hd,
                                scale=1.0 / math.sqrt(hd), kv_len=kv_len)
            a_q = dict(M=n * Q, a_rows=(Q, S * C)) if Lt else {}
            hq32, hq16 = _post_ln(ctx, pk["attn_out"], hq32, dt, **a_q)
            if Lt:
                ht32, ht16 = _post_ln(ctx[Q:], pk["attn_out"], ht32, dt, M=n * Lt, a_rows=(Lt, S * C))
            # ---- cross-attention, query rows only, even layers (Qformer.py:430-444) -----------------
            if "cq_w" in pk:
                cq = hip.gemm(hq16, pk["cq_w"], dtype=dt, bias=pk["cq_b"])
                w_all, b_all, j, n_cross = pk["ckv_all"]
                if ckv_all is None:
                    ckv_all = hip.gemm(enc16, w_all, dtype=dt, bias=b_all)
                ckv = ckv_all[:, j * 2 * C:(j + 1) * 2 * C]
                cctx = hip.attention(cq, ckv[:, :C], ckv[:, C:], B=n, H=H, Sq=Q, Skv=P, D=hd, scale=1.0 / math.sqrt(hd))
                hq32, hq16 = _post_ln(cctx, pk["cross_out"], hq32, dt)
            # ---- FFN: query rows -> *_query weights, text rows -> text weights (Qformer.py:449-462) ----
            f = pk["ffn_q"]
            g = hip.gemm(hq16, f["w1"], dtype=dt, bias=f["b1"], act=hip.ACT_GELU)
            hq32, hq16 = _post_ln(g, f["out"], hq32, dt)
            if Lt:
                f = pk["ffn_t"]
                g = hip.gemm(ht16, f["w1"], dtype=dt, bias=f["b1"], act=hip.ACT_GELU)
                ht32, ht16 = _post_ln(g, f["out"], ht32, dt)
        return hq32, hq16, ht32

    def forward(self, input_ids=None, attention_mask=None, position_ids=None, head_mask=None, query_embeds=None,
                encoder_hidden_states=None, encoder_attention_mask=None, past_key_values=None, use_cache=None,
                output_attentions=None, output_hidden_states=None, return_dict=None, is_decoder=False):
        """BertModel.forward (Qformer.py:804-965) for the call shapes ST-LLM uses (st_llm.py:353-367):
        query_embeds [N,32,768] (an expand of query_tokens), encoder_hidden_states [N,P,1408] fp32,
        optional input_ids [N,Lt] + attention_mask [N,32+Lt].  Returns last_hidden_state [N,32+Lt,768]."""
        if input_ids is None:
            assert query_embeds is not None, "You have to specify query_embeds when input_ids is None"
        if is_decoder or past_key_values is not None or output_attentions:
            raise NotImplementedError("decoder mode / kv-cache / attention maps are outside the hot path")
        n, Q, C = query_embeds.shape
        qt = query_embeds[0]
        if n > 1 and query_embeds.stride(0) != 0 and not torch.equal(query_embeds[0], query_embeds[-1]):
            raise NotImplementedError("per-sample query_embeds are not used by ST-LLM")
        dt = runtime.compute_dtype()
        enc = encoder_hidden_states
        enc16 = hip.cast_rows(enc.reshape(-1, enc.shape[-1]).float().contiguous(), dt)
        tmask = None
        if input_ids is not None:
            tmask = attention_mask[:, Q:].cpu() if attention_mask is not None else torch.ones_like(input_ids)
            input_ids = input_ids.cpu()
        hq32, _, ht32 = self.encode(qt, enc16, n, input_ids, tmask)
        out = hq32.view(n, Q, C)
        if ht32 is not None:
            out = torch.cat([out, ht32.view(n, -1, C)], dim=1)
        return Output(last_hidden_state=out)


class BertLMHeadModel(nn.Module):
    """Container named like the reference's (`Qformer.bert.*` keys); the LM head (`cls`) is unused and dropped
    by ST-LLM (st_llm.py:288)."""

    def __init__(self, config, device=None):
        super().__init__()
        self.config = config
        self.bert = BertModel(config, add_pooling_layer=False, device=device)
        self.cls = None

    def resize_token_embeddings(self, n):
        emb = self.bert.embeddings.word_embeddings
        old = emb.weight
        if old.shape[0] != n:
            new = torch.zeros((n, old.shape[1]), device=old.device, dtype=old.dtype)
            k = min(n, old.shape[0])
            new[:k] = old[:k]
            emb.weight = nn.Parameter(new, requires_grad=False)
            self.config.vocab_size = n
        return emb
