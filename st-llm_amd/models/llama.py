"""Vicuna-7B / Llama-1 decoder PREFILL on the HIP C ABI, HF parameter names.

The reference imports HF ``transformers.models.llama`` (st_llm.py:20; pinned 4.28.0) — third-party code
that is not in the reference tree; its in-tree statement is stllm/models/modeling_llama_mem.py:61-144.
Per layer:

    RMSNorm -> [fused QKV GEMM + rotate-half RoPE epilogue] -> causal flash attention (+ right-pad mask)
    -> [o_proj GEMM + fp32 residual] -> RMSNorm -> [gate/up GEMM + SiLU(gate)*up epilogue]
    -> [down_proj GEMM + fp32 residual]

then the final RMSNorm and lm_head on ALL positions (st_llm.py:122).

Decode with a KV cache (the step after the prefill in Chat.answer -> generate): the fused QKV
buffer of every layer IS the cache — ``KVCache`` owns one [B, max_len, 3*D] buffer per layer; the prefill GEMM writes rows
[0, S) of it through the 2-level row indexing, a decode step writes row `len` (RoPE at that position) and attends over
rows [0, len] in place: no copies, no re-layout.  Decode reuses the prefill kernels (small-M tiles); GEMV-regime kernels
are future work.
"""
import os

import torch
import torch.nn as nn

from .. import hip, pack, runtime
from .layers import Embedding, Linear, Output, ParamList, RMSNorm, params_fingerprint


def _frag(pk, key):
    """keyword of hip.gemm for the fragment-major copy of pk[key], when the pack made one (pack.frag32_or_none)"""
    f = pk.get(key + "_frag")
    return {"w_frag": f} if f is not None else {}


class LlamaConfig:
    model_type = "llama"

    def __init__(self, hidden_size=4096, intermediate_size=11008, num_hidden_layers=32, num_attention_heads=32,
                 vocab_size=32000, rms_norm_eps=1e-6, max_position_embeddings=2048, rope_theta=10000.0, **kw):
        self.hidden_size, self.intermediate_size = hidden_size, intermediate_size
        self.num_hidden_layers, self.num_attention_heads = num_hidden_layers, num_attention_heads
        self.vocab_size, self.rms_norm_eps = vocab_size, rms_norm_eps
        self.max_position_embeddings, self.rope_theta = max_position_embeddings, rope_theta
        self.__dict__.update(kw)


class LlamaAttention(nn.Module):
    def __init__(self, cfg, device):
        super().__init__()
        d = cfg.hidden_size
        self.q_proj = Linear(d, d, bias=False, device=device)
        self.k_proj = Linear(d, d, bias=False, device=device)
        self.v_proj = Linear(d, d, bias=False, device=device)
        self.o_proj = Linear(d, d, bias=False, device=device)


class LlamaMLP(nn.Module):
    def __init__(self, cfg, device):
        super().__init__()
        self.gate_proj = Linear(cfg.hidden_size, cfg.intermediate_size, bias=False, device=device)
        self.up_proj = Linear(cfg.hidden_size, cfg.intermediate_size, bias=False, device=device)
        self.down_proj = Linear(cfg.intermediate_size, cfg.hidden_size, bias=False, device=device)


class LlamaDecoderLayer(nn.Module):
    def __init__(self, cfg, device):
        super().__init__()
        self.self_attn = LlamaAttention(cfg, device)
        self.mlp = LlamaMLP(cfg, device)
        self.input_layernorm = RMSNorm(cfg.hidden_size, cfg.rms_norm_eps, device)
        self.post_attention_layernorm = RMSNorm(cfg.hidden_size, cfg.rms_norm_eps, device)

    def pack(self, dt, n_heads, frag=True):
        a, m = self.self_attn, self.mlp
        pk = dict(ln1=self.input_layernorm.weight, ln2=self.post_attention_layernorm.weight,
                  wqkv=pack.llama_qkv(a.q_proj.weight, a.k_proj.weight, a.v_proj.weight, dt, n_heads),
                  wo=pack.linear(a.o_proj.weight, dt),
                  wgu=pack.llama_gate_up(m.gate_proj.weight, m.up_proj.weight, dt),
                  wdown=pack.linear(m.down_proj.weight, dt))
        # fragment-major copy of wqkv for the W-direct prefill GEMM (csrc/gemm_wd.inc, round 6): + 100 MB per layer (3.2 GB at 7B in 16 bits).  The gate/up
        # weight gets none: its 86 column blocks never make the one-round plans the dispatcher takes (measured equal or slower, profiles/r06_bench_ab_wd.log);
        # the C entry point and hip.gemm accept one all the same (stllm_llama_layer_weights.wgu_frag, w_frag=)
        pk["wqkv_frag"] = pack.frag32_or_none(pk["wqkv"]) if (frag and dt != torch.float32) else None   # (fp32 / bf16x3: other kernels)
        pk["wgu_frag"] = None
        return pk


def cfg_max_len(cfg):
    return int(getattr(cfg, "max_position_embeddings", 2048))


class KVCache:
    """Per-layer fused [q | k | v] rows (compute dtype, k/q in the packed RoPE head layout).  Equal-length sequences only
    (a clip, or the beams of one clip): right-padded batches would need per-row positions."""

    def __init__(self, n_layers, batch, max_len, hidden, dtype, device):
        self.max_len, self.batch, self.hidden = max_len, batch, hidden
        self.qkv = [torch.empty((batch, max_len, 3 * hidden), device=device, dtype=dtype) for _ in range(n_layers)]
        self.len = 0


STACK_ENTRY = os.environ.get("STLLM_STACK_ENTRY", "1") != "0"   # 0: one C-ABI call per op instead of stllm_llama_layers / stllm_vit_blocks (A/B, tests)
FUSE_NORM_ROWS = int(os.environ.get("STLLM_DECODE_FUSE_ROWS", "2"))   # decode steps with at most this many rows fuse RMSNorm into the GEMVs


class LlamaModel(nn.Module):
    def __init__(self, config, device=None):
        super().__init__()
        self.config = config
        if config.hidden_size // config.num_attention_heads != 128:
            raise NotImplementedError("RoPE epilogue / attention kernels are specialised for head_dim 128")
        self.embed_tokens = Embedding(config.vocab_size, config.hidden_size, device)
        self.layers = nn.ModuleList([LlamaDecoderLayer(config, device) for _ in range(config.num_hidden_layers)])
        self.norm = RMSNorm(config.hidden_size, config.rms_norm_eps, device)
        self._packed = {}
        self._rope = {}
        self._carr = {}   # C-side table of the packed layers (+ the cache it points into): rebuilt when either changes
        self._plist = ParamList(lambda: self.layers.parameters())
        self.wd_frag = True   # keep fragment-major copies of wqkv / wgu next to the packed weights (W-direct prefill GEMM); the training step switches this off:
                              # its weights change every step and its taped forward does not use the copies

    def pack(self, dtype=None):
        dt = hip.torch_dtype(dtype) if dtype is not None else runtime.compute_dtype()
        fp = params_fingerprint(self._plist.get())   # a stale packed copy after p.data.copy_ / .to(device) would run silently
        hit = self._packed.get(dt)
        if hit is None or hit[0] != fp:
            self._packed = {}  # one packed copy at a time (13.5 GB at 7B) ...
            self._carr = {}    # ... including the C-side table, which holds a reference to the list it was built from
            hit = (fp, [l.pack(dt, self.config.num_attention_heads, frag=self.wd_frag) for l in self.layers])
            self._packed[dt] = hit
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

    def rope(self, S, device):
        """cos / sin tables [S, 64] on the device, cached per (S, device) — a handful of entries, oldest dropped.  The MVM forward alternates
        two sequence lengths inside ONE step (masked and un-masked prefill, st_llm.py:56-92): with a single-entry cache (rounds 1-3) every
        prefill rebuilt the tables on the host and copied them from pageable memory — a device synchronisation that cost configs c4 / c5
        40-50 ms of idle GPU per step (profiles/r04_mvm_host_stall.md)."""
        key = (S, str(device))
        hit = self._rope.get(key)
        if hit is None:
            d = self.config.hidden_size // self.config.num_attention_heads
            hit = pack.rope_tables(S, d, self.config.rope_theta, device)
            if len(self._rope) >= 8:
                self._rope.pop(next(iter(self._rope)))
            self._rope[key] = hit
        return hit

    def prefill(self, inputs_embeds, attention_mask=None, cache=None):
        """inputs_embeds f32 [B,S,D]; attention_mask [B,S] (1 = token, right-padded) or None; cache: KVCache to fill.
        Returns (hidden f32 [B,S,D] after model.norm == hidden_states[-1], hidden in compute dtype [B*S,D])."""
        cfg = self.config
        dt = runtime.compute_dtype()
        layers = self.pack(dt)
        B, S, D = inputs_embeds.shape
        H = cfg.num_attention_heads
        hd = D // H
        dev = inputs_embeds.device
        x = inputs_embeds.reshape(B * S, D).float().clone()
        kv_len = None
        if attention_mask is not None:
            m = hip.host_mask(attention_mask).long()   # the assembler built it on the host: no D2H read, no stall
            if not bool((m[:, 1:] <= m[:, :-1]).all()):
                raise NotImplementedError("only right-padded attention masks occur on this path (st_llm.py:400-404)")
            if int(m.sum()) != m.numel():
                kv_len = hip.h2d(m.sum(dim=1).to(torch.int32), dev)
        cos, sin = self.rope(S, dev)
        if cache is not None:
            if kv_len is not None:
                raise NotImplementedError("KV cache needs equal-length sequences")
            assert cache.batch == B and cache.max_len >= S and cache.qkv[0].dtype == dt
            cache.len = S
        # the 32-layer loop is ONE call into the C ABI (stllm_llama_layers; == prefill_layer_by_layer, bit for bit); with a cache the
        # fused QKV rows of every layer are written straight into its cache buffer (rows (b, s) at b * max_len + s)
        if cache is not None:
            carr = hip.llama_layer_array(layers, cache)       # once per generate(): not worth caching (and it would pin the cache)
        else:
            if self._carr.get("layers") is not layers:
                self._carr = {"layers": layers, "carr": hip.llama_layer_array(layers)}
            carr = self._carr["carr"]
        if STACK_ENTRY:
            hip.llama_layers(x, layers, carr, B=B, S=S, n_heads=H, eps=cfg.rms_norm_eps, rope=(cos, sin), dtype=dt, kv_len=kv_len, cache=cache)
        else:
            self.prefill_layers_per_op(x, layers, B, S, cos, sin, kv_len, cache, dt)
        h16, h32 = hip.rmsnorm(x, self.norm.weight, cfg.rms_norm_eps, dtype=dt, want_f32=True)
        return h32.view(B, S, D), h16

    def prefill_layers_per_op(self, x, layers, B, S, cos, sin, kv_len, cache, dt):
        """The decoder-layer loop as one C-ABI call per op — what stllm_llama_layers issues from C.  Kept as the reference the stack
        entry point is tested against (-m gpu: bit-identical) and as the body the test-only CPU contract backend runs."""
        cfg = self.config
        D = cfg.hidden_size
        H = cfg.num_attention_heads
        hd = D // H
        for li_, pk in enumerate(layers):
            h, _ = hip.rmsnorm(x, pk["ln1"], cfg.rms_norm_eps, dtype=dt)
            if cache is None:
                qkv = hip.gemm(h, pk["wqkv"], dtype=dt, epilogue=hip.EPI_ROPE, rope=(cos, sin), rope_seq=S, rope_cols=2 * D, **_frag(pk, "wqkv"))
                strides = None
            else:  # the cache buffer is the GEMM's output: rows (b, s) at b*max_len + s
                qkv = cache.qkv[li_].view(B * cache.max_len, 3 * D)
                hip.gemm(h, pk["wqkv"], dtype=dt, epilogue=hip.EPI_ROPE, rope=(cos, sin), rope_seq=S, rope_cols=2 * D,
                         out=qkv, M=B * S, o_rows=(S, cache.max_len * 3 * D), **_frag(pk, "wqkv"))
                strides = (cache.max_len * 3 * D, 3 * D)
            a = hip.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], B=B, H=H, Sq=S, Skv=S, D=hd,
                              scale=hd ** -0.5, causal=True, kv_len=kv_len, q_strides=strides, k_strides=strides,
                              v_strides=strides)
            hip.gemm(a, pk["wo"], dtype=dt, epilogue=hip.EPI_RESID, resid=x)
            h, _ = hip.rmsnorm(x, pk["ln2"], cfg.rms_norm_eps, dtype=dt)
            g = hip.gemm(h, pk["wgu"], dtype=dt, epilogue=hip.EPI_SWIGLU, **_frag(pk, "wgu"))
            hip.gemm(g, pk["wdown"], dtype=dt, epilogue=hip.EPI_RESID, resid=x)
        return x

    def sp_layer_part(self, part, x, layers, li, qkv, s0, s1, cos_l, sin_l, dt, carr):
        """one half of a decoder layer on this rank's rows [s0, s1) (prefill_sp): part 0 = RMSNorm + QKV GEMM + RoPE into qkv[s0:s1], part 1 = attention over
        the s1 rows + o_proj + RMSNorm + gate/up + down.  carr: the C-side layer table (one stllm_llama_layer_sp call) or None (the per-op body)."""
        cfg = self.config
        D, H = cfg.hidden_size, cfg.num_attention_heads
        hd = D // H
        pk = layers[li]
        if carr is not None:
            return hip.llama_layer_sp(x, carr, li, qkv, s0=s0, s1=s1, part=part, n_heads=H, eps=cfg.rms_norm_eps, rope=(cos_l, sin_l), dtype=dt,
                                      inter=pk["wgu"].shape[0] // 2)
        if part == 0:
            h, _ = hip.rmsnorm(x, pk["ln1"], cfg.rms_norm_eps, dtype=dt)
            hip.gemm(h, pk["wqkv"], dtype=dt, epilogue=hip.EPI_ROPE, rope=(cos_l, sin_l), rope_seq=s1 - s0, rope_cols=2 * D, out=qkv[s0:s1], **_frag(pk, "wqkv"))
            return x
        a = hip.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], B=1, H=H, Sq=s1, Skv=s1, D=hd, scale=hd ** -0.5, causal=True)
        hip.gemm(a[s0:s1], pk["wo"], dtype=dt, epilogue=hip.EPI_RESID, resid=x)
        h, _ = hip.rmsnorm(x, pk["ln2"], cfg.rms_norm_eps, dtype=dt)
        g = hip.gemm(h, pk["wgu"], dtype=dt, epilogue=hip.EPI_SWIGLU, **_frag(pk, "wgu"))
        hip.gemm(g, pk["wdown"], dtype=dt, epilogue=hip.EPI_RESID, resid=x)
        return x

    def prefill_sp(self, inputs_embeds, sp):
        """Sequence-parallel prefill of ONE sequence inside a clip team (stllm_amd.parallel, round 5): this rank runs the decoder layers on the
        positions [s0, s1) = parallel.sp_row_ranges(S, k)[j] only.  Causal attention makes the team's dependency one-directional: per layer the
        K | V rows of the members before this one arrive point-to-point (posted before this rank's own QKV GEMM, awaited in front of its
        attention), its own K | V rows leave for the members behind it right after the QKV GEMM — the transfers ride under the o_proj / MLP
        GEMMs of the sender and the RMSNorm / QKV GEMM of the receiver.  No kernel knows about any of this: the fused QKV buffer simply has rows
        [0, s1) — foreign K | V, own q | k | v — and the attention runs causal over s1 x s1 (the s0 foreign query rows are zero, their outputs unused:
        +7 us per layer at S = 580, k = 2, against a second set of attention kernels with a query offset).
        sp: dict(index=j, size=k, ranks=[global ranks of the team], rank=this rank, group=None, mailbox=None | parallel.Mailbox).
        inputs_embeds f32 [1, S, D] (every member assembles the whole sequence: one 9.5 MB gather).  Returns (hidden f32 [1, s1 - s0, D] after
        model.norm, the same rows in the compute dtype, (s0, s1))."""
        from .. import parallel
        cfg = self.config
        dt = runtime.compute_dtype()
        layers = self.pack(dt)
        B, S, D = inputs_embeds.shape
        if B != 1:
            raise NotImplementedError("sequence-parallel prefill: one sequence per team (stllm_amd.parallel.TeamPlan)")
        H = cfg.num_attention_heads
        hd = D // H
        dev = inputs_embeds.device
        j, k, ranks, me, group, box = sp["index"], sp["size"], sp["ranks"], sp["rank"], sp.get("group"), sp.get("mailbox")
        rr = parallel.sp_row_ranges(S, k)
        s0, s1 = rr[j]
        x = inputs_embeds.reshape(S, D)[s0:s1].float().clone()
        cos, sin = self.rope(S, dev)
        cos_l, sin_l = cos[s0:s1], sin[s0:s1]
        qkv = torch.empty((s1, 3 * D), device=dev, dtype=dt)
        if s0 > 0:
            qkv[:s0, :D].zero_()          # query columns of the foreign rows: never written again, their attention outputs are never read
        n_loc = s1 - s0
        if n_loc == 0:                     # more members than 32-row groups: this member has no rows (and nobody waits for any from it)
            return x.view(1, 0, D), x.to(dt), (s0, s1)
        pending = []                       # send handles + the packed buffers they read: kept until the end of the prefill
        # round 6 (VERDICT r05 missing #4): the two halves of a layer — up to the QKV rows, and from the attention on — are ONE C call each
        # (stllm_llama_layer_sp) instead of 2 + 5 per-op calls; the split verify mode and STACK_ENTRY = 0 keep the per-op body (bit-identical)
        carr = None
        if STACK_ENTRY and not runtime.gemm_split():
            if self._carr.get("layers") is not layers:
                self._carr = {"layers": layers, "carr": hip.llama_layer_array(layers)}
            carr = self._carr["carr"]
        for li_, pk in enumerate(layers):
            recvs = [(torch.empty((rr[i][1] - rr[i][0], 2 * D), device=dev, dtype=dt), ranks[i], ("kv", li_)) for i in range(j) if rr[i][1] > rr[i][0]]
            works = parallel.p2p_exchange([], recvs, me, group, box) if recvs else []
            self.sp_layer_part(0, x, layers, li_, qkv, s0, s1, cos_l, sin_l, dt, carr)
            later = [i for i in range(j + 1, k) if rr[i][1] > rr[i][0]]
            if later:
                kv = qkv[s0:s1, D:].contiguous()
                pending.append((kv, parallel.p2p_exchange([(kv, ranks[i], ("kv", li_)) for i in later], [], me, group, box)))
            for w in works:
                w.wait()
            for (buf, _, _), i in zip(recvs, [i for i in range(j) if rr[i][1] > rr[i][0]]):
                qkv[rr[i][0]:rr[i][1], D:].copy_(buf)
            self.sp_layer_part(1, x, layers, li_, qkv, s0, s1, cos_l, sin_l, dt, carr)
        for _, ws in pending:
            for w in ws:
                w.wait()
        h16, h32 = hip.rmsnorm(x, self.norm.weight, cfg.rms_norm_eps, dtype=dt, want_f32=True)
        return h32.view(1, n_loc, D), h16, (s0, s1)

    def decode_step(self, x_new, cache):
        """One token per sequence: x_new f32 [B,1,D] (embedding of the token at position cache.len).  Appends its K/V to the
        cache and returns (hidden f32 [B,1,D] after model.norm, hidden compute-dtype [B,D])."""
        cfg = self.config
        dt = runtime.compute_dtype()
        layers = self.pack(dt)
        B, _, D = x_new.shape
        H = cfg.num_attention_heads
        hd = D // H
        pos = cache.len
        assert pos < cache.max_len, "KV cache full"
        cos, sin = self.rope(cache.max_len, x_new.device)
        cpos, spos = cos[pos:pos + 1], sin[pos:pos + 1]
        x = x_new.reshape(B, D).float().clone()
        ML3 = cache.max_len * 3 * D
        # 16-bit modes, <= 8 rows: both RMSNorms ride inside the GEMV that consumes them (2 launches per layer fewer)
        # (every workgroup of the GEMV recomputes the norm of all its rows: worth it for 1-2 rows — 3.70 -> 3.29 ms/token —, a loss for
        #  the 5 rows of beam search, where 2752 workgroups x 6 staged rows re-read 0.5 GB through L2)
        fuse = dt != torch.float32 and B <= FUSE_NORM_ROWS
        for li_, pk in enumerate(layers):
            row = cache.qkv[li_][:, pos]                                   # [B, 3D] view, row stride max_len*3D
            if fuse:
                hip.gemm(None, pk["wqkv"], dtype=dt, epilogue=hip.EPI_ROPE, rope=(cpos, spos), rope_seq=1, rope_cols=2 * D, out=row,
                         a_norm=(x, pk["ln1"], cfg.rms_norm_eps))
            else:
                h, _ = hip.rmsnorm(x, pk["ln1"], cfg.rms_norm_eps, dtype=dt)
                hip.gemm(h, pk["wqkv"], dtype=dt, epilogue=hip.EPI_ROPE, rope=(cpos, spos), rope_seq=1, rope_cols=2 * D, out=row)
            full = cache.qkv[li_].view(B * cache.max_len, 3 * D)
            a = hip.attention(row[:, :D], full[:, D:2 * D], full[:, 2 * D:], B=B, H=H, Sq=1, Skv=pos + 1, D=hd,
                              scale=hd ** -0.5, causal=False, q_strides=(ML3, 3 * D), k_strides=(ML3, 3 * D),
                              v_strides=(ML3, 3 * D))
            hip.gemm(a, pk["wo"], dtype=dt, epilogue=hip.EPI_RESID, resid=x)
            if fuse:
                g = hip.gemm(None, pk["wgu"], dtype=dt, epilogue=hip.EPI_SWIGLU, a_norm=(x, pk["ln2"], cfg.rms_norm_eps))
            else:
                h, _ = hip.rmsnorm(x, pk["ln2"], cfg.rms_norm_eps, dtype=dt)
                g = hip.gemm(h, pk["wgu"], dtype=dt, epilogue=hip.EPI_SWIGLU)
            hip.gemm(g, pk["wdown"], dtype=dt, epilogue=hip.EPI_RESID, resid=x)
        cache.len = pos + 1
        h16, h32 = hip.rmsnorm(x, self.norm.weight, cfg.rms_norm_eps, dtype=dt, want_f32=True)
        return h32.view(B, 1, D), h16

    def new_cache(self, batch, max_len, device):
        return KVCache(len(self.layers), batch, max_len, self.config.hidden_size, runtime.compute_dtype(), device)

    def forward(self, input_ids=None, attention_mask=None, inputs_embeds=None, use_cache=False,
                output_hidden_states=False, return_dict=True, past_key_values=None, sp=None, **kw):
        if inputs_embeds is None:
            inputs_embeds = self.embed_tokens(input_ids)
        if sp is not None:   # this rank's position range of a sequence-parallel prefill (no cache: generate() prefills on one GPU)
            if use_cache or past_key_values is not None:
                raise NotImplementedError("sequence-parallel prefill does not fill a KV cache")
            hidden, h16, rows = self.prefill_sp(inputs_embeds, sp)
            out = Output(last_hidden_state=hidden, past_key_values=None, hidden_states=(hidden,) if output_hidden_states else None, attentions=None)
            out._h16, out._sp_rows = h16, rows
            return out
        if isinstance(past_key_values, KVCache) and past_key_values.len > 0:   # decode step(s), one token at a time
            hs = []
            for t in range(inputs_embeds.shape[1]):
                hidden, h16 = self.decode_step(inputs_embeds[:, t:t + 1], past_key_values)
                hs.append(hidden)
            hidden = torch.cat(hs, dim=1) if len(hs) > 1 else hs[0]
            out = Output(last_hidden_state=hidden, past_key_values=past_key_values,
                         hidden_states=(hidden,) if output_hidden_states else None, attentions=None)
            out._h16 = h16
            return out
        cache = None
        if use_cache:
            cache = past_key_values if isinstance(past_key_values, KVCache) else \
                self.new_cache(inputs_embeds.shape[0], min(cfg_max_len(self.config), inputs_embeds.shape[1] + 512), inputs_embeds.device)
        hidden, h16 = self.prefill(inputs_embeds, attention_mask, cache=cache)
        out = Output(last_hidden_state=hidden, past_key_values=cache,
                     hidden_states=(hidden,) if output_hidden_states else None, attentions=None)
        out._h16 = h16
        return out


class LlamaForCausalLM(nn.Module):
    def __init__(self, config, device=None):
        super().__init__()
        self.config = config
        self.model = LlamaModel(config, device)
        self.vocab_size = config.vocab_size
        self.lm_head = Linear(config.hidden_size, config.vocab_size, bias=False, device=device)
        self._lm_packed = {}

    def lm_weight(self, dt):
        ver = (self.lm_head.weight._version, self.lm_head.weight.data_ptr(), runtime.gemm_split())
        hit = self._lm_packed.get(dt)
        if hit is None or hit[0] != ver:
            hit = (ver, pack.pad_rows(pack.linear(self.lm_head.weight, dt), 128))
            self._lm_packed = {dt: hit}
        return hit[1]

    def logits_from(self, h16, B, S):
        dt = runtime.compute_dtype()
        lg = hip.gemm(h16, self.lm_weight(dt), dtype=dt, out_f32=True)
        return lg.view(B, S, -1)[..., : self.vocab_size]
