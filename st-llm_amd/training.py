"""Training step of `STLLMForCausalLM.forward(samples)`: loss + gradients of everything the reference trains.

What the reference does (train_hf.py -> HF Trainer -> DeepSpeed ZeRO, config/*_stllm_qa.yaml: `freeze_LLM: False`, bf16,
`use_grad_checkpoint: True`, AdamW lr 2e-5): autograd through st_llm.py:116-146 (shifted CE + loss_mvm) with the ViT and the
Q-Former frozen (st_llm.py:257-296), i.e. gradients for
    llama_proj, down_proj / up_proj (video_input == "residual"), mvm_decoder, and the whole LLM (embed_tokens, 32 layers, norm,
    lm_head).
There is no autograd here: the backward graph is written out explicitly on the HIP C ABI.

MI355X-first choices:
  * no activation recomputation — 288 GB keeps every per-layer activation of a 16 x 576-token batch (about 1.2 GB per layer in
    bf16) resident; the reference needs gradient checkpointing on 80 GB parts;
  * every dgrad / wgrad is the SAME `stllm_gemm` (C = A @ W^T, both operands K-contiguous) fed with transposed operands
    (`stllm_transpose`, zero-padded to the K granularity): dX = dY @ W -> gemm(dY, W^T);  dW = dY^T @ X -> gemm(dY^T, X^T),
    fp32 out;
  * the residual-stream gradient stays fp32 and is accumulated in place by the RESID epilogue (like the forward stream); the
    other inter-op gradients travel in the compute dtype (what autocast-bf16 does in the reference);
  * gate/up are stored raw (STORE epilogue) and SiLU(gate)*up is a separate elementwise kernel in training, so the backward
    needs no recomputation of the largest GEMM;
  * q/k gradients are rotated back by `stllm_rope_bwd` in the packed head layout, weight gradients are produced in the packed
    layouts of pack.py and un-permuted once per step into the reference's parameter layout (index permutations only).
BT-Adapter parameters (model_type *_btadapter — 4 of the 5 shipped training configs): their gradient goes back through llama_proj,
the frozen Q-Former and ln_vision into the adapter branch (training_vision.py).  The adapter blocks' train-mode stochastic depth
(DropPath 0.1, eva_btadapter.py:259) is applied when per-sample factors are passed (`drop_path=`); default: the deterministic step.

Layout of the result: {reference parameter name: fp32 gradient in the reference's layout}.
"""
import torch

from . import hip, pack, runtime


# ---- small helpers --------------------------------------------------------------------------------------------------
def _kgran(dt):
    return 32 if dt == torch.float32 else 64


def linear_bwd(dy, x, w, dt, *, need_dx=True, dx_resid=None, dx_f32=False, need_dw=True, dw_out=None):
    """y = x @ w^T.  dy [M,N], x [M,K], w [N,K] — all in the compute dtype `dt`.
    Returns (dx, dw): dx [M,K] (compute dtype, or fp32 when dx_f32; accumulated into `dx_resid` fp32 [M,K] when given),
    dw fp32 [N,K] — written straight into `dw_out` (a contiguous fp32 [N,K], e.g. the optimizer's slice of its flat gradient buffer) when given."""
    dx = dw = None
    if need_dx:
        wt = hip.transpose(w, pad=_kgran(dt))                    # [K, N]
        if dx_resid is not None:
            dx = hip.gemm(dy, wt, dtype=dt, epilogue=hip.EPI_RESID, resid=dx_resid)
        else:
            dx = hip.gemm(dy, wt, dtype=dt, out_f32=dx_f32)
    if need_dw:
        dyt = hip.transpose(dy, pad=_kgran(dt))                  # [N, Mp]
        xt = hip.transpose(x, pad=_kgran(dt))                    # [K, Mp]
        if dw_out is not None and (tuple(dw_out.shape) != (dy.shape[1], x.shape[1]) or not dw_out.is_contiguous()):
            dw_out = None                                        # a padded weight (lm_head of a 32001-token vocabulary): the caller slices and copies
        dw = hip.gemm(dyt, xt, dtype=dt, out_f32=True, out=dw_out)   # [N, K]
    return dx, dw


def unpack_qkv_grad(dw, n_heads, out=(None, None, None)):
    """inverse of pack.llama_qkv on a [3D, D] gradient -> (dq, dk, dv) in the reference's row order; `out`: contiguous fp32 [D, D]
    destinations (the optimizer's gradient slices) or None each"""
    D = dw.shape[0] // 3
    perm = pack.rope_head_perm(n_heads, D // n_heads, dw.device)
    dq, dk, dv = (torch.empty_like(dw[:D]) if o is None else o for o in out)
    dq[perm] = dw[:D]
    dk[perm] = dw[D:2 * D]
    dv.copy_(dw[2 * D:])
    return dq, dk, dv


def unpack_gate_up_grad(dw, out=(None, None)):
    """inverse of pack.llama_gate_up on a [2I, D] gradient -> (dgate, dup); `out` as in unpack_qkv_grad"""
    n2, k = dw.shape
    v = dw.view(n2 // 64, 2, 32, k)
    dg, du = (torch.empty((n2 // 2, k), device=dw.device, dtype=dw.dtype) if o is None else o for o in out)
    dg.view(n2 // 64, 32, k).copy_(v[:, 0])
    du.view(n2 // 64, 32, k).copy_(v[:, 1])
    return dg, du


# ---- the LLM: forward that keeps activations, and its backward ---------------------------------------------------------
class LlamaTape:
    """activations of one taped prefill (compute dtype unless noted)"""

    def __init__(self):
        self.layers = []          # per layer: dict(x0 f32, h1, qkv, a, x1 f32, h2, gu, g)
        self.x_final = None       # f32 [M,D] residual stream entering model.norm
        self.h16 = self.h32 = None
        self.B = self.S = 0
        self.kv_len = None
        self.cos = self.sin = None


def llama_forward_taped(lm, inputs_embeds, attention_mask=None):
    """LlamaModel.prefill (models/llama.py) with every activation kept.  Returns (h32 [B,S,D], h16 [B*S,D], tape)."""
    cfg = lm.config
    dt = runtime.compute_dtype()
    lm.wd_frag = False   # (the weights are re-packed after every optimizer step: no second, fragment-major copy of wqkv / wgu for kernels this path does not call)
    packs = lm.pack(dt)
    B, S, D = inputs_embeds.shape
    H = cfg.num_attention_heads
    hd = D // H
    dev = inputs_embeds.device
    t = LlamaTape()
    t.B, t.S = B, S
    x = inputs_embeds.reshape(B * S, D).float().clone()
    if attention_mask is not None:
        m = attention_mask.to("cpu").long()
        if not bool((m[:, 1:] <= m[:, :-1]).all()):
            raise NotImplementedError("only right-padded attention masks occur on this path (st_llm.py:400-404)")
        if int(m.sum()) != m.numel():
            t.kv_len = m.sum(dim=1).to(torch.int32).to(dev)
    t.cos, t.sin = lm.rope(S, dev)
    for pk in packs:
        rec = dict(x0=x.clone())
        rec["h1"], _ = hip.rmsnorm(x, pk["ln1"], cfg.rms_norm_eps, dtype=dt)
        rec["qkv"] = hip.gemm(rec["h1"], pk["wqkv"], dtype=dt, epilogue=hip.EPI_ROPE, rope=(t.cos, t.sin), rope_seq=S, rope_cols=2 * D)
        q = rec["qkv"]
        rec["a"] = hip.attention(q[:, :D], q[:, D:2 * D], q[:, 2 * D:], B=B, H=H, Sq=S, Skv=S, D=hd, scale=hd ** -0.5, causal=True,
                                 kv_len=t.kv_len)
        hip.gemm(rec["a"], pk["wo"], dtype=dt, epilogue=hip.EPI_RESID, resid=x)
        rec["x1"] = x.clone()
        rec["h2"], _ = hip.rmsnorm(x, pk["ln2"], cfg.rms_norm_eps, dtype=dt)
        rec["gu"] = hip.gemm(rec["h2"], pk["wgu"], dtype=dt)                 # raw [32 gate | 32 up] groups
        rec["g"] = hip.swiglu(rec["gu"])
        hip.gemm(rec["g"], pk["wdown"], dtype=dt, epilogue=hip.EPI_RESID, resid=x)
        t.layers.append(rec)
    t.x_final = x
    t.h16, t.h32 = hip.rmsnorm(x, lm.norm.weight, cfg.rms_norm_eps, dtype=dt, want_f32=True)
    return t.h32.view(B, S, D), t.h16, t


def _no_sink(name):
    return None


def llama_backward(lm, tape, d_h16=None, d_h32=None, prefix="model.", sink=_no_sink):
    """Backward of llama_forward_taped.  d_h16: gradient w.r.t. the compute-dtype output of model.norm ([M,D], compute dtype),
    d_h32: w.r.t. its fp32 twin ([M,D] fp32); either may be None.  Returns (d_inputs_embeds fp32 [M,D], grads dict).
    sink(name) -> the fp32 destination of that parameter's gradient (AdamW.grad_sink: a slice of the flat gradient buffer) or None:
    the 25.8 GB of layer weight gradients of a 7B step are then written once, by the wgrad GEMM / the un-permutation, instead of being
    produced in a temporary and copied by AdamW.step."""
    cfg = lm.config
    dt = runtime.compute_dtype()
    packs = lm.pack(dt)
    B, S = tape.B, tape.S
    D = cfg.hidden_size
    H = cfg.num_attention_heads
    hd = D // H
    M = B * S
    grads = {}
    dev = tape.x_final.device
    # ---- model.norm ---------------------------------------------------------------------------------------------
    dx = torch.zeros((M, D), device=dev, dtype=torch.float32)
    dgam = None
    for dy in (d_h16, d_h32):
        if dy is not None:
            d = hip.rmsnorm_bwd(tape.x_final, lm.norm.weight, cfg.rms_norm_eps, dy, dx, accumulate=True)
            dgam = d if dgam is None else dgam + d
    grads[prefix + "norm.weight"] = dgam
    # ---- layers, last to first ---------------------------------------------------------------------------------------
    for li in range(len(packs) - 1, -1, -1):
        pk, rec = packs[li], tape.layers[li]
        lp = f"{prefix}layers.{li}."
        dx16 = hip.cast_rows(dx, dt)
        # x2 = x1 + g @ Wdown^T
        dg, dw = linear_bwd(dx16, rec["g"], pk["wdown"], dt, dw_out=sink(lp + "mlp.down_proj.weight"))
        grads[lp + "mlp.down_proj.weight"] = dw
        dgu = hip.swiglu_bwd(rec["gu"], dg)
        dh2, dw = linear_bwd(dgu, rec["h2"], pk["wgu"], dt)
        grads[lp + "mlp.gate_proj.weight"], grads[lp + "mlp.up_proj.weight"] = unpack_gate_up_grad(
            dw, (sink(lp + "mlp.gate_proj.weight"), sink(lp + "mlp.up_proj.weight")))
        grads[lp + "post_attention_layernorm.weight"] = hip.rmsnorm_bwd(rec["x1"], pk["ln2"], cfg.rms_norm_eps, dh2, dx, accumulate=True)
        # x1 = x0 + a @ Wo^T          (dx is now dL/dx1)
        dx16 = hip.cast_rows(dx, dt)
        da, dw = linear_bwd(dx16, rec["a"], pk["wo"], dt, dw_out=sink(lp + "self_attn.o_proj.weight"))
        grads[lp + "self_attn.o_proj.weight"] = dw
        qkv = rec["qkv"]
        dqkv = torch.empty_like(qkv)
        hip.attention_bwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], rec["a"], da, dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:],
                          B=B, H=H, S=S, D=hd, scale=hd ** -0.5, causal=True, kv_len=tape.kv_len)
        hip.rope_bwd(dqkv, tape.cos, tape.sin, rope_seq=S, rope_cols=2 * D)
        dh1, dw = linear_bwd(dqkv, rec["h1"], pk["wqkv"], dt)
        a = lp + "self_attn."
        grads[a + "q_proj.weight"], grads[a + "k_proj.weight"], grads[a + "v_proj.weight"] = unpack_qkv_grad(
            dw, H, (sink(a + "q_proj.weight"), sink(a + "k_proj.weight"), sink(a + "v_proj.weight")))
        grads[lp + "input_layernorm.weight"] = hip.rmsnorm_bwd(rec["x0"], pk["ln1"], cfg.rms_norm_eps, dh1, dx, accumulate=True)
    return dx, grads


# ---- the whole training forward + backward ------------------------------------------------------------------------------
PHASES = None   # set to a list to collect (phase name, torch.cuda.Event) marks of the next loss_and_grads call (tools/train_bench.py)


def _mark(name):
    if PHASES is not None and torch.cuda.is_available():
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        PHASES.append((name, ev))


def loss_and_grads(model, samples, freeze_btadapter=False, drop_path=None, sink=None):
    """model: STLLMForCausalLM.  Returns (loss fp32 scalar tensor, loss_mvm or None, grads {reference name: fp32 tensor}).
    sink: AdamW.grad_sink() — the LLM's weight gradients (lm_head, 32 x q/k/v/o/gate/up/down, the embedding table) are then produced IN the
    optimizer's flat gradient buffer and the returned tensors alias it (train_step does this); None: every gradient is its own tensor.
    On the eva_btadapter_g backbone the reference also trains the `visual_encoder.BTAdapter*` parameters (st_llm.py:257-261): their
    gradient is carried back through llama_proj, the frozen Q-Former and ln_vision into the adapter branch (training_vision.py);
    freeze_btadapter=True skips that and treats the adapter as frozen.  drop_path: the adapter blocks' train-mode stochastic depth as
    per-sample factors (training_vision.drop_path_factors / btadapter_forward_taped); None = the deterministic step."""
    lmw = model                                  # lm_head owner
    lm = model.model                             # STLLMLlamaModel (LlamaModel + stllm_model)
    sm = lm.stllm_model
    train_adapter = sm.vit_model != "eva_clip_g" and not freeze_btadapter
    if sm.frame_parallel is not None:
        raise NotImplementedError("training is data-parallel (one micro-batch per rank); frame-parallel is the inference path")
    if runtime.gemm_split():
        # compute_dtype() is fp32 there but the packed weights are split bf16 [N, 3 K]: the backward's transposes / GEMMs would see a "weight" of the
        # wrong width.  The split mode is an inference verify mode; train in bf16 / fp16 / fp32.
        raise NotImplementedError('training does not run in the "bf16x3" split mode: use runtime.set_compute_dtype("bf16" | "fp16" | "fp32")')
    dt = runtime.compute_dtype()
    cfg = lm.config
    D = cfg.hidden_size
    _mark("start")
    sm._tape = tape = {"want_vision": train_adapter, "drop_path": drop_path if train_adapter else None}
    try:
        inputs_embeds, attention_mask, un_e, un_a, labels = sm(samples)
    finally:
        sm._tape = None
    B, S, _ = inputs_embeds.shape
    dev = inputs_embeds.device
    _mark("vision forward + assembly")
    h32, h16, lt = llama_forward_taped(lm, inputs_embeds, attention_mask)
    _mark("LLM forward")
    grads = {}
    # ---- shifted CE (st_llm.py:125-135) ---------------------------------------------------------------------------
    Wlm = lmw.lm_weight(dt)                                                  # [Vp, D]
    logits = hip.gemm(h16, Wlm, dtype=dt, out_f32=True)                      # [M, Vp]
    V = lmw.vocab_size
    shift = torch.full_like(labels, -100)
    shift[:, :-1] = labels[:, 1:]
    lab = shift.reshape(-1).to(torch.int32)
    n_valid = int((shift != -100).sum().clamp(min=1))
    rows = hip.cross_entropy_rows(logits[:, :V], lab)
    loss = rows.sum() / n_valid
    dlogits = hip.cross_entropy_bwd(logits, lab, 1.0 / n_valid, dtype=dt, vocab=V)
    sink = sink or _no_sink
    d_h16, dw = linear_bwd(dlogits, h16, Wlm, dt, dw_out=sink("lm_head.weight"))
    grads["lm_head.weight"] = dw[:V]
    # ---- MVM branch (st_llm.py:71-91) -------------------------------------------------------------------------------
    loss_mvm = None
    d_h32 = None
    if un_e is not None:
        img_start = 0 if sm.qformer_text_input else 8
        Lk = sm.mask_img_len
        rows_a = (torch.arange(B).view(B, 1) * S + img_start + torch.arange(Lk).view(1, Lk)).reshape(-1).to(torch.int32).to(dev)
        a_in = hip.gather_rows(h32.reshape(B * S, D), rows_a)                # f32 [B*Lk, D]
        p = "model.stllm_model.mvm_decoder."
        has_dec = hasattr(sm, "mvm_decoder")
        if has_dec:
            dec = sm.mvm_decoder
            w, b = dec.head.packed(dt)
            a16 = hip.cast_rows(a_in, dt)
            lin = hip.gemm(a16, w, dtype=dt, bias=b, out_f32=True)
            _, a = hip.layernorm(lin, dec.norm.weight, dec.norm.bias, dec.norm.eps, dtype=torch.float32, want_t=False, want_f32=True)
        else:
            a = a_in
        un_out, _ = lm.prefill(un_e, un_a)                                   # the (detached) target pass, st_llm.py:77-83
        S2 = un_out.shape[1]
        keep = ~sm.mask.squeeze(1)
        pos = torch.stack([torch.nonzero(keep[b]).flatten() for b in range(B)])
        idx_b = (torch.arange(B).view(B, 1) * S2 + img_start + pos).reshape(-1).to(torch.int32).to(dev)
        ub = un_out.reshape(B * S2, D)
        n = B * Lk
        loss_mvm = hip.cosine_rows(a, ub, None, idx_b, n_rows=n).mean()
        loss = loss + loss_mvm
        da = hip.cosine_rows_bwd(a, ub, None, idx_b, n_rows=n, scale=1.0 / n)
        if has_dec:
            dlin, dgam, dbet = hip.layernorm_bwd(lin, dec.norm.weight, dec.norm.eps, da)
            grads[p + "norm.weight"], grads[p + "norm.bias"] = dgam, dbet
            dlin16 = hip.cast_rows(dlin, dt)
            da_in, dw = linear_bwd(dlin16, a16, w, dt, dx_f32=True)
            grads[p + "head.weight"], grads[p + "head.bias"] = dw, hip.colsum(dlin16)
        else:
            da_in = da
        d_h32 = torch.zeros((B * S, D), device=dev, dtype=torch.float32)
        hip.scatter_add_rows(da_in, rows_a, d_h32)
    # ---- the LLM ----------------------------------------------------------------------------------------------------
    _mark("lm_head + losses (+ MVM target pass)")
    d_emb, g_llm = llama_backward(lm, lt, d_h16, d_h32, sink=sink)
    _mark("LLM backward")
    grads.update(g_llm)
    # ---- token-block assembly: gather_rows^T (visual rows | embedding-table rows) -------------------------------------
    d_vis = torch.zeros((tape["vis_rows"], D), device=dev, dtype=torch.float32)
    d_table = sink("model.embed_tokens.weight")
    d_table = torch.zeros_like(lm.embed_tokens.weight, dtype=torch.float32) if d_table is None else d_table.zero_()
    hip.scatter_add_rows(d_emb, tape["gather_idx"][0], d_vis, d_table)        # ([1], the un-masked assembly, carries no gradient)
    grads["model.embed_tokens.weight"] = d_table
    # ---- pooling (st_llm.py:463-478) --------------------------------------------------------------------------------
    p = "model.stllm_model."
    if tape.get("pooled", False) and "pool_shape" in tape:
        Bp, T, Lq, _ = tape["pool_shape"]
        if sm.video_input == "all":
            d_tok = d_vis                                                     # a view change only
        elif sm.video_input == "mean":
            d_tok = torch.zeros((Bp, T, Lq * D), device=dev, dtype=torch.float32)
            hip.bcast_add_t(d_tok, d_vis.view(Bp, Lq * D), 1.0 / T)
            d_tok = d_tok.view(Bp * T * Lq, D)
        elif sm.video_input == "residual":
            d_tok = torch.zeros((Bp * T * Lq, D), device=dev, dtype=torch.float32)
            hip.scatter_add_rows(d_vis, tape["pool_idx"], d_tok)              # local = emb[:, idx]
            d_gg = torch.zeros((Bp * Lq, D), device=dev, dtype=torch.float32)
            hip.scatter_add_rows(d_vis, tape["pool_idx_add"], d_gg)           # + global, broadcast over the R copies
            wd, _ = sm.down_proj.packed(dt)
            wu, _ = sm.up_proj.packed(dt)
            d_gg16 = hip.cast_rows(d_gg, dt)
            dh, dw = linear_bwd(d_gg16, tape["pool_h"], wu, dt)
            grads[p + "up_proj.weight"], grads[p + "up_proj.bias"] = dw, hip.colsum(d_gg16)
            dh = hip.relu_bwd(dh, tape["pool_h"])
            d_mean, dw = linear_bwd(dh, tape["pool_g16"], wd, dt, dx_f32=True)
            grads[p + "down_proj.weight"], grads[p + "down_proj.bias"] = dw, hip.colsum(dh)
            d_tok = d_tok.view(Bp, T, Lq * D)
            hip.bcast_add_t(d_tok, d_mean.view(Bp, Lq * D), 1.0 / T)
            d_tok = d_tok.view(Bp * T * Lq, D)
        else:
            d_tok = d_vis
    else:
        d_tok = d_vis
    # ---- projector (st_llm.py:368): inputs_llama = hq @ W^T + b; the Q-Former below it is frozen ---------------------------
    w, _ = sm.llama_proj.packed(dt)
    d_tok16 = hip.cast_rows(d_tok, dt)
    d_hq, dw = linear_bwd(d_tok16, tape["hq16"], w, dt, need_dx=train_adapter, dx_f32=True)
    grads[p + "llama_proj.weight"], grads[p + "llama_proj.bias"] = dw, hip.colsum(d_tok16)
    # ---- BT-Adapter: back through the frozen Q-Former and ln_vision into the adapter branch (training_vision.py) ------------------
    if train_adapter:
        from . import training_vision
        d_enc = training_vision.qformer_backward(sm.Qformer.bert, tape["qf_tape"], d_hq)
        d_feats, _, _ = hip.layernorm_bwd(tape["feats"], sm.ln_vision.weight, sm.ln_vision.eps, d_enc)
        grads.update(training_vision.btadapter_backward(sm.visual_encoder, tape["bt_tape"], d_feats, p + "visual_encoder."))
    _mark("assembly / pooling / projector / vision backward")
    return loss, loss_mvm, grads


# ---- optimizer --------------------------------------------------------------------------------------------------------
def trainable_parameters(model, freeze_btadapter=False):
    """(name, parameter) of what the reference leaves trainable (st_llm.py:182-186, 257-296 with the shipped configs): everything
    but the ViT, ln_vision, the Q-Former and its query tokens — the `BTAdapter*` parameters inside visual_encoder stay trainable
    (st_llm.py:259) unless freeze_btadapter."""
    frozen = ("model.stllm_model.visual_encoder", "model.stllm_model.ln_vision", "model.stllm_model.Qformer",
              "model.stllm_model.query_tokens")
    seen = set()
    for n, prm in model.named_parameters():
        if (n.startswith(frozen) and ("BTAdapter" not in n or freeze_btadapter)) or id(prm) in seen:
            continue
        seen.add(id(prm))
        yield n, prm


class AdamW:
    """torch.optim.AdamW semantics (what HF Trainer builds for the reference: lr 2e-5, betas (0.9, 0.999), eps 1e-8,
    weight_decay 0., max_grad_norm 1.0) on ONE flat fp32 buffer per state, sharded ZeRO-1 style when a process group is given:

        grads --reduce-scatter(avg)--> each rank's 1/N slice --stllm_adamw--> its slice of the masters --all-gather--> everyone.

    Masters / m / v of a 7B model: 28 + 56 GB un-sharded (fits one 288 GB MI355X); with N ranks each holds 1/N of m and v.
    The flat buffer is padded to a multiple of N * 64 elements; parameters alias slices of it, so the model sees the update."""

    def __init__(self, named_params, lr=2e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_grad_norm=1.0, group=None,
                 world_size=1, rank=0):
        self.names, self.params = zip(*named_params)
        self.lr, self.betas, self.eps, self.wd, self.max_norm = lr, betas, eps, weight_decay, max_grad_norm
        self.group, self.world, self.rank = group, world_size, rank
        self.step_no = 0
        self.steps = [0] * len(self.params)   # torch.optim.AdamW keeps state['step'] PER PARAMETER: it only advances when the parameter has a gradient
        n = sum(p.numel() for p in self.params)
        gran = world_size * 64
        self.n, self.n_pad = n, (n + gran - 1) // gran * gran
        dev = self.params[0].device
        self.flat = torch.zeros(self.n_pad, device=dev, dtype=torch.float32)
        self.offsets = []
        off = 0
        for p in self.params:                       # masters alias the flat buffer
            self.flat[off: off + p.numel()] = p.data.reshape(-1)
            p.data = self.flat[off: off + p.numel()].view(p.shape)
            self.offsets.append(off)
            off += p.numel()
        self.shard = self.n_pad // world_size
        self.m = torch.zeros(self.shard, device=dev, dtype=torch.float32)
        self.v = torch.zeros(self.shard, device=dev, dtype=torch.float32)
        self.gflat = torch.zeros(self.n_pad, device=dev, dtype=torch.float32)
        self._slot = {name: (off, p.shape) for name, off, p in zip(self.names, self.offsets, self.params)}

    def grad_sink(self):
        """name -> this parameter's slice of the flat gradient buffer, shaped like the parameter (None for names that are not optimised):
        a producer that writes a gradient there (training.loss_and_grads(sink=...)) saves step() the copy.
        CONTRACT: tensors produced through the sink ALIAS this buffer.  They are this step's gradients until the next step() — which
        averages them over the ranks in place — or the next producer run, whichever comes first; keep `.clone()`s to keep values.  A grads dict whose
        aliasing entries were already consumed by a step() is refused by the next step() instead of silently applying whatever the buffer holds now."""
        self._sink_live = True      # a producer is about to fill the buffer: its aliasing gradients are valid for ONE step()

        def sink(name):
            slot = self._slot.get(name)
            if slot is None:
                return None
            off, shape = slot
            return self.gflat[off: off + shape.numel()].view(shape)
        return sink

    def step(self, grads):
        """grads: {name: fp32 tensor} of THIS rank's micro-batch.  Returns the global gradient norm (before clipping).
        A parameter without an entry (or None) had no gradient in this step — e.g. down_proj / up_proj on an image batch (T == 1
        skips the pooling), mvm_decoder.* when no mask was drawn: like torch.optim.AdamW (`if p.grad is None: continue`) it is
        left untouched (no decay, no moment update, its own step count does not advance) and does not enter the gradient norm."""
        import torch.distributed as dist
        present = []
        for name, off, p in zip(self.names, self.offsets, self.params):
            g = grads.get(name)
            present.append(g is not None)
            if g is None:
                self.gflat[off: off + p.numel()].zero_()
            elif not (g.is_contiguous() and g.numel() == p.numel() and g.data_ptr() == self.gflat.data_ptr() + 4 * off):   # else: produced in place (grad_sink)
                self.gflat[off: off + p.numel()] = g.reshape(-1)
            elif not getattr(self, "_sink_live", False):
                raise RuntimeError(f"AdamW.step: the gradient of {name!r} aliases the optimizer's flat buffer but no producer has filled it since the last step() "
                                   "(a stale grads dict from an earlier train step?): the buffer now holds that step's AVERAGED gradients — re-run "
                                   "loss_and_grads(sink=optimizer.grad_sink()) or pass clones")
        self._sink_live = False
        lo = self.rank * self.shard
        if self.world > 1:
            # presence is a property of the AVERAGED gradient: a parameter that got a gradient on ANY rank (mixed image / video
            # batches, a mask drawn on one rank only) is updated on the rank that owns its shard and its step count advances on
            # every rank alike — torch DDP semantics; a per-rank view would skip updates and let the bias corrections diverge
            pm = torch.tensor([1 if x else 0 for x in present], dtype=torch.int32, device=self.gflat.device)
            dist.all_reduce(pm, op=dist.ReduceOp.MAX, group=self.group)
            present = [bool(x) for x in pm.tolist()]
            gshard = torch.empty(self.shard, device=self.gflat.device, dtype=torch.float32)
            if dist.get_backend(self.group) == "gloo":          # gloo has no reduce_scatter: all-reduce + slice (tests only)
                dist.all_reduce(self.gflat, group=self.group)
                gshard.copy_(self.gflat[lo: lo + self.shard])
            else:
                dist.reduce_scatter_tensor(gshard, self.gflat, group=self.group)
            gshard /= self.world
        else:
            gshard = self.gflat
        sq = hip.sumsq(gshard)
        if self.world > 1:
            dist.all_reduce(sq, group=self.group)
        norm = float(sq.sqrt())
        scale = 1.0
        if self.max_norm is not None and self.max_norm > 0:
            scale = min(1.0, self.max_norm / (norm + 1e-6))     # torch.nn.utils.clip_grad_norm_
        self.step_no += 1
        pshard = self.flat[lo: lo + self.shard]
        kw = dict(lr=self.lr, beta1=self.betas[0], beta2=self.betas[1], eps=self.eps, weight_decay=self.wd, grad_scale=scale)
        if all(present) and len(set(self.steps)) == 1:
            # the usual case: every parameter has a gradient and the same history -> ONE launch over the whole shard
            self.steps = [self.steps[0] + 1] * len(self.steps)
            hip.adamw(pshard, gshard, self.m, self.v, step=self.steps[0], **kw)
        else:
            # per-parameter step counts / skipped parameters: one launch per run of adjacent parameters that share a step count,
            # restricted to this rank's slice of the flat buffer
            runs = []   # [start, end, step]
            for i, (off, p) in enumerate(zip(self.offsets, self.params)):
                if not present[i]:
                    continue
                self.steps[i] += 1
                a, b = off, off + p.numel()
                if runs and runs[-1][1] == a and runs[-1][2] == self.steps[i]:
                    runs[-1][1] = b
                else:
                    runs.append([a, b, self.steps[i]])
            for a, b, st in runs:
                a, b = max(a, lo), min(b, lo + self.shard)
                if a < b:
                    hip.adamw(self.flat[a:b], gshard[a - lo: b - lo], self.m[a - lo: b - lo], self.v[a - lo: b - lo], step=st, **kw)
        if self.world > 1:
            dist.all_gather_into_tensor(self.flat, pshard.clone(), group=self.group)
        return norm

    def state_dict(self):
        """this rank's optimizer state (HF Trainer / DeepSpeed write one optimizer shard per rank too): step count, hyper-parameters,
        the rank's slices of the moments; the masters live in the model's own state dict (trainable_state_dict)."""
        return dict(step=self.step_no, steps=list(self.steps), lr=self.lr, betas=self.betas, eps=self.eps, weight_decay=self.wd, world_size=self.world,
                    rank=self.rank, n=self.n, m=self.m.detach().cpu().clone(), v=self.v.detach().cpu().clone())

    def load_state_dict(self, sd):
        if sd["world_size"] != self.world or sd["rank"] != self.rank or sd["n"] != self.n:
            raise ValueError(f"optimizer shard of rank {sd['rank']}/{sd['world_size']} ({sd['n']} parameters) does not fit rank "
                             f"{self.rank}/{self.world} ({self.n})")
        self.step_no = int(sd["step"])
        self.steps = [int(x) for x in sd["steps"]] if "steps" in sd else [self.step_no] * len(self.params)
        self.m.copy_(sd["m"].to(self.m.device))
        self.v.copy_(sd["v"].to(self.v.device))


def cosine_lr(step, total_steps, base_lr, warmup_ratio=0.03):
    """HF `get_cosine_schedule_with_warmup` as configured by config/*_stllm_qa.yaml (`lr_scheduler_type: cosine`,
    `warmup_ratio: 0.03`; Trainer uses ceil(total * ratio) warm-up steps): the learning rate of optimizer step `step` (0-based)."""
    import math
    warm = math.ceil(total_steps * warmup_ratio)
    if step < warm:
        return base_lr * step / max(1, warm)
    progress = (step - warm) / max(1, total_steps - warm)
    return base_lr * max(0.0, 0.5 * (1.0 + math.cos(math.pi * progress)))


def trainable_state_dict(model):
    """what train_hf.py:188-203 writes at a checkpoint: the parameters that require grad in the reference, by its names"""
    return {n: p.detach().clone() for n, p in trainable_parameters(model)}


def train_step(model, samples, optimizer, freeze_btadapter=False, drop_path=None):
    """One optimisation step (HF Trainer.training_step + optimizer.step for gradient_accumulation_steps = 1).  Returns (loss, loss_mvm, gradient norm) —
    deliberately NOT the gradients: the LLM's weight gradients live in the optimizer's flat buffer (AdamW.grad_sink) and are consumed by the step."""
    loss, loss_mvm, grads = loss_and_grads(model, samples, freeze_btadapter, drop_path, sink=optimizer.grad_sink())
    norm = optimizer.step(grads)
    invalidate_packed(model)
    if loss.is_cuda:   # optimizer.step() read the gradient norm back: the stream is idle, check the split-K exchanges of this step
        hip.gemm_workspace_check(loss.device, wait=True)
    return loss, loss_mvm, norm


def invalidate_packed(model):
    """drop the cached compute-dtype copies of the TRAINABLE parameters (re-packed lazily from the updated masters; the kernels
    update the masters behind torch's version counters).  The frozen ViT / Q-Former copies stay."""
    model._lm_packed = {}
    model.model.repack()
    sm = model.model.stllm_model
    for name in ("llama_proj", "down_proj", "up_proj"):
        if hasattr(sm, name):
            getattr(sm, name)._packed = {}
    if hasattr(sm, "mvm_decoder"):
        sm.mvm_decoder.head._packed = {}
    if hasattr(sm.visual_encoder, "_bt_packed"):
        sm.visual_encoder._bt_packed = {}
