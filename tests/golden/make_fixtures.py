"""Golden-vector generator — runs ONLY in the build container (needs /root/reference).

    python tests/golden/make_fixtures.py [name ...]      # default: all small fixtures
    python tests/golden/make_fixtures.py c1_full         # full-size config-1 summary (~15 min, 40 GB RAM)
    python tests/golden/make_fixtures.py c2_full         # full-size config-2 (= bench.py's workload) summary

Imports the reference's *own* model code (via ref_shim.py), fills its parameters with the
deterministic generator ``stllm_amd.synth`` (so the weights never need to be stored: both the
oracle and the HIP path regenerate them from (name, seed)), runs the reference on seeded inputs
on CPU in fp32 and writes inputs that cannot be regenerated + (sub-sampled) outputs to
``tests/golden/<name>.npz``.  Sub-sampling: full-width outputs are MBs each; we keep strided
slices plus L2 norms / abs-max, which pins every element class (all rows/cols strides hit all
MFMA lanes/regs) at a few tens of KB per fixture.

A fixture is data (inputs + expected outputs); no reference source is stored.
"""
import os
import sys
import time

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import ref_shim  # noqa: E402
from stllm_amd import synth  # noqa: E402

torch.set_grad_enabled(False)
SEED = 0


def T(name, shape, std=1.0, seed=SEED):
    return synth.normal_(torch.empty(shape), name, seed, std)


def sub(x, *strides):
    """strided sub-sample + stats"""
    x = x.detach().float()
    idx = tuple(slice(None, None, s) for s in strides)
    return x[idx].contiguous().numpy()


def stats(x):
    x = x.detach().double()
    return np.array([x.norm().item(), x.abs().max().item(), x.mean().item()])


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"[fixture] {name}: {os.path.getsize(path) / 1024:.1f} KiB  keys={list(arrs)}")


# ------------------------------------------------------------------------------------------
def fx_vit_ops():
    """A1-A5: PatchEmbed, Attention, Block, forward_features(depth 2), ln_vision."""
    ref = ref_shim.load_reference()
    vit = ref_shim.build_ref_vit(depth=2)
    synth.fill_module_(vit, SEED, "visual_encoder.")
    frames = T("input.frames", (2, 3, 224, 224))
    pe = vit.patch_embed(frames)
    h0 = T("input.h0", (2, 257, 1408))
    attn = vit.blocks[0].attn(h0)
    blk = vit.blocks[0](h0)
    mlp = vit.blocks[0].mlp(h0)
    feat = vit(frames)
    ln = ref.blip2.LayerNorm(1408)
    synth.fill_module_(ln, SEED, "ln_vision.")
    lnv = ln(feat)
    save("vit_ops",
         patch_embed=sub(pe, 1, 5, 7), patch_embed_stats=stats(pe),
         attn=sub(attn, 1, 4, 9), attn_stats=stats(attn),
         mlp=sub(mlp, 1, 4, 9), mlp_stats=stats(mlp),
         block=sub(blk, 1, 4, 9), block_stats=stats(blk),
         feat=sub(feat, 1, 4, 9), feat_stats=stats(feat),
         ln_vision=sub(lnv, 1, 4, 9), ln_vision_stats=stats(lnv))


def _qf_inputs():
    enc = T("input.image_embeds", (2, 257, 1408))
    g = torch.Generator().manual_seed(SEED)
    ids = torch.randint(3, 30000, (2, 12), generator=g)
    tmask = torch.ones(2, 12, dtype=torch.long)
    tmask[1, 9:] = 0  # ragged text (padding='longest')
    ids[1, 9:] = 0
    return enc, ids, tmask


def fx_qformer():
    """A6-A9: full 12-layer Q-Former with and without text; stripped == full-without-text."""
    qf, qt = ref_shim.build_ref_qformer(text=True)
    synth.fill_module_(qf, SEED, "Qformer.")
    synth.normal_(qt.data, "query_tokens", SEED, 0.02)
    enc, ids, tmask = _qf_inputs()
    q = qt.expand(2, -1, -1)
    ones = torch.ones(2, 257, dtype=torch.long)
    att = torch.cat([torch.ones(2, 32, dtype=torch.long), tmask], dim=1)
    o_text = qf.bert(ids, attention_mask=att, query_embeds=q, encoder_hidden_states=enc,
                     encoder_attention_mask=ones, return_dict=True).last_hidden_state
    o_plain = qf.bert(query_embeds=q, encoder_hidden_states=enc, encoder_attention_mask=ones,
                      return_dict=True).last_hidden_state
    # stripped (MiniGPT4) variant: same weights, text modules removed
    qf2, _ = ref_shim.build_ref_qformer(text=False)
    synth.fill_module_(qf2, SEED, "Qformer.")
    o_strip = qf2.bert(query_embeds=q, encoder_hidden_states=enc, encoder_attention_mask=ones,
                       return_dict=True).last_hidden_state
    assert torch.equal(o_plain, o_strip)
    # single layers in isolation (even = with cross-attention, odd = without)
    h = T("input.qf_h", (2, 44, 768))
    ext = qf.bert.get_extended_attention_mask(att, h.shape[:-1], h.device, False)
    l0 = qf.bert.encoder.layer[0](h, ext, None, enc, None, None, False, 32)[0]
    l1 = qf.bert.encoder.layer[1](h, ext, None, enc, None, None, False, 32)[0]
    save("qformer", input_ids=ids.numpy(), text_mask=tmask.numpy(),
         out_text=sub(o_text, 1, 1, 3), out_text_stats=stats(o_text),
         out_plain=sub(o_plain, 1, 1, 3), out_plain_stats=stats(o_plain),
         layer0=sub(l0, 1, 1, 3), layer1=sub(l1, 1, 1, 3))


class _Cfg(dict):
    def get(self, k, d=None):
        return dict.get(self, k, d)


def _build_ref_stllm(cfg, vit_depth, qf_layers, llm_layers, bt_depth=3):
    """The reference's own STLLMForCausalLM(+STLLMModel) with network/weight loading bypassed."""
    ref = ref_shim.load_reference()
    st, blip2 = ref.st, ref.blip2
    text = cfg.get("qformer_text_input", False)

    def init_vision_encoder(cls, model_name, img_size, dpr, ckpt, precision):
        if model_name == "eva_clip_g":
            v = ref_shim.build_ref_vit(vit_depth)
        else:
            v = _build_ref_btadapter(vit_depth, bt_depth)
        return v, blip2.LayerNorm(v.num_features)

    def init_qformer(cls, num_query_token, vision_width, cross_attention_freq=2):
        return ref_shim.build_ref_qformer(num_query_token, vision_width, qf_layers, text=True,
                                          vocab=30522, keep_cls=True)

    blip2.Blip2Base.init_tokenizer = classmethod(lambda cls, truncation_side="right": ref_shim.FakeTokenizer())
    blip2.Blip2Base.init_vision_encoder = classmethod(init_vision_encoder)
    blip2.Blip2Base.init_Qformer = classmethod(init_qformer)
    blip2.Blip2Base.load_from_pretrained = lambda self, url_or_filename: None
    st.LlamaTokenizer.from_pretrained = classmethod(
        lambda cls, *a, **k: ref_shim.FakeTokenizer(pad_token_id=0, bos_token_id=1, eos_token="2"))
    lcfg = st.StllmConfig(hidden_size=4096, intermediate_size=11008, num_hidden_layers=llm_layers,
                          num_attention_heads=32, num_key_value_heads=32, vocab_size=32000,
                          rms_norm_eps=1e-6, max_position_embeddings=2048, attn_implementation="eager")
    model = st.STLLMForCausalLM(lcfg)
    orig_resize = st.STLLMLlamaModel.resize_token_embeddings
    st.STLLMLlamaModel.resize_token_embeddings = lambda self, n, *a, **k: None  # FakeTokenizer len == vocab
    model.get_model().initialize_vision_modules(cfg)
    st.STLLMLlamaModel.resize_token_embeddings = orig_resize
    sm = model.model.stllm_model
    # Q-Former embedding table was resized to len(tokenizer)=32000 by the reference; irrelevant to math
    model.eval()
    return model


def _build_ref_btadapter(vit_depth, bt_depth):
    ref = ref_shim.load_reference()
    orig = ref.bt.create_eva_vit_g
    ref.bt.create_eva_vit_g = lambda *a, **k: ref_shim.build_ref_vit(vit_depth)
    try:
        m = ref.bt.EVAVisionTransformer_BTAdapter(depth=bt_depth)
    finally:
        ref.bt.create_eva_vit_g = orig
    return m.eval()


def fill_stllm(model, seed=SEED):
    """Reference-named parameters <- synth.  Non-zero up_proj / temporal_fc on purpose
    (zero-init branches would otherwise be untested)."""
    synth.fill_module_(model, seed, "")
    return model


def _samples(B, Tn, text, seed=SEED):
    g = torch.Generator().manual_seed(seed + 1)
    image = T("input.video", (B, Tn, 3, 224, 224))

    def ids(n):
        return torch.randint(3, 32000, (n,), generator=g).tolist()

    before = [ids(7) for _ in range(B)]  # 7 ids (+BOS => img_start 8, st_llm.py:71)
    after = [ids(5 + i) for i in range(B)]  # ragged
    answer = [ids(6 + 2 * i) for i in range(B)]
    qtext = [ids(9 - 2 * i) for i in range(B)]
    s = lambda r: " ".join(str(x) for x in r)
    if text:
        # qformer text = instruction.split('Human: ')[1].split(' ###')[0]   (st_llm.py:458)
        instr = [f"{s(before[i])}<ImageHere>{s(after[i])} Human: {s(qtext[i])} ###" for i in range(B)]
    else:
        instr = [f"{s(before[i])}<ImageHere>{s(after[i])}" for i in range(B)]
    samples = {"image": image, "instruction_input": instr, "answer": [s(a) for a in answer]}
    meta = dict(before=before, after=after, answer=answer, qtext=qtext)
    return samples, meta


def _ragged(rows):
    L = max(len(r) for r in rows)
    a = np.full((len(rows), L), -1, dtype=np.int64)
    for i, r in enumerate(rows):
        a[i, :len(r)] = r
    return a


def fx_stllm(name, cfg, Tn, vit_depth=2, qf_layers=2, llm_layers=2):
    """A10,A12-A16 end-to-end through the reference's STLLMForCausalLM.forward(samples)."""
    cfg = _Cfg(cfg)
    model = fill_stllm(_build_ref_stllm(cfg, vit_depth, qf_layers, llm_layers))
    text = cfg.get("qformer_text_input", False)
    samples, meta = _samples(2, Tn, text)
    sm = model.model.stllm_model
    if text:
        # the reference appends ' 2' (eos) to answers; for text Q-Former, after-ids get BOS via add_special_tokens
        pass
    # --- inject the mask: capture what the reference draws from numpy's global RNG --------------
    np.random.seed(1234)
    out = model(samples=samples)
    extra = {}
    if cfg.get("use_mask", False):
        extra["mask"] = sm.mask.squeeze(1).numpy()
        extra["img_len"] = np.array([sm.img_len, sm.mask_img_len])
    # separate pieces for narrower checks
    np.random.seed(1234)
    ie, am, ue, ua, tg = sm(samples)
    enc = sm.encode_img(samples["image"],
                        [it.split('Human: ')[1].split(' ###')[0] for it in samples["instruction_input"]] if text else None)[0]
    outs, loss_mvm, labels = model.model(samples) if False else (None, None, None)
    np.random.seed(1234)
    o2, loss_mvm, _ = model.model(samples)
    loss_total = out.loss.item()
    # effective id streams seen by the reference's tokenizer calls (st_llm.py:387-390, 498-508):
    # text mode: p_after is tokenised with add_special_tokens=True (BOS first) and contains the
    # Q-Former text ids; answers get eos/end_sym (id 2) appended.
    B = len(meta["after"])
    after_eff = [([1] if text else []) + meta["after"][i] + (meta["qtext"][i] if text else []) for i in range(B)]
    answer_eff = [a + [2] for a in meta["answer"]]
    save(name,
         before=_ragged(meta["before"]), after=_ragged(after_eff), answer=_ragged(answer_eff),
         qtext=_ragged(meta["qtext"]),
         inputs_llama=sub(enc, 1, 1, 4, 16), inputs_llama_stats=stats(enc),
         inputs_embeds=sub(ie, 1, 1, 16), inputs_embeds_stats=stats(ie),
         attention_mask=am.numpy(), targets=tg.numpy(),
         hidden=sub(o2[0], 1, 1, 16), hidden_stats=stats(o2[0]),
         logits=sub(out.logits, 1, 1, 61), logits_stats=stats(out.logits),
         loss=np.array([loss_total, -1.0 if loss_mvm is None else loss_mvm.item()]),
         **extra)


def fx_stllm_minigpt4():
    fx_stllm("stllm_minigpt4", dict(vit_model="eva_clip_g", image_size=224, num_query_token=32,
                                    llama_model="", video_input="all", use_mask=True, mvm_decode=True,
                                    qformer_text_input=False, max_txt_len=32, end_sym=" 2",
                                    vit_precision="fp32"), Tn=4)


def fx_stllm_no_qformer():
    """st_llm.py:299-301, 369-373: has_qformer=False — 4 concatenated patch tokens per LLM token through llama_proj(5632 -> 4096), 64 tokens per
    frame; 'mean' pooling over T = 2 frames (no shipped yaml sets it: the branch is pinned here so that the product's implementation has a reference)."""
    fx_stllm("stllm_no_qformer", dict(vit_model="eva_clip_g", image_size=224, num_query_token=32, llama_model="", video_input="mean",
                                      use_mask=False, mvm_decode=False, qformer_text_input=False, has_qformer=False,
                                      max_txt_len=32, end_sym=" 2", vit_precision="fp32"), Tn=2)


def fx_stllm_pre_encoding():
    """st_llm.py:452-455: pre_encoding=True — samples["image"] holds pre-extracted Q-Former features [B, T, 32, 768]; forward() applies llama_proj only
    (the vision tower and the Q-Former are built but never run), 'all' pooling over T = 3 frames.  No shipped yaml sets it: pinned here."""
    cfg = _Cfg(dict(vit_model="eva_clip_g", image_size=224, num_query_token=32, llama_model="", video_input="all", use_mask=False, mvm_decode=False,
                    qformer_text_input=False, pre_encoding=True, max_txt_len=32, end_sym=" 2", vit_precision="fp32"))
    model = fill_stllm(_build_ref_stllm(cfg, 1, 1, 2))
    samples, meta = _samples(2, 3, False)
    samples["image"] = T("input.features", (2, 3, 32, 768), 0.5)
    sm = model.model.stllm_model
    assert sm.pre_encoding
    out = model(samples=samples)
    ie, am, ue, ua, tg = sm(samples)
    o2, _, _ = model.model(samples)
    save("stllm_pre_encoding",
         before=_ragged(meta["before"]), after=_ragged(meta["after"]), answer=_ragged([a + [2] for a in meta["answer"]]), qtext=_ragged(meta["qtext"]),
         inputs_embeds=sub(ie, 1, 1, 16), inputs_embeds_stats=stats(ie), attention_mask=am.numpy(), targets=tg.numpy(),
         hidden=sub(o2[0], 1, 1, 16), hidden_stats=stats(o2[0]),
         logits=sub(out.logits, 1, 1, 61), logits_stats=stats(out.logits), loss=np.array([out.loss.item(), -1.0]))


def fx_stllm_instructblip():
    fx_stllm("stllm_instructblip", dict(vit_model="eva_clip_g", image_size=224, num_query_token=32,
                                        llama_model="", video_input="residual", residual_size=4,
                                        use_mask=False, mvm_decode=False, qformer_text_input=True,
                                        max_txt_len=32, end_sym=" 2", vit_precision="fp32"), Tn=8)


def fx_pooling():
    """A12/A13/A17: pooling variants incl. the inference twin (Chat.upload_video tensor math)."""
    ref = ref_shim.load_reference()
    emb = T("input.inputs_llama", (2, 8, 32, 4096), 0.5)
    down, up = nn.Linear(4096, 1024), nn.Linear(1024, 4096)
    synth.fill_named_(list(down.named_parameters()), SEED, "down_proj.")
    synth.fill_named_(list(up.named_parameters()), SEED, "up_proj.")
    R = 4
    seg = 8.0 / R
    idx = torch.from_numpy(np.array([int((seg / 2) + np.round(seg * i)) for i in range(R)]))
    g = emb.mean(dim=1, keepdim=True).expand((-1, R, -1, -1))
    res = (emb[:, idx] + up(torch.relu(down(g)))).view(2, 1, -1, 4096)
    idx_tab = {f"idx_{r}_{t}": np.array([int((float(t) / r / 2) + np.round(float(t) / r * i)) for i in range(r)])
               for (r, t) in [(4, 8), (16, 64), (4, 16), (16, 16), (3, 10), (4, 6)]}
    np.random.seed(7)
    mask = ref.utils.RandomMaskingGenerator(256, 0.37, 2, "cpu")
    allv = emb.view(2, 1, -1, 4096)
    kept = allv[~mask.unsqueeze(1)].reshape(2, 1, -1, 4096)
    save("pooling", residual=sub(res, 1, 1, 1, 32), residual_stats=stats(res),
         mean=sub(emb.mean(dim=1, keepdim=True), 1, 1, 1, 32),
         mask=mask.numpy(), kept=sub(kept, 1, 1, 1, 64), **idx_tab)


def fx_llama():
    """A15/A16: HF LlamaModel (2 layers, full width) prefill with right-padding + lm_head."""
    ref = ref_shim.load_reference()
    st = ref.st
    lcfg = st.StllmConfig(hidden_size=4096, intermediate_size=11008, num_hidden_layers=2,
                          num_attention_heads=32, num_key_value_heads=32, vocab_size=32000,
                          rms_norm_eps=1e-6, max_position_embeddings=2048, attn_implementation="eager")
    model = st.STLLMForCausalLM(lcfg).eval()
    synth.fill_module_(model, SEED, "")
    S = 45
    x = T("input.inputs_embeds", (2, S, 4096), 0.05)
    am = torch.ones(2, S, dtype=torch.long)
    am[1, 37:] = 0
    out = model(samples=None, inputs_embeds=x, attention_mask=am, output_hidden_states=True, use_cache=False)
    h1 = out.hidden_states[1]
    hl = out.hidden_states[-1]
    save("llama", attention_mask=am.numpy(),
         layer0=sub(h1, 1, 1, 16), layer0_stats=stats(h1),
         hidden=sub(hl, 1, 1, 16), hidden_stats=stats(hl),
         logits=sub(out.logits, 1, 1, 61), logits_stats=stats(out.logits))


def fx_btadapter():
    """A11: BT-Adapter on a 5-block ViT with non-zero temporal_fc; 5-D and 4-D inputs."""
    m = _build_ref_btadapter(5, 3)
    synth.fill_module_(m, SEED, "visual_encoder.")
    x5 = T("input.video", (2, 4, 3, 224, 224))
    branches = []
    orig = m.forward_branch

    def spy(x, branch_x, num_layer, mask=None):
        r = orig(x, branch_x, num_layer, mask)
        branches.append(r)
        return r
    m.forward_branch = spy
    o5 = m(x5)
    b5 = list(branches)
    branches.clear()
    o4 = m(x5[0])
    save("btadapter", out5=sub(o5, 1, 4, 9), out5_stats=stats(o5), out4=sub(o4, 1, 4, 9), out4_stats=stats(o4),
         **{f"branch{j}": sub(b, 1, 16, 9) for j, b in enumerate(b5)})


def fx_chat():
    """A17: Chat.upload_video tensor math (4-D frames -> video_emb) + get_context_emb_sim concat +
    prefill logits through generate()'s first forward (inputs_embeds path)."""
    cfg = _Cfg(dict(vit_model="eva_clip_g", image_size=224, num_query_token=32, llama_model="",
                    video_input="residual", residual_size=2, use_mask=False, mvm_decode=False,
                    qformer_text_input=True, max_txt_len=32, end_sym=" 2", vit_precision="fp32"))
    model = fill_stllm(_build_ref_stllm(cfg, 2, 2, 2))
    sm = model.model.stllm_model
    frames = T("input.frames4", (4, 3, 224, 224))
    g = torch.Generator().manual_seed(5)
    qtext = torch.randint(3, 30000, (6,), generator=g).tolist()
    question = torch.randint(3, 32000, (11,), generator=g).tolist()
    ref = ref_shim.load_reference()
    video_emb, _, _ = sm.encode_img(frames, " ".join(map(str, qtext)))
    # conversation.py:285-293 ('residual' branch), restated with the reference's modules
    Tn, R = video_emb.size(0), sm.residual_size
    seg = float(Tn) / R
    ridx = torch.from_numpy(np.array([int((seg / 2) + np.round(seg * i)) for i in range(R)]))
    glob = video_emb.mean(dim=0, keepdim=True).expand((R, -1, -1))
    glob = sm.up_proj(sm.non_linear_func(sm.down_proj(glob)))
    vemb = (video_emb[ridx] + glob).view(1, -1, video_emb.size(-1))
    qids = torch.tensor([[1] + question])  # add_special_tokens => BOS first (conversation.py:327-328)
    mixed = torch.cat((vemb, sm.embed_tokens(qids)), dim=1)
    out = model(samples=None, inputs_embeds=mixed, use_cache=False)
    save("chat", qtext=np.array(qtext), question=np.array(question),
         video_emb=sub(vemb, 1, 1, 16), video_emb_stats=stats(vemb),
         logits=sub(out.logits, 1, 1, 61), logits_stats=stats(out.logits),
         last_logits=out.logits[0, -1].numpy()[::7])


GEN_CASES = [(4.0, 3), (4.0, 4), (8.0, 4), (8.0, 5)]   # (lm_head scale, prompt seed): beam search != greedy in two of them
GEN_MODES = [dict(num_beams=1), dict(num_beams=5), dict(num_beams=3, repetition_penalty=1.3, length_penalty=2.0)]


def fx_generate():
    """§8f rank 1: `llama_model.generate(inputs_embeds=...)` exactly as Chat.answer calls it (conversation.py:231-243; demo.py runs
    num_beams=5, do_sample=False) on the reference's own STLLMForCausalLM (2-layer Llama, synthetic weights, lm_head scaled so that
    the next-token distribution has structure).  Stores only the generated ids; prompts regenerate from (name, seed)."""
    cfg = _Cfg(dict(vit_model="eva_clip_g", image_size=224, num_query_token=32, llama_model="", video_input="mean", use_mask=False,
                    mvm_decode=False, qformer_text_input=False, max_txt_len=32, end_sym=" 2", vit_precision="fp32"))
    model = fill_stllm(_build_ref_stllm(cfg, 1, 2, 2))
    w0 = model.lm_head.weight.detach().clone()
    out = {}
    for scale, seed in GEN_CASES:
        with torch.no_grad():
            model.lm_head.weight.copy_(w0 * scale)
        emb = T(f"gen.emb{seed}", (1, 9, 4096), 0.05)
        for mi, kw in enumerate(GEN_MODES):
            k = dict(dict(max_new_tokens=6, do_sample=False, min_length=1, top_p=0.9, repetition_penalty=1.0, length_penalty=1,
                          temperature=1.0), **kw)
            ids = model.generate(inputs_embeds=emb, **k)[0]
            out[f"s{scale:g}_p{seed}_m{mi}"] = ids.numpy().astype(np.int64)
            print(scale, seed, kw, ids.tolist())
    save("generate", **out)


def fx_backward():
    """gradients of the reference's training loss (st_llm.py:125-138, CE + loss_mvm) w.r.t. every parameter
    the reference leaves trainable with freeze_vit / freeze_qformer = True and freeze_LLM = False (config/*_stllm_qa.yaml):
    llama_proj, down/up_proj, mvm_decoder, and the whole LLM (embed_tokens, layers, norm, lm_head).  loss.backward() through the
    reference's own STLLMForCausalLM.forward(samples) on CPU fp32; per parameter we keep (L2 norm, abs-max, mean) and a strided
    slice.  The injected mask is the one the reference drew (numpy seed 1234)."""
    cases = {
        "mvm": (dict(vit_model="eva_clip_g", image_size=224, num_query_token=32, llama_model="", video_input="all",
                     use_mask=True, mvm_decode=True, qformer_text_input=False, max_txt_len=32, end_sym=" 2",
                     vit_precision="fp32"), 4),
        "residual": (dict(vit_model="eva_clip_g", image_size=224, num_query_token=32, llama_model="",
                          video_input="residual", residual_size=4, use_mask=False, mvm_decode=False,
                          qformer_text_input=True, max_txt_len=32, end_sym=" 2", vit_precision="fp32"), 8),
        # the backbone of 4 of the 5 shipped training configs: visual_encoder.BTAdapter* stay trainable (st_llm.py:257-261)
        # and with the model block of config/instructblipbase_stllm_qa.yaml: video_input all, use_mask, mvm_decode, qformer_text_input
        "btadapter": (dict(vit_model="eva_btadapter_g", image_size=224, num_query_token=32, llama_model="", video_input="all",
                           use_mask=True, mvm_decode=True, qformer_text_input=True, max_txt_len=32, end_sym=" 2",
                           vit_precision="fp32"), 4),
    }
    depths = {"btadapter": (4, 2, 1)}                      # (ViT blocks, Q-Former layers, Llama layers); default (1, 2, 2)
    only = os.environ.get("STLLM_FX_BACKWARD_ONLY")        # regenerate one case, keep the others
    arrs = {}
    if only:
        old = np.load(os.path.join(HERE, "backward.npz"))
        arrs = {k: old[k] for k in old.files if not k.startswith(only + ".")}
        cases = {only: cases[only]}
    for tag, (cfg, Tn) in cases.items():
        cfg = _Cfg(cfg)
        model = fill_stllm(_build_ref_stllm(cfg, *depths.get(tag, (1, 2, 2))))
        samples, meta = _samples(2, Tn, cfg.get("qformer_text_input", False))
        np.random.seed(1234)
        with torch.enable_grad():
            out = model(samples=samples)
            out.loss.backward()
        sm = model.model.stllm_model
        names = []
        for n, prm in model.named_parameters():
            if prm.grad is None:
                continue
            names.append(n)
            g = prm.grad
            arrs[f"{tag}.stats.{n}"] = stats(g)
            arrs[f"{tag}.slice.{n}"] = sub(g, 97, 101) if g.dim() == 2 else sub(g, 29)
        frozen = [n for n, prm in model.named_parameters() if prm.grad is None]
        assert all(n.startswith(("model.stllm_model.visual_encoder", "model.stllm_model.ln_vision", "model.stllm_model.Qformer",
                                 "model.stllm_model.query_tokens")) and "BTAdapter" not in n for n in frozen), frozen[:5]
        arrs[f"{tag}.names"] = np.array(names)
        arrs[f"{tag}.loss"] = np.array([out.loss.item()])
        text = cfg.get("qformer_text_input", False)       # effective id streams, as in fx_stllm
        arrs[f"{tag}.before"] = _ragged(meta["before"])
        arrs[f"{tag}.after"] = _ragged([([1] if text else []) + meta["after"][i] + (meta["qtext"][i] if text else []) for i in range(2)])
        arrs[f"{tag}.answer"] = _ragged([a + [2] for a in meta["answer"]])
        arrs[f"{tag}.qtext"] = _ragged(meta["qtext"])
        if cfg.get("use_mask", False):
            arrs[f"{tag}.mask"] = sm.mask.squeeze(1).numpy()
        print(tag, "loss", out.loss.item(), "trainable tensors", len(names), "frozen", len(frozen))
    save("backward", **arrs)


def fx_c1_full():
    """Config 1, FULL SIZE: B1 T4, EVA-CLIP-g 39 blocks + 12-layer Q-Former + Vicuna-7B 32 layers,
    InstructBLIP-style (residual R=4 of T=4 == all frames, text Q-Former).  Stores a logits summary."""
    t0 = time.time()
    cfg = _Cfg(dict(vit_model="eva_clip_g", image_size=224, num_query_token=32, llama_model="",
                    video_input="all", use_mask=False, mvm_decode=False,
                    qformer_text_input=False, max_txt_len=32, end_sym=" 2", vit_precision="fp32"))
    model = _build_ref_stllm(cfg, 39, 12, 32)
    print("built", time.time() - t0)
    fill_stllm(model)
    print("filled", time.time() - t0)
    samples, meta = _samples(1, 4, False)
    t1 = time.time()
    out = model(samples=samples)
    print("forward", time.time() - t1)
    lg = out.logits[0]
    top = lg.topk(5, dim=-1)
    save("c1_full", before=_ragged(meta["before"]), after=_ragged(meta["after"]), answer=_ragged(meta["answer"]),
         logits_slice=lg[::3, ::499].numpy(), logits_stats=stats(lg),
         top_ids=top.indices.numpy(), top_vals=top.values.numpy(),
         row_norms=lg.norm(dim=-1).numpy(), loss=np.array([out.loss.item()]),
         ref_forward_seconds=np.array([time.time() - t1]))


def fx_pos_embed():
    """the reference's interpolate_pos_embed (eva_vit.py:373-394) — a checkpoint whose position table was
    trained at another resolution (here 26 x 26 = 364 px and 8 x 8 = 112 px) resampled to the model's 16 x 16 grid."""
    ref = ref_shim.load_reference()
    D = 24
    import types
    model = types.SimpleNamespace(patch_embed=types.SimpleNamespace(num_patches=256), pos_embed=torch.zeros(1, 257, D))
    arrs = {}
    for tag, n in (("up", 8), ("down", 26), ("same", 16)):
        ck = {"pos_embed": T(f"pos_embed.{tag}", (1, 1 + n * n, D))}
        arrs[f"{tag}.in"] = ck["pos_embed"].numpy().copy()
        ref.eva.interpolate_pos_embed(model, ck)
        arrs[f"{tag}.out"] = ck["pos_embed"].numpy()
    save("pos_embed", **arrs)


def fx_c2_full():
    """Config 2 = the BENCHMARKED size: B1 T16, EVA-CLIP-g 39 blocks + 12-layer Q-Former + Vicuna-7B 32 layers, 'all' pooling,
    S = 576 — exactly the samples bench.py times (bench.make_samples), through the reference's own forward on CPU (fp32).
    Stores the same logits summary as c1_full."""
    import bench   # repo root is on sys.path (see the top of this file)
    t0 = time.time()
    cfg = _Cfg(dict(vit_model="eva_clip_g", image_size=224, num_query_token=32, llama_model="",
                    video_input="all", use_mask=False, mvm_decode=False,
                    qformer_text_input=False, max_txt_len=32, end_sym=" 2", vit_precision="fp32"))
    model = _build_ref_stllm(cfg, 39, 12, 32)
    print("built", time.time() - t0, flush=True)
    fill_stllm(model)
    print("filled", time.time() - t0, flush=True)
    samples = bench.make_samples(1, 16, "cpu")
    t1 = time.time()
    out = model(samples=samples)
    print("forward", time.time() - t1, flush=True)
    lg = out.logits[0]
    top = lg.topk(5, dim=-1)
    save("c2_full", logits_slice=lg[::3, ::499].numpy(), logits_stats=stats(lg), seq_len=np.array([lg.shape[0]]),
         top_ids=top.indices.numpy(), top_vals=top.values.numpy(),
         row_norms=lg.norm(dim=-1).numpy(), loss=np.array([out.loss.item()]),
         ref_forward_seconds=np.array([time.time() - t1]))



def fx_stllm_flagship():
    """The reference's flagship SHIPPED combination (config/instructblipbase_stllm_conversation.yaml:11,14-17): text-conditioned Q-Former +
    global-local 'residual' pooling + dynamic masking over the pooled R*32 block + the MVM branch, whose slices start at img_start = 0
    (st_llm.py:71: BOS and the 'before' ids are inside the slice — a quirk of the reference that a drop-in must reproduce)."""
    fx_stllm("stllm_flagship", dict(vit_model="eva_clip_g", image_size=224, num_query_token=32, llama_model="",
                                    video_input="residual", residual_size=4, use_mask=True, mvm_decode=True,
                                    qformer_text_input=True, max_txt_len=32, end_sym=" 2", vit_precision="fp32"), Tn=8)


FULL_CFGS = {
    # BASELINE configs[2] (config/instructblipbase_stllm_conversation.yaml:11,14-15,21 without the mask: bench.py's c3 leg): B=4 x T=64, text Q-Former, residual R=16
    "c3": (dict(vit_model="eva_clip_g", video_input="residual", residual_size=16, use_mask=False, mvm_decode=False, qformer_text_input=True,
                max_txt_len=64), 4, 64, True),
    # BASELINE configs[3]: dynamic masking + MVM forward at T=32 ('all' pooling: L=1024 visual tokens, two prefills S ~ 560 + 1100; st_llm.py:56-92, 480-493)
    "c4": (dict(vit_model="eva_clip_g", video_input="all", use_mask=True, mvm_decode=True, qformer_text_input=False, max_txt_len=32), 1, 32, False),
    # BASELINE configs[4] (config/minigpt4base_stllm_qa.yaml:3,7,11,13-14): BT-Adapter backbone, 'all', mask + MVM, T=16
    "c5": (dict(vit_model="eva_btadapter_g", video_input="all", use_mask=True, mvm_decode=True, qformer_text_input=False, max_txt_len=32), 1, 16, False),
}
MASK_SEED = 1234   # numpy's global RNG before the reference's forward: st_llm.py:482-484 draws the rate and the shuffles from it


def fx_full(tag):
    """BASELINE configs[2..4] at FULL SIZE (39 + 12 + 32 layers) through the reference's own forward on CPU (fp32) on exactly the samples
    bench.py --config <tag> times (bench.make_samples + bench.CONFIGS).  Same logits summary as c2_full, per clip; the mask the reference
    drew (numpy global RNG seeded with MASK_SEED) is stored so that the device path can be given the same one."""
    import bench
    t0 = time.time()
    mcfg, B, Tn, text = FULL_CFGS[tag]
    cfg = _Cfg(dict(dict(image_size=224, num_query_token=32, llama_model="", end_sym=" 2", vit_precision="fp32"), **mcfg))
    model = _build_ref_stllm(cfg, 39, 12, 32)
    print("built", time.time() - t0, flush=True)
    fill_stllm(model)
    print("filled", time.time() - t0, flush=True)
    samples = bench.make_samples(B, Tn, "cpu", text=text)
    sm = model.model.stllm_model
    np.random.seed(MASK_SEED)
    t1 = time.time()
    out = model(samples=samples)
    dt = time.time() - t1
    print("forward", dt, flush=True)
    extra = {}
    if cfg.get("use_mask", False):
        extra["mask"] = sm.mask.squeeze(1).numpy()
        extra["img_len"] = np.array([sm.img_len, sm.mask_img_len])
        np.random.seed(MASK_SEED)
        _, loss_mvm, _ = model.model(samples)          # the MVM term on its own (second full forward)
        extra["loss_mvm"] = np.array([loss_mvm.item()])
    lg = out.logits
    top = lg.topk(5, dim=-1)
    save(f"{tag}_full", logits_slice=lg[:, ::3, ::499].numpy(), logits_stats=stats(lg), seq_len=np.array([lg.shape[1]]),
         top_ids=top.indices.numpy(), top_vals=top.values.numpy(), row_norms=lg.norm(dim=-1).numpy(),
         loss=np.array([out.loss.item()]), ref_forward_seconds=np.array([dt]), **extra)


ALL = dict(vit_ops=fx_vit_ops, qformer=fx_qformer, pooling=fx_pooling, llama=fx_llama,
           stllm_minigpt4=fx_stllm_minigpt4, stllm_instructblip=fx_stllm_instructblip, stllm_no_qformer=fx_stllm_no_qformer, stllm_pre_encoding=fx_stllm_pre_encoding,
           stllm_flagship=fx_stllm_flagship, btadapter=fx_btadapter, chat=fx_chat, generate=fx_generate, backward=fx_backward, pos_embed=fx_pos_embed)
SLOW = dict(c1_full=fx_c1_full, c2_full=fx_c2_full, c3_full=lambda: fx_full("c3"), c4_full=lambda: fx_full("c4"), c5_full=lambda: fx_full("c5"))

if __name__ == "__main__":
    names = sys.argv[1:] or list(ALL)
    for n in names:
        t0 = time.time()
        {**ALL, **SLOW}[n]()
        print(f"  ({n}: {time.time() - t0:.1f}s)")
