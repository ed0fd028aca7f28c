"""CPU: the oracle (oracle/stllm_oracle.py) against the golden vectors captured from the
reference's own model code (tests/golden/make_fixtures.py).  fp32 both sides; tolerance 2e-5
relative to the tensor's abs-max (summation order differs between the reference's torch graph and
the restatement only in a few places, e.g. the explicit im2col GEMM)."""
import numpy as np
import pytest
import torch

import shapes
import stllm_oracle as O
from _util import T, golden, sd_from, stats, sub, unragged, assert_close

torch.set_grad_enabled(False)
RTOL = 2e-5


def close(got, want, what):
    want = np.asarray(want)
    assert_close(got, want, atol=RTOL * max(1.0, float(np.abs(want).max())), what=what)


def test_vit_ops():
    g = golden("vit_ops")
    sd = sd_from({**shapes.vit_shapes(2), "ln_vision.weight": (1408,), "ln_vision.bias": (1408,)})
    frames = T("input.frames", (2, 3, 224, 224))
    h0 = T("input.h0", (2, 257, 1408))
    p = "visual_encoder."
    pe = O.vit_patch_embed(frames, sd, p)
    close(sub(pe, 1, 5, 7), g["patch_embed"], "patch_embed")
    close(stats(pe), g["patch_embed_stats"], "patch_embed stats")
    close(sub(O.vit_attention(h0, sd, p + "blocks.0.attn."), 1, 4, 9), g["attn"], "attention")
    close(sub(O.vit_mlp(h0, sd, p + "blocks.0.mlp."), 1, 4, 9), g["mlp"], "mlp")
    close(sub(O.vit_block(h0, sd, p + "blocks.0."), 1, 4, 9), g["block"], "block")
    feat = O.vit_forward(frames, sd, p)
    close(sub(feat, 1, 4, 9), g["feat"], "forward_features")
    close(stats(feat), g["feat_stats"], "forward_features stats")
    close(sub(O.ln_vision(feat, sd, "ln_vision"), 1, 4, 9), g["ln_vision"], "ln_vision")


def test_qformer():
    g = golden("qformer")
    sd = sd_from({**shapes.qformer_shapes(12, True, 30523), "query_tokens": (1, 32, 768)})
    enc = T("input.image_embeds", (2, 257, 1408))
    q = sd["query_tokens"].expand(2, -1, -1)
    ids = torch.from_numpy(g["input_ids"])
    tmask = torch.from_numpy(g["text_mask"])
    att = torch.cat([torch.ones(2, 32, dtype=torch.long), tmask], dim=1)
    o_text = O.qformer_forward(q, enc, sd, "Qformer.bert.", ids, att)
    close(sub(o_text, 1, 1, 3), g["out_text"], "qformer with text")
    close(stats(o_text), g["out_text_stats"], "qformer with text stats")
    o_plain = O.qformer_forward(q, enc, sd, "Qformer.bert.")
    close(sub(o_plain, 1, 1, 3), g["out_plain"], "qformer without text")
    # stripped (MiniGPT4) == full without text: same oracle call on a state dict without text params
    sd2 = {k: v for k, v in sd.items() if k in shapes.qformer_shapes(12, False) or k == "query_tokens"}
    assert torch.equal(O.qformer_forward(q, enc, sd2, "Qformer.bert."), o_plain)
    h = T("input.qf_h", (2, 44, 768))
    add = (1.0 - att[:, None, None, :].float()) * -10000.0
    close(sub(O.bert_layer(h, add, enc, 0, sd, "Qformer.bert.", 32), 1, 1, 3), g["layer0"], "even BertLayer")
    close(sub(O.bert_layer(h, add, enc, 1, sd, "Qformer.bert.", 32), 1, 1, 3), g["layer1"], "odd BertLayer")


def test_pooling_and_masking():
    g = golden("pooling")
    sd = sd_from({"down_proj.weight": (1024, 4096), "down_proj.bias": (1024,),
                  "up_proj.weight": (4096, 1024), "up_proj.bias": (4096,)})
    emb = T("input.inputs_llama", (2, 8, 32, 4096), 0.5)
    res = O.video_pool(emb, "residual", sd, "", 4)
    close(sub(res, 1, 1, 1, 32), g["residual"], "residual pooling")
    close(stats(res), g["residual_stats"], "residual pooling stats")
    close(sub(O.video_pool(emb, "mean", sd), 1, 1, 1, 32), g["mean"], "mean pooling")
    assert O.video_pool(emb, "all", sd).shape == (2, 1, 256, 4096)
    for k in g.files:
        if k.startswith("idx_"):
            r, t = map(int, k.split("_")[1:])
            assert np.array_equal(O.get_residual_index(r, t), g[k]), k
    # masking: same numpy RNG stream as the reference => identical mask; then the gather
    np.random.seed(7)
    mask = O.random_masking_generator(256, 0.37, 2)
    assert np.array_equal(mask, g["mask"])
    kept = O.apply_mask(emb.reshape(2, 1, -1, 4096), torch.from_numpy(mask))
    assert kept.shape == (2, 1, 256 - int(0.37 * 256), 4096)
    close(sub(kept, 1, 1, 1, 64), g["kept"], "masked gather")
    # inference twin (Chat.upload_video): [T,32,D] -> [1,L,D]
    close(sub(O.video_pool_infer(emb[0], "residual", sd, "", 4), 1, 1, 32), g["residual"][0], "infer residual")


def test_llama_prefill():
    g = golden("llama")
    sd = sd_from(shapes.llama_shapes(2))
    x = T("input.inputs_embeds", (2, 45, 4096), 0.05)
    am = torch.from_numpy(g["attention_mask"])
    h1 = O.llama_forward(x, am, {k: v for k, v in sd.items() if "layers.1." not in k}, final_norm=False)
    close(sub(h1, 1, 1, 16), g["layer0"], "llama layer 0")
    hid = O.llama_forward(x, am, sd)
    close(sub(hid, 1, 1, 16), g["hidden"], "llama hidden")
    close(stats(hid), g["hidden_stats"], "llama hidden stats")
    lg = O.lm_logits(hid, sd)
    close(sub(lg, 1, 1, 61), g["logits"], "logits")
    close(stats(lg), g["logits_stats"], "logits stats")


def test_btadapter():
    g = golden("btadapter")
    sd = sd_from(shapes.btadapter_shapes(5, 3))
    x5 = T("input.video", (2, 4, 3, 224, 224))
    o5, br = O.btadapter_forward(x5, sd, "visual_encoder.", 3, return_branches=True)
    close(sub(o5, 1, 4, 9), g["out5"], "btadapter 5-D")
    close(stats(o5), g["out5_stats"], "btadapter 5-D stats")
    for j, b in enumerate(br):
        close(sub(b, 1, 16, 9), g[f"branch{j}"], f"branch {j}")
    o4 = O.btadapter_forward(x5[0], sd, "visual_encoder.", 3)
    close(sub(o4, 1, 4, 9), g["out4"], "btadapter 4-D")


def _e2e(name, cfg, Tn, text):
    g = golden(name)
    sdshape = {**shapes.stllm_model_shapes(2, 2, text, cfg["video_input"], cfg.get("mvm_decode", False), qf_vocab=32000,
                                           has_qformer=cfg.get("has_qformer", True)), **shapes.llama_shapes(2)}
    sd = sd_from(sdshape)
    image = T("input.features", (2, Tn, 32, 768), 0.5) if cfg.get("pre_encoding") else T("input.video", (2, Tn, 3, 224, 224))
    samples = {"image": image, "before_ids": unragged(g["before"]),
               "after_ids": unragged(g["after"]), "answer_ids": unragged(g["answer"])}
    if text:
        # the reference's BERT tokenizer call uses add_special_tokens=True (st_llm.py:344): the fixture's
        # FakeTokenizer prepends its BOS id (1)
        qt = [[1] + r for r in unragged(g["qtext"])]
        L = max(len(r) for r in qt)
        ids = torch.zeros(2, L, dtype=torch.long)
        m = torch.zeros(2, L, dtype=torch.long)
        for i, r in enumerate(qt):
            ids[i, :len(r)] = torch.tensor(r)
            m[i, :len(r)] = 1
        samples["qformer_ids"], samples["qformer_mask"] = ids, m
    if cfg.get("use_mask"):
        samples["mask"] = torch.from_numpy(g["mask"])
    out = O.stllm_forward(samples, sd, dict(cfg, pad_id=0, bos_id=1))
    assert np.array_equal(out["attention_mask"].numpy(), g["attention_mask"])
    assert np.array_equal(out["targets"].numpy(), g["targets"])
    close(sub(out["inputs_embeds"], 1, 1, 16), g["inputs_embeds"], "inputs_embeds")
    close(stats(out["inputs_embeds"]), g["inputs_embeds_stats"], "inputs_embeds stats")
    close(sub(out["hidden"], 1, 1, 16), g["hidden"], "hidden")
    close(sub(out["logits"], 1, 1, 61), g["logits"], "logits")
    close(stats(out["logits"]), g["logits_stats"], "logits stats")
    assert abs(out["loss"].item() - g["loss"][0]) < 2e-4 * max(1.0, abs(g["loss"][0]))
    if g["loss"][1] >= 0:
        assert abs(out["loss_mvm"].item() - g["loss"][1]) < 2e-5


CFG_MINIGPT4 = dict(vit_model="eva_clip_g", video_input="all", use_mask=True, mvm_decode=True, qformer_text_input=False)
CFG_INSTRUCTBLIP = dict(vit_model="eva_clip_g", video_input="residual", residual_size=4, use_mask=False, mvm_decode=False,
                        qformer_text_input=True)


def test_stllm_minigpt4_style():
    _e2e("stllm_minigpt4", CFG_MINIGPT4, 4, False)


def test_stllm_instructblip_style():
    _e2e("stllm_instructblip", CFG_INSTRUCTBLIP, 8, True)


def test_stllm_without_qformer():
    """st_llm.py:299-301, 369-373 (has_qformer=False): CLS dropped, 4 patch tokens concatenated per LLM token, llama_proj(5632 -> 4096) — the
    oracle's branch against the reference's own forward."""
    _e2e("stllm_no_qformer", dict(vit_model="eva_clip_g", video_input="mean", use_mask=False, mvm_decode=False, qformer_text_input=False,
                                  has_qformer=False), 2, False)


def test_stllm_pre_encoding():
    """st_llm.py:452-455 (pre_encoding=True): samples["image"] = pre-extracted features [B, T, 32, 768], llama_proj only — the oracle's branch against
    the reference's own forward."""
    _e2e("stllm_pre_encoding", dict(vit_model="eva_clip_g", video_input="all", use_mask=False, mvm_decode=False, qformer_text_input=False,
                                    pre_encoding=True), 3, False)


CFG_FLAGSHIP = dict(vit_model="eva_clip_g", video_input="residual", residual_size=4, use_mask=True, mvm_decode=True, qformer_text_input=True)


def test_stllm_flagship_yaml_combination():
    """config/instructblipbase_stllm_conversation.yaml:11,14-17: text Q-Former + residual pooling + mask over the pooled block + MVM with
    img_start = 0 (st_llm.py:71) — the oracle against the reference's own forward (loss AND loss_mvm)."""
    _e2e("stllm_flagship", CFG_FLAGSHIP, 8, True)


def test_chat_upload_video_path():
    g = golden("chat")
    sd = sd_from({**shapes.stllm_model_shapes(2, 2, True, "residual", False, qf_vocab=32000), **shapes.llama_shapes(2)})
    p = "model.stllm_model."
    frames = T("input.frames4", (4, 3, 224, 224))
    qt = torch.tensor([1] + g["qtext"].tolist()).view(1, -1).repeat(4, 1)
    emb = O.encode_img(frames, sd, p, "eva_clip_g", qt, torch.ones_like(qt))
    vemb = O.video_pool_infer(emb, "residual", sd, p, 2)
    close(sub(vemb, 1, 1, 16), g["video_emb"], "video_emb")
    qids = torch.tensor([[1] + g["question"].tolist()])
    mixed = torch.cat((vemb, sd["model.embed_tokens.weight"][qids]), dim=1)
    lg = O.lm_logits(O.llama_forward(mixed, None, sd), sd)
    close(sub(lg, 1, 1, 61), g["logits"], "prefill logits")
    close(lg[0, -1].numpy()[::7], g["last_logits"], "first-step logits")
