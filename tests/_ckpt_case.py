"""§8(f4) checkpoint I/O end to end (shared by the CPU contract-backend test and the -m gpu test): write the files the reference's loaders
read (st_llm.py:149-158 sharded `pytorch_model-XXXXX-of-YYYYY.bin`; :189-201 / :595-603 `ckpt` with the BLIP-2 projector name `llm_proj`;
:52-53, 180-181 the 32001-row tables of the '[PAD]' tokenizer; eva_vit.py:373-394 a position table of another resolution), load them
through STLLMForCausalLM.from_config on `device`, run a forward and compare with the oracle fed the same state dict (position table
resampled on the oracle side by torch's own bicubic interpolation).  Test infrastructure."""
import json

import torch
import torch.nn.functional as F

import shapes
import stllm_oracle as O
from _util import T, sd_from


def run(tmp_path, device, dtype_ctx):
    from stllm_amd.models import st_llm
    from stllm_amd.models.blip2 import Blip2Base
    from stllm_amd.tokenizer import IdTokenizer
    cfg = dict(vit_model="eva_clip_g", image_size=224, num_query_token=32, video_input="residual", residual_size=2, use_mask=False,
               mvm_decode=False, qformer_text_input=True, max_txt_len=32, end_sym=" 2")
    V = 32001
    shp = {**shapes.stllm_model_shapes(1, 2, True, "residual", False, qf_vocab=32000), **shapes.llama_shapes(1, vocab=V)}
    sd = sd_from(shp)
    pfx = "model.stllm_model."
    big = T("ckpt.pos_embed_24", (1, 1 + 24 * 24, 1408), 0.02)
    # ---- the files ------------------------------------------------------------------------------------------------------------
    llm_dir = tmp_path / "vicuna"
    llm_dir.mkdir()
    llm_keys = sorted(k for k in sd if not k.startswith(pfx))
    with open(llm_dir / "config.json", "w") as f:
        json.dump(dict(hidden_size=4096, intermediate_size=11008, num_hidden_layers=1, num_attention_heads=32, vocab_size=32000,
                       rms_norm_eps=1e-6, max_position_embeddings=2048), f)
    for i in range(3):    # the base LLM: 32000 words, three shards
        part = {k: (sd[k][:32000] if k in ("model.embed_tokens.weight", "lm_head.weight") else sd[k]).clone() for k in llm_keys[i::3]}
        torch.save(part, llm_dir / f"pytorch_model-{i + 1:05d}-of-00003.bin")
    ck = {k: v.clone() for k, v in sd.items() if k.startswith(pfx) or k in ("model.embed_tokens.weight", "lm_head.weight")}
    ck["llm_proj.weight"] = ck.pop(pfx + "llama_proj.weight")       # un-prefixed BLIP-2 name: consumed by STLLMModel.from_config (st_llm.py:595-603)
    ck["llm_proj.bias"] = ck.pop(pfx + "llama_proj.bias")
    ck[pfx + "visual_encoder.pos_embed"] = big
    ckpt_file = tmp_path / "stllm_ckpt.pth"
    torch.save({"model": ck}, ckpt_file)
    # ---- the product's loader ---------------------------------------------------------------------------------------------------
    old = (Blip2Base.vit_depth, Blip2Base.qformer_layers, Blip2Base.init_tokenizer, IdTokenizer.hf_special_tokens)
    Blip2Base.vit_depth, Blip2Base.qformer_layers = 1, 2
    Blip2Base.init_tokenizer = classmethod(lambda cls, truncation_side="right": IdTokenizer(0, 1, 2, 32000))
    IdTokenizer.hf_special_tokens = True      # the product default: '[PAD]' becomes id 32000 (tests/conftest.py switches it off for fixture replays)
    try:
        model = st_llm.STLLMForCausalLM.from_config(dict(cfg, llama_model=str(llm_dir), ckpt=str(ckpt_file)), device=device).eval()
        assert model.config.vocab_size == V and model.lm_head.weight.shape[0] == V
        got_sd = model.state_dict()
        missing = sorted(set(sd) - set(got_sd))
        # (position_ids is an arange buffer built by the constructor; stllm_model.embed_tokens aliases model.embed_tokens, st_llm.py:54)
        extra = sorted(k for k in set(got_sd) - set(sd) if not k.endswith("embeddings.position_ids") and k != pfx + "embed_tokens.weight")
        assert not missing and not extra, (missing, extra)
        osd = dict(sd)
        grid = big[:, 1:].reshape(1, 24, 24, 1408).permute(0, 3, 1, 2)
        small = F.interpolate(grid, size=(16, 16), mode="bicubic", align_corners=False).permute(0, 2, 3, 1).reshape(1, 256, 1408)
        osd[pfx + "visual_encoder.pos_embed"] = torch.cat((big[:, :1], small), dim=1)
        for k in sd:      # every tensor arrived from a file (nothing was filled by hand): exact, except the resampled position table
            if k == pfx + "visual_encoder.pos_embed":
                assert float((got_sd[k].cpu() - osd[k]).abs().max()) <= 1e-5, k
            else:
                assert torch.equal(got_sd[k].cpu(), sd[k]), k
        B, Tn = 2, 4
        g = torch.Generator().manual_seed(8)
        ids = lambda n: torch.randint(3, 30000, (n,), generator=g).tolist()
        before, after, answer, qtext = [ids(3)] * B, [ids(4), ids(6)], [ids(5), ids(3)], [ids(6), ids(4)]
        s = lambda r: " ".join(map(str, r))
        image = T("input.video", (B, Tn, 3, 224, 224))
        samples = {"image": image.to(device), "answer": [s(a) for a in answer],
                   "instruction_input": [f"{s(before[i])}<ImageHere>{s(after[i])} Human: {s(qtext[i])} ###" for i in range(B)]}
        L = max(len(q) + 1 for q in qtext)
        qi, qm = torch.zeros(B, L, dtype=torch.long), torch.zeros(B, L, dtype=torch.long)
        for i, q in enumerate(qtext):
            qi[i, :len(q) + 1] = torch.tensor([1] + q)
            qm[i, :len(q) + 1] = 1
        pad_id = model.model.stllm_model.llama_tokenizer.pad_token_id
        assert pad_id == 32000
        ref = O.stllm_forward({"image": image, "before_ids": before, "after_ids": [[1] + after[i] + qtext[i] for i in range(B)],
                               "answer_ids": [a + [2] for a in answer], "qformer_ids": qi, "qformer_mask": qm}, osd,
                              dict(cfg, pad_id=pad_id, bos_id=1))
        with torch.no_grad(), dtype_ctx():
            out = model(samples=samples)
    finally:
        Blip2Base.vit_depth, Blip2Base.qformer_layers, Blip2Base.init_tokenizer, IdTokenizer.hf_special_tokens = old
    assert out.logits.shape == ref["logits"].shape and out.logits.shape[-1] == V
    valid = ref["attention_mask"].bool()
    err = float((out.logits.cpu() - ref["logits"])[valid].abs().max())
    return err, abs(out.loss.item() - ref["loss"].item())
