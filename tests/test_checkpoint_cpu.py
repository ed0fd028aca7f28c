"""CPU: checkpoint I/O compatibility (st_llm.py:149-158, 160-203): a HF-style sharded directory
(config.json + pytorch_model-XXXXX-of-YYYYY.bin, or safetensors shards) and a BLIP-2 style `ckpt` file with the
`llm_proj -> llama_proj` rename load into the product modules by the reference's parameter names."""
import json
import os

import pytest
import torch

from test_host_orchestration_cpu import CFGS


def _build(cfg, llama_model):
    from stllm_amd.models import st_llm
    from stllm_amd.models.blip2 import Blip2Base
    from stllm_amd.tokenizer import IdTokenizer
    old = (Blip2Base.vit_depth, Blip2Base.qformer_layers, Blip2Base.init_tokenizer)
    Blip2Base.vit_depth, Blip2Base.qformer_layers = 1, 2
    Blip2Base.init_tokenizer = classmethod(lambda cls, truncation_side="right": IdTokenizer(0, 1, 2, 32000))
    try:
        return st_llm.STLLMForCausalLM.from_config(dict(cfg, llama_model=llama_model), device="cpu")
    finally:
        Blip2Base.vit_depth, Blip2Base.qformer_layers, Blip2Base.init_tokenizer = old


@pytest.mark.parametrize("fmt", ["bin", "safetensors"])
def test_sharded_directory_round_trip(tmp_path, fmt):
    from stllm_amd import synth
    cfg = CFGS["mean_pooling"]
    lcfg = dict(num_hidden_layers=1, hidden_size=512, intermediate_size=1024, num_attention_heads=4, vocab_size=2048)   # small LLM: this test is about names and files
    a = _build(cfg, lcfg)
    synth.fill_module_(a, 0, "")
    sd = {k: v.detach().clone() for k, v in a.state_dict().items()}
    d = tmp_path / "stllm_ckpt"
    d.mkdir()
    with open(d / "config.json", "w") as f:
        json.dump(dict(lcfg, rms_norm_eps=1e-6, max_position_embeddings=2048, architectures=["LlamaForCausalLM"], torch_dtype="float16"), f)
    keys = sorted(sd)
    shards = [keys[0::3], keys[1::3], keys[2::3]]
    for i, ks in enumerate(shards):
        part = {k: sd[k].contiguous() for k in ks}
        if fmt == "bin":
            torch.save(part, d / f"pytorch_model-{i + 1:05d}-of-{len(shards):05d}.bin")
        else:
            from safetensors.torch import save_file
            save_file(part, str(d / f"model-{i + 1:05d}-of-{len(shards):05d}.safetensors"))
    # as in the released demo configs: `llama_model` and `ckpt` both point at the ST-LLM directory — the first pass fills the
    # LLM (the vision modules do not exist yet, st_llm.py:171-187), the second everything under model.stllm_model.*
    b = _build(dict(cfg, ckpt=str(d)), str(d))
    sb = b.state_dict()
    assert set(sb) == set(sd)
    for k in keys:
        assert torch.equal(sb[k], sd[k]), k
    assert b.config.num_hidden_layers == 1 and b.config.vocab_size == 2048 and b.config.hidden_size == 512


def test_ckpt_file_with_llm_proj_rename(tmp_path):
    """st_llm.py:189-201: `ckpt` may be a single file holding {'model': state_dict} whose projector is still called llm_proj."""
    from stllm_amd import synth
    cfg = CFGS["mean_pooling"]
    small = dict(num_hidden_layers=1, hidden_size=512, intermediate_size=1024, num_attention_heads=4, vocab_size=2048)
    a = _build(cfg, small)
    synth.fill_module_(a, 3, "")
    sd = a.state_dict()
    pw = "model.stllm_model.llama_proj.weight"
    assert pw in sd
    # exactly what the reference's loader consumes: top-level keys, projector under its BLIP-2 name
    ck = {"model": {"llm_proj.weight": sd[pw].clone() * 2, "llm_proj.bias": sd[pw.replace("weight", "bias")].clone() + 1,
                    "model.stllm_model.query_tokens": sd["model.stllm_model.query_tokens"].clone() + 5}}
    path = tmp_path / "pretrained_minigpt4.pth"
    torch.save(ck, path)
    b = _build(dict(cfg, ckpt=str(path)), small)
    sb = b.state_dict()
    # the rename happens before load_state_dict(strict=False); like the reference, un-prefixed keys of the causal-LM wrapper are ignored
    assert torch.equal(sb["model.stllm_model.query_tokens"], sd["model.stllm_model.query_tokens"] + 5)
    assert "llama_proj.weight" not in sb and "llm_proj.weight" not in sb


def test_text_qformer_models_grow_the_vocabulary_like_the_reference(tmp_path):
    """st_llm.py:306-310, :52-53, :180-181: with qformer_text_input the LLaMA tokenizer gains '[PAD]' (id 32000), the input
    table and lm_head grow to 32001 rows BEFORE `ckpt` is loaded, padding / target masking use id 32000 — so a trained
    InstructBLIP-style ST-LLM checkpoint (32001 rows) loads onto a 32000-word Vicuna `llama_model`."""
    from stllm_amd import synth
    from stllm_amd.tokenizer import IdTokenizer
    cfg = dict(CFGS["instructblip_residual_text"])
    small = dict(num_hidden_layers=1, hidden_size=512, intermediate_size=1024, num_attention_heads=4, vocab_size=32000)
    old = IdTokenizer.hf_special_tokens
    IdTokenizer.hf_special_tokens = True        # the product default (conftest switches it off for the fixture replays)
    try:
        a = _build(cfg, small)
        sm = a.model.stllm_model
        assert len(sm.llama_tokenizer) == 32001 and sm.llama_tokenizer.pad_token_id == 32000
        assert a.config.vocab_size == a.vocab_size == 32001
        assert a.model.embed_tokens.weight.shape == (32001, 512) and a.lm_head.weight.shape == (32001, 512)
        assert sm.embed_tokens is a.model.embed_tokens
        synth.fill_module_(a, 5, "")
        sd = {k: v.detach().clone() for k, v in a.state_dict().items()}
        # the trained checkpoint as a single file + the base LLM as a 32000-word HF directory
        base = tmp_path / "vicuna"
        base.mkdir()
        with open(base / "config.json", "w") as f:
            json.dump(dict(small, rms_norm_eps=1e-6, max_position_embeddings=2048), f)
        llm = {k: (v[:32000].clone() if k in ("model.embed_tokens.weight", "lm_head.weight") else v.clone())
               for k, v in sd.items() if "stllm_model" not in k}
        torch.save(llm, base / "pytorch_model-00001-of-00001.bin")
        path = tmp_path / "stllm_text.pth"
        torch.save({"model": sd}, path)
        b = _build(dict(cfg, ckpt=str(path)), str(base))
        sb = b.state_dict()
        assert b.config.vocab_size == 32001 and set(sb) == set(sd)
        for k in sd:
            assert torch.equal(sb[k], sd[k]), k
        # targets: the pad id (32000) is masked, id 0 is an ordinary token now
        rows, att, targets = b.model.stllm_model._assemble(4, [[0, 1, 2, 3]] * 2, ["5 6<ImageHere>7", "5 6<ImageHere>7 8 9"], [[0, 11, 2], [12, 2]], 2)
        assert (targets[0] >= 0).sum() == 3 and (targets[1] >= 0).sum() == 2 and int(targets[0][targets[0] >= 0][0]) == 0
        # a non-text model follows a checkpoint whose tables have another size instead of failing in load_state_dict
        c = _build(dict(CFGS["mean_pooling"], ckpt=str(path)), str(base))
        assert c.config.vocab_size == 32001 and torch.equal(c.state_dict()["lm_head.weight"], sd["lm_head.weight"])
    finally:
        IdTokenizer.hf_special_tokens = old


def test_eva_pos_embed_interpolation_matches_reference():
    """SURVEY §8(f4), eva_vit.py:373-394: position tables of another resolution (8 x 8, 26 x 26) are resampled to the model's
    16 x 16 grid exactly as the reference's interpolate_pos_embed did (fixture from the reference function; torch's own
    F.interpolate as a second opinion), and load_state_dict applies it on the way in."""
    import types
    import numpy as np
    import torch.nn.functional as F
    from _util import golden
    from stllm_amd.models import eva_vit
    g = golden("pos_embed")
    model = types.SimpleNamespace(patch_embed=types.SimpleNamespace(num_patches=256), pos_embed=torch.zeros(1, 257, 24))
    for tag in ("up", "down", "same"):
        ck = {"pos_embed": torch.from_numpy(g[f"{tag}.in"]).clone()}
        eva_vit.interpolate_pos_embed(model, ck)
        assert ck["pos_embed"].shape == (1, 257, 24)
        assert np.abs(ck["pos_embed"].numpy() - g[f"{tag}.out"]).max() <= 2e-6, tag
        n = int((g[f"{tag}.in"].shape[1] - 1) ** 0.5)
        want = F.interpolate(torch.from_numpy(g[f"{tag}.in"])[:, 1:].reshape(1, n, n, 24).permute(0, 3, 1, 2), size=(16, 16), mode="bicubic",
                             align_corners=False).permute(0, 2, 3, 1).reshape(1, 256, 24)
        assert (ck["pos_embed"][:, 1:] - want).abs().max() <= 2e-6
    # through load_state_dict: a 364-px checkpoint (26 x 26 patches) into the 224-px model (full width: the HIP patch-embed is
    # specialised for EVA-CLIP-g; one block keeps it small)
    vit = eva_vit.VisionTransformer(img_size=224, patch_size=14, embed_dim=1408, depth=1, num_heads=16, device="cpu")
    sd = {k: torch.zeros_like(v) for k, v in vit.state_dict().items()}
    big = torch.zeros(1, 1 + 26 * 26, 1408)
    big[..., :24] = torch.from_numpy(g["down.in"])
    sd["pos_embed"] = big
    vit.load_state_dict(sd)
    assert vit.pos_embed.shape == (1, 257, 1408)
    assert np.abs(vit.pos_embed.detach().numpy()[..., :24] - g["down.out"]).max() <= 2e-6 and float(vit.pos_embed[..., 24:].abs().max()) == 0.0


def test_checkpoint_files_to_forward_on_contract_backend(tmp_path):
    """the -m gpu test `test_checkpoint_io_on_device` with the kernels replaced by the contract backend: the same files, the same loader,
    the same oracle comparison (tests/_ckpt_case.py) — keeps the case itself honest without a GPU."""
    import _ckpt_case
    import _cpu_backend
    from stllm_amd import runtime

    import contextlib

    @contextlib.contextmanager
    def ctx():
        with _cpu_backend.installed(), runtime.use_dtype("fp32"):
            yield
    err, loss_err = _ckpt_case.run(tmp_path, "cpu", ctx)
    assert err <= 5e-4 and loss_err <= 1e-4, (err, loss_err)
