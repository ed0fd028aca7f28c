"""CPU (-m "not gpu"): host-side logic of the product path — no kernel is launched.

  * the C-ABI library loads and exports every symbol include/stllm_hip.h declares;
  * synthetic-weight generator: deterministic, dtype/device independent bits, sane statistics;
  * weight packers are pure permutations (RoPE head permutation, gate/up interleave, patch padding);
  * parameter names/shapes of the product modules == the reference's (checkpoint drop-in, Appendix D);
  * token-block assembly index tables (the product's one-gather form of prompt_wrap / concat_emb_input_output /
    BOS / targets) reproduce the oracle's (== reference's) tensors when the gather is emulated with torch indexing;
  * residual index / masking generator semantics; tokenizer stand-in; the HIP path refuses CPU tensors loudly.
"""
import os
import re

import numpy as np
import pytest
import torch

import shapes
import stllm_oracle as O
from _util import T, golden, sd_from, unragged

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_exports_match_header():
    from stllm_amd import hip
    L = hip.lib()
    header = open(os.path.join(ROOT, "include", "stllm_hip.h")).read()
    declared = set(re.findall(r"^(?:int|int64_t|const char\*)\s+(stllm_[a-z0-9_]+)\s*\(", header, re.M))
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(L, name), f"libstllm_hip.so does not export {name}"
    assert set(hip.EXPORTS) == declared, (set(hip.EXPORTS) ^ declared)
    assert L.stllm_abi_version() == 7   # round 6: stllm_gemm_args.w_frag, stllm_llama_layer_weights.wqkv_frag / wgu_frag


def test_phased_gemm_schedule_invariants():
    """Host logic of the phased GEMM (st-llm_amd/csrc/gemm_p8.inc, make_plan): every tile is covered exactly once, the K
    slices of a remainder tile fit one XCD's 32 workgroups, and the hot-path shapes get the schedules DESIGN.md describes."""
    from stllm_amd import hip
    shapes_ = [(4112, 4224, 1408), (4112, 1408, 1408), (4112, 6144, 1408), (4112, 1408, 6144), (576, 12288, 4096),
               (576, 4096, 4096), (576, 22016, 4096), (576, 4096, 11008), (576, 32000, 4096), (4096, 4096, 4096),
               (1, 128, 64), (300, 384, 192), (100000, 256, 64), (257, 128 * 77, 64 * 9)]
    for rows in (192, 256):
        for M, N, K in shapes_:
            for heavy in (0, 1, 2):
                q, r, s, cap, est = hip.gemm_plan(M, N, K, heavy, rows)
                tiles = -(-M // rows) * -(-N // 256)
                assert q * 256 + r == tiles and 0 <= r < 256, (M, N, K, rows, q, r)
                assert 1 <= s <= 32 and cap == 32 // s and est > 0
                if r:
                    assert 8 * cap >= r, "every remainder tile needs its own group of s workgroups on one XCD"
                    assert s <= K // 64, "no empty K slice"
                else:
                    assert s == 1
    # the Llama prefill shapes of config 2 (M = 576 = 3 x 192: no padding with the 192-row tile)
    assert hip.gemm_plan(576, 4096, 11008, 1, 192)[:4] == (0, 48, 5, 6)       # down_proj: 48 tiles x 5 K-slices = 240 workgroups
    assert hip.gemm_plan(576, 22016, 4096, 0, 192)[:3] == (1, 2, 16)          # gate/up: one full round + 2 tiles split 16 ways (cost model: 0.3 us per peer)
    assert hip.gemm_plan(576, 12288, 4096, 0, 192)[:3] == (0, 144, 1)         # qkv: 144 tiles admit no XCD-local split
    assert hip.gemm_plan(4096, 4096, 4096, 0, 256)[:3] == (1, 0, 1)           # 256 tiles: pure data-parallel
    with pytest.raises(RuntimeError):
        hip.gemm_plan(576, 4096, 100, 0, 192)                                  # K must be a multiple of 64


def test_w4_gemm_schedule_invariants():
    """Host logic of the one-wave-per-SIMD GEMM (st-llm_amd/csrc/gemm_w4.inc, w4_make_plan): same invariants as the phased kernel."""
    from stllm_amd import hip
    shapes_ = [(4112, 4224, 1408), (4112, 6144, 1408), (4112, 1408, 6144), (576, 12288, 4096), (576, 22016, 4096), (576, 4096, 11008),
               (4096, 4096, 4096), (1, 128, 64), (300, 384, 192), (100000, 256, 64), (3072, 8192, 1408)]
    for shape, rows, cols in ((34, 192, 256), (44, 256, 256), (32, 192, 128), (42, 256, 128), (43, 256, 192), (33, 192, 192), (24, 128, 256)):
        for M, N, K in shapes_:
            for heavy in (0, 1, 2):
                q, r, s, cap, est = hip.gemm_w4_plan(M, N, K, heavy, shape)
                tiles = -(-M // rows) * -(-N // cols)
                assert q * 256 + r == tiles and 0 <= r < 256, (M, N, K, rows, q, r)
                assert 1 <= s <= 32 and cap == 32 // s and est > 0
                if r:
                    assert 8 * cap >= r and s <= K // 64
                else:
                    assert s == 1
    assert hip.gemm_w4_plan(3072, 8192, 1408, 0, 34)[:3] == (2, 0, 1)          # 512 tiles of 192 x 256: two whole rounds
    assert hip.gemm_w4_plan(4096, 4096, 4096, 0, 44)[:3] == (1, 0, 1)
    assert hip.gemm_w4_plan(4112, 1408, 6144, 1, 32)[:3] == (0, 242, 1)        # ViT fc2 / proj: 242 tiles of 192 x 128 = one partial round, no K split
    # heavy bit 3 = the epilogue allows thin tail rows: a last tile row of <= 32 rows is computed outside the tile grid
    assert hip.gemm_w4_plan(4112, 6144, 1408, 2 | 8, 42)[:3] == (3, 0, 1)     # ViT fc1: 16 x 48 tiles of 256 x 128 = three whole rounds (+ 16 thin rows)
    assert hip.gemm_w4_plan(4112, 6144, 1408, 2, 42)[:2] == (3, 48)           # without: a 17th tile row
    assert hip.gemm_w4_plan(4112 + 17, 6144, 1408, 2 | 8, 42)[:2] == (3, 48)  # 33 tail rows: too many for the thin path
    assert hip.gemm_w4_plan(200, 128, 64, 8, 32)[:2] == (0, 1)                # 8 tail rows past one 192-row tile
    assert hip.gemm_w4_plan(20, 128, 64, 8, 32)[:2] == (0, 1)                 # fewer rows than a tile: an ordinary ragged tile
    # round 3: 192-column tiles (16-bit STORE epilogues only)
    assert hip.gemm_w4_plan(4112, 6144, 1408, 2 | 8, 43)[:3] == (2, 0, 1)     # ViT fc1: 16 x 32 tiles of 256 x 192 = TWO whole rounds (+ 16 thin rows)
    assert hip.gemm_w4_plan(4112, 4224, 1408, 8, 33)[:3] == (1, 228, 1)       # ViT qkv: 22 x 22 tiles of 192 x 192 = 1.9 rounds, no K split
    assert hip.gemm_w4_plan(576, 12288, 4096, 0, 24)[:3] == (0, 240, 1)        # round 4, Llama prefill qkv: 5 x 48 tiles of 128 x 256 = ONE partial round, nothing exchanged
    with pytest.raises(RuntimeError):
        hip.gemm_w4_plan(576, 4096, 4096, 0, 35)


def test_hip_path_has_no_cpu_fallback():
    from stllm_amd import hip
    a = torch.zeros(8, 64, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="no CPU fallback|must be a CUDA"):
        hip.gemm(a, torch.zeros(128, 64, dtype=torch.bfloat16), dtype="bf16")
    with pytest.raises(RuntimeError):
        hip.layernorm(torch.zeros(4, 64), torch.ones(64), torch.zeros(64), 1e-5, dtype="bf16")


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "st-llm_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py") and f != "smoke.py":  # smoke.py is __graft_entry__.smoke()'s checker leg
                src = open(os.path.join(dp, f)).read()
                assert "stllm_oracle" not in src and "import oracle" not in src, f"{f} touches the oracle"


def test_synth_deterministic_and_normalish():
    from stllm_amd import synth
    a = synth.normal_(torch.empty(257, 1408), "x.weight", 0, 0.02)
    b = synth.normal_(torch.empty(257 * 1408), "x.weight", 0, 0.02).view(257, 1408)
    assert torch.equal(a, b)
    assert not torch.equal(a, synth.normal_(torch.empty(257, 1408), "y.weight", 0, 0.02))
    assert not torch.equal(a, synth.normal_(torch.empty(257, 1408), "x.weight", 1, 0.02))
    assert abs(a.std().item() - 0.02) < 2e-4 and abs(a.mean().item()) < 2e-4
    h = synth.normal_(torch.empty(1000, dtype=torch.bfloat16), "x.weight", 0, 0.02)
    assert torch.equal(h, a.reshape(-1)[:1000].to(torch.bfloat16))
    m = torch.nn.LayerNorm(16)
    synth.fill_module_(m, 0, "norm1.")
    assert abs(m.weight.mean().item() - 1.0) < 0.2


def test_packers_are_permutations():
    from stllm_amd import pack
    H, D, K = 4, 128, 64
    wq, wk, wv = T("wq", (H * D, K)), T("wk", (H * D, K)), T("wv", (H * D, K))
    qkv = pack.llama_qkv(wq, wk, wv, torch.float32, n_heads=H)
    perm = pack.rope_head_perm(H)
    assert sorted(perm.tolist()) == list(range(H * D))
    assert torch.equal(qkv[:H * D], wq[perm]) and torch.equal(qkv[H * D:2 * H * D], wk[perm]) and torch.equal(qkv[2 * H * D:], wv)
    # partner columns (i, i+64) of a head sit 32 apart inside one 64-column group
    p1 = pack.rope_head_perm(1).tolist()
    for g in range(2):
        for j in range(32):
            assert p1[g * 64 + 32 + j] == p1[g * 64 + j] + 64
    wg, wu = T("wg", (256, K)), T("wu", (256, K))
    gu = pack.llama_gate_up(wg, wu, torch.float32)
    for g in range(256 // 32):
        assert torch.equal(gu[g * 64:g * 64 + 32], wg[g * 32:(g + 1) * 32])
        assert torch.equal(gu[g * 64 + 32:(g + 1) * 64], wu[g * 32:(g + 1) * 32])
    pw = pack.patch_weight(T("pw", (1408, 3, 14, 14)), "bf16")
    assert pw.shape == (1408, 640) and pw[:, 588:].abs().max() == 0
    assert pack.patch_weight(T("pw", (1408, 3, 14, 14)), "fp32").shape == (1408, 608)
    c, s = pack.rope_tables(9)
    oc, os_ = O.rope_tables(9, 128)
    assert torch.equal(c, oc[:, :64]) and torch.equal(s, os_[:, :64])
    assert pack.pad_rows(torch.zeros(32001, 8)).shape[0] == 32128
    # round 6: the fragment-major copy of the W-direct GEMM (stllm_gemm_args.w_frag): lane l of fragment (nb, ks) = w[32 nb + (l & 31), 16 ks + 8 (l >> 5) : + 8]
    w = torch.arange(64 * 48, dtype=torch.float32).view(64, 48).to(torch.bfloat16)
    f = pack.frag32(w).view(2, 3, 64, 8)
    for nb, ks, l in [(0, 0, 0), (0, 0, 31), (0, 0, 32), (1, 2, 63), (1, 1, 37), (0, 2, 5)]:
        assert torch.equal(f[nb, ks, l], w[32 * nb + (l & 31), 16 * ks + 8 * (l >> 5): 16 * ks + 8 * (l >> 5) + 8])
    assert sorted(f.reshape(-1).float().tolist()) == sorted(w.reshape(-1).float().tolist())     # a permutation
    assert pack.frag32_or_none(w) is None        # host tensors / shapes outside the kernel's domain: no copy (the other kernels run)


@pytest.mark.parametrize("text,video_input,mvm,vit", [(False, "all", True, "eva_clip_g"), (True, "residual", False, "eva_clip_g"),
                                                      (False, "all", True, "eva_btadapter_g")])
def test_param_names_match_reference(text, video_input, mvm, vit):
    from stllm_amd.models import st_llm
    from stllm_amd.models.blip2 import Blip2Base
    old = (Blip2Base.vit_depth, Blip2Base.qformer_layers)
    Blip2Base.vit_depth, Blip2Base.qformer_layers = 4, 2
    try:
        m = st_llm.STLLMForCausalLM.from_config(dict(vit_model=vit, image_size=224, num_query_token=32, video_input=video_input,
                                                     use_mask=mvm, mvm_decode=mvm, qformer_text_input=text,
                                                     llama_model=dict(num_hidden_layers=2)), device="cpu")
    finally:
        Blip2Base.vit_depth, Blip2Base.qformer_layers = old
    got = {k: tuple(v.shape) for k, v in m.named_parameters()}
    want = {**shapes.stllm_model_shapes(4, 2, text, video_input, mvm, vit_model=vit), **shapes.llama_shapes(2)}
    assert got == {k: tuple(v) for k, v in want.items()}
    # attribute paths used by the reference's callers (conversation.py:185-190, 281-293)
    sm = m.model.stllm_model
    assert sm.embed_tokens is m.model.embed_tokens and hasattr(sm, "llama_tokenizer") and sm.video_input == video_input
    from stllm_amd.common.registry import registry
    assert registry.get_model_class("st_llm_hf") is st_llm.STLLMForCausalLM


def _emulate_gather(vis_flat, table, rows):
    idx = torch.tensor(rows)
    out = torch.empty(idx.shape + (vis_flat.shape[-1],))
    pos = idx >= 0
    out[pos] = vis_flat[idx[pos]]
    out[~pos] = table[-idx[~pos] - 1]
    return out


@pytest.mark.parametrize("name,text,use_mask", [("stllm_minigpt4", False, True), ("stllm_instructblip", True, False), ("stllm_flagship", True, True)])
def test_assembly_index_tables_vs_oracle(name, text, use_mask):
    """STLLMModel._assemble (host) + an emulated gather == oracle.assemble == the reference's attention_mask/targets."""
    from stllm_amd.models import st_llm
    from stllm_amd.models.blip2 import Blip2Base
    g = golden(name)
    old = (Blip2Base.vit_depth, Blip2Base.qformer_layers)
    Blip2Base.vit_depth, Blip2Base.qformer_layers = 1, 1
    try:
        m = st_llm.STLLMForCausalLM.from_config(dict(vit_model="eva_clip_g", video_input="all", use_mask=use_mask,
                                                     qformer_text_input=text, llama_model=dict(num_hidden_layers=1, vocab_size=32000)),
                                                device="cpu")
    finally:
        Blip2Base.vit_depth, Blip2Base.qformer_layers = old
    sm = m.model.stllm_model
    before, after, answer, qtext = [unragged(g[k]) for k in ("before", "after", "answer", "qtext")]
    B = len(before)
    s = lambda r: " ".join(map(str, r))
    if text:
        instr = [f"{s(before[i])}<ImageHere>{s(after[i][1:len(after[i]) - len(qtext[i])])} Human: {s(qtext[i])} ###" for i in range(B)]
    else:
        instr = [f"{s(before[i])}<ImageHere>{s(after[i])}" for i in range(B)]
    L, D = 24, 8
    table = T("emb", (32000, D))
    vis = T("vis", (B, 1, L, D))
    kept = [list(range(L))] * B
    mask = None
    if use_mask:
        np.random.seed(5)
        mask = torch.from_numpy(O.random_masking_generator(L, 0.4, B))
        kept = [torch.nonzero(~mask[b]).flatten().tolist() for b in range(B)]
    rows, att, tg = sm._assemble(L, kept, instr, answer, B)
    ie = _emulate_gather(vis.reshape(B * L, D), table, rows)
    vis_in = O.apply_mask(vis, mask) if use_mask else vis
    r_ie, r_att, r_ue, r_ua, r_tg = O.assemble(vis_in, before, after, answer, table, 0, 1, not text, vis if use_mask else None)
    assert torch.equal(ie, r_ie) and torch.equal(att, r_att.long()) and torch.equal(tg, r_tg)
    # and the mask/targets layout equals the reference's own (fixture) when the visual length matches
    if not use_mask:
        rows2, att2, tg2 = sm._assemble(g["attention_mask"].shape[1] - (att.shape[1] - L), [list(range(g["attention_mask"].shape[1] - (att.shape[1] - L)))] * B, instr, answer, B)
        assert np.array_equal(att2.numpy(), g["attention_mask"]) and np.array_equal(tg2.numpy(), g["targets"])
    if use_mask:
        urows, uatt, _ = sm._assemble(L, [list(range(L))] * B, instr, answer, B)
        assert torch.equal(_emulate_gather(vis.reshape(B * L, D), table, urows), r_ue) and torch.equal(uatt, r_ua.long())


def test_residual_index_and_masking_semantics():
    from stllm_amd.models.st_llm import get_residual_index
    from stllm_amd.models.utils import RandomMaskingGenerator
    g = golden("pooling")
    for k in g.files:
        if k.startswith("idx_"):
            r, t = map(int, k.split("_")[1:])
            assert np.array_equal(get_residual_index(r, t), g[k])
    np.random.seed(7)
    m = RandomMaskingGenerator(256, 0.37, 2)
    assert np.array_equal(m.numpy(), g["mask"]) and m.dtype == torch.bool and int(m[0].sum()) == int(0.37 * 256)


def test_id_tokenizer_contract():
    from stllm_amd.tokenizer import IdTokenizer
    tk = IdTokenizer()
    o = tk(["5 6 7", "8"], padding="longest", add_special_tokens=False)
    assert o.input_ids.tolist() == [[5, 6, 7], [8, 0, 0]] and o.attention_mask.tolist() == [[1, 1, 1], [1, 0, 0]]
    assert tk("###Human: 9 10 ###", add_special_tokens=True).input_ids.tolist() == [[1, 9, 10]]
    assert tk(["1 2 3 4"], truncation=True, max_length=2, add_special_tokens=False).input_ids.tolist() == [[1, 2]]
    assert tk("", add_special_tokens=False).input_ids.shape == (1, 0)


def test_synth_host_generator_is_bit_identical_to_the_torch_recipe():
    """stllm_synth_normal_f32 (error.cpp) is what fills CPU tensors since round 2; the torch recipe (the one that fills device
    tensors) must give the same bits: (name, seed) keys, chunk boundaries of the threaded loop, non-zero mean, tiny tensors."""
    from stllm_amd import synth
    saved = synth._CACHE
    synth._CACHE = None
    try:
        for n, std, mean in [(1, 0.02, 0.0), (4097, 1.0, 0.5), ((1 << 22) + 13, 0.02, 0.0), (3 * 1408 * 64, 1e-3, -2.0)]:
            a, b = torch.empty(n), torch.empty(n)
            synth.normal_(a, "model.layers.0.mlp.down_proj.weight", 7, std, mean)
            host = synth._host_fill
            synth._host_fill = lambda *k: False
            try:
                synth.normal_(b, "model.layers.0.mlp.down_proj.weight", 7, std, mean)
            finally:
                synth._host_fill = host
            assert torch.equal(a, b), (n, std, mean)
    finally:
        synth._CACHE = saved


def test_mixed_numerics_mode_scopes_the_vit_only():
    """runtime "mixed" (round 5): the split verify mode with the ViT blocks in fp16 — vit_scope() switches the mode for the visual encoder's
    forward and restores it, every other mode leaves vit_scope() a no-op"""
    import torch
    from stllm_amd import runtime
    with runtime.use_dtype("mixed"):
        assert runtime.mode_name() == "mixed" and runtime.compute_dtype() == torch.float32 and runtime.gemm_split()
        with runtime.vit_scope():
            assert runtime.mode_name() == "fp16" and runtime.compute_dtype() == torch.float16 and not runtime.gemm_split()
        assert runtime.mode_name() == "mixed" and runtime.gemm_split()
    for m in ("bf16", "fp16", "fp32", "bf16x3"):
        with runtime.use_dtype(m):
            with runtime.vit_scope():
                assert runtime.mode_name() == m
    assert runtime.mode_name() == "bf16"
