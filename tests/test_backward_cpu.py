"""CPU (-m "not gpu"): SURVEY.md §8f rank 3 — gradients of the training loss.

1. The oracle (autograd over oracle/stllm_oracle.py) against the gradients the REFERENCE produced here
   (tests/golden/backward.npz, written by make_fixtures.py fx_backward: loss.backward() through the reference's own
   STLLMForCausalLM.forward(samples), CPU fp32) — pins the gradient oracle.
2. The product's explicit backward graph (stllm_amd/training.py) on the test-only CPU backend against the oracle.
Tolerances: fp32 everywhere, 2e-4 relative to each tensor's abs-max (sums over up to ~350 tokens x 4096 features)."""
import numpy as np
import pytest
import torch

import shapes
import stllm_oracle as O
from _util import T, golden, sd_from, stats, sub, unragged

CASES = {
    "mvm": (dict(vit_model="eva_clip_g", video_input="all", use_mask=True, mvm_decode=True, qformer_text_input=False), 4),
    "residual": (dict(vit_model="eva_clip_g", video_input="residual", residual_size=4, use_mask=False, mvm_decode=False,
                      qformer_text_input=True), 8),
}
BT_CASE = dict(vit_model="eva_btadapter_g", video_input="all", use_mask=True, mvm_decode=True, qformer_text_input=True)   # = instructblipbase_stllm_qa.yaml
FROZEN = ("model.stllm_model.visual_encoder", "model.stllm_model.ln_vision", "model.stllm_model.Qformer",
          "model.stllm_model.query_tokens")


def case_inputs(tag):
    g = golden("backward")
    cfg, Tn = CASES[tag]
    text = cfg["qformer_text_input"]
    sdshape = {**shapes.stllm_model_shapes(1, 2, text, cfg["video_input"], cfg.get("mvm_decode", False), qf_vocab=32000),
               **shapes.llama_shapes(2)}
    sd = sd_from(sdshape)
    samples = {"image": T("input.video", (2, Tn, 3, 224, 224)), "before_ids": unragged(g[f"{tag}.before"]),
               "after_ids": unragged(g[f"{tag}.after"]), "answer_ids": unragged(g[f"{tag}.answer"])}
    if text:
        qt = [[1] + r for r in unragged(g[f"{tag}.qtext"])]
        L = max(len(r) for r in qt)
        ids = torch.zeros(2, L, dtype=torch.long)
        m = torch.zeros(2, L, dtype=torch.long)
        for i, r in enumerate(qt):
            ids[i, :len(r)] = torch.tensor(r)
            m[i, :len(r)] = 1
        samples["qformer_ids"], samples["qformer_mask"] = ids, m
    if cfg.get("use_mask"):
        samples["mask"] = torch.from_numpy(g[f"{tag}.mask"])
    return g, cfg, sd, samples


def oracle_grads(cfg, sd, samples):
    train = [n for n in sd if not n.startswith(FROZEN) or "BTAdapter" in n]      # st_llm.py:257-261: the adapter stays trainable
    for n in train:
        sd[n].requires_grad_(True)
    with torch.enable_grad():
        out = O.stllm_forward(samples, sd, dict(cfg, pad_id=0, bos_id=1))
        out["loss"].backward()
    grads = {n: sd[n].grad for n in train if sd[n].grad is not None}
    for n in train:
        sd[n].requires_grad_(False)
    return out["loss"].item(), grads


def check_against_fixture(g, tag, loss, grads, rtol):
    names = [str(n) for n in g[f"{tag}.names"]]
    assert sorted(names) == sorted(grads), set(names) ^ set(grads)
    assert abs(loss - g[f"{tag}.loss"][0]) < 2e-4 * max(1.0, abs(g[f"{tag}.loss"][0]))
    for n in names:
        gr = grads[n]
        want = g[f"{tag}.slice.{n}"]
        got = sub(gr, 97, 101) if gr.dim() == 2 else sub(gr, 29)
        st = g[f"{tag}.stats.{n}"]
        assert np.abs(got - want).max() <= rtol * st[1], (n, np.abs(got - want).max(), st[1])
        assert abs(stats(gr)[0] - st[0]) <= rtol * st[0], (n, stats(gr)[0], st[0])


@pytest.mark.parametrize("tag", list(CASES))
def test_oracle_gradients_match_reference(tag):
    g, cfg, sd, samples = case_inputs(tag)
    loss, grads = oracle_grads(cfg, sd, samples)
    check_against_fixture(g, tag, loss, grads, 2e-4)


# ---- the product's explicit backward graph on the test-only CPU backend ------------------------------------------------
def product_samples(g, tag, text):
    before = unragged(g[f"{tag}.before"])
    qtext = unragged(g[f"{tag}.qtext"])
    after_eff = unragged(g[f"{tag}.after"])
    after = [a[1: len(a) - len(q)] for a, q in zip(after_eff, qtext)] if text else after_eff
    answer = [a[:-1] for a in unragged(g[f"{tag}.answer"])]
    s = lambda r: " ".join(map(str, r))
    if text:
        instr = [f"{s(before[i])}<ImageHere>{s(after[i])} Human: {s(qtext[i])} ###" for i in range(2)]
    else:
        instr = [f"{s(before[i])}<ImageHere>{s(after[i])}" for i in range(2)]
    return instr, [s(a) for a in answer]


@pytest.mark.parametrize("tag", list(CASES))
def test_product_backward_matches_reference(tag):
    """stllm_amd.training.loss_and_grads (host graph + kernel contracts of include/stllm_hip.h, restated in tests/_cpu_backend.py)
    against the reference's gradients, fp32."""
    import _cpu_backend
    from test_host_orchestration_cpu import build
    from stllm_amd import runtime, training
    g = golden("backward")
    cfg, Tn = CASES[tag]
    text = cfg["qformer_text_input"]
    model = build(dict(cfg, image_size=224, num_query_token=32, max_txt_len=32, end_sym=" 2"), vit_depth=1, qf_layers=2, llm_layers=2)
    instr, answers = product_samples(g, tag, text)
    samples = {"image": T("input.video", (2, Tn, 3, 224, 224)), "instruction_input": instr, "answer": answers}
    if cfg.get("use_mask"):
        samples["mask"] = torch.from_numpy(g[f"{tag}.mask"])
    with _cpu_backend.installed(), runtime.use_dtype("fp32"):
        loss, loss_mvm, grads = training.loss_and_grads(model, samples)
        fwd = model(samples=samples)                       # the inference forward computes the same loss
    assert abs(loss.item() - fwd.loss.item()) <= 2e-5 * max(1.0, abs(fwd.loss.item()))
    check_against_fixture(g, tag, loss.item(), grads, 3e-4)
    names = {n for n, _ in training.trainable_parameters(model)}
    assert names == set(grads), names ^ set(grads)


def test_mean_pooling_backward_and_one_optimizer_step():
    """`video_input: mean` (not in the reference fixture) against autograd over the pinned oracle, then one AdamW step through
    train_step: the masters alias the optimizer's flat buffer, packed copies are rebuilt, and the loss goes down."""
    import _cpu_backend
    from test_host_orchestration_cpu import CFGS, build, make_inputs
    from stllm_amd import runtime, training
    cfg = CFGS["mean_pooling"]
    model = build(cfg, vit_depth=1, qf_layers=2, llm_layers=1)
    samples, osamples = make_inputs(2, 4, False)
    sd = sd_from({**shapes.stllm_model_shapes(1, 2, False, "mean", False, qf_vocab=32000), **shapes.llama_shapes(1)})
    want_loss, want = oracle_grads(cfg, sd, osamples)
    with _cpu_backend.installed(), runtime.use_dtype("fp32"):
        loss, _, grads = training.loss_and_grads(model, samples)
        assert abs(loss.item() - want_loss) <= 1e-4
        assert set(grads) == set(want)
        for n, gr in grads.items():
            scale = want[n].abs().max().item()
            assert (gr - want[n]).abs().max().item() <= 3e-4 * scale, n
        opt = training.AdamW(list(training.trainable_parameters(model)), lr=1e-3, max_grad_norm=1.0)
        before = model.lm_head.weight.detach().clone()
        l0, _, norm = training.train_step(model, samples, opt)
        assert abs(l0.item() - want_loss) <= 1e-4 and norm > 0
        assert not torch.equal(before, model.lm_head.weight)
        l1 = model(samples=samples).loss
    assert l1.item() < l0.item() - 0.05, (l0.item(), l1.item())


def test_gradients_produced_in_the_optimizer_buffer_are_the_same_gradients():
    """train_step hands AdamW.grad_sink() to loss_and_grads: the LLM's weight gradients (lm_head, q/k/v/o/gate/up/down of every layer, the
    embedding table) are written by the wgrad GEMMs / the un-permutations INTO the flat gradient buffer.  Same bits as the stand-alone tensors,
    the returned tensors alias the buffer, step() finds them in place (and still copies the rest), and a second step overwrites them."""
    import _cpu_backend
    from test_host_orchestration_cpu import CFGS, build, make_inputs
    from stllm_amd import runtime, training
    model = build(CFGS["mean_pooling"], vit_depth=1, qf_layers=2, llm_layers=1)
    samples, _ = make_inputs(2, 4, False)
    with _cpu_backend.installed(), runtime.use_dtype("fp32"):
        _, _, want = training.loss_and_grads(model, samples)
        opt = training.AdamW(list(training.trainable_parameters(model)), lr=1e-3, max_grad_norm=1.0)
        opt.gflat.fill_(float("nan"))                       # whatever is not written in place must arrive by step()'s copy
        _, _, got = training.loss_and_grads(model, samples, sink=opt.grad_sink())
        assert set(got) == set(want)
        lo, hi = opt.gflat.data_ptr(), opt.gflat.data_ptr() + 4 * opt.gflat.numel()
        in_place = {n for n, g in got.items() if lo <= g.data_ptr() < hi}
        llm = {n for n in want if n.endswith("_proj.weight") and ".layers." in n} | {"lm_head.weight", "model.embed_tokens.weight"}
        assert in_place == llm, in_place ^ llm
        for n in want:
            assert torch.equal(got[n], want[n]), n
        norm = opt.step(got)
        for n, off, p in zip(opt.names, opt.offsets, opt.params):
            assert torch.equal(opt.gflat[off: off + p.numel()].view(p.shape), want[n]), n
        assert abs(norm - float(torch.sqrt(sum((g.double() ** 2).sum() for g in want.values())))) <= 1e-4 * norm
        # a STALE grads dict: its aliasing entries were consumed by the step above — a second step() with it would silently apply whatever
        # the buffer holds now; it is refused instead.  Clones, and a fresh producer run, are fine.
        with pytest.raises(RuntimeError, match="aliases the optimizer's flat buffer"):
            opt.step(got)
        opt.step({n: g.clone() for n, g in got.items()})
        l1, _, _ = training.train_step(model, samples, opt)  # the next step: every slot overwritten, nothing stale, nothing NaN
        assert torch.isfinite(opt.gflat).all() and torch.isfinite(l1)
    assert opt.grad_sink()("no.such.parameter") is None


def test_btadapter_backbone_end_to_end_matches_reference():
    """The reference's main training config (config/instructblipbase_stllm_qa.yaml: eva_btadapter_g backbone, Q-Former text input,
    video_input all, dynamic masking + MVM loss): loss_and_grads carries the gradient through llama_proj, the
    frozen Q-Former and ln_vision into the adapter branch; every trainable tensor (LLM, projector, all BTAdapter*) matches the
    REFERENCE's own loss.backward() (tests/golden/backward.npz, case "btadapter": 4 ViT blocks, 3 adapter layers, 1 Llama layer).
    freeze_btadapter=True names exactly the non-adapter entries."""
    import _cpu_backend
    from test_host_orchestration_cpu import build
    from stllm_amd import runtime, training
    g = golden("backward")
    model = build(dict(BT_CASE, image_size=224, num_query_token=32, max_txt_len=32, end_sym=" 2"), vit_depth=4, qf_layers=2, llm_layers=1)
    instr, answers = product_samples(g, "btadapter", True)
    samples = {"image": T("input.video", (2, 4, 3, 224, 224)), "instruction_input": instr, "answer": answers,
               "mask": torch.from_numpy(g["btadapter.mask"])}
    with _cpu_backend.installed(), runtime.use_dtype("fp32"):
        loss, _, grads = training.loss_and_grads(model, samples)
    assert any("BTAdapter" in n for n in grads)
    check_against_fixture(g, "btadapter", loss.item(), grads, 5e-4)
    assert set(grads) == {n for n, _ in training.trainable_parameters(model)}
    assert {n for n, _ in training.trainable_parameters(model, freeze_btadapter=True)} == {n for n in grads if "BTAdapter" not in n}


@pytest.mark.parametrize("text", [True])       # the text-free variant runs inside the end-to-end BT-Adapter test
def test_qformer_backward_to_image_tokens(text):
    """training_vision.qformer_backward: the dgrad-only sweep through the frozen Q-Former (2 layers: one with cross-attention, one
    without; with and without the text stream and its key mask) against autograd over the oracle's qformer_forward."""
    import _cpu_backend
    from test_host_orchestration_cpu import CFGS, build
    from stllm_amd import runtime, training_vision
    cfg = CFGS["instructblip_residual_text" if text else "mean_pooling"]
    model = build(cfg, vit_depth=1, qf_layers=2, llm_layers=1)
    sm = model.model.stllm_model
    sd = sd_from(shapes.stllm_model_shapes(1, 2, text, cfg["video_input"], False, qf_vocab=32000))
    p = "model.stllm_model."
    n, P = 3, 257
    enc = T("input.qf_enc", (n, P, 1408), 0.7)
    R = T("input.qf_dout", (n, 32, 768), 1.0)
    ids = mask = att = None
    if text:
        ids = torch.tensor([[1, 17, 23, 9, 4], [1, 8, 0, 0, 0], [1, 5, 6, 7, 0]])
        mask = torch.tensor([[1, 1, 1, 1, 1], [1, 1, 0, 0, 0], [1, 1, 1, 1, 0]])
        att = torch.cat([torch.ones(n, 32, dtype=torch.long), mask], dim=1)
    ev = enc.clone().requires_grad_(True)
    with torch.enable_grad():
        out = O.qformer_forward(sd[p + "query_tokens"].expand(n, -1, -1), ev, sd, p + "Qformer.bert.", ids, att)[:, :32]
        (out * R).sum().backward()
    with _cpu_backend.installed(), runtime.use_dtype("fp32"):
        hq32, _, tape = training_vision.qformer_forward_taped(sm.Qformer.bert, sm.query_tokens[0], enc.reshape(n * P, 1408).clone(), n, ids, mask)
        d_enc = training_vision.qformer_backward(sm.Qformer.bert, tape, R.reshape(n * 32, 768))
    assert (hq32.view(n, 32, 768) - out.detach()).abs().max() <= 2e-5 * out.abs().max()
    want = ev.grad.reshape(n * P, 1408)
    assert (d_enc - want).abs().max().item() <= 3e-4 * want.abs().max().item()


def test_btadapter_branch_backward_with_stochastic_depth():
    """training_vision.btadapter_backward with injected DropPath factors at the three call sites of every adapter layer (incl. a
    sample whose whole block output is dropped): gradients of every BTAdapter* parameter for a random output gradient against
    autograd over the oracle's btadapter_forward with the same factors (4-block ViT, 3 adapter layers, B = 2, T = 4)."""
    import _cpu_backend
    from test_host_orchestration_cpu import CFGS, build
    from stllm_amd import runtime, training_vision
    cfg = CFGS["btadapter"]
    model = build(cfg, vit_depth=4, qf_layers=2, llm_layers=1)
    vit = model.model.stllm_model.visual_encoder
    p = "model.stllm_model.visual_encoder."
    sd = sd_from(shapes.stllm_model_shapes(4, 2, False, cfg["video_input"], False, vit_model=cfg["vit_model"], qf_vocab=32000))
    x = T("input.video", (2, 4, 3, 224, 224))
    drop = training_vision.drop_path_factors(2, 4, depth=3, drop_prob=0.3, generator=torch.Generator().manual_seed(5))
    drop[1]["o"] = torch.tensor([0.0, 1.0 / 0.7])                      # sample 0 loses its whole branch state after layer 1
    assert any((d["t"] == 0).any() for d in drop) and any((d["s"] == 0).any() for d in drop)
    names = [n for n in sd if n.startswith(p) and "BTAdapter" in n]
    for n in names:
        sd[n].requires_grad_(True)
    R = T("input.bt_dout", (2 * 4 * 257, 1408), 1.0)
    with torch.enable_grad():
        out = O.btadapter_forward(x, sd, p, 3, drop=drop).reshape(-1, 1408)
        (out * R).sum().backward()
    with _cpu_backend.installed(), runtime.use_dtype("fp32"):
        got_out, tape = training_vision.btadapter_forward_taped(vit, x, drop)
        grads = training_vision.btadapter_backward(vit, tape, R)
    assert (got_out - out.detach()).abs().max() <= 3e-5 * out.abs().max()
    assert set(grads) == set(names)
    for n in names:
        want = sd[n].grad
        assert (grads[n] - want).abs().max().item() <= 3e-4 * max(want.abs().max().item(), 1e-6), n


def test_train_step_on_an_image_batch_skips_the_pooling_parameters():
    """T == 1 (`use_image`, st_llm.py:324-326): the global-local module is bypassed, so down_proj / up_proj get no gradient —
    loss_and_grads leaves them out and the optimizer (like torch.optim.AdamW with grad None) leaves them untouched."""
    import _cpu_backend
    from test_host_orchestration_cpu import CFGS, build, make_inputs
    from stllm_amd import runtime, training
    cfg = CFGS["instructblip_residual_text"]
    model = build(cfg, vit_depth=1, qf_layers=2, llm_layers=1)
    samples, _ = make_inputs(2, 1, True)
    with _cpu_backend.installed(), runtime.use_dtype("fp32"):
        opt = training.AdamW(list(training.trainable_parameters(model)), lr=1e-3, max_grad_norm=1.0)
        sm = model.model.stllm_model
        pool = lambda n: n.startswith(("model.stllm_model.down_proj", "model.stllm_model.up_proj"))   # (not the LLM's mlp.down_proj / up_proj)
        frozen_before = {n: p.detach().clone() for n, p in model.named_parameters() if pool(n)}
        assert frozen_before
        head_before = model.lm_head.weight.detach().clone()
        loss, _, grads = training.loss_and_grads(model, samples)
        assert not any(pool(n) for n in grads) and "lm_head.weight" in grads
        l0, _, norm = training.train_step(model, samples, opt)
        assert norm > 0 and not torch.equal(head_before, model.lm_head.weight)
        for n, p in model.named_parameters():
            if n in frozen_before:
                assert torch.equal(p, frozen_before[n]), n
        i_skip = [i for i, n in enumerate(opt.names) if pool(n)]
        assert i_skip and all(opt.steps[i] == 0 for i in i_skip) and max(opt.steps) == 1
