"""CPU: stllm_amd.training.AdamW — the update rule against torch.optim.AdamW + clip_grad_norm_ (what HF Trainer runs for the
reference), and the ZeRO-1 sharding (reduce-scatter / local update / all-gather) under gloo with world_size 2 against the
single-process result on the averaged gradients.  Kernel entry points come from the test-only contract backend."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import _cpu_backend


def _params(seed=0):
    g = torch.Generator().manual_seed(seed)
    shapes = [(37, 11), (64,), (5, 7, 3), (129,)]
    return [(f"p{i}", torch.nn.Parameter(torch.randn(s, generator=g), requires_grad=False)) for i, s in enumerate(shapes)]


def _grads(named, seed):
    g = torch.Generator().manual_seed(100 + seed)
    return {n: torch.randn(p.shape, generator=g) * 3.0 for n, p in named}


@pytest.mark.parametrize("wd,clip", [(0.0, 1.0), (0.05, None), (0.01, 0.3)])
def test_adamw_matches_torch(wd, clip):
    from stllm_amd import training
    named = _params()
    ref_p = [torch.nn.Parameter(p.detach().clone()) for _, p in named]
    ref = torch.optim.AdamW(ref_p, lr=3e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd)
    with _cpu_backend.installed():
        opt = training.AdamW(named, lr=3e-3, weight_decay=wd, max_grad_norm=clip)
        for step in range(4):
            grads = _grads(named, step)
            for rp, (n, _) in zip(ref_p, named):
                rp.grad = grads[n].clone()
            want_norm = torch.nn.utils.clip_grad_norm_(ref_p, clip if clip else 1e30).item()
            ref.step()
            got_norm = opt.step(grads)
            assert abs(got_norm - want_norm) <= 1e-5 * want_norm
            for rp, (n, p) in zip(ref_p, named):
                assert torch.allclose(p, rp, rtol=2e-6, atol=2e-7), (n, step, (p - rp).abs().max())


def test_adamw_skips_parameters_without_a_gradient_like_torch():
    """Image batches (T == 1: no pooling -> no down_proj / up_proj gradient) and un-masked batches (no mvm_decoder gradient) give
    loss_and_grads dictionaries WITHOUT those names: torch.optim.AdamW skips a parameter whose .grad is None (no decay, no moment
    update, its per-parameter step count stays), clip_grad_norm_ ignores it."""
    from stllm_amd import training
    named = _params()
    ref_p = [torch.nn.Parameter(p.detach().clone()) for _, p in named]
    ref = torch.optim.AdamW(ref_p, lr=3e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.02)
    missing_at = {0: ("p1",), 1: (), 2: ("p1", "p2"), 3: ("p0",), 4: ()}
    with _cpu_backend.installed():
        opt = training.AdamW(named, lr=3e-3, weight_decay=0.02, max_grad_norm=0.7)
        for step in range(5):
            grads = {n: g for n, g in _grads(named, step).items() if n not in missing_at[step]}
            for rp, (n, _) in zip(ref_p, named):
                rp.grad = grads[n].clone() if n in grads else None
            want_norm = torch.nn.utils.clip_grad_norm_(ref_p, 0.7).item()
            ref.step()
            got_norm = opt.step(grads)
            assert abs(got_norm - want_norm) <= 1e-5 * want_norm
            for rp, (n, p) in zip(ref_p, named):
                assert torch.allclose(p, rp, rtol=2e-6, atol=2e-7), (n, step, (p - rp).abs().max())
        assert opt.steps == [4, 3, 4, 5]
        sd = opt.state_dict()
        opt2 = training.AdamW(_params(), lr=3e-3, weight_decay=0.02, max_grad_norm=0.7)
        opt2.load_state_dict(sd)
        assert opt2.steps == opt.steps


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from stllm_amd import training
    named = _params()
    with _cpu_backend.installed():
        opt = training.AdamW(named, lr=3e-3, weight_decay=0.01, max_grad_norm=0.5, group=dist.group.WORLD, world_size=world, rank=rank)
        norms = []
        for step in range(3):
            norms.append(opt.step(_grads(named, 10 * step + rank)))      # every rank has its own micro-batch gradient
    q.put((rank, [p.detach().numpy().copy() for _, p in named], norms, opt.m.numel()))   # by value: the worker may exit first
    dist.barrier()
    dist.destroy_process_group()


def _presence_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from stllm_amd import training
    named = _params()
    with _cpu_backend.installed():
        opt = training.AdamW(named, lr=3e-3, weight_decay=0.01, max_grad_norm=0.5, group=dist.group.WORLD, world_size=world, rank=rank)
        for step in range(3):
            g = _grads(named, 10 * step + rank)
            if rank == 1:
                g.pop("p1")                      # e.g. an image batch on rank 1: no pooling-MLP gradient there
                g.pop("p3", None) if step == 1 else None
            if step == 2:
                g.pop("p2")                      # nobody has p2 in the last step: skipped everywhere
            opt.step(g)
    q.put((rank, [p.detach().numpy().copy() for _, p in named], list(opt.steps)))
    dist.barrier()
    dist.destroy_process_group()


def test_zero1_gradient_presence_is_global():
    """A parameter with a gradient on ONE rank only (mixed image / video batches): the averaged gradient is non-zero, so the owner
    of its shard must apply it and every rank must advance that parameter's step count — as torch DDP does."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_presence_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from stllm_amd import training
    named = _params()
    with _cpu_backend.installed():
        opt = training.AdamW(named, lr=3e-3, weight_decay=0.01, max_grad_norm=0.5)
        for step in range(3):
            ga, gb = _grads(named, 10 * step), _grads(named, 10 * step + 1)
            gb["p1"] = torch.zeros_like(gb["p1"])
            if step == 1:
                gb["p3"] = torch.zeros_like(gb["p3"])
            avg = {n: (ga[n] + gb[n]) / 2 for n in ga}
            if step == 2:
                avg.pop("p2")
            opt.step(avg)
    for rank, params, steps in res:
        assert steps == [3, 3, 2, 3] == opt.steps, (rank, steps)
        for got, (n, want) in zip(params, named):
            assert torch.allclose(torch.from_numpy(got), want, rtol=1e-5, atol=1e-6), (rank, n)


def test_zero1_sharded_step_matches_single_process():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from stllm_amd import training
    named = _params()
    with _cpu_backend.installed():
        opt = training.AdamW(named, lr=3e-3, weight_decay=0.01, max_grad_norm=0.5)
        norms = []
        for step in range(3):
            ga, gb = _grads(named, 10 * step), _grads(named, 10 * step + 1)
            norms.append(opt.step({n: (ga[n] + gb[n]) / 2 for n in ga}))
    total = sum(p.numel() for _, p in named)
    for rank, params, rnorms, shard in res:
        assert shard * world >= total and shard * world - total < world * 64       # optimizer state really is sharded
        for a, b in zip(rnorms, norms):
            assert abs(a - b) <= 1e-5 * b
        for got, (n, want) in zip(params, named):
            assert torch.allclose(torch.from_numpy(got), want, rtol=1e-5, atol=1e-6), (rank, n)


def test_cosine_schedule_matches_hf():
    import math
    from transformers import get_cosine_schedule_with_warmup
    from stllm_amd import training
    total, base = 137, 2e-5
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=base)
    sch = get_cosine_schedule_with_warmup(opt, math.ceil(total * 0.03), total)
    for step in range(total):
        assert abs(sch.get_last_lr()[0] - training.cosine_lr(step, total, base)) <= 1e-12, step
        opt.step()
        sch.step()


def test_optimizer_resume_is_exact():
    """checkpoint after 2 steps (trainable masters + optimizer shard), rebuild, resume: steps 3-4 are bit-identical"""
    from stllm_amd import training
    with _cpu_backend.installed():
        named = _params()
        opt = training.AdamW(named, lr=3e-3, weight_decay=0.01, max_grad_norm=0.5)
        for step in range(2):
            opt.step(_grads(named, step))
        ckpt_p = [p.detach().clone() for _, p in named]
        ckpt_o = opt.state_dict()
        for step in range(2, 4):
            opt.step(_grads(named, step))
        want = [p.detach().clone() for _, p in named]
        named2 = _params(seed=99)
        for (_, p), c in zip(named2, ckpt_p):
            p.data.copy_(c)
        opt2 = training.AdamW(named2, lr=3e-3, weight_decay=0.01, max_grad_norm=0.5)
        opt2.load_state_dict(ckpt_o)
        for step in range(2, 4):
            opt2.step(_grads(named2, step))
    for (n, p), w in zip(named2, want):
        assert torch.equal(p, w), n
    with pytest.raises(ValueError):
        training.AdamW(_params()[:2], lr=1e-3).load_state_dict(ckpt_o)


# ---- data-parallel training of a small Llama: real gradients + ZeRO-1 step under gloo == one process on the averaged gradients ----
def _tiny_lm():
    from stllm_amd import synth
    from stllm_amd.models.st_llm import STLLMForCausalLM, StllmConfig
    m = STLLMForCausalLM(StllmConfig(hidden_size=256, intermediate_size=384, num_hidden_layers=2, num_attention_heads=2, vocab_size=256),
                         device="cpu")
    synth.fill_module_(m, 0, "")
    return m


def _lm_named(m):
    return [(n, p) for n, p in m.named_parameters()]


def _lm_grads(model, rank, step):
    """CE gradients of the micro-batch (rank, step): taped forward + explicit backward on the contract backend"""
    from stllm_amd import hip, training
    g = torch.Generator().manual_seed(1000 + 17 * step + rank)
    B, S = 2, 24
    ids = torch.randint(0, 256, (B, S), generator=g)
    emb = model.model.embed_tokens(ids)
    labels = torch.randint(0, 256, (B * S,), generator=g).to(torch.int32)
    h32, h16, tape = training.llama_forward_taped(model.model, emb, None)
    W = model.lm_weight(torch.float32)
    logits = hip.gemm(h16, W, dtype=torch.float32, out_f32=True)
    loss = hip.cross_entropy_rows(logits, labels).mean()
    dlog = hip.cross_entropy_bwd(logits, labels, 1.0 / (B * S), dtype=torch.float32, vocab=256)
    d_h16, dw = training.linear_bwd(dlog, h16, W, torch.float32)
    d_emb, grads = training.llama_backward(model.model, tape, d_h16, None)
    grads["lm_head.weight"] = dw
    d_table = torch.zeros_like(model.model.embed_tokens.weight)
    hip.scatter_add_rows(d_emb, (-(ids.reshape(-1)) - 1).to(torch.int32), d_table, d_table)
    grads["model.embed_tokens.weight"] = d_table
    return loss.item(), grads


def _dp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from stllm_amd import runtime, training
    torch.set_grad_enabled(False)
    model = _tiny_lm()
    with _cpu_backend.installed(), runtime.use_dtype("fp32"):
        opt = training.AdamW(_lm_named(model), lr=1e-2, max_grad_norm=1.0, group=dist.group.WORLD, world_size=world, rank=rank)
        losses = []
        for step in range(3):
            loss, grads = _lm_grads(model, rank, step)
            opt.step(grads)
            model._lm_packed = {}
            model.model.repack()
            losses.append(loss)
    q.put((rank, losses, {n: p.detach().numpy().copy() for n, p in model.named_parameters()}))   # by value: the worker may exit first
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_llama_training_matches_single_process():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from stllm_amd import runtime, training
    torch.set_grad_enabled(False)
    model = _tiny_lm()
    with _cpu_backend.installed(), runtime.use_dtype("fp32"):
        opt = training.AdamW(_lm_named(model), lr=1e-2, max_grad_norm=1.0)
        for step in range(3):
            (l0, g0), (l1, g1) = _lm_grads(model, 0, step), _lm_grads(model, 1, step)
            assert abs(l0 - res[0][1][step]) < 1e-5 and abs(l1 - res[1][1][step]) < 1e-5      # same weights on every rank, every step
            opt.step({n: (g0[n] + g1[n]) / 2 for n in g0})
            model._lm_packed = {}
            model.model.repack()
    for rank, _, params in res:
        for n, p in model.named_parameters():
            assert torch.allclose(torch.from_numpy(params[n]), p, rtol=1e-5, atol=1e-6), (rank, n)


def test_weight_gradient_unpackers_invert_the_packers():
    """training.unpack_qkv_grad / unpack_gate_up_grad are the exact inverses of pack.llama_qkv / pack.llama_gate_up (index permutations)"""
    from stllm_amd import pack, training
    g = torch.Generator().manual_seed(7)
    wq, wk, wv = (torch.randn(256, 64, generator=g) for _ in range(3))
    q, k, v = training.unpack_qkv_grad(pack.llama_qkv(wq, wk, wv, torch.float32, n_heads=2), 2)
    assert torch.equal(q, wq) and torch.equal(k, wk) and torch.equal(v, wv)
    wg, wu = torch.randn(96, 40, generator=g), torch.randn(96, 40, generator=g)
    a, b = training.unpack_gate_up_grad(pack.llama_gate_up(wg, wu, torch.float32))
    assert torch.equal(a, wg) and torch.equal(b, wu)


def test_trainable_state_dict_round_trip():
    """what a checkpoint holds (train_hf.py:188-203: the parameters that require grad) loads back by name into a fresh model"""
    from stllm_amd import training
    a, b = _tiny_lm(), _tiny_lm()
    with torch.no_grad():
        for _, p in training.trainable_parameters(a):
            p.add_(1.0)
    sd = training.trainable_state_dict(a)
    assert set(sd) == {n for n, _ in a.named_parameters()}            # a plain Llama: everything is trainable
    missing, unexpected = b.load_state_dict(sd, strict=False)
    assert not unexpected and not missing
    for (n, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
        assert torch.equal(p, q), n
