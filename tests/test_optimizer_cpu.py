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
    q.put((rank, [p.detach().clone() for _, p in named], norms, opt.m.numel()))
    dist.barrier()
    dist.destroy_process_group()


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
            assert torch.allclose(got, want, rtol=1e-5, atol=1e-6), (rank, n, (got - want).abs().max())


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
