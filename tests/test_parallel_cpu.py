"""CPU, world_size 2, gloo: the frame-parallel collective logic of stllm_amd.parallel (the compute is injected, so no
GPU is needed): gathered token block bit-identical to the single-process result, ragged frame counts, clip ownership."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _encode(frames):
    # stand-in for ViT -> Q-Former -> projector: deterministic per-frame function with a cross-feature mix
    x = frames.reshape(frames.shape[0], -1)[:, : 32 * 8].reshape(-1, 32, 8)
    return torch.tanh(x * 1.7) + x.flip(-1) * 0.25


def _worker(rank, world, port, n_frames, q, extra=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from stllm_amd import parallel
    torch.manual_seed(0)
    frames = torch.randn(n_frames, 3, 16, 16)
    tokens = parallel.encode_frames_parallel(_encode, frames, rank, world, token_shape=(32, 8), extra=extra)
    q.put((rank, tokens.numpy().copy(), parallel.frame_range(n_frames, rank, world, extra), parallel.clips_of_rank(5, rank, world)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_frames,extra", [(32, None), (7, None), (1, None), (32, [12.0, 0.0]), (9, [0.0, 30.0])])
def test_frame_parallel_allgather_matches_single_process(n_frames, extra):
    """extra = the prefill load of every rank in frame units: uneven (even empty) frame ranges, same gathered block"""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_frames, q, extra)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    torch.manual_seed(0)
    ref = _encode(torch.randn(n_frames, 3, 16, 16))
    ranges = {}
    for rank, tokens, fr, clips in res:
        assert torch.equal(torch.from_numpy(tokens), ref), f"rank {rank}: gathered block differs from the 1-process result"
        ranges[rank] = fr
        assert clips == [c for c in range(5) if c % world == rank]
    assert ranges[0][0] == 0 and ranges[0][1] == ranges[1][0] and ranges[1][1] == n_frames
    if extra == [12.0, 0.0]:
        assert ranges[0] == (0, 10) and ranges[1] == (10, 32)        # 10 + 12 == 22 + 0: both ranks finish together
    if extra == [0.0, 30.0]:
        assert ranges[0] == (0, 9) and ranges[1] == (9, 9)           # the loaded rank only takes part in the collective


def test_frame_range_and_clip_ownership():
    from stllm_amd import parallel
    for n in (1, 7, 16, 64, 256):
        for w in (1, 2, 4, 8):
            r = [parallel.frame_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n and all(r[i][1] == r[i + 1][0] for i in range(w - 1))
            assert max(e - s for s, e in r) - min(e - s for s, e in r) <= 1
    owned = sorted(c for k in range(8) for c in parallel.clips_of_rank(4, k, 8))
    assert owned == [0, 1, 2, 3]
    # config 3 on 8 GPUs: 4 clips x 64 frames, ranks 0-3 also prefill one clip (~12 frames of work each)
    assert parallel.frame_counts(256, 8, [12.0] * 4 + [0.0] * 4) == [26, 26, 26, 26, 38, 38, 38, 38]
    assert parallel.frame_counts(256, 8, [22.0] * 4 + [0.0] * 4) == [21, 21, 21, 21, 43, 43, 43, 43]   # STLLMModel.prefill_cost_frames as measured in round 4
    assert parallel.frame_counts(256, 8, [12.0] * 8) == [32] * 8 == parallel.frame_counts(256, 8)
    for n in (1, 7, 16, 64, 256):
        for w in (2, 4, 8):
            for ex in ([5.0 * (k % 2) for k in range(w)], [100.0] + [0.0] * (w - 1), [float(k) for k in range(w)]):
                c = parallel.frame_counts(n, w, ex)
                assert sum(c) == n and min(c) >= 0
                lv = [c[k] + ex[k] for k in range(w) if c[k] > 0]
                assert max(lv) - min(lv) <= 1.0 + 1e-9, (n, w, ex, c)    # levelled to within one frame among the ranks that got frames


def test_gather_needed_only_when_a_rank_prefills_frames_it_did_not_encode():
    from stllm_amd import parallel as P
    # one clip per GPU (weak scaling): ranges == clips, nothing to exchange
    for world in (2, 4, 8):
        assert not P.gather_needed(world * 16, 16, world, [12.0] * world)
        assert not P.gather_needed(world * 16, 16, world, None)
    assert P.gather_needed(4 * 16, 16, 2, [24.0, 24.0])                            # 4 clips on 2 ranks: rank 0 prefills clips 0 and 2, encodes 0 and 1
    assert P.gather_needed(4 * 64, 64, 8, [12.0] * 4 + [0.0] * 4)                  # config 3 on 8 GPUs: frames levelled, clips split
    assert P.gather_needed(3 * 2, 2, 2, None)                                      # 3 clips on 2 ranks
    assert not P.gather_needed(16, 16, 1, None)


# ---- round 5: clip teams, point-to-point exchange per clip ------------------------------------------------------------------
def test_team_plan_tables():
    from stllm_amd import parallel as P
    # config 3 on 8 GPUs: 4 clips x 64 frames -> 4 teams of 2 (rank c and rank c + 4), 32 / 32 frames, the prefill shared (sequence-parallel)
    p = P.TeamPlan(4, 64, 8)
    assert p.team == [[0, 4], [1, 5], [2, 6], [3, 7]] and all(p.sp)
    assert p.frames == [[(0, 32), (32, 64)]] * 4
    assert [p.clips_of(r) for r in range(8)] == [[0], [1], [2], [3], [0], [1], [2], [3]]
    assert p.encodes(5) == [(1, 32, 64)] and p.receivers(1) == [1, 5] and p.exchange_needed()
    # owner-only prefill (MVM forward): latency = equal shares, throughput = the owner encodes fewer frames (water-filled against its prefill)
    p = P.TeamPlan(4, 64, 8, sp=False, balance="latency", prefill_cost_frames=22.0)
    assert p.frames[0] == [(0, 32), (32, 64)] and not any(p.sp) and p.receivers(2) == [2]
    assert [p.clips_of(r) for r in range(8)] == [[0], [1], [2], [3], [], [], [], []]
    p = P.TeamPlan(4, 64, 8, sp=False, balance="throughput", prefill_cost_frames=22.0)
    assert p.frames[3] == [(0, 21), (21, 64)]
    # fewer ranks than clips: whole clips per rank, teams of one, nothing to exchange
    p = P.TeamPlan(4, 64, 2)
    assert p.team == [[0], [1], [0], [1]] and not p.exchange_needed() and not any(p.sp)
    assert p.encodes(1) == [(1, 0, 64), (3, 0, 64)] and p.clips_of(0) == [0, 2]
    # one clip per rank (bench.py's weak-scaling c2 at N > 1), a team of three with a ragged split, more members than frames
    assert not P.TeamPlan(8, 16, 8).exchange_needed()
    p = P.TeamPlan(3, 64, 8)
    assert p.team == [[0, 3, 6], [1, 4, 7], [2, 5]] and p.frames[0] == [(0, 22), (22, 43), (43, 64)] and p.frames[2] == [(0, 32), (32, 64)]
    p = P.TeamPlan(1, 2, 4)
    assert p.frames[0] == [(0, 1), (1, 2), (2, 2), (2, 2)] and p.encodes(3) == [] and p.clips_of(3) == [0]
    for clips in (1, 2, 3, 4, 5, 8):
        for world in (1, 2, 3, 4, 8):
            for T in (1, 4, 16, 64):
                p = P.TeamPlan(clips, T, world, sp=(clips + world) % 2 == 0)
                got = sorted((c, f) for r in range(world) for c, f0, f1 in p.encodes(r) for f in range(f0, f1))
                assert got == [(c, f) for c in range(clips) for f in range(T)], "every frame is encoded exactly once"
                pre = [c for r in range(world) for c in p.clips_of(r) if not p.sp[c]]
                assert sorted(pre) == [c for c in range(clips) if not p.sp[c]], "every clip without a shared prefill has exactly one prefill rank"
    assert P.sp_row_ranges(580, 2) == [(0, 288), (288, 580)] and P.sp_row_ranges(580, 1) == [(0, 580)]
    assert P.sp_row_ranges(85, 2) == [(0, 32), (32, 85)] and P.sp_row_ranges(20, 3) == [(0, 0), (0, 0), (0, 20)]
    rr = P.sp_row_ranges(2320, 4)
    assert rr[0][0] == 0 and rr[-1][1] == 2320 and all(a[1] == b[0] for a, b in zip(rr, rr[1:])) and all(e % 32 == 0 for _, e in rr[:-1])


def _team_worker(rank, world, port, clips, T, sp, q, wire=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from stllm_amd import parallel
    torch.manual_seed(0)
    frames = torch.randn(clips * T, 3, 16, 16)
    plan = parallel.TeamPlan(clips, T, world, sp=sp, balance="throughput", prefill_cost_frames=1.0)
    local = {c: _encode(frames[c * T + f0: c * T + f1]) for c, f0, f1 in plan.encodes(rank)}
    blocks = parallel.exchange_clip_tokens(local, plan, rank, token_shape=(32, 8), wire_dtype=wire)
    q.put((rank, {c: b.numpy().copy() for c, b in blocks.items()}, plan.clips_of(rank)))   # arrays: pickled by value (a tensor travels as a shared-memory handle that dies with the worker)
    dist.barrier()
    dist.destroy_process_group()


def test_clip_team_exchange_on_a_16_bit_wire():
    """VERDICT r05 #8b: wire_dtype = bfloat16 — the sub-blocks travel in 16 bits and the sender rounds its own copy through the same dtype: every member of
    a team assembles the SAME bits (the fp32 block rounded through bf16), half the bytes per link."""
    world, clips, T = 4, 2, 6
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_team_worker, args=(r, world, port, clips, T, True, q, torch.bfloat16)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    torch.manual_seed(0)
    ref = _encode(torch.randn(clips * T, 3, 16, 16)).view(clips, T, 32, 8)
    ref16 = ref.to(torch.bfloat16).float()
    assert not torch.equal(ref, ref16)
    n = 0
    for rank, blocks, own in res:
        for c, b in blocks.items():
            assert torch.equal(torch.from_numpy(b), ref16[c]), f"rank {rank} clip {c}: not the fp32 block rounded through the wire dtype"
            n += 1
    assert n == 4      # two teams of two, both members hold their clip's block


@pytest.mark.parametrize("world,clips,T,sp", [(2, 1, 5, True), (4, 2, 6, True), (4, 2, 6, False), (4, 1, 3, True), (2, 3, 2, True)])
def test_clip_team_exchange_matches_single_process(world, clips, T, sp):
    """gloo, 2 and 4 processes: after the point-to-point exchange every rank that prefills (a share of) a clip holds that clip's whole token
    block, bit-identical to the 1-process encode; ranks outside the clip's receivers hold nothing of it (no byte moved for them)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_team_worker, args=(r, world, port, clips, T, sp, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from stllm_amd import parallel
    torch.manual_seed(0)
    ref = _encode(torch.randn(clips * T, 3, 16, 16)).view(clips, T, 32, 8)
    plan = parallel.TeamPlan(clips, T, world, sp=sp, balance="throughput", prefill_cost_frames=1.0)
    holders = {c: [] for c in range(clips)}
    for rank, blocks, own in res:
        assert sorted(blocks) == sorted(own) == plan.clips_of(rank)
        for c, b in blocks.items():
            assert torch.equal(torch.from_numpy(b), ref[c]), f"rank {rank} clip {c}"
            holders[c].append(rank)
    for c in range(clips):
        assert sorted(holders[c]) == sorted(plan.receivers(c))
