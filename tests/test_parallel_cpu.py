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
    q.put((rank, tokens.clone(), parallel.frame_range(n_frames, rank, world, extra), parallel.clips_of_rank(5, rank, world)))
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
        assert torch.equal(tokens, ref), f"rank {rank}: gathered block differs from the 1-process result"
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
