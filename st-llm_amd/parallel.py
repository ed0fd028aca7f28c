"""Frame-parallel visual encode + clip-parallel prefill over the GPUs of one node (RCCL over xGMI).

New functionality with no reference counterpart (the reference only has ZeRO/DDP data parallelism,
SURVEY.md §2b); correctness criterion: the gathered token block is bit-identical to the 1-GPU path.

  1. the B*T frames of a batch are split into `world` contiguous ranges; each rank runs
     ViT -> ln_vision -> Q-Former -> projector on its range (frames are independent for eva_clip_g;
     BT-Adapter's temporal attention couples the frames of a clip => that backbone shards by clip only);
  2. ONE all-gather of the projected tokens [frames_per_rank, 32, 4096] (8.4 MB/rank fp32 at T=16) — every
     rank then holds all B*T*32 tokens.  xGMI is point-to-point, the message is tiny, so this is
     latency- not bandwidth-bound: a single ncclAllGather on the compute stream, no overlap machinery;
  3. pooling needs all T frames of a clip (mean over T / residual index) => done after the gather, on the
     rank that owns the clip: clip c is prefilled by rank c % world (replicating the prefill would cap the
     speed-up at the ViT/LLM FLOP ratio, SURVEY.md §7 hard-part 3).
"""
import torch
import torch.distributed as dist


def frame_range(n_frames, rank, world):
    """contiguous, balanced split (first n_frames % world ranks get one extra frame)"""
    q, r = divmod(n_frames, world)
    start = rank * q + min(rank, r)
    return start, start + q + (1 if rank < r else 0)


def clips_of_rank(n_clips, rank, world):
    return [c for c in range(n_clips) if c % world == rank]


def all_gather_frames(local, n_frames, rank, world, group=None):
    """local: [n_local, ...] tokens of this rank's frame range -> [n_frames, ...] on every rank.
    Ragged ranges are padded to the largest range for the collective and trimmed afterwards."""
    if world == 1:
        return local
    sizes = [frame_range(n_frames, r, world) for r in range(world)]
    mx = max(e - s for s, e in sizes)
    pad = local
    if local.shape[0] < mx:
        pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        pad[: local.shape[0]] = local
    out = torch.empty((world * mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad.contiguous(), group=group)
    if all(e - s == mx for s, e in sizes):
        return out
    return torch.cat([out[r * mx: r * mx + (e - s)] for r, (s, e) in enumerate(sizes)], dim=0)


def encode_frames_parallel(encode_fn, frames, rank, world, group=None, token_shape=(32, 4096)):
    """frames: [N, 3, 224, 224] (the full batch, or anything indexable by the frame range);
    encode_fn(frames_slice) -> tokens [n, 32, D] fp32.  Returns tokens of all N frames on every rank."""
    n = frames.shape[0]
    s, e = frame_range(n, rank, world)
    if e > s:
        local = encode_fn(frames[s:e])
    else:  # more ranks than frames: this rank only takes part in the collective
        local = torch.zeros((0,) + tuple(token_shape), dtype=torch.float32, device=frames.device)
    return all_gather_frames(local, n, rank, world, group)
