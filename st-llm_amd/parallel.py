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
     speed-up at the ViT/LLM FLOP ratio, SURVEY.md §7 hard-part 3);
  4. with fewer clips than GPUs (config 3: 4 clips on 8 GPUs) the frame ranges are NOT equal: a rank that also prefills a clip
     gets fewer frames than a rank that does not (frame_counts: one prefill ~ 12 frames of encode at S = 576), so that all
     ranks finish together — 21 / 43 frames instead of 32 / 32 in config 3 at N = 8 (one prefill ~ 22 marginal frames: measured, round 4);
  5. when the frame ranges coincide with the clips every rank prefills (one clip per GPU: bench.py's weak-scaling config 2 at
     N > 1) the all-gather would move 8.4 MB per rank that nobody reads: gather_needed() is False and the collective is skipped.
"""
import torch
import torch.distributed as dist


def frame_counts(n_frames, world, extra=None):
    """Frames per rank.  extra[r] = other work rank r has in the same step, in units of one frame's encode time (the prefill of
    the clips it owns): the counts level `frames_r + extra_r` over the ranks (water-filling), so that with FEWER clips than
    GPUs (config 3 on 8 GPUs: 4 clips) the ranks without a prefill encode more frames than the ranks with one.  extra = None or
    all-equal -> the balanced split (first n_frames % world ranks get one extra frame).  Deterministic: every rank computes the
    same table."""
    if extra is None or len(set(extra)) <= 1:
        q, r = divmod(n_frames, world)
        return [q + (1 if k < r else 0) for k in range(world)]
    extra = [float(e) for e in extra]
    order = sorted(range(world), key=lambda k: extra[k])
    level, active = 0.0, 0
    for i, k in enumerate(order):            # raise the water level until the n_frames are placed
        nxt = extra[order[i + 1]] if i + 1 < world else float("inf")
        active = i + 1
        have = sum(max(0.0, extra[k] - extra[j]) for j in order[:active])   # frames placed when the level reaches extra[k]
        room = (nxt - extra[k]) * active
        if have + room >= n_frames or i + 1 == world:
            level = extra[k] + (n_frames - have) / active
            break
    want = [max(0.0, level - extra[k]) for k in range(world)]
    counts = [int(w) for w in want]
    rest = n_frames - sum(counts)
    for k in sorted(range(world), key=lambda k: -(want[k] - counts[k]))[:rest]:   # largest remainders first
        counts[k] += 1
    return counts


def frame_range(n_frames, rank, world, extra=None):
    """contiguous range of rank `rank` under frame_counts(n_frames, world, extra)"""
    c = frame_counts(n_frames, world, extra)
    start = sum(c[:rank])
    return start, start + c[rank]


def clips_of_rank(n_clips, rank, world):
    return [c for c in range(n_clips) if c % world == rank]


def gather_needed(n_frames, T, world, extra=None):
    """False when every rank's frame range is exactly the frames of the clips it prefills (clip c -> rank c % world): the all-gather
    would then carry nothing any rank needs — the weak-scaling case of one clip per GPU — and is skipped.  Deterministic: every rank
    computes the same answer, so either all of them enter the collective or none."""
    n_clips = n_frames // T
    for r in range(world):
        s, e = frame_range(n_frames, r, world, extra)
        for c in clips_of_rank(n_clips, r, world):
            if c * T < s or (c + 1) * T > e:
                return True
    return False


def all_gather_frames(local, n_frames, rank, world, group=None, extra=None):
    """local: [n_local, ...] tokens of this rank's frame range -> [n_frames, ...] on every rank.
    Ragged ranges are padded to the largest range for the collective and trimmed afterwards."""
    if world == 1:
        return local
    sizes = [frame_range(n_frames, r, world, extra) for r in range(world)]
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


def encode_frames_parallel(encode_fn, frames, rank, world, group=None, token_shape=(32, 4096), extra=None, simulate=None):
    """frames: [N, 3, 224, 224] (the full batch, or anything indexable by the frame range);
    encode_fn(frames_slice) -> tokens [n, 32, D] fp32.  Returns tokens of all N frames on every rank.
    extra: see frame_counts (prefill load of every rank in frame units).
    simulate: a pre-computed token block [N, 32, D] standing in for the peers' ranges — ONE process measuring rank `rank`'s share of a
    `world`-rank step without a process group (bench.py's frame_parallel_projection): the collective becomes a device copy of the block
    with this rank's own tokens written into their range."""
    n = frames.shape[0]
    s, e = frame_range(n, rank, world, extra)
    if e > s:
        local = encode_fn(frames[s:e])
    else:  # more ranks than frames: this rank only takes part in the collective
        local = torch.zeros((0,) + tuple(token_shape), dtype=torch.float32, device=frames.device)
    if simulate is not None:
        out = simulate.clone()
        out[s:e] = local
        return out
    return all_gather_frames(local, n, rank, world, group, extra)
