"""CPU restatement of the frame preprocessing in front of the hot path — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product path
(st-llm_amd/processors.py -> stllm_preprocess_frames in libstllm_hip.so) never does.

What the reference does (stllm/conversation/conversation.py:190-198, 276-279):

    T.Compose([GroupScale(224, interpolation=BICUBIC),      # stllm/test/video_transforms.py:110-124 -> torchvision Resize
               GroupCenterCrop(224),                        # video_transforms.py:54-60        -> torchvision CenterCrop
               Stack(),                                     # :367-384  concat along channels: [H, W, 3T]
               ToTorchFormatTensor(),                       # :387-407  HWC uint8 -> CHW float / 255
               GroupNormalize(mean, std)])                  # :94-107   (x - mean) / std per channel, in place, fp32
    video_frames.view(bt // 3, 3, w, h)                     # conversation.py:278-279

Two third-party pieces are not in /root/reference and are restated from their published sources:

  * torchvision == 0.15.1 (requirement.txt:3): transforms.functional.resize on a PIL image =
    `img.resize((new_w, new_h), BICUBIC)` with `new_short = size, new_long = int(size * long / short)`
    (functional._compute_resized_output_size), untouched when the short side already equals `size`;
    CenterCrop offsets `int(round((h - 224) / 2.0))`, `int(round((w - 224) / 2.0))` (Python round: half to even).
  * Pillow (unpinned by the reference; pinned here by tests/golden/preprocess.npz, generated with Pillow 12.2.0):
    Image.resize(BICUBIC) on an 8-bit RGB image = libImaging/Resample.c, ImagingResampleInner: a horizontal pass then a
    vertical pass, each a convolution with per-output-pixel coefficient rows from `precompute_coeffs` (bicubic a = -0.5,
    support 2 * max(scale, 1): antialiased when shrinking), coefficients normalised to sum 1 and quantised to
    22-bit fixed point (`normalize_coeffs_8bpc`, PRECISION_BITS = 32 - 8 - 2), accumulator started at 1 << 21,
    result `clip8(acc >> 22)`; the intermediate image between the passes is uint8.

Pinned: tests/test_preprocess_cpu.py checks `pil_resize_bicubic_u8` bit-for-bit against Pillow's own output on
random images (up- and down-scaling, odd sizes) and the whole chain against tests/golden/preprocess.npz.
"""
import numpy as np

PRECISION_BITS = 32 - 8 - 2
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)     # conversation.py:190
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)      # conversation.py:191


def _bicubic(x):
    """Resample.c bicubic_filter, a = -0.5 (double arithmetic, same operation order)."""
    a = -0.5
    x = np.abs(x)
    near = ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    far = (((x - 5) * x + 8) * x - 4) * a
    return np.where(x < 1.0, near, np.where(x < 2.0, far, 0.0))


def precompute_coeffs(in_size, out_size):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the full box [0, in_size).
    Returns (ksize, xmin[out], xcnt[out], kk int32 [out, ksize])."""
    in0, in1 = np.float32(0.0), np.float32(in_size)
    scale = float(in1 - in0) / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    xmin = np.zeros(out_size, np.int32)
    xcnt = np.zeros(out_size, np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = float(in0) + (xx + 0.5) * scale
        lo = int(center - support + 0.5)          # C (int) cast truncates toward zero; the operand is > -1 here or clamped next
        lo = max(lo, 0)
        hi = int(center + support + 0.5)
        hi = min(hi, in_size)
        n = hi - lo
        x = np.arange(n, dtype=np.float64)
        w = _bicubic((x + lo - center + 0.5) * ss)
        ww = 0.0
        for v in w:                                # sequential double sum, as in the C loop
            ww += float(v)
        if ww != 0.0:
            w = w / ww
        q = np.where(w < 0, (-0.5 + w * (1 << PRECISION_BITS)), (0.5 + w * (1 << PRECISION_BITS)))
        kk[xx, :n] = np.trunc(q).astype(np.int64).astype(np.int32)
        xmin[xx], xcnt[xx] = lo, n
    return ksize, xmin, xcnt, kk


def _resample_axis_u8(img, out_size, axis):
    """One pass of ImagingResample{Horizontal,Vertical}_8bpc along `axis` of a uint8 [H, W, C] image."""
    in_size = img.shape[axis]
    ksize, xmin, xcnt, kk = precompute_coeffs(in_size, out_size)
    src = np.moveaxis(img, axis, 0).astype(np.int64)              # [in, other, C]
    out = np.empty((out_size,) + src.shape[1:], np.uint8)
    for xx in range(out_size):
        n, lo = int(xcnt[xx]), int(xmin[xx])
        acc = np.full(src.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        acc = acc + np.tensordot(kk[xx, :n].astype(np.int64), src[lo:lo + n], axes=(0, 0))
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)   # clip8: arithmetic shift, then clamp
    return np.moveaxis(out, 0, axis)


def pil_resize_bicubic_u8(img, out_w, out_h):
    """PIL.Image.resize((out_w, out_h), BICUBIC) for an 8-bit RGB image [H, W, 3]: horizontal pass, then vertical
    (Resample.c ImagingResampleInner; a pass whose size does not change is skipped, as in Pillow)."""
    h, w = img.shape[:2]
    if (out_w, out_h) == (w, h):
        return img.copy()
    tmp = _resample_axis_u8(img, out_w, 1) if out_w != w else img
    return _resample_axis_u8(tmp, out_h, 0) if out_h != h else tmp


def tv_resized_size(h, w, size=224):
    """torchvision 0.15.1 functional._compute_resized_output_size for size=[224]: returns (new_h, new_w)."""
    short, long = (w, h) if w <= h else (h, w)
    new_short, new_long = size, int(size * long / short)
    return (new_long, new_short) if w <= h else (new_short, new_long)


def center_crop_offsets(h, w, crop=224):
    """torchvision CenterCrop: int(round((h - crop) / 2.0)) with Python's round-half-to-even."""
    return int(round((h - crop) / 2.0)), int(round((w - crop) / 2.0))


def video_transform(frames_u8):
    """Chat.transform (conversation.py:192-198) on uint8 frames [T, H, W, 3] (H, W >= 224 after the resize).
    Returns float32 [T*3, 224, 224] exactly as the reference hands it to upload_video's `.view(bt // 3, 3, w, h)`."""
    frames_u8 = np.asarray(frames_u8)
    T, H, W, _ = frames_u8.shape
    nh, nw = tv_resized_size(H, W)
    top, left = center_crop_offsets(nh, nw)
    out = np.empty((T * 3, 224, 224), np.float32)
    mean = np.array(CLIP_MEAN * T, np.float32)
    std = np.array(CLIP_STD * T, np.float32)
    for t in range(T):
        r = frames_u8[t] if (nh, nw) == (H, W) else pil_resize_bicubic_u8(frames_u8[t], nw, nh)
        c = r[top:top + 224, left:left + 224]                         # GroupCenterCrop
        x = c.transpose(2, 0, 1).astype(np.float32) / np.float32(255)  # ToTorchFormatTensor: .float().div(255)
        out[3 * t:3 * t + 3] = x
    out = (out - mean[:, None, None]) / std[:, None, None]            # GroupNormalize: t.sub_(m).div_(s), fp32
    return out.astype(np.float32)
