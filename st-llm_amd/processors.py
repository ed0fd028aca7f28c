"""Host mirror of the reference's demo-time frame transform, running on the GPU.

Reference (stllm/conversation/conversation.py:190-198):

    self.transform = T.Compose([GroupScale(224, interpolation=BICUBIC), GroupCenterCrop(224), Stack(),
                                ToTorchFormatTensor(), GroupNormalize(input_mean, input_std)])
    video_frames = self.transform(raw_frames).to(self.device)          # :277  [T*3, 224, 224] fp32

Here the raw uint8 frames go to the GPU first (a quarter of the bytes of the fp32 tensor, and no CPU resampling) and
`stllm_preprocess_frames` (st-llm_amd/csrc/preprocess.hip) produces the same fp32 tensor bit for bit — see
oracle/preprocess_oracle.py and tests/test_preprocess_*.py.
"""
import numpy as np
import torch

from . import hip


def _to_uint8_thwc(frames):
    """list of PIL images / list of HWC arrays / numpy or torch uint8 [T, H, W, 3] -> torch uint8 [T, H, W, 3] (host or device)"""
    if isinstance(frames, (list, tuple)):
        frames = np.stack([np.asarray(f) for f in frames])
    if isinstance(frames, np.ndarray):
        frames = torch.from_numpy(np.ascontiguousarray(frames))
    if not isinstance(frames, torch.Tensor) or frames.dtype != torch.uint8 or frames.dim() != 4 or frames.shape[-1] != 3:
        raise TypeError("raw frames must be uint8 RGB [T, H, W, 3] (or a list of PIL images / HWC arrays of one size)")
    return frames


class VideoTransform:
    """`Chat.transform`: raw RGB frames -> CLIP-normalised float32 [T*3, 224, 224] on `device` (the layout upload_video
    reshapes with `.view(bt // 3, 3, w, h)`, conversation.py:278-279)."""

    def __init__(self, device="cuda:0"):
        self.device = device

    def __call__(self, raw_frames):
        f = _to_uint8_thwc(raw_frames).to(self.device, non_blocking=True)
        out = hip.preprocess_frames(f)            # [T, 3, 224, 224]
        return out.view(-1, 224, 224)


def is_raw_frames(video):
    """True for what the reference calls `raw_frames` (decoded uint8 frames), False for an already transformed tensor."""
    if isinstance(video, (list, tuple)):
        return True
    if isinstance(video, np.ndarray):
        return video.dtype == np.uint8
    return isinstance(video, torch.Tensor) and video.dtype == torch.uint8
