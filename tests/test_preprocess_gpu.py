"""GPU: stllm_preprocess_frames (through the C ABI) is bit-identical to the CPU oracle / the reference's transform chain."""
import os

import numpy as np
import pytest
import torch

import preprocess_oracle as P

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hip():
    from stllm_amd import hip as h
    return h


@pytest.mark.parametrize("name", ["up", "down", "portrait", "same", "odd"])
def test_kernel_vs_reference_fixture(hip, name):
    fx = np.load(os.path.join(ROOT, "tests", "golden", "preprocess.npz"))
    frames = fx[f"{name}.frames"]
    out = hip.preprocess_frames(torch.from_numpy(frames).cuda()).cpu().numpy().reshape(-1, 224, 224)
    assert np.array_equal(out[:, ::5, ::5], fx[f"{name}.out_sub"])
    assert np.array_equal(out, P.video_transform(frames)), "HIP preprocessing is not bit-identical to the oracle"


@pytest.mark.parametrize("T_,H,W", [(3, 240, 320), (2, 360, 640), (1, 1080, 1920), (2, 400, 300), (1, 224, 640), (2, 77, 91), (1, 231, 229)])
def test_kernel_vs_oracle_random(hip, T_, H, W):
    rng = np.random.default_rng(H * 10007 + W)
    frames = rng.integers(0, 256, (T_, H, W, 3), dtype=np.uint8)
    got = hip.preprocess_frames(torch.from_numpy(frames).cuda()).cpu().numpy().reshape(-1, 224, 224)
    ref = P.video_transform(frames)
    assert np.array_equal(got, ref), f"max abs diff {np.abs(got - ref).max()}"


def test_chat_accepts_raw_frames(hip):
    """Chat.upload_video on decoded uint8 frames == on the pre-transformed tensor (conversation.py:276-279)."""
    from stllm_amd.processors import VideoTransform, is_raw_frames
    rng = np.random.default_rng(5)
    frames = rng.integers(0, 256, (4, 180, 320, 3), dtype=np.uint8)
    assert is_raw_frames(frames) and not is_raw_frames(torch.zeros(12, 224, 224))
    t = VideoTransform("cuda:0")(frames)
    assert t.shape == (12, 224, 224) and t.dtype == torch.float32
    assert np.array_equal(t.cpu().numpy(), P.video_transform(frames))
    with pytest.raises(RuntimeError):
        hip.preprocess_frames(torch.zeros((1, 10, 10, 4), dtype=torch.uint8, device="cuda"))
