"""CPU (-m "not gpu"): the preprocessing oracle is pinned against (a) golden vectors produced by the reference's own
transform chain (tests/golden/make_preprocess_fixtures.py) and (b) Pillow itself when it is importable."""
import os

import numpy as np
import pytest

import preprocess_oracle as P

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = ["up", "down", "portrait", "same", "odd"]


@pytest.fixture(scope="module")
def fx():
    return np.load(os.path.join(ROOT, "tests", "golden", "preprocess.npz"))


@pytest.mark.parametrize("name", CASES)
def test_oracle_chain_vs_reference_fixture(fx, name):
    frames = fx[f"{name}.frames"]
    got = P.video_transform(frames)
    assert got.dtype == np.float32 and got.shape == (frames.shape[0] * 3, 224, 224)
    assert np.array_equal(got[:, ::5, ::5], fx[f"{name}.out_sub"]), "float output differs from the reference chain (bit-exact expected)"
    s = np.array([got.astype(np.float64).sum(), (got.astype(np.float64) ** 2).sum()])
    assert np.allclose(s, fx[f"{name}.out_sum"], rtol=1e-13, atol=0)
    if f"{name}.crop_u8" in fx:
        H, W = frames.shape[1:3]
        nh, nw = P.tv_resized_size(H, W)
        top, left = P.center_crop_offsets(nh, nw)
        r = frames[0] if (nh, nw) == (H, W) else P.pil_resize_bicubic_u8(frames[0], nw, nh)
        assert np.array_equal(r[top:top + 224, left:left + 224], fx[f"{name}.crop_u8"])


def test_size_and_crop_rules():
    assert P.tv_resized_size(360, 640) == (224, 398) and P.tv_resized_size(640, 360) == (398, 224)
    assert P.tv_resized_size(224, 224) == (224, 224) and P.tv_resized_size(225, 301) == (224, 299)
    assert P.center_crop_offsets(224, 398) == (0, 87)
    assert P.center_crop_offsets(224, 299) == (0, 38)      # 37.5 -> 38 (half to even)
    assert P.center_crop_offsets(224, 297) == (0, 36)      # 36.5 -> 36
    assert P.center_crop_offsets(297, 224) == (36, 0)


def test_resize_vs_pillow_live():
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(3)
    for (h, w, oh, ow) in [(240, 320, 224, 298), (100, 150, 224, 336), (333, 251, 297, 224), (540, 960, 224, 398), (225, 225, 224, 224)]:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        ref = np.asarray(Image.fromarray(img, "RGB").resize((ow, oh), Image.BICUBIC))
        assert np.array_equal(P.pil_resize_bicubic_u8(img, ow, oh), ref), (h, w, oh, ow)


def test_workspace_geometry_host_side():
    """stllm_preprocess_workspace_bytes is pure host arithmetic (no GPU): coefficient tables + the uint8 image between the passes."""
    from stllm_amd import hip
    L = hip.lib()
    # 360x640 -> 224x398: ksize_h = 2*ceil(2*640/398)+1 = 9, ksize_v = 2*ceil(2*360/224)+1 = 9
    tabs = 2 * (2 * 224 + 224 * 9) * 4
    assert L.stllm_preprocess_workspace_bytes(16, 360, 640) == -(-tabs // 256) * 256 + 16 * 360 * 224 * 3
    # no resampling at all: identity tables (one tap per output)
    assert L.stllm_preprocess_workspace_bytes(1, 224, 224) == -(-(2 * (2 * 224 + 224) * 4) // 256) * 256 + 224 * 224 * 3
    assert L.stllm_preprocess_workspace_bytes(0, 360, 640) == -1
    assert L.stllm_preprocess_workspace_bytes(1, 224 * 40, 224 * 40) == -1          # > 31x down-scaling: more taps than the kernel holds
    assert L.stllm_preprocess_workspace_bytes(1, 100, 150) > 0                       # up-scaling is fine
