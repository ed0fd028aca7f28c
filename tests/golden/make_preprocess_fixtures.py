"""Golden vectors for the frame preprocessing — runs ONLY in the build container.

    python tests/golden/make_preprocess_fixtures.py      -> tests/golden/preprocess.npz

Runs the reference's OWN transform chain (`stllm/test/video_transforms.py`: GroupScale, GroupCenterCrop, Stack,
ToTorchFormatTensor, GroupNormalize, composed exactly as `Chat.__init__` does at stllm/conversation/conversation.py:190-198)
on synthetic uint8 frames.  torchvision (== 0.15.1 in the reference's requirement.txt) is not installed here, so the two
torchvision workers those classes delegate to are supplied by a stub that restates torchvision's PIL code path
(`functional._compute_resized_output_size` + `img.resize(size[::-1], BICUBIC)`; `CenterCrop` offsets
`int(round((h - th) / 2.0))`); the resampling itself is done by the real Pillow (12.2.0 here).

A fixture is data: input frames + expected outputs (the float output sub-sampled with stride 5 plus full checksums, and
the full uint8 resized+cropped image of the first frame of three cases).  No reference source is stored.
"""
import os
import sys
import types

import numpy as np
import torch
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def _install_torchvision_stub():
    tv = types.ModuleType("torchvision")
    tr = types.ModuleType("torchvision.transforms")

    class Resize:   # torchvision 0.15.1 transforms.Resize on PIL images, size = int
        def __init__(self, size, interpolation=Image.BILINEAR):
            self.size, self.interpolation = size, interpolation

        def __call__(self, img):
            w, h = img.size
            short, long = (w, h) if w <= h else (h, w)
            new_short, new_long = self.size, int(self.size * long / short)
            new_w, new_h = (new_short, new_long) if w <= h else (new_long, new_short)
            if (w, h) == (new_w, new_h):
                return img
            return img.resize((new_w, new_h), self.interpolation)

    class CenterCrop:   # torchvision 0.15.1 transforms.CenterCrop (inputs here are never smaller than the crop)
        def __init__(self, size):
            self.size = (int(size), int(size))

        def __call__(self, img):
            w, h = img.size
            th, tw = self.size
            top, left = int(round((h - th) / 2.0)), int(round((w - tw) / 2.0))
            return img.crop((left, top, left + tw, top + th))

    class Compose:
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x

    tr.Resize, tr.CenterCrop, tr.Compose = Resize, CenterCrop, Compose
    tv.transforms = tr
    sys.modules["torchvision"], sys.modules["torchvision.transforms"] = tv, tr
    return tr


def synth_frames(T, H, W, seed):
    """deterministic frames with smooth gradients, sinusoids, hard edges and a band of noise"""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:H, 0:W].astype(np.float64)
    out = np.empty((T, H, W, 3), np.uint8)
    for t in range(T):
        for c in range(3):
            f = 127.5 + 60 * np.sin(x * (0.05 + 0.02 * c) + t) * np.cos(y * (0.07 - 0.01 * c)) + 40 * ((x + 2 * y + 13 * t) % 64 < 20)
            f += 25 * (((x // 9 + y // 7 + c) % 2) - 0.5)
            f[H // 3: H // 3 + 5, :] = 255 * (c == t % 3)
            f[:, W // 2: W // 2 + 3] = 0
            f[: H // 5] += rng.integers(-6, 7, (H // 5, W))   # noise on a band only: keeps the fixture compressible
            out[t, :, :, c] = np.clip(np.rint(f), 0, 255).astype(np.uint8)
    return out


def main():
    T = _install_torchvision_stub()
    sys.path.insert(0, REF)
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_video_transforms", os.path.join(REF, "stllm", "test", "video_transforms.py"))
    vt = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(vt)
    input_mean = [0.48145466, 0.4578275, 0.40821073]
    input_std = [0.26862954, 0.26130258, 0.27577711]
    transform = T.Compose([vt.GroupScale(int(224), interpolation=Image.BICUBIC), vt.GroupCenterCrop(224), vt.Stack(),
                           vt.ToTorchFormatTensor(), vt.GroupNormalize(input_mean, input_std)])
    crop_only = T.Compose([vt.GroupScale(int(224), interpolation=Image.BICUBIC), vt.GroupCenterCrop(224)])
    cases = {"up": (2, 120, 160), "down": (1, 180, 320), "portrait": (1, 250, 180), "same": (1, 224, 224), "odd": (1, 225, 301)}
    fx = {}
    for i, (name, (t, h, w)) in enumerate(cases.items()):
        frames = synth_frames(t, h, w, 100 + i)
        imgs = [Image.fromarray(f, "RGB") for f in frames]
        out = transform(imgs)                                   # float32 [t*3, 224, 224]
        assert out.shape == (t * 3, 224, 224) and out.dtype == torch.float32
        o = out.numpy()
        fx[f"{name}.frames"] = frames
        fx[f"{name}.out_sub"] = o[:, ::5, ::5].copy()
        fx[f"{name}.out_sum"] = np.array([o.astype(np.float64).sum(), (o.astype(np.float64) ** 2).sum()])
        if name in ("down", "portrait", "odd"):
            fx[f"{name}.crop_u8"] = np.asarray(crop_only(imgs)[0]).copy()
        print(name, frames.shape, "->", o.shape, "abs-max", float(np.abs(o).max()))
    np.savez_compressed(os.path.join(HERE, "preprocess.npz"), **fx)
    print("wrote", os.path.join(HERE, "preprocess.npz"), os.path.getsize(os.path.join(HERE, "preprocess.npz")), "bytes")


if __name__ == "__main__":
    main()
