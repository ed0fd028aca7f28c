#!/usr/bin/env python3
"""Frame preprocessing (stllm_preprocess_frames): GPU time per 16-frame clip vs the reference's CPU chain (Pillow resize +
numpy crop / normalise on one core, as `Chat.transform` runs it).  GPU only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from stllm_amd import hip

for (T, H, W) in [(16, 360, 640), (16, 720, 1280), (16, 1080, 1920), (64, 360, 640)]:
    frames = torch.randint(0, 256, (T, H, W, 3), dtype=torch.uint8)
    d = frames.cuda()
    out = hip.preprocess_frames(d); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): hip.preprocess_frames(d, out=out)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / 20 * 1e3
    algo = T * H * W * 3 + T * 3 * 224 * 224 * 4            # input read once + fp32 output written once
    t0 = time.perf_counter(); h2d = frames.cuda(); torch.cuda.synchronize(); t_h2d = (time.perf_counter() - t0) * 1e6
    line = f"T={T:3d} {H}x{W}: GPU {us:8.1f} us ({algo / us / 1e3:6.1f} GB/s algorithmic), H2D of the uint8 frames {t_h2d:8.1f} us"
    try:
        from PIL import Image
        mean = np.array([0.48145466, 0.4578275, 0.40821073], np.float32); std = np.array([0.26862954, 0.26130258, 0.27577711], np.float32)
        t0 = time.perf_counter()
        for f in frames.numpy():
            im = Image.fromarray(f, "RGB")
            w, h = im.size; nl = int(224 * max(w, h) / min(w, h)); nw, nh = (224, nl) if w <= h else (nl, 224)
            im = im.resize((nw, nh), Image.BICUBIC)
            top, left = int(round((nh - 224) / 2.0)), int(round((nw - 224) / 2.0))
            a = np.asarray(im)[top:top + 224, left:left + 224].transpose(2, 0, 1).astype(np.float32) / 255
            a = (a - mean[:, None, None]) / std[:, None, None]
        line += f", CPU (Pillow, 1 core) {(time.perf_counter() - t0) * 1e6:9.1f} us"
    except ImportError:
        pass
    print(line)
