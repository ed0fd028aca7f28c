"""models/utils.py of the reference: RandomMaskingGenerator (models/utils.py:4-16) — numpy global RNG on the
host, True = dropped; kept identical so a seeded numpy stream reproduces the reference's masks."""
import numpy as np
import torch


def RandomMaskingGenerator(num_patches, mask_ratio, batch, device="cuda"):
    num_mask = int(mask_ratio * num_patches)
    rows = []
    for _ in range(batch):
        m = np.hstack([np.zeros(num_patches - num_mask), np.ones(num_mask)])
        np.random.shuffle(m)
        rows.append(m)
    return torch.from_numpy(np.array(rows)).to(torch.bool)  # stays on the host: it only builds gather indices
