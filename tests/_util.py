"""Shared helpers for the test-suite (test infrastructure)."""
import os

import numpy as np
import torch

from stllm_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SEED = 0


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def T(name, shape, std=1.0, seed=SEED, device="cpu", dtype=torch.float32):
    return synth.normal_(torch.empty(shape, device=device, dtype=dtype), name, seed, std)


def sd_from(shapes, device="cpu", dtype=torch.float32, seed=SEED):
    return synth.state_dict_from_shapes(shapes, seed, device, dtype)


def sub(x, *strides):
    idx = tuple(slice(None, None, s) for s in strides)
    return x.detach().float().cpu()[idx].numpy()


def stats(x):
    x = x.detach().double().cpu()
    return np.array([x.norm().item(), x.abs().max().item(), x.mean().item()])


def unragged(a):
    return [[int(v) for v in row if v >= 0] for row in a]


def max_abs(a, b):
    return float(np.max(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64))))


def assert_close(got, want, atol, rtol=0.0, what=""):
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, f"{what}: shape {got.shape} vs {want.shape}"
    err = np.abs(got - want)
    tol = atol + rtol * np.abs(want)
    bad = err > tol
    assert not bad.any(), (f"{what}: {bad.sum()}/{bad.size} elements out of tolerance; max err {err.max():.3e} "
                           f"(atol {atol}, rtol {rtol}); ref abs-max {np.abs(want).max():.3e}")
