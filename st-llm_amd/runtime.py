"""Process-wide numerics mode of the HIP path.

  "bf16" — MFMA bf16 inputs / fp32 accumulate, fp32 residual stream + norm/softmax statistics (default,
           BASELINE.json config 2)
  "fp16" — same with fp16 MFMA inputs (the reference's own production dtype: demo.py:46, blip2.py:36)
  "fp32" — exact-fp32 MFMA (v_mfma_f32_32x32x2_f32): the "verify" mode that meets the <=1e-2 logits bar
"""
import contextlib

import torch

from .hip import torch_dtype

_state = {"dtype": torch.bfloat16}


def set_compute_dtype(d):
    _state["dtype"] = torch_dtype(d)


def compute_dtype():
    return _state["dtype"]


@contextlib.contextmanager
def use_dtype(d):
    old = _state["dtype"]
    set_compute_dtype(d)
    try:
        yield
    finally:
        _state["dtype"] = old
