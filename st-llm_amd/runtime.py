"""Process-wide numerics mode of the HIP path.

  "bf16"   — MFMA bf16 inputs / fp32 accumulate, fp32 residual stream + norm/softmax statistics (default,
             BASELINE.json config 2)
  "fp16"   — same with fp16 MFMA inputs (the reference's own production dtype: demo.py:46, blip2.py:36)
  "fp32"   — exact-fp32 MFMA (v_mfma_f32_32x32x2_f32): the "verify" mode that meets the <=1e-2 logits bar
  "bf16x3" — the SPLIT verify mode (round 4): activations, norms, softmax and attention as in "fp32", every Linear as three bf16
             matrix-core products of split operands (x = hi + lo; stllm_hip.h STLLM_BF16X3): the fp32 accuracy class at ~3x the
             bf16 GEMM time instead of 16x.  compute_dtype() is torch.float32 in this mode; gemm_split() tells the packers to
             store the weights split.
  "mixed"  — (round 5) "bf16x3" everywhere EXCEPT the ViT blocks, which run in "fp16": the per-stage error ladder (profiles/r04_parity_ladder.log)
             shows the ViT is the stage whose fp16 error stays under the bar (9.5e-3 alone; the Q-Former's is 1.3e-2, Llama's 2.3e-2), and it is
             45 % of the GEMM time — the step costs 1.8x the bf16 step instead of 2.8x.  Measured at full size: c2 9.4e-3, c3 9.8e-3, c4 1.02e-2 — AT the 1e-2 bar, not
             safely under it: "bf16x3" stays the verify mode.  The visual encoder enters `vit_scope()` for its forward (the BT-Adapter backbone does not: it stays bf16x3).
"""
import contextlib

import torch

from .hip import torch_dtype

_state = {"dtype": torch.bfloat16, "split": False, "vit": None}
SPLIT_NAMES = ("bf16x3", "split")
MIXED_VIT = "fp16"   # the ViT's mode inside "mixed"


def set_compute_dtype(d):
    if isinstance(d, str) and d == "mixed":
        _state.update(dtype=torch.float32, split=True, vit=MIXED_VIT)
    elif isinstance(d, str) and d in SPLIT_NAMES:
        _state.update(dtype=torch.float32, split=True, vit=None)
    else:
        _state.update(dtype=torch_dtype(d), split=False, vit=None)


@contextlib.contextmanager
def vit_scope():
    """the numerics mode of the visual encoder's blocks: the process-wide mode, or — in "mixed" — the ViT's own"""
    v = _state["vit"]
    if v is None:
        yield
        return
    old = dict(_state)
    set_compute_dtype(v)
    try:
        yield
    finally:
        _state.update(old)


def compute_dtype():
    return _state["dtype"]


def gemm_split():
    """True in the "bf16x3" mode: pack.* store GEMM weights as split bf16 [N, 3 K] and hip.gemm runs them as STLLM_BF16X3"""
    return _state["split"]


def mode_name():
    if _state["split"]:
        return "mixed" if _state["vit"] else "bf16x3"
    return {torch.bfloat16: "bf16", torch.float16: "fp16", torch.float32: "fp32"}[_state["dtype"]]


@contextlib.contextmanager
def use_dtype(d):
    old = dict(_state)
    set_compute_dtype(d)
    try:
        yield
    finally:
        _state.update(old)
