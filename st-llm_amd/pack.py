"""Weight packers: reference-named fp32 parameters -> the layouts the HIP kernels consume.

All packers are pure index permutations / concatenations / casts (done once, on the device the
parameter lives on); none changes the mathematical function:

* ``vit_qkv_bias``   — eva_vit.py:120-124: bias = cat(q_bias, zeros, v_bias).
* ``patch_weight``   — Conv2d weight [1408,3,14,14] -> GEMM weight [1408, 588] zero-padded in K to a
                       whole number of 128-byte LDS panels (640 for 16-bit, 608 for fp32).
* ``llama_qkv``      — q/k/v_proj fused to one [3*4096, 4096] weight.  Inside every 128-wide q/k head
                       the rows are re-ordered to [0:32 | 64:96 | 32:64 | 96:128] so that the rotate-half
                       partners (i, i+64) land in the two 32-column halves of one 64-column wave tile
                       and RoPE is an in-lane epilogue.  q and k get the SAME permutation, so q.k is
                       unchanged (spec of the rotation: modeling_llama_mem.py:113-127).
* ``llama_gate_up``  — gate/up_proj interleaved in groups of 32 rows ([32 gate | 32 up] per 64-column
                       wave tile) for the fused SiLU(gate)*up epilogue (modeling_llama_mem.py:143-144).
* ``bert_qkv`` / ``bert_kv`` — Q-Former query/key/value fused (Qformer.py:127-133).
"""
import os

import torch

from . import runtime
from .hip import torch_dtype


def split3_weight(w):
    """fp32 [N, K] -> bf16 [N, 3 K] = (hi | lo | hi) along K, hi = bf16(w), lo = bf16(w - hi): the weight operand of the "bf16x3" GEMMs
    (stllm_hip.h STLLM_BF16X3; the same rounding as stllm_split3_rows side 1)."""
    wf = w.detach().float()
    hi = wf.to(torch.bfloat16)
    lo = (wf - hi.float()).to(torch.bfloat16)
    return torch.cat((hi, lo, hi), dim=1).contiguous()


def _cast(w, dtype):
    td = torch_dtype(dtype)
    if td == torch.float32 and runtime.gemm_split() and w.dim() == 2 and w.shape[1] % 64 == 0:
        return split3_weight(w)
    return w.detach().to(td).contiguous()


def linear(w, dtype):
    return _cast(w, dtype)


def f32(b):
    return None if b is None else b.detach().float().contiguous()


def vit_qkv_bias(q_bias, v_bias):
    return torch.cat((q_bias.detach().float(), torch.zeros_like(v_bias, dtype=torch.float32), v_bias.detach().float())).contiguous()


def patch_k_padded(dtype):
    eb = 4 if torch_dtype(dtype) == torch.float32 else 2
    panel = 128 // eb
    return ((588 + panel - 1) // panel) * panel


def patch_weight(w, dtype):
    n = w.shape[0]
    out = torch.zeros((n, patch_k_padded(dtype)), device=w.device, dtype=torch_dtype(dtype))
    out[:, :588] = w.detach().reshape(n, 588).to(out.dtype)
    return out


def rope_head_perm(n_heads, head_dim=128, device="cpu"):
    assert head_dim == 128
    base = torch.cat([torch.arange(0, 32), torch.arange(64, 96), torch.arange(32, 64), torch.arange(96, 128)])
    return (torch.arange(n_heads)[:, None] * head_dim + base[None, :]).reshape(-1).to(device)


def llama_qkv(wq, wk, wv, dtype, n_heads=32):
    perm = rope_head_perm(n_heads, wq.shape[0] // n_heads, wq.device)
    return torch.cat((_cast(wq, dtype)[perm], _cast(wk, dtype)[perm], _cast(wv, dtype)), dim=0).contiguous()


def llama_gate_up(wg, wu, dtype):
    n, k = wg.shape
    assert n % 32 == 0
    g = _cast(wg, dtype)
    u = _cast(wu, dtype)
    k = g.shape[1]   # 3 K in the split mode
    return torch.stack((g.view(n // 32, 32, k), u.view(n // 32, 32, k)), dim=1).reshape(2 * n, k).contiguous()


def frag32(w):
    """fragment-major copy of a packed 16-bit weight [N, K] (N % 32 == 0, K % 16 == 0) for the W-direct GEMM (csrc/gemm_wd.inc, stllm_gemm_args.w_frag):
    [N / 32][K / 16][64 lanes][8 elements], lane l of fragment (nb, ks) = w[32 nb + (l & 31), 16 ks + 8 (l >> 5) : + 8] — the operand of one
    v_mfma_f32_32x32x16 as one contiguous KiB."""
    n, k = w.shape
    assert n % 32 == 0 and k % 16 == 0 and w.element_size() == 2
    #            nb      row      ks     half   e            nb ks half row e  -> lane = half * 32 + row
    return w.reshape(n // 32, 32, k // 16, 2, 8).permute(0, 2, 3, 1, 4).contiguous().reshape(-1)


def frag32_or_none(w):
    """frag32(w) where the W-direct kernel can use it: a 16-bit weight on the GPU with N % 256 == 0 and K % 256 == 0 (un-padded, contiguous rows);
    STLLM_WD_FRAG=0 in the environment: never (the prefill runs on the other kernels, no second copy of the weights)."""
    if os.environ.get("STLLM_WD_FRAG", "1") == "0" or not w.is_cuda or w.dim() != 2 or w.element_size() != 2 or not w.is_contiguous():
        return None
    n, k = w.shape
    if n % 256 or k % 256 or n * k * 2 >= (1 << 32):
        return None
    return frag32(w)


def bert_qkv(q, k, v, dtype):
    w = torch.cat((_cast(q.weight, dtype), _cast(k.weight, dtype), _cast(v.weight, dtype)), dim=0).contiguous()
    b = torch.cat((f32(q.bias), f32(k.bias), f32(v.bias))).contiguous()
    return w, b


def bert_kv(k, v, dtype):
    w = torch.cat((_cast(k.weight, dtype), _cast(v.weight, dtype)), dim=0).contiguous()
    b = torch.cat((f32(k.bias), f32(v.bias))).contiguous()
    return w, b


def pad_rows(w, mult=128):
    """zero-pad the out_features dim to a multiple of `mult` (lm_head with a 32001-token vocab)."""
    n = w.shape[0]
    n_pad = ((n + mult - 1) // mult) * mult
    if n_pad == n:
        return w
    out = torch.zeros((n_pad,) + tuple(w.shape[1:]), device=w.device, dtype=w.dtype)
    out[:n] = w
    return out


def rope_tables(S, head_dim=128, base=10000.0, device="cpu"):
    """cos/sin [S, head_dim/2] fp32 — LlamaRotaryEmbedding (modeling_llama_mem.py:81-110); computed on the
    host in fp32 exactly as the reference does, then copied to the device."""
    inv = 1.0 / (base ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    f = torch.outer(torch.arange(S, dtype=torch.float32), inv)
    return f.cos().contiguous().to(device), f.sin().contiguous().to(device)
