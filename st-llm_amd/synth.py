"""Deterministic synthetic weights, bit-identical on CPU and GPU.

There are no checkpoints offline (BASELINE.json: random-init), and full-width tensors are far too
large to commit as fixtures, so every parity fixture names its weights by (tensor name, seed) and
both sides regenerate them with this counter-based generator.  Only exact integer ops and one
IEEE multiply are used, so CPU (fixture generation / oracle) and GPU (product path) produce the
same bits.

Distribution: Irwin–Hall(8) of hashed bytes, centred and scaled to the requested std — a close
stand-in for the reference's N(0, 0.02) initialisers (eva_vit.py:308-315, Qformer.py:664-674,
HF Llama initializer_range) with support ±4.9 sigma.
"""
import zlib

import torch

_M32 = 0xFFFFFFFF
_CHUNK = 1 << 24
_CACHE = None          # optional {(name, seed, numel, std, mean): fp32 CPU tensor}; test-suites regenerate the same tensors often
_CACHE_BYTES = 0
_CACHE_LIMIT = 12 << 30


def enable_cache(limit_bytes=12 << 30):
    """Memoise generated CPU tensors (generation is ~20 M elements/s on the host)."""
    global _CACHE, _CACHE_LIMIT
    if _CACHE is None:
        _CACHE = {}
    _CACHE_LIMIT = limit_bytes


def _hash32(x: torch.Tensor) -> torch.Tensor:
    # values stay < 2^32 and the multiplier < 2^27, so int64 never overflows
    x = (((x >> 16) ^ x) * 0x45D9F3B) & _M32
    x = (((x >> 16) ^ x) * 0x45D9F3B) & _M32
    return (x >> 16) ^ x


def _bytesum(u: torch.Tensor) -> torch.Tensor:
    return (u & 0xFF) + ((u >> 8) & 0xFF) + ((u >> 16) & 0xFF) + ((u >> 24) & 0xFF)


def name_key(name: str, seed: int) -> int:
    return (zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & _M32


def _host_fill(flat, n, key, scale, mean) -> bool:
    """CPU fp32 tensors: the same stream from the C loop in libstllm_hip.so (stllm_synth_normal_f32, ~50x the torch recipe on a host
    core; bit-identical — tests/test_host_cpu.py).  False when the library is not built: the torch recipe below takes over."""
    try:
        import ctypes
        from . import hip
        import numpy as np
        # scale / mean as torch would apply them: the python floats rounded to fp32
        rc = hip.lib().stllm_synth_normal_f32(ctypes.c_void_p(flat.data_ptr()), n, 0, key, float(np.float32(scale)), float(np.float32(mean)))
        return rc == 0
    except Exception:
        return False


def normal_(t: torch.Tensor, name: str, seed: int = 0, std: float = 0.02, mean: float = 0.0) -> torch.Tensor:
    """Fill `t` in place (any float dtype, any device) from (name, seed)."""
    global _CACHE_BYTES
    key = name_key(name, seed)
    flat = t.view(-1)
    n = flat.numel()
    ck = (name, seed, n, float(std), float(mean))
    if _CACHE is not None and t.device.type == "cpu":
        hit = _CACHE.get(ck)
        if hit is not None:
            flat.copy_(hit)
            return t
    scale = std / 209.02152999054  # sqrt(8 * (256^2 - 1) / 12)
    if t.device.type == "cpu" and t.dtype == torch.float32 and flat.is_contiguous() and _host_fill(flat, n, key, scale, mean):
        if _CACHE is not None and _CACHE_BYTES + 4 * n <= _CACHE_LIMIT:
            _CACHE[ck] = flat.clone()
            _CACHE_BYTES += 4 * n
        return t
    for s in range(0, n, _CHUNK):
        e = min(n, s + _CHUNK)
        idx = torch.arange(s, e, dtype=torch.int64, device=t.device)
        a = _hash32((idx ^ key) & _M32)
        b = _hash32((a + 0x68E31DA4 + idx) & _M32)
        z = (_bytesum(a) + _bytesum(b)).to(torch.float32) - 1020.0
        v = z * scale
        if mean != 0.0:
            v = v + mean
        flat[s:e] = v.to(t.dtype)
    if _CACHE is not None and t.device.type == "cpu" and t.dtype == torch.float32 and _CACHE_BYTES + 4 * n <= _CACHE_LIMIT:
        _CACHE[ck] = flat.clone()
        _CACHE_BYTES += 4 * n
    return t


def _rule(name: str):
    """(mean, std) by reference parameter name (Appendix D of SURVEY.md).  Biases and norm
    parameters are deliberately non-trivial so that parity tests exercise them."""
    leaf = name.rsplit(".", 2)[-2:] if "." in name else [name]
    last = leaf[-1]
    parent = leaf[0] if len(leaf) == 2 else ""
    is_norm = ("norm" in parent.lower()) or parent in ("ln_vision",)
    if last == "weight" and is_norm:
        return 1.0, 0.1
    if last == "bias" and is_norm:
        return 0.0, 0.05
    if last in ("bias", "q_bias", "v_bias"):
        return 0.0, 0.02
    return 0.0, 0.02


def fill_named_(named, seed: int = 0, prefix: str = "", overrides=None):
    """Fill an iterable of (name, tensor) in place.  `overrides`: {substring: (mean, std)}."""
    with torch.no_grad():
        for name, t in named:
            if not torch.is_floating_point(t):
                continue
            mean, std = _rule(prefix + name)
            for k, ms in (overrides or {}).items():
                if k in prefix + name:
                    mean, std = ms
            normal_(t.data if hasattr(t, "data") else t, prefix + name, seed, std, mean)


def fill_module_(module: torch.nn.Module, seed: int = 0, prefix: str = "", overrides=None):
    fill_named_(module.named_parameters(), seed, prefix, overrides)
    return module


def state_dict_from_shapes(shapes: dict, seed: int = 0, device="cpu", dtype=torch.float32, overrides=None):
    sd = {k: torch.empty(v, device=device, dtype=dtype) for k, v in shapes.items()}
    fill_named_(sd.items(), seed, "", overrides)
    return sd
