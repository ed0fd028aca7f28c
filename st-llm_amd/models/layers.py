"""Parameter-holder modules with the reference's parameter names; arithmetic goes to the HIP C ABI.

Parameters are fp32 masters (checkpoint-compatible: ``load_state_dict`` of reference checkpoints works,
SURVEY.md Appendix D); the kernels consume per-dtype packed copies built lazily by the owning model's
``pack()`` and cached until the parameters change.
"""
import torch
import torch.nn as nn

from .. import hip, pack, runtime


def _dev(device):
    return device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu")


def params_fingerprint(params):
    """Cheap identity of a set of master parameters for the packed-weight caches: in-place edits bump `_version`, `.to(device)`
    / re-assignment changes `data_ptr()`.  (Kernels that write through raw pointers — the optimizer — do neither: training calls
    repack() explicitly, stllm_amd.training.invalidate_packed.)"""
    v = a = 0
    for p in params:
        v += p._version
        a ^= p.data_ptr()
    return (v, a, runtime.gemm_split())   # the "bf16x3" mode shares torch.float32 with the plain verify mode but packs split weights


class ParamList:
    """The flat list of the master parameters a packed-weight cache depends on, built ONCE: walking `module.parameters()` costs ~1.5 us per
    parameter (435 us for Vicuna's 291, ~1.5 ms per step over the three stacks — a third of the host's enqueue time of a c2 step), reading
    `_version` / `data_ptr()` off a list 0.2 us.  The Parameter OBJECTS must stay the ones the module was built with: `load_state_dict`, `.to()`,
    in-place edits and optimizers keep them (and are seen through `_version` / `data_ptr()`); code that assigns a NEW nn.Parameter to a layer
    must call the model's repack(), which drops this list too."""

    def __init__(self, walk):
        self._walk, self._ps, self._calls = walk, None, 0

    def get(self):
        # every 64th call re-walks the module and compares identities (a layer.weight = nn.Parameter(...) without repack() would otherwise
        # leave the fingerprint — and the packed weights — on the replaced tensors for ever): ~7 us per call on average instead of 435 us
        self._calls += 1
        if self._ps is None or (self._calls & 63) == 0:
            ps = list(self._walk())
            if self._ps is None or len(ps) != len(self._ps) or any(a is not b for a, b in zip(ps, self._ps)):
                self._ps = ps
        return self._ps

    def reset(self):
        self._ps = None


class Linear(nn.Module):
    """nn.Linear-named holder.  ``forward`` is the generic (unfused) path: y = x @ W^T + b in the current
    compute dtype, fp32 in / fp32 out."""

    def __init__(self, in_features, out_features, bias=True, device=None):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features, device=_dev(device)), requires_grad=False)
        self.bias = nn.Parameter(torch.empty(out_features, device=_dev(device)), requires_grad=False) if bias else None
        self._packed = {}

    def packed(self, dtype):
        key = hip.torch_dtype(dtype)
        ver = (self.weight._version, self.weight.data_ptr(), runtime.gemm_split())
        hit = self._packed.get(key)
        if hit is None or hit[0] != ver:
            hit = (ver, pack.linear(self.weight, key), pack.f32(self.bias))
            self._packed = {key: hit}
        return hit[1], hit[2]

    def forward(self, x, act=hip.ACT_NONE):
        dt = runtime.compute_dtype()
        w, b = self.packed(dt)
        shp = x.shape
        x2 = x.reshape(-1, shp[-1]).float().contiguous()
        y = hip.gemm(hip.cast_rows(x2, dt), w, dtype=dt, bias=b, out_f32=True, act=act)
        return y.view(*shp[:-1], self.out_features)


class LayerNorm(nn.Module):
    """nn.LayerNorm-named holder; fp32 statistics (also the fp32 ``blip2.LayerNorm`` wrapper, blip2.py:103-109)."""

    def __init__(self, dim, eps=1e-5, device=None):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.empty(dim, device=_dev(device)), requires_grad=False)
        self.bias = nn.Parameter(torch.empty(dim, device=_dev(device)), requires_grad=False)

    def forward(self, x):
        shp = x.shape
        x2 = x.reshape(-1, shp[-1]).float().contiguous()
        _, y = hip.layernorm(x2, self.weight, self.bias, self.eps, dtype=torch.float32, want_t=False, want_f32=True)
        return y.view(shp)


class RMSNorm(nn.Module):
    def __init__(self, dim, eps=1e-6, device=None):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.empty(dim, device=_dev(device)), requires_grad=False)


class Embedding(nn.Module):
    """nn.Embedding-named holder; lookups are stllm_gather_rows on the fp32 table."""

    def __init__(self, num, dim, device=None):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(num, dim, device=_dev(device)), requires_grad=False)

    def forward(self, ids):
        ids = torch.as_tensor(ids)
        shp = ids.shape
        idx = hip.h2d((-(ids.reshape(-1).to(torch.int64)) - 1).to(torch.int32), self.weight.device)
        out = hip.gather_rows(self.weight, idx, src_b=self.weight)
        return out.view(*shp, self.weight.shape[1])


class Output(dict):
    """Tiny stand-in for HF ModelOutput: attribute + integer access (``outputs[0]``, ``outputs.logits``)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __getitem__(self, k):
        if isinstance(k, int):
            return [v for v in self.values() if v is not None][k]
        return dict.__getitem__(self, k)
