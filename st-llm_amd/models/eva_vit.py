"""EVA-CLIP-g ViT behind the reference's surface (stllm/models/eva_vit.py), running on the HIP C ABI.

Module / parameter names follow the reference (eva_vit.py:76-82,115,157-165,196,263-265) so its
checkpoints load unchanged.  Per block (eva_vit.py:173-180, gamma_1/2 None for eva_clip_g):

    LN(eps 1e-6) -> [QKV GEMM + (q_bias,0,v_bias)] -> fused attention (16 heads x 88, scale 88^-0.5 folded
    into the softmax) -> [proj GEMM + bias + fp32 residual] -> LN -> [fc1 GEMM + bias + exact-erf GELU]
    -> [fc2 GEMM + bias + fp32 residual]

The residual stream is fp32; GEMM/attention operands are the compute dtype (runtime.compute_dtype()).
"""

import math

import torch
import torch.nn as nn

from .. import hip, pack, runtime
from .layers import LayerNorm, Linear, ParamList, _dev, params_fingerprint


class Attention(nn.Module):
    def __init__(self, dim, num_heads, device=None):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = Linear(dim, dim * 3, bias=False, device=device)
        self.q_bias = nn.Parameter(torch.empty(dim, device=_dev(device)), requires_grad=False)
        self.v_bias = nn.Parameter(torch.empty(dim, device=_dev(device)), requires_grad=False)
        self.proj = Linear(dim, dim, device=device)


class Mlp(nn.Module):
    def __init__(self, dim, hidden, device=None):
        super().__init__()
        self.fc1 = Linear(dim, hidden, device=device)
        self.fc2 = Linear(hidden, dim, device=device)


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4.3637, eps=1e-6, device=None):
        super().__init__()
        self.norm1 = LayerNorm(dim, eps, device)
        self.attn = Attention(dim, num_heads, device)
        self.norm2 = LayerNorm(dim, eps, device)
        self.mlp = Mlp(dim, int(dim * mlp_ratio), device)

    def pack(self, dtype):
        a, m = self.attn, self.mlp
        return dict(n1w=self.norm1.weight, n1b=self.norm1.bias, e1=self.norm1.eps,
                    wqkv=pack.linear(a.qkv.weight, dtype), bqkv=pack.vit_qkv_bias(a.q_bias, a.v_bias),
                    wproj=pack.linear(a.proj.weight, dtype), bproj=pack.f32(a.proj.bias),
                    n2w=self.norm2.weight, n2b=self.norm2.bias, e2=self.norm2.eps,
                    wfc1=pack.linear(m.fc1.weight, dtype), bfc1=pack.f32(m.fc1.bias),
                    wfc2=pack.linear(m.fc2.weight, dtype), bfc2=pack.f32(m.fc2.bias))


def block_forward(x, pk, n_seq, seq_len, num_heads, dt):
    """One pre-LN transformer block on the flat fp32 stream x [n_seq*seq_len, dim], in place."""
    dim = x.shape[1]
    hd = dim // num_heads
    h, _ = hip.layernorm(x, pk["n1w"], pk["n1b"], pk["e1"], dtype=dt)
    qkv = hip.gemm(h, pk["wqkv"], dtype=dt, bias=pk["bqkv"])
    a = hip.attention(qkv[:, :dim], qkv[:, dim:2 * dim], qkv[:, 2 * dim:], B=n_seq, H=num_heads, Sq=seq_len,
                      Skv=seq_len, D=hd, scale=hd ** -0.5)
    hip.gemm(a, pk["wproj"], dtype=dt, epilogue=hip.EPI_RESID, bias=pk["bproj"], resid=x)
    h, _ = hip.layernorm(x, pk["n2w"], pk["n2b"], pk["e2"], dtype=dt)
    g = hip.gemm(h, pk["wfc1"], dtype=dt, bias=pk["bfc1"], act=hip.ACT_GELU)
    hip.gemm(g, pk["wfc2"], dtype=dt, epilogue=hip.EPI_RESID, bias=pk["bfc2"], resid=x)
    return x


class PatchEmbed(nn.Module):
    def __init__(self, img_size=224, patch_size=14, in_chans=3, embed_dim=1408, device=None):
        super().__init__()
        self.img_size = (img_size, img_size)
        self.patch_size = (patch_size, patch_size)
        self.num_patches = (img_size // patch_size) ** 2
        self.patch_shape = (img_size // patch_size, img_size // patch_size)
        self.proj = nn.Module()
        self.proj.weight = nn.Parameter(torch.empty(embed_dim, in_chans, patch_size, patch_size, device=_dev(device)),
                                        requires_grad=False)
        self.proj.bias = nn.Parameter(torch.empty(embed_dim, device=_dev(device)), requires_grad=False)


class VisionTransformer(nn.Module):
    """eva_vit.VisionTransformer (eva_vit.py:246-370) for the eva_clip_g configuration only
    (img 224, patch 14, dim 1408, 16 heads, abs pos-embed, no rel-pos bias, no LayerScale, no final norm)."""

    def __init__(self, img_size=224, patch_size=14, embed_dim=1408, depth=39, num_heads=16, mlp_ratio=4.3637,
                 eps=1e-6, device=None, **unused):
        super().__init__()
        if img_size != 224 or patch_size != 14 or embed_dim != 1408:
            raise NotImplementedError("HIP patch-embed kernel is specialised for EVA-CLIP-g (224/14/1408)")
        self.image_size = img_size
        self.num_features = self.embed_dim = embed_dim
        self.num_heads = num_heads
        self.patch_embed = PatchEmbed(img_size, patch_size, 3, embed_dim, device)
        self.cls_token = nn.Parameter(torch.empty(1, 1, embed_dim, device=_dev(device)), requires_grad=False)
        self.pos_embed = nn.Parameter(torch.empty(1, self.patch_embed.num_patches + 1, embed_dim, device=_dev(device)),
                                      requires_grad=False)
        self.blocks = nn.ModuleList([Block(embed_dim, num_heads, mlp_ratio, eps, device) for _ in range(depth)])
        self._packed = {}
        self._plist = ParamList(lambda: [self.patch_embed.proj.weight, self.patch_embed.proj.bias, self.pos_embed, self.cls_token, *self.blocks.parameters()])

    # -- packing ---------------------------------------------------------------------------------
    def pack(self, dtype=None):
        dt = hip.torch_dtype(dtype) if dtype is not None else runtime.compute_dtype()
        fp = params_fingerprint(self._plist.get())
        hit = self._packed.get(dt)
        if hit is None or hit[0] != fp:
            hit = (fp, dict(
                wpatch=pack.patch_weight(self.patch_embed.proj.weight, dt), bpatch=pack.f32(self.patch_embed.proj.bias),
                pos=self.pos_embed.detach().view(-1, self.embed_dim).float().contiguous(),
                cls=self.cls_token.detach().view(-1).float().contiguous(),
                blocks=[b.pack(dt) for b in self.blocks]))
            self._packed = {dt: hit}
        return hit[1]

    def repack(self):
        self._packed = {}
        self._plist.reset()

    def _load_from_state_dict(self, state_dict, prefix, *a, **k):
        self._packed = {}
        self._plist.reset()
        interpolate_pos_embed(self, state_dict, prefix + "pos_embed")   # eva_vit.py:435: before the tensors are copied in
        return super()._load_from_state_dict(state_dict, prefix, *a, **k)

    # -- forward ---------------------------------------------------------------------------------
    def embed_flat(self, x, pk, dt, out=None):
        """patch-embed implicit GEMM + CLS + pos_embed -> flat fp32 stream [N*257, 1408]"""
        N, C, H, W = x.shape
        assert H == self.patch_embed.img_size[0] and W == self.patch_embed.img_size[1], \
            f"Input image size ({H}*{W}) doesn't match model ({self.patch_embed.img_size[0]}*{self.patch_embed.img_size[1]})."
        x = x.float().contiguous()
        if out is None:
            out = torch.empty((N * 257, self.embed_dim), device=x.device, dtype=torch.float32)
        hip.gemm(None, pk["wpatch"], dtype=dt, epilogue=hip.EPI_PATCH, bias=pk["bpatch"], out=out, frames=x,
                 pos_embed=pk["pos"], n_frames=N)
        hip.vit_cls_rows(pk["cls"], pk["pos"], out, N)
        return out

    frame_streams = 1   # >1: run groups of frames on concurrent HIP streams (measured slower on MI355X: 31.7 ms vs 33.1 / 38.2 ms at 2 / 4 streams in round 1;
                        # 23.12 vs 25.75-25.92 ms at 2 streams on the round-4 kernels, `bench.py --vit-streams 2`: one workgroup per CU owns the whole LDS, so the
                        # second stream's kernels only get the CUs the first leaves idle and every weight matrix is streamed twice)

    def _features_group(self, x, pk, dt, out):
        n = x.shape[0]
        self.embed_flat(x, pk, dt, out=out)
        from . import llama
        if not llama.STACK_ENTRY:
            for bp in pk["blocks"]:
                block_forward(out, bp, n, 257, self.num_heads, dt)
            return
        if "cblocks" not in pk:   # the C-side table of the packed blocks lives next to them (dropped with them on repack)
            pk["cblocks"] = hip.vit_block_array(pk["blocks"])
        # eva_vit.py:336-339: the whole block loop is ONE call into the C ABI (stllm_vit_blocks; == block_forward per block, bit for bit)
        hip.vit_blocks(out, pk["blocks"], pk["cblocks"], n_seq=n, seq_len=257, num_heads=self.num_heads, dtype=dt)

    def forward_features_flat(self, x):
        """[N,3,224,224] -> flat fp32 stream [N*257, 1408].  The N frames are split into `frame_streams` groups that run on
        separate HIP streams: every launch is a persistent grid that ends with a partially filled last round of tiles, and
        kernels of one stream serialise — a second independent stream fills those tails (same kernels, same numerics:
        rows of different frames never interact)."""
        with runtime.vit_scope():     # the "mixed" verify mode runs the ViT in its own numerics mode (runtime.py)
            return self._forward_features_flat(x)

    def _forward_features_flat(self, x):
        dt = runtime.compute_dtype()
        pk = self.pack(dt)
        N = x.shape[0]
        x = x.float().contiguous()
        out = torch.empty((N * 257, self.embed_dim), device=x.device, dtype=torch.float32)
        ns = min(self.frame_streams, N) if x.is_cuda else 1
        if ns <= 1:
            self._features_group(x, pk, dt, out)
            return out
        if getattr(self, "_streams", None) is None or len(self._streams) != ns:
            self._streams = [torch.cuda.Stream(device=x.device) for _ in range(ns)]
        cur = torch.cuda.current_stream(x.device)
        bounds = [round(i * N / ns) for i in range(ns + 1)]
        for i, st in enumerate(self._streams):
            a, b = bounds[i], bounds[i + 1]
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                self._features_group(x[a:b], pk, dt, out[a * 257: b * 257])
        for st in self._streams:
            cur.wait_stream(st)
        return out

    def forward_features(self, x):
        return self.forward_features_flat(x).view(x.shape[0], 257, self.embed_dim)

    def forward(self, x):
        return self.forward_features(x)

    def get_num_layers(self):
        return len(self.blocks)


def _bicubic_matrix(n_out, n_in):
    """[n_out, n_in] weights of torch's bicubic resampling along one axis (align_corners=False, cubic-convolution A = -0.75,
    border taps clamped): out = W @ in.  The 2-D resample is separable: W @ P @ W^T."""
    A = -0.75
    scale = n_in / n_out
    W = torch.zeros(n_out, n_in, dtype=torch.float64)
    for o in range(n_out):
        src = (o + 0.5) * scale - 0.5
        x0 = math.floor(src)
        t = src - x0
        coef = (((A * (t + 1) - 5 * A) * (t + 1) + 8 * A) * (t + 1) - 4 * A,
                ((A + 2) * t - (A + 3)) * t * t + 1,
                ((A + 2) * (1 - t) - (A + 3)) * (1 - t) * (1 - t) + 1,
                ((A * (2 - t) - 5 * A) * (2 - t) + 8 * A) * (2 - t) - 4 * A)
        for k, c in enumerate(coef):
            W[o, min(max(x0 - 1 + k, 0), n_in - 1)] += c
    return W


def interpolate_pos_embed(model, checkpoint_model, key="pos_embed"):
    """eva_vit.py:373-394: a checkpoint whose position table belongs to another input resolution is resampled (bicubic, class
    token kept) to this model's patch grid, in place in `checkpoint_model` — host-side, once, at load time."""
    if key not in checkpoint_model:
        return
    ck = checkpoint_model[key].float()
    D = ck.shape[-1]
    num_patches = model.patch_embed.num_patches
    extra = model.pos_embed.shape[-2] - num_patches
    orig = int((ck.shape[-2] - extra) ** 0.5)
    new = int(num_patches ** 0.5)
    if orig == new:
        return
    W = _bicubic_matrix(new, orig)
    grid = ck[:, extra:].reshape(-1, orig, orig, D).double()
    out = torch.einsum("oi,bijd,pj->bopd", W, grid, W).reshape(ck.shape[0], new * new, D).float()
    checkpoint_model[key] = torch.cat((ck[:, :extra], out), dim=1)


def create_eva_vit_g(img_size=224, drop_path_rate=0.4, use_checkpoint=False, precision="fp16", depth=39, device=None):
    """eva_vit.create_eva_vit_g (eva_vit.py:415-443) without the checkpoint download (no network): random
    parameters are left uninitialised for the caller to fill (synth / load_state_dict)."""
    return VisionTransformer(img_size=img_size, patch_size=14, embed_dim=1408, depth=depth, num_heads=1408 // 88,
                             mlp_ratio=4.3637, eps=1e-6, device=device)
