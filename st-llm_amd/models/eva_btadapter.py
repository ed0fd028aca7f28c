"""EVA-CLIP-g + BT-Adapter visual backbone (stllm/models/eva_btadapter.py) on the HIP C ABI — BASELINE config 5
(config/minigpt4base_stllm_qa.yaml: ``vit_model: eva_btadapter_g``).

After each of the last ``depth`` ViT blocks a side branch runs a temporal block (attention over the T frames
of every patch position, then ``temporal_fc``) and a spatial block (a clone of the ViT block, attention over
the 257 tokens of every frame, CLS shared across frames); the output is (ViT + branch) / 2
(eva_btadapter.py:147-207).  Frames of a clip are coupled => this backbone shards by clip, not by frame.

Layouts (all fp32 streams, D = 1408, P = 256 patches):
  main stream  h  [(b t) l]        rows (b*T + t)*257 + l                         (as the ViT)
  branch       br [B*P*T + B, D]   rows (b*P + p)*T + t = patch tokens in the reference's 'b (p t)' order
                                   (so every temporal sequence is T consecutive rows), then B CLS rows.
Re-orderings between the two layouts, the CLS means over T, and the (x + y)/2 averages are index-table
gathers (stllm_gather_rows with add / scale) and stllm_mean_t — no arithmetic outside the C ABI.
"""
import torch
import torch.nn as nn

from .. import hip, pack, runtime
from .eva_vit import Attention, Block, VisionTransformer, block_forward
from .layers import Embedding, LayerNorm, Linear, _dev, params_fingerprint


class BTAdapter_Spatial(Block):
    """eva_btadapter.py:257-281.  Built through Block's default norm_layer => LayerNorm eps 1e-5 (not the ViT's 1e-6)."""

    def __init__(self, d_model, n_head, device=None):
        super().__init__(d_model, n_head, mlp_ratio=4.3637, eps=1e-5, device=device)


class BTAdapter_Temp(nn.Module):
    """eva_btadapter.py:283-310: attention over T per patch position + temporal_fc (zero-initialised upstream)."""

    def __init__(self, d_model, n_head, device=None):
        super().__init__()
        self.attn = Attention(d_model, n_head, device)
        self.norm1 = LayerNorm(d_model, 1e-6, device)
        self.temporal_fc = Linear(d_model, d_model, device=device)

    def pack(self, dt):
        a = self.attn
        return dict(n1w=self.norm1.weight, n1b=self.norm1.bias, e1=self.norm1.eps,
                    wqkv=pack.linear(a.qkv.weight, dt), bqkv=pack.vit_qkv_bias(a.q_bias, a.v_bias),
                    wproj=pack.linear(a.proj.weight, dt), bproj=pack.f32(a.proj.bias),
                    wfc=pack.linear(self.temporal_fc.weight, dt), bfc=pack.f32(self.temporal_fc.bias))


class EVAVisionTransformer_BTAdapter(VisionTransformer):
    def __init__(self, depth=3, vit_depth=39, device=None, mask_rate=0):
        super().__init__(depth=vit_depth, device=device)
        self.depth = depth
        self.num_layers = vit_depth
        self.mask_rate = mask_rate  # TubeMasking is unreachable upstream (mask_rate=0, eva_btadapter.py:49,249)
        self.BTAdapter_cls = nn.Parameter(torch.empty(1, 1, self.embed_dim, device=_dev(device)), requires_grad=False)
        self.BTAdapter_S = nn.ModuleList([BTAdapter_Spatial(self.embed_dim, self.num_heads, device) for _ in range(depth)])
        self.BTAdapter_T = nn.ModuleList([BTAdapter_Temp(self.embed_dim, self.num_heads, device) for _ in range(depth)])
        self.BTAdapter_position = Embedding(64, self.embed_dim, device)
        self._bt_packed = {}
        self._idx = {}

    def init_weights(self):
        """eva_btadapter.py:89-99: the spatial adapter blocks start as clones of the last `depth` ViT blocks."""
        start = len(self.blocks) - self.depth
        for j in range(self.depth):
            self.BTAdapter_S[j].load_state_dict(self.blocks[start + j].state_dict())

    def repack(self):
        super().repack()
        self._bt_packed = {}

    def pack_bt(self, dt):
        fp = params_fingerprint([*self.BTAdapter_S.parameters(), *self.BTAdapter_T.parameters()])
        hit = self._bt_packed.get(dt)
        if hit is None or hit[0] != fp:
            hit = (fp, dict(S=[m.pack(dt) for m in self.BTAdapter_S], T=[m.pack(dt) for m in self.BTAdapter_T]))
            self._bt_packed = {dt: hit}
        return hit[1]

    # ---- index tables (host, cached per (B, T)) -------------------------------------------------------
    def _tables(self, B, T, dev):
        key = (B, T, str(dev))
        if key in self._idx:
            return self._idx[key]
        P, L = 256, 257
        b = torch.arange(B).view(B, 1, 1)
        p = torch.arange(P).view(1, P, 1)
        t = torch.arange(T).view(1, 1, T)
        i32 = lambda x: x.reshape(-1).to(torch.int32).to(dev)
        main_of_bpt = (b * T + t) * L + 1 + p            # main-stream row of branch patch (b,p,t)
        br_of_bpt = (b * P + p) * T + t                  # branch row of (b,p,t)
        cls_main = (torch.arange(B).view(B, 1) * T + torch.arange(T).view(1, T)) * L   # [B,T] main rows of CLS tokens
        # spatial-block sequence layout '(b t) l': row (b*T+t)*257 + l  <-  CLS of b (branch row B*P*T + b) | patch (b,p,t)
        bt = torch.arange(B * T).view(B * T, 1)
        l = torch.arange(L).view(1, L)
        bb, tt = bt // T, bt % T
        sp_src = torch.where(l == 0, B * P * T + bb + 0 * l, (bb * P + (l - 1).clamp(min=0)) * T + tt)
        tb = dict(P=P, L=L,
                  main_of_bpt=i32(main_of_bpt), br_of_bpt=i32(br_of_bpt), cls_main=i32(cls_main),
                  pt_of_bpt=i32((p * T + t).expand(B, P, T)),          # row of the (pos_embed + time) table
                  sp_src=i32(sp_src),                                   # spatial layout <- branch rows
                  sp_patch_rows=i32((b * T + t) * L + 1 + p),           # rows of the spatial layout holding (b,p,t)
                  sp_cls_rows=i32(cls_main),
                  arange_bpt=i32(torch.arange(B * P * T)), arange_b=i32(torch.arange(B)),
                  zeros_b=i32(torch.zeros(B)),
                  # init_input's (pos_embed + time embedding) table: row (p, t) <- pos_embed[1 + p] + BTAdapter_position[t].  Built HERE, once: as
                  # `torch.arange(...).to(dev)` inside forward_flat they were two pageable host -> device copies per step = two device
                  # synchronisations in front of the adapter (the host stood still for the 36 plain blocks: 9.9 ms of a 33 ms step, round 5)
                  pos_of_pt=i32(1 + torch.arange(P).view(P, 1).expand(P, T)), time_of_pt=i32(torch.arange(T).view(1, T).expand(P, T)),
                  # final merge: main row (b,t,l) <- branch row
                  out_src=i32(sp_src), arange_main=i32(torch.arange(B * T * L)))
        self._idx = {key: tb}
        return tb

    def _cls_mean(self, stream, rows_idx, B, T):
        """mean over the T frames of a clip of the CLS rows of a '(b t) l' stream -> [B, D]"""
        g = hip.gather_rows(stream, rows_idx)                 # [B*T, D]
        return hip.mean_t(g.view(B, T, -1))

    def forward_flat(self, x):
        """x: [B,T,3,224,224] (or [B,3,T,..] when dim-1 == 3: upstream quirk, :235-237) or 4-D [T,3,224,224].
        Returns the flat fp32 stream [(B*T)*257, 1408]."""
        if x.ndim == 5:
            if x.shape[1] == 3:
                x = x.permute(0, 2, 1, 3, 4)
            B, T = x.shape[0], x.shape[1]
            x = x.reshape((-1,) + tuple(x.shape[2:]))
        elif x.ndim == 4:
            T, B = x.shape[0], 1
        else:
            raise ValueError("expected 4-D or 5-D input")
        self.T = T
        assert T <= 64, "BTAdapter_position has 64 entries (eva_btadapter.py:84)"
        dt = runtime.compute_dtype()
        pk, bt = self.pack(dt), self.pack_bt(dt)
        dev = x.device
        tb = self._tables(B, T, dev)
        P, L, D, H = tb["P"], tb["L"], self.embed_dim, self.num_heads
        N = B * T
        nbr = B * P * T
        h = self.embed_flat(x, pk, dt)
        br = None
        # the blocks in front of the adapter (36 of 39) have no side branch: ONE C call (stllm_vit_blocks) instead of 7 host calls per block — the
        # per-op loop made config 5's encode host-bound (16 ms of enqueue for a 34 ms step, the stream idle in front of the adapter: round 5)
        n_plain = self.num_layers - self.depth
        from . import llama
        if llama.STACK_ENTRY and n_plain > 0:
            if "cblocks" not in pk:
                pk["cblocks"] = hip.vit_block_array(pk["blocks"])
            hip.vit_blocks(h, pk["blocks"][:n_plain], pk["cblocks"], n_seq=N, seq_len=L, num_heads=H, dtype=dt)
        else:
            n_plain = 0
        for i, bp_ in enumerate(pk["blocks"]):
            if i < n_plain:
                continue
            block_forward(h, bp_, N, L, H, dt)
            if i < self.num_layers - self.depth:
                continue
            j = i + self.depth - self.num_layers
            new = torch.empty((nbr + B, D), device=dev, dtype=torch.float32)
            cls_mean = self._cls_mean(h, tb["cls_main"], B, T)                      # x[:,:,0].mean(dim=1)
            if br is None:
                # init_input (eva_btadapter.py:209-231): patches + pos_embed (again) + time embedding; CLS averaged
                # with (BTAdapter_cls + pos_embed[0])
                pt = hip.gather_rows(pk["pos"], tb["pos_of_pt"], add=self.BTAdapter_position.weight, idx_add=tb["time_of_pt"])
                hip.gather_rows(h, tb["main_of_bpt"], add=pt, idx_add=tb["pt_of_bpt"], out=new[:nbr])
                cls_br = hip.gather_rows(self.BTAdapter_cls.view(1, D).float().contiguous(), tb["zeros_b"][:1], add=pk["pos"],
                                         idx_add=tb["zeros_b"][:1])                # BTAdapter_cls + pos_embed[0]
                hip.gather_rows(cls_mean, tb["arange_b"], add=cls_br, idx_add=tb["zeros_b"], out=new[nbr:], scale=0.5)
            else:
                # forward_branch (eva_btadapter.py:188-196): re-ordered main stream + previous branch
                hip.gather_rows(h, tb["main_of_bpt"], add=br, idx_add=tb["arange_bpt"], out=new[:nbr])
                hip.gather_rows(cls_mean, tb["arange_b"], add=br[nbr:], idx_add=tb["arange_b"], out=new[nbr:])
            br = new
            # ---- temporal block: attention over the T frames of every (b, p) ---------------------------
            t_ = bt["T"][j]
            patches = br[:nbr]
            hn, _ = hip.layernorm(patches, t_["n1w"], t_["n1b"], t_["e1"], dtype=dt)
            qkv = hip.gemm(hn, t_["wqkv"], dtype=dt, bias=t_["bqkv"])
            a = hip.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], B=B * P, H=H, Sq=T, Skv=T, D=D // H,
                              scale=(D // H) ** -0.5)
            pr = hip.gemm(a, t_["wproj"], dtype=dt, bias=t_["bproj"])
            hip.gemm(pr, t_["wfc"], dtype=dt, epilogue=hip.EPI_RESID, bias=t_["bfc"], resid=patches)
            # ---- spatial block: attention over the 257 tokens of every frame, CLS shared across frames ------
            s_ = bt["S"][j]
            sx = hip.gather_rows(br, tb["sp_src"])                                       # '(b t) l' layout
            hn, _ = hip.layernorm(sx, s_["n1w"], s_["n1b"], s_["e1"], dtype=dt)
            qkv = hip.gemm(hn, s_["wqkv"], dtype=dt, bias=s_["bqkv"])
            a = hip.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], B=N, H=H, Sq=L, Skv=L, D=D // H,
                              scale=(D // H) ** -0.5)
            res = hip.gemm(a, s_["wproj"], dtype=dt, bias=s_["bproj"], out_f32=True)     # res_spatial
            nxt = torch.empty_like(br)
            hip.gather_rows(res, tb["sp_patch_rows"], add=br, idx_add=tb["arange_bpt"], out=nxt[:nbr])
            cls_res = self._cls_mean(res, tb["sp_cls_rows"], B, T)
            hip.gather_rows(cls_res, tb["arange_b"], add=br[nbr:], idx_add=tb["arange_b"], out=nxt[nbr:])
            br = nxt
            hn, _ = hip.layernorm(br, s_["n2w"], s_["n2b"], s_["e2"], dtype=dt)
            g = hip.gemm(hn, s_["wfc1"], dtype=dt, bias=s_["bfc1"], act=hip.ACT_GELU)
            hip.gemm(g, s_["wfc2"], dtype=dt, epilogue=hip.EPI_RESID, bias=s_["bfc2"], resid=br)
        # ---- (x + branch) / 2 in the main '(b t) l' layout (eva_btadapter.py:179-184) ---------------------
        return hip.gather_rows(br, tb["out_src"], add=h, idx_add=tb["arange_main"], scale=0.5)

    def forward(self, x, return_all_features=False):
        out = self.forward_flat(x)
        return out.view(-1, 257, self.embed_dim)


def create_eva_btadapter(precision="fp16", depth=39, adapter_depth=3, device=None):
    """eva_btadapter.create_eva_btadapter (eva_btadapter.py:312-317)."""
    return EVAVisionTransformer_BTAdapter(depth=adapter_depth, vit_depth=depth, device=device)
