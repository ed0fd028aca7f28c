"""ST-LLM model behind the reference's surface (stllm/models/st_llm.py) on the HIP C ABI.

Same classes, attribute paths and call signatures as the reference:
    registry.get_model_class("st_llm_hf").from_config(cfg)                      (st_llm.py:94, 160-203)
    STLLMForCausalLM.forward(samples=None, inputs_embeds=None, **kw)            (st_llm.py:116-146)
    STLLMLlamaModel.forward(samples=None, inputs_embeds=None, **kw)             (st_llm.py:56-92)
    STLLMModel.forward(samples) / encode_img(image, text=None)                  (st_llm.py:447-546, 321-377)
    model.model.stllm_model.{embed_tokens, llama_tokenizer, video_input, residual_size, up_proj, ...}

What differs by design (MI355X-first):
  * all arithmetic is HIP kernels; the residual streams are fp32, GEMM/attention operands bf16/fp16/fp32
    by ``stllm_amd.runtime`` (no autocast);
  * token-block assembly (prompt_wrap + concat_emb_input_output + BOS + masking, st_llm.py:379-432,
    486-530) is ONE gather kernel driven by an index table built on the host from the token ids —
    visual tokens and embedding-table rows are gathered straight into inputs_embeds;
  * the mask / mask rate can be injected through ``samples["mask"]`` (host numpy RNG otherwise, exactly as
    the reference draws it, st_llm.py:484-486).
Tokenizers are host-side text utilities: ``llama_tokenizer`` / ``tokenizer`` are the offline IdTokenizer
unless real HF tokenizer files are supplied.
"""
import os
import re

import numpy as np
import torch
import torch.nn as nn

from .. import hip, runtime
from ..common.registry import registry
from ..tokenizer import IdTokenizer
from .blip2 import BaseModel, Blip2Base
from .layers import LayerNorm, Linear, Output
from .llama import LlamaConfig, LlamaForCausalLM, LlamaModel
from .utils import RandomMaskingGenerator


class StllmConfig(LlamaConfig):
    model_type = "st_llm_hf"


class Linear_Decoder(nn.Module):
    """st_llm.py:35-43: LayerNorm(Linear(4096,4096)), eps 1e-5."""

    def __init__(self, output_dim=4096, embed_dim=4096, device=None):
        super().__init__()
        self.head = Linear(embed_dim, output_dim, device=device)
        self.norm = LayerNorm(output_dim, 1e-5, device)

    def forward(self, x):
        return self.norm(self.head(x))


def get_residual_index(sample_segments, total_segments):
    """st_llm.py:434-445 / conversation.py:118-125 (numpy round = half-to-even)."""
    seg = float(total_segments) / sample_segments
    return np.array([int((seg / 2) + np.round(seg * i)) for i in range(sample_segments)])


class STLLMModel(Blip2Base):
    _tape = None   # stllm_amd.training sets a dict here: forward() then records what its backward needs (no effect otherwise)

    def __init__(self, vit_model="eva_clip_g", q_former_model="", img_size=224, pre_encoding=False, use_mask=False,
                 mvm_decode=False, video_input=None, residual_size=4, qformer_text_input=False, drop_path_rate=0,
                 use_grad_checkpoint=False, vit_precision="fp16", freeze_vit=True, has_qformer=True,
                 freeze_qformer=True, num_query_token=32, llama_model="", max_txt_len=32, end_sym="\n", device=None):
        super().__init__()
        self.tokenizer = self.init_tokenizer(truncation_side="left")
        self.pre_encoding, self.video_input, self.use_mask = pre_encoding, video_input, use_mask
        self.mvm_decode, self.qformer_text_input, self.residual_size = mvm_decode, qformer_text_input, residual_size
        if self.video_input == "residual":
            self.down_proj = Linear(4096, 1024, device=device)
            self.non_linear_func = nn.ReLU()
            self.up_proj = Linear(1024, 4096, device=device)
        if self.mvm_decode:
            self.mvm_decoder = Linear_Decoder(device=device)
        self.vit_model = vit_model
        self.visual_encoder, self.ln_vision = self.init_vision_encoder(vit_model, img_size, drop_path_rate,
                                                                       use_grad_checkpoint, vit_precision, device=device)
        self.has_qformer = has_qformer
        if has_qformer:
            self.Qformer, self.query_tokens = self.init_Qformer(num_query_token, self.visual_encoder.num_features, device=device)
            if not qformer_text_input:  # st_llm.py:277-283
                self.Qformer.bert.embeddings.word_embeddings = None
                self.Qformer.bert.embeddings.position_embeddings = None
                for layer in self.Qformer.bert.encoder.layer:
                    layer.output = None
                    layer.intermediate = None
            else:
                self.Qformer.resize_token_embeddings(len(self.tokenizer))
            self.Qformer.cls = None
            img_f_dim, self.tokens_per_frame = self.Qformer.config.hidden_size, num_query_token
        else:   # st_llm.py:299-301, 369-373: four concatenated patch tokens per LLM token, 256 / 4 = 64 tokens per frame
            if qformer_text_input:
                raise ValueError("qformer_text_input needs the Q-Former (st_llm.py:337: the text goes through Qformer.bert)")
            img_f_dim, self.tokens_per_frame = self.visual_encoder.num_features * 4, 64
        self.llama_tokenizer = IdTokenizer(pad_token_id=0, bos_token_id=1, eos_token_id=2, vocab_size=32000)
        if qformer_text_input:  # st_llm.py:306-310: '[PAD]' becomes token 32000 (the LLM's tables grow in initialize_vision_modules)
            self.llama_tokenizer.add_special_tokens({"pad_token": "[PAD]"})
            for k in ("bos_token", "eos_token", "unk_token"):
                self.llama_tokenizer.add_special_tokens({k: "</s>"})
        else:
            self.llama_tokenizer.pad_token = "$$"
        self.llama_proj = Linear(img_f_dim, 4096, device=device)
        self.max_txt_len, self.end_sym = max_txt_len, end_sym
        self.embed_tokens = None  # set by STLLMLlamaModel.initialize_vision_modules (st_llm.py:54)
        self.frame_parallel = None  # (rank, world, group): see stllm_amd.parallel
        self._fp_local_clips = False
        # how the ranks share a batch (stllm_amd.parallel): "team" = clip teams, point-to-point exchange per clip, sequence-parallel prefill inside a team
        # (round 5) | "allgather" = flat frame ranges + ONE padded all-gather, the clip's owner prefills alone (rounds 1-4: kept as the fallback exchange)
        self.fp_mode, self.fp_sp, self.fp_balance, self.fp_mailbox = "team", True, "latency", None
        self._sp_state = None

    def set_frame_parallel(self, rank, world, group=None, mode=None, sp=None, balance=None, mailbox=None):
        """Shard the work of a batch over `world` GPUs (stllm_amd.parallel).  eva_clip_g: the frames of a clip are encoded by the clip's TEAM
        of ranks, the token sub-blocks are exchanged point-to-point inside the team and the clip's prefill runs sequence-parallel over the team
        (mode "team", the default) — or, mode "allgather", flat frame ranges + one all-gather + owner-only prefill as in rounds 1-4.
        BT-Adapter backbone: its temporal attention and CLS averaging couple the T frames of a clip (eva_btadapter.py:162-169, 189-190), so the
        unit of work is the whole clip — rank r encodes AND prefills the clips c with c % world == r, no collective (SURVEY.md §8e "shard by
        clip").  mailbox: a parallel.Mailbox standing in for the wire when one process plays the ranks one after another (tests, bench.py's shares)."""
        self.frame_parallel = (rank, world, group) if world > 1 else None
        if mode is not None:
            assert mode in ("team", "allgather")
            self.fp_mode = mode
        if sp is not None:
            self.fp_sp = bool(sp)
        if balance is not None:
            assert balance in ("latency", "throughput")
            self.fp_balance = balance
        self.fp_mailbox = mailbox
        self._sp_state = None

    def fp_wire_dtype(self):
        """dtype of the token sub-blocks on the wire (parallel.exchange_clip_tokens): the compute dtype in the 16-bit modes (half the bytes per xGMI link; the
        first RMSNorm rounds these tokens to it anyway), fp32 (None) in the verify modes; `fp_wire16 = False` keeps fp32 everywhere."""
        dt = runtime.compute_dtype()
        return dt if (getattr(self, "fp_wire16", True) and dt in (torch.bfloat16, torch.float16) and not runtime.gemm_split()) else None

    def _team_plan(self, n_clips, T):
        """the TeamPlan of a batch of n_clips x T frames on this model's ranks.  The prefill is shared inside a team (sequence-parallel) unless the
        forward needs two prefills and a loss over rows of both (MVM: use_mask, st_llm.py:71-91) — then the owner prefills alone."""
        from .. import parallel
        _, world, _ = self.frame_parallel
        tpf = self.tokens_per_frame   # 32 query tokens per frame, 64 pooled patches without the Q-Former
        if self.video_input == "residual":
            lvis = self.residual_size * tpf
        elif self.video_input == "mean":
            lvis = tpf
        else:
            lvis = T * tpf
        return parallel.TeamPlan(n_clips, T, world, sp=self.fp_sp and not self.use_mask, balance=self.fp_balance,
                                 prefill_cost_frames=self.prefill_cost_frames * (lvis + 64) / 576.0)

    # encode time of one frame = 1; one prefilled clip of ~576 positions costs about this many frames; used to level the frame ranges when a
    # batch has fewer clips than ranks.  Measured on one MI355X (round 4, bench.py frame_parallel_projection, config 3 at N = 8): a rank
    # with 38 frames and no prefill takes 28.6 ms = 0.75 ms per frame at that batch size, a rank with 26 frames + one prefill (S = 580)
    # 32.4 ms; with 24 / 40 frames: 31.9 / 28.4 ms.  A rank's encode is NOT proportional to its frames — t(F) ~ 7 ms + 0.53 ms x F (launch-
    # bound small kernels, per-GEMM prologues / epilogues, tile quantisation at M = 257 F rows) — so what levels the ranks is the prefill's
    # 12 ms over the MARGINAL frame cost: 12 / 0.53 = 22 frames.  (Rounds 1-3 used 12, from c2's 11.8 ms of LLM time and 0.96 ms per frame
    # at 16 frames: the 26 / 38 split it gave left the prefill ranks 3.8 ms behind; 22 gives 21 / 43.)
    prefill_cost_frames = 22.0

    def _prefill_load(self, n_clips, T, world):
        """frames-equivalent of the prefill work of every rank (clip c -> rank c % world), scaled with the visual tokens per clip"""
        from .. import parallel
        tpf = self.tokens_per_frame   # 32 query tokens per frame, 64 pooled patches without the Q-Former
        if self.video_input == "residual":
            lvis = self.residual_size * tpf
        elif self.video_input == "mean":
            lvis = tpf
        else:
            lvis = T * tpf
        per_clip = self.prefill_cost_frames * (lvis + 64) / 576.0
        return [len(parallel.clips_of_rank(n_clips, r, world)) * per_clip for r in range(world)]

    # ------------------------------------------------------------------------------------------
    def _qformer_ids(self, text, n_frames, T):
        """st_llm.py:337-350: BERT-tokenise the instruction, one copy per frame."""
        assert text
        if isinstance(text, str):
            text = [text] * n_frames
        elif len(text) != n_frames:
            text = [t for t in text for _ in range(T)]
        tok = self.tokenizer(text, padding="longest", truncation=True, max_length=self.max_txt_len, return_tensors="pt")
        return tok.input_ids, tok.attention_mask

    def encode_img(self, image, text=None):
        """st_llm.py:321-377.  5-D [B,T,3,224,224] -> [B,T,32,4096]; 4-D [T,3,224,224] -> [T,32,4096] (fp32)."""
        dt = runtime.compute_dtype()
        T = image.shape[1]
        infer = image.dim() == 4
        use_image = True if T == 1 or infer else False
        if self.frame_parallel is not None and not infer and self.vit_model == "eva_clip_g" and self.fp_mode == "team" and not use_image:
            # clip teams (stllm_amd.parallel): this rank encodes its sub-ranges of its teams' clips in ONE batch, the sub-blocks travel
            # point-to-point to the team members that prefill the clip; the result holds exactly the clips this rank prefills (a share of)
            from .. import parallel
            rank, world, group = self.frame_parallel
            B = image.shape[0]
            plan = self._team_plan(B, T)
            frames = image.reshape((-1,) + tuple(image.shape[2:]))
            enc = plan.encodes(rank)
            local = {}
            if enc:
                idx = [c * T + f for c, f0, f1 in enc for f in range(f0, f1)]
                contiguous = idx == list(range(idx[0], idx[0] + len(idx)))
                fr = frames[idx[0]: idx[0] + len(idx)] if contiguous else frames[torch.as_tensor(idx, device=frames.device)]
                t_local = None
                if self.qformer_text_input:
                    all_t = [text] * frames.shape[0] if isinstance(text, str) else [t for t in text for _ in range(T)]
                    t_local = [all_t[i] for i in idx]
                toks = self._encode_frames(fr, t_local, T, dt)
                o = 0
                for c, f0, f1 in enc:
                    local[c] = toks[o: o + (f1 - f0)]
                    o += f1 - f0
            blocks = parallel.exchange_clip_tokens(local, plan, rank, group, self.fp_mailbox, token_shape=(self.tokens_per_frame, 4096), device=image.device,
                                                   wire_dtype=self.fp_wire_dtype())
            need = plan.clips_of(rank)
            self._fp_local_clips = True
            if getattr(self, "_fp_keep_tokens", False):   # bench.py / tests: the blocks as they arrived, per clip
                self._fp_last_tokens = {c: blocks[c] for c in need}
            if not need:
                inputs_llama = torch.zeros((0, T, self.tokens_per_frame, 4096), dtype=torch.float32, device=image.device)
            elif len(need) == 1:
                inputs_llama = blocks[need[0]].view(1, T, -1, 4096)
            else:
                inputs_llama = torch.stack([blocks[c] for c in need], dim=0)
            atts_llama = torch.ones(inputs_llama.shape[:-1], dtype=torch.long, device=image.device)
            return inputs_llama, atts_llama, use_image
        if self.frame_parallel is not None and not infer and self.vit_model == "eva_clip_g":
            from .. import parallel
            rank, world, group = self.frame_parallel
            frames = image.reshape((-1,) + tuple(image.shape[2:]))
            qtext = text
            load = self._prefill_load(image.shape[0], T, world)

            def enc_local(fr, _s=[0]):
                s0, _ = parallel.frame_range(frames.shape[0], rank, world, load)
                t_local = None
                if self.qformer_text_input:  # each rank needs the text of the clips its frames belong to
                    all_t = [qtext] * frames.shape[0] if isinstance(qtext, str) else [t for t in qtext for _ in range(T)]
                    t_local = all_t[s0: s0 + fr.shape[0]]
                return self._encode_frames(fr, t_local, T, dt)
            # image batches (T == 1 -> use_image) are not clip-sharded by forward(): every rank prefills the whole batch and so needs
            # every image's tokens — the collective may only be skipped on the video path
            self._fp_local_clips = (not use_image) and not parallel.gather_needed(frames.shape[0], T, world, load)
            if self._fp_local_clips:   # this rank's frames ARE the clips it prefills (one clip per GPU): nothing to exchange
                s0, e0 = parallel.frame_range(frames.shape[0], rank, world, load)
                tokens = enc_local(frames[s0:e0]) if e0 > s0 else torch.zeros((0, self.tokens_per_frame, 4096), dtype=torch.float32, device=image.device)
            else:
                tokens = parallel.encode_frames_parallel(enc_local, frames, rank, world, group, token_shape=(self.tokens_per_frame, 4096), extra=load,
                                                         simulate=getattr(self, "_fp_sim_tokens", None))
            if getattr(self, "_fp_keep_tokens", False):   # bench.py's one-off check of the gathered block against a single-GPU encode
                self._fp_last_tokens = tokens
            inputs_llama = tokens.view(-1, T, tokens.shape[1], 4096)
            atts_llama = torch.ones(inputs_llama.shape[:-1], dtype=torch.long, device=image.device)
            return inputs_llama, atts_llama, use_image
        taped_vision = self._tape is not None and self._tape.get("want_vision", False) and self.vit_model != "eva_clip_g"
        if self.vit_model == "eva_clip_g":
            frames = image.reshape((-1,) + tuple(image.shape[2:])) if image.dim() == 5 else image
            feats = self.visual_encoder.forward_features_flat(frames)
            n = frames.shape[0]
        else:
            if taped_vision:   # training of the BTAdapter* parameters: keep the branch's activations (training_vision.py)
                from .. import training_vision
                feats, self._tape["bt_tape"] = training_vision.btadapter_forward_taped(self.visual_encoder, image, self._tape.get("drop_path"))
            else:
                feats = self.visual_encoder.forward_flat(image)
            n = feats.shape[0] // 257
        enc16, _ = hip.layernorm(feats, self.ln_vision.weight, self.ln_vision.bias, self.ln_vision.eps, dtype=dt)
        if not self.has_qformer:
            if self._tape is not None:
                raise NotImplementedError("training through the no-Q-Former projector is not implemented")
            inputs_llama = self._project_patches(enc16, n, dt)
            if not infer:
                inputs_llama = inputs_llama.view(-1, T, inputs_llama.shape[1], 4096)
            atts_llama = torch.ones(inputs_llama.shape[:-1], dtype=torch.long, device=image.device)
            return inputs_llama, atts_llama, use_image
        ids = tmask = None
        if self.qformer_text_input:
            ids, tmask = self._qformer_ids(text, n, T)
        if taped_vision:       # ... and the frozen Q-Former's, for the dgrad sweep back to the image tokens
            _, hq16, self._tape["qf_tape"] = training_vision.qformer_forward_taped(self.Qformer.bert, self.query_tokens[0], enc16, n, ids, tmask)
            self._tape["feats"] = feats
        else:
            _, hq16, _ = self.Qformer.bert.encode(self.query_tokens[0], enc16, n, ids, tmask)
        if self._tape is not None:
            self._tape["hq16"] = hq16
        w, b = self.llama_proj.packed(dt)
        inputs_llama = hip.gemm(hq16, w, dtype=dt, bias=b, out_f32=True).view(n, -1, 4096)
        if not infer:
            inputs_llama = inputs_llama.view(-1, T, inputs_llama.shape[1], 4096)
        atts_llama = torch.ones(inputs_llama.shape[:-1], dtype=torch.long, device=image.device)
        return inputs_llama, atts_llama, use_image

    def _project_patches(self, enc16, n, dt):
        """st_llm.py:369-373 (has_qformer=False): drop CLS, view the 256 patch tokens of a frame as 64 rows of 4 concatenated tokens and project
        them — llama_proj(5632 -> 4096).  No copy: rows 1..256 of a frame are contiguous in the ln_vision output [n * 257, 1408], so the GEMM's A
        operand is that buffer read through the 2-level row indexing (64 rows of 5632 per frame, frame stride 257 * 1408)."""
        C = enc16.shape[1]
        a = enc16.reshape(-1)[C: C + 64 * 4 * C].view(64, 4 * C)
        w, b = self.llama_proj.packed(dt)
        return hip.gemm(a, w, dtype=dt, bias=b, out_f32=True, M=n * 64, a_rows=(64, 257 * C)).view(n, 64, 4096)

    def _project_features(self, feats):
        """st_llm.py:452-454 (pre_encoding): llama_proj on pre-extracted features [B, T, L, C] -> [B, T, L, 4096] fp32 — one GEMM with the bias fused."""
        return self.llama_proj(feats)

    def _encode_frames(self, frames, text_per_frame, T, dt):
        """ViT -> ln_vision -> Q-Former -> projector for a flat list of frames -> [n,32,4096] fp32."""
        n = frames.shape[0]
        feats = self.visual_encoder.forward_features_flat(frames)
        enc16, _ = hip.layernorm(feats, self.ln_vision.weight, self.ln_vision.bias, self.ln_vision.eps, dtype=dt)
        hook = getattr(self, "_after_vit_hook", None)
        if hook is not None:   # forward(): the host-side plan is built here, under the ViT's kernels
            hook()
        if not self.has_qformer:
            return self._project_patches(enc16, n, dt)
        ids = tmask = None
        if self.qformer_text_input:
            tok = self.tokenizer(text_per_frame, padding="longest", truncation=True, max_length=self.max_txt_len, return_tensors="pt")
            ids, tmask = tok.input_ids, tok.attention_mask
        _, hq16, _ = self.Qformer.bert.encode(self.query_tokens[0], enc16, n, ids, tmask)
        w, b = self.llama_proj.packed(dt)
        return hip.gemm(hq16, w, dtype=dt, bias=b, out_f32=True).view(n, -1, 4096)

    # ------------------------------------------------------------------------------------------
    def pool_video(self, img_embeds):
        """st_llm.py:463-478 — [B,T,32,D] -> [B,1,L,D] ('all' | 'mean' | 'residual' global-local module)."""
        B, T, Lq, D = img_embeds.shape
        if self._tape is not None:
            self._tape["pool_shape"] = (B, T, Lq, D)
        if self.video_input == "all":
            return img_embeds.reshape(B, 1, T * Lq, D).contiguous()
        if self.video_input == "mean":
            return hip.mean_t(img_embeds.contiguous()).view(B, 1, Lq, D)
        if self.video_input == "residual":
            dt = runtime.compute_dtype()
            R = self.residual_size
            ridx = self.get_residual_index(R, T, img_embeds.device)
            g = hip.mean_t(img_embeds.contiguous()).view(B * Lq, D)  # the reference expands to R copies first: same values
            wd, bd = self.down_proj.packed(dt)
            wu, bu = self.up_proj.packed(dt)
            g16 = hip.cast_rows(g, dt)
            h = hip.gemm(g16, wd, dtype=dt, bias=bd, act=hip.ACT_RELU)
            gg = hip.gemm(h, wu, dtype=dt, bias=bu, out_f32=True)
            b_i = torch.arange(B).view(B, 1, 1)
            r_i = torch.as_tensor(ridx).view(1, R, 1)
            l_i = torch.arange(Lq).view(1, 1, Lq)
            idx = hip.h2d(((b_i * T + r_i) * Lq + l_i).reshape(-1).to(torch.int32), img_embeds.device)
            idx_add = hip.h2d((b_i * Lq + l_i).expand(B, R, Lq).reshape(-1).to(torch.int32), img_embeds.device)
            out = hip.gather_rows(img_embeds.reshape(-1, D), idx, add=gg, idx_add=idx_add)
            if self._tape is not None:
                self._tape.update(pool_g16=g16, pool_h=h, pool_idx=idx, pool_idx_add=idx_add)
            return out.view(B, 1, R * Lq, D)
        return img_embeds

    def get_residual_index(self, sample_segments, total_segments, devices=None):
        # the reference caches the first (R, T) it sees in a buffer (st_llm.py:435-445);
        # recomputing per call gives the same observable result for a fixed T and is correct when T changes.
        return get_residual_index(sample_segments, total_segments)

    # ------------------------------------------------------------------------------------------
    def _upload_rows(self, rows, device):
        """rows: list (B) of lists of gather indices (>=0: row of the visual-token block; <0: -(token id)-1) — all the same length.
        Host -> device through the pinned ring (asynchronous): forward() does this BEFORE the encode is enqueued, so the table is
        resident long before the gather that reads it and the host never makes the GPU wait between the projector and the prefill."""
        return hip.h2d(torch.tensor(rows, dtype=torch.int32).reshape(-1), device), len(rows), len(rows[0])

    def _gather_tokens(self, vis_flat, rows):
        """One kernel assembles inputs_embeds [B,S,D] from visual tokens + embedding-table rows (rows: index lists, or the
        (idx, B, S) triple _upload_rows returned for them)."""
        idx, B, S = rows if isinstance(rows, tuple) else self._upload_rows(rows, vis_flat.device)
        out = hip.gather_rows(vis_flat, idx, src_b=self.embed_tokens.weight)
        if self._tape is not None:
            self._tape.setdefault("gather_idx", []).append(idx)
        return out.view(B, S, vis_flat.shape[-1])

    def _assemble(self, L_total, kept, instruction, answers_ids, B):
        """Index-table form of prompt_wrap (st_llm.py:379-407) + concat_emb_input_output (:409-432) +
        BOS (:519-530) + targets (:532-542).  kept[b] = positions (within the L_total visual tokens of sample b)
        that enter the sequence."""
        tk = self.llama_tokenizer
        pad, bos = tk.pad_token_id, tk.bos_token_id
        tok = lambda t: -(int(t)) - 1
        prompts = [instruction] * B if isinstance(instruction, str) else list(instruction)
        ins = []
        for b in range(B):
            p_before, p_after = prompts[b].split("<ImageHere>")
            before = tk(p_before, return_tensors="pt", add_special_tokens=False).input_ids[0].tolist()
            after = tk(p_after, return_tensors="pt", add_special_tokens=self.qformer_text_input).input_ids[0].tolist()
            ins.append([tok(t) for t in before] + [b * L_total + int(k) for k in kept[b]] + [tok(t) for t in after])
        max_in = max(len(r) for r in ins)
        La = max(len(a) for a in answers_ids)
        prepend = not self.qformer_text_input
        rows, atts, input_lens = [], [], []
        for b in range(B):
            n = len(ins[b])
            input_lens.append(n)
            ans = list(answers_ids[b]) + [pad] * (La - len(answers_ids[b]))
            ans_att = [1] * len(answers_ids[b]) + [0] * (La - len(answers_ids[b]))
            row = ins[b] + [tok(t) for t in ans] + [tok(pad)] * (max_in - n)
            att = [1] * n + ans_att + [0] * (max_in - n)
            if prepend:
                row, att = [tok(bos)] + row, [1] + att
            rows.append(row)
            atts.append(att)
        S = len(rows[0])
        targets = torch.full((B, S), -100, dtype=torch.long)
        off = 1 if prepend else 0
        for b in range(B):
            a = torch.tensor(list(answers_ids[b]) + [pad] * (La - len(answers_ids[b])), dtype=torch.long)
            a = a.masked_fill(a == pad, -100)
            targets[b, input_lens[b] + off: input_lens[b] + La + off] = a
        return rows, torch.tensor(atts, dtype=torch.long), targets

    def forward(self, samples):
        """st_llm.py:447-546.  Returns (inputs_embeds, attention_mask, unmask_inputs_embeds,
        unmask_attention_mask, targets) — embeddings fp32 on the device, masks/targets on the host side too."""
        image = samples["image"]
        instruction = samples["instruction_input"] if "instruction_input" in samples else None
        if self.qformer_text_input:
            qtext = [it.split("Human: ")[1].split(" ###")[0] for it in instruction]
        else:
            qtext = None
        clip_sharded = False
        self._sp_state = None
        pre = bool(self.pre_encoding)
        if pre:   # st_llm.py:452-455: `image` = pre-extracted features [B, T, L, C]; only llama_proj runs
            if image.dim() != 4 or image.shape[-1] != self.llama_proj.weight.shape[1]:
                raise ValueError(f"pre_encoding: samples['image'] must hold features [B, T, L, {self.llama_proj.weight.shape[1]}], got {tuple(image.shape)}")
            if self.frame_parallel is not None:
                raise NotImplementedError("pre_encoding: nothing is encoded, so there are no frames to share between ranks — shard the clips at the caller")
        if self.frame_parallel is not None and self.vit_model != "eva_clip_g" and image.dim() == 5:
            # BT-Adapter: clip-parallel from the first kernel on — this rank's clips only, then the single-GPU path
            from .. import parallel
            rank, world, _ = self.frame_parallel
            own = parallel.clips_of_rank(image.shape[0], rank, world)
            self.owned_clips = own
            if not own:
                return None
            image = image[own].contiguous()
            if instruction is not None and not isinstance(instruction, str):
                instruction = [instruction[c] for c in own]
            if qtext is not None:
                qtext = [qtext[c] for c in own]
            samples = dict(samples, answer=[samples["answer"][c] for c in own])
            if "mask" in samples and samples["mask"] is not None:
                samples["mask"] = torch.as_tensor(samples["mask"])[own]
            clip_sharded = True
        # ---- the host-side plan (tokenizers, _assemble, index tables) depends on shapes and token ids only, not on a single device result.
        # Order (round 5): who-prefills-what first (trivial), then the ENCODE is enqueued — two C calls, ~1 ms of host time for >= 10 ms of GPU work —,
        # then the plan is built and its tables uploaded WHILE the GPU encodes, then pooling + gather.  Rounds 1-4 built the plan after the encode's
        # per-op Python launches had eaten the host's lead (0.65 ms of idle GPU per step under rocprofv3, profiles/r04_bench_gaps_final.md); building it
        # BEFORE the encode (first version of this round) hides it in steady state but exposes it whenever the GPU is idle at the start of a step
        # (a single request, the first step after a synchronisation, a profiler-slowed host: 0.9-1.2 ms, profiles/r05_bench_gaps.md).
        dev = image.device
        T = image.shape[1]
        use_image = bool(T == 1 or image.dim() == 4) and not pre   # encode_img's rule (st_llm.py:326-328); pre_encoding never sets it (st_llm.py:451)
        Lq = image.shape[2] if pre else self.tokens_per_frame
        if use_image:
            L = Lq
        elif self.video_input == "all":
            L = T * Lq
        elif self.video_input == "residual":
            L = self.residual_size * Lq
        else:                                                # "mean"; None leaves [B,T,32,D] — which the reference's assembly cannot take either
            L = Lq
        B = image.shape[0]
        answers_txt = list(samples["answer"])
        own = None
        if self.frame_parallel is not None and not use_image and not clip_sharded:
            # clip-parallel prefill: this rank continues with the clips it owns (clip c -> rank c % world)
            from .. import parallel
            rank, world, group = self.frame_parallel
            self._sp_state = None
            if self.fp_mode == "team" and self.vit_model == "eva_clip_g":
                plan = self._team_plan(image.shape[0], T)
                own = plan.clips_of(rank)
                if len(own) == 1 and plan.sp[own[0]]:    # this rank runs ITS position range of the clip's prefill (LlamaModel.prefill_sp)
                    team = plan.team[own[0]]
                    self._sp_state = dict(index=team.index(rank), size=len(team), ranks=team, rank=rank, group=group, mailbox=self.fp_mailbox)
            else:
                own = parallel.clips_of_rank(image.shape[0], rank, world)
            self.owned_clips = own
            if own:
                if instruction is not None and not isinstance(instruction, str):
                    instruction = [instruction[c] for c in own]
                answers_txt = [answers_txt[c] for c in own]
                if "mask" in samples and samples["mask"] is not None:
                    samples = dict(samples, mask=torch.as_tensor(samples["mask"])[own])
                B = len(own)
        # ---- device work, part 1: the encode (+ the team's token exchange) goes out now; the host runs ahead of it ----------------
        # The plan (tokenizers, _assemble, index tables: Python) is built by a hook that _encode_frames calls right after the ViT + ln_vision have been
        # enqueued and BEFORE the Q-Former's ~100 short launches (round 6): the GPU is busy with >= 10 ms of ViT kernels then.  Behind the whole encode
        # (round 5) it was free only while the host kept a lead through the Q-Former's 5-18 us kernels — which a profiler-slowed host does not: 0.3-0.57 ms
        # of idle GPU in front of gather_rows under rocprofv3 (profiles/r06_bench_gaps.md).
        def build_plan():
            kept = [list(range(L)) for _ in range(B)]
            mask = None
            if not use_image and self.use_mask:
                self.img_len = L
                if "mask" in samples and samples["mask"] is not None:
                    mask = torch.as_tensor(samples["mask"]).to(torch.bool).cpu().view(B, L)
                else:
                    rate = np.random.normal(0.5, 0.1)
                    mask = RandomMaskingGenerator(L, float(np.clip(rate, 0.1, 0.7)), B)
                self.mask = mask.unsqueeze(1)
                kept = [torch.nonzero(~mask[b]).flatten().tolist() for b in range(B)]
                self.mask_img_len = len(kept[0])
                assert all(len(k) == self.mask_img_len for k in kept)
            self.llama_tokenizer.padding_side = "right"
            text = [t + self.llama_tokenizer.eos_token for t in answers_txt] if self.qformer_text_input \
                else [t + self.end_sym for t in answers_txt]
            tr = self.llama_tokenizer(text, return_tensors="pt", padding="longest", truncation=True,
                                      max_length=self.max_txt_len, add_special_tokens=False)
            answers = [tr.input_ids[b][: int(tr.attention_mask[b].sum())].tolist() for b in range(B)]
            rows, attention_mask, targets = self._assemble(L, kept, instruction, answers, B)
            plan = dict(rows=self._upload_rows(rows, dev), attention_mask=hip.with_host(attention_mask, dev), targets=hip.with_host(targets, dev))
            if mask is not None:
                urows, un_a, _ = self._assemble(L, [list(range(L))] * B, instruction, answers, B)
                plan.update(urows=self._upload_rows(urows, dev), un_a=hip.with_host(un_a, dev))
            return plan

        plan_box = []

        def after_vit():
            if not plan_box and (own is None or own):
                plan_box.append(build_plan())

        self._after_vit_hook = after_vit
        try:
            if pre:
                img_embeds, use_image_enc = self._project_features(image), False
            else:
                img_embeds, atts_img, use_image_enc = self.encode_img(image, qtext)
        finally:
            self._after_vit_hook = None
        assert use_image_enc == use_image
        if own is not None and not own:     # this rank only encoded (and sent) frames: no clip of the batch is prefilled here
            return None
        after_vit()                          # (paths that never reach _encode_frames: pre_encoding, the BT-Adapter backbone)
        plan = plan_box[0] if plan_box else None
        # ---- device work, part 2: pooling -> ONE gather per sequence block ------------------------------------------------------------
        if not use_image:
            img_embeds = self.pool_video(img_embeds)
        elif img_embeds.dim() == 3:
            img_embeds = img_embeds.unsqueeze(1)
        if own is not None:
            if not getattr(self, "_fp_local_clips", False):   # (skipped all-gather: img_embeds holds exactly the owned clips already)
                img_embeds = img_embeds[own].contiguous()
        assert tuple(img_embeds.shape[:3]) == (B, 1, L), (tuple(img_embeds.shape), B, L)
        D = img_embeds.shape[-1]
        vis_flat = img_embeds.reshape(B * L, D)
        if self._tape is not None:
            self._tape.update(vis_rows=B * L, pooled=not use_image)
        inputs_embeds = self._gather_tokens(vis_flat, plan["rows"])
        un_e = un_a = None
        if "urows" in plan:
            un_e = self._gather_tokens(vis_flat, plan["urows"])
            un_a = plan["un_a"]
        return inputs_embeds, plan["attention_mask"], un_e, un_a, plan["targets"]

    @classmethod
    def from_config(cls, cfg, device=None):
        g = cfg.get
        model = cls(vit_model=g("vit_model", "eva_clip_g"), q_former_model=g("q_former_model", ""),
                    img_size=g("image_size", 224), pre_encoding=g("pre_encoding", False), use_mask=g("use_mask", False),
                    mvm_decode=g("mvm_decode", False), video_input=g("video_input", None),
                    residual_size=g("residual_size", 4), qformer_text_input=g("qformer_text_input", False),
                    drop_path_rate=g("drop_path_rate", 0), use_grad_checkpoint=g("use_grad_checkpoint", False),
                    vit_precision=g("vit_precision", "fp16"), freeze_vit=g("freeze_vit", True),
                    has_qformer=g("has_qformer", True), freeze_qformer=g("freeze_qformer", True),
                    num_query_token=g("num_query_token", 32), llama_model=g("llama_model", ""),
                    max_txt_len=g("max_txt_len", 32), end_sym=g("end_sym", "\n"), device=device)
        ckpt_path = g("ckpt", "")
        if ckpt_path and os.path.isfile(ckpt_path):
            ckpt = torch.load(ckpt_path, map_location="cpu")
            ckpt = ckpt.get("model", ckpt)
            if "llm_proj.weight" in ckpt:  # st_llm.py:601-603
                ckpt["llama_proj.weight"] = ckpt.pop("llm_proj.weight")
                ckpt["llama_proj.bias"] = ckpt.pop("llm_proj.bias")
            model.load_state_dict(ckpt, strict=False)
        return model


def _resized_rows(w, n, name):
    """[old, D] -> [n, D]: the first min(old, n) rows are kept, new rows are drawn like HF's _init_weights (normal, std 0.02)"""
    old = w.shape[0]
    if old == n:
        return w
    new = torch.empty((n, w.shape[1]), dtype=w.dtype, device=w.device)
    k = min(old, n)
    new[:k] = w.data[:k]
    if n > old:
        g = torch.Generator().manual_seed(20230911 + old)
        new[old:] = (torch.randn((n - old, w.shape[1]), generator=g) * 0.02).to(new.device, new.dtype)
    return nn.Parameter(new, requires_grad=w.requires_grad)


class STLLMLlamaModel(LlamaModel):
    config_class = StllmConfig

    def initialize_vision_modules(self, cfg, device=None):
        self.stllm_model = STLLMModel.from_config(cfg, device=device)
        if cfg.get("qformer_text_input", False):   # st_llm.py:52-53
            self.resize_token_embeddings(len(self.stllm_model.llama_tokenizer))
        self.stllm_model.embed_tokens = self.embed_tokens  # shared module (st_llm.py:54)

    def resize_token_embeddings(self, n):
        """HF PreTrainedModel.resize_token_embeddings for the input table: old rows kept, new rows ~ N(0, initializer_range)
        from a fixed generator; config.vocab_size follows (the outer model then grows lm_head to it, st_llm.py:180-181)."""
        self.embed_tokens.weight = _resized_rows(self.embed_tokens.weight, n, "model.embed_tokens.weight")
        self.config.vocab_size = n
        return self.embed_tokens

    def forward(self, samples=None, inputs_embeds=None, **kwargs):
        if samples is None:
            return super().forward(inputs_embeds=inputs_embeds, **kwargs)
        sm = self.stllm_model
        res = sm(samples)
        if res is None:  # frame-parallel run and this rank owns no clip of the batch
            return None, None, None
        inputs_embeds, attention_mask, un_e, un_a, labels = res
        outputs = super().forward(attention_mask=attention_mask, inputs_embeds=inputs_embeds, use_cache=False,
                                  output_hidden_states=un_e is not None, return_dict=True, sp=sm._sp_state if un_e is None else None)
        if un_e is None:
            return outputs, None, labels
        # ---- MVM branch (st_llm.py:71-91) ---------------------------------------------------------
        dt = runtime.compute_dtype()
        img_start = 0 if sm.qformer_text_input else 8
        mask_output = outputs.hidden_states[-1]
        B, S1, D = mask_output.shape
        Lk = sm.mask_img_len
        dev = mask_output.device
        rows = (torch.arange(B).view(B, 1) * S1 + img_start + torch.arange(Lk).view(1, Lk)).reshape(-1)
        a = hip.gather_rows(mask_output.reshape(B * S1, D), hip.h2d(rows.to(torch.int32), dev))
        if hasattr(sm, "mvm_decoder"):
            a = sm.mvm_decoder(a)
        un_out = super().forward(inputs_embeds=un_e, attention_mask=un_a, return_dict=True, use_cache=False,
                                 output_hidden_states=True).hidden_states[-1]
        S2 = un_out.shape[1]
        keep = (~sm.mask.squeeze(1))
        pos = torch.stack([torch.nonzero(keep[b]).flatten() for b in range(B)])  # [B, Lk]
        idx_b = hip.h2d((torch.arange(B).view(B, 1) * S2 + img_start + pos).reshape(-1).to(torch.int32), dev)
        loss_rows = hip.cosine_rows(a, un_out.reshape(B * S2, D), None, idx_b, n_rows=B * Lk)
        return outputs, loss_rows.mean(), labels


@registry.register_model("st_llm_hf")
class STLLMForCausalLM(LlamaForCausalLM, BaseModel):
    config_class = StllmConfig
    PRETRAINED_MODEL_CONFIG_DICT = {
        "instructblip_vicuna0": "configs/models/instructblip_vicuna0.yaml",
        "instructblip_vicuna0_btadapter": "configs/models/instructblip_vicuna0_btadapter.yaml",
        "minigpt4_vicuna0": "configs/models/minigpt4_vicuna0.yaml",
        "minigpt4_vicuna0_btadapter": "configs/models/minigpt4_vicuna0_btadapter.yaml",
    }

    def __init__(self, config, device=None):
        nn.Module.__init__(self)
        self.config = config
        self.model = STLLMLlamaModel(config, device)
        self.vocab_size = config.vocab_size
        self.lm_head = Linear(config.hidden_size, config.vocab_size, bias=False, device=device)
        self._lm_packed = {}

    def get_model(self):
        return self.model

    def resize_token_embeddings(self, n):
        """st_llm.py:180-181 (`model.resize_token_embeddings(model.config.vocab_size)`): both tables to n rows."""
        self.model.resize_token_embeddings(n)
        self.lm_head.weight = _resized_rows(self.lm_head.weight, n, "lm_head.weight")
        self.vocab_size = self.config.vocab_size = n
        self._lm_packed = {}
        return self.model.embed_tokens

    def forward(self, samples=None, inputs_embeds=None, **kwargs):
        if samples is None:  # plain causal-LM forward used by generate() (st_llm.py:118-119)
            kwargs.pop("labels", None)
            out = self.model(samples=None, inputs_embeds=inputs_embeds, **kwargs)
            B, S, _ = out.last_hidden_state.shape
            h16 = out._h16
            if h16.shape[0] != B * S:   # decode steps return the compute-dtype hidden of the LAST token only
                S = h16.shape[0] // B
            logits = self.logits_from(h16, B, S)
            hip.gemm_workspace_check(logits.device) if logits.is_cuda else None   # non-blocking: reports a timed-out split-K exchange of the PREVIOUS call
            return Output(loss=None, logits=logits, past_key_values=out.past_key_values,
                          hidden_states=out.hidden_states, attentions=None)
        outputs, loss_pretrain, labels = self.model(samples)
        if outputs is None:
            return Output(loss=None, logits=None, past_key_values=None, hidden_states=None, attentions=None)
        B, S, _ = outputs.last_hidden_state.shape
        logits = self.logits_from(outputs._h16, B, S)
        loss = None
        sp_rows = getattr(outputs, "_sp_rows", None)   # sequence-parallel prefill: this rank holds the positions [s0, s1) of the clip only
        if sp_rows is not None and labels is not None:
            # lm_head + CE follow the rows.  The clip's loss is the sum over the team's row losses, accumulated along the team (member j adds its
            # rows to what member j - 1 sent and passes the sum on): complete on the LAST member — which also holds the answer positions.
            from .. import parallel
            sp = self.model.stllm_model._sp_state
            s0, s1 = sp_rows
            lab_h = hip.host_mask(labels)
            shift_h = torch.full_like(lab_h, -100)
            shift_h[:, :-1] = lab_h[:, 1:]
            part = torch.zeros((1,), dtype=torch.float32, device=logits.device)
            if S > 0:
                rows = hip.cross_entropy_rows(logits.reshape(B * S, -1), hip.h2d(shift_h[:, s0:s1].reshape(-1).to(torch.int32), logits.device))
                part = (rows.sum() / max(int((shift_h != -100).sum()), 1)).reshape(1)
            j, k, ranks = sp["index"], sp["size"], sp["ranks"]
            if j > 0:
                prev = torch.zeros_like(part)
                for w in parallel.p2p_exchange([], [(prev, ranks[j - 1], ("loss", 0))], sp["rank"], sp.get("group"), sp.get("mailbox")):
                    w.wait()
                part = part + prev
            if j + 1 < k:
                for w in parallel.p2p_exchange([(part, ranks[j + 1], ("loss", 0))], [], sp["rank"], sp.get("group"), sp.get("mailbox")):
                    w.wait()
            hip.gemm_workspace_check(logits.device) if logits.is_cuda else None
            # a partial sum must not look like the clip's loss — `loss` is the scalar on the LAST member only and None before it
            # (the running sum of members 0 .. j stays readable as `loss_partial`)
            res = Output(loss=part[0] if j + 1 == k else None, logits=logits, past_key_values=None, hidden_states=outputs.hidden_states, attentions=None)
            object.__setattr__(res, "loss_mvm", None)
            object.__setattr__(res, "sp_rows", (s0, s1))            # logits = rows [s0, s1) of the clip's sequence
            object.__setattr__(res, "loss_complete", j + 1 == k)
            object.__setattr__(res, "loss_partial", part[0])
            return res
        if labels is not None:  # shifted CE (st_llm.py:125-135)
            lab_h = getattr(labels, "_stllm_host", None)
            if lab_h is not None:   # the targets were built on the host (st_llm.py:532-542): shift them there, one asynchronous H2D copy
                shift_h = torch.full_like(lab_h, -100)
                shift_h[:, :-1] = lab_h[:, 1:]
                rows = hip.cross_entropy_rows(logits.reshape(B * S, -1), hip.h2d(shift_h.reshape(-1).to(torch.int32), logits.device))
                loss = rows.sum() / max(int((shift_h != -100).sum()), 1)
            else:
                shift = torch.full_like(labels, -100)
                shift[:, :-1] = labels[:, 1:]
                rows = hip.cross_entropy_rows(logits.reshape(B * S, -1), shift.reshape(-1).to(torch.int32))
                loss = rows.sum() / (shift != -100).sum().clamp(min=1)
        if loss_pretrain is not None:
            loss = loss + loss_pretrain
        hip.gemm_workspace_check(logits.device) if logits.is_cuda else None   # non-blocking (see hip.gemm_workspace_check)
        res = Output(loss=loss, logits=logits, past_key_values=None, hidden_states=outputs.hidden_states, attentions=None)
        object.__setattr__(res, "loss_mvm", loss_pretrain)   # not a key (integer indexing stays HF's): the MVM term on its own, for tests / logging
        return res

    @torch.no_grad()
    def generate(self, inputs_embeds=None, max_new_tokens=16, num_beams=1, do_sample=False, stopping_criteria=None,
                 attention_mask=None, use_cache=True, min_length=0, top_p=1.0, repetition_penalty=1.0, length_penalty=1.0,
                 temperature=1.0, eos_token_id=2, pad_token_id=0, generator=None, **unused):
        """`llama_model.generate(inputs_embeds=..., ...)` as Chat.answer calls it (conversation.py:231-243; demo.py runs
        num_beams=5, do_sample=False): HIP prefill of `inputs_embeds` into a KV cache, then one decode step per token with
        HF's greedy / sampling / beam-search bookkeeping restated in stllm_amd/generation.py.  eos / pad default to the
        Vicuna generation config (2 / 0).  A padded batch (attention_mask rows of different valid lengths, left- or right-padded) is generated by length groups
        (see below).  Returns the generated ids [B, n] (the prompt has no ids)."""
        from .. import generation
        if attention_mask is not None and inputs_embeds.shape[0] > 1:
            # Ragged prompts (round 5).  The device KV cache holds equal-length rows (one position counter, RoPE by row index), and the reference itself only
            # ever sends one prompt x beams (conversation.py:231-243) — so a padded batch is served by LENGTH GROUPS: HF derives position_ids from the mask
            # (positions count real tokens only), i.e. every row generates exactly as its unpadded prompt would alone; rows of equal length share one
            # batched call, the results are re-assembled in row order, padded with pad_token_id like HF's finished rows.  Greedy and beam search are
            # deterministic: identical to per-row generation; sampling draws per group, not per batch (a different but equally valid stream).
            m = hip.host_mask(attention_mask).to(torch.bool)
            lens = m.sum(dim=1).tolist()
            if len(set(lens)) > 1 or not bool(m.all()):
                left = [bool(m[b, -1]) and not bool(m[b, 0]) for b in range(m.shape[0])]      # left-padded rows keep their LAST tokens
                groups = {}
                for b, n in enumerate(lens):
                    groups.setdefault(n, []).append(b)
                outs = [None] * m.shape[0]
                for n, rows in sorted(groups.items()):
                    emb = torch.stack([inputs_embeds[b, m.shape[1] - n:] if left[b] else inputs_embeds[b, :n] for b in rows], dim=0)
                    ids = self.generate(inputs_embeds=emb, max_new_tokens=max_new_tokens, num_beams=num_beams, do_sample=do_sample, stopping_criteria=stopping_criteria,
                                        use_cache=use_cache, min_length=min_length, top_p=top_p, repetition_penalty=repetition_penalty, length_penalty=length_penalty,
                                        temperature=temperature, eos_token_id=eos_token_id, pad_token_id=pad_token_id, generator=generator)
                    for j, b in enumerate(rows):
                        outs[b] = ids[j]
                width = max(o.shape[0] for o in outs)
                res = torch.full((len(outs), width), pad_token_id, dtype=torch.long, device=outs[0].device)
                for b, o in enumerate(outs):
                    res[b, : o.shape[0]] = o
                return res
        return generation.generate(self, inputs_embeds, max_new_tokens=max_new_tokens, num_beams=num_beams, do_sample=do_sample,
                                   min_length=min_length, top_p=top_p, temperature=temperature,
                                   repetition_penalty=repetition_penalty, length_penalty=length_penalty,
                                   stopping_criteria=stopping_criteria, eos_token_id=eos_token_id, pad_token_id=pad_token_id,
                                   use_cache=use_cache, generator=generator)

    @classmethod
    def get_state_dict(cls, path, prefix="pytorch_model"):
        """st_llm.py:149-158: merge the `pytorch_model-XXXXX-of-YYYYY.bin` shards of a HF directory.  Beyond the reference:
        `model-XXXXX-of-YYYYY.safetensors` shards (what current HF exports write) are merged the same way."""
        pattern = re.compile(f"{prefix}-(\\d+)-of-(\\d+).bin")
        sd = {}
        for fn in sorted(f for f in os.listdir(path) if pattern.match(f)):
            sd.update(torch.load(os.path.join(path, fn), map_location="cpu"))
        st_pattern = re.compile(r"model-(\d+)-of-(\d+)\.safetensors")
        st_files = sorted(f for f in os.listdir(path) if st_pattern.match(f))
        if st_files and not sd:
            from safetensors.torch import load_file
            for fn in st_files:
                sd.update(load_file(os.path.join(path, fn), device="cpu"))
        return sd

    @classmethod
    def from_config(cls, cfg, device=None):
        """st_llm.py:160-203.  ``llama_model``: a dict of LlamaConfig fields / "" (Vicuna-7B dims) => parameters
        left for the caller to fill (random-init benchmarks, BASELINE.json); a directory => HF sharded weights."""
        llama_model = cfg.get("llama_model", "")
        lcfg, sd = StllmConfig(), None
        if isinstance(llama_model, dict):
            lcfg = StllmConfig(**llama_model)
        elif llama_model and os.path.isdir(llama_model):
            import json
            with open(os.path.join(llama_model, "config.json")) as f:
                lcfg = StllmConfig(**{k: v for k, v in json.load(f).items() if k in
                                      ("hidden_size", "intermediate_size", "num_hidden_layers", "num_attention_heads",
                                       "vocab_size", "rms_norm_eps", "max_position_embeddings")})
            sd = cls.get_state_dict(llama_model)
        if cfg.get("lora_r", 0) > 0:
            raise NotImplementedError("LoRA (peft) is not on the hot path: no shipped config sets lora_r")
        model = cls(lcfg, device=device)
        if sd:
            model.load_state_dict(sd, strict=False)
        model.get_model().initialize_vision_modules(cfg, device=device)
        if cfg.get("qformer_text_input", False):   # st_llm.py:180-181: lm_head follows the input table (32001 rows with '[PAD]')
            model.resize_token_embeddings(model.config.vocab_size)
        ckpt_path = cfg.get("ckpt", "")
        if ckpt_path and os.path.exists(ckpt_path):
            ckpt = cls.get_state_dict(ckpt_path) if os.path.isdir(ckpt_path) else torch.load(ckpt_path, map_location="cpu")
            ckpt = ckpt.get("model", ckpt)
            if "llm_proj.weight" in ckpt:
                ckpt["llama_proj.weight"] = ckpt.pop("llm_proj.weight")
                ckpt["llama_proj.bias"] = ckpt.pop("llm_proj.bias")
            rows = ckpt.get("model.embed_tokens.weight")
            if rows is not None and rows.shape[0] != model.config.vocab_size:
                # a checkpoint trained with a different vocabulary (e.g. saved by a tokenizer without '[PAD]'): strict=False does
                # not skip shape mismatches, so follow the checkpoint instead of failing in load_state_dict
                model.resize_token_embeddings(rows.shape[0])
            model.load_state_dict(ckpt, strict=False)
        return model
