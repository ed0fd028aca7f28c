"""Blip2Base of the reference (stllm/models/blip2.py): builds ViT / Q-Former / tokenizer.

No network here: ``init_vision_encoder`` / ``init_Qformer`` construct the architectures with uninitialised
parameters (filled by ``load_state_dict`` or ``stllm_amd.synth``); ``init_tokenizer`` returns the offline
IdTokenizer unless a real HF tokenizer directory is given.  ``maybe_autocast`` (blip2.py:36-44) has no
counterpart: the numerics mode is process-wide (``stllm_amd.runtime``).
"""
import contextlib

import torch
import torch.nn as nn

from ..tokenizer import IdTokenizer
from .eva_vit import create_eva_vit_g
from .layers import LayerNorm as _LN
from .layers import _dev
from .Qformer import BertConfig, BertLMHeadModel


class LayerNorm(_LN):
    """blip2.LayerNorm (blip2.py:103-109): fp32 LayerNorm, default eps 1e-5."""


def disabled_train(self, mode=True):
    return self


class BaseModel(nn.Module):
    @property
    def device(self):
        return list(self.parameters())[-1].device


class Blip2Base(BaseModel):
    vit_depth = 39          # overridable for reduced-depth parity tests
    qformer_layers = 12
    bt_adapter_depth = 3

    @classmethod
    def init_tokenizer(cls, truncation_side="right"):
        return IdTokenizer(pad_token_id=0, bos_token_id=101, eos_token_id=102, vocab_size=30523)

    def maybe_autocast(self, dtype=None):
        return contextlib.nullcontext()

    @classmethod
    def init_Qformer(cls, num_query_token, vision_width, cross_attention_freq=2, device=None):
        cfg = BertConfig(encoder_width=vision_width, add_cross_attention=True, cross_attention_freq=cross_attention_freq,
                         query_length=num_query_token, num_hidden_layers=cls.qformer_layers)
        q = BertLMHeadModel(cfg, device=device)
        query_tokens = nn.Parameter(torch.empty(1, num_query_token, cfg.hidden_size, device=_dev(device)), requires_grad=False)
        return q, query_tokens

    @classmethod
    def init_vision_encoder(cls, model_name, img_size, drop_path_rate, use_grad_checkpoint, precision, device=None):
        assert model_name in ["eva_clip_g", "eva_btadapter_g"], "vit model must be eva_clip_g for current version of MiniGPT-4"
        if model_name == "eva_clip_g":
            v = create_eva_vit_g(img_size, drop_path_rate, use_grad_checkpoint, precision, depth=cls.vit_depth, device=device)
        else:
            from .eva_btadapter import create_eva_btadapter
            v = create_eva_btadapter(precision, depth=cls.vit_depth, adapter_depth=cls.bt_adapter_depth, device=device)
        return v, LayerNorm(v.num_features, device=device)

    def load_from_pretrained(self, url_or_filename):
        import os
        if not (url_or_filename and os.path.isfile(url_or_filename)):
            return None  # offline: nothing to download; parameters come from load_state_dict / synth
        ckpt = torch.load(url_or_filename, map_location="cpu")
        return self.load_state_dict(ckpt["model"], strict=False)
