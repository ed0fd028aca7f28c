"""Build-container-only loader for the *reference* ST-LLM model code.

Used ONLY by ``tests/golden/make_fixtures.py`` (golden-vector generation) in the
container that has ``/root/reference``.  Nothing here is shipped to or executed on
the GPU box: ``-m gpu`` tests, ``smoke()`` and ``bench.py`` never import this file.

The reference package cannot be imported as-is here (its ``__init__`` pulls
omegaconf / webdataset / timm / peft, none installed), so the model files are loaded
by path under a skeleton ``stllm`` package with the minimum stubs of SURVEY.md §8c:

  * timm.models.layers.{drop_path,to_2tuple,trunc_normal_}, timm.models.registry,
    timm.models.hub                                (eva_vit.py:15-16, dist_utils.py:14)
  * transformers.modeling_utils.{apply_chunking_to_forward, prune_linear_layer,
    find_pruneable_heads_and_indices}              (Qformer.py:39-44)
  * BertPreTrainedModel.init_weights / get_head_mask (removed in HF 5; Qformer.py:697,935)
  * omegaconf.OmegaConf, peft.*                    (base_model.py:16, st_llm.py:26-29)
  * stllm.common.utils.{is_url,get_abs_path}       (real one needs iopath/torchvision)

No reference source text is copied: modules are exec'd from where they lie.
"""
import importlib.util
import math
import os
import sys
import types

import torch
import torch.nn as nn

REF = os.environ.get("STLLM_REFERENCE", "/root/reference")


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _load(name, relpath):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, relpath))
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def _install_stubs():
    # transformers probes optional packages with importlib.find_spec at import time: import it
    # (and the submodules the reference touches) BEFORE the spec-less stub modules exist.
    import transformers  # noqa: F401
    import transformers.modeling_utils  # noqa: F401
    import transformers.models.llama.modeling_llama  # noqa: F401
    import transformers.models.bert.configuration_bert  # noqa: F401
    from transformers import LlamaTokenizer, BertTokenizer  # noqa: F401
    # ---- timm ---------------------------------------------------------------
    def to_2tuple(x):
        return tuple(x) if isinstance(x, (tuple, list)) else (x, x)

    def drop_path(x, drop_prob=0.0, training=False):
        assert drop_prob == 0.0 or not training
        return x

    def trunc_normal_(t, mean=0.0, std=1.0, a=-2.0, b=2.0):
        return nn.init.trunc_normal_(t, mean=mean, std=std, a=a, b=b)

    _mod("timm")
    _mod("timm.models")
    _mod("timm.models.layers", drop_path=drop_path, to_2tuple=to_2tuple, trunc_normal_=trunc_normal_)
    _mod("timm.layers", drop_path=drop_path, to_2tuple=to_2tuple, trunc_normal_=trunc_normal_)
    _mod("timm.models.registry", register_model=lambda f: f)
    _mod("timm.models.hub", download_cached_file=None, get_cache_dir=lambda *a, **k: "/tmp")

    # ---- transformers helpers that moved -------------------------------------
    import transformers.modeling_utils as mu
    import transformers.pytorch_utils as pu
    for n in ("apply_chunking_to_forward", "prune_linear_layer"):
        if not hasattr(mu, n):
            setattr(mu, n, getattr(pu, n))
    if not hasattr(mu, "find_pruneable_heads_and_indices"):
        mu.find_pruneable_heads_and_indices = getattr(pu, "find_pruneable_heads_and_indices", lambda *a, **k: None)

    # ---- omegaconf / peft ----------------------------------------------------
    _mod("omegaconf", OmegaConf=type("OmegaConf", (), {}))
    _mod("peft", LoraConfig=object, get_peft_model=lambda m, c: m)
    _mod("peft.utils", PeftType=type("PeftType", (), {}))
    _mod("peft.peft_model", PeftModelForCausalLM=type("PeftModelForCausalLM", (), {}))


def load_reference():
    """Returns a namespace with the reference's model modules (eva, qf, bt, st, blip2, utils)."""
    if "stllm.models.st_llm" in sys.modules:
        return _ns()
    _install_stubs()
    pkg = _mod("stllm"); pkg.__path__ = [os.path.join(REF, "stllm")]
    common = _mod("stllm.common"); common.__path__ = [os.path.join(REF, "stllm/common")]
    models = _mod("stllm.models"); models.__path__ = [os.path.join(REF, "stllm/models")]
    _load("stllm.common.registry", "stllm/common/registry.py")
    _mod("stllm.common.utils", is_url=lambda s: False, get_abs_path=lambda p: p)
    _load("stllm.common.dist_utils", "stllm/common/dist_utils.py")
    _load("stllm.common.logger", "stllm/common/logger.py")
    bm = _load("stllm.models.base_model", "stllm/models/base_model.py")
    models.BaseModel = bm.BaseModel
    _load("stllm.models.utils", "stllm/models/utils.py")
    _load("stllm.models.eva_vit", "stllm/models/eva_vit.py")
    qf = _load("stllm.models.Qformer", "stllm/models/Qformer.py")
    # HF 5 removed these two PreTrainedModel helpers the reference relies on
    if not hasattr(qf.BertPreTrainedModel, "init_weights") or True:
        qf.BertPreTrainedModel.init_weights = lambda self: self.apply(self._init_weights)
    qf.BertModel.get_head_mask = lambda self, head_mask, n, *a, **k: [None] * n
    _load("stllm.models.eva_btadapter", "stllm/models/eva_btadapter.py")
    _load("stllm.models.blip2", "stllm/models/blip2.py")
    _load("stllm.models.peft_model", "stllm/models/peft_model.py")
    _load("stllm.models.st_llm", "stllm/models/st_llm.py")
    return _ns()


def _ns():
    return types.SimpleNamespace(
        eva=sys.modules["stllm.models.eva_vit"],
        qf=sys.modules["stllm.models.Qformer"],
        bt=sys.modules["stllm.models.eva_btadapter"],
        blip2=sys.modules["stllm.models.blip2"],
        st=sys.modules["stllm.models.st_llm"],
        utils=sys.modules["stllm.models.utils"],
    )


# ----------------------------------------------------------------------------------
# Construction helpers that bypass network / weight downloads
# ----------------------------------------------------------------------------------
def build_ref_vit(depth=39):
    """EVA-CLIP-g exactly as eva_vit.py:416-428 (random init, fp32), with `depth` blocks."""
    from functools import partial
    ref = load_reference()
    return ref.eva.VisionTransformer(
        img_size=224, patch_size=14, use_mean_pooling=False, embed_dim=1408, depth=depth,
        num_heads=1408 // 88, mlp_ratio=4.3637, qkv_bias=True, drop_path_rate=0.0,
        norm_layer=partial(nn.LayerNorm, eps=1e-6), use_checkpoint=False).eval()


def build_ref_qformer(num_query_token=32, vision_width=1408, num_layers=12, text=True, vocab=30523,
                      keep_cls=False):
    """blip2.py:46-59 with BertConfig() defaults (== bert-base-uncased)."""
    ref = load_reference()
    cfg = ref.qf.BertConfig()
    cfg.num_hidden_layers = num_layers
    cfg.encoder_width = vision_width
    cfg.add_cross_attention = True
    cfg.cross_attention_freq = 2
    cfg.query_length = num_query_token
    cfg.vocab_size = vocab
    q = ref.qf.BertLMHeadModel(config=cfg)
    query_tokens = nn.Parameter(torch.zeros(1, num_query_token, cfg.hidden_size))
    query_tokens.data.normal_(mean=0.0, std=cfg.initializer_range)
    if not text:  # st_llm.py:277-283
        q.bert.embeddings.word_embeddings = None
        q.bert.embeddings.position_embeddings = None
        for layer in q.bert.encoder.layer:
            layer.output = None
            layer.intermediate = None
    if not keep_cls:  # st_llm.py:288 (the reference drops the LM head itself when built through STLLMModel)
        q.cls = None
    return q.eval(), query_tokens


class FakeTokenizer:
    """Deterministic stand-in for LlamaTokenizer/BertTokenizer: a 'text' is a string of
    space-separated integer ids.  Lets the reference's own prompt_wrap / forward run on
    fixed token-id arrays."""

    def __init__(self, pad_token_id=0, bos_token_id=1, eos_token="2", bos=False):
        self.pad_token_id = pad_token_id
        self.bos_token_id = bos_token_id
        self.eos_token = " " + eos_token
        self.padding_side = "right"
        self.pad_token = None
        self._bos = bos

    def __len__(self):
        return 32000

    def add_special_tokens(self, d):
        return 0

    def _ids(self, s, add_special_tokens):
        ids = [int(t) for t in s.split() if t.isdigit()]  # non-numeric words (role tags) carry no ids
        if add_special_tokens:
            ids = [self.bos_token_id] + ids
        return ids

    def __call__(self, text, return_tensors="pt", add_special_tokens=True, padding=None,
                 truncation=False, max_length=None, **kw):
        single = isinstance(text, str)
        texts = [text] if single else list(text)
        rows = [self._ids(t, add_special_tokens) for t in texts]
        if truncation and max_length is not None:
            rows = [r[:max_length] for r in rows]
        L = max(len(r) for r in rows) if rows else 0
        ids = torch.full((len(rows), L), self.pad_token_id, dtype=torch.long)
        att = torch.zeros((len(rows), L), dtype=torch.long)
        for i, r in enumerate(rows):
            ids[i, :len(r)] = torch.tensor(r, dtype=torch.long)
            att[i, :len(r)] = 1
        out = types.SimpleNamespace(input_ids=ids, attention_mask=att)
        out.to = lambda dev: out
        return out
