"""Reference parameter names & shapes for the hot path — test infrastructure.

Lets tests build a full-width state dict for the oracle from ``stllm_amd.synth`` without
instantiating either the reference or the product modules.  tests/test_param_names.py checks the
product's nn.Modules expose exactly these names/shapes (checkpoint drop-in compatibility).
"""


def vit_shapes(depth=39, p="visual_encoder."):
    """eva_vit.py:76-82,115,157-165,196,263-265."""
    s = {p + "cls_token": (1, 1, 1408), p + "pos_embed": (1, 257, 1408),
         p + "patch_embed.proj.weight": (1408, 3, 14, 14), p + "patch_embed.proj.bias": (1408,)}
    for i in range(depth):
        s.update(vit_block_shapes(f"{p}blocks.{i}."))
    return s


def vit_block_shapes(b):
    return {b + "norm1.weight": (1408,), b + "norm1.bias": (1408,),
            b + "attn.q_bias": (1408,), b + "attn.v_bias": (1408,),
            b + "attn.qkv.weight": (4224, 1408),
            b + "attn.proj.weight": (1408, 1408), b + "attn.proj.bias": (1408,),
            b + "norm2.weight": (1408,), b + "norm2.bias": (1408,),
            b + "mlp.fc1.weight": (6144, 1408), b + "mlp.fc1.bias": (6144,),
            b + "mlp.fc2.weight": (1408, 6144), b + "mlp.fc2.bias": (1408,)}


def btadapter_shapes(depth=39, adapter_depth=3, p="visual_encoder."):
    """eva_btadapter.py:81-84, 257-292."""
    s = vit_shapes(depth, p)
    s[p + "BTAdapter_cls"] = (1, 1, 1408)
    s[p + "BTAdapter_position.weight"] = (64, 1408)
    for j in range(adapter_depth):
        s.update(vit_block_shapes(f"{p}BTAdapter_S.{j}."))
        t = f"{p}BTAdapter_T.{j}."
        s.update({t + "attn.q_bias": (1408,), t + "attn.v_bias": (1408,), t + "attn.qkv.weight": (4224, 1408),
                  t + "attn.proj.weight": (1408, 1408), t + "attn.proj.bias": (1408,),
                  t + "norm1.weight": (1408,), t + "norm1.bias": (1408,),
                  t + "temporal_fc.weight": (1408, 1408), t + "temporal_fc.bias": (1408,)})
    return s


def qformer_shapes(layers=12, text=True, vocab=30523, p="Qformer.bert."):
    """Qformer.py:56-65,127-133,281-282,352,367-368,384-400 (cls=None, st_llm.py:288)."""
    s = {p + "embeddings.LayerNorm.weight": (768,), p + "embeddings.LayerNorm.bias": (768,)}
    if text:
        s[p + "embeddings.word_embeddings.weight"] = (vocab, 768)
        s[p + "embeddings.position_embeddings.weight"] = (512, 768)

    def attn(a, kv):
        return {a + "self.query.weight": (768, 768), a + "self.query.bias": (768,),
                a + "self.key.weight": (768, kv), a + "self.key.bias": (768,),
                a + "self.value.weight": (768, kv), a + "self.value.bias": (768,),
                a + "output.dense.weight": (768, 768), a + "output.dense.bias": (768,),
                a + "output.LayerNorm.weight": (768,), a + "output.LayerNorm.bias": (768,)}

    def ffn(i, o):
        return {i + "dense.weight": (3072, 768), i + "dense.bias": (3072,),
                o + "dense.weight": (768, 3072), o + "dense.bias": (768,),
                o + "LayerNorm.weight": (768,), o + "LayerNorm.bias": (768,)}
    for i in range(layers):
        lp = f"{p}encoder.layer.{i}."
        s.update(attn(lp + "attention.", 768))
        if i % 2 == 0:
            s.update(attn(lp + "crossattention.", 1408))
        if text:
            s.update(ffn(lp + "intermediate.", lp + "output."))
        s.update(ffn(lp + "intermediate_query.", lp + "output_query."))
    return s


def stllm_model_shapes(vit_depth=39, qf_layers=12, text=False, video_input=None, mvm_decode=False,
                       vit_model="eva_clip_g", adapter_depth=3, qf_vocab=30523, p="model.stllm_model.", has_qformer=True):
    """STLLMModel parameters (st_llm.py:240-250, 254, 273, 314) under the prefix set at st_llm.py:51."""
    s = {}
    s.update(vit_shapes(vit_depth, p + "visual_encoder.") if vit_model == "eva_clip_g"
             else btadapter_shapes(vit_depth, adapter_depth, p + "visual_encoder."))
    s[p + "ln_vision.weight"] = (1408,)
    s[p + "ln_vision.bias"] = (1408,)
    if has_qformer:
        s[p + "query_tokens"] = (1, 32, 768)
        s.update(qformer_shapes(qf_layers, text, qf_vocab, p + "Qformer.bert."))
    s[p + "llama_proj.weight"] = (4096, 768 if has_qformer else 4 * 1408)   # st_llm.py:299-301, 314
    s[p + "llama_proj.bias"] = (4096,)
    if video_input == "residual":
        s.update({p + "down_proj.weight": (1024, 4096), p + "down_proj.bias": (1024,),
                  p + "up_proj.weight": (4096, 1024), p + "up_proj.bias": (4096,)})
    if mvm_decode:
        s.update({p + "mvm_decoder.head.weight": (4096, 4096), p + "mvm_decoder.head.bias": (4096,),
                  p + "mvm_decoder.norm.weight": (4096,), p + "mvm_decoder.norm.bias": (4096,)})
    return s


def llama_shapes(layers=32, hidden=4096, inter=11008, vocab=32000, p="model."):
    """HF naming, st_llm.py:104-108; spec modeling_llama_mem.py:163-166,138-140,261-262,444-446,602."""
    s = {p + "embed_tokens.weight": (vocab, hidden), p + "norm.weight": (hidden,),
         "lm_head.weight": (vocab, hidden)}
    for i in range(layers):
        lp = f"{p}layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "o_proj"):
            s[lp + f"self_attn.{n}.weight"] = (hidden, hidden)
        s[lp + "mlp.gate_proj.weight"] = (inter, hidden)
        s[lp + "mlp.up_proj.weight"] = (inter, hidden)
        s[lp + "mlp.down_proj.weight"] = (hidden, inter)
        s[lp + "input_layernorm.weight"] = (hidden,)
        s[lp + "post_attention_layernorm.weight"] = (hidden,)
    return s
