"""__graft_entry__.smoke(): one tiny end-to-end invocation of the hot path on cuda:0, checked against the
CPU oracle (the oracle is only the checker here)."""
import torch


def run(verbose=True):
    import shapes
    import stllm_oracle as O
    from stllm_amd import runtime, synth
    from stllm_amd.models import st_llm
    from stllm_amd.models.blip2 import Blip2Base

    torch.set_grad_enabled(False)
    Blip2Base.vit_depth, Blip2Base.qformer_layers = 1, 2
    cfg = dict(vit_model="eva_clip_g", image_size=224, num_query_token=32, llama_model=dict(num_hidden_layers=1),
               video_input="all", use_mask=False, mvm_decode=False, qformer_text_input=False, max_txt_len=32, end_sym=" 2")
    try:
        model = st_llm.STLLMForCausalLM.from_config(cfg, device="cuda:0")
    finally:
        Blip2Base.vit_depth, Blip2Base.qformer_layers = 39, 12
    synth.fill_module_(model, 0, "")
    frames = synth.normal_(torch.empty(1, 2, 3, 224, 224), "input.video", 0, 1.0)
    samples = {"image": frames.cuda(), "instruction_input": ["5 6 7 8 9 10 11<ImageHere>12 13 14"], "answer": ["20 21 22"]}
    sd = synth.state_dict_from_shapes({**shapes.stllm_model_shapes(1, 2, False, "all", False), **shapes.llama_shapes(1)}, 0)
    ref = O.stllm_forward({"image": frames, "before_ids": [[5, 6, 7, 8, 9, 10, 11]], "after_ids": [[12, 13, 14]],
                           "answer_ids": [[20, 21, 22, 2]]}, sd,
                          dict(cfg, pad_id=0, bos_id=1))
    scale = ref["logits"].abs().max().item()
    for mode, tol in (("fp32", 1e-3), ("bf16", 0.25), ("fp16", 0.05)):
        with runtime.use_dtype(mode):
            for m in (model.model.stllm_model.visual_encoder, model.model.stllm_model.Qformer.bert, model.model):
                m.repack()
            out = model(samples=samples)
        torch.cuda.synchronize()
        err = (out.logits.cpu() - ref["logits"]).abs().max().item()
        lerr = abs(out.loss.item() - ref["loss"].item())
        if verbose:
            print(f"[smoke] {mode}: logits max-abs err {err:.3e} (abs-max {scale:.2f}), loss err {lerr:.2e}")
        assert err <= tol * max(1.0, scale), f"smoke {mode}: logits err {err} > {tol}"
    # the tiny model's GEMMs stay on the 128x128 kernels: push one small problem through the phased split-K kernel as well
    from stllm_amd import hip
    a = synth.normal_(torch.empty(300, 256), "smoke.a", 0, 1.0).to(torch.bfloat16).cuda()
    w = synth.normal_(torch.empty(512, 256), "smoke.w", 0, 0.05).to(torch.bfloat16).cuda()
    hip.set_option("gemm_p8", 3)
    try:
        got = hip.gemm(a, w, dtype="bf16", out_f32=True)
    finally:
        hip.set_option("gemm_p8", -1)
    assert hip.lib().stllm_last_kernel().decode().startswith("gemm_p8_kernel<")
    gerr = (got.cpu() - a.float().cpu() @ w.float().cpu().t()).abs().max().item()
    if verbose:
        print(f"[smoke] phased GEMM 300x512x256 (K split across one XCD's workgroups): max-abs err {gerr:.2e}")
    assert gerr < 1e-4 and hip.gemm_workspace_ok(), hip.lib().stllm_last_error().decode()
    return True
