"""Training-step timing on one MI355X: forward(samples) with the activation tape + explicit backward + AdamW.

    python tools/train_bench.py [--layers 32] [--batch 4] [--frames 16] [--steps 3] [--dtype bf16]

Prints per-phase wall times (HIP events on the current stream) and the tokens/s of the LLM part; use under
`rocprofv3 --kernel-trace --stats` for the per-kernel split.  Random-init weights, synthetic frames and ids."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--vit-depth", type=int, default=39)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--backbone", default="eva_clip_g", choices=["eva_clip_g", "eva_btadapter_g"],
                    help="eva_btadapter_g = the reference's main training config: the BTAdapter* parameters train too")
    ap.add_argument("--no-sink", action="store_true", help="gradients as stand-alone tensors, copied by AdamW.step (the pre-round-4 path)")
    ap.add_argument("--text", action="store_true", help="Q-Former text input (instructblip_* model types)")
    a = ap.parse_args()
    from stllm_amd import runtime, synth, training
    from stllm_amd.models import st_llm
    from stllm_amd.models.blip2 import Blip2Base
    Blip2Base.vit_depth = a.vit_depth
    cfg = dict(vit_model=a.backbone, image_size=224, num_query_token=32, video_input="all", use_mask=True, mvm_decode=True,
               qformer_text_input=a.text, max_txt_len=64, end_sym=" 2", llama_model=dict(num_hidden_layers=a.layers))
    model = st_llm.STLLMForCausalLM.from_config(cfg, device="cuda")
    synth.fill_module_(model, 0, "")
    g = torch.Generator().manual_seed(0)
    ids = lambda n: " ".join(str(int(x)) for x in torch.randint(3, 32000, (n,), generator=g))
    samples = {"image": torch.randn(a.batch, a.frames, 3, 224, 224, device="cuda"),
               "instruction_input": [f"{ids(7)}<ImageHere>{ids(24)}" + (f" Human: {ids(12)} ###" if a.text else "") for _ in range(a.batch)], "answer": [ids(31) for _ in range(a.batch)]}
    opt = training.AdamW(list(training.trainable_parameters(model)), lr=2e-5)
    with runtime.use_dtype(a.dtype):
        for step in range(a.steps + 1):
            torch.cuda.synchronize()
            t0 = time.time()
            training.PHASES = []
            loss, loss_mvm, grads = training.loss_and_grads(model, samples, sink=None if a.no_sink else opt.grad_sink())
            torch.cuda.synchronize()
            t1 = time.time()
            marks, training.PHASES = training.PHASES, None
            print("   " + "  |  ".join(f"{n}: {a.elapsed_time(b):.1f} ms" for (_, a), (n, b) in zip(marks[:-1], marks[1:])), flush=True)
            norm = opt.step(grads)
            training.invalidate_packed(model)
            torch.cuda.synchronize()
            t2 = time.time()
            print(f"step {step}: loss {loss.item():.4f} mvm {None if loss_mvm is None else round(loss_mvm.item(), 4)} |g| {norm:.3f}  "
                  f"fwd+bwd {1e3 * (t1 - t0):.1f} ms  optimizer {1e3 * (t2 - t1):.1f} ms  "
                  f"mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)


if __name__ == "__main__":
    main()
