#!/usr/bin/env python3
"""Which stage of the path makes the fast numerics modes miss the 1e-2 logits bar?  

Runs BASELINE config 2 at full size (bench.make_samples(1, 16)) with the compute dtype chosen PER STAGE — ViT, Q-Former
(+ ln_vision + projector), Llama (+ lm_head) — and prints the logits error against the reference's own CPU fp32 forward
(tests/golden/c2_full.npz).  GPU only.

    python tools/parity_ladder.py [--modes fp16,bf16] [--extra bf16x3/bf16x3/bf16x3,bf16x3/fp32/fp32]
(stage modes: fp32 | fp16 | bf16 | bf16x3 = the split verify mode of round 4; every row also prints the step time of that combination)
"""
import argparse
import contextlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--modes", default="fp16,bf16")
    ap.add_argument("--extra", default="", help="comma list of extra (vit,qf,llm) triples, e.g. fp16/fp16/fp32")
    a = ap.parse_args()
    torch.set_grad_enabled(False)
    from stllm_amd import runtime
    g = np.load(os.path.join(ROOT, "tests", "golden", "c2_full.npz"))
    args = argparse.Namespace(vit_depth=39, qformer_layers=12, llm_layers=32)
    model = bench.build_model(torch.device("cuda:0"), args)
    sm = model.model.stllm_model
    samples = bench.make_samples(1, 16, "cuda:0")
    stage = {"vit": "fp32", "qf": "fp32", "llm": "fp32"}

    def wrap(fn, key):
        def inner(*x, **k):
            with runtime.use_dtype(stage[key]):
                return fn(*x, **k)
        return inner
    sm.visual_encoder.forward_features_flat = wrap(sm.visual_encoder.forward_features_flat, "vit")
    enc0 = sm.encode_img
    sm.encode_img = wrap(enc0, "qf")                      # ln_vision, Q-Former, projector take encode_img's dtype; the ViT call inside overrides it

    def run(v, q, l):
        stage.update(vit=v, qf=q, llm=l)
        for m in (sm.visual_encoder, sm.Qformer.bert, model.model):
            m.repack()
        model._lm_packed = {}
        import time
        with runtime.use_dtype(l):
            out = model(samples=samples)          # packs the weights of the three stages
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                out = model(samples=samples)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 3 * 1e3
        lg = out.logits[0].float().cpu()
        err = float(np.abs(lg[::3, ::499].numpy() - g["logits_slice"]).max())
        agree = float((lg.argmax(-1).numpy() == g["top_ids"][:, 0]).mean())
        print(f"ViT {v:6s} Q-Former {q:6s} Llama {l:6s}: logits max-abs err {err:.3e}  top-1 {agree:.4f}  loss err {abs(out.loss.item() - float(g['loss'][0])):.2e}  {ms:7.2f} ms/step", flush=True)

    run("fp32", "fp32", "fp32")
    for m in a.modes.split(","):
        run(m, "fp32", "fp32")
        run("fp32", m, "fp32")
        run("fp32", "fp32", m)
        run(m, m, m)
    for t in [t for t in a.extra.split(",") if t]:
        run(*t.split("/"))


if __name__ == "__main__":
    main()
