#!/usr/bin/env python3
"""Decode latency of the KV-cache path: prefill S=576, then one-token steps.  GPU only.
HBM floor for Vicuna-7B in bf16: 13.5 GB of weights per token / ~6.3 TB/s achievable = 2.1 ms."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--llm-layers", type=int, default=32)
ap.add_argument("--vit-depth", type=int, default=1)
ap.add_argument("--qformer-layers", type=int, default=1)
ap.add_argument("--tokens", type=int, default=16)
ap.add_argument("--rows", type=int, default=1, help="sequences decoded together (5 = demo.py's beam search)")
ap.add_argument("--gemv", type=int, default=-1, help="stllm_set_option('gemm_gemv'): -1 default (M <= 8), 1 = M <= 4, 0 off")
ap.add_argument("--attn-single", type=int, default=1, help="stllm_set_option('attn_decode_single')")
args = ap.parse_args()
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
model = bench.build_model(dev, args)
lm = model.model
S = 576
from stllm_amd import hip
hip.set_option("gemm_gemv", args.gemv)
hip.set_option("attn_decode_single", args.attn_single)
R = args.rows
emb = (torch.randn(1, S, 4096, device=dev) * 0.02).expand(R, S, 4096).contiguous()
cache = lm.new_cache(R, S + args.tokens + 8, dev)
hidden, h16 = lm.prefill(emb, None, cache=cache)
tok = torch.randn(R, 1, 4096, device=dev) * 0.02
for _ in range(3):
    lm.decode_step(tok, cache)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.tokens):
    _, h = lm.decode_step(tok, cache)
    logits = model.logits_from(h, R, 1)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / args.tokens * 1e3
wbytes = sum(p.numel() for n, p in lm.named_parameters() if "layers" in n) * 2 + 32000 * 4096 * 2
print(f"decode ({R} rows, gemm_gemv {args.gemv}, attn_single {args.attn_single}, fuse_norm_rows {os.environ.get('STLLM_DECODE_FUSE_ROWS', '2')}): {ms:.2f} ms/step ({1e3 / ms:.1f} tok/s), weights streamed per token {wbytes / 1e9:.2f} GB => {wbytes / ms / 1e9:.2f} TB/s")
