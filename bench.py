#!/usr/bin/env python3
"""bench.py — video-tokens/sec through ViT + Q-Former + projector + Vicuna-7B prefill (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--dtype bf16|fp16|fp32] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path over one batch of synthetic input: per GPU one clip of T=16 random
224x224 frames (already resident in HBM) -> EVA-CLIP-g (39 blocks) -> ln_vision -> Q-Former (12 layers) ->
llama_proj -> 'all' pooling (512 video tokens) -> token-block assembly (BOS + 7 + 512 + 40 + 16 = 576
positions) -> Vicuna-7B (32 layers) prefill -> lm_head on all positions -> shifted CE.  Random-init weights
(stllm_amd.synth), full sizes: BASELINE.json configs[1].  N > 1: weak scaling — N clips; the B*T frames are
sharded over the ranks (frame-parallel), ONE RCCL all-gather of the visual tokens, clip c prefilled on rank c.

Prints ONE JSON line (rank 0).  `value` = total video tokens (B * 512) / step time, inputs resident in HBM.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

MFMA_PEAK = {"bf16": 2500.0, "fp16": 2500.0, "fp32": 157.3}  # dense TFLOP/s, MI355X_MICROARCH.md


def build_model(device, args):
    from stllm_amd import synth
    from stllm_amd.models import st_llm
    from stllm_amd.models.blip2 import Blip2Base
    Blip2Base.vit_depth, Blip2Base.qformer_layers = args.vit_depth, args.qformer_layers
    cfg = dict(vit_model="eva_clip_g", image_size=224, num_query_token=32, video_input="all", use_mask=False,
               mvm_decode=False, qformer_text_input=False, max_txt_len=32, end_sym=" 2",
               llama_model=dict(num_hidden_layers=args.llm_layers))
    model = st_llm.STLLMForCausalLM.from_config(cfg, device=device)
    synth.fill_module_(model, 0, "")
    return model.eval()


def make_samples(B, T, device, seed=0):
    from stllm_amd import synth
    g = torch.Generator().manual_seed(seed)
    ids = lambda n: " ".join(str(int(x)) for x in torch.randint(3, 32000, (n,), generator=g))
    frames = synth.normal_(torch.empty(B, T, 3, 224, 224, device=device), "input.video", seed, 1.0)
    return {"image": frames, "instruction_input": [ids(7) + "<ImageHere>" + ids(40) for _ in range(B)],
            "answer": [ids(15) for _ in range(B)]}  # +eos => 16 answer ids; BOS is prepended by the model


def algorithmic_flops(T, S):
    """2*MAC, unpadded."""
    vit = T * 520.72e9
    qf = T * 12.75e9 + T * 0.201e9
    llm = S * 2 * 32 * (4 * 4096 ** 2 + 3 * 4096 * 11008) + 2 * S * S * 4096 * 32 + 2 * 4096 * 32000 * S
    return vit + qf + llm


def cpu_baseline(T, S, budget_s=25.0):
    """The oracle (a port of the reference's CPU path, oracle/stllm_oracle.py) timed on this box's host cores
    on a BOUNDED sample of the same workload, extrapolated per stage (reported, never the target)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import shapes
    import stllm_oracle as O
    cores = min(os.cpu_count() or 1, 64)   # more threads than this only adds fork/join overhead at these sizes
    torch.set_num_threads(cores)
    rnd = lambda shp: {k: torch.randn(v) * 0.02 for k, v in shp.items()}
    NF, VB, QL, LL = 16, 8, 12, 8    # sample: 16 frames x 8/39 ViT blocks, 16 frames x 12/12 Q-Former layers, 8/32 Llama layers (~10-20 s of CPU work)
    with torch.no_grad():
        sd = rnd(shapes.vit_shapes(VB, "v."))
        fr = torch.randn(NF, 3, 224, 224)
        O.vit_forward(fr[:2], sd, "v.")  # warm-up (thread pool, allocator)
        t0 = time.perf_counter(); O.vit_forward(fr, sd, "v."); t_vit2 = time.perf_counter() - t0
        vit_per_frame = t_vit2 / NF / VB * 39
        sd = rnd({**shapes.qformer_shapes(QL, False, p="q."), "qt": (1, 32, 768)})
        enc = torch.randn(NF, 257, 1408)
        O.qformer_forward(sd["qt"].expand(2, -1, -1), enc[:2], sd, "q.")
        t0 = time.perf_counter(); O.qformer_forward(sd["qt"].expand(NF, -1, -1), enc, sd, "q."); t_qf = time.perf_counter() - t0
        qf_per_frame = t_qf / NF / QL * 12
        sd = rnd({k: v for k, v in shapes.llama_shapes(LL).items() if "embed" not in k})
        x = torch.randn(1, S, 4096) * 0.05
        O.llama_forward(x[:, :64], None, sd)
        t0 = time.perf_counter(); h = O.llama_forward(x, None, sd); t_l1 = (time.perf_counter() - t0) / LL
        t0 = time.perf_counter(); O.lm_logits(h, sd); t_head = time.perf_counter() - t0
    clip_s = T * (vit_per_frame + qf_per_frame) + 32 * t_l1 + t_head
    return {"value": round(T * 32 / clip_s, 3), "unit": "video-tokens/s", "cores": cores, "kind": "port",
            "dtype": "f32",
            "sample": f"oracle on CPU fp32, {cores} threads: ViT {NF} frames x {VB}/39 blocks ({t_vit2:.2f}s), Q-Former {NF} frames x "
                      f"{QL}/12 layers ({t_qf:.2f}s), Llama {LL}/32 layers at S={S} ({t_l1 * LL:.2f}s) + lm_head ({t_head:.2f}s); "
                      f"extrapolated linearly to T={T}, 39/12/32 layers => {clip_s:.1f} s/clip"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--vit-depth", type=int, default=39)
    ap.add_argument("--qformer-layers", type=int, default=12)
    ap.add_argument("--llm-layers", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--vit-streams", type=int, default=1)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
        args.gpus = world
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group(backend="nccl", device_id=device)
    torch.set_grad_enabled(False)

    from stllm_amd import hip, runtime
    runtime.set_compute_dtype(args.dtype)
    model = build_model(device, args)
    sm = model.model.stllm_model
    sm.set_frame_parallel(rank, world)
    sm.visual_encoder.frame_streams = args.vit_streams
    B, T = world, args.frames
    samples = make_samples(B, T, device)
    Lvis = T * 32

    def step():
        return model(samples=samples)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- warmup (first step also packs the weights); calibration pass finds the dominant GEMM kernel ----
    out = None
    for i in range(max(args.warmup, 1)):
        out = step()
    torch.cuda.synchronize()
    S = out.logits.shape[1]
    prof = None
    if not args.no_roofline:
        # calibration step on EVERY rank (it contains the all-gather collective); only rank 0 times its GEMM launches
        if rank == 0:
            prof = hip.GemmProfiler()
            hip.set_profiler(prof)
        step()
        torch.cuda.synchronize()
        if rank == 0:
            cal = prof.summary()
            prof.target = max(cal, key=lambda k: cal[k]["total_ms"])
            prof.mode, prof.records = "target", {}
    # ---- timed region: EXACTLY K steps between barrier + synchronize ---------------------------------
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    sync()
    dt_s = time.perf_counter() - t0
    hip.set_profiler(None)
    if not hip.gemm_workspace_ok(device):   # a split-K exchange gave up waiting for a peer workgroup: the numbers would be meaningless
        raise RuntimeError(hip.lib().stllm_last_error().decode())
    t = torch.tensor([dt_s], device=device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt_s = float(t.item())
    ms_per_step = dt_s / args.steps * 1e3
    loss = float(out.loss.item()) if out.loss is not None else float("nan")

    if rank == 0:
        res = {"metric": "video-tokens/sec (ViT+Qformer+LLM-prefill) at T=16, Vicuna-7B", "value": round(B * Lvis / (dt_s / args.steps), 2),
               "unit": "video-tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": args.dtype, "data": "synthetic (random 224x224 frames, random-init EVA-CLIP-g + Q-Former + Vicuna-7B, fixed token ids)",
               "config": {"workload": f"BASELINE configs[1]: B={B} clip(s)/step ({B // world} per GPU), T={T} frames, ViT {args.vit_depth} blocks + "
                                      f"Q-Former {args.qformer_layers} layers + Llama {args.llm_layers} layers prefill S={S} + lm_head(all positions)",
                          "global_batch": B, "frames": T, "video_tokens_per_clip": Lvis, "seq_len": S,
                          "parallelism": "single GPU" if world == 1 else f"frame-parallel x{world} + RCCL all-gather + clip-parallel prefill"},
               "frames_per_s": round(B * T / (dt_s / args.steps), 2), "loss": round(loss, 5),
               "algorithmic_tflop_per_step": round(B * algorithmic_flops(T, S) / 1e12, 3),
               "end_to_end_tflops_per_gpu": round(algorithmic_flops(T, S) / (dt_s / args.steps) / 1e12, 1)}
        full = (args.vit_depth, args.qformer_layers, args.llm_layers, T) == (39, 12, 32, 16)
        if not full:
            res["config"]["workload"] += "  [REDUCED — not the BASELINE config; for debugging only]"
        if prof is not None and prof.records:
            s = prof.summary()[prof.target]
            avg_ms = s["total_ms"] / s["launches"]
            ach = s["flops"] / (s["total_ms"] * 1e-3) / 1e12
            traffic = None
            tp = os.path.join(ROOT, "profiles", "traffic_r01.json")
            if os.path.exists(tp):
                traffic = json.load(open(tp)).get(prof.target)
            res["roofline"] = {"bound": "mfma", "kernel": prof.target, "achieved": round(ach, 1), "peak": MFMA_PEAK[args.dtype],
                               "unit": "TFLOP/s", "frac": round(ach / MFMA_PEAK[args.dtype], 4), "traffic": traffic,
                               "launches_timed": s["launches"], "avg_launch_ms": round(avg_ms, 5),
                               "algorithmic_gflop_per_launch": round(s["flops"] / s["launches"] / 1e9, 2),
                               "all_gemm_kernels_one_step": {k: {"launches": v["launches"], "ms": round(v["total_ms"], 3),
                                                                 "tflops": round(v["flops"] / max(v["total_ms"], 1e-9) / 1e9, 1)}
                                                             for k, v in sorted(cal.items(), key=lambda kv: -kv[1]["total_ms"])}}
        if not args.no_cpu_baseline and world == 1:   # reported baseline: rank 0, N=1 only
            res["cpu_baseline"] = cpu_baseline(T, S)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
