#!/usr/bin/env python3
"""Is the host ahead of the GPU?  Wraps the stllm_amd.hip entry points of one bench step and records, at every call, the host clock and
whether the stream is already idle (stream.query() == True: everything enqueued so far has finished = the GPU is waiting for the host).

    python tools/host_timeline.py [--steps 3] [--config c2|c3|c4|c5]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--config", default="c2", choices=sorted(bench.CONFIGS))
    a = ap.parse_args()
    torch.set_grad_enabled(False)
    from stllm_amd import hip, runtime
    runtime.set_compute_dtype("bf16")
    args = argparse.Namespace(vit_depth=39, qformer_layers=12, llm_layers=32)
    conf = bench.CONFIGS[a.config]
    model = bench.build_model(torch.device("cuda:0"), args, conf["model"])
    B, T = conf["clips"] or 1, conf["frames"]
    samples = bench.make_samples(B, T, "cuda:0", text=conf["model"]["qformer_text_input"])
    if conf["model"].get("use_mask"):
        samples["mask"] = bench.draw_mask((conf["model"].get("residual_size") if conf["model"]["video_input"] == "residual" else T) * 32, B)
    for _ in range(3):
        model(samples=samples)
    torch.cuda.synchronize()
    log = []
    names = ["gemm", "layernorm", "rmsnorm", "attention", "gather_rows", "vit_blocks", "llama_layers", "h2d", "cross_entropy_rows", "vit_cls_rows",
             "cast_rows", "mean_t"]
    st = torch.cuda.current_stream()
    t0 = [0.0]

    def wrap(name, fn):
        def inner(*x, **k):
            idle = st.query()
            t = time.perf_counter()
            r = fn(*x, **k)
            log.append((name, (t - t0[0]) * 1e6, (time.perf_counter() - t) * 1e6, idle))
            return r
        return inner
    for n in names:
        setattr(hip, n, wrap(n, getattr(hip, n)))
    for s in range(a.steps):
        log.clear()
        t0[0] = time.perf_counter()
        out = model(samples=samples)
        t_host = (time.perf_counter() - t0[0]) * 1e3
        torch.cuda.synchronize()
        t_all = (time.perf_counter() - t0[0]) * 1e3
        idle_calls = [(n, round(t), round(d)) for n, t, d, i in log if i]
        print(f"step {s}: host enqueue {t_host:.2f} ms, GPU done at {t_all:.2f} ms, {len(log)} calls, {len(idle_calls)} found the stream idle")
        print("   calls that found the GPU idle (name, host us since step start, call us):", idle_calls[:40])
        slow = sorted(log, key=lambda e: -e[2])[:8]
        print("   slowest host calls:", [(n, round(t), round(d)) for n, t, d, _ in slow])
        # host time between consecutive calls (python glue)
        gaps = sorted(((log[i + 1][1] - log[i][1] - log[i][2], log[i][0], log[i + 1][0], round(log[i][1])) for i in range(len(log) - 1)), reverse=True)[:6]
        print("   largest python gaps between calls (us, after, before, at):", [(round(g), x, y, t) for g, x, y, t in gaps])


if __name__ == "__main__":
    main()
