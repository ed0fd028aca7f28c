#!/usr/bin/env python3
"""Are the norm kernels at their floor?  (VERDICT r04 "Next round" #7: "norm_* <= 0.8 ms / step or a measured negative".)
Times stllm_layernorm / stllm_rmsnorm at the model's shapes next to stllm_cast_rows on the same rows — the same bytes (fp32 row in, 16-bit row out)
with no statistics at all: the floor of ANY stand-alone kernel that reads the fp32 stream and writes the GEMM operand.  GPU only.
    python tools/norm_floor.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from stllm_amd import hip  # noqa: E402


def timed(fn, iters=200):
    for _ in range(10):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def main():
    dt = torch.bfloat16
    for name, M, D, rms, per_step in (("ViT LayerNorm", 4112, 1408, False, 79), ("Llama RMSNorm", 576, 4096, True, 65), ("Q-Former LayerNorm", 512, 768, False, 31)):
        # 8 distinct row blocks in rotation: the rows come from L2 / MALL / HBM as in the model (x was just written by a GEMM), not from one hot buffer
        xs = [torch.randn(M, D, device="cuda") for _ in range(8)]
        g, b = torch.rand(D, device="cuda"), torch.rand(D, device="cuda")
        out = torch.empty(M, D, device="cuda", dtype=dt)
        i = [0]

        def norm():
            x = xs[i[0] & 7]; i[0] += 1
            if rms:
                hip.rmsnorm(x, g, 1e-6, dtype=dt, out_t=out)
            else:
                hip.layernorm(x, g, b, 1e-6, dtype=dt, out_t=out)

        def cast():
            x = xs[i[0] & 7]; i[0] += 1
            hip.cast_rows(x, dt, out=out)
        t_n, t_c = timed(norm), timed(cast)
        mb = M * D * 6 / 1e6
        print(f"{name:20s} {M:5d} x {D:5d}: norm {t_n:6.2f} us ({mb / t_n * 1e-3 * 1e3:5.2f} TB/s)   cast_rows (same bytes, no statistics) {t_c:6.2f} us   "
              f"-> norm - floor = {t_n - t_c:5.2f} us x {per_step} per step = {(t_n - t_c) * per_step / 1e3:5.3f} ms")


if __name__ == "__main__":
    main()
