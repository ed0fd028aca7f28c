#!/usr/bin/env python3
"""In-kernel timeline of the ViT / Llama prefill attention kernels (trace build: tools/attn_trace.sh).
    STLLM_LIB=st-llm_amd/attn_trace/libstllm_hip.so python tools/attn_trace.py vit|llama
Per wave 8 slots of s_memtime (shader cycles).  ViT (attn_dma88): 0 start, 1 Q + all windows requested, 2 / 4 / 6 window 0 / 1 / 2 ready (after the
barrier), 3 / 5 / 7 window computed.  Llama (attn_dma): 0 start, 1 first window requested, 2 first window ready, 3 key loop done, 4 merge done,
5 stores issued; slot 6 = number of windows, 7 = cycles spent in the per-window wait + barrier."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from stllm_amd import hip  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "vit"
B, H, S, D, causal = (16, 16, 257, 88, False) if which == "vit" else (1, 32, 576, 128, True)
buf = torch.randn(B * S, 3 * H * D, device="cuda").to(torch.bfloat16)
q, k, v = buf[:, :H * D], buf[:, H * D:2 * H * D], buf[:, 2 * H * D:]
L = hip.lib()
L.stllm_attn_trace_read.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int]
n = 512 * 12 * 8
host = np.zeros(n, dtype=np.uint64)
run = lambda: hip.attention(q, k, v, B=B, H=H, Sq=S, Skv=S, D=D, scale=D ** -0.5, causal=causal)
for _ in range(5): run()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20): run()
e.record(); torch.cuda.synchronize()
us = s.elapsed_time(e) / 20 * 1e3
print(f"{which}: {us:.1f} us per launch [{L.stllm_last_kernel().decode()}]")
assert L.stllm_attn_trace_read(host.ctypes.data, n * 8, 1) == 0
run()
assert L.stllm_attn_trace_read(host.ctypes.data, n * 8, 0) == 0
t = host.reshape(512, 12, 8).astype(np.int64)
live = t[:, :, 0] > 0
t0 = t[:, :, 0][live].min()
nwg = int(live.any(axis=1).sum())
print(f"workgroups {nwg}, waves per workgroup {int(live.sum() / max(nwg, 1))}")
rel = lambda a: a - t0
starts = rel(t[:, :, 0][live])
print(f"wave start: min 0, median {np.median(starts):.0f}, p90 {np.percentile(starts, 90):.0f}, max {starts.max():.0f} cycles after the first")
if which == "vit":
    names = ["start", "requested", "w0 ready", "w0 done", "w1 ready", "w1 done", "w2 ready", "w2 done"]
    for sl in range(1, 8):
        m = live & (t[:, :, sl] > 0)
        d = (t[:, :, sl] - t[:, :, 0])[m]
        print(f"  slot {sl} {names[sl]:10s}: since own start mean {d.mean():8.0f}  p10 {np.percentile(d, 10):8.0f}  p90 {np.percentile(d, 90):8.0f}  max {d.max():8.0f}   (n = {m.sum()})")
    for w in range(12):
        m = live[:, w] & (t[:, w, 7] > 0)
        if m.any():
            print(f"  wave {w:2d}: w0 wait {np.mean(t[m, w, 2] - t[m, w, 1]):7.0f}  w0 compute {np.mean(t[m, w, 3] - t[m, w, 2]):7.0f}  w1 wait {np.mean(t[m, w, 4] - t[m, w, 3]):7.0f}  "
                  f"w1 compute {np.mean(t[m, w, 5] - t[m, w, 4]):7.0f}  w2 wait {np.mean(t[m, w, 6] - t[m, w, 5]):7.0f}  w2 compute {np.mean(t[m, w, 7] - t[m, w, 6]):7.0f}")
    ends = rel(t[:, :, 7][live & (t[:, :, 7] > 0)])
else:
    names = ["start", "w0 requested", "w0 ready", "loop done", "merge done", "stored"]
    for sl in range(1, 6):
        m = live & (t[:, :, sl] > 0)
        d = (t[:, :, sl] - t[:, :, 0])[m]
        print(f"  slot {sl} {names[sl]:12s}: since own start mean {d.mean():8.0f}  p10 {np.percentile(d, 10):8.0f}  p90 {np.percentile(d, 90):8.0f}  max {d.max():8.0f}   (n = {m.sum()})")
    # by number of windows (causal: the chunk of the workgroup)
    for nw in sorted(set(t[:, :, 6][live].tolist())):
        m = live & (t[:, :, 6] == nw) & (t[:, :, 3] > 0)
        print(f"  {nw} windows: n = {m.sum():4d} waves, loop {np.mean((t[:, :, 3] - t[:, :, 2])[m]):8.0f} cycles of which waiting {np.mean(t[:, :, 7][m]):8.0f} (incl. the first window {np.mean((t[:, :, 2] - t[:, :, 1])[m]):6.0f}); "
              f"start -> loop done {np.mean((t[:, :, 3] - t[:, :, 0])[m]):8.0f}; wave start after first {np.mean(rel(t[:, :, 0])[m]):7.0f}")
    ends = rel(t[:, :, 5][live & (t[:, :, 5] > 0)])
d1 = np.sort((t[:, :, 1] - t[:, :, 0])[live & (t[:, :, 1] > 0)])
print("start -> slot 1, sorted quantiles (min, 1 %, 5 %, 10 %, 25 %, 50 %, 75 %, 100 %):", [int(d1[min(len(d1) - 1, int(q * len(d1)))]) for q in (0, 0.01, 0.05, 0.1, 0.25, 0.5, 0.75, 1.0)])
print(f"last stamp of a wave, cycles after the first wave start: median {np.median(ends):.0f}, p90 {np.percentile(ends, 90):.0f}, max {ends.max():.0f}  ({ends.max() / us:.0f} cycles per us of the event time)")
