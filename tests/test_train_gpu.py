"""GPU (-m gpu): the training entry points on the device.

First run on hardware: the round-1 driver run (GPUTEST_r01.json: all three XPASS on an MI355X) — the staging xfail marker is
gone, a failure here is a regression.  Every check still runs in a child process with a timeout, so that a device fault cannot
take the rest of the GPU suite down.

1. every kernel case of test_kernels_emulated_cpu.py, through the real C ABI on cuda:0 (same tolerances);
2. loss_and_grads on the device against the reference's gradient fixture — MVM/mask, residual pooling + text, and the BT-Adapter
   backbone with its adapter parameters — (fp32 mode, 5e-4 of each tensor's abs-max), and
   bf16 mode against the same fixture with the tolerance of bf16 GEMM operands (cosine similarity of every gradient tensor
   >= 0.995, loss within 2e-2)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _child(args, timeout, env_extra=None):
    env = dict(os.environ, **(env_extra or {}))
    r = subprocess.run([sys.executable] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    tail = (r.stdout[-3000:] + "\n" + r.stderr[-3000:])
    assert r.returncode == 0, f"child exited with {r.returncode}:\n{tail}"
    return r.stdout


@pytest.mark.gpu
def test_training_kernels_on_device():
    out = _child(["-m", "pytest", "tests/test_kernels_emulated_cpu.py", "-q", "-x", "-k", "not entirely_on_emulated_kernels", "-p", "no:cacheprovider"],
                 timeout=900, env_extra={"STLLM_TRAIN_KERNELS_ON_DEVICE": "1"})
    assert " passed" in out and "failed" not in out, out[-2000:]


_STEP = r'''
import sys
sys.path[:0] = ["tests", "oracle", "."]
import numpy as np, torch
import test_backward_cpu as TB
from stllm_amd import runtime, training
from test_model_gpu import build_stllm
mode = sys.argv[1]
CASES = dict(TB.CASES, btadapter=(TB.BT_CASE, 4))
DEPTHS = {"btadapter": (4, 2, 1)}
for tag in sys.argv[2].split(","):
    g = TB.golden("backward")
    cfg, Tn = CASES[tag]
    text = cfg["qformer_text_input"]
    vd, ql, ll = DEPTHS.get(tag, (1, 2, 2))
    model = build_stllm(dict(cfg, image_size=224, num_query_token=32, max_txt_len=32, end_sym=" 2"), vit_depth=vd, qf_layers=ql, llm_layers=ll)
    instr, answers = TB.product_samples(g, tag, text)
    samples = {"image": TB.T("input.video", (2, Tn, 3, 224, 224)).cuda(), "instruction_input": instr, "answer": answers}
    if cfg.get("use_mask"):
        samples["mask"] = torch.from_numpy(g[f"{tag}.mask"])     # (the BT-Adapter case is the masked / MVM / text config too)
    with runtime.use_dtype(mode):
        loss, loss_mvm, grads = training.loss_and_grads(model, samples)
    torch.cuda.synchronize()
    grads = {n: v.float().cpu() for n, v in grads.items()}
    if mode == "fp32":
        TB.check_against_fixture(g, tag, loss.item(), grads, 5e-4)
    else:
        assert abs(loss.item() - g[f"{tag}.loss"][0]) < 2e-2 * abs(g[f"{tag}.loss"][0]), (loss.item(), g[f"{tag}.loss"][0])
        for n in (str(x) for x in g[f"{tag}.names"]):
            gr = grads[n]
            got = TB.sub(gr, 97, 101) if gr.dim() == 2 else TB.sub(gr, 29)
            want = g[f"{tag}.slice.{n}"]
            if np.linalg.norm(want) < 1e-12:            # e.g. embed_tokens: the sampled rows may hold no token of the batch
                assert np.linalg.norm(got) < 1e-6, n
            else:
                cs = float((got * want).sum() / (np.linalg.norm(got) * np.linalg.norm(want) + 1e-30))
                assert cs >= 0.995, (n, cs)
            st = g[f"{tag}.stats.{n}"]
            assert abs(TB.stats(gr)[0] - st[0]) <= 5e-2 * st[0], (n, TB.stats(gr)[0], st[0])
    print("ok", tag, mode, loss.item())
'''


@pytest.mark.gpu
@pytest.mark.parametrize("mode,tags", [("fp32", "mvm,residual,btadapter"), ("bf16", "mvm")])
def test_training_step_on_device_matches_reference_gradients(mode, tags):
    out = _child(["-c", _STEP, mode, tags], timeout=900)
    assert out.count("ok ") == len(tags.split(",")), out[-2000:]
