"""CPU: the GELU polynomial of the 16-bit GEMM epilogues (csrc/common.h STLLM_GELU_*) — the constants in the header are what
tools/fit_gelu_poly.py produces, and evaluated in float32 Horner arithmetic as the kernels run it they stay within the documented
error of the exact-erf GELU of the reference (nn.GELU(), eva_vit.py:54-61): |error| <= 1.25e-5 |x| + 2e-7 inside the clamp (9.9e-6 |x| for the fit in exact arithmetic + float32 rounding), both tails exact."""
import os
import re
import subprocess
import sys

import numpy as np
from scipy.special import erf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_constants():
    text = open(os.path.join(ROOT, "st-llm_amd", "csrc", "common.h")).read()
    vals = dict(re.findall(r"#define STLLM_GELU_(X0|C\d) (-?[\d.e+-]+)f", text))
    assert set(vals) == {"X0"} | {f"C{i}" for i in range(9)}, vals
    return np.float32(vals["X0"]), [np.float32(vals[f"C{i}"]) for i in range(8, -1, -1)]   # Horner order: C8 first


def gelu_poly32(x, x0, coef):
    x = x.astype(np.float32)
    xc = np.clip(x, -x0, x0)
    t = (xc * xc).astype(np.float32)
    p = np.full_like(x, coef[0])
    for c in coef[1:]:
        p = (p.astype(np.float64) * t + c).astype(np.float32)          # fmaf: one rounding
    phi = (xc.astype(np.float64) * p + 0.5).astype(np.float32)
    return (x * phi).astype(np.float32)


def test_header_constants_are_the_fit():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fit_gelu_poly.py"), "4.3", "8"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    fitted = [np.float32(v) for v in re.findall(r"np\.float32\((-?[\d.e+-]+)\)\s+-?\d", r.stdout)]
    clamp = np.float32(re.search(r"clamp at xc = np\.float32\(([\d.]+)\)", r.stdout).group(1))
    x0, coef = header_constants()
    assert len(fitted) == 9, r.stdout
    # the LP solver may move the last bits between library versions: the header must agree to ~1e-6 relative, the clamp to 1e-3
    for got, want in zip(coef, fitted):
        assert abs(float(got) - float(want)) <= 2e-6 * abs(float(want)), (coef, fitted)
    assert abs(float(x0) - float(clamp)) <= 1e-3, (x0, clamp)


def test_polynomial_error_profile():
    x0, coef = header_constants()
    xs = np.linspace(-12.0, 12.0, 960001)
    ref = xs * 0.5 * (1.0 + erf(xs / np.sqrt(2.0)))
    err = np.abs(gelu_poly32(xs, x0, coef).astype(np.float64) - ref)
    inside = np.abs(xs) <= float(x0)
    assert (err[inside] <= 1.25e-5 * np.abs(xs[inside]) + 2e-7).all(), float((err[inside] - 1.25e-5 * np.abs(xs[inside])).max())
    assert err.max() <= 5.2e-5, err.max()
    # tails: exactly 0 * x and 1 * x up to one float32 rounding of Phi
    assert np.abs(gelu_poly32(np.array([-50.0, -12.0]), x0, coef)).max() <= 1e-6
    assert np.array_equal(gelu_poly32(np.array([12.0, 50.0]), x0, coef), np.array([12.0, 50.0], dtype=np.float32))
