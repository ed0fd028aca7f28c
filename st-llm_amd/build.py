"""Build libstllm_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libstllm_hip.so")
SOURCES = ["gemm.hip", "gemm_p8_bf16.hip", "gemm_p8_f16.hip", "gemm_w4_bf16.hip", "gemm_w4_f16.hip", "gemm_t1_bf16.hip", "gemm_t1_f16.hip", "gemm_wd_bf16.hip", "gemm_wd_f16.hip", "gemv.hip", "norm.hip", "attention.hip", "elementwise.hip", "split3.hip", "preprocess.hip", "train_ops.hip", "attention_bwd.hip", "error.cpp", "stacks.cpp", "profile.cpp"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    return "hipcc"


def _digest():
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode() + fh.read())
    with open(os.path.join(HERE, "..", "include", "stllm_hip.h"), "rb") as fh:
        h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force=False, verbose=True, trace=False):
    """Compile every translation unit to an object (in parallel), then link.  Skips when up to date.
    trace=True: the same library with the in-kernel timeline stamps of gemm_w4.inc compiled in (-DSTLLM_W4_TRACE), written to
    st-llm_amd/trace/libstllm_hip.so — for tools/gemm_harness only (LD_LIBRARY_PATH=st-llm_amd/trace:...), never loaded by the package."""
    if trace and trace is not True:   # "--trace-x N": a trace build with a part of the w4 epilogue cut out (timeline experiments; results are garbage)
        return _build_to(os.path.join(HERE, f"trace_x{int(trace)}", "libstllm_hip.so"), os.path.join(HERE, "build", f"trace_x{int(trace)}"),
                         FLAGS + ["-DSTLLM_W4_TRACE", f"-DSTLLM_W4_EXPERIMENT={int(trace)}"], verbose)
    if trace:
        return _build_to(os.path.join(HERE, "trace", "libstllm_hip.so"), os.path.join(HERE, "build", "trace"), FLAGS + ["-DSTLLM_W4_TRACE"], verbose)
    stamp = LIB + ".stamp"
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == dig:
        return LIB
    _build_to(LIB, os.path.join(HERE, "build"), FLAGS, verbose)
    with open(stamp, "w") as f:
        f.write(dig)
    return LIB


def _build_to(LIB, objdir, FLAGS, verbose):
    hipcc = _hipcc()
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    os.makedirs(objdir, exist_ok=True)
    procs = []
    for s in SOURCES:
        obj = os.path.join(objdir, s + ".o")
        cmd = [hipcc] + FLAGS + ["-x", "hip", "-c", os.path.join(CSRC, s), "-o", obj]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        procs.append((s, obj, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    objs = []
    for s, obj, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {s}:\n{out.decode()}")
        if verbose and out.strip():
            print(out.decode())
        objs.append(obj)
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    tr = "--trace" in sys.argv
    if "--trace-x" in sys.argv:
        tr = int(sys.argv[sys.argv.index("--trace-x") + 1])
    print(build(force="--force" in sys.argv, trace=tr))
