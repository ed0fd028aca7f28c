"""Build tests/hipemu/_build/libstllm_emu.so: the training kernels of st-llm_amd/csrc compiled for the HOST against the
emulation shim (tests/hipemu/hip/hip_runtime.h).  Test infrastructure only (see the shim's header)."""
import os
import re
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "st-llm_amd", "csrc")
OUT = os.path.join(HERE, "_build")
KERNEL_SOURCES = ["train_ops.hip", "attention_bwd.hip", "attention.hip", "gemv.hip", "norm.hip", "elementwise.hip", "split3.hip", "preprocess.hip", "gemm.hip"]
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
DYN = "#include <hip/hip_runtime.h>\nnamespace {{ alignas(16) {type} smem[{n}]; }}\n"   # 160 KB of dynamic LDS


def build(force=False):
    os.makedirs(OUT, exist_ok=True)
    lib = os.path.join(OUT, "libstllm_emu.so")
    srcs = [os.path.join(CSRC, f) for f in KERNEL_SOURCES + ["common.h", "gemm_common.h", "error.cpp", "stacks.cpp"]] + [os.path.join(HERE, "hip", "hip_runtime.h"), os.path.join(HERE, "p8_stubs.cpp"),
                                                                                                          os.path.join(ROOT, "include", "stllm_hip.h"), os.path.join(CSRC, "options.h")]
    if not force and os.path.exists(lib) and all(os.path.getmtime(lib) >= os.path.getmtime(s) for s in srcs):
        return lib
    tus = []
    # headers with inline ISA get the same treatment, as a patched copy that shadows the original for the emulated TUs only
    hdr = open(os.path.join(CSRC, "gemm_common.h")).read()
    hdr = re.sub(r'asm volatile\("s_waitcnt vmcnt\(%0\)" ::"n"\(N\) : "memory"\);', "", hdr)
    with open(os.path.join(OUT, "gemm_common.h"), "w") as fh:
        fh.write(hdr)
    for f in KERNEL_SOURCES:
        text = open(os.path.join(CSRC, f)).read()
        m = re.search(r"extern\s+__shared__[^;]*?(\w+)\s+smem\[\]", text)
        head = ""
        if m:
            ty = m.group(1)
            head = DYN.format(type=ty, n=163840 // (4 if ty == "float" else 1))
        text = re.sub(r"extern\s+__shared__", "EMU_DYN_SHARED", text)
        # the column reductions split the rows over 64 workgroups to fill the chip; 4 keep the emulation cheap (same code path)
        text = text.replace("constexpr int kColBlocks = 64;", "constexpr int kColBlocks = 4;")
        # inline ISA: a barrier that does not drain the LDS-DMA queue is a plain barrier here; bare waits vanish (copies are synchronous)
        text = re.sub(r'asm volatile\("s_waitcnt lgkmcnt\(0\)\\n\\ts_barrier" ::: "memory"\)', "__syncthreads()", text)
        text = re.sub(r'asm volatile\("s_waitcnt [a-z]+cnt\(\d+\)" ::: "memory"\)', "((void)0)", text)
        # attention.hip: the pair of hardware transposing LDS reads -> the shim's emulation (hip_runtime.h: emu_ds_read_tr_b16)
        text = re.sub(r'const unsigned a = \(unsigned\)\(uintptr_t\)\(__attribute__\(\(address_space\(3\)\)\) const char\*\)p;\s*asm volatile\("ds_read_b64_tr_b16[^;]*;',
                      "lo = emu_ds_read_tr_b16(p); hi = emu_ds_read_tr_b16(p + OFF);", text, count=1)
        text = re.sub(r'const unsigned a = \(unsigned\)\(uintptr_t\)\(__attribute__\(\(address_space\(3\)\)\) const char\*\)p;\s*asm volatile\("ds_read_b64_tr_b16[^;]*;',
                      "l0 = emu_ds_read_tr_b16(p); h0 = emu_ds_read_tr_b16(p + OFF); l1 = emu_ds_read_tr_b16(p + 64); h1 = emu_ds_read_tr_b16(p + OFF + 64); "
                      "l2 = emu_ds_read_tr_b16(p + 128); h2 = emu_ds_read_tr_b16(p + OFF + 128);", text, count=1)
        # ... and the four-pair block of the 8-wave prefill kernel (four addresses, one wait)
        text = re.sub(r'typedef __attribute__\(\(address_space\(3\)\)\) const char\* lp;\s*const unsigned a0 =[^;]*;\s*asm volatile\("ds_read_b64_tr_b16[^;]*;',
                      "lo[0] = emu_ds_read_tr_b16(p0); hi[0] = emu_ds_read_tr_b16(p0 + OFF); lo[1] = emu_ds_read_tr_b16(p1); hi[1] = emu_ds_read_tr_b16(p1 + OFF); "
                      "lo[2] = emu_ds_read_tr_b16(p2); hi[2] = emu_ds_read_tr_b16(p2 + OFF); lo[3] = emu_ds_read_tr_b16(p3); hi[3] = emu_ds_read_tr_b16(p3 + OFF);", text, count=1)
        tu = os.path.join(OUT, f.replace(".hip", ".emu.cpp"))
        with open(tu, "w") as fh:
            fh.write(head + text)
        tus.append(tu)
    flags = ["-x", "c++", "-std=c++20", "-O1", "-ffp-contract=off", "-fPIC", "-pthread", "-Wno-unused-value", "-Wno-comment", "-Wno-psabi",
             "-I", HERE, "-I", OUT, "-I", CSRC]
    units = tus + [os.path.join(HERE, "p8_stubs.cpp"), os.path.join(CSRC, "error.cpp"), os.path.join(CSRC, "stacks.cpp")]   # stacks.cpp: host code over the C ABI

    def compile_one(src):
        obj = os.path.join(OUT, os.path.basename(src) + ".o")
        r = subprocess.run([CLANG] + flags + ["-c", src, "-o", obj], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"emulation build failed for {src}:\n" + r.stderr[-4000:])
        return obj
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(8, len(units))) as ex:       # one clang per translation unit, in parallel
        objs = list(ex.map(compile_one, units))
    r = subprocess.run([CLANG, "-shared", "-pthread", "-o", lib] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("emulation link failed:\n" + r.stderr[-4000:])
    return lib


if __name__ == "__main__":
    print(build(force=True))
