#!/usr/bin/env python3
"""Second half of tools/pmc_traffic.sh: counter_collection CSVs of the two PMC passes -> profiles/traffic_rNN.json."""
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_launch(outdir, counter):
    files = glob.glob(os.path.join(outdir, counter, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {outdir}/{counter}")
    vals, name = [], None
    for r in csv.DictReader(open(files[0])):
        if r["Counter_Name"] == counter and re.search(r"gemm_(p8|w4|sk|t1|wd)?_?kernel", r["Kernel_Name"]):
            vals.append(float(r["Counter_Value"]))
            name = r["Kernel_Name"]
    if not vals:
        raise SystemExit(f"no GEMM dispatch with {counter} in {files[0]}")
    vals = vals[1:] if len(vals) > 1 else vals      # drop the first launch (cold caches, first-touch)
    return sum(vals) / len(vals), len(vals), name


def symbol(kernel_name):
    """the demangled device symbol -> the short name bench.py gets from stllm_last_kernel()"""
    m = re.search(r"(gemm_(?:p8_|w4_|sk_|t1_|wd_)?kernel)<([^>]*)>", kernel_name)
    fam, args = m.group(1), [a.strip() for a in m.group(2).split(",")]
    epi = ["STORE", "RESID", "SWIGLU", "ROPE", "PATCH"]
    b = lambda x: "1" if x == "true" else "0" if x == "false" else x
    if fam in ("gemm_p8_kernel", "gemm_t1_kernel", "gemm_wd_kernel"):      # <T, MIW | NRW | WM, EPI, ACT, OF32>
        return f"{fam}<{args[0]},{args[1]},{epi[int(args[2])]},{args[3]},{b(args[4])}>"
    if fam == "gemm_w4_kernel":      # <T, WM, WN, EPI, ACT, OF32>
        return f"{fam}<{args[0]},{args[1]},{args[2]},{epi[int(args[3])]},{args[4]},{b(args[5])}>"
    return f"{fam}<{args[0]},{args[1]},{args[2]},{epi[int(args[3])]},{args[4]},{b(args[5])}>"   # <T, BM, BN, EPI, ACT, OF32>


def main():
    rnd, outdir, M, N, K, epi = int(sys.argv[1]), sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
    fetch_kib, n1, kname = per_launch(outdir, "FETCH_SIZE")
    write_kib, n2, _ = per_launch(outdir, "WRITE_SIZE")
    fetch_b, write_b = fetch_kib * 1024 * 2, write_kib * 1024
    out_b = M * N * (4 if epi == "resid" else 2)
    algo = (M + N) * K * 2 + out_b + (M * N * 4 if epi == "resid" else 0)
    path = os.path.join(ROOT, "profiles", f"traffic_r{rnd:02d}.json")
    d = json.load(open(path)) if os.path.exists(path) else {}
    if d.get("round") != rnd:
        d = {"round": rnd, "_comment": "HBM-side bytes per launch from rocprofv3 PMC passes (tools/pmc_traffic.sh): FETCH_SIZE [KiB] x 1024 x 2 "
                                        "(gfx950: 128-B requests tallied as 64 B, MI355X_MICROARCH.md) + WRITE_SIZE [KiB] x 1024", "kernels": {}}
    sym = symbol(kname)
    one = {"hbm_bytes_per_launch": round(fetch_b + write_b), "fetch_bytes": round(fetch_b), "write_bytes": round(write_b),
           "algorithmic_bytes": algo, "shape": [M, N, K], "epilogue": epi, "launches_averaged": min(n1, n2)}
    # one symbol can serve several shapes of the model (w4 192x128 RESID = ViT proj AND fc2, equally often): keep every shape and
    # report the mean over them, which is what bench.py's per-launch average of that symbol corresponds to
    e = d["kernels"].get(sym, {})
    shapes = e.get("per_shape", {})
    if not shapes and "shape" in e:
        shapes["x".join(map(str, e["shape"]))] = {k: e[k] for k in one if k in e}
    shapes[f"{M}x{N}x{K}"] = one
    n = len(shapes)
    d["kernels"][sym] = {"hbm_bytes_per_launch": round(sum(v["hbm_bytes_per_launch"] for v in shapes.values()) / n),
                         "algorithmic_bytes": round(sum(v["algorithmic_bytes"] for v in shapes.values()) / n),
                         "per_shape": shapes}
    json.dump(d, open(path, "w"), indent=1)
    print(f"{sym}: fetch {fetch_b / 1e6:.1f} MB + write {write_b / 1e6:.1f} MB = {(fetch_b + write_b) / 1e6:.1f} MB per launch "
          f"(algorithmic {algo / 1e6:.1f} MB) -> {path}")


if __name__ == "__main__":
    main()
