#!/bin/bash
# Matrix-pipe busy fraction of the four ViT GEMMs and the four Llama prefill GEMMs (every GEMM >= 1 ms per step) from SQ counters (one counter-only rocprofv3 pass per shape over tools/gemm_one.py):
#   tools/pmc_mfma.sh   ->  gpurun_out/pmc_mfma/summary.md   (copy to profiles/r0N_mfma_busy_pmc.md)
# SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the SIMDs (= 32 x number of 32x32x16 MFMAs); busy fraction = that / (1024 SIMDs x
# kernel cycles); kernel cycles = GRBM_GUI_ACTIVE of the dispatch / 8 (the counter comes summed over the 8 XCDs: / 8 it is 2.15-2.2 GHz x
# the kernel-trace duration of the same dispatch for the 55-85 us kernels; MI355X_MICROARCH.md "rocprofv3 PMC slots").
set -uo pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/pmc_mfma
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=${TMPDIR:-/tmp}
cd "$ROOT"
CTR="SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
for S in "4112 6144 1408 gelu" "4112 4224 1408 store" "4112 1408 1408 resid" "4112 1408 6144 resid" "576 12288 4096 rope" "576 4096 4096 resid" "576 22016 4096 swiglu" "576 4096 11008 resid"; do
  set -- $S
  D=$OUT/$1x$2x$3_$4
  timeout 120 rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d "$D" -- python tools/gemm_one.py $1 $2 $3 6 $4 > "$D.log" 2>&1 || echo "pass failed: $S (see $D.log)"
done
python - "$OUT" <<'PY'
import csv, glob, os, re, sys
out = sys.argv[1]
rows = ["| shape (M x N x K, epilogue) | kernel | duration us (kernel trace, under the counters) | MFMA busy cycles / SIMD | kernel cycles (GRBM_GUI_ACTIVE / 8) | matrix pipe busy | waves parked (WAIT_ANY / WAVE_CYCLES) | issue stall (WAIT_INST_ANY / WAVE_CYCLES) |", "|---|---|---|---|---|---|---|---|"]
for d in sorted(glob.glob(os.path.join(out, "*x*_*"))):
    if not os.path.isdir(d):
        continue
    fs = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not fs:
        rows.append(f"| {os.path.basename(d)} | no counter file | | | | | | |")
        continue
    per = {}
    for r in csv.DictReader(open(fs[0])):
        if re.search(r"gemm_(p8|w4|sk|t1|wd)?_?kernel", r["Kernel_Name"]):
            per.setdefault(r["Dispatch_Id"], {"k": r["Kernel_Name"]})[r["Counter_Name"]] = float(r["Counter_Value"])
    disp = [per[k] for k in sorted(per, key=int)][1:]          # drop the first (cold) launch
    kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    durs = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt[0])) if re.search(r"gemm_(p8|w4|sk|t1|wd)?_?kernel", r["Kernel_Name"])][1:] if kt else []
    dur_us = sum(durs) / len(durs) / 1e3 if durs else 0.0
    if not disp:
        continue
    avg = lambda c: sum(x.get(c, 0.0) for x in disp) / len(disp)
    mf, gui, wc, wa, wi = avg("SQ_VALU_MFMA_BUSY_CYCLES"), avg("GRBM_GUI_ACTIVE") / 8, avg("SQ_WAVE_CYCLES"), avg("SQ_WAIT_ANY"), avg("SQ_WAIT_INST_ANY")
    name = re.sub(r"\(anonymous namespace\)::|void |\(sg::GemmParams\)", "", disp[0]["k"])
    rows.append(f"| {os.path.basename(d).replace('_', ', ')} | `{name}` | {dur_us:.1f} | {mf / 1024:.0f} | {gui:.0f} | **{mf / 1024 / gui:.3f}** | {wa / wc if wc else 0:.3f} | {wi / wc if wc else 0:.3f} |")
open(os.path.join(out, "summary.md"), "w").write("\n".join(rows) + "\n")
print("\n".join(rows))
PY
