#!/bin/bash
# Re-measure the HBM-side traffic of the GEMM symbol bench.py reports as dominant (measure, do not replay).
#
#   tools/pmc_traffic.sh <round> [M N K epilogue]        (defaults: the ViT fc1 GEMM, 4112 6144 1408 gelu)
#
# Two counter-only rocprofv3 passes (FETCH_SIZE costs 3 of the 4 TCC slots, WRITE_SIZE 2: they cannot share a pass;
# MI355X_MICROARCH.md "rocprofv3 PMC slots") over `tools/gemm_one.py`, then tools/pmc_traffic.py turns the per-dispatch rows of the
# GEMM kernel into profiles/traffic_rNN.json = { "round": NN, "kernels": { "<symbol as stllm_last_kernel() names it>":
# { "hbm_bytes_per_launch": FETCH_SIZE[KiB] * 1024 * 2 (gfx950 counts a 128-B request as 64 B) + WRITE_SIZE[KiB] * 1024, ... } } }.
# bench.py only accepts a file whose round matches its own ROUND constant and that holds the symbol it found dominant.
set -euo pipefail
ROUND=${1:?round number}
M=${2:-4112}; N=${3:-6144}; K=${4:-1408}; EPI=${5:-gelu}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/pmc_r$(printf %02d "$ROUND")/${M}x${N}x${K}_$EPI
rm -rf "$OUT"
mkdir -p "$OUT"
export TMPDIR=${TMPDIR:-/tmp}
cd "$ROOT"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/$C" -- python tools/gemm_one.py $M $N $K 5 $EPI > "$OUT/$C.log" 2>&1
done
python tools/pmc_traffic.py "$ROUND" "$OUT" $M $N $K $EPI
