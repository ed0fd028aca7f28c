#!/bin/bash
# round-4 GPU call E: smoke(), bench lines of configs c3 / c4 / c5 at N = 1, HBM-traffic PMC passes (profiles/traffic_r04.json), matrix-pipe busy PMC
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04e
mkdir -p $O
cd $R
(timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3) > $O/smoke.log; tail -1 $O/smoke.log
for c in c3 c4 c5; do
  timeout 900 python bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err; echo "bench $c rc $?"
done
export TMPDIR=/tmp
for S in "4112 1408 1408 resid" "4112 1408 6144 resid" "4112 6144 1408 gelu" "4112 4224 1408 store"; do
  timeout 300 bash tools/pmc_traffic.sh 4 $S > $O/pmc_traffic_$(echo $S | tr ' ' '_').log 2>&1; echo "pmc $S rc $?"
done
cp profiles/traffic_r04.json $O/ 2>/dev/null
timeout 600 bash tools/pmc_mfma.sh > $O/pmc_mfma.log 2>&1; cp gpurun_out/pmc_mfma/summary.md $O/mfma_busy.md 2>/dev/null
find gpurun_out/pmc_r04 gpurun_out/pmc_mfma -type f -size +2M -delete 2>/dev/null
python - <<'PY'
import json
for c in ("c3","c4","c5"):
    try:
        d=json.load(open(f"gpurun_out/r04e/bench_{c}.json"))
        print(c, d["ms_per_step"], d["value"], d["config"]["seq_len"], d.get("parity",{}).get("logits_max_abs_err"), d.get("parity",{}).get("top1_agreement"), d.get("parity",{}).get("mask_equals_reference"), d["roofline"]["kernel"], d["roofline"]["frac"], d["end_to_end_tflops_per_gpu"])
    except Exception as e: print(c, "ERR", e)
PY
cat $O/mfma_busy.md 2>/dev/null | head -12
