#!/bin/bash
# re-calibrated cost models (large-M regimes) in the model: bench A/B previous / new library on c2 (must not move), c4, c3 (one GPU) and the training step
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04ah; mkdir -p $O; cd $R
for cfg in c2 c4 c3; do
  st=40; [ $cfg = c3 ] && st=8; [ $cfg = c4 ] && st=20
  for i in 1 2; do for lib in prev new; do
    if [ $lib = prev ]; then export STLLM_LIB=$R/st-llm_amd/prev/libstllm_hip.so; else unset STLLM_LIB; fi
    timeout 600 python bench.py --config $cfg --steps $st --warmup 3 --no-extra-legs --no-cpu-baseline --no-projection --no-roofline > $O/b.json 2>/dev/null
    python -c "import json; d=json.load(open('$O/b.json')); print('$cfg $lib', d['ms_per_step'], d['ms_per_step_blocks']['ms'], d['parity']['logits_max_abs_err'], d['telemetry']['sclk_mhz']['mean'])"
  done; done
done 2>&1 | tee $O/bench_ab.log
for lib in prev new prev new; do
  if [ $lib = prev ]; then export STLLM_LIB=$R/st-llm_amd/prev/libstllm_hip.so; else unset STLLM_LIB; fi
  echo "== train $lib"; timeout 600 python tools/train_bench.py --layers 32 --batch 16 --steps 6 2>&1 | grep "^step" | tail -5
done 2>&1 | tee $O/train_ab.log
