# small-M exchange-free one-wave plans: audits at c2 / c4 / c5 prefill row counts (mismatch lines only) + c2 / c5 bench A/B against the previous library
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4am; mkdir -p $O; cd $R
( for r in 4112,576 4112,528 4096,296 4112,178 4112,400 4112,700 4112,900; do echo "== rows $r"; timeout 200 python tools/gemm_bench.py --audit --rows $r --iters 15 --only llm_qkv,llm_o,llm_gu,llm_down,lm_head 2>&1 | grep -v amdgpu | cut -c1-200; done ) > $O/audit_small_m.log 2>&1
grep "^[a-z=]\|faster" $O/audit_small_m.log | grep -B1 "faster" | grep -v "^--" | cut -c1-200
for cfg in c2 c5; do for i in 1 2; do for lib in prev new; do
  if [ $lib = prev ]; then export STLLM_LIB=$R/st-llm_amd/prev/libstllm_hip.so; else unset STLLM_LIB; fi
  timeout 600 python bench.py --config $cfg --steps 40 --warmup 3 --no-extra-legs --no-cpu-baseline --no-projection --no-roofline > $O/b.json 2>/dev/null
  python -c "import json; d=json.load(open('$O/b.json')); print('$cfg $lib', d['ms_per_step'], d['ms_per_step_blocks']['ms'], d['parity']['logits_max_abs_err'], d['telemetry']['sclk_mhz']['mean'])"
done; done; done 2>&1 | tee $O/bench_ab.log
