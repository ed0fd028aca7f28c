# round-5 device script (rewritten per call; the invocations worth keeping are listed in profiles/README.md)
O=$GRAFT_REPO_ROOT/gpurun_out/r05e; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $GRAFT_REPO_ROOT/bench.py --dtype bf16x3 --steps 4 --warmup 2 --no-extra-legs --no-projection --no-cpu-baseline --no-roofline > $O/bench_x3_under_rocprof.json 2> $O/prof.err
cd $GRAFT_REPO_ROOT
S=$(ls $O/prof/*/*kernel_stats.csv | head -1)
python tools/prof_summary.py $S --div 6 --top 40 --title "rocprofv3 --kernel-trace --stats of bench.py --dtype bf16x3 --steps 4 --warmup 2 (round 5 start)" > $O/x3_kernel_stats.md
cp $S $O/x3_kernel_stats.csv; rm -rf $O/prof
head -40 $O/x3_kernel_stats.md | cut -c1-200
python -m pytest tests/test_model_gpu.py -x -q -k "no_qformer" 2>&1 | tail -3
