# round-5 device script (rewritten per call; the invocations worth keeping are listed in profiles/README.md)
set -x
O=$GRAFT_REPO_ROOT/gpurun_out/r05b; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-extra-legs --no-projection --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/prof.err
cd $GRAFT_REPO_ROOT
T=$(ls $O/prof/*/*kernel_trace.csv | head -1); S=$(ls $O/prof/*/*kernel_stats.csv | head -1)
python tools/trace_gaps.py $T --steps 3 > $O/gaps.md
python tools/prof_summary.py $S --div 14 --top 40 --title "rocprofv3 --kernel-trace --stats of bench.py --steps 10 --warmup 3 (round 5, Q-Former stack entry + host plan first)" > $O/kernel_stats.md
cp $S $O/kernel_stats.csv; rm -rf $O/prof
head -30 $O/gaps.md
