# rocprofv3 kernel trace + stats of a short bench run on the GPU box (round 6):  bash tools/gpu_profile.sh <outdir> <streams> [extra bench flags]
#   /usr/local/graft/bin/gpurun --timeout 900 -- "bash tools/gpu_profile.sh prof_s1 1"     (outputs under gpurun_out/<outdir>)
O=gpurun_out/${1:-prof}; S=${2:-1}; shift 2; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
(timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 4 --streams $S --no-cpu-baseline --no-extra-legs --no-projection "$@" > $GRAFT_REPO_ROOT/$O/prof_bench.json 2> $GRAFT_REPO_ROOT/$O/prof.err)
cd $GRAFT_REPO_ROOT
DIV=25; if [ "$S" != "1" ]; then DIV=37; fi     # steps in the trace: 4 warm-up + 1 calibration + 20 timed (+ 2 + 10 of the single-stream bracket with streams > 1)
find $O/prof -name "*kernel_trace.csv" | head -1 | xargs -I{} python tools/trace_gaps.py {} --steps 3 > $O/gaps.md; head -8 $O/gaps.md
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c "python tools/prof_summary.py {} --div $DIV --top 40 > $O/kernel_stats.md; cp {} $O/kernel_stats.csv"
rm -rf $O/prof
head -c 300 $O/prof_bench.json; echo; head -30 $O/kernel_stats.md
