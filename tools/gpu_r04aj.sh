# kernel table of the training step (16 clips, mask + MVM) under rocprofv3, final round-4 code
mkdir -p gpurun_out/r4aj
export TMPDIR=/tmp; cd /tmp; (timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4aj/prof -- python $GRAFT_REPO_ROOT/tools/train_bench.py --layers 32 --batch 16 --steps 4 > $GRAFT_REPO_ROOT/gpurun_out/r4aj/train.log 2> $GRAFT_REPO_ROOT/gpurun_out/r4aj/prof.err); cd $GRAFT_REPO_ROOT
find gpurun_out/r4aj/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r4aj/kernel_stats.csv
find gpurun_out/r4aj/prof -name "*kernel_trace.csv" -delete
python tools/prof_summary.py gpurun_out/r4aj/kernel_stats.csv --div 5 --top 32 --title "training step, 16 clips x 16 frames, mask + MVM, bf16: rocprofv3 --kernel-trace --stats of tools/train_bench.py --steps 4 (5 steps incl. the first; totals / 5)" > gpurun_out/r4aj/train_step.md
grep "^step" gpurun_out/r4aj/train.log; head -45 gpurun_out/r4aj/train_step.md | cut -c1-170
