# gradient sink (wgrad straight into AdamW's flat buffer): training tests on the device + the 16-clip training step
mkdir -p gpurun_out/r4q

(timeout 600 python tools/train_bench.py --layers 32 --batch 16 --steps 8 2>&1 | tail -18) > gpurun_out/r4q/train_bench_b16.log; grep step gpurun_out/r4q/train_bench_b16.log
(timeout 600 python tools/train_bench.py --layers 32 --batch 16 --steps 8 --no-sink 2>&1 | tail -18) > gpurun_out/r4q/train_bench_b16_nosink.log; grep step gpurun_out/r4q/train_bench_b16_nosink.log
