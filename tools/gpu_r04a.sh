#!/bin/bash
# round-4 GPU call A: split verify mode (kernel + model tests), new fixtures' tests, default bench line, rocprof of the split step
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04a
mkdir -p $O
cd $R
python - > $O/smi_probe.log 2>&1 <<'PY'
import torch, json
import bench
t = bench.Telemetry(0)
print("src", t.src, getattr(t, "err", None))
if t.src:
    print("read", t._read())
    a = t._smi
    try: print(a.amdsmi_get_clock_info(t._h, a.AmdSmiClkType.GFX))
    except Exception as e: print("clk err", e)
    try: print(a.amdsmi_get_power_info(t._h))
    except Exception as e: print("pw err", e)
pr = torch.cuda.get_device_properties(0)
print(pr.name, getattr(pr, "pci_bus_id", None), getattr(pr, "pci_device_id", None), getattr(pr, "pci_domain_id", None), pr.multi_processor_count)
PY
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -s -k "bf16x3 or rope" > $O/t_kernels.log 2>&1; echo "kernels rc $?" >> $O/t_kernels.log
timeout 1500 python -m pytest tests/test_model_gpu.py -q -x -s -k "flagship or split_verify or full_size_configs" > $O/t_model.log 2>&1; echo "model rc $?" >> $O/t_model.log
export LD_LIBRARY_PATH=$R/st-llm_amd:/opt/rocm/lib:${LD_LIBRARY_PATH:-}
for w in 33 32; do timeout 120 tools/gemm_harness 50 6 0 1 $w 0 700; done > $O/harness_llm_qkv.log 2>&1
timeout 900 python bench.py --steps 30 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?" >> $O/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_x3 -- python $R/bench.py --dtype bf16x3 --steps 3 --warmup 2 --no-extra-legs --no-cpu-baseline --no-roofline > $O/prof_x3.log 2>&1
cd $R
find $O/prof_x3 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/x3_kernel_stats.csv
find $O/prof_x3 -type f ! -name "*kernel_stats.csv" -size +1M -delete
tail -n 4 $O/harness_llm_qkv.log; tail -n 5 $O/t_kernels.log $O/t_model.log $O/bench.err
head -c 1500 $O/bench.json
