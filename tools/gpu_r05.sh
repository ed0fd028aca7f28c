# round-5 device script (rewritten per call; the invocations worth keeping are listed in profiles/README.md)
O=gpurun_out/r05d; mkdir -p $O
python tools/gemm_bench.py --only llm_qkv,llm_o,llm_gu,llm_down,lm_head --rows 8224,288 --audit > $O/audit_288.log 2>&1
python tools/gemm_bench.py --only llm_qkv,llm_o,llm_gu,llm_down,lm_head --rows 8224,292 --audit > $O/audit_292.log 2>&1
python tools/gemm_bench.py --only llm_qkv,llm_o,llm_gu,llm_down,lm_head --rows 8224,384 --audit > $O/audit_384.log 2>&1
python tools/gemm_bench.py --only llm_qkv,llm_o,llm_gu,llm_down,lm_head --rows 8224,192 --audit > $O/audit_192.log 2>&1
grep -h "^llm\|^lm_head\|<--" $O/audit_*.log
