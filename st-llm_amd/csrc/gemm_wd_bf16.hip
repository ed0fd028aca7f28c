// bf16 instantiations of the W-direct GEMM (gemm_wd.inc)
#define STLLM_WD_TYPE bf16_t
#define STLLM_WD_ENTRY stllm_gemm_wd_launch_bf16
#include "gemm_wd.inc"
