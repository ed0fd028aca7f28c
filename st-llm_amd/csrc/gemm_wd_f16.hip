// fp16 instantiations of the W-direct GEMM (gemm_wd.inc)
#define STLLM_WD_TYPE f16_t
#define STLLM_WD_ENTRY stllm_gemm_wd_launch_f16
#include "gemm_wd.inc"
