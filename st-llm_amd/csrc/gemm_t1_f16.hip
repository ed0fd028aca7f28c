// fp16 instantiations of the tall-tile one-round GEMM (gemm_t1.inc)
#define STLLM_T1_TYPE f16_t
#define STLLM_T1_ENTRY stllm_gemm_t1_launch_f16
#include "gemm_t1.inc"
