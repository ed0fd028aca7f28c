// bf16 instantiations of the tall-tile one-round GEMM (gemm_t1.inc)
#define STLLM_T1_TYPE bf16_t
#define STLLM_T1_ENTRY stllm_gemm_t1_launch_bf16
#include "gemm_t1.inc"
