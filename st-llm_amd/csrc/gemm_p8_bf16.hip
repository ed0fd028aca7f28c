// bf16 instantiations of the phased GEMM (gemm_p8.inc)
#define STLLM_P8_TYPE bf16_t
#define STLLM_P8_ENTRY stllm_gemm_p8_launch_bf16
#define STLLM_P8_DEFINE_ESTIMATE 1
#include "gemm_p8.inc"
