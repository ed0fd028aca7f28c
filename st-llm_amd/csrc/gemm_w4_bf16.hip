// bf16 instantiations of the one-wave-per-SIMD GEMM (gemm_w4.inc)
#define STLLM_W4_TYPE bf16_t
#define STLLM_W4_ENTRY stllm_gemm_w4_launch_bf16
#define STLLM_W4_DEFINE_ESTIMATE 1
#include "gemm_w4.inc"
