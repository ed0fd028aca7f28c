// fp16 instantiations of the one-wave-per-SIMD GEMM (gemm_w4.inc)
#define STLLM_W4_TYPE f16_t
#define STLLM_W4_ENTRY stllm_gemm_w4_launch_f16
#include "gemm_w4.inc"
