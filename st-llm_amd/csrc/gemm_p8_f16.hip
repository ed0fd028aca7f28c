// fp16 instantiations of the phased GEMM (gemm_p8.inc)
#define STLLM_P8_TYPE f16_t
#define STLLM_P8_ENTRY stllm_gemm_p8_launch_f16
#include "gemm_p8.inc"
