// TEST-ONLY: the phased GEMM (gemm_p8.inc: LDS-DMA rings, hardware registers, cross-workgroup flags) cannot run in the emulator;
// the emulated library reports it as unsupported so that stllm_gemm falls back to the 128x128 kernels of gemm.hip, which can.
#include "gemm_common.h"

int stllm_gemm_p8_launch_bf16(int, int, const sg::GemmParams&, hipStream_t) { return STLLM_ERR_UNSUPPORTED; }
int stllm_gemm_p8_launch_f16(int, int, const sg::GemmParams&, hipStream_t) { return STLLM_ERR_UNSUPPORTED; }
float stllm_gemm_p8_estimate_us(int, int, int, int, int* miw) { if (miw) *miw = 4; return 1.0e30f; }
extern "C" int stllm_gemm_plan(int, int, int, int, int, int*) { return STLLM_ERR_UNSUPPORTED; }
int stllm_gemm_w4_launch_bf16(int, int, const sg::GemmParams&, hipStream_t) { return STLLM_ERR_UNSUPPORTED; }
int stllm_gemm_w4_launch_f16(int, int, const sg::GemmParams&, hipStream_t) { return STLLM_ERR_UNSUPPORTED; }
// round 6: the tall-tile and the W-direct kernels (inline-asm MFMAs / global loads, loader waves) are hardware-only too
int stllm_gemm_t1_launch_bf16(int, int, const sg::GemmParams&, hipStream_t) { return STLLM_ERR_UNSUPPORTED; }
int stllm_gemm_t1_launch_f16(int, int, const sg::GemmParams&, hipStream_t) { return STLLM_ERR_UNSUPPORTED; }
int stllm_gemm_wd_launch_bf16(int, int, const sg::GemmParams&, hipStream_t) { return STLLM_ERR_UNSUPPORTED; }
int stllm_gemm_wd_launch_f16(int, int, const sg::GemmParams&, hipStream_t) { return STLLM_ERR_UNSUPPORTED; }
float stllm_gemm_w4_estimate_us(int, int, int, int, int* shape, int* split) { if (shape) *shape = 44; if (split) *split = 1; return 1.0e30f; }
// profile.cpp (HIP events) is not part of the emulated library either
int stllm_prof_begin(const stllm_gemm_args*, void*) { return -1; }
void stllm_prof_end(int, int, const stllm_gemm_args*, void*) {}
extern "C" int stllm_gemm_w4_plan(int, int, int, int, int, int*) { return STLLM_ERR_UNSUPPORTED; }
