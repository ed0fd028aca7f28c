// Power-capped MFMA ceiling of an MI355X: a register-only loop of independent 32x32x16 bf16 MFMAs on random operands.
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/mfma_peak tools/mfma_peak.hip
//   run:   tools/mfma_peak [waves_per_simd 1|2] [iters] [zero_operands 0|1]
// Answers: how far below the 2.5 PFLOP/s "peak" does the chip run when NOTHING but the matrix pipes works?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) int i32x4;

template <int NACC>
__global__ __launch_bounds__(512) void mfma_loop(float* out, int iters, int zero) {
  const int lane = threadIdx.x;
  unsigned h = (lane * 2654435761u) ^ (blockIdx.x * 40503u);
  i32x4 a[4], b[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      h = h * 1664525u + 1013904223u;
      // random bf16 pairs in [-2, 2): sign + exponent 0x3f / 0x40 + random mantissa
      unsigned lo = (h & 0x807f) | 0x3f80, hi = ((h >> 16) & 0x807f) | 0x3f00;
      a[i][j] = zero ? 0 : (int)(lo | (hi << 16));
      h = h * 1664525u + 1013904223u;
      lo = (h & 0x807f) | 0x3f00; hi = ((h >> 16) & 0x807f) | 0x3f80;
      b[i][j] = zero ? 0 : (int)(lo | (hi << 16));
    }
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i)
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i & 3]), __builtin_bit_cast(bf16x8, b[(i >> 2) & 3]), acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 123.456f) out[0] = s;
}

int main(int argc, char** argv) {
  const int wps = argc > 1 ? atoi(argv[1]) : 1;
  const int iters = argc > 2 ? atoi(argv[2]) : 20000;
  const int zero = argc > 3 ? atoi(argv[3]) : 0;
  float* d;
  hipMalloc(&d, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  constexpr int NACC = 8;
  const int threads = 256 * wps;
  for (int rep = 0; rep < 6; ++rep) {
    hipEventRecord(e0);
    for (int l = 0; l < 10; ++l) mfma_loop<NACC><<<256, threads>>>(d, iters, zero);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = 10.0 * 256 * (threads / 64) * (double)iters * NACC * 2.0 * 32 * 32 * 16;
    printf("waves/SIMD %d zero %d: %.1f ms  %.1f TFLOP/s (%.2f of 2500)\n", wps, zero, ms, flop / ms * 1e-9, flop / ms * 1e-9 / 2500.0);
  }
  return 0;
}
