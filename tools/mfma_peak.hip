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

// the same loop with NR ds_read_b128 per NACC MFMAs feeding the operands (LDS filled with random bf16): what do the LDS reads of
// a GEMM inner loop cost in clock (power), with no global traffic at all?
template <int NACC, int NR>
__global__ __launch_bounds__(512) void mfma_lds_loop(float* out, int iters) {
  __shared__ i32x4 lds[8192];   // 128 KiB
  const int lane = threadIdx.x;
  unsigned h = (lane * 2654435761u) ^ (blockIdx.x * 40503u);
  for (int i = lane; i < 8192; i += blockDim.x) {
    i32x4 v;
    for (int j = 0; j < 4; ++j) {
      h = h * 1664525u + 1013904223u;
      v[j] = (int)(((h & 0x807f) | 0x3f80) | ((((h >> 16) & 0x807f) | 0x3f00) << 16));
    }
    lds[i] = v;
  }
  __syncthreads();
  i32x4 f[8];
  for (int i = 0; i < 8; ++i) f[i] = lds[(lane + 64 * i) & 8191];
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  unsigned off = lane;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f[i & 7]), __builtin_bit_cast(bf16x8, f[(i + 3) & 7]), acc[i], 0, 0, 0);
      if (i < NR) {
        off = (off + 64 * 5) & 8191;
        f[(i + 5) & 7] = lds[off];   // conflict-free: consecutive lanes, consecutive 16-byte slots
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 123.456f) out[0] = s;
}

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
  const int nr = argc > 4 ? atoi(argv[4]) : -1;   // >= 0: LDS-fed variant with nr ds_read_b128 per 8 MFMAs
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    for (int l = 0; l < 10; ++l) {
      if (nr < 0) mfma_loop<NACC><<<256, threads>>>(d, iters, zero);
      else if (nr == 0) mfma_lds_loop<NACC, 0><<<256, threads>>>(d, iters);
      else if (nr == 2) mfma_lds_loop<NACC, 2><<<256, threads>>>(d, iters);
      else if (nr == 4) mfma_lds_loop<NACC, 4><<<256, threads>>>(d, iters);
      else if (nr == 6) mfma_lds_loop<NACC, 6><<<256, threads>>>(d, iters);
      else mfma_lds_loop<NACC, 8><<<256, threads>>>(d, iters);
    }
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = 10.0 * 256 * (threads / 64) * (double)iters * NACC * 2.0 * 32 * 32 * 16;
    printf("waves/SIMD %d zero %d lds_reads_per_8 %d: %.1f ms  %.1f TFLOP/s (%.2f of 2500)\n", wps, zero, nr, ms, flop / ms * 1e-9, flop / ms * 1e-9 / 2500.0);
  }
  return 0;
}
