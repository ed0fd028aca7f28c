// What does one K step of the tall-tile GEMM cost without any memory traffic?  (gfx950, one wave per SIMD, round 6)
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma16_probe tools/mfma16_probe.hip && tools/mfma16_probe
// Per iteration: NMF independent MFMAs (MODE 0: v_mfma_f32_16x16x32_bf16, 1: v_mfma_f32_32x32x16_bf16) [+ one s_barrier when BAR]
// in 256-thread workgroups, one per CU.  Prints shader cycles per iteration (s_memtime of wave 0) and per MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) int i32x4;

template <int MODE, int NMF, bool BAR, bool PING>
__global__ __launch_bounds__(256, 1) void probe(unsigned long long* out, int iters, const i32x4* in) {
  const int lane = threadIdx.x;
  i32x4 a[2], b[9];
  for (int i = 0; i < 2; ++i) a[i] = in[(lane + 64 * i) & 1023];
  for (int i = 0; i < 9; ++i) b[i] = in[(lane * 3 + 64 * i + 17) & 1023];
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if constexpr (MODE == 0) {
    f32x4 acc[NMF], acc2[NMF];
    for (int i = 0; i < NMF; ++i) acc[i] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
      if constexpr (BAR) { asm volatile("s_barrier" ::: "memory"); }
      if constexpr (PING) {
#pragma unroll
        for (int i = 0; i < NMF; ++i) acc2[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i & 1]), __builtin_bit_cast(bf16x8, b[i % 9]), acc[i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (BAR) { asm volatile("s_barrier" ::: "memory"); }
#pragma unroll
        for (int i = 0; i < NMF; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i & 1]), __builtin_bit_cast(bf16x8, b[i % 9]), acc2[i], 0, 0, 0);
      } else {
#pragma unroll
        for (int i = 0; i < NMF; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i & 1]), __builtin_bit_cast(bf16x8, b[i % 9]), acc[i], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    float s = 0;
    for (int i = 0; i < NMF; ++i) s += acc[i][0] + acc[i][3];
    if (s == 123.456f) out[1] = 1;
  } else {
    f32x16 acc[NMF];
    for (int i = 0; i < NMF; ++i)
      for (int r = 0; r < 16; ++r) acc[i][r] = 0;
    for (int it = 0; it < iters; ++it) {
      if constexpr (BAR) { asm volatile("s_barrier" ::: "memory"); }
#pragma unroll
      for (int i = 0; i < NMF; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i & 1]), __builtin_bit_cast(bf16x8, b[i % 9]), acc[i], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    float s = 0;
    for (int i = 0; i < NMF; ++i) s += acc[i][0] + acc[i][7];
    if (s == 123.456f) out[1] = 1;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = t1 - t0;
}

template <int MODE, int NMF, bool BAR, bool PING>
void run(const char* name, int iters, unsigned long long* d, const i32x4* in) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  probe<MODE, NMF, BAR, PING><<<256, 256>>>(d, iters, in);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<MODE, NMF, BAR, PING><<<256, 256>>>(d, iters, in);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long cyc = 0;
  hipMemcpy(&cyc, d, 8, hipMemcpyDeviceToHost);
  const int per = PING ? 2 * NMF : NMF;
  const double flop = 256.0 * 4 * (double)iters * per * 2.0 * (MODE == 0 ? 16 * 16 * 32 : 32 * 32 * 16);
  printf("%-44s %7.1f ticks per iteration, %5.2f per MFMA, %7.1f ns per iteration, %7.1f TFLOP/s wall\n", name, (double)cyc / iters, (double)cyc / iters / per,
         ms * 1e6 / iters, flop / ms * 1e-9);
}

int main() {
  unsigned long long* d;
  hipMalloc(&d, 16);
  i32x4* in;
  hipMalloc(&in, 1024 * 16);
  unsigned h[4096];
  unsigned s = 12345u;
  for (int i = 0; i < 4096; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((s & 0x807f) | 0x3f80) | ((((s >> 16) & 0x807f) | 0x3f00) << 16); }
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  const int iters = 20000;
  run<0, 18, false, false>("16x16x32 x 18, no barrier", iters, d, in);
  run<0, 18, true, false>("16x16x32 x 18, barrier", iters, d, in);
  run<0, 18, true, true>("16x16x32 x 18, barrier, acc ping-pong (x2)", iters, d, in);
  run<0, 18, false, true>("16x16x32 x 18, no barrier, ping-pong (x2)", iters, d, in);
  run<0, 36, true, false>("16x16x32 x 36, barrier", iters, d, in);
  run<1, 9, false, false>("32x32x16 x 9, no barrier", iters, d, in);
  run<1, 9, true, false>("32x32x16 x 9, barrier", iters, d, in);
  run<1, 12, true, false>("32x32x16 x 12, barrier", iters, d, in);
  return 0;
}
