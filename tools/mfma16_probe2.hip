// Exact issue rate of back-to-back independent MFMAs (inline asm: no compiler-made moves), gfx950, one wave per SIMD, 256 workgroups
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma16_probe2 tools/mfma16_probe2.hip && tools/mfma16_probe2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) int i32x4;

template <int MODE>
__global__ __launch_bounds__(256, 1) void probe(unsigned long long* out, int iters, const i32x4* in) {
  const int lane = threadIdx.x;
  i32x4 a0 = in[lane & 1023], a1 = in[(lane + 64) & 1023], b0 = in[(lane * 3 + 17) & 1023], b1 = in[(lane * 3 + 81) & 1023], b2 = in[(lane * 3 + 145) & 1023];
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if constexpr (MODE == 0 || MODE == 2) {
    f32x4 c[18];
    for (int i = 0; i < 18; ++i) c[i] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
      if constexpr (MODE == 2) asm volatile("s_barrier" ::: "memory");
      asm volatile(
          "v_mfma_f32_16x16x32_bf16 %0, %18, %20, %0\n v_mfma_f32_16x16x32_bf16 %1, %19, %20, %1\n v_mfma_f32_16x16x32_bf16 %2, %18, %21, %2\n"
          "v_mfma_f32_16x16x32_bf16 %3, %19, %21, %3\n v_mfma_f32_16x16x32_bf16 %4, %18, %22, %4\n v_mfma_f32_16x16x32_bf16 %5, %19, %22, %5\n"
          "v_mfma_f32_16x16x32_bf16 %6, %18, %20, %6\n v_mfma_f32_16x16x32_bf16 %7, %19, %20, %7\n v_mfma_f32_16x16x32_bf16 %8, %18, %21, %8\n"
          "v_mfma_f32_16x16x32_bf16 %9, %19, %21, %9\n v_mfma_f32_16x16x32_bf16 %10, %18, %22, %10\n v_mfma_f32_16x16x32_bf16 %11, %19, %22, %11\n"
          "v_mfma_f32_16x16x32_bf16 %12, %18, %20, %12\n v_mfma_f32_16x16x32_bf16 %13, %19, %20, %13\n v_mfma_f32_16x16x32_bf16 %14, %18, %21, %14\n"
          "v_mfma_f32_16x16x32_bf16 %15, %19, %21, %15\n v_mfma_f32_16x16x32_bf16 %16, %18, %22, %16\n v_mfma_f32_16x16x32_bf16 %17, %19, %22, %17\n"
          : "+a"(c[0]), "+a"(c[1]), "+a"(c[2]), "+a"(c[3]), "+a"(c[4]), "+a"(c[5]), "+a"(c[6]), "+a"(c[7]), "+a"(c[8]), "+a"(c[9]), "+a"(c[10]), "+a"(c[11]),
            "+a"(c[12]), "+a"(c[13]), "+a"(c[14]), "+a"(c[15]), "+a"(c[16]), "+a"(c[17])
          : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(b2));
    }
    float s = 0;
    for (int i = 0; i < 18; ++i) s += c[i][0] + c[i][3];
    if (s == 123.456f) out[1] = 1;
  } else {
    f32x16 c[9];
    for (int i = 0; i < 9; ++i)
      for (int r = 0; r < 16; ++r) c[i][r] = 0;
    for (int it = 0; it < iters; ++it) {
      if constexpr (MODE == 3) asm volatile("s_barrier" ::: "memory");
      asm volatile(
          "v_mfma_f32_32x32x16_bf16 %0, %9, %11, %0\n v_mfma_f32_32x32x16_bf16 %1, %10, %11, %1\n v_mfma_f32_32x32x16_bf16 %2, %9, %12, %2\n"
          "v_mfma_f32_32x32x16_bf16 %3, %10, %12, %3\n v_mfma_f32_32x32x16_bf16 %4, %9, %13, %4\n v_mfma_f32_32x32x16_bf16 %5, %10, %13, %5\n"
          "v_mfma_f32_32x32x16_bf16 %6, %9, %11, %6\n v_mfma_f32_32x32x16_bf16 %7, %10, %11, %7\n v_mfma_f32_32x32x16_bf16 %8, %9, %12, %8\n"
          : "+a"(c[0]), "+a"(c[1]), "+a"(c[2]), "+a"(c[3]), "+a"(c[4]), "+a"(c[5]), "+a"(c[6]), "+a"(c[7]), "+a"(c[8])
          : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(b2));
    }
    float s = 0;
    for (int i = 0; i < 9; ++i) s += c[i][0] + c[i][7];
    if (s == 123.456f) out[1] = 1;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = t1 - t0;
}

template <int MODE>
void run(const char* name, int iters, unsigned long long* d, const i32x4* in) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  probe<MODE><<<256, 256>>>(d, iters, in);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<MODE><<<256, 256>>>(d, iters, in);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long cyc = 0;
  (void)hipMemcpy(&cyc, d, 8, hipMemcpyDeviceToHost);
  const int per = (MODE == 0 || MODE == 2) ? 18 : 9;
  const double flop = 256.0 * 4 * (double)iters * per * 2.0 * 16384;
  printf("%-36s %7.1f ticks per iteration, %5.2f per MFMA, %7.1f ns per iteration, %7.1f TFLOP/s wall\n", name, (double)cyc / iters, (double)cyc / iters / per, ms * 1e6 / iters,
         flop / ms * 1e-9);
}

int main() {
  unsigned long long* d;
  (void)hipMalloc(&d, 16);
  i32x4* in;
  (void)hipMalloc(&in, 1024 * 16);
  unsigned h[4096];
  unsigned s = 12345u;
  for (int i = 0; i < 4096; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((s & 0x807f) | 0x3f80) | ((((s >> 16) & 0x807f) | 0x3f00) << 16); }
  (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  const int iters = 20000;
  run<0>("16x16x32 x 18 (asm)", iters, d, in);
  run<2>("16x16x32 x 18 (asm) + barrier", iters, d, in);
  run<1>("32x32x16 x 9 (asm)", iters, d, in);
  run<3>("32x32x16 x 9 (asm) + barrier", iters, d, in);
  return 0;
}
