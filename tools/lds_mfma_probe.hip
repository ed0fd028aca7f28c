// What does ONE LDS fragment read cost the matrix pipe of its SIMD?  (gfx950, one wave per SIMD, round 4)
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/lds_mfma_probe tools/lds_mfma_probe.hip
//   run:   tools/lds_mfma_probe [iters]
// A loop of 8 independent v_mfma_f32_32x32x16_bf16 per iteration with NR "fragment reads" (16 bytes per lane each) placed one per MFMA gap,
// the read being: 0 ds_read_b128 | 1 two ds_read_b64 | 2 one ds_read2_b64 | 3 two ds_read_b64_tr_b16 | 4 ds_read_b128 into an AGPR tuple |
// 5 four ds_read_b32.  Prints shader cycles per MFMA (s_memtime of one wave) and wall-clock TFLOP/s: the in-kernel timelines of gemm_w4.inc
// show ~14 cycles of lost matrix-pipe time per ds_read_b128 at every tile shape (profiles/r04_w4_128x256_qkv.md) — is that the instruction,
// the bytes, or the destination?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(2))) int i32x2;

template <int MODE>
__device__ __forceinline__ i32x4 frag_read(unsigned a16, unsigned a8) {
  i32x4 v;
  if constexpr (MODE == 0) {
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a16));
  } else if constexpr (MODE == 1) {
    i32x2 lo, hi;
    asm volatile("ds_read_b64 %0, %1" : "=v"(lo) : "v"(a8));
    asm volatile("ds_read_b64 %0, %1 offset:4096" : "=v"(hi) : "v"(a8));
    v = i32x4{lo[0], lo[1], hi[0], hi[1]};
  } else if constexpr (MODE == 2) {
    asm volatile("ds_read2_b64 %0, %1 offset0:0 offset1:64" : "=v"(v) : "v"(a8));
  } else if constexpr (MODE == 3) {
    i32x2 lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a8));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:4096" : "=v"(hi) : "v"(a8));
    v = i32x4{lo[0], lo[1], hi[0], hi[1]};
  } else if constexpr (MODE == 4) {
    asm volatile("ds_read_b128 %0, %1" : "=a"(v) : "v"(a16));
  } else {
    int x0, x1, x2, x3;
    const unsigned a4 = a8 >> 1;
    asm volatile("ds_read_b32 %0, %1" : "=v"(x0) : "v"(a4));
    asm volatile("ds_read_b32 %0, %1 offset:1024" : "=v"(x1) : "v"(a4));
    asm volatile("ds_read_b32 %0, %1 offset:2048" : "=v"(x2) : "v"(a4));
    asm volatile("ds_read_b32 %0, %1 offset:3072" : "=v"(x3) : "v"(a4));
    v = i32x4{x0, x1, x2, x3};
  }
  return v;
}

template <int MODE, int NR>
__global__ __launch_bounds__(256) void probe(unsigned long long* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) i32x4 lds[];   // 64 KiB
  const int lane = threadIdx.x;
  unsigned h = (lane * 2654435761u) ^ (blockIdx.x * 40503u);
  for (int i = lane; i < 4096; i += blockDim.x) {
    i32x4 v;
    for (int j = 0; j < 4; ++j) {
      h = h * 1664525u + 1013904223u;
      v[j] = (int)(((h & 0x807f) | 0x3f80) | ((((h >> 16) & 0x807f) | 0x3f00) << 16));
    }
    lds[i] = v;
  }
  __syncthreads();
  i32x4 f[8];
  for (int i = 0; i < 8; ++i) f[i] = lds[(lane + 64 * i) & 4095];
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const unsigned wl = lane & 63;
  unsigned a16 = wl * 16, a8 = wl * 8;   // conflict-free for every mode (consecutive lanes, consecutive slots)
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  i32x4 g[4];
  for (int i = 0; i < 4; ++i) g[i] = lds[(lane * 3 + 64 * i + 17) & 4095];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {   // the read behind MFMA i overwrites the A operand that MFMA just consumed; its next use is one iteration away
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f[i]), __builtin_bit_cast(bf16x8, g[i & 3]), acc[i], 0, 0, 0);
      if (i < NR) f[i] = frag_read<MODE>(a16 + 1024 * i, a8 + 512 * i);
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    a16 ^= 16384;
    a8 ^= 8192;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 8; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 123.456f) out[1] = (unsigned long long)s;
  if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = t1 - t0;
}

template <int MODE, int NR>
void run(const char* name, int iters, unsigned long long* d) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(probe<MODE, NR>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  probe<MODE, NR><<<256, 256, 65536>>>(d, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int l = 0; l < 5; ++l) probe<MODE, NR><<<256, 256, 65536>>>(d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long cyc = 0;
  hipMemcpy(&cyc, d, 8, hipMemcpyDeviceToHost);
  const double flop = 5.0 * 256 * 4 * (double)iters * 8 * 2.0 * 32 * 32 * 16;
  printf("%-26s reads per 8 MFMAs %d: %6.2f cycles per MFMA (+%5.2f per read)  %7.1f TFLOP/s wall\n", name, NR, (double)cyc / iters / 8.0,
         NR ? ((double)cyc / iters / 8.0 - 32.0) * 8.0 / NR : 0.0, flop / ms * 1e-9);
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  unsigned long long* d;
  hipMalloc(&d, 16);
  run<0, 0>("no reads", iters, d);
  run<0, 2>("ds_read_b128", iters, d);
  run<0, 4>("ds_read_b128", iters, d);
  run<0, 6>("ds_read_b128", iters, d);
  run<0, 8>("ds_read_b128", iters, d);
  run<4, 4>("ds_read_b128 -> AGPR", iters, d);
  run<4, 8>("ds_read_b128 -> AGPR", iters, d);
  run<1, 4>("2 x ds_read_b64", iters, d);
  run<1, 8>("2 x ds_read_b64", iters, d);
  run<2, 4>("ds_read2_b64", iters, d);
  run<2, 8>("ds_read2_b64", iters, d);
  run<3, 4>("2 x ds_read_b64_tr_b16", iters, d);
  run<3, 8>("2 x ds_read_b64_tr_b16", iters, d);
  run<5, 4>("4 x ds_read_b32", iters, d);
  run<5, 8>("4 x ds_read_b32", iters, d);
  return 0;
}
