// How fast can ONE CU fill its LDS by LDS-DMA (global_load_lds, 16 bytes per lane), and does a second workgroup on the CU add to it?  (gfx950, round 5)
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/fill_probe tools/fill_probe.hip
//   run:   tools/fill_probe [iters]
// The question behind VERDICT r04 "Next round" #3(b) — two 4-wave workgroups per CU on smaller tiles, drifting apart so that one's epilogue hides
// under the other's K loop: a 128 x 128 (192 x 128) tile moves 32 (40) KiB per 64-deep K unit for 2.1 (3.1) MFLOP, so two such workgroups that keep the
// matrix pipes 90 % busy need ~118 (98) GB/s of fill per CU, where one 256 x 192 workgroup needs ~70.  Every GEMM of the library was OBSERVED to
// fill at 40-60 GB/s per CU (NOTES round 4); this probe measures the ceiling itself, with no MFMA in the way:
//   workgroups of W waves, each wave issuing 1-KiB pieces (64 lanes x 16 B) back to back with at most D pieces in flight (counted vmcnt wait: the
//   GEMMs' ring), G workgroups per CU (the launch's LDS size sets the residency), source either L2-resident (each workgroup re-reads its own
//   64 KiB) or streamed from HBM (distinct 1 KiB pieces over 1 GiB).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

template <int DEPTH>
__device__ __forceinline__ void wait_depth() {
  if constexpr (DEPTH >= 32) asm volatile("s_waitcnt vmcnt(31)" ::: "memory");
  else if constexpr (DEPTH >= 16) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
  else if constexpr (DEPTH >= 8) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
}

// every wave: `iters` pieces of 1 KiB; piece i of wave w of workgroup b comes from src + ((b * waves + w) * stride_w + (i % wrap) * 1024) % span
template <int DEPTH>
__global__ __launch_bounds__(512) void fill_kernel(const char* __restrict__ src, size_t span, size_t stride_w, int wrap, int iters, unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  char* dst = smem + wave * 2048;   // two 1-KiB landing slots per wave, alternating (nobody reads them)
  // span and wrap are powers of two: the address of a piece is two masks and an add (a runtime `%` would cost more than the load it feeds)
  const unsigned smask = (unsigned)(span - 1), wmask = (unsigned)wrap - 1;
  const unsigned base = (unsigned)(((size_t)(blockIdx.x * nw + wave) * stride_w) & smask) + lane * 16;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const unsigned off = (base + (((unsigned)(i + u) & wmask) << 10)) & smask;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + off),
                                       (__attribute__((address_space(3))) void*)(dst + (u & 1) * 1024), 16, 0, 0);
    }
    wait_depth<DEPTH>();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int DEPTH>
static void run(const char* label, const char* src, size_t span, size_t stride_w, int wrap, int waves, int wg_per_cu, int iters, unsigned long long* cyc) {
  const int cus = 256;
  const int lds = 160 * 1024 / wg_per_cu - (wg_per_cu > 1 ? 1024 : 0);   // residency by LDS size
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(fill_kernel<DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  int occ = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fill_kernel<DEPTH>, waves * 64, lds));
  const int grid = cus * wg_per_cu;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(fill_kernel<DEPTH>, dim3(grid), dim3(waves * 64), lds, 0, src, span, stride_w, wrap, iters / 8, cyc);   // warm-up
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(fill_kernel<DEPTH>, dim3(grid), dim3(waves * 64), lds, 0, src, span, stride_w, wrap, iters, cyc);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = (double)grid * waves * iters * 1024.0;
  printf("%-10s depth %2d  %d waves/wg x %d wg/CU (occupancy query %d): %7.1f GB/s per CU, %6.2f TB/s chip, %8.1f us\n", label, DEPTH, waves, wg_per_cu, occ,
         bytes / (ms * 1e-3) / cus / 1e9, bytes / (ms * 1e-3) / 1e12, ms * 1e3);
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 4096;
  const size_t big = (size_t)1 << 31;
  char* buf;
  CK(hipMalloc(&buf, big));
  CK(hipMemset(buf, 1, big));
  unsigned long long* cyc;
  CK(hipMalloc(&cyc, 4096 * sizeof(unsigned long long)));
  // L2-resident: each wave re-reads 4 KiB of its own (32 CUs x 16 waves x 4 KiB = 2 MiB per XCD, inside its 4 MiB L2)
  for (int waves : {4, 8}) {
    for (int wg : {1, 2, 4}) {
      if (waves * wg > 16) continue;
      run<8>("L2 hot", buf, (size_t)32 << 20, 4096, 4, waves, wg, iters, cyc);
      run<16>("L2 hot", buf, (size_t)32 << 20, 4096, 4, waves, wg, iters, cyc);
      run<32>("L2 hot", buf, (size_t)32 << 20, 4096, 4, waves, wg, iters, cyc);
    }
  }
  // streamed: every piece a new 1-KiB line group, waves far apart
  for (int waves : {4, 8}) {
    for (int wg : {1, 2}) {
      run<16>("HBM stream", buf, big, (size_t)iters * 1024, iters, waves, wg, iters, cyc);   // (iters a power of two)
      run<32>("HBM stream", buf, big, (size_t)iters * 1024, iters, waves, wg, iters, cyc);
    }
  }
  // the GEMM-like mix: rows shared by the workgroups of an XCD (every workgroup reads the SAME 2 MiB window: L2 hits after the first touch)
  for (int wg : {1, 2}) run<16>("shared 2MB", buf, (size_t)2 << 20, 0, 2048, 4, wg, iters, cyc);
  return 0;
}
