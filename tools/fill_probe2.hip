// Does the SHAPE of an LDS-DMA piece set the fill rate of a CU?  (gfx950, round 6)
//   hipcc --offload-arch=gfx950 -O3 -o tools/fill_probe2 tools/fill_probe2.hip && tools/fill_probe2
// The traffic of the tall-tile GEMM at 576 x 4096 x 4096 (256 workgroups: tile (tm, tn) reads A rows 144 tm .. + 143 and W rows 64 tn .. + 63, row stride
// 8 KiB, the whole K extent) with no MFMA and no LDS read in the way: 4 loader waves per workgroup, one workgroup per CU, every wave keeps DEPTH pieces
// (64 lanes x 16 B = 1 KiB) in flight.  A piece covers ROWS rows x (1024 / ROWS) contiguous bytes:
//   ROWS = 8: the library's piece (8 rows x 128 B = one 64-deep K unit), with or without the 16-byte chunk swizzle on the source address;
//   ROWS = 4 / 2 / 1: 256 / 512 / 1024 contiguous bytes per row (a K unit of 128 / 256 / 512 elements).
// Every mode moves the same bytes (208 rows x 8 KiB per workgroup).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

template <int ROWS, bool SWZ, int DEPTH>
__global__ __launch_bounds__(256) void fill2(const char* __restrict__ A, const char* __restrict__ W, int ldb, int kbytes, unsigned long long* cyc, int reps = 1) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.x, g = (b & 7) * 32 + (b >> 3), tm = g & 3, tn = g >> 2;
  constexpr int SEG = 1024 / ROWS;          // contiguous bytes per row and piece
  constexpr int LPR = 64 / ROWS;            // lanes per row
  const int rin = lane / LPR, cin = lane % LPR;
  int chunk = cin;
  if (SWZ) chunk = cin ^ (rin & 7);         // (ROWS == 8: the library's swizzle pattern, up to the row pairing)
  // 208 rows = 26 groups of 8 rows; wave w takes groups w, w + 4, ..; a group of 8 rows = 8 / ROWS pieces per SEG-wide k step
  char* dst = smem + wave * 8192;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  int issued = 0;
  for (int rep = 0; rep < reps; ++rep)
  for (int k = 0; k < kbytes; k += SEG) {
    for (int grp = wave; grp < 26; grp += 4) {
#pragma unroll
      for (int pp = 0; pp < 8 / ROWS; ++pp) {
        const int row = grp * 8 + pp * ROWS + rin;
        const char* base = row < 144 ? A + (size_t)(tm * 144 + row) * ldb : W + (size_t)(tn * 64 + row - 144) * ldb;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + k + chunk * 16),
                                         (__attribute__((address_space(3))) void*)(dst + (issued & 7) * 1024), 16, 0, 0);
        ++issued;
        if constexpr (DEPTH >= 32) asm volatile("s_waitcnt vmcnt(31)" ::: "memory");
        else if constexpr (DEPTH >= 16) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// the same traffic by plain 16-byte loads into registers (MODE 1: dropped there, 2: + ds_write_b128 into the LDS): is the LDS-DMA path the limit?
// NW waves per workgroup; wave w takes the 8-row groups w, w + NW, ..; per 128-byte K step it loads its groups (<= 7 loads of 1 KiB per wave in flight), then
// consumes them — the other waves of the CU cover the wait.
typedef __attribute__((ext_vector_type(4))) int i32x4;
template <int ROWS, int MODE, int NW>
__global__ __launch_bounds__(NW * 64) void fill_plain(const char* __restrict__ A, const char* __restrict__ W, int ldb, int kbytes, unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.x, g = (b & 7) * 32 + (b >> 3), tm = g & 3, tn = g >> 2;
  static_assert(ROWS == 8, "8-row pieces only");
  const int rin = lane >> 3, cin = lane & 7;
  constexpr int NG = (26 + NW - 1) / NW;
  const char* rowp[NG];
#pragma unroll
  for (int gi = 0; gi < NG; ++gi) {
    int grp = wave + NW * gi;
    grp = grp < 26 ? grp : 25;
    const int row = grp * 8 + rin;
    rowp[gi] = (row < 144 ? A + (size_t)(tm * 144 + row) * ldb : W + (size_t)(tn * 64 + row - 144) * ldb) + cin * 16;
  }
  char* dst = smem + wave * 8192 + lane * 16;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int k = 0; k < kbytes; k += 256) {   // two K steps per iteration: 2 NG loads in flight per wave
    i32x4 v[2 * NG];
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
      v[2 * gi] = *reinterpret_cast<const i32x4*>(rowp[gi] + k);
      v[2 * gi + 1] = *reinterpret_cast<const i32x4*>(rowp[gi] + k + 128);
    }
#pragma unroll
    for (int u = 0; u < 2 * NG; ++u) {
      if constexpr (MODE == 2) *reinterpret_cast<i32x4*>(dst + (u & 7) * 1024) = v[u];
      else asm volatile("" ::"v"(v[u]));
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int ROWS, int MODE, int NW>
static void run_plain(const char* label, const char* A, const char* W, int ldb, int kbytes, unsigned long long* cyc, int grid) {
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(fill_plain<ROWS, MODE, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((fill_plain<ROWS, MODE, NW>), dim3(grid), dim3(NW * 64), 128 * 1024, 0, A, W, ldb, kbytes, cyc);
  CK(hipEventRecord(e0));
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((fill_plain<ROWS, MODE, NW>), dim3(grid), dim3(NW * 64), 128 * 1024, 0, A, W, ldb, kbytes, cyc);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= 5;
  const double bytes = (double)grid * (double)(((26 + NW - 1) / NW) * NW * 8) * kbytes;   // (clamped groups re-load the last one)
  printf("%-34s %d waves ld %5d grid %3d: %7.1f GB/s per CU, %6.2f TB/s chip, %7.1f us\n", label, NW, ldb, grid, bytes / (ms * 1e-3) / grid / 1e9, bytes / (ms * 1e-3) / 1e12, ms * 1e3);
}

template <int ROWS, bool SWZ, int DEPTH>
static void run(const char* label, const char* A, const char* W, int ldb, int kbytes, unsigned long long* cyc, int grid, int reps = 1) {
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(fill2<ROWS, SWZ, DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((fill2<ROWS, SWZ, DEPTH>), dim3(grid), dim3(256), 128 * 1024, 0, A, W, ldb, kbytes, cyc, reps);
  CK(hipEventRecord(e0));
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((fill2<ROWS, SWZ, DEPTH>), dim3(grid), dim3(256), 128 * 1024, 0, A, W, ldb, kbytes, cyc, reps);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= 5;
  const double bytes = (double)grid * 208.0 * kbytes * reps;
  printf("%-34s depth %2d ld %5d grid %3d: %7.1f GB/s per CU, %6.2f TB/s chip, %7.1f us  (%5.1f cycles per piece and CU at 2.1 GHz)\n", label, DEPTH, ldb, grid, bytes / (ms * 1e-3) / grid / 1e9,
         bytes / (ms * 1e-3) / 1e12, ms * 1e3, ms * 1e-3 * 2.1e9 / (208.0 * kbytes * reps / 1024.0));
}

int main() {
  char *A, *W;
  const int ldb = 8192 + 128;
  CK(hipMalloc(&A, (size_t)576 * ldb + 65536));
  CK(hipMalloc(&W, (size_t)4096 * ldb + 65536));
  CK(hipMemset(A, 1, (size_t)576 * ldb));
  CK(hipMemset(W, 1, (size_t)4096 * ldb));
  unsigned long long* cyc;
  CK(hipMalloc(&cyc, 4096 * sizeof(unsigned long long)));
  for (int ld : {8192, 8192 + 128}) {
    if (ld != 8192) break;
    run<8, true, 16>("8 rows x 128 B, chunk swizzle", A, W, ld, 8192, cyc, 256);
    run<8, false, 16>("8 rows x 128 B, plain", A, W, ld, 8192, cyc, 256);
    run<4, false, 16>("4 rows x 256 B", A, W, ld, 8192, cyc, 256);
    run<2, false, 16>("2 rows x 512 B", A, W, ld, 8192, cyc, 256);
    run<1, false, 16>("1 row x 1 KiB", A, W, ld, 8192, cyc, 256);
  }
  run<8, true, 8>("8 rows x 128 B, chunk swizzle", A, W, 8192, 8192, cyc, 256);
  run<8, true, 32>("8 rows x 128 B, chunk swizzle", A, W, 8192, 8192, cyc, 256);
  run<1, false, 32>("1 row x 1 KiB", A, W, 8192, 8192, cyc, 256);
  // the same pieces re-reading a 1-KiB-wide K window 8 / 32 times: per XCD 0.6 MB of A + 0.5 MB of W = L2 hits after the first pass (per CU 208 KB: no L1 hits)
  run<8, true, 16>("8 x 128 B swizzle, L2-resident x8", A, W, 8192, 1024, cyc, 256, 8);
  run<8, true, 16>("8 x 128 B swizzle, L2-resident x32", A, W, 8192, 1024, cyc, 256, 32);
  run<8, true, 32>("8 x 128 B swizzle, L2-resident x32", A, W, 8192, 1024, cyc, 256, 32);
  run<1, false, 16>("1 x 1 KiB, L2-resident x32", A, W, 8192, 1024, cyc, 256, 32);
  run_plain<8, 1, 4>("plain loads 8 x 128 B -> VGPR", A, W, 8192, 8192, cyc, 256);
  run_plain<8, 1, 8>("plain loads 8 x 128 B -> VGPR", A, W, 8192, 8192, cyc, 256);
  run_plain<8, 1, 13>("plain loads 8 x 128 B -> VGPR", A, W, 8192, 8192, cyc, 256);
  run_plain<8, 2, 8>("plain loads 8 x 128 B -> ds_write", A, W, 8192, 8192, cyc, 256);
  run_plain<8, 2, 13>("plain loads 8 x 128 B -> ds_write", A, W, 8192, 8192, cyc, 256);
  run<8, true, 16>("8 rows x 128 B, chunk swizzle", A, W, 8192, 8192, cyc, 128);
  run<1, false, 16>("1 row x 1 KiB", A, W, 8192, 8192, cyc, 128);
  return 0;
}
