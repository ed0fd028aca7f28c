// Stand-alone probe of a ONE-WAVE-PER-SIMD GEMM main loop for gfx950 (no torch, no library): 256-thread workgroups =
// 4 waves as 2(M) x 2(N), each wave owns a (32*WM) x (32*WN) block of a (64*WM) x (64*WN) tile in up to 256 accumulator
// registers (the whole 512-entry register file belongs to one wave per SIMD, so nobody shares the SIMD's matrix pipe and no
// hand-over barriers are needed).  K in 64-element (128-byte) units through a 2-buffer LDS ring filled by LDS-DMA; ONE
// workgroup barrier per unit.  Prints time / TFLOP/s per shape and checks the result against a naive fp32 kernel.
//   build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/w4_probe tools/w4_probe.hip
//   run:    tools/w4_probe [reps]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <cmath>
#include <vector>

#define CK(x)                                                                                                         \
  do {                                                                                                                \
    hipError_t e_ = (x);                                                                                              \
    if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); }  \
  } while (0)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) int i32x4;

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + k;
}
__device__ __forceinline__ void tile_coords(int w, int tiles_m, int tiles_n, int& tm, int& tn) {
  const int gsz = 8 * tiles_n;
  const int g = w / gsz, rem = w - g * gsz;
  const int first = g * 8;
  const int gm = min(8, tiles_m - first);
  tn = rem / gm;
  tm = first + (rem - tn * gm);
}
__device__ __forceinline__ void glds16(const void* g, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ uint16_t pack_bf16(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }

#define W4_BAR()                                \
  do {                                          \
    __builtin_amdgcn_sched_barrier(0);          \
    asm volatile("s_barrier" ::: "memory");     \
    __builtin_amdgcn_sched_barrier(0);          \
  } while (0)

// DMODE: 0 = all LDS-DMA of the next-next unit issued between the MFMAs of k-step 2, 1 = spread over k-steps 2 and 3,
//        2 = issued in one burst right after the barrier (before the MFMAs of k-step 2)
template <int WM, int WN, int DMODE>
__global__ __launch_bounds__(256) void w4_kernel(const char* __restrict__ A, const char* __restrict__ W, uint16_t* __restrict__ C, int M, int N, int K,
                                                 int tiles_m, int tiles_n, unsigned long long* dbg) {
  constexpr int BM = 64 * WM, BN = 64 * WN;
  constexpr int kBuf = (BM + BN) * 128;
  constexpr int NP = (BM + BN) / 32;   // LDS-DMA pieces (8 rows x 128 B) per wave per unit
  constexpr int NPA = BM / 32;         // ... of which A rows
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const unsigned long long t_entry = dbg ? __builtin_amdgcn_s_memtime() : 0;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int nk = K / 64;
  const int G = gridDim.x, g = xcd_remap(blockIdx.x, G);
  const int T = tiles_m * tiles_n;
  const int n_tiles = (T - g + G - 1) / G;   // tiles g, g + G, ...
  if (n_tiles <= 0) return;
  const int n_units = n_tiles * nk;
  const int64_t lda_b = (int64_t)K * 2, ldw_b = (int64_t)K * 2;

  // ---- LDS-DMA plan: piece p = wave + 4 j covers rows 8p .. 8p+7 of the combined [A rows | W rows] tile ----
  unsigned goff[NP];
  auto plan = [&](int tile) {
    int ln = lane;
    asm volatile("" : "+v"(ln));
    int tm, tn;
    tile_coords(tile, tiles_m, tiles_n, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int r = (wave + 4 * j) * 8 + (ln >> 3);
      const int lc = (ln & 7) ^ ((r >> 1) & 7);
      if (j < NPA) {
        int gr = m0 + r;
        gr = gr < M ? gr : M - 1;
        goff[j] = (unsigned)((int64_t)gr * lda_b) + lc * 16;
      } else {
        int gr = n0 + r - BM;
        gr = gr < N ? gr : N - 1;
        goff[j] = (unsigned)((int64_t)gr * ldw_b) + lc * 16;
      }
    }
  };
  int l_tile = g, l_k = 0;   // next unit to issue
  plan(l_tile);
  const char* a_k = A;
  const char* w_k = W;
  auto issue_piece = [&](int j, int lu) {   // j compile-time after unrolling
    char* dst = smem + (lu & 1) * kBuf + (wave + 4 * j) * 1024;
    glds16((j < NPA ? a_k : w_k) + goff[j], dst);
  };
  auto unit_issued = [&]() {   // bookkeeping after the last piece of a unit
    a_k += 128;
    w_k += 128;
    if (++l_k == nk) {
      l_k = 0;
      l_tile += G;
      a_k = A;
      w_k = W;
      if (l_tile < T) plan(l_tile);
    }
  };

  // ---- fragment read addresses ----
  typedef __attribute__((address_space(3))) const char* lds_ptr;
  const int sw = (li >> 1) & 7;
  const unsigned ko = (unsigned)(((lh ^ sw) & 7) << 4) + (unsigned)(uintptr_t)(lds_ptr)smem;
  const unsigned aoff0 = (unsigned)((wm * 32 * WM + li) * 128) + ko;
  const unsigned boff0 = (unsigned)((BM + wn * 32 * WN + li) * 128) + ko;
  auto lds16 = [&](unsigned addr, int kk, int imm) {
    return *reinterpret_cast<const __attribute__((address_space(3))) i32x4*>((lds_ptr)(uintptr_t)(addr ^ (unsigned)(kk << 5)) + imm);
  };

  f32x16 acc[WM][WN];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  };
  i32x4 fa[3][WM], fb[3][WN];
  auto rd = [&](auto sc, int buf, int kk) {
    constexpr int s = decltype(sc)::value;
#pragma unroll
    for (int i = 0; i < WM; ++i) fa[s][i] = lds16(aoff0 + buf * kBuf, kk, i * 4096);
#pragma unroll
    for (int j = 0; j < WN; ++j) fb[s][j] = lds16(boff0 + buf * kBuf, kk, j * 4096);
  };

  int c_tile = g;
  auto epilogue = [&]() {
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int li = ln & 31, lh = ln >> 5;
    int tm, tn;
    tile_coords(c_tile, tiles_m, tiles_n, tm, tn);
    const int rowb = tm * BM + wm * 32 * WM, colb = tn * BN + wn * 32 * WN;
    uint16_t* s16 = reinterpret_cast<uint16_t*>(smem + 2 * kBuf + wave * 4096);
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int jp = 0; jp < WN / 2; ++jp) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int lrow = (r & 3) + 8 * (r >> 2) + 4 * lh;
          const int prow = lrow ^ ((lrow >> 2) & 1);
          s16[prow * 64 + li] = pack_bf16(acc[i][2 * jp][r]);
          s16[prow * 64 + 32 + li] = pack_bf16(acc[i][2 * jp + 1][r]);
        }
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) {
          const int prow = t4 * 8 + (ln >> 3), lrow = prow ^ ((prow >> 2) & 1), c8 = (ln & 7) * 8;
          const i32x4 v = *reinterpret_cast<const i32x4*>(reinterpret_cast<const char*>(s16) + prow * 128 + c8 * 2);
          const int row = rowb + i * 32 + lrow, col = colb + jp * 64 + c8;
          if (row < M && col < N) *reinterpret_cast<i32x4*>(C + (int64_t)row * N + col) = v;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0x0f70);
  };

  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;

  // ---- prologue: units 0 and 1 in flight, unit 0 landed, k-step 0 fragments of unit 0 in set 0 ----
#pragma unroll
  for (int j = 0; j < NP; ++j) issue_piece(j, 0);
  unit_issued();
  if (n_units > 1) {
#pragma unroll
    for (int j = 0; j < NP; ++j) issue_piece(j, 1);
    unit_issued();
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  W4_BAR();
  rd(I0{}, 0, 0);
  __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0), visible to the compiler's scoreboard

#define MM(S, I, J) acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[S][I]), __builtin_bit_cast(bf16x8, fb[S][J]), acc[I][J], 0, 0, 0)

  int it = 0;
  unsigned long long stall = 0, t_first = 0, t_last = 0, t_epi = 0;
  for (int ct = 0; ct < n_tiles; ++ct) {
  zero_acc();
  for (int ck = 0; ck < nk; ++ck, ++it) {
    const int buf = it & 1;
    // Hand-ordered stream: every MFMA is followed by at most two LDS reads or one LDS-DMA and a scheduling fence, so the
    // emitted order is the source order.  Fragment sets: S0 = k-steps 0 / 2, S1 = k-step 1, S2 = k-step 3.
    // read order within a k-step = order of first use by the row-major MFMA sweep: a0 b0 b1 .. b(WN-1) a1 a2 ..
    auto rd1 = [&](auto sc, int bufx, int kk, int idx) {   // idx-th fragment of a k-step in first-use order
      constexpr int s_ = decltype(sc)::value;
      if (idx == 0) fa[s_][0] = lds16(aoff0 + bufx * kBuf, kk, 0);
      else if (idx <= WN) fb[s_][idx - 1] = lds16(boff0 + bufx * kBuf, kk, (idx - 1) * 4096);
      else if (idx < WM + WN) fa[s_][idx - WN] = lds16(aoff0 + bufx * kBuf, kk, (idx - WN) * 4096);
    };
    constexpr int NF = WM + WN, NM = WM * WN;
    // P0: MFMA k0 (S0); reads k1 -> S1 then k3 -> S2, two per MFMA
    {
      int q = 0;
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) {
          MM(0, i, j);
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int f = 2 * q + u;
            if (f < NF) rd1(I1{}, buf, 1, f);
            else if (f < 2 * NF) rd1(I2{}, buf, 3, f - NF);
          }
          __builtin_amdgcn_sched_barrier(0);
          ++q;
        }
#pragma unroll
      for (int f = 2 * NM; f < 2 * NF; ++f) { if (f < NF) rd1(I1{}, buf, 1, f); else rd1(I2{}, buf, 3, f - NF); }
    }
    __builtin_amdgcn_sched_barrier(0);
    // P1: MFMA k1 (S1); reads k2 -> S0, one per MFMA
    {
      int q = 0;
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) {
          MM(1, i, j);
          if (q < NF) rd1(I0{}, buf, 2, q);
          __builtin_amdgcn_sched_barrier(0);
          ++q;
        }
#pragma unroll
      for (int f = NM; f < NF; ++f) rd1(I0{}, buf, 2, f);
    }
    __builtin_amdgcn_sched_barrier(0);
    // every LDS read of this unit's buffer has been issued: retire them, make sure the next unit landed, one barrier
    unsigned long long t1 = 0, t2 = 0;
    if (dbg) t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    W4_BAR();
    if (dbg) { t2 = __builtin_amdgcn_s_memtime(); stall += t2 - t1; if (it == 0) t_first = t1; t_last = t1; }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    const bool more = it + 2 < n_units;
    // P2 / P3: MFMA k2 (S0) and k3 (S2); the NP LDS-DMA pieces of unit it + 2 spread over the 2 NM MFMAs (DMODE 1) or over
    // the first NP MFMAs (DMODE 0); reads of the next unit's k0 -> S0 behind the first MFMAs of P3
    {
      int q = 0;
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) {
          MM(0, i, j);
          if constexpr (DMODE == 0) { if (q < NP) issue_piece(q, it); }
          else { if ((q * NP) / (2 * NM) != ((q + 1) * NP) / (2 * NM)) issue_piece((q * NP) / (2 * NM), it); }
          __builtin_amdgcn_sched_barrier(0);
          ++q;
        }
    }
    {
      int q = 0;
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) {
          MM(2, i, j);
          if (q < NF) rd1(I0{}, buf ^ 1, 0, q);   // (reads stale LDS after the last unit: harmless)
          if constexpr (DMODE == 0) { if (NM + q < NP) issue_piece(NM + q, it); }
          else { const int qq = NM + q; if ((qq * NP) / (2 * NM) != ((qq + 1) * NP) / (2 * NM)) issue_piece((qq * NP) / (2 * NM), it); }
          __builtin_amdgcn_sched_barrier(0);
          ++q;
        }
#pragma unroll
      for (int f = NM; f < NF; ++f) rd1(I0{}, buf ^ 1, 0, f);
    }
    if (more) unit_issued();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): k-step 0 of the next unit landed long ago; tells the compiler so
  }
  unsigned long long te = 0;
  if (dbg) te = __builtin_amdgcn_s_memtime();
  epilogue();
  if (dbg) t_epi += __builtin_amdgcn_s_memtime() - te;
  c_tile += G;
  }
  if (dbg && lane == 0) {
    unsigned long long* d = dbg + ((int64_t)g * 4 + wave) * 4;
    d[0] = (unsigned long long)n_units; d[1] = stall; d[2] = t_last - t_first; d[3] = t_epi;
    if (wave == 0) { dbg[4096 + g * 2] = t_first; dbg[4096 + g * 2 + 1] = __builtin_amdgcn_s_memtime(); }
  }
}

__global__ void ref_kernel(const uint16_t* A, const uint16_t* W, float* C, int M, int N, int K) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
  if (n >= N || m >= M) return;
  float s = 0.f;
  for (int k = 0; k < K; ++k) {
    const float a = __builtin_bit_cast(float, (uint32_t)A[(int64_t)m * K + k] << 16), w = __builtin_bit_cast(float, (uint32_t)W[(int64_t)n * K + k] << 16);
    s = fmaf(a, w, s);
  }
  C[(int64_t)m * N + n] = s;
}

static uint16_t f2bf(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static float bf2f(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
static uint32_t rng_state = 12345;
static float urand() {
  rng_state = rng_state * 1664525u + 1013904223u;
  return (float)((rng_state >> 8) & 0xffff) / 32768.0f - 1.0f;
}

template <int WM, int WN, int DMODE>
void run(const char* name, int M, int N, int K, int reps, const uint16_t* dA, const uint16_t* dW, uint16_t* dC, const float* dRef, hipStream_t st) {
  constexpr int BM = 64 * WM, BN = 64 * WN;
  const int lds = 2 * (BM + BN) * 128 + 4 * 4096;
  auto kern = w4_kernel<WM, WN, DMODE>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  const int T = tiles_m * tiles_n, grid = T < 256 ? T : 256;
  CK(hipMemsetAsync(dC, 0xff, (size_t)M * N * 2, st));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, (const char*)dA, (const char*)dW, dC, M, N, K, tiles_m, tiles_n, (unsigned long long*)nullptr);
  CK(hipGetLastError());
  CK(hipStreamSynchronize(st));
  std::vector<uint16_t> h((size_t)M * N);
  std::vector<float> r((size_t)M * N);
  CK(hipMemcpy(h.data(), dC, h.size() * 2, hipMemcpyDeviceToHost));
  CK(hipMemcpy(r.data(), dRef, r.size() * 4, hipMemcpyDeviceToHost));
  size_t nbad = 0, first = 0;
  double maxabs = 0;
  for (size_t i = 0; i < h.size(); ++i) {
    const double x = r[i], y = bf2f(h[i]), d = fabs(x - y);
    if (!(d <= 2e-2 + 1e-2 * fabs(x))) { if (!nbad) first = i; ++nbad; }
    if (d > maxabs) maxabs = d;
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float best = 1e30f, sum = 0;
  for (int rr = 0; rr < 3; ++rr) {
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, (const char*)dA, (const char*)dW, dC, M, N, K, tiles_m, tiles_n, (unsigned long long*)nullptr);
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const float us = ms * 1000.f / reps;
    best = us < best ? us : best;
    sum += us;
  }
  {
    unsigned long long* ddbg;
    CK(hipMalloc(&ddbg, (4096 + 512) * 8));
    CK(hipMemset(ddbg, 0, (4096 + 512) * 8));
    hipEvent_t d0, d1;
    CK(hipEventCreate(&d0)); CK(hipEventCreate(&d1));
    CK(hipEventRecord(d0, st));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, (const char*)dA, (const char*)dW, dC, M, N, K, tiles_m, tiles_n, ddbg);
    CK(hipEventRecord(d1, st));
    CK(hipStreamSynchronize(st));
    float dms; CK(hipEventElapsedTime(&dms, d0, d1));
    std::vector<unsigned long long> hd(4096 + 512);
    CK(hipMemcpy(hd.data(), ddbg, hd.size() * 8, hipMemcpyDeviceToHost));
    double su = 0, ss = 0, sp = 0, se = 0; int nw = 0;
    for (int w = 0; w < grid * 4; ++w) { if (hd[w * 4] < 2) continue; su += hd[w * 4]; ss += hd[w * 4 + 1]; sp += hd[w * 4 + 2]; se += hd[w * 4 + 3]; ++nw; }
    unsigned long long tmin = ~0ull, tmax = 0;
    for (int b = 0; b < grid; ++b) { if (hd[4096 + 2 * b] && hd[4096 + 2 * b] < tmin) tmin = hd[4096 + 2 * b]; if (hd[4096 + 2 * b + 1] > tmax) tmax = hd[4096 + 2 * b + 1]; }
    printf("   [dbg] kernel span %llu ticks, event %.1f us => %.0f ticks/us\n", tmax - tmin, dms * 1000.f, (double)(tmax - tmin) / (dms * 1000.f));
    if (nw) printf("   [dbg] per unit: period %.0f ticks, wait+barrier stall %.0f ticks; epilogue %.0f ticks per wave-launch (memtime = 100 MHz? ticks)\n", sp / (su - nw), ss / su, se / nw);
    hipFree(ddbg);
  }
  const double tf = 2.0 * M * N * K * 1e-6;
  printf("%-8s tile %dx%d dmode %d  M=%5d N=%5d K=%5d  tiles %4d  mean %8.1f us (%7.1f TF)  best %8.1f us (%7.1f TF)  maxabs %.3g bad %zu", name, BM, BN, DMODE, M, N,
         K, T, sum / 3, tf / (sum / 3), best, tf / best, maxabs, nbad);
  if (nbad) printf(" first bad row %zu col %zu", first / N, first % N);
  printf(" %s\n", nbad ? "FAIL" : "ok");
  fflush(stdout);
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 20;
  struct Shape { const char* name; int M, N, K; };
  const Shape shapes[] = {{"sq4096", 4096, 4096, 4096}, {"vit_fc1", 4112, 6144, 1408}, {"vit_fc2", 4112, 1408, 6144}, {"llm_gu", 576, 22016, 4096}, {"small", 300, 384, 192}};
  hipStream_t st;
  CK(hipStreamCreate(&st));
  for (const Shape& s : shapes) {
    const int M = s.M, N = s.N, K = s.K;
    std::vector<uint16_t> hA((size_t)M * K), hW((size_t)N * K);
    for (auto& v : hA) v = f2bf(urand());
    for (auto& v : hW) v = f2bf(urand() * 0.05f);
    uint16_t *dA, *dW, *dC;
    float* dRef;
    CK(hipMalloc(&dA, hA.size() * 2));
    CK(hipMalloc(&dW, hW.size() * 2));
    CK(hipMalloc(&dC, (size_t)M * N * 2));
    CK(hipMalloc(&dRef, (size_t)M * N * 4));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(ref_kernel, dim3((N + 255) / 256, M), dim3(256), 0, st, dA, dW, dRef, M, N, K);
    CK(hipStreamSynchronize(st));
    run<4, 4, 0>(s.name, M, N, K, reps, dA, dW, dC, dRef, st);
    run<4, 4, 1>(s.name, M, N, K, reps, dA, dW, dC, dRef, st);
    run<3, 4, 0>(s.name, M, N, K, reps, dA, dW, dC, dRef, st);
    run<3, 4, 1>(s.name, M, N, K, reps, dA, dW, dC, dRef, st);
    run<4, 2, 1>(s.name, M, N, K, reps, dA, dW, dC, dRef, st);
    hipFree(dA); hipFree(dW); hipFree(dC); hipFree(dRef);
  }
  return 0;
}
