// 256x256 "phased" stream-K GEMM for gfx950 (bf16 / fp16):  C[M,N] = epilogue(A[M,K] @ W[N,K]^T)
//
// One 512-thread workgroup per CU = 8 waves as 2(M) x 4(N); every wave owns a 128 x 64 output block
// (4 x 2 MFMA 32x32 fragments = 128 accumulator registers).  K is walked in 64-element (128-byte) tiles.
//
// LDS ring (128 KiB): 2 K-tile buffers x 4 HALF-TILES of 16 KiB = 128 rows x 128 B, swizzled like gemm.hip
// (physical 16-B chunk = logical chunk ^ ((row >> 1) & 7), applied to the LDS-DMA SOURCE address):
//     A0 / A1 = the first / second 64 rows of BOTH wave rows   (local row = wm * 64 + r)
//     B0 / B1 = the first / second 32 W rows of all FOUR wave columns (local row = wn * 32 + r)
// in buffer order [A0 | B0 | B1 | A1].  A K-tile is consumed in four quadrant phases
//     q0: read A0 (8 x ds_read_b128) + B0 (4)  -> acc[0..1][0]      q1: read B1 (4)      -> acc[0..1][1]
//     q2: read A1 (8)                          -> acc[2..3][1]      q3: (B0 kept in VGPRs) -> acc[2..3][0]
// and every phase is   { ds_reads ; 2 x global_load_lds (one half-tile share) ; s_waitcnt vmcnt(8) ; s_barrier ;
//                        s_waitcnt lgkmcnt(0) ; 8 x MFMA 32x32x16 ; s_barrier }.
// The two wave rows (wm = 0 / 1: one wave of each per SIMD) run STAGGERED by one barrier, so while one group
// issues its 8 MFMAs (256 cycles of the SIMD's matrix pipe) the other group's LDS reads and LDS-DMA issue run
// underneath; the matrix pipe is handed back and forth at every barrier.
//
// Load stream: half-tiles are issued in consumption order S = A0(0) B0(0) B1(0) A1(0) A0(1) ...; phase q of unit t
// issues S[4t + q + 6], i.e. 6 half-tiles (12 KiB per wave) stay in flight and every load has >= 4 phases to land.
//   RAW: `vmcnt(8)` in phase q retires S[4t + q + 2] in the issuing wave; the first reader (the other group is one
//        barrier behind) touches that half-tile one full phase later, after both groups passed a barrier.
//   WAR: a half-tile is re-issued >= 3 barriers after the phase that read it last (its readers executed
//        lgkmcnt(0) after their first barrier of that phase).
// Stream-K: the flattened (tile, K-tile) space is cut into G equal ranges (G <= #CUs, all resident).  A range that
// starts inside a tile writes its raw accumulators to a per-workgroup slab in REGISTER ORDER (float4 per lane,
// fully coalesced, no LDS) and publishes it with an agent-scope release + epoch flag; the workgroup that owns
// K-tile 0 of the tile polls, acquires, adds the slabs in workgroup order (deterministic) and runs the epilogue.
// Epilogue: per-wave 4 KiB LDS scratch (no workgroup barriers): accumulators are transposed through it so that
// global stores are 16 bytes per lane on whole 128-byte rows; bias / GELU / RoPE / SwiGLU are applied on the
// accumulator registers, the fp32 residual add on the row-major side.
#include <type_traits>

#include "gemm_common.h"

namespace {
using namespace sg;

constexpr int kNT = 512;
constexpr int BM = 256, BN = 256;
constexpr int kHalf = 128 * kRowBytes;   // 16 KiB
constexpr int kBuf = 4 * kHalf;          // 64 KiB per K-tile
constexpr int kRing = 2 * kBuf;          // 128 KiB
constexpr int kScr = 4096;               // per-wave epilogue scratch
constexpr int kLds = kRing + 8 * kScr;   // 160 KiB
constexpr int kSlabFloats = BM * BN;     // 256 KiB per workgroup

#define P8_BAR()                                    \
  do {                                              \
    __builtin_amdgcn_sched_barrier(0);              \
    asm volatile("s_barrier" ::: "memory");         \
    __builtin_amdgcn_sched_barrier(0);              \
  } while (0)
#define P8_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

template <int V> using IC = std::integral_constant<int, V>;

template <typename T, int EPI, int ACT, bool OF32>
__global__ __launch_bounds__(kNT) void gemm_p8_kernel(const GemmParams p) {
  constexpr int EB = 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int li = lane & 31, lh = lane >> 5;

  const int ntiles = p.tiles_m * p.tiles_n;
  const int nk = p.K / 64;
  const int64_t total = (int64_t)ntiles * nk;
  const int G = gridDim.x;
  const int g = xcd_remap(blockIdx.x, G);
  auto range_start = [&](int gg) -> int64_t { return total * gg / G; };
  const int64_t u_begin = range_start(g), u_end = range_start(g + 1);
  const int n_units = (int)(u_end - u_begin);
  if (n_units <= 0) return;

  unsigned* flags = reinterpret_cast<unsigned*>(p.ws);
  float* slabs = reinterpret_cast<float*>(p.ws + kSkFlagBytes);
  // timeline instrumentation (gemm_debug bit 16): lane 0 of the workgroup appends (tag, cycle) pairs to the buffer passed
  // in `frames` (unused by every epilogue this kernel supports): 64 x 8-byte slots per workgroup, slot 0 = count
  unsigned long long* dbg = (p.debug & 16) ? reinterpret_cast<unsigned long long*>(const_cast<float*>(p.frames)) + (int64_t)g * 64 : nullptr;
  int dbg_n = 1;
  auto stamp = [&](int tag) {
    if (dbg && tid == 0 && dbg_n < 64) {
      dbg[dbg_n] = ((unsigned long long)tag << 56) | (__builtin_amdgcn_s_memtime() & 0x00ffffffffffffffull);
      ++dbg_n;
      dbg[0] = (unsigned long long)dbg_n;
    }
  };
  stamp(1);

  // ---- load plan: per half-tile h (0 A0, 1 B0, 2 B1, 3 A1) two 1-KiB pieces per wave (pieces wave, wave + 8) -------
  const char* gsrc[4][2];
  auto a_row = [&](int gr) -> const char* {
    gr = gr < p.M ? gr : p.M - 1;
    int64_t off = (int64_t)gr * p.lda_b;
    if (p.a_rpb > 0) { const int bb = gr / p.a_rpb; off = (int64_t)bb * p.a_bs_b + (int64_t)(gr - bb * p.a_rpb) * p.lda_b; }
    return p.A + off;
  };
  auto plan = [&](int tile, int k0) {
    int tm, tn;
    tile_coords(tile, p.tiles_m, p.tiles_n, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int lr = (wave + 8 * j) * 8 + (lane >> 3);   // local row of the half-tile
      const int lc = (lane & 7) ^ ((lr >> 1) & 7);       // logical chunk fetched into physical chunk lane & 7
      const int64_t kofs = (int64_t)k0 * kRowBytes + lc * 16;
      const int ar = m0 + (lr >> 6) * 128 + (lr & 63);
      gsrc[0][j] = a_row(ar) + kofs;
      gsrc[3][j] = a_row(ar + 64) + kofs;
      int wr0 = n0 + (lr >> 5) * 64 + (lr & 31), wr1 = wr0 + 32;
      wr0 = wr0 < p.N ? wr0 : p.N - 1;
      wr1 = wr1 < p.N ? wr1 : p.N - 1;
      gsrc[1][j] = p.W + (int64_t)wr0 * p.ldw_b + kofs;
      gsrc[2][j] = p.W + (int64_t)wr1 * p.ldw_b + kofs;
    }
  };
  int l_tile = (int)(u_begin / nk), l_k = (int)(u_begin - (int64_t)l_tile * nk);
  plan(l_tile, l_k);
  // issue this wave's share of half-tile h of unit lu (caller guarantees lu < n_units)
  auto issue = [&](auto hc, int lu) {
    constexpr int h = decltype(hc)::value;
    char* dst = smem + (lu & 1) * kBuf + h * kHalf + wave * 1024;
    glds16(gsrc[h][0], dst);
    glds16(gsrc[h][1], dst + 8 * 1024);
    gsrc[h][0] += kRowBytes;
    gsrc[h][1] += kRowBytes;
    if constexpr (h == 3) {   // unit lu completely issued
      if (++l_k == nk) {
        l_k = 0;
        ++l_tile;
        if (lu + 1 < n_units) plan(l_tile, 0);
      }
    }
  };

  // ---- fragment read offsets (bytes from smem): row term + swizzled chunk; buffer bit toggled per unit -------------
  const int sw = (li >> 1) & 7;
  unsigned aoff[4], boff[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    const unsigned ko = (unsigned)(((kk * 2 + lh) ^ sw) << 4);
    aoff[kk] = (unsigned)((wm * 64 + li) * kRowBytes) + ko;
    boff[kk] = (unsigned)((wn * 32 + li) * kRowBytes) + ko;
  }
  auto lds16 = [&](unsigned off) { return *reinterpret_cast<const i32x4*>(smem + off); };

  auto out_off = [&](int row) -> int64_t {
    if (p.o_rpb > 0) { const int bb = row / p.o_rpb; return (int64_t)bb * p.o_bs + (int64_t)(row - bb * p.o_rpb) * p.ldo; }
    return (int64_t)row * p.ldo;
  };

  f32x16 acc[4][2];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  };
  zero_acc();
  i32x4 af[2][4], bf[2][4];

  int c_tile = (int)(u_begin / nk), c_k = (int)(u_begin - (int64_t)c_tile * nk);
  int seg_k0 = c_k;

  // ================= segment end: partial -> slab, or (gather partials +) fused epilogue ============================
  auto segment_end = [&]() {
    int tm, tn;
    tile_coords(c_tile, p.tiles_m, p.tiles_n, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    const bool contrib = seg_k0 > 0;      // tile was started by a lower-numbered workgroup
    int ncontrib = 0;                       // partials this workgroup must add (it owns K-tile 0)
    stamp(contrib ? 3 : 4);
    if (!contrib && c_k < nk) {
      const int64_t tile_end = (int64_t)(c_tile + 1) * nk;
      while (g + 1 + ncontrib < G && range_start(g + 1 + ncontrib) < tile_end) ++ncontrib;
      if (wave == 0) {   // one wave polls (relaxed, bounded), then one agent-scope acquire for the whole CU
        for (int q = 0; q < ncontrib; ++q) {
          unsigned spins = 0;
          while (__hip_atomic_load(&flags[g + 1 + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)p.epoch &&
                 ++spins < (1u << 22))
            __builtin_amdgcn_s_sleep(8);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        wait_vmcnt<0>();
      }
      P8_BAR();   // wave 0 is in the leading group: nobody starts its epilogue before the acquire
      stamp(5);
    }
    const int wslab = wave * (kSlabFloats / 8);
    if (contrib) {
      float* my = slabs + (int64_t)g * kSlabFloats + wslab;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
            *reinterpret_cast<f32x4*>(my + (((i * 2 + j) * 4 + q) * 64 + lane) * 4) = v;
          }
      wait_vmcnt<0>();   // this wave's slab stores reached L2
      stamp(6);
      P8_BAR();
      // tid 256 is in the TRAILING group: when it passes this barrier every wave of both groups has drained
      if (tid == 256) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        wait_vmcnt<0>();
        __hip_atomic_store(&flags[g], (unsigned)p.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      stamp(10);
      return;
    }
    for (int q = 0; q < ncontrib; ++q) {
      const float* sl = slabs + (int64_t)(g + 1 + q) * kSlabFloats + wslab;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(sl + (((i * 2 + j) * 4 + qq) * 64 + lane) * 4);
            acc[i][j][4 * qq] += v[0]; acc[i][j][4 * qq + 1] += v[1]; acc[i][j][4 * qq + 2] += v[2]; acc[i][j][4 * qq + 3] += v[3];
          }
    }
    if (ncontrib) stamp(7);

    // ---- fused epilogue through this wave's private scratch ---------------------------------------------------------
    char* scr = smem + kRing + wave * kScr;
    constexpr bool kOutT = (EPI == STLLM_EPI_SWIGLU || EPI == STLLM_EPI_ROPE);
    constexpr bool f32out = (EPI == STLLM_EPI_RESID) || (!kOutT && OF32);
    const int colw = n0 + wn * 64;          // first column of this wave
    const int rowb = m0 + wm * 128;         // first row of this wave
    auto gf4 = [&](const float* ptr) { return *reinterpret_cast<const f32x4*>(ptr); };

    if constexpr (f32out) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          float* sf = reinterpret_cast<float*>(scr);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int lrow = (r & 3) + 8 * (r >> 2) + 4 * lh;
            sf[(lrow ^ ((lrow >> 2) & 1)) * 32 + li] = acc[i][j][r];
          }
#pragma unroll
          for (int t4 = 0; t4 < 4; ++t4) {
            const int prow = t4 * 8 + (lane >> 3), lrow = prow ^ ((prow >> 2) & 1), c4 = (lane & 7) * 4;
            f32x4 v = *reinterpret_cast<const f32x4*>(sf + prow * 32 + c4);
            const int row = rowb + i * 32 + lrow, col = colw + j * 32 + c4;
            if (row >= p.M || col >= p.N) continue;
            if (p.bias) v += gf4(p.bias + col);
            if constexpr (EPI == STLLM_EPI_RESID) {
              v += gf4(p.resid + (int64_t)row * p.ldr + col);
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                if constexpr (ACT == STLLM_ACT_GELU) v[e] = gelu_erf(v[e]);
                if constexpr (ACT == STLLM_ACT_RELU) v[e] = fmaxf(v[e], 0.0f);
              }
            }
            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + out_off(row) + col) = v;
          }
        }
    } else if constexpr (EPI == STLLM_EPI_SWIGLU) {
      // the wave's 64 columns are one [32 gate | 32 up] group: silu(g) * u on the registers -> 32 outputs
      const int gcol = colw + li;
      const float bg = (p.bias && gcol < p.N) ? p.bias[gcol] : 0.0f, bu = (p.bias && gcol + 32 < p.N) ? p.bias[gcol + 32] : 0.0f;
      uint16_t* s16 = reinterpret_cast<uint16_t*>(scr);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int lrow = (r & 3) + 8 * (r >> 2) + 4 * lh;
          s16[(lrow ^ ((lrow >> 2) & 1)) * 32 + li] = Elem<T>::pack(silu_f(acc[i][0][r] + bg) * (acc[i][1][r] + bu));
        }
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
          const int prow = t2 * 16 + (lane >> 2), lrow = prow ^ ((prow >> 2) & 1), c8 = (lane & 3) * 8;
          const i32x4 v = *reinterpret_cast<const i32x4*>(scr + prow * 64 + c8 * 2);
          const int row = rowb + i * 32 + lrow, col = (colw >> 1) + c8;
          if (row >= p.M || colw >= p.N) continue;
          *reinterpret_cast<i32x4*>(reinterpret_cast<char*>(p.out) + (out_off(row) + col) * EB) = v;
        }
      }
    } else {
      // 16-bit outputs: STORE (bias, act) and ROPE (bias, rotate-half on the [x_lo | x_hi] halves = the two fragments)
      const int c0 = colw + li;
      const float b0 = (p.bias && c0 < p.N) ? p.bias[c0] : 0.0f, b1 = (p.bias && c0 + 32 < p.N) ? p.bias[c0 + 32] : 0.0f;
      uint16_t* s16 = reinterpret_cast<uint16_t*>(scr);
      bool rope = false;
      int fi = 0;
      if constexpr (EPI == STLLM_EPI_ROPE) {
        rope = colw < p.rope_cols;
        fi = ((colw >> 6) & 1) * 32 + li;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int lrow = (r & 3) + 8 * (r >> 2) + 4 * lh;
          float x = acc[i][0][r] + b0, y = acc[i][1][r] + b1;
          if constexpr (EPI == STLLM_EPI_ROPE) {
            if (rope) {
              int row = rowb + i * 32 + lrow;
              row = row < p.M ? row : p.M - 1;
              const int pos = row % p.rope_seq;
              const float c = p.aux0[pos * 64 + fi], s = p.aux1[pos * 64 + fi];
              const float xr = x * c - y * s;
              y = y * c + x * s;
              x = xr;
            }
          } else {
            if constexpr (ACT == STLLM_ACT_GELU) { x = gelu_erf(x); y = gelu_erf(y); }
            if constexpr (ACT == STLLM_ACT_RELU) { x = fmaxf(x, 0.0f); y = fmaxf(y, 0.0f); }
          }
          const int prow = lrow ^ ((lrow >> 2) & 1);
          s16[prow * 64 + li] = Elem<T>::pack(x);
          s16[prow * 64 + 32 + li] = Elem<T>::pack(y);
        }
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) {
          const int prow = t4 * 8 + (lane >> 3), lrow = prow ^ ((prow >> 2) & 1), c8 = (lane & 7) * 8;
          const i32x4 v = *reinterpret_cast<const i32x4*>(scr + prow * 128 + c8 * 2);
          const int row = rowb + i * 32 + lrow, col = colw + c8;
          if (row >= p.M || col >= p.N) continue;
          *reinterpret_cast<i32x4*>(reinterpret_cast<char*>(p.out) + (out_off(row) + col) * EB) = v;
        }
      }
    }
    // stores and loads share vmcnt and may retire out of order with each other: drain, so that the counted
    // vmcnt(8) of the following phases only ever sees LDS-DMA loads
    wait_vmcnt<0>();
    stamp(8);
  };

  // ================= prologue: S[0..5] in flight, A0(0)/B0(0) landed ===================================================
  issue(IC<0>{}, 0); issue(IC<1>{}, 0); issue(IC<2>{}, 0); issue(IC<3>{}, 0);
  if (n_units >= 2) { issue(IC<0>{}, 1); issue(IC<1>{}, 1); wait_vmcnt<8>(); } else wait_vmcnt<0>();
  P8_BAR();
  stamp(2);
  if (wm == 1) P8_BAR();   // stagger: the wm = 1 group runs one barrier behind

// The MFMA builtins are pure values to the optimiser: without the two pins below it sinks them past the barriers.
// Pin 1 (after the LDS wait) re-defines the fragments, pin 2 consumes the accumulators -> the 8 MFMAs stay between them.
#define P8_MFMA(MI_, NI_)                                                                                        \
  do {                                                                                                           \
    asm volatile("" : "+v"(af[0][0]), "+v"(af[0][1]), "+v"(af[0][2]), "+v"(af[0][3]), "+v"(af[1][0]),           \
                 "+v"(af[1][1]), "+v"(af[1][2]), "+v"(af[1][3]), "+v"(bf[NI_][0]), "+v"(bf[NI_][1]),             \
                 "+v"(bf[NI_][2]), "+v"(bf[NI_][3]));                                                            \
    __builtin_amdgcn_s_setprio(1);                                                                               \
    _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) {                                                           \
      acc[2 * MI_][NI_] = Elem<T>::mfma(af[0][kk], bf[NI_][kk], acc[2 * MI_][NI_]);                              \
      acc[2 * MI_ + 1][NI_] = Elem<T>::mfma(af[1][kk], bf[NI_][kk], acc[2 * MI_ + 1][NI_]);                      \
    }                                                                                                            \
    asm volatile("" : "+v"(acc[2 * MI_][NI_]), "+v"(acc[2 * MI_ + 1][NI_]));                                     \
    __builtin_amdgcn_s_setprio(0);                                                                               \
  } while (0)
#define P8_ISSUE(H_, LU_)                                                   \
  do {                                                                      \
    const int lu_ = (LU_);                                                  \
    if (lu_ < n_units) { issue(IC<H_>{}, lu_); wait_vmcnt<8>(); }           \
    else wait_vmcnt<0>();                                                   \
  } while (0)

  for (int it = 0; it < n_units; ++it) {
    // ---- phase 0: A0 + B0 -> quadrant (0, 0) ------------------------------------------------------------------------
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) bf[0][kk] = lds16(boff[kk] + 1 * kHalf);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      af[0][kk] = lds16(aoff[kk]);
      af[1][kk] = lds16(aoff[kk] + 32 * kRowBytes);
    }
    P8_ISSUE(2, it + 1);
    P8_BAR();
    P8_LGKM0();
    P8_MFMA(0, 0);
    P8_BAR();
    // ---- phase 1: B1 -> quadrant (0, 1) -----------------------------------------------------------------------------
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) bf[1][kk] = lds16(boff[kk] + 2 * kHalf);
    P8_ISSUE(3, it + 1);
    P8_BAR();
    P8_LGKM0();
    P8_MFMA(0, 1);
    P8_BAR();
    // ---- phase 2: A1 -> quadrant (1, 1) -----------------------------------------------------------------------------
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      af[0][kk] = lds16(aoff[kk] + 3 * kHalf);
      af[1][kk] = lds16(aoff[kk] + 3 * kHalf + 32 * kRowBytes);
    }
    P8_ISSUE(0, it + 2);
    P8_BAR();
    P8_LGKM0();
    P8_MFMA(1, 1);
    P8_BAR();
    // ---- phase 3: (B0 still in registers) -> quadrant (1, 0) ----------------------------------------------------------
    P8_ISSUE(1, it + 2);
    P8_BAR();
    P8_MFMA(1, 0);
    P8_BAR();

#pragma unroll
    for (int kk = 0; kk < 4; ++kk) { aoff[kk] ^= (unsigned)kBuf; boff[kk] ^= (unsigned)kBuf; }
    ++c_k;
    if (c_k < nk && it + 1 < n_units) continue;
    segment_end();
    if (c_k == nk) { c_k = 0; ++c_tile; }
    seg_k0 = c_k;
    zero_acc();
  }
  if (wm == 0) P8_BAR();   // pairs with the trailing group's last barrier
  wait_vmcnt<0>();
  stamp(9);
#undef P8_MFMA
#undef P8_ISSUE
}

template <typename T, int EPI, int ACT, bool OF32>
int launch_p8(const GemmParams& p0, hipStream_t stream) {
  GemmParams p = p0;
  p.tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.N + BN - 1) / BN;
  auto kern = gemm_p8_kernel<T, EPI, ACT, OF32>;
  static int max_wg = 0;
  if (max_wg == 0) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    if (e != hipSuccess) {
      stllm_set_error("stllm_gemm(p8): cannot reserve %d bytes of LDS: %s", kLds, hipGetErrorString(e));
      return STLLM_ERR_HIP;
    }
    // every workgroup must be RESIDENT (finalisers wait for contributors): size the grid from the occupancy query
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    (void)hipGetDevice(&dev);
    (void)hipGetDeviceProperties(&prop, dev);
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, kNT, kLds);
    if (per_cu < 1) {
      stllm_set_error("stllm_gemm(p8): kernel does not fit a CU (occupancy query returned %d)", per_cu);
      return STLLM_ERR_HIP;
    }
    max_wg = prop.multiProcessorCount;   // one workgroup per CU (160 KiB LDS)
  }
  const int64_t total = (int64_t)p.tiles_m * p.tiles_n * (p.K / 64);
  const int grid = total < max_wg ? (int)total : max_wg;
  if (p.ws == nullptr || p.ws_bytes < kSkFlagBytes + (int64_t)grid * kSlabFloats * 4) {
    stllm_set_error("stllm_gemm(p8): workspace too small (%lld bytes)", (long long)p.ws_bytes);
    return STLLM_ERR_BAD_SHAPE;
  }
  p.epoch = stllm_sk_next_epoch();
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kNT), kLds, stream, p);
  STLLM_CHECK_LAUNCH("stllm_gemm(p8)");
  {
    static const char* kEpi[] = {"STORE", "RESID", "SWIGLU", "ROPE", "PATCH"};
    static char name[96];
    static bool named = false;
    if (!named) {
      snprintf(name, sizeof(name), "gemm_p8_kernel<%s,%s,%d,%d>", std::is_same<T, bf16_t>::value ? "bf16_t" : "f16_t", kEpi[EPI], ACT, (int)OF32);
      named = true;
    }
    stllm_set_last_kernel(name);
  }
  return STLLM_OK;
}

template <typename T>
int dispatch_p8(int epilogue, const GemmParams& p, hipStream_t stream) {
  switch (epilogue) {
    case STLLM_EPI_STORE:
      if (p.act == STLLM_ACT_NONE) return p.out_is_f32 ? launch_p8<T, STLLM_EPI_STORE, 0, true>(p, stream) : launch_p8<T, STLLM_EPI_STORE, 0, false>(p, stream);
      if (p.act == STLLM_ACT_GELU && !p.out_is_f32) return launch_p8<T, STLLM_EPI_STORE, 1, false>(p, stream);
      break;
    case STLLM_EPI_RESID: return launch_p8<T, STLLM_EPI_RESID, 0, false>(p, stream);
    case STLLM_EPI_SWIGLU: return launch_p8<T, STLLM_EPI_SWIGLU, 0, false>(p, stream);
    case STLLM_EPI_ROPE: return launch_p8<T, STLLM_EPI_ROPE, 0, false>(p, stream);
  }
  return STLLM_ERR_UNSUPPORTED;   // caller falls back to the 128x128 kernels
}

}  // namespace

int stllm_gemm_p8_launch(int dtype, int epilogue, const sg::GemmParams& p, hipStream_t stream) {
  if (dtype == STLLM_BF16) return dispatch_p8<bf16_t>(epilogue, p, stream);
#ifndef STLLM_P8_BF16_ONLY
  if (dtype == STLLM_F16) return dispatch_p8<f16_t>(epilogue, p, stream);
#endif
  return STLLM_ERR_UNSUPPORTED;
}
