// MFMA GEMM for gfx950:  C[M,N] = epilogue(A[M,K] @ W[N,K]^T)
//
// Structure (per 256-thread workgroup = 4 waves as 2(M) x 2(N)):
//   * K is walked in 128-BYTE panels (64 bf16/fp16 or 32 fp32 per row) — one LDS row = 128 B.
//   * both operand tiles are copied HBM->LDS with `global_load_lds_dwordx4` (no VGPR round trip);
//     the LDS image is lane-linear, so the bank-conflict swizzle is applied to the per-lane
//     SOURCE address and undone on the ds_read_b128 side (cdna_hip_programming.md §5.4 rule 21):
//         physical 16-B chunk = logical chunk ^ ((row >> 1) & 7)
//     which makes every 16-lane ds_read_b128 group hit 16 distinct 16-B slots of the 256-B bank row.
//   * 2-stage LDS ring: panel t+1 streams in while panel t feeds the matrix cores.
//   * MFMA 32x32x16 (bf16/fp16, fp32 accumulate) or 4x exact-fp32 32x32x2 per fragment pair.
//   * epilogues fused on the accumulator registers (bias / GELU / ReLU / fp32 residual add /
//     SwiGLU / rotate-half RoPE / patch-embed scatter + pos_embed).
//   * block -> tile map is XCD-aware (block b runs on XCD b % 8): each XCD gets a contiguous run of
//     tiles that share operand panels in its private L2.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace {

struct GemmParams {
  const char* A; int64_t lda_b;    // bytes
  const char* W; int64_t ldw_b;
  const float* bias;
  void* out; int64_t ldo;          // elements
  const float* resid; int64_t ldr;
  const float* aux0; const float* aux1;
  const float* frames;
  int rope_seq, rope_cols;
  int M, N, K;                     // K in elements (padded)
  int act, out_is_f32;
  int tiles_m, tiles_n;
  int debug;                       // ablation bits (env STLLM_GEMM_DEBUG): 1 skip staging, 2 skip MFMA loop, 4 skip copy-out
  int a_rpb; int64_t a_bs_b;       // A 2-level rows: rows per batch, batch stride (bytes)
  int o_rpb; int64_t o_bs;         // out 2-level rows (elements)
};

constexpr int kRowBytes = 128;  // one K panel row
constexpr bool kUsePrefetchWave = false;  // experimental 5th wave that pulls future K panels into L2
constexpr int kThreads = kUsePrefetchWave ? 320 : 256;   // 4 MFMA waves (+ 1 L2-prefetch wave)
constexpr int kPrefetchDist = 6; // panels the prefetch wave runs ahead of the MFMA waves

template <int BM, int BN> struct Tile {
  static constexpr int WM = BM / 2, WN = BN / 2, MI = WM / 32, NI = WN / 32;
  static constexpr int kStageBytes = (BM + BN) * kRowBytes;
  static constexpr int kPfScratch = 256;                    // landing pad of the L2-prefetch wave's LDS-DMA
  static constexpr int kLdsBytes = 2 * kStageBytes + kPfScratch;
  // resident workgroups per CU: LDS-limited (160 KiB), at most 6 (5 waves each, 32 waves per CU)
  static constexpr int kWavesPerWG = kThreads / 64;
  static constexpr int kPerCU = (160 * 1024 / kLdsBytes) < (32 / kWavesPerWG) ? (160 * 1024 / kLdsBytes) : (32 / kWavesPerWG);
  static constexpr int kMaxPersistent = kPerCU * 256;
};

// XCD-aware bijective remap of the linear block id (guide §5: "XCD swizzle must be bijective")
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + k;
}

// ---- patch-embed implicit-GEMM A loader (eva_vit.py:196-204) ---------------------------------
// logical element k of patch-row m:  frames[n][c][py*14+dy][px*14+dx],
//   n = m/256, py = (m%256)/16, px = m%16, c = k/196, dy = (k%196)/14, dx = k%14; k >= 588 -> 0
template <typename T>
__device__ __forceinline__ i32x4 patch_chunk(const float* __restrict__ frames, int m, int M, int k0) {
  // returns 16 bytes = 8 (16-bit) or 4 (fp32) consecutive k starting at k0
  constexpr int NE = 16 / Elem<T>::kBytes;
  float v[NE];
  const int mm = m < M ? m : M - 1;
  const int n = mm >> 8, p = mm & 255, py = p >> 4, px = p & 15;
  const float* base = frames + (int64_t)n * (3 * 224 * 224) + (py * 14) * 224 + px * 14;
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    const int k = k0 + e;
    const int c = k / 196, rem = k - c * 196, dy = rem / 14, dx = rem - dy * 14;
    v[e] = (k < 588) ? base[c * (224 * 224) + dy * 224 + dx] : 0.0f;
  }
  i32x4 r;
  if constexpr (Elem<T>::kIsF32) {
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = __builtin_bit_cast(int, v[e]);
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      r[e] = (int)((uint32_t)Elem<T>::pack(v[2 * e]) | ((uint32_t)Elem<T>::pack(v[2 * e + 1]) << 16));
  }
  return r;
}

constexpr int kGroupM = 8;  // tile rows per L2 locality group

// work id -> (tm, tn): groups of kGroupM tile rows, tm fastest inside a group.  With the XCD remap
// applied to the PERSISTENT block id, the 64 tiles an XCD runs concurrently form an ~8x8 patch that
// shares 8 A panels and 8 W panels in that XCD's private L2.
__device__ __forceinline__ void tile_coords(int w, int tiles_m, int tiles_n, int& tm, int& tn) {
  const int gsz = kGroupM * tiles_n;
  const int g = w / gsz, rem = w - g * gsz;
  const int first = g * kGroupM;
  const int gm = min(kGroupM, tiles_m - first);
  tn = rem / gm;
  tm = first + (rem - tn * gm);
}

template <typename T, int BM, int BN, int EPI, int ACT, bool OF32>
__global__ __launch_bounds__(kThreads, kUsePrefetchWave ? 3 : 2) void gemm_kernel(const GemmParams p) {
  using TL = Tile<BM, BN>;
  constexpr int MI = TL::MI, NI = TL::NI, WM = TL::WM, WN = TL::WN;
  constexpr int EB = Elem<T>::kBytes;
  constexpr int kElemsPerPanel = kRowBytes / EB;
  constexpr bool kPatch = (EPI == STLLM_EPI_PATCH);
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is_pf = kUsePrefetchWave && (wave == 4);  // wave 4 never touches the matrix pipe: it pulls future panels into L2
  const int wm = (wave >> 1) & 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  // workgroup barrier that does NOT drain outstanding LDS-DMA (a __syncthreads() would emit vmcnt(0))
#define STLLM_BAR() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

  // ---- persistent work list: this block runs tiles pb, pb + G, pb + 2G, ... ----------------------
  const int ntiles = p.tiles_m * p.tiles_n;
  const int G = gridDim.x;
  const int pb = xcd_remap(blockIdx.x, G);
  const int nk = p.K / kElemsPerPanel;

  // ---- staging plan: piece = 8 rows x 128 B = one wave-wide global_load_lds --------------------
  // combined tile rows [0,BM) = A, [BM,BM+BN) = W; wave w owns pieces w, w+4, ...
  constexpr int kPieces = (BM + BN) / 8;
  constexpr int kPiecesPerWave = kPieces / 4;
  const char* gsrc[kPiecesPerWave];
  auto a_row = [&](int gr) -> const char* {
    gr = gr < p.M ? gr : p.M - 1;
    int64_t off = (int64_t)gr * p.lda_b;
    if (p.a_rpb > 0) { const int bb = gr / p.a_rpb; off = (int64_t)bb * p.a_bs_b + (int64_t)(gr - bb * p.a_rpb) * p.lda_b; }
    return p.A + off;
  };
  auto plan = [&](int m0, int n0) {
#pragma unroll
    for (int j = 0; j < kPiecesPerWave; ++j) {
      const int piece = wave + 4 * j;
      const int r = piece * 8 + (lane >> 3);          // row in combined tile
      const int c = lane & 7;                          // physical 16-B chunk in the LDS row
      const int lc = c ^ ((r >> 1) & 7);               // logical chunk to fetch
      if (r < BM) {
        gsrc[j] = a_row(m0 + r) + lc * 16;
      } else {
        int gr = n0 + (r - BM); gr = gr < p.N ? gr : p.N - 1;
        gsrc[j] = p.W + (int64_t)gr * p.ldw_b + lc * 16;
      }
    }
  };
  auto stage = [&](int m0, int t, int buf) {
    if (p.debug & 1) return;
    char* dst = smem + buf * TL::kStageBytes;
#pragma unroll
    for (int j = 0; j < kPiecesPerWave; ++j) {
      const int piece = wave + 4 * j;
      if constexpr (kPatch) {
        if (piece * 8 < BM) {  // A rows: gather + convert through registers
          const int r = piece * 8 + (lane >> 3), c = lane & 7, lc = c ^ ((r >> 1) & 7);
          const i32x4 v = patch_chunk<T>(p.frames, m0 + r, p.M, t * kElemsPerPanel + lc * (16 / EB));
          *reinterpret_cast<i32x4*>(dst + r * kRowBytes + c * 16) = v;
          continue;
        }
      }
      glds16(gsrc[j] + (int64_t)t * kRowBytes, dst + piece * 1024);
    }
  };

  // ---- fragment read offsets (bytes) ---------------------------------------------------------
  const int sw = (li >> 1) & 7;
  int koff[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) koff[kk] = ((kk * 2 + lh) ^ sw) << 4;
  const int a_row_off = (wm * WM + li) * kRowBytes;
  const int b_row_off = (BM + wn * WN + li) * kRowBytes;

  auto out_off = [&](int row) -> int64_t {
    if (p.o_rpb > 0) { const int bb = row / p.o_rpb; return (int64_t)bb * p.o_bs + (int64_t)(row - bb * p.o_rpb) * p.ldo; }
    return (int64_t)row * p.ldo;
  };

  int w = pb;
  if (w >= ntiles) return;
  int tm, tn;
  tile_coords(w, p.tiles_m, p.tiles_n, tm, tn);
  if (!is_pf) {
    plan(tm * BM, tn * BN);
    stage(tm * BM, 0, 0);
  }
  int it = 0;  // running panel counter: panel `it` lives in LDS stage it & 1
  char* pf_pad = smem + 2 * TL::kStageBytes;
  const int pos = blockIdx.x >> 3;  // position of this workgroup inside its XCD's contiguous run of tiles

  // L2 prefetch of one K panel of a tile: one 4-byte LDS-DMA per 128-byte line (the data is discarded; the
  // line stays in this XCD's L2 so that the MFMA waves' global_load_lds hit).  Only the first tile of a row /
  // column inside the XCD's patch prefetches the A / W panel the whole patch shares.
  auto prefetch = [&](int ww, int ptm, int ptn, int t) {
    const int gsz = kGroupM * p.tiles_n;
    const int rem = ww % gsz;
    const int gm = min(kGroupM, p.tiles_m - (ww / gsz) * kGroupM);
    const bool lead_a = (pos < gm) || (rem < gm);
    const bool lead_w = (rem % gm == 0) || (pos == 0);
    if (lead_a && !kPatch) {
#pragma unroll
      for (int j = 0; j < (BM + 63) / 64; ++j) {
        const int r = lane + 64 * j;
        if (r < BM) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_row(ptm * BM + r) + (int64_t)t * kRowBytes),
                                                     (__attribute__((address_space(3))) void*)pf_pad, 4, 0, 0);
      }
    }
    if (lead_w) {
#pragma unroll
      for (int j = 0; j < (BN + 63) / 64; ++j) {
        const int r = lane + 64 * j;
        int gr = ptn * BN + r; gr = gr < p.N ? gr : p.N - 1;
        if (r < BN) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.W + (int64_t)gr * p.ldw_b + (int64_t)t * kRowBytes),
                                                     (__attribute__((address_space(3))) void*)pf_pad, 4, 0, 0);
      }
    }
  };

  if (is_pf) {
    // ---- prefetch wave: mirrors the MFMA waves' barrier sequence exactly, kPrefetchDist panels ahead -------
    constexpr int kPasses = (BM * BN * 4 > TL::kStageBytes) ? 2 : 1;
    while (w < ntiles) {
      const int w_next = w + G;
      int tm_n = 0, tn_n = 0;
      if (w_next < ntiles) tile_coords(w_next, p.tiles_m, p.tiles_n, tm_n, tn_n);
      for (int t = 0; t < nk; ++t) {
        const int tp = t + kPrefetchDist;
        if (tp < nk) prefetch(w, tm, tn, tp);
        else if (w_next < ntiles && tp - nk < nk) prefetch(w_next, tm_n, tn_n, tp - nk);
        STLLM_BAR();
      }
#pragma unroll
      for (int q = 0; q < 2 * kPasses; ++q) STLLM_BAR();
      w = w_next; tm = tm_n; tn = tn_n;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }

  while (w < ntiles) {
    const int m0 = tm * BM, n0 = tn * BN;
    const int w_next = w + G;
    int tm_n = 0, tn_n = 0;
    if (w_next < ntiles) tile_coords(w_next, p.tiles_m, p.tiles_n, tm_n, tn_n);

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    for (int t = 0; t < nk; ++t, ++it) {
      const int cur = it & 1;
      // panel `it` has landed (each wave waits for its own pieces, the barrier publishes all of them)
      // and every wave is done reading stage cur^1 (panel it-1 / the previous tile's epilogue)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      STLLM_BAR();
      if (t + 1 < nk) {
        stage(m0, t + 1, cur ^ 1);
      } else if (w_next < ntiles) {  // flattened stream: next tile's first panel flies under this tile's epilogue
        plan(tm_n * BM, tn_n * BN);
        stage(tm_n * BM, 0, cur ^ 1);
      }
      const char* base = smem + cur * TL::kStageBytes;
      if (p.debug & 2) continue;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        i32x4 af[MI], bf[NI];
#pragma unroll
        for (int i = 0; i < MI; ++i)
          af[i] = *reinterpret_cast<const i32x4*>(base + a_row_off + i * 32 * kRowBytes + koff[kk]);
#pragma unroll
        for (int j = 0; j < NI; ++j)
          bf[j] = *reinterpret_cast<const i32x4*>(base + b_row_off + j * 32 * kRowBytes + koff[kk]);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j) acc[i][j] = Elem<T>::mfma(af[i], bf[j], acc[i][j]);
      }
    }

    // ---- epilogue: raw fp32 accumulators -> LDS stage just consumed -> (bias / act / RoPE / SwiGLU / residual)
    //      applied by the copy-out threads on whole rows -> coalesced 16-byte global stores.
    // C layout of 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    char* ep = smem + ((it - 1) & 1) * TL::kStageBytes;  // the other stage is receiving the next tile's panel
    constexpr bool kOutT = (EPI == STLLM_EPI_SWIGLU || EPI == STLLM_EPI_ROPE);   // always compute-dtype out
    constexpr bool kOutF = (EPI == STLLM_EPI_RESID || EPI == STLLM_EPI_PATCH);   // always fp32 out
    constexpr bool f32out = kOutF || (!kOutT && (OF32 || Elem<T>::kIsF32));
    constexpr int oes = f32out ? 4 : EB;                                          // output element bytes
    constexpr int passes = (BM * BN * 4 > TL::kStageBytes) ? 2 : 1;               // fp32 tile vs one LDS stage
    constexpr int rows_pp = BM / passes;                                          // == WM when 2 passes
    constexpr int pitch = BN * 4;
    for (int pass = 0; pass < passes; ++pass) {
      STLLM_BAR();  // stage `ep` free: all waves finished their last MFMA reads / previous pass copy-out
      if (passes == 1 || wm == pass) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int lrow = wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh - pass * rows_pp;
#pragma unroll
            for (int j = 0; j < NI; ++j)
              *reinterpret_cast<float*>(ep + lrow * pitch + (wn * WN + j * 32 + li) * 4) = acc[i][j][r];
          }
      }
      STLLM_BAR();
      if (p.debug & 4) continue;
      // ---- copy-out --------------------------------------------------------------------------------
      auto ldf4 = [&](int lrow, int col) { return *reinterpret_cast<const f32x4*>(ep + lrow * pitch + col * 4); };
      auto gf4 = [&](const float* ptr) { return *reinterpret_cast<const f32x4*>(ptr); };
      auto pack4 = [&](f32x4 a, f32x4 b) {  // 8 values -> 16 bytes of the compute dtype
        i32x4 o;
        o[0] = (int)((uint32_t)Elem<T>::pack(a[0]) | ((uint32_t)Elem<T>::pack(a[1]) << 16));
        o[1] = (int)((uint32_t)Elem<T>::pack(a[2]) | ((uint32_t)Elem<T>::pack(a[3]) << 16));
        o[2] = (int)((uint32_t)Elem<T>::pack(b[0]) | ((uint32_t)Elem<T>::pack(b[1]) << 16));
        o[3] = (int)((uint32_t)Elem<T>::pack(b[2]) | ((uint32_t)Elem<T>::pack(b[3]) << 16));
        return o;
      };
      f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
      if constexpr (EPI == STLLM_EPI_SWIGLU) {
        // staged row = [32 gate | 32 up] per 64-column group; 8 outputs per work item
        constexpr int IPR = BN / 16;  // items per row: BN/2 outputs / 8
        for (int c = tid; c < rows_pp * IPR; c += 256) {
          const int lrow = c / IPR, q = c - lrow * IPR;
          const int row = m0 + pass * rows_pp + lrow;
          if (row >= p.M) continue;
          const int g = q >> 2, within = (q & 3) * 8;      // 4 items per 32-output group
          const int gc = g * 64 + within, uc = gc + 32;
          f32x4 ga = ldf4(lrow, gc), gb = ldf4(lrow, gc + 4), ua = ldf4(lrow, uc), ub = ldf4(lrow, uc + 4);
          if (p.bias) { ga += gf4(p.bias + n0 + gc); gb += gf4(p.bias + n0 + gc + 4); ua += gf4(p.bias + n0 + uc); ub += gf4(p.bias + n0 + uc + 4); }
          f32x4 oa, ob;
#pragma unroll
          for (int e = 0; e < 4; ++e) { oa[e] = silu_f(ga[e]) * ua[e]; ob[e] = silu_f(gb[e]) * ub[e]; }
          char* dst = reinterpret_cast<char*>(p.out) + (out_off(row) + (n0 >> 1) + g * 32 + within) * EB;
          if constexpr (Elem<T>::kIsF32) { *reinterpret_cast<f32x4*>(dst) = oa; *reinterpret_cast<f32x4*>(dst + 16) = ob; }
          else *reinterpret_cast<i32x4*>(dst) = pack4(oa, ob);
        }
      } else if constexpr (EPI == STLLM_EPI_ROPE) {
        // staged row = [x_lo(32) | x_hi(32)] per 64-column group (rotate-half partners 32 columns apart)
        constexpr int IPR = BN / 16;  // BN/64 groups x 4 items of 8 partner pairs
        for (int c = tid; c < rows_pp * IPR; c += 256) {
          const int lrow = c / IPR, q = c - lrow * IPR;
          const int row = m0 + pass * rows_pp + lrow;
          if (row >= p.M) continue;
          const int g = q >> 2, within = (q & 3) * 8;
          const int c1 = g * 64 + within, c2 = c1 + 32;
          f32x4 xa = ldf4(lrow, c1), xb = ldf4(lrow, c1 + 4), ya = ldf4(lrow, c2), yb = ldf4(lrow, c2 + 4);
          if (p.bias) { xa += gf4(p.bias + n0 + c1); xb += gf4(p.bias + n0 + c1 + 4); ya += gf4(p.bias + n0 + c2); yb += gf4(p.bias + n0 + c2 + 4); }
          if (n0 + c1 < p.rope_cols) {
            const int fi = (((n0 + c1) >> 6) & 1) * 32 + within;   // frequency index of the first pair
            const int pos = row % p.rope_seq;
            const f32x4 ca = gf4(p.aux0 + pos * 64 + fi), cb = gf4(p.aux0 + pos * 64 + fi + 4);
            const f32x4 sa = gf4(p.aux1 + pos * 64 + fi), sb = gf4(p.aux1 + pos * 64 + fi + 4);
            const f32x4 ra = xa * ca - ya * sa, rb = xb * cb - yb * sb;
            ya = ya * ca + xa * sa; yb = yb * cb + xb * sb;
            xa = ra; xb = rb;
          }
          char* dst = reinterpret_cast<char*>(p.out) + (out_off(row) + n0 + c1) * EB;
          if constexpr (Elem<T>::kIsF32) {
            *reinterpret_cast<f32x4*>(dst) = xa; *reinterpret_cast<f32x4*>(dst + 16) = xb;
            *reinterpret_cast<f32x4*>(dst + 32 * 4) = ya; *reinterpret_cast<f32x4*>(dst + 32 * 4 + 16) = yb;
          } else {
            *reinterpret_cast<i32x4*>(dst) = pack4(xa, xb);
            *reinterpret_cast<i32x4*>(dst + 32 * EB) = pack4(ya, yb);
          }
        }
      } else if constexpr (f32out) {
        constexpr int IPR = BN / 4;
        for (int c = tid; c < rows_pp * IPR; c += 256) {
          const int lrow = c / IPR, cc = (c - lrow * IPR) * 4;
          const int row = m0 + pass * rows_pp + lrow;
          if (row >= p.M) continue;
          const int col = n0 + cc;
          f32x4 v = ldf4(lrow, cc) + (p.bias ? gf4(p.bias + col) : zero4);
          if constexpr (EPI == STLLM_EPI_RESID) {
            v += gf4(p.resid + (int64_t)row * p.ldr + col);
            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + out_off(row) + col) = v;
          } else if constexpr (EPI == STLLM_EPI_PATCH) {
            const int n = row >> 8, pp = row & 255;
            v += gf4(p.aux0 + (int64_t)(1 + pp) * p.N + col);
            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + ((int64_t)n * 257 + 1 + pp) * p.ldo + col) = v;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              if constexpr (ACT == STLLM_ACT_GELU) v[e] = gelu_erf(v[e]);
              if constexpr (ACT == STLLM_ACT_RELU) v[e] = fmaxf(v[e], 0.0f);
            }
            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + out_off(row) + col) = v;
          }
        }
      } else {  // STORE, compute-dtype out (16-bit): 8 columns per work item
        constexpr int IPR = BN / 8;
        for (int c = tid; c < rows_pp * IPR; c += 256) {
          const int lrow = c / IPR, cc = (c - lrow * IPR) * 8;
          const int row = m0 + pass * rows_pp + lrow;
          if (row >= p.M) continue;
          const int col = n0 + cc;
          f32x4 a = ldf4(lrow, cc), b = ldf4(lrow, cc + 4);
          if (p.bias) { a += gf4(p.bias + col); b += gf4(p.bias + col + 4); }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if constexpr (ACT == STLLM_ACT_GELU) { a[e] = gelu_erf(a[e]); b[e] = gelu_erf(b[e]); }
            if constexpr (ACT == STLLM_ACT_RELU) { a[e] = fmaxf(a[e], 0.0f); b[e] = fmaxf(b[e], 0.0f); }
          }
          *reinterpret_cast<i32x4*>(reinterpret_cast<char*>(p.out) + (out_off(row) + col) * EB) = pack4(a, b);
        }
      }
    }
    w = w_next; tm = tm_n; tn = tn_n;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the prefetch wave's LDS-DMA must land before the LDS is released
#undef STLLM_BAR
}

template <typename T, int BM, int BN, int EPI, int ACT = 0, bool OF32 = false>
int launch(const GemmParams& p0, hipStream_t stream) {
  GemmParams p = p0;
  p.tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = p.N / BN;
  const int lds = Tile<BM, BN>::kLdsBytes;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_kernel<T, BM, BN, EPI, ACT, OF32>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  // persistent grid: every workgroup is resident (LDS-limited workgroups per CU x 256 CUs)
  const int ntiles = p.tiles_m * p.tiles_n;
  const int grid = ntiles < Tile<BM, BN>::kMaxPersistent ? ntiles : Tile<BM, BN>::kMaxPersistent;
  hipLaunchKernelGGL((gemm_kernel<T, BM, BN, EPI, ACT, OF32>), dim3(grid), dim3(kThreads), lds, stream, p);
  STLLM_CHECK_LAUNCH("stllm_gemm");
  {
    static const char* kEpi[] = {"STORE", "RESID", "SWIGLU", "ROPE", "PATCH"};
    static char name[96];
    static bool named = false;
    if (!named) {
      snprintf(name, sizeof(name), "gemm_kernel<%s,%d,%d,%s,%d,%d>",
               Elem<T>::kIsF32 ? "float" : (std::is_same<T, bf16_t>::value ? "bf16_t" : "f16_t"), BM, BN, kEpi[EPI], ACT, (int)OF32);
      named = true;
    }
    stllm_set_last_kernel(name);
  }
  return STLLM_OK;
}

template <typename T, int EPI, int ACT = 0, bool OF32 = false>
int dispatch_tile(const GemmParams& p, hipStream_t stream) {
  if constexpr (EPI == STLLM_EPI_SWIGLU || EPI == STLLM_EPI_ROPE) {
    // these epilogues pair columns inside a 64-column wave tile
    if (p.M <= 64) return launch<T, 64, 128, EPI>(p, stream);
    return launch<T, 128, 128, EPI>(p, stream);
  } else {
    // small problems: smaller tiles so that more CUs get work
    const int64_t t128 = (int64_t)((p.M + 127) / 128) * (p.N / 128);
    if (t128 < 192) return launch<T, 64, 64, EPI, ACT, OF32>(p, stream);
    return launch<T, 128, 128, EPI, ACT, OF32>(p, stream);
  }
}

template <typename T>
int dispatch_store(const GemmParams& p, hipStream_t stream) {
  const int key = p.act * 2 + (p.out_is_f32 ? 1 : 0);
  switch (key) {
    case 0: return dispatch_tile<T, STLLM_EPI_STORE, 0, false>(p, stream);
    case 1: return dispatch_tile<T, STLLM_EPI_STORE, 0, true>(p, stream);
    case 2: return dispatch_tile<T, STLLM_EPI_STORE, 1, false>(p, stream);
    case 3: return dispatch_tile<T, STLLM_EPI_STORE, 1, true>(p, stream);
    case 4: return dispatch_tile<T, STLLM_EPI_STORE, 2, false>(p, stream);
    case 5: return dispatch_tile<T, STLLM_EPI_STORE, 2, true>(p, stream);
  }
  stllm_set_error("stllm_gemm: bad act %d", p.act);
  return STLLM_ERR_UNSUPPORTED;
}

template <typename T>
int dispatch_epi(const stllm_gemm_args* a, const GemmParams& p, hipStream_t stream) {
  switch (a->epilogue) {
    case STLLM_EPI_STORE: return dispatch_store<T>(p, stream);
    case STLLM_EPI_RESID: return dispatch_tile<T, STLLM_EPI_RESID>(p, stream);
    case STLLM_EPI_SWIGLU: return dispatch_tile<T, STLLM_EPI_SWIGLU>(p, stream);
    case STLLM_EPI_ROPE: return dispatch_tile<T, STLLM_EPI_ROPE>(p, stream);
    case STLLM_EPI_PATCH: return dispatch_tile<T, STLLM_EPI_PATCH>(p, stream);
  }
  stllm_set_error("stllm_gemm: unknown epilogue %d", a->epilogue);
  return STLLM_ERR_UNSUPPORTED;
}

}  // namespace

extern "C" int stllm_gemm(const stllm_gemm_args* a, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  STLLM_CHECK_ARG(a != nullptr, "stllm_gemm: null args");
  STLLM_CHECK_ARG(a->dtype >= STLLM_BF16 && a->dtype <= STLLM_F32, "stllm_gemm: bad dtype %d", a->dtype);
  const int eb = a->dtype == STLLM_F32 ? 4 : 2;
  const int panel = kRowBytes / eb;
  STLLM_CHECK_ARG(a->M > 0 && a->N > 0 && a->K > 0, "stllm_gemm: empty problem M=%d N=%d K=%d", a->M, a->N, a->K);
  STLLM_CHECK_ARG(a->N % 128 == 0, "stllm_gemm: N=%d must be a multiple of 128", a->N);
  const bool patch = a->epilogue == STLLM_EPI_PATCH;
  int K = a->K;
  if (patch) {
    STLLM_CHECK_ARG(a->K == 588 && a->M % 256 == 0 && a->frames && a->aux0, "stllm_gemm(PATCH): need K=588, M=n_frames*256, frames, pos_embed");
    K = ((588 + panel - 1) / panel) * panel;
    STLLM_CHECK_ARG(a->ldw >= K, "stllm_gemm(PATCH): W must be zero-padded to ldw >= %d", K);
  } else {
    STLLM_CHECK_ARG(a->A && aligned16(a->A), "stllm_gemm: A null or not 16-byte aligned");
    STLLM_CHECK_ARG(a->K % panel == 0, "stllm_gemm: K=%d must be a multiple of %d", a->K, panel);
    STLLM_CHECK_ARG((a->lda * eb) % 16 == 0 && a->lda >= a->K, "stllm_gemm: bad lda %lld", (long long)a->lda);
  }
  STLLM_CHECK_ARG(a->W && aligned16(a->W) && (a->ldw * eb) % 16 == 0 && a->ldw >= K, "stllm_gemm: bad W/ldw");
  STLLM_CHECK_ARG(a->out != nullptr && aligned16(a->out), "stllm_gemm: out null or not 16-byte aligned");
  {
    const bool f32o = a->epilogue == STLLM_EPI_RESID || a->epilogue == STLLM_EPI_PATCH ||
                      (a->epilogue == STLLM_EPI_STORE && a->out_is_f32);
    const int oes = f32o ? 4 : eb;
    STLLM_CHECK_ARG((a->ldo * oes) % 16 == 0 && (a->o_batch_stride * oes) % 16 == 0,
                    "stllm_gemm: output row/batch stride must be a multiple of 16 bytes (ldo=%lld)", (long long)a->ldo);
  }
  if (a->epilogue == STLLM_EPI_RESID)
    STLLM_CHECK_ARG(a->resid != nullptr && aligned16(a->resid) && a->ldr % 4 == 0, "stllm_gemm(RESID): resid null / misaligned");
  if (a->epilogue == STLLM_EPI_PATCH) STLLM_CHECK_ARG(aligned16(a->aux0), "stllm_gemm(PATCH): pos_embed misaligned");
  if (a->epilogue == STLLM_EPI_ROPE)
    STLLM_CHECK_ARG(a->aux0 && a->aux1 && a->rope_seq > 0 && a->rope_cols % 128 == 0, "stllm_gemm(ROPE): need cos/sin tables, rope_seq, rope_cols%%128==0");

  GemmParams p{};
  p.A = reinterpret_cast<const char*>(a->A); p.lda_b = a->lda * eb;
  p.W = reinterpret_cast<const char*>(a->W); p.ldw_b = a->ldw * eb;
  p.bias = a->bias; p.out = a->out; p.ldo = a->ldo; p.resid = a->resid; p.ldr = a->ldr;
  p.aux0 = a->aux0; p.aux1 = a->aux1; p.frames = a->frames;
  p.rope_seq = a->rope_seq; p.rope_cols = a->rope_cols;
  p.M = a->M; p.N = a->N; p.K = K; p.act = a->act; p.out_is_f32 = a->out_is_f32;
  { static int dbg = -1; if (dbg < 0) { const char* e = getenv("STLLM_GEMM_DEBUG"); dbg = e ? atoi(e) : 0; } p.debug = dbg; }
  p.a_rpb = a->a_rows_per_batch; p.a_bs_b = a->a_batch_stride * eb;
  p.o_rpb = a->o_rows_per_batch; p.o_bs = a->o_batch_stride;
  switch (a->dtype) {
    case STLLM_BF16: return dispatch_epi<bf16_t>(a, p, stream);
    case STLLM_F16: return dispatch_epi<f16_t>(a, p, stream);
    default: return dispatch_epi<float>(a, p, stream);
  }
}
