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
#include <string.h>

#include <atomic>
#include <type_traits>

#include "gemm_common.h"

namespace {
using namespace sg;

constexpr bool kUsePrefetchWave = false;  // experimental 5th wave that pulls future K panels into L2
constexpr int kThreads = kUsePrefetchWave ? 320 : 256;   // 4 MFMA waves (+ 1 L2-prefetch wave)
constexpr int kPrefetchDist = 6; // panels the prefetch wave runs ahead of the MFMA waves

#ifndef STLLM_GEMM_RING64
#define STLLM_GEMM_RING64 4   // LDS ring depth of the 64 x 64 tile (2 = rounds 1-3; A/B builds: -DSTLLM_GEMM_RING64=2)
#endif
template <int BM, int BN> struct Tile {
  static constexpr int WM = BM / 2, WN = BN / 2, MI = WM / 32, NI = WN / 32;
  static constexpr int kStageBytes = (BM + BN) * kRowBytes;
  // Ring depth.  The 64 x 64 tile serves the small problems (Q-Former: M = 512 rows, K = 768 / 3072): a K panel of it is ~200 cycles of
  // matrix work per wave against a ~1000-cycle L2 round trip, so with two stages ("wait for everything, barrier, request the next panel")
  // the loop ran at one L2 latency per panel: 0.55 us x 12 .. 48 panels.  Four stages (16 KiB each) keep three panels in flight behind a
  // counted wait (round 4; 4 x 16 KiB + pad = 65 KiB: two workgroups per CU instead of four).  The 128-row tiles stay at two stages (32-48 KiB per stage, two workgroups per CU).
  static constexpr int kStages = (BM == 64 && BN == 64) ? STLLM_GEMM_RING64 : 2;
  static constexpr int kPfScratch = 1024;                   // landing pad: the L2-prefetch wave's LDS-DMA / the pieces issued past the end of the panel stream
  static constexpr int kLdsBytes = kStages * kStageBytes + kPfScratch;
  // resident workgroups per CU: LDS-limited (160 KiB), at most 6 (5 waves each, 32 waves per CU)
  static constexpr int kWavesPerWG = kThreads / 64;
  static constexpr int kPerCU = (160 * 1024 / kLdsBytes) < (32 / kWavesPerWG) ? (160 * 1024 / kLdsBytes) : (32 / kWavesPerWG);
  static constexpr int kMaxPersistent = kPerCU * 256;
};

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
      r[e] = (int)(Elem<T>::pack2(v[2 * e], v[2 * e + 1]));
  }
  return r;
}

static int g_debug_early() { return stllm_options().gemm_debug; }

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
  constexpr int NS = TL::kStages;
  int it = 0;  // running panel counter: panel `it` lives in LDS stage it % NS
  char* pf_pad = smem + NS * TL::kStageBytes;
  // ---- the panel stream of this workgroup, flattened over its tiles: an ISSUE cursor (tile iw, panel it_iss) runs NS - 1 panels ahead of
  // the compute loop.  Every call requests exactly kPiecesPerWave pieces per wave — past the end of the stream into the landing pad — so
  // that the loop's counted wait (vmcnt retires in order) always means "panel `it` has landed".
  int iw = w, it_iss = 0, im = tm, in_ = tn;
  auto issue_next = [&](int buf) {
    if (iw < ntiles) {
      stage(im * BM, it_iss, buf);
      if (++it_iss == nk) {
        iw += G;
        it_iss = 0;
        if (iw < ntiles) {
          tile_coords(iw, p.tiles_m, p.tiles_n, im, in_);
          plan(im * BM, in_ * BN);
        }
      }
    } else if (!(p.debug & 1)) {
#pragma unroll
      for (int j = 0; j < kPiecesPerWave; ++j) glds16(p.W + (lane & 7) * 16, pf_pad);
    }
  };
  if (!is_pf) {
    plan(tm * BM, tn * BN);
#pragma unroll
    for (int s_ = 0; s_ < NS - 1; ++s_) issue_next(s_);
  }
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
      const int cur = it % NS;
      // panel `it` has landed (each wave waits for its own pieces — the NS - 2 younger panels may stay in flight —, the barrier publishes
      // all of them) and every wave is done reading the stage of panel it - 1 (or the previous tile's epilogue scratch), which the
      // request below overwrites.  The stream is flattened over the tiles: the next tile's first panels fly under this tile's epilogue.
      wait_vmcnt<(NS - 2) * kPiecesPerWave>();
      STLLM_BAR();
      issue_next((it + NS - 1) % NS);
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
    char* ep = smem + ((it - 1) % NS) * TL::kStageBytes;  // the stage consumed last; the others are receiving the next panels
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
        o[0] = (int)(Elem<T>::pack2(a[0], a[1]));
        o[1] = (int)(Elem<T>::pack2(a[2], a[3]));
        o[2] = (int)(Elem<T>::pack2(b[0], b[1]));
        o[3] = (int)(Elem<T>::pack2(b[2], b[3]));
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
  static StllmPerDevice attr_dev;   // the dynamic-LDS opt-in is a per-device attribute
  bool attr_first;
  const int attr_d = attr_dev.enter(&attr_first);
  if (attr_first) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_kernel<T, BM, BN, EPI, ACT, OF32>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_dev.done(attr_d);
  }
  // persistent grid: every workgroup is resident (LDS-limited workgroups per CU x 256 CUs)
  const int ntiles = p.tiles_m * p.tiles_n;
  static StllmPerDevice occ_dev;   // resident workgroups per device ordinal
  bool occ_first;
  const int occ_d = occ_dev.enter(&occ_first);
  if (occ_first) {
    int max_wg;
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    (void)hipGetDevice(&dev);
    (void)hipGetDeviceProperties(&prop, dev);
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gemm_kernel<T, BM, BN, EPI, ACT, OF32>, kThreads, lds);
    if (per_cu < 1) per_cu = 1;
    max_wg = per_cu * prop.multiProcessorCount;
    if (g_debug_early() & 8) fprintf(stderr, "[stllm] gemm<%d,%d> occupancy %d WG/CU x %d CUs\n", BM, BN, per_cu, prop.multiProcessorCount);
    occ_dev.value[occ_d] = max_wg;
    occ_dev.done(occ_d);
  }
  const int max_wg = occ_dev.value[occ_d];
  const int grid = ntiles < max_wg ? ntiles : max_wg;
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


// =====================================================================================================
// Stream-K variant: 512 threads = 8 waves as 2(M) x 4(N), tile BM x 256, one workgroup per CU.
//
// Why (profiles/r01_gemm_ablation.md): with a 128x128 tile the HBM/L2 -> LDS panel stream takes as long as the
// MFMA loop (64 FLOP per LDS-filled byte).  A BM x 256 tile raises that to 85 (BM=128) / 128 (BM=256) FLOP/B and
// the 3-stage (BM=128) ring keeps two panels in flight.  Tiles this large quantise badly on 256 CUs for this
// workload (M = 4112 or 576, N = 1408 ... 32000), so the flattened (tile, K-panel) space is cut into G equal
// ranges (stream-K): a workgroup may start and end in the middle of a tile.
//   * a range that starts inside a tile produces a PARTIAL: raw fp32 accumulators -> its slab in the workspace,
//     then an agent-scope release + flag (the workgroup does this first, so partials are ready early);
//   * the workgroup that owns panel 0 of a tile FINALISES it: after its own panels it polls the flags of the
//     contributors (relaxed, one lane, bounded spin), one agent-scope acquire, adds their slabs in workgroup
//     order (deterministic) and runs the fused epilogue.
// Finalisers only ever wait for higher-numbered workgroups doing their FIRST segment; all G <= 256 workgroups
// are resident (one per CU: 128-144 KiB LDS each), so the protocol is placement- and order-independent.
// Flags carry the launch epoch (host counter), so no memset is needed between launches on one stream.
// =====================================================================================================
constexpr int kSkMaxSlabWG = 512;

// tile configurations: 128x128 (4 waves, 2 stages, 2 workgroups/CU), 128x256 (8 waves, 3 stages), 256x256 (8 waves, 2 stages)
template <int BM, int BN> struct SkTile {
  static constexpr int NWN = BN / 64, NWAVES = 2 * NWN, NT = 64 * NWAVES;
  static constexpr int MI = BM / 64, NI = 2;
  static constexpr int kStageBytes = (BM + BN) * kRowBytes;
  static constexpr int NS = (BM + BN == 384) ? 3 : 2;
  static constexpr int kLdsBytes = NS * kStageBytes;
  static constexpr int kPPW = (BM + BN) / 8 / NWAVES;       // LDS-DMA pieces per wave per panel
  static constexpr int kMaxWG = (160 * 1024 / kLdsBytes) * 256;
  static constexpr int64_t kSlabBytes = (int64_t)BM * BN * 4;
};


template <typename T, int BM, int BN, int EPI, int ACT, bool OF32>
__global__ __launch_bounds__(2 * BN, 2) void gemm_sk_kernel(const GemmParams p) {
  using TL = SkTile<BM, BN>;
  constexpr int NT = TL::NT, NWAVES = TL::NWAVES, NWN = TL::NWN;
  constexpr int MI = TL::MI, NI = TL::NI, NS = TL::NS, PPW = TL::kPPW;
  constexpr int EB = Elem<T>::kBytes;
  constexpr int kElemsPerPanel = kRowBytes / EB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
#define STLLM_BAR() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / NWN, wn = wave % NWN;
  const int li = lane & 31, lh = lane >> 5;

  const int ntiles = p.tiles_m * p.tiles_n;
  const int nk = p.K / kElemsPerPanel;
  const int64_t total = (int64_t)ntiles * nk;
  const int G = gridDim.x;
  const int g = xcd_remap(blockIdx.x, G);
  auto range_start = [&](int gg) -> int64_t { return total * gg / G; };
  const int64_t u_begin = range_start(g), u_end = range_start(g + 1);
  const int n_units = (int)(u_end - u_begin);
  if (n_units <= 0) return;

  unsigned* flags = reinterpret_cast<unsigned*>(p.ws);
  float* slabs = reinterpret_cast<float*>(p.ws + kSkFlagBytes);

  // ---- staging plan (per tile): combined tile rows [0,BM) = A, [BM,BM+BN) = W; wave w owns pieces w, w+NWAVES, ...
  const char* gsrc[PPW];
  auto a_row = [&](int gr) -> const char* {
    gr = gr < p.M ? gr : p.M - 1;
    int64_t off = (int64_t)gr * p.lda_b;
    if (p.a_rpb > 0) { const int bb = gr / p.a_rpb; off = (int64_t)bb * p.a_bs_b + (int64_t)(gr - bb * p.a_rpb) * p.lda_b; }
    return p.A + off;
  };
  auto plan = [&](int tile) {
    int tm, tn;
    tile_coords(tile, p.tiles_m, p.tiles_n, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
      const int piece = wave + NWAVES * j;
      const int r = piece * 8 + (lane >> 3);
      const int c = lane & 7;
      const int lc = c ^ ((r >> 1) & 7);
      if (r < BM) {
        gsrc[j] = a_row(m0 + r) + lc * 16;
      } else {
        int gr = n0 + (r - BM); gr = gr < p.N ? gr : p.N - 1;
        gsrc[j] = p.W + (int64_t)gr * p.ldw_b + lc * 16;
      }
    }
  };
  auto stage = [&](int k, int buf) {
    char* dst = smem + buf * TL::kStageBytes;
#pragma unroll
    for (int j = 0; j < PPW; ++j) glds16(gsrc[j] + (int64_t)k * kRowBytes, dst + (wave + NWAVES * j) * 1024);
  };

  // ---- load cursor (runs NS-1 panels ahead of the compute cursor) ------------------------------------------
  int l_left = n_units;                 // panels not yet issued
  int l_tile = (int)(u_begin / nk), l_k = (int)(u_begin - (int64_t)l_tile * nk);
  int l_cnt = 0;
  plan(l_tile);
  auto issue = [&]() {
    if (l_left <= 0) return;
    stage(l_k, l_cnt % NS);
    ++l_cnt; --l_left; ++l_k;
    if (l_k == nk) { l_k = 0; ++l_tile; if (l_left > 0) plan(l_tile); }
  };
#pragma unroll
  for (int q = 0; q < NS - 1; ++q) issue();

  // ---- fragment read offsets ------------------------------------------------------------------------------
  const int sw = (li >> 1) & 7;
  int koff[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) koff[kk] = ((kk * 2 + lh) ^ sw) << 4;
  const int a_row_off = (wm * (BM / 2) + li) * kRowBytes;
  const int b_row_off = (BM + wn * 64 + li) * kRowBytes;

  auto out_off = [&](int row) -> int64_t {
    if (p.o_rpb > 0) { const int bb = row / p.o_rpb; return (int64_t)bb * p.o_bs + (int64_t)(row - bb * p.o_rpb) * p.ldo; }
    return (int64_t)row * p.ldo;
  };

  int c_tile = (int)(u_begin / nk), c_k = (int)(u_begin - (int64_t)c_tile * nk);
  int seg_k0 = c_k;
  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  for (int it = 0; it < n_units; ++it) {
    const int cur = it % NS;
    // panel `it` landed; up to NS-2 younger panels may stay in flight across the barrier
    if (NS == 3 && it + 1 < n_units) wait_vmcnt<PPW>(); else wait_vmcnt<0>();
    STLLM_BAR();
    issue();  // into the stage consumed in the previous iteration
    const char* base = smem + cur * TL::kStageBytes;
    if (!(p.debug & 2)) {
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
    ++c_k;
    if (c_k < nk && it + 1 < n_units) continue;

    // =================== segment [seg_k0, c_k) of tile c_tile finished ===================================
    int tm, tn;
    tile_coords(c_tile, p.tiles_m, p.tiles_n, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    const bool contrib = seg_k0 > 0;                  // tile was started by a lower-numbered workgroup
    int ncontrib = 0;                                   // partials this workgroup must add (it owns panel 0)
    if (!contrib && c_k < nk) {
      const int64_t tile_end = (int64_t)(c_tile + 1) * nk;
      while (g + 1 + ncontrib < G && range_start(g + 1 + ncontrib) < tile_end) ++ncontrib;
      if (wave == 0) {  // one wave polls, relaxed; bounded so that a broken run ends instead of hanging the GPU
        for (int q = 0; q < ncontrib; ++q) {
          unsigned spins = 0;
          while (__hip_atomic_load(&flags[g + 1 + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)p.epoch) {
            if (++spins >= (1u << 22)) {   // contributor never showed up: flag it (stllm_gemm_workspace_status), never hang
              if (lane == 0) __hip_atomic_fetch_or(&flags[kSkErrWord], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              break;
            }
            __builtin_amdgcn_s_sleep(8);
          }
        }
        if (lane == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
    }
    char* ep = smem + cur * TL::kStageBytes;           // the stage just consumed; the others hold future panels
    constexpr bool kOutT = (EPI == STLLM_EPI_SWIGLU || EPI == STLLM_EPI_ROPE);
    constexpr bool f32out = (EPI == STLLM_EPI_RESID) || (!kOutT && (OF32 || Elem<T>::kIsF32));
    constexpr int pitch = BN * 4;
    float* my_slab = slabs + (int64_t)g * BM * BN;

    for (int pass = 0; pass < 2 * MI; ++pass) {
      const int pwm = pass / MI, pi = pass - pwm * MI;
      STLLM_BAR();  // `ep` free (last MFMA reads / previous pass copy-out done); also orders the acquire above
      if (wm == pwm) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          if (i != pi) continue;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int lrow = (r & 3) + 8 * (r >> 2) + 4 * lh;
#pragma unroll
            for (int j = 0; j < NI; ++j)
              *reinterpret_cast<float*>(ep + lrow * pitch + (wn * 64 + j * 32 + li) * 4) = acc[i][j][r];
          }
        }
      }
      STLLM_BAR();
      if (p.debug & 4) continue;
      const int trow0 = pwm * (BM / 2) + pi * 32;       // first tile row of this pass
      auto ldf4 = [&](int lrow, int col) {
        f32x4 v = *reinterpret_cast<const f32x4*>(ep + lrow * pitch + col * 4);
        for (int q = 0; q < ncontrib; ++q)
          v += *reinterpret_cast<const f32x4*>(slabs + (int64_t)(g + 1 + q) * BM * BN + (int64_t)(trow0 + lrow) * BN + col);
        return v;
      };
      auto gf4 = [&](const float* ptr) { return *reinterpret_cast<const f32x4*>(ptr); };
      auto pack4 = [&](f32x4 a, f32x4 b) {
        i32x4 o;
        o[0] = (int)(Elem<T>::pack2(a[0], a[1]));
        o[1] = (int)(Elem<T>::pack2(a[2], a[3]));
        o[2] = (int)(Elem<T>::pack2(b[0], b[1]));
        o[3] = (int)(Elem<T>::pack2(b[2], b[3]));
        return o;
      };
      const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
      if (contrib) {
        // raw fp32 partial -> this workgroup's slab (whole 32 x 256 pass, unmasked: the slab is private scratch)
        for (int c = tid; c < 32 * (BN / 4); c += NT) {
          const int lrow = c / (BN / 4), cc = (c - lrow * (BN / 4)) * 4;
          *reinterpret_cast<f32x4*>(my_slab + (int64_t)(trow0 + lrow) * BN + cc) =
              *reinterpret_cast<const f32x4*>(ep + lrow * pitch + cc * 4);
        }
      } else if constexpr (EPI == STLLM_EPI_SWIGLU) {
        constexpr int IPR = BN / 16;
        for (int c = tid; c < 32 * IPR; c += NT) {
          const int lrow = c / IPR, q = c - lrow * IPR;
          const int row = m0 + trow0 + lrow;
          const int gq = q >> 2, within = (q & 3) * 8;
          const int gc = gq * 64 + within, uc = gc + 32;
          if (row >= p.M || n0 + gc >= p.N) continue;
          f32x4 ga = ldf4(lrow, gc), gb = ldf4(lrow, gc + 4), ua = ldf4(lrow, uc), ub = ldf4(lrow, uc + 4);
          if (p.bias) { ga += gf4(p.bias + n0 + gc); gb += gf4(p.bias + n0 + gc + 4); ua += gf4(p.bias + n0 + uc); ub += gf4(p.bias + n0 + uc + 4); }
          f32x4 oa, ob;
#pragma unroll
          for (int e = 0; e < 4; ++e) { oa[e] = silu_f(ga[e]) * ua[e]; ob[e] = silu_f(gb[e]) * ub[e]; }
          char* dst = reinterpret_cast<char*>(p.out) + (out_off(row) + (n0 >> 1) + gq * 32 + within) * EB;
          if constexpr (Elem<T>::kIsF32) { *reinterpret_cast<f32x4*>(dst) = oa; *reinterpret_cast<f32x4*>(dst + 16) = ob; }
          else *reinterpret_cast<i32x4*>(dst) = pack4(oa, ob);
        }
      } else if constexpr (EPI == STLLM_EPI_ROPE) {
        constexpr int IPR = BN / 16;
        for (int c = tid; c < 32 * IPR; c += NT) {
          const int lrow = c / IPR, q = c - lrow * IPR;
          const int row = m0 + trow0 + lrow;
          const int gq = q >> 2, within = (q & 3) * 8;
          const int c1 = gq * 64 + within, c2 = c1 + 32;
          if (row >= p.M || n0 + c1 >= p.N) continue;
          f32x4 xa = ldf4(lrow, c1), xb = ldf4(lrow, c1 + 4), ya = ldf4(lrow, c2), yb = ldf4(lrow, c2 + 4);
          if (p.bias) { xa += gf4(p.bias + n0 + c1); xb += gf4(p.bias + n0 + c1 + 4); ya += gf4(p.bias + n0 + c2); yb += gf4(p.bias + n0 + c2 + 4); }
          if (n0 + c1 < p.rope_cols) {
            const int fi = (((n0 + c1) >> 6) & 1) * 32 + within;
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
        for (int c = tid; c < 32 * IPR; c += NT) {
          const int lrow = c / IPR, cc = (c - lrow * IPR) * 4;
          const int row = m0 + trow0 + lrow, col = n0 + cc;
          if (row >= p.M || col >= p.N) continue;
          f32x4 v = ldf4(lrow, cc) + (p.bias ? gf4(p.bias + col) : zero4);
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
      } else {
        constexpr int IPR = BN / 8;
        for (int c = tid; c < 32 * IPR; c += NT) {
          const int lrow = c / IPR, cc = (c - lrow * IPR) * 8;
          const int row = m0 + trow0 + lrow, col = n0 + cc;
          if (row >= p.M || col >= p.N) continue;
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
    if (contrib) {
      // publish the partial: every wave drains its slab stores, then ONE lane releases at agent scope and sets the flag
      wait_vmcnt<0>();
      STLLM_BAR();
      if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(&flags[g], (unsigned)p.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    // next segment
    if (c_k == nk) { c_k = 0; ++c_tile; }
    seg_k0 = c_k;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  }
  wait_vmcnt<0>();
#undef STLLM_BAR
}

static std::atomic<int> g_sk_epoch{0};
}  // namespace
int stllm_sk_next_epoch() { return ++g_sk_epoch; }   // one epoch counter for every stream-K kernel: they share the flag array
namespace {

template <typename T, int BM, int BN, int EPI, int ACT = 0, bool OF32 = false>
int launch_sk(const GemmParams& p0, hipStream_t stream) {
  GemmParams p = p0;
  p.tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.N + BN - 1) / BN;
  const int lds = SkTile<BM, BN>::kLdsBytes;
  static StllmPerDevice attr_dev;   // the dynamic-LDS opt-in is a per-device attribute
  bool attr_first;
  const int attr_d = attr_dev.enter(&attr_first);
  if (attr_first) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_sk_kernel<T, BM, BN, EPI, ACT, OF32>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_dev.done(attr_d);
  }
  const int64_t total = (int64_t)p.tiles_m * p.tiles_n * (p.K / (kRowBytes / Elem<T>::kBytes));
  // every workgroup must be RESIDENT (finalisers wait for contributors): size the grid from the occupancy query, never
  // from an assumption (MI355X_MICROARCH "Residency and cooperative launch"); one fewer per CU keeps a margin
  static StllmPerDevice occ_dev;   // resident workgroups per device ordinal
  bool occ_first;
  const int occ_d = occ_dev.enter(&occ_first);
  if (occ_first) {
    int max_wg;
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    (void)hipGetDevice(&dev);
    (void)hipGetDeviceProperties(&prop, dev);
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gemm_sk_kernel<T, BM, BN, EPI, ACT, OF32>, SkTile<BM, BN>::NT, lds);
    const int want = SkTile<BM, BN>::kMaxWG / 256;
    if (per_cu > want) per_cu = want;
    if (per_cu < 1) per_cu = 1;
    max_wg = per_cu * prop.multiProcessorCount;
    if (g_debug_early() & 8) fprintf(stderr, "[stllm] gemm_sk<%d,%d> occupancy %d WG/CU x %d CUs\n", BM, BN, per_cu, prop.multiProcessorCount);
    occ_dev.value[occ_d] = max_wg;
    occ_dev.done(occ_d);
  }
  const int max_wg = occ_dev.value[occ_d];
  const int grid = total < max_wg ? (int)total : max_wg;
  p.epoch = stllm_sk_next_epoch();
  hipLaunchKernelGGL((gemm_sk_kernel<T, BM, BN, EPI, ACT, OF32>), dim3(grid), dim3(SkTile<BM, BN>::NT), lds, stream, p);
  STLLM_CHECK_LAUNCH("stllm_gemm(stream-K)");
  {
    static const char* kEpi[] = {"STORE", "RESID", "SWIGLU", "ROPE", "PATCH"};
    static char name[96];
    static bool named = false;
    if (!named) {
      snprintf(name, sizeof(name), "gemm_sk_kernel<%s,%d,%d,%s,%d,%d>",
               Elem<T>::kIsF32 ? "float" : (std::is_same<T, bf16_t>::value ? "bf16_t" : "f16_t"), BM, BN, kEpi[EPI], ACT, (int)OF32);
      named = true;
    }
    stllm_set_last_kernel(name);
  }
  return STLLM_OK;
}

// stream-K eligibility + tile: returns 0 (off) | 1 = 128x128 | 2 = 128x256 | 3 = 256x256
static int sk_choice(const GemmParams& p, int eb) {
  const int mode = stllm_options().gemm_sk;
  if (mode == 0 || p.ws == nullptr) return 0;
  if (p.ws_bytes < kSkFlagBytes + (int64_t)kSkMaxSlabWG * 128 * 128 * 4) return 0;
  if (mode >= 1 && mode <= 3) return mode;
  // auto (measured, profiles/r01_gemm_streamk.md): the split only pays where whole tiles cannot fill the chip AND K is
  // long enough to amortise the hand-off — the Llama down_proj shape (M=576, N=4096, K=11008: 160 tiles of 128x128,
  // 172 panels): 128x256 stream-K 114 us vs 186 us.  Everywhere else the persistent whole-tile kernel is faster.
  const int64_t t128 = (int64_t)((p.M + 127) / 128) * (p.N / 128);
  const int nk = p.K * eb / kRowBytes;
  if (p.ws_bytes < kSkFlagBytes + (int64_t)256 * 128 * 256 * 4) return 0;
  return (t128 < 192 && t128 >= 32 && nk >= 128) ? 2 : 0;
}
template <typename T, int EPI, int ACT = 0, bool OF32 = false>
int dispatch_tile(const GemmParams& p, hipStream_t stream) {
  if constexpr (EPI != STLLM_EPI_PATCH) {
    const int sk = sk_choice(p, Elem<T>::kBytes);
    if (sk == 1) return launch_sk<T, 128, 128, EPI, ACT, OF32>(p, stream);
    if (sk == 2) return launch_sk<T, 128, 256, EPI, ACT, OF32>(p, stream);
    if (sk == 3) return launch_sk<T, 256, 256, EPI, ACT, OF32>(p, stream);
  }
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

// Phased kernel (gemm_p8.inc): 16-bit dtypes, needs the workspace.  mode 1 / 3 / 4: always (cost model / 192 / 256 rows);
// auto: when its cost model beats the estimate for the 128x128 kernels by a margin (both calibrated on MI355X, see
// profiles/r01_gemm_p8.md; a wrong guess near the margin costs a few percent either way).
// estimate for the 128x128 / 64x64 persistent kernels (the model p8_wanted compares with)
static float old_kernels_estimate_us(const GemmParams& p) {
  const int nk = p.K / 64;
  const int64_t t128 = (int64_t)((p.M + 127) / 128) * (p.N / 128);
  if (t128 < 192) {   // 64x64 tiles: with the 4-stage ring (65 KiB of LDS per workgroup) TWO workgroups are resident per CU (hipOccupancy: launch<>() asks), not the
    // four of the 2-stage ring — up to 512 tiles run at once.  The per-256-tiles factor below is an EMPIRICAL fit of the measured launches (one "round" = the tiles
    // one workgroup per CU covers: the second resident workgroup shares its CU's LDS fill path and roughly doubles the unit time), not a residency count.
    // Round 4 re-calibration (profiles/r04_gemm64_ring4_harness.log, r04_gemm_vs_vendor.log): 6.9-8.4 us at
    // 12 K units, 17.7 at 48 (<= 256 tiles: one tile per CU slot), 54-68 us for the 576 tiles x 64 units of the Llama o_proj shape — the old
    // 5 + 0.55 nk over-estimated the short-K / few-tile case by 13 us and sent the Q-Former's 512 x 768 x 3072 residual GEMM to the phased
    // kernel's K-split (27.8 us, 12 launches per step) although this kernel runs it in 18-19
    const int64_t t64 = (int64_t)((p.M + 63) / 64) * (p.N / 64);
    return 3.5f + nk * 0.30f * (float)((t64 + 255) / 256);
  }
  return 5.0f + (float)((t128 + 511) / 512) * nk * (t128 >= 512 ? 1.33f : 1.17f);
}

// the phased kernel's ROPE epilogue runs 1.2-1.45x its plain-store estimate (166 vs 116 us at 1088 rows, 942-1039 vs 779 at 9216: dispatch audits of round 4)
constexpr float kP8RopePenalty = 1.3f;
static bool p8_wanted(const GemmParams& p, int heavy, int* miw, bool rope = false) {
  const int g_p8_mode = stllm_options().gemm_p8, g_sk_mode = stllm_options().gemm_sk;
  if (g_p8_mode == 0 || p.ws == nullptr) return false;
  if (p.ws_bytes < kSkFlagBytes + (int64_t)256 * 256 * 256 * 4) return false;
  const float est = stllm_gemm_p8_estimate_us(p.M, p.N, p.K, heavy, miw) * (rope ? kP8RopePenalty : 1.0f);
  if (g_p8_mode == 3 || g_p8_mode == 4) { *miw = g_p8_mode; return true; }
  if (g_p8_mode == 1) return true;
  if (g_sk_mode >= 1) return false;   // a forced stream-K tile (tests / experiments) wins over the automatic choice
  return est < 0.93f * old_kernels_estimate_us(p);
}

// One-wave-per-SIMD kernel (gemm_w4.inc): same eligibility as the phased kernel.  mode 1 / 32 / 34 / 44: always (cost model /
// 192 x 128 / 192 x 256 / 256 x 256 tile).  auto: only plans WITHOUT a K-split exchange (whole or partial rounds of whole tiles:
// there it measured 3-30 % faster than the alternatives, profiles/r02_w4_probe.md) on large-M problems, when its calibrated
// estimate beats both other kernel families — in the bench path that is the ViT's N = 1408 GEMMs (proj, fc2) as ONE round
// of 242 tiles of 192 x 128.
static bool w4_wanted(const GemmParams& p, int heavy, int* shape, float p8_est_us) {
  const int g_w4_mode = stllm_options().gemm_w4, g_p8_mode = stllm_options().gemm_p8, g_sk_mode = stllm_options().gemm_sk;
  if (g_w4_mode == 0 || p.ws == nullptr) return false;
  if (p.ws_bytes < kSkFlagBytes + (int64_t)256 * 256 * 256 * 4) return false;
  int split = 1;
  const float est = stllm_gemm_w4_estimate_us(p.M, p.N, p.K, heavy, shape, &split);
  if (g_w4_mode == 32 || g_w4_mode == 34 || g_w4_mode == 44 || g_w4_mode == 42 || g_w4_mode == 43 || g_w4_mode == 33 || g_w4_mode == 24 || g_w4_mode == 22) { *shape = g_w4_mode; return true; }
  if (g_w4_mode == 1) return true;
  if (g_sk_mode >= 1 || g_p8_mode == 1 || g_p8_mode == 3 || g_p8_mode == 4 || g_p8_mode == 0) return false;   // a forced / disabled kernel family (tests / experiments) wins
  // Round 2, first version (2-buffer LDS ring): no gain inside bench.py, where the weights come from HBM (proj + fc2 5.39 ms per
  // step against 5.46 ms); with the 3-buffer ring of the 192 x 128 tile the harness measures the same time on cold weights as on
  // warm ones (tools/gemm_harness ... <cold MiB>, profiles/r02_w4_ring3.md) and the rule below is the default (-1 == 2).
  if (g_w4_mode != 2 && g_w4_mode != -1 && g_w4_mode != 3 && g_w4_mode != 4) return false;
  // exchange-free plans only: the one candidate with a K-split that the estimates favour, the Llama qkv GEMM (576 x 12288 x 4096 as
  // 256 whole 192 x 128 tiles + 32 tiles split 8 ways), measured 76.8 vs 81.3 us in the harness but 81.7 us inside bench.py
  // (profiles/r02c_bench_kernel_stats.md) — no gain, so the 128 x 128 kernel keeps it and no model GEMM depends on a w4 exchange
  // round 4: with at least one whole round in front of it (and >= 1024 rows) the end-of-launch reduction waits for nobody and its traffic hides behind the rounds — at 2304 rows the
  // Llama qkv / o / down GEMMs measure 222 / 87 / 189 us on such plans against 298 / 97 / 228 on the phased kernel (profiles/r04_gemm_dispatch_audit_c3_1gpu.log)
  int plan5[5] = {0, 0, 1, 32, 0};
  if (split != 1 && stllm_gemm_w4_plan(p.M, p.N, p.K, heavy, *shape, plan5) != STLLM_OK) return false;
  const float other = p8_est_us < old_kernels_estimate_us(p) ? p8_est_us : old_kernels_estimate_us(p);
  if (p.M < 1024) {
    // a few hundred rows (prefill): exchange-free plans of the even tiles only — one partial round of whole tiles is robust inside the model (the 128 x 256 rule above
    // is the same idea), a K-split at these sizes is not (round 2).  At 296 rows (c5's masked prefill) the qkv GEMM runs 49 us as 192 tiles of 192 x 128 against 68 us on
    // the phased kernel's ROPE epilogue (profiles/r04_gemm_dispatch_audit_c5.log)
    if (p.M < 128) return false;
    static const int kEven[4] = {32, 42, 34, 24};
    float best = 1.0e30f;
    for (int i = 0; i < 4; ++i) {
      int q5[5];
      if (kEven[i] == 24 && !stllm_options().gemm_w4_wide) continue;   // the A/B switch of the 128 x 256 tile covers this rule too
      if (stllm_gemm_w4_plan(p.M, p.N, p.K, heavy, kEven[i], q5) != STLLM_OK || q5[2] != 1 || q5[0] != 0) continue;   // ONE partial round (q = 0): the audited case (lm_head as 2.9 rounds of 192 x 128 lost 24 us to the phased kernel)
      if ((float)q5[4] < best) { best = (float)q5[4]; *shape = kEven[i]; }
    }
    if (best < 0.97f * other) return true;
    // A/B switch (STLLM_GEMM_W4 = 4, round 6): plans with ONE whole round + a K-split remainder below 1024 rows too — the Llama gate / up GEMM at 576 rows is
    // 258 tiles of 192 x 256 = one round + 2 tiles split 32 ways inside an XCD (the phased kernel runs the same plan at 1.25 us per unit, this kernel at 1.10)
    if (g_w4_mode == 4 && split != 1 && plan5[0] >= 1 && est < 0.97f * other) return true;
    return false;
  }
  if (split != 1 && plan5[0] < 1) return false;
  return est < 0.97f * other;
}

template <typename T>
int dispatch_epi(const stllm_gemm_args* a, const GemmParams& p, hipStream_t stream) {
  if constexpr (!Elem<T>::kIsF32) {
    const StllmOptions& o_ = stllm_options();   // this thread's options: env parsed once, before any decision below
    const int gemv_mode = o_.gemm_gemv, g_sk_mode = o_.gemm_sk, g_p8_mode = o_.gemm_p8, g_w4_mode = o_.gemm_w4;
    const bool forced_tiles = g_sk_mode >= 1 || g_p8_mode == 1 || g_p8_mode == 3 || g_p8_mode == 4 || g_w4_mode == 1 || g_w4_mode == 32 || g_w4_mode == 34 || g_w4_mode == 44 || g_w4_mode == 42 || g_w4_mode == 43 || g_w4_mode == 33 || g_w4_mode == 24 || g_w4_mode == 22;   // tests / experiments
    // M <= 8 since round 2: 5-row decode steps 6.99 -> 6.02 ms at Vicuna-7B size (profiles/r02_decode_bench.log)
    if (p.nx) {   // fused RMSNorm operand: only the GEMV kernel computes it
      const int rc = a->epilogue != STLLM_EPI_PATCH ? stllm_gemv_launch(a->dtype, a->epilogue, p, stream) : STLLM_ERR_UNSUPPORTED;
      if (rc == STLLM_ERR_UNSUPPORTED) stllm_set_error("stllm_gemm(a_norm): shape outside the decode regime (M=%d N=%d K=%d): run stllm_rmsnorm first", p.M, p.N, p.K);
      return rc;
    }
    if (p.M <= (gemv_mode == 1 ? 4 : 16) && a->epilogue != STLLM_EPI_PATCH && gemv_mode != 0 && !forced_tiles) {   // 9..16 rows: matrix-core GEMV only (gemv.hip)
      const int rc = stllm_gemv_launch(a->dtype, a->epilogue, p, stream);
      if (rc != STLLM_ERR_UNSUPPORTED) return rc;
    }
    // Tall-tile one-round kernel (gemm_t1.inc, round 6): prefill-sized problems whose 144-row tiles fill ONE round of the 256 CUs with the
    // whole K extent per workgroup — the Llama o_proj / down GEMMs at 576 rows as 4 x 64 = 256 tiles of 144 x 64 (nothing exchanged between
    // workgroups; the other kernels split K over 2.7-5 workgroups per tile there).  STLLM_GEMM_T1 = 2 / 4: forced wherever eligible (tests).
    if (o_.gemm_t1 != 0 && (a->epilogue == STLLM_EPI_RESID || (a->epilogue == STLLM_EPI_STORE && a->act == STLLM_ACT_NONE))) {
      int t1_shape = 0;
      if (o_.gemm_t1 > 0) t1_shape = o_.gemm_t1;
      else if (!forced_tiles && p.M >= 128 && p.M <= 1152 && p.K <= 6144) {   // (K = 11008, the down projection: 56-64 us in the harness but 88 us inside the model with cold weights,
                                                                              //  against 71 us on the phased kernel — profiles/r06_bench_ab_t1.log; o_proj: 38 vs 44 us in the model)
        const int tm = (p.M + 143) / 144;
        for (int s = 2; s <= 4 && !t1_shape; s += 2) {
          const int tiles = tm * (p.N / (32 * s));
          if (p.N % (32 * s) == 0 && tiles >= 208 && tiles <= 256 && tm * 144 - p.M < 72) t1_shape = s;
        }
      }
      if (t1_shape) {
        const int rc = std::is_same<T, bf16_t>::value ? stllm_gemm_t1_launch_bf16(a->epilogue, t1_shape, p, stream)
                                                      : stllm_gemm_t1_launch_f16(a->epilogue, t1_shape, p, stream);
        if (rc != STLLM_ERR_UNSUPPORTED) return rc;
      }
    }
    // W-direct kernel (gemm_wd.inc, round 6): the caller supplied the fragment-major copy of W and the (32 WM) x 256 tiles make ONE round of the chip —
    // the Llama qkv GEMM at 576 rows as 5 x 48 = 240 tiles of 128 x 256.  STLLM_GEMM_WD = 4 / 6: forced wherever eligible (tests) | 0: off.
    if (o_.gemm_wd != 0 && p.Wf != nullptr && (a->epilogue == STLLM_EPI_ROPE || a->epilogue == STLLM_EPI_SWIGLU || (a->epilogue == STLLM_EPI_STORE && a->act == STLLM_ACT_NONE))) {
      int wd_shape = 0;
      if (o_.gemm_wd > 0) wd_shape = o_.gemm_wd;
      else if (!forced_tiles && p.M >= 128 && p.M <= 1536 && p.N % 256 == 0) {
        const int cands[2] = {4, 6};
        for (int ci = 0; ci < 2 && !wd_shape; ++ci) {
          const int bm = 32 * cands[ci], tmr = (p.M + bm - 1) / bm, tiles = tmr * (p.N / 256);
          if (tiles >= 200 && tiles <= 256 && tmr * bm - p.M <= bm / 2) wd_shape = cands[ci];
        }
      }
      if (wd_shape) {
        const int rc = std::is_same<T, bf16_t>::value ? stllm_gemm_wd_launch_bf16(a->epilogue, wd_shape, p, stream)
                                                      : stllm_gemm_wd_launch_f16(a->epilogue, wd_shape, p, stream);
        if (rc != STLLM_ERR_UNSUPPORTED) return rc;
      }
    }
    int miw = 4;
    const int heavy = (a->epilogue == STLLM_EPI_STORE && a->act == STLLM_ACT_GELU) ? 2
                    : (a->epilogue == STLLM_EPI_RESID || (a->epilogue == STLLM_EPI_STORE && a->out_is_f32)) ? 1 : 0;
    const bool rope_epi = a->epilogue == STLLM_EPI_ROPE;
    const bool p8_ok = a->epilogue != STLLM_EPI_PATCH && p8_wanted(p, heavy, &miw, rope_epi);
    if (a->epilogue != STLLM_EPI_PATCH) {
      int shape = 44, miw2 = 4;
      const float p8_est = stllm_gemm_p8_estimate_us(p.M, p.N, p.K, heavy, &miw2) * (rope_epi ? kP8RopePenalty : 1.0f);
      const int thin_bit = (a->epilogue == STLLM_EPI_STORE || a->epilogue == STLLM_EPI_RESID) ? 8 : 0;   // gemm_w4.inc: thin tail rows allowed
      // experiment switch (STLLM_GEMM_W4=3): the automatic rule, plus the Llama prefill qkv GEMM (ROPE epilogue, M < 1024) on the
      // 192 x 128 one-wave tile with its K-split remainder — 78.5 vs 82.9 us in the harness, re-measured in the model every round
      bool w4_go = w4_wanted(p, heavy | thin_bit, &shape, p8_est);
      if (!w4_go && g_w4_mode == 3 && a->epilogue == STLLM_EPI_ROPE && p.M < 1024 && p.ws != nullptr &&
          p.ws_bytes >= kSkFlagBytes + (int64_t)256 * 256 * 256 * 4) { shape = 32; w4_go = true; }
      // Llama prefill qkv (ROPE epilogue, a few hundred rows): the 128 x 256 one-wave tile when its tiles make ONE well-filled round and nothing is
      // exchanged — 576 x 12288 x 4096: 5 x 48 = 240 tiles, 68.8-72 us against 80-82 us on the 128 x 128 kernel (480 tiles = 1.9 rounds) in the
      // harness, profiles/r04_w4_128x256_qkv.md (the vendor library picks the same macro tile for this shape)
      // (any epilogue: the split verify mode's inner GEMM of the same layer is a plain fp32 store at K' = 3 K — 175 vs 230 us on this tile, profiles/r04_gemm_dispatch_audit_x3.log)
      if (!w4_go && (g_w4_mode == -1 || g_w4_mode == 2 || g_w4_mode == 3 || g_w4_mode == 4) && o_.gemm_w4_wide && p.M <= 640 && !forced_tiles && p.ws != nullptr &&
          p.ws_bytes >= kSkFlagBytes + (int64_t)256 * 256 * 256 * 4 && p.N % 256 == 0) {
        const int t24 = ((p.M + 127) / 128) * (p.N / 256);
        if (t24 >= 192 && t24 <= 256) { shape = 24; w4_go = true; }
      }
      if (w4_go) {
        const int rc = std::is_same<T, bf16_t>::value ? stllm_gemm_w4_launch_bf16(a->epilogue, shape, p, stream)
                                                      : stllm_gemm_w4_launch_f16(a->epilogue, shape, p, stream);
        if (rc != STLLM_ERR_UNSUPPORTED) return rc;
      }
    }
    if (p8_ok) {
      const int rc = std::is_same<T, bf16_t>::value ? stllm_gemm_p8_launch_bf16(a->epilogue, miw, p, stream)
                                                    : stllm_gemm_p8_launch_f16(a->epilogue, miw, p, stream);
      if (rc != STLLM_ERR_UNSUPPORTED) return rc;
    }
  }
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

int stllm_gemm_bf16x3(const stllm_gemm_args* a, void* stream);   // split3.hip
int stllm_prof_begin(const stllm_gemm_args* a, void* stream);     // profile.cpp
void stllm_prof_end(int idx, int rc, const stllm_gemm_args* a, void* stream);

extern "C" int stllm_gemm(const stllm_gemm_args* a, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  STLLM_CHECK_ARG(a != nullptr, "stllm_gemm: null args");
  STLLM_CHECK_ARG(a->dtype >= STLLM_BF16 && a->dtype <= STLLM_BF16X3, "stllm_gemm: bad dtype %d", a->dtype);
  if (a->dtype == STLLM_BF16X3) {   // split3.hip: split A, ONE bf16 GEMM with K' = 3 K (back through this entry point), fp32 post-epilogue
    STLLM_CHECK_ARG(a->M > 0 && a->N > 0 && a->K > 0 && a->N % 128 == 0 && a->out != nullptr && aligned16(a->out) && a->W && aligned16(a->W),
                    "stllm_gemm(BF16X3): empty problem, N %% 128 != 0 or null / misaligned W / out (M=%d N=%d K=%d)", a->M, a->N, a->K);
    return stllm_gemm_bf16x3(a, stream_);
  }
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
  } else if (a->a_norm_x) {
    STLLM_CHECK_ARG(a->dtype != STLLM_F32 && a->M <= 8 && a->a_norm_gamma && aligned16(a->a_norm_x) && aligned16(a->a_norm_gamma) &&
                        a->a_norm_ldx >= a->K && a->a_norm_ldx % 4 == 0 && a->K % panel == 0 && a->a_rows_per_batch == 0,
                    "stllm_gemm(a_norm): the fused RMSNorm operand needs a 16-bit dtype, M <= 8, flat 16-byte aligned fp32 rows (M=%d)", a->M);
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
  p.debug = stllm_options().gemm_debug;
  p.ws = reinterpret_cast<char*>(a->workspace); p.ws_bytes = a->workspace_bytes;
  p.nx = a->a_norm_x; p.nx_ld = a->a_norm_ldx; p.ngamma = a->a_norm_gamma; p.neps = a->a_norm_eps;
  p.Wf = reinterpret_cast<const char*>(a->w_frag);
  p.a_rpb = a->a_rows_per_batch; p.a_bs_b = a->a_batch_stride * eb;
  p.o_rpb = a->o_rows_per_batch; p.o_bs = a->o_batch_stride;
  const int prof_rec = stllm_prof_begin(a, stream_);   // profile.cpp: HIP events around this launch when the caller asked for them
  int rc;
  switch (a->dtype) {
    case STLLM_BF16: rc = dispatch_epi<bf16_t>(a, p, stream); break;
    case STLLM_F16: rc = dispatch_epi<f16_t>(a, p, stream); break;
    default: rc = dispatch_epi<float>(a, p, stream); break;
  }
  stllm_prof_end(prof_rec, rc, a, stream_);
  return rc;
}


extern "C" int stllm_gemm_workspace_status(const void* workspace, void* stream_) {
  if (!workspace) return 0;
  unsigned w = 0;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (hipMemcpyAsync(&w, reinterpret_cast<const unsigned*>(workspace) + kSkErrWord, sizeof(w), hipMemcpyDeviceToHost, stream) != hipSuccess ||
      hipStreamSynchronize(stream) != hipSuccess) {
    stllm_set_error("stllm_gemm_workspace_status: cannot read the workspace");
    return STLLM_ERR_HIP;
  }
  if (w) stllm_set_error("stllm_gemm: a split-K workgroup timed out waiting for a peer (flags 0x%x): results of that launch are invalid", w);
  return (int)w;
}

extern "C" int64_t stllm_gemm_workspace_bytes(void) { return kSkFlagBytes + (int64_t)256 * 256 * 256 * 4; }  // = 512 slabs of 128x128 too

