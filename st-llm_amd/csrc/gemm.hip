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
  int a_rpb; int64_t a_bs_b;       // A 2-level rows: rows per batch, batch stride (bytes)
  int o_rpb; int64_t o_bs;         // out 2-level rows (elements)
};

constexpr int kRowBytes = 128;  // one K panel row

template <int BM, int BN> struct Tile {
  static constexpr int WM = BM / 2, WN = BN / 2, MI = WM / 32, NI = WN / 32;
  static constexpr int kStageBytes = (BM + BN) * kRowBytes;
  static constexpr int kLdsBytes = 2 * kStageBytes;
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

template <typename T, int BM, int BN, int EPI>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmParams p) {
  using TL = Tile<BM, BN>;
  constexpr int MI = TL::MI, NI = TL::NI, WM = TL::WM, WN = TL::WN;
  constexpr int EB = Elem<T>::kBytes;
  constexpr int kElemsPerPanel = kRowBytes / EB;
  constexpr bool kPatch = (EPI == STLLM_EPI_PATCH);
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;

  // ---- tile coordinates -------------------------------------------------------------------
  const int nwg = p.tiles_m * p.tiles_n;
  const int id = xcd_remap(blockIdx.x, nwg);
  int tm, tn;
  if (p.tiles_m <= p.tiles_n) { tm = id % p.tiles_m; tn = id / p.tiles_m; }
  else                        { tn = id % p.tiles_n; tm = id / p.tiles_n; }
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- staging plan: piece = 8 rows x 128 B = one wave-wide global_load_lds --------------------
  // combined tile rows [0,BM) = A, [BM,BM+BN) = W; wave w owns pieces w, w+4, ...
  constexpr int kPieces = (BM + BN) / 8;
  constexpr int kPiecesPerWave = kPieces / 4;
  const char* gsrc[kPiecesPerWave];
#pragma unroll
  for (int j = 0; j < kPiecesPerWave; ++j) {
    const int piece = wave + 4 * j;
    const int r = piece * 8 + (lane >> 3);          // row in combined tile
    const int c = lane & 7;                          // physical 16-B chunk in the LDS row
    const int lc = c ^ ((r >> 1) & 7);               // logical chunk to fetch
    if (r < BM) {
      int gr = m0 + r; gr = gr < p.M ? gr : p.M - 1;
      int64_t off = (int64_t)gr * p.lda_b;
      if (p.a_rpb > 0) { const int bb = gr / p.a_rpb; off = (int64_t)bb * p.a_bs_b + (int64_t)(gr - bb * p.a_rpb) * p.lda_b; }
      gsrc[j] = p.A + off + lc * 16;
    } else {
      int gr = n0 + (r - BM); gr = gr < p.N ? gr : p.N - 1;
      gsrc[j] = p.W + (int64_t)gr * p.ldw_b + lc * 16;
    }
  }

  auto stage = [&](int t, int buf) {
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

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const int nk = p.K / kElemsPerPanel;
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int t = 0; t < nk; ++t) {
    const int cur = t & 1;
    if (t + 1 < nk) stage(t + 1, cur ^ 1);
    const char* base = smem + cur * TL::kStageBytes;
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
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ---- epilogue ------------------------------------------------------------------------------
  // C layout of 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  const int col_l = li;
  auto out_off = [&](int row) -> int64_t {
    if (p.o_rpb > 0) { const int bb = row / p.o_rpb; return (int64_t)bb * p.o_bs + (int64_t)(row - bb * p.o_rpb) * p.ldo; }
    return (int64_t)row * p.ldo;
  };
  const int cbase = n0 + wn * WN;
  float bv[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) bv[j] = p.bias ? p.bias[cbase + j * 32 + col_l] : 0.0f;

  if constexpr (EPI == STLLM_EPI_STORE) {
    auto body = [&](auto act_c, auto f32_c) {
      constexpr int ACT = decltype(act_c)::value;
      constexpr bool F32 = decltype(f32_c)::value;
#pragma unroll
      for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (row >= p.M) continue;
          const int64_t ro = out_off(row);
#pragma unroll
          for (int j = 0; j < NI; ++j) {
            float v = acc[i][j][r] + bv[j];
            if constexpr (ACT == STLLM_ACT_GELU) v = gelu_erf(v);
            if constexpr (ACT == STLLM_ACT_RELU) v = fmaxf(v, 0.0f);
            const int64_t o = ro + cbase + j * 32 + col_l;
            if constexpr (F32) reinterpret_cast<float*>(p.out)[o] = v;
            else store_elem<T>(p.out, o, v);
          }
        }
      }
    };
    using std::integral_constant;
    const int key = p.act * 2 + (p.out_is_f32 ? 1 : 0);
    switch (key) {
      case 0: body(integral_constant<int, 0>{}, integral_constant<bool, false>{}); break;
      case 1: body(integral_constant<int, 0>{}, integral_constant<bool, true>{}); break;
      case 2: body(integral_constant<int, 1>{}, integral_constant<bool, false>{}); break;
      case 3: body(integral_constant<int, 1>{}, integral_constant<bool, true>{}); break;
      case 4: body(integral_constant<int, 2>{}, integral_constant<bool, false>{}); break;
      default: body(integral_constant<int, 2>{}, integral_constant<bool, true>{}); break;
    }
  } else {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row >= p.M) continue;
        if constexpr (EPI == STLLM_EPI_RESID) {
          const int64_t ro = out_off(row);
#pragma unroll
          for (int j = 0; j < NI; ++j) {
            const int col = cbase + j * 32 + col_l;
            reinterpret_cast<float*>(p.out)[ro + col] = acc[i][j][r] + bv[j] + p.resid[(int64_t)row * p.ldr + col];
          }
        } else if constexpr (EPI == STLLM_EPI_PATCH) {
          const int n = row >> 8, pp = row & 255;
          const int64_t orow = (int64_t)n * 257 + 1 + pp;
#pragma unroll
          for (int j = 0; j < NI; ++j) {
            const int col = cbase + j * 32 + col_l;
            reinterpret_cast<float*>(p.out)[orow * p.ldo + col] = acc[i][j][r] + bv[j] + p.aux0[(int64_t)(1 + pp) * p.N + col];
          }
        } else if constexpr (EPI == STLLM_EPI_SWIGLU) {
          static_assert(EPI != STLLM_EPI_SWIGLU || NI == 2, "SwiGLU epilogue needs a 64-column wave tile");
          const int g = cbase >> 6;  // 64-column group: [32 gate | 32 up]
          const float gate = acc[i][0][r] + bv[0];
          const float up = acc[i][NI - 1][r] + bv[NI - 1];
          store_elem<T>(p.out, out_off(row) + g * 32 + col_l, silu_f(gate) * up);
        } else if constexpr (EPI == STLLM_EPI_ROPE) {
          static_assert(EPI != STLLM_EPI_ROPE || NI == 2, "RoPE epilogue needs a 64-column wave tile");
          float x1 = acc[i][0][r] + bv[0], x2 = acc[i][NI - 1][r] + bv[NI - 1];
          if (cbase < p.rope_cols) {
            const int g = cbase >> 6;              // 64-col group: head = g/2, half = g&1
            const int fi = (g & 1) * 32 + col_l;   // frequency index 0..63
            const int pos = row % p.rope_seq;
            const float c = p.aux0[pos * 64 + fi], sn = p.aux1[pos * 64 + fi];
            const float y1 = x1 * c - x2 * sn, y2 = x2 * c + x1 * sn;
            x1 = y1; x2 = y2;
          }
          const int64_t ro = out_off(row);
          store_elem<T>(p.out, ro + cbase + col_l, x1);
          store_elem<T>(p.out, ro + cbase + 32 + col_l, x2);
        }
      }
    }
  }
}

template <typename T, int BM, int BN, int EPI>
int launch(const GemmParams& p0, hipStream_t stream) {
  GemmParams p = p0;
  p.tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = p.N / BN;
  const int lds = Tile<BM, BN>::kLdsBytes;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_kernel<T, BM, BN, EPI>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_kernel<T, BM, BN, EPI>), dim3(p.tiles_m * p.tiles_n), dim3(256), lds, stream, p);
  STLLM_CHECK_LAUNCH("stllm_gemm");
  return STLLM_OK;
}

template <typename T, int EPI>
int dispatch_tile(const GemmParams& p, hipStream_t stream) {
  if constexpr (EPI == STLLM_EPI_SWIGLU || EPI == STLLM_EPI_ROPE) {
    // these epilogues pair columns inside a 64-column wave tile
    if (p.M <= 64) return launch<T, 64, 128, EPI>(p, stream);
    return launch<T, 128, 128, EPI>(p, stream);
  } else {
    // small problems: smaller tiles so that more CUs get work
    const int64_t t128 = (int64_t)((p.M + 127) / 128) * (p.N / 128);
    if (t128 < 192) return launch<T, 64, 64, EPI>(p, stream);
    return launch<T, 128, 128, EPI>(p, stream);
  }
}

template <typename T>
int dispatch_epi(const stllm_gemm_args* a, const GemmParams& p, hipStream_t stream) {
  switch (a->epilogue) {
    case STLLM_EPI_STORE: return dispatch_tile<T, STLLM_EPI_STORE>(p, stream);
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
  STLLM_CHECK_ARG(a->out != nullptr, "stllm_gemm: null out");
  if (a->epilogue == STLLM_EPI_RESID) STLLM_CHECK_ARG(a->resid != nullptr, "stllm_gemm(RESID): null resid");
  if (a->epilogue == STLLM_EPI_ROPE)
    STLLM_CHECK_ARG(a->aux0 && a->aux1 && a->rope_seq > 0 && a->rope_cols % 128 == 0, "stllm_gemm(ROPE): need cos/sin tables, rope_seq, rope_cols%%128==0");

  GemmParams p{};
  p.A = reinterpret_cast<const char*>(a->A); p.lda_b = a->lda * eb;
  p.W = reinterpret_cast<const char*>(a->W); p.ldw_b = a->ldw * eb;
  p.bias = a->bias; p.out = a->out; p.ldo = a->ldo; p.resid = a->resid; p.ldr = a->ldr;
  p.aux0 = a->aux0; p.aux1 = a->aux1; p.frames = a->frames;
  p.rope_seq = a->rope_seq; p.rope_cols = a->rope_cols;
  p.M = a->M; p.N = a->N; p.K = K; p.act = a->act; p.out_is_f32 = a->out_is_f32;
  p.a_rpb = a->a_rows_per_batch; p.a_bs_b = a->a_batch_stride * eb;
  p.o_rpb = a->o_rows_per_batch; p.o_bs = a->o_batch_stride;
  switch (a->dtype) {
    case STLLM_BF16: return dispatch_epi<bf16_t>(a, p, stream);
    case STLLM_F16: return dispatch_epi<f16_t>(a, p, stream);
    default: return dispatch_epi<float>(a, p, stream);
  }
}
