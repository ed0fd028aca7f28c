// Skinny GEMM for the decode regime:  C[M,N] = epilogue(A[M,K] @ W[N,K]^T) with M <= 4 (M <= 8 with
// stllm_set_option("gemm_gemv", 2): the 5 beams of demo.py's beam search; staged, off by default until timed), bf16 / fp16.
//
// One token per sequence means every weight matrix is streamed from HBM once per step and used for M <= 4 rows: the
// problem is HBM-bound (Vicuna-7B: 13.2 GB per token), the matrix cores are useless (a 64-row MFMA tile would be 98 % padding).
//   * A (M x K, <= 88 KB) is staged once per workgroup in LDS;
//   * every WAVE owns the column pair (c, c + 32) of a 64-column group — exactly the [32 gate | 32 up] / [x_lo | x_hi]
//     partners of the packed SwiGLU / RoPE layouts (pack.py), so every epilogue is local to the wave;
//   * the wave walks K in 512-element steps: each lane streams 16 bytes of both W rows per step, two batches of four steps in
//     flight (16 x 16-byte loads per lane), the first batch requested before A is staged,
//     widens bf16 / fp16 to fp32 and accumulates M x 2 dot products in registers; a 6-step butterfly reduces across lanes;
//   * lane 0 applies bias / GELU / residual / SwiGLU / RoPE and stores.
// Algorithmic bytes per launch: N*K*2 (W) + M*K*2 (A) + outputs; the roofline is HBM (~6.3 TB/s achievable).
#include <cstdlib>
#include <type_traits>

#include "gemm_common.h"

namespace {
using namespace sg;

// VALU kernel: the weights are read exactly once per launch, in whole 128-byte lines per instruction: non-temporal loads (decode
// 3.23 -> 3.08 ms/token).  NOT in the matrix-core kernel, whose two loads per step touch the two halves of the same lines: there
// the hint costs the second half its cache hit (5 beams: 4.34 -> 4.66 ms).
__device__ __forceinline__ i32x4 ldw(const char* ptr) { return __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(ptr)); }

template <typename T> __device__ __forceinline__ void widen8(i32x4 v, float* f);
template <> __device__ __forceinline__ void widen8<bf16_t>(i32x4 v, float* f) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    f[2 * e] = __builtin_bit_cast(float, (uint32_t)v[e] << 16);
    f[2 * e + 1] = __builtin_bit_cast(float, (uint32_t)v[e] & 0xffff0000u);
  }
}
template <> __device__ __forceinline__ void widen8<f16_t>(i32x4 v, float* f) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {   // (bit_cast<f16x2>(v[e]) mis-compiles here with ROCm 7.2: every e read element 0)
    const uint32_t u = (uint32_t)v[e];
    f[2 * e] = (float)__builtin_bit_cast(_Float16, (uint16_t)(u & 0xffffu));
    f[2 * e + 1] = (float)__builtin_bit_cast(_Float16, (uint16_t)(u >> 16));
  }
}

// epilogue of one output row for the column pair (c0, c1 = c0 + 32) — the [32 gate | 32 up] / [x_lo | x_hi] partners
// acc + a.lo * b.lo + a.hi * b.hi on packed 16-bit pairs (products are exact in fp32)
typedef __attribute__((ext_vector_type(2))) __bf16 gemv_bf16x2;
typedef __attribute__((ext_vector_type(2))) _Float16 gemv_f16x2;
template <typename T> __device__ __forceinline__ float dot2_acc(int a, int b, float c);
template <> __device__ __forceinline__ float dot2_acc<bf16_t>(int a, int b, float c) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(gemv_bf16x2, a), __builtin_bit_cast(gemv_bf16x2, b), c, false);
}
template <> __device__ __forceinline__ float dot2_acc<f16_t>(int a, int b, float c) {
  return __builtin_amdgcn_fdot2(__builtin_bit_cast(gemv_f16x2, a), __builtin_bit_cast(gemv_f16x2, b), c, false);
}

template <typename T, int EPI, int ACT, bool OF32>
__device__ __forceinline__ void gemv_store(const GemmParams& p, int m, int c0, int c1, float x, float y) {
  int64_t oo = (int64_t)m * p.ldo;
  if (p.o_rpb > 0) { const int bb = m / p.o_rpb; oo = (int64_t)bb * p.o_bs + (int64_t)(m - bb * p.o_rpb) * p.ldo; }
  if constexpr (EPI == STLLM_EPI_RESID) {
    float* o = reinterpret_cast<float*>(p.out) + oo;
    o[c0] = p.resid[(int64_t)m * p.ldr + c0] + x;
    o[c1] = p.resid[(int64_t)m * p.ldr + c1] + y;
  } else if constexpr (EPI == STLLM_EPI_SWIGLU) {
    store_elem<T>(p.out, oo + (c0 >> 6) * 32 + (c0 & 31), silu_f(x) * y);
  } else {
    if constexpr (EPI == STLLM_EPI_ROPE) {
      if (c0 < p.rope_cols) {
        const int fi = ((c0 >> 6) & 1) * 32 + (c0 & 31);
        const int pos = m % p.rope_seq;
        const float c = p.aux0[pos * 64 + fi], sn = p.aux1[pos * 64 + fi];
        const float xr = x * c - y * sn;
        y = y * c + x * sn;
        x = xr;
      }
    } else {
      if constexpr (ACT == STLLM_ACT_GELU) { x = gelu_erf(x); y = gelu_erf(y); }
      if constexpr (ACT == STLLM_ACT_RELU) { x = fmaxf(x, 0.0f); y = fmaxf(y, 0.0f); }
    }
    if constexpr (OF32 && EPI == STLLM_EPI_STORE) {
      float* o = reinterpret_cast<float*>(p.out) + oo;
      o[c0] = x;
      o[c1] = y;
    } else {
      store_elem<T>(p.out, oo + c0, x);
      store_elem<T>(p.out, oo + c1, y);
    }
  }
}

template <typename T, int EPI, int ACT, bool OF32, int MR>
__global__ __launch_bounds__(256) void gemv_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // A: MR rows x K x 2 bytes
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = p.K;
  const int row_bytes = K * 2;
  // ---- this wave's column pair; its first kU steps of both W rows are requested BEFORE A is staged (the weight stream does not
  // depend on A: the HBM round trip overlaps the staging — and, with the fused RMSNorm operand, the two passes over x) -------------
  const int q = blockIdx.x * 4 + wave;           // column-pair index
  const int c0 = (q >> 5) * 64 + (q & 31), c1 = c0 + 32;
  const bool active = c0 < p.N;
  const char* w0 = p.W + (active ? (int64_t)c0 * p.ldw_b : 0);
  const char* w1 = p.W + (active ? (int64_t)c1 * p.ldw_b : 0);
  constexpr int kU = 4;                           // steps (1024 bytes of K per wave) per batch; two batches in flight
  const int lane_b = lane * 16;
  auto load_batch = [&](i32x4* wa, i32x4* wb, int kbase) {
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int kb = kbase + u * 1024 + lane_b;
      if (kb < row_bytes) {
        wa[u] = ldw(w0 + kb);
        wb[u] = ldw(w1 + kb);
      }
    }
  };
  i32x4 wa0[kU], wb0[kU], wa1[kU], wb1[kU];
  load_batch(wa0, wb0, 0);
  // ---- stage A (2-level row indexing honoured) -------------------------------------------------------------------
  if (p.nx) {
    // A := RMSNorm(x) * gamma, computed here (Llama's input / post-attention norm fused into the projection of the decode step):
    // same arithmetic as norm_row_kernel<T, true> (variance in fp32 over the row, (x * rstd) * gamma, one rounding to T).  Every
    // workgroup recomputes it for the <= 8 rows — 16-32 KB of L2 reads against the weight panel it then streams from HBM.
    __shared__ float red[8];
    const int nvec = K >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(p.ngamma);
    for (int m = 0; m < MR; ++m) {
      const int gr = m < p.M ? m : p.M - 1;
      const float4* xr = reinterpret_cast<const float4*>(p.nx + (int64_t)gr * p.nx_ld);
      float ss = 0.0f;
      for (int c = tid; c < nvec; c += 256) {
        const float4 v = xr[c];
        ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
      }
      ss = wave_sum(ss);
      if (lane == 0) red[(m & 1) * 4 + wave] = ss;      // two slots: row m + 1 may write while a slow wave still reads row m's
      __syncthreads();
      ss = red[(m & 1) * 4] + red[(m & 1) * 4 + 1] + red[(m & 1) * 4 + 2] + red[(m & 1) * 4 + 3];
      const float rstd = rsqrtf(ss / (float)K + p.neps);
      for (int c = tid; c < nvec; c += 256) {
        const float4 v = xr[c], g = g4[c];
        uint2 pk;
        pk.x = Elem<T>::pack2(v.x * rstd * g.x, v.y * rstd * g.y);
        pk.y = Elem<T>::pack2(v.z * rstd * g.z, v.w * rstd * g.w);
        *reinterpret_cast<uint2*>(smem + m * row_bytes + c * 8) = pk;
      }
    }
  } else {
    for (int m = 0; m < MR; ++m) {
      int gr = m < p.M ? m : p.M - 1;
      int64_t off = (int64_t)gr * p.lda_b;
      if (p.a_rpb > 0) { const int bb = gr / p.a_rpb; off = (int64_t)bb * p.a_bs_b + (int64_t)(gr - bb * p.a_rpb) * p.lda_b; }
      const char* src = p.A + off;
      for (int c = tid * 16; c < row_bytes; c += 256 * 16)
        *reinterpret_cast<i32x4*>(smem + m * row_bytes + c) = *reinterpret_cast<const i32x4*>(src + c);
    }
  }
  __syncthreads();

  if (!active) return;                            // (no barrier after this point)

  float acc[MR][2];
#pragma unroll
  for (int m = 0; m < MR; ++m) acc[m][0] = acc[m][1] = 0.0f;

  // 8 elements of both W rows against the MR rows of A: v_dot2c_f32_{bf16,f16} multiplies two 16-bit pairs and adds them to an
  // fp32 accumulator in one instruction — no widening, 8 instructions per row instead of 24 (5 beams: the kernel is HBM-bound
  // again instead of VALU-bound)
  auto fma8 = [&](i32x4 wa, i32x4 wb, int kb) {   // kb = byte offset of this lane's 8 elements
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      const i32x4 x = *reinterpret_cast<const i32x4*>(smem + m * row_bytes + kb);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc[m][0] = dot2_acc<T>(x[e], wa[e], acc[m][0]);
        acc[m][1] = dot2_acc<T>(x[e], wb[e], acc[m][1]);
      }
    }
  };
  auto use_batch = [&](const i32x4* wa, const i32x4* wb, int kbase) {
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int kb = kbase + u * 1024 + lane_b;
      if (kb < row_bytes) fma8(wa[u], wb[u], kb);   // ragged last step (K % 512 != 0): the lanes past the row end sit out
    }
  };
  // lane l owns bytes [l*16 + 1024*step, +16) of both W rows; two batches of kU steps (2 x 8 x 16-byte loads per lane) in flight
  for (int kbase = 0; kbase < row_bytes; kbase += 2 * kU * 1024) {
    load_batch(wa1, wb1, kbase + kU * 1024);
    use_batch(wa0, wb0, kbase);
    load_batch(wa0, wb0, kbase + 2 * kU * 1024);
    use_batch(wa1, wb1, kbase + kU * 1024);
  }
#pragma unroll
  for (int m = 0; m < MR; ++m) {
    acc[m][0] = wave_sum(acc[m][0]);
    acc[m][1] = wave_sum(acc[m][1]);
  }
  if (lane != 0) return;

  const float b0 = p.bias ? p.bias[c0] : 0.0f, b1 = p.bias ? p.bias[c1] : 0.0f;
  for (int m = 0; m < MR; ++m) {
    if (m >= p.M) break;
    gemv_store<T, EPI, ACT, OF32>(p, m, c0, c1, acc[m][0] + b0, acc[m][1] + b1);
  }
}

// ---- 3 <= M <= 16 rows: the same stream on the matrix cores -----------------------------------------------------------------------
// With more than two rows the VALU kernel above is bound by its own FMAs (5 beams: 160 VALU instructions per 2 KB of weights),
// not by HBM.  One v_mfma_f32_16x16x32 multiplies 16 rows of A with 32 k x 16 columns of W (1 KB of weights) in 8 cycles; rows
// M..15 are padding that costs nothing.
//   * a workgroup of 8 waves owns 16 columns — or, for the SwiGLU / RoPE epilogues, the 32 columns [c, c + 16) and [c + 32, c + 48)
//     of a 64-column group, so that every lane ends up with the (c, c + 32) partners of the packed layouts in two accumulators;
//   * lane l = (column n = l % 16, k-group g = l / 16) loads bytes [16 g, 16 g + 16) of both 64-byte halves of W row n's 128-byte
//     step: the four k-groups of a row cover one contiguous 64-byte sector per instruction, and the two MFMAs of a step take
//     k = [0, 32) and [32, 64) in natural order; A's rows are read with the same addressing;
//   * the 8 waves take every 8th 64-element k step (together 1 KB of every row per round); their partial sums meet in LDS and
//     wave 0 adds them in wave order (deterministic) and runs the epilogue.
constexpr int kGmWaves = 8;
template <typename T> __device__ __forceinline__ f32x4 mfma16(i32x4 a, i32x4 b, f32x4 c);
template <> __device__ __forceinline__ f32x4 mfma16<bf16_t>(i32x4 a, i32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x4 mfma16<f16_t>(i32x4 a, i32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// one output element of the column-local epilogues (STORE / RESID)
template <typename T, int EPI, int ACT, bool OF32>
__device__ __forceinline__ void gemv_store1(const GemmParams& p, int m, int c, float x) {
  int64_t oo = (int64_t)m * p.ldo;
  if (p.o_rpb > 0) { const int bb = m / p.o_rpb; oo = (int64_t)bb * p.o_bs + (int64_t)(m - bb * p.o_rpb) * p.ldo; }
  if constexpr (EPI == STLLM_EPI_RESID) {
    reinterpret_cast<float*>(p.out)[oo + c] = p.resid[(int64_t)m * p.ldr + c] + x;
  } else {
    if constexpr (ACT == STLLM_ACT_GELU) x = gelu_erf(x);
    if constexpr (ACT == STLLM_ACT_RELU) x = fmaxf(x, 0.0f);
    if constexpr (OF32) reinterpret_cast<float*>(p.out)[oo + c] = x;
    else store_elem<T>(p.out, oo + c, x);
  }
}

template <typename T, int EPI, int ACT, bool OF32>
__global__ __launch_bounds__(64 * kGmWaves, 4) void gemv_mfma_kernel(const GemmParams p) {
  // PAIR: the epilogue combines columns c and c + 32 (SwiGLU, RoPE) => the workgroup owns two 16-column blocks; otherwise ONE
  // block of 16 columns, which doubles the number of workgroups (o_proj / down_proj, N = 4096: 256 instead of 128 — one per CU)
  constexpr bool PAIR = (EPI == STLLM_EPI_SWIGLU || EPI == STLLM_EPI_ROPE);
  constexpr int NL = PAIR ? 6 : 4;              // 16-byte loads per lane and step
  __shared__ float red[kGmWaves * (PAIR ? 8 : 4) * 64];   // partial sums of the 8 waves
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n16 = lane & 15, g = lane >> 4;
  const int unit = blockIdx.x;
  const int col0 = PAIR ? (unit >> 1) * 64 + (unit & 1) * 16 + n16 : unit * 16 + n16, col1 = col0 + 32;
  const char* w0 = p.W + (int64_t)col0 * p.ldw_b + g * 16;
  const char* w1 = p.W + (int64_t)(PAIR ? col1 : col0) * p.ldw_b + g * 16;
  // A is NOT staged: the <= 16 rows (<= 350 KB, L2 / L1 resident) are read straight into the operand registers next to the weight
  // stream — no staging pass, no barrier before the first MFMA, no LDS footprint (K = 11008 with 8 or 16 rows fits as well).
  // Rows >= M of the 16-row operand: any valid row (their products land in output rows that are never stored).
  const int am = n16 < p.M ? n16 : p.M - 1;
  int64_t aoff = (int64_t)am * p.lda_b;
  if (p.a_rpb > 0) { const int bb = am / p.a_rpb; aoff = (int64_t)bb * p.a_bs_b + (int64_t)(am - bb * p.a_rpb) * p.lda_b; }
  const char* ax = p.A + aoff + g * 16;
  const int nsteps = p.K / 64;
  constexpr int U = 2;                           // steps per batch; two batches (2 x 2 NL x 16-byte loads per lane) in flight
  i32x4 wq[2][U][NL];
  // Steps past the end of K are NOT skipped (a load or an MFMA under a wave-uniform `if (st < nsteps)` was observed to read
  // stale registers on gfx950 whenever only part of a batch was valid — the waits the compiler counts for the full batch let
  // younger data slip): they re-read the last step and multiply it with a zeroed A operand instead.
  auto load_batch = [&](int b, int st0) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int st = st0 + kGmWaves * u;
      st = st < nsteps ? st : nsteps - 1;
      wq[b][u][0] = *reinterpret_cast<const i32x4*>(ax + st * 128);
      wq[b][u][1] = *reinterpret_cast<const i32x4*>(ax + st * 128 + 64);
      wq[b][u][2] = *reinterpret_cast<const i32x4*>(w0 + st * 128);
      wq[b][u][3] = *reinterpret_cast<const i32x4*>(w0 + st * 128 + 64);
      if constexpr (PAIR) {
        wq[b][u][4] = *reinterpret_cast<const i32x4*>(w1 + st * 128);
        wq[b][u][5] = *reinterpret_cast<const i32x4*>(w1 + st * 128 + 64);
      }
    }
  };
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  auto use_batch = [&](int b, int st0) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int keep = (st0 + kGmWaves * u < nsteps) ? -1 : 0;   // wave-uniform mask
      i32x4 x0 = wq[b][u][0], x1 = wq[b][u][1];
#pragma unroll
      for (int e = 0; e < 4; ++e) { x0[e] &= keep; x1[e] &= keep; }
      acc0 = mfma16<T>(x0, wq[b][u][2], acc0);
      acc0 = mfma16<T>(x1, wq[b][u][3], acc0);
      if constexpr (PAIR) {
        acc1 = mfma16<T>(x0, wq[b][u][4], acc1);
        acc1 = mfma16<T>(x1, wq[b][u][5], acc1);
      }
    }
  };
  load_batch(0, wave);
  for (int st = wave; st < nsteps; st += 2 * U * kGmWaves) {
    load_batch(1, st + U * kGmWaves);
    use_batch(0, st);
    load_batch(0, st + 2 * U * kGmWaves);
    use_batch(1, st + U * kGmWaves);
  }
  constexpr int NR = PAIR ? 8 : 4;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    red[(wave * NR + r) * 64 + lane] = acc0[r];
    if constexpr (PAIR) red[(wave * NR + 4 + r) * 64 + lane] = acc1[r];
  }
  __syncthreads();
  if (wave != 0) return;
  const float b0 = p.bias ? p.bias[col0] : 0.0f, b1 = (PAIR && p.bias) ? p.bias[col1] : 0.0f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int m = 4 * g + r;   // D register r of lane l is D[4 (l / 16) + r][l % 16]
    float x = 0.0f, y = 0.0f;
#pragma unroll
    for (int w = 0; w < kGmWaves; ++w) {
      x += red[(w * NR + r) * 64 + lane];
      if constexpr (PAIR) y += red[(w * NR + 4 + r) * 64 + lane];
    }
    if (m < p.M) {
      if constexpr (PAIR) gemv_store<T, EPI, ACT, OF32>(p, m, col0, col1, x + b0, y + b1);
      else gemv_store1<T, EPI, ACT, OF32>(p, m, col0, x + b0);
    }
  }
}

template <typename T, int EPI, int ACT, bool OF32>
int launch_gemv_mfma(const GemmParams& p, hipStream_t stream) {
  auto kern = gemv_mfma_kernel<T, EPI, ACT, OF32>;
  constexpr bool PAIR = (EPI == STLLM_EPI_SWIGLU || EPI == STLLM_EPI_ROPE);
  hipLaunchKernelGGL(kern, dim3(p.N / (PAIR ? 32 : 16)), dim3(64 * kGmWaves), 0, stream, p);
  STLLM_CHECK_LAUNCH("stllm_gemm(gemv_mfma)");
  {
    static const char* kEpi[] = {"STORE", "RESID", "SWIGLU", "ROPE", "PATCH"};
    static char name[96];
    static bool named = false;
    if (!named) {
      snprintf(name, sizeof(name), "gemv_mfma_kernel<%s,%s,%d,%d>", std::is_same<T, bf16_t>::value ? "bf16_t" : "f16_t", kEpi[EPI], ACT, (int)OF32);
      named = true;
    }
    stllm_set_last_kernel(name);
  }
  return STLLM_OK;
}

template <typename T>
int dispatch_gemv_mfma(int epilogue, const GemmParams& p, hipStream_t stream) {
  switch (epilogue) {
    case STLLM_EPI_STORE:
      if (p.act == STLLM_ACT_NONE) return p.out_is_f32 ? launch_gemv_mfma<T, STLLM_EPI_STORE, 0, true>(p, stream) : launch_gemv_mfma<T, STLLM_EPI_STORE, 0, false>(p, stream);
      if (p.act == STLLM_ACT_GELU && !p.out_is_f32) return launch_gemv_mfma<T, STLLM_EPI_STORE, 1, false>(p, stream);
      break;
    case STLLM_EPI_RESID: return launch_gemv_mfma<T, STLLM_EPI_RESID, 0, false>(p, stream);
    case STLLM_EPI_SWIGLU: return launch_gemv_mfma<T, STLLM_EPI_SWIGLU, 0, false>(p, stream);
    case STLLM_EPI_ROPE: return launch_gemv_mfma<T, STLLM_EPI_ROPE, 0, false>(p, stream);
  }
  return STLLM_ERR_UNSUPPORTED;
}

template <typename T, int EPI, int ACT, bool OF32, int MR>
int launch_gemv(const GemmParams& p, hipStream_t stream) {
  auto kern = gemv_kernel<T, EPI, ACT, OF32, MR>;
  const int lds = MR * p.K * 2;
  static StllmPerDevice lds_dev;   // largest dynamic-LDS size opted into, per device ordinal
  bool lds_first;
  const int lds_d = lds_dev.enter(&lds_first);
  if (lds_first || lds > lds_dev.value[lds_d]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return STLLM_ERR_UNSUPPORTED;
    lds_dev.value[lds_d] = lds;
    lds_dev.done(lds_d);
  }
  const int pairs = p.N / 2;
  hipLaunchKernelGGL(kern, dim3((pairs + 3) / 4), dim3(256), lds, stream, p);
  STLLM_CHECK_LAUNCH("stllm_gemm(gemv)");
  {
    static const char* kEpi[] = {"STORE", "RESID", "SWIGLU", "ROPE", "PATCH"};
    static char name[96];
    static bool named = false;
    if (!named) {
      snprintf(name, sizeof(name), "gemv_kernel<%s,%s,%d,%d,%d>", std::is_same<T, bf16_t>::value ? "bf16_t" : "f16_t", kEpi[EPI], ACT, (int)OF32, MR);
      named = true;
    }
    stllm_set_last_kernel(name);
  }
  return STLLM_OK;
}

template <typename T, int EPI, int ACT, bool OF32>
int launch_gemv_m(const GemmParams& p, hipStream_t stream) {
  switch (p.M) {
    case 1: return launch_gemv<T, EPI, ACT, OF32, 1>(p, stream);
    case 2: return launch_gemv<T, EPI, ACT, OF32, 2>(p, stream);
    case 3: case 4: return launch_gemv<T, EPI, ACT, OF32, 4>(p, stream);
    case 5: case 6: return launch_gemv<T, EPI, ACT, OF32, 6>(p, stream);
    default: return launch_gemv<T, EPI, ACT, OF32, 8>(p, stream);   // M = 7, 8
  }
}

template <typename T>
int dispatch_gemv(int epilogue, const GemmParams& p, hipStream_t stream) {
  switch (epilogue) {
    case STLLM_EPI_STORE:
      if (p.act == STLLM_ACT_NONE) return p.out_is_f32 ? launch_gemv_m<T, STLLM_EPI_STORE, 0, true>(p, stream) : launch_gemv_m<T, STLLM_EPI_STORE, 0, false>(p, stream);
      if (p.act == STLLM_ACT_GELU && !p.out_is_f32) return launch_gemv_m<T, STLLM_EPI_STORE, 1, false>(p, stream);
      break;
    case STLLM_EPI_RESID: return launch_gemv_m<T, STLLM_EPI_RESID, 0, false>(p, stream);
    case STLLM_EPI_SWIGLU: return launch_gemv_m<T, STLLM_EPI_SWIGLU, 0, false>(p, stream);
    case STLLM_EPI_ROPE: return launch_gemv_m<T, STLLM_EPI_ROPE, 0, false>(p, stream);
  }
  return STLLM_ERR_UNSUPPORTED;
}

}  // namespace

// M <= 16 (the caller decides how far it goes), 16-bit dtypes.  M >= 3 (option "gemv_mfma": 0 never, 1 from M = 1): the matrix-core
// kernel (K % 64 == 0; A is read straight from global memory, no LDS limit); otherwise, for M <= 8, the VALU kernel,
// whose staged rows (1, 2, 4, 6 or 8 x K x 2 bytes) must fit the LDS.  Returns STLLM_ERR_UNSUPPORTED when neither applies (the caller
// falls back to the tile kernels).
int stllm_gemv_launch(int dtype, int epilogue, const sg::GemmParams& p, hipStream_t stream) {
  if (p.M < 1 || p.M > 16) return STLLM_ERR_UNSUPPORTED;
  if (p.N % 64 || p.K % 8 || (p.ldw_b % 16) || (!p.nx && (p.lda_b % 16))) return STLLM_ERR_UNSUPPORTED;
  const int g_gemv_mfma = stllm_options().gemv_mfma;
  const int from = g_gemv_mfma == 0 ? 17 : g_gemv_mfma >= 1 ? g_gemv_mfma : 3;   // n >= 1: from M = n
  if (!p.nx && p.M >= from && p.K % 64 == 0) {
    if (dtype == STLLM_BF16) return dispatch_gemv_mfma<bf16_t>(epilogue, p, stream);
    if (dtype == STLLM_F16) return dispatch_gemv_mfma<f16_t>(epilogue, p, stream);
    return STLLM_ERR_UNSUPPORTED;
  }
  if (p.M > 8) return STLLM_ERR_UNSUPPORTED;
  const int mr = p.M <= 2 ? p.M : (p.M + 1) / 2 * 2;
  if ((int64_t)mr * p.K * 2 > 150 * 1024) return STLLM_ERR_UNSUPPORTED;
  if (p.nx && (p.K % 4 || p.nx_ld % 4 || p.a_rpb > 0)) return STLLM_ERR_UNSUPPORTED;
  if (dtype == STLLM_BF16) return dispatch_gemv<bf16_t>(epilogue, p, stream);
  if (dtype == STLLM_F16) return dispatch_gemv<f16_t>(epilogue, p, stream);
  return STLLM_ERR_UNSUPPORTED;
}
