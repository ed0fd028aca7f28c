// Skinny GEMM for the decode regime:  C[M,N] = epilogue(A[M,K] @ W[N,K]^T) with M <= 4 (M <= 8 with
// stllm_set_option("gemm_gemv", 2): the 5 beams of demo.py's beam search; staged, off by default until timed), bf16 / fp16.
//
// One token per sequence means every weight matrix is streamed from HBM once per step and used for M <= 4 rows: the
// problem is HBM-bound (Vicuna-7B: 13.2 GB per token), the matrix cores are useless (a 64-row MFMA tile would be 98 % padding).
//   * A (M x K, <= 88 KB) is staged once per workgroup in LDS;
//   * every WAVE owns the column pair (c, c + 32) of a 64-column group — exactly the [32 gate | 32 up] / [x_lo | x_hi]
//     partners of the packed SwiGLU / RoPE layouts (pack.py), so every epilogue is local to the wave;
//   * the wave walks K in 512-element steps: each lane streams 16 bytes of both W rows per step (two steps in flight),
//     widens bf16 / fp16 to fp32 and accumulates M x 2 dot products in registers; a 6-step butterfly reduces across lanes;
//   * lane 0 applies bias / GELU / residual / SwiGLU / RoPE and stores.
// Algorithmic bytes per launch: N*K*2 (W) + M*K*2 (A) + outputs; the roofline is HBM (~6.3 TB/s achievable).
#include <type_traits>

#include "gemm_common.h"

namespace {
using namespace sg;

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

template <typename T, int EPI, int ACT, bool OF32, int MR>
__global__ __launch_bounds__(256) void gemv_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // A: MR rows x K x 2 bytes
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = p.K;
  const int row_bytes = K * 2;
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

  const int q = blockIdx.x * 4 + wave;           // column-pair index
  const int c0 = (q >> 5) * 64 + (q & 31), c1 = c0 + 32;
  if (c0 >= p.N) return;                          // (no barrier after this point)
  const char* w0 = p.W + (int64_t)c0 * p.ldw_b;
  const char* w1 = p.W + (int64_t)c1 * p.ldw_b;

  float acc[MR][2];
#pragma unroll
  for (int m = 0; m < MR; ++m) acc[m][0] = acc[m][1] = 0.0f;

  auto fma8 = [&](i32x4 wa, i32x4 wb, int kb) {   // kb = byte offset of this lane's 8 elements
    float fa[8], fb[8];
    widen8<T>(wa, fa);
    widen8<T>(wb, fb);
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      float x[8];
      widen8<T>(*reinterpret_cast<const i32x4*>(smem + m * row_bytes + kb), x);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        acc[m][0] = fmaf(x[e], fa[e], acc[m][0]);
        acc[m][1] = fmaf(x[e], fb[e], acc[m][1]);
      }
    }
  };
  // lane l owns bytes [l*16 + 1024*step, +16) of both W rows; two steps (4 x 16-byte loads per lane) in flight per iteration
  int kb = lane * 16;
  for (; kb + 1024 < row_bytes; kb += 2048) {
    const i32x4 a0 = *reinterpret_cast<const i32x4*>(w0 + kb), b0 = *reinterpret_cast<const i32x4*>(w1 + kb);
    const i32x4 a1 = *reinterpret_cast<const i32x4*>(w0 + kb + 1024), b1 = *reinterpret_cast<const i32x4*>(w1 + kb + 1024);
    fma8(a0, b0, kb);
    fma8(a1, b1, kb + 1024);
  }
  if (kb < row_bytes) {   // odd number of steps, or the ragged last step of K % 512 != 0
    const i32x4 a0 = *reinterpret_cast<const i32x4*>(w0 + kb), b0 = *reinterpret_cast<const i32x4*>(w1 + kb);
    fma8(a0, b0, kb);
  }
#pragma unroll
  for (int m = 0; m < MR; ++m) {
    acc[m][0] = wave_sum(acc[m][0]);
    acc[m][1] = wave_sum(acc[m][1]);
  }
  if (lane != 0) return;

  auto out_off = [&](int row) -> int64_t {
    if (p.o_rpb > 0) { const int bb = row / p.o_rpb; return (int64_t)bb * p.o_bs + (int64_t)(row - bb * p.o_rpb) * p.ldo; }
    return (int64_t)row * p.ldo;
  };
  const float b0 = p.bias ? p.bias[c0] : 0.0f, b1 = p.bias ? p.bias[c1] : 0.0f;
  for (int m = 0; m < MR; ++m) {
    if (m >= p.M) break;
    float x = acc[m][0] + b0, y = acc[m][1] + b1;
    if constexpr (EPI == STLLM_EPI_RESID) {
      float* o = reinterpret_cast<float*>(p.out) + out_off(m);
      o[c0] = p.resid[(int64_t)m * p.ldr + c0] + x;
      o[c1] = p.resid[(int64_t)m * p.ldr + c1] + y;
    } else if constexpr (EPI == STLLM_EPI_SWIGLU) {
      store_elem<T>(p.out, out_off(m) + (c0 >> 6) * 32 + (c0 & 31), silu_f(x) * y);
    } else {
      if constexpr (EPI == STLLM_EPI_ROPE) {
        if (c0 < p.rope_cols) {
          const int fi = ((c0 >> 6) & 1) * 32 + (c0 & 31);
          const int pos = m % p.rope_seq;
          const float c = p.aux0[pos * 64 + fi], s = p.aux1[pos * 64 + fi];
          const float xr = x * c - y * s;
          y = y * c + x * s;
          x = xr;
        }
      } else {
        if constexpr (ACT == STLLM_ACT_GELU) { x = gelu_erf(x); y = gelu_erf(y); }
        if constexpr (ACT == STLLM_ACT_RELU) { x = fmaxf(x, 0.0f); y = fmaxf(y, 0.0f); }
      }
      if constexpr (OF32 && EPI == STLLM_EPI_STORE) {
        float* o = reinterpret_cast<float*>(p.out) + out_off(m);
        o[c0] = x;
        o[c1] = y;
      } else {
        store_elem<T>(p.out, out_off(m) + c0, x);
        store_elem<T>(p.out, out_off(m) + c1, y);
      }
    }
  }
}

template <typename T, int EPI, int ACT, bool OF32, int MR>
int launch_gemv(const GemmParams& p, hipStream_t stream) {
  auto kern = gemv_kernel<T, EPI, ACT, OF32, MR>;
  const int lds = MR * p.K * 2;
  static int lds_set = 0;
  if (lds > lds_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return STLLM_ERR_UNSUPPORTED;
    lds_set = lds;
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

// M <= 8 (the caller decides how far it goes), 16-bit dtypes, the staged rows of A (1, 2, 4, 6 or 8 x K x 2 bytes) must fit the LDS;
// returns STLLM_ERR_UNSUPPORTED otherwise (caller falls back to the tile kernels)
int stllm_gemv_launch(int dtype, int epilogue, const sg::GemmParams& p, hipStream_t stream) {
  if (p.M < 1 || p.M > 8) return STLLM_ERR_UNSUPPORTED;
  const int mr = p.M <= 2 ? p.M : (p.M + 1) / 2 * 2;
  if (p.N % 64 || p.K % 8 || (int64_t)mr * p.K * 2 > 150 * 1024) return STLLM_ERR_UNSUPPORTED;
  if ((!p.nx && (p.lda_b % 16)) || (p.ldw_b % 16)) return STLLM_ERR_UNSUPPORTED;
  if (p.nx && (p.K % 4 || p.nx_ld % 4 || p.a_rpb > 0)) return STLLM_ERR_UNSUPPORTED;
  if (dtype == STLLM_BF16) return dispatch_gemv<bf16_t>(epilogue, p, stream);
  if (dtype == STLLM_F16) return dispatch_gemv<f16_t>(epilogue, p, stream);
  return STLLM_ERR_UNSUPPORTED;
}
