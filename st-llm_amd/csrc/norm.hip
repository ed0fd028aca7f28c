// LayerNorm / RMSNorm over fp32 rows: HBM-bound streaming kernels.
// One wave per row, row cached in registers (float4 per lane per step), fp32 two-pass statistics
// via wave shuffles; writes the compute-dtype copy (next GEMM's A operand) and/or an fp32 copy.
#include "common.h"

#include <type_traits>

namespace {

constexpr int kMaxVec = 32;  // D <= 64 lanes * 4 floats * 32 = 8192

// dtype STLLM_BF16X3 (the split verify mode): out_t is the A operand of the bf16x3 GEMM that follows — bf16 [M, ldo_t >= 3 D] = (hi | hi | lo) of the
// fp32 result, hi = bf16(v), lo = bf16(v - hi): exactly what stllm_split3_rows would make of the fp32 copy, without writing and re-reading it.
struct split3_t {};

// four consecutive columns 4 c .. 4 c + 3 of row `row` in the compute dtype
template <typename T>
__device__ __forceinline__ void store_row4(void* out_t, int64_t row, int64_t ldo_t, int c, int D, float4 o) {
  if constexpr (std::is_same<T, split3_t>::value) {
    uint16_t h[4], l[4];
    const float v[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      h[e] = f32_to_bf16_bits(v[e]);
      l[e] = f32_to_bf16_bits(v[e] - bf16_bits_to_f32(h[e]));
    }
    uint2 hv, lv;
    hv.x = h[0] | ((uint32_t)h[1] << 16); hv.y = h[2] | ((uint32_t)h[3] << 16);
    lv.x = l[0] | ((uint32_t)l[1] << 16); lv.y = l[2] | ((uint32_t)l[3] << 16);
    uint16_t* op = reinterpret_cast<uint16_t*>(out_t) + row * ldo_t + 4 * c;
    *reinterpret_cast<uint2*>(op) = hv;
    *reinterpret_cast<uint2*>(op + D) = hv;
    *reinterpret_cast<uint2*>(op + 2 * D) = lv;
  } else if constexpr (Elem<T>::kIsF32) {
    reinterpret_cast<float4*>(reinterpret_cast<float*>(out_t) + row * ldo_t)[c] = o;
  } else {
    uint2 pk;
    pk.x = Elem<T>::pack2(o.x, o.y);
    pk.y = Elem<T>::pack2(o.z, o.w);
    reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(out_t) + row * ldo_t)[c] = pk;
  }
}

template <typename T, bool RMS, int NV, int RW = 1>
__global__ __launch_bounds__(256) void norm_kernel(const float* __restrict__ x, int64_t ldx,
                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                   float eps, void* __restrict__ out_t, int64_t ldo_t,
                                                   float* __restrict__ out_f, int64_t ldo_f, int M, int D) {
  // RW rows per wave (RW = 2 for the short rows of the ViT: the kernel is one dependent chain per wave — load, two reductions, store —
  // and a launch of 4112 one-row waves was mostly dispatch + that chain's latency; two rows in flight per wave share gamma / beta and
  // overlap their chains)
  const int lane = threadIdx.x & 63;
  const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RW;
  if (row0 >= M) return;
  const int nvec = D >> 2;
  float4 v[RW][NV];
#pragma unroll
  for (int q = 0; q < RW; ++q) {
    const int row = row0 + q < M ? row0 + q : M - 1;   // (a clamped duplicate of the last row: never stored)
    const float4* xr = reinterpret_cast<const float4*>(x + (int64_t)row * ldx);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + i * 64;
      v[q][i] = (c < nvec) ? xr[c] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  // gamma / beta are requested together with the rows (round 3): behind the two reductions their L2 round trip was a second, fully exposed
  // latency in a kernel that is one dependent chain per wave (9.2 -> ~8 us at 4112 x 1408)
  const float4* g4 = reinterpret_cast<const float4*>(gamma);
  const float4* b4 = reinterpret_cast<const float4*>(beta);
  float4 gv[NV], bv[RMS ? 1 : NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + i * 64;
    const int cc = c < nvec ? c : 0;
    gv[i] = g4[cc];
    if constexpr (!RMS) bv[i] = b4[cc];
  }
  float s[RW], mean[RW], rstd[RW];
#pragma unroll
  for (int q = 0; q < RW; ++q) {
    s[q] = 0.0f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if constexpr (RMS) s[q] += v[q][i].x * v[q][i].x + v[q][i].y * v[q][i].y + v[q][i].z * v[q][i].z + v[q][i].w * v[q][i].w;
      else s[q] += v[q][i].x + v[q][i].y + v[q][i].z + v[q][i].w;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {   // the RW reductions step by step side by side (wave_sum's order per row)
#pragma unroll
    for (int q = 0; q < RW; ++q) s[q] += __shfl_xor(s[q], o, 64);
  }
  if constexpr (RMS) {
#pragma unroll
    for (int q = 0; q < RW; ++q) { mean[q] = 0.0f; rstd[q] = rsqrtf(s[q] / (float)D + eps); }
  } else {
    float qq[RW];
#pragma unroll
    for (int q = 0; q < RW; ++q) {
      mean[q] = s[q] / (float)D;
      qq[q] = 0.0f;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = lane + i * 64;
        if (c < nvec) {
          const float a = v[q][i].x - mean[q], b = v[q][i].y - mean[q], cc = v[q][i].z - mean[q], d = v[q][i].w - mean[q];
          qq[q] += a * a + b * b + cc * cc + d * d;
        }
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
      for (int q = 0; q < RW; ++q) qq[q] += __shfl_xor(qq[q], o, 64);
    }
#pragma unroll
    for (int q = 0; q < RW; ++q) rstd[q] = rsqrtf(qq[q] / (float)D + eps);
  }
#pragma unroll
  for (int q = 0; q < RW; ++q) {
    const int row = row0 + q;
    if (row >= M) continue;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + i * 64;
      if (c >= nvec) continue;
      const float4 g = gv[i];
      float4 o;
      o.x = (v[q][i].x - mean[q]) * rstd[q] * g.x; o.y = (v[q][i].y - mean[q]) * rstd[q] * g.y;
      o.z = (v[q][i].z - mean[q]) * rstd[q] * g.z; o.w = (v[q][i].w - mean[q]) * rstd[q] * g.w;
      if constexpr (!RMS) {
        const float4 b = bv[i];
        o.x += b.x; o.y += b.y; o.z += b.z; o.w += b.w;
      }
      if (out_f) reinterpret_cast<float4*>(out_f + (int64_t)row * ldo_f)[c] = o;
      if (out_t) store_row4<T>(out_t, row, ldo_t, c, D, o);
    }
  }
}

// Few-row variant (Llama prefill: 576 rows x 4096): one WORKGROUP (4 waves) per row so that 576 rows still put 576
// workgroups on the 256 CUs; cross-wave reduction through LDS.
template <typename T, bool RMS, int NV>
__global__ __launch_bounds__(256) void norm_row_kernel(const float* __restrict__ x, int64_t ldx,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float eps, void* __restrict__ out_t, int64_t ldo_t,
                                                       float* __restrict__ out_f, int64_t ldo_f, int M, int D) {
  __shared__ float red[2][4];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int row = blockIdx.x;
  const int nvec = D >> 2;
  const float4* xr = reinterpret_cast<const float4*>(x + (int64_t)row * ldx);
  float4 v[NV];
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = tid + i * 256;
    v[i] = (c < nvec) ? xr[c] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const float4* g4 = reinterpret_cast<const float4*>(gamma);
  const float4* b4 = reinterpret_cast<const float4*>(beta);
  float4 gv[NV], bv[RMS ? 1 : NV];   // requested with the row, consumed behind the reductions (see norm_kernel)
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = tid + i * 256;
    const int cc = c < nvec ? c : 0;
    gv[i] = g4[cc];
    if constexpr (!RMS) bv[i] = b4[cc];
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    if constexpr (RMS) s += v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w;
    else s += v[i].x + v[i].y + v[i].z + v[i].w;
  }
  s = wave_sum(s);
  if (lane == 0) red[0][w] = s;
  __syncthreads();
  s = red[0][0] + red[0][1] + red[0][2] + red[0][3];
  float mean = 0.0f, rstd;
  if constexpr (RMS) {
    rstd = rsqrtf(s / (float)D + eps);
  } else {
    mean = s / (float)D;
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = tid + i * 256;
      if (c < nvec) {
        const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
        q += a * a + b * b + cc * cc + d * d;
      }
    }
    q = wave_sum(q);
    if (lane == 0) red[1][w] = q;
    __syncthreads();
    q = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    rstd = rsqrtf(q / (float)D + eps);
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = tid + i * 256;
    if (c >= nvec) continue;
    const float4 g = gv[i];
    float4 o;
    o.x = (v[i].x - mean) * rstd * g.x; o.y = (v[i].y - mean) * rstd * g.y;
    o.z = (v[i].z - mean) * rstd * g.z; o.w = (v[i].w - mean) * rstd * g.w;
    if constexpr (!RMS) {
      const float4 b = bv[i];
      o.x += b.x; o.y += b.y; o.z += b.z; o.w += b.w;
    }
    if (out_f) reinterpret_cast<float4*>(out_f + (int64_t)row * ldo_f)[c] = o;
    if (out_t) store_row4<T>(out_t, row, ldo_t, c, D, o);
  }
}

template <typename T, bool RMS>
int launch_nv(const float* x, int64_t ldx, const float* gamma, const float* beta, float eps, void* out_t,
              int64_t ldo_t, float* out_f, int64_t ldo_f, int M, int D, hipStream_t stream) {
  if (M <= 2048 && (D >> 2) <= 256 * 8) {   // few rows: one workgroup per row
    const int nvr = ((D >> 2) + 255) / 256;
    dim3 g2(M), b2(256);
#define STLLM_NORMR_CASE(NV)                                                                                     \
  hipLaunchKernelGGL((norm_row_kernel<T, RMS, NV>), g2, b2, 0, stream, x, ldx, gamma, beta, eps, out_t, ldo_t, \
                     out_f, ldo_f, M, D)
    if (nvr <= 1) STLLM_NORMR_CASE(1);
    else if (nvr <= 2) STLLM_NORMR_CASE(2);
    else if (nvr <= 4) STLLM_NORMR_CASE(4);
    else STLLM_NORMR_CASE(8);
#undef STLLM_NORMR_CASE
    STLLM_CHECK_LAUNCH(RMS ? "stllm_rmsnorm" : "stllm_layernorm");
    return STLLM_OK;
  }
  const int nv = ((D >> 2) + 63) / 64;
  dim3 grid((M + 3) / 4), block(256);
#define STLLM_NORM_CASE(NV)                                                                                  \
  hipLaunchKernelGGL((norm_kernel<T, RMS, NV>), grid, block, 0, stream, x, ldx, gamma, beta, eps, out_t, ldo_t, \
                     out_f, ldo_f, M, D)
  if (nv <= 6 && M >= 2048 && stllm_options().norm_fast == 2) {   // two rows per wave (option norm_fast = 2)
    dim3 g2((M + 7) / 8);
    if (nv <= 3) hipLaunchKernelGGL((norm_kernel<T, RMS, 3, 2>), g2, block, 0, stream, x, ldx, gamma, beta, eps, out_t, ldo_t, out_f, ldo_f, M, D);
    else hipLaunchKernelGGL((norm_kernel<T, RMS, 6, 2>), g2, block, 0, stream, x, ldx, gamma, beta, eps, out_t, ldo_t, out_f, ldo_f, M, D);
    STLLM_CHECK_LAUNCH(RMS ? "stllm_rmsnorm" : "stllm_layernorm");
    return STLLM_OK;
  }
  if (nv <= 3) STLLM_NORM_CASE(3);
  else if (nv <= 6) STLLM_NORM_CASE(6);
  else if (nv <= 16) STLLM_NORM_CASE(16);
  else STLLM_NORM_CASE(kMaxVec);
#undef STLLM_NORM_CASE
  STLLM_CHECK_LAUNCH(RMS ? "stllm_rmsnorm" : "stllm_layernorm");
  return STLLM_OK;
}

template <bool RMS>
int norm_entry(int dtype, const float* x, int64_t ldx, const float* gamma, const float* beta, float eps,
               void* out_t, int64_t ldo_t, float* out_f, int64_t ldo_f, int M, int D, hipStream_t stream) {
  const char* nm = RMS ? "stllm_rmsnorm" : "stllm_layernorm";
  STLLM_CHECK_ARG(M > 0 && D > 0 && D % 4 == 0 && D <= 64 * 4 * kMaxVec, "%s: bad M=%d D=%d", nm, M, D);
  STLLM_CHECK_ARG(x && gamma && (RMS || beta), "%s: null input", nm);
  STLLM_CHECK_ARG(out_t || out_f, "%s: no output requested", nm);
  STLLM_CHECK_ARG(ldx % 4 == 0 && aligned16(x) && aligned16(gamma), "%s: x/gamma not 16-byte aligned", nm);
  if (out_f) STLLM_CHECK_ARG(ldo_f % 4 == 0 && aligned16(out_f), "%s: out_f32 misaligned", nm);
  if (out_t) STLLM_CHECK_ARG(ldo_t % 4 == 0 && (reinterpret_cast<uintptr_t>(out_t) & 7) == 0, "%s: out_t misaligned", nm);
  switch (dtype) {
    case STLLM_BF16: return launch_nv<bf16_t, RMS>(x, ldx, gamma, beta, eps, out_t, ldo_t, out_f, ldo_f, M, D, stream);
    case STLLM_F16: return launch_nv<f16_t, RMS>(x, ldx, gamma, beta, eps, out_t, ldo_t, out_f, ldo_f, M, D, stream);
    case STLLM_F32: return launch_nv<float, RMS>(x, ldx, gamma, beta, eps, out_t, ldo_t, out_f, ldo_f, M, D, stream);
    case STLLM_BF16X3:
      if (out_t) STLLM_CHECK_ARG(ldo_t >= 3 * (int64_t)D, "%s(BF16X3): out_t is the split image bf16 [M, ldo_t >= 3 D]", nm);
      return launch_nv<split3_t, RMS>(x, ldx, gamma, beta, eps, out_t, ldo_t, out_f, ldo_f, M, D, stream);
  }
  stllm_set_error("%s: bad dtype %d", nm, dtype);
  return STLLM_ERR_BAD_DTYPE;
}

}  // namespace

extern "C" int stllm_layernorm(int dtype, const float* x, int64_t ldx, const float* gamma, const float* beta,
                               float eps, void* out_t, int64_t ldo_t, float* out_f32, int64_t ldo_f, int M,
                               int D, void* stream) {
  return norm_entry<false>(dtype, x, ldx, gamma, beta, eps, out_t, ldo_t, out_f32, ldo_f, M, D,
                           reinterpret_cast<hipStream_t>(stream));
}

extern "C" int stllm_rmsnorm(int dtype, const float* x, int64_t ldx, const float* gamma, float eps, void* out_t,
                             int64_t ldo_t, float* out_f32, int64_t ldo_f, int M, int D, void* stream) {
  return norm_entry<true>(dtype, x, ldx, gamma, nullptr, eps, out_t, ldo_t, out_f32, ldo_f, M, D,
                          reinterpret_cast<hipStream_t>(stream));
}
