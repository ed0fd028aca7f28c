// HBM-bound glue kernels: row gather (token-block assembly / masking / embedding lookup),
// temporal mean, ViT CLS rows, MVM cosine loss rows.  All fp32, float4-vectorised, coalesced.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src_a, int64_t ld_a,
                                                          const float* __restrict__ src_b, int64_t ld_b,
                                                          const int32_t* __restrict__ idx_a,
                                                          const float* __restrict__ add, int64_t ld_add,
                                                          const int32_t* __restrict__ idx_add,
                                                          float* __restrict__ dst, int64_t ld_dst, int n_rows, int D,
                                                          float scale) {
  const int row = blockIdx.x;
  if (row >= n_rows) return;
  const int ia = idx_a[row];
  const float4* s = (ia >= 0) ? reinterpret_cast<const float4*>(src_a + (int64_t)ia * ld_a)
                              : reinterpret_cast<const float4*>(src_b + (int64_t)(-ia - 1) * ld_b);
  const float4* a = add ? reinterpret_cast<const float4*>(add + (int64_t)idx_add[row] * ld_add) : nullptr;
  float4* d = reinterpret_cast<float4*>(dst + (int64_t)row * ld_dst);
  for (int c = threadIdx.x; c < (D >> 2); c += blockDim.x) {
    float4 v = s[c];
    if (a) { const float4 w = a[c]; v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w; }
    if (scale != 1.0f) { v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale; }
    d[c] = v;
  }
}

__global__ __launch_bounds__(256) void mean_t_kernel(const float* __restrict__ x, float* __restrict__ out, int T,
                                                     int64_t J4) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (j >= J4) return;
  const float4* xb = reinterpret_cast<const float4*>(x) + (int64_t)b * T * J4 + j;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int t = 0; t < T; ++t) {
    const float4 v = xb[(int64_t)t * J4];
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  const float inv = 1.0f / (float)T;
  acc.x *= inv; acc.y *= inv; acc.z *= inv; acc.w *= inv;
  reinterpret_cast<float4*>(out)[(int64_t)b * J4 + j] = acc;
}

__global__ __launch_bounds__(256) void cls_rows_kernel(const float* __restrict__ cls, const float* __restrict__ pos,
                                                       float* __restrict__ x, int64_t ldx, int D) {
  const int n = blockIdx.x;
  float* d = x + (int64_t)n * 257 * ldx;
  for (int c = threadIdx.x; c < D; c += blockDim.x) d[c] = cls[c] + pos[c];
}

__global__ __launch_bounds__(256) void cosine_rows_kernel(const float* __restrict__ a, int64_t lda,
                                                          const int32_t* __restrict__ idx_a,
                                                          const float* __restrict__ b, int64_t ldb,
                                                          const int32_t* __restrict__ idx_b, float* __restrict__ out,
                                                          int n_rows, int D) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_rows) return;
  const float4* ar = reinterpret_cast<const float4*>(a + (int64_t)(idx_a ? idx_a[row] : row) * lda);
  const float4* br = reinterpret_cast<const float4*>(b + (int64_t)(idx_b ? idx_b[row] : row) * ldb);
  float ab = 0.f, aa = 0.f, bb = 0.f;
  for (int c = lane; c < (D >> 2); c += 64) {
    const float4 u = ar[c], v = br[c];
    ab += u.x * v.x + u.y * v.y + u.z * v.z + u.w * v.w;
    aa += u.x * u.x + u.y * u.y + u.z * u.z + u.w * u.w;
    bb += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  ab = wave_sum(ab); aa = wave_sum(aa); bb = wave_sum(bb);
  if (lane == 0) out[row] = 2.0f - 2.0f * ab / (sqrtf(aa) * sqrtf(bb));
}

// per-row cross-entropy: loss[i] = logsumexp(logits[i,:]) - logits[i,label[i]]; label < 0 -> 0
__global__ __launch_bounds__(256) void ce_rows_kernel(const float* __restrict__ logits, int64_t ldl,
                                                      const int32_t* __restrict__ labels, float* __restrict__ loss,
                                                      int V) {
  __shared__ float red[4];
  const int row = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int lab = labels[row];
  if (lab < 0) { if (threadIdx.x == 0) loss[row] = 0.0f; return; }
  const float* x = logits + (int64_t)row * ldl;
  float mx = -3.0e38f;
  for (int c = threadIdx.x; c < V; c += 256) mx = fmaxf(mx, x[c]);
  mx = wave_max(mx);
  if (lane == 0) red[w] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float s = 0.0f;
  for (int c = threadIdx.x; c < V; c += 256) s += expf(x[c] - mx);
  s = wave_sum(s);
  if (lane == 0) red[w] = s;
  __syncthreads();
  if (threadIdx.x == 0) loss[row] = logf(red[0] + red[1] + red[2] + red[3]) + mx - x[lab];
}

// fp32 -> compute dtype rows (A operand of a GEMM whose input lives in the fp32 residual stream)
template <typename T>
__global__ __launch_bounds__(256) void cast_rows_kernel(const float* __restrict__ x, int64_t ldx, void* __restrict__ out,
                                                        int64_t ldo, int D) {
  const int row = blockIdx.x;
  const float4* xr = reinterpret_cast<const float4*>(x + (int64_t)row * ldx);
  for (int c = threadIdx.x; c < (D >> 2); c += blockDim.x) {
    const float4 v = xr[c];
    if constexpr (Elem<T>::kIsF32) {
      reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + (int64_t)row * ldo)[c] = v;
    } else {
      uint2 pk;
      pk.x = Elem<T>::pack2(v.x, v.y);
      pk.y = Elem<T>::pack2(v.z, v.w);
      reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(out) + (int64_t)row * ldo)[c] = pk;
    }
  }
}

}  // namespace

extern "C" int stllm_gather_rows(const float* src_a, int64_t ld_a, const float* src_b, int64_t ld_b,
                                 const int32_t* idx_a, const float* add, int64_t ld_add, const int32_t* idx_add,
                                 float* dst, int64_t ld_dst, int n_rows, int D, float scale, void* stream) {
  STLLM_CHECK_ARG(src_a && idx_a && dst, "stllm_gather_rows: null pointer");
  STLLM_CHECK_ARG(n_rows > 0 && D > 0 && D % 4 == 0, "stllm_gather_rows: bad n_rows=%d D=%d", n_rows, D);
  STLLM_CHECK_ARG(ld_a % 4 == 0 && ld_dst % 4 == 0 && aligned16(src_a) && aligned16(dst), "stllm_gather_rows: misaligned");
  STLLM_CHECK_ARG(!src_b || (ld_b % 4 == 0 && aligned16(src_b)), "stllm_gather_rows: src_b misaligned");
  STLLM_CHECK_ARG(!add || (idx_add && ld_add % 4 == 0 && aligned16(add)), "stllm_gather_rows: add needs idx_add, aligned");
  hipLaunchKernelGGL(gather_rows_kernel, dim3(n_rows), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src_a, ld_a,
                     src_b, ld_b, idx_a, add, ld_add, idx_add, dst, ld_dst, n_rows, D, scale);
  STLLM_CHECK_LAUNCH("stllm_gather_rows");
  return STLLM_OK;
}

extern "C" int stllm_mean_t(const float* x, float* out, int B, int T, int64_t J, void* stream) {
  STLLM_CHECK_ARG(x && out && B > 0 && T > 0 && J > 0 && J % 4 == 0, "stllm_mean_t: bad args");
  STLLM_CHECK_ARG(aligned16(x) && aligned16(out), "stllm_mean_t: misaligned");
  const int64_t J4 = J / 4;
  hipLaunchKernelGGL(mean_t_kernel, dim3((unsigned)((J4 + 255) / 256), B), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), x, out, T, J4);
  STLLM_CHECK_LAUNCH("stllm_mean_t");
  return STLLM_OK;
}

extern "C" int stllm_vit_cls_rows(const float* cls, const float* pos, float* x, int64_t ldx, int n_frames, int D,
                                  void* stream) {
  STLLM_CHECK_ARG(cls && pos && x && n_frames > 0 && D > 0, "stllm_vit_cls_rows: bad args");
  hipLaunchKernelGGL(cls_rows_kernel, dim3(n_frames), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), cls, pos, x,
                     ldx, D);
  STLLM_CHECK_LAUNCH("stllm_vit_cls_rows");
  return STLLM_OK;
}

extern "C" int stllm_cosine_rows(const float* a, int64_t lda, const int32_t* idx_a, const float* b, int64_t ldb,
                                 const int32_t* idx_b, float* out, int n_rows, int D, void* stream) {
  STLLM_CHECK_ARG(a && b && out && n_rows > 0 && D > 0 && D % 4 == 0, "stllm_cosine_rows: bad args");
  STLLM_CHECK_ARG(lda % 4 == 0 && ldb % 4 == 0 && aligned16(a) && aligned16(b), "stllm_cosine_rows: misaligned");
  hipLaunchKernelGGL(cosine_rows_kernel, dim3((n_rows + 3) / 4), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a,
                     lda, idx_a, b, ldb, idx_b, out, n_rows, D);
  STLLM_CHECK_LAUNCH("stllm_cosine_rows");
  return STLLM_OK;
}

extern "C" int stllm_cross_entropy_rows(const float* logits, int64_t ldl, const int32_t* labels, float* loss,
                                        int n_rows, int V, void* stream) {
  STLLM_CHECK_ARG(logits && labels && loss && n_rows > 0 && V > 0, "stllm_cross_entropy_rows: bad args");
  hipLaunchKernelGGL(ce_rows_kernel, dim3(n_rows), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), logits, ldl,
                     labels, loss, V);
  STLLM_CHECK_LAUNCH("stllm_cross_entropy_rows");
  return STLLM_OK;
}

extern "C" int stllm_cast_rows(int dtype, const float* x, int64_t ldx, void* out, int64_t ldo, int M, int D,
                               void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  STLLM_CHECK_ARG(x && out && M > 0 && D > 0 && D % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0, "stllm_cast_rows: bad args");
  STLLM_CHECK_ARG(aligned16(x) && (reinterpret_cast<uintptr_t>(out) & 7) == 0, "stllm_cast_rows: misaligned");
  switch (dtype) {
    case STLLM_BF16: hipLaunchKernelGGL(cast_rows_kernel<bf16_t>, dim3(M), dim3(256), 0, stream, x, ldx, out, ldo, D); break;
    case STLLM_F16: hipLaunchKernelGGL(cast_rows_kernel<f16_t>, dim3(M), dim3(256), 0, stream, x, ldx, out, ldo, D); break;
    case STLLM_F32: hipLaunchKernelGGL(cast_rows_kernel<float>, dim3(M), dim3(256), 0, stream, x, ldx, out, ldo, D); break;
    default: stllm_set_error("stllm_cast_rows: bad dtype %d", dtype); return STLLM_ERR_BAD_DTYPE;
  }
  STLLM_CHECK_LAUNCH("stllm_cast_rows");
  return STLLM_OK;
}
