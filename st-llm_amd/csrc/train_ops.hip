// Backward / optimizer kernels that are NOT GEMM-shaped (gradients of the trainable parts + AdamW).
// All of them are HBM-bound streaming or reduction kernels: the design rules are coalesced 16-byte accesses, one wave per
// row for row reductions, deterministic two-stage column reductions (bias / gamma gradients), and no re-reads beyond what
// L2 serves.  GEMM-shaped gradients (dgrad / wgrad) reuse stllm_gemm on operands re-laid-out by stllm_transpose.
//
// Status: compiled for gfx950 and covered by the contract tests of tests/_cpu_backend.py through the host graph
// (tests/test_backward_cpu.py); the on-device parity tests live in tests/test_train_gpu.py.
#include "common.h"

namespace {

#define STLLM_DISPATCH_DTYPE(dtype, what, CALL)                                  \
  switch (dtype) {                                                               \
    case STLLM_BF16: { using T = bf16_t; CALL; } break;                          \
    case STLLM_F16: { using T = f16_t; CALL; } break;                            \
    case STLLM_F32: { using T = float; CALL; } break;                            \
    default: stllm_set_error("%s: bad dtype %d", what, dtype); return STLLM_ERR_BAD_DTYPE; \
  }

// ---- transpose -----------------------------------------------------------------------------------------------------------
// dst[c, r] = src[r, c] (0 for r >= R), 64x64 tiles through LDS (+1 padding column: conflict-free both ways).
template <typename U>
__global__ __launch_bounds__(256) void transpose_kernel(const U* __restrict__ src, int64_t lds_, U* __restrict__ dst, int64_t ldd,
                                                        int R, int C, int Rp) {
  __shared__ U tile[64][65];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
#pragma unroll 4
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < R && c < C) ? src[(int64_t)r * lds_ + c] : (U)0;
  }
  __syncthreads();
#pragma unroll 4
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + tx;
    if (c < C && r < Rp) dst[(int64_t)c * ldd + r] = tile[tx][i];
  }
}

// ---- norm backward: dx (+ per-row statistics), then the column reductions ------------------------------------------------
// One wave per row.  RMS: xh = x * r, r = rsqrt(mean(x^2) + eps);  LN: xh = (x - mu) * r.
// dx (+)= r * (g - [mean(g)] - xh * mean(g * xh)),  g = gamma * dy.   stats[row] = (mu, r) for the column kernel.
template <typename TY, bool RMS>
__global__ __launch_bounds__(256) void norm_bwd_dx_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ gamma,
                                                          float eps, const void* __restrict__ dy, int64_t lddy, float* __restrict__ dx,
                                                          int64_t lddx, int accumulate, float* __restrict__ stats, int M, int D) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int nvec = D >> 2;
  const float4* xr = reinterpret_cast<const float4*>(x + (int64_t)row * ldx);
  const float4* g4 = reinterpret_cast<const float4*>(gamma);
  float mu = 0.0f, s = 0.0f;
  if constexpr (!RMS) {
    for (int c = lane; c < nvec; c += 64) { const float4 v = xr[c]; s += v.x + v.y + v.z + v.w; }
    mu = wave_sum(s) / (float)D;
    s = 0.0f;
  }
  for (int c = lane; c < nvec; c += 64) {
    const float4 v = xr[c];
    const float a = v.x - mu, b = v.y - mu, cc = v.z - mu, d = v.w - mu;
    s += a * a + b * b + cc * cc + d * d;
  }
  const float r = rsqrtf(wave_sum(s) / (float)D + eps);
  float c0 = 0.0f, c1 = 0.0f;
  for (int c = lane; c < nvec; c += 64) {
    const float4 v = xr[c], g = g4[c];
    float y[4];
    load4<TY>(dy, (int64_t)row * lddy + 4 * c, y);
    const float gx = g.x * y[0], gy = g.y * y[1], gz = g.z * y[2], gw = g.w * y[3];
    c0 += gx + gy + gz + gw;
    c1 += gx * (v.x - mu) + gy * (v.y - mu) + gz * (v.z - mu) + gw * (v.w - mu);
  }
  c0 = RMS ? 0.0f : wave_sum(c0) / (float)D;
  c1 = wave_sum(c1) * r / (float)D;                  // mean(g * xh)
  float4* dxr = reinterpret_cast<float4*>(dx + (int64_t)row * lddx);
  for (int c = lane; c < nvec; c += 64) {
    const float4 v = xr[c], g = g4[c];
    float y[4];
    load4<TY>(dy, (int64_t)row * lddy + 4 * c, y);
    float4 o;
    o.x = r * (g.x * y[0] - c0 - (v.x - mu) * r * c1);
    o.y = r * (g.y * y[1] - c0 - (v.y - mu) * r * c1);
    o.z = r * (g.z * y[2] - c0 - (v.z - mu) * r * c1);
    o.w = r * (g.w * y[3] - c0 - (v.w - mu) * r * c1);
    if (accumulate) { const float4 p = dxr[c]; o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w; }
    dxr[c] = o;
  }
  if (lane == 0) { stats[2 * row] = mu; stats[2 * row + 1] = r; }
}

// Column partial sums over a slab of rows: block (64 columns x 4 row lanes), grid (ceil(N/64), kColBlocks).
// MODE 0: sum_r y[r,c];  MODE 1: also sum_r y[r,c] * xh[r,c] with xh from x and stats (norm gamma / beta gradients).
constexpr int kColBlocks = 64;
template <typename TY, int MODE>
__global__ __launch_bounds__(256) void col_partial_kernel(const void* __restrict__ y, int64_t ldy, const float* __restrict__ x, int64_t ldx,
                                                          const float* __restrict__ stats, float* __restrict__ part_a,
                                                          float* __restrict__ part_b, int M, int N) {
  __shared__ float red[2][4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx;
  const int rows_per = (M + kColBlocks - 1) / kColBlocks;
  const int r0 = blockIdx.y * rows_per, r1 = min(M, r0 + rows_per);
  float sa = 0.0f, sb = 0.0f;
  if (c < N) {
    for (int r = r0 + ty; r < r1; r += 4) {
      const float v = load_elem<TY>(y, (int64_t)r * ldy + c);
      sa += v;
      if constexpr (MODE == 1) sb += v * (x[(int64_t)r * ldx + c] - stats[2 * r]) * stats[2 * r + 1];
    }
  }
  red[0][ty][tx] = sa;
  red[1][ty][tx] = sb;
  __syncthreads();
  if (ty == 0 && c < N) {
    part_a[(int64_t)blockIdx.y * N + c] = red[0][0][tx] + red[0][1][tx] + red[0][2][tx] + red[0][3][tx];
    if constexpr (MODE == 1) part_b[(int64_t)blockIdx.y * N + c] = red[1][0][tx] + red[1][1][tx] + red[1][2][tx] + red[1][3][tx];
  }
}
__global__ __launch_bounds__(256) void col_final_kernel(const float* __restrict__ part, float* __restrict__ out, int N) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= N) return;
  float s = 0.0f;
  for (int b = 0; b < kColBlocks; ++b) s += part[(int64_t)b * N + c];   // fixed order: bit-reproducible
  out[c] = s;
}

// ---- SwiGLU on the packed [32 gate | 32 up] groups ---------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void swiglu_kernel(const void* __restrict__ gu, int64_t ldgu, void* __restrict__ out, int64_t ldo,
                                                     int M, int I) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int per_row = I >> 3;
  if (t >= (int64_t)M * per_row) return;
  const int row = (int)(t / per_row), j = (int)(t % per_row) * 8;          // 8 outputs inside one 32-group
  const int64_t src = (int64_t)row * ldgu + (j >> 5) * 64 + (j & 31);
  float g[8], u[8], o[8];
  load8<T>(gu, src, g);
  load8<T>(gu, src + 32, u);
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = silu_f(g[e]) * u[e];
  store8<T>(out, (int64_t)row * ldo + j, o);
}
template <typename T>
__global__ __launch_bounds__(256) void swiglu_bwd_kernel(const void* __restrict__ gu, int64_t ldgu, const void* __restrict__ dg, int64_t lddg,
                                                         void* __restrict__ out, int64_t ldo, int M, int I) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int per_row = I >> 3;
  if (t >= (int64_t)M * per_row) return;
  const int row = (int)(t / per_row), j = (int)(t % per_row) * 8;
  const int col = (j >> 5) * 64 + (j & 31);
  float g[8], u[8], d[8], og[8], ou[8];
  load8<T>(gu, (int64_t)row * ldgu + col, g);
  load8<T>(gu, (int64_t)row * ldgu + col + 32, u);
  load8<T>(dg, (int64_t)row * lddg + j, d);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float sg = __frcp_rn(1.0f + __expf(-g[e]));
    og[e] = d[e] * u[e] * sg * (1.0f + g[e] * (1.0f - sg));
    ou[e] = d[e] * g[e] * sg;
  }
  store8<T>(out, (int64_t)row * ldo + col, og);
  store8<T>(out, (int64_t)row * ldo + col + 32, ou);
}

// ---- RoPE^T in the packed head layout (forward: y1 = x1 c - x2 s, y2 = x2 c + x1 s) -------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void rope_bwd_kernel(void* __restrict__ d, int64_t ld, const float* __restrict__ cos_t,
                                                       const float* __restrict__ sin_t, int M, int rope_seq, int groups) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int per_row = groups * 4;                                         // 4 threads x 8 pairs per 64-column group
  if (t >= (int64_t)M * per_row) return;
  const int row = (int)(t / per_row), q = (int)(t % per_row);
  const int grp = q >> 2, j = (q & 3) * 8;
  const int64_t base = (int64_t)row * ld + grp * 64 + j;
  const int pos = row % rope_seq, fi = (grp & 1) * 32 + j;
  float y1[8], y2[8], x1[8], x2[8];
  load8<T>(d, base, y1);
  load8<T>(d, base + 32, y2);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float c = cos_t[pos * 64 + fi + e], s = sin_t[pos * 64 + fi + e];
    x1[e] = y1[e] * c + y2[e] * s;
    x2[e] = y2[e] * c - y1[e] * s;
  }
  store8<T>(d, base, x1);
  store8<T>(d, base + 32, x2);
}

// ---- cross-entropy backward: (softmax - onehot) * scale, one workgroup per row ------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* __restrict__ logits, int64_t ldl, const int32_t* __restrict__ labels,
                                                     float scale, void* __restrict__ out, int64_t ldo, int V, int Vp) {
  __shared__ float red[4];
  const int row = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int lab = labels[row];
  const int64_t ob = (int64_t)row * ldo;
  if (lab < 0) {
    for (int c = threadIdx.x; c < Vp; c += 256) store_elem<T>(out, ob + c, 0.0f);
    return;
  }
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
  const float inv = scale / (red[0] + red[1] + red[2] + red[3]);
  for (int c = threadIdx.x; c < Vp; c += 256) {
    float v = 0.0f;
    if (c < V) v = expf(x[c] - mx) * inv - (c == lab ? scale : 0.0f);
    store_elem<T>(out, ob + c, v);
  }
}

// ---- scatter-add of rows (gather_rows^T), fp32 atomics -----------------------------------------------------------------
__global__ __launch_bounds__(256) void scatter_add_kernel(const float* __restrict__ src, int64_t lds_, const int32_t* __restrict__ idx,
                                                          float* __restrict__ dst_a, int64_t lda, float* __restrict__ dst_b, int64_t ldb,
                                                          int D, float scale) {
  const int i = blockIdx.x;
  const int j = idx[i];
  float* d = j >= 0 ? dst_a + (int64_t)j * lda : dst_b + (int64_t)(-j - 1) * ldb;
  const float* s = src + (int64_t)i * lds_;
  for (int c = threadIdx.x; c < D; c += 256) atomicAdd(d + c, s[c] * scale);
}

// ---- d(2 - 2 cos(a, b)) / da, one wave per row --------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cosine_bwd_kernel(const float* __restrict__ a, int64_t lda, const int32_t* __restrict__ idx_a,
                                                         const float* __restrict__ b, int64_t ldb, const int32_t* __restrict__ idx_b,
                                                         float scale, float* __restrict__ out, int64_t ldo, int n_rows, int D) {
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
  const float na = sqrtf(aa), nb = sqrtf(bb);
  const float cs = ab / (na * nb);
  const float k = -2.0f * scale / na;                                    // d/da = k * (b/|b| - cos * a/|a|)
  float4* o = reinterpret_cast<float4*>(out + (int64_t)row * ldo);
  for (int c = lane; c < (D >> 2); c += 64) {
    const float4 u = ar[c], v = br[c];
    float4 r;
    r.x = k * (v.x / nb - cs * u.x / na); r.y = k * (v.y / nb - cs * u.y / na);
    r.z = k * (v.z / nb - cs * u.z / na); r.w = k * (v.w / nb - cs * u.w / na);
    o[c] = r;
  }
}

// ---- exact-erf GELU on raw pre-activations (training keeps them: the forward's fused GELU epilogue does not) ------------------
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void gelu_kernel(const void* __restrict__ x, int64_t ldx, const void* __restrict__ dy, int64_t lddy,
                                                   void* __restrict__ out, int64_t ldo, int M, int N) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int per_row = N >> 3;
  if (t >= (int64_t)M * per_row) return;
  const int row = (int)(t / per_row), j = (int)(t % per_row) * 8;
  float a[8], g[8];
  load8<T>(x, (int64_t)row * ldx + j, a);
  if constexpr (BWD) load8<T>(dy, (int64_t)row * lddy + j, g);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    if constexpr (BWD) {
      const float cdf = 0.5f * (1.0f + erf_as(a[e] * 0.70710678118654752f));
      a[e] = g[e] * (cdf + a[e] * 0.3989422804014327f * __expf(-0.5f * a[e] * a[e]));
    } else {
      a[e] = gelu_erf(a[e]);
    }
  }
  store8<T>(out, (int64_t)row * ldo + j, a);
}

// ---- per-row scaling by a per-group factor (stochastic depth: x * keep_mask / keep_prob per sample; its own transpose) ----------
template <typename T>
__global__ __launch_bounds__(256) void scale_rows_kernel(void* __restrict__ x, int64_t ldx, const float* __restrict__ scale,
                                                         const int32_t* __restrict__ idx, int rows_per_group, int M, int N) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int per_row = N >> 3;
  if (t >= (int64_t)M * per_row) return;
  const int row = (int)(t / per_row), j = (int)(t % per_row) * 8;
  const float s = scale[idx ? idx[row] : row / rows_per_group];
  float a[8];
  load8<T>(x, (int64_t)row * ldx + j, a);
#pragma unroll
  for (int e = 0; e < 8; ++e) a[e] *= s;
  store8<T>(x, (int64_t)row * ldx + j, a);
}

// ---- small elementwise pieces -------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void relu_bwd_kernel(const void* __restrict__ dy, int64_t lddy, const void* __restrict__ y, int64_t ldy,
                                                       void* __restrict__ out, int64_t ldo, int M, int N) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int per_row = N >> 3;
  if (t >= (int64_t)M * per_row) return;
  const int row = (int)(t / per_row), j = (int)(t % per_row) * 8;
  float a[8], b[8];
  load8<T>(dy, (int64_t)row * lddy + j, a);
  load8<T>(y, (int64_t)row * ldy + j, b);
#pragma unroll
  for (int e = 0; e < 8; ++e) a[e] = b[e] > 0.0f ? a[e] : 0.0f;
  store8<T>(out, (int64_t)row * ldo + j, a);
}

__global__ __launch_bounds__(256) void bcast_add_t_kernel(float* __restrict__ dst, const float* __restrict__ src, int T_, int64_t J4,
                                                          int64_t total4, float scale) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
    const int64_t b = i / (T_ * J4), j = i % J4;
    const float4 s = reinterpret_cast<const float4*>(src)[b * J4 + j];
    float4 d = reinterpret_cast<float4*>(dst)[i];
    d.x += scale * s.x; d.y += scale * s.y; d.z += scale * s.z; d.w += scale * s.w;
    reinterpret_cast<float4*>(dst)[i] = d;
  }
}

// torch.optim.AdamW (decoupled weight decay, bias-corrected moments): one pass over p, g, m, v (16 B read + 12 B written per
// parameter, + 2 B for the optional compute-dtype copy): HBM roofline = 28 n bytes.
template <int P16>   // 0: none, 1: bf16, 2: f16
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, uint16_t* __restrict__ p16, int64_t n, float decay, float b1,
                                                    float b2, float eps, float step_size, float inv_sqrt_bc2, float gscale) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float gi = g[i] * gscale;
    float pi = p[i] * decay;
    const float mi = b1 * m[i] + (1.0f - b1) * gi;
    const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    pi -= step_size * mi / (sqrtf(vi) * inv_sqrt_bc2 + eps);
    p[i] = pi; m[i] = mi; v[i] = vi;
    if constexpr (P16 == 1) p16[i] = Elem<bf16_t>::pack(pi);
    if constexpr (P16 == 2) p16[i] = Elem<f16_t>::pack(pi);
  }
}

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ out) {
  __shared__ float red[4];
  float s = 0.0f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) s += x[i] * x[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }
inline int eb_of(int dtype) { return dtype == STLLM_F32 ? 4 : 2; }
inline bool vec_ok(const void* p, int64_t ld, int dtype) { return aligned16(p) && (ld * eb_of(dtype)) % 16 == 0; }

template <typename TY, bool RMS>
int norm_bwd_launch(const float* x, int64_t ldx, const float* gamma, float eps, const void* dy, int64_t lddy, float* dx, int64_t lddx,
                    int accumulate, float* dgamma, float* dbeta, float* ws, int M, int D, hipStream_t st) {
  float* stats = ws;
  float* part_a = ws + 2 * (int64_t)M;
  float* part_b = part_a + (int64_t)kColBlocks * D;
  hipLaunchKernelGGL((norm_bwd_dx_kernel<TY, RMS>), dim3((M + 3) / 4), dim3(256), 0, st, x, ldx, gamma, eps, dy, lddy, dx, lddx, accumulate,
                     stats, M, D);
  hipLaunchKernelGGL((col_partial_kernel<TY, 1>), dim3((D + 63) / 64, kColBlocks), dim3(256), 0, st, dy, lddy, x, ldx, stats, part_a, part_b,
                     M, D);
  hipLaunchKernelGGL(col_final_kernel, dim3((D + 255) / 256), dim3(256), 0, st, part_b, dgamma, D);
  if (dbeta) hipLaunchKernelGGL(col_final_kernel, dim3((D + 255) / 256), dim3(256), 0, st, part_a, dbeta, D);
  return STLLM_OK;
}

template <bool RMS>
int norm_bwd_entry(int dy_dtype, const float* x, int64_t ldx, const float* gamma, float eps, const void* dy, int64_t lddy, float* dx,
                   int64_t lddx, int accumulate, float* dgamma, float* dbeta, void* ws, int64_t ws_bytes, int M, int D, void* stream) {
  const char* nm = RMS ? "stllm_rmsnorm_bwd" : "stllm_layernorm_bwd";
  STLLM_CHECK_ARG(M > 0 && D > 0 && D % 4 == 0, "%s: bad M=%d D=%d", nm, M, D);
  STLLM_CHECK_ARG(x && gamma && dy && dx && dgamma && (RMS || dbeta), "%s: null argument", nm);
  STLLM_CHECK_ARG(aligned16(x) && ldx % 4 == 0 && aligned16(dx) && lddx % 4 == 0 && aligned16(gamma), "%s: x / dx / gamma misaligned", nm);
  STLLM_CHECK_ARG((reinterpret_cast<uintptr_t>(dy) & 7) == 0 && lddy % 4 == 0, "%s: dy misaligned", nm);
  STLLM_CHECK_ARG(ws && ws_bytes >= stllm_norm_bwd_workspace_bytes(M, D), "%s: workspace too small", nm);
  STLLM_DISPATCH_DTYPE(dy_dtype, nm, (norm_bwd_launch<T, RMS>(x, ldx, gamma, eps, dy, lddy, dx, lddx, accumulate, dgamma, RMS ? nullptr : dbeta,
                                                              reinterpret_cast<float*>(ws), M, D, S(stream))));
  STLLM_CHECK_LAUNCH(nm);
  return STLLM_OK;
}

}  // namespace

// =========================================================================================================================
extern "C" int stllm_transpose(int dtype, const void* src, int64_t lds_, void* dst, int64_t ldd, int rows, int cols, int rows_padded,
                               void* stream) {
  STLLM_CHECK_ARG(src && dst && rows > 0 && cols > 0 && rows_padded >= rows && ldd >= rows_padded && lds_ >= cols,
                  "stllm_transpose: bad shape rows=%d cols=%d rows_padded=%d", rows, cols, rows_padded);
  const dim3 grid((cols + 63) / 64, (rows_padded + 63) / 64), block(256);
  if (dtype == STLLM_F32)
    hipLaunchKernelGGL(transpose_kernel<uint32_t>, grid, block, 0, S(stream), reinterpret_cast<const uint32_t*>(src), lds_,
                       reinterpret_cast<uint32_t*>(dst), ldd, rows, cols, rows_padded);
  else if (dtype == STLLM_BF16 || dtype == STLLM_F16)
    hipLaunchKernelGGL(transpose_kernel<uint16_t>, grid, block, 0, S(stream), reinterpret_cast<const uint16_t*>(src), lds_,
                       reinterpret_cast<uint16_t*>(dst), ldd, rows, cols, rows_padded);
  else { stllm_set_error("stllm_transpose: bad dtype %d", dtype); return STLLM_ERR_BAD_DTYPE; }
  STLLM_CHECK_LAUNCH("stllm_transpose");
  return STLLM_OK;
}

extern "C" int64_t stllm_norm_bwd_workspace_bytes(int rows, int cols) {
  return ((int64_t)2 * rows + (int64_t)2 * kColBlocks * cols) * 4;
}

extern "C" int stllm_rmsnorm_bwd(int dy_dtype, const float* x, int64_t ldx, const float* gamma, float eps, const void* dy, int64_t lddy,
                                 float* dx, int64_t lddx, int accumulate, float* dgamma, void* workspace, int64_t workspace_bytes, int rows,
                                 int cols, void* stream) {
  return norm_bwd_entry<true>(dy_dtype, x, ldx, gamma, eps, dy, lddy, dx, lddx, accumulate, dgamma, nullptr, workspace, workspace_bytes, rows,
                              cols, stream);
}

extern "C" int stllm_layernorm_bwd(int dy_dtype, const float* x, int64_t ldx, const float* gamma, float eps, const void* dy, int64_t lddy,
                                   float* dx, int64_t lddx, int accumulate, float* dgamma, float* dbeta, void* workspace,
                                   int64_t workspace_bytes, int rows, int cols, void* stream) {
  return norm_bwd_entry<false>(dy_dtype, x, ldx, gamma, eps, dy, lddy, dx, lddx, accumulate, dgamma, dbeta, workspace, workspace_bytes, rows,
                               cols, stream);
}

extern "C" int stllm_swiglu(int dtype, const void* gu, int64_t ldgu, void* out, int64_t ldo, int rows, int inter, void* stream) {
  STLLM_CHECK_ARG(gu && out && rows > 0 && inter > 0 && inter % 32 == 0, "stllm_swiglu: bad shape rows=%d inter=%d", rows, inter);
  STLLM_CHECK_ARG(vec_ok(gu, ldgu, dtype) && vec_ok(out, ldo, dtype), "stllm_swiglu: rows must be 16-byte aligned");
  const int64_t n = (int64_t)rows * (inter / 8);
  STLLM_DISPATCH_DTYPE(dtype, "stllm_swiglu",
                       hipLaunchKernelGGL(swiglu_kernel<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, S(stream), gu, ldgu, out, ldo, rows, inter));
  STLLM_CHECK_LAUNCH("stllm_swiglu");
  return STLLM_OK;
}

extern "C" int stllm_swiglu_bwd(int dtype, const void* gu, int64_t ldgu, const void* dg, int64_t lddg, void* dgu, int64_t lddgu, int rows,
                                int inter, void* stream) {
  STLLM_CHECK_ARG(gu && dg && dgu && rows > 0 && inter > 0 && inter % 32 == 0, "stllm_swiglu_bwd: bad shape rows=%d inter=%d", rows, inter);
  STLLM_CHECK_ARG(vec_ok(gu, ldgu, dtype) && vec_ok(dg, lddg, dtype) && vec_ok(dgu, lddgu, dtype), "stllm_swiglu_bwd: rows must be 16-byte aligned");
  const int64_t n = (int64_t)rows * (inter / 8);
  STLLM_DISPATCH_DTYPE(dtype, "stllm_swiglu_bwd",
                       hipLaunchKernelGGL(swiglu_bwd_kernel<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, S(stream), gu, ldgu, dg, lddg, dgu,
                                          lddgu, rows, inter));
  STLLM_CHECK_LAUNCH("stllm_swiglu_bwd");
  return STLLM_OK;
}

extern "C" int stllm_rope_bwd(int dtype, void* d, int64_t ld, const float* cos_t, const float* sin_t, int rows, int cols, int rope_seq,
                              int rope_cols, void* stream) {
  STLLM_CHECK_ARG(d && cos_t && sin_t && rows > 0 && rope_seq > 0 && rope_cols > 0 && rope_cols % 128 == 0 && rope_cols <= cols,
                  "stllm_rope_bwd: bad shape rows=%d cols=%d rope_cols=%d", rows, cols, rope_cols);
  STLLM_CHECK_ARG(vec_ok(d, ld, dtype), "stllm_rope_bwd: rows must be 16-byte aligned");
  const int groups = rope_cols / 64;
  const int64_t n = (int64_t)rows * groups * 4;
  STLLM_DISPATCH_DTYPE(dtype, "stllm_rope_bwd",
                       hipLaunchKernelGGL(rope_bwd_kernel<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, S(stream), d, ld, cos_t, sin_t, rows,
                                          rope_seq, groups));
  STLLM_CHECK_LAUNCH("stllm_rope_bwd");
  return STLLM_OK;
}

extern "C" int stllm_cross_entropy_bwd(int dtype, const float* logits, int64_t ldl, const int32_t* labels, float scale, void* dlogits,
                                       int64_t lddl, int rows, int vocab, int cols_padded, void* stream) {
  STLLM_CHECK_ARG(logits && labels && dlogits && rows > 0 && vocab > 0 && cols_padded >= vocab && lddl >= cols_padded && ldl >= vocab,
                  "stllm_cross_entropy_bwd: bad shape rows=%d vocab=%d padded=%d", rows, vocab, cols_padded);
  STLLM_DISPATCH_DTYPE(dtype, "stllm_cross_entropy_bwd",
                       hipLaunchKernelGGL(ce_bwd_kernel<T>, dim3(rows), dim3(256), 0, S(stream), logits, ldl, labels, scale, dlogits, lddl, vocab,
                                          cols_padded));
  STLLM_CHECK_LAUNCH("stllm_cross_entropy_bwd");
  return STLLM_OK;
}

extern "C" int stllm_scatter_add_rows(const float* src, int64_t lds_, const int32_t* idx, float* dst_a, int64_t lda, float* dst_b, int64_t ldb,
                                      int n, int cols, float scale, void* stream) {
  STLLM_CHECK_ARG(src && idx && dst_a && n > 0 && cols > 0, "stllm_scatter_add_rows: bad args");
  hipLaunchKernelGGL(scatter_add_kernel, dim3(n), dim3(256), 0, S(stream), src, lds_, idx, dst_a, lda, dst_b, ldb, cols, scale);
  STLLM_CHECK_LAUNCH("stllm_scatter_add_rows");
  return STLLM_OK;
}

extern "C" int stllm_cosine_rows_bwd(const float* a, int64_t lda, const int32_t* idx_a, const float* b, int64_t ldb, const int32_t* idx_b,
                                     float scale, float* da, int64_t ldda, int n_rows, int D, void* stream) {
  STLLM_CHECK_ARG(a && b && da && n_rows > 0 && D > 0 && D % 4 == 0, "stllm_cosine_rows_bwd: bad args");
  STLLM_CHECK_ARG(lda % 4 == 0 && ldb % 4 == 0 && ldda % 4 == 0 && aligned16(a) && aligned16(b) && aligned16(da), "stllm_cosine_rows_bwd: misaligned");
  hipLaunchKernelGGL(cosine_bwd_kernel, dim3((n_rows + 3) / 4), dim3(256), 0, S(stream), a, lda, idx_a, b, ldb, idx_b, scale, da, ldda, n_rows, D);
  STLLM_CHECK_LAUNCH("stllm_cosine_rows_bwd");
  return STLLM_OK;
}

extern "C" int stllm_colsum(int dtype, const void* x, int64_t ldx, float* out, int rows, int cols, void* workspace, int64_t workspace_bytes,
                            void* stream) {
  STLLM_CHECK_ARG(x && out && rows > 0 && cols > 0, "stllm_colsum: bad args");
  STLLM_CHECK_ARG(workspace && workspace_bytes >= stllm_norm_bwd_workspace_bytes(rows, cols), "stllm_colsum: workspace too small");
  float* part = reinterpret_cast<float*>(workspace);
  STLLM_DISPATCH_DTYPE(dtype, "stllm_colsum",
                       hipLaunchKernelGGL((col_partial_kernel<T, 0>), dim3((cols + 63) / 64, kColBlocks), dim3(256), 0, S(stream), x, ldx,
                                          (const float*)nullptr, (int64_t)0, (const float*)nullptr, part, (float*)nullptr, rows, cols));
  hipLaunchKernelGGL(col_final_kernel, dim3((cols + 255) / 256), dim3(256), 0, S(stream), part, out, cols);
  STLLM_CHECK_LAUNCH("stllm_colsum");
  return STLLM_OK;
}

extern "C" int stllm_relu_bwd(int dtype, const void* dy, int64_t lddy, const void* y, int64_t ldy, void* dx, int64_t lddx, int rows, int cols,
                              void* stream) {
  STLLM_CHECK_ARG(dy && y && dx && rows > 0 && cols > 0 && cols % 8 == 0, "stllm_relu_bwd: bad shape rows=%d cols=%d", rows, cols);
  STLLM_CHECK_ARG(vec_ok(dy, lddy, dtype) && vec_ok(y, ldy, dtype) && vec_ok(dx, lddx, dtype), "stllm_relu_bwd: rows must be 16-byte aligned");
  const int64_t n = (int64_t)rows * (cols / 8);
  STLLM_DISPATCH_DTYPE(dtype, "stllm_relu_bwd",
                       hipLaunchKernelGGL(relu_bwd_kernel<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, S(stream), dy, lddy, y, ldy, dx, lddx,
                                          rows, cols));
  STLLM_CHECK_LAUNCH("stllm_relu_bwd");
  return STLLM_OK;
}

extern "C" int stllm_gelu(int dtype, const void* x, int64_t ldx, void* out, int64_t ldo, int rows, int cols, void* stream) {
  STLLM_CHECK_ARG(x && out && rows > 0 && cols > 0 && cols % 8 == 0, "stllm_gelu: bad shape rows=%d cols=%d", rows, cols);
  STLLM_CHECK_ARG(vec_ok(x, ldx, dtype) && vec_ok(out, ldo, dtype), "stllm_gelu: rows must be 16-byte aligned");
  const int64_t n = (int64_t)rows * (cols / 8);
  STLLM_DISPATCH_DTYPE(dtype, "stllm_gelu",
                       hipLaunchKernelGGL((gelu_kernel<T, false>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, S(stream), x, ldx, (const void*)nullptr,
                                          (int64_t)0, out, ldo, rows, cols));
  STLLM_CHECK_LAUNCH("stllm_gelu");
  return STLLM_OK;
}

extern "C" int stllm_gelu_bwd(int dtype, const void* x, int64_t ldx, const void* dy, int64_t lddy, void* dx, int64_t lddx, int rows, int cols,
                              void* stream) {
  STLLM_CHECK_ARG(x && dy && dx && rows > 0 && cols > 0 && cols % 8 == 0, "stllm_gelu_bwd: bad shape rows=%d cols=%d", rows, cols);
  STLLM_CHECK_ARG(vec_ok(x, ldx, dtype) && vec_ok(dy, lddy, dtype) && vec_ok(dx, lddx, dtype), "stllm_gelu_bwd: rows must be 16-byte aligned");
  const int64_t n = (int64_t)rows * (cols / 8);
  STLLM_DISPATCH_DTYPE(dtype, "stllm_gelu_bwd",
                       hipLaunchKernelGGL((gelu_kernel<T, true>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, S(stream), x, ldx, dy, lddy, dx, lddx,
                                          rows, cols));
  STLLM_CHECK_LAUNCH("stllm_gelu_bwd");
  return STLLM_OK;
}

extern "C" int stllm_scale_rows(int dtype, void* x, int64_t ldx, const float* scale, const int32_t* idx, int rows_per_group, int rows, int cols,
                                void* stream) {
  STLLM_CHECK_ARG(x && scale && rows > 0 && cols > 0 && cols % 8 == 0 && (idx || rows_per_group > 0), "stllm_scale_rows: bad args");
  STLLM_CHECK_ARG(vec_ok(x, ldx, dtype), "stllm_scale_rows: rows must be 16-byte aligned");
  const int64_t n = (int64_t)rows * (cols / 8);
  STLLM_DISPATCH_DTYPE(dtype, "stllm_scale_rows",
                       hipLaunchKernelGGL(scale_rows_kernel<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, S(stream), x, ldx, scale, idx,
                                          rows_per_group, rows, cols));
  STLLM_CHECK_LAUNCH("stllm_scale_rows");
  return STLLM_OK;
}

extern "C" int stllm_bcast_add_t(float* dst, const float* src, int B, int T_, int64_t J, float scale, void* stream) {
  STLLM_CHECK_ARG(dst && src && B > 0 && T_ > 0 && J > 0 && J % 4 == 0 && aligned16(dst) && aligned16(src), "stllm_bcast_add_t: bad args");
  const int64_t total4 = (int64_t)B * T_ * (J / 4);
  const unsigned blocks = (unsigned)((total4 + 255) / 256 > 65536 ? 65536 : (total4 + 255) / 256);
  hipLaunchKernelGGL(bcast_add_t_kernel, dim3(blocks), dim3(256), 0, S(stream), dst, src, T_, J / 4, total4, scale);
  STLLM_CHECK_LAUNCH("stllm_bcast_add_t");
  return STLLM_OK;
}

extern "C" int stllm_adamw(float* p, const float* g, float* m, float* v, void* p16, int p16_dtype, int64_t n, float lr, float beta1, float beta2,
                           float eps, float weight_decay, int step, float grad_scale, void* stream) {
  STLLM_CHECK_ARG(p && g && m && v && n > 0 && step >= 1, "stllm_adamw: bad args (n=%lld step=%d)", (long long)n, step);
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  const float step_size = (float)((double)lr / bc1), inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2)), decay = 1.0f - lr * weight_decay;
  const unsigned blocks = (unsigned)((n + 255) / 256 > 16384 ? 16384 : (n + 255) / 256);
  uint16_t* h = reinterpret_cast<uint16_t*>(p16);
#define STLLM_ADAMW(P) hipLaunchKernelGGL(adamw_kernel<P>, dim3(blocks), dim3(256), 0, S(stream), p, g, m, v, h, n, decay, beta1, beta2, eps, step_size, inv_sqrt_bc2, grad_scale)
  if (!p16) STLLM_ADAMW(0);
  else if (p16_dtype == STLLM_BF16) STLLM_ADAMW(1);
  else if (p16_dtype == STLLM_F16) STLLM_ADAMW(2);
  else { stllm_set_error("stllm_adamw: bad p16 dtype %d", p16_dtype); return STLLM_ERR_BAD_DTYPE; }
#undef STLLM_ADAMW
  STLLM_CHECK_LAUNCH("stllm_adamw");
  return STLLM_OK;
}

extern "C" int stllm_sumsq(const float* x, int64_t n, float* out, void* stream) {
  STLLM_CHECK_ARG(x && out && n > 0, "stllm_sumsq: bad args");
  const unsigned blocks = (unsigned)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
  hipLaunchKernelGGL(sumsq_kernel, dim3(blocks), dim3(256), 0, S(stream), x, n, out);
  STLLM_CHECK_LAUNCH("stllm_sumsq");
  return STLLM_OK;
}
