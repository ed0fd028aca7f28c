// "bf16 x 3" GEMM operands (stllm_dtype STLLM_BF16X3): fp32 accuracy class on the 16-bit matrix cores.
//
//   x = hi + lo + O(2^-17 |x|),  hi = bf16(x), lo = bf16(x - hi)            (both round-to-nearest-even; x - hi is exact in fp32)
//   A W^T  ~=  A_hi W_hi^T + A_hi W_lo^T + A_lo W_hi^T                      (the lo x lo term, 2^-18 relative, is dropped)
//
// The three products are ONE bf16 GEMM with K' = 3 K: the A operand is laid out [hi | hi | lo] along K (stllm_split3_rows, side 0), the
// weight [hi | lo | hi] (side 1, packed once).  No new matrix-core kernel: the phased / one-wave GEMMs of this library run it, fp32
// accumulate, fp32 output — at 3x the bf16 FLOPs instead of the 16x of the exact v_mfma_f32_32x32x2_f32 path, and with longer K loops
// (prologue / epilogue / exchange amortised over three times the work).  Activations stay fp32 between the GEMMs (norms, softmax,
// GELU / SiLU / RoPE are the fp32 kernels of the verify mode).  This is the "split-bf16 x 3" mode of SURVEY.md §7 #1.
#include "common.h"
#include "../../include/stllm_hip.h"

namespace {

__device__ __forceinline__ void split1(float x, uint16_t& hi, uint16_t& lo) {
  hi = f32_to_bf16_bits(x);
  lo = f32_to_bf16_bits(x - bf16_bits_to_f32(hi));
}

// one thread = four consecutive k of one row: a float4 load, three 8-byte stores
template <int SIDE>
__global__ __launch_bounds__(256) void split3_rows_kernel(const float* __restrict__ x, int64_t ldx, int rpb, int64_t bs,
                                                          uint16_t* __restrict__ out, int64_t ldo, int M, int K) {
  const int k4 = K >> 2;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)M * k4) return;
  const int row = (int)(idx / k4), c = (int)(idx - (int64_t)row * k4);
  int64_t off = (int64_t)row * ldx;
  if (rpb > 0) { const int b = row / rpb; off = (int64_t)b * bs + (int64_t)(row - b * rpb) * ldx; }
  const float4 v = reinterpret_cast<const float4*>(x + off)[c];
  uint16_t h[4], l[4];
  split1(v.x, h[0], l[0]); split1(v.y, h[1], l[1]); split1(v.z, h[2], l[2]); split1(v.w, h[3], l[3]);
  uint2 hv, lv;
  hv.x = h[0] | ((uint32_t)h[1] << 16); hv.y = h[2] | ((uint32_t)h[3] << 16);
  lv.x = l[0] | ((uint32_t)l[1] << 16); lv.y = l[2] | ((uint32_t)l[3] << 16);
  uint16_t* o = out + (int64_t)row * ldo + 4 * c;
  *reinterpret_cast<uint2*>(o) = hv;
  *reinterpret_cast<uint2*>(o + K) = SIDE == 0 ? hv : lv;
  *reinterpret_cast<uint2*>(o + 2 * K) = SIDE == 0 ? lv : hv;
}

__device__ __forceinline__ int64_t row_off(int row, int64_t ld, int rpb, int64_t bs) {
  if (rpb > 0) { const int b = row / rpb; return (int64_t)b * bs + (int64_t)(row - b * rpb) * ld; }
  return (int64_t)row * ld;
}

// fp32 post-epilogues of the bf16x3 GEMM (the inner GEMM stores acc + bias as fp32)
//   ACT    : x <- gelu_erf(x) | relu(x), in place                                   (eva_vit.py:45 nn.GELU, Qformer.py:359)
//   SWIGLU : out[m, 32 g + c] = silu(t[m, 64 g + c]) * t[m, 64 g + 32 + c]          (packed [32 gate | 32 up] columns, modeling_llama_mem.py:143-144)
//   ROPE   : rotate-half on the packed [x_lo | x_hi] halves of every 64-column group below rope_cols, in place (modeling_llama_mem.py:113-127)
__device__ __forceinline__ void store_split4(uint16_t* o, int K, float4 v) {   // (hi | hi | lo) of four consecutive columns at o, o + K, o + 2 K
  uint16_t h[4], l[4];
  split1(v.x, h[0], l[0]); split1(v.y, h[1], l[1]); split1(v.z, h[2], l[2]); split1(v.w, h[3], l[3]);
  uint2 hv, lv;
  hv.x = h[0] | ((uint32_t)h[1] << 16); hv.y = h[2] | ((uint32_t)h[3] << 16);
  lv.x = l[0] | ((uint32_t)l[1] << 16); lv.y = l[2] | ((uint32_t)l[3] << 16);
  *reinterpret_cast<uint2*>(o) = hv;
  *reinterpret_cast<uint2*>(o + K) = hv;
  *reinterpret_cast<uint2*>(o + 2 * K) = lv;
}

// MODE 3 / 4 (STLLM_SPLIT_OUT): the activation / SwiGLU of modes 0 / 1 from the fp32 temporary t, written straight as the split A operand
// of the NEXT bf16x3 GEMM — bf16 [M, ldo >= 3 N'] = (hi | hi | lo) — instead of an fp32 tensor that stllm_split3_rows would read again.
template <int MODE>
__global__ __launch_bounds__(256) void post_split_rows_kernel(const float* __restrict__ t, int64_t ldt, uint16_t* __restrict__ out, int64_t ldo, int o_rpb,
                                                              int64_t o_bs, int M, int N, int act) {
  const int row = blockIdx.x;
  if (row >= M) return;
  uint16_t* o = out + row_off(row, ldo, o_rpb, o_bs);
  const float* s = t + (int64_t)row * ldt;
  if constexpr (MODE == 3) {
    for (int c = threadIdx.x; c < (N >> 2); c += blockDim.x) {
      float4 v = reinterpret_cast<const float4*>(s)[c];
      if (act == STLLM_ACT_GELU) { v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w); }
      else if (act == STLLM_ACT_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      store_split4(o + 4 * c, N, v);
    }
  } else {
    const int No = N >> 1;
    for (int c = threadIdx.x; c < (N >> 3); c += blockDim.x) {
      const int g = c >> 3, q = (c & 7) * 4;
      const float4 a = *reinterpret_cast<const float4*>(s + 64 * g + q), b = *reinterpret_cast<const float4*>(s + 64 * g + 32 + q);
      float4 r;
      r.x = silu_f(a.x) * b.x; r.y = silu_f(a.y) * b.y; r.z = silu_f(a.z) * b.z; r.w = silu_f(a.w) * b.w;
      store_split4(o + 32 * g + q, No, r);
    }
  }
}

template <int MODE>
__global__ __launch_bounds__(256) void post_rows_kernel(const float* __restrict__ t, int64_t ldt, float* __restrict__ out, int64_t ldo, int o_rpb,
                                                        int64_t o_bs, int M, int N, int act, const float* __restrict__ cosb,
                                                        const float* __restrict__ sinb, int rope_seq, int rope_cols) {
  const int row = blockIdx.x;
  if (row >= M) return;
  float* o = out + row_off(row, ldo, o_rpb, o_bs);
  if constexpr (MODE == 0) {
    for (int c = threadIdx.x; c < (N >> 2); c += blockDim.x) {
      float4 v = reinterpret_cast<float4*>(o)[c];
      if (act == STLLM_ACT_GELU) { v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w); }
      else { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      reinterpret_cast<float4*>(o)[c] = v;
    }
  } else if constexpr (MODE == 1) {
    const float* s = t + (int64_t)row * ldt;
    for (int c = threadIdx.x; c < (N >> 3); c += blockDim.x) {   // four outputs per thread: group g = c / 8, columns 4 (c % 8) ..
      const int g = c >> 3, q = (c & 7) * 4;
      const float4 a = *reinterpret_cast<const float4*>(s + 64 * g + q), b = *reinterpret_cast<const float4*>(s + 64 * g + 32 + q);
      float4 r;
      r.x = silu_f(a.x) * b.x; r.y = silu_f(a.y) * b.y; r.z = silu_f(a.z) * b.z; r.w = silu_f(a.w) * b.w;
      *reinterpret_cast<float4*>(o + 32 * g + q) = r;
    }
  } else {
    const int pos = row % rope_seq;
    for (int c = threadIdx.x; c < (rope_cols >> 3); c += blockDim.x) {
      const int g = c >> 3, q = (c & 7) * 4, fi = (g & 1) * 32 + q;
      float4 a = *reinterpret_cast<float4*>(o + 64 * g + q), b = *reinterpret_cast<float4*>(o + 64 * g + 32 + q);
      const float4 cs = *reinterpret_cast<const float4*>(cosb + pos * 64 + fi), sn = *reinterpret_cast<const float4*>(sinb + pos * 64 + fi);
      float4 x, y;
      x.x = a.x * cs.x - b.x * sn.x; y.x = b.x * cs.x + a.x * sn.x;
      x.y = a.y * cs.y - b.y * sn.y; y.y = b.y * cs.y + a.y * sn.y;
      x.z = a.z * cs.z - b.z * sn.z; y.z = b.z * cs.z + a.z * sn.z;
      x.w = a.w * cs.w - b.w * sn.w; y.w = b.w * cs.w + a.w * sn.w;
      *reinterpret_cast<float4*>(o + 64 * g + q) = x;
      *reinterpret_cast<float4*>(o + 64 * g + 32 + q) = y;
    }
  }
}

inline int64_t up256(int64_t v) { return (v + 255) / 256 * 256; }

}  // namespace

extern "C" int stllm_split3_rows(const float* x, int64_t ldx, int rows_per_batch, int64_t batch_stride, void* out, int64_t ldo, int M, int K,
                                 int weight_side, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  STLLM_CHECK_ARG(M > 0 && K > 0 && K % 4 == 0, "stllm_split3_rows: bad M=%d K=%d (K %% 4 == 0)", M, K);
  STLLM_CHECK_ARG(x && out && aligned16(x) && ldx % 4 == 0 && batch_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 7) == 0 && ldo % 4 == 0 && ldo >= 3 * (int64_t)K,
                  "stllm_split3_rows: null / misaligned buffers or ldo < 3 K");
  const int64_t n = (int64_t)M * (K >> 2);
  dim3 grid((unsigned)((n + 255) / 256)), block(256);
  if (weight_side) hipLaunchKernelGGL((split3_rows_kernel<1>), grid, block, 0, stream, x, ldx, rows_per_batch, batch_stride, reinterpret_cast<uint16_t*>(out), ldo, M, K);
  else hipLaunchKernelGGL((split3_rows_kernel<0>), grid, block, 0, stream, x, ldx, rows_per_batch, batch_stride, reinterpret_cast<uint16_t*>(out), ldo, M, K);
  STLLM_CHECK_LAUNCH("stllm_split3_rows");
  return STLLM_OK;
}

extern "C" int64_t stllm_gemm_split_ws_bytes(int M, int N, int K, int epilogue, int split_flags) {
  if (M <= 0 || N <= 0 || K <= 0) return -1;
  return ((split_flags & STLLM_SPLIT_A_PRESPLIT) ? 0 : up256((int64_t)M * 3 * K * 2)) +
         ((epilogue == STLLM_EPI_SWIGLU || (split_flags & STLLM_SPLIT_OUT)) ? up256((int64_t)M * N * 4) : 0);
}

// stllm_gemm with dtype STLLM_BF16X3 (called from gemm.hip after the common argument checks): A f32 [M, K] (or, STLLM_SPLIT_A_PRESPLIT, already the
// split image bf16 [M, 3 K]), W bf16 [N, 3 K] (side 1 layout), every output fp32 (or, STLLM_SPLIT_OUT, the split image of it).
int stllm_gemm_bf16x3(const stllm_gemm_args* a, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  const bool presplit = (a->split_flags & STLLM_SPLIT_A_PRESPLIT) != 0, out_split = (a->split_flags & STLLM_SPLIT_OUT) != 0;
  STLLM_CHECK_ARG(a->epilogue != STLLM_EPI_PATCH && a->a_norm_x == nullptr,
                  "stllm_gemm(BF16X3): the patch-embed gather and the fused RMSNorm operand have no split form (use STLLM_F32 / stllm_rmsnorm)");
  STLLM_CHECK_ARG(a->K % 64 == 0 && a->ldw >= 3 * (int64_t)a->K && (a->ldw * 2) % 16 == 0, "stllm_gemm(BF16X3): K %% 64 == 0 and W = bf16 [N, ldw >= 3 K] (stllm_split3_rows side 1)");
  const int K3 = 3 * a->K;
  if (presplit)
    STLLM_CHECK_ARG(a->A && aligned16(a->A) && (a->lda * 2) % 16 == 0 && a->lda >= K3 && (a->a_batch_stride * 2) % 16 == 0,
                    "stllm_gemm(BF16X3, A pre-split): A must be bf16 rows of 3 K elements, 16-byte aligned");
  else
    STLLM_CHECK_ARG(a->A && aligned16(a->A) && a->lda % 4 == 0 && a->lda >= a->K && a->a_batch_stride % 4 == 0, "stllm_gemm(BF16X3): A must be fp32 rows, 16-byte aligned");
  STLLM_CHECK_ARG(!out_split || a->epilogue == STLLM_EPI_STORE || a->epilogue == STLLM_EPI_SWIGLU, "stllm_gemm(BF16X3): STLLM_SPLIT_OUT goes with STORE / SWIGLU only");
  const int64_t need = stllm_gemm_split_ws_bytes(a->M, a->N, a->K, a->epilogue, a->split_flags);
  STLLM_CHECK_ARG(need == 0 || (a->split_ws && aligned16(a->split_ws) && a->split_ws_bytes >= need),
                  "stllm_gemm(BF16X3): split_ws of %lld bytes needed (stllm_gemm_split_ws_bytes), %lld given", (long long)need, (long long)a->split_ws_bytes);
  STLLM_CHECK_ARG(a->epilogue != STLLM_EPI_SWIGLU || a->N % 64 == 0, "stllm_gemm(BF16X3, SWIGLU): N %% 64");
  // the inner GEMM runs as a plain STORE, so the checks of the epilogue that the post pass applies have to be repeated here, before anything is launched
  // (a C caller with rope_seq = 0 or NULL tables got a device fault out of post_rows_kernel<2>, a bad `act` silently ran as ReLU)
  STLLM_CHECK_ARG(a->epilogue != STLLM_EPI_ROPE || (a->aux0 && a->aux1 && a->rope_seq > 0 && a->rope_cols % 128 == 0),
                  "stllm_gemm(BF16X3, ROPE): need cos/sin tables, rope_seq, rope_cols%%128==0");
  STLLM_CHECK_ARG(a->act == STLLM_ACT_NONE || a->act == STLLM_ACT_GELU || a->act == STLLM_ACT_RELU, "stllm_gemm(BF16X3): bad act %d", a->act);
  if (out_split) {   // (checked before anything is launched)
    const int n_out = a->epilogue == STLLM_EPI_SWIGLU ? a->N / 2 : a->N;
    STLLM_CHECK_ARG(a->ldo >= 3 * (int64_t)n_out && a->ldo % 4 == 0 && a->o_batch_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(a->out) & 7) == 0,
                    "stllm_gemm(BF16X3, STLLM_SPLIT_OUT): out = bf16 [M, ldo >= 3 x %d], 8-byte aligned", n_out);
  } else {
    STLLM_CHECK_ARG(a->ldo % 4 == 0 && a->o_batch_stride % 4 == 0, "stllm_gemm(BF16X3): fp32 output rows must be 16-byte aligned (ldo=%lld)", (long long)a->ldo);
  }
  char* ws = reinterpret_cast<char*>(a->split_ws);
  int rc;
  stllm_gemm_args g = *a;
  g.dtype = STLLM_BF16;
  g.K = K3;
  g.split_ws = nullptr; g.split_ws_bytes = 0; g.split_flags = 0;
  int64_t ws_off = 0;
  if (!presplit) {
    rc = stllm_split3_rows(reinterpret_cast<const float*>(a->A), a->lda, a->a_rows_per_batch, a->a_batch_stride, ws, K3, a->M, a->K, 0, stream_);
    if (rc != STLLM_OK) return rc;
    g.A = ws; g.lda = K3;
    g.a_rows_per_batch = 0; g.a_batch_stride = 0;
    ws_off = up256((int64_t)a->M * K3 * 2);
  }
  float* tmp = nullptr;
  if (a->epilogue == STLLM_EPI_RESID) {
    // out = resid + acc + bias in fp32: the 16-bit kernels' own epilogue
  } else {
    g.epilogue = STLLM_EPI_STORE; g.act = STLLM_ACT_NONE; g.out_is_f32 = 1;
    if (a->epilogue == STLLM_EPI_SWIGLU || out_split) {
      tmp = reinterpret_cast<float*>(ws + ws_off);
      g.out = tmp; g.ldo = a->N; g.o_rows_per_batch = 0; g.o_batch_stride = 0;
    }
  }
  rc = stllm_gemm(&g, stream_);
  if (rc != STLLM_OK) return rc;
  dim3 grid(a->M), block(256);
  float* out = reinterpret_cast<float*>(a->out);
  if (out_split) {
    uint16_t* o16 = reinterpret_cast<uint16_t*>(a->out);
    if (a->epilogue == STLLM_EPI_SWIGLU)
      hipLaunchKernelGGL((post_split_rows_kernel<4>), grid, block, 0, stream, tmp, a->N, o16, a->ldo, a->o_rows_per_batch, a->o_batch_stride, a->M, a->N, 0);
    else
      hipLaunchKernelGGL((post_split_rows_kernel<3>), grid, block, 0, stream, tmp, a->N, o16, a->ldo, a->o_rows_per_batch, a->o_batch_stride, a->M, a->N, a->act);
  } else if (a->epilogue == STLLM_EPI_STORE && a->act != STLLM_ACT_NONE) {
    hipLaunchKernelGGL((post_rows_kernel<0>), grid, block, 0, stream, nullptr, 0, out, a->ldo, a->o_rows_per_batch, a->o_batch_stride, a->M, a->N, a->act,
                       nullptr, nullptr, 1, 0);
  } else if (a->epilogue == STLLM_EPI_SWIGLU) {
    hipLaunchKernelGGL((post_rows_kernel<1>), grid, block, 0, stream, tmp, a->N, out, a->ldo, a->o_rows_per_batch, a->o_batch_stride, a->M, a->N, 0,
                       nullptr, nullptr, 1, 0);
  } else if (a->epilogue == STLLM_EPI_ROPE) {
    const int rc_cols = a->rope_cols < a->N ? a->rope_cols : a->N;
    hipLaunchKernelGGL((post_rows_kernel<2>), grid, block, 0, stream, nullptr, 0, out, a->ldo, a->o_rows_per_batch, a->o_batch_stride, a->M, a->N, 0,
                       a->aux0, a->aux1, a->rope_seq, rc_cols);
  }
  STLLM_CHECK_LAUNCH("stllm_gemm(BF16X3 post-epilogue)");
  return STLLM_OK;
}
