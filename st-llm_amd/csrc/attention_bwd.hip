// Attention backward for the training step: gradients of  O = softmax(scale * Q K^T + masks) V
// causal and / or key-length masked (right-padded batches), operands in the strided fused-QKV layout of
// the forward kernels (include/stllm_hip.h: element (b, s, h, d) at base[b * batch_stride + s * row_stride + h * D + d]).
//
// Flash-attention-2 backward structure, three kernels, nothing S x S ever touches HBM:
//   1. stats: per query row  lse = log sum_j exp(s_ij),  delta = dO_i . O_i                (workspace, 8 bytes per row)
//   2. dq   : one workgroup per 32 queries, sweeps the visible key tiles: P = exp(s - lse), dS = P * (dO V^T - delta),
//             dQ = scale * dS K
//   3. dkv  : one workgroup per 32 keys, sweeps the query tiles that see them: dV = P^T dO,  dK = scale * dS^T Q
// Tiles are staged in LDS as fp32 (row stride 132 floats: the float4 reads of 8 consecutive rows hit 32 distinct banks).
// Two implementations of the same structure:
//   * VALU (fp32 FMA on fp32 LDS tiles, row stride 132 floats): exact-fp32 "verify" mode, and the reference the MFMA path is
//     compared with (STLLM_ATTN_BWD_VALU=1 forces it for 16-bit operands too);
//   * MFMA (bf16 / f16, v_mfma_f32_32x32x16, "everything transposed" like the forward kernel in attention.hip): a wave owns 32
//     queries (dQ kernel) or 32 keys (dK/dV kernel) as the COLUMNS of every accumulator tile, so the per-row softmax statistics
//     are per-lane scalars (dQ) and the P / dS accumulators are already in B-operand form for the second product:
//        dQ kernel :  S^T = K Q^T,  dP^T = V dO^T           (A = K / V rows from LDS, B = Q / dO fragments in VGPRs)
//                     dQ^T += K^T dS^T                      (A = K^T from LDS, B = dS^T accumulator registers, packed to 16 bit)
//        dKV kernel:  S = Q K^T,   dP = dO V^T              (A = Q / dO rows from LDS, B = K / V fragments in VGPRs)
//                     dV^T += dO^T P,  dK^T += Q^T dS       (A = dO^T / Q^T from LDS, B = P / dS accumulator registers)
//     The statistics pass (log-sum-exp in the log2 domain + dO.o) is the first sweep of the dQ kernel, written to the workspace
//     for the dK/dV kernel.  24 MFMAs per (32 queries x 32 keys) in the dQ sweep + 8 in its statistics sweep, 32 in the dK/dV sweep.
// Algorithmic FLOPs: 5 * 2 * S^2/2 * 128 per (b, h) for causal masks.
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int kT = 32;        // tile edge (queries / keys)

struct Ptr { const void* p; int64_t bs, rs; };
struct MPtr { void* p; int64_t bs, rs; };

// ---- VALU path: any head dim D <= DP (DP in {64, 96, 128}; 88 is zero-padded to 96 in LDS only), Sq != Skv allowed ------------
// rows [row0, row0 + 32) of head (b, h) -> lds[32][DP + 4] as fp32, zero beyond `rows` and beyond D
template <typename T, int DP>
__device__ __forceinline__ void load_tile(float* lds, const Ptr& t, int b, int h, int D, int row0, int rows, int tid) {
  constexpr int LD = DP + 4, CPR = DP / 8;
  const int64_t off = (int64_t)b * t.bs + (int64_t)h * D;
  for (int v = tid; v < kT * CPR; v += 256) {
    const int r = v / CPR, c = (v % CPR) * 8;
    float f[8];
    if (row0 + r < rows && c < D) load8<T>(t.p, off + (int64_t)(row0 + r) * t.rs + c, f);
    else {
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = 0.0f;
    }
    float4* d = reinterpret_cast<float4*>(lds + r * LD + c);
    d[0] = make_float4(f[0], f[1], f[2], f[3]);
    d[1] = make_float4(f[4], f[5], f[6], f[7]);
  }
}

template <int DP> __device__ __forceinline__ float dot_dp(const float* a, const float* b) {
  float s = 0.0f;
#pragma unroll 8
  for (int d = 0; d < DP; d += 4) {
    const float4 x = *reinterpret_cast<const float4*>(a + d), y = *reinterpret_cast<const float4*>(b + d);
    s = fmaf(x.x, y.x, s); s = fmaf(x.y, y.y, s); s = fmaf(x.z, y.z, s); s = fmaf(x.w, y.w, s);
  }
  return s;
}
__device__ __forceinline__ float oct_sum(float v) {   // over the 8 consecutive lanes that share a row
  v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
  return v;
}
__device__ __forceinline__ float oct_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 1, 64)); v = fmaxf(v, __shfl_xor(v, 2, 64)); v = fmaxf(v, __shfl_xor(v, 4, 64));
  return v;
}
// this thread's DP/8 consecutive dims of one output row (d < D only)
template <typename T, int DP>
__device__ __forceinline__ void store_dims(const MPtr& t, int b, int h, int D, int row, int d0, const float* acc, float mul) {
  const int64_t off = (int64_t)b * t.bs + (int64_t)row * t.rs + (int64_t)h * D;
#pragma unroll
  for (int e = 0; e < DP / 8; ++e)
    if (d0 + e < D) store_elem<T>(t.p, off + d0 + e, acc[e] * mul);
}

struct BwdDims { int H, Sq, Skv, D; float scale; int causal; const int32_t* kv_len; };

// ---- 1. statistics -------------------------------------------------------------------------------------------------------
template <typename T, int DP>
__global__ __launch_bounds__(256) void attn_bwd_stats_kernel(Ptr q, Ptr k, Ptr o, Ptr dO, float* __restrict__ ws, BwdDims p) {
  constexpr int LD = DP + 4, DPT = DP / 8;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Qs = smem;
  float* Ks = smem + kT * LD;
  const int tid = threadIdx.x, qi = tid >> 3, kg = tid & 7;
  const int q0 = blockIdx.x * kT, h = blockIdx.y, b = blockIdx.z;
  const int kmax = p.kv_len ? min(p.Skv, p.kv_len[b]) : p.Skv;
  const int kend = p.causal ? min(kmax, q0 + kT) : kmax;
  load_tile<T, DP>(Qs, q, b, h, p.D, q0, p.Sq, tid);
  float m = -3.0e38f, l = 0.0f;
  for (int k0 = 0; k0 < kend; k0 += kT) {
    __syncthreads();
    load_tile<T, DP>(Ks, k, b, h, p.D, k0, p.Skv, tid);
    __syncthreads();
    float s[4], mt = -3.0e38f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int kj = kg * 4 + t, kk = k0 + kj;
      const bool ok = kk < kend && (!p.causal || kk <= q0 + qi);
      s[t] = ok ? dot_dp<DP>(Qs + qi * LD, Ks + kj * LD) * p.scale : -3.0e38f;
      mt = fmaxf(mt, s[t]);
    }
    mt = oct_max(mt);
    const float mn = fmaxf(m, mt);
    float ps = 0.0f;
#pragma unroll
    for (int t = 0; t < 4; ++t) ps += s[t] > -1.0e38f ? __expf(s[t] - mn) : 0.0f;
    ps = oct_sum(ps);
    if (mn > -1.0e38f) { l = l * __expf(m - mn) + ps; m = mn; }
  }
  // delta = dO . O over this thread's DPT of the DP dims
  float dl = 0.0f;
  if (q0 + qi < p.Sq) {
    const int64_t oo = (int64_t)b * o.bs + (int64_t)(q0 + qi) * o.rs + (int64_t)h * p.D;
    const int64_t od = (int64_t)b * dO.bs + (int64_t)(q0 + qi) * dO.rs + (int64_t)h * p.D;
#pragma unroll
    for (int e = 0; e < DPT; ++e) {
      const int d = kg * DPT + e;
      if (d < p.D) dl = fmaf(load_elem<T>(o.p, oo + d), load_elem<T>(dO.p, od + d), dl);
    }
  }
  dl = oct_sum(dl);
  if (kg == 0 && q0 + qi < p.Sq) {
    float* w = ws + ((int64_t)(b * p.H + h) * p.Sq + q0 + qi) * 2;
    w[0] = m + __logf(l);
    w[1] = dl;
  }
}

// ---- 2. dQ -----------------------------------------------------------------------------------------------------------------
template <typename T, int DP>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(Ptr q, Ptr k, Ptr v, Ptr dO, MPtr dq, const float* __restrict__ ws, BwdDims p) {
  constexpr int LD = DP + 4, DPT = DP / 8, TILE = kT * LD;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Qs = smem;
  float* Os = smem + TILE;
  float* Ks = smem + 2 * TILE;
  float* Vs = smem + 3 * TILE;
  float* Ds = smem + 4 * TILE;            // [32][33]
  const int tid = threadIdx.x, qi = tid >> 3, kg = tid & 7;
  const int q0 = blockIdx.x * kT, h = blockIdx.y, b = blockIdx.z;
  const int kmax = p.kv_len ? min(p.Skv, p.kv_len[b]) : p.Skv;
  const int kend = p.causal ? min(kmax, q0 + kT) : kmax;
  load_tile<T, DP>(Qs, q, b, h, p.D, q0, p.Sq, tid);
  load_tile<T, DP>(Os, dO, b, h, p.D, q0, p.Sq, tid);
  float lse = 0.0f, delta = 0.0f;
  if (q0 + qi < p.Sq) {
    const float* w = ws + ((int64_t)(b * p.H + h) * p.Sq + q0 + qi) * 2;
    lse = w[0]; delta = w[1];
  }
  float acc[DPT];
#pragma unroll
  for (int e = 0; e < DPT; ++e) acc[e] = 0.0f;
  for (int k0 = 0; k0 < kend; k0 += kT) {
    __syncthreads();
    load_tile<T, DP>(Ks, k, b, h, p.D, k0, p.Skv, tid);
    load_tile<T, DP>(Vs, v, b, h, p.D, k0, p.Skv, tid);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int kj = kg * 4 + t, kk = k0 + kj;
      const bool ok = kk < kend && (!p.causal || kk <= q0 + qi) && q0 + qi < p.Sq;
      float ds = 0.0f;
      if (ok) {
        const float s = dot_dp<DP>(Qs + qi * LD, Ks + kj * LD) * p.scale;
        const float dp = dot_dp<DP>(Os + qi * LD, Vs + kj * LD);
        ds = __expf(s - lse) * (dp - delta);
      }
      Ds[qi * 33 + kj] = ds;
    }
    __syncthreads();
#pragma unroll 4
    for (int j = 0; j < kT; ++j) {
      const float w = Ds[qi * 33 + j];
      const float* kr = Ks + j * LD + kg * DPT;
#pragma unroll
      for (int e = 0; e < DPT; e += 4) {
        const float4 x = *reinterpret_cast<const float4*>(kr + e);
        acc[e] = fmaf(w, x.x, acc[e]); acc[e + 1] = fmaf(w, x.y, acc[e + 1]);
        acc[e + 2] = fmaf(w, x.z, acc[e + 2]); acc[e + 3] = fmaf(w, x.w, acc[e + 3]);
      }
    }
  }
  if (q0 + qi < p.Sq) store_dims<T, DP>(dq, b, h, p.D, q0 + qi, kg * DPT, acc, p.scale);
}

// ---- 3. dK, dV ---------------------------------------------------------------------------------------------------------------
template <typename T, int DP>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(Ptr q, Ptr k, Ptr v, Ptr dO, MPtr dk, MPtr dv, const float* __restrict__ ws,
                                                           BwdDims p) {
  constexpr int LD = DP + 4, DPT = DP / 8, TILE = kT * LD;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Qs = smem;
  float* Os = smem + TILE;
  float* Ks = smem + 2 * TILE;
  float* Vs = smem + 3 * TILE;
  float* Ps = smem + 4 * TILE;            // [32 q][33]
  float* Ds = Ps + kT * 33;               // [32 q][33]
  const int tid = threadIdx.x, kj = tid >> 3, qg = tid & 7;
  const int k0 = blockIdx.x * kT, h = blockIdx.y, b = blockIdx.z;
  const int kmax = p.kv_len ? min(p.Skv, p.kv_len[b]) : p.Skv;
  const int kk = k0 + kj;
  load_tile<T, DP>(Ks, k, b, h, p.D, k0, p.Skv, tid);
  load_tile<T, DP>(Vs, v, b, h, p.D, k0, p.Skv, tid);
  float av[DPT], ak[DPT];
#pragma unroll
  for (int e = 0; e < DPT; ++e) av[e] = ak[e] = 0.0f;
  const float* wrow = ws + (int64_t)(b * p.H + h) * p.Sq * 2;
  for (int q0 = p.causal ? k0 : 0; q0 < p.Sq; q0 += kT) {
    __syncthreads();
    load_tile<T, DP>(Qs, q, b, h, p.D, q0, p.Sq, tid);
    load_tile<T, DP>(Os, dO, b, h, p.D, q0, p.Sq, tid);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int qi = qg * 4 + t, qq = q0 + qi;
      const bool ok = qq < p.Sq && kk < kmax && (!p.causal || kk <= qq);
      float pr = 0.0f, ds = 0.0f;
      if (ok) {
        const float s = dot_dp<DP>(Qs + qi * LD, Ks + kj * LD) * p.scale;
        const float dp = dot_dp<DP>(Os + qi * LD, Vs + kj * LD);
        pr = __expf(s - wrow[2 * qq]);
        ds = pr * (dp - wrow[2 * qq + 1]);
      }
      Ps[qi * 33 + kj] = pr;
      Ds[qi * 33 + kj] = ds;
    }
    __syncthreads();
#pragma unroll 2
    for (int i = 0; i < kT; ++i) {
      const float pr = Ps[i * 33 + kj], ds = Ds[i * 33 + kj];
      const float* orow = Os + i * LD + qg * DPT;
      const float* qrow = Qs + i * LD + qg * DPT;
#pragma unroll
      for (int e = 0; e < DPT; e += 4) {
        const float4 x = *reinterpret_cast<const float4*>(orow + e), y = *reinterpret_cast<const float4*>(qrow + e);
        av[e] = fmaf(pr, x.x, av[e]); av[e + 1] = fmaf(pr, x.y, av[e + 1]); av[e + 2] = fmaf(pr, x.z, av[e + 2]); av[e + 3] = fmaf(pr, x.w, av[e + 3]);
        ak[e] = fmaf(ds, y.x, ak[e]); ak[e + 1] = fmaf(ds, y.y, ak[e + 1]); ak[e + 2] = fmaf(ds, y.z, ak[e + 2]); ak[e + 3] = fmaf(ds, y.w, ak[e + 3]);
      }
    }
  }
  if (kk < p.Skv) {
    store_dims<T, DP>(dk, b, h, p.D, kk, qg * DPT, ak, p.scale);
    store_dims<T, DP>(dv, b, h, p.D, kk, qg * DPT, av, 1.0f);
  }
}

// =========================================================================================================================
// MFMA path (16-bit operands)
// =========================================================================================================================
constexpr int kNW = 4;                  // waves per workgroup: 128 queries (dQ) / 128 keys (dK/dV)
// head dim D <= DP, DP in {64, 96, 128} (88 is zero-padded to 96 in LDS / registers only, as in the forward kernel)
template <int DP> struct MC {
  static constexpr int KS = DP / 16;            // k-steps of a DP-deep contraction
  static constexpr int DB = DP / 32;            // 32-row blocks of a [DP x 32] transposed accumulator
  static constexpr int RowPitch = DP * 2 + 16;  // bytes per row-major tile row in LDS (conflict-free ds_read_b128)
  static constexpr int NCH = 32 * (DP / 8);     // 16-byte chunks per tile
  static constexpr int CPT = (NCH + 64 * kNW - 1) / (64 * kNW);   // ... per thread
};
constexpr int kTPitch = 32 * 2 + 8;     // bytes per row of a transposed tile [d][32 rows]
constexpr float kNegBig = -1.0e30f;

// Tiles are register-staged one tile AHEAD (as in the forward kernel): fetch_tile16 issues the global loads of the next tile before the
// MFMA work of the current one, put_tile16 writes them to LDS at the top of the next iteration — HBM / L2 latency hides under compute.
template <typename T, int DP>
__device__ __forceinline__ void fetch_tile16(i32x4* reg, const Ptr& t, int b, int h, int D, int row0, int S, int tid) {
  const char* base = reinterpret_cast<const char*>(t.p) + ((int64_t)b * t.bs + (int64_t)h * D) * 2;
#pragma unroll
  for (int c = 0; c < MC<DP>::CPT; ++c) {
    const int ch = tid + c * 64 * kNW;
    const int cc = ch >> 5, row = ch & 31;      // consecutive lanes = consecutive rows: conflict-free transposed writes
    const i32x4 z = {0, 0, 0, 0};
    reg[c] = (ch < MC<DP>::NCH && row0 + row < S && cc * 8 < D)
                 ? *reinterpret_cast<const i32x4*>(base + ((int64_t)(row0 + row) * t.rs + cc * 8) * 2) : z;
  }
}
// row-major copy (rows_lds) and, if t_lds, the transposed copy [d][row]
template <int DP>
__device__ __forceinline__ void put_tile16(const i32x4* reg, char* rows_lds, char* t_lds, int tid) {
#pragma unroll
  for (int c = 0; c < MC<DP>::CPT; ++c) {
    const int ch = tid + c * 64 * kNW;
    if (ch >= MC<DP>::NCH) continue;
    const int cc = ch >> 5, row = ch & 31;
    const i32x4 x = reg[c];
    *reinterpret_cast<i32x4*>(rows_lds + row * MC<DP>::RowPitch + cc * 16) = x;
    if (t_lds) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const uint32_t w = (uint32_t)x[e];
        *reinterpret_cast<uint16_t*>(t_lds + (cc * 8 + 2 * e) * kTPitch + row * 2) = (uint16_t)(w & 0xffff);
        *reinterpret_cast<uint16_t*>(t_lds + (cc * 8 + 2 * e + 1) * kTPitch + row * 2) = (uint16_t)(w >> 16);
      }
    }
  }
}
// this lane's B-operand fragments of row `row`: X[row][ks*16 + lh*8 .. +8], zero beyond S
template <typename T, int DP>
__device__ __forceinline__ void load_frags(i32x4* f, const Ptr& t, int b, int h, int D, int row, int S, int lh) {
  const char* base = reinterpret_cast<const char*>(t.p) + ((int64_t)b * t.bs + (int64_t)(row < S ? row : 0) * t.rs + (int64_t)h * D) * 2;
#pragma unroll
  for (int ks = 0; ks < MC<DP>::KS; ++ks) {
    const i32x4 z = {0, 0, 0, 0};
    const int d0 = ks * 16 + lh * 8;
    f[ks] = (row < S && d0 < D) ? *reinterpret_cast<const i32x4*>(base + d0 * 2) : z;
  }
}
template <typename T> __device__ __forceinline__ float dot8(i32x4 a, i32x4 b) {
  float s = 0.0f;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const uint32_t ua = (uint32_t)a[e], ub = (uint32_t)b[e];
    s = fmaf(Elem<T>::unpack((uint16_t)(ua & 0xffffu)), Elem<T>::unpack((uint16_t)(ub & 0xffffu)), s);
    s = fmaf(Elem<T>::unpack((uint16_t)(ua >> 16)), Elem<T>::unpack((uint16_t)(ub >> 16)), s);
  }
  return s;
}
// X . Y^T for one 32x32 tile over the DP-deep contraction: A rows from a row-major LDS tile, B fragments in registers
template <typename T, int DP> __device__ __forceinline__ f32x16 tile_nt(const char* rows_lds, const i32x4* bf, int li, int lh) {
  f32x16 s;
#pragma unroll
  for (int r = 0; r < 16; ++r) s[r] = 0.0f;
#pragma unroll
  for (int ks = 0; ks < MC<DP>::KS; ++ks) {
    const i32x4 af = *reinterpret_cast<const i32x4*>(rows_lds + li * MC<DP>::RowPitch + (ks * 2 + lh) * 16);
    s = Elem<T>::mfma(af, bf[ks], s);
  }
  return s;
}
// acc^T[d][col] += X^T[d][row] . W[row][col]: A from the transposed LDS tile, B = the 32x32 accumulator `w` (rows x this lane's
// column) packed to 16 bit; the accumulator register order IS the contraction order (see attention.hip)
template <typename T, int DP> __device__ __forceinline__ void tile_tn(f32x16* acc, const char* t_lds, const f32x16& w, int li, int lh) {
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    i32x4 bf;
#pragma unroll
    for (int e = 0; e < 4; ++e) bf[e] = (int)(Elem<T>::pack2(w[a * 8 + 2 * e], w[a * 8 + 2 * e + 1]));
#pragma unroll
    for (int i = 0; i < MC<DP>::DB; ++i) {
      const char* tp = t_lds + (i * 32 + li) * kTPitch + (16 * a + 4 * lh) * 2;
      const i32x2 lo = *reinterpret_cast<const i32x2*>(tp);
      const i32x2 hi = *reinterpret_cast<const i32x2*>(tp + 16);
      const i32x4 af = {lo[0], lo[1], hi[0], hi[1]};
      acc[i] = Elem<T>::mfma(af, bf, acc[i]);
    }
  }
}
// lane holds acc^T[d = i*32 + 8g + 4lh + (0..3)][row]: 8-byte stores of 4 consecutive d
template <typename T, int DP>
__device__ __forceinline__ void store_acc_t(const f32x16* acc, const MPtr& t, int b, int h, int D, int row, int lh, float mul) {
  uint16_t* op = reinterpret_cast<uint16_t*>(t.p) + (int64_t)b * t.bs + (int64_t)row * t.rs + (int64_t)h * D;
#pragma unroll
  for (int i = 0; i < MC<DP>::DB; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (i * 32 + 8 * g + 4 * lh >= D) continue;      // D % 4 == 0
      uint2 pk;
      pk.x = Elem<T>::pack2(acc[i][4 * g + 0] * mul, acc[i][4 * g + 1] * mul);
      pk.y = Elem<T>::pack2(acc[i][4 * g + 2] * mul, acc[i][4 * g + 3] * mul);
      *reinterpret_cast<uint2*>(op + i * 32 + 8 * g + 4 * lh) = pk;
    }
}

template <typename T, int DP>
__global__ __launch_bounds__(64 * kNW) void attn_bwd_dq_mfma_kernel(Ptr q, Ptr k, Ptr v, Ptr o, Ptr dO, MPtr dq, float* __restrict__ ws, int H,
                                                                   int S, int Skv, int D, float scale, int causal, const int32_t* __restrict__ kv_len) {
  constexpr int kKS = MC<DP>::KS, kDB = MC<DP>::DB, kCPT = MC<DP>::CPT;
  __shared__ __attribute__((aligned(16))) char k_lds[32 * MC<DP>::RowPitch];
  __shared__ __attribute__((aligned(16))) char v_lds[32 * MC<DP>::RowPitch];
  __shared__ __attribute__((aligned(16))) char kt_lds[DP * kTPitch];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  const int q_blk0 = blockIdx.x * (32 * kNW);
  const int qrow = q_blk0 + wave * 32 + li;                 // this lane's query (column of every accumulator tile)
  const int wave_q_last = q_blk0 + wave * 32 + 31;
  const int kvlen = kv_len ? min(kv_len[b], Skv) : Skv;
  const int kv_end = causal ? min(kvlen, q_blk0 + 32 * kNW) : kvlen;
  const float scale_log2 = scale * 1.4426950408889634f;
  i32x4 qf[kKS], dof[kKS];
  load_frags<T, DP>(qf, q, b, h, D, qrow, S, lh);
  load_frags<T, DP>(dof, dO, b, h, D, qrow, S, lh);
  float delta = 0.0f;
  {
    i32x4 of[kKS];
    load_frags<T, DP>(of, o, b, h, D, qrow, S, lh);
#pragma unroll
    for (int ks = 0; ks < kKS; ++ks) delta += dot8<T>(dof[ks], of[ks]);
    delta += __shfl_xor(delta, 32, 64);
  }
  // ---- sweep 1: log-sum-exp of the scaled scores, log2 domain ------------------------------------------------------------
  float m_run = kNegBig, l_run = 0.0f;
  i32x4 kreg[kCPT], vreg[kCPT];
  if (kv_end > 0) fetch_tile16<T, DP>(kreg, k, b, h, D, 0, Skv, tid);
  for (int kv0 = 0; kv0 < kv_end; kv0 += 32) {
    __syncthreads();                       // previous tile fully consumed
    put_tile16<DP>(kreg, k_lds, nullptr, tid);
    __syncthreads();
    if (kv0 + 32 < kv_end) fetch_tile16<T, DP>(kreg, k, b, h, D, kv0 + 32, Skv, tid);
    if (causal && kv0 > wave_q_last) continue;
    f32x16 s = tile_nt<T, DP>(k_lds, qf, li, lh);
    float mx = kNegBig;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kv = kv0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      const bool dead = (kv >= kvlen) || (causal && kv > qrow);
      s[r] = dead ? kNegBig : s[r];
      mx = fmaxf(mx, s[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    float rs = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) rs += s[r] > -1.0e29f ? __builtin_amdgcn_exp2f((s[r] - m_new) * scale_log2) : 0.0f;
    rs += __shfl_xor(rs, 32, 64);
    l_run = l_run * __builtin_amdgcn_exp2f((m_run - m_new) * scale_log2) + rs;
    m_run = m_new;
  }
  const float lse2 = m_run * scale_log2 + __log2f(l_run);
  if (lh == 0 && qrow < S) {
    float* w = ws + ((int64_t)(b * H + h) * S + qrow) * 2;
    w[0] = lse2;
    w[1] = delta;
  }
  // ---- sweep 2: dQ^T += K^T dS^T ---------------------------------------------------------------------------------------------
  f32x16 acc[kDB];
#pragma unroll
  for (int i = 0; i < kDB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
  if (kv_end > 0) {
    fetch_tile16<T, DP>(kreg, k, b, h, D, 0, Skv, tid);
    fetch_tile16<T, DP>(vreg, v, b, h, D, 0, Skv, tid);
  }
  for (int kv0 = 0; kv0 < kv_end; kv0 += 32) {
    __syncthreads();
    put_tile16<DP>(kreg, k_lds, kt_lds, tid);
    put_tile16<DP>(vreg, v_lds, nullptr, tid);
    __syncthreads();
    if (kv0 + 32 < kv_end) {
      fetch_tile16<T, DP>(kreg, k, b, h, D, kv0 + 32, Skv, tid);
      fetch_tile16<T, DP>(vreg, v, b, h, D, kv0 + 32, Skv, tid);
    }
    if (causal && kv0 > wave_q_last) continue;
    f32x16 s = tile_nt<T, DP>(k_lds, qf, li, lh);
    const f32x16 dp = tile_nt<T, DP>(v_lds, dof, li, lh);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kv = kv0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      const bool dead = (kv >= kvlen) || (causal && kv > qrow) || qrow >= S;
      const float pr = dead ? 0.0f : __builtin_amdgcn_exp2f(fmaf(s[r], scale_log2, -lse2));
      s[r] = pr * (dp[r] - delta);
    }
    tile_tn<T, DP>(acc, kt_lds, s, li, lh);
  }
  if (qrow < S) store_acc_t<T, DP>(acc, dq, b, h, D, qrow, lh, scale);
}

template <typename T, int DP>
__global__ __launch_bounds__(64 * kNW) void attn_bwd_dkv_mfma_kernel(Ptr q, Ptr k, Ptr v, Ptr dO, MPtr dk, MPtr dv, const float* __restrict__ ws,
                                                                    int H, int S, int Skv, int D, float scale, int causal,
                                                                    const int32_t* __restrict__ kv_len) {
  constexpr int kKS = MC<DP>::KS, kDB = MC<DP>::DB, kCPT = MC<DP>::CPT;
  __shared__ __attribute__((aligned(16))) char q_lds[32 * MC<DP>::RowPitch];
  __shared__ __attribute__((aligned(16))) char do_lds[32 * MC<DP>::RowPitch];
  __shared__ __attribute__((aligned(16))) char qt_lds[DP * kTPitch];
  __shared__ __attribute__((aligned(16))) char dot_lds[DP * kTPitch];
  __shared__ float st_lds[64];                               // (lse2, delta) of the 32 queries of the tile
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  const int k_blk0 = blockIdx.x * (32 * kNW);
  const int krow = k_blk0 + wave * 32 + li;                  // this lane's key (column of every accumulator tile)
  const int wave_k_first = k_blk0 + wave * 32;
  const int kvlen = kv_len ? min(kv_len[b], Skv) : Skv;
  const float scale_log2 = scale * 1.4426950408889634f;
  i32x4 kf[kKS], vf[kKS];
  load_frags<T, DP>(kf, k, b, h, D, krow, Skv, lh);
  load_frags<T, DP>(vf, v, b, h, D, krow, Skv, lh);
  f32x16 accv[kDB], acck[kDB];
#pragma unroll
  for (int i = 0; i < kDB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) accv[i][r] = acck[i][r] = 0.0f;
  const float* wrow = ws + (int64_t)(b * H + h) * S * 2;
  const int q_first = causal ? k_blk0 : 0;
  i32x4 qreg[kCPT], oreg[kCPT];
  float streg = 0.0f;
  auto fetch = [&](int q0) {
    fetch_tile16<T, DP>(qreg, q, b, h, D, q0, S, tid);
    fetch_tile16<T, DP>(oreg, dO, b, h, D, q0, S, tid);
    if (tid < 64) streg = (q0 + (tid >> 1) < S) ? wrow[2 * (q0 + (tid >> 1)) + (tid & 1)] : 0.0f;
  };
  if (q_first < S) fetch(q_first);
  for (int q0 = q_first; q0 < S; q0 += 32) {
    __syncthreads();
    put_tile16<DP>(qreg, q_lds, qt_lds, tid);
    put_tile16<DP>(oreg, do_lds, dot_lds, tid);
    if (tid < 64) st_lds[tid] = streg;
    __syncthreads();
    if (q0 + 32 < S) fetch(q0 + 32);
    if (causal && q0 + 31 < wave_k_first) continue;          // every query of the tile precedes every key of this wave
    f32x16 s = tile_nt<T, DP>(q_lds, kf, li, lh);                // s[r] = S[query (r&3)+8(r>>2)+4lh][key krow]
    f32x16 ds = tile_nt<T, DP>(do_lds, vf, li, lh);              // dP, same layout
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ql = (r & 3) + 8 * (r >> 2) + 4 * lh, qq = q0 + ql;
      const bool dead = qq >= S || krow >= kvlen || (causal && krow > qq);
      const float pr = dead ? 0.0f : __builtin_amdgcn_exp2f(fmaf(s[r], scale_log2, -st_lds[2 * ql]));
      ds[r] = pr * (ds[r] - st_lds[2 * ql + 1]);
      s[r] = pr;
    }
    tile_tn<T, DP>(accv, dot_lds, s, li, lh);                    // dV^T += dO^T P
    tile_tn<T, DP>(acck, qt_lds, ds, li, lh);                    // dK^T += Q^T dS
  }
  if (krow < Skv) {
    store_acc_t<T, DP>(accv, dv, b, h, D, krow, lh, 1.0f);
    store_acc_t<T, DP>(acck, dk, b, h, D, krow, lh, scale);
  }
}

template <typename T, int DP>
int launch_bwd_mfma_dp(Ptr q, Ptr k, Ptr v, Ptr o, Ptr dO, MPtr dq, MPtr dk, MPtr dv, float* ws, int B, int H, int S, int Skv, int D,
                       float scale, int causal, const int32_t* kv_len, hipStream_t st) {
  const dim3 gq((S + 32 * kNW - 1) / (32 * kNW), H, B), gk((Skv + 32 * kNW - 1) / (32 * kNW), H, B), block(64 * kNW);
  hipLaunchKernelGGL((attn_bwd_dq_mfma_kernel<T, DP>), gq, block, 0, st, q, k, v, o, dO, dq, ws, H, S, Skv, D, scale, causal, kv_len);
  hipLaunchKernelGGL((attn_bwd_dkv_mfma_kernel<T, DP>), gk, block, 0, st, q, k, v, dO, dk, dv, ws, H, S, Skv, D, scale, causal, kv_len);
  return STLLM_OK;
}
template <typename T>
int launch_bwd_mfma(Ptr q, Ptr k, Ptr v, Ptr o, Ptr dO, MPtr dq, MPtr dk, MPtr dv, float* ws, int B, int H, int S, int Skv, int D, float scale,
                    int causal, const int32_t* kv_len, hipStream_t st) {
  if (D <= 64) return launch_bwd_mfma_dp<T, 64>(q, k, v, o, dO, dq, dk, dv, ws, B, H, S, Skv, D, scale, causal, kv_len, st);
  if (D <= 96) return launch_bwd_mfma_dp<T, 96>(q, k, v, o, dO, dq, dk, dv, ws, B, H, S, Skv, D, scale, causal, kv_len, st);
  return launch_bwd_mfma_dp<T, 128>(q, k, v, o, dO, dq, dk, dv, ws, B, H, S, Skv, D, scale, causal, kv_len, st);
}

template <typename T, int DP>
int launch_bwd(Ptr q, Ptr k, Ptr v, Ptr o, Ptr dO, MPtr dq, MPtr dk, MPtr dv, float* ws, int B, const BwdDims& p, hipStream_t st) {
  constexpr int LD = DP + 4;
  constexpr int lds_stats = 2 * kT * LD * 4, lds_dq = (4 * kT * LD + kT * 33) * 4, lds_dkv = (4 * kT * LD + 2 * kT * 33) * 4;
  static StllmPerDevice attr_dev;   // the dynamic-LDS opt-in is a per-device attribute
  bool attr_first;
  const int attr_d = attr_dev.enter(&attr_first);
  if (attr_first) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dq_kernel<T, DP>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_dq) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkv_kernel<T, DP>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_dkv) != hipSuccess) {
      stllm_set_error("stllm_attention_bwd: cannot raise the dynamic LDS limit");
      return STLLM_ERR_HIP;
    }
    attr_dev.done(attr_d);
  }
  const dim3 gq((p.Sq + kT - 1) / kT, p.H, B), gk((p.Skv + kT - 1) / kT, p.H, B), block(256);
  hipLaunchKernelGGL((attn_bwd_stats_kernel<T, DP>), gq, block, lds_stats, st, q, k, o, dO, ws, p);
  hipLaunchKernelGGL((attn_bwd_dq_kernel<T, DP>), gq, block, lds_dq, st, q, k, v, dO, dq, ws, p);
  hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, DP>), gk, block, lds_dkv, st, q, k, v, dO, dk, dv, ws, p);
  return STLLM_OK;
}
template <typename T>
int launch_bwd_dp(int D, Ptr q, Ptr k, Ptr v, Ptr o, Ptr dO, MPtr dq, MPtr dk, MPtr dv, float* ws, int B, const BwdDims& p, hipStream_t st) {
  if (D <= 64) return launch_bwd<T, 64>(q, k, v, o, dO, dq, dk, dv, ws, B, p, st);
  if (D <= 96) return launch_bwd<T, 96>(q, k, v, o, dO, dq, dk, dv, ws, B, p, st);
  return launch_bwd<T, 128>(q, k, v, o, dO, dq, dk, dv, ws, B, p, st);
}

}  // namespace

extern "C" int64_t stllm_attention_bwd_workspace_bytes(int B, int H, int Sq) { return (int64_t)B * H * Sq * 2 * 4; }

extern "C" int stllm_attention_bwd(int dtype, const void* q, int64_t q_bs, int64_t q_rs, const void* k, int64_t k_bs, int64_t k_rs, const void* v,
                                   int64_t v_bs, int64_t v_rs, const void* o, int64_t o_bs, int64_t o_rs, const void* dO, int64_t do_bs,
                                   int64_t do_rs, void* dq, int64_t dq_bs, int64_t dq_rs, void* dk, int64_t dk_bs, int64_t dk_rs, void* dv,
                                   int64_t dv_bs, int64_t dv_rs, int B, int H, int Sq, int Skv, int D, float scale, int causal,
                                   const int32_t* kv_len, void* workspace, int64_t workspace_bytes, void* stream) {
  STLLM_CHECK_ARG(q && k && v && o && dO && dq && dk && dv && B > 0 && H > 0 && Sq > 0 && Skv > 0, "stllm_attention_bwd: bad args");
  STLLM_CHECK_ARG(D > 0 && D <= 128 && D % 8 == 0, "stllm_attention_bwd: head_dim %d unsupported (multiples of 8 up to 128)", D);
  STLLM_CHECK_ARG(!causal || Sq == Skv, "stllm_attention_bwd: causal needs Sq == Skv");
  STLLM_CHECK_ARG(workspace && workspace_bytes >= stllm_attention_bwd_workspace_bytes(B, H, Sq), "stllm_attention_bwd: workspace too small");
  const int eb = dtype == STLLM_F32 ? 4 : 2;
  const void* ps[8] = {q, k, v, o, dO, dq, dk, dv};
  const int64_t st_[16] = {q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs, do_bs, do_rs, dq_bs, dq_rs, dk_bs, dk_rs, dv_bs, dv_rs};
  for (int i = 0; i < 8; ++i)
    STLLM_CHECK_ARG(aligned16(ps[i]) && (st_[2 * i] * eb) % 16 == 0 && (st_[2 * i + 1] * eb) % 16 == 0,
                    "stllm_attention_bwd: operand %d must be 16-byte aligned with 16-byte-multiple strides", i);
  const Ptr Q{q, q_bs, q_rs}, K{k, k_bs, k_rs}, V{v, v_bs, v_rs}, O{o, o_bs, o_rs}, DO{dO, do_bs, do_rs};
  const MPtr DQ{dq, dq_bs, dq_rs}, DK{dk, dk_bs, dk_rs}, DV{dv, dv_bs, dv_rs};
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  float* ws = reinterpret_cast<float*>(workspace);
  const BwdDims dims{H, Sq, Skv, D, scale, causal, kv_len};
  int rc;
  const bool mfma = stllm_options().attn_bwd_valu != 1;      // 16-bit operands: MFMA kernels; fp32 always runs the fp32-FMA kernels
  switch (dtype) {
    case STLLM_BF16: rc = mfma ? launch_bwd_mfma<bf16_t>(Q, K, V, O, DO, DQ, DK, DV, ws, B, H, Sq, Skv, D, scale, causal, kv_len, s)
                               : launch_bwd_dp<bf16_t>(D, Q, K, V, O, DO, DQ, DK, DV, ws, B, dims, s); break;
    case STLLM_F16: rc = mfma ? launch_bwd_mfma<f16_t>(Q, K, V, O, DO, DQ, DK, DV, ws, B, H, Sq, Skv, D, scale, causal, kv_len, s)
                              : launch_bwd_dp<f16_t>(D, Q, K, V, O, DO, DQ, DK, DV, ws, B, dims, s); break;
    case STLLM_F32: rc = launch_bwd_dp<float>(D, Q, K, V, O, DO, DQ, DK, DV, ws, B, dims, s); break;
    default: stllm_set_error("stllm_attention_bwd: bad dtype %d", dtype); return STLLM_ERR_BAD_DTYPE;
  }
  if (rc != STLLM_OK) return rc;
  STLLM_CHECK_LAUNCH("stllm_attention_bwd");
  return STLLM_OK;
}
