// Fused attention  out = softmax(scale * Q K^T + mask) V   (never materialises S)
//
// 16-bit path (bf16/fp16, MFMA 32x32x16, fp32 online softmax) — "everything transposed" layout:
//   * a wave owns 32 query rows; a workgroup = NW waves = 32*NW rows of one (batch, head).
//   * per 32-key tile:  S^T[kv][q] = K_tile[kv][d] . Q^T[d][q]      (A = K from LDS, B = Q in VGPRs)
//     so every lane holds 16 scores of ONE query (q = lane & 31): row max / row sum are in-lane
//     reductions plus a single lane<->lane+32 exchange (wave shuffles), no LDS round trip.
//   * P^T stays in registers and is already in B-operand form for
//                         O^T[d][q] += V^T[d][kv] . P^T[kv][q]       (A = V^T from LDS)
//     the key<->k-slot map of the MFMA is free as long as A and B agree, so the S^T accumulator
//     register order {0-3,8-11 | 4-7,12-15}+16a is used as the contraction order directly: V^T
//     fragments are two 8-byte LDS reads at [d][16a+4h] and [d][16a+8+4h].
//   * O^T columns are queries too, so the online-softmax rescale is an in-lane multiply.
//   * K tile rows are padded by 16 B and V^T rows by 8 B: conflict-free ds_read_b128 / ds_read_b64.
//   * head_dim 88 (EVA-CLIP-g) is zero-padded to 96 in LDS/registers only; HBM traffic stays 88.
// fp32 path ("verify" numerics): attn_mfma_f32_kernel — the same schedule on exact-fp32 MFMAs (32x32x2) — from 8 query rows on; an
// exact-fp32 vector kernel, one wave per query row, for the one-row decode step.
#include <cstdlib>
#include "common.h"

namespace {

struct AttnParams {
  const char* q; int64_t q_bs, q_rs;   // strides in elements
  const char* k; int64_t k_bs, k_rs;
  const char* v; int64_t v_bs, v_rs;
  char* o; int64_t o_bs, o_rs;
  int B, H, Sq, Skv, D;
  float scale_log2;                     // scale * log2(e)
  float scale;
  int causal;
  const int32_t* kv_len;
  int q_lds;                            // attn_dma_kernel: the query tiles arrive through the LDS (set by launch_dma when they fit)
  int merge_par;                        // attn_dma_kernel, KS2 > 1: every partner wave has its own merge slot (one barrier pair instead of one per partner)
};

constexpr float kNeg = -1.0e30f;

// in-kernel timeline (tools/attn_trace.sh builds st-llm_amd/attn_trace/libstllm_hip.so with -DSTLLM_ATTN_TRACE; never in the shipped library):
// lane 0 of every wave stamps s_memtime into 8 slots of g_attn_trace[workgroup][wave]; tools/attn_trace.py reads them back.
#ifdef STLLM_ATTN_TRACE
__device__ unsigned long long g_attn_trace[512 * 12 * 8];
#define ATTN_STAMP(slot) do { if ((threadIdx.x & 63) == 0) g_attn_trace[((size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 12 + (threadIdx.x >> 6)) * 8 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#define ATTN_STAMP_VAL(slot, val) do { if ((threadIdx.x & 63) == 0) g_attn_trace[((size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 12 + (threadIdx.x >> 6)) * 8 + (slot)] = (unsigned long long)(val); } while (0)
#else
#define ATTN_STAMP(slot) do { } while (0)
#define ATTN_STAMP_VAL(slot, val) do { } while (0)
#endif

// XCD-aware (batch, head, query-chunk) assignment.  The hardware deals workgroups round-robin over the 8 XCDs, each with its own
// L2; with the natural order neighbouring heads of a frame (whose 176-byte K / V rows share cache lines at head_dim 88) and the
// query chunks of one head (which re-read the same K / V) land on DIFFERENT XCDs, and every line is fetched 2-5 times from the
// fabric — the ViT attention spent 38 % of its 26 us waiting for its first 36 KB (timeline: tools/attn_probe.hip).  The bijective
// remap makes the linear ids that one XCD sees consecutive: all chunks of a head, then the next head of the same (frame, batch).
__device__ __forceinline__ void attn_block_coords(int& chunk_x, int& h, int& b) {
  const int gx = gridDim.x, gy = gridDim.y, n = gx * gy * gridDim.z;
  const int lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
  const int q = n >> 3, r = n & 7, xcd = lin & 7, k = lin >> 3;
  const int id = ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  chunk_x = id % gx;
  h = (id / gx) % gy;
  b = id / (gx * gy);
}

template <typename T, int DP, int NW>
__global__ __launch_bounds__(64 * NW) void attn_mfma_kernel(const AttnParams p) {
  constexpr int KS = DP / 16;          // k-steps of the S^T MFMA chain
  constexpr int DB = DP / 32;          // 32-row blocks of O^T
  constexpr int KPITCH = DP * 2 + 16;  // bytes per K row in LDS
  constexpr int VPITCH = 32 * 2 + 8;   // bytes per V^T row in LDS
  constexpr int NT = 64 * NW;
  constexpr int CPR = DP / 8;          // 16-byte chunks per K/V row
  constexpr int NCH = 32 * CPR;        // chunks per tile
  constexpr int CPT = (NCH + NT - 1) / NT;
  __shared__ __attribute__((aligned(16))) char k_lds[32 * KPITCH];
  __shared__ __attribute__((aligned(16))) char v_lds[DP * VPITCH];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  const int q_blk0 = blockIdx.x * (32 * NW);
  const int qrow = q_blk0 + wave * 32 + li;  // this lane's query
  const int D = p.D;
  const int kvlen = p.kv_len ? min(p.kv_len[b], p.Skv) : p.Skv;

  // ---- Q fragments (B operand of S^T): lane holds Q[q][ks*16 + lh*8 .. +8] ------------------------
  i32x4 qf[KS];
  {
    const char* qp = p.q + ((int64_t)b * p.q_bs + (int64_t)(qrow < p.Sq ? qrow : 0) * p.q_rs + (int64_t)h * D) * 2;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int d0 = ks * 16 + lh * 8;
      i32x4 z = {0, 0, 0, 0};
      qf[ks] = (qrow < p.Sq && d0 < D) ? *reinterpret_cast<const i32x4*>(qp + d0 * 2) : z;
    }
  }

  f32x16 o[DB];
#pragma unroll
  for (int i = 0; i < DB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[i][r] = 0.0f;
  float m_run = kNeg, l_run = 0.0f;

  int kv_end = kvlen;
  if (p.causal) kv_end = min(kv_end, q_blk0 + 32 * NW);  // keys beyond the last query of the block are masked
  const char* kbase = p.k + ((int64_t)b * p.k_bs + (int64_t)h * D) * 2;
  const char* vbase = p.v + ((int64_t)b * p.v_bs + (int64_t)h * D) * 2;

  // K/V tiles are register-staged one tile AHEAD: the global loads of tile t+1 are issued before the MFMA/softmax work of
  // tile t and only consumed (written to LDS) at the top of the next iteration — HBM/L2 latency hides under compute.
  i32x4 kreg[CPT], vreg[CPT];
  auto load_tile = [&](int kv0) {
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
      const int ch = tid + c * NT;
      const int cc = ch >> 5, row = ch & 31;   // consecutive lanes = consecutive keys: conflict-free transposed V writes
      const int kv = kv0 + row;
      const bool ok = (ch < NCH) && (kv < p.Skv) && (cc * 8 < D);
      i32x4 z = {0, 0, 0, 0};
      kreg[c] = ok ? *reinterpret_cast<const i32x4*>(kbase + ((int64_t)kv * p.k_rs + cc * 8) * 2) : z;
      vreg[c] = ok ? *reinterpret_cast<const i32x4*>(vbase + ((int64_t)kv * p.v_rs + cc * 8) * 2) : z;
    }
  };
  const int wave_q_last = q_blk0 + wave * 32 + 31;  // last query row of this wave (causal tile skipping)
  if (kv_end > 0) load_tile(0);
  for (int kv0 = 0; kv0 < kv_end; kv0 += 32) {
    __syncthreads();  // previous tile fully consumed
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
      const int ch = tid + c * NT;
      if (ch < NCH) {
        const int cc = ch >> 5, row = ch & 31;
        *reinterpret_cast<i32x4*>(k_lds + row * KPITCH + cc * 16) = kreg[c];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const uint32_t w = (uint32_t)vreg[c][e];
          *reinterpret_cast<uint16_t*>(v_lds + (cc * 8 + 2 * e) * VPITCH + row * 2) = (uint16_t)(w & 0xffff);
          *reinterpret_cast<uint16_t*>(v_lds + (cc * 8 + 2 * e + 1) * VPITCH + row * 2) = (uint16_t)(w >> 16);
        }
      }
    }
    __syncthreads();
    if (kv0 + 32 < kv_end) load_tile(kv0 + 32);
    if (p.causal && kv0 > wave_q_last) continue;  // every key of this tile is in the future of every query of this wave

    // ---- S^T = K . Q^T ------------------------------------------------------------------------------
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.0f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const i32x4 kf = *reinterpret_cast<const i32x4*>(k_lds + li * KPITCH + (ks * 2 + lh) * 16);
      s = Elem<T>::mfma(kf, qf[ks], s);
    }
    // ---- mask + online softmax (per-lane query); scores stay RAW, the scale is folded into the exp2 argument ----
    const bool need_mask = (kv0 + 32 > kvlen) || (p.causal && kv0 + 31 > q_blk0 + wave * 32);  // wave-uniform
    if (need_mask) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kv = kv0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const bool dead = (kv >= kvlen) || (p.causal && kv > qrow);
        s[r] = dead ? kNeg : s[r];
      }
    }
    float mx = s[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);               // running max of RAW scores (scale > 0)
    const float mc = m_new * p.scale_log2;
    float rs = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float pv = __builtin_amdgcn_exp2f(fmaf(s[r], p.scale_log2, -mc));   // masked: exp2(-huge) == 0
      s[r] = pv;
      rs += pv;
    }
    rs += __shfl_xor(rs, 32, 64);
    if (__any(m_new != m_run)) {   // rescale only when some query's running max moved (wave-uniform branch)
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * p.scale_log2);
      l_run *= alpha;
#pragma unroll
      for (int i = 0; i < DB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
      m_run = m_new;
    }
    l_run += rs;

    // ---- O^T += V^T . P^T ----------------------------------------------------------------------------
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      i32x4 pf;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        pf[e] = (int)(Elem<T>::pack2(s[a * 8 + 2 * e], s[a * 8 + 2 * e + 1]));
#pragma unroll
      for (int i = 0; i < DB; ++i) {
        const char* vp = v_lds + (i * 32 + li) * VPITCH + (16 * a + 4 * lh) * 2;
        const i32x2 lo = *reinterpret_cast<const i32x2*>(vp);
        const i32x2 hi = *reinterpret_cast<const i32x2*>(vp + 16);
        const i32x4 vf = {lo[0], lo[1], hi[0], hi[1]};
        o[i] = Elem<T>::mfma(vf, pf, o[i]);
      }
    }
  }

  // ---- normalise and store: lane holds O[q][d = i*32 + (r&3) + 8*(r>>2) + 4*lh] ------------------------
  if (qrow < p.Sq) {
    const float inv = 1.0f / l_run;
    uint16_t* op = reinterpret_cast<uint16_t*>(p.o) + (int64_t)b * p.o_bs + (int64_t)qrow * p.o_rs + (int64_t)h * D;
#pragma unroll
    for (int i = 0; i < DB; ++i) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = i * 32 + 8 * g + 4 * lh;
        if (d0 < D) {  // D % 4 == 0
          uint2 pk;
          pk.x = Elem<T>::pack2(o[i][4 * g + 0] * inv, o[i][4 * g + 1] * inv);
          pk.y = Elem<T>::pack2(o[i][4 * g + 2] * inv, o[i][4 * g + 3] * inv);
          *reinterpret_cast<uint2*>(op + d0) = pk;
        }
      }
    }
  }
}

// ---- resident-K/V variant for short key sequences (ViT: 257 keys, Q-Former cross-attention) ----------------
// The whole K (row-major, padded rows) and V^T of one (batch, head) are staged into LDS ONCE (<= 116 KiB for 288 keys x
// 96 dims), one barrier, then every wave walks all key tiles for its own 32 query rows with NO further barriers: waves
// run decoupled, so LDS/MFMA/VALU latencies of one wave hide under the others (PMC on the tiled kernel: 55 % of wave
// time parked at barriers/waits).  One workgroup per (batch, head): 16 frames x 16 heads = 256 workgroups = one per CU.
// KS2 = 2 / 4: every query tile is worked on by KS2 waves that take every KS2-th 32-key tile (the causal prefill has few, long
// query tiles: 32 heads x 18 tiles walk up to 18 key tiles each — the serial walk, not the FLOPs, sets the time); the
// partial (m, l, O) states are merged through LDS at the end.
template <typename T, int DP, int SKV_MAX, int KS2 = 1>
__global__ __launch_bounds__(768) void attn_resident_kernel(const AttnParams p) {
  // SKV_MAX = keys staged per pass (the "window").  Keys beyond one window are handled by further passes (one pair of
  // barriers per window instead of per 32-key tile); blockIdx.x selects a chunk of blockDim.x/64 query tiles.
  constexpr int KS = DP / 16, DB = DP / 32;
  constexpr int KPITCH = DP * 2 + 16;
  constexpr int VPITCH = SKV_MAX * 2 + 8;
  constexpr int CPR = DP / 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* k_lds = smem;
  char* v_lds = smem + SKV_MAX * KPITCH;

  const int tid = threadIdx.x, lane = tid & 63, wave_all = tid >> 6, nwaves = (blockDim.x >> 6) / KS2;
  const int wave = wave_all % nwaves, kpar = wave_all / nwaves;   // query tile within the chunk, key-tile parity (KS2 = 2)
  const int li = lane & 31, lh = lane >> 5;
  int bx, b, h;
  attn_block_coords(bx, h, b);
  const int D = p.D;
  const int kvlen = p.kv_len ? min(p.kv_len[b], p.Skv) : p.Skv;
  const char* kbase = p.k + ((int64_t)b * p.k_bs + (int64_t)h * D) * 2;
  const char* vbase = p.v + ((int64_t)b * p.v_bs + (int64_t)h * D) * 2;
  // heavy (late, causal) chunks first: the chunk index counts down
  const int chunk = (int)gridDim.x - 1 - bx;
  const int qt = chunk * nwaves + wave;                 // this wave's query tile
  const bool q_live = qt * 32 < p.Sq;
  const int qrow = qt * 32 + li;
  unsigned long long t_wait = 0;
  (void)t_wait;
  int kv_block_end = kvlen;                             // keys any wave of this block can see
  if (p.causal) kv_block_end = min(kv_block_end, (chunk + 1) * nwaves * 32);

  i32x4 qf[KS];
  {
    const char* qp = p.q + ((int64_t)b * p.q_bs + (int64_t)(qrow < p.Sq ? qrow : 0) * p.q_rs + (int64_t)h * D) * 2;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int d0 = ks * 16 + lh * 8;
      i32x4 z = {0, 0, 0, 0};
      qf[ks] = (qrow < p.Sq && d0 < D) ? *reinterpret_cast<const i32x4*>(qp + d0 * 2) : z;
    }
  }
  f32x16 o[DB];
#pragma unroll
  for (int i = 0; i < DB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[i][r] = 0.0f;
  float m_run = kNeg, l_run = 0.0f;

  for (int win0 = 0; win0 < kv_block_end; win0 += SKV_MAX) {
    const int win_keys = min(SKV_MAX, kv_block_end - win0);
    const int n_tiles = (win_keys + 31) >> 5;
    if (win0 > 0) __syncthreads();   // previous window fully consumed
    // ---- stage this window's K and V^T (rows >= Skv and dims >= D are zero) ------------------------------
    // kStageUn chunks per thread are requested before the first one is written: the staging pass pays ONE memory round trip per
    // group instead of one per chunk (ViT: 3456 chunks of K and of V over 576 threads = 6 per thread)
    constexpr int kStageUn = 3;
    const int n_chunks = n_tiles * 32 * CPR;
    for (int ch0 = tid; ch0 < n_chunks; ch0 += blockDim.x * kStageUn) {
      i32x4 kq[kStageUn], vq[kStageUn];
      int lrow_[kStageUn], cc_[kStageUn];
#pragma unroll
      for (int u = 0; u < kStageUn; ++u) {
        const int ch = ch0 + u * (int)blockDim.x;
        const int chc = ch < n_chunks ? ch : n_chunks - 1;
        const int t = chc / (32 * CPR), rem = chc - t * (32 * CPR);
        const int cc = rem >> 5, lrow = t * 32 + (rem & 31);
        const int row = win0 + lrow;
        const bool ok = (ch < n_chunks) && (row < p.Skv) && (cc * 8 < D);
        const i32x4 z = {0, 0, 0, 0};
        kq[u] = ok ? *reinterpret_cast<const i32x4*>(kbase + ((int64_t)row * p.k_rs + cc * 8) * 2) : z;
        vq[u] = ok ? *reinterpret_cast<const i32x4*>(vbase + ((int64_t)row * p.v_rs + cc * 8) * 2) : z;
        lrow_[u] = lrow;
        cc_[u] = cc;
      }
#pragma unroll
      for (int u = 0; u < kStageUn; ++u) {
        if (ch0 + u * (int)blockDim.x >= n_chunks) break;
        const int lrow = lrow_[u], cc = cc_[u];
        *reinterpret_cast<i32x4*>(k_lds + lrow * KPITCH + cc * 16) = kq[u];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const uint32_t w = (uint32_t)vq[u][e];
          *reinterpret_cast<uint16_t*>(v_lds + (cc * 8 + 2 * e) * VPITCH + lrow * 2) = (uint16_t)(w & 0xffff);
          *reinterpret_cast<uint16_t*>(v_lds + (cc * 8 + 2 * e + 1) * VPITCH + lrow * 2) = (uint16_t)(w >> 16);
        }
      }
    }
    __syncthreads();
    if (!q_live) continue;
    {
    int t_end = n_tiles;
    if (p.causal) t_end = min(n_tiles, qt + 1 - (win0 >> 5));   // tiles up to this wave's diagonal
    for (int t = 0; t < t_end; ++t) {
      if (KS2 > 1 && (((win0 >> 5) + t) & (KS2 - 1)) != kpar) continue;   // the partner wave takes this key tile
      const int kv0 = win0 + t * 32;   // global key index of the tile; LDS rows are window-relative
      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.0f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const i32x4 kf = *reinterpret_cast<const i32x4*>(k_lds + (t * 32 + li) * KPITCH + (ks * 2 + lh) * 16);
        s = Elem<T>::mfma(kf, qf[ks], s);
      }
      const bool need_mask = (kv0 + 32 > kvlen) || (p.causal && kv0 + 31 > qt * 32);
      if (need_mask) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kv = kv0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          const bool dead = (kv >= kvlen) || (p.causal && kv > qrow);
          s[r] = dead ? kNeg : s[r];
        }
      }
      float mx = s[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run, mx);
      const float mc = m_new * p.scale_log2;
      float rs = 0.0f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = __builtin_amdgcn_exp2f(fmaf(s[r], p.scale_log2, -mc));
        s[r] = pv;
        rs += pv;
      }
      rs += __shfl_xor(rs, 32, 64);
      if (__any(m_new != m_run)) {
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * p.scale_log2);
        l_run *= alpha;
#pragma unroll
        for (int i = 0; i < DB; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
        m_run = m_new;
      }
      l_run += rs;
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        i32x4 pf;
#pragma unroll
        for (int e = 0; e < 4; ++e) pf[e] = (int)(Elem<T>::pack2(s[a * 8 + 2 * e], s[a * 8 + 2 * e + 1]));
#pragma unroll
        for (int i = 0; i < DB; ++i) {
          const char* vp = v_lds + (i * 32 + li) * VPITCH + (t * 32 + 16 * a + 4 * lh) * 2;
          const i32x2 lo = *reinterpret_cast<const i32x2*>(vp);
          const i32x2 hi = *reinterpret_cast<const i32x2*>(vp + 16);
          const i32x4 vf = {lo[0], lo[1], hi[0], hi[1]};
          o[i] = Elem<T>::mfma(vf, pf, o[i]);
        }
      }
    }
    }
  }
  if constexpr (KS2 > 1) {
    // merge the two key-parity partials of every query tile: parity 1 parks (m, l, O) in LDS (the K/V windows are dead),
    // parity 0 combines:  m = max(m0, m1);  O = O0 * 2^((m0-m)*c) + O1 * 2^((m1-m)*c);  l likewise
    float* mbuf = reinterpret_cast<float*>(smem);
    for (int pp = 1; pp < KS2; ++pp) {   // one partner at a time: the merge buffer has to fit the (dead) K/V windows
      __syncthreads();
      if (kpar == pp) {
#pragma unroll
        for (int i = 0; i < DB; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) mbuf[((wave * (DB * 16 + 2) + i * 16 + r) << 6) + lane] = o[i][r];
        mbuf[((wave * (DB * 16 + 2) + DB * 16) << 6) + lane] = m_run;
        mbuf[((wave * (DB * 16 + 2) + DB * 16 + 1) << 6) + lane] = l_run;
      }
      __syncthreads();
      if (kpar == 0) {
        const float m1 = mbuf[((wave * (DB * 16 + 2) + DB * 16) << 6) + lane];
        const float l1 = mbuf[((wave * (DB * 16 + 2) + DB * 16 + 1) << 6) + lane];
        const float m = fmaxf(m_run, m1);
        const float a0 = __builtin_amdgcn_exp2f((m_run - m) * p.scale_log2), a1 = __builtin_amdgcn_exp2f((m1 - m) * p.scale_log2);
        l_run = l_run * a0 + l1 * a1;
        m_run = m;
#pragma unroll
        for (int i = 0; i < DB; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[i][r] = o[i][r] * a0 + mbuf[((wave * (DB * 16 + 2) + i * 16 + r) << 6) + lane] * a1;
      }
    }
    if (kpar != 0) return;
  }
  if (q_live && qrow < p.Sq) {
    const float inv = 1.0f / l_run;
    uint16_t* op = reinterpret_cast<uint16_t*>(p.o) + (int64_t)b * p.o_bs + (int64_t)qrow * p.o_rs + (int64_t)h * D;
#pragma unroll
    for (int i = 0; i < DB; ++i) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = i * 32 + 8 * g + 4 * lh;
        if (d0 < D) {
          uint2 pk;
          pk.x = Elem<T>::pack2(o[i][4 * g + 0] * inv, o[i][4 * g + 1] * inv);
          pk.y = Elem<T>::pack2(o[i][4 * g + 2] * inv, o[i][4 * g + 3] * inv);
          *reinterpret_cast<uint2*>(op + d0) = pk;
        }
      }
    }
  }
}

template <typename T, int DP, int SKV_MAX, int KS2 = 1>
int launch_resident(const AttnParams& p, hipStream_t stream, int nw_req = 0) {
  constexpr int lds = SKV_MAX * (DP * 2 + 16) + DP * (SKV_MAX * 2 + 8);

  static StllmPerDevice attr_dev;   // the dynamic-LDS opt-in is a per-device attribute
  bool attr_first;
  const int attr_d = attr_dev.enter(&attr_first);
  if (attr_first) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_resident_kernel<T, DP, SKV_MAX, KS2>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_dev.done(attr_d);
  }
  const int q_tiles = (p.Sq + 31) / 32;
  int nw = nw_req > 0 ? nw_req : q_tiles;
  if (nw > 12) nw = 12;
  if (nw > q_tiles) nw = q_tiles;
  if (KS2 > 1 && nw * (DP / 32 * 16 + 2) * 256 > lds) {   // merge buffer (one parked (m, l, O) state per query tile) reuses the K/V windows
    stllm_set_error("stllm_attention: key-split merge buffer does not fit (nw=%d)", nw);
    return STLLM_ERR_UNSUPPORTED;
  }
  dim3 grid((q_tiles + nw - 1) / nw, p.H, p.B), block(64 * nw * KS2);
  hipLaunchKernelGGL((attn_resident_kernel<T, DP, SKV_MAX, KS2>), grid, block, lds, stream, p);
  STLLM_CHECK_LAUNCH("stllm_attention(resident)");
  return STLLM_OK;
}

// ---- D = 128 prefill (Llama), round 2: windows staged by LDS-DMA, V read through the hardware transpose ----------------------------
// Same work split as attn_resident_kernel<T, 128, 128, 4> (nw query tiles per workgroup, KS2 waves per query tile taking every
// KS2-th 32-key tile, 128-key windows, LDS merge at the end), but the staging no longer passes through registers:
//   * K and V rows of a window go global -> LDS with global_load_lds_dwordx4 (1 KiB = 4 rows per instruction, asynchronous) into
//     ROW-MAJOR images; window w + 1 is requested right after the barrier that starts window w and lands during its MFMAs:
//     ONE barrier per window, no ds_write, no V^T scatter (the old pass: 24 ds_write_b16 per thread and window);
//   * bank conflicts are avoided on the SOURCE side: LDS slot (row, 16-byte chunk c) holds logical chunk c ^ (row & 15) for K
//     (ds_read_b128 of 16 consecutive rows -> 16 different chunks) and c ^ ((row & 7) << 1) for V;
//   * V^T fragments come from ds_read_b64_tr_b16: in a 16-lane group lane i fetches 8 bytes of row (key) base + i / 4 at column
//     piece i % 4 and RECEIVES the four keys base .. base + 3 of column 4 (i / 4) + i % 4 = i — measured with tools/tr_probe.hip.
// two transposing reads, at p and p + OFF bytes (8 rows of the image further down); see the layout note above
template <int OFF = 2048>
__device__ __forceinline__ void lds_read_tr_pair(const char* p, unsigned long long& lo, unsigned long long& hi) {
  const unsigned a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
  asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:%3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(lo), "=&v"(hi) : "v"(a), "n"(OFF) : "memory");
}

// three pairs (d-blocks 0, 1, 2 of an UNSWIZZLED row: 64 bytes apart) with a single wait
template <int OFF>
__device__ __forceinline__ void lds_read_tr_3pairs(const char* p, unsigned long long& l0, unsigned long long& h0, unsigned long long& l1,
                                                   unsigned long long& h1, unsigned long long& l2, unsigned long long& h2) {
  const unsigned a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
  asm volatile("ds_read_b64_tr_b16 %0, %6\n\tds_read_b64_tr_b16 %1, %6 offset:%7\n\t"
               "ds_read_b64_tr_b16 %2, %6 offset:64\n\tds_read_b64_tr_b16 %3, %6 offset:%8\n\t"
               "ds_read_b64_tr_b16 %4, %6 offset:128\n\tds_read_b64_tr_b16 %5, %6 offset:%9\n\ts_waitcnt lgkmcnt(0)"
               : "=&v"(l0), "=&v"(h0), "=&v"(l1), "=&v"(h1), "=&v"(l2), "=&v"(h2) : "v"(a), "n"(OFF), "n"(OFF + 64), "n"(OFF + 128) : "memory");
}

// NT: the launch's thread bound.  768 (12 waves: 3 per SIMD) caps the kernel at 168 registers and costs it 2-3 spilled ones; the prefill's 2 x 4 split
// (8 waves) is instantiated with 512: 256 registers, no scratch (round 6)
// four pairs at four addresses (the d-blocks of a SWIZZLED 256-byte row), ONE wait: eight reads in flight instead of four serialised round trips per 16-key step
template <int OFF>
__device__ __forceinline__ void lds_read_tr_4pairs(const char* p0, const char* p1, const char* p2, const char* p3, unsigned long long (&lo)[4], unsigned long long (&hi)[4]) {
  typedef __attribute__((address_space(3))) const char* lp;
  const unsigned a0 = (unsigned)(uintptr_t)(lp)p0, a1 = (unsigned)(uintptr_t)(lp)p1, a2 = (unsigned)(uintptr_t)(lp)p2, a3 = (unsigned)(uintptr_t)(lp)p3;
  asm volatile("ds_read_b64_tr_b16 %0, %8\n\tds_read_b64_tr_b16 %1, %8 offset:%12\n\t"
               "ds_read_b64_tr_b16 %2, %9\n\tds_read_b64_tr_b16 %3, %9 offset:%12\n\t"
               "ds_read_b64_tr_b16 %4, %10\n\tds_read_b64_tr_b16 %5, %10 offset:%12\n\t"
               "ds_read_b64_tr_b16 %6, %11\n\tds_read_b64_tr_b16 %7, %11 offset:%12\n\ts_waitcnt lgkmcnt(0)"
               : "=&v"(lo[0]), "=&v"(hi[0]), "=&v"(lo[1]), "=&v"(hi[1]), "=&v"(lo[2]), "=&v"(hi[2]), "=&v"(lo[3]), "=&v"(hi[3])
               : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "n"(OFF) : "memory");
}

template <typename T, int KS2, int DP = 128, int NT = 768>
__global__ __launch_bounds__(NT) void attn_dma_kernel(const AttnParams p) {
  ATTN_STAMP(0);
  // DP = 128: Llama (256-byte rows, 128-key windows); DP = 96: head_dim 88 of the EVA ViT (rows of 11 chunks in a 12-chunk = 192-byte
  // pitch, chunk 11 a copy of chunk 10 that meets zero-padded Q / unstored columns; 96-key windows, two workgroups per CU)
  constexpr int KS = DP / 16, DB = DP / 32, CH = DP / 8, PITCH = CH * 16, W = DP == 128 ? 128 : 96;
  constexpr int kPieces = W * CH / 64;              // 1-KiB DMA pieces per image (32 / 18)
  constexpr int kImg = W * PITCH;                   // one K or V window image
  constexpr int kBuf = 2 * kImg;                   // [K | V]
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, nwv = blockDim.x >> 6, nwaves = nwv / KS2;
  const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave = wave_all % nwaves, kpar = wave_all / nwaves;
  const int li = lane & 31, lh = lane >> 5;
  int bx, b, h;
  attn_block_coords(bx, h, b);
  const int kvlen = p.kv_len ? min(p.kv_len[b], p.Skv) : p.Skv;
  const int D = p.D;
  const char* kbase = p.k + ((int64_t)b * p.k_bs + (int64_t)h * D) * 2;
  const char* vbase = p.v + ((int64_t)b * p.v_bs + (int64_t)h * D) * 2;
  auto swk = [](int row) { return DP == 128 ? (row & 15) : ((row >> 2) & 3); };          // K: 16-byte chunk XOR (bank spread of ds_read_b128 over 16 rows)
  auto swv = [](int row) { return DP == 128 ? ((row & 7) << 1) : 0; };                    // V: for the transposing reads (192-byte pitch needs none)
  const int chunk = (int)gridDim.x - 1 - bx;   // heavy (late, causal) chunks first
  const int qt = chunk * nwaves + wave;
  const bool q_live = qt * 32 < p.Sq;
  const int qrow = qt * 32 + li;
  unsigned long long t_wait = 0;
  (void)t_wait;
  int kv_block_end = kvlen;
  if (p.causal) kv_block_end = min(kv_block_end, (chunk + 1) * nwaves * 32);

  // The query fragments.  Straight from global memory every load instruction touches 32 rows = 32 cache lines for 32 bytes each, and the KS2 waves
  // of a tile fetch the same fragments: 8 x 32 line look-ups per wave at ~4.75 cycles each sat in FRONT of the first K / V window in the CU's
  // memory pipeline (timeline, tools/attn_trace.py: 7.7 k cycles from the start of a wave to its last request, profiles/r06_attn_trace.md).
  // p.q_lds: the tile's 32 rows arrive ONCE, row-contiguous, by LDS-DMA (CH / 2 one-KiB pieces shared by the tile's KS2 waves, K's chunk swizzle)
  // in an image behind the window buffers and are read as fragments after the first barrier.
  i32x4 qf[KS];
  if (!p.q_lds) {
    const char* qp = p.q + ((int64_t)b * p.q_bs + (int64_t)(qrow < p.Sq ? qrow : 0) * p.q_rs + (int64_t)h * D) * 2;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int d0 = ks * 16 + lh * 8;
      i32x4 z = {0, 0, 0, 0};
      qf[ks] = (qrow < p.Sq && d0 < D) ? *reinterpret_cast<const i32x4*>(qp + d0 * 2) : z;
    }
  }
  f32x16 o[DB];
#pragma unroll
  for (int i = 0; i < DB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[i][r] = 0.0f;
  float m_run = kNeg, l_run = 0.0f;
  char* qimg = smem + 2 * kBuf + wave * (32 * PITCH);
  if (p.q_lds) {
    const char* qb = p.q + ((int64_t)b * p.q_bs + (int64_t)h * D) * 2;
    const int lastc = (D * 2 + 15) / 16 - 1;
    for (int pc = kpar; pc < CH / 2; pc += KS2) {
      const int idx = pc * 64 + lane;
      const int r = idx / CH, pch = idx - r * CH;
      int lc = pch ^ swk(r);
      lc = lc < lastc ? lc : lastc;
      int row = qt * 32 + r;
      row = row < p.Sq ? row : p.Sq - 1;                            // rows past the end: a valid row, never stored
      glds16(qb + (int64_t)row * p.q_rs * 2 + lc * 16, qimg + (idx - lane) * 16);
    }
  }

  // ---- window DMA: 2 x kPieces pieces of 1 KiB per window (K image, then V image); piece pc is issued by wave pc % nwv -------------
  const int last_chunk = (D * 2 + 15) / 16 - 1;
  auto issue_window = [&](int win0, int buf) {
    for (int pc = wave_all; pc < 2 * kPieces; pc += nwv) {
      const bool is_v = pc >= kPieces;
      const int idx = (is_v ? pc - kPieces : pc) * 64 + lane;      // physical 16-byte slot of the image
      const int r4 = idx / CH, pch = idx - r4 * CH;                // window-relative row, physical chunk
      int lc = pch ^ (is_v ? swv(r4) : swk(r4));
      lc = lc < last_chunk ? lc : last_chunk;
      int row = win0 + r4;
      row = row < p.Skv ? row : p.Skv - 1;                         // rows past the end: a valid row (masked keys / zero weights)
      const char* src = (is_v ? vbase + (int64_t)row * p.v_rs * 2 : kbase + (int64_t)row * p.k_rs * 2) + lc * 16;
      glds16(src, smem + buf * kBuf + (is_v ? kImg : 0) + (idx - lane) * 16);
    }
  };
  const int n_win = (kv_block_end + W - 1) / W;
  if (n_win > 0) issue_window(0, 0);
  ATTN_STAMP(1);
  for (int w = 0; w < n_win; ++w) {
    const int win0 = w * W;
    const int n_tiles = (min(W, kv_block_end - win0) + 31) >> 5;
#ifdef STLLM_ATTN_TRACE
    const unsigned long long tw0 = __builtin_amdgcn_s_memtime();
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my pieces of window w landed ...
    __syncthreads();                                    // ... everybody's did, and everybody is done with window w - 1
#ifdef STLLM_ATTN_TRACE
    t_wait += __builtin_amdgcn_s_memtime() - tw0;
    if (w == 0) ATTN_STAMP(2);
#endif
    if (w + 1 < n_win) issue_window(win0 + W, (w + 1) & 1);
    if (p.q_lds && w == 0) {   // (every piece of the tile landed before the barrier above: each wave drained its own queue first)
      const char* qa = qimg + li * PITCH;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const i32x4 z = {0, 0, 0, 0};
        const i32x4 f = *reinterpret_cast<const i32x4*>(qa + (((ks * 2 + lh) ^ swk(li)) << 4));
        qf[ks] = (ks * 16 + lh * 8 < D) ? f : z;                      // dims past D: K's chunk there is a copy of the last real one
      }
    }
    if (q_live) {
      const char* kimg = smem + (w & 1) * kBuf;
      const char* vimg = kimg + kImg;
      int t_end = n_tiles;
      if (p.causal) t_end = min(n_tiles, qt + 1 - (win0 >> 5));
      for (int t = 0; t < t_end; ++t) {
        if (KS2 > 1 && (((win0 >> 5) + t) & (KS2 - 1)) != kpar) continue;
        const int kv0 = win0 + t * 32;
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.0f;
        {
          const int row = t * 32 + li;
          const char* ka = kimg + row * PITCH;
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            const i32x4 kf = *reinterpret_cast<const i32x4*>(ka + (((ks * 2 + lh) ^ swk(row)) << 4));
            s = Elem<T>::mfma(kf, qf[ks], s);
          }
        }
        const bool need_mask = (kv0 + 32 > kvlen) || (p.causal && kv0 + 31 > qt * 32);
        if (need_mask) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int kv = kv0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const bool dead = (kv >= kvlen) || (p.causal && kv > qrow);
            s[r] = dead ? kNeg : s[r];
          }
        }
        float mx = s[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float mc = m_new * p.scale_log2;
        float rs = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pv = __builtin_amdgcn_exp2f(fmaf(s[r], p.scale_log2, -mc));
          s[r] = pv;
          rs += pv;
        }
        rs += __shfl_xor(rs, 32, 64);
        if (__any(m_new != m_run)) {
          const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * p.scale_log2);
          l_run *= alpha;
#pragma unroll
          for (int i = 0; i < DB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
          m_run = m_new;
        }
        l_run += rs;
        // P V: lane (d = li, lh) needs the keys 16 a + 4 lh + {0..3, 8..11} of column d: two transposing reads of four keys each
        const int i16 = lane & 15, jrow = i16 >> 2, piece = i16 & 3, cb = (li >> 4) * 4;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          i32x4 pf;
#pragma unroll
          for (int e = 0; e < 4; ++e) pf[e] = (int)(Elem<T>::pack2(s[a * 8 + 2 * e], s[a * 8 + 2 * e + 1]));
          if constexpr (NT == 512 && DB == 4) {
            // 256 registers (the 8-wave instantiation): the eight transposing reads of a 16-key step in flight together, one wait (under the 768-thread
            // bound's 168 registers the same block spilled and measured slower: 24.8 vs 22.7 us, round 2)
            const int row = t * 32 + 16 * a + 4 * lh + jrow;
            const char* rp = vimg + row * PITCH;
            auto ad = [&](int i) { const int pc8 = i * 8 + cb + piece; return rp + (((pc8 >> 1) ^ swv(row)) << 4) + ((pc8 & 1) << 3); };
            unsigned long long lo[4], hi[4];
            lds_read_tr_4pairs<8 * PITCH>(ad(0), ad(1), ad(2), ad(3), lo, hi);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const i32x4 vf = {(int)(unsigned)lo[i], (int)(unsigned)(lo[i] >> 32), (int)(unsigned)hi[i], (int)(unsigned)(hi[i] >> 32)};
              o[i] = Elem<T>::mfma(vf, pf, o[i]);
            }
          } else {
#pragma unroll
          for (int i = 0; i < DB; ++i) {
            unsigned long long lo, hi;
            {
              const int row = t * 32 + 16 * a + 4 * lh + jrow;       // (+ 8 for the second read: same row & 7)
              const int pc8 = i * 8 + cb + piece;                    // 8-byte piece of the row: column 32 i + 16 (li >> 4) + 4 piece
              lds_read_tr_pair<8 * PITCH>(vimg + row * PITCH + (((pc8 >> 1) ^ swv(row)) << 4) + ((pc8 & 1) << 3), lo, hi);
            }
            const i32x4 vf = {(int)(unsigned)lo, (int)(unsigned)(lo >> 32), (int)(unsigned)hi, (int)(unsigned)(hi >> 32)};
            o[i] = Elem<T>::mfma(vf, pf, o[i]);
          }
          }
        }
      }
    }
  }
  ATTN_STAMP(3);
  ATTN_STAMP_VAL(6, n_win);
  ATTN_STAMP_VAL(7, t_wait);
  if constexpr (KS2 > 1) {
    // merge of the KS2 partial (m, l, O) states of a query tile.  p.merge_par (launch_dma: (KS2 - 1) x nwaves parked states fit in the LDS): every
    // partner parks its state in its OWN slot, ONE barrier pair, the tile's first wave folds them in the fixed order 1, 2, .. (bit-identical to the
    // sequential rounds below, which cost two barriers per partner: 5.4 k cycles at KS2 = 4 in the timeline, profiles/r06_attn_trace.md)
    float* mbuf = reinterpret_cast<float*>(smem);
    constexpr int kState = DB * 16 + 2;
    auto fold = [&](int slot) {
      const float m1 = mbuf[((slot * kState + DB * 16) << 6) + lane];
      const float l1 = mbuf[((slot * kState + DB * 16 + 1) << 6) + lane];
      const float m = fmaxf(m_run, m1);
      const float a0 = __builtin_amdgcn_exp2f((m_run - m) * p.scale_log2), a1 = __builtin_amdgcn_exp2f((m1 - m) * p.scale_log2);
      l_run = l_run * a0 + l1 * a1;
      m_run = m;
#pragma unroll
      for (int i = 0; i < DB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = o[i][r] * a0 + mbuf[((slot * kState + i * 16 + r) << 6) + lane] * a1;
    };
    auto park = [&](int slot) {
#pragma unroll
      for (int i = 0; i < DB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) mbuf[((slot * kState + i * 16 + r) << 6) + lane] = o[i][r];
      mbuf[((slot * kState + DB * 16) << 6) + lane] = m_run;
      mbuf[((slot * kState + DB * 16 + 1) << 6) + lane] = l_run;
    };
    if (p.merge_par) {
      __syncthreads();
      if (kpar > 0) park((kpar - 1) * nwaves + wave);
      __syncthreads();
      if (kpar == 0)
        for (int pp = 1; pp < KS2; ++pp) fold((pp - 1) * nwaves + wave);
    } else {
      for (int pp = 1; pp < KS2; ++pp) {
        __syncthreads();
        if (kpar == pp) park(wave);
        __syncthreads();
        if (kpar == 0) fold(wave);
      }
    }
    ATTN_STAMP(4);
    if (kpar != 0) return;
  }
  if (q_live && qrow < p.Sq) {
    const float inv = 1.0f / l_run;
    uint16_t* op = reinterpret_cast<uint16_t*>(p.o) + (int64_t)b * p.o_bs + (int64_t)qrow * p.o_rs + (int64_t)h * D;
#pragma unroll
    for (int i = 0; i < DB; ++i) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = i * 32 + 8 * g + 4 * lh;
        if (d0 < D) {
          uint2 pk;
          pk.x = Elem<T>::pack2(o[i][4 * g + 0] * inv, o[i][4 * g + 1] * inv);
          pk.y = Elem<T>::pack2(o[i][4 * g + 2] * inv, o[i][4 * g + 3] * inv);
          *reinterpret_cast<uint2*>(op + d0) = pk;
        }
      }
    }
  }
  ATTN_STAMP(5);
}

// ---- D = 88 (EVA ViT: 257 keys, padded to 96 dims x 288 keys): the same staging for the resident-K/V kernel -------------------------
// One workgroup per (frame, head) and query chunk, one wave per 32-row query tile, the whole K and V of the head in LDS as
// ROW-MAJOR images of 288 rows x 12 chunks of 16 bytes (chunk 11 = dims 88..95: a copy of chunk 10 — finite values that meet the
// zero-padded Q / unstored output columns).  The 108 KiB arrive by LDS-DMA in three windows of 96 keys (4 one-KiB pieces per wave
// and window at 9 waves); the waves start on window 0 while windows 1 and 2 are still in flight (counted vmcnt + one barrier per
// window).  K slot (row, c) holds logical chunk c ^ ((row >> 2) & 3) (rows are 48 dwords apart: rows r and r + 4 share banks);
// V needs no swizzle for the transposing reads (four consecutive rows = four disjoint 16-bank spans).
template <typename T>
__global__ __launch_bounds__(768) void attn_dma88_kernel(const AttnParams p) {
  ATTN_STAMP(0);
  constexpr int DP = 96, KS = DP / 16, DB = DP / 32, PITCH = 192, ROWS = 288;
  constexpr int kImg = ROWS * PITCH;                 // 55 296 bytes = 54 pieces of 1 KiB
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, nwaves = blockDim.x >> 6;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  int bx, b, h;
  attn_block_coords(bx, h, b);
  const int D = p.D;
  const int kvlen = p.kv_len ? min(p.kv_len[b], p.Skv) : p.Skv;
  const char* kbase = p.k + ((int64_t)b * p.k_bs + (int64_t)h * D) * 2;
  const char* vbase = p.v + ((int64_t)b * p.v_bs + (int64_t)h * D) * 2;
  const int qt = bx * nwaves + wave;
  const bool q_live = qt * 32 < p.Sq;
  const int qrow = qt * 32 + li;

  // ---- staging: window w = rows 96 w .. 96 w + 95 of K (18 pieces) and of V (18 pieces); piece pc of a window by wave pc % nwaves
  auto issue_window = [&](int w) {
    for (int pc = wave; pc < 36; pc += nwaves) {
      const bool is_v = pc >= 18;
      const int idx = (w * 18 + (is_v ? pc - 18 : pc)) * 64 + lane;   // physical 16-byte slot of the image
      const int r = idx / 12, pch = idx - r * 12;
      int lc = is_v ? pch : (pch ^ ((r >> 2) & 3));
      lc = lc < 11 ? lc : 10;
      const int row = r < p.Skv ? r : p.Skv - 1;
      const char* src = (is_v ? vbase + (int64_t)row * p.v_rs * 2 : kbase + (int64_t)row * p.k_rs * 2) + lc * 16;
      glds16(src, smem + (is_v ? kImg : 0) + (idx - lane) * 16);
    }
  };
  i32x4 qf[KS];
  {
    const char* qp = p.q + ((int64_t)b * p.q_bs + (int64_t)(qrow < p.Sq ? qrow : 0) * p.q_rs + (int64_t)h * D) * 2;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int d0 = ks * 16 + lh * 8;
      i32x4 z = {0, 0, 0, 0};
      qf[ks] = (qrow < p.Sq && d0 < D) ? *reinterpret_cast<const i32x4*>(qp + d0 * 2) : z;
    }
  }
  asm volatile("" ::: "memory");   // the Q loads are OLDER than the window DMAs: the first counted wait below covers them
  const int n_win = (min(kvlen, ROWS) + 95) / 96;
  for (int w = 0; w < n_win; ++w) issue_window(w);
  ATTN_STAMP(1);
  f32x16 o[DB];
#pragma unroll
  for (int i = 0; i < DB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[i][r] = 0.0f;
  float m_run = kNeg, l_run = 0.0f;
  const char* kimg = smem;
  const char* vimg = smem + kImg;
  const int i16 = lane & 15, jrow = i16 >> 2, piece = i16 & 3, cb = (li >> 4) * 4;
  const int n_tiles = (min(kvlen, ROWS) + 31) >> 5;

  for (int w = 0; w < n_win; ++w) {
    // my pieces of windows 0..w landed (pieces are issued window by window, the same number per window for a given wave) ...
    // (counted only in the ViT's shape — 9 waves: exactly 4 pieces per wave and window; any other wave count drains everything)
    const int later = n_win - 1 - w;
    if (nwaves == 9 && later == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (nwaves == 9 && later == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                    // ... and everybody else's
    ATTN_STAMP(2 + 2 * (w < 2 ? w : 2));
    if (!q_live) continue;
    const int t_hi = min(n_tiles, 3 * (w + 1));
    for (int t = 3 * w; t < t_hi; ++t) {
      const int kv0 = t * 32;
      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.0f;
      {
        const int row = t * 32 + li;
        const char* ka = kimg + row * PITCH;
        const int sw = (row >> 2) & 3;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const i32x4 kf = *reinterpret_cast<const i32x4*>(ka + (((ks * 2 + lh) ^ sw) << 4));
          s = Elem<T>::mfma(kf, qf[ks], s);
        }
      }
      if (kv0 + 32 > kvlen) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kv = kv0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          s[r] = kv >= kvlen ? kNeg : s[r];
        }
      }
      float mx = s[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run, mx);
      const float mc = m_new * p.scale_log2;
      float rs = 0.0f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = __builtin_amdgcn_exp2f(fmaf(s[r], p.scale_log2, -mc));
        s[r] = pv;
        rs += pv;
      }
      rs += __shfl_xor(rs, 32, 64);
      if (__any(m_new != m_run)) {
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * p.scale_log2);
        l_run *= alpha;
#pragma unroll
        for (int i = 0; i < DB; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
        m_run = m_new;
      }
      l_run += rs;
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        i32x4 pf;
#pragma unroll
        for (int e = 0; e < 4; ++e) pf[e] = (int)(Elem<T>::pack2(s[a * 8 + 2 * e], s[a * 8 + 2 * e + 1]));
        // the three pairs of transposing reads of this 16-key step in one block, ONE wait (registers to spare here: 120 of 170)
        unsigned long long l0, h0, l1, h1, l2, h2;
        {
          const int row = t * 32 + 16 * a + 4 * lh + jrow;
          lds_read_tr_3pairs<8 * PITCH>(vimg + row * PITCH + (cb + piece) * 8, l0, h0, l1, h1, l2, h2);
        }
        {
          const i32x4 v0 = {(int)(unsigned)l0, (int)(unsigned)(l0 >> 32), (int)(unsigned)h0, (int)(unsigned)(h0 >> 32)};
          const i32x4 v1 = {(int)(unsigned)l1, (int)(unsigned)(l1 >> 32), (int)(unsigned)h1, (int)(unsigned)(h1 >> 32)};
          const i32x4 v2 = {(int)(unsigned)l2, (int)(unsigned)(l2 >> 32), (int)(unsigned)h2, (int)(unsigned)(h2 >> 32)};
          o[0] = Elem<T>::mfma(v0, pf, o[0]);
          o[1] = Elem<T>::mfma(v1, pf, o[1]);
          o[2] = Elem<T>::mfma(v2, pf, o[2]);
        }
      }
    }
    ATTN_STAMP(3 + 2 * (w < 2 ? w : 2));
  }
  if (q_live && qrow < p.Sq) {
    const float inv = 1.0f / l_run;
    uint16_t* op = reinterpret_cast<uint16_t*>(p.o) + (int64_t)b * p.o_bs + (int64_t)qrow * p.o_rs + (int64_t)h * D;
#pragma unroll
    for (int i = 0; i < DB; ++i) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = i * 32 + 8 * g + 4 * lh;
        if (d0 < D) {
          uint2 pk;
          pk.x = Elem<T>::pack2(o[i][4 * g + 0] * inv, o[i][4 * g + 1] * inv);
          pk.y = Elem<T>::pack2(o[i][4 * g + 2] * inv, o[i][4 * g + 3] * inv);
          *reinterpret_cast<uint2*>(op + d0) = pk;
        }
      }
    }
  }
}

template <typename T>
int launch_dma88(const AttnParams& p, hipStream_t stream) {
  constexpr int lds = 2 * 288 * 192;
  static StllmPerDevice attr_dev;   // the dynamic-LDS opt-in is a per-device attribute
  bool attr_first;
  const int attr_d = attr_dev.enter(&attr_first);
  if (attr_first) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_dma88_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_dev.done(attr_d);
  }
  const int q_tiles = (p.Sq + 31) / 32;
  const int nw = q_tiles < 12 ? q_tiles : 12;
  dim3 grid((q_tiles + nw - 1) / nw, p.H, p.B), block(64 * nw);
  hipLaunchKernelGGL((attn_dma88_kernel<T>), grid, block, lds, stream, p);
  STLLM_CHECK_LAUNCH("stllm_attention(dma88)");
  return STLLM_OK;
}

// env STLLM_ATTN_DMA / option "attn_dma" (stllm_options().attn_dma): 1 (default) LDS-DMA kernels for head_dim 128 (Llama prefill) and 88 (ViT) |
                              // 0 register-staged kernels everywhere
template <typename T, int KS2, int DP = 128, int NT = 768>
int launch_dma_nt(const AttnParams& p, hipStream_t stream, int nw_req) {
  constexpr int W = DP == 128 ? 128 : 96;
  constexpr int lds_win = 2 * 2 * W * (DP * 2);   // two buffers of [K | V] windows (128 KiB at 128 dims, 72 KiB at 96: two workgroups per CU)
  constexpr int lds_max = DP == 128 ? 160 * 1024 : 80 * 1024;
  static StllmPerDevice attr_dev;   // the dynamic-LDS opt-in is a per-device attribute
  bool attr_first;
  const int attr_d = attr_dev.enter(&attr_first);
  if (attr_first) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_dma_kernel<T, KS2, DP, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    attr_dev.done(attr_d);
  }
  const int q_tiles = (p.Sq + 31) / 32;
  int nw = nw_req < q_tiles ? nw_req : q_tiles;
  if (KS2 > 1 && nw * (DP / 32 * 16 + 2) * 256 > lds_win) {
    stllm_set_error("stllm_attention(dma): key-split merge buffer does not fit (nw=%d)", nw);
    return STLLM_ERR_UNSUPPORTED;
  }
  // query tiles through the LDS (attn_dma_kernel) when their images fit behind the window buffers; option attn_q_lds = 0: never (A/B)
  AttnParams pp = p;
  const int q_img = nw * 32 * (DP * 2);
  pp.q_lds = (stllm_options().attn_q_lds != 0 && lds_win + q_img <= lds_max && ((int64_t)p.q_rs * 2) % 16 == 0 && (p.D * 2) % 16 == 0) ? 1 : 0;
  const int lds = lds_win + (pp.q_lds ? q_img : 0);
  pp.merge_par = (KS2 > 1 && stllm_options().attn_q_lds != 0 && (KS2 - 1) * nw * (DP / 32 * 16 + 2) * 256 <= lds) ? 1 : 0;
  dim3 grid((q_tiles + nw - 1) / nw, p.H, p.B), block(64 * nw * KS2);
  hipLaunchKernelGGL((attn_dma_kernel<T, KS2, DP, NT>), grid, block, lds, stream, pp);
  STLLM_CHECK_LAUNCH("stllm_attention(dma)");
  return STLLM_OK;
}

template <typename T, int KS2, int DP = 128>
int launch_dma(const AttnParams& p, hipStream_t stream, int nw_req) {
  const int q_tiles = (p.Sq + 31) / 32;
  const int nw = nw_req < q_tiles ? nw_req : q_tiles;
  if (64 * nw * KS2 <= 512 && stllm_options().attn_q_lds != 0) return launch_dma_nt<T, KS2, DP, 512>(p, stream, nw_req);
  return launch_dma_nt<T, KS2, DP, 768>(p, stream, nw_req);
}

// ---- exact fp32 path: one wave per query row -----------------------------------------------------
constexpr int kF32MaxKv = 2048;

__global__ __launch_bounds__(256) void attn_f32_kernel(const AttnParams p) {
  __shared__ float q_s[4][128];
  __shared__ float sc[4][kF32MaxKv];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int b = blockIdx.z, h = blockIdx.y;
  const int qrow = blockIdx.x * 4 + w;
  if (qrow >= p.Sq) return;  // no block-level sync below
  const int D = p.D;
  const int kvlen = p.kv_len ? min(p.kv_len[b], p.Skv) : p.Skv;
  const int kv_end = p.causal ? min(kvlen, qrow + 1) : kvlen;
  const float* q = reinterpret_cast<const float*>(p.q) + (int64_t)b * p.q_bs + (int64_t)qrow * p.q_rs + (int64_t)h * D;
  const float* kb = reinterpret_cast<const float*>(p.k) + (int64_t)b * p.k_bs + (int64_t)h * D;
  const float* vb = reinterpret_cast<const float*>(p.v) + (int64_t)b * p.v_bs + (int64_t)h * D;
  for (int d = lane; d < D; d += 64) q_s[w][d] = q[d];
  __builtin_amdgcn_wave_barrier();
  float mx = kNeg;
  for (int j = lane; j < kv_end; j += 64) {
    const float4* kr = reinterpret_cast<const float4*>(kb + (int64_t)j * p.k_rs);
    float acc = 0.0f;
    for (int d4 = 0; d4 < (D >> 2); ++d4) {
      const float4 kk = kr[d4];
      acc = fmaf(q_s[w][4 * d4 + 0], kk.x, acc);
      acc = fmaf(q_s[w][4 * d4 + 1], kk.y, acc);
      acc = fmaf(q_s[w][4 * d4 + 2], kk.z, acc);
      acc = fmaf(q_s[w][4 * d4 + 3], kk.w, acc);
    }
    acc *= p.scale;
    sc[w][j] = acc;
    mx = fmaxf(mx, acc);
  }
  mx = wave_max(mx);
  float sum = 0.0f;
  for (int j = lane; j < kv_end; j += 64) {
    const float e = expf(sc[w][j] - mx);
    sc[w][j] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  __builtin_amdgcn_wave_barrier();
  const float inv = 1.0f / sum;
  float* o = reinterpret_cast<float*>(p.o) + (int64_t)b * p.o_bs + (int64_t)qrow * p.o_rs + (int64_t)h * D;
  for (int d = lane; d < D; d += 64) {
    float acc = 0.0f;
    for (int j = 0; j < kv_end; ++j) acc = fmaf(sc[w][j], vb[(int64_t)j * p.v_rs + d], acc);
    o[d] = acc * inv;
  }
}

// ---- exact fp32 path on the MATRIX cores (round 4): v_mfma_f32_32x32x2_f32, bit-equal to an fmaf chain per output --------------------
// The one-wave-per-row kernel above re-reads the whole K and V of a head from L2 for every query row (181 KB per ViT row: ~12 GB per
// layer) and ran at 3.6 % of the fp32 vector peak: 58 of the 117 ms of a split-verify step (profiles/r04_split_step.md).  This kernel is
// attn_mfma_kernel's schedule in fp32: a wave owns 32 query rows, 32-key tiles of K (row-major) and V^T go through LDS once per
// WORKGROUP (register-staged one tile ahead), S^T = K Q^T and O^T += V^T P^T are chains of 32x32x2 fp32 MFMAs (K = 2 per instruction:
// DP / 2 + 16 DP / 32 of them per tile), softmax statistics per lane exactly as in the 16-bit kernels.  A 16-byte LDS read feeds four MFMAs
// (the k <-> (step, half) slot map only has to agree between the A and the B operand, Elem<float>::mfma).
template <int DP, int NW>
__global__ __launch_bounds__(64 * NW) void attn_mfma_f32_kernel(const AttnParams p) {
  constexpr int KS8 = DP / 8;           // 16-byte K fragments per row half-pair: lane half lh holds d = 8 ks + 4 lh .. + 3
  constexpr int DB = DP / 32;           // 32-row blocks of O^T
  constexpr int KPITCH = DP * 4 + 16;   // bytes per K row in LDS
  constexpr int VPITCH = 32 * 4 + 16;   // bytes per V^T row (32 keys) in LDS
  constexpr int NT = 64 * NW;
  constexpr int CPR = DP / 4;           // 16-byte chunks per K / V row
  constexpr int NCH = 32 * CPR;
  constexpr int CPT = (NCH + NT - 1) / NT;
  __shared__ __attribute__((aligned(16))) char k_lds[32 * KPITCH];
  __shared__ __attribute__((aligned(16))) char v_lds[DP * VPITCH];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  const int q_blk0 = blockIdx.x * (32 * NW);
  const int qrow = q_blk0 + wave * 32 + li;
  const int D = p.D;
  const int kvlen = p.kv_len ? min(p.kv_len[b], p.Skv) : p.Skv;

  f32x4 qf[KS8];   // B operand of S^T: Q[q = li][8 ks + 4 lh .. + 3]
  {
    const float* qp = reinterpret_cast<const float*>(p.q) + (int64_t)b * p.q_bs + (int64_t)(qrow < p.Sq ? qrow : 0) * p.q_rs + (int64_t)h * D;
#pragma unroll
    for (int ks = 0; ks < KS8; ++ks) {
      const int d0 = ks * 8 + lh * 4;
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      qf[ks] = (qrow < p.Sq && d0 < D) ? *reinterpret_cast<const f32x4*>(qp + d0) : z;   // D % 4 == 0
    }
  }
  f32x16 o[DB];
#pragma unroll
  for (int i = 0; i < DB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[i][r] = 0.0f;
  float m_run = kNeg, l_run = 0.0f;

  int kv_end = kvlen;
  if (p.causal) kv_end = min(kv_end, q_blk0 + 32 * NW);
  const float* kbase = reinterpret_cast<const float*>(p.k) + (int64_t)b * p.k_bs + (int64_t)h * D;
  const float* vbase = reinterpret_cast<const float*>(p.v) + (int64_t)b * p.v_bs + (int64_t)h * D;
  f32x4 kreg[CPT], vreg[CPT];
  auto load_tile = [&](int kv0) {
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
      const int ch = tid + c * NT;
      const int cc = ch >> 5, row = ch & 31;   // consecutive lanes = consecutive keys: conflict-free transposed V writes
      const int kv = kv0 + row;
      const bool ok = (ch < NCH) && (kv < p.Skv) && (cc * 4 < D);
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      kreg[c] = ok ? *reinterpret_cast<const f32x4*>(kbase + (int64_t)kv * p.k_rs + cc * 4) : z;
      vreg[c] = ok ? *reinterpret_cast<const f32x4*>(vbase + (int64_t)kv * p.v_rs + cc * 4) : z;
    }
  };
  const int wave_q_last = q_blk0 + wave * 32 + 31;
  if (kv_end > 0) load_tile(0);
  for (int kv0 = 0; kv0 < kv_end; kv0 += 32) {
    __syncthreads();
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
      const int ch = tid + c * NT;
      if (ch < NCH) {
        const int cc = ch >> 5, row = ch & 31;
        *reinterpret_cast<f32x4*>(k_lds + row * KPITCH + cc * 16) = kreg[c];
#pragma unroll
        for (int e = 0; e < 4; ++e) *reinterpret_cast<float*>(v_lds + (cc * 4 + e) * VPITCH + row * 4) = vreg[c][e];
      }
    }
    __syncthreads();
    if (kv0 + 32 < kv_end) load_tile(kv0 + 32);
    if (p.causal && kv0 > wave_q_last) continue;

    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.0f;
#pragma unroll
    for (int ks = 0; ks < KS8; ++ks) {
      const f32x4 kf = *reinterpret_cast<const f32x4*>(k_lds + li * KPITCH + (ks * 2 + lh) * 16);
#pragma unroll
      for (int e = 0; e < 4; ++e) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], qf[ks][e], s, 0, 0, 0);
    }
    const bool need_mask = (kv0 + 32 > kvlen) || (p.causal && kv0 + 31 > q_blk0 + wave * 32);
    if (need_mask) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kv = kv0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const bool dead = (kv >= kvlen) || (p.causal && kv > qrow);
        s[r] = dead ? kNeg : s[r];
      }
    }
    float mx = s[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float mc = m_new * p.scale_log2;
    float rs = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float pv = exp2f(fmaf(s[r], p.scale_log2, -mc));   // the library exp2f (full fp32 accuracy), not the hardware approximation
      s[r] = pv;
      rs += pv;
    }
    rs += __shfl_xor(rs, 32, 64);
    if (__any(m_new != m_run)) {
      const float alpha = exp2f((m_run - m_new) * p.scale_log2);
      l_run *= alpha;
#pragma unroll
      for (int i = 0; i < DB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
      m_run = m_new;
    }
    l_run += rs;
    // O^T[d][q] += V^T[d][kv] P^T[kv][q]: the lane's score register r is key (r & 3) + 8 (r >> 2) + 4 lh — registers 4 g .. 4 g + 3 are four
    // consecutive keys 8 g + 4 lh ..: one 16-byte read of the V^T row feeds their four MFMAs
#pragma unroll
    for (int i = 0; i < DB; ++i) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 vf = *reinterpret_cast<const f32x4*>(v_lds + (i * 32 + li) * VPITCH + (8 * g + 4 * lh) * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[e], s[4 * g + e], o[i], 0, 0, 0);
      }
    }
  }
  if (qrow < p.Sq) {
    const float inv = 1.0f / l_run;
    float* op = reinterpret_cast<float*>(p.o) + (int64_t)b * p.o_bs + (int64_t)qrow * p.o_rs + (int64_t)h * D;
#pragma unroll
    for (int i = 0; i < DB; ++i) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = i * 32 + 8 * g + 4 * lh;
        if (d0 < D) {
          const f32x4 ov = {o[i][4 * g + 0] * inv, o[i][4 * g + 1] * inv, o[i][4 * g + 2] * inv, o[i][4 * g + 3] * inv};
          *reinterpret_cast<f32x4*>(op + d0) = ov;
        }
      }
    }
  }
}

template <int DP, int NW>
int launch_mfma_f32(const AttnParams& p, hipStream_t stream) {
  dim3 grid((p.Sq + 32 * NW - 1) / (32 * NW), p.H, p.B), block(64 * NW);
  hipLaunchKernelGGL((attn_mfma_f32_kernel<DP, NW>), grid, block, 0, stream, p);
  STLLM_CHECK_LAUNCH("stllm_attention(fp32 mfma)");
  return STLLM_OK;
}

// fp32: matrix-core kernel from 8 query rows on (prefill shapes), the vector kernel for the one-row decode step and tiny problems
int dispatch_f32(const AttnParams& p, hipStream_t stream, int mode) {
  if (mode != 0 && p.Sq >= 8) {
    const int nw = p.Sq <= 32 ? 1 : (p.Sq <= 64 ? 2 : 3);
    if (p.D == 64) return nw == 1 ? launch_mfma_f32<64, 1>(p, stream) : nw == 2 ? launch_mfma_f32<64, 2>(p, stream) : launch_mfma_f32<64, 3>(p, stream);
    if (p.D == 88) return nw == 1 ? launch_mfma_f32<96, 1>(p, stream) : nw == 2 ? launch_mfma_f32<96, 2>(p, stream) : launch_mfma_f32<96, 3>(p, stream);
    if (p.D == 128) return nw == 1 ? launch_mfma_f32<128, 1>(p, stream) : nw == 2 ? launch_mfma_f32<128, 2>(p, stream) : launch_mfma_f32<128, 3>(p, stream);
  }
  if (p.Skv > kF32MaxKv) {
    stllm_set_error("stllm_attention(fp32, vector kernel): Skv %d > %d", p.Skv, kF32MaxKv);
    return STLLM_ERR_BAD_SHAPE;
  }
  dim3 grid((p.Sq + 3) / 4, p.H, p.B), block(256);
  hipLaunchKernelGGL(attn_f32_kernel, grid, block, 0, stream, p);
  STLLM_CHECK_LAUNCH("stllm_attention(fp32)");
  return STLLM_OK;
}

template <typename T, int DP, int NW>
int launch_mfma(const AttnParams& p, hipStream_t stream) {
  dim3 grid((p.Sq + 32 * NW - 1) / (32 * NW), p.H, p.B), block(64 * NW);
  hipLaunchKernelGGL((attn_mfma_kernel<T, DP, NW>), grid, block, 0, stream, p);
  STLLM_CHECK_LAUNCH("stllm_attention");
  return STLLM_OK;
}

template <typename T>
int dispatch(const AttnParams& p, hipStream_t stream) {
  // NW picked so that 32*NW divides the common sequence lengths with little waste:
  //   ViT 257 -> 3 waves (96 rows, 3 blocks), Q-Former 32/44 -> 1-2 waves, Llama -> 3 waves (576 = 6*96)
  if (p.D == 88) {
    if (p.Sq <= 32) return launch_mfma<T, 96, 1>(p, stream);                       // BT-Adapter temporal attention
    if (p.Skv <= 288 && !p.causal) {
      const int g_attn_dma = stllm_options().attn_dma;
      const bool al = ((reinterpret_cast<uintptr_t>(p.k) | reinterpret_cast<uintptr_t>(p.v)) & 15) == 0 && (p.k_rs % 8) == 0 && (p.v_rs % 8) == 0 &&
                      (p.k_bs % 8) == 0 && (p.v_bs % 8) == 0;
      // LDS-DMA staging in three 96-key windows + transposing V reads: 23.2 vs 24.1 us for the register-staged resident kernel (both
      // after the XCD-aware block mapping; 26.3 vs 27.1 before it)
      if (g_attn_dma != 0 && al) return launch_dma88<T>(p, stream);
      return launch_resident<T, 96, 288>(p, stream);
    }  // ViT: K/V of a head resident in LDS (a key split here re-stages K/V per query chunk: 26 -> 37 us)
    return launch_mfma<T, 96, 3>(p, stream);
  }
  if (p.D == 64) return p.Sq <= 32 ? launch_mfma<T, 64, 1>(p, stream) : launch_mfma<T, 64, 2>(p, stream);
  if (p.D == 128) {
    const int g_attn_dma = stllm_options().attn_dma;
    const bool al = ((reinterpret_cast<uintptr_t>(p.k) | reinterpret_cast<uintptr_t>(p.v)) & 15) == 0 && (p.k_rs % 8) == 0 && (p.v_rs % 8) == 0 &&
                    (p.k_bs % 8) == 0 && (p.v_bs % 8) == 0;
    if (g_attn_dma != 0 && al && p.Skv >= 1) {   // LDS-DMA windows + transposing V reads (round 2)
      // query tiles per workgroup x waves per query tile, measured at S = 576, 32 heads (us at B = 1 / B = 4): 12 x 1: 27.1 / 33.0,
      // 6 x 2: 27.1 / 48.6, 4 x 2: 21.5 / 53.4, 3 x 4: 22.7 / 62.1 (register-staged kernel: 28.0 / 75.3) — the fewest key-split
      // waves that still give >= 128 workgroups
      const int q_tiles = (p.Sq + 31) / 32, bh = p.B * p.H;
      if (g_attn_dma >= 10) {   // experiments (tools/attn_bench.py --audit): attn_dma = 10 x (query tiles per workgroup) + (waves per query tile), at most 12 waves
        const int nwf = g_attn_dma / 10, ks = g_attn_dma % 10;
        if (nwf >= 1 && nwf * ks <= 12) {
          if (ks == 1) return launch_dma<T, 1>(p, stream, nwf);
          if (ks == 2) return launch_dma<T, 2>(p, stream, nwf);
          if (ks == 4) return launch_dma<T, 4>(p, stream, nwf);
        }
      }
      // round 4 (tools/attn_bench.py --audit, profiles/r04_attn_dispatch_audit.log): one sequence of <= 640 positions (32 heads) runs best as 2 query tiles x 4 key-split
      // waves per workgroup — 10.8 / 13.0 / 14.9 / 19.0 / 19.1 us at S = 178 / 296 / 400 / 528 / 576 against 14.2 / 16.2 / 17.0 / 20.5 / 20.6 on the rules below
      if (bh <= 32 && q_tiles <= 20) return launch_dma<T, 4>(p, stream, 2);
      if (bh * ((q_tiles + 11) / 12) >= 128) return launch_dma<T, 1>(p, stream, 12);
      if (bh * ((q_tiles + 3) / 4) >= 128) return launch_dma<T, 2>(p, stream, 4);
      return launch_dma<T, 4>(p, stream, 3);
    }
    return launch_resident<T, 128, 128, 4>(p, stream, 3);
  }   // + four waves per query tile (every 4th key tile each)  // 128-key windows (2 workgroups/CU), 96 queries per workgroup
  stllm_set_error("stllm_attention: unsupported head_dim %d", p.D);
  return STLLM_ERR_UNSUPPORTED;
}

}  // namespace

// =====================================================================================================
// Decode attention (Sq = 1 against a KV cache): HBM-bound streaming of K and V.
// The tile kernels above give one query tile per head = 32 workgroups walking the cache serially (38 us per layer
// at Skv = 580).  Here the keys are split across `nsplit` workgroups per (batch, head):
//   partial kernel: 4 waves per workgroup, wave w takes keys k0 + w, k0 + w + 4, ...; lane l owns dims 2l, 2l+1
//     (one 4-byte load per lane = one coalesced 256-byte row per wave), score = butterfly-reduced dot, online softmax
//     in the exp2 domain, O accumulated in 2 registers per lane; the 4 waves merge through LDS and write (m, l, O[128]);
//   merge kernel: one workgroup per (batch, head) combines the nsplit partials and writes the output row.
// =====================================================================================================
constexpr int kDecD = 128;
constexpr int kDecRec = kDecD + 2;   // floats per partial record: m, l, O[128]

template <typename T> __device__ __forceinline__ void widen2(uint32_t u, float& a, float& b);
template <> __device__ __forceinline__ void widen2<bf16_t>(uint32_t u, float& a, float& b) {
  a = __builtin_bit_cast(float, u << 16);
  b = __builtin_bit_cast(float, u & 0xffff0000u);
}
template <> __device__ __forceinline__ void widen2<f16_t>(uint32_t u, float& a, float& b) {
  a = (float)__builtin_bit_cast(_Float16, (uint16_t)(u & 0xffffu));
  b = (float)__builtin_bit_cast(_Float16, (uint16_t)(u >> 16));
}

template <typename T>
__global__ __launch_bounds__(256) void attn_decode_partial_kernel(const AttnParams p, float* __restrict__ ws, int nsplit, int keys_per_split) {
  __shared__ float part[4][kDecRec];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sp = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int k0 = sp * keys_per_split, k1 = min(p.Skv, k0 + keys_per_split);
  float q0, q1;
  widen2<T>(*reinterpret_cast<const uint32_t*>(p.q + ((int64_t)b * p.q_bs + (int64_t)h * kDecD + 2 * lane) * 2), q0, q1);
  q0 *= p.scale_log2;
  q1 *= p.scale_log2;
  const char* kb = p.k + ((int64_t)b * p.k_bs + (int64_t)h * kDecD + 2 * lane) * 2;
  const char* vb = p.v + ((int64_t)b * p.v_bs + (int64_t)h * kDecD + 2 * lane) * 2;
  float m = kNeg, l = 0.0f, o0 = 0.0f, o1 = 0.0f;
  for (int key = k0 + wave; key < k1; key += 4) {
    float ka, kc, va, vc;
    widen2<T>(*reinterpret_cast<const uint32_t*>(kb + (int64_t)key * p.k_rs * 2), ka, kc);
    widen2<T>(*reinterpret_cast<const uint32_t*>(vb + (int64_t)key * p.v_rs * 2), va, vc);
    const float s = wave_sum(fmaf(q0, ka, q1 * kc));   // log2-domain score, identical in every lane
    const float mn = fmaxf(m, s);
    const float alpha = __builtin_amdgcn_exp2f(m - mn), pr = __builtin_amdgcn_exp2f(s - mn);
    l = fmaf(l, alpha, pr);
    o0 = fmaf(o0, alpha, pr * va);
    o1 = fmaf(o1, alpha, pr * vc);
    m = mn;
  }
  if (lane == 0) { part[wave][0] = m; part[wave][1] = l; }
  part[wave][2 + 2 * lane] = o0;
  part[wave][3 + 2 * lane] = o1;
  __syncthreads();
  if (wave == 0) {
    float mm = fmaxf(fmaxf(part[0][0], part[1][0]), fmaxf(part[2][0], part[3][0]));
    float ll = 0.0f, a0 = 0.0f, a1 = 0.0f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float f = __builtin_amdgcn_exp2f(part[w][0] - mm);
      ll = fmaf(part[w][1], f, ll);
      a0 = fmaf(part[w][2 + 2 * lane], f, a0);
      a1 = fmaf(part[w][3 + 2 * lane], f, a1);
    }
    float* rec = ws + (((int64_t)b * p.H + h) * nsplit + sp) * kDecRec;
    if (lane == 0) { rec[0] = mm; rec[1] = ll; }
    rec[2 + 2 * lane] = a0;
    rec[3 + 2 * lane] = a1;
  }
}

template <typename T>
__global__ __launch_bounds__(64) void attn_decode_merge_kernel(const AttnParams p, const float* __restrict__ ws, int nsplit) {
  const int lane = threadIdx.x, h = blockIdx.x, b = blockIdx.y;
  const float* rec = ws + ((int64_t)b * p.H + h) * nsplit * kDecRec;
  float mm = kNeg;
  for (int s = 0; s < nsplit; ++s) mm = fmaxf(mm, rec[s * kDecRec]);
  float ll = 0.0f, a0 = 0.0f, a1 = 0.0f;
  for (int s = 0; s < nsplit; ++s) {
    const float f = __builtin_amdgcn_exp2f(rec[s * kDecRec] - mm);
    ll = fmaf(rec[s * kDecRec + 1], f, ll);
    a0 = fmaf(rec[s * kDecRec + 2 + 2 * lane], f, a0);
    a1 = fmaf(rec[s * kDecRec + 3 + 2 * lane], f, a1);
  }
  const float inv = 1.0f / ll;
  uint32_t* op = reinterpret_cast<uint32_t*>(p.o + ((int64_t)b * p.o_bs + (int64_t)h * kDecD + 2 * lane) * 2);
  *op = Elem<T>::pack2(a0 * inv, a1 * inv);
}

// ---- single-pass decode attention (round 2): ONE workgroup of 16 waves per (batch, head) ---------------------------------------
// The split-KV pair above costs two launches (10.5 + 7.7 us per layer at Skv = 580) for 0.3 MB of K/V per head.  Here a wave works on
// FOUR keys at a time: 16 lanes per key, 8 dims = one 16-byte load per lane for K and for V (4x wider than the 4-byte loads above),
// a 4-step butterfly inside the 16-lane group, an online-softmax state per lane group; the next step's K/V are requested before the
// current step is consumed.  The 4 x 16 partial states of the workgroup are merged through LDS (8.3 KB).  Used when the cache is
// short enough that 16 waves per head cover it in a few steps (Skv <= kDecSingleMax) or when B * H alone fills the chip.
constexpr int kDecSingleMax = 1536;
constexpr int kDecWaves = 16;

template <typename T> __device__ __forceinline__ void widen8v(i32x4 v, float* f);
template <> __device__ __forceinline__ void widen8v<bf16_t>(i32x4 v, float* f) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    f[2 * e] = __builtin_bit_cast(float, (uint32_t)v[e] << 16);
    f[2 * e + 1] = __builtin_bit_cast(float, (uint32_t)v[e] & 0xffff0000u);
  }
}
template <> __device__ __forceinline__ void widen8v<f16_t>(i32x4 v, float* f) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const uint32_t u = (uint32_t)v[e];
    f[2 * e] = (float)__builtin_bit_cast(_Float16, (uint16_t)(u & 0xffffu));
    f[2 * e + 1] = (float)__builtin_bit_cast(_Float16, (uint16_t)(u >> 16));
  }
}

template <typename T>
__global__ __launch_bounds__(64 * kDecWaves) void attn_decode_single_kernel(const AttnParams p) {
  __shared__ float part[kDecWaves * 4][kDecRec];   // one (m, l, O[128]) record per 16-lane group
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int grp = lane >> 4, l16 = lane & 15;     // key within the wave's quad, dims 8 * l16 .. + 7
  const int h = blockIdx.x, b = blockIdx.y;
  float q[8];
  widen8v<T>(*reinterpret_cast<const i32x4*>(p.q + ((int64_t)b * p.q_bs + (int64_t)h * kDecD + 8 * l16) * 2), q);
#pragma unroll
  for (int e = 0; e < 8; ++e) q[e] *= p.scale_log2;
  const char* kb = p.k + ((int64_t)b * p.k_bs + (int64_t)h * kDecD + 8 * l16) * 2;
  const char* vb = p.v + ((int64_t)b * p.v_bs + (int64_t)h * kDecD + 8 * l16) * 2;
  float m = kNeg, l = 0.0f, o[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = 0.0f;
  const int step = kDecWaves * 4;
  int kbase = wave * 4;                              // wave-uniform: all four lane groups run the same number of steps
  i32x4 kn = {0, 0, 0, 0}, vn = {0, 0, 0, 0};
  if (kbase + grp < p.Skv) {
    kn = *reinterpret_cast<const i32x4*>(kb + (int64_t)(kbase + grp) * p.k_rs * 2);
    vn = *reinterpret_cast<const i32x4*>(vb + (int64_t)(kbase + grp) * p.v_rs * 2);
  }
  for (; kbase < p.Skv; kbase += step) {
    const int key = kbase + grp;
    const bool live = key < p.Skv;
    const i32x4 kc = kn, vc = vn;
    const int nk = key + step;
    if (nk < p.Skv) {   // request the next quad of this wave before this one is consumed
      kn = *reinterpret_cast<const i32x4*>(kb + (int64_t)nk * p.k_rs * 2);
      vn = *reinterpret_cast<const i32x4*>(vb + (int64_t)nk * p.v_rs * 2);
    }
    float kf[8], vf[8];
    widen8v<T>(kc, kf);
    widen8v<T>(vc, vf);
    float s = 0.0f;
#pragma unroll
    for (int e = 0; e < 8; ++e) s = fmaf(q[e], kf[e], s);
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    s += __shfl_xor(s, 8, 64);                       // log2-domain score of this group's key, identical in its 16 lanes
    if (live) {
      const float mn = fmaxf(m, s);
      const float alpha = __builtin_amdgcn_exp2f(m - mn), pr = __builtin_amdgcn_exp2f(s - mn);
      l = fmaf(l, alpha, pr);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = fmaf(o[e], alpha, pr * vf[e]);
      m = mn;
    }
  }
  float* rec = part[wave * 4 + grp];
  if (l16 == 0) { rec[0] = m; rec[1] = l; }
#pragma unroll
  for (int e = 0; e < 8; ++e) rec[2 + 8 * l16 + e] = o[e];
  __syncthreads();
  if (threadIdx.x < kDecD) {
    const int d = threadIdx.x;
    float mm = kNeg;
    for (int r = 0; r < kDecWaves * 4; ++r) mm = fmaxf(mm, part[r][0]);
    float ll = 0.0f, acc = 0.0f;
    for (int r = 0; r < kDecWaves * 4; ++r) {
      const float f = __builtin_amdgcn_exp2f(part[r][0] - mm);
      ll = fmaf(part[r][1], f, ll);
      acc = fmaf(part[r][2 + d], f, acc);
    }
    store_elem<T>(p.o, (int64_t)b * p.o_bs + (int64_t)h * kDecD + d, acc / ll);
  }
}

// stllm_set_option("attn_decode_single", 0): always the split-KV pair (tests compare both)
static int decode_splits(int Skv) {
  int n = (Skv + 47) / 48;   // ~48 keys (12 per wave) per workgroup
  return n < 1 ? 1 : (n > 64 ? 64 : n);
}

extern "C" int stllm_attention(int dtype, const void* q, int64_t q_bs, int64_t q_rs, const void* k, int64_t k_bs,
                               int64_t k_rs, const void* v, int64_t v_bs, int64_t v_rs, void* out, int64_t o_bs,
                               int64_t o_rs, int B, int H, int Sq, int Skv, int D, float scale, int causal,
                               const int32_t* kv_len, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  STLLM_CHECK_ARG(q && k && v && out, "stllm_attention: null pointer");
  STLLM_CHECK_ARG(B > 0 && H > 0 && Sq > 0 && Skv > 0, "stllm_attention: empty problem");
  STLLM_CHECK_ARG(D == 64 || D == 88 || D == 128, "stllm_attention: head_dim %d not in {64,88,128}", D);
  STLLM_CHECK_ARG(!causal || Sq == Skv, "stllm_attention: causal needs Sq == Skv");
  STLLM_CHECK_ARG(aligned16(q) && aligned16(k) && aligned16(v) && aligned16(out), "stllm_attention: pointers must be 16-byte aligned");
  STLLM_CHECK_ARG(q_rs % 8 == 0 && k_rs % 8 == 0 && v_rs % 8 == 0 && o_rs % 8 == 0 && q_bs % 8 == 0 && k_bs % 8 == 0 &&
                      v_bs % 8 == 0 && o_bs % 8 == 0, "stllm_attention: strides must be multiples of 8 elements");
  AttnParams p{};
  p.q = (const char*)q; p.q_bs = q_bs; p.q_rs = q_rs;
  p.k = (const char*)k; p.k_bs = k_bs; p.k_rs = k_rs;
  p.v = (const char*)v; p.v_bs = v_bs; p.v_rs = v_rs;
  p.o = (char*)out; p.o_bs = o_bs; p.o_rs = o_rs;
  p.B = B; p.H = H; p.Sq = Sq; p.Skv = Skv; p.D = D;
  p.scale = scale; p.scale_log2 = scale * 1.44269504088896340736f;
  p.causal = causal; p.kv_len = kv_len;
  switch (dtype) {
    case STLLM_BF16: return dispatch<bf16_t>(p, stream);
    case STLLM_F16: return dispatch<f16_t>(p, stream);
    case STLLM_F32: return dispatch_f32(p, stream, stllm_options().attn_f32_mfma);
  }
  stllm_set_error("stllm_attention: bad dtype %d", dtype);
  return STLLM_ERR_BAD_DTYPE;
}


extern "C" int64_t stllm_attention_decode_workspace_bytes(int B, int H, int Skv) {
  if (B <= 0 || H <= 0 || Skv <= 0) return -1;
  return (int64_t)B * H * decode_splits(Skv) * kDecRec * 4;
}

extern "C" int stllm_attention_decode(int dtype, const void* q, int64_t q_bs, const void* k, int64_t k_bs, int64_t k_rs,
                                      const void* v, int64_t v_bs, int64_t v_rs, void* out, int64_t o_bs, int B, int H, int Skv,
                                      int D, float scale, void* workspace, int64_t workspace_bytes, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  STLLM_CHECK_ARG(q && k && v && out && workspace, "stllm_attention_decode: null pointer");
  STLLM_CHECK_ARG(B > 0 && H > 0 && Skv > 0, "stllm_attention_decode: empty problem");
  STLLM_CHECK_ARG(D == kDecD, "stllm_attention_decode: head_dim %d != 128", D);
  STLLM_CHECK_ARG(dtype == STLLM_BF16 || dtype == STLLM_F16, "stllm_attention_decode: 16-bit dtypes only");
  STLLM_CHECK_ARG(((uintptr_t)q & 3) == 0 && ((uintptr_t)k & 3) == 0 && ((uintptr_t)v & 3) == 0 && ((uintptr_t)out & 3) == 0 &&
                      q_bs % 2 == 0 && k_bs % 2 == 0 && k_rs % 2 == 0 && v_bs % 2 == 0 && v_rs % 2 == 0 && o_bs % 2 == 0,
                  "stllm_attention_decode: pointers / strides must be 4-byte aligned");
  const int nsplit = decode_splits(Skv);
  STLLM_CHECK_ARG(workspace_bytes >= stllm_attention_decode_workspace_bytes(B, H, Skv) && aligned16(workspace),
                  "stllm_attention_decode: workspace too small or misaligned");
  AttnParams p{};
  p.q = (const char*)q; p.q_bs = q_bs;
  p.k = (const char*)k; p.k_bs = k_bs; p.k_rs = k_rs;
  p.v = (const char*)v; p.v_bs = v_bs; p.v_rs = v_rs;
  p.o = (char*)out; p.o_bs = o_bs;
  p.B = B; p.H = H; p.Sq = 1; p.Skv = Skv; p.D = D;
  p.scale = scale; p.scale_log2 = scale * 1.44269504088896340736f;
  if ((Skv <= kDecSingleMax || B * H >= 256) && aligned16(q) && aligned16(k) && aligned16(v) && q_bs % 8 == 0 && k_bs % 8 == 0 &&
      k_rs % 8 == 0 && v_bs % 8 == 0 && v_rs % 8 == 0 && stllm_options().attn_decode_single) {
    if (dtype == STLLM_BF16) hipLaunchKernelGGL(attn_decode_single_kernel<bf16_t>, dim3(H, B), dim3(64 * kDecWaves), 0, stream, p);
    else hipLaunchKernelGGL(attn_decode_single_kernel<f16_t>, dim3(H, B), dim3(64 * kDecWaves), 0, stream, p);
    STLLM_CHECK_LAUNCH("stllm_attention_decode(single)");
    return STLLM_OK;
  }
  float* ws = reinterpret_cast<float*>(workspace);
  const int kps = (Skv + nsplit - 1) / nsplit;
  if (dtype == STLLM_BF16) {
    hipLaunchKernelGGL(attn_decode_partial_kernel<bf16_t>, dim3(nsplit, H, B), dim3(256), 0, stream, p, ws, nsplit, kps);
    hipLaunchKernelGGL(attn_decode_merge_kernel<bf16_t>, dim3(H, B), dim3(64), 0, stream, p, ws, nsplit);
  } else {
    hipLaunchKernelGGL(attn_decode_partial_kernel<f16_t>, dim3(nsplit, H, B), dim3(256), 0, stream, p, ws, nsplit, kps);
    hipLaunchKernelGGL(attn_decode_merge_kernel<f16_t>, dim3(H, B), dim3(64), 0, stream, p, ws, nsplit);
  }
  STLLM_CHECK_LAUNCH("stllm_attention_decode");
  return STLLM_OK;
}

#ifdef STLLM_ATTN_TRACE
// trace build only: the stamps of the last launch(es) -> host (tools/attn_trace.py)
extern "C" int stllm_attn_trace_read(void* dst, int64_t bytes, int clear) {
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_attn_trace), (size_t)bytes) != hipSuccess) return -2;
  if (clear) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_attn_trace)) != hipSuccess || hipMemset(p, 0, sizeof(g_attn_trace)) != hipSuccess) return -3;
  }
  return 0;
}
#endif
