// Shared between the GEMM translation units (gemm.hip: 128x128 persistent + stream-K kernels, gemm_p8_*.hip: the
// phased kernel).
#pragma once
#include "common.h"

namespace sg {

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
  char* ws; int64_t ws_bytes; int epoch;   // stream-K workspace: [4 KiB flags | per-workgroup fp32 slabs], launch epoch
  int debug;                       // ablation bits (env STLLM_GEMM_DEBUG): 1 skip staging, 2 skip MFMA loop, 4 skip copy-out
  int a_rpb; int64_t a_bs_b;       // A 2-level rows: rows per batch, batch stride (bytes)
  int o_rpb; int64_t o_bs;         // out 2-level rows (elements)
  int p8_q, p8_r, p8_s, p8_cap;    // phased kernel schedule: DP rounds, remainder tiles, K-slices per remainder tile, groups per XCD
  const float* nx; int64_t nx_ld; const float* ngamma; float neps;   // GEMV only: A := RMSNorm(nx) * ngamma (fp32 rows, stride nx_ld elements)
  const char* Wf;                  // gemm_wd only: fragment-major copy of W (pack.frag32: [N / 32][K / 16][64 lanes x 16 B]) or nullptr
  int w4_thin;                     // gemm_w4 only: the last M % tile_rows (<= 32) rows are computed outside the tile grid (0 = none)
};

constexpr int kRowBytes = 128;  // one K panel row
// XCD-aware bijective remap of the linear block id (guide §5: "XCD swizzle must be bijective")
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + k;
}

#ifndef STLLM_GROUP_M
#define STLLM_GROUP_M 8   // (experiment builds: -DSTLLM_GROUP_M=4, profiles/r06_group_m.md)
#endif
constexpr int kGroupM = STLLM_GROUP_M;  // tile rows per L2 locality group

// work id -> (tm, tn): groups of kGroupM tile rows, tm fastest inside a group.  With the XCD remap
// applied to the PERSISTENT block id, the 64 tiles an XCD runs concurrently form an ~8x8 patch that
// shares 8 A panels and 8 W panels in that XCD's private L2.
__device__ __forceinline__ void tile_coords(int w, int tiles_m, int tiles_n, int& tm, int& tn) {
  const int gsz = kGroupM * tiles_n;
  const int g = w / gsz, rem = w - g * gsz;
  const int first = g * kGroupM;
  const int gm = min(kGroupM, tiles_m - first);
  tn = rem / gm;
  tm = first + (rem - tn * gm);
}

constexpr int kSkFlagBytes = 4096;   // workspace: [1024 flag words | per-workgroup fp32 slabs]
constexpr int kSkErrWord = 1000;     // flag word set by a kernel whose bounded poll for a peer workgroup expired (stllm_gemm_workspace_status)

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

}  // namespace sg

// (64*MIW) x 256 phased kernel (gemm_p8.inc, one TU per 16-bit dtype); every epilogue except PATCH.
// Returns STLLM_OK, STLLM_ERR_UNSUPPORTED (caller falls back to the 128x128 kernels) or an error code.
int stllm_sk_next_epoch();   // gemm.hip: launch epoch shared by all kernels that use the workspace flag array
int stllm_gemm_p8_launch_bf16(int epilogue, int miw, const sg::GemmParams& p, hipStream_t stream);
int stllm_gemm_p8_launch_f16(int epilogue, int miw, const sg::GemmParams& p, hipStream_t stream);
// skinny GEMM of the decode regime (gemv.hip): M <= 4, 16-bit dtypes, every epilogue except PATCH
int stllm_gemv_launch(int dtype, int epilogue, const sg::GemmParams& p, hipStream_t stream);
float stllm_gemm_p8_estimate_us(int M, int N, int K, int heavy_epilogue, int* miw);   // cost model of the schedule; picks MIW (4: 256 rows, 3: 192 rows)
// one-wave-per-SIMD kernel (gemm_w4.inc): 4 waves x up to 256 accumulator registers, tile shape 34 (192 x 256) or 44 (256 x 256);
// same workspace, same epilogues and return codes as the phased kernel
int stllm_gemm_w4_launch_bf16(int epilogue, int shape, const sg::GemmParams& p, hipStream_t stream);
int stllm_gemm_w4_launch_f16(int epilogue, int shape, const sg::GemmParams& p, hipStream_t stream);
// tall-tile one-round kernel (gemm_t1.inc): 144-row tiles x 32 shape columns, whole K per workgroup, no workspace; STORE (no activation) / RESID
int stllm_gemm_t1_launch_bf16(int epilogue, int shape, const sg::GemmParams& p, hipStream_t stream);
int stllm_gemm_t1_launch_f16(int epilogue, int shape, const sg::GemmParams& p, hipStream_t stream);
// W-direct kernel (gemm_wd.inc): (32 shape) x 256 tiles, A through the LDS, W fragments straight into registers from GemmParams::Wf; STORE (no activation) / SWIGLU / ROPE
int stllm_gemm_wd_launch_bf16(int epilogue, int shape, const sg::GemmParams& p, hipStream_t stream);
int stllm_gemm_wd_launch_f16(int epilogue, int shape, const sg::GemmParams& p, hipStream_t stream);
float stllm_gemm_w4_estimate_us(int M, int N, int K, int heavy_epilogue, int* shape, int* split);   // split = K slices of the remainder tiles (1: none)
