// thread-local last-error string + ABI version for libstllm_hip.so
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <thread>
#include <vector>

#include "../../include/stllm_hip.h"
#include "options.h"

void stllm_set_error(const char* fmt, ...);

static thread_local char g_err[512] = "";

void stllm_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* stllm_last_error(void) { return g_err; }
extern "C" int stllm_abi_version(void) { return 7; }   // 7 (round 6): stllm_gemm_args.w_frag, stllm_llama_layer_weights.wqkv_frag / wgu_frag (W-direct GEMM).  6 (round 5): stllm_qformer_layers (the third whole-stack family).  5 (round 4): STLLM_BF16X3 + stllm_gemm_args.split_ws / stllm_split3_rows / stllm_gemm_split_ws_bytes; the fold_* fields, stllm_row_stats and stllm_gemm_fold_supported of ABI 4 are gone.  2: stllm_gemm_args gained the trailing a_norm_* fields (round 2); 3: whole-stack entry points, stllm_gemm_profile*, per-thread options; 4: stllm_gemm_args fold_* (LayerNorm folded into the GEMMs), stllm_row_stats (round 3)

// ---- per-thread dispatch options (common.h) ----
static int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}
StllmOptions& stllm_options() {
  static thread_local StllmOptions o;
  static thread_local bool init = false;
  if (!init) {   // every variable in ONE place, before the first dispatch decision of this thread
    o.gemm_sk = env_int("STLLM_GEMM_SK", -1);
    o.gemm_debug = env_int("STLLM_GEMM_DEBUG", 0);
    o.gemm_gemv = env_int("STLLM_GEMM_GEMV", -1);
    o.gemm_p8 = env_int("STLLM_GEMM_P8", -1);
    o.gemm_w4 = env_int("STLLM_GEMM_W4", -1);
    o.gemv_mfma = env_int("STLLM_GEMV_MFMA", -1);
    o.attn_decode_single = 1;
    o.attn_dma = env_int("STLLM_ATTN_DMA", 1);
    o.attn_bwd_valu = env_int("STLLM_ATTN_BWD_VALU", 0);
    o.norm_fast = env_int("STLLM_NORM_FAST", 1);
    o.gemm_t1 = env_int("STLLM_GEMM_T1", -1);
    o.gemm_wd = env_int("STLLM_GEMM_WD", -1);
    o.attn_q_lds = env_int("STLLM_ATTN_Q_LDS", 1);
    o.gemm_w4_odd = env_int("STLLM_GEMM_W4_ODD", 1);
    o.gemm_w4_wide = env_int("STLLM_GEMM_W4_WIDE", 1);
    o.attn_f32_mfma = env_int("STLLM_ATTN_F32_MFMA", 1);
    init = true;
  }
  return o;
}

extern "C" int stllm_set_option(const char* key, int value) {
  if (!key) return STLLM_ERR_BAD_SHAPE;
  StllmOptions& o = stllm_options();
  if (!strcmp(key, "gemm_sk")) { o.gemm_sk = value; return STLLM_OK; }
  if (!strcmp(key, "gemm_debug")) { o.gemm_debug = value; return STLLM_OK; }
  if (!strcmp(key, "gemm_p8")) { o.gemm_p8 = value; return STLLM_OK; }
  if (!strcmp(key, "gemm_w4")) { o.gemm_w4 = value; return STLLM_OK; }
  if (!strcmp(key, "gemm_gemv")) { o.gemm_gemv = value; return STLLM_OK; }
  if (!strcmp(key, "gemv_mfma")) { o.gemv_mfma = value; return STLLM_OK; }
  if (!strcmp(key, "attn_decode_single")) { o.attn_decode_single = value; return STLLM_OK; }
  if (!strcmp(key, "attn_dma")) { o.attn_dma = value; return STLLM_OK; }
  if (!strcmp(key, "attn_bwd_valu")) { o.attn_bwd_valu = value; return STLLM_OK; }
  if (!strcmp(key, "norm_fast")) { o.norm_fast = value; return STLLM_OK; }
  if (!strcmp(key, "gemm_t1")) { o.gemm_t1 = value; return STLLM_OK; }
  if (!strcmp(key, "gemm_wd")) { o.gemm_wd = value; return STLLM_OK; }
  if (!strcmp(key, "attn_q_lds")) { o.attn_q_lds = value; return STLLM_OK; }
  if (!strcmp(key, "gemm_w4_odd")) { o.gemm_w4_odd = value; return STLLM_OK; }
  if (!strcmp(key, "gemm_w4_wide")) { o.gemm_w4_wide = value; return STLLM_OK; }
  if (!strcmp(key, "attn_f32_mfma")) { o.attn_f32_mfma = value; return STLLM_OK; }
  stllm_set_error("stllm_set_option: unknown key %s", key);
  return STLLM_ERR_UNSUPPORTED;
}

static thread_local const char* g_last_kernel = "";
void stllm_set_last_kernel(const char* name) { g_last_kernel = name; }
extern "C" const char* stllm_last_kernel(void) { return g_last_kernel; }

// ---- host side of the synthetic-weight generator (stllm_amd/synth.py) -----------------------------------------------------------
// out[i] = ((bytesum(a) + bytesum(b)) - 1020) * scale (+ mean), a = hash32((start + i) ^ key), b = hash32(a + 0x68E31DA4 + start + i):
// exactly the integer recipe synth.normal_ runs with torch ops (on the GPU for the product, on the CPU for the oracle / fixtures) —
// the torch version manages ~20 M elements/s on a host core, which made the CPU test-suite spend most of its time generating
// the same full-width tensors; this loop does ~1 G/s.  Exact integer ops + one IEEE multiply (+ one add): bit-identical.
static inline uint32_t synth_hash32(uint32_t x) {
  x = ((x >> 16) ^ x) * 0x45D9F3Bu;
  x = ((x >> 16) ^ x) * 0x45D9F3Bu;
  return (x >> 16) ^ x;
}
static inline uint32_t synth_bytesum(uint32_t u) { return (u & 0xFF) + ((u >> 8) & 0xFF) + ((u >> 16) & 0xFF) + (u >> 24); }

extern "C" int stllm_synth_normal_f32(float* out, int64_t n, int64_t start, uint32_t key, float scale, float mean) {
  if (!out || n < 0 || start < 0) { stllm_set_error("stllm_synth_normal_f32: bad arguments"); return STLLM_ERR_BAD_SHAPE; }
  auto work = [=](int64_t lo, int64_t hi) {
    for (int64_t i = lo; i < hi; ++i) {
      const uint64_t idx = (uint64_t)(start + i);
      const uint32_t a = synth_hash32((uint32_t)((idx ^ (uint64_t)key) & 0xFFFFFFFFull));
      const uint32_t b = synth_hash32((uint32_t)(((uint64_t)a + 0x68E31DA4ull + idx) & 0xFFFFFFFFull));
      const float z = (float)(int)(synth_bytesum(a) + synth_bytesum(b)) - 1020.0f;
      float v = z * scale;
      if (mean != 0.0f) v = v + mean;
      out[i] = v;
    }
  };
  const int nt = n >= (1 << 22) ? 8 : 1;
  if (nt == 1) { work(0, n); return STLLM_OK; }
  std::vector<std::thread> th;
  const int64_t per = (n + nt - 1) / nt;
  for (int t = 0; t < nt; ++t) {
    const int64_t lo = t * per, hi = lo + per < n ? lo + per : n;
    if (lo < hi) th.emplace_back(work, lo, hi);
  }
  for (auto& t : th) t.join();
  return STLLM_OK;
}
