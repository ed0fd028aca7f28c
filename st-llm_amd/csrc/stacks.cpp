// Whole-stack entry points (SURVEY.md §8b "whole-stack entry points taking a packed-weights handle"): the ViT block loop and the Llama
// decoder-layer loop issued from ONE C call each, so that a step costs a handful of host calls instead of ~540 ctypes round trips and
// the launch queue never runs dry behind the host.  Pure host code: every kernel goes through the same public entry points
// (stllm_layernorm / stllm_rmsnorm / stllm_gemm / stllm_attention) with the same arguments the per-op Python path passes —
// results are bit-identical to that path by construction.
#include <stdint.h>
#include <string.h>

#include "../../include/stllm_hip.h"

void stllm_set_error(const char* fmt, ...);

namespace {

inline int64_t up256(int64_t v) { return (v + 255) / 256 * 256; }
inline int esize(int dtype) { return (dtype == STLLM_F32 || dtype == STLLM_BF16X3) ? 4 : 2; }   // activations between the GEMMs
inline int act_dtype(int dtype) { return dtype == STLLM_BF16X3 ? STLLM_F32 : dtype; }            // split mode: fp32 norms / attention
inline bool dtype_ok(int dtype) { return dtype >= STLLM_BF16 && dtype <= STLLM_BF16X3; }

struct Carver {
  char* base;
  int64_t off = 0, cap;
  Carver(void* p, int64_t bytes) : base(reinterpret_cast<char*>(p)), cap(bytes) {}
  void* take(int64_t bytes) {
    void* r = base + off;
    off += up256(bytes);
    return r;
  }
  bool ok() const { return off <= cap; }
};

stllm_gemm_args gemm_base(int dtype, void* ws, int64_t ws_bytes, void* split_ws, int64_t split_ws_bytes) {
  stllm_gemm_args g;
  memset(&g, 0, sizeof(g));
  g.dtype = dtype;
  g.workspace = ws;
  g.workspace_bytes = ws_bytes;
  g.split_ws = split_ws;
  g.split_ws_bytes = split_ws_bytes;
  return g;
}

// Split verify mode (STLLM_BF16X3): the norms write the split image of their output straight away (h: bf16 [M, 3 D]), fc1 / gate-up write the
// split image of their activation (STLLM_SPLIT_OUT), so only the attention outputs (fp32 from the attention kernel) are split inside their GEMM.
// split workspace of a stack = the larger of: the split A operand of proj / o_proj, the fp32 temporary of fc1 / gate-up.
inline int64_t vit_split_ws(int M, int dim, int hidden) {
  const int64_t a = stllm_gemm_split_ws_bytes(M, dim, dim, STLLM_EPI_RESID, 0), b = stllm_gemm_split_ws_bytes(M, hidden, dim, STLLM_EPI_STORE, STLLM_SPLIT_A_PRESPLIT | STLLM_SPLIT_OUT);
  return a > b ? a : b;
}
inline int64_t llama_split_ws(int M, int hidden, int inter) {
  const int64_t a = stllm_gemm_split_ws_bytes(M, hidden, hidden, STLLM_EPI_RESID, 0), b = stllm_gemm_split_ws_bytes(M, 2 * inter, hidden, STLLM_EPI_SWIGLU, STLLM_SPLIT_A_PRESPLIT | STLLM_SPLIT_OUT);
  return a > b ? a : b;
}
inline int64_t hsize(int dtype) { return dtype == STLLM_BF16X3 ? 6 : esize(dtype); }   // bytes per element of a GEMM's A operand written by a norm / an activation

}  // namespace

#define STACK_TRY(call)          \
  do {                           \
    const int rc_ = (call);      \
    if (rc_ != STLLM_OK) return rc_; \
  } while (0)

extern "C" int64_t stllm_vit_blocks_scratch_bytes(int dtype, int n_seq, int seq_len, int dim, int hidden) {
  if (n_seq <= 0 || seq_len <= 0 || dim <= 0 || hidden <= 0 || !dtype_ok(dtype)) return -1;
  const int64_t M = (int64_t)n_seq * seq_len, e = esize(dtype), eh = hsize(dtype);
  int64_t need = up256(M * dim * eh) + up256(M * dim * e) + up256(M * 3 * dim * e) + up256(M * hidden * eh);
  if (dtype == STLLM_BF16X3) need += up256(vit_split_ws((int)M, dim, hidden));
  return need;
}

// eva_vit.py:173-180 (Block.forward, gamma_1 / gamma_2 None) x n_blocks on the flat fp32 stream, in place:
//   LN -> [qkv GEMM + (q_bias, 0, v_bias)] -> attention (scale head_dim^-0.5) -> [proj GEMM + bias + residual] -> LN ->
//   [fc1 GEMM + bias + GELU] -> [fc2 GEMM + bias + residual]
extern "C" int stllm_vit_blocks(const stllm_vit_blocks_args* a, const stllm_vit_block_weights* blocks, int n_blocks, void* stream) {
  if (!a || (!blocks && n_blocks > 0) || n_blocks < 0) { stllm_set_error("stllm_vit_blocks: null arguments"); return STLLM_ERR_BAD_SHAPE; }
  if (!dtype_ok(a->dtype)) { stllm_set_error("stllm_vit_blocks: bad dtype %d", a->dtype); return STLLM_ERR_BAD_DTYPE; }
  if (a->num_heads <= 0 || a->dim <= 0 || a->hidden <= 0 || a->n_seq <= 0 || a->seq_len <= 0 || a->dim % a->num_heads != 0 || a->ldx < a->dim || !a->x || !a->scratch) {
    stllm_set_error("stllm_vit_blocks: bad dims (heads %d, dim %d, hidden %d, ldx %lld) / null buffers", a->num_heads, a->dim, a->hidden, (long long)a->ldx);
    return STLLM_ERR_BAD_SHAPE;
  }
  const int64_t need = stllm_vit_blocks_scratch_bytes(a->dtype, a->n_seq, a->seq_len, a->dim, a->hidden);
  if (need < 0 || a->scratch_bytes < need) {
    stllm_set_error("stllm_vit_blocks: scratch of %lld bytes needed, %lld given", (long long)need, (long long)a->scratch_bytes);
    return STLLM_ERR_BAD_SHAPE;
  }
  const int M = a->n_seq * a->seq_len, D = a->dim, hd = D / a->num_heads, e = esize(a->dtype), eh = (int)hsize(a->dtype);
  const bool x3 = a->dtype == STLLM_BF16X3;
  const int adt = act_dtype(a->dtype);   // attention: the activations' dtype (fp32 in the split mode)
  const int64_t ldh = x3 ? 3 * (int64_t)D : D, ldg = x3 ? 3 * (int64_t)a->hidden : a->hidden;   // row strides of the norm / GELU outputs (split images in the split mode)
  Carver c(a->scratch, a->scratch_bytes);
  char* h = reinterpret_cast<char*>(c.take((int64_t)M * D * eh));
  char* att = reinterpret_cast<char*>(c.take((int64_t)M * D * e));
  char* qkv = reinterpret_cast<char*>(c.take((int64_t)M * 3 * D * e));
  char* g1 = reinterpret_cast<char*>(c.take((int64_t)M * a->hidden * eh));
  void* sws = nullptr;
  int64_t sws_bytes = 0;
  if (x3) {
    sws_bytes = vit_split_ws(M, D, a->hidden);
    sws = c.take(sws_bytes);
  }
  float scale = 1.0f;
  {   // hd ** -0.5 exactly as the host path computes it (Python float -> float32)
    double s = 1.0;
    s = 1.0 / __builtin_sqrt((double)hd);
    scale = (float)s;
  }
  for (int b = 0; b < n_blocks; ++b) {
    const stllm_vit_block_weights& w = blocks[b];
    stllm_gemm_args g = gemm_base(a->dtype, a->workspace, a->workspace_bytes, sws, sws_bytes);
    STACK_TRY(stllm_layernorm(a->dtype, a->x, a->ldx, w.n1w, w.n1b, w.e1, h, ldh, nullptr, 0, M, D, stream));
    g.epilogue = STLLM_EPI_STORE; g.A = h; g.lda = ldh; g.W = w.wqkv; g.ldw = w.ld_qkv; g.bias = w.bqkv;
    if (x3) g.split_flags = STLLM_SPLIT_A_PRESPLIT;
    g.out = qkv; g.ldo = 3 * D; g.M = M; g.N = 3 * D; g.K = D;
    STACK_TRY(stllm_gemm(&g, stream));
    const int64_t rs = 3 * D, bs = (int64_t)a->seq_len * rs;
    STACK_TRY(stllm_attention(adt, qkv, bs, rs, qkv + (int64_t)D * e, bs, rs, qkv + (int64_t)2 * D * e, bs, rs, att,
                              (int64_t)a->seq_len * D, D, a->n_seq, a->num_heads, a->seq_len, a->seq_len, hd, scale, 0, nullptr, stream));
    g = gemm_base(a->dtype, a->workspace, a->workspace_bytes, sws, sws_bytes);
    g.epilogue = STLLM_EPI_RESID; g.A = att; g.lda = D; g.W = w.wproj; g.ldw = w.ld_proj; g.bias = w.bproj;
    g.out = a->x; g.ldo = a->ldx; g.resid = a->x; g.ldr = a->ldx; g.M = M; g.N = D; g.K = D;
    STACK_TRY(stllm_gemm(&g, stream));
    g = gemm_base(a->dtype, a->workspace, a->workspace_bytes, sws, sws_bytes);
    STACK_TRY(stllm_layernorm(a->dtype, a->x, a->ldx, w.n2w, w.n2b, w.e2, h, ldh, nullptr, 0, M, D, stream));
    g.epilogue = STLLM_EPI_STORE; g.act = STLLM_ACT_GELU; g.A = h; g.lda = ldh; g.W = w.wfc1; g.ldw = w.ld_fc1; g.bias = w.bfc1;
    if (x3) g.split_flags = STLLM_SPLIT_A_PRESPLIT | STLLM_SPLIT_OUT;   // GELU(fc1) leaves as the split A operand of fc2
    g.out = g1; g.ldo = ldg; g.M = M; g.N = a->hidden; g.K = D;
    STACK_TRY(stllm_gemm(&g, stream));
    g = gemm_base(a->dtype, a->workspace, a->workspace_bytes, sws, sws_bytes);
    g.epilogue = STLLM_EPI_RESID; g.A = g1; g.lda = ldg; g.W = w.wfc2; g.ldw = w.ld_fc2; g.bias = w.bfc2;
    if (x3) g.split_flags = STLLM_SPLIT_A_PRESPLIT;
    g.out = a->x; g.ldo = a->ldx; g.resid = a->x; g.ldr = a->ldx; g.M = M; g.N = D; g.K = a->hidden;
    STACK_TRY(stllm_gemm(&g, stream));
  }
  return STLLM_OK;
}

extern "C" int64_t stllm_llama_layers_scratch_bytes(int dtype, int B, int S, int hidden, int inter) {
  if (B <= 0 || S <= 0 || hidden <= 0 || inter <= 0 || !dtype_ok(dtype)) return -1;
  const int64_t M = (int64_t)B * S, e = esize(dtype), eh = hsize(dtype);
  int64_t need = up256(M * hidden * eh) + up256(M * hidden * e) + up256(M * 3 * hidden * e) + up256(M * inter * eh);
  if (dtype == STLLM_BF16X3) need += up256(llama_split_ws((int)M, hidden, inter));
  return need;
}

// HF LlamaDecoderLayer x n_layers in prefill form (spec modeling_llama_mem.py:61-316) on the flat fp32 stream, in place:
//   RMSNorm -> [fused QKV GEMM + rotate-half RoPE] -> causal attention (+ right-padding kv_len) -> [o_proj + residual] -> RMSNorm ->
//   [gate/up GEMM + SiLU(gate) * up] -> [down_proj + residual].   With kv_cache pointers the fused QKV rows of layer l are written
//   into its cache buffer [B, cache_max_len, 3 * hidden] (rows (b, s) at b * cache_max_len + s) and attended in place.
extern "C" int stllm_llama_layers(const stllm_llama_layers_args* a, const stllm_llama_layer_weights* layers, int n_layers, void* stream) {
  if (!a || (!layers && n_layers > 0) || n_layers < 0) { stllm_set_error("stllm_llama_layers: null arguments"); return STLLM_ERR_BAD_SHAPE; }
  if (!dtype_ok(a->dtype)) { stllm_set_error("stllm_llama_layers: bad dtype %d", a->dtype); return STLLM_ERR_BAD_DTYPE; }
  if (a->n_heads <= 0 || a->hidden <= 0 || a->inter <= 0 || a->B <= 0 || a->S <= 0 || a->hidden % a->n_heads != 0 || a->ldx < a->hidden ||
      !a->x || !a->scratch || !a->rope_cos || !a->rope_sin) {
    stllm_set_error("stllm_llama_layers: bad dims (heads %d, hidden %d, inter %d, ldx %lld) / null buffers", a->n_heads, a->hidden, a->inter, (long long)a->ldx);
    return STLLM_ERR_BAD_SHAPE;
  }
  const int64_t need = stllm_llama_layers_scratch_bytes(a->dtype, a->B, a->S, a->hidden, a->inter);
  if (need < 0 || a->scratch_bytes < need) {
    stllm_set_error("stllm_llama_layers: scratch of %lld bytes needed, %lld given", (long long)need, (long long)a->scratch_bytes);
    return STLLM_ERR_BAD_SHAPE;
  }
  const int M = a->B * a->S, D = a->hidden, hd = D / a->n_heads, e = esize(a->dtype), eh = (int)hsize(a->dtype);
  const bool x3 = a->dtype == STLLM_BF16X3;
  const int64_t ldh = x3 ? 3 * (int64_t)D : D, ldg = x3 ? 3 * (int64_t)a->inter : a->inter;
  if (a->cache_max_len != 0 && (a->cache_max_len < a->S || a->kv_len != nullptr)) {
    stllm_set_error("stllm_llama_layers: the KV cache needs max_len >= S and equal-length sequences");
    return STLLM_ERR_BAD_SHAPE;
  }
  Carver c(a->scratch, a->scratch_bytes);
  char* h = reinterpret_cast<char*>(c.take((int64_t)M * D * eh));
  char* att = reinterpret_cast<char*>(c.take((int64_t)M * D * e));
  char* qkv_s = reinterpret_cast<char*>(c.take((int64_t)M * 3 * D * e));
  char* gu = reinterpret_cast<char*>(c.take((int64_t)M * a->inter * eh));
  void* sws = nullptr;
  int64_t sws_bytes = 0;
  if (a->dtype == STLLM_BF16X3) {
    sws_bytes = llama_split_ws(M, D, a->inter);
    sws = c.take(sws_bytes);
  }
  const int adt = act_dtype(a->dtype);
  const float scale = (float)(1.0 / __builtin_sqrt((double)hd));
  for (int l = 0; l < n_layers; ++l) {
    const stllm_llama_layer_weights& w = layers[l];
    STACK_TRY(stllm_rmsnorm(a->dtype, a->x, a->ldx, w.ln1, a->eps, h, ldh, nullptr, 0, M, D, stream));
    stllm_gemm_args g = gemm_base(a->dtype, a->workspace, a->workspace_bytes, sws, sws_bytes);
    g.epilogue = STLLM_EPI_ROPE; g.A = h; g.lda = ldh; g.W = w.wqkv; g.ldw = w.ld_qkv;
    if (x3) g.split_flags = STLLM_SPLIT_A_PRESPLIT;
    else g.w_frag = w.wqkv_frag;
    g.aux0 = a->rope_cos; g.aux1 = a->rope_sin; g.rope_seq = a->S; g.rope_cols = 2 * D; g.M = M; g.N = 3 * D; g.K = D; g.ldo = 3 * D;
    char* qkv = qkv_s;
    int64_t bs = (int64_t)a->S * 3 * D;
    if (a->cache_max_len != 0) {
      if (!w.kv_cache) { stllm_set_error("stllm_llama_layers: layer %d has no cache buffer", l); return STLLM_ERR_BAD_SHAPE; }
      qkv = reinterpret_cast<char*>(w.kv_cache);
      bs = a->cache_max_len * 3 * D;
      g.o_rows_per_batch = a->S; g.o_batch_stride = bs;
    }
    g.out = qkv;
    STACK_TRY(stllm_gemm(&g, stream));
    const int64_t rs = 3 * D;
    STACK_TRY(stllm_attention(adt, qkv, bs, rs, qkv + (int64_t)D * e, bs, rs, qkv + (int64_t)2 * D * e, bs, rs, att,
                              (int64_t)a->S * D, D, a->B, a->n_heads, a->S, a->S, hd, scale, 1, a->kv_len, stream));
    g = gemm_base(a->dtype, a->workspace, a->workspace_bytes, sws, sws_bytes);
    g.epilogue = STLLM_EPI_RESID; g.A = att; g.lda = D; g.W = w.wo; g.ldw = w.ld_o;
    g.out = a->x; g.ldo = a->ldx; g.resid = a->x; g.ldr = a->ldx; g.M = M; g.N = D; g.K = D;
    STACK_TRY(stllm_gemm(&g, stream));
    STACK_TRY(stllm_rmsnorm(a->dtype, a->x, a->ldx, w.ln2, a->eps, h, ldh, nullptr, 0, M, D, stream));
    g = gemm_base(a->dtype, a->workspace, a->workspace_bytes, sws, sws_bytes);
    g.epilogue = STLLM_EPI_SWIGLU; g.A = h; g.lda = ldh; g.W = w.wgu; g.ldw = w.ld_gu;
    if (x3) g.split_flags = STLLM_SPLIT_A_PRESPLIT | STLLM_SPLIT_OUT;   // SiLU(gate) * up leaves as the split A operand of down_proj
    else g.w_frag = w.wgu_frag;
    g.out = gu; g.ldo = ldg; g.M = M; g.N = 2 * a->inter; g.K = D;
    STACK_TRY(stllm_gemm(&g, stream));
    g = gemm_base(a->dtype, a->workspace, a->workspace_bytes, sws, sws_bytes);
    g.epilogue = STLLM_EPI_RESID; g.A = gu; g.lda = ldg; g.W = w.wdown; g.ldw = w.ld_down;
    if (x3) g.split_flags = STLLM_SPLIT_A_PRESPLIT;
    g.out = a->x; g.ldo = a->ldx; g.resid = a->x; g.ldr = a->ldx; g.M = M; g.N = D; g.K = a->inter;
    STACK_TRY(stllm_gemm(&g, stream));
  }
  return STLLM_OK;
}

extern "C" int64_t stllm_llama_layer_sp_scratch_bytes(int dtype, int s0, int s1, int hidden, int inter) {
  if (s0 < 0 || s1 <= s0 || hidden <= 0 || inter <= 0 || !dtype_ok(dtype) || dtype == STLLM_BF16X3) return -1;
  const int64_t n = s1 - s0, e = esize(dtype);
  return up256(n * hidden * e) + up256((int64_t)s1 * hidden * e) + up256(n * inter * e);
}

// One decoder layer of the sequence-parallel prefill, in the two parts between which the team's K | V rows travel (stllm_hip.h).
extern "C" int stllm_llama_layer_sp(const stllm_llama_layers_args* a, const stllm_llama_layer_weights* wp, void* qkv_, int s0, int s1, int part, void* stream) {
  if (!a || !wp || !qkv_) { stllm_set_error("stllm_llama_layer_sp: null arguments"); return STLLM_ERR_BAD_SHAPE; }
  if (!dtype_ok(a->dtype) || a->dtype == STLLM_BF16X3) { stllm_set_error("stllm_llama_layer_sp: dtype %d (the split mode runs the per-op path)", a->dtype); return STLLM_ERR_BAD_DTYPE; }
  const int n = s1 - s0;
  if (a->B != 1 || a->S != n || n <= 0 || s0 < 0 || a->n_heads <= 0 || a->hidden <= 0 || a->inter <= 0 || a->hidden % a->n_heads != 0 || a->ldx < a->hidden ||
      !a->x || !a->scratch || !a->rope_cos || !a->rope_sin || a->kv_len || a->cache_max_len != 0 || (part != 0 && part != 1)) {
    stllm_set_error("stllm_llama_layer_sp: bad arguments (B %d, S %d, rows [%d, %d), part %d)", a->B, a->S, s0, s1, part);
    return STLLM_ERR_BAD_SHAPE;
  }
  const int64_t need = stllm_llama_layer_sp_scratch_bytes(a->dtype, s0, s1, a->hidden, a->inter);
  if (need < 0 || a->scratch_bytes < need) {
    stllm_set_error("stllm_llama_layer_sp: scratch of %lld bytes needed, %lld given", (long long)need, (long long)a->scratch_bytes);
    return STLLM_ERR_BAD_SHAPE;
  }
  const stllm_llama_layer_weights& w = *wp;
  const int D = a->hidden, hd = D / a->n_heads, e = esize(a->dtype);
  Carver c(a->scratch, a->scratch_bytes);
  char* h = reinterpret_cast<char*>(c.take((int64_t)n * D * e));
  char* att = reinterpret_cast<char*>(c.take((int64_t)s1 * D * e));
  char* gu = reinterpret_cast<char*>(c.take((int64_t)n * a->inter * e));
  char* qkv = reinterpret_cast<char*>(qkv_);
  const int64_t rs = 3 * (int64_t)D;
  if (part == 0) {
    STACK_TRY(stllm_rmsnorm(a->dtype, a->x, a->ldx, w.ln1, a->eps, h, D, nullptr, 0, n, D, stream));
    stllm_gemm_args g = gemm_base(a->dtype, a->workspace, a->workspace_bytes, nullptr, 0);
    g.epilogue = STLLM_EPI_ROPE; g.A = h; g.lda = D; g.W = w.wqkv; g.ldw = w.ld_qkv; g.w_frag = w.wqkv_frag;
    g.aux0 = a->rope_cos; g.aux1 = a->rope_sin; g.rope_seq = n; g.rope_cols = 2 * D; g.M = n; g.N = 3 * D; g.K = D; g.ldo = 3 * D;
    g.out = qkv + (int64_t)s0 * rs * e;
    STACK_TRY(stllm_gemm(&g, stream));
    return STLLM_OK;
  }
  const float scale = (float)(1.0 / __builtin_sqrt((double)hd));
  STACK_TRY(stllm_attention(a->dtype, qkv, (int64_t)s1 * rs, rs, qkv + (int64_t)D * e, (int64_t)s1 * rs, rs, qkv + (int64_t)2 * D * e, (int64_t)s1 * rs, rs, att,
                            (int64_t)s1 * D, D, 1, a->n_heads, s1, s1, hd, scale, 1, nullptr, stream));
  stllm_gemm_args g = gemm_base(a->dtype, a->workspace, a->workspace_bytes, nullptr, 0);
  g.epilogue = STLLM_EPI_RESID; g.A = att + (int64_t)s0 * D * e; g.lda = D; g.W = w.wo; g.ldw = w.ld_o;
  g.out = a->x; g.ldo = a->ldx; g.resid = a->x; g.ldr = a->ldx; g.M = n; g.N = D; g.K = D;
  STACK_TRY(stllm_gemm(&g, stream));
  STACK_TRY(stllm_rmsnorm(a->dtype, a->x, a->ldx, w.ln2, a->eps, h, D, nullptr, 0, n, D, stream));
  g = gemm_base(a->dtype, a->workspace, a->workspace_bytes, nullptr, 0);
  g.epilogue = STLLM_EPI_SWIGLU; g.A = h; g.lda = D; g.W = w.wgu; g.ldw = w.ld_gu; g.w_frag = w.wgu_frag;
  g.out = gu; g.ldo = a->inter; g.M = n; g.N = 2 * a->inter; g.K = D;
  STACK_TRY(stllm_gemm(&g, stream));
  g = gemm_base(a->dtype, a->workspace, a->workspace_bytes, nullptr, 0);
  g.epilogue = STLLM_EPI_RESID; g.A = gu; g.lda = a->inter; g.W = w.wdown; g.ldw = w.ld_down;
  g.out = a->x; g.ldo = a->ldx; g.resid = a->x; g.ldr = a->ldx; g.M = n; g.N = D; g.K = a->inter;
  STACK_TRY(stllm_gemm(&g, stream));
  return STLLM_OK;
}

namespace {

inline int64_t max64(int64_t a, int64_t b) { return a > b ? a : b; }

// split workspace of the Q-Former stack in the split mode: the largest split A operand among its GEMMs (none is pre-split)
int64_t qformer_split_ws(int64_t nq, int64_t nt, int64_t nenc, int C, int inter, int enc_dim, int n_cross) {
  const int64_t rows = max64(nq, nt);
  int64_t w = max64(stllm_gemm_split_ws_bytes((int)rows, 3 * C, C, STLLM_EPI_STORE, 0), stllm_gemm_split_ws_bytes((int)rows, C, C, STLLM_EPI_RESID, 0));
  w = max64(w, stllm_gemm_split_ws_bytes((int)rows, inter, C, STLLM_EPI_STORE, 0));
  w = max64(w, stllm_gemm_split_ws_bytes((int)rows, C, inter, STLLM_EPI_RESID, 0));
  if (n_cross > 0) w = max64(w, stllm_gemm_split_ws_bytes((int)nenc, n_cross * 2 * C, enc_dim, STLLM_EPI_STORE, 0));
  return w;
}

}  // namespace

extern "C" int64_t stllm_qformer_layers_scratch_bytes(int dtype, int n_seq, int n_query, int n_text, int hidden, int inter, int enc_len, int enc_dim, int n_cross) {
  if (n_seq <= 0 || n_query <= 0 || n_text < 0 || hidden <= 0 || inter <= 0 || enc_len < 0 || enc_dim < 0 || n_cross < 0 || !dtype_ok(dtype)) return -1;
  const int64_t e = esize(dtype), S = n_query + n_text, nq = (int64_t)n_seq * n_query, nt = (int64_t)n_seq * n_text, rows = max64(nq, nt);
  int64_t need = up256((int64_t)n_seq * S * 3 * hidden * e) + up256((int64_t)n_seq * S * hidden * e) + up256(rows * hidden * 4) + up256(rows * inter * e);
  if (n_cross > 0) need += 2 * up256(nq * hidden * e) + up256((int64_t)n_seq * enc_len * n_cross * 2 * hidden * e);
  if (dtype == STLLM_BF16X3) need += up256(qformer_split_ws(nq, nt, (int64_t)n_seq * enc_len, hidden, inter, enc_dim, n_cross));
  return need;
}

extern "C" int stllm_qformer_layers(const stllm_qformer_layers_args* a, const stllm_qformer_layer_weights* layers, int n_layers, void* stream) {
  if (!a || (!layers && n_layers > 0) || n_layers < 0) { stllm_set_error("stllm_qformer_layers: null arguments"); return STLLM_ERR_BAD_SHAPE; }
  if (!dtype_ok(a->dtype)) { stllm_set_error("stllm_qformer_layers: bad dtype %d", a->dtype); return STLLM_ERR_BAD_DTYPE; }
  if (a->n_seq <= 0 || a->n_query <= 0 || a->n_text < 0 || a->n_heads <= 0 || a->hidden <= 0 || a->inter <= 0 || a->hidden % a->n_heads != 0 || a->n_cross < 0 ||
      !a->hq32 || !a->hq16 || !a->scratch || (a->n_text > 0 && (!a->ht32 || !a->ht16)) ||
      (a->n_cross > 0 && (!a->enc16 || !a->ckv_w || a->enc_len <= 0 || a->enc_dim <= 0 || a->ld_enc < a->enc_dim))) {
    stllm_set_error("stllm_qformer_layers: bad dims (seqs %d, queries %d, text %d, heads %d, hidden %d, inter %d, image tokens %d x %d, cross layers %d) / null buffers",
                    a->n_seq, a->n_query, a->n_text, a->n_heads, a->hidden, a->inter, a->enc_len, a->enc_dim, a->n_cross);
    return STLLM_ERR_BAD_SHAPE;
  }
  int cross_seen = 0;
  for (int l = 0; l < n_layers; ++l) {
    const stllm_qformer_layer_weights& w = layers[l];
    const bool cross_bad = w.has_cross && (!w.cq_w || !w.cross_out.w || w.ckv_index < 0 || w.ckv_index >= a->n_cross);
    if (!w.wqkv || !w.attn_out.w || !w.fq_w1 || !w.fq_out.w || cross_bad || (a->n_text > 0 && (!w.ft_w1 || !w.ft_out.w))) {
      stllm_set_error("stllm_qformer_layers: layer %d lacks a weight the call needs (cross %d of %d, text rows %d)", l, w.has_cross ? w.ckv_index : -1, a->n_cross, a->n_text);
      return STLLM_ERR_BAD_SHAPE;
    }
    cross_seen += w.has_cross ? 1 : 0;
  }
  (void)cross_seen;
  const int64_t need = stllm_qformer_layers_scratch_bytes(a->dtype, a->n_seq, a->n_query, a->n_text, a->hidden, a->inter, a->enc_len, a->enc_dim, a->n_cross);
  if (need < 0 || a->scratch_bytes < need) {
    stllm_set_error("stllm_qformer_layers: scratch of %lld bytes needed, %lld given", (long long)need, (long long)a->scratch_bytes);
    return STLLM_ERR_BAD_SHAPE;
  }
  const int n = a->n_seq, Q = a->n_query, Lt = a->n_text, S = Q + Lt, C = a->hidden, H = a->n_heads, hd = C / H, P = a->enc_len;
  const int e = esize(a->dtype), adt = act_dtype(a->dtype);
  const int64_t nq = (int64_t)n * Q, nt = (int64_t)n * Lt, rows = max64(nq, nt);
  const int64_t ld_ckv_out = (int64_t)a->n_cross * 2 * C;
  Carver c(a->scratch, a->scratch_bytes);
  char* qkv = reinterpret_cast<char*>(c.take((int64_t)n * S * 3 * C * e));
  char* ctx = reinterpret_cast<char*>(c.take((int64_t)n * S * C * e));
  float* tmp = reinterpret_cast<float*>(c.take(rows * C * 4));
  char* g1 = reinterpret_cast<char*>(c.take(rows * a->inter * e));
  char *cq = nullptr, *cctx = nullptr, *ckv = nullptr;
  if (a->n_cross > 0) {
    cq = reinterpret_cast<char*>(c.take(nq * C * e));
    cctx = reinterpret_cast<char*>(c.take(nq * C * e));
    ckv = reinterpret_cast<char*>(c.take((int64_t)n * P * ld_ckv_out * e));
  }
  void* sws = nullptr;
  int64_t sws_bytes = 0;
  if (a->dtype == STLLM_BF16X3) {
    sws_bytes = qformer_split_ws(nq, nt, (int64_t)n * P, C, a->inter, a->enc_dim, a->n_cross);
    sws = c.take(sws_bytes);
  }
  const float scale = (float)(1.0 / __builtin_sqrt((double)hd));
  // LayerNorm(dense(x) + input) of `m` rows: x = A (2-level rows when a_rpb > 0), input / result = the fp32 stream h32 with its compute-dtype copy h16
  auto post_ln = [&](const void* A, int64_t lda, int a_rpb, int64_t a_bs, int K, const stllm_bert_output_weights& o, float* h32, void* h16, int64_t m) -> int {
    stllm_gemm_args g = gemm_base(a->dtype, a->workspace, a->workspace_bytes, sws, sws_bytes);
    g.epilogue = STLLM_EPI_RESID; g.A = A; g.lda = lda; g.a_rows_per_batch = a_rpb; g.a_batch_stride = a_bs; g.W = o.w; g.ldw = o.ldw; g.bias = o.b;
    g.resid = h32; g.ldr = C; g.out = tmp; g.ldo = C; g.M = (int)m; g.N = C; g.K = K;
    STACK_TRY(stllm_gemm(&g, stream));
    return stllm_layernorm(adt, tmp, C, o.g, o.beta, o.eps, h16, C, h32, C, (int)m, C, stream);
  };
  bool ckv_done = false;
  for (int l = 0; l < n_layers; ++l) {
    const stllm_qformer_layer_weights& w = layers[l];
    // ---- self-attention over [queries | text] (Qformer.py:417-424): both row groups write their rows of ONE fused [n, S, 3 C] buffer ----
    stllm_gemm_args g = gemm_base(a->dtype, a->workspace, a->workspace_bytes, sws, sws_bytes);
    g.epilogue = STLLM_EPI_STORE; g.A = a->hq16; g.lda = C; g.W = w.wqkv; g.ldw = w.ld_qkv; g.bias = w.bqkv; g.out = qkv; g.ldo = 3 * C; g.M = (int)nq; g.N = 3 * C; g.K = C;
    if (Lt) { g.o_rows_per_batch = Q; g.o_batch_stride = (int64_t)S * 3 * C; }
    STACK_TRY(stllm_gemm(&g, stream));
    if (Lt) {
      g.A = a->ht16; g.out = qkv + (int64_t)Q * 3 * C * e; g.M = (int)nt; g.o_rows_per_batch = Lt;
      STACK_TRY(stllm_gemm(&g, stream));
    }
    const int64_t rs = 3 * C, bs = (int64_t)S * rs;
    STACK_TRY(stllm_attention(adt, qkv, bs, rs, qkv + (int64_t)C * e, bs, rs, qkv + (int64_t)2 * C * e, bs, rs, ctx, (int64_t)S * C, C, n, H, S, S, hd, scale, 0, a->kv_len, stream));
    STACK_TRY(post_ln(ctx, C, Lt ? Q : 0, Lt ? (int64_t)S * C : 0, C, w.attn_out, a->hq32, a->hq16, nq));
    if (Lt) STACK_TRY(post_ln(ctx + (int64_t)Q * C * e, C, Lt, (int64_t)S * C, C, w.attn_out, a->ht32, a->ht16, nt));
    // ---- cross-attention, query rows only (Qformer.py:430-444) ----
    if (w.has_cross) {
      g = gemm_base(a->dtype, a->workspace, a->workspace_bytes, sws, sws_bytes);
      g.epilogue = STLLM_EPI_STORE; g.A = a->hq16; g.lda = C; g.W = w.cq_w; g.ldw = w.ld_cq; g.bias = w.cq_b; g.out = cq; g.ldo = C; g.M = (int)nq; g.N = C; g.K = C;
      STACK_TRY(stllm_gemm(&g, stream));
      if (!ckv_done) {   // the K / V projections of every cross layer in one GEMM over the image tokens
        g = gemm_base(a->dtype, a->workspace, a->workspace_bytes, sws, sws_bytes);
        g.epilogue = STLLM_EPI_STORE; g.A = a->enc16; g.lda = a->ld_enc; g.W = a->ckv_w; g.ldw = a->ld_ckv; g.bias = a->ckv_b; g.out = ckv; g.ldo = ld_ckv_out;
        g.M = n * P; g.N = (int)ld_ckv_out; g.K = a->enc_dim;
        STACK_TRY(stllm_gemm(&g, stream));
        ckv_done = true;
      }
      const char* kk = ckv + (int64_t)w.ckv_index * 2 * C * e;
      STACK_TRY(stllm_attention(adt, cq, (int64_t)Q * C, C, kk, (int64_t)P * ld_ckv_out, ld_ckv_out, kk + (int64_t)C * e, (int64_t)P * ld_ckv_out, ld_ckv_out, cctx, (int64_t)Q * C, C,
                                n, H, Q, P, hd, scale, 0, nullptr, stream));
      STACK_TRY(post_ln(cctx, C, 0, 0, C, w.cross_out, a->hq32, a->hq16, nq));
    }
    // ---- FFN: query rows through *_query, text rows through the text weights (Qformer.py:449-462) ----
    g = gemm_base(a->dtype, a->workspace, a->workspace_bytes, sws, sws_bytes);
    g.epilogue = STLLM_EPI_STORE; g.act = STLLM_ACT_GELU; g.A = a->hq16; g.lda = C; g.W = w.fq_w1; g.ldw = w.ld_fq1; g.bias = w.fq_b1; g.out = g1; g.ldo = a->inter;
    g.M = (int)nq; g.N = a->inter; g.K = C;
    STACK_TRY(stllm_gemm(&g, stream));
    STACK_TRY(post_ln(g1, a->inter, 0, 0, a->inter, w.fq_out, a->hq32, a->hq16, nq));
    if (Lt) {
      g.A = a->ht16; g.W = w.ft_w1; g.ldw = w.ld_ft1; g.bias = w.ft_b1; g.M = (int)nt;
      STACK_TRY(stllm_gemm(&g, stream));
      STACK_TRY(post_ln(g1, a->inter, 0, 0, a->inter, w.ft_out, a->ht32, a->ht16, nt));
    }
  }
  return STLLM_OK;
}
