/*
 * stllm_hip.h — C ABI of libstllm_hip.so: hand-written HIP kernels (gfx950 / MI355X) for the
 * ST-LLM video-token hot path (EVA-CLIP-g ViT -> Q-Former -> projector -> Vicuna-7B prefill).
 *
 * The reference (TencentARC/ST-LLM) has NO native/FFI layer for this path:
 * the arithmetic is torch.nn ops dispatched to vendor BLAS/DNN kernels.  Each entry point below
 * therefore cites the reference torch call site(s) it replaces (paths relative to the reference
 * root) rather than an existing FFI symbol.  The Python binding a maintainer would add is the
 * ctypes stub in st-llm_amd/hip.py (shown in INTEGRATION.md).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no torch types.  All data pointers are DEVICE pointers
 *     owned by the caller (torch tensors in the shipped host code); nothing is allocated inside.
 *   - every call takes the hipStream_t to launch on (void* here so the header needs no HIP include).
 *   - return 0 on success, negative stllm_status on error; stllm_last_error() gives the message
 *     (thread-local).  Kernels are launched asynchronously; launch errors are reported, execution
 *     errors surface at the caller's next synchronisation as usual.
 *   - "T" below is the compute dtype selected by `dtype`: bf16 / fp16 (MFMA 32x32x16, fp32
 *     accumulate) or fp32 (exact-fp32 MFMA 32x32x2 — the "verify" numerics mode).
 *   - the residual stream / normalisation inputs are always fp32 (SURVEY.md §7 hard-part 1).
 *   - matrices are row-major; weights are [N, K] (out_features, in_features) exactly as
 *     torch.nn.Linear stores them, possibly row-permuted by the packers in st-llm_amd/pack.py.
 */
#ifndef STLLM_HIP_H
#define STLLM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* STLLM_BF16X3 (ABI >= 5; stllm_gemm and the whole-stack entry points only): the "split" verify mode — fp32 activations and outputs, every GEMM
 * run as THREE bf16 matrix-core products (A_hi W_hi + A_hi W_lo + A_lo W_hi, x = hi + lo in bf16) in one bf16 GEMM with K' = 3 K; the
 * fp32 accuracy class (~2^-16 relative per product) at 3x the bf16 FLOPs instead of the 16x of the exact fp32 MFMA.  See stllm_gemm. */
typedef enum { STLLM_BF16 = 0, STLLM_F16 = 1, STLLM_F32 = 2, STLLM_BF16X3 = 3 } stllm_dtype;

typedef enum {
  STLLM_OK = 0,
  STLLM_ERR_BAD_SHAPE = -1,   /* size/stride/alignment constraint violated */
  STLLM_ERR_BAD_DTYPE = -2,
  STLLM_ERR_HIP = -3,         /* hip runtime error at launch */
  STLLM_ERR_UNSUPPORTED = -4
} stllm_status;

/* GEMM epilogues (fused into the producing kernel) */
typedef enum {
  STLLM_EPI_STORE = 0,   /* out = act(acc + bias), out dtype T or f32                              */
  STLLM_EPI_RESID = 1,   /* out_f32 = resid_f32 + acc + bias        (in place when out == resid)   */
  STLLM_EPI_SWIGLU = 2,  /* out[:, j] = silu(gate_j) * up_j; W rows packed [32 gate | 32 up] x N/64 */
  STLLM_EPI_ROPE = 3,    /* fused QKV: rotate-half RoPE on columns < rope_cols (packed layout)     */
  STLLM_EPI_PATCH = 4    /* patch-embed: out_f32[n*257+1+p] = acc + bias + pos_embed[1+p]          */
} stllm_epilogue;

typedef enum { STLLM_ACT_NONE = 0, STLLM_ACT_GELU = 1, STLLM_ACT_RELU = 2 } stllm_act;

const char* stllm_last_error(void);
int stllm_abi_version(void);
/* HOST utility (no GPU involved): n values of the deterministic synthetic-weight stream (stllm_amd/synth.py: random-init benchmarks,
 * parity fixtures) for element indices start .. start + n - 1 of the tensor whose (name, seed) hash to `key`; bit-identical to the
 * torch recipe that fills device tensors. */
int stllm_synth_normal_f32(float* out, int64_t n, int64_t start, uint32_t key, float scale, float mean);
/* name of the kernel template instantiation chosen by the last stllm_gemm call on this thread (for profiling) */
const char* stllm_last_kernel(void);

/*
 * C[M,N] = epilogue(A[M,K] @ W[N,K]^T)  on MFMA, LDS-tiled (128-byte K panels, XOR-swizzled).
 * Replaces every nn.Linear / F.linear on the path:
 *   eva_vit.py:124 (qkv), :146 (proj), :55-59 (fc1/fc2); Qformer.py:186-198 (query/key/value),
 *   :286 (attention output dense), :359 (intermediate), :372 (output); st_llm.py:368 (llama_proj),
 *   :475 (down/up_proj), :38-42 (mvm_decoder.head), :122 (lm_head); HF LlamaAttention q/k/v/o_proj
 *   and LlamaMLP gate/up/down_proj (spec: modeling_llama_mem.py:163-166, 138-144).
 * Constraints: K*sizeof(T) % 128 == 0; N % 128 == 0; lda/ldw*sizeof(T) % 16 == 0; A, W 16-byte aligned.
 * A: T[M,lda]   W: T[N,ldw]   bias: f32[N] or NULL
 * STORE : out = T or f32 (out_is_f32) [M,ldo]; act = stllm_act
 * RESID : resid/out f32 [M,ldo]
 * SWIGLU: out T[M,ldo], N/2 columns
 * ROPE  : out T[M,ldo]; rope_cos/rope_sin f32[rope_seq, 64]; row m is position m % rope_seq;
 *         head_dim fixed at 128; columns >= rope_cols are stored unrotated
 * PATCH : A is ignored — the A operand is gathered from `frames` f32 [n_frames,3,224,224]
 *         (implicit GEMM, eva_vit.py:196-204); M = n_frames*256; K = 588 padded (W zero-padded to
 *         ldw); out_f32 = x[n_frames*257, ldo]; `aux` = pos_embed f32[257,N].
 * dtype STLLM_BF16X3: A is f32 [M,lda]; W is bf16 [N, ldw >= 3 K] = stllm_split3_rows(weight, side 1) — (hi | lo | hi) along K;
 *         every output is f32 (STORE / SWIGLU / ROPE write f32 [M,ldo]; act = exact erf GELU / ReLU); K % 64 == 0; PATCH and a_norm_*
 *         are rejected.  Needs split_ws (below).  Internally: split A -> bf16 [M, 3 K] (hi | hi | lo), ONE bf16 GEMM with K' = 3 K and
 *         fp32 output on the kernels above, then the activation / SwiGLU / RoPE as an fp32 row pass.
 */
typedef struct {
  int dtype;          /* stllm_dtype */
  int epilogue;       /* stllm_epilogue */
  int act;            /* stllm_act (STORE only) */
  int out_is_f32;     /* STORE only */
  const void* A; int64_t lda;
  const void* W; int64_t ldw;
  const float* bias;
  void* out; int64_t ldo;
  const float* resid; int64_t ldr;
  const float* aux0;  /* ROPE: cos table; PATCH: pos_embed */
  const float* aux1;  /* ROPE: sin table */
  const float* frames;/* PATCH */
  int rope_seq; int rope_cols;
  int M, N, K;
  /* optional 2-level row indexing (0 = flat): logical row m lives at
   *   (m / rows_per_batch) * batch_stride + (m % rows_per_batch) * ld      [elements]
   * used for the Q-Former's query / text row groups inside a [N, 32+Lt, C] buffer
   * (Qformer.py:430-462 slices hidden states by query_length). */
  int a_rows_per_batch; int64_t a_batch_stride;
  int o_rows_per_batch; int64_t o_batch_stride;
  /* optional scratch for the phased / stream-K kernels (large problems): >= stllm_gemm_workspace_bytes() bytes of device
   * memory, 16-byte aligned, private to the launch stream (launches on one stream may share it).  NULL => the
   * small-tile kernel is used for every shape.  Contents need no initialisation beyond one memset at allocation. */
  void* workspace; int64_t workspace_bytes;
  /* optional, ABI version >= 2 (decode regime: M <= 8, 16-bit dtypes, every epilogue but PATCH): the A operand is not read
   * from memory but computed on the fly as  RMSNorm(x) * gamma  cast to `dtype` — Llama's input_layernorm /
   * post_attention_layernorm (modeling_llama_mem.py:61-78) fused into the projection that follows it in the one-token step.
   * a_norm_x: fp32 rows [M, K], row stride a_norm_ldx elements; a_norm_gamma: fp32 [K].  A / lda are ignored (may be NULL / 0).
   * Outside the decode regime the call is rejected (STLLM_ERR_UNSUPPORTED): run stllm_rmsnorm first. */
  const float* a_norm_x; int64_t a_norm_ldx; const float* a_norm_gamma; float a_norm_eps;
  /* ABI version >= 5, dtype STLLM_BF16X3 only: >= stllm_gemm_split_ws_bytes(M, N, K, epilogue) bytes of device memory, 16-byte aligned, private
   * to the launch stream, no initialisation (the split A operand and, for SWIGLU, the fp32 gate/up columns).  (ABI 4's fold_* fields —
   * LayerNorm folded into the GEMMs, measured slower than the LayerNorm kernels — were removed with their kernels in ABI 5.) */
  void* split_ws; int64_t split_ws_bytes;
  /* dtype STLLM_BF16X3 only — chaining two split GEMMs without an fp32 round trip between them:
   *   STLLM_SPLIT_A_PRESPLIT: A is ALREADY the split image, bf16 [M, lda >= 3 K] = (hi | hi | lo) — written by stllm_layernorm / stllm_rmsnorm called with
   *     dtype STLLM_BF16X3, by stllm_split3_rows, or by a previous GEMM with STLLM_SPLIT_OUT;
   *   STLLM_SPLIT_OUT (STORE and SWIGLU epilogues): `out` receives the split image of the result, bf16 [M, ldo >= 3 N'] (N' = N, or N / 2 for SWIGLU), after the
   *     activation — the A operand of the next GEMM — instead of the fp32 tensor.  The values are bit-identical to splitting the fp32 result afterwards. */
  int split_flags;
  /* optional, ABI version >= 7 (16-bit dtypes; STORE without activation, SWIGLU, ROPE): a FRAGMENT-MAJOR copy of W — [N / 32][K / 16][64 lanes][8 elements],
   * lane l of fragment (nb, ks) = W[32 nb + (l & 31)][16 ks + 8 (l >> 5) .. + 7], i.e. one contiguous KiB per v_mfma_f32_32x32x16 operand
   * (st-llm_amd/pack.py: frag32).  When present (N % 256 == 0, K % 256 == 0) prefill-sized problems may run on the W-direct kernel (csrc/gemm_wd.inc):
   * W fragments go straight into registers, only A is staged through the LDS.  W itself must still be valid (other shapes / kernels read it).
   * Replaces nothing in the reference: a layout of the nn.Linear weight of modeling_llama_mem.py:130-144, 172-248. */
  const void* w_frag;
} stllm_gemm_args;
enum { STLLM_SPLIT_A_PRESPLIT = 1, STLLM_SPLIT_OUT = 2 };
int64_t stllm_gemm_split_ws_bytes(int M, int N, int K, int epilogue, int split_flags);
/* x f32 [M, K] (row stride ldx; rows_per_batch / batch_stride: the 2-level row indexing of stllm_gemm_args, 0 = flat) ->
 * out bf16 [M, ldo >= 3 K]: hi = bf16(x), lo = bf16(x - hi), laid out (hi | hi | lo) along K for weight_side == 0 (the A operand) and
 * (hi | lo | hi) for weight_side != 0 (the weight: packed once).  K % 4 == 0. */
int stllm_split3_rows(const float* x, int64_t ldx, int rows_per_batch, int64_t batch_stride, void* out, int64_t ldo, int M, int K,
                      int weight_side, void* stream);
int64_t stllm_gemm_workspace_bytes(void);
/* Synchronises `stream` and returns 0 when no GEMM launch that used `workspace` ever gave up waiting for a peer workgroup
 * (the split-K exchanges poll with a bound instead of hanging the GPU when not all workgroups are resident, e.g. on a
 * GPU shared with another process), non-zero otherwise (sticky; stllm_last_error() explains). */
int stllm_gemm_workspace_status(const void* workspace, void* stream);
/* Host-only introspection of the phased kernel's schedule for an M x N x K problem (no GPU needed; used by the CPU tests):
 * plan5 = { data-parallel rounds q, remainder tiles r, K slices per remainder tile s, slice groups per XCD cap, estimated us }
 * for tile_rows = 192 | 256; heavy = 0 plain 16-bit output, 1 fp32 output / residual, 2 GELU.  T = q * 256 + r tiles of
 * tile_rows x 256; the s workgroups of a remainder tile sit on one XCD (s <= 32, 8 * cap >= r, cap = 32 / s). */
int stllm_gemm_plan(int M, int N, int K, int heavy, int tile_rows, int* plan5);
/* The same for the one-wave-per-SIMD kernel (st-llm_amd/csrc/gemm_w4.inc): shape = 32 (192 x 128 tile) | 42 (256 x 128) | 34 (192 x 256) |
 * 44 (256 x 256) | 22 (128 x 128, two workgroups per CU: T = q * 512 + r).  heavy bit 3 (| 8): the epilogue is STORE / RESID, so a last tile row of <= 32 rows is computed outside the tile
 * grid ("thin tail": ViT fc1's 4112 rows = 16 tile rows + 16 rows) and does not count as tiles. */
int stllm_gemm_w4_plan(int M, int N, int K, int heavy, int shape, int* plan5);
/* tuning / test hooks.  Option state is PER THREAD (thread-local, like the error string): the first stllm_* call of a host thread reads the
 * STLLM_GEMM_SK / _DEBUG / _GEMV / _P8 / _W4, STLLM_GEMV_MFMA, STLLM_ATTN_DMA, STLLM_ATTN_BWD_VALU, STLLM_NORM_FAST environment variables once,
 * stllm_set_option() changes the calling thread's copy only — two host threads driving different streams / devices never race on it.
 * Per-device launch state (dynamic-LDS opt-in, occupancy answers) is kept per device ordinal.
 * The fused-RMSNorm operand (a_norm_*) exists in the GEMV kernels only: "gemm_gemv" = 0 and forced tile kernels do not apply to it.
 *   "gemm_p8"    = -1 auto (cost model) | 0 off | 1 always (cost model picks the tile height) | 3 / 4 always, 192 / 256-row tile:
 *                  the phased 192|256 x 256 kernel (st-llm_amd/csrc/gemm_p8.inc; 16-bit dtypes, needs `workspace`)
 *   "gemm_w4"    = -1 auto = 2 | 0 off | 1 always (cost model picks the tile) | 2 where its plan beats the other kernels' estimates (from 1024 rows on also
 *                  plans whose K-split hides behind >= 1 whole round; below 1024 rows exchange-free plans of ONE partial round only) | 32 / 42 / 34 / 24 / 43 / 33 always,
 *                  192 x 128 / 256 x 128 / 192 x 256 / 128 x 256 / 256 x 192 / 192 x 192 tile (44 = 256 x 256 was retired in round 3: unsupported, falls back) | 22 (round 5, never chosen
 *                  automatically): the 128 x 128 tile as TWO workgroups per CU (two waves per SIMD, 80 KiB of LDS each, 512 persistent workgroups, exchange-free plans):
 *                  the one-wave-per-SIMD kernel (st-llm_amd/csrc/gemm_w4.inc; 16-bit dtypes, needs `workspace`)
 *   "gemm_gemv"  = -1 on (M <= 16) | 0 off | 1 only M <= 4 | 2 = -1: the skinny kernels of the decode regime (st-llm_amd/csrc/gemv.hip):
 *                  the 5 beams of demo.py's beam search, small serving batches (5-row step 6.99 -> 4.34 ms on MI355X)
 *   "gemv_mfma"  = -1 matrix-core GEMV (v_mfma_f32_16x16x32) from M = 3, the VALU kernel (v_dot2c) below | 0 never (M > 8 then runs on the
 *                  tile kernels) | 1 from M = 1
 *   "attn_dma"   = 1 (default) attention at head_dim 128 (Llama prefill) and 88 (ViT) staged by LDS-DMA, V through the transposing LDS read |
 *                  0 register-staged kernels | 10 nw + ks (nw x ks <= 12, ks in 1 / 2 / 4): force the head_dim-128 kernel's work split — nw query tiles per
 *                  workgroup x ks key-split waves per tile (tools/attn_bench.py --audit)
 *   "attn_decode_single" = 1 (default) one-workgroup-per-head decode attention for Skv <= 1536 | 0 always the split-KV pair
 *   "gemm_sk"    = -1 auto | 0 off | 1 (128x128) | 2 (128x256) | 3 (256x256): stream-K tile of the older kernels
 *   "gemm_debug" = ablation bits of the 128x128 kernels; bit 16 = in-kernel timeline of the phased kernel (tools/gemm_harness.cpp)
 *   "attn_bwd_valu" = 0 (default) MFMA attention backward for 16-bit operands | 1 the VALU twins
 *   "norm_fast"  = 1 (default) one row per wave | 2: two rows per wave for >= 2048 short rows (bit-identical results, same speed: an A/B switch)
 *   "gemm_w4_odd" = 1 (default) the 192-column tiles (43 = 256 x 192, 33 = 192 x 192; 16-bit STORE epilogues) take part in the automatic
 *                  choice | 0 the round-2 choice (A/B inside one process: bench.py --ab)
 *   "gemm_w4_wide" = 1 (default) a 16-bit GEMM of at most 640 rows whose 128 x 256 tiles make ONE round of 192..256 tiles (the Llama prefill qkv
 *                  GEMM at 385..640 rows: 5 x 48 = 240 tiles at S = 576) runs on the one-wave kernel's 128 x 256 tile (code 24) | 0 on the 128 x 128 kernel */
int stllm_set_option(const char* key, int value);
int stllm_gemm(const stllm_gemm_args* args, void* stream);

/* HIP-event timing of stllm_gemm launches on their launch stream, per calling thread (bench.py's roofline leg; also sees the launches of
 * the whole-stack entry points below).  mode 0 off | 1 every launch | 2 only launches whose kernel symbol — learned per (dtype, epilogue,
 * act, M, N, K) while mode 1 was on — equals target_symbol | 3 every 7th of those (a sample: the event records sit between the kernels on
 * the stream and cost the measured step ~3 us per timed launch; bench.py's timed region uses 3).  Every call starts a new, empty session.  _read synchronises record i's end
 * event and returns its kernel symbol (stllm_last_kernel naming), duration, algorithmic FLOPs (2 M N K) and shape. */
int stllm_gemm_profile(int mode, const char* target_symbol);
int stllm_gemm_profile_count(void);
int stllm_gemm_profile_read(int i, char* symbol, int symbol_cap, float* ms, double* flops, int* mnk3 /* M, N, K or NULL */);

/* =========================================================================================================================
 * Whole-stack entry points: one host call issues every launch of a layer stack, through the entry points of this header
 * with the arguments the per-op host code passes (bit-identical results; ~540 host round trips per step become a handful).
 * All weight pointers are device pointers to tensors packed as for stllm_gemm (st-llm_amd/pack.py); ld_* = row stride in elements.
 * ========================================================================================================================= */

/* one EVA ViT block (eva_vit.py:157-180: norm1, attn.qkv + (q_bias, 0, v_bias), attn.proj, norm2, mlp.fc1, mlp.fc2) */
typedef struct {
  const float* n1w; const float* n1b; float e1;
  const void* wqkv; int64_t ld_qkv; const float* bqkv;     /* [3 dim, dim], bias f32 [3 dim] = (q_bias, 0, v_bias) (eva_vit.py:120-124) */
  const void* wproj; int64_t ld_proj; const float* bproj;
  const float* n2w; const float* n2b; float e2;
  const void* wfc1; int64_t ld_fc1; const float* bfc1;     /* [hidden, dim] */
  const void* wfc2; int64_t ld_fc2; const float* bfc2;     /* [dim, hidden] */
} stllm_vit_block_weights;
typedef struct {
  int dtype; int n_seq; int seq_len; int num_heads; int dim; int hidden;
  float* x; int64_t ldx;                       /* fp32 residual stream [n_seq * seq_len, dim], updated in place */
  void* scratch; int64_t scratch_bytes;        /* >= stllm_vit_blocks_scratch_bytes(...), 256-byte aligned, no initialisation */
  void* workspace; int64_t workspace_bytes;    /* the stllm_gemm workspace of the launch stream */
} stllm_vit_blocks_args;
int64_t stllm_vit_blocks_scratch_bytes(int dtype, int n_seq, int seq_len, int dim, int hidden);
/* VisionTransformer.forward_features' block loop (eva_vit.py:336-339; Block.forward :173-180) for n_blocks consecutive blocks */
int stllm_vit_blocks(const stllm_vit_blocks_args* args, const stllm_vit_block_weights* blocks, int n_blocks, void* stream);

/* one HF LlamaDecoderLayer (spec modeling_llama_mem.py:61-316): RMSNorm weights, fused [q | k | v] rows in the packed RoPE layout,
 * o_proj, packed [32 gate | 32 up] rows, down_proj; kv_cache: NULL or this layer's [B, cache_max_len, 3 hidden] cache buffer */
typedef struct {
  const float* ln1; const void* wqkv; int64_t ld_qkv; const void* wo; int64_t ld_o;
  const float* ln2; const void* wgu; int64_t ld_gu; const void* wdown; int64_t ld_down;
  void* kv_cache;
  const void* wqkv_frag; const void* wgu_frag;   /* ABI >= 7: NULL or the fragment-major copies of wqkv / wgu (stllm_gemm_args.w_frag) */
} stllm_llama_layer_weights;
typedef struct {
  int dtype; int B; int S; int n_heads; int hidden; int inter; float eps;
  float* x; int64_t ldx;                       /* fp32 residual stream [B * S, hidden], updated in place */
  const float* rope_cos; const float* rope_sin;/* f32 [S, 64] */
  const int32_t* kv_len;                       /* int32 [B] valid lengths of right-padded sequences, or NULL */
  int64_t cache_max_len;                       /* 0: no KV cache */
  void* scratch; int64_t scratch_bytes;        /* >= stllm_llama_layers_scratch_bytes(...) */
  void* workspace; int64_t workspace_bytes;
} stllm_llama_layers_args;
int64_t stllm_llama_layers_scratch_bytes(int dtype, int B, int S, int hidden, int inter);
/* the decoder-layer loop of the PREFILL (st_llm.py:56-92 -> HF LlamaModel.forward, use_cache False or filling a fresh cache) */
int stllm_llama_layers(const stllm_llama_layers_args* args, const stllm_llama_layer_weights* layers, int n_layers, void* stream);
/* ONE decoder layer of a SEQUENCE-PARALLEL prefill in two parts (ABI >= 7, round 6; st-llm_amd/models/llama.py: LlamaModel.prefill_sp — no reference counterpart, the
 * reference has no sequence parallelism: SURVEY.md 2b): this rank owns the positions [s0, s1) of one sequence (args->B == 1, args->S == s1 - s0 rows in x,
 * args->rope_cos / rope_sin pointing at position s0's table row), `qkv` is the fused buffer [s1, 3 * hidden] (compute dtype) whose rows [0, s0) hold the earlier
 * members' K | V (their q columns zero).
 *   part 0: RMSNorm(x) -> fused QKV GEMM + RoPE at positions s0 .. -> rows [s0, s1) of `qkv`            (the caller then ships / receives K | V rows)
 *   part 1: causal attention of the s1 query rows over the s1 keys, rows [s0, s1) -> o_proj + residual -> RMSNorm -> gate/up + SiLU(gate) * up -> down + residual
 * The same launches, in the same order, as the per-op path: bit-identical.  Scratch: stllm_llama_layer_sp_scratch_bytes(dtype, s0, s1, hidden, inter). */
int64_t stllm_llama_layer_sp_scratch_bytes(int dtype, int s0, int s1, int hidden, int inter);
int stllm_llama_layer_sp(const stllm_llama_layers_args* args, const stllm_llama_layer_weights* layer, void* qkv, int s0, int s1, int part, void* stream);

/* BertSelfOutput / BertOutput (Qformer.py:281-289, 384-400): LayerNorm(dense(x) + input) */
typedef struct {
  const void* w; int64_t ldw; const float* b;  /* dense [hidden, in], bias f32 [hidden] */
  const float* g; const float* beta; float eps;/* LayerNorm weight / bias f32 [hidden] */
} stllm_bert_output_weights;
/* one Q-Former BertLayer (Qformer.py:367-484): fused self-attention [q | k | v] rows, attention.output, on the layers with cross-attention
 * (has_cross: layer_num % cross_attention_freq == 0) crossattention.self.query + crossattention.output — the cross K / V projections of ALL
 * layers live in ONE weight (stllm_qformer_layers_args.ckv_w; ckv_index = this layer's [k | v] column block in its output) —, the query
 * rows' FFN (intermediate_query / output_query) and, with text rows, the text FFN (intermediate / output; ft_w1 NULL otherwise) */
typedef struct {
  const void* wqkv; int64_t ld_qkv; const float* bqkv;
  stllm_bert_output_weights attn_out;
  int has_cross; int ckv_index;
  const void* cq_w; int64_t ld_cq; const float* cq_b;
  stllm_bert_output_weights cross_out;
  const void* fq_w1; int64_t ld_fq1; const float* fq_b1; stllm_bert_output_weights fq_out;
  const void* ft_w1; int64_t ld_ft1; const float* ft_b1; stllm_bert_output_weights ft_out;
} stllm_qformer_layer_weights;
typedef struct {
  int dtype; int n_seq; int n_query; int n_text; int n_heads; int hidden; int inter; int enc_len; int enc_dim; int n_cross;
  float* hq32; void* hq16;                     /* query rows [n_seq * n_query, hidden]: fp32 post-LN stream + its compute-dtype copy, both updated in place */
  float* ht32; void* ht16;                     /* text rows [n_seq * n_text, hidden] likewise (NULL when n_text == 0) */
  const void* enc16; int64_t ld_enc;           /* ln_vision'd image tokens, compute dtype [n_seq * enc_len, enc_dim] */
  const void* ckv_w; int64_t ld_ckv; const float* ckv_b;   /* [n_cross * 2 hidden, enc_dim]: (key | value) rows of the cross layers in layer order */
  const int32_t* kv_len;                       /* int32 [n_seq] = n_query + valid text tokens (right-padded text), or NULL */
  void* scratch; int64_t scratch_bytes;        /* >= stllm_qformer_layers_scratch_bytes(...), 256-byte aligned, no initialisation */
  void* workspace; int64_t workspace_bytes;    /* the stllm_gemm workspace of the launch stream */
} stllm_qformer_layers_args;
int64_t stllm_qformer_layers_scratch_bytes(int dtype, int n_seq, int n_query, int n_text, int hidden, int inter, int enc_len, int enc_dim, int n_cross);
/* BertEncoder.forward's layer loop (Qformer.py:495-589; BertLayer.forward :402-484) on the embedded rows: self-attention over [queries | text],
 * cross-attention of the query rows against the image tokens on the has_cross layers, the two FFNs.  STLLM_BF16X3: fp32 activations, every
 * Linear split inside its GEMM (no pre-split images: the rows are too few for the split passes to matter). */
int stllm_qformer_layers(const stllm_qformer_layers_args* args, const stllm_qformer_layer_weights* layers, int n_layers, void* stream);

/*
 * Decode attention: ONE query row per (batch, head) against Skv cached keys (the one-token step of generate(), SURVEY §8f
 * rank 1; HF LlamaAttention with past_key_values, spec modeling_llama_mem.py:172-248).  HBM-bound: the keys are split over
 * several workgroups per head and the partial softmax states are merged by a second launch.  head_dim 128, bf16 / fp16.
 * q: element (b, h, d) at q[b*q_bs + h*128 + d];  k, v: (b, s, h, d) at k[b*k_bs + s*k_rs + h*128 + d];  out like q.
 * workspace: >= stllm_attention_decode_workspace_bytes(B, H, Skv) bytes, 16-byte aligned, no initialisation needed.
 */
int64_t stllm_attention_decode_workspace_bytes(int B, int H, int Skv);
int stllm_attention_decode(int dtype, const void* q, int64_t q_bs, const void* k, int64_t k_bs, int64_t k_rs,
                           const void* v, int64_t v_bs, int64_t v_rs, void* out, int64_t o_bs, int B, int H, int Skv,
                           int D, float scale, void* workspace, int64_t workspace_bytes, void* stream);

/*
 * Frame preprocessing in front of the path — replaces the CPU transform chain of Chat.__init__ / upload_video
 * (stllm/conversation/conversation.py:190-198, 276-279; stllm/test/video_transforms.py:54-60, 94-124, 367-407):
 *   GroupScale(224, BICUBIC) -> GroupCenterCrop(224) -> Stack -> ToTorchFormatTensor -> GroupNormalize(CLIP mean / std)
 * frames: uint8 RGB [n, H, W, 3] (frame n at frames + n * frame_stride_bytes);  out: f32 [n, 3, 224, 224].
 * Bit-identical to torchvision 0.15.1 + Pillow on the CPU: short side -> 224 with Pillow's antialiased bicubic (22-bit
 * fixed-point coefficients, horizontal then vertical pass, uint8 in between), centre crop with round-half-even offsets,
 * x / 255, (x - mean) / std in fp32.  Down-scaling up to ~31x.  `workspace`: >= stllm_preprocess_workspace_bytes(n, H, W)
 * bytes of device memory, 16-byte aligned, no initialisation needed (coefficient tables + the uint8 image between the passes).
 */
int64_t stllm_preprocess_workspace_bytes(int n_frames, int H, int W);   /* -1: unsupported geometry */
int stllm_preprocess_frames(const uint8_t* frames, int64_t frame_stride_bytes, int n_frames, int H, int W, float* out,
                            void* workspace, int64_t workspace_bytes, void* stream);

/*
 * LayerNorm over the last dim of x f32[M,D] (ldx), fp32 statistics (two-pass), affine.
 * Writes out_t T[M,D] (ldo_t) and/or out_f32 [M,D] (ldo_f); either may be NULL.
 * dtype STLLM_BF16X3 (ABI >= 5): out_t is the split image bf16 [M, ldo_t >= 3 D] = (hi | hi | lo) of the fp32 result — the A operand of a bf16x3 GEMM
 * called with STLLM_SPLIT_A_PRESPLIT (same for stllm_rmsnorm).
 * Replaces nn.LayerNorm at eva_vit.py:157,163 (eps 1e-6), blip2.py:103-109 (ln_vision, eps 1e-5),
 * Qformer.py:65,106,282,288,368,374 (eps 1e-12), st_llm.py:39 (mvm_decoder.norm).
 * D % 4 == 0, D <= 8192.
 */
int stllm_layernorm(int dtype, const float* x, int64_t ldx, const float* gamma, const float* beta,
                    float eps, void* out_t, int64_t ldo_t, float* out_f32, int64_t ldo_f,
                    int M, int D, void* stream);

/* RMSNorm (HF LlamaRMSNorm, spec modeling_llama_mem.py:61-78): out = w * x * rsqrt(mean(x^2)+eps). */
int stllm_rmsnorm(int dtype, const float* x, int64_t ldx, const float* gamma, float eps,
                  void* out_t, int64_t ldo_t, float* out_f32, int64_t ldo_f, int M, int D, void* stream);

/*
 * Fused softmax(scale * Q K^T + mask) V, FlashAttention-style (online softmax, LDS-tiled K/V,
 * MFMA 32x32x16; fp32 dtype uses an exact-fp32 vector kernel).  Never materialises S.
 * Replaces eva_vit.py:128-145 (ViT, 16 heads x 88), Qformer.py:205-268 (self: 12x64, cross),
 * HF LlamaAttention softmax(QK^T/sqrt(128) + causal/pad mask) V (spec modeling_llama_mem.py:172-248).
 * q: T, element (b, s, h, d) at q[b*q_bs + s*q_rs + h*D + d]; same for k, v (Skv rows) and out.
 * kv_len: int32[B] or NULL — keys >= kv_len[b] are masked (right-padding / Q-Former text mask,
 *         Qformer.py:785-801 with -10000 == exact 0 after fp32 softmax).
 * causal: key j visible to query i iff j <= i (Sq == Skv).
 * D in {64, 88, 128}; all strides multiples of 8 elements; pointers 16-byte aligned.
 */
int stllm_attention(int dtype, const void* q, int64_t q_bs, int64_t q_rs,
                    const void* k, int64_t k_bs, int64_t k_rs,
                    const void* v, int64_t v_bs, int64_t v_rs,
                    void* out, int64_t o_bs, int64_t o_rs,
                    int B, int H, int Sq, int Skv, int D, float scale, int causal,
                    const int32_t* kv_len, void* stream);

/*
 * Row gather (token-block assembly, dynamic masking, residual-index selection, embedding lookup):
 *   dst[i, :] = scale * ((idx_a[i] >= 0 ? src_a[idx_a[i], :] : src_b[-idx_a[i]-1, :]) + (add ? add[idx_add[i], :] : 0))
 * all f32, D % 4 == 0.  The same kernel re-orders tokens between the BT-Adapter's spatial '(b t) p' and temporal
 * 'b (p t)' layouts and averages branches (eva_btadapter.py:179-196, 261-310).  Replaces torch.cat / index / embed_tokens glue at st_llm.py:391-404,
 * 416-431, 473-476, 491, 509, 524-530 and conversation.py:288-293, 336-337.
 */
int stllm_gather_rows(const float* src_a, int64_t ld_a, const float* src_b, int64_t ld_b,
                      const int32_t* idx_a, const float* add, int64_t ld_add, const int32_t* idx_add,
                      float* dst, int64_t ld_dst, int n_rows, int D, float scale, void* stream);

/* out[b, j] = mean_t x[b, t, j]  (x f32 [B,T,J] contiguous) — st_llm.py:468,471; conversation.py:282,287 */
int stllm_mean_t(const float* x, float* out, int B, int T, int64_t J, void* stream);

/* x[n*257, :] = cls_token + pos_embed[0]  for every frame — eva_vit.py:328-331 (CLS row) */
int stllm_vit_cls_rows(const float* cls, const float* pos, float* x, int64_t ldx, int n_frames, int D,
                       void* stream);

/* MVM loss pieces (st_llm.py:89-91): out[i] = 2 - 2 * <a_i/|a_i|, b_i/|b_i|>, rows gathered by index */
int stllm_cosine_rows(const float* a, int64_t lda, const int32_t* idx_a, const float* b, int64_t ldb,
                      const int32_t* idx_b, float* out, int n_rows, int D, void* stream);

/* shifted-label cross entropy rows (st_llm.py:127-135): loss[i] = logsumexp(logits[i]) - logits[i,labels[i]];
 * labels[i] < 0 (ignore_index) -> 0.  The caller averages over the valid rows. */
int stllm_cross_entropy_rows(const float* logits, int64_t ldl, const int32_t* labels, float* loss,
                             int n_rows, int V, void* stream);

/* out T[M,D] = cast(x f32[M,D]) — the .to(dtype)/type_as casts at st_llm.py:453,474 and the autocast
 * boundary of blip2.py:36-44, when a GEMM consumes rows of the fp32 stream without a norm in between. */
int stllm_cast_rows(int dtype, const float* x, int64_t ldx, void* out, int64_t ldo, int M, int D, void* stream);

/* =========================================================================================================================
 * Backward / optimizer entry points.  The reference has no such code of its own: the gradients come
 * from torch.autograd over st_llm.py:116-146 (shifted CE + loss_mvm) driven by HF Trainer + DeepSpeed (train_hf.py,
 * stllm_trainer.py:317, config/(model)_stllm_qa.yaml: freeze_LLM False, bf16, AdamW lr 2e-5).  Each entry point below is the
 * transpose (vector-Jacobian product) of one forward entry point above, with the same pointer / stride / stream conventions;
 * GEMM-shaped gradients reuse stllm_gemm on operands re-laid-out by stllm_transpose:
 *      y = x W^T :   dx = gemm(A = dy,   W = W^T [K, N])          dW = gemm(A = dy^T [N, Mp], W = x^T [K, Mp])  (fp32 out)
 * ========================================================================================================================= */

/* dst[c, r] = src[r, c] for r < rows, 0 for rows <= r < rows_padded (zero K-padding for the GEMM).  dtype sizes 2 or 4. */
int stllm_transpose(int dtype, const void* src, int64_t ld_src, void* dst, int64_t ld_dst, int rows, int cols, int rows_padded,
                    void* stream);

/* scratch of the two norm-backward entry points and of stllm_colsum: per-row statistics + column partial sums */
int64_t stllm_norm_bwd_workspace_bytes(int rows, int cols);

/* Transpose of stllm_rmsnorm (modeling_llama_mem.py:61-78): with r = rsqrt(mean(x^2) + eps), xh = x r, g = gamma * dy:
 *   dx (+)= r * (g - xh * mean(g * xh))      dgamma[c] = sum_rows dy * xh       (dy: T_dy [rows, cols], T_dy any dtype code)
 * accumulate != 0 adds into dx (the fp32 residual-stream gradient), else overwrites.  Deterministic (no atomics). */
int stllm_rmsnorm_bwd(int dy_dtype, const float* x, int64_t ldx, const float* gamma, float eps, const void* dy, int64_t lddy,
                      float* dx, int64_t lddx, int accumulate, float* dgamma, void* workspace, int64_t workspace_bytes, int rows,
                      int cols, void* stream);

/* Transpose of stllm_layernorm (st_llm.py:39 mvm_decoder.norm): xh = (x - mu) r;
 *   dx (+)= r * (g - mean(g) - xh * mean(g * xh))      dgamma = sum dy * xh      dbeta = sum dy */
int stllm_layernorm_bwd(int dy_dtype, const float* x, int64_t ldx, const float* gamma, float eps, const void* dy, int64_t lddy,
                        float* dx, int64_t lddx, int accumulate, float* dgamma, float* dbeta, void* workspace,
                        int64_t workspace_bytes, int rows, int cols, void* stream);

/* SiLU(gate) * up on RAW gate/up activations in the packed [32 gate | 32 up] column groups of the SWIGLU epilogue
 * (modeling_llama_mem.py:143-144).  Training keeps the raw activations (STORE epilogue) so the backward needs no
 * recomputation:  gu T[rows, 2*inter] -> out T[rows, inter]. */
int stllm_swiglu(int dtype, const void* gu, int64_t ldgu, void* out, int64_t ldo, int rows, int inter, void* stream);

/* d gate = dg * up * s * (1 + gate * (1 - s)),  d up = dg * gate * s,  s = sigmoid(gate); output in the packed layout of gu. */
int stllm_swiglu_bwd(int dtype, const void* gu, int64_t ldgu, const void* dg, int64_t lddg, void* dgu, int64_t lddgu, int rows,
                     int inter, void* stream);

/* Transpose of the ROPE epilogue's rotation, in place on the first rope_cols columns of d T[rows, cols] (packed head layout,
 * position = row % rope_seq; cos/sin as for stllm_gemm):  dx1 = dy1 c + dy2 s,  dx2 = dy2 c - dy1 s. */
int stllm_rope_bwd(int dtype, void* d, int64_t ld, const float* cos_t, const float* sin_t, int rows, int cols, int rope_seq,
                   int rope_cols, void* stream);

/* Gradients of stllm_attention: dq, dk, dv from q, k, v, the forward output o and its gradient dO; same element addressing as
 * stllm_attention for all eight operands; masks as in the forward (causal needs Sq == Skv).  D % 8 == 0, D <= 128.
 * bf16 / f16 run on MFMA kernels (the Llama prefill, the Q-Former's 64-wide self- and cross-attention heads, EVA's 88-wide heads padded
 * to 96 in LDS), fp32 on fp32-FMA kernels (also for 16-bit operands with STLLM_ATTN_BWD_VALU=1 in the environment).
 * workspace: stllm_attention_bwd_workspace_bytes(B, H, Sq) (log-sum-exp and dO.o per query row). */
int64_t stllm_attention_bwd_workspace_bytes(int B, int H, int Sq);
int stllm_attention_bwd(int dtype, const void* q, int64_t q_bs, int64_t q_rs, const void* k, int64_t k_bs, int64_t k_rs,
                        const void* v, int64_t v_bs, int64_t v_rs, const void* o, int64_t o_bs, int64_t o_rs,
                        const void* dO, int64_t do_bs, int64_t do_rs, void* dq, int64_t dq_bs, int64_t dq_rs,
                        void* dk, int64_t dk_bs, int64_t dk_rs, void* dv, int64_t dv_bs, int64_t dv_rs,
                        int B, int H, int Sq, int Skv, int D, float scale, int causal, const int32_t* kv_len,
                        void* workspace, int64_t workspace_bytes, void* stream);

/* Transpose of stllm_cross_entropy_rows summed with weight `scale` (st_llm.py:127-135; scale = 1 / #valid rows):
 * dlogits[i, c] = scale * (softmax(logits[i, :vocab])[c] - [c == labels[i]]) for labels[i] >= 0, else 0; columns
 * vocab .. cols_padded-1 (lm_head padding) are written as 0.  dlogits: T[rows, cols_padded]. */
int stllm_cross_entropy_bwd(int dtype, const float* logits, int64_t ldl, const int32_t* labels, float scale, void* dlogits,
                            int64_t lddl, int rows, int vocab, int cols_padded, void* stream);

/* Transpose of stllm_gather_rows: (idx[i] >= 0 ? dst_a[idx[i]] : dst_b[-idx[i]-1]) += scale * src[i]  (fp32 atomics; rows that
 * are hit more than once — repeated token ids — make the summation order, not the set of summands, run-dependent). */
int stllm_scatter_add_rows(const float* src, int64_t ld_src, const int32_t* idx, float* dst_a, int64_t ld_a, float* dst_b,
                           int64_t ld_b, int n_rows, int D, float scale, void* stream);

/* d/da of  scale * sum_i stllm_cosine_rows(a, b)[i]  (st_llm.py:89-91; b is the detached target):
 * da[i] = -2 scale / |a_i| * (b_i/|b_i| - cos_i * a_i/|a_i|),  rows gathered by index as in the forward. */
int stllm_cosine_rows_bwd(const float* a, int64_t lda, const int32_t* idx_a, const float* b, int64_t ldb, const int32_t* idx_b,
                          float scale, float* da, int64_t ldda, int n_rows, int D, void* stream);

/* out[c] = sum_rows x[r, c]  (bias gradients); workspace: stllm_norm_bwd_workspace_bytes(rows, cols).  Deterministic. */
int stllm_colsum(int dtype, const void* x, int64_t ldx, float* out, int rows, int cols, void* workspace, int64_t workspace_bytes,
                 void* stream);

/* dx = dy * (y > 0)  — the ReLU between down_proj and up_proj (st_llm.py:473-475); cols % 8 == 0. */
int stllm_relu_bwd(int dtype, const void* dy, int64_t lddy, const void* y, int64_t ldy, void* dx, int64_t lddx, int rows, int cols,
                   void* stream);

/* exact-erf GELU (nn.GELU(), eva_vit.py:45; Qformer.py:347 ACT2FN["gelu"]) on RAW pre-activations and its derivative: training keeps the
 * pre-activations (STORE epilogue) because the fused GELU epilogue of the forward does not.  cols % 8 == 0. */
int stllm_gelu(int dtype, const void* x, int64_t ldx, void* out, int64_t ldo, int rows, int cols, void* stream);
int stllm_gelu_bwd(int dtype, const void* x, int64_t ldx, const void* dy, int64_t lddy, void* dx, int64_t lddx, int rows, int cols,
                   void* stream);

/* x[r, :] *= scale[idx ? idx[r] : r / rows_per_group], in place (T [rows, cols], cols % 8 == 0): timm's drop_path as called by the
 * BT-Adapter blocks in train mode (eva_vit.py:30-38, eva_btadapter.py:274-280, 303) — per-sample keep mask / keep_prob — and, being
 * linear and diagonal, its own backward. */
int stllm_scale_rows(int dtype, void* x, int64_t ldx, const float* scale, const int32_t* idx, int rows_per_group, int rows, int cols,
                     void* stream);

/* Transpose of stllm_mean_t up to the 1/T factor: dst[b, t, j] += scale * src[b, j]  (f32, J % 4 == 0). */
int stllm_bcast_add_t(float* dst, const float* src, int B, int T, int64_t J, float scale, void* stream);

/* torch.optim.AdamW step on flat fp32 state (what HF Trainer instantiates for the reference's runs):
 *   g' = grad_scale * g;  p *= 1 - lr * weight_decay;  m = b1 m + (1-b1) g';  v = b2 v + (1-b2) g'^2;
 *   p -= lr / (1 - b1^step) * m / (sqrt(v) / sqrt(1 - b2^step) + eps);   p16 (optional, dtype code p16_dtype) = cast(p). */
int stllm_adamw(float* p, const float* g, float* m, float* v, void* p16, int p16_dtype, int64_t n, float lr, float beta1,
                float beta2, float eps, float weight_decay, int step, float grad_scale, void* stream);

/* out[0] += sum x^2  (gradient-norm clipping, torch.nn.utils.clip_grad_norm_); the caller zeroes out[0]. */
int stllm_sumsq(const float* x, int64_t n, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* STLLM_HIP_H */
