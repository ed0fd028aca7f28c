// Per-thread dispatch options of libstllm_hip.so (no HIP dependency: shared by common.h and error.cpp).
#pragma once
struct StllmOptions {
  int gemm_sk;             // STLLM_GEMM_SK: -1 auto | 0 off | 1 / 2 / 3 force a stream-K tile of the older kernels
  int gemm_debug;          // STLLM_GEMM_DEBUG: ablation bits
  int gemm_gemv;           // STLLM_GEMM_GEMV: -1 / 2 GEMV kernels up to 16 rows | 1 up to 4 rows | 0 off
  int gemm_p8;             // STLLM_GEMM_P8: -1 auto | 0 off | 1 phased kernel (3 / 4: force 192 / 256 rows)
  int gemm_w4;             // STLLM_GEMM_W4: -1 auto | 0 off | 1 one-wave-per-SIMD kernel (32 / 34 / 42 / 24 / 43 / 33: force the tile) | 3 auto + the Llama prefill qkv GEMM on it (A/B switch)
  int gemv_mfma;           // STLLM_GEMV_MFMA: -1 matrix-core GEMV from 3 rows | 0 never | n from n rows
  int attn_decode_single;  // 1 one-workgroup-per-head decode attention for Skv <= 1536 | 0 always the split-KV pair
  int attn_dma;            // STLLM_ATTN_DMA: 1 LDS-DMA attention kernels | 0 register-staged | 2 ...
  int attn_bwd_valu;       // STLLM_ATTN_BWD_VALU: 1 = VALU attention backward also for 16-bit operands
  int gemm_w4_odd;         // STLLM_GEMM_W4_ODD: 1 (default) the 192-column w4 tiles (256 x 192, 192 x 192) take part in the automatic choice | 0 round-2 choice
  int gemm_w4_wide;        // STLLM_GEMM_W4_WIDE: 1 (default) prefill-sized GEMMs (<= 640 rows) whose 128 x 256 tiles fill ONE round (192..256 tiles: the Llama qkv GEMM at 385..640 rows) run on the one-wave kernel's 128 x 256 tile | 0 the 128 x 128 kernel (round 1-3)
  int attn_f32_mfma;       // STLLM_ATTN_F32_MFMA: 1 (default) fp32 attention on the exact-fp32 matrix-core kernel from 8 query rows on | 0 the vector kernel (round 1-3) everywhere
  int gemm_t1;             // STLLM_GEMM_T1: -1 auto | 0 off | 2 / 4 / 6 force the tall-tile one-round kernel (gemm_t1.inc) with that many column fragments per wave wherever it is eligible
  int gemm_wd;             // STLLM_GEMM_WD: -1 auto | 0 off | 4 / 6 force the W-direct kernel (gemm_wd.inc) with that many 32-row fragments per tile wherever it is eligible
  int attn_q_lds;          // STLLM_ATTN_Q_LDS: 1 (default) the LDS-DMA attention kernels stage their query tiles through the LDS (row-contiguous requests, one copy per tile) | 0 per-lane fragment loads from global memory (rounds 2-5)
  int norm_fast;           // STLLM_NORM_FAST: 1 (default) one row per wave | 2 two rows per wave for >= 2048 short rows (bit-identical; measured equal: 22.36 / 22.47 / 22.41 / 22.40 ms per step)
};
StllmOptions& stllm_options();
