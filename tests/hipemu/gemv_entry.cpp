// TEST-ONLY: `stllm_gemm` of the emulated library = the argument marshalling of the real entry point (st-llm_amd/csrc/gemm.hip)
// for the one kernel family the emulator can run, the skinny GEMM of gemv.hip.  Anything else is "unsupported".
#include "gemm_common.h"

extern "C" int stllm_gemm(const stllm_gemm_args* a, void* stream) {
  if (!a || a->epilogue == STLLM_EPI_PATCH || a->dtype == STLLM_F32) return STLLM_ERR_UNSUPPORTED;
  sg::GemmParams p{};
  p.A = reinterpret_cast<const char*>(a->A); p.lda_b = a->lda * 2;
  p.W = reinterpret_cast<const char*>(a->W); p.ldw_b = a->ldw * 2;
  p.bias = a->bias; p.out = a->out; p.ldo = a->ldo; p.resid = a->resid; p.ldr = a->ldr;
  p.aux0 = a->aux0; p.aux1 = a->aux1;
  p.rope_seq = a->rope_seq; p.rope_cols = a->rope_cols;
  p.M = a->M; p.N = a->N; p.K = a->K; p.act = a->act; p.out_is_f32 = a->out_is_f32;
  p.a_rpb = a->a_rows_per_batch; p.a_bs_b = a->a_batch_stride * 2;
  p.o_rpb = a->o_rows_per_batch; p.o_bs = a->o_batch_stride;
  return stllm_gemv_launch(a->dtype, a->epilogue, p, reinterpret_cast<hipStream_t>(stream));
}
extern "C" int64_t stllm_gemm_workspace_bytes(void) { return 16; }
