// HIP-event timing of stllm_gemm launches on the launch stream (bench.py's roofline leg: `achieved` = algorithmic FLOPs / average
// launch duration of the dominant kernel symbol, measured live over the timed region).  Lives behind stllm_gemm itself so that launches
// issued by the whole-stack entry points (stacks.cpp) are seen exactly like launches issued one by one from the host language.
// Per-thread state (like the options): the thread that enables profiling is the thread whose launches are timed.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <map>
#include <tuple>
#include <vector>

#include "../../include/stllm_hip.h"

void stllm_set_error(const char* fmt, ...);

namespace {
struct Rec { hipEvent_t s, e; const char* sym; double flops; int m, n, k; };
typedef std::tuple<int, int, int, int, int, int, int> Key;   // dtype, epilogue, act, out_is_f32, M, N, K
constexpr unsigned kSampleEvery = 7;
constexpr size_t kMaxRecs = 1 << 16;   // records (event pairs) per session
struct Prof {
  int mode = 0;                       // 0 off | 1 every launch | 2 launches whose symbol (learned in mode 1) equals `target` | 3: every 7th of those
  unsigned seen = 0;                  // mode 3: launches of the target so far
  char target[160] = "";
  std::vector<Rec> recs;
  std::vector<hipEvent_t> pool;       // events of earlier sessions, reused
  std::map<Key, const char*> sym_of;  // shape -> kernel symbol (static-lifetime strings of stllm_last_kernel)
};
Prof& prof() {
  static thread_local Prof p;
  return p;
}
hipEvent_t take_event(Prof& p) {
  if (!p.pool.empty()) {
    hipEvent_t e = p.pool.back();
    p.pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}
}  // namespace

// called by stllm_gemm around its dispatch: returns a record index or -1
int stllm_prof_begin(const stllm_gemm_args* a, void* stream) {
  Prof& p = prof();
  if (p.mode == 0) return -1;
  if (p.mode >= 2) {
    auto it = p.sym_of.find(Key(a->dtype, a->epilogue, a->act, a->out_is_f32, a->M, a->N, a->K));
    if (it == p.sym_of.end() || strcmp(it->second, p.target) != 0) return -1;
    // mode 3: every kSampleEvery-th launch of the target only.  An event pair around EVERY launch of a symbol that runs 78 times per step
    // cost the step it measures 0.5 ms of 23.6 (the records sit between the kernels on the stream); the period is odd so that a symbol
    // serving two alternating shapes (ViT proj / fc2) is sampled on both.
    if (p.mode == 3 && (p.seen++ % kSampleEvery) != 0) return -1;
  }
  if (p.recs.size() >= kMaxRecs) return -1;   // a session left in mode 1 for a long run: keep what we have, stop recording (stllm_gemm_profile starts afresh)
  Rec r;
  r.s = take_event(p);
  r.e = take_event(p);
  if (!r.s || !r.e) return -1;                // event creation failed: no record rather than a read error later
  r.sym = "";
  r.flops = 2.0 * a->M * a->N * a->K;
  r.m = a->M; r.n = a->N; r.k = a->K;
  if (hipEventRecord(r.s, reinterpret_cast<hipStream_t>(stream)) != hipSuccess) { p.pool.push_back(r.s); p.pool.push_back(r.e); return -1; }
  p.recs.push_back(r);
  return (int)p.recs.size() - 1;
}

void stllm_prof_end(int idx, int rc, const stllm_gemm_args* a, void* stream) {
  if (idx < 0) return;
  Prof& p = prof();
  if (rc != STLLM_OK || idx != (int)p.recs.size() - 1) {   // the dispatch failed: nothing ran, stllm_last_kernel() still names the PREVIOUS launch —
    if (idx == (int)p.recs.size() - 1) {                   // drop the record, learn no symbol for this shape
      p.pool.push_back(p.recs[idx].s); p.pool.push_back(p.recs[idx].e);
      p.recs.pop_back();
    }
    return;
  }
  Rec& r = p.recs[idx];
  (void)hipEventRecord(r.e, reinterpret_cast<hipStream_t>(stream));
  r.sym = stllm_last_kernel();
  p.sym_of[Key(a->dtype, a->epilogue, a->act, a->out_is_f32, a->M, a->N, a->K)] = r.sym;
}

extern "C" int stllm_gemm_profile(int mode, const char* target_symbol) {
  Prof& p = prof();
  if (mode < 0 || mode > 3 || (mode >= 2 && !target_symbol)) { stllm_set_error("stllm_gemm_profile: bad mode %d", mode); return STLLM_ERR_BAD_SHAPE; }
  for (Rec& r : p.recs) { p.pool.push_back(r.s); p.pool.push_back(r.e); }   // a new session starts empty
  p.recs.clear();
  p.mode = mode;
  p.target[0] = 0;
  p.seen = 0;
  if (mode >= 2) { strncpy(p.target, target_symbol, sizeof(p.target) - 1); p.target[sizeof(p.target) - 1] = 0; }
  return STLLM_OK;
}

extern "C" int stllm_gemm_profile_count(void) { return (int)prof().recs.size(); }

extern "C" int stllm_gemm_profile_read(int i, char* symbol, int symbol_cap, float* ms, double* flops, int* mnk3) {
  Prof& p = prof();
  if (i < 0 || i >= (int)p.recs.size()) { stllm_set_error("stllm_gemm_profile_read: record %d of %d", i, (int)p.recs.size()); return STLLM_ERR_BAD_SHAPE; }
  Rec& r = p.recs[i];
  if (hipEventSynchronize(r.e) != hipSuccess) { stllm_set_error("stllm_gemm_profile_read: event synchronise failed"); return STLLM_ERR_HIP; }
  float t = 0.f;
  if (hipEventElapsedTime(&t, r.s, r.e) != hipSuccess) { stllm_set_error("stllm_gemm_profile_read: elapsed time unavailable"); return STLLM_ERR_HIP; }
  if (ms) *ms = t;
  if (flops) *flops = r.flops;
  if (mnk3) { mnk3[0] = r.m; mnk3[1] = r.n; mnk3[2] = r.k; }
  if (symbol && symbol_cap > 0) { strncpy(symbol, r.sym, symbol_cap - 1); symbol[symbol_cap - 1] = 0; }
  return STLLM_OK;
}
