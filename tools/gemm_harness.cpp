// Stand-alone GEMM checker / timer over the C ABI (no torch): compares the 256x256 phased stream-K kernel
// ("gemm_p8" = 1) with the 128x128 kernels ("gemm_p8" = 0, "gemm_sk" = 0) on the same random operands, screens for
// races (every repeat must be bit-identical to the first) and times both.
//   build:  hipcc --offload-arch=gfx950 -O2 -o tools/gemm_harness tools/gemm_harness.cpp -Lst-llm_amd -lstllm_hip
//   run:    LD_LIBRARY_PATH=st-llm_amd tools/gemm_harness [reps]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <cmath>
#include <vector>

#include "../include/stllm_hip.h"

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } \
  } while (0)

static uint16_t f2bf(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static float bf2f(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
static uint32_t rng_state = 12345;
static float urand() {   // [-1, 1)
  rng_state = rng_state * 1664525u + 1013904223u;
  return (float)((rng_state >> 8) & 0xffff) / 32768.0f - 1.0f;
}

struct Case { const char* name; int M, N, K, epi, act, of32; };

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 20;
  const int only = argc > 2 ? atoi(argv[2]) : -1;
  const int trace = argc > 3 ? atoi(argv[3]) : 0;
  const int p8opt = argc > 4 ? atoi(argv[4]) : 1;   // 1: cost model picks the tile height, 3 / 4: force 192 / 256 rows
  const int w4opt = argc > 5 ? atoi(argv[5]) : 0;   // 0: test the phased kernel; 1 / 34 / 44: test the one-wave-per-SIMD kernel (gemm_w4)
  const int baseopt = argc > 6 ? atoi(argv[6]) : 0; // baseline ("old" column): 0 = 128x128 kernels, 1 = phased kernel (cost model)
  const int cold_mb = argc > 7 ? atoi(argv[7]) : 0; // > 0: the timed loop rotates over copies of W worth this many MiB (weights come from HBM, as in the model)
  std::vector<Case> cases = {
      {"tiny_store", 256, 256, 128, STLLM_EPI_STORE, 0, 0},
      {"edge_store", 300, 384, 192, STLLM_EPI_STORE, 0, 0},
      {"vit_qkv", 4112, 4224, 1408, STLLM_EPI_STORE, 0, 0},
      {"vit_proj", 4112, 1408, 1408, STLLM_EPI_RESID, 0, 0},
      {"vit_fc1", 4112, 6144, 1408, STLLM_EPI_STORE, 1, 0},
      {"vit_fc2", 4112, 1408, 6144, STLLM_EPI_RESID, 0, 0},
      {"llm_qkv", 576, 12288, 4096, STLLM_EPI_ROPE, 0, 0},
      {"llm_o", 576, 4096, 4096, STLLM_EPI_RESID, 0, 0},
      {"llm_gu", 576, 22016, 4096, STLLM_EPI_SWIGLU, 0, 0},
      {"llm_down", 576, 4096, 11008, STLLM_EPI_RESID, 0, 0},
      {"lm_head", 576, 32000, 4096, STLLM_EPI_STORE, 0, 1},
      {"sq4096", 4096, 4096, 4096, STLLM_EPI_STORE, 0, 0},
      {"dp_k1408", 4096, 4096, 1408, STLLM_EPI_STORE, 0, 0},
      {"sk_k1408", 4096, 4224, 1408, STLLM_EPI_STORE, 0, 0},
      {"dp_k6144", 4096, 4096, 6144, STLLM_EPI_STORE, 0, 0},
      {"sk_k4096", 4096, 4352, 4096, STLLM_EPI_STORE, 0, 0},
      {"dp2_192", 3072, 8192, 1408, STLLM_EPI_STORE, 0, 0},       // 512 tiles of 192 x 256: two whole rounds, no remainder
      {"dp2_192g", 3072, 8192, 1408, STLLM_EPI_STORE, 1, 0},      // ... with the GELU epilogue
      {"dp1_192", 3072, 4096, 1408, STLLM_EPI_STORE, 0, 0},       // 256 tiles of 192 x 256: one round
      // Q-Former at T = 16 (16 frames x 32 queries = 512 rows): self-attention qkv / out, FFN, cross-attention K|V of one layer / all six
      {"llm_qkv_store", 576, 12288, 4096, STLLM_EPI_STORE, 0, 0},
      {"llm_gu_255", 576, 21760, 4096, STLLM_EPI_SWIGLU, 0, 0},     // 255 tiles of 192 x 256: the gate/up GEMM without its 2 remainder tiles   // the Llama qkv shape without the ROPE epilogue
      {"qf_qkv", 512, 2304, 768, STLLM_EPI_STORE, 0, 0},
      {"qf_out", 512, 768, 768, STLLM_EPI_STORE, 0, 1},
      {"qf_ffn1", 512, 3072, 768, STLLM_EPI_STORE, 1, 0},
      {"qf_ffn2", 512, 768, 3072, STLLM_EPI_STORE, 0, 1},
      {"qf_xkv1", 4112, 1536, 1408, STLLM_EPI_STORE, 0, 0},
      {"qf_xkv6", 4112, 9216, 1408, STLLM_EPI_STORE, 0, 0},
      // training step, 16 clips x 576 tokens = 9216 rows: dgrad = gemm(dY, W^T), wgrad = gemm(dY^T, X^T) with fp32 output
      {"tr_dgrad_down", 9216, 11008, 4096, STLLM_EPI_STORE, 0, 0},
      {"tr_dgrad_gu", 9216, 4096, 22016, STLLM_EPI_STORE, 0, 0},
      {"tr_dgrad_qkv", 9216, 4096, 12288, STLLM_EPI_STORE, 0, 0},
      {"tr_wgrad_down", 4096, 11008, 9216, STLLM_EPI_STORE, 0, 1},
      {"tr_wgrad_gu", 22016, 4096, 9216, STLLM_EPI_STORE, 0, 1},
      {"tr_wgrad_qkv", 12288, 4096, 9216, STLLM_EPI_STORE, 0, 1},
      {"tr_wgrad_lm", 32000, 4096, 9216, STLLM_EPI_STORE, 0, 1},
      {"vit_fc1_noact", 4112, 6144, 1408, STLLM_EPI_STORE, 0, 0},   // the fc1 shape without its GELU (epilogue timeline experiments)
      // round 5, two workgroups per CU on 128 x 128 tiles (w4 tile code 22): shapes whose 128 x 128 tiles make whole rounds of 512
      {"dual_fc1", 4096, 6144, 1408, STLLM_EPI_STORE, 1, 0},        // 32 x 48 = 1536 tiles = 3 rounds of 512 (256 x 192: 512 tiles = 2 rounds of 256)
      {"dual_qkv", 4096, 4096, 1408, STLLM_EPI_STORE, 0, 0},        // 1024 tiles = 2 rounds
      {"dual_fc2", 4096, 2048, 6144, STLLM_EPI_RESID, 0, 0},        // 512 tiles = 1 round, long K
  };
  int dev_lds = 0;
  CK(hipDeviceGetAttribute(&dev_lds, hipDeviceAttributeMaxSharedMemoryPerBlock, 0));
  printf("max LDS per block: %d bytes\n", dev_lds);
  const int64_t ws_bytes = stllm_gemm_workspace_bytes();
  void* ws;
  CK(hipMalloc(&ws, ws_bytes));
  CK(hipMemset(ws, 0, ws_bytes));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  int bad = 0;
  for (size_t ci = 0; ci < cases.size(); ++ci) {
    if (only >= 0 && (int)ci != only) continue;
    const Case& c = cases[ci];
    const int M = c.M, N = c.N, K = c.K;
    const int No = c.epi == STLLM_EPI_SWIGLU ? N / 2 : N;
    const bool f32o = c.epi == STLLM_EPI_RESID || c.of32;
    const size_t oes = f32o ? 4 : 2;
    const int ldpad = getenv("HARNESS_LDPAD") ? atoi(getenv("HARNESS_LDPAD")) : 0;   // elements added to the row stride of A and W (channel-aliasing experiments: K = 4096 rows are 8 KiB apart)
    const int LD = K + ldpad;
    std::vector<uint16_t> hA((size_t)M * LD), hW((size_t)N * LD);
    for (auto& v : hA) v = f2bf(urand());
    for (auto& v : hW) v = f2bf(urand() * 0.05f);
    std::vector<float> hb(N), hres((size_t)M * N), hcos(576 * 64), hsin(576 * 64);
    for (auto& v : hb) v = urand() * 0.5f;
    for (auto& v : hres) v = urand();
    for (int i = 0; i < 576 * 64; ++i) { hcos[i] = cosf(0.37f * i); hsin[i] = sinf(0.37f * i); }
    void *dA, *dA2, *dW, *dout0, *dout1, *dfirst;
    float *db, *dres, *dcos, *dsin;
    CK(hipMalloc(&dA, hA.size() * 2)); CK(hipMalloc(&dA2, hA.size() * 2)); CK(hipMalloc(&dW, hW.size() * 2));
    CK(hipMalloc(&db, N * 4)); CK(hipMalloc(&dres, hres.size() * 4));
    CK(hipMalloc(&dcos, hcos.size() * 4)); CK(hipMalloc(&dsin, hsin.size() * 4));
    const size_t obytes = (size_t)M * No * oes;
    CK(hipMalloc(&dout0, obytes)); CK(hipMalloc(&dout1, obytes)); CK(hipMalloc(&dfirst, obytes));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
    for (auto& v : hA) v = f2bf(urand());   // a second operand set: alternating launches expose stale slab reads
    CK(hipMemcpy(dA2, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), N * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dres, hres.data(), hres.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dcos, hcos.data(), hcos.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dsin, hsin.data(), hsin.size() * 4, hipMemcpyHostToDevice));

    std::vector<void*> dWs;   // cold-weight mode: more copies of W than the 256 MiB of MALL hold
    if (cold_mb > 0) {
      const size_t wb = hW.size() * 2;
      const int n = (int)(((size_t)cold_mb << 20) / wb) + 2;
      for (int i = 0; i < n; ++i) {
        void* q;
        CK(hipMalloc(&q, wb));
        CK(hipMemcpy(q, dW, wb, hipMemcpyDeviceToDevice));
        dWs.push_back(q);
      }
    }
    stllm_gemm_args a;
    memset(&a, 0, sizeof(a));
    a.dtype = STLLM_BF16; a.epilogue = c.epi; a.act = c.act; a.out_is_f32 = c.of32;
    a.A = dA; a.lda = LD; a.W = dW; a.ldw = LD; a.bias = db; a.ldo = No;
    a.resid = dres; a.ldr = N; a.aux0 = dcos; a.aux1 = dsin; a.rope_seq = 576; a.rope_cols = (N / 3) * 2 / 128 * 128;
    a.M = M; a.N = N; a.K = K; a.workspace = ws; a.workspace_bytes = ws_bytes;

    auto run = [&](int p8, void* out, float* us, const char** kname) -> int {
      stllm_set_option("gemm_p8", p8 ? p8opt : baseopt);
      stllm_set_option("gemm_w4", p8 ? w4opt : 0);
      stllm_set_option("gemm_sk", p8 ? -1 : 0);
      a.out = out;
      CK(hipMemsetAsync(out, 0xff, obytes, st));
      int rc = stllm_gemm(&a, st);
      if (rc != STLLM_OK) { printf("  stllm_gemm rc=%d: %s\n", rc, stllm_last_error()); return rc; }
      *kname = stllm_last_kernel();
      CK(hipStreamSynchronize(st));
      if (p8) {   // race / staleness screen: alternate two operand sets; every repeat bit-identical to its first result
        std::vector<char> h0(obytes), h2(obytes), h1(obytes);
        CK(hipMemcpy(h0.data(), out, obytes, hipMemcpyDeviceToHost));
        a.A = dA2;
        stllm_gemm(&a, st);
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(h2.data(), out, obytes, hipMemcpyDeviceToHost));
        for (int r = 0; r < 8; ++r) {
          a.A = (r & 1) ? dA2 : dA;
          CK(hipMemsetAsync(out, 0xff, obytes, st));
          stllm_gemm(&a, st);
          CK(hipStreamSynchronize(st));
          CK(hipMemcpy(h1.data(), out, obytes, hipMemcpyDeviceToHost));
          const std::vector<char>& ref = (r & 1) ? h2 : h0;
          if (memcmp(ref.data(), h1.data(), obytes) != 0) {
            size_t nd = 0, first = 0;
            for (size_t i = 0; i < obytes; ++i) if (ref[i] != h1[i]) { if (!nd) first = i; ++nd; }
            printf("  RACE: repeat %d differs from the first run in %zu bytes (first at element %zu = row %zu col %zu)\n", r, nd,
                   first / oes, first / oes / No, first / oes % No);
            a.A = dA;
            return -100;
          }
        }
        a.A = dA;
        CK(hipMemsetAsync(out, 0xff, obytes, st));
        stllm_gemm(&a, st);
        CK(hipStreamSynchronize(st));
      }
      CK(hipEventRecord(e0, st));
      for (int r = 0; r < reps; ++r) {
        if (!dWs.empty()) a.W = dWs[r % dWs.size()];
        stllm_gemm(&a, st);
      }
      CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
      a.W = dW;
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      *us = ms * 1000.0f / reps;
      return 0;
    };
    float us0 = 0, us1 = 0;
    const char *k0 = "", *k1 = "";
    int rc0 = run(0, dout0, &us0, &k0);
    int rc1 = run(1, dout1, &us1, &k1);
    double maxabs = 0, maxref = 0;
    size_t nbad = 0, firstbad = 0;
    if (rc0 == 0 && rc1 == 0) {
      std::vector<char> h0(obytes), h1(obytes);
      CK(hipMemcpy(h0.data(), dout0, obytes, hipMemcpyDeviceToHost));
      CK(hipMemcpy(h1.data(), dout1, obytes, hipMemcpyDeviceToHost));
      const size_t n = (size_t)M * No;
      for (size_t i = 0; i < n; ++i) {
        float x, y;
        if (f32o) { x = ((float*)h0.data())[i]; y = ((float*)h1.data())[i]; }
        else { x = bf2f(((uint16_t*)h0.data())[i]); y = bf2f(((uint16_t*)h1.data())[i]); }
        const double d = fabs((double)x - y), tol = f32o ? 2e-3 + 1e-4 * fabs(x) : 2e-2 + 1.6e-2 * fabs(x);
        if (!(d <= tol)) { if (!nbad) firstbad = i; ++nbad; }
        if (d > maxabs) maxabs = d;
        if (fabs(x) > maxref) maxref = fabs(x);
      }
    }
    if (trace && rc1 == 0) {   // in-kernel timeline of one p8 launch (gemm_debug bit 16)
      unsigned long long* ddbg;
      const size_t dbytes = 256 * 64 * 8;
      CK(hipMalloc(&ddbg, dbytes));
      CK(hipMemset(ddbg, 0, dbytes));
      stllm_set_option("gemm_p8", p8opt);
      stllm_set_option("gemm_debug", 16);
      a.frames = (const float*)ddbg;
      a.out = dout1;
      stllm_gemm(&a, st);
      CK(hipStreamSynchronize(st));
      stllm_set_option("gemm_debug", 0);
      a.frames = nullptr;
      std::vector<unsigned long long> h(256 * 64);
      CK(hipMemcpy(h.data(), ddbg, dbytes, hipMemcpyDeviceToHost));
      unsigned long long t0 = ~0ull, t1 = 0;
      for (int g = 0; g < 256; ++g) {
        const int n = (int)h[g * 64];
        for (int i = 1; i < n; ++i) {
          if (h[g * 64 + i] == 0) continue;
          const unsigned long long t = h[g * 64 + i] & 0x00ffffffffffffffull;
          if (t < t0) t0 = t;
          if (t > t1) t1 = t;
        }
      }
      printf("  trace: kernel span %llu ticks (events said %.1f us => %.1f ticks/us)\n", t1 - t0, us1, (double)(t1 - t0) / us1);
      double sum[16][16] = {{0}};
      int cnt[16][16] = {{0}};
      for (int g = 0; g < 256; ++g) {
        const int n = (int)h[g * 64];
        if (n < 2) continue;
        if (g < 6 || g % 37 == 0) printf("  wg %3d:", g);
        unsigned long long prev = 0;   // unused slots stay 0: skipped
        for (int i = 1; i < n; ++i) {
          if (h[g * 64 + i] == 0) continue;
          const int tag = (int)(h[g * 64 + i] >> 56);
          const unsigned long long t = h[g * 64 + i] & 0x00ffffffffffffffull;
          if (g < 6 || g % 37 == 0) printf(" [%d]%llu", tag, t - t0);
          if (prev) {
            const int ptag = (int)(prev >> 56);
            sum[ptag][tag] += (double)(t - (prev & 0x00ffffffffffffffull));
            cnt[ptag][tag]++;
          }
          prev = h[g * 64 + i];
        }
        if (g < 6 || g % 37 == 0) printf("\n");
      }
      for (int x = 0; x < 16; ++x)
        for (int y = 0; y < 16; ++y)
          if (cnt[x][y]) printf("  tag %2d -> %2d : n=%4d  mean %9.0f ticks\n", x, y, cnt[x][y], sum[x][y] / cnt[x][y]);
      hipFree(ddbg);
    }
    const double tf = 2.0 * M * N * K * 1e-6;
    printf("%-10s M=%5d N=%6d K=%6d  old %7.1f us (%6.1f TF) [%s]   p8 %7.1f us (%6.1f TF) [%s]  maxabs %.3g (ref max %.3g) bad %zu", c.name, M,
           N, K, us0, tf / us0, k0, us1, tf / us1, k1, maxabs, maxref, nbad);
    if (nbad) printf(" first bad row %zu col %zu", firstbad / No, firstbad % No);
    printf(" %s\n", (rc0 || rc1 || nbad) ? "FAIL" : "ok");
    fflush(stdout);
    if (rc0 || rc1 || nbad) ++bad;
    hipFree(dA); hipFree(dA2); hipFree(dW); hipFree(db); hipFree(dres); hipFree(dcos); hipFree(dsin); hipFree(dout0); hipFree(dout1); hipFree(dfirst);
    for (void* q : dWs) hipFree(q);
    if (rc1 == -100) break;
  }
  printf("%s\n", bad ? "HARNESS FAIL" : "HARNESS OK");
  return bad ? 1 : 0;
}
