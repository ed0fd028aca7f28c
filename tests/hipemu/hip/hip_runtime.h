// TEST-ONLY host emulation of the small HIP surface used by train_ops.hip / attention_bwd.hip, so that the LOGIC of those
// kernels (indexing, reductions, barriers, masks) is checked by the CPU suite against the contract backend
// (tests/test_kernels_emulated_cpu.py).  One OS thread per HIP thread, blocks run one after another;
// __syncthreads = std::barrier over the block, wave shuffles = exchange buffer + per-wave barrier (64 lanes).
// Nothing in st-llm_amd/ includes this file; it says nothing about performance or ISA-level behaviour.
#pragma once
#include <algorithm>
#include <atomic>
#include <barrier>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static              /* blocks run one at a time: a static IS block-shared */
#define EMU_DYN_SHARED extern          /* build step rewrites `extern __shared__` to this */

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};
struct emu_idx { unsigned x, y, z; };
inline thread_local emu_idx threadIdx, blockIdx;
inline thread_local dim3 blockDim, gridDim;

struct float4 { float x, y, z, w; };
struct uint4 { uint32_t x, y, z, w; };
struct uint2 { uint32_t x, y; };
inline float4 make_float4(float a, float b, float c, float d) { return float4{a, b, c, d}; }

using std::max;
using std::min;
inline float __expf(float x) { return expf(x); }
inline float __logf(float x) { return logf(x); }
inline float __log2f(float x) { return log2f(x); }
inline float __frcp_rn(float x) { return 1.0f / x; }
inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
// correctly rounded single operations (no contraction): plain IEEE ops on the host (-ffp-contract=off in the emulated build)
inline double __dadd_rn(double a, double b) { volatile double r = a + b; return r; }
inline double __dmul_rn(double a, double b) { volatile double r = a * b; return r; }
inline double __ddiv_rn(double a, double b) { volatile double r = a / b; return r; }
inline float __fdiv_rn(float a, float b) { volatile float r = a / b; return r; }
inline float __fsub_rn(float a, float b) { volatile float r = a - b; return r; }
inline float emu_med3(float a, float b, float c) { return std::max(std::min(a, b), std::min(std::max(a, b), c)); }
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z) emu_mfma_32x32<8>(a, b, c)
#define __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, x, y, z) emu_mfma_32x32<8>(a, b, c)
#define __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, x, y, z) emu_mfma_32x32_f32(a, b, c)
#define __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, x, y, z) emu_mfma_16x16x32(a, b, c)
#define __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, x, y, z) emu_mfma_16x16x32(a, b, c)
#define __builtin_amdgcn_fmed3f(a, b, c) emu_med3(a, b, c)
#define __builtin_amdgcn_global_load_lds(g, l, n, o, a) emu_global_load_lds(g, l, n, o)
#define __builtin_amdgcn_readfirstlane(x) (x)
#define __builtin_amdgcn_s_sleep(n) ((void)0)
#define __builtin_amdgcn_fence(...) ((void)0)
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __hip_atomic_load(p, order, scope) (*(p))
#define __hip_atomic_store(p, v, order, scope) (*(p) = (v))
#define __hip_atomic_fetch_or(p, v, order, scope) (*(p) |= (v))
#define __builtin_amdgcn_exp2f(x) exp2f(x)
#define __builtin_amdgcn_wave_barrier() emu_wave_barrier()   /* lanes are independent threads here: make it a real barrier */

// ---- block / wave state ---------------------------------------------------------------------------------------------
struct emu_block {
  std::barrier<> all;
  std::vector<std::unique_ptr<std::barrier<>>> wave;
  // wave collectives exchange through DOUBLE-buffered arrays: every lane of a wave runs the same sequence of collectives, so a
  // per-thread phase bit selects the buffer and ONE barrier per collective suffices (a buffer is rewritten only two collectives
  // later, after every lane has passed the barrier of the collective in between, i.e. finished reading it)
  std::vector<float> xch;               // [2][thread]
  std::vector<float> ma, mb;            // MFMA operand exchange: [2][thread][8]
  int n;
  explicit emu_block(int n_) : all(n_), xch(2 * n_), ma(16 * n_), mb(16 * n_), n(n_) {
    for (int w = 0; w < (n + 63) / 64; ++w) wave.emplace_back(new std::barrier<>(std::min(64, n - 64 * w)));
  }
};
inline thread_local emu_block* emu_cur = nullptr;
inline thread_local int emu_phase = 0;   // toggles on every wave collective of this thread
inline void __syncthreads() { emu_cur->all.arrive_and_wait(); }
inline float __shfl_xor(float v, int mask, int width = 64) {
  (void)width;
  const int t = threadIdx.x;
  float* buf = emu_cur->xch.data() + (emu_phase ^= 1) * emu_cur->n;
  buf[t] = v;
  emu_cur->wave[t >> 6]->arrive_and_wait();
  return buf[t ^ mask];
}
// LDS-DMA (global_load_lds): every lane copies `n` bytes from its own global address to the wave-uniform LDS base + lane * n
template <class G, class L> inline void emu_global_load_lds(G g, L l, int n, int off) {
  std::memcpy((char*)(l) + off + (threadIdx.x & 63) * n, (const char*)(g), n);
}
inline int __any(int pred) {                                  // wave vote
  const int t = threadIdx.x, w0 = t & ~63;
  float* buf = emu_cur->xch.data() + (emu_phase ^= 1) * emu_cur->n;
  buf[t] = pred ? 1.0f : 0.0f;
  emu_cur->wave[t >> 6]->arrive_and_wait();
  int r = 0;
  for (int l = 0; l < 64 && w0 + l < emu_cur->n; ++l) r |= buf[w0 + l] != 0.0f;
  return r;
}
inline void emu_wave_barrier() { emu_cur->wave[threadIdx.x >> 6]->arrive_and_wait(); }

// v_mfma_f32_32x32x{8,16}_{f16,bf16}: D[32x32] += A[32xK] B[Kx32], K = 2*KPL; lane l holds A[l%32][KPL*(l/32) .. +KPL) and
// B[KPL*(l/32) .. +KPL)[l%32]; D register j of lane l is D[8*(j/4) + 4*(l/32) + j%4][l%32]  (CDNA3/4 ISA guide, 32x32 layouts).
template <int KPL, typename VA, typename VC> inline VC emu_mfma_32x32(VA a, VA b, VC c) {
  const int t = threadIdx.x, lane = t & 63, w0 = t & ~63;
  const int ph = (emu_phase ^= 1) * 8 * emu_cur->n;
  float* A = emu_cur->ma.data() + ph;
  float* B = emu_cur->mb.data() + ph;
  for (int e = 0; e < KPL; ++e) { A[8 * t + e] = (float)a[e]; B[8 * t + e] = (float)b[e]; }
  emu_cur->wave[t >> 6]->arrive_and_wait();
  const int n = lane & 31;
  for (int j = 0; j < 16; ++j) {
    const int m = 8 * (j / 4) + 4 * (lane / 32) + (j % 4);
    float acc = c[j];
    for (int k = 0; k < 2 * KPL; ++k) acc += A[8 * (w0 + m + 32 * (k / KPL)) + k % KPL] * B[8 * (w0 + n + 32 * (k / KPL)) + k % KPL];
    c[j] = acc;
  }
  return c;
}
// v_mfma_f32_16x16x32_{f16,bf16}: D[16x16] += A[16x32] B[32x16]; lane l holds 8 k values of A row l%16 and of B column l%16 for
// k-group l/16 (which 8 of the 32 does not matter: A and B use the same grouping); D register j of lane l is D[4*(l/16) + j][l%16].
template <typename VA, typename VC> inline VC emu_mfma_16x16x32(VA a, VA b, VC c) {
  const int t = threadIdx.x, lane = t & 63, w0 = t & ~63;
  const int ph = (emu_phase ^= 1) * 8 * emu_cur->n;
  float* A = emu_cur->ma.data() + ph;
  float* B = emu_cur->mb.data() + ph;
  for (int e = 0; e < 8; ++e) { A[8 * t + e] = (float)a[e]; B[8 * t + e] = (float)b[e]; }
  emu_cur->wave[t >> 6]->arrive_and_wait();
  const int n = lane & 15;
  for (int j = 0; j < 4; ++j) {
    const int m = 4 * (lane / 16) + j;
    float acc = c[j];
    for (int g = 0; g < 4; ++g)
      for (int e = 0; e < 8; ++e) acc += A[8 * (w0 + m + 16 * g) + e] * B[8 * (w0 + n + 16 * g) + e];
    c[j] = acc;
  }
  return c;
}
// v_dot2c_f32_{bf16,f16}: c + a[0] * b[0] + a[1] * b[1]
template <typename V2> inline float emu_fdot2(V2 a, V2 b, float c) { return c + ((float)a[0] * (float)b[0] + (float)a[1] * (float)b[1]); }
#define __builtin_amdgcn_fdot2_f32_bf16(a, b, c, clamp) emu_fdot2(a, b, c)
#define __builtin_amdgcn_fdot2(a, b, c, clamp) emu_fdot2(a, b, c)
// ds_read_b64_tr_b16: every lane fetches 8 bytes (4 x 16 bit) at its own address; within a 16-lane group lane i receives element
// i % 4 of the fetches of lanes i / 4, i / 4 + 4, i / 4 + 8, i / 4 + 12 (measured on gfx950, tools/tr_probe.hip)
inline unsigned long long emu_ds_read_tr_b16(const char* p) {
  const int t = threadIdx.x, w0 = t & ~63, lane = t & 63;
  const int ph = (emu_phase ^= 1) * 8 * emu_cur->n;
  float* X = emu_cur->ma.data() + ph;
  const uint16_t* src = reinterpret_cast<const uint16_t*>(p);
  for (int e = 0; e < 4; ++e) X[8 * t + e] = (float)src[e];
  emu_cur->wave[t >> 6]->arrive_and_wait();
  const int grp = lane & ~15, i = lane & 15;
  unsigned long long v = 0;
  for (int j = 0; j < 4; ++j) v |= (unsigned long long)(uint16_t)X[8 * (w0 + grp + (i / 4) + 4 * j) + (i % 4)] << (16 * j);
  return v;
}
template <typename VC> inline VC emu_mfma_32x32_f32(float a, float b, VC c) {   // 32x32x2: lane l holds A[l%32][l/32], B[l/32][l%32]
  const int t = threadIdx.x, lane = t & 63, w0 = t & ~63;
  const int ph = (emu_phase ^= 1) * 8 * emu_cur->n;
  float* A = emu_cur->ma.data() + ph;
  float* B = emu_cur->mb.data() + ph;
  A[8 * t] = a;
  B[8 * t] = b;
  emu_cur->wave[t >> 6]->arrive_and_wait();
  const int n = lane & 31;
  for (int j = 0; j < 16; ++j) {
    const int m = 8 * (j / 4) + 4 * (lane / 32) + (j % 4);
    float acc = c[j];
    for (int k = 0; k < 2; ++k) acc += A[8 * (w0 + m + 32 * k)] * B[8 * (w0 + n + 32 * k)];
    c[j] = acc;
  }
  return c;
}

inline float atomicAdd(float* addr, float v) {
  std::atomic_ref<float> a(*addr);
  float old = a.load();
  while (!a.compare_exchange_weak(old, old + v)) {}
  return old;
}

// ---- runtime API stubs ------------------------------------------------------------------------------------------------
typedef void* hipStream_t;
enum hipError_t { hipSuccess = 0, hipErrorUnknown = 999 };
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
enum hipMemcpyKind { hipMemcpyDeviceToHost = 2 };
inline hipError_t hipGetLastError() { return hipSuccess; }
inline const char* hipGetErrorString(hipError_t) { return "emulated"; }
inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }
struct hipDeviceProp_t { int multiProcessorCount = 6; };          // a small persistent grid keeps the emulation cheap
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { *p = hipDeviceProp_t(); return hipSuccess; }
template <class K> inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, K, int, size_t) { *n = 1; return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { std::memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }

template <typename K, typename... A>
void emu_launch(K kern, dim3 grid, dim3 block, size_t /*lds*/, hipStream_t /*stream*/, A... args) {
  const int n = block.x * block.y * block.z;
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        emu_block blk(n);
        std::vector<std::thread> th;
        th.reserve(n);
        for (int t = 0; t < n; ++t)
          th.emplace_back([&, t]() {
            threadIdx = emu_idx{(unsigned)t, 0, 0};
            blockIdx = emu_idx{bx, by, bz};
            blockDim = block;
            gridDim = grid;
            emu_cur = &blk;
            emu_phase = 0;
            kern(args...);
            blk.wave[t >> 6]->arrive_and_drop();   // exited lanes / waves no longer take part in barriers (as in hardware)
            blk.all.arrive_and_drop();
          });
        for (auto& x : th) x.join();
      }
}
#define hipLaunchKernelGGL(kern, grid, block, lds, stream, ...) emu_launch(kern, grid, block, lds, stream, __VA_ARGS__)
