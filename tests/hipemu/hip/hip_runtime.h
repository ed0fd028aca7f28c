// TEST-ONLY host emulation of the small HIP surface used by train_ops.hip / attention_bwd.hip, so that the LOGIC of those
// kernels (indexing, reductions, barriers, masks) is checked by the CPU suite against the contract backend
// (tests/test_train_kernels_emulated_cpu.py).  One OS thread per HIP thread, blocks run one after another;
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
inline float __frcp_rn(float x) { return 1.0f / x; }
inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
inline float emu_med3(float a, float b, float c) { return std::max(std::min(a, b), std::min(std::max(a, b), c)); }
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z) (c)
#define __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, x, y, z) (c)
#define __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, x, y, z) (c)
#define __builtin_amdgcn_fmed3f(a, b, c) emu_med3(a, b, c)
#define __builtin_amdgcn_global_load_lds(g, l, n, o, a) ((void)0)

// ---- block / wave state ---------------------------------------------------------------------------------------------
struct emu_block {
  std::barrier<> all;
  std::vector<std::unique_ptr<std::barrier<>>> wave;
  std::vector<float> xch;
  explicit emu_block(int n) : all(n), xch(n) {
    for (int w = 0; w < (n + 63) / 64; ++w) wave.emplace_back(new std::barrier<>(std::min(64, n - 64 * w)));
  }
};
inline thread_local emu_block* emu_cur = nullptr;
inline void __syncthreads() { emu_cur->all.arrive_and_wait(); }
inline float __shfl_xor(float v, int mask, int width = 64) {
  (void)width;
  const int t = threadIdx.x;
  auto& bar = *emu_cur->wave[t >> 6];
  emu_cur->xch[t] = v;
  bar.arrive_and_wait();
  const float r = emu_cur->xch[t ^ mask];
  bar.arrive_and_wait();
  return r;
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
            kern(args...);
            blk.wave[t >> 6]->arrive_and_drop();   // exited lanes / waves no longer take part in barriers (as in hardware)
            blk.all.arrive_and_drop();
          });
        for (auto& x : th) x.join();
      }
}
#define hipLaunchKernelGGL(kern, grid, block, lds, stream, ...) emu_launch(kern, grid, block, lds, stream, __VA_ARGS__)
