// Shared device/host helpers for libstllm_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/stllm_hip.h"

// ---------------------------------------------------------------------------------------------
// host: error plumbing
// ---------------------------------------------------------------------------------------------
void stllm_set_error(const char* fmt, ...);
void stllm_set_last_kernel(const char* name);  // static-lifetime string: symbol family of the last launch

#define STLLM_CHECK_ARG(cond, ...)                 \
  do {                                             \
    if (!(cond)) {                                 \
      stllm_set_error(__VA_ARGS__);                \
      return STLLM_ERR_BAD_SHAPE;                  \
    }                                              \
  } while (0)

#define STLLM_CHECK_LAUNCH(what)                                                   \
  do {                                                                             \
    hipError_t e_ = hipGetLastError();                                             \
    if (e_ != hipSuccess) {                                                        \
      stllm_set_error("%s: HIP launch error: %s", what, hipGetErrorString(e_));    \
      return STLLM_ERR_HIP;                                                        \
    }                                                                              \
  } while (0)

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// ---------------------------------------------------------------------------------------------
// host: dispatch options — per THREAD (include/stllm_hip.h: no process-global state besides thread-local data).  The first use in
// a thread reads the STLLM_* environment variables ONCE, all of them, before any dispatch decision; stllm_set_option() writes the
// calling thread's copy (a test / experiment hook: two host threads with different options do not race).
// ---------------------------------------------------------------------------------------------
#include "options.h"

// Function attributes (dynamic LDS opt-in) and occupancy answers are PER DEVICE: one bit per device ordinal and call site, so a
// process that drives several GPUs sets them on each (a process-wide `static bool` left device 1 without the opt-in).
struct StllmPerDevice {
  unsigned long long seen = 0;
  int value[64] = {0};
  // current device ordinal; *first = true when this call site has not run on that device yet
  int enter(bool* first) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev &= 63;
    *first = !((__atomic_load_n(&seen, __ATOMIC_ACQUIRE) >> dev) & 1ull);
    return dev;
  }
  void done(int dev) { __atomic_fetch_or(&seen, 1ull << dev, __ATOMIC_RELEASE); }
};

// ---------------------------------------------------------------------------------------------
// device: vector types
// ---------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(2))) int i32x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;

struct bf16_t { uint16_t v; };   // tag types for templates (storage = 2 bytes)
struct f16_t { uint16_t v; };

// fp32 -> bf16 round-to-nearest-even (matches torch .to(bfloat16))
__device__ __forceinline__ uint16_t f32_to_bf16_bits(float f) {
  uint32_t u = __builtin_bit_cast(uint32_t, f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float bf16_bits_to_f32(uint16_t h) {
  return __builtin_bit_cast(float, (uint32_t)h << 16);
}
__device__ __forceinline__ uint16_t f32_to_f16_bits(float f) {
  _Float16 h = (_Float16)f;  // v_cvt_f16_f32, RNE
  return __builtin_bit_cast(uint16_t, h);
}
__device__ __forceinline__ float f16_bits_to_f32(uint16_t h) {
  return (float)__builtin_bit_cast(_Float16, h);
}

template <typename T> struct Elem;
template <> struct Elem<bf16_t> {
  static constexpr int kBytes = 2;
  static constexpr bool kIsF32 = false;
  __device__ static __forceinline__ uint16_t pack(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }  // RNE
  // two values -> one dword: a single v_cvt_pk_bf16_f32 (round-to-nearest-even, == torch .to(bfloat16))
  __device__ static __forceinline__ uint32_t pack2(float lo, float hi) {
    f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
  }
  __device__ static __forceinline__ float unpack(uint16_t h) { return bf16_bits_to_f32(h); }
  __device__ static __forceinline__ f32x16 mfma(i32x4 a, i32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};
template <> struct Elem<f16_t> {
  static constexpr int kBytes = 2;
  static constexpr bool kIsF32 = false;
  __device__ static __forceinline__ uint16_t pack(float f) { return f32_to_f16_bits(f); }
  __device__ static __forceinline__ uint32_t pack2(float lo, float hi) {   // v_cvt_pk_f16_f32, RNE
    f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
  }
  __device__ static __forceinline__ float unpack(uint16_t h) { return f16_bits_to_f32(h); }
  __device__ static __forceinline__ f32x16 mfma(i32x4 a, i32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
};
template <> struct Elem<float> {
  static constexpr int kBytes = 4;
  static constexpr bool kIsF32 = true;
  __device__ static __forceinline__ uint16_t pack(float) { return 0; }  // never used: fp32 outputs are stored as-is
  __device__ static __forceinline__ uint32_t pack2(float, float) { return 0; }
  // One 16-byte fragment = 4 consecutive k of this lane's half; the k <-> (step, half) slot map is
  // the same for A and B, so 4 exact-fp32 MFMAs (K=2 each) consume one fragment pair.
  __device__ static __forceinline__ f32x16 mfma(i32x4 a, i32x4 b, f32x16 c) {
    f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0], bf[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1], bf[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(af[2], bf[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(af[3], bf[3], c, 0, 0, 0);
    return c;
  }
};

// store one value of the compute dtype
template <typename T> __device__ __forceinline__ void store_elem(void* base, int64_t idx, float v) {
  if constexpr (Elem<T>::kIsF32) reinterpret_cast<float*>(base)[idx] = v;
  else reinterpret_cast<uint16_t*>(base)[idx] = Elem<T>::pack(v);
}
template <typename T> __device__ __forceinline__ float load_elem(const void* base, int64_t idx) {
  if constexpr (Elem<T>::kIsF32) return reinterpret_cast<const float*>(base)[idx];
  else return Elem<T>::unpack(reinterpret_cast<const uint16_t*>(base)[idx]);
}

// 8- / 4-element row vectors of the compute dtype <-> fp32 (16-byte / 8-byte accesses; idx in elements, suitably aligned)
template <typename T> __device__ __forceinline__ void load8(const void* base, int64_t idx, float* f) {
  if constexpr (Elem<T>::kIsF32) {
    const float4* p = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + idx);
    const float4 a = p[0], b = p[1];
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
  } else {
    const uint4 u = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(base) + idx);
    const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      f[2 * e] = Elem<T>::unpack((uint16_t)(w[e] & 0xffffu));
      f[2 * e + 1] = Elem<T>::unpack((uint16_t)(w[e] >> 16));
    }
  }
}
template <typename T> __device__ __forceinline__ void store8(void* base, int64_t idx, const float* f) {
  if constexpr (Elem<T>::kIsF32) {
    float4* p = reinterpret_cast<float4*>(reinterpret_cast<float*>(base) + idx);
    p[0] = make_float4(f[0], f[1], f[2], f[3]);
    p[1] = make_float4(f[4], f[5], f[6], f[7]);
  } else {
    uint4 u;
    u.x = Elem<T>::pack2(f[0], f[1]); u.y = Elem<T>::pack2(f[2], f[3]);
    u.z = Elem<T>::pack2(f[4], f[5]); u.w = Elem<T>::pack2(f[6], f[7]);
    *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(base) + idx) = u;
  }
}
template <typename T> __device__ __forceinline__ void load4(const void* base, int64_t idx, float* f) {
  if constexpr (Elem<T>::kIsF32) {
    const float4 a = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + idx);
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w;
  } else {
    const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(base) + idx);
    f[0] = Elem<T>::unpack((uint16_t)(u.x & 0xffffu)); f[1] = Elem<T>::unpack((uint16_t)(u.x >> 16));
    f[2] = Elem<T>::unpack((uint16_t)(u.y & 0xffffu)); f[3] = Elem<T>::unpack((uint16_t)(u.y >> 16));
  }
}

// async global -> LDS, 16 bytes per lane; LDS destination = wave-uniform base + lane*16
__device__ __forceinline__ void glds16(const void* g, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// exact-erf GELU (nn.GELU()): erf by Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7)
__device__ __forceinline__ float erf_as(float x) {
  float ax = fabsf(x);
  float t = __frcp_rn(fmaf(0.3275911f, ax, 1.0f));
  float p = fmaf(fmaf(fmaf(fmaf(1.061405429f, t, -1.453152027f), t, 1.421413741f), t, -0.284496736f), t, 0.254829592f) * t;
  float r = 1.0f - p * __expf(-ax * ax);
  return copysignf(r, x);
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752f)); }
// GELU for 16-bit OUTPUTS of the GEMM epilogues: Phi(x) = 0.5 + xc * P(xc^2), xc = clamp(x, -X0, X0), P of degree 8 — a minimax fit of
// 0.5 * erf(x / sqrt2) / x on |x| <= 4.3 (max |Phi error| 9.9e-6) under the END-POINT CONSTRAINT X0 * P(X0^2) = 0.5 (tools/fit_gelu_poly.py),
// so that the clamp alone gives both tails: Phi(-X0) = 8.7e-11 and Phi(X0) = 1 in float32 at X0 = 4.2983036 (no compare + select per
// element as in round 2).  |GELU error| <= 1.2e-5 |x| in float32: 1.7e-5 on |x| <= 2, 5.1e-5 at the clamp (below the fp16 / bf16 rounding of any
// result of magnitude >= 0.1; in the far negative tail, where GELU itself is < 1e-3, the relative error reaches 1e-2).  13 packed VALU ops
// per PAIR of values (round 2: 17) against ~12 + two quarter-rate transcendentals per value for the erf form.
#define STLLM_GELU_X0 4.2983036f
#define STLLM_GELU_C8 5.146456e-11f
#define STLLM_GELU_C7 -5.025569e-09f
#define STLLM_GELU_C6 2.1660718e-07f
#define STLLM_GELU_C5 -5.48717e-06f
#define STLLM_GELU_C4 9.219258e-05f
#define STLLM_GELU_C3 -0.0011024966f
#define STLLM_GELU_C2 0.009800259f
#define STLLM_GELU_C1 -0.0663265f
#define STLLM_GELU_C0 0.39889646f
__device__ __forceinline__ float gelu_poly16(float x) {
  const float xc = __builtin_amdgcn_fmed3f(x, -STLLM_GELU_X0, STLLM_GELU_X0);
  const float t = xc * xc;
  float p = fmaf(STLLM_GELU_C8, t, STLLM_GELU_C7);
  p = fmaf(p, t, STLLM_GELU_C6);
  p = fmaf(p, t, STLLM_GELU_C5);
  p = fmaf(p, t, STLLM_GELU_C4);
  p = fmaf(p, t, STLLM_GELU_C3);
  p = fmaf(p, t, STLLM_GELU_C2);
  p = fmaf(p, t, STLLM_GELU_C1);
  p = fmaf(p, t, STLLM_GELU_C0);
  return x * fmaf(xc, p, 0.5f);
}
// NP pairs of values at once on the packed fp32 VALU ops (v_pk_fma_f32 / v_pk_mul_f32), the NP polynomial chains INTERLEAVED step by step:
// a single chain is a string of dependent packed ops, each waiting out its predecessor's latency (+ one hazard s_nop) — with one wave per
// SIMD nothing else fills those slots and the GELU of a 256 x 192 tile cost 12.6 k cycles instead of ~6 k (profiles/r03_w4_epilogue.md).
// Bit-identical to gelu_poly16 on each element.
typedef __attribute__((ext_vector_type(2))) float stllm_f32x2;
template <int NP>
__device__ __forceinline__ void gelu_poly16_xn(stllm_f32x2 (&v)[NP]) {
  auto c = [](float k) { stllm_f32x2 r = {k, k}; return r; };
  stllm_f32x2 xc[NP], t[NP], q[NP];
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    xc[k][0] = __builtin_amdgcn_fmed3f(v[k][0], -STLLM_GELU_X0, STLLM_GELU_X0);
    xc[k][1] = __builtin_amdgcn_fmed3f(v[k][1], -STLLM_GELU_X0, STLLM_GELU_X0);
  }
#pragma unroll
  for (int k = 0; k < NP; ++k) t[k] = xc[k] * xc[k];
#pragma unroll
  for (int k = 0; k < NP; ++k) q[k] = __builtin_elementwise_fma(c(STLLM_GELU_C8), t[k], c(STLLM_GELU_C7));
  constexpr float kC[7] = {STLLM_GELU_C6, STLLM_GELU_C5, STLLM_GELU_C4, STLLM_GELU_C3, STLLM_GELU_C2, STLLM_GELU_C1, STLLM_GELU_C0};
#pragma unroll
  for (int d = 0; d < 7; ++d) {
#pragma unroll
    for (int k = 0; k < NP; ++k) q[k] = __builtin_elementwise_fma(q[k], t[k], c(kC[d]));
  }
#pragma unroll
  for (int k = 0; k < NP; ++k) q[k] = __builtin_elementwise_fma(xc[k], q[k], c(0.5f));
#pragma unroll
  for (int k = 0; k < NP; ++k) v[k] = v[k] * q[k];
}
__device__ __forceinline__ void gelu_poly16_x2(float& x0, float& x1) {
  stllm_f32x2 v[1] = {{x0, x1}};
  gelu_poly16_xn<1>(v);
  x0 = v[0][0];
  x1 = v[0][1];
}
__device__ __forceinline__ float silu_f(float x) { return x * __frcp_rn(1.0f + __expf(-x)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
