// Frame preprocessing in front of the hot path: uint8 RGB frames -> CLIP-normalised fp32
// [n, 3, 224, 224], bit-identical to the reference's CPU transform chain
//     GroupScale(224, BICUBIC) -> GroupCenterCrop(224) -> Stack -> ToTorchFormatTensor -> GroupNormalize
// (stllm/conversation/conversation.py:190-198, stllm/test/video_transforms.py:54-60, 94-124, 367-407), i.e. torchvision
// 0.15.1's PIL resize (short side -> 224, long side int(224 * long / short)) = Pillow's antialiased bicubic
// (libImaging/Resample.c: per-output-pixel coefficient rows, normalised, quantised to 22-bit fixed point, horizontal
// pass -> uint8 -> vertical pass -> uint8), centre crop with round-half-even offsets, x / 255, (x - mean) / std.
//
// Three launches, nothing allocated, all asynchronous on the caller's stream:
//   1. coeff_kernel   — the two coefficient tables (only the 224 output columns / rows the crop keeps), computed ON THE
//      DEVICE in IEEE double with explicitly rounded operations (__dadd_rn / __dmul_rn / __ddiv_rn: no FMA contraction),
//      in Pillow's operation order, so the quantised int32 coefficients equal the CPU's bit for bit;
//   2. hpass_kernel   — horizontal pass of every input row, cropped columns only: uint8 [n, H, 224, 3];
//   3. vpass_kernel   — vertical pass of the cropped rows + /255 + normalise: fp32 [n, 3, 224, 224].
// HBM-bound byte work: reads n*H*W*3 bytes once, writes n*H*224*3 + n*3*224*224*4 bytes.
#include "common.h"

namespace {

constexpr int kOut = 224;
constexpr int kPrec = 32 - 8 - 2;       // Resample.c PRECISION_BITS
constexpr int kMaxTaps = 128;           // ksize = 2 * ceil(2 * scale) + 1 <= 128  <=>  down-scaling up to ~31x

struct Axis {           // one resampling axis of the crop
  int in_size;          // input extent
  int out_size;         // extent after the resize (before the crop)
  int crop0;            // first kept output index
};

__device__ __forceinline__ double bicubic_rn(double x) {   // Resample.c bicubic_filter, a = -0.5, same operation order
  x = fabs(x);
  if (x < 1.0) {
    const double t = __dadd_rn(__dmul_rn(1.5, x), -2.5);                 // (a + 2) * x - (a + 3)
    return __dadd_rn(__dmul_rn(__dmul_rn(t, x), x), 1.0);               // ... * x * x + 1
  }
  if (x < 2.0) {
    const double u = __dadd_rn(__dmul_rn(__dadd_rn(__dmul_rn(__dadd_rn(x, -5.0), x), 8.0), x), -4.0);   // ((x - 5) * x + 8) * x - 4
    return __dmul_rn(u, -0.5);
  }
  return 0.0;
}

// tables per axis: xmin[224], cnt[224], kk[224][ksize]  (ksize = row pitch, passed in)
__global__ void coeff_kernel(Axis ax_h, Axis ax_v, int ksize_h, int ksize_v, int* __restrict__ tab_h, int* __restrict__ tab_v) {
  const Axis ax = blockIdx.x == 0 ? ax_h : ax_v;
  const int ksize = blockIdx.x == 0 ? ksize_h : ksize_v;
  int* tab = blockIdx.x == 0 ? tab_h : tab_v;
  const int o = threadIdx.x;
  if (o >= kOut) return;
  int* xmin_p = tab;
  int* cnt_p = tab + kOut;
  int* kk = tab + 2 * kOut + o * ksize;
  const int xx = ax.crop0 + o;
  if (ax.in_size == ax.out_size) {   // Pillow skips a pass whose size does not change: identity
    xmin_p[o] = xx;
    cnt_p[o] = 1;
    kk[0] = 1 << kPrec;
    return;
  }
  // precompute_coeffs: scale = (double)(in1 - in0) / outSize with float in0 = 0, in1 = in_size
  const double scale = __ddiv_rn((double)(float)ax.in_size, (double)ax.out_size);
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = __dmul_rn(2.0, filterscale);
  const double ss = __ddiv_rn(1.0, filterscale);
  const double center = __dadd_rn(0.0, __dmul_rn(__dadd_rn((double)xx, 0.5), scale));
  int lo = (int)__dadd_rn(__dadd_rn(center, -support), 0.5);
  if (lo < 0) lo = 0;
  int hi = (int)__dadd_rn(__dadd_rn(center, support), 0.5);
  if (hi > ax.in_size) hi = ax.in_size;
  const int n = hi - lo;
  double w[kMaxTaps];
  double ww = 0.0;
  for (int x = 0; x < n; ++x) {
    const double arg = __dmul_rn(__dadd_rn(__dadd_rn((double)(x + lo), -center), 0.5), ss);   // (x + xmin - center + 0.5) * ss
    w[x] = bicubic_rn(arg);
    ww = __dadd_rn(ww, w[x]);
  }
  for (int x = 0; x < n; ++x) {
    double k = w[x];
    if (ww != 0.0) k = __ddiv_rn(k, ww);
    // normalize_coeffs_8bpc: (int)(+-0.5 + k * (1 << PRECISION_BITS)), truncation toward zero
    const double q = __dmul_rn(k, (double)(1 << kPrec));
    kk[x] = k < 0 ? (int)__dadd_rn(-0.5, q) : (int)__dadd_rn(0.5, q);
  }
  xmin_p[o] = lo;
  cnt_p[o] = n;
}

__device__ __forceinline__ uint8_t clip8(int acc) {
  const int v = acc >> kPrec;   // arithmetic shift, then clamp (Resample.c clip8 lookup)
  return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// tmp[n][y][o][c] = horizontal pass of input row y at cropped output column o
__global__ __launch_bounds__(256) void hpass_kernel(const uint8_t* __restrict__ frames, int64_t frame_stride, int H, int W,
                                                     const int* __restrict__ tab, int ksize, uint8_t* __restrict__ tmp) {
  const int o = threadIdx.x;              // 224 of 256 threads active
  const int y = blockIdx.x, n = blockIdx.y;
  if (o >= kOut) return;
  const int lo = tab[o], cnt = tab[kOut + o];
  const int* kk = tab + 2 * kOut + o * ksize;
  const uint8_t* row = frames + (int64_t)n * frame_stride + ((int64_t)y * W + lo) * 3;
  int s0 = 1 << (kPrec - 1), s1 = s0, s2 = s0;
  for (int i = 0; i < cnt; ++i) {
    const int k = kk[i];
    s0 += (int)row[3 * i + 0] * k;
    s1 += (int)row[3 * i + 1] * k;
    s2 += (int)row[3 * i + 2] * k;
  }
  uint8_t* dst = tmp + (((int64_t)n * H + y) * kOut + o) * 3;
  dst[0] = clip8(s0);
  dst[1] = clip8(s1);
  dst[2] = clip8(s2);
}

// out[n][c][r][o] = ((clip8(vertical pass) / 255) - mean[c]) / std[c]     (fp32, correctly rounded divisions)
__global__ __launch_bounds__(256) void vpass_kernel(const uint8_t* __restrict__ tmp, int H, const int* __restrict__ tab, int ksize,
                                                     float* __restrict__ out) {
  const int o = threadIdx.x;
  const int r = blockIdx.x, n = blockIdx.y;
  if (o >= kOut) return;
  const int lo = tab[r], cnt = tab[kOut + r];
  const int* kk = tab + 2 * kOut + r * ksize;
  const uint8_t* col = tmp + (((int64_t)n * H + lo) * kOut + o) * 3;
  int s0 = 1 << (kPrec - 1), s1 = s0, s2 = s0;
  for (int i = 0; i < cnt; ++i) {
    const int k = kk[i];
    const uint8_t* px = col + (int64_t)i * kOut * 3;
    s0 += (int)px[0] * k;
    s1 += (int)px[1] * k;
    s2 += (int)px[2] * k;
  }
  // conversation.py:190-191
  const float mean[3] = {0.48145466f, 0.4578275f, 0.40821073f};
  const float stdv[3] = {0.26862954f, 0.26130258f, 0.27577711f};
  const int acc[3] = {s0, s1, s2};
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float v = __fdiv_rn((float)clip8(acc[c]), 255.0f);            // ToTorchFormatTensor: .float().div(255)
    v = __fdiv_rn(__fsub_rn(v, mean[c]), stdv[c]);                // GroupNormalize: t.sub_(m).div_(s)
    out[(((int64_t)n * 3 + c) * kOut + r) * kOut + o] = v;
  }
}

// host: torchvision 0.15.1 _compute_resized_output_size + CenterCrop offsets (Python round = half to even)
static int round_half_even_div2(int d) {   // round(d / 2.0)
  if (d % 2 == 0) return d / 2;
  const int f = (d - 1) / 2;               // floor for d > 0
  return (f % 2 == 0) ? f : f + 1;
}
static int ksize_of(int in_size, int out_size) {
  if (in_size == out_size) return 1;
  const double scale = (double)in_size / out_size;
  const double fs = scale < 1.0 ? 1.0 : scale;
  int c = (int)(2.0 * fs);
  if ((double)c < 2.0 * fs) ++c;           // ceil
  return c * 2 + 1;
}
static bool geometry(int H, int W, Axis* ah, Axis* av, int* kh, int* kv) {
  const int shortv = W <= H ? W : H, longv = W <= H ? H : W;
  const int new_long = (int)((double)(kOut * (int64_t)longv) / shortv);   // int(size * long / short): true division, truncation
  const int nw = W <= H ? kOut : new_long, nh = W <= H ? new_long : kOut;
  if (nw < kOut || nh < kOut) return false;
  ah->in_size = W; ah->out_size = nw; ah->crop0 = round_half_even_div2(nw - kOut);
  av->in_size = H; av->out_size = nh; av->crop0 = round_half_even_div2(nh - kOut);
  *kh = ksize_of(W, nw);
  *kv = ksize_of(H, nh);
  return *kh <= kMaxTaps && *kv <= kMaxTaps;
}
static int64_t tab_ints(int ksize) { return 2 * kOut + (int64_t)kOut * ksize; }

}  // namespace

extern "C" int64_t stllm_preprocess_workspace_bytes(int n_frames, int H, int W) {
  Axis ah, av;
  int kh, kv;
  if (n_frames <= 0 || H <= 0 || W <= 0 || !geometry(H, W, &ah, &av, &kh, &kv)) return -1;
  const int64_t tabs = (tab_ints(kh) + tab_ints(kv)) * 4;
  return ((tabs + 255) / 256) * 256 + (int64_t)n_frames * H * kOut * 3;
}

extern "C" int stllm_preprocess_frames(const uint8_t* frames, int64_t frame_stride_bytes, int n_frames, int H, int W, float* out,
                                       void* workspace, int64_t workspace_bytes, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  STLLM_CHECK_ARG(frames && out && workspace, "stllm_preprocess_frames: null pointer");
  STLLM_CHECK_ARG(n_frames > 0 && H > 0 && W > 0 && frame_stride_bytes >= (int64_t)H * W * 3, "stllm_preprocess_frames: bad shape n=%d H=%d W=%d", n_frames, H, W);
  Axis ah, av;
  int kh, kv;
  STLLM_CHECK_ARG(geometry(H, W, &ah, &av, &kh, &kv), "stllm_preprocess_frames: %dx%d frames are outside the supported range (short side -> 224, <= 31x down-scaling)", H, W);
  const int64_t need = stllm_preprocess_workspace_bytes(n_frames, H, W);
  STLLM_CHECK_ARG(workspace_bytes >= need && aligned16(workspace), "stllm_preprocess_frames: workspace too small (%lld < %lld bytes) or misaligned",
                  (long long)workspace_bytes, (long long)need);
  int* tab_h = reinterpret_cast<int*>(workspace);
  int* tab_v = tab_h + tab_ints(kh);
  uint8_t* tmp = reinterpret_cast<uint8_t*>(workspace) + (((tab_ints(kh) + tab_ints(kv)) * 4 + 255) / 256) * 256;
  hipLaunchKernelGGL(coeff_kernel, dim3(2), dim3(256), 0, stream, ah, av, kh, kv, tab_h, tab_v);
  hipLaunchKernelGGL(hpass_kernel, dim3(H, n_frames), dim3(256), 0, stream, frames, frame_stride_bytes, H, W, tab_h, kh, tmp);
  hipLaunchKernelGGL(vpass_kernel, dim3(kOut, n_frames), dim3(256), 0, stream, tmp, H, tab_v, kv, out);
  STLLM_CHECK_LAUNCH("stllm_preprocess_frames");
  return STLLM_OK;
}
