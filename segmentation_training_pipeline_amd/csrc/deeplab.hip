// Kernels that only the in-tree DeepLabV3+ (segmentation_pipeline/impl/deeplab/model.py) needs: depthwise convolution
// (model.py:136, 255-259) forward / data-gradient / weight-gradient with stride, dilation and TF 'same' padding,
// align_corners=True bilinear resize (model.py:94-100) and its gradient, inverted Dropout (model.py:461) with a
// counter-based mask, element-wise sigmoid and its gradient, and the loss on PROBABILITIES (the model applies its
// activation inside the last 1x1 convolution and upsamples the probabilities, model.py:485-486).
// All of them are HBM-bound streaming kernels; reductions are two-stage and fixed-order (deterministic).
#include "common.h"

template <typename T, int V> __device__ __forceinline__ void dl_ldv(const T* p, float (&o)[V]);
template <> __device__ __forceinline__ void dl_ldv<float, 4>(const float* p, float (&o)[4]) {
  const f32x4 v = *reinterpret_cast<const f32x4*>(p);
  o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
}
template <> __device__ __forceinline__ void dl_ldv<bf16_t, 4>(const bf16_t* p, float (&o)[4]) {
  const f32x4 v = load4(p);
  o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
}
template <typename T> __device__ __forceinline__ void dl_stv(T* p, const float (&o)[4]) { store4(p, f32x4{o[0], o[1], o[2], o[3]}); }

struct DwArgs {
  int N, H, W, C, k, stride, pad_t, pad_l, dil, Ho, Wo;
};

// y[n,ho,wo,c] = sum_{kh,kw} x[n, ho*s - pt + kh*d, wo*s - pl + kw*d, c] * w[kh][kw][c]
template <typename T>
__global__ __launch_bounds__(256) void dwconv_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w, T* __restrict__ y, DwArgs a) {
  const int cg = a.C >> 2;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= a.Wo * cg) return;
  const int wo = t / cg, c = (t - wo * cg) * 4;
  const int n = blockIdx.y / a.Ho, ho = blockIdx.y - n * a.Ho;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int kh = 0; kh < a.k; ++kh) {
    const int h = ho * a.stride - a.pad_t + kh * a.dil;
    if ((unsigned)h >= (unsigned)a.H) continue;
    for (int kw = 0; kw < a.k; ++kw) {
      const int ww = wo * a.stride - a.pad_l + kw * a.dil;
      if ((unsigned)ww >= (unsigned)a.W) continue;
      float xv[4];
      dl_ldv<T, 4>(x + (((size_t)n * a.H + h) * a.W + ww) * a.C + c, xv);
      const f32x4 wv = *reinterpret_cast<const f32x4*>(w + (size_t)(kh * a.k + kw) * a.C + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = fmaf(xv[e], wv[e], acc[e]);
    }
  }
  dl_stv<T>(y + (((size_t)n * a.Ho + ho) * a.Wo + wo) * a.C + c, acc);
}

// dx[n,h,w,c] (+)= sum over the taps whose window covers (h,w): ho = (h + pt - kh*d) / s when divisible and in range
template <typename T>
__global__ __launch_bounds__(256) void dwconv_dgrad_kernel(const T* __restrict__ dy, const float* __restrict__ w, T* __restrict__ dx, DwArgs a,
                                                           int accumulate) {
  const int cg = a.C >> 2;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= a.W * cg) return;
  const int wi = t / cg, c = (t - wi * cg) * 4;
  const int n = blockIdx.y / a.H, h = blockIdx.y - n * a.H;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int kh = 0; kh < a.k; ++kh) {
    const int hh = h + a.pad_t - kh * a.dil;
    if (hh < 0 || hh % a.stride) continue;
    const int ho = hh / a.stride;
    if (ho >= a.Ho) continue;
    for (int kw = 0; kw < a.k; ++kw) {
      const int ww = wi + a.pad_l - kw * a.dil;
      if (ww < 0 || ww % a.stride) continue;
      const int wo = ww / a.stride;
      if (wo >= a.Wo) continue;
      float gv[4];
      dl_ldv<T, 4>(dy + (((size_t)n * a.Ho + ho) * a.Wo + wo) * a.C + c, gv);
      const f32x4 wv = *reinterpret_cast<const f32x4*>(w + (size_t)(kh * a.k + kw) * a.C + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = fmaf(gv[e], wv[e], acc[e]);
    }
  }
  T* o = dx + (((size_t)n * a.H + h) * a.W + wi) * a.C + c;
  if (accumulate) {
    float p[4];
    dl_ldv<T, 4>(o, p);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] += p[e];
  }
  dl_stv<T>(o, acc);
}

// dw[kh][kw][c] = sum_{n,ho,wo} dy * x: workgroup b sums output rows b, b+grid, ... for one 64-channel slab (blockIdx.y)
// -> partial[b][tap][C]; a second kernel adds the partials in order.
#define DW_MAX_BLOCKS 128
template <typename T>
__global__ __launch_bounds__(256) void dwconv_wgrad_partial_kernel(const T* __restrict__ x, const T* __restrict__ dy, float* __restrict__ partial,
                                                                   DwArgs a) {
  __shared__ float red[16][64];
  const int cl = threadIdx.x & 63, seg = threadIdx.x >> 6;   // 64 channels x 4 pixel segments
  const int c = blockIdx.y * 64 + cl;
  const int taps = a.k * a.k;
  const int64_t rows = (int64_t)a.N * a.Ho;
  for (int tp = 0; tp < taps; ++tp) {
    const int kh = tp / a.k, kw = tp - kh * a.k;
    float acc = 0.f;
    if (c < a.C) {
      for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
        const int n = (int)(r / a.Ho), ho = (int)(r - (int64_t)n * a.Ho);
        const int h = ho * a.stride - a.pad_t + kh * a.dil;
        if ((unsigned)h >= (unsigned)a.H) continue;
        for (int wo = seg; wo < a.Wo; wo += 4) {
          const int ww = wo * a.stride - a.pad_l + kw * a.dil;
          if ((unsigned)ww >= (unsigned)a.W) continue;
          acc = fmaf(Elem<T>::load(dy + (((size_t)n * a.Ho + ho) * a.Wo + wo) * a.C + c),
                     Elem<T>::load(x + (((size_t)n * a.H + h) * a.W + ww) * a.C + c), acc);
        }
      }
    }
    red[seg][cl] = acc;
    __syncthreads();
    if (seg == 0 && c < a.C) partial[((size_t)blockIdx.x * taps + tp) * a.C + c] = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
    __syncthreads();
  }
}
__global__ __launch_bounds__(256) void dwconv_wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw, int blocks, int count,
                                                                  int accumulate) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= count) return;
  float s = 0.f;
  for (int b = 0; b < blocks; ++b) s += partial[(size_t)b * count + i];
  dw[i] = accumulate ? dw[i] + s : s;
}

static bool dw_fill(DwArgs& a, int N, int H, int W, int C, int k, int stride, int pad_t, int pad_l, int dil, int Ho, int Wo) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3) || k < 1 || stride < 1 || dil < 1 || Ho <= 0 || Wo <= 0) return false;
  a = DwArgs{N, H, W, C, k, stride, pad_t, pad_l, dil, Ho, Wo};
  return true;
}

extern "C" int stp_dwconv(const void* x, const float* w, void* y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t k, int32_t stride,
                          int32_t pad_t, int32_t pad_l, int32_t dilation, int32_t Ho, int32_t Wo, int32_t dtype, void* stream) {
  if (!stp_dtype_ok(dtype)) return STP_E_BADARG;      // (the other build's 16-bit code, or garbage)
  DwArgs a;
  if (!x || !w || !y || !dw_fill(a, N, H, W, C, k, stride, pad_t, pad_l, dilation, Ho, Wo) || (int64_t)N * Ho > 65535) return STP_E_BADARG;
  const dim3 grid(ceil_div(Wo * (C >> 2), 256), N * Ho);
  if (dtype == STP_H16) hipLaunchKernelGGL(dwconv_fwd_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, w, (bf16_t*)y, a);
  else if (dtype == STP_F32) hipLaunchKernelGGL(dwconv_fwd_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, w, (float*)y, a);
  else return STP_E_BADARG;
  STP_LAUNCH_CHECK();
  return STP_OK;
}

extern "C" int stp_dwconv_dgrad(const void* dy, const float* w, void* dx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t k,
                                int32_t stride, int32_t pad_t, int32_t pad_l, int32_t dilation, int32_t Ho, int32_t Wo, int32_t dtype,
                                int32_t accumulate, void* stream) {
  if (!stp_dtype_ok(dtype)) return STP_E_BADARG;      // (the other build's 16-bit code, or garbage)
  DwArgs a;
  if (!dy || !w || !dx || !dw_fill(a, N, H, W, C, k, stride, pad_t, pad_l, dilation, Ho, Wo) || (int64_t)N * H > 65535) return STP_E_BADARG;
  const dim3 grid(ceil_div(W * (C >> 2), 256), N * H);
  if (dtype == STP_H16) hipLaunchKernelGGL(dwconv_dgrad_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, w, (bf16_t*)dx, a, accumulate);
  else if (dtype == STP_F32) hipLaunchKernelGGL(dwconv_dgrad_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)dy, w, (float*)dx, a, accumulate);
  else return STP_E_BADARG;
  STP_LAUNCH_CHECK();
  return STP_OK;
}

extern "C" size_t stp_dwconv_wgrad_workspace_bytes(int32_t C, int32_t k) { return (size_t)DW_MAX_BLOCKS * k * k * C * sizeof(float); }

extern "C" int stp_dwconv_wgrad(const void* x, const void* dy, float* dw, int32_t N, int32_t H, int32_t W, int32_t C, int32_t k,
                                int32_t stride, int32_t pad_t, int32_t pad_l, int32_t dilation, int32_t Ho, int32_t Wo, int32_t dtype,
                                int32_t accumulate, void* workspace, size_t workspace_bytes, void* stream) {
  if (!stp_dtype_ok(dtype)) return STP_E_BADARG;      // (the other build's 16-bit code, or garbage)
  DwArgs a;
  if (!x || !dy || !dw || !workspace || !dw_fill(a, N, H, W, C, k, stride, pad_t, pad_l, dilation, Ho, Wo)) return STP_E_BADARG;
  if (workspace_bytes < stp_dwconv_wgrad_workspace_bytes(C, k)) return STP_E_WORKSPACE;
  int blocks = N * Ho < DW_MAX_BLOCKS ? N * Ho : DW_MAX_BLOCKS;
  const dim3 grid(blocks, ceil_div(C, 64));
  hipStream_t s = (hipStream_t)stream;
  if (dtype == STP_H16) hipLaunchKernelGGL(dwconv_wgrad_partial_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)dy, (float*)workspace, a);
  else if (dtype == STP_F32) hipLaunchKernelGGL(dwconv_wgrad_partial_kernel<float>, grid, dim3(256), 0, s, (const float*)x, (const float*)dy, (float*)workspace, a);
  else return STP_E_BADARG;
  STP_LAUNCH_CHECK();
  const int count = k * k * C;
  hipLaunchKernelGGL(dwconv_wgrad_reduce_kernel, dim3(ceil_div(count, 256)), dim3(256), 0, s, (const float*)workspace, dw, blocks, count, accumulate);
  STP_LAUNCH_CHECK();
  return STP_OK;
}

// ------------------------------------------------------------------------------------------
// tf.image.resize_bilinear(align_corners=True) to an arbitrary output size (BilinearUpsampling, model.py:94-100):
// src = dst * (in - 1) / (out - 1) (0 when out == 1).  C contiguous channels per pixel, any C.
__device__ __forceinline__ float ac_scale(int in, int out) { return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f; }

template <typename T>
__global__ __launch_bounds__(256) void resize_ac_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int C, int Ho, int Wo) {
  const int64_t total = (int64_t)N * Ho * Wo * C;
  const float sy = ac_scale(H, Ho), sx = ac_scale(W, Wo);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    const int xo = (int)((i / C) % Wo);
    const int yo = (int)((i / ((int64_t)C * Wo)) % Ho);
    const int n = (int)(i / ((int64_t)C * Wo * Ho));
    const float fy_ = (float)yo * sy, fx_ = (float)xo * sx;
    const int y0 = (int)floorf(fy_), x0 = (int)floorf(fx_);
    const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
    const float fy = fy_ - (float)y0, fx = fx_ - (float)x0;
    const T* b = x + (int64_t)n * H * W * C + c;
    const float v00 = Elem<T>::load(b + ((int64_t)y0 * W + x0) * C), v01 = Elem<T>::load(b + ((int64_t)y0 * W + x1) * C);
    const float v10 = Elem<T>::load(b + ((int64_t)y1 * W + x0) * C), v11 = Elem<T>::load(b + ((int64_t)y1 * W + x1) * C);
    const float top = v00 + (v01 - v00) * fx, bot = v10 + (v11 - v10) * fx;
    Elem<T>::store(y + i, top + (bot - top) * fy);
  }
}

// gradient: every input pixel gathers, in a fixed order, the outputs whose two taps along each axis include it
template <typename T>
__global__ __launch_bounds__(256) void resize_ac_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int N, int H, int W, int C, int Ho, int Wo,
                                                            int accumulate) {
  const int64_t total = (int64_t)N * H * W * C;
  const float sy = ac_scale(H, Ho), sx = ac_scale(W, Wo);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    const int w = (int)((i / C) % W);
    const int h = (int)((i / ((int64_t)C * W)) % H);
    const int n = (int)(i / ((int64_t)C * W * H));
    // candidate outputs: src in (h-1, h+1)  ->  yo in [(h-1)/sy, (h+1)/sy]; everything when the scale is 0 (in == 1)
    const int ylo = sy > 0.f ? max((int)floorf((float)(h - 1) / sy), 0) : 0, yhi = sy > 0.f ? min((int)ceilf((float)(h + 1) / sy), Ho - 1) : Ho - 1;
    const int xlo = sx > 0.f ? max((int)floorf((float)(w - 1) / sx), 0) : 0, xhi = sx > 0.f ? min((int)ceilf((float)(w + 1) / sx), Wo - 1) : Wo - 1;
    const T* b = dy + (int64_t)n * Ho * Wo * C + c;
    float acc = 0.f;
    for (int yo = ylo; yo <= yhi; ++yo) {
      const float fy_ = (float)yo * sy;
      const int y0 = (int)floorf(fy_), y1 = min(y0 + 1, H - 1);
      const float fy = fy_ - (float)y0;
      float wy = 0.f;
      if (y0 == h) wy += 1.f - fy;
      if (y1 == h) wy += fy;
      if (wy == 0.f) continue;
      for (int xo = xlo; xo <= xhi; ++xo) {
        const float fx_ = (float)xo * sx;
        const int x0 = (int)floorf(fx_), x1 = min(x0 + 1, W - 1);
        const float fx = fx_ - (float)x0;
        float wx = 0.f;
        if (x0 == w) wx += 1.f - fx;
        if (x1 == w) wx += fx;
        if (wx != 0.f) acc += wy * wx * Elem<T>::load(b + ((int64_t)yo * Wo + xo) * C);
      }
    }
    if (accumulate) acc += Elem<T>::load(dx + i);
    Elem<T>::store(dx + i, acc);
  }
}

static int dl_grid(int64_t items) {
  int64_t g = (items + 255) / 256;
  return (int)(g > 16384 ? 16384 : (g < 1 ? 1 : g));
}

extern "C" int stp_resize_bilinear_ac(const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t Ho, int32_t Wo,
                                      int32_t dtype, void* stream) {
  if (!stp_dtype_ok(dtype)) return STP_E_BADARG;      // (the other build's 16-bit code, or garbage)
  if (!x || !y || N <= 0 || H <= 0 || W <= 0 || C <= 0 || Ho <= 0 || Wo <= 0) return STP_E_BADARG;
  const int g = dl_grid((int64_t)N * Ho * Wo * C);
  if (dtype == STP_H16) hipLaunchKernelGGL(resize_ac_kernel<bf16_t>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, N, H, W, C, Ho, Wo);
  else if (dtype == STP_F32) hipLaunchKernelGGL(resize_ac_kernel<float>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)y, N, H, W, C, Ho, Wo);
  else return STP_E_BADARG;
  STP_LAUNCH_CHECK();
  return STP_OK;
}

extern "C" int stp_resize_bilinear_ac_bwd(const void* dy, void* dx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t Ho, int32_t Wo,
                                          int32_t dtype, int32_t accumulate, void* stream) {
  if (!stp_dtype_ok(dtype)) return STP_E_BADARG;      // (the other build's 16-bit code, or garbage)
  if (!dy || !dx || N <= 0 || H <= 0 || W <= 0 || C <= 0 || Ho <= 0 || Wo <= 0) return STP_E_BADARG;
  const int g = dl_grid((int64_t)N * H * W * C);
  if (dtype == STP_H16) hipLaunchKernelGGL(resize_ac_bwd_kernel<bf16_t>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, (bf16_t*)dx, N, H, W, C, Ho, Wo, accumulate);
  else if (dtype == STP_F32) hipLaunchKernelGGL(resize_ac_bwd_kernel<float>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const float*)dy, (float*)dx, N, H, W, C, Ho, Wo, accumulate);
  else return STP_E_BADARG;
  STP_LAUNCH_CHECK();
  return STP_OK;
}

// ------------------------------------------------------------------------------------------
// Inverted dropout (keras Dropout(rate) in the training phase): keep with probability 1 - rate and scale by 1/(1 - rate).
// The mask is a counter-based hash of (state[0] = step counter, salt, element index): the backward recomputes it, a
// hipGraph replay draws a fresh mask every step (stp_counter_tick), and the numpy oracle reproduces it exactly.
__device__ __forceinline__ uint32_t drop_hash(uint32_t seed, uint32_t i) {
  uint32_t h = seed ^ (i * 0x9E3779B1u);
  h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
  return h;
}
__global__ void counter_tick_kernel(int32_t* state) {
  if (threadIdx.x == 0 && blockIdx.x == 0) state[0] += 1;
}
template <typename T>
__global__ __launch_bounds__(256) void dropout_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t count, uint32_t thresh, float scale,
                                                      const int32_t* state, uint32_t salt) {
  const uint32_t seed = (uint32_t)state[0] * 0x85EBCA77u + salt;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) {
    const bool keep = (drop_hash(seed, (uint32_t)i) >> 8) >= thresh;
    Elem<T>::store(y + i, keep ? Elem<T>::load(x + i) * scale : 0.f);
  }
}

extern "C" int stp_counter_tick(int32_t* state, void* stream) {
  if (!state) return STP_E_BADARG;
  hipLaunchKernelGGL(counter_tick_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state);
  STP_LAUNCH_CHECK();
  return STP_OK;
}

// SpatialDropout2D (noise_shape (batch, 1, 1, channels): a whole feature map of a sample is kept or dropped): the mask index is
// (sample, channel) instead of the element
template <typename T>
__global__ __launch_bounds__(256) void dropout_spatial_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t count, int64_t hwc, int C,
                                                              uint32_t thresh, float scale, const int32_t* state, uint32_t salt) {
  const uint32_t seed = (uint32_t)state[0] * 0x85EBCA77u + salt;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) {
    const int64_t n = i / hwc;
    const uint32_t m = (uint32_t)(n * C + (i % C));
    const bool keep = (drop_hash(seed, m) >> 8) >= thresh;
    Elem<T>::store(y + i, keep ? Elem<T>::load(x + i) * scale : 0.f);
  }
}

extern "C" int stp_dropout_spatial(const void* x, void* y, int32_t N, int64_t HW, int32_t C, float rate, const int32_t* state, uint32_t salt,
                                   int32_t dtype, void* stream) {
  if (!stp_dtype_ok(dtype)) return STP_E_BADARG;      // (the other build's 16-bit code, or garbage)
  if (!x || !y || !state || N <= 0 || HW <= 0 || C <= 0 || rate < 0.f || rate >= 1.f) return STP_E_BADARG;
  const uint32_t thresh = (uint32_t)lrintf(rate * 16777216.f);
  const float scale = 1.f / (1.f - rate);
  const int64_t count = (int64_t)N * HW * C;
  const int g = dl_grid(count);
  if (dtype == STP_H16) hipLaunchKernelGGL(dropout_spatial_kernel<bf16_t>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, count, HW * C, C, thresh, scale, state, salt);
  else if (dtype == STP_F32) hipLaunchKernelGGL(dropout_spatial_kernel<float>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)y, count, HW * C, C, thresh, scale, state, salt);
  else return STP_E_BADARG;
  STP_LAUNCH_CHECK();
  return STP_OK;
}

// forward and backward are the same map (y = mask * x / (1 - rate)); x == y (in place) is allowed
extern "C" int stp_dropout(const void* x, void* y, int64_t count, float rate, const int32_t* state, uint32_t salt, int32_t dtype, void* stream) {
  if (!stp_dtype_ok(dtype)) return STP_E_BADARG;      // (the other build's 16-bit code, or garbage)
  if (!x || !y || !state || count <= 0 || rate < 0.f || rate >= 1.f) return STP_E_BADARG;
  const uint32_t thresh = (uint32_t)lrintf(rate * 16777216.f);
  const float scale = 1.f / (1.f - rate);
  const int g = dl_grid(count);
  if (dtype == STP_H16) hipLaunchKernelGGL(dropout_kernel<bf16_t>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, count, thresh, scale, state, salt);
  else if (dtype == STP_F32) hipLaunchKernelGGL(dropout_kernel<float>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)y, count, thresh, scale, state, salt);
  else return STP_E_BADARG;
  STP_LAUNCH_CHECK();
  return STP_OK;
}

// ------------------------------------------------------------------------------------------
// Activation('sigmoid') as a tensor op (the model's last convolution carries it, model.py:485) and its gradient
// dz = dp * p * (1 - p), both over [rows][ld] tensors of which the first `channels` columns are used.
template <typename T>
__global__ __launch_bounds__(256) void sigmoid_act_kernel(const T* __restrict__ z, T* __restrict__ p, int64_t rows, int channels, int ldz, int ldp) {
  const int64_t total = rows * channels;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / channels;
    const int c = (int)(i - r * channels);
    Elem<T>::store(p + r * ldp + c, 1.f / (1.f + expf(-Elem<T>::load(z + r * ldz + c))));
  }
}
template <typename T>
__global__ __launch_bounds__(256) void sigmoid_act_bwd_kernel(const T* __restrict__ p, const T* __restrict__ dp, T* __restrict__ dz, int64_t rows,
                                                              int channels, int ldp, int ldg) {
  const int64_t total = rows * ldg;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / ldg;
    const int c = (int)(i - r * ldg);
    float g = 0.f;
    if (c < channels) {
      const float pv = Elem<T>::load(p + r * ldp + c);
      g = Elem<T>::load(dp + i) * pv * (1.f - pv);
    }
    Elem<T>::store(dz + i, g);
  }
}

extern "C" int stp_sigmoid_act(const void* z, void* p, int64_t rows, int32_t channels, int32_t ldz, int32_t ldp, int32_t dtype, void* stream) {
  if (!stp_dtype_ok(dtype)) return STP_E_BADARG;      // (the other build's 16-bit code, or garbage)
  if (!z || !p || rows <= 0 || channels <= 0 || ldz < channels || ldp < channels) return STP_E_BADARG;
  const int g = dl_grid(rows * channels);
  if (dtype == STP_H16) hipLaunchKernelGGL(sigmoid_act_kernel<bf16_t>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)z, (bf16_t*)p, rows, channels, ldz, ldp);
  else if (dtype == STP_F32) hipLaunchKernelGGL(sigmoid_act_kernel<float>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const float*)z, (float*)p, rows, channels, ldz, ldp);
  else return STP_E_BADARG;
  STP_LAUNCH_CHECK();
  return STP_OK;
}

// dp and dz are [rows][ldg] (padded gradient channels; columns >= channels of dz are written as 0); dp == dz allowed
extern "C" int stp_sigmoid_act_bwd(const void* p, const void* dp, void* dz, int64_t rows, int32_t channels, int32_t ldp, int32_t ldg,
                                   int32_t dtype, void* stream) {
  if (!stp_dtype_ok(dtype)) return STP_E_BADARG;      // (the other build's 16-bit code, or garbage)
  if (!p || !dp || !dz || rows <= 0 || channels <= 0 || ldp < channels || ldg < channels) return STP_E_BADARG;
  const int g = dl_grid(rows * ldg);
  if (dtype == STP_H16) hipLaunchKernelGGL(sigmoid_act_bwd_kernel<bf16_t>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)p, (const bf16_t*)dp, (bf16_t*)dz, rows, channels, ldp, ldg);
  else if (dtype == STP_F32) hipLaunchKernelGGL(sigmoid_act_bwd_kernel<float>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const float*)p, (const float*)dp, (float*)dz, rows, channels, ldp, ldg);
  else return STP_E_BADARG;
  STP_LAUNCH_CHECK();
  return STP_OK;
}

// ------------------------------------------------------------------------------------------
// Activation('softmax') as a tensor op for the multi-class head (the class convolution carries the activation, model.py:485;
// 2..32 classes) and its gradient dz_c = p_c (dp_c - sum_k p_k dp_k).  One thread per pixel row.
template <typename T>
__global__ __launch_bounds__(256) void softmax_act_kernel(const T* __restrict__ z, T* __restrict__ p, int64_t rows, int classes, int ldz, int ldp) {
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < rows; r += (int64_t)gridDim.x * 256) {
    const T* zr = z + r * ldz;
    float m = -3.4e38f;
    for (int c = 0; c < classes; ++c) m = fmaxf(m, Elem<T>::load(zr + c));
    float sum = 0.f;
    for (int c = 0; c < classes; ++c) sum += expf(Elem<T>::load(zr + c) - m);
    const float inv = 1.f / sum;
    for (int c = 0; c < classes; ++c) Elem<T>::store(p + r * ldp + c, expf(Elem<T>::load(zr + c) - m) * inv);
  }
}
template <typename T>
__global__ __launch_bounds__(256) void softmax_act_bwd_kernel(const T* __restrict__ p, const T* __restrict__ dp, T* __restrict__ dz, int64_t rows,
                                                              int classes, int ldp, int ldg) {
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < rows; r += (int64_t)gridDim.x * 256) {
    const T* pr = p + r * ldp;
    const T* gr = dp + r * ldg;
    float dot = 0.f;
    for (int c = 0; c < classes; ++c) dot += Elem<T>::load(pr + c) * Elem<T>::load(gr + c);
    for (int c = 0; c < ldg; ++c)
      Elem<T>::store(dz + r * ldg + c, c < classes ? Elem<T>::load(pr + c) * (Elem<T>::load(gr + c) - dot) : 0.f);
  }
}

extern "C" int stp_softmax_act(const void* z, void* p, int64_t rows, int32_t classes, int32_t ldz, int32_t ldp, int32_t dtype, void* stream) {
  if (!stp_dtype_ok(dtype)) return STP_E_BADARG;      // (the other build's 16-bit code, or garbage)
  if (!z || !p || rows <= 0 || classes < 2 || classes > 32 || ldz < classes || ldp < classes) return STP_E_BADARG;
  const int g = dl_grid(rows);
  if (dtype == STP_H16) hipLaunchKernelGGL(softmax_act_kernel<bf16_t>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)z, (bf16_t*)p, rows, classes, ldz, ldp);
  else if (dtype == STP_F32) hipLaunchKernelGGL(softmax_act_kernel<float>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const float*)z, (float*)p, rows, classes, ldz, ldp);
  else return STP_E_BADARG;
  STP_LAUNCH_CHECK();
  return STP_OK;
}

// dp and dz are [rows][ldg] (columns >= classes of dz are written as 0); dp == dz allowed (a row is read before it is written)
extern "C" int stp_softmax_act_bwd(const void* p, const void* dp, void* dz, int64_t rows, int32_t classes, int32_t ldp, int32_t ldg,
                                   int32_t dtype, void* stream) {
  if (!stp_dtype_ok(dtype)) return STP_E_BADARG;      // (the other build's 16-bit code, or garbage)
  if (!p || !dp || !dz || rows <= 0 || classes < 2 || classes > 32 || ldp < classes || ldg < classes) return STP_E_BADARG;
  const int g = dl_grid(rows);
  if (dtype == STP_H16) hipLaunchKernelGGL(softmax_act_bwd_kernel<bf16_t>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)p, (const bf16_t*)dp, (bf16_t*)dz, rows, classes, ldp, ldg);
  else if (dtype == STP_F32) hipLaunchKernelGGL(softmax_act_bwd_kernel<float>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const float*)p, (const float*)dp, (float*)dz, rows, classes, ldp, ldg);
  else return STP_E_BADARG;
  STP_LAUNCH_CHECK();
  return STP_OK;
}

// ------------------------------------------------------------------------------------------
// w_bce * binary_crossentropy + w_dice * dice_loss on PROBABILITIES (1 class), scalars as stp_sigmoid_bce_dice, and the
// gradient w.r.t. the probabilities into column 0 of dprobs [count][dl_channels]:
//   d bce / d p = (p - y) / (p (1 - p)) / count inside the Keras clip [1e-7, 1 - 1e-7], 0 outside;  d dice_loss / d p = -(2 y den - num) / den^2
#define PL_MAX_BLOCKS 1024
#define PL_NSUM 8
template <typename T>
__global__ __launch_bounds__(256) void prob_loss_partial_kernel(const T* __restrict__ probs, const uint8_t* __restrict__ target, int64_t count,
                                                                float* partial) {
  float a[PL_NSUM] = {0, 0, 0, 0, 0, 0, 0, 0};
  const int64_t per = (count + gridDim.x - 1) / gridDim.x;
  const int64_t i0 = (int64_t)blockIdx.x * per, i1 = i0 + per < count ? i0 + per : count;
  for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) {
    const float p = Elem<T>::load(probs + i);
    const float y = target[i] ? 1.f : 0.f;
    const float pc = fminf(fmaxf(p, 1e-7f), 1.f - 1e-7f);
    const float z = logf(pc / (1.f - pc));
    a[0] += fmaxf(z, 0.f) - z * y + log1pf(expf(-fabsf(z)));
    a[1] += p;
    a[2] += y;
    a[3] += p * y;
    const float t = p > 0.5f ? 1.f : 0.f;
    a[4] += t;
    a[5] += t * y;
    a[6] += (t == y) ? 1.f : 0.f;
  }
  __shared__ float red[4][PL_NSUM];
#pragma unroll
  for (int e = 0; e < PL_NSUM; ++e) a[e] = wave_sum(a[e]);
  if ((threadIdx.x & 63) == 0)
#pragma unroll
    for (int e = 0; e < PL_NSUM; ++e) red[threadIdx.x >> 6][e] = a[e];
  __syncthreads();
  if (threadIdx.x < PL_NSUM)
    partial[(size_t)blockIdx.x * PL_NSUM + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}
__global__ __launch_bounds__(256) void prob_loss_finalize_kernel(const float* partial, int blocks, double inv_count, float w_bce, float w_dice,
                                                                 float* scalars) {
  __shared__ double sh[32][PL_NSUM];
  const int e = threadIdx.x & 7, lane = threadIdx.x >> 3;
  double a = 0.0;
  for (int b = lane; b < blocks; b += 32) a += (double)partial[(size_t)b * PL_NSUM + e];
  sh[lane][e] = a;
  __syncthreads();
  for (int w = 16; w > 0; w >>= 1) {
    if (lane < w) sh[lane][e] += sh[lane + w][e];
    __syncthreads();
  }
  if (threadIdx.x != 0) return;
  const double* s = sh[0];
  const double bce = s[0] * inv_count;
  const double dice_l = 1.0 - (2.0 * s[3] + 1.0) / (s[2] + s[1] + 1.0);
  scalars[0] = (float)(w_bce * bce + w_dice * dice_l);
  scalars[1] = (float)bce;
  scalars[2] = (float)dice_l;
  scalars[3] = (float)((2.0 * s[5] + 1.0) / (s[2] + s[4] + 1.0));
  scalars[4] = (float)(s[6] * inv_count);
  scalars[5] = (float)s[1];
  scalars[6] = (float)s[2];
  scalars[7] = (float)s[3];
  scalars[8] = (float)((s[3] + 1.0) / (s[2] + s[1] - s[3] + 1.0));
  scalars[9] = (float)((s[5] + 1.0) / (s[2] + s[4] - s[5] + 1.0));
}
template <typename T>
__global__ __launch_bounds__(256) void prob_loss_grad_kernel(const T* __restrict__ probs, const uint8_t* __restrict__ target, int64_t count,
                                                             const float* scalars, float w_bce, float w_dice, float inv_count, T* __restrict__ dp,
                                                             int dlc) {
  const float sp = scalars[5], sy = scalars[6], spy = scalars[7];
  const float den = sy + sp + 1.f, inv_den2 = 1.f / (den * den), num = 2.f * spy + 1.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) {
    const float p = Elem<T>::load(probs + i);
    const float y = target[i] ? 1.f : 0.f;
    const bool inr = p >= 1e-7f && p <= 1.f - 1e-7f;
    float g = inr ? w_bce * (p - y) / (p * (1.f - p)) * inv_count : 0.f;
    g += w_dice * (-(2.f * y * den - num) * inv_den2);
    T* o = dp + i * dlc;
    Elem<T>::store(o, g);
    for (int c = 1; c < dlc; ++c) Elem<T>::store(o + c, 0.f);
  }
}

extern "C" int stp_prob_bce_dice(const void* probs, const uint8_t* target, int64_t count, int32_t dtype, float w_bce, float w_dice,
                                 float* scalars, void* dprobs, int32_t dl_channels, void* workspace, size_t workspace_bytes, void* stream) {
  if (!stp_dtype_ok(dtype)) return STP_E_BADARG;      // (the other build's 16-bit code, or garbage)
  if (!probs || !target || !scalars || !workspace || count <= 0) return STP_E_BADARG;
  if (workspace_bytes < (size_t)PL_MAX_BLOCKS * PL_NSUM * sizeof(float)) return STP_E_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  int64_t b = count / 1024;
  if (b < 1) b = 1;
  if (b > PL_MAX_BLOCKS) b = PL_MAX_BLOCKS;
  const int blocks = (int)b;
  float* partial = (float*)workspace;
  if (dtype == STP_H16) hipLaunchKernelGGL(prob_loss_partial_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, (const bf16_t*)probs, target, count, partial);
  else if (dtype == STP_F32) hipLaunchKernelGGL(prob_loss_partial_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)probs, target, count, partial);
  else return STP_E_BADARG;
  STP_LAUNCH_CHECK();
  hipLaunchKernelGGL(prob_loss_finalize_kernel, dim3(1), dim3(256), 0, s, partial, blocks, 1.0 / (double)count, w_bce, w_dice, scalars);
  STP_LAUNCH_CHECK();
  if (dprobs) {
    if (dl_channels < 1) return STP_E_BADARG;
    const int g = dl_grid(count);
    const float inv_count = (float)(1.0 / (double)count);
    if (dtype == STP_H16) hipLaunchKernelGGL(prob_loss_grad_kernel<bf16_t>, dim3(g), dim3(256), 0, s, (const bf16_t*)probs, target, count, scalars, w_bce, w_dice, inv_count, (bf16_t*)dprobs, dl_channels);
    else hipLaunchKernelGGL(prob_loss_grad_kernel<float>, dim3(g), dim3(256), 0, s, (const float*)probs, target, count, scalars, w_bce, w_dice, inv_count, (float*)dprobs, dl_channels);
    STP_LAUNCH_CHECK();
  }
  return STP_OK;
}

// ------------------------------------------------------------------------------------------
// Multi-class loss on PROBABILITIES (the model resizes the softmax output, model.py:485-486): Keras categorical_crossentropy
// (p <- p / sum p, clip to [1e-7, 1 - 1e-7], -log p_target) + w_dice * musket dice over every (pixel, class) element of the
// one-hot target.  probs [pixels][ldc], target = class index per pixel; scalars as stp_softmax_cce_dice; the gradient
// w.r.t. the probabilities goes to dprobs [pixels][dl_channels] (zero padding).  One thread per pixel.
//   d cce / d p_k = -([k == t] / p_t - 1 / S) / pixels  where the clip is inactive on q_t = p_t / S, else 0
template <typename T>
__global__ __launch_bounds__(256) void prob_cce_partial_kernel(const T* __restrict__ probs, const uint8_t* __restrict__ target, int64_t pixels,
                                                               int classes, int ldc, float* partial) {
  float a[PL_NSUM] = {0, 0, 0, 0, 0, 0, 0, 0};
  const int64_t per = (pixels + gridDim.x - 1) / gridDim.x;
  const int64_t i0 = (int64_t)blockIdx.x * per, i1 = i0 + per < pixels ? i0 + per : pixels;
  for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) {
    const T* pr = probs + i * ldc;
    const int t = target[i];
    float S = 0.f;
    for (int c = 0; c < classes; ++c) S += Elem<T>::load(pr + c);
    for (int c = 0; c < classes; ++c) {
      const float p = Elem<T>::load(pr + c), y = c == t ? 1.f : 0.f;
      a[1] += p;
      a[3] += p * y;
      const float th = p > 0.5f ? 1.f : 0.f;
      a[4] += th;
      a[5] += th * y;
      a[6] += (th == y) ? 1.f : 0.f;
      if (c == t) a[0] += -logf(fminf(fmaxf(p / S, 1e-7f), 1.f - 1e-7f));
    }
    a[2] += (t < classes) ? 1.f : 0.f;
  }
  __shared__ float red[4][PL_NSUM];
#pragma unroll
  for (int e = 0; e < PL_NSUM; ++e) a[e] = wave_sum(a[e]);
  if ((threadIdx.x & 63) == 0)
#pragma unroll
    for (int e = 0; e < PL_NSUM; ++e) red[threadIdx.x >> 6][e] = a[e];
  __syncthreads();
  if (threadIdx.x < PL_NSUM)
    partial[(size_t)blockIdx.x * PL_NSUM + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}
// sum 0 is per PIXEL, sum 6 per (pixel, class) element
__global__ __launch_bounds__(256) void prob_cce_finalize_kernel(const float* partial, int blocks, double inv_pixels, double inv_elems, float w_cce,
                                                                float w_dice, float* scalars) {
  __shared__ double sh[32][PL_NSUM];
  const int e = threadIdx.x & 7, lane = threadIdx.x >> 3;
  double a = 0.0;
  for (int b = lane; b < blocks; b += 32) a += (double)partial[(size_t)b * PL_NSUM + e];
  sh[lane][e] = a;
  __syncthreads();
  for (int w = 16; w > 0; w >>= 1) {
    if (lane < w) sh[lane][e] += sh[lane + w][e];
    __syncthreads();
  }
  if (threadIdx.x != 0) return;
  const double* s = sh[0];
  const double cce = s[0] * inv_pixels;
  const double dice_l = 1.0 - (2.0 * s[3] + 1.0) / (s[2] + s[1] + 1.0);
  scalars[0] = (float)(w_cce * cce + w_dice * dice_l);
  scalars[1] = (float)cce;
  scalars[2] = (float)dice_l;
  scalars[3] = (float)((2.0 * s[5] + 1.0) / (s[2] + s[4] + 1.0));
  scalars[4] = (float)(s[6] * inv_elems);
  scalars[5] = (float)s[1];
  scalars[6] = (float)s[2];
  scalars[7] = (float)s[3];
  scalars[8] = (float)((s[3] + 1.0) / (s[2] + s[1] - s[3] + 1.0));
  scalars[9] = (float)((s[5] + 1.0) / (s[2] + s[4] - s[5] + 1.0));
}
template <typename T>
__global__ __launch_bounds__(256) void prob_cce_grad_kernel(const T* __restrict__ probs, const uint8_t* __restrict__ target, int64_t pixels, int classes,
                                                            int ldc, const float* scalars, float w_cce, float w_dice, float inv_pixels,
                                                            T* __restrict__ dp, int dlc) {
  const float sp = scalars[5], sy = scalars[6], spy = scalars[7];
  const float den = sy + sp + 1.f, inv_den2 = 1.f / (den * den), num = 2.f * spy + 1.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < pixels; i += (int64_t)gridDim.x * 256) {
    const T* pr = probs + i * ldc;
    const int t = target[i];
    float S = 0.f;
    for (int c = 0; c < classes; ++c) S += Elem<T>::load(pr + c);
    const float pt = t < classes ? Elem<T>::load(pr + t) : 0.f;
    const float q = pt / S;
    const bool inr = t < classes && q >= 1e-7f && q <= 1.f - 1e-7f;
    T* o = dp + i * dlc;
    for (int c = 0; c < dlc; ++c) {
      float g = 0.f;
      if (c < classes) {
        const float y = c == t ? 1.f : 0.f;
        if (inr) g = -w_cce * (y / pt - 1.f / S) * inv_pixels;
        g += w_dice * (-(2.f * y * den - num) * inv_den2);
      }
      Elem<T>::store(o + c, g);
    }
  }
}

extern "C" int stp_prob_cce_dice(const void* probs, const uint8_t* target, int64_t pixels, int32_t classes, int32_t ldc, int32_t dtype, float w_cce,
                                 float w_dice, float* scalars, void* dprobs, int32_t dl_channels, void* workspace, size_t workspace_bytes,
                                 void* stream) {
  if (!stp_dtype_ok(dtype)) return STP_E_BADARG;      // (the other build's 16-bit code, or garbage)
  if (!probs || !target || !scalars || !workspace || pixels <= 0 || classes < 2 || classes > 32 || ldc < classes) return STP_E_BADARG;
  if (workspace_bytes < (size_t)PL_MAX_BLOCKS * PL_NSUM * sizeof(float)) return STP_E_WORKSPACE;
  if (dtype != STP_H16 && dtype != STP_F32) return STP_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  int64_t b = pixels / 512;
  if (b < 1) b = 1;
  if (b > PL_MAX_BLOCKS) b = PL_MAX_BLOCKS;
  const int blocks = (int)b;
  float* partial = (float*)workspace;
  if (dtype == STP_H16) hipLaunchKernelGGL(prob_cce_partial_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, (const bf16_t*)probs, target, pixels, classes, ldc, partial);
  else hipLaunchKernelGGL(prob_cce_partial_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)probs, target, pixels, classes, ldc, partial);
  STP_LAUNCH_CHECK();
  hipLaunchKernelGGL(prob_cce_finalize_kernel, dim3(1), dim3(256), 0, s, partial, blocks, 1.0 / (double)pixels, 1.0 / ((double)pixels * classes), w_cce,
                     w_dice, scalars);
  STP_LAUNCH_CHECK();
  if (dprobs) {
    if (dl_channels < classes) return STP_E_BADARG;
    const int g = dl_grid(pixels);
    if (dtype == STP_H16) hipLaunchKernelGGL(prob_cce_grad_kernel<bf16_t>, dim3(g), dim3(256), 0, s, (const bf16_t*)probs, target, pixels, classes, ldc, scalars, w_cce, w_dice, (float)(1.0 / (double)pixels), (bf16_t*)dprobs, dl_channels);
    else hipLaunchKernelGGL(prob_cce_grad_kernel<float>, dim3(g), dim3(256), 0, s, (const float*)probs, target, pixels, classes, ldc, scalars, w_cce, w_dice, (float)(1.0 / (double)pixels), (float*)dprobs, dl_channels);
    STP_LAUNCH_CHECK();
  }
  return STP_OK;
}
