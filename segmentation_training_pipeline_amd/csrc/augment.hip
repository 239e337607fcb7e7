// On-device augmentation: one fused pass per batch (replaces the imgaug worker processes of the
// reference, schemas/augmenters.raml:43-133 / README.md:247-268).
//
// Geometry (Fliplr / Flipud / Affine{scale,translate,rotate,shear} / crop / resize) is composed on
// the host into ONE 2x3 matrix per sample that maps output pixel centres to input pixel
// coordinates.  Sampling uses cv2.warpAffine-style fixed point so that results are integer-exact
// and reproducible on any device: coordinates carry 10 fractional bits, bilinear weights 5 bits
// per axis, border = constant 0.  Masks use nearest neighbour.  The point operations of the catalogue (Add, Multiply,
// their Elementwise forms, AdditiveGaussianNoise, Dropout, Grayscale, Invert) follow on the uint8 result, saturating
// like imgaug; crop / pad augmenters are part of the matrix.
#include "common.h"

// Per-sample record, STP_AUG_RECORD = 24 floats (integers are stored as exactly representable floats):
//   0-5   2x3 output->input matrix
//   6-8   Add per channel (int)            9-11  Multiply per channel (float)
//   12    flags: 1 Invert | 2 noise per channel | 4 dropout per channel | 8 AddElementwise per channel |
//                16 MultiplyElementwise per channel | 32 AddElementwise on | 64 MultiplyElementwise on
//   13    Grayscale alpha in 1/256 (0..256)
//   14    AdditiveGaussianNoise: k = rint(sigma * 65536 / 147.8) (sigma in 8-bit units; noise = Irwin-Hall sum of 4 bytes)
//   15    Dropout: threshold on 24 random bits, rint(p * 2^24)
//   16-17 AddElementwise integer range [lo, hi]     18-19 MultiplyElementwise range [lo, hi]
//   20    seed (integer < 2^24)                      21-23 reserved
// Point operations run in this fixed order after the warp: Add, Multiply, MultiplyElementwise, AddElementwise,
// AdditiveGaussianNoise, Dropout, Grayscale, Invert - all in integer / single-rounded float arithmetic, so the numpy
// oracle (oracle/augment.py) reproduces them bit for bit; randomness is a counter-based hash of (seed, pixel, channel, op).
struct AugSample {
  float m[6];
  float add[3], mul[3];
  float flags, gray_q, noise_k, drop_t;
  float adde_lo, adde_hi, mule_lo, mule_hi;
  float seed, r0, r1, r2;
};

__device__ __forceinline__ uint32_t aug_hash(uint32_t seed, uint32_t pix, uint32_t ch, uint32_t op) {
  uint32_t h = seed ^ (pix * 0x9E3779B1u) ^ (ch * 0x85EBCA77u) ^ (op * 0xC2B2AE3Du);
  h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
  return h;
}
__device__ __forceinline__ int clip255(int v) { return min(max(v, 0), 255); }

__device__ __forceinline__ int64_t fix10(double v) { return (int64_t)__double2ll_rn(v); }

__global__ __launch_bounds__(256) void augment_kernel(const uint8_t* __restrict__ img, const uint8_t* __restrict__ mask,
                                                      uint8_t* __restrict__ img_out, uint8_t* __restrict__ mask_out,
                                                      const AugSample* __restrict__ prm, int N, int Hin, int Win, int Hout,
                                                      int Wout, int C) {
  const int64_t total = (int64_t)N * Hout * Wout;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int xo = (int)(i % Wout);
    const int yo = (int)((i / Wout) % Hout);
    const int n = (int)(i / ((int64_t)Wout * Hout));
    const AugSample s = prm[n];
    // no fused multiply-add here: the CPU oracle rounds each product and sum separately
    const int64_t X0 = fix10(__dmul_rn(__dmul_rn((double)s.m[0], (double)xo), 1024.0)) +
                       fix10(__dmul_rn(__dadd_rn(__dmul_rn((double)s.m[1], (double)yo), (double)s.m[2]), 1024.0));
    const int64_t Y0 = fix10(__dmul_rn(__dmul_rn((double)s.m[3], (double)xo), 1024.0)) +
                       fix10(__dmul_rn(__dadd_rn(__dmul_rn((double)s.m[4], (double)yo), (double)s.m[5]), 1024.0));
    // image: bilinear, 5 fractional bits
    const int64_t X = (X0 + 16) >> 5, Y = (Y0 + 16) >> 5;
    const int64_t ix = X >> 5, iy = Y >> 5;
    const int fx = (int)(X & 31), fy = (int)(Y & 31);
    const int w00 = (32 - fx) * (32 - fy), w01 = fx * (32 - fy), w10 = (32 - fx) * fy, w11 = fx * fy;
    const bool x0ok = ix >= 0 && ix < Win, x1ok = ix + 1 >= 0 && ix + 1 < Win;
    const bool y0ok = iy >= 0 && iy < Hin, y1ok = iy + 1 >= 0 && iy + 1 < Hin;
    const uint8_t* b = img + (int64_t)n * Hin * Win * C;
    const uint32_t flags = (uint32_t)s.flags, seed = (uint32_t)s.seed, pix = (uint32_t)(yo * Wout + xo);
    const int noise_k = (int)s.noise_k;
    const uint32_t drop_t = (uint32_t)s.drop_t;
    const int ae_lo = (int)s.adde_lo, ae_n = (int)s.adde_hi - ae_lo + 1;
    int px[4] = {0, 0, 0, 0};
    for (int c = 0; c < C; ++c) {
      const int v00 = (x0ok && y0ok) ? b[(iy * Win + ix) * C + c] : 0;
      const int v01 = (x1ok && y0ok) ? b[(iy * Win + ix + 1) * C + c] : 0;
      const int v10 = (x0ok && y1ok) ? b[((iy + 1) * Win + ix) * C + c] : 0;
      const int v11 = (x1ok && y1ok) ? b[((iy + 1) * Win + ix + 1) * C + c] : 0;
      int v = (w00 * v00 + w01 * v01 + w10 * v10 + w11 * v11 + 512) >> 10;
      const int cc = c < 3 ? c : 2;
      v = clip255(v + (int)s.add[cc]);
      if (s.mul[cc] != 1.f) v = clip255((int)rintf(__fmul_rn((float)v, s.mul[cc])));
      if (flags & 64u) {
        const float u = __fmul_rn((float)(aug_hash(seed, pix, (flags & 16u) ? c : 0, 1u) >> 8), 5.9604644775390625e-8f);
        const float m = __fadd_rn(s.mule_lo, __fmul_rn(__fsub_rn(s.mule_hi, s.mule_lo), u));
        v = clip255((int)rintf(__fmul_rn((float)v, m)));
      }
      if (flags & 32u) v = clip255(v + ae_lo + (int)(aug_hash(seed, pix, (flags & 8u) ? c : 0, 2u) % (uint32_t)ae_n));
      if (noise_k) {
        const uint32_t h = aug_hash(seed, pix, (flags & 2u) ? c : 0, 3u);
        const int sum = (int)(h & 255u) + (int)((h >> 8) & 255u) + (int)((h >> 16) & 255u) + (int)(h >> 24);
        v = clip255(v + (((sum - 510) * noise_k + 32768) >> 16));
      }
      if (drop_t && (aug_hash(seed, pix, (flags & 4u) ? c : 0, 4u) >> 8) < drop_t) v = 0;
      if (c < 4) px[c] = v;
      if (C != 3) img_out[i * C + c] = (uint8_t)((flags & 1u) ? 255 - v : v);
    }
    if (C == 3) {
      const int gq = (int)s.gray_q;
      if (gq) {  // cv2 RGB2GRAY fixed point, blended with alpha = gq/256
        const int gray = (px[0] * 4899 + px[1] * 9617 + px[2] * 1868 + 8192) >> 14;
#pragma unroll
        for (int c = 0; c < 3; ++c) px[c] = (gq * gray + (256 - gq) * px[c] + 128) >> 8;
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) img_out[i * 3 + c] = (uint8_t)((flags & 1u) ? 255 - px[c] : px[c]);
    }
    if (mask) {
      const int64_t mx = (X0 + 512) >> 10, my = (Y0 + 512) >> 10;
      uint8_t mv = 0;
      if (mx >= 0 && mx < Win && my >= 0 && my < Hin) mv = mask[((int64_t)n * Hin + my) * Win + mx];
      mask_out[i] = mv;
    }
  }
}

extern "C" int stp_augment_u8(const uint8_t* img, const uint8_t* mask, uint8_t* img_out, uint8_t* mask_out, const float* params,
                              int32_t N, int32_t Hin, int32_t Win, int32_t Hout, int32_t Wout, int32_t C, void* stream) {
  if (!img || !img_out || !params || N <= 0 || C <= 0 || (mask && !mask_out)) return STP_E_BADARG;
  int64_t g = ((int64_t)N * Hout * Wout + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(augment_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, img, mask, img_out, mask_out,
                     (const AugSample*)params, N, Hin, Win, Hout, Wout, C);
  STP_LAUNCH_CHECK();
  return STP_OK;
}

// ------------------------------------------------------------------------------------------
// Neighbourhood filters of the catalogue (GaussianBlur, AverageBlur, Sharpen, Emboss, EdgeDetect = a K x K linear
// filter; MedianBlur = rank selection) on the augmented uint8 batch, one per-image record:
//   int32[STP_FILTER_RECORD = 4 + 13*13]: K (odd, 0 = copy, <= 13), mode (0 linear, 1 median), 0, 0, then K*K weights
//   in 1/16384 (row-major; ignored for the median).  Border: reflect-101 (cv2.filter2D / cv2.blur default).
// Integer arithmetic throughout -> bit-exact against the numpy oracle.  Masks are not filtered (as in imgaug).
#define STP_FILTER_RECORD 173
__device__ __forceinline__ int reflect101(int i, int n) {
  if (n == 1) return 0;
  while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
  return i;
}

__global__ __launch_bounds__(256) void filter_u8_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                        const int32_t* __restrict__ prm, int N, int H, int W, int C) {
  const int64_t total = (int64_t)N * H * W * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    const int x = (int)((i / C) % W);
    const int y = (int)((i / ((int64_t)C * W)) % H);
    const int n = (int)(i / ((int64_t)C * W * H));
    const int32_t* r = prm + (size_t)n * STP_FILTER_RECORD;
    const int K = r[0];
    if (K <= 0) { dst[i] = src[i]; continue; }
    const int rad = K >> 1;
    const uint8_t* b = src + (int64_t)n * H * W * C + c;
    if (r[1] == 0) {
      int acc = 0;
      for (int ky = 0; ky < K; ++ky) {
        const int yy = reflect101(y + ky - rad, H);
        for (int kx = 0; kx < K; ++kx) acc += r[4 + ky * K + kx] * (int)b[((int64_t)yy * W + reflect101(x + kx - rad, W)) * C];
      }
      dst[i] = (uint8_t)min(max((acc + 8192) >> 14, 0), 255);
    } else {
      // median of K*K bytes by counting: the smallest value v with #(values <= v) > K*K/2
      int hist_lo = 0, lo = 0, hi = 255;
      const int need = (K * K) / 2 + 1;
      while (lo < hi) {            // binary search over the value range, one pass over the window per step
        const int mid = (lo + hi) >> 1;
        int cnt = 0;
        for (int ky = 0; ky < K; ++ky) {
          const int yy = reflect101(y + ky - rad, H);
          for (int kx = 0; kx < K; ++kx) cnt += (int)b[((int64_t)yy * W + reflect101(x + kx - rad, W)) * C] <= mid;
        }
        if (cnt >= need) hi = mid; else lo = mid + 1;
      }
      (void)hist_lo;
      dst[i] = (uint8_t)lo;
    }
  }
}

extern "C" int stp_filter_u8(const uint8_t* src, uint8_t* dst, const int32_t* params, int32_t N, int32_t H, int32_t W, int32_t C,
                             void* stream) {
  if (!src || !dst || !params || src == dst || N <= 0 || H <= 0 || W <= 0 || C <= 0) return STP_E_BADARG;
  int64_t g = ((int64_t)N * H * W * C + 255) / 256;
  if (g > 16384) g = 16384;
  hipLaunchKernelGGL(filter_u8_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, src, dst, params, N, H, W, C);
  STP_LAUNCH_CHECK();
  return STP_OK;
}
