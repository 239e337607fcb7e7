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
#include <cstdlib>

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
                                                      int Wout, int C, const int32_t* __restrict__ field) {
  const int64_t total = (int64_t)N * Hout * Wout;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int xo = (int)(i % Wout);
    const int yo = (int)((i / Wout) % Hout);
    const int n = (int)(i / ((int64_t)Wout * Hout));
    const AugSample s = prm[n];
    // displacement field (PiecewiseAffine / ElasticTransformation): the output pixel samples the warped canvas at p + D(p),
    // D in 1/64 pixel (two int16 per pixel) - exact in double, so the matrix arithmetic below is unchanged
    double xd = (double)xo, yd = (double)yo;
    if (field) {
      const int32_t f = field[i];
      xd = __dadd_rn(xd, __dmul_rn((double)(int16_t)(f & 0xffff), 0.015625));
      yd = __dadd_rn(yd, __dmul_rn((double)(int16_t)(f >> 16), 0.015625));
    }
    // no fused multiply-add here: the CPU oracle rounds each product and sum separately
    const int64_t X0 = fix10(__dmul_rn(__dmul_rn((double)s.m[0], xd), 1024.0)) +
                       fix10(__dmul_rn(__dadd_rn(__dmul_rn((double)s.m[1], yd), (double)s.m[2]), 1024.0));
    const int64_t Y0 = fix10(__dmul_rn(__dmul_rn((double)s.m[3], xd), 1024.0)) +
                       fix10(__dmul_rn(__dadd_rn(__dmul_rn((double)s.m[4], yd), (double)s.m[5]), 1024.0));
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

static int launch_augment(const uint8_t* img, const uint8_t* mask, uint8_t* img_out, uint8_t* mask_out, const float* params,
                          const int32_t* field, int32_t N, int32_t Hin, int32_t Win, int32_t Hout, int32_t Wout, int32_t C, void* stream) {
  if (!img || !img_out || !params || N <= 0 || C <= 0 || (mask && !mask_out)) return STP_E_BADARG;
  int64_t g = ((int64_t)N * Hout * Wout + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(augment_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, img, mask, img_out, mask_out,
                     (const AugSample*)params, N, Hin, Win, Hout, Wout, C, field);
  STP_LAUNCH_CHECK();
  return STP_OK;
}

extern "C" int stp_augment_u8(const uint8_t* img, const uint8_t* mask, uint8_t* img_out, uint8_t* mask_out, const float* params,
                              int32_t N, int32_t Hin, int32_t Win, int32_t Hout, int32_t Wout, int32_t C, void* stream) {
  return launch_augment(img, mask, img_out, mask_out, params, nullptr, N, Hin, Win, Hout, Wout, C, stream);
}

// the same pass with a per-pixel displacement field [N][Hout][Wout] (stp_field_piecewise / stp_field_elastic; NULL = none)
extern "C" int stp_augment_field_u8(const uint8_t* img, const uint8_t* mask, uint8_t* img_out, uint8_t* mask_out, const float* params,
                                    const int32_t* field, int32_t N, int32_t Hin, int32_t Win, int32_t Hout, int32_t Wout, int32_t C,
                                    void* stream) {
  return launch_augment(img, mask, img_out, mask_out, params, field, N, Hin, Win, Hout, Wout, C, stream);
}

// ------------------------------------------------------------------------------------------
// Displacement fields of the two non-affine geometric augmenters (schemas/augmenters.raml:126-133), integer arithmetic
// throughout (bit-exact against oracle/augment.py); a field entry packs (dx, dy) as two int16 in 1/64 pixel.
__device__ __forceinline__ int32_t pack_disp(int dx, int dy) {
  dx = min(max(dx, -32768), 32767);
  dy = min(max(dy, -32768), 32767);
  return (int32_t)(((uint32_t)dx & 0xffffu) | ((uint32_t)dy << 16));
}

// PiecewiseAffine: a rows x cols grid of control points at linspace(0, H, rows) x linspace(0, W, cols) (imgaug / skimage
// PiecewiseAffineTransform), each moved by the sampled jitter grid[n][row][col] = (dx, dy) in 1/64 pixel; inside a cell the
// displacement is affine on each of the two triangles the (0,0)-(1,1) diagonal cuts (10 fractional bits of cell position).
__global__ __launch_bounds__(256) void field_piecewise_kernel(int32_t* __restrict__ field, const int32_t* __restrict__ grid, int N, int H, int W,
                                                              int R, int Cg) {
  const int64_t total = (int64_t)N * H * W;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int x = (int)(i % W), y = (int)((i / W) % H), n = (int)(i / ((int64_t)W * H));
    const int64_t u = (int64_t)x * (Cg - 1) * 1024 / W, v = (int64_t)y * (R - 1) * 1024 / H;
    const int cx = min((int)(u >> 10), Cg - 2), cy = min((int)(v >> 10), R - 2);
    const int64_t fu = u - ((int64_t)cx << 10), fv = v - ((int64_t)cy << 10);
    const int32_t* g = grid + (((int64_t)n * R + cy) * Cg + cx) * 2;
    int d[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int64_t d00 = g[a], d10 = g[2 + a], d01 = g[2 * Cg + a], d11 = g[2 * Cg + 2 + a];
      const int64_t t = fu >= fv ? d00 * 1024 + fu * (d10 - d00) + fv * (d11 - d10) : d00 * 1024 + fv * (d01 - d00) + fu * (d11 - d01);
      d[a] = (int)((t + 512) >> 10);
    }
    field[i] = pack_disp(d[0], d[1]);
  }
}

extern "C" int stp_field_piecewise(int32_t* field, const int32_t* grid, int32_t N, int32_t H, int32_t W, int32_t rows, int32_t cols,
                                   void* stream) {
  if (!field || !grid || N <= 0 || H <= 0 || W <= 0 || rows < 2 || cols < 2) return STP_E_BADARG;
  int64_t g = ((int64_t)N * H * W + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(field_piecewise_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, field, grid, N, H, W, rows, cols);
  STP_LAUNCH_CHECK();
  return STP_OK;
}

// ElasticTransformation (imgaug: uniform(-1, 1) noise per pixel and axis -> gaussian_filter(sigma, mode='constant') -> * alpha).
// Record int32[STP_ELASTIC_RECORD = 4 + 65]: seed, alpha in 1/64 pixel, radius r (<= 64; scipy truncate = 4 sigma), 0, then the
// one-sided kernel w[0..r] in 1/32768 (the full kernel sums to exactly 32768).  Noise = 16 hash bits - 32768; separable blur in
// two passes, each rounded to 1/32768; displacement = (alpha * v + 2^14) >> 15.
#define STP_ELASTIC_RECORD 69
__global__ __launch_bounds__(256) void field_elastic_h_kernel(int32_t* __restrict__ tmp, const int32_t* __restrict__ prm, int N, int H, int W) {
  const int64_t total = (int64_t)N * H * W;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int x = (int)(i % W), y = (int)((i / W) % H), n = (int)(i / ((int64_t)W * H));
    const int32_t* p = prm + (int64_t)n * STP_ELASTIC_RECORD;
    const uint32_t seed = (uint32_t)p[0];
    const int r = p[2];
    int t[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      int sum = 0;
      for (int k = -r; k <= r; ++k) {
        const int xx = x + k;
        if (xx < 0 || xx >= W) continue;
        const int nz = (int)(aug_hash(seed, (uint32_t)(y * W + xx), (uint32_t)a, 7u) >> 16) - 32768;
        sum += p[4 + (k < 0 ? -k : k)] * nz;
      }
      t[a] = (sum + 16384) >> 15;
    }
    tmp[i] = pack_disp(t[0], t[1]);
  }
}
__global__ __launch_bounds__(256) void field_elastic_v_kernel(int32_t* __restrict__ field, const int32_t* __restrict__ tmp,
                                                              const int32_t* __restrict__ prm, int N, int H, int W) {
  const int64_t total = (int64_t)N * H * W;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int x = (int)(i % W), y = (int)((i / W) % H), n = (int)(i / ((int64_t)W * H));
    const int32_t* p = prm + (int64_t)n * STP_ELASTIC_RECORD;
    const int alpha = p[1], r = p[2];
    int s0 = 0, s1 = 0;
    for (int k = -r; k <= r; ++k) {
      const int yy = y + k;
      if (yy < 0 || yy >= H) continue;
      const int32_t f = tmp[((int64_t)n * H + yy) * W + x];
      const int wk = p[4 + (k < 0 ? -k : k)];
      s0 += wk * (int)(int16_t)(f & 0xffff);
      s1 += wk * (int)(int16_t)(f >> 16);
    }
    const int v0 = (s0 + 16384) >> 15, v1 = (s1 + 16384) >> 15;
    field[i] = pack_disp((alpha * v0 + 16384) >> 15, (alpha * v1 + 16384) >> 15);
  }
}

extern "C" int stp_field_elastic(int32_t* field, int32_t* tmp, const int32_t* params, int32_t N, int32_t H, int32_t W, void* stream) {
  if (!field || !tmp || !params || field == tmp || N <= 0 || H <= 0 || W <= 0) return STP_E_BADARG;
  int64_t g = ((int64_t)N * H * W + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(field_elastic_h_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, tmp, params, N, H, W);
  STP_LAUNCH_CHECK();
  hipLaunchKernelGGL(field_elastic_v_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, field, tmp, params, N, H, W);
  STP_LAUNCH_CHECK();
  return STP_OK;
}

// ------------------------------------------------------------------------------------------
// BackgroundReplacer (musket's augmenter for background-removal tasks, reference README.md:270-278, FAQ.md:24-38): pixels outside
// the (eroded) mask take the pixel of a background image already resized to the item's size.  Erosion e = a (2e+1) x (2e+1)
// minimum over the mask (pixels outside the image do not erode, cv2.erode's default border); the mask itself is not changed.
__global__ __launch_bounds__(256) void background_replace_kernel(const uint8_t* __restrict__ img, const uint8_t* __restrict__ mask,
                                                                 const uint8_t* __restrict__ bg, uint8_t* __restrict__ out, int N, int H, int W,
                                                                 int C, int erosion) {
  const int64_t total = (int64_t)N * H * W;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int x = (int)(i % W), y = (int)((i / W) % H);
    const int64_t base = i - ((int64_t)y * W + x);
    bool fg = mask[i] != 0;
    for (int dy = -erosion; fg && dy <= erosion; ++dy) {
      const int yy = y + dy;
      if (yy < 0 || yy >= H) continue;
      for (int dx = -erosion; dx <= erosion; ++dx) {
        const int xx = x + dx;
        if (xx >= 0 && xx < W && mask[base + (int64_t)yy * W + xx] == 0) { fg = false; break; }
      }
    }
    const uint8_t* src = fg ? img : bg;
    for (int c = 0; c < C; ++c) out[i * C + c] = src[i * C + c];
  }
}

extern "C" int stp_background_replace_u8(const uint8_t* img, const uint8_t* mask, const uint8_t* bg, uint8_t* out, int32_t N, int32_t H,
                                         int32_t W, int32_t C, int32_t erosion, void* stream) {
  if (!img || !mask || !bg || !out || out == img || N <= 0 || H <= 0 || W <= 0 || C <= 0 || erosion < 0 || erosion > 32) return STP_E_BADARG;
  int64_t g = ((int64_t)N * H * W + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(background_replace_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, img, mask, bg, out, N, H, W, C, erosion);
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

// one output element (image n, row y, column x, channel c) the direct way: K x K byte loads from global memory (median: 8 passes over them)
__device__ __forceinline__ uint8_t filter_one(const uint8_t* __restrict__ src, const int32_t* __restrict__ r, int n, int y, int x, int c, int H, int W, int C) {
  const int K = r[0];
  const int rad = K >> 1;
  const uint8_t* b = src + (int64_t)n * H * W * C + c;
  if (r[1] == 0) {
    int acc = 0;
    for (int ky = 0; ky < K; ++ky) {
      const int yy = reflect101(y + ky - rad, H);
      for (int kx = 0; kx < K; ++kx) acc += r[4 + ky * K + kx] * (int)b[((int64_t)yy * W + reflect101(x + kx - rad, W)) * C];
    }
    return (uint8_t)min(max((acc + 8192) >> 14, 0), 255);
  }
  // median of K*K bytes by counting: the smallest value v with #(values <= v) > K*K/2
  int lo = 0, hi = 255;
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
  return (uint8_t)lo;
}

// Round 6: the linear filters on a 64 x 16 pixel tile staged in LDS.  The element-per-thread kernel above was the largest non-GEMM launch
// of configs[4]'s step (534 us for 8 x 768 x 768 x 3 with the S4 GaussianBlur, K = 5 .. 11: K^2 byte loads through reflect-101 index
// arithmetic, a weight load and a quarter-rate 32-bit multiply per tap and element).  Here the tile's (16 + K - 1) x (64 + K - 1) patch
// goes to LDS once as one plane per channel (border reflection paid once per patch byte); a thread owns FOUR adjacent pixels of a row: per
// kernel row it reads 16 patch bytes as 4 aligned dwords and feeds every byte to the <= 4 outputs that use it with v_mad_i32_i24
// (weights are wave-uniform: scalar loads; |weight| < 2^23 / 16384 = 512, far above every filter of the catalogue), K a compile-time
// constant per image (uniform branch).  Same integers as the direct form: the sums are exact in any order.
#define FT_W 64
#define FT_H 16
#define FT_PITCH 80      // bytes per patch row: 64 + 12 columns of the widest kernel, padded for the dword reads (col0 <= 60, 16 bytes)
#define FT_ROWS 28       // 16 + 12
template <int K>
__device__ __forceinline__ void filter_tile_linear(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, const int32_t* __restrict__ r, int n,
                                                   int ty0, int tx0, int H, int W, int C, uint8_t* patch) {
  constexpr int R = K / 2;
  const int rows = min(FT_H, H - ty0) + K - 1, cols = min(FT_W, W - tx0) + K - 1;
  const uint8_t* img = src + (int64_t)n * H * W * C;
  // patch[c][row][col] = image(reflect(ty0 - R + row), reflect(tx0 - R + col), c); channel fastest in the loop = the order in memory
  for (int i = threadIdx.x; i < rows * cols * C; i += 256) {
    const int c = i % C, pc = i / C;
    const int row = pc / cols, col = pc - row * cols;
    patch[(c * FT_ROWS + row) * FT_PITCH + col] = img[((int64_t)reflect101(ty0 - R + row, H) * W + reflect101(tx0 - R + col, W)) * C + c];
  }
  __syncthreads();
  const int ly = threadIdx.x >> 4, lx0 = (threadIdx.x & 15) * 4;
  const int y = ty0 + ly, x = tx0 + lx0;
  if (y < H && x < W) {
    const int32_t* w = r + 4;
    for (int c = 0; c < C; ++c) {
      int acc[4] = {0, 0, 0, 0};
      const uint8_t* pl = patch + (c * FT_ROWS + ly) * FT_PITCH + lx0;
#pragma unroll
      for (int ky = 0; ky < K; ++ky) {
        const uint32_t* q = reinterpret_cast<const uint32_t*>(pl + ky * FT_PITCH);
        uint32_t d[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) d[j] = (j * 4 < K + 3) ? q[j] : 0u;
        int b[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) b[j] = (int)__builtin_amdgcn_ubfe(d[j >> 2], 8 * (j & 3), 8);
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
          const int wt = w[ky * K + kx];
#pragma unroll
          for (int p = 0; p < 4; ++p) acc[p] += __mul24(wt, b[p + kx]);
        }
      }
#pragma unroll
      for (int p = 0; p < 4; ++p)
        if (x + p < W) dst[(((int64_t)n * H + y) * W + x + p) * C + c] = (uint8_t)min(max((acc[p] + 8192) >> 14, 0), 255);
    }
  }
}

__global__ __launch_bounds__(256) void filter_u8_tile_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                             const int32_t* __restrict__ prm, int H, int W, int C) {
  __shared__ __attribute__((aligned(16))) uint8_t patch[4 * FT_ROWS * FT_PITCH];
  const int n = blockIdx.z, ty0 = blockIdx.y * FT_H, tx0 = blockIdx.x * FT_W;
  const int32_t* r = prm + (size_t)n * STP_FILTER_RECORD;
  const int K = r[0], mode = r[1];
  if (K > 0 && mode == 0 && (K & 1) && K <= 13 && C <= 4) {
    switch (K) {
      case 1: filter_tile_linear<1>(src, dst, r, n, ty0, tx0, H, W, C, patch); break;
      case 3: filter_tile_linear<3>(src, dst, r, n, ty0, tx0, H, W, C, patch); break;
      case 5: filter_tile_linear<5>(src, dst, r, n, ty0, tx0, H, W, C, patch); break;
      case 7: filter_tile_linear<7>(src, dst, r, n, ty0, tx0, H, W, C, patch); break;
      case 9: filter_tile_linear<9>(src, dst, r, n, ty0, tx0, H, W, C, patch); break;
      case 11: filter_tile_linear<11>(src, dst, r, n, ty0, tx0, H, W, C, patch); break;
      default: filter_tile_linear<13>(src, dst, r, n, ty0, tx0, H, W, C, patch); break;
    }
    return;
  }
  // no filter (copy), the median, or a shape the tile form does not serve: element by element over the tile
  const int th = min(FT_H, H - ty0), tw = min(FT_W, W - tx0);
  for (int i = threadIdx.x; i < th * tw * C; i += 256) {
    const int c = i % C, pc = i / C;
    const int y = ty0 + pc / tw, x = tx0 + pc % tw;
    const int64_t o = (((int64_t)n * H + y) * W + x) * C + c;
    dst[o] = K <= 0 ? src[o] : filter_one(src, r, n, y, x, c, H, W, C);
  }
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
    dst[i] = r[0] <= 0 ? src[i] : filter_one(src, r, n, y, x, c, H, W, C);
  }
}

extern "C" int stp_filter_u8(const uint8_t* src, uint8_t* dst, const int32_t* params, int32_t N, int32_t H, int32_t W, int32_t C,
                             void* stream) {
  if (!src || !dst || !params || src == dst || N <= 0 || H <= 0 || W <= 0 || C <= 0) return STP_E_BADARG;
  static const bool tile_on = !(getenv("STP_FILTER_TILE") && atoi(getenv("STP_FILTER_TILE")) == 0);
  const int gx = (W + FT_W - 1) / FT_W, gy = (H + FT_H - 1) / FT_H;
  if (tile_on && C <= 4 && N <= 65535 && gy <= 65535) {
    hipLaunchKernelGGL(filter_u8_tile_kernel, dim3(gx, gy, N), dim3(256), 0, (hipStream_t)stream, src, dst, params, H, W, C);
    STP_LAUNCH_CHECK();
    return STP_OK;
  }
  int64_t g = ((int64_t)N * H * W * C + 255) / 256;
  if (g > 16384) g = 16384;
  hipLaunchKernelGGL(filter_u8_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, src, dst, params, N, H, W, C);
  STP_LAUNCH_CHECK();
  return STP_OK;
}
