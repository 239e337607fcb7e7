// On-device augmentation: one fused pass per batch (replaces the imgaug worker processes of the
// reference, schemas/augmenters.raml:43-133 / README.md:247-268).
//
// Geometry (Fliplr / Flipud / Affine{scale,translate,rotate,shear} / crop / resize) is composed on
// the host into ONE 2x3 matrix per sample that maps output pixel centres to input pixel
// coordinates.  Sampling uses cv2.warpAffine-style fixed point so that results are integer-exact
// and reproducible on any device: coordinates carry 10 fractional bits, bilinear weights 5 bits
// per axis, border = constant 0.  Masks use nearest neighbour.  Colour ops Add (integer) and
// Multiply (float, round-half-even) follow on the uint8 result, saturating like imgaug.
#include "common.h"

struct AugSample {
  float m[6];
  float add, mul, r0, r1;
};

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
    const int addi = (int)s.add;
    for (int c = 0; c < C; ++c) {
      const int v00 = (x0ok && y0ok) ? b[(iy * Win + ix) * C + c] : 0;
      const int v01 = (x1ok && y0ok) ? b[(iy * Win + ix + 1) * C + c] : 0;
      const int v10 = (x0ok && y1ok) ? b[((iy + 1) * Win + ix) * C + c] : 0;
      const int v11 = (x1ok && y1ok) ? b[((iy + 1) * Win + ix + 1) * C + c] : 0;
      int v = (w00 * v00 + w01 * v01 + w10 * v10 + w11 * v11 + 512) >> 10;
      v = min(max(v + addi, 0), 255);
      if (s.mul != 1.f) v = min(max((int)rintf((float)v * s.mul), 0), 255);
      img_out[i * C + c] = (uint8_t)v;
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
