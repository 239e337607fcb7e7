// lovasz_loss of the loss registry (reference segmentation.py:18; README.md:379,411 use it as a stage loss): the binary
// Lovasz hinge of Berman et al., per image, on the logits recovered from the probabilities the Keras way
// (clip to [1e-7, 1 - 1e-7], log(p / (1 - p))), mean over the images.  Per image, with sign = 2y - 1:
//   err_i = 1 - logit_i * sign_i;  sort descending;  loss = sum_r relu(err_(r)) * g_r,
//   g_r = J_r - J_(r-1),  J_r = 1 - (G - c1_r) / (G + c0_r)   (G = positives of the image, c1 / c0 = positives / negatives among
//   the first r + 1 sorted elements).  g_r is evaluated in closed form (no cancellation):
//   positive at rank r: 1 / (G + c0_r);   negative: (G - c1_r) / ((G + c0_r - 1) (G + c0_r)).
// The gradient treats g as a constant (the surrogate's sub-gradient): d loss / d logit_i = -sign_i [err_i > 0] g_rank(i) / images.
//
// Device path: one key kernel -> ONE chip-wide radix sort of 64-bit composite keys (image index above the order-reversed error
// bits; rocPRIM through hipCUB, temporary storage from the caller's workspace, so nothing allocates or synchronises and the
// launches capture into a hipGraph) -> one workgroup per image scans its sorted run (block scan with a carry, fixed order) ->
// a one-thread finalize.  The result is ADDED to what stp_sigmoid_bce_dice / stp_sigmoid_loss_ex left in scalars[0] and in
// dlogits, so any `a+w*lovasz_loss` composite works.
#include <hipcub/hipcub.hpp>

#include "common.h"

namespace {

__device__ __forceinline__ uint32_t f32_sortable(float f) {       // monotone float -> unsigned
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float f32_unsortable(uint32_t s) {
  return __uint_as_float((s & 0x80000000u) ? (s & 0x7fffffffu) : ~s);
}

__device__ __forceinline__ float keras_logit(float z, bool* in_range) {
  const float eps = 1e-7f, hi = 1.f - 1e-7f;
  const float p = 1.f / (1.f + expf(-z));
  *in_range = (p >= eps) && (p <= hi);
  const float pc = fminf(fmaxf(p, eps), hi);
  return logf(pc / (1.f - pc));
}

template <typename T>
__global__ __launch_bounds__(256) void lovasz_keys_kernel(const T* __restrict__ logits, const uint8_t* __restrict__ target, int64_t count,
                                                          int64_t per_image, uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) {
    const uint32_t y = target[i] ? 1u : 0u;
    bool inr;
    const float zz = keras_logit(Elem<T>::load(logits + i), &inr);
    const float err = 1.f - zz * (y ? 1.f : -1.f);
    const int64_t n = i / per_image;
    keys[i] = ((uint64_t)n << 32) | (uint64_t)(~f32_sortable(err));        // ascending key = descending error inside the image
    vals[i] = ((uint32_t)(i - n * per_image) << 1) | y;
  }
}

#define LV_T 1024
template <typename T>
__global__ __launch_bounds__(LV_T) void lovasz_scan_kernel(const T* __restrict__ logits, const uint64_t* __restrict__ keys,
                                                           const uint32_t* __restrict__ vals, int64_t per_image, float gscale,
                                                           T* __restrict__ dl, int dlc, float* __restrict__ image_loss) {
  __shared__ uint32_t wsum[LV_T / 64];
  __shared__ float fsum[LV_T / 64];
  __shared__ uint32_t carry_s;
  const int n = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
  const uint64_t* k = keys + (int64_t)n * per_image;
  const uint32_t* v = vals + (int64_t)n * per_image;
  // positives of the image
  uint32_t g = 0;
  for (int64_t r = t; r < per_image; r += LV_T) g += v[r] & 1u;
  for (int o = 32; o > 0; o >>= 1) g += __shfl_xor(g, o);
  if (lane == 0) wsum[w] = g;
  __syncthreads();
  uint32_t G = 0;
  for (int i = 0; i < LV_T / 64; ++i) G += wsum[i];
  __syncthreads();
  if (t == 0) carry_s = 0;
  __syncthreads();
  const float Gf = (float)G;
  float acc = 0.f;
  for (int64_t r0 = 0; r0 < per_image; r0 += LV_T) {
    const int64_t r = r0 + t;
    const bool live = r < per_image;
    const uint32_t val = live ? v[r] : 0u;
    const uint32_t y = val & 1u;
    // inclusive scan of y over the chunk: wave scan + wave totals
    uint32_t s = y;
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t u = __shfl_up(s, o);
      if (lane >= o) s += u;
    }
    if (lane == 63) wsum[w] = s;
    __syncthreads();
    uint32_t base = carry_s;
    for (int i = 0; i < w; ++i) base += wsum[i];
    const uint32_t c1 = base + s;                              // positives among ranks 0..r
    __syncthreads();
    if (t == LV_T - 1) carry_s = c1;
    if (live) {
      const float c0 = (float)((uint32_t)(r + 1) - c1);        // negatives among ranks 0..r
      const float den = Gf + c0;
      // (an image without positives: J = 1 at every rank, so the whole weight sits on the largest error)
      const float gr = G == 0 ? (r == 0 ? 1.f : 0.f) : (y ? 1.f / den : (Gf - (float)c1) / ((den - 1.f) * den));
      const float err = f32_unsortable(~(uint32_t)(k[r] & 0xffffffffu));
      if (err > 0.f) {
        acc += err * gr;
        if (dl) {
          const int64_t i = (int64_t)n * per_image + (val >> 1);
          bool inr;
          keras_logit(Elem<T>::load(logits + i), &inr);
          if (inr) {
            T* o = dl + i * dlc;
            Elem<T>::store(o, Elem<T>::load(o) - (y ? 1.f : -1.f) * gr * gscale);
          }
        }
      }
    }
    __syncthreads();
  }
  // fixed-order block sum
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane == 0) fsum[w] = acc;
  __syncthreads();
  if (t == 0) {
    float s = 0.f;
    for (int i = 0; i < LV_T / 64; ++i) s += fsum[i];
    image_loss[n] = s;
  }
}

// scalars[12] = lovasz_loss, scalars[0] += weight * lovasz_loss
__global__ void lovasz_finalize_kernel(const float* image_loss, int images, float weight, float* scalars) {
  if (threadIdx.x || blockIdx.x) return;
  double s = 0.0;
  for (int i = 0; i < images; ++i) s += (double)image_loss[i];
  const float l = (float)(s / (double)images);
  scalars[12] = l;
  scalars[0] += weight * l;
}

struct LvLayout { size_t keys_in, keys_out, vals_in, vals_out, image_loss, temp, temp_bytes, total; };

static int lv_end_bit(int images) {
  int b = 0;
  while ((1 << b) < images) ++b;
  return 32 + b;
}

static int lv_layout(int64_t count, int images, LvLayout* L) {
  size_t temp = 0;
  if (hipcub::DeviceRadixSort::SortPairs(nullptr, temp, (const uint64_t*)nullptr, (uint64_t*)nullptr, (const uint32_t*)nullptr,
                                         (uint32_t*)nullptr, (int)count, 0, lv_end_bit(images), (hipStream_t)0) != hipSuccess)
    return STP_E_LAUNCH;
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  size_t o = 0;
  L->keys_in = o; o += up((size_t)count * 8);
  L->keys_out = o; o += up((size_t)count * 8);
  L->vals_in = o; o += up((size_t)count * 4);
  L->vals_out = o; o += up((size_t)count * 4);
  L->image_loss = o; o += up((size_t)images * 4);
  L->temp = o; L->temp_bytes = temp; o += up(temp);
  L->total = o;
  return STP_OK;
}

}  // namespace

extern "C" size_t stp_lovasz_workspace_bytes(int64_t count, int32_t images) {
  LvLayout L;
  if (count <= 0 || images <= 0 || count >= ((int64_t)1 << 31) || lv_layout(count, images, &L) != STP_OK) return 0;
  return L.total;
}

extern "C" int stp_lovasz_hinge(const void* logits, const uint8_t* target, int32_t images, int64_t per_image, int32_t dtype, float weight,
                                float* scalars, void* dlogits, int32_t dl_channels, void* workspace, size_t workspace_bytes, void* stream) {
  if (!stp_dtype_ok(dtype)) return STP_E_BADARG;      // (the other build's 16-bit code, or garbage)
  const int64_t count = (int64_t)images * per_image;
  if (!logits || !target || !scalars || !workspace || images <= 0 || per_image <= 0 || count >= ((int64_t)1 << 31) ||
      per_image >= ((int64_t)1 << 31) || (dtype != STP_H16 && dtype != STP_F32) || (dlogits && dl_channels < 1))
    return STP_E_BADARG;
  LvLayout L;
  if (lv_layout(count, images, &L) != STP_OK) return STP_E_LAUNCH;
  if (workspace_bytes < L.total) return STP_E_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  char* ws = (char*)workspace;
  uint64_t *kin = (uint64_t*)(ws + L.keys_in), *kout = (uint64_t*)(ws + L.keys_out);
  uint32_t *vin = (uint32_t*)(ws + L.vals_in), *vout = (uint32_t*)(ws + L.vals_out);
  float* il = (float*)(ws + L.image_loss);
  int64_t g = (count + 255) / 256;
  if (g > 8192) g = 8192;
  if (dtype == STP_H16)
    hipLaunchKernelGGL(lovasz_keys_kernel<bf16_t>, dim3((int)g), dim3(256), 0, s, (const bf16_t*)logits, target, count, per_image, kin, vin);
  else
    hipLaunchKernelGGL(lovasz_keys_kernel<float>, dim3((int)g), dim3(256), 0, s, (const float*)logits, target, count, per_image, kin, vin);
  STP_LAUNCH_CHECK();
  size_t temp = L.temp_bytes;
  if (hipcub::DeviceRadixSort::SortPairs(ws + L.temp, temp, (const uint64_t*)kin, kout, (const uint32_t*)vin, vout, (int)count, 0,
                                         lv_end_bit(images), s) != hipSuccess)
    return STP_E_LAUNCH;
  const float gscale = weight / (float)images;
  if (dtype == STP_H16)
    hipLaunchKernelGGL(lovasz_scan_kernel<bf16_t>, dim3(images), dim3(LV_T), 0, s, (const bf16_t*)logits, kout, vout, per_image, gscale,
                       (bf16_t*)dlogits, dl_channels, il);
  else
    hipLaunchKernelGGL(lovasz_scan_kernel<float>, dim3(images), dim3(LV_T), 0, s, (const float*)logits, kout, vout, per_image, gscale,
                       (float*)dlogits, dl_channels, il);
  STP_LAUNCH_CHECK();
  hipLaunchKernelGGL(lovasz_finalize_kernel, dim3(1), dim3(64), 0, s, il, images, weight, scalars);
  STP_LAUNCH_CHECK();
  return STP_OK;
}
