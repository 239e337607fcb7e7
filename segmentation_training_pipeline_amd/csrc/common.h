// Shared device helpers for the gfx950 kernels (wave = 64 lanes, 16-byte vector access).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/stp_hip.h"

#define STP_LAUNCH_CHECK()                       \
  do {                                           \
    if (hipGetLastError() != hipSuccess) return STP_E_LAUNCH; \
  } while (0)

// The 16-bit storage format is a BUILD parameter of the whole kernel set: libstp_hip.so stores bfloat16 (STP_BF16),
// libstp_hip_f16.so - the same sources compiled with -DSTP_STORAGE_F16=1 - stores IEEE half (STP_F16) and feeds the
// v_mfma_*_f16 instructions (same rate and fp32 accumulation as the bf16 ones on gfx950).  `bf16_t` is "the 16-bit storage
// word" in either build; every conversion goes through the helpers below, STP_H16 is the dtype code this build accepts.
#ifndef STP_STORAGE_F16
#define STP_STORAGE_F16 0
#endif
#define STP_H16 (STP_STORAGE_F16 ? STP_F16 : STP_BF16)
// every entry point with a `dtype` argument starts with this: fp32 or THIS build's 16-bit format
static inline bool stp_dtype_ok(int dtype) { return dtype == STP_F32 || dtype == STP_H16; }

typedef uint16_t bf16_t;  // raw 16-bit storage word (bf16 or IEEE half, see above)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));

#if STP_STORAGE_F16
// IEEE half: v_cvt_f32_f16 / v_cvt_f16_f32 (round-to-nearest-even, as torch.float16 casts)
__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
__device__ __forceinline__ float h16lo_to_f32(uint32_t w) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(w & 0xffffu)); }
__device__ __forceinline__ float h16hi_to_f32(uint32_t w) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(w >> 16)); }
// Stores SATURATE at +-65504 (one v_med3_f32 per element): a value that overflows the format becomes its largest finite value, not
// an inf that the next BatchNormalization turns into NaN for the whole map - e.g. the inference-phase pass of a barely trained
// network, whose moving statistics do not normalise yet.  (NaN inputs stay NaN.)
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  const f32x2 v = {__builtin_amdgcn_fmed3f(lo, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(hi, -65504.f, 65504.f)};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_t));
}
template <typename V> __device__ __forceinline__ f32x4 mfma16_16x16x32(V a, V b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
#else
__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ float h16lo_to_f32(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float h16hi_to_f32(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
// fp32 -> bf16, round-to-nearest-even: the gfx950 hardware conversion (v_cvt_pk_bf16_f32), same rule
// as torch / ml_dtypes bf16 casts.
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  const f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
template <typename V> __device__ __forceinline__ f32x4 mfma16_16x16x32(V a, V b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
#endif
// v_mfma_f32_32x32x16 of the build's 16-bit storage format
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <typename V> __device__ __forceinline__ f32x16 mfma16_32x32x16(V a, V b, f32x16 c) {
#if STP_STORAGE_F16
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
#else
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
#endif
}
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return (bf16_t)(pack_bf16x2(f, 0.f) & 0xffffu); }

template <typename T> struct Elem;
template <> struct Elem<float> {
  static constexpr int VEC = 4;  // elements per 16 bytes
  static constexpr int DTYPE = STP_F32;
  __device__ static __forceinline__ float load(const float* p) { return *p; }
  __device__ static __forceinline__ void store(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
  static constexpr int VEC = 8;
  static constexpr int DTYPE = STP_H16;
  __device__ static __forceinline__ float load(const bf16_t* p) { return bf16_to_f32(*p); }
  __device__ static __forceinline__ void store(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};

// 4 consecutive elements <-> f32x4 (8-byte access for bf16, 16-byte for fp32)
__device__ __forceinline__ f32x4 load4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 load4(const bf16_t* p) {
  u32x2 r = *reinterpret_cast<const u32x2*>(p);
  f32x4 o;
  o.x = h16lo_to_f32(r.x);
  o.y = h16hi_to_f32(r.x);
  o.z = h16lo_to_f32(r.y);
  o.w = h16hi_to_f32(r.y);
  return o;
}
__device__ __forceinline__ void store4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ void store4(bf16_t* p, f32x4 v) {
  u32x2 r;
  r.x = pack_bf16x2(v.x, v.y);
  r.y = pack_bf16x2(v.z, v.w);
#if defined(STP_EXP) && STP_EXP == 41      // what-if: write-through stores (no dirty lines left for the end-of-kernel L2 write-back)
  asm volatile("global_store_dwordx2 %0, %1, off sc1\n s_nop 1" ::"v"(p), "v"(r) : "memory");
#elif defined(STP_EXP) && STP_EXP == 42
  __builtin_nontemporal_store(r, reinterpret_cast<u32x2*>(p));
#else
  *reinterpret_cast<u32x2*>(p) = r;
#endif
}

// LDS-only workgroup barrier: __syncthreads() also fences global memory, i.e. waits for every outstanding global load AND store
// (vmcnt counts stores on CDNA) - in the epilogue that would serialise the operand prefetch, the output stores and the reduction
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Sum over each aligned group of 16 lanes (a DPP "row") with 4 v_add_f32 row_shr: the total lands in lane 15
// of the row (other lanes hold partial prefix sums).  No LDS traffic (unlike __shfl_xor -> ds_bpermute).
__device__ __forceinline__ float row_sum16_to_lane15(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));  // row_shr:1
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));  // row_shr:2
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));  // row_shr:4
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));  // row_shr:8
  return v;
}

// Exact unsigned 32-bit division by a runtime constant (round-up method, Granlund-Montgomery):
// l = ceil(log2 d), m = floor(2^32 (2^l - d) / d) + 1, q = (t + ((n - t) >> 1)) >> (l - 1), t = mulhi(m, n).
struct FastDiv {
  uint32_t d, magic, shift;
};
__host__ __device__ static inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f;
  f.d = d;
  f.magic = 0;
  f.shift = 0;
  if (d > 1) {
    uint32_t l = 0;
    while ((1ull << l) < d) ++l;
    f.magic = (uint32_t)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
    f.shift = l - 1;
  }
  return f;
}
__device__ __forceinline__ uint32_t fdiv(uint32_t n, FastDiv f) {
  if (f.d <= 1) return n;
  const uint32_t t = __umulhi(n, f.magic);
  return (t + ((n - t) >> 1)) >> f.shift;
}

static inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
static inline int round_up(int a, int b) { return (a + b - 1) / b * b; }

// XCD-aware work order: consecutive workgroup ids round-robin over the 8 XCDs (each with its own L2), so map
// them onto 8 contiguous runs of the logical tile order - neighbouring tiles (shared 3x3 halo rows, or the
// same pixel range of a weight-gradient split) then meet in one L2.  Bijective for any nblk.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, x = bid & 7, j = bid >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
}

// x*scale + shift exactly as the BatchNormalization forward computed it: the backward passes re-derive the
// ReLU mask from this expression, so every user must round identically (single fma).
__device__ __forceinline__ float bn_affine(float x, float scale, float shift) { return fmaf(x, scale, shift); }
// activation fused behind a BatchNormalization: 0 none, 1 ReLU, 2 ReLU6 (K.relu(x, max_value=6): MobileNetV2 blocks).
// bn_act_on = "the gradient passes" (strictly inside the linear range, as TF's relu6 gradient does)
// (one v_med3_f32 against bounds that depend only on the mode: -inf..inf, 0..inf, 0..6)
__device__ __forceinline__ float bn_act(float v, int relu) {
  const float lo = relu ? 0.f : -__builtin_inff(), hi = relu == 2 ? 6.f : __builtin_inff();
  return __builtin_amdgcn_fmed3f(v, lo, hi);
}
__device__ __forceinline__ bool bn_act_on(float v, int relu) { return relu == 0 || (v > 0.f && (relu == 1 || v < 6.f)); }

// Fused BatchNormalization sums, "slot" form: instead of one partial per pixel tile (reduced by a finalize kernel), the
// epilogue adds its tile's sum into one of a few int64 FIXED-POINT slots per channel (2^-24 units).  Integer addition is
// associative, so the result does not depend on the order in which workgroups arrive: deterministic without a second
// kernel; the consumer (BN apply / BN-backward apply) sums the <= 16 slots of each channel in its prologue.
#define STP_SLOT_SCALE 16777216.0
__device__ __forceinline__ void slot_add(long long* slots, int nslots, int channel, int tile, float v) {
  atomicAdd(reinterpret_cast<unsigned long long*>(slots) + (size_t)channel * nslots + (tile & (nslots - 1)),
            (unsigned long long)__double2ll_rn((double)v * STP_SLOT_SCALE));
}
__device__ __forceinline__ double slot_sum(const long long* slots, int nslots, int channel) {
  long long s = 0;
  for (int k = 0; k < nslots; ++k) s += slots[(size_t)channel * nslots + k];
  return (double)s * (1.0 / STP_SLOT_SCALE);
}

// BatchNormalization-backward partial sums fused into the epilogue of the data-gradient convolution that
// produces dY of the BN output:  g = dY * [relu mask],  sum(g) and sum(g * xhat) per channel, g stored in
// place of dY (see stp_conv_params.bnb_x).
struct BnBack {
  const char* x;  // BN input, same [pixels][C] layout and dtype as the convolution's destination
  const float *mean, *rstd, *gamma, *beta;
  int relu;
};
struct BnBackCh {  // per-lane constants of 4 consecutive channels
  f32x4 sc, sh, mu, rs;
};
__device__ __forceinline__ BnBackCh bnback_load(const BnBack& b, int c) {
  BnBackCh k;
  k.mu = *reinterpret_cast<const f32x4*>(b.mean + c);
  k.rs = *reinterpret_cast<const f32x4*>(b.rstd + c);
  k.sc = b.gamma ? k.rs * *reinterpret_cast<const f32x4*>(b.gamma + c) : k.rs;
  const f32x4 be = b.beta ? *reinterpret_cast<const f32x4*>(b.beta + c) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int e = 0; e < 4; ++e) k.sh[e] = be[e] - k.mu[e] * k.sc[e];
  return k;
}
// dy (as stored) -> masked g; accumulates the two sums
__device__ __forceinline__ f32x4 bnback_apply(const BnBackCh& k, int relu, const f32x4& xv, const f32x4& dy, f32x4& ss, f32x4& qq) {
  f32x4 g;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const bool on = bn_act_on(bn_affine(xv[e], k.sc[e], k.sh[e]), relu);
    g[e] = on ? dy[e] : 0.f;
    ss[e] += g[e];
    qq[e] += g[e] * ((xv[e] - k.mu[e]) * k.rs[e]);
  }
  return g;
}

