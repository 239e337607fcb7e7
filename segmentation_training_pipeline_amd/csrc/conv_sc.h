// Shared declarations of the small-channel kernels (conv_sc.hip: generic streaming / single-shot / wide forms; conv_sc_lean.hip: the
// lean streaming kernel of round 4).
#pragma once
#include "common.h"
#include <cstdlib>
#include <type_traits>
#include <utility>

struct ScArgs {
  const char* src;     // [N,Hs,Ws,CIN]  (Hs = H/2 when upsampling)
  const char* weight;  // [Cout_pad16][9*CIN]
  const float* bias;
  char* dst;           // [N,H,W,Cout]
  int N, H, W, Hs, Ws, Cout, up, accumulate, relu;
  int tiles_x, tiles_y;
  FastDiv divTx, divTy;
  float* stats;        // optional fused BatchNorm statistics [2][Cout][tiles] (or int64 slots, see stat_slots)
  int stat_slots;
  BnBack bnb;          // see stp_conv_params.bnb_x
  int sum2;            // see stp_conv_params.dst_sum2x2: dst is [N,H/2,W/2,Cout]
  uint32_t src_bytes;  // size of src (buffer descriptor of the streaming kernel's LDS-DMA)
  BnBack pbn;          // see stp_conv_params.src_bn_mean: src is normalised while it is staged (pbn.x unused)
};

// BatchNormalization (+activation) of one staged 16-byte vector: V consecutive channels with per-lane constants.  Same fma,
// activation and bf16 rounding as bn_apply_kernel, so the staged tile equals what stp_bn_apply would have stored.
template <typename T> struct ScStageBn;
template <> struct ScStageBn<float> {
  f32x4 sc, sh;
  __device__ __forceinline__ void load(const BnBack& b, int c) { const BnBackCh k = bnback_load(b, c); sc = k.sc; sh = k.sh; }
  __device__ __forceinline__ void load_tab(const float* tsc, const float* tsh, int c) {
    sc = *reinterpret_cast<const f32x4*>(tsc + c); sh = *reinterpret_cast<const f32x4*>(tsh + c);
  }
  __device__ __forceinline__ u32x4 apply(const u32x4& r, int relu) const {
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = __float_as_uint(bn_act(bn_affine(__uint_as_float(r[e]), sc[e], sh[e]), relu));
    return o;
  }
};
template <> struct ScStageBn<bf16_t> {
  f32x2 sc[4], sh[4];   // channel pairs: one v_pk_fma_f32 each
  __device__ __forceinline__ void load(const BnBack& b, int c) {
    const BnBackCh k0 = bnback_load(b, c), k1 = bnback_load(b, c + 4);
    sc[0] = f32x2{k0.sc[0], k0.sc[1]}; sc[1] = f32x2{k0.sc[2], k0.sc[3]}; sc[2] = f32x2{k1.sc[0], k1.sc[1]}; sc[3] = f32x2{k1.sc[2], k1.sc[3]};
    sh[0] = f32x2{k0.sh[0], k0.sh[1]}; sh[1] = f32x2{k0.sh[2], k0.sh[3]}; sh[2] = f32x2{k1.sh[0], k1.sh[1]}; sh[3] = f32x2{k1.sh[2], k1.sh[3]};
  }
  __device__ __forceinline__ void load_tab(const float* tsc, const float* tsh, int c) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { sc[e] = *reinterpret_cast<const f32x2*>(tsc + c + 2 * e); sh[e] = *reinterpret_cast<const f32x2*>(tsh + c + 2 * e); }
  }
  __device__ __forceinline__ u32x4 apply(const u32x4& r, int relu) const {
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      f32x2 v = {h16lo_to_f32(r[e]), h16hi_to_f32(r[e])};
      v = __builtin_elementwise_fma(v, sc[e], sh[e]);               // = bn_affine per element (single rounding)
      o[e] = pack_bf16x2(bn_act(v.x, relu), bn_act(v.y, relu));
    }
    return o;
  }
};


constexpr int SC_TH = 8, SC_TW = 32, SC_HW = SC_TW + 2, SC_HH = SC_TH + 2;

// compile-time loop: f(std::integral_constant<int, 0>) ... f(std::integral_constant<int, N - 1>)
template <int... Ks, typename F>
__device__ __forceinline__ void sc_unroll_seq(std::integer_sequence<int, Ks...>, F&& f) { (f(std::integral_constant<int, Ks>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void sc_unroll(F&& f) { sc_unroll_seq(std::make_integer_sequence<int, N>{}, f); }
#ifndef SC_RING
#define SC_RING 8
#endif

// workgroups per CU of the 16-bit instantiations with <= 16 input and output channels (what-if builds: -DSC_WPE_SMALL=3 / 5)
#ifndef SC_WPE_SMALL
#define SC_WPE_SMALL 4
#endif

static inline bool sc_stream_on() {
  static const bool on = !(getenv("STP_SC_STREAM") && atoi(getenv("STP_SC_STREAM")) == 0);
  return on;
}
static inline int sc_cu_count() {
  static const int cus = [] {
    int d = 0, n = 0;
    if (hipGetDevice(&d) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || n <= 0) n = 256;
    return n;      // MI355X (also the answer on a build host without a GPU: plan sizes must not depend on where they are computed)
  }();
  return cus;
}
// workgroups of the streaming kernel: CUs x the co-resident workgroups its launch bound leaves room for, at most one per tile
static inline int sc_stream_blocks(int dtype, int cin, int cout, int ntiles) {
  const int per_cu = cout <= 16 ? (dtype == STP_H16 && cin <= 16 ? SC_WPE_SMALL : 3) : 2;
  const int64_t b = (int64_t)sc_cu_count() * per_cu;
  return (int)(b < ntiles ? b : ntiles);
}


struct ScWgArgs {
  const char* src;  // x  [N,Hs,Ws,CIN]
  const char* dy;   // dY [N,H,W,COUT]
  float* slabs;     // [gridDim][Cout][9*CIN]
  int N, H, W, Hs, Ws, Cout, up;
  int tiles_x, tiles_y, ntiles;
  int ctot, coff;   // this source occupies channels [coff, coff+CIN) of the Ctot-channel concatenated input
  uint32_t src_bytes, dy_bytes;   // buffer descriptors of the streaming kernel's LDS-DMA
  BnBack pbn;       // see stp_wgrad_params.src_bn_mean
};

constexpr int ST_HH = (SC_TH - 1) * 2 + 7, ST_HW = 70, ST_WROW = 464, ST_HALO = ST_HH * ST_HW * 8, ST_WBYTES = 64 * ST_WROW;

struct StemArgs {
  const char* src;     // [N,H,W,4] bf16
  const char* weight;  // [64][7][8][4] bf16
  char* dst;           // [N,Ho,Wo,64] bf16
  int N, H, W, Ho, Wo, tiles_x, tiles_y;
  FastDiv divTx, divTy;
  float* stats;        // optional fused BatchNorm statistics [2][64][tiles]
  uint32_t src_bytes;  // size of src (buffer descriptor of the persistent kernel's LDS-DMA); 0 = too large
};

// conv_sc_lean.hip, persistent stem kernel: does it serve the shape (switch, even width, 32-bit offsets) / its workgroups / launch (1 = not served)
bool stem_lean_serves(int N, int H, int W);
int stem_lean_blocks(int ntiles);
int stem_lean_launch(const StemArgs& a, hipStream_t s);
// stem weight gradient (conv_stem_wgrad_lean_kernel): shape served / workgroups (= slabs [64][224]) / launch
bool stem_wg_lean_serves(int N, int H, int W, int Ho, int Wo);
int stem_wg_lean_blocks(int N, int Ho, int Wo);
int stem_wg_lean_launch(const void* src, const void* dy, float* slabs, int N, int H, int W, int Ho, int Wo, hipStream_t s);

// conv_sc_lean.hip: launches the lean kernel if it serves this configuration (returns STP_OK / an error), or returns 1 = "not served"
int sc_lean_launch(const ScArgs& a, int cin, int dtype, hipStream_t s);
int sc_wg_lean_launch(const ScWgArgs& a, int cin, int cout, int dtype, int blocks, hipStream_t s);
