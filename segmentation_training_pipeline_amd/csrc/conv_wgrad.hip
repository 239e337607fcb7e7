// Convolution weight gradient as a pixel-reduction GEMM on MFMA (gfx950).
//
//   dW[co][k] = sum_p dY[p][co] * V[p][k]      p = (n,ho,wo),  k = (kh*KW + kw)*Ctot + c
//
// Both operands live in HBM with the reduction index (pixels) as the SLOW dimension and
// channels contiguous (NHWC), the opposite of what an MFMA fragment wants (8 consecutive
// reduction elements per lane).  Tiles are therefore staged pixel-major in LDS and
//   * bf16: fetched with ds_read_b64_tr_b16, the gfx950 transpose read: a 16-lane group reads a
//     [4 pixels][16 channels] block and lane i receives channel i of the 4 pixels;
//   * fp32: v_mfma_f32_16x16x4_f32 takes ONE element per lane, so plain ds_read_b32 suffices.
// Only the (lane, slot) -> pixel assignment has to agree between the A and B fragments; it is
// chosen so that a 32-lane access touches 8 consecutive pixel rows, which together with a
// 32-byte-unit XOR swizzle makes the transpose reads bank-conflict free.
// The im2col operand V is gathered on the fly exactly as in conv_igemm.hip (nearest-2x /
// zero-insert / concat folded in).  Pixels are partitioned over `splits` slabs (split-K) that
// a second kernel sums in fixed order: deterministic, no atomics.
#include "common.h"

typedef short s16x4 __attribute__((ext_vector_type(4)));

struct WgradArgs {
  const char* src0;
  const char* src1;
  const char* dy;
  float* out;  // slabs [splits][Cout][K]
  int N, Hs0, Ws0, Hv, Wv, C0, C1, Ctot, mode;
  int KH, KW, stride, pad, Ho, Wo, Cout;
  int K, P, HoWo;
  int ntile_m, ntile_n, steps_per_split, nsteps;
  FastDiv divC, divKW, divHoWo, divWo;
};

// byte address of (row, byte-in-row) in a pixel-major tile whose rows are ROWB bytes
template <int ROWB> __device__ __forceinline__ int tile_addr(int row, int byte) {
  constexpr int U = ROWB / 32;
  constexpr int R = U >= 8 ? 1 : 8 / U;
  constexpr int M = (U >= 8 ? 8 : U) - 1;
  const int s = (row / R) & M;
  return row * ROWB + (byte ^ (s << 5));
}

template <typename T, int BM, int BN, int WM, int WN, bool C4>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradArgs a) {
  static_assert(WM * WN == 4, "4 waves");
  constexpr int SZ = (int)sizeof(T);
  constexpr int VEC = Elem<T>::VEC;
  constexpr int PK = 128 / SZ;  // pixels per step: 64 bf16 / 32 fp32
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
  constexpr int ROWA = BM * SZ, ROWB = BN * SZ;
  constexpr int VPRA = BM / VEC, VPRB = BN / VEC;       // 16-byte vectors per pixel row
  constexpr int NVA = (PK * VPRA + 255) / 256, NVB = (PK * VPRB + 255) / 256;  // vectors per thread
  constexpr int STAGE = PK * (ROWA + ROWB);

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;

  const int tiles = a.ntile_m * a.ntile_n;
  const int split = blockIdx.x / tiles;
  const int t = blockIdx.x - split * tiles;
  const int tile_m = t % a.ntile_m, tile_n = t / a.ntile_m;
  const int cout0 = tile_m * BM, k0 = tile_n * BN;
  const int step0 = split * a.steps_per_split;
  const int step1 = min(step0 + a.steps_per_split, a.nsteps);

  // ---- fixed per-thread column metadata ---------------------------------------------------
  const int colA = (tid % VPRA) * VEC;  // cout offset within tile
  const int colB = (tid % VPRB) * VEC;  // k offset within tile
  const int rowA0 = tid / VPRA, rowB0 = tid / VPRB;
  const bool coA_ok = (cout0 + colA) < a.Cout;
  const int kB = k0 + colB;
  const bool kB_ok = kB < a.K;
  int kh, kw, ci, cs, mode, Hs, Ws;
  const char* base;
  if constexpr (C4) {
    const uint32_t pos = (uint32_t)kB >> 2;
    kh = (int)fdiv(pos, a.divKW);
    kw = (int)pos - kh * a.KW;
    ci = 0; cs = 4; mode = 0; Hs = a.Hs0; Ws = a.Ws0; base = a.src0;
  } else {
    const uint32_t pos = fdiv((uint32_t)kB, a.divC);
    ci = kB - (int)pos * a.Ctot;
    kh = (int)fdiv(pos, a.divKW);
    kw = (int)pos - kh * a.KW;
    const bool first = ci < a.C0;
    base = first ? a.src0 : a.src1;
    cs = first ? a.C0 : a.C1;
    mode = first ? a.mode : 0;
    Hs = first ? a.Hs0 : a.Hv;
    Ws = first ? a.Ws0 : a.Wv;
    if (!first) ci -= a.C0;
  }

  u32x4 ra[NVA], rb[NVB];

  auto load_tile = [&](int step) {
    const int p0 = step * PK;
#pragma unroll
    for (int i = 0; i < NVA; ++i) {
      const int row = rowA0 + i * (256 / VPRA);
      const int p = p0 + row;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (coA_ok && row < PK && p < a.P)
        v = *reinterpret_cast<const u32x4*>(a.dy + ((size_t)p * a.Cout + cout0 + colA) * SZ);
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NVB; ++i) {
      const int row = rowB0 + i * (256 / VPRB);
      const int p = p0 + row;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (kB_ok && row < PK && p < a.P) {
        const uint32_t n = fdiv((uint32_t)p, a.divHoWo);
        const uint32_t rem = (uint32_t)p - n * (uint32_t)a.HoWo;
        const uint32_t ho = fdiv(rem, a.divWo);
        const uint32_t wo = rem - ho * (uint32_t)a.Wo;
        const int hv = (int)ho * a.stride - a.pad + kh;
        const int wv = (int)wo * a.stride - a.pad + kw;
        if constexpr (C4) {
          if ((unsigned)hv < (unsigned)a.Hv) {
            const char* rowp = base + ((size_t)((int)n * Hs + hv) * Ws) * (4 * SZ);
            if ((unsigned)wv < (unsigned)a.Wv) {
              u32x2 q = *reinterpret_cast<const u32x2*>(rowp + (size_t)wv * (4 * SZ));
              v.x = q.x; v.y = q.y;
            }
            if ((unsigned)(wv + 1) < (unsigned)a.Wv) {
              u32x2 q = *reinterpret_cast<const u32x2*>(rowp + (size_t)(wv + 1) * (4 * SZ));
              v.z = q.x; v.w = q.y;
            }
          }
        } else {
          bool ok = (unsigned)hv < (unsigned)a.Hv && (unsigned)wv < (unsigned)a.Wv;
          if (mode == STP_SRC_ZEROINS2X) ok = ok && (((hv | wv) & 1) == 0);
          const int hs = mode ? (hv >> 1) : hv, ws = mode ? (wv >> 1) : wv;
          if (ok) v = *reinterpret_cast<const u32x4*>(base + (((size_t)((int)n * Hs + hs) * Ws + ws) * cs + ci) * SZ);
        }
      }
      rb[i] = v;
    }
  };

  auto store_tile = [&](int buf) {
    char* sa = smem + buf * STAGE;
    char* sb = sa + PK * ROWA;
#pragma unroll
    for (int i = 0; i < NVA; ++i) {
      const int row = rowA0 + i * (256 / VPRA);
      if (row < PK) *reinterpret_cast<u32x4*>(sa + tile_addr<ROWA>(row, colA * SZ)) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < NVB; ++i) {
      const int row = rowB0 + i * (256 / VPRB);
      if (row < PK) *reinterpret_cast<u32x4*>(sb + tile_addr<ROWB>(row, colB * SZ)) = rb[i];
    }
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int lr = lane & 15, lg = lane >> 4;
  auto compute = [&](int buf) {
    const char* sa = smem + buf * STAGE;
    const char* sb = sa + PK * ROWA;
    const int ca = (wm * (BM / WM)) * SZ, cb = (wn * (BN / WN)) * SZ;  // wave column origin, bytes
    if constexpr (sizeof(T) == 2) {
      // 64 pixels per step = 2 MFMA k-steps of 32 pixels.  Lane group g owns pixels
      // {4g..4g+3} and {16+4g..16+4g+3} of the 32: two transpose reads of a [4][16] block.
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        u32x4 fa[TM], fb[TN];
        const int prow = c * 32 + lg * 4 + (lr >> 2);
        const int qb = (lr & 3) * 8;  // this lane's 8-byte quad inside the 32-byte block
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int byte = ca + i * 32 + qb;
          s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4*)(sa + tile_addr<ROWA>(prow, byte)));
          s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4*)(sa + tile_addr<ROWA>(prow + 16, byte)));
          u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
          fa[i] = u32x4{l2.x, l2.y, h2.x, h2.y};
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int byte = cb + j * 32 + qb;
          s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4*)(sb + tile_addr<ROWB>(prow, byte)));
          s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4*)(sb + tile_addr<ROWB>(prow + 16, byte)));
          u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
          fb[j] = u32x4{l2.x, l2.y, h2.x, h2.y};
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[i]),
                                                                __builtin_bit_cast(bf16x8, fb[j]), acc[i][j], 0, 0, 0);
      }
    } else {
      // fp32: 32 pixels per step = 8 MFMA k-steps of 4 pixels; lane group g owns pixel 4s+g.
#pragma unroll
      for (int s = 0; s < PK / 4; ++s) {
        const int prow = s * 4 + lg;
        float fa[TM], fb[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
          fa[i] = *reinterpret_cast<const float*>(sa + tile_addr<ROWA>(prow, ca + (i * 16 + lr) * 4));
#pragma unroll
        for (int j = 0; j < TN; ++j)
          fb[j] = *reinterpret_cast<const float*>(sb + tile_addr<ROWB>(prow, cb + (j * 16 + lr) * 4));
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
      }
    }
  };

  if (step0 < step1) {
    load_tile(step0);
    store_tile(0);
    __syncthreads();
    for (int st = step0; st < step1; ++st) {
      const int cur = (st - step0) & 1;
      if (st + 1 < step1) load_tile(st + 1);
      compute(cur);
      if (st + 1 < step1) store_tile(cur ^ 1);
      __syncthreads();
    }
  }

  // ---- write the fp32 slab (zeros if this split had no pixels) ---------------------------
  float* out = a.out + (size_t)split * a.Cout * a.K;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int kc = k0 + wn * (BN / WN) + j * 16 + lr;
      const int co = cout0 + wm * (BM / WM) + i * 16 + lg * 4;
      if (kc >= a.K) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (co + r < a.Cout) out[(size_t)(co + r) * a.K + kc] = acc[i][j][r];
    }
  }
}

// dw[i] (+)= sum_k slabs[k][i], 4 floats per thread (count is a multiple of 4: Cout*K with K % 4 == 0)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ dw, int64_t count,
                                                           int splits, int accumulate) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= count) return;
  f32x4 s = *reinterpret_cast<const f32x4*>(slabs + i);
  for (int k = 1; k < splits; ++k) s += *reinterpret_cast<const f32x4*>(slabs + (size_t)k * count + i);
  if (accumulate) s += *reinterpret_cast<const f32x4*>(dw + i);
  *reinterpret_cast<f32x4*>(dw + i) = s;
}

struct WgradPlan {
  int tile, bm, bn, ntile_m, ntile_n, splits, steps_per_split, nsteps;
};

static WgradPlan plan_wgrad(const stp_wgrad_params* p) {
  WgradPlan w;
  const int K = p->KH * p->KW * (p->C0 + p->C1);
  const int pk = p->dtype == STP_BF16 ? 64 : 32;
  const int64_t P = (int64_t)p->N * p->Ho * p->Wo;
  if (p->Cout <= 16) { w.tile = 4; w.bm = 16; w.bn = 256; }
  else if (p->Cout <= 32) { w.tile = 3; w.bm = 32; w.bn = 256; }
  else if (p->Cout <= 64) { w.tile = 2; w.bm = 64; w.bn = 128; }
  else { w.tile = 1; w.bm = 128; w.bn = 128; }
  w.ntile_m = ceil_div(p->Cout, w.bm);
  w.ntile_n = ceil_div(K, w.bn);
  w.nsteps = ceil_div(P, pk);
  const int tiles = w.ntile_m * w.ntile_n;
  int splits = p->splits;
  if (splits <= 0) {
    splits = ceil_div(1024, tiles);                   // ~4 blocks per CU
    const int max_by_steps = w.nsteps / 8 > 0 ? w.nsteps / 8 : 1;  // keep >= 8 steps per split
    if (splits > max_by_steps) splits = max_by_steps;
    if (splits > 256) splits = 256;
  }
  if (splits > w.nsteps) splits = w.nsteps;
  if (splits < 1) splits = 1;
  w.steps_per_split = ceil_div(w.nsteps, splits);
  w.splits = ceil_div(w.nsteps, w.steps_per_split);
  return w;
}

extern "C" size_t stp_conv2d_wgrad_workspace_bytes(const stp_wgrad_params* p) {
  if (!p) return 0;
  const WgradPlan w = plan_wgrad(p);
  const size_t K = (size_t)p->KH * p->KW * (p->C0 + p->C1);
  return (size_t)w.splits * p->Cout * K * sizeof(float);
}

template <typename T, int BM, int BN, int WM, int WN, bool C4>
static int launch_wgrad(WgradArgs& a, int splits, hipStream_t s) {
  constexpr int PK = 128 / (int)sizeof(T);
  const size_t lds = 2 * PK * (BM + BN) * sizeof(T);
  auto kern = conv_wgrad_kernel<T, BM, BN, WM, WN, C4>;
  if (lds > 64 * 1024) {
    static bool attr_set = false;
    if (!attr_set) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return STP_E_LAUNCH;
      attr_set = true;
    }
  }
  hipLaunchKernelGGL(kern, dim3(a.ntile_m * a.ntile_n * splits), dim3(256), lds, s, a);
  STP_LAUNCH_CHECK();
  return STP_OK;
}

template <typename T, bool C4>
static int launch_wgrad_tile(WgradArgs& a, const WgradPlan& w, hipStream_t s) {
  switch (w.tile) {
    case 1: return launch_wgrad<T, 128, 128, 2, 2, C4>(a, w.splits, s);
    case 2: return launch_wgrad<T, 64, 128, 1, 4, C4>(a, w.splits, s);
    case 3: return launch_wgrad<T, 32, 256, 1, 4, C4>(a, w.splits, s);
    case 4: return launch_wgrad<T, 16, 256, 1, 4, C4>(a, w.splits, s);
    default: return STP_E_BADARG;
  }
}

extern "C" int stp_conv2d_wgrad(const stp_wgrad_params* p, void* workspace, size_t workspace_bytes, void* stream) {
  if (!p || !p->src0 || !p->dy || !p->dw || !workspace) return STP_E_BADARG;
  if (p->dtype != STP_F32 && p->dtype != STP_BF16) return STP_E_BADARG;
  const int vec = p->dtype == STP_BF16 ? 8 : 4;
  const bool c4 = (p->dtype == STP_BF16) && p->C0 == 4 && p->C1 == 0;
  if (c4) {
    if ((p->KW & 1) || p->src0_mode != STP_SRC_DIRECT) return STP_E_BADARG;
  } else if ((p->C0 % vec) || (p->C1 % vec)) {
    return STP_E_BADARG;
  }
  if (p->Cout % vec) return STP_E_BADARG;
  if (p->C1 > 0 && !p->src1) return STP_E_BADARG;
  if (stp_conv2d_wgrad_workspace_bytes(p) > workspace_bytes) return STP_E_WORKSPACE;
  const WgradPlan w = plan_wgrad(p);
  WgradArgs a;
  a.src0 = (const char*)p->src0; a.src1 = (const char*)p->src1; a.dy = (const char*)p->dy; a.out = (float*)workspace;
  a.N = p->N; a.Hs0 = p->Hs0; a.Ws0 = p->Ws0; a.Hv = p->Hv; a.Wv = p->Wv; a.C0 = p->C0; a.C1 = p->C1;
  a.Ctot = p->C0 + p->C1; a.mode = p->src0_mode;
  a.KH = p->KH; a.KW = p->KW; a.stride = p->stride; a.pad = p->pad; a.Ho = p->Ho; a.Wo = p->Wo; a.Cout = p->Cout;
  a.K = p->KH * p->KW * a.Ctot;
  const int64_t P = (int64_t)p->N * p->Ho * p->Wo;
  if (P >= (1ll << 31)) return STP_E_BADARG;
  a.P = (int)P; a.HoWo = p->Ho * p->Wo;
  a.ntile_m = w.ntile_m; a.ntile_n = w.ntile_n; a.steps_per_split = w.steps_per_split; a.nsteps = w.nsteps;
  a.divC = make_fastdiv((uint32_t)a.Ctot); a.divKW = make_fastdiv((uint32_t)a.KW);
  a.divHoWo = make_fastdiv((uint32_t)a.HoWo); a.divWo = make_fastdiv((uint32_t)a.Wo);
  hipStream_t s = (hipStream_t)stream;
  int rc;
  if (p->dtype == STP_BF16) rc = c4 ? launch_wgrad_tile<bf16_t, true>(a, w, s) : launch_wgrad_tile<bf16_t, false>(a, w, s);
  else rc = launch_wgrad_tile<float, false>(a, w, s);
  if (rc != STP_OK) return rc;
  const int64_t count = (int64_t)p->Cout * a.K;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(ceil_div(count, 1024)), dim3(256), 0, s, (const float*)workspace, p->dw,
                     count, w.splits, p->accumulate);
  STP_LAUNCH_CHECK();
  return STP_OK;
}
