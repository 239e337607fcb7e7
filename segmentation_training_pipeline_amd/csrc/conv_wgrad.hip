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
#include <cstdlib>
#include <cstring>
#include <vector>
#include <type_traits>

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
  int xcd;   // XCD-contiguous block order
  int rowu;  // every pixel step lies inside one image and starts on an output-row boundary pattern (see ROWU)
  uint32_t bytes0, bytes1, bytesdy;
  FastDiv divC, divKW, divHoWo, divWo;
  BnBack pbn;   // row-of-taps kernel: src0 is the tensor BEFORE a BatchNormalization(+activation), normalised in LDS (pbn.x unused)
};

#define STP_OOB 0x80000000u  // buffer voffset beyond any descriptor -> the load returns 0
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// byte address of (row, byte-in-row) in a pixel-major tile whose rows are ROWB bytes
template <int ROWB> __device__ __forceinline__ int tile_addr(int row, int byte) {
  constexpr int U = ROWB / 32;
  constexpr int R = U >= 8 ? 1 : 8 / U;
  constexpr int M = (U >= 8 ? 8 : U) - 1;
  const int s = (row / R) & M;
  return row * ROWB + (byte ^ (s << 5));
}

// One pixel-step of MFMAs for this wave (transpose reads for bf16, scalar reads for fp32).
template <typename T, int BM, int BN, int WM, int WN>
__device__ __forceinline__ void wgrad_compute(const char* sa, int wm, int wn, int lr, int lg,
                                              f32x4 (&acc)[BM / WM / 16][BN / WN / 16]) {
  constexpr int SZ = (int)sizeof(T);
  constexpr int PK = 128 / SZ;
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
  constexpr int ROWA = BM * SZ, ROWB = BN * SZ;
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
            acc[i][j] = mfma16_16x16x32(fa[i], fb[j], acc[i][j]);
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
  }

template <int BM, int BN, int WM, int WN>
__device__ __forceinline__ void wgrad_write_slab(const WgradArgs& a, int split, int cout0, int k0, int wm, int wn, int lr, int lg,
                                                 f32x4 (&acc)[BM / WM / 16][BN / WN / 16]) {
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
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
  const int bid = a.xcd ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;  // the tiles of one split (same pixels) share an XCD's L2
  const int split = bid / tiles;
  const int t = bid - split * tiles;
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
  auto compute = [&](int buf) { wgrad_compute<T, BM, BN, WM, WN>(smem + buf * STAGE, wm, wn, lr, lg, acc); };

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

  wgrad_write_slab<BM, BN, WM, WN>(a, split, cout0, k0, wm, wn, lr, lg, acc);
}


// ------------------------------------------------------------------------------------------------
// Direct-to-LDS variant: every 16-byte vector of both tiles is fetched with `buffer_load_dwordx4 ... lds`
// (32-bit offsets, out-of-range -> 0 for padding / tails), STAGES-deep ring with counted vmcnt.  The LDS
// image of an LDS-DMA is lane-linear, so the 32-byte-unit XOR swizzle is applied to the SOURCE column:
// a thread owns a fixed PHYSICAL 16-byte slot and fetches the logical column that lives there (the
// swizzle key only depends on row&7 and every pass advances the row by a multiple of 8).
//
// ROWU ("uniform rows"): when Wo % PK == 0, or PK % Wo == 0 and Ho*Wo % PK == 0 (every power-of-two feature map), the
// PK pixels of a step are (ho0 + r / Wo, wo0 + r % Wo) of ONE image with wave-uniform (n, ho0, wo0): the two
// divisions per gathered row become per-thread constants plus scalar arithmetic - the address VALU work was
// what bounded this kernel (scratch/whatif_bench.py: loads-only took 80% of the full time).
template <typename T, int BM, int BN, int WM, int WN, int STAGES, bool ROWU>
__global__ __launch_bounds__(256) void conv_wgrad_dma_kernel(const WgradArgs a) {
  static_assert(WM * WN == 4 && STAGES >= 2, "config");
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int SZ = (int)sizeof(T);
  constexpr int PK = 128 / SZ;
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
  constexpr int ROWA = BM * SZ, ROWB = BN * SZ;
  constexpr int VPRA = ROWA / 16, VPRB = ROWB / 16;
  constexpr int RPA = 256 / VPRA, RPB = 256 / VPRB;            // rows per pass
  constexpr int NVA = (PK + RPA - 1) / RPA, NVB = (PK + RPB - 1) / RPB;
  static_assert((NVA == 1 || RPA % 8 == 0) && (NVB == 1 || RPB % 8 == 0), "swizzle key must not change between passes");
  constexpr int STAGE = PK * (ROWA + ROWB);
  constexpr int L = NVA + NVB;
  constexpr int DUMP = STAGES * STAGE;  // 4 KiB sink for the lane groups that have no row in a pass

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  const int tiles = a.ntile_m * a.ntile_n;
  const int bid = a.xcd ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;  // the tiles of one split (same pixels) share an XCD's L2
  const int split = bid / tiles;
  const int t = bid - split * tiles;
  const int tile_m = t % a.ntile_m, tile_n = t / a.ntile_m;
  const int cout0 = tile_m * BM, k0 = tile_n * BN;
  const int step0 = split * a.steps_per_split;
  const int step1 = min(step0 + a.steps_per_split, a.nsteps);

  const __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, a.bytesdy, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)a.src0, 0, a.bytes0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.src1 ? a.src1 : a.src0), 0, a.src1 ? a.bytes1 : 0u, 0x00020000);

  // fixed per-thread columns (logical column stored at this thread's physical slot)
  const int rowA0 = tid / VPRA, rowB0 = tid / VPRB;
  const int colA = (tile_addr<ROWA>(rowA0, (tid % VPRA) * 16) - rowA0 * ROWA) / SZ;
  const int colB = (tile_addr<ROWB>(rowB0, (tid % VPRB) * 16) - rowB0 * ROWB) / SZ;
  const bool coA_ok = (cout0 + colA) < a.Cout;
  const int kB = k0 + colB;
  const bool kB_ok = kB < a.K;
  const uint32_t pos = fdiv((uint32_t)kB, a.divC);
  int ci = kB - (int)pos * a.Ctot;
  const int kh = (int)fdiv(pos, a.divKW);
  const int kw = (int)pos - kh * a.KW;
  const bool first = ci < a.C0;
  if (!first) ci -= a.C0;
  const int cs = first ? a.C0 : a.C1;
  const int sh = (first && a.mode) ? 1 : 0;
  const bool zins = first && a.mode == STP_SRC_ZEROINS2X;
  const int Hs = first ? a.Hs0 : a.Hv, Ws = first ? a.Ws0 : a.Wv;

  // ROWU: per-thread constants of its rows
  int hc[NVB], wc[NVB];
  uint32_t rowa[NVA];
  const uint32_t pixb = (uint32_t)cs * SZ, imgb = (uint32_t)Hs * (uint32_t)Ws * pixb, cib = (uint32_t)ci * SZ;
  if constexpr (ROWU) {
#pragma unroll
    for (int i = 0; i < NVB; ++i) {
      const int row = rowB0 + i * RPB;
      const int dho = a.Wo >= PK ? 0 : row / a.Wo;
      const int dwo = a.Wo >= PK ? row : row - dho * a.Wo;
      hc[i] = dho * a.stride - a.pad + kh;
      wc[i] = dwo * a.stride - a.pad + kw;
    }
#pragma unroll
    for (int i = 0; i < NVA; ++i) rowa[i] = ((uint32_t)(rowA0 + i * RPA) * (uint32_t)a.Cout + (uint32_t)(cout0 + colA)) * SZ;
  }

  auto issue_tile = [&](int step, int buf) {
    const int p0 = step * PK;
    char* sa = smem + buf * STAGE;
    char* sb = sa + PK * ROWA;
    if constexpr (ROWU) {
      // wave-uniform decomposition of the step's first pixel (P % PK == 0 here: no ragged tail)
      const uint32_t n = fdiv((uint32_t)p0, a.divHoWo);
      const uint32_t rem = (uint32_t)p0 - n * (uint32_t)a.HoWo;
      const uint32_t ho0 = fdiv(rem, a.divWo);
      const uint32_t wo0 = rem - ho0 * (uint32_t)a.Wo;
      const int hs = (int)ho0 * a.stride, wsb = (int)wo0 * a.stride;
      const uint32_t abase = (uint32_t)p0 * (uint32_t)a.Cout * SZ;
      const uint32_t nb = n * imgb + cib;
#pragma unroll
      for (int i = 0; i < NVA; ++i) {
        const bool act = (i * 256 + wave * 64) / VPRA < PK;
        const bool rok = NVA * RPA == PK || (rowA0 + i * RPA) < PK;
        char* dst = act ? sa + (i * 256 + wave * 64) * 16 : smem + DUMP + wave * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsd, (__attribute__((address_space(3))) void*)dst, 16,
                                                 (coA_ok && rok) ? abase + rowa[i] : STP_OOB, 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < NVB; ++i) {
        const bool act = (i * 256 + wave * 64) / VPRB < PK;
        const bool rok = NVB * RPB == PK || (rowB0 + i * RPB) < PK;
        const int hv = hs + hc[i], wv = wsb + wc[i];
        bool ok = kB_ok && rok && (unsigned)hv < (unsigned)a.Hv && (unsigned)wv < (unsigned)a.Wv;
        if (zins) ok = ok && (((hv | wv) & 1) == 0);
        const uint32_t off = nb + ((uint32_t)(hv >> sh) * (uint32_t)Ws + (uint32_t)(wv >> sh)) * pixb;
        char* dst = act ? sb + (i * 256 + wave * 64) * 16 : smem + DUMP + wave * 1024;
        if (first) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, (__attribute__((address_space(3))) void*)dst, 16, ok ? off : STP_OOB, 0, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (__attribute__((address_space(3))) void*)dst, 16, ok ? off : STP_OOB, 0, 0, 0);
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < NVA; ++i) {
      const int row = rowA0 + i * RPA;
      const int p = p0 + row;
      const bool act = (i * 256 + wave * 64) / VPRA < PK;   // wave-uniform: does this pass have rows for this wave?
      const uint32_t off = (coA_ok && row < PK && p < a.P) ? ((uint32_t)p * (uint32_t)a.Cout + (uint32_t)(cout0 + colA)) * SZ : STP_OOB;
      char* dst = act ? sa + (i * 256 + wave * 64) * 16 : smem + DUMP + wave * 1024;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsd, (__attribute__((address_space(3))) void*)dst, 16, off, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NVB; ++i) {
      const int row = rowB0 + i * RPB;
      const int p = p0 + row;
      const bool act = (i * 256 + wave * 64) / VPRB < PK;
      uint32_t off = STP_OOB;
      if (kB_ok && row < PK && p < a.P) {
        const uint32_t n = fdiv((uint32_t)p, a.divHoWo);
        const uint32_t rem = (uint32_t)p - n * (uint32_t)a.HoWo;
        const uint32_t ho = fdiv(rem, a.divWo);
        const uint32_t wo = rem - ho * (uint32_t)a.Wo;
        const int hv = (int)ho * a.stride - a.pad + kh;
        const int wv = (int)wo * a.stride - a.pad + kw;
        bool ok = (unsigned)hv < (unsigned)a.Hv && (unsigned)wv < (unsigned)a.Wv;
        if (zins) ok = ok && (((hv | wv) & 1) == 0);
        if (ok) off = ((n * (uint32_t)Hs + (uint32_t)(hv >> sh)) * (uint32_t)Ws + (uint32_t)(wv >> sh)) * (uint32_t)cs * SZ + (uint32_t)ci * SZ;
      }
      char* dst = act ? sb + (i * 256 + wave * 64) * 16 : smem + DUMP + wave * 1024;
      if (first) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, (__attribute__((address_space(3))) void*)dst, 16, off, 0, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (__attribute__((address_space(3))) void*)dst, 16, off, 0, 0, 0);
    }
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int lr = lane & 15, lg = lane >> 4;

  const int nst = step1 - step0;
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s)
    if (s < nst) issue_tile(step0 + s, s);
  int buf = 0, nbuf = STAGES - 1;
  for (int st = 0; st < nst; ++st) {
    const int ahead = nst - 1 - st;
    if (STAGES >= 3 && ahead >= STAGES - 2) wait_vmcnt<(STAGES >= 3 ? (STAGES - 2) : 0) * L>();
    else if (STAGES >= 4 && ahead == 1) wait_vmcnt<L>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
#if !defined(STP_EXP) || STP_EXP != 2  // what-if builds (scratch/exp_build.sh): 1 = no MFMA work, 2 = no loads in the loop
    if (st + STAGES - 1 < nst) issue_tile(step0 + st + STAGES - 1, nbuf);
#endif
#if !defined(STP_EXP) || STP_EXP != 1
    wgrad_compute<T, BM, BN, WM, WN>(smem + buf * STAGE, wm, wn, lr, lg, acc);
#endif
    buf = (buf + 1 == STAGES) ? 0 : buf + 1;
    nbuf = (nbuf + 1 == STAGES) ? 0 : nbuf + 1;
  }
  wgrad_write_slab<BM, BN, WM, WN>(a, split, cout0, k0, wm, wn, lr, lg, acc);
#endif
}

// ------------------------------------------------------------------------------------------------
// ROW-OF-TAPS variant (bf16, 3x3 / stride 1 / pad 1, source channel counts multiples of 64, uniform-row feature maps).
// The DMA kernel above is bound by the L2 -> LDS path (what-if build: loads only = 80 % of the launch; 64 FLOP per staged
// byte): every k-tile re-stages dY, and the im2col columns of the three taps of a kernel row are three copies of the same
// pixels shifted by one.  Here a workgroup's k-tile is ONE KERNEL ROW of a 64-channel input block - 192 columns
// (kw = 0..2) x (ci 0..63) - and the operand tile of a 64-pixel step is the step's input pixels WITH THEIR HALO: rows x
// (seg + 2) pixels of 128 bytes (seg = min(64, Wo) pixels of `rows` image rows).  The B fragment of tap kw is the same LDS
// block read one pixel row further: 28 KB staged per 2 x 128 x 192 x 64 FLOP = 112 FLOP per byte, 7 LDS-DMA instructions
// per thread per 48 MFMAs (8 per 32 above).  LDS rows keep the 32-byte-unit XOR swizzle of tile_addr<> (a transpose read
// touches 8 consecutive pixel rows from any start: conflict-free for every kw).
//   tile: BM output channels x 192 columns of kernel row kh, input-channel block cib   (tile_n = kh * (C0 / 64) + cib)
//   dW column of (kw, ci): (kh * 3 + kw) * Ctot + cib * 64 + ci.  Two concatenated sources (each a multiple of 64 channels, the first optionally
//   nearest-2x upsampled: the decoder's UpSampling2D + Concatenate) are handled per 64-channel block: a block lies in one source.
// PBN (fused producer BatchNormalization of src0) is a compile-time parameter of the body: 20 VGPRs the plain launches do not carry.
// which instances run on v_mfma_f32_32x32x16 (STP_ROW_M32=0: a what-if build on the 16 x 16 x 32 form everywhere)
#ifndef STP_ROW_M32
#define STP_ROW_M32 0
#endif
template <int BM, int WM, int WN> struct RowM32 { static constexpr bool value = STP_ROW_M32 && BM == 128 && WM == 2 && WN == 2; };

// element e (float4) of a tile's fragment-major slab -> first of its 4 output channels (relative to the tile) and its column (0..191)
template <int BM, int WM, int WN>
__device__ __forceinline__ void row_slab_decode(int e, int& co, int& col) {
  const int lane = e & 63;
  int q = e >> 6;
  if constexpr (RowM32<BM, WM, WN>::value) {
    constexpr int TM = BM / WM / 32, TN = 192 / WN / 32;
    const int qq = q & 3; q >>= 2;
    const int ij = q % (TM * TN), wave = q / (TM * TN);
    const int i = ij / TN, j = ij - i * TN, wm = wave / WN, wn = wave % WN;
    co = wm * (BM / WM) + i * 32 + 8 * qq + 4 * (lane >> 5);
    col = wn * (192 / WN) + j * 32 + (lane & 31);
  } else {
    constexpr int TM = BM / WM / 16, TN = 192 / WN / 16;
    const int ij = q % (TM * TN), wave = q / (TM * TN);
    const int i = ij / TN, j = ij - i * TN, wm = wave / WN, wn = wave % WN;
    co = wm * (BM / WM) + i * 16 + (lane >> 4) * 4;
    col = wn * (192 / WN) + j * 16 + (lane & 15);
  }
}

// (t, step0, step1, out_tile): the tile of the layer (tile_m fastest), the range of 64-pixel steps summed and the fragment-major
// partial slab [wave][i][j][lane] float4 that receives the sums - the single-layer kernel derives them from the block id, the
// grouped kernel (below) from its work list.
template <int BM, int WM, int WN, int STAGES, bool PBN>
__device__ __forceinline__ void conv_wgrad_row_body(const WgradArgs& a, const int t, const int step0, const int step1, f32x4* const out_tile) {
  static_assert(WM * WN == 4, "4 waves");
#if defined(__HIP_DEVICE_COMPILE__)
  typedef bf16_t T;
  constexpr int SZ = 2, PK = 64, BN = 192;
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
  constexpr int ROWA = BM * SZ, ROWB = 128;                    // LDS row bytes: dY pixel / halo pixel (64 input channels)
  constexpr int VPRA = ROWA / 16, RPA = 256 / VPRA, NVA = PK / RPA;
  constexpr int HROWS = 72, NVB = 3;                           // halo rows: rows x (seg + 2) <= 72; passes of 32 rows, the third has 8
  constexpr int STAGE = PK * ROWA + HROWS * ROWB;
  constexpr int L = NVA + NVB;                                 // LDS-DMA instructions per stage: wave 0 (waves 1-3 skip the third halo pass)
  static_assert(NVA * RPA == PK && (NVA == 1 || RPA % 8 == 0), "A passes");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int lr = lane & 15, lg = lane >> 4;

  const int tile_m = t % a.ntile_m, tile_n = t / a.ntile_m;
  const int ncib = a.Ctot >> 6;
  const int kh = tile_n / ncib, cib = tile_n - kh * ncib;       // cib: 64-channel block of the CONCATENATED input
  const bool first = cib * 64 < a.C0;                            // wave-uniform: which source holds this block
  const int cs = first ? a.C0 : a.C1, cb_src = first ? cib : cib - (a.C0 >> 6);
  const int sh = (first && a.mode == STP_SRC_NEAREST2X) ? 1 : 0;  // UpSampling2D(2) folded into the gather: source pixel = (h >> 1, w >> 1)
  const int Hs = first ? a.Hs0 : a.Hv, Ws = first ? a.Ws0 : a.Wv;
  const int cout0 = tile_m * BM;

  const __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, a.bytesdy, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)(first ? a.src0 : a.src1), 0, first ? a.bytes0 : a.bytes1, 0x00020000);

  // geometry of a step: `rows` image rows of `seg` pixels (wave-uniform)
  const int seg = a.Wo >= PK ? PK : a.Wo, rows = PK / seg, hw = seg + 2;

  // ---- LDS-DMA constants.  A (dY): as in the DMA kernel.  B: halo row h = pass * 32 + tid / 8, physical slot tid & 7 -------
  const int rowA0 = tid / VPRA;
  const int colA = (tile_addr<ROWA>(rowA0, (tid % VPRA) * 16) - rowA0 * ROWA) / SZ;
  const bool coA_ok = (cout0 + colA) < a.Cout;
  uint32_t rowa[NVA];
#pragma unroll
  for (int i = 0; i < NVA; ++i) rowa[i] = ((uint32_t)(rowA0 + i * RPA) * (uint32_t)a.Cout + (uint32_t)(cout0 + colA)) * SZ;
  const int hb0 = tid >> 3;
  const uint32_t colB = (uint32_t)(tile_addr<ROWB>(hb0, (tid & 7) * 16) - hb0 * ROWB);   // logical byte column held by this slot
  int hr[NVB], hx[NVB];          // image row / column of the halo pixel relative to the step's first pixel (-1 = unused row)
#pragma unroll
  for (int i = 0; i < NVB; ++i) {
    const int h = hb0 + i * 32;
    const int r = h / hw;
    hr[i] = r < rows ? r + kh - 1 : -0x4000;
    hx[i] = h - r * hw - 1;
  }
  const uint32_t pixb = (uint32_t)cs * SZ, imgb = (uint32_t)Hs * (uint32_t)Ws * pixb, cbyte = (uint32_t)cb_src * 128u + colB;
  // fused PRODUCER BatchNormalization (+activation): the thread that DMA'd a 16-byte vector of the halo tile normalises it in LDS
  // once it has landed (own data: its vmcnt orders the read-modify-write, the step's barrier publishes it) - same fma, activation
  // and bf16 rounding as stp_bn_apply, so the operand equals the tensor that launch would have stored; padding stays zero
  const bool pbn = PBN && a.pbn.mean != nullptr;
  f32x2 psc[4], psh[4];
  if (pbn) {
    const int c0 = cb_src * 64 + (int)(colB >> 1);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float r = a.pbn.rstd[c0 + e], k = a.pbn.gamma ? r * a.pbn.gamma[c0 + e] : r;
      psc[e >> 1][e & 1] = k;
      psh[e >> 1][e & 1] = (a.pbn.beta ? a.pbn.beta[c0 + e] : 0.f) - a.pbn.mean[c0 + e] * k;
    }
  }
  auto transform_tile = [&](int step, int buf) {
    char* sb = smem + buf * STAGE + PK * ROWA;
    const int p0 = step * PK;
    const uint32_t n = fdiv((uint32_t)p0, a.divHoWo);
    const uint32_t rem = (uint32_t)p0 - n * (uint32_t)a.HoWo;
    const uint32_t ho0 = fdiv(rem, a.divWo);
    const uint32_t wo0 = rem - ho0 * (uint32_t)a.Wo;
#pragma unroll
    for (int i = 0; i < NVB; ++i) {
      if (i * 32 + wave * 8 >= HROWS) continue;                 // wave-uniform
      const int hv = (int)ho0 + hr[i], wv = (int)wo0 + hx[i];
      if (!((unsigned)hv < (unsigned)a.Hv && (unsigned)wv < (unsigned)a.Wv)) continue;
      u32x4* vp = reinterpret_cast<u32x4*>(sb + (i * 256 + tid) * 16);
      const u32x4 v = *vp;
      u32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        f32x2 t = {h16lo_to_f32(v[e]), h16hi_to_f32(v[e])};
        t = __builtin_elementwise_fma(t, psc[e], psh[e]);
        o[e] = pack_bf16x2(bn_act(t.x, a.pbn.relu), bn_act(t.y, a.pbn.relu));
      }
      *vp = o;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };

  auto issue_tile = [&](int step, int buf) {
    const int p0 = step * PK;
    char* sa = smem + buf * STAGE;
    char* sb = sa + PK * ROWA;
    const uint32_t n = fdiv((uint32_t)p0, a.divHoWo);
    const uint32_t rem = (uint32_t)p0 - n * (uint32_t)a.HoWo;
    const uint32_t ho0 = fdiv(rem, a.divWo);
    const uint32_t wo0 = rem - ho0 * (uint32_t)a.Wo;
    const uint32_t abase = (uint32_t)p0 * (uint32_t)a.Cout * SZ;
    const uint32_t nb = n * imgb + cbyte;
#pragma unroll
    for (int i = 0; i < NVA; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsd, (__attribute__((address_space(3))) void*)(sa + (i * 256 + wave * 64) * 16), 16,
                                               coA_ok ? abase + rowa[i] : STP_OOB, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < NVB; ++i) {
      const int hv = (int)ho0 + hr[i], wv = (int)wo0 + hx[i];
      const bool ok = (unsigned)hv < (unsigned)a.Hv && (unsigned)wv < (unsigned)a.Wv;
      const uint32_t off = nb + ((uint32_t)(hv >> sh) * (uint32_t)Ws + (uint32_t)(wv >> sh)) * pixb;
      if (i * 32 + wave * 8 < HROWS)     // wave-uniform: rows 72.. of the third pass do not exist
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, (__attribute__((address_space(3))) void*)(sb + (i * 256 + wave * 64) * 16), 16,
                                                 ok ? off : STP_OOB, 0, 0, 0);
    }
  };

  // ---- fragment addresses.  Pixel q of the step = image row q / seg, column q % seg -> halo row (q / seg) * hw + q % seg + kw.
  // A 16-lane group reads 4 consecutive pixels (seg >= 16: they share an image row); lane: pixel (lr >> 2), quad (lr & 3).
  const int qb = (lr & 3) * 8;
  int prowA[4], hrowB[4];      // [chunk * 2 + half]: dY row / halo row (kw = 0) of this lane's pixel
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int q = e * 16 + lg * 4 + (lr >> 2);
    prowA[e] = q;
    const int r = q / seg;
    hrowB[e] = r * hw + (q - r * seg);
  }
  int kwB[TN], cbB[TN];        // tap and byte column (64-channel block) of this wave's column blocks
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = wn * (BN / WN) + j * 16;
    kwB[j] = col >> 6;
    cbB[j] = (col & 63) * SZ + qb;
  }
  const int ca = (wm * (BM / WM)) * SZ + qb;

  // M32 (128-channel class, 2 x 2 waves of 64 x 96): v_mfma_f32_32x32x16 - half the MFMA instructions per step for the same
  // transpose reads (the operand bytes of a wave tile do not depend on the MFMA shape), which frees issue slots for the reads.
  // Operand layout of the 32 x 32 x 16 form: lane group g = lane >> 4 holds rows / columns 16 (g & 1) .. +15 and k = 8 (g >> 1) .. +7;
  // a k-step = 16 pixels: pixel 16 kk + 8 (g >> 1) + {0..3} from the first transpose read, + {4..7} from the second - the same map
  // for both operands, which is all the contraction needs.
  constexpr bool M32 = RowM32<BM, WM, WN>::value;
  constexpr int TM32 = M32 ? BM / WM / 32 : 1, TN32 = M32 ? BN / WN / 32 : 1;
  int prowA32[8], hrowB32[8], kwB32[TN32], cbB32[TN32];
  int ca32 = 0;
  if constexpr (M32) {
    const int hh = lg & 1, kh2 = lg >> 1;
#pragma unroll
    for (int e = 0; e < 8; ++e) {      // e = kk * 2 + half
      const int q = (e >> 1) * 16 + kh2 * 8 + (e & 1) * 4 + (lr >> 2);
      prowA32[e] = q;
      const int r = q / seg;
      hrowB32[e] = r * hw + (q - r * seg);
    }
#pragma unroll
    for (int j = 0; j < TN32; ++j) {
      const int col = wn * (BN / WN) + j * 32 + hh * 16;
      kwB32[j] = col >> 6;
      cbB32[j] = (col & 63) * SZ + qb;
    }
    ca32 = (wm * (BM / WM) + hh * 16) * SZ + qb;
  }

  f32x4 acc[M32 ? 1 : TM][M32 ? 1 : TN];
  f32x16 acc32[TM32][TN32];
  if constexpr (M32) {
#pragma unroll
    for (int i = 0; i < TM32; ++i)
#pragma unroll
      for (int j = 0; j < TN32; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc32[i][j][e] = 0.f;
  } else {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  auto compute = [&](const char* sa) {
    const char* sb = sa + PK * ROWA;
    if constexpr (M32) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        u32x4 fa[TM32], fb[TN32];
#pragma unroll
        for (int i = 0; i < TM32; ++i) {
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sa + tile_addr<ROWA>(prowA32[2 * kk], ca32 + i * 64)));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sa + tile_addr<ROWA>(prowA32[2 * kk + 1], ca32 + i * 64)));
          const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
          fa[i] = u32x4{l2.x, l2.y, h2.x, h2.y};
        }
#pragma unroll
        for (int j = 0; j < TN32; ++j) {
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sb + tile_addr<ROWB>(hrowB32[2 * kk] + kwB32[j], cbB32[j])));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sb + tile_addr<ROWB>(hrowB32[2 * kk + 1] + kwB32[j], cbB32[j])));
          const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
          fb[j] = u32x4{l2.x, l2.y, h2.x, h2.y};
        }
#pragma unroll
        for (int i = 0; i < TM32; ++i)
#pragma unroll
          for (int j = 0; j < TN32; ++j)
            acc32[i][j] = mfma16_32x32x16(fa[i], fb[j], acc32[i][j]);
      }
      return;
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      u32x4 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sa + tile_addr<ROWA>(prowA[2 * c], ca + i * 32)));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sa + tile_addr<ROWA>(prowA[2 * c + 1], ca + i * 32)));
        const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
        fa[i] = u32x4{l2.x, l2.y, h2.x, h2.y};
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sb + tile_addr<ROWB>(hrowB[2 * c] + kwB[j], cbB[j])));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sb + tile_addr<ROWB>(hrowB[2 * c + 1] + kwB[j], cbB[j])));
        const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
        fb[j] = u32x4{l2.x, l2.y, h2.x, h2.y};
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = mfma16_16x16x32(fa[i], fb[j], acc[i][j]);
    }
  };

  // STAGES-deep ring, one barrier per step: tile st+STAGES-1 is issued after the barrier of step st (every wave has left
  // compute(st-1), the last reader of that slot); the wait leaves the STAGES-2 younger tiles in flight
  const int nst = step1 - step0;
#pragma unroll
  for (int q = 0; q < STAGES - 1; ++q)
    if (q < nst) issue_tile(step0 + q, q);
  int buf = 0, nbuf = STAGES - 1;
  for (int st = 0; st < nst; ++st) {
    const int ahead = nst - 1 - st;      // tiles after this one
    if (STAGES >= 3 && ahead >= STAGES - 2) {
      if (wave == 0) wait_vmcnt<(STAGES - 2) * L>(); else wait_vmcnt<(STAGES - 2) * (L - 1)>();
    } else {
      wait_vmcnt<0>();
    }
    if (pbn) transform_tile(step0 + st, buf);
    __builtin_amdgcn_s_barrier();
#if !defined(STP_EXP) || STP_EXP != 2
    if (st + STAGES - 1 < nst) issue_tile(step0 + st + STAGES - 1, nbuf);
#endif
#if !defined(STP_EXP) || STP_EXP != 1
    compute(smem + buf * STAGE);
#endif
    buf = (buf + 1 == STAGES) ? 0 : buf + 1;
    nbuf = (nbuf + 1 == STAGES) ? 0 : nbuf + 1;
  }

  // ---- slab, FRAGMENT-MAJOR: [split][tile][wave][i][j][lane] float4 (the lane's 4 output channels of one column) - every store is
  // a 16-byte lane-contiguous vector (1 KB per wave instruction) instead of 4-byte stores K floats apart (measured: the scattered
  // slab write was a third of the launch); wgrad_reduce_row_kernel sums the splits in this order and scatters into dW once
  f32x4* out = out_tile + (size_t)wave * (TM * TN * 64) + lane;
  if constexpr (M32) {      // [wave][i][j][q][lane]: the lane's rows 8 q + 4 (lane >> 5) + {0..3} of column lane & 31 (row_slab_decode)
#pragma unroll
    for (int i = 0; i < TM32; ++i)
#pragma unroll
      for (int j = 0; j < TN32; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          out[((i * TN32 + j) * 4 + q) * 64] = f32x4{acc32[i][j][4 * q], acc32[i][j][4 * q + 1], acc32[i][j][4 * q + 2], acc32[i][j][4 * q + 3]};
  } else {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) out[(i * TN + j) * 64] = acc[i][j];
  }
#endif
}

// single layer: block = (split, tile), the tiles of one split (same pixels) next to each other
template <int BM, int WM, int WN, int STAGES, bool PBN>
__device__ __forceinline__ void conv_wgrad_row_single(const WgradArgs& a) {
  const int tiles = a.ntile_m * a.ntile_n;
  const int bid = a.xcd ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
  const int split = bid / tiles;
  const int t = bid - split * tiles;
  const int step0 = split * a.steps_per_split;
  const int step1 = min(step0 + a.steps_per_split, a.nsteps);
  conv_wgrad_row_body<BM, WM, WN, STAGES, PBN>(a, t, step0, step1, reinterpret_cast<f32x4*>(a.out) + (size_t)(split * tiles + t) * (BM * 192 / 4));
}
template <int BM, int WM, int WN, int STAGES>
__global__ __launch_bounds__(256) void conv_wgrad_row_kernel(const WgradArgs a) { conv_wgrad_row_single<BM, WM, WN, STAGES, false>(a); }
template <int BM, int WM, int WN, int STAGES>
__global__ __launch_bounds__(256) void conv_wgrad_row_pbn_kernel(const WgradArgs a) { conv_wgrad_row_single<BM, WM, WN, STAGES, true>(a); }

// ------------------------------------------------------------------------------------------------
// GROUPED launch (round 3).  A layer launched alone has 6-96 output tiles for 512-768 workgroup slots, so its pixel reduction was
// split 5-85 ways: ~12 steps per workgroup, a 3-stage pipeline fill and a 96 KB partial slab per 12 steps - the launch ran at
// 520-700 TFLOP/s and the slabs were 3.65x the algorithmic bytes (profiles/r02g_*).  The same kernel body with 130+ steps per
// workgroup reaches 1060-1070 TFLOP/s (scratch/r03_probe.py).  The weight gradients feed nothing but the optimizer, so the plan
// collects the layers of a stage and issues ONE launch for them: the (layer, tile, step) space of the group is laid out as a line
// (layer-major, tile, step), cut into one contiguous chunk per workgroup slot (stream-K), and a workgroup walks the segments of
// its chunk.  Every segment writes one partial slab; a tile is covered by 1-3 segments that `wgrad_group_reduce_kernel` sums in
// line order (fixed partition, fixed order: deterministic, no atomics) and scatters into dW.
struct WgLayer {
  WgradArgs a;
  float* dw;
  int accumulate, tile0, pad0, pad1;
};
struct WgSeg { int layer, tile, step0, step1, slot, pad0, pad1, pad2; };
struct WgTile { int layer, t, list0, nslots; };        // the tile's partial slabs: slot_list[list0 .. list0 + nslots), in line order
struct WgGroupHeader {
  int magic, bm, n_layers, n_segs, n_wg, n_tiles;
  int off_layers, off_segs, off_first, off_tiles;      // byte offsets from the start of the table
  int total_bytes, off_slots;
  int reduce_lanes, xcd, pbn, taps9;                    // lanes per element of the reduce launch (1 or 4); xcd: contiguous line runs per XCD;
};                                                      // pbn: some layer normalises its src0 in LDS (fused producer BatchNormalization);
                                                        // taps9: all-taps tiles (conv_wgrad_taps9_group_kernel) instead of row-of-taps tiles
#define WG_GROUP_MAGIC 0x57474733

// dword-wise copy of a descriptor through the constant address space (scalar, invariant loads)
template <typename T> __device__ __forceinline__ void load_constant(T& dst, const T* src) {
  static_assert(sizeof(T) % 4 == 0, "dwords");
  typedef const __attribute__((address_space(4))) uint32_t* cptr_t;
  const cptr_t q = (cptr_t)(reinterpret_cast<const uint32_t*>(src));
  uint32_t* d = reinterpret_cast<uint32_t*>(&dst);
#pragma unroll
  for (int i = 0; i < (int)(sizeof(T) / 4); ++i) d[i] = q[i];
}

// PBN: the instance for groups in which some layer reads the tensor BEFORE a BatchNormalization(+activation) and normalises its halo
// tile in LDS (per layer: WgradArgs::pbn.mean != nullptr) - the stp_bn_apply launch of that BatchNormalization is gone (graph.py)
template <int BM, int WM, int WN, int STAGES, bool PBN>
__device__ __forceinline__ void conv_wgrad_row_group_body(const char* __restrict__ table, f32x4* __restrict__ slabs) {
  const WgGroupHeader* const hd = reinterpret_cast<const WgGroupHeader*>(table);
  const WgLayer* const layers = reinterpret_cast<const WgLayer*>(table + hd->off_layers);
  const WgSeg* const segs = reinterpret_cast<const WgSeg*>(table + hd->off_segs);
  const int* const first = reinterpret_cast<const int*>(table + hd->off_first);
  // consecutive block ids round-robin over the XCDs: give every XCD one contiguous run of the line (the tiles of a layer read
  // the same dY / x pixels at the same time: they meet in one L2)
  // hd->xcd: 0 = identity (line neighbours on different XCDs), 1 = one contiguous run of the line per XCD, G >= 2 = runs of G line
  // neighbours per XCD, the runs dealt round-robin (G workgroups share their operands in one L2, no more: 64 workgroups fetching the
  // same lines at the same time from ONE L2 measured slower for the 512-channel layers)
  int w = blockIdx.x;
  if (hd->xcd == 1) {
    w = xcd_remap(blockIdx.x, gridDim.x);
  } else if (hd->xcd >= 2) {
    const int G = hd->xcd, nmain = (int)gridDim.x / (8 * G) * (8 * G);
    if ((int)blockIdx.x < nmain) {
      const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
      w = ((j / G) * 8 + x) * G + (j % G);
    }
  }
  const int s0 = __builtin_amdgcn_readfirstlane(first[w]), s1 = __builtin_amdgcn_readfirstlane(first[w + 1]);
  for (int s = s0; s < s1; ++s) {
    // The descriptors are read through the CONSTANT address space: invariant scalar loads the compiler may hoist out of the step
    // loop and keep in SGPRs, as it does for kernel arguments.  Through a global pointer every `memory`-clobbering s_waitcnt of the
    // loop forced a reload (s_load + lgkmcnt(0) in every step, which also drains the fragment reads: -17 % on the first build).
    WgSeg sg;
    load_constant(sg, segs + s);
    WgradArgs a;
    load_constant(a, &layers[sg.layer].a);
    if (s != s0) __syncthreads();        // the previous segment's last fragment reads precede this segment's first LDS-DMA
    conv_wgrad_row_body<BM, WM, WN, STAGES, PBN>(a, sg.tile, sg.step0, sg.step1, slabs + (size_t)sg.slot * (BM * 192 / 4));
  }
}
template <int BM, int WM, int WN, int STAGES>
__global__ __launch_bounds__(256) void conv_wgrad_row_group_kernel(const char* __restrict__ table, f32x4* __restrict__ slabs) {
  conv_wgrad_row_group_body<BM, WM, WN, STAGES, false>(table, slabs);
}
template <int BM, int WM, int WN, int STAGES>
__global__ __launch_bounds__(256) void conv_wgrad_row_group_pbn_kernel(const char* __restrict__ table, f32x4* __restrict__ slabs) {
  conv_wgrad_row_group_body<BM, WM, WN, STAGES, true>(table, slabs);
}

// ------------------------------------------------------------------------------------------------
// ALL-TAPS tile (round 4): BM output channels x 576 columns = the NINE taps of one 64-channel input block.
// The row-of-taps tile above (BM x 192: one kernel row) re-stages dY for each of the three kernel rows and the same input pixels
// three times with a one-row shift: 25 KB through L2 -> LDS per 3.1 MFLOP, and the 24-96 tiles over a pixel range each fetch their
// own copy (profiles/r03final_pmc_traffic.json: 4.85x the algorithmic bytes reach the fabric; the kernel waits on memory 42 % of
// its time).  Here a step stages dY once (64 pixels x BM channels) and ONE halo of the input block - (rows + 2) image rows of
// (seg + 2) pixels - for all nine taps: 43.5 KB (seg = 64) / 33 KB (seg = 16, 32) per 9.4 MFLOP = 1.7-2.3x fewer staged bytes per
// FLOP, and a third of the tiles per pixel range.
//   * 512 threads = 8 waves as 2 (output channels: BM / 2 each) x 4 (the four 16-channel quarters of the input block): a wave's
//     columns are (tap 0..8) x (its 16 channels) - the swizzle slot of its B reads is then a LANE CONSTANT and the tap is
//       kw -> one of 3 address registers per pixel group (the halo row shifts by one),  kh -> an INSTRUCTION IMMEDIATE:
//     the halo row pitch is padded to a multiple of 8 rows (hwp = 24 / 40 / 72 for seg = 16 / 32 / 64), so kh * hwp never moves the
//     swizzle key ((row >> 1) & 3).  12 + 4 * TM address registers serve all 36 + 4 * TM transpose reads of a step.
//   * SEG (pixels of an image row per step: min(64, Wo)) is a template parameter: pitch, immediates and the pixel -> halo row map
//     are compile-time.
//   * accumulators TM x 9 tiles of 16 x 16 (144 registers at BM = 128), one workgroup per CU, 3-stage LDS-DMA ring
//     (3 x 48 KB at BM = 128 / seg = 64), one barrier per step; every wave issues the same NVA + NPASS LDS-DMA instructions per step
//     (rows past the halo are fetched out of bounds = zeros), so the counted vmcnt is a compile-time constant.
//   * partial slab per segment: fragment-major [wave][i][tap][lane] float4, summed in line order by wgrad_group9_reduce_kernel.
// Eligibility = the row-of-taps rules (3x3 / s1 / p1, 64-channel blocks, uniform-row maps), no fused producer BatchNormalization.
template <int SEG> struct Taps9Geo {
  static constexpr int ROWS = 64 / SEG, HW = SEG + 2, HWP = (HW + 7) / 8 * 8, HR = (ROWS + 2) * HWP;
  static constexpr int NPASS = (HR + 63) / 64, LROWS = NPASS * 64;
};

template <int BM, int SEG, int STAGES, bool M32>
__device__ __forceinline__ void conv_wgrad_taps9_body(const WgradArgs& a, const int t, const int step0, const int step1, f32x4* const out_tile) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef Taps9Geo<SEG> G;
  constexpr int SZ = 2, PK = 64;
  constexpr int TM = BM / 2 / 16, TN = 9;
  constexpr int ROWA = BM * SZ, ROWB = 128;
  constexpr int VPRA = ROWA / 16, RPA = 512 / VPRA, NVA = (PK + RPA - 1) / RPA;
  constexpr int ABYTES = (PK * ROWA > NVA * 512 * 16) ? PK * ROWA : NVA * 512 * 16;     // (BM = 32: the single pass covers 128 rows)
  constexpr int STAGE = ABYTES + G::LROWS * ROWB;
  constexpr int L = NVA + G::NPASS;                               // LDS-DMA instructions per thread per step (every wave)
  static_assert(BM == 128 || BM == 64 || BM == 32, "output-channel tile");
  static_assert(!M32 || BM == 128, "the 32 x 32 x 16 form: 4 x 2 waves of 32 channels x (9 taps x 32 input channels)");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  // The thread index passes through an opaque asm: the lane constants below (fragment and LDS-DMA addresses) depend only on it and
  // on SEG, so the compiler would hoist those of ALL THREE bodies of conv_wgrad_taps9_group_kernel out of its segment loop and keep
  // them alive across it - ~50 registers that the BM = 128 instance does not have (42 spills).  Recomputed per segment instead.
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int lr = lane & 15, lg = lane >> 4;

  const int tile_m = t % a.ntile_m, cib = t / a.ntile_m;         // cib: 64-channel block of the CONCATENATED input
  const bool first = cib * 64 < a.C0;                            // wave-uniform: which source holds this block
  const int cs = first ? a.C0 : a.C1, cb_src = first ? cib : cib - (a.C0 >> 6);
  const int sh = (first && a.mode == STP_SRC_NEAREST2X) ? 1 : 0;
  const int Hs = first ? a.Hs0 : a.Hv, Ws = first ? a.Ws0 : a.Wv;
  const int cout0 = tile_m * BM;

  const __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, a.bytesdy, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)(first ? a.src0 : a.src1), 0, first ? a.bytes0 : a.bytes1, 0x00020000);

  // ---- LDS-DMA constants.  A (dY): pass i = pixel rows i * RPA + tid / VPRA.  B: LDS halo row h = pass * 64 + tid / 8 -> image row
  // h / HWP - 1, column h % HWP - 1 relative to the step's first pixel (pad columns and rows past the halo: out of bounds)
  const int rowA0 = tid / VPRA;
  const int colA = (tile_addr<ROWA>(rowA0, (tid % VPRA) * 16) - rowA0 * ROWA) / SZ;
  const bool coA_ok = (cout0 + colA) < a.Cout;
  uint32_t rowa[NVA];
#pragma unroll
  for (int i = 0; i < NVA; ++i)
    rowa[i] = (rowA0 + i * RPA) < PK ? ((uint32_t)(rowA0 + i * RPA) * (uint32_t)a.Cout + (uint32_t)(cout0 + colA)) * SZ : STP_OOB;
  const int hb0 = tid >> 3;
  const uint32_t colB = (uint32_t)(tile_addr<ROWB>(hb0, (tid & 7) * 16) - hb0 * ROWB);    // logical byte column held by this slot (hb0 + 64 p: same key)
  // (image row / column of halo row pass * 64 + hb0 relative to the step's first pixel: recomputed per piece from hb0 - compile-time
  //  divisor, ~6 VALU - instead of 2 x NPASS registers: the BM = 128 instance sits at the 256-register limit)
  const uint32_t pixb = (uint32_t)cs * SZ, imgb = (uint32_t)Hs * (uint32_t)Ws * pixb, cbyte = (uint32_t)cb_src * 128u + colB;

  // a tile = L pieces (LDS-DMA instructions): pieces 0 .. NVA-1 = dY, NVA .. L-1 = halo passes.  The step's scalar geometry
  // (image, first row / column, byte bases) is computed once (tile_geo), a piece adds its lane constants.
  struct TileGeo { uint32_t ho0, wo0, abase, nb; };
  auto tile_geo = [&](int step) {
    const int p0 = step * PK;
    const uint32_t n = fdiv((uint32_t)p0, a.divHoWo);
    const uint32_t rem = (uint32_t)p0 - n * (uint32_t)a.HoWo;
    TileGeo tg;
    tg.ho0 = fdiv(rem, a.divWo);
    tg.wo0 = rem - tg.ho0 * (uint32_t)a.Wo;
    tg.abase = (uint32_t)p0 * (uint32_t)a.Cout * SZ;
    tg.nb = n * imgb + cbyte;
    return tg;
  };
  auto issue_piece = [&](const TileGeo& tg, int buf, int piece) __attribute__((always_inline)) {
    char* sa = smem + buf * STAGE;
    if (piece < NVA) {
      const int i = piece;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsd, (__attribute__((address_space(3))) void*)(sa + (i * 512 + wave * 64) * 16), 16,
                                               (coA_ok && rowa[i] != STP_OOB) ? tg.abase + rowa[i] : STP_OOB, 0, 0, 0);
    } else {
      const int i = piece - NVA;
      const int h = hb0 + i * 64;
      const int r = h / G::HWP, c = h - r * G::HWP;
      const int hv = (int)tg.ho0 + r - 1, wv = (int)tg.wo0 + c - 1;
      const bool ok = r < G::ROWS + 2 && c < G::HW && (unsigned)hv < (unsigned)a.Hv && (unsigned)wv < (unsigned)a.Wv;
      const uint32_t off = tg.nb + ((uint32_t)(hv >> sh) * (uint32_t)Ws + (uint32_t)(wv >> sh)) * pixb;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, (__attribute__((address_space(3))) void*)(sa + ABYTES + (i * 512 + wave * 64) * 16), 16,
                                               ok ? off : STP_OOB, 0, 0, 0);
    }
  };
  auto issue_tile = [&](int step, int buf) {
    const TileGeo tg = tile_geo(step);
#pragma unroll
    for (int pc = 0; pc < L; ++pc) issue_piece(tg, buf, pc);
  };

  // The transpose reads are INLINE ASM with hand-counted lgkmcnt waits (see the 16 x 16 x 32 path below for why).
  auto tr_read = [&](uint32_t addr, auto off) __attribute__((always_inline)) {
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(decltype(off)::value));
    return v;
  };
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;      // LDS byte address of the ring (the asm reads take addresses)

  if constexpr (M32) {
    // ---- v_mfma_f32_32x32x16 form (BM = 128): 8 waves = 4 (32 output channels) x 2 (32 of the block's 64 input channels); a wave's
    // tile = 32 channels x (9 taps x 32 channels) = 9 accumulators of 32 x 32.  The 16 x 16 x 32 form tops out at 80 % of the MFMA
    // peak (MI355X_MICROARCH.md: ~5 vs ~8 cycles per CU for half the FLOP), and the all-taps loop is MFMA-issue bound
    // (profiles/r04h_pmc_sq.json: issue stalls 46 % of the wave cycles, MFMA pipe busy 44 %).
    // Operand layout: lane group g = lane >> 4 holds rows / columns 16 (g & 1) .. +15 and k = 8 (g >> 1) .. +7 of a 16-pixel k-step:
    // pixels 16 kk + 8 (g >> 1) + {0..3} from the first transpose read, + {4..7} from the second.  kk and the kernel row are
    // instruction immediates (16 pixels = 16 LDS rows, a kernel row = HWP rows: multiples of 8 rows, same swizzle key); the channel
    // block (g & 1) and the tap column kw sit in the 2 + 6 address registers.
    const int wm4 = wave >> 1, wn2 = wave & 1;
    const int hh = lg & 1, kh2 = lg >> 1;
    const int qb32 = (lr & 3) * 8;
    uint32_t a32[2], b32[2][3];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int q0 = kh2 * 8 + half * 4 + (lr >> 2);
      a32[half] = (uint32_t)tile_addr<ROWA>(q0, (wm4 * 32 + hh * 16) * SZ + qb32);
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) b32[half][kw] = (uint32_t)(ABYTES + tile_addr<ROWB>(q0 + kw, (wn2 * 32 + hh * 16) * SZ + qb32));
    }
    f32x16 acc32[9];
#pragma unroll
    for (int j = 0; j < 9; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc32[j][e] = 0.f;
    auto read_a32 = [&](uint32_t st, auto kkc, u32x4& fa) __attribute__((always_inline)) {
      constexpr int OFF = decltype(kkc)::value * 16 * ROWA;
      const u32x2 l2 = tr_read(st + a32[0], std::integral_constant<int, OFF>{});
      const u32x2 h2 = tr_read(st + a32[1], std::integral_constant<int, OFF>{});
      fa = u32x4{l2.x, l2.y, h2.x, h2.y};
    };
    auto read_b32 = [&](uint32_t st, auto kkc, auto khc, u32x4 (&fb)[3]) __attribute__((always_inline)) {
      constexpr int Q = decltype(kkc)::value * 16;
      constexpr int OFF = ((Q / SEG) * G::HWP + (Q % SEG) + decltype(khc)::value * G::HWP) * ROWB;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const u32x2 l2 = tr_read(st + b32[0][kw], std::integral_constant<int, OFF>{});
        const u32x2 h2 = tr_read(st + b32[1][kw], std::integral_constant<int, OFF>{});
        fb[kw] = u32x4{l2.x, l2.y, h2.x, h2.y};
      }
    };
    auto wait32 = [&](auto young, u32x4& fa, u32x4 (&fb)[3]) __attribute__((always_inline)) {
      asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(fa), "+v"(fb[0]), "+v"(fb[1]), "+v"(fb[2]) : "n"(decltype(young)::value));
    };
    static_assert(L <= 6, "one LDS-DMA piece per two MFMA groups");
    // 12 groups (k-step kk = 0..3) x (kernel row kh = 0..2) of 3 MFMAs; the fragments of group g + 1 (and the A fragment of the next
    // k-step) are requested before the MFMAs of group g; an LDS-DMA piece of the next tile follows every second group.
    auto compute32 = [&](uint32_t st, bool next, int nstep, int nb_) {
      u32x4 fa[2], fb[2][3];
      TileGeo tg = {0u, 0u, 0u, 0u};
      if (next) tg = tile_geo(nstep);
      read_a32(st, std::integral_constant<int, 0>{}, fa[0]);
      read_b32(st, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, fb[0]);
#define STP_T9M_GROUP(G_, KK_, KH_, KKN_, KHN_)                                                                                    \
      __builtin_amdgcn_sched_barrier(0);                                                                                            \
      if (KH_ == 0 && KK_ < 3) read_a32(st, std::integral_constant<int, (KK_ + 1) & 3>{}, fa[(KK_ + 1) & 1]);                       \
      if (G_ + 1 < 12) read_b32(st, std::integral_constant<int, KKN_>{}, std::integral_constant<int, KHN_>{}, fb[(G_ + 1) & 1]);    \
      wait32(std::integral_constant<int, (G_ + 1 < 12 ? 6 : 0) + ((KH_ == 0 && KK_ < 3) ? 2 : 0)>{}, fa[KK_ & 1], fb[G_ & 1]);      \
      __builtin_amdgcn_sched_barrier(0);                                                                                            \
      _Pragma("unroll") for (int kw = 0; kw < 3; ++kw)                                                                              \
        acc32[KH_ * 3 + kw] = mfma16_32x32x16(fa[KK_ & 1], fb[G_ & 1][kw], acc32[KH_ * 3 + kw]);                                    \
      __builtin_amdgcn_sched_barrier(0);                                                                                            \
      if ((G_ & 1) == 0 && G_ / 2 < L && next) issue_piece(tg, nb_, G_ / 2);
      STP_T9M_GROUP(0, 0, 0, 0, 1)
      STP_T9M_GROUP(1, 0, 1, 0, 2)
      STP_T9M_GROUP(2, 0, 2, 1, 0)
      STP_T9M_GROUP(3, 1, 0, 1, 1)
      STP_T9M_GROUP(4, 1, 1, 1, 2)
      STP_T9M_GROUP(5, 1, 2, 2, 0)
      STP_T9M_GROUP(6, 2, 0, 2, 1)
      STP_T9M_GROUP(7, 2, 1, 2, 2)
      STP_T9M_GROUP(8, 2, 2, 3, 0)
      STP_T9M_GROUP(9, 3, 0, 3, 1)
      STP_T9M_GROUP(10, 3, 1, 3, 2)
      STP_T9M_GROUP(11, 3, 2, 0, 0)
#undef STP_T9M_GROUP
      __builtin_amdgcn_sched_barrier(0);
    };
    const int nst = step1 - step0;
#pragma unroll
    for (int q = 0; q < STAGES - 1; ++q)
      if (q < nst) issue_tile(step0 + q, q);
    int buf = 0, nbuf = STAGES - 1;
    for (int st = 0; st < nst; ++st) {
      const int ahead = nst - 1 - st;
      if (STAGES >= 3 && ahead >= STAGES - 2) wait_vmcnt<(STAGES - 2) * L>();
      else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      compute32(lds0 + (uint32_t)(buf * STAGE), st + STAGES - 1 < nst, step0 + st + STAGES - 1, nbuf);
      buf = (buf + 1 == STAGES) ? 0 : buf + 1;
      nbuf = (nbuf + 1 == STAGES) ? 0 : nbuf + 1;
    }
    // slab: [wave][tap][q][lane] float4 = rows 8 q + 4 (lane >> 5) + {0..3} (output channels) of column lane & 31 (taps9_slab_decode)
    f32x4* out = out_tile + (size_t)wave * (9 * 4 * 64) + lane;
#pragma unroll
    for (int j = 0; j < 9; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        out[(j * 4 + q) * 64] = f32x4{acc32[j][4 * q], acc32[j][4 * q + 1], acc32[j][4 * q + 2], acc32[j][4 * q + 3]};
    return;
  }

  // ---- fragment addresses (bytes from the stage base).  Pixel q of the step (row q / SEG, column q % SEG); a 16-lane group reads 4
  // consecutive pixels: lane -> pixel (lr >> 2), 8-byte quad (lr & 3) of the 32-byte channel block.
  const int qb = (lr & 3) * 8;
  // A: the 32-byte channel block i of a wave sits in byte bits 5.. of the row, the swizzle key XORs the same bits: block i = block 0
  // XOR (i * 32) - one address register per pixel group (4 instead of 4 x TM)
  int aaddr[4], baddr[4][3];           // [pixel group e = chunk * 2 + half] / [e][kw]
  static_assert(ROWA >= 64 && ((BM / 2) * SZ) % (TM * 32) == 0, "the wave's channel blocks occupy an aligned bit field of the row");
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int q = e * 16 + lg * 4 + (lr >> 2);
    aaddr[e] = tile_addr<ROWA>(q, (wm * (BM / 2)) * SZ + qb);
    const int r = q / SEG, c = q - r * SEG;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) baddr[e][kw] = ABYTES + tile_addr<ROWB>(r * G::HWP + c + kw, wn * 32 + qb);
  }

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // A step = 2 chunks of 32 pixels x 3 kernel rows = 6 groups of 3 taps x TM MFMAs.  The B fragments of group g + 1 are requested
  // BEFORE the MFMAs of group g (two sets of 12 registers, pinned by sched_barrier: left alone, the scheduler hoists every read of
  // the step to its top - 52 fragment registers on top of 144 accumulators + 28 addresses = 36 spills at BM = 128).
  // The transpose reads are INLINE ASM with hand-counted lgkmcnt waits: through the builtin, the compiler's wait-count pass puts
  // `s_waitcnt vmcnt(0)` in front of the first LDS read after every LDS-DMA instruction (it cannot tell that the ring slot being
  // filled is not the one being read) - with the pieces of the next tile issued between the MFMA groups that drained the whole DMA
  // queue six times per step (first build: 5600 cycles per step for 2304 cycles of MFMA work).  The step's barrier + counted vmcnt
  // order the slot that IS read.
  auto read_a = [&](uint32_t st, int c, u32x4 (&fa)[TM]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const u32x2 l2 = tr_read(st + (uint32_t)(aaddr[2 * c] ^ (i * 32)), std::integral_constant<int, 0>{});
      const u32x2 h2 = tr_read(st + (uint32_t)(aaddr[2 * c + 1] ^ (i * 32)), std::integral_constant<int, 0>{});
      fa[i] = u32x4{l2.x, l2.y, h2.x, h2.y};
    }
  };
  // (kernel row kh = a compile-time byte offset: a multiple of 8 LDS rows, same swizzle key)
  auto read_b = [&](uint32_t st, int c, auto khc, u32x4 (&fb)[3]) __attribute__((always_inline)) {
    constexpr int OFF = decltype(khc)::value * G::HWP * ROWB;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const u32x2 l2 = tr_read(st + (uint32_t)baddr[2 * c][kw], std::integral_constant<int, OFF>{});
      const u32x2 h2 = tr_read(st + (uint32_t)baddr[2 * c + 1][kw], std::integral_constant<int, OFF>{});
      fb[kw] = u32x4{l2.x, l2.y, h2.x, h2.y};
    }
  };
  // everything but the `young` most recent LDS reads has returned; the fragments are operands of the asm, so no MFMA that consumes
  // them can be scheduled above the wait
  auto wait_frags = [&](auto young, u32x4 (&fa)[TM], u32x4 (&fb)[3]) __attribute__((always_inline)) {
    if constexpr (TM == 4)
      asm volatile("s_waitcnt lgkmcnt(%7)" : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fa[3]), "+v"(fb[0]), "+v"(fb[1]), "+v"(fb[2]) : "n"(decltype(young)::value));
    else if constexpr (TM == 2)
      asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(fa[0]), "+v"(fa[1]), "+v"(fb[0]), "+v"(fb[1]), "+v"(fb[2]) : "n"(decltype(young)::value));
    else
      asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(fa[0]), "+v"(fb[0]), "+v"(fb[1]), "+v"(fb[2]) : "n"(decltype(young)::value));
  };
  static_assert(L <= 6, "one LDS-DMA piece per MFMA group");
  // A step = 2 chunks of 32 pixels x 3 kernel rows = 6 groups of 3 taps x TM MFMAs.  The B fragments of group g + 1 are requested
  // BEFORE the MFMAs of group g (two sets of 12 registers).  `next`: the pieces of tile `nstep` go into ring slot `nb_` BETWEEN the
  // MFMA groups, one per group (an LDS-DMA instruction costs ~60 issue cycles among MFMAs, 100-180 in a burst that opens the step).
  // Steps are software-pipelined ACROSS the step barrier (-DSTP_T9_NOPIPE: the what-if build without): the barrier of step st + 1 sits in front of the
  // LAST MFMA group of step st - every wave has waited for its own pieces of tile st + 1 and for every fragment read of the current
  // slot - and the first fragments of step st + 1 are requested before / right behind those MFMAs.  With the barrier at the top of the
  // step every wave sat through the first read round trip of every step with the MFMA pipe empty (~400 of ~3700 cycles per step).
  constexpr int P5 = L < 5 ? L : 5;                          // pieces of the next-but-one tile issued before the last group
#if defined(STP_T9_NOPIPE)
  constexpr bool PIPE = false;
#else
  constexpr bool PIPE = true;
#endif
  u32x4 fa[TM], fb[2][3];
  auto compute = [&](uint32_t st, uint32_t st_nx, bool more, bool next, int nstep, int nb_) {
    TileGeo tg = {0u, 0u, 0u, 0u};
    if (next) tg = tile_geo(nstep);
#define STP_T9_MFMAS(G_, KH_)                                                                                    \
    _Pragma("unroll") for (int kw = 0; kw < 3; ++kw)                                                             \
      _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                             \
        acc[i][KH_ * 3 + kw] = mfma16_16x16x32(fa[i], fb[G_ & 1][kw], acc[i][KH_ * 3 + kw]);
#define STP_T9_GROUP(G_, C_, KH_, CN_, KHN_)                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                           \
    read_b(st, CN_, std::integral_constant<int, KHN_>{}, fb[(G_ + 1) & 1]);                                      \
    wait_frags(std::integral_constant<int, 6>{}, fa, fb[G_ & 1]);                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                           \
    STP_T9_MFMAS(G_, KH_)                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                           \
    if (G_ < L && next) issue_piece(tg, nb_, G_);                                                                \
    if (G_ == 2) {   /* the A fragments of the second chunk, behind the last MFMAs that read the first */        \
      __builtin_amdgcn_sched_barrier(0);                                                                         \
      read_a(st, 1, fa);                                                                                         \
    }
    // (fa = A fragments of chunk 0 and fb[0] = group 0 of THIS step are in flight: requested by the prologue / the previous step)
    // (group 2 -> 3: the 2 * TM reads of read_a(1) are OLDER than the 6 of read_b(group 4), so "all but the youngest 6" covers them)
    STP_T9_GROUP(0, 0, 0, 0, 1)
    STP_T9_GROUP(1, 0, 1, 0, 2)
    STP_T9_GROUP(2, 0, 2, 1, 0)
    STP_T9_GROUP(3, 1, 0, 1, 1)
    STP_T9_GROUP(4, 1, 1, 1, 2)
    // ---- group 5 (its B fragments were requested in group 4) + the hand-over to the next step
    __builtin_amdgcn_sched_barrier(0);
    wait_frags(std::integral_constant<int, 0>{}, fa, fb[1]);        // every LDS read of this wave has returned (the current slot is done with)
    if (PIPE && more) {
      if (next) wait_vmcnt<P5>(); else wait_vmcnt<0>();              // own pieces of tile st + 1 have landed (the P5 youngest: tile st + 2)
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      read_b(st_nx, 0, std::integral_constant<int, 0>{}, fb[0]);
    }
    __builtin_amdgcn_sched_barrier(0);
    STP_T9_MFMAS(5, 2)
    __builtin_amdgcn_sched_barrier(0);
    if (5 < L && next) issue_piece(tg, nb_, 5);
    if (PIPE && more) {
      __builtin_amdgcn_sched_barrier(0);
      read_a(st_nx, 0, fa);
    }
#undef STP_T9_GROUP
#undef STP_T9_MFMAS
    __builtin_amdgcn_sched_barrier(0);
  };

  const int nst = step1 - step0;
#pragma unroll
  for (int q = 0; q < STAGES - 1; ++q)
    if (q < nst) issue_tile(step0 + q, q);
  static_assert(STAGES == 3, "the pipelined hand-over assumes the 3-slot ring");
  if (nst > 1) wait_vmcnt<L>(); else wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  read_a(lds0, 0, fa);
  read_b(lds0, 0, std::integral_constant<int, 0>{}, fb[0]);
  int buf = 0, nbuf = STAGES - 1;
  for (int st = 0; st < nst; ++st) {
    const int nx = (buf + 1 == STAGES) ? 0 : buf + 1;
    if (!PIPE && st > 0) {                                   // (what-if build: barrier and first reads at the top of every step)
      if (st + 1 < nst) wait_vmcnt<L>(); else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      read_a(lds0 + (uint32_t)(buf * STAGE), 0, fa);
      read_b(lds0 + (uint32_t)(buf * STAGE), 0, std::integral_constant<int, 0>{}, fb[0]);
    }
    compute(lds0 + (uint32_t)(buf * STAGE), lds0 + (uint32_t)(nx * STAGE), st + 1 < nst, st + STAGES - 1 < nst, step0 + st + STAGES - 1, nbuf);
    buf = nx;
    nbuf = (nbuf + 1 == STAGES) ? 0 : nbuf + 1;
  }

  // ---- slab, fragment-major: [wave][i][tap][lane] float4 = the lane's 4 output channels of one column (taps9_slab_decode)
  f32x4* out = out_tile + (size_t)wave * (TM * TN * 64) + lane;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) out[(i * TN + j) * 64] = acc[i][j];
#endif
}

// element e (float4) of an all-taps tile's slab -> first of its 4 output channels (relative to the tile), tap and input channel (0..63)
template <int BM, bool M32> __device__ __forceinline__ void taps9_slab_decode(int e, int& co, int& tap, int& ci) {
  const int lane = e & 63;
  if constexpr (M32) {
    const int t2 = e >> 6, q = t2 & 3, t3 = t2 >> 2;
    tap = t3 % 9;
    const int wave = t3 / 9;
    co = (wave >> 1) * 32 + 8 * q + 4 * (lane >> 5);
    ci = (wave & 1) * 32 + (lane & 31);
  } else {
    constexpr int TM = BM / 2 / 16;
    const int q = e >> 6;
    const int ij = q % (TM * 9), wave = q / (TM * 9);
    const int i = ij / 9;
    tap = ij - i * 9;
    co = (wave >> 2) * (BM / 2) + i * 16 + (lane >> 4) * 4;
    ci = (wave & 3) * 16 + (lane & 15);
  }
}

template <int BM, int STAGES, bool M32 = false>
__global__ __launch_bounds__(512) void conv_wgrad_taps9_group_kernel(const char* __restrict__ table, f32x4* __restrict__ slabs) {
  const WgGroupHeader* const hd = reinterpret_cast<const WgGroupHeader*>(table);
  const WgLayer* const layers = reinterpret_cast<const WgLayer*>(table + hd->off_layers);
  const WgSeg* const segs = reinterpret_cast<const WgSeg*>(table + hd->off_segs);
  const int* const first = reinterpret_cast<const int*>(table + hd->off_first);
  const int w = hd->xcd == 1 ? xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int s0 = __builtin_amdgcn_readfirstlane(first[w]), s1 = __builtin_amdgcn_readfirstlane(first[w + 1]);
  for (int s = s0; s < s1; ++s) {
    WgSeg sg;
    load_constant(sg, segs + s);
    WgradArgs a;
    load_constant(a, &layers[sg.layer].a);
    if (s != s0) __syncthreads();        // the previous segment's last fragment reads precede this segment's first LDS-DMA
    f32x4* const out = slabs + (size_t)sg.slot * (BM * 576 / 4);
    // a group mixes feature-map widths (decoder + encoder stages): the segment's layer picks the body (wave-uniform)
    if (a.Wo >= 64) conv_wgrad_taps9_body<BM, 64, STAGES, M32>(a, sg.tile, sg.step0, sg.step1, out);
    else if (a.Wo == 32) conv_wgrad_taps9_body<BM, 32, STAGES, M32>(a, sg.tile, sg.step0, sg.step1, out);
    else conv_wgrad_taps9_body<BM, 16, STAGES, M32>(a, sg.tile, sg.step0, sg.step1, out);
  }
}

template <int BM, int SL, bool M32 = false>
__global__ __launch_bounds__(256) void wgrad_group9_reduce_kernel(const char* __restrict__ table, const f32x4* __restrict__ slabs) {
  constexpr int PER = BM * 576 / 4, EPB = 256 / SL, BPT = PER / EPB;
  static_assert(PER % EPB == 0, "whole blocks per tile");
  __shared__ f32x4 shm[SL][EPB];
  const WgGroupHeader* const hd = reinterpret_cast<const WgGroupHeader*>(table);
  const WgLayer* const layers = reinterpret_cast<const WgLayer*>(table + hd->off_layers);
  const int* const slot_list = reinterpret_cast<const int*>(table + hd->off_slots);
  const WgTile tl = reinterpret_cast<const WgTile*>(table + hd->off_tiles)[blockIdx.x / BPT];
  const int ev = threadIdx.x % EPB, sl = threadIdx.x / EPB;
  const int e = (blockIdx.x % BPT) * EPB + ev;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  {
    int k = sl;
    for (; k + 3 * SL < tl.nslots; k += 4 * SL) {
      const f32x4 a0 = slabs[(size_t)slot_list[tl.list0 + k] * PER + e], a1 = slabs[(size_t)slot_list[tl.list0 + k + SL] * PER + e];
      const f32x4 a2 = slabs[(size_t)slot_list[tl.list0 + k + 2 * SL] * PER + e], a3 = slabs[(size_t)slot_list[tl.list0 + k + 3 * SL] * PER + e];
      s += a0; s += a1; s += a2; s += a3;
    }
    for (; k < tl.nslots; k += SL) s += slabs[(size_t)slot_list[tl.list0 + k] * PER + e];
  }
  if (SL > 1) {
    shm[sl][ev] = s;
    __syncthreads();
#pragma unroll
    for (int w = SL / 2; w > 0; w >>= 1) {
      if (sl < w) shm[sl][ev] += shm[sl + w][ev];
      __syncthreads();
    }
    s = shm[0][ev];
    if (sl != 0) return;
  }
  const WgLayer& L = layers[tl.layer];
  int co, tap, ci;
  taps9_slab_decode<BM, M32>(e, co, tap, ci);
  const int tile_m = tl.t % L.a.ntile_m, cib = tl.t / L.a.ntile_m;
  const int kc = tap * L.a.Ctot + cib * 64 + ci;
  co += tile_m * BM;
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (co + r < L.a.Cout) {
      float* d = L.dw + (size_t)(co + r) * L.a.K + kc;
      *d = L.accumulate ? *d + s[r] : s[r];
    }
}

// sum of a tile's partial slabs in line order, scattered into dW (the element decode of wgrad_reduce_row_kernel).  SL lanes walk the
// slab list of an element in parallel (fixed assignment, fixed-shape LDS tree: deterministic) - stage-1 tiles have 40+ slabs.
template <int BM, int WM, int WN, int SL>
__global__ __launch_bounds__(256) void wgrad_group_reduce_kernel(const char* __restrict__ table, const f32x4* __restrict__ slabs) {
  constexpr int PER = BM * 192 / 4, EPB = 256 / SL, BPT = PER / EPB;
  static_assert(PER % EPB == 0, "whole blocks per tile");
  __shared__ f32x4 sh[SL][EPB];
  const WgGroupHeader* const hd = reinterpret_cast<const WgGroupHeader*>(table);
  const WgLayer* const layers = reinterpret_cast<const WgLayer*>(table + hd->off_layers);
  const int* const slot_list = reinterpret_cast<const int*>(table + hd->off_slots);
  const WgTile tl = reinterpret_cast<const WgTile*>(table + hd->off_tiles)[blockIdx.x / BPT];
  const int ev = threadIdx.x % EPB, sl = threadIdx.x / EPB;
  const int e = (blockIdx.x % BPT) * EPB + ev;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  {
    // (four slab loads in flight per lane; same order of additions)
    int k = sl;
    for (; k + 3 * SL < tl.nslots; k += 4 * SL) {
      const f32x4 a0 = slabs[(size_t)slot_list[tl.list0 + k] * PER + e], a1 = slabs[(size_t)slot_list[tl.list0 + k + SL] * PER + e];
      const f32x4 a2 = slabs[(size_t)slot_list[tl.list0 + k + 2 * SL] * PER + e], a3 = slabs[(size_t)slot_list[tl.list0 + k + 3 * SL] * PER + e];
      s += a0; s += a1; s += a2; s += a3;
    }
    for (; k < tl.nslots; k += SL) s += slabs[(size_t)slot_list[tl.list0 + k] * PER + e];
  }
  if (SL > 1) {
    sh[sl][ev] = s;
    __syncthreads();
#pragma unroll
    for (int w = SL / 2; w > 0; w >>= 1) {
      if (sl < w) sh[sl][ev] += sh[sl + w][ev];
      __syncthreads();
    }
    s = sh[0][ev];
    if (sl != 0) return;
  }
  const WgLayer& L = layers[tl.layer];
  int col, co;
  row_slab_decode<BM, WM, WN>(e, co, col);
  const int tile_m = tl.t % L.a.ntile_m, tile_n = tl.t / L.a.ntile_m;
  const int ncib = L.a.Ctot >> 6, kh = tile_n / ncib, cib = tile_n - kh * ncib;
  const int kc = (kh * 3 + (col >> 6)) * L.a.Ctot + cib * 64 + (col & 63);
  co += tile_m * BM;
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (co + r < L.a.Cout) {
      float* d = L.dw + (size_t)(co + r) * L.a.K + kc;
      *d = L.accumulate ? *d + s[r] : s[r];
    }
}

// Reduction of the row-of-taps kernel's fragment-major slabs: element e = ((tile * 4 + wave) * TM*TN + i*TN + j) * 64 + lane.
// SL lanes walk the splits of an element in parallel (fixed assignment, fixed-shape LDS tree: deterministic), then the 4 floats
// go to dW[co + r][(kh*3 + kw) * C0 + cib*64 + ci].
template <int BM, int WM, int WN, int SL>
__global__ __launch_bounds__(256) void wgrad_reduce_row_kernel(const f32x4* __restrict__ slabs, float* __restrict__ dw, int64_t per_split,
                                                               int splits, int accumulate, int ntile_m, int C0, int Cout, int K) {
  constexpr int EPB = 256 / SL;
  __shared__ f32x4 sh[SL][EPB];
  const int ev = threadIdx.x % EPB, sl = threadIdx.x / EPB;
  const int64_t e = (int64_t)blockIdx.x * EPB + ev;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (e < per_split)
    for (int k = sl; k < splits; k += SL) s += slabs[(size_t)k * per_split + e];
  if (SL > 1) {
    sh[sl][ev] = s;
    __syncthreads();
#pragma unroll
    for (int w = SL / 2; w > 0; w >>= 1) {
      if (sl < w) sh[sl][ev] += sh[sl + w][ev];
      __syncthreads();
    }
    s = sh[0][ev];
  }
  if (sl != 0 || e >= per_split) return;
  constexpr int PERT = BM * 192 / 4;                       // float4 elements per tile
  const int t = (int)(e / PERT);
  int col, co;
  row_slab_decode<BM, WM, WN>((int)(e - (int64_t)t * PERT), co, col);
  const int tile_m = t % ntile_m, tile_n = t / ntile_m;
  const int ncib = C0 >> 6, kh = tile_n / ncib, cib = tile_n - kh * ncib;
  const int kc = (kh * 3 + (col >> 6)) * C0 + cib * 64 + (col & 63);
  co += tile_m * BM;
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (co + r < Cout) {
      float* d = dw + (size_t)(co + r) * K + kc;
      *d = accumulate ? *d + s[r] : s[r];
    }
}

// dw[i] (+)= sum_k slabs[k][i], 4 floats per thread (count is a multiple of 4: Cout*K with K % 4 == 0)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ dw, int64_t count,
                                                           int splits, int accumulate) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= count) return;
  f32x4 s = *reinterpret_cast<const f32x4*>(slabs + i);
  int k = 1;
  for (; k + 3 < splits; k += 4) {      // four slab loads in flight; same order of additions
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(slabs + (size_t)k * count + i), a1 = *reinterpret_cast<const f32x4*>(slabs + (size_t)(k + 1) * count + i);
    const f32x4 a2 = *reinterpret_cast<const f32x4*>(slabs + (size_t)(k + 2) * count + i), a3 = *reinterpret_cast<const f32x4*>(slabs + (size_t)(k + 3) * count + i);
    s += a0; s += a1; s += a2; s += a3;
  }
  for (; k < splits; ++k) s += *reinterpret_cast<const f32x4*>(slabs + (size_t)k * count + i);
  if (accumulate) s += *reinterpret_cast<const f32x4*>(dw + i);
  *reinterpret_cast<f32x4*>(dw + i) = s;
}

// Many slabs, few elements (small-channel layers: up to 1024 slabs of a few thousand floats): 16 lanes walk
// the slabs in parallel for each group of 4 elements, then a fixed-shape LDS tree combines them.
__global__ __launch_bounds__(256) void wgrad_reduce_wide_kernel(const float* __restrict__ slabs, float* __restrict__ dw, int64_t count,
                                                                int splits, int accumulate) {
  __shared__ f32x4 sh[16][16];
  const int ev = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int64_t i = ((int64_t)blockIdx.x * 16 + ev) * 4;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (i < count) {
    // eight loads in flight per lane (the loop is latency-bound otherwise: 13.5 us for 10-20 MB of slabs); same order of additions
    int k = sl;
    for (; k + 112 < splits; k += 128) {
      f32x4 a[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] = *reinterpret_cast<const f32x4*>(slabs + (size_t)(k + 16 * u) * count + i);
#pragma unroll
      for (int u = 0; u < 8; ++u) s += a[u];
    }
    for (; k < splits; k += 16) s += *reinterpret_cast<const f32x4*>(slabs + (size_t)k * count + i);
  }
  sh[sl][ev] = s;
  __syncthreads();
  for (int w = 8; w > 0; w >>= 1) {
    if (sl < w) sh[sl][ev] += sh[sl + w][ev];
    __syncthreads();
  }
  if (sl == 0 && i < count) {
    f32x4 r = sh[0][ev];
    if (accumulate) r += *reinterpret_cast<const f32x4*>(dw + i);
    *reinterpret_cast<f32x4*>(dw + i) = r;
  }
}

struct WgradPlan {
  int tile, bm, bn, ntile_m, ntile_n, splits, steps_per_split, nsteps;
};

extern "C" int stp_wgrad_sc_eligible(const stp_wgrad_params* p);
extern "C" int stp_wgrad_sc_slabs(const stp_wgrad_params* p);
extern "C" int stp_wgrad_sc_partial(const stp_wgrad_params* p, void* workspace, void* stream);
#define WG_TILE_SC 100  // small-channel halo-tile kernel (conv_sc.hip): splits = number of persistent workgroups

static int device_cu_count() {
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
    n = 256;   // MI355X (also the answer on a build host without a GPU: plan sizes must not depend on where they are computed)
  return n;
}

static WgradPlan plan_wgrad(const stp_wgrad_params* p) {
  WgradPlan w;
  if (p->splits == 0 && stp_wgrad_sc_eligible(p)) {
    w.tile = WG_TILE_SC; w.bm = w.bn = 0; w.ntile_m = w.ntile_n = 1;
    w.splits = stp_wgrad_sc_slabs(p); w.steps_per_split = 0; w.nsteps = 0;
    return w;
  }
  const int K = p->KH * p->KW * (p->C0 + p->C1);
  const int pk = p->dtype == STP_H16 ? 64 : 32;
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
    // Fill the machine EXACTLY once: workgroup slots = CUs x co-resident workgroups (LDS-limited: 2 for the 128x128 tile,
    // 3 for 64x128), splits = floor(slots / tiles).  One workgroup more than the slots costs a whole extra round
    // (scratch/wgrad_split_sweep.py: 128->128 @64x64, 9 tiles: 57 splits = 513 workgroups 49 us, 56 splits = 504 workgroups 33 us);
    // every split also costs a slab write + read of Cout*K floats, so fewer is better at equal fill.
    static const int cus = device_cu_count();
    static const int target = getenv("STP_WGRAD_BLOCKS") ? atoi(getenv("STP_WGRAD_BLOCKS")) : 0;
    const int slots = target > 0 ? target : cus * (w.bm == 64 ? 3 : 2);
    splits = slots / tiles;
    if (splits < 1) splits = 1;
    const int max_by_steps = w.nsteps / 8 > 0 ? w.nsteps / 8 : 1;  // keep >= 8 steps per split
    if (splits > max_by_steps) splits = max_by_steps;
    if (splits > 512) splits = 512;   // (the stem: 2 tiles x 384 splits = 768 workgroups, 113 -> 88 us)
  }
  if (splits > w.nsteps) splits = w.nsteps;
  if (splits < 1) splits = 1;
  w.steps_per_split = ceil_div(w.nsteps, splits);
  w.splits = ceil_div(w.nsteps, w.steps_per_split);
  return w;
}

static WgradPlan plan_wgrad_gemm(const stp_wgrad_params* p);

// ---- row-of-taps kernel (conv_wgrad_row_kernel): eligibility and plan ------------------------------------------------------
#define WG_TILE_ROW128 5
#define WG_TILE_ROW64 6
static bool wgrad_row_eligible(const stp_wgrad_params* p) {
  if (!p || p->dtype != STP_H16 || p->KH != 3 || p->KW != 3 || p->stride != 1 || p->pad != 1 || (p->C0 & 63) || (p->C1 & 63) ||
      p->Ho != p->Hv || p->Wo != p->Wv || (p->Cout & 7))
    return false;
  if (p->src_bn_mean && (!p->src_bn_rstd || p->C1 != 0 || p->src0_mode != STP_SRC_DIRECT)) return false;
  if (p->src0_mode == STP_SRC_DIRECT ? (p->Hs0 != p->Hv || p->Ws0 != p->Wv)
                                     : (p->src0_mode != STP_SRC_NEAREST2X || p->Hv != 2 * p->Hs0 || p->Wv != 2 * p->Ws0))
    return false;
  const int hw = p->Ho * p->Wo;
  return (p->Wo % 64 == 0) || (p->Wo >= 16 && 64 % p->Wo == 0 && hw % 64 == 0);
}
static bool wgrad_row_auto(const stp_wgrad_params* p) {
  static const bool on = !(getenv("STP_WGRAD_ROW") && atoi(getenv("STP_WGRAD_ROW")) == 0);
  return on && p->splits == 0 && wgrad_row_eligible(p);
}
static WgradPlan plan_wgrad_row(const stp_wgrad_params* p) {
  WgradPlan w;
  w.bm = p->Cout <= 64 ? 64 : 128; w.bn = 192;
  w.tile = w.bm == 64 ? WG_TILE_ROW64 : WG_TILE_ROW128;
  w.ntile_m = ceil_div(p->Cout, w.bm);
  w.ntile_n = 3 * ((p->C0 + p->C1) / 64);
  const int64_t P = (int64_t)p->N * p->Ho * p->Wo;
  w.nsteps = (int)(P / 64);
  const int tiles = w.ntile_m * w.ntile_n;
  int splits = p->splits;
  if (splits <= 0) {
    static const int cus = device_cu_count();
    static const int target = getenv("STP_WGRAD_ROW_BLOCKS") ? atoi(getenv("STP_WGRAD_ROW_BLOCKS")) : 0;
    const int slots = target > 0 ? target : cus * (w.bm == 64 ? 3 : 2);   // co-resident workgroups (LDS: 40 / 56 KB each)
    splits = slots / tiles;
    if (splits < 1) splits = 1;
    const int max_by_steps = w.nsteps / 8 > 0 ? w.nsteps / 8 : 1;
    if (splits > max_by_steps) splits = max_by_steps;
    if (splits > 512) splits = 512;
  }
  if (splits > w.nsteps) splits = w.nsteps;
  if (splits < 1) splits = 1;
  w.steps_per_split = ceil_div(w.nsteps, splits);
  w.splits = ceil_div(w.nsteps, w.steps_per_split);
  return w;
}

// enough for either kernel family (the automatic choice and the forced GEMM variants)
extern "C" size_t stp_conv2d_wgrad_workspace_bytes(const stp_wgrad_params* p) {
  if (!p) return 0;
  const WgradPlan w = plan_wgrad(p), g = plan_wgrad_gemm(p);
  const size_t K = (size_t)p->KH * p->KW * (p->C0 + p->C1);
  int splits = w.splits > g.splits ? w.splits : g.splits;
  size_t bytes = (size_t)splits * p->Cout * K * sizeof(float);
  if (wgrad_row_eligible(p)) {
    const WgradPlan r = plan_wgrad_row(p);
    const size_t rb = (size_t)r.splits * r.ntile_m * r.ntile_n * r.bm * 192 * sizeof(float);   // fragment-major, whole tiles
    if (rb > bytes) bytes = rb;
  }
  return bytes;
}

template <typename K>
static int launch_wg(K kern, WgradArgs& a, size_t lds, int splits, bool& attr_set, hipStream_t s) {
  if (lds > 64 * 1024 && !attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return STP_E_LAUNCH;
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(a.ntile_m * a.ntile_n * splits), dim3(256), lds, s, a);
  STP_LAUNCH_CHECK();
  return STP_OK;
}

template <typename T, int BM, int BN, int WM, int WN, bool C4>
static int launch_wgrad(WgradArgs& a, int splits, hipStream_t s) {
  static bool attr_set = false;
  constexpr int PK = 128 / (int)sizeof(T);
  return launch_wg(conv_wgrad_kernel<T, BM, BN, WM, WN, C4>, a, (size_t)2 * PK * (BM + BN) * sizeof(T), splits, attr_set, s);
}

template <typename T, int BM, int BN, int WM, int WN, int STAGES>
static int launch_wgrad_dma(WgradArgs& a, int splits, hipStream_t s) {
  static bool attr_set = false;
  constexpr int PK = 128 / (int)sizeof(T);
  static bool attr_set_u = false;
  const size_t lds = (size_t)STAGES * PK * (BM + BN) * sizeof(T) + 4096;
  if (a.rowu) return launch_wg(conv_wgrad_dma_kernel<T, BM, BN, WM, WN, STAGES, true>, a, lds, splits, attr_set_u, s);
  return launch_wg(conv_wgrad_dma_kernel<T, BM, BN, WM, WN, STAGES, false>, a, lds, splits, attr_set, s);
}

// variant: 0 = auto, 1 = legacy register-staged, 2/3 = DMA ring with that many stages
template <typename T, bool C4>
static int launch_wgrad_tile(WgradArgs& a, const WgradPlan& w, bool dma_ok, int variant, hipStream_t s) {
  if constexpr (C4) {
    switch (w.tile) {
      case 1: return launch_wgrad<T, 128, 128, 2, 2, true>(a, w.splits, s);
      case 2: return launch_wgrad<T, 64, 128, 1, 4, true>(a, w.splits, s);
      case 3: return launch_wgrad<T, 32, 256, 1, 4, true>(a, w.splits, s);
      case 4: return launch_wgrad<T, 16, 256, 1, 4, true>(a, w.splits, s);
      default: return STP_E_BADARG;
    }
  } else {
    const bool f32 = sizeof(T) == 4;
    int v = variant;
    // (STP_WGRAD_DMA_STAGES=3: the 3-stage ring for the automatic choice - experiments on the lone 1x1 layers of the bottleneck ResNets,
    //  whose 18-step workgroups wait for every 64-pixel step's LDS-DMA with only one other step in flight)
    static const int dma_auto = (getenv("STP_WGRAD_DMA_STAGES") && atoi(getenv("STP_WGRAD_DMA_STAGES")) == 3) ? 3 : 2;
    if (v == 0) v = dma_ok ? dma_auto : 1;
    if (v >= 2 && (!dma_ok || (f32 && w.tile >= 3))) v = 1;  // fp32 256-column tiles: swizzle key varies per pass
    if (v == 1) {
      switch (w.tile) {
        case 1: return launch_wgrad<T, 128, 128, 2, 2, false>(a, w.splits, s);
        case 2: return launch_wgrad<T, 64, 128, 1, 4, false>(a, w.splits, s);
        case 3: return launch_wgrad<T, 32, 256, 1, 4, false>(a, w.splits, s);
        case 4: return launch_wgrad<T, 16, 256, 1, 4, false>(a, w.splits, s);
        default: return STP_E_BADARG;
      }
    }
    if constexpr (sizeof(T) == 2) {
      switch (w.tile * 4 + v) {
        case 1 * 4 + 2: return launch_wgrad_dma<T, 128, 128, 2, 2, 2>(a, w.splits, s);
        case 1 * 4 + 3: return launch_wgrad_dma<T, 128, 128, 2, 2, 3>(a, w.splits, s);
        case 2 * 4 + 2: return launch_wgrad_dma<T, 64, 128, 1, 4, 2>(a, w.splits, s);
        case 2 * 4 + 3: return launch_wgrad_dma<T, 64, 128, 1, 4, 3>(a, w.splits, s);
        case 3 * 4 + 2: return launch_wgrad_dma<T, 32, 256, 1, 4, 2>(a, w.splits, s);
        case 3 * 4 + 3: return launch_wgrad_dma<T, 32, 256, 1, 4, 3>(a, w.splits, s);
        case 4 * 4 + 2: return launch_wgrad_dma<T, 16, 256, 1, 4, 2>(a, w.splits, s);
        case 4 * 4 + 3: return launch_wgrad_dma<T, 16, 256, 1, 4, 3>(a, w.splits, s);
        default: return STP_E_BADARG;
      }
    } else {
      switch (w.tile * 4 + v) {
        case 1 * 4 + 2: return launch_wgrad_dma<T, 128, 128, 2, 2, 2>(a, w.splits, s);
        case 1 * 4 + 3: return launch_wgrad_dma<T, 128, 128, 2, 2, 3>(a, w.splits, s);
        case 2 * 4 + 2: return launch_wgrad_dma<T, 64, 128, 1, 4, 2>(a, w.splits, s);
        case 2 * 4 + 3: return launch_wgrad_dma<T, 64, 128, 1, 4, 3>(a, w.splits, s);
        default: return STP_E_BADARG;
      }
    }
  }
}

static WgradPlan plan_wgrad_gemm(const stp_wgrad_params* p) {
  stp_wgrad_params q = *p;
  if (q.splits == 0 && stp_wgrad_sc_eligible(p)) q.splits = -1;  // any non-zero value bypasses the small-channel plan...
  WgradPlan w = plan_wgrad(&q);
  return w;
}

static int wgrad_fill(const stp_wgrad_params* p, void* workspace, size_t workspace_bytes, WgradArgs& a, WgradPlan& w, bool* c4_out,
                      bool* dma_out) {
  if (!p || !p->src0 || !p->dy || !p->dw || !workspace) return STP_E_BADARG;
  if (p->dtype != STP_F32 && p->dtype != STP_H16) return STP_E_BADARG;
  const int vec = p->dtype == STP_H16 ? 8 : 4;
  const int sz = p->dtype == STP_H16 ? 2 : 4;
  const bool c4 = (p->dtype == STP_H16) && p->C0 == 4 && p->C1 == 0;
  if (c4) {
    if ((p->KW & 1) || p->src0_mode != STP_SRC_DIRECT) return STP_E_BADARG;
  } else if ((p->C0 % vec) || (p->C1 % vec)) {
    return STP_E_BADARG;
  }
  if (p->Cout % vec) return STP_E_BADARG;
  if (p->C1 > 0 && !p->src1) return STP_E_BADARG;
  if (stp_conv2d_wgrad_workspace_bytes(p) > workspace_bytes) return STP_E_WORKSPACE;
  w = plan_wgrad_gemm(p);
  if ((size_t)w.splits * p->Cout * p->KH * p->KW * (p->C0 + p->C1) * sizeof(float) > workspace_bytes) return STP_E_WORKSPACE;
  a.src0 = (const char*)p->src0; a.src1 = (const char*)p->src1; a.dy = (const char*)p->dy; a.out = (float*)workspace;
  a.N = p->N; a.Hs0 = p->Hs0; a.Ws0 = p->Ws0; a.Hv = p->Hv; a.Wv = p->Wv; a.C0 = p->C0; a.C1 = p->C1;
  a.Ctot = p->C0 + p->C1; a.mode = p->src0_mode;
  a.KH = p->KH; a.KW = p->KW; a.stride = p->stride; a.pad = p->pad; a.Ho = p->Ho; a.Wo = p->Wo; a.Cout = p->Cout;
  a.K = p->KH * p->KW * a.Ctot;
  const int64_t P = (int64_t)p->N * p->Ho * p->Wo;
  if (P >= (1ll << 31)) return STP_E_BADARG;
  a.P = (int)P; a.HoWo = p->Ho * p->Wo;
  a.ntile_m = w.ntile_m; a.ntile_n = w.ntile_n; a.steps_per_split = w.steps_per_split; a.nsteps = w.nsteps;
  // measured per layer (scratch/launch_table.py with and without): the XCD-contiguous order gains 5-10% when one
  // Cout tile covers the layer (Cout <= 128), is neutral for Cout = 512 and loses 35% for Cout = 256
  a.xcd = w.ntile_m == 1;
  {
    const int pk = 128 / sz;
    a.rowu = (p->Wo % pk == 0) || (pk % p->Wo == 0 && a.HoWo % pk == 0);
  }
  a.divC = make_fastdiv((uint32_t)a.Ctot); a.divKW = make_fastdiv((uint32_t)a.KW);
  a.divHoWo = make_fastdiv((uint32_t)a.HoWo); a.divWo = make_fastdiv((uint32_t)a.Wo);
  const int64_t lim = 1ll << 31;
  const int64_t b0 = (int64_t)p->N * p->Hs0 * p->Ws0 * p->C0 * sz, b1 = (int64_t)p->N * p->Hv * p->Wv * p->C1 * sz;
  const int64_t bd = P * p->Cout * sz;
  a.bytes0 = (uint32_t)(b0 < lim ? b0 : 0); a.bytes1 = (uint32_t)(b1 < lim ? b1 : 0); a.bytesdy = (uint32_t)(bd < lim ? bd : 0);
  *c4_out = c4;
  *dma_out = !c4 && b0 < lim && b1 < lim && bd < lim;
  a.pbn.x = nullptr; a.pbn.mean = p->src_bn_mean; a.pbn.rstd = p->src_bn_rstd; a.pbn.gamma = p->src_bn_gamma; a.pbn.beta = p->src_bn_beta;
  a.pbn.relu = p->src_bn_relu;
  return STP_OK;
}

// ---- grouped row-of-taps launch: host side -----------------------------------------------------------------------------------
// class of a layer = the output-channel tile of its kernel instance (layers of one group share it); 0 = not eligible
static int wg_group_bm(const stp_wgrad_params* p) {
  if (!p || !wgrad_row_eligible(p) || p->splits != 0) return 0;      // (a fused producer BatchNormalization: wgrad_row_eligible's rules)
  const int64_t lim = 1ll << 31;
  const int64_t P = (int64_t)p->N * p->Ho * p->Wo;
  if ((int64_t)p->N * p->Hs0 * p->Ws0 * p->C0 * 2 >= lim || (int64_t)p->N * p->Hv * p->Wv * p->C1 * 2 >= lim || P * p->Cout * 2 >= lim) return 0;
  return p->Cout <= 32 ? 32 : p->Cout <= 64 ? 64 : 128;
}
extern "C" int stp_wgrad_group_class(const stp_wgrad_params* p) { return wg_group_bm(p); }

struct WgGroupPlan {
  int bm = 0, max_slots_per_tile = 1;
  bool taps9 = false;      // all-taps tiles (BM x 576, one workgroup per CU) instead of row-of-taps tiles (BM x 192)
  std::vector<int> ntile_m, ntile_n, nsteps, tile0;
  std::vector<WgSeg> segs;
  std::vector<int> first;
  std::vector<WgTile> tiles;
  std::vector<int> slot_list;
};

static int wg_group_slots(int bm, bool taps9) {
  static const int cus = device_cu_count();
  static const int target = getenv("STP_WGRAD_GROUP_SLOTS") ? atoi(getenv("STP_WGRAD_GROUP_SLOTS")) : 0;
  if (target > 0) return target;
  if (taps9) return cus;                                 // 512 threads, 120-144 KB of LDS: one workgroup per CU
  return cus * (bm == 128 ? 2 : bm == 64 ? 3 : 4);     // co-resident workgroups: 77 KB of LDS / 224 registers at 128 channels, 52 KB / 144 and 40 KB / 114 below
}
// all-taps tiles for a group: every layer on a 16 / 32 / >= 64 pixel wide map (the kernel's three bodies), none with a fused producer
// BatchNormalization (that stays on the row-of-taps instance).  Default for the 128-channel class (profiles/r04f_*: the three groups
// 265 / 234 / 150 -> 210 / 173 / 117 us, 1150-1230 TFLOP/s, step 6.78 -> 6.64 ms on one box); STP_WGRAD_TAPS9=0 keeps the round-3 kernel.
static bool wg_group_taps9(const stp_wgrad_params* const* L, int n) {
  static const int mode = getenv("STP_WGRAD_TAPS9") ? atoi(getenv("STP_WGRAD_TAPS9")) : 1;      // 0: off; 1 (default): 128-channel class; 2: every class (tests)
  if (!mode) return false;
  // (the 64 / 32-channel classes: 32 / 16 output channels per wave against 9 taps x 16 input channels = 1.2 / 2 transpose reads per
  //  MFMA - LDS-bound, measured 25 % / 45 % slower than their row-of-taps instances, whose fabric traffic is already ~1x algorithmic)
  if (mode != 2 && wg_group_bm(L[0]) != 128) return false;
  for (int l = 0; l < n; ++l)
    if (L[l]->src_bn_mean || !(L[l]->Wo == 16 || L[l]->Wo == 32 || L[l]->Wo % 64 == 0)) return false;
  return true;
}

// The line: layer-major; inside a layer RANGE-major (a range = ~one workgroup chunk of steps), tile-minor - the tiles of a layer over
// the same pixel range are neighbours on the line, i.e. run at the same time on the same XCD (contiguous runs of the line per XCD)
// and share dY / x through its L2.  With the tile-major line of the first build a stage-1 layer (3 tiles x 4096 steps) put 43
// workgroups on 43 different pixel ranges of one tile: every operand byte came from HBM three times (541 TFLOP/s, 5.9 TB/s).
static int wg_group_plan(const stp_wgrad_params* const* L, int n, WgGroupPlan& g) {
  if (!L || n <= 0) return STP_E_BADARG;
  g.bm = wg_group_bm(L[0]);
  if (!g.bm) return STP_E_BADARG;
  g.taps9 = wg_group_taps9(L, n);
  int64_t total = 0;
  int ntiles = 0;
  for (int l = 0; l < n; ++l) {
    if (wg_group_bm(L[l]) != g.bm || !L[l]->src0 || !L[l]->dy || !L[l]->dw) return STP_E_BADARG;
    const int tm = ceil_div(L[l]->Cout, g.bm), tn = (g.taps9 ? 1 : 3) * ((L[l]->C0 + L[l]->C1) / 64);
    const int ns = (int)((int64_t)L[l]->N * L[l]->Ho * L[l]->Wo / 64);
    g.ntile_m.push_back(tm); g.ntile_n.push_back(tn); g.nsteps.push_back(ns); g.tile0.push_back(ntiles);
    ntiles += tm * tn;
    total += (int64_t)tm * tn * ns;
  }
  const int slots = wg_group_slots(g.bm, g.taps9);
  const int MINSEG = 8;       // no segment shorter than this unless its item is (a 3-stage pipeline fill + a slab per segment)
  int64_t chunk = (total + slots - 1) / slots;
  if (chunk < 2 * MINSEG) chunk = 2 * MINSEG;
  std::vector<std::vector<int>> lists;
  for (;; ++chunk) {
    g.segs.clear(); g.first.clear();
    lists.assign(ntiles, std::vector<int>());
    // items of the line: (layer, range, tile) -> steps [r * R, min(ns, (r + 1) * R))
    int l = 0, r = 0, t = 0, s = 0;
    auto ranges_of = [&](int l_) { int64_t nr = (g.nsteps[l_] + chunk / 2) / chunk; return (int)(nr < 1 ? 1 : nr); };
    auto rlen_of = [&](int l_) { return ceil_div(g.nsteps[l_], ranges_of(l_)); };
    while (l < n) {
      g.first.push_back((int)g.segs.size());
      int64_t rem = chunk;
      int mine = 0;
      while (rem > 0 && l < n) {
        const int R = rlen_of(l), i0 = r * R, i1 = (i0 + R < g.nsteps[l]) ? i0 + R : g.nsteps[l];
        const int avail = i1 - i0 - s;
        int take = (int)(rem < avail ? rem : avail);
        if (avail - take > 0 && avail - take < MINSEG) take = avail;             // do not leave a sliver of the item behind
        if (take < MINSEG && take < avail && mine > 0) break;                     // a sliver at the end of the chunk: the next workgroup takes it
        WgSeg sg = {l, t, i0 + s, i0 + s + take, (int)g.segs.size(), 0, 0, 0};
        lists[g.tile0[l] + t].push_back(sg.slot);
        g.segs.push_back(sg);
        ++mine;
        rem -= take; s += take;
        if (s == i1 - i0) {
          s = 0;
          if (++t == g.ntile_m[l] * g.ntile_n[l]) { t = 0; if (++r == ranges_of(l)) { r = 0; ++l; } }
        }
      }
    }
    g.first.push_back((int)g.segs.size());
    if ((int)g.first.size() - 1 <= slots) break;       // (slivers moved between neighbours can cost one workgroup more than the slots)
  }
  g.tiles.clear(); g.slot_list.clear(); g.max_slots_per_tile = 1;
  for (int l = 0; l < n; ++l)
    for (int t = 0; t < g.ntile_m[l] * g.ntile_n[l]; ++t) {
      const std::vector<int>& v = lists[g.tile0[l] + t];
      g.tiles.push_back(WgTile{l, t, (int)g.slot_list.size(), (int)v.size()});
      g.slot_list.insert(g.slot_list.end(), v.begin(), v.end());
      if ((int)v.size() > g.max_slots_per_tile) g.max_slots_per_tile = (int)v.size();
    }
  return STP_OK;
}

static size_t wg_align16(size_t v) { return (v + 15) & ~(size_t)15; }

extern "C" size_t stp_wgrad_group_table_bytes(const stp_wgrad_params* const* layers, int32_t n) {
  WgGroupPlan g;
  if (wg_group_plan(layers, n, g) != STP_OK) return 0;
  return wg_align16(sizeof(WgGroupHeader)) + wg_align16(sizeof(WgLayer) * n) + wg_align16(sizeof(WgSeg) * g.segs.size()) +
         wg_align16(sizeof(int) * g.first.size()) + wg_align16(sizeof(WgTile) * g.tiles.size()) + wg_align16(sizeof(int) * g.slot_list.size());
}

extern "C" size_t stp_wgrad_group_workspace_bytes(const stp_wgrad_params* const* layers, int32_t n) {
  WgGroupPlan g;
  if (wg_group_plan(layers, n, g) != STP_OK) return 0;
  return g.segs.size() * (size_t)g.bm * (g.taps9 ? 576 : 192) * sizeof(float);
}

// Fills `host_table` (stp_wgrad_group_table_bytes): the caller copies it to device memory and passes both to the launches
// (the header is read on the host, everything else on the device; the table holds the layers' device pointers).
extern "C" int stp_wgrad_group_build(const stp_wgrad_params* const* layers, int32_t n, void* host_table, size_t table_bytes) {
  WgGroupPlan g;
  int rc = wg_group_plan(layers, n, g);
  if (rc != STP_OK) return rc;
  if (!host_table || table_bytes < stp_wgrad_group_table_bytes(layers, n)) return STP_E_WORKSPACE;
  char* tb = (char*)host_table;
  WgGroupHeader hd = {};
  hd.magic = WG_GROUP_MAGIC; hd.bm = g.bm; hd.n_layers = n; hd.n_segs = (int)g.segs.size(); hd.n_wg = (int)g.first.size() - 1;
  hd.n_tiles = (int)g.tiles.size();
  size_t off = wg_align16(sizeof(WgGroupHeader));
  hd.off_layers = (int)off; off += wg_align16(sizeof(WgLayer) * n);
  hd.off_segs = (int)off; off += wg_align16(sizeof(WgSeg) * g.segs.size());
  hd.off_first = (int)off; off += wg_align16(sizeof(int) * g.first.size());
  hd.off_tiles = (int)off; off += wg_align16(sizeof(WgTile) * g.tiles.size());
  hd.off_slots = (int)off; off += wg_align16(sizeof(int) * g.slot_list.size());
  hd.total_bytes = (int)off;
  hd.reduce_lanes = g.max_slots_per_tile >= 12 ? 4 : 1;
  {
    static const int xcd_env = getenv("STP_WGRAD_GROUP_XCD") ? atoi(getenv("STP_WGRAD_GROUP_XCD")) : -1;
    // measured per group (profiles/r03f_group_xcd_ab.txt, r03g_*): one contiguous run per XCD wins (stage 1: 140 vs 181 us, stage 3: 224
    // vs 244) except where most of the work sits in layers of 64+ tiles over ONE pixel range (the 512-channel layers at 16 x 16 x 16:
    // 96 tiles x 64 steps - an XCD's 64 workgroups then fetch the same lines at the same time: 275 vs 247 us) -> identity there
    int64_t wide = 0, all = 0;
    for (int l = 0; l < n; ++l) {
      const int64_t wk = (int64_t)g.ntile_m[l] * g.ntile_n[l] * g.nsteps[l];
      all += wk;
      if (g.ntile_m[l] * g.ntile_n[l] >= 64) wide += wk;
    }
    hd.xcd = xcd_env >= 0 ? xcd_env : (2 * wide > all ? 0 : 1);
  }
  for (int l = 0; l < n; ++l)
    if (layers[l]->src_bn_mean) hd.pbn = 1;
  {
    // the v_mfma_f32_32x32x16 form of the 128-channel instance: OPT-IN (STP_WGRAD_TAPS9_M32=1) - measured equal to the 16 x 16 x 32 form
    // (profiles/r04j_m32_ab.txt: 7.33 vs 7.33 ms per step; the loop is not bound by the MFMA shape)
    static const bool m32 = getenv("STP_WGRAD_TAPS9_M32") && atoi(getenv("STP_WGRAD_TAPS9_M32")) == 1;
    hd.taps9 = g.taps9 ? ((m32 && g.bm == 128) ? 2 : 1) : 0;
  }
  memset(tb, 0, off);
  memcpy(tb, &hd, sizeof(hd));
  for (int l = 0; l < n; ++l) {
    WgLayer wl = {};
    WgradPlan w;
    bool c4, dma;
    rc = wgrad_fill(layers[l], (void*)tb, ~(size_t)0, wl.a, w, &c4, &dma);     // (workspace arguments: validation only, unused by the group)
    if (rc != STP_OK) return rc;
    if (!dma) return STP_E_BADARG;
    wl.a.out = nullptr;
    wl.a.ntile_m = g.ntile_m[l]; wl.a.ntile_n = g.ntile_n[l]; wl.a.nsteps = g.nsteps[l]; wl.a.steps_per_split = g.nsteps[l];
    wl.a.xcd = 0;
    wl.dw = layers[l]->dw; wl.accumulate = layers[l]->accumulate; wl.tile0 = g.tile0[l];
    memcpy(tb + hd.off_layers + sizeof(WgLayer) * l, &wl, sizeof(wl));
  }
  memcpy(tb + hd.off_segs, g.segs.data(), sizeof(WgSeg) * g.segs.size());
  memcpy(tb + hd.off_first, g.first.data(), sizeof(int) * g.first.size());
  memcpy(tb + hd.off_tiles, g.tiles.data(), sizeof(WgTile) * g.tiles.size());
  memcpy(tb + hd.off_slots, g.slot_list.data(), sizeof(int) * g.slot_list.size());
  return STP_OK;
}

static const WgGroupHeader* wg_group_header(const void* host_table) {
  const WgGroupHeader* hd = (const WgGroupHeader*)host_table;
  return (hd && hd->magic == WG_GROUP_MAGIC && (hd->bm == 32 || hd->bm == 64 || hd->bm == 128) && hd->n_wg > 0 && hd->n_tiles > 0) ? hd : nullptr;
}

extern "C" int stp_wgrad_group_partial(const void* host_table, const void* dev_table, void* workspace, size_t workspace_bytes, void* stream) {
  const WgGroupHeader* hd = wg_group_header(host_table);
  if (!hd || !dev_table || !workspace) return STP_E_BADARG;
  if ((size_t)hd->n_segs * hd->bm * (hd->taps9 ? 576 : 192) * sizeof(float) > workspace_bytes) return STP_E_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  if (hd->taps9) {
    // 3 stages x (dY tile + 256 halo rows of 128 bytes - the seg = 64 geometry, the largest)
    const size_t lds9 = (size_t)3 * ((hd->bm == 128 ? 16384 : 8192) + 256 * 128);
    static bool a128 = false, a64 = false, a32 = false;
#define STP_GROUP9_LAUNCH(BM_, FLAG_)                                                                                               \
  do {                                                                                                                              \
    auto kern = conv_wgrad_taps9_group_kernel<BM_, 3>;                                                                              \
    if (!FLAG_) {                                                                                                                   \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds9) != hipSuccess) \
        return STP_E_LAUNCH;                                                                                                        \
      FLAG_ = true;                                                                                                                 \
    }                                                                                                                               \
    hipLaunchKernelGGL(kern, dim3(hd->n_wg), dim3(512), lds9, s, (const char*)dev_table, (f32x4*)workspace);                        \
  } while (0)
    if (hd->bm == 128 && hd->taps9 == 2) {
      static bool a128m = false;
      auto kern = conv_wgrad_taps9_group_kernel<128, 3, true>;
      if (!a128m) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds9) != hipSuccess) return STP_E_LAUNCH;
        a128m = true;
      }
      hipLaunchKernelGGL(kern, dim3(hd->n_wg), dim3(512), lds9, s, (const char*)dev_table, (f32x4*)workspace);
    } else if (hd->bm == 128) STP_GROUP9_LAUNCH(128, a128);
    else if (hd->bm == 64) STP_GROUP9_LAUNCH(64, a64);
    else STP_GROUP9_LAUNCH(32, a32);
#undef STP_GROUP9_LAUNCH
    STP_LAUNCH_CHECK();
    return STP_OK;
  }
  const size_t lds = (size_t)3 * (64 * hd->bm * 2 + 72 * 128);
  static bool attr128 = false, attr64 = false, attr32 = false, attr128p = false, attr64p = false, attr32p = false;
#define STP_GROUP_LAUNCH(BM_, WM_, WN_, FLAG_)                                                                                      \
  do {                                                                                                                              \
    auto kern = hd->pbn ? conv_wgrad_row_group_pbn_kernel<BM_, WM_, WN_, 3> : conv_wgrad_row_group_kernel<BM_, WM_, WN_, 3>;        \
    bool& done = hd->pbn ? FLAG_##p : FLAG_;                                                                                        \
    if (lds > 64 * 1024 && !done) {                                                                                                 \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
        return STP_E_LAUNCH;                                                                                                        \
      done = true;                                                                                                                  \
    }                                                                                                                               \
    hipLaunchKernelGGL(kern, dim3(hd->n_wg), dim3(256), lds, s, (const char*)dev_table, (f32x4*)workspace);                         \
  } while (0)
  if (hd->bm == 128) STP_GROUP_LAUNCH(128, 2, 2, attr128);
  else if (hd->bm == 64) STP_GROUP_LAUNCH(64, 1, 4, attr64);
  else STP_GROUP_LAUNCH(32, 1, 4, attr32);
#undef STP_GROUP_LAUNCH
  STP_LAUNCH_CHECK();
  return STP_OK;
}

extern "C" int stp_wgrad_group_reduce(const void* host_table, const void* dev_table, const void* workspace, void* stream) {
  const WgGroupHeader* hd = wg_group_header(host_table);
  if (!hd || !dev_table || !workspace) return STP_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  const int sl = hd->reduce_lanes == 4 ? 4 : 1;
  if (hd->taps9) {
    const dim3 grid9(hd->n_tiles * (hd->bm * 576 / 4 / (256 / sl)));
#define STP_GROUP9_REDUCE(BM_)                                                                                                                       \
  do {                                                                                                                                               \
    if (sl == 4) hipLaunchKernelGGL((wgrad_group9_reduce_kernel<BM_, 4>), grid9, dim3(256), 0, s, (const char*)dev_table, (const f32x4*)workspace);  \
    else hipLaunchKernelGGL((wgrad_group9_reduce_kernel<BM_, 1>), grid9, dim3(256), 0, s, (const char*)dev_table, (const f32x4*)workspace);          \
  } while (0)
    if (hd->bm == 128 && hd->taps9 == 2) {
      if (sl == 4) hipLaunchKernelGGL((wgrad_group9_reduce_kernel<128, 4, true>), grid9, dim3(256), 0, s, (const char*)dev_table, (const f32x4*)workspace);
      else hipLaunchKernelGGL((wgrad_group9_reduce_kernel<128, 1, true>), grid9, dim3(256), 0, s, (const char*)dev_table, (const f32x4*)workspace);
    } else if (hd->bm == 128) STP_GROUP9_REDUCE(128);
    else if (hd->bm == 64) STP_GROUP9_REDUCE(64);
    else STP_GROUP9_REDUCE(32);
#undef STP_GROUP9_REDUCE
    STP_LAUNCH_CHECK();
    return STP_OK;
  }
  const int bpt = hd->bm * 192 / 4 / (256 / sl);
  const dim3 grid(hd->n_tiles * bpt);
#define STP_GROUP_REDUCE(BM_, WM_, WN_)                                                                                                     \
  do {                                                                                                                                      \
    if (sl == 4) hipLaunchKernelGGL((wgrad_group_reduce_kernel<BM_, WM_, WN_, 4>), grid, dim3(256), 0, s, (const char*)dev_table, (const f32x4*)workspace); \
    else hipLaunchKernelGGL((wgrad_group_reduce_kernel<BM_, WM_, WN_, 1>), grid, dim3(256), 0, s, (const char*)dev_table, (const f32x4*)workspace);         \
  } while (0)
  if (hd->bm == 128) STP_GROUP_REDUCE(128, 2, 2);
  else if (hd->bm == 64) STP_GROUP_REDUCE(64, 1, 4);
  else STP_GROUP_REDUCE(32, 1, 4);
#undef STP_GROUP_REDUCE
  STP_LAUNCH_CHECK();
  return STP_OK;
}

extern "C" int stp_conv2d_wgrad_kernel_id(const stp_wgrad_params* p) {
  if (!p) return 0;
  if (p->splits == 0 && stp_wgrad_sc_eligible(p)) return 1;
  if (wgrad_row_auto(p)) return p->Cout <= 64 ? 3 : 2;
  return 0;
}

// Phase 1: per-split partial sums into the workspace slabs.  `variant` as in launch_wgrad_tile.
extern "C" int stp_conv2d_wgrad_partial(const stp_wgrad_params* p, void* workspace, size_t workspace_bytes, int32_t variant,
                                        void* stream) {
  if (p && variant == 0 && p->splits == 0 && stp_wgrad_sc_eligible(p)) {
    if (!workspace || stp_conv2d_wgrad_workspace_bytes(p) > workspace_bytes) return STP_E_WORKSPACE;
    return stp_wgrad_sc_partial(p, workspace, stream);
  }
  if (p && p->src_bn_mean && !(variant == 4 || (variant == 0 && wgrad_row_auto(p)))) return STP_E_BADARG;   // fused producer BatchNormalization:
  WgradArgs a;                                                                                              // small-channel and row-of-taps kernels only
  WgradPlan w;
  bool c4, dma;
  const int rc = wgrad_fill(p, workspace, workspace_bytes, a, w, &c4, &dma);
  if (rc != STP_OK) return rc;
  hipStream_t s = (hipStream_t)stream;
  if (variant == 4 || (variant == 0 && wgrad_row_auto(p))) {   // row-of-taps kernel
    if (!wgrad_row_eligible(p) || !dma) return STP_E_BADARG;
    w = plan_wgrad_row(p);
    a.ntile_m = w.ntile_m; a.ntile_n = w.ntile_n; a.steps_per_split = w.steps_per_split; a.nsteps = w.nsteps;
    a.xcd = w.ntile_m == 1;
    static const int stages = getenv("STP_WGRAD_ROW_STAGES") ? atoi(getenv("STP_WGRAD_ROW_STAGES")) : 3;
    const size_t lds = (size_t)(stages == 2 ? 2 : 3) * (64 * w.bm * 2 + 72 * 128);    // 25 / 17 KB per stage
    static bool attr128 = false, attr64 = false, attr_dummy = true;
    if (stages == 2) {
      if (a.pbn.mean) return STP_E_BADARG;       // (the 2-stage what-if build has no fused producer BatchNormalization)
      if (w.tile == WG_TILE_ROW128) return launch_wg(conv_wgrad_row_kernel<128, 2, 2, 2>, a, lds, w.splits, attr_dummy, s);
      return launch_wg(conv_wgrad_row_kernel<64, 1, 4, 2>, a, lds, w.splits, attr_dummy, s);
    }
    if (a.pbn.mean) {      // fused producer BatchNormalization: its own instances
      static bool attr128p = false, attr64p = false;
      if (w.tile == WG_TILE_ROW128) return launch_wg(conv_wgrad_row_pbn_kernel<128, 2, 2, 3>, a, lds, w.splits, attr128p, s);
      return launch_wg(conv_wgrad_row_pbn_kernel<64, 1, 4, 3>, a, lds, w.splits, attr64p, s);
    }
    if (w.tile == WG_TILE_ROW128) return launch_wg(conv_wgrad_row_kernel<128, 2, 2, 3>, a, lds, w.splits, attr128, s);
    return launch_wg(conv_wgrad_row_kernel<64, 1, 4, 3>, a, lds, w.splits, attr64, s);
  }
  if (p->dtype == STP_H16)
    return c4 ? launch_wgrad_tile<bf16_t, true>(a, w, dma, variant, s) : launch_wgrad_tile<bf16_t, false>(a, w, dma, variant, s);
  return launch_wgrad_tile<float, false>(a, w, dma, variant, s);
}

// Phase 2: dw (+)= sum over slabs, fixed order.  `variant` must be the one given to the partial launch.
extern "C" int stp_conv2d_wgrad_reduce(const stp_wgrad_params* p, const void* workspace, int32_t variant, void* stream) {
  if (!p || !p->dw || !workspace) return STP_E_BADARG;
  if (variant == 4 || (variant == 0 && wgrad_row_auto(p) && !stp_wgrad_sc_eligible(p))) {
    if (!wgrad_row_eligible(p)) return STP_E_BADARG;
    const WgradPlan r = plan_wgrad_row(p);
    const int64_t per_split = (int64_t)r.ntile_m * r.ntile_n * r.bm * 192 / 4;   // float4 elements
    const int K = 9 * (p->C0 + p->C1);
    const int sl = r.splits >= 32 ? 16 : r.splits >= 8 ? 4 : 1;
    const dim3 grid(ceil_div(per_split, 256 / sl));
    hipStream_t s = (hipStream_t)stream;
    const f32x4* slabs = (const f32x4*)workspace;
#define STP_ROW_REDUCE(BM_, WM_, WN_, SL_) \
    hipLaunchKernelGGL((wgrad_reduce_row_kernel<BM_, WM_, WN_, SL_>), grid, dim3(256), 0, s, slabs, p->dw, per_split, r.splits, p->accumulate, \
                       r.ntile_m, p->C0 + p->C1, p->Cout, K)
    if (r.bm == 128) { if (sl == 16) STP_ROW_REDUCE(128, 2, 2, 16); else if (sl == 4) STP_ROW_REDUCE(128, 2, 2, 4); else STP_ROW_REDUCE(128, 2, 2, 1); }
    else { if (sl == 16) STP_ROW_REDUCE(64, 1, 4, 16); else if (sl == 4) STP_ROW_REDUCE(64, 1, 4, 4); else STP_ROW_REDUCE(64, 1, 4, 1); }
#undef STP_ROW_REDUCE
    STP_LAUNCH_CHECK();
    return STP_OK;
  }
  const WgradPlan w = variant == 0 ? plan_wgrad(p) : plan_wgrad_gemm(p);
  const int64_t count = (int64_t)p->Cout * p->KH * p->KW * (p->C0 + p->C1);
  if ((w.splits >= 64 && count <= (1 << 16)) || (w.splits >= 32 && count <= (1 << 19)))
    hipLaunchKernelGGL(wgrad_reduce_wide_kernel, dim3(ceil_div(count, 64)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)workspace, p->dw, count, w.splits, p->accumulate);
  else
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(ceil_div(count, 1024)), dim3(256), 0, (hipStream_t)stream, (const float*)workspace,
                       p->dw, count, w.splits, p->accumulate);
  STP_LAUNCH_CHECK();
  return STP_OK;
}

// ---- batched reduce of lone weight gradients (round 6) ---------------------------------------------------------------------------------
// The bottleneck ResNets' 1x1 layers never neighbour a layer of their own class, so each ran partial launch + reduce launch: 59 reduces per
// step on FPN/ResNet50, 30 on PSPNet/ResNet101 at ~9 us each (profiles/r06a_floor_config{3,4}.txt: 576 / 260 us = 3.7 % / 2.9 % of the
// step) for 1 - 16 MB of slabs - launch latency, not bytes.  The weight gradient feeds nothing but the optimizer: a layer whose slabs are
// the plain [splits][Cout * K] form keeps them in a workspace OF ITS OWN and its reduce joins a table; one launch (grid y = layer) reduces
// up to STP_WGRAD_REDUCE_BATCH layers.  Same 16-lane walk + fixed-shape tree as wgrad_reduce_wide_kernel: deterministic.
struct WgReduceDesc {
  const float* slabs;
  float* dw;
  int64_t count;
  int32_t splits, accumulate;
};

__global__ __launch_bounds__(256) void wgrad_reduce_batched_kernel(const WgReduceDesc* __restrict__ D) {
  const WgReduceDesc d = D[blockIdx.y];
  if ((int64_t)blockIdx.x * 64 >= d.count) return;                      // (workgroup-uniform: the grid is sized for the largest layer)
  __shared__ f32x4 sh[16][16];
  const int ev = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int64_t i = ((int64_t)blockIdx.x * 16 + ev) * 4;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (i < d.count) {
    int k = sl;
    for (; k + 112 < d.splits; k += 128) {      // eight loads in flight per lane; same order of additions as the one-by-one loop
      f32x4 a[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] = *reinterpret_cast<const f32x4*>(d.slabs + (size_t)(k + 16 * u) * d.count + i);
#pragma unroll
      for (int u = 0; u < 8; ++u) s += a[u];
    }
    for (; k < d.splits; k += 16) s += *reinterpret_cast<const f32x4*>(d.slabs + (size_t)k * d.count + i);
  }
  sh[sl][ev] = s;
  __syncthreads();
  for (int w = 8; w > 0; w >>= 1) {
    if (sl < w) sh[sl][ev] += sh[sl + w][ev];
    __syncthreads();
  }
  if (sl == 0 && i < d.count) {
    f32x4 r = sh[0][ev];
    if (d.accumulate) r += *reinterpret_cast<const f32x4*>(d.dw + i);
    *reinterpret_cast<f32x4*>(d.dw + i) = r;
  }
}

extern "C" size_t stp_wgrad_reduce_desc_bytes() { return sizeof(WgReduceDesc); }

// Fills descriptor `index` of a host table for the layer `p` whose partial launch (stp_conv2d_wgrad_partial, variant 0) wrote `workspace`.
// Returns the element count of the layer (> 0), or 0 when its slabs are not the plain [splits][Cout * KH * KW * C] form (row-of-taps
// kernels: fragment-major slabs) - such a layer keeps its own stp_conv2d_wgrad_reduce launch.
extern "C" int64_t stp_wgrad_reduce_desc_fill(void* host_table, int32_t index, const stp_wgrad_params* p, const void* workspace) {
  if (!host_table || index < 0 || !p || !p->dw || !workspace) return 0;
  if (wgrad_row_auto(p) && !stp_wgrad_sc_eligible(p)) return 0;
  const WgradPlan w = plan_wgrad(p);
  const int64_t count = (int64_t)p->Cout * p->KH * p->KW * (p->C0 + p->C1);
  if (w.splits < 1 || count <= 0 || (count & 3)) return 0;
  WgReduceDesc& d = reinterpret_cast<WgReduceDesc*>(host_table)[index];
  d.slabs = (const float*)workspace; d.dw = p->dw; d.count = count; d.splits = w.splits; d.accumulate = p->accumulate;
  return count;
}

extern "C" int stp_wgrad_reduce_batched(const void* table_dev, int32_t n, int64_t max_count, void* stream) {
  if (!table_dev || n <= 0 || max_count <= 0) return STP_E_BADARG;
  hipLaunchKernelGGL(wgrad_reduce_batched_kernel, dim3(ceil_div(max_count, 64), n), dim3(256), 0, (hipStream_t)stream, (const WgReduceDesc*)table_dev);
  STP_LAUNCH_CHECK();
  return STP_OK;
}

extern "C" int stp_conv2d_wgrad(const stp_wgrad_params* p, void* workspace, size_t workspace_bytes, void* stream) {
  const int rc = stp_conv2d_wgrad_partial(p, workspace, workspace_bytes, 0, stream);
  if (rc != STP_OK) return rc;
  return stp_conv2d_wgrad_reduce(p, workspace, 0, stream);
}
