// Shared pieces of the MFMA convolution kernels (conv_igemm.hip: per-tap operand tiles; conv_halo.hip: halo-resident
// activation slabs): argument block, MFMA wrappers, the fused epilogue.
#pragma once
#include <type_traits>
#include "common.h"
#include <cstdlib>

struct ConvArgs {
  const char* src0;
  const char* src1;
  const char* weight;
  const char* residual;
  const float* bias;
  char* dst0;
  char* dst1;
  int N, Hs0, Ws0, Hv, Wv, C0, C1, Ctot, mode;
  int KH, KW, stride, pad, Ho, Wo, Cout, Cd0, Cd1;
  int acc0, acc1, relu;
  int K, P, HoWo, wrows;
  int ntile_m, ntile_n;
  uint32_t bytes0, bytes1, bytesw;
  float* stats;  // optional [2][Cout][ntile_n] floats, or (stat_slots > 0) int64 fixed-point slots [2][Cout][stat_slots]
  int stat_slots;
  BnBack bnb;    // bnb.x != NULL: stats are the BatchNormalization-backward sums and dst receives the masked gradient
  BnBack pbn;    // pbn.mean != NULL (halo kernel): src0 is the tensor BEFORE a BatchNormalization(+activation), normalised in LDS
  FastDiv divC, divKW, divHoWo, divWo, divNtm;   // magic-number division: a runtime integer divide costs ~30 VALU instructions
  // zperm (uniform-tap buffer-DMA kernel, zero-inserted source = data gradient of a stride-2 convolution): the logical pixel order
  // is PARITY-CLASS major - class c = (ho & 1) * 2 + (wo & 1), then (n, ho >> 1, wo >> 1) - so every pixel of a tile meets the
  // zero-inserted grid the same way and the K loop visits only the taps that hit real samples (1, 2, 2 or 4 of 9; 1 or 0 of 1)
  int zperm, zPc, zH2W2, zW2, zcpt;              // pixels per class, (Ho/2)*(Wo/2), Wo/2, K-tiles per tap
  FastDiv divPc, divH2W2, divW2, divCpt;
  // upc (same kernel and pixel order, NEAREST-2x src0 of a 3x3 / pad 1 forward convolution): the taps over src0 collapse to the
  // 2 x 2 low-resolution pixels a parity class touches, multiplied by the class-summed weights weight_up (stp_conv_params)
  const char* weight_up;
  uint32_t byteswu;
  int upc, ucpt0, ucpt1;                          // K-tiles per collapsed tap (C0 / KE) and per src1 tap (C1 / KE)
  FastDiv divU0, divU1;
  // folded shortcut data gradient (zperm launches): fold_cpt more K-tiles for the (even, even) class, read from fold_src at the
  // centre tap with the rows of fold_weight ([wrows][C0])
  const char* fold_src;
  const char* fold_weight;
  int fold_cpt;
  uint32_t bytes_fold, bytesw_fold;
  int no_rm;      // 1: keep the fragment-order epilogue (STP_IGEMM_RM=0: A/B of the row-major one)
  int ep_generic; // 1: the halo kernel's epilogue takes its all-operands loop (STP_EPILOGUE_SPECIAL=0: A/B of the per-combination copies)
  int sum2x2;     // halo kernel, two destinations: dst0 = [N][Ho/2][Wo/2][Cd0] receives the 2 x 2 block sums of the first Cd0 channels (+ bnb)
  // group-level pre-reduction of the statistic columns (stp_conv_params.stats_group): see stats_group_finish
  float* sg_out;
  unsigned* sg_cnt;
  int sg_G;
  int d2s;        // halo kernel, stp_conv_params.s2d_dgrad: depth-to-space store of the 4 x Cq parity-class-major channels
};

// ---- group-level pre-reduction of the fused sums (round 5) ---------------------------------------------------------------------------
// The BatchNormalization that follows a convolution reads [2][C][tiles] partial sums.  Up to 128 columns the apply kernels reduce them in
// their own prologue (stp_bn_finalize_apply); the 256 ... 2048-column tables of the large feature maps needed a finalize LAUNCH in
// between - ~5 us of pure latency on the critical chain, 50 times per step.  Here the last-arriving workgroup of every G consecutive
// tiles sums the group's columns in tile order into a [2][C][tiles / G] table.  Visibility across CUs / XCDs (per-XCD L2s are not
// coherent: MI355X_MICROARCH.md, inter-workgroup visibility): the columns are stored WRITE-THROUGH (sc1), every wave drains its
// stores (asm s_waitcnt vmcnt(0): invisible to the wait-count pass), the workgroup meets at a barrier, ONE lane draws a device-scope
// ticket; the workgroup that draws the last ticket reads the columns with sc1 loads (L1 bypassed, the write-through stores dropped the
// lines from every L2).  Fixed membership, fixed order: the sums do not depend on who arrives last.
__device__ __forceinline__ void stats_store(const ConvArgs& a, size_t idx, float v) {
  if (a.sg_out) __hip_atomic_store(a.stats + idx, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else a.stats[idx] = v;
}
// Every thread of the workgroup calls this AFTER the column `tile_n` (rows cout0 .. cout0 + nch - 1 of both statistics, Cs channels per
// statistic, ntile columns) has been written with stats_store.  `word`: one LDS word nobody else touches until the workgroup ends.
template <int NT>
__device__ __forceinline__ void stats_group_finish(const ConvArgs& a, int Cs, int cout0, int nch, int tile_m, int tile_n, int ntile, int tid,
                                                   unsigned* word) {
  if (!a.sg_out) return;                                       // (launch-uniform)
  const int G = a.sg_G, g = tile_n / G, ngroups = (ntile + G - 1) / G;
  const int members = (ntile - g * G) < G ? (ntile - g * G) : G;
  unsigned* const cnt = a.sg_cnt + (size_t)tile_m * ngroups + g;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // this wave's column stores have left the CU
  __syncthreads();
  if (tid == 0) *word = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (*word != (unsigned)(members - 1)) return;                 // (workgroup-uniform) not the last of the group
  if (tid == 0) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch / graph replay
  for (int r = tid; r < 2 * nch; r += NT) {
    const int stat = r / nch, ch = cout0 + (r - stat * nch);
    if (ch >= Cs) continue;
    const float* col = a.stats + ((size_t)stat * Cs + ch) * ntile + (size_t)g * G;
    float v[16];                                                // G <= 16: all loads in flight before the first addition
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = j < members ? __hip_atomic_load(col + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
    float sum = v[0];
#pragma unroll
    for (int j = 1; j < 16; ++j) sum += v[j];                   // tile order; the zeros of a short group change nothing
    a.sg_out[((size_t)stat * Cs + ch) * ngroups + g] = sum;
  }
}
// group size that brings `cols` columns to <= 128 (0: nothing to do)
static inline int stats_group_size(int cols) {
  if (cols <= 128) return 0;
  int G = 2;
  while ((cols + G - 1) / G > 128) G *= 2;
  return G <= 16 ? G : 0;
}

// logical (parity-class major) pixel -> n, ho, wo
__device__ __forceinline__ void zperm_decode(const ConvArgs& a, int pl, int& n, int& ho, int& wo) {
  const int c = (int)fdiv((uint32_t)pl, a.divPc);
  const int q = pl - c * a.zPc;
  n = (int)fdiv((uint32_t)q, a.divH2W2);
  const int rem = q - n * a.zH2W2;
  const int y2 = (int)fdiv((uint32_t)rem, a.divW2);
  ho = 2 * y2 + (c >> 1);
  wo = 2 * (rem - y2 * a.zW2) + (c & 1);
}
__device__ __forceinline__ int zperm_pixel(const ConvArgs& a, int pl) {
  int n, ho, wo;
  zperm_decode(a, pl, n, ho, wo);
  return (n * a.Ho + ho) * a.Wo + wo;
}

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  __device__ static __forceinline__ void run(const u32x4& a, const u32x4& b, f32x4& c) {
    c = mfma16_16x16x32(a, b, c);
  }
};
template <> struct Mma<float> {
  // A lane holds k = 4*(lane>>4) + s, s = 0..3, of a 16-wide K chunk; step s multiplies the s-th
  // components.  A and B use the same (lane, s) -> k map, which is all the contraction needs.
  __device__ static __forceinline__ void run(const u32x4& a, const u32x4& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
  }
};

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }


// One K-tile of MFMAs for this wave: TM x TN fragments of 16x16, two 64-byte chunks per 128-byte row.
template <typename T, int BM, int BN, int WM, int WN>
__device__ __forceinline__ void compute_tile(const char* stage, int wm, int wn, int lr, int lg,
                                             f32x4 (&acc)[BM / WM / 16][BN / WN / 16]) {
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
  const char* sa = stage + (wm * (BM / WM)) * 128;
  const char* sb = stage + BM * 128 + (wn * (BN / WN)) * 128;
  // all fragment reads of the K-tile are issued before its first MFMA (the compiler would otherwise sink them
  // next to their users to save registers, exposing one LDS round trip per four MFMAs); the s_waitcnt counters
  // it inserts then release the first 64-byte chunk while the second is still in flight
  u32x4 fa[2][TM], fb[2][TN];
#if defined(STP_EXP) && STP_EXP == 3  // what-if: no LDS fragment reads either (pure MFMA issue rate)
#pragma unroll
  for (int c = 0; c < 2; ++c) {
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[c][i] = u32x4{(uint32_t)lr, (uint32_t)lg, (uint32_t)c, (uint32_t)i};
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[c][j] = u32x4{(uint32_t)lg, (uint32_t)lr, (uint32_t)j, (uint32_t)c};
  }
  asm volatile("" ::: "memory");
#else
#pragma unroll
  for (int c = 0; c < 2; ++c) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int row = i * 16 + lr;  // (wave row offset is a multiple of 16 -> row&7 unchanged)
      fa[c][i] = *reinterpret_cast<const u32x4*>(sa + row * 128 + (((c * 4 + lg) ^ (row & 7)) << 4));
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int row = j * 16 + lr;
      fb[c][j] = *reinterpret_cast<const u32x4*>(sb + row * 128 + (((c * 4 + lg) ^ (row & 7)) << 4));
    }
  }
#endif
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) Mma<T>::run(fa[c][i], fb[c][j], acc[i][j]);
}

// value as it will be read back from memory (the BatchNorm that follows normalises the STORED tensor)
__device__ __forceinline__ f32x4 stored(f32x4 v, const float*) { return v; }
__device__ __forceinline__ f32x4 stored(f32x4 v, const bf16_t*) {
  const uint32_t a = pack_bf16x2(v.x, v.y), b = pack_bf16x2(v.z, v.w);
  return f32x4{h16lo_to_f32(a), h16hi_to_f32(a), h16lo_to_f32(b), h16hi_to_f32(b)};
}

// bias, residual, ReLU, dual destination, optional accumulate, optional fused BatchNormalization statistics.
//
// Statistics: per-channel sum and sum of squares (of the values as STORED) over this workgroup's pixels.  The
// 16 lanes of a DPP row hold 16 different pixels of the same 4 channels -> row reduction with DPP adds, then
// the WN waves that share the channels are combined through LDS in a fixed order.  Layout written:
// stats[stat][channel][tile] (tile index contiguous, so the finalize reads coalesced).  The channel-tile loop
// is the OUTER loop so that only one pair of accumulators is live at a time (register pressure).
// PRE: the residual / BatchNormalization-backward x values of this lane's outputs were fetched before the K loop (bf16,
// buffer-DMA kernel) - the epilogue's own loads sit behind per-fragment branches and would be latency-serialised.
__device__ __forceinline__ f32x4 unpack_bf16x4(const u32x2 r) {
  return f32x4{h16lo_to_f32(r.x), h16hi_to_f32(r.x), h16lo_to_f32(r.y), h16hi_to_f32(r.y)};
}

// pixf(j) -> the flattened output pixel (n*Ho*Wo + ho*Wo + wo) of this lane's j-th fragment column, or < 0 to skip it.
// cof(i)  -> first of the 4 consecutive output channels (relative to the BM-channel tile) of this lane's i-th fragment row.
// A lane's accumulators are viewed as TM x TN fragments of 4 channels x 1 pixel; the 16 lanes of a DPP row hold 16 different
// pixels of the same channels.  `vn` of WNN = which of the WNN groups of DPP rows that share this lane's channels (waves along
// the pixel dimension, times the pixel halves inside a 32x32 MFMA tile), NT threads per workgroup.
// prea (optional): the ACCUMULATE operands of dst0 (what the destination holds), fetched before the K loop like `pre` - the
// parity-class data gradients that complete a multi-consumer gradient (accumulate0 + BatchNormalization-backward sums) otherwise
// pay one dependent load round trip per fragment here.
template <typename T, int TM, int TN, int BM, int WNN, int NT, bool PRE, typename PixF, typename CoF>
__device__ __forceinline__ void epilogue_cf(const ConvArgs& a, int cout0, int vn, int lr, f32x4 (&acc)[TM][TN], char* smem,
                                            int tile_n, const u32x2 (*pre)[TN], PixF pixf, CoF cof, const u32x2 (*prea)[TN] = nullptr) {
  const T* res = reinterpret_cast<const T*>(a.residual);
  float* red = reinterpret_cast<float*>(smem);  // [WN][BM][2], valid after the barrier below
  if (a.stats) __syncthreads();                 // the K-loop's LDS tiles are dead from here on
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int co = cout0 + cof(i);
    f32x4 ss = {0.f, 0.f, 0.f, 0.f}, qq = {0.f, 0.f, 0.f, 0.f};
    if (co < a.Cout) {
      BnBackCh bk;
      if (a.bnb.x) bk = bnback_load(a.bnb, co);  // Cout % 4 == 0 in this mode
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int pm = pixf(j);
        if (pm < 0) continue;
        f32x4 v = acc[i][j];
        if (co + 3 < a.Cout) {
          if (a.bias) v += *reinterpret_cast<const f32x4*>(a.bias + co);
          if (res) {
            if constexpr (PRE) v += unpack_bf16x4(pre[i][j]);
            else v += load4(res + (size_t)pm * a.Cout + co);
          }
          T* d;
          bool accum;
          if (co < a.Cd0) { d = reinterpret_cast<T*>(a.dst0) + (size_t)pm * a.Cd0 + co; accum = a.acc0; }
          else { d = reinterpret_cast<T*>(a.dst1) + (size_t)pm * a.Cd1 + (co - a.Cd0); accum = a.acc1; }
          if (accum) {
            if (PRE && prea && co < a.Cd0) v += unpack_bf16x4(prea[i][j]);
            else v += load4(d);
          }
          if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
          if (a.bnb.x) {
            f32x4 xv;
            if constexpr (PRE) xv = unpack_bf16x4(pre[i][j]);
            else xv = load4(reinterpret_cast<const T*>(a.bnb.x) + (size_t)pm * a.Cout + co);
            v = bnback_apply(bk, a.bnb.relu, xv, stored(v, (const T*)nullptr), ss, qq);
          }
#if defined(STP_EXP) && STP_EXP == 4   // what-if: no output stores (the branch is never taken, the values stay live)
          if (a.P < 0)
#endif
          store4(d, v);
          if (a.stats && !a.bnb.x) {
            const f32x4 sv = stored(v, (const T*)nullptr);
            ss += sv;
            qq += sv * sv;
          }
        } else {
          // ragged channel tail (e.g. the 1-class head): scalar path
          for (int r = 0; r < 4 && co + r < a.Cout; ++r) {
            const int c1 = co + r;
            float x = v[r];
            if (a.bias) x += a.bias[c1];
            if (res) x += Elem<T>::load(res + (size_t)pm * a.Cout + c1);
            T* d;
            bool accum;
            if (c1 < a.Cd0) { d = reinterpret_cast<T*>(a.dst0) + (size_t)pm * a.Cd0 + c1; accum = a.acc0; }
            else { d = reinterpret_cast<T*>(a.dst1) + (size_t)pm * a.Cd1 + (c1 - a.Cd0); accum = a.acc1; }
            if (accum) x += Elem<T>::load(d);
            if (a.relu) x = fmaxf(x, 0.f);
            Elem<T>::store(d, x);
          }
        }
      }
    }
    if (a.stats) {  // wave-uniform
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float s = row_sum16_to_lane15(ss[e]), q = row_sum16_to_lane15(qq[e]);
        if (lr == 15) {
          const int cl = cof(i) + e;
          red[(vn * BM + cl) * 2] = s;
          red[(vn * BM + cl) * 2 + 1] = q;
        }
      }
    }
  }
  if (a.stats) {
    __syncthreads();
    for (int c = threadIdx.x; c < BM; c += NT) {
      if (cout0 + c >= a.Cout) continue;
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int w = 0; w < WNN; ++w) { s += red[(w * BM + c) * 2]; q += red[(w * BM + c) * 2 + 1]; }
      if (a.stat_slots) {
        long long* sl = reinterpret_cast<long long*>(a.stats);
        slot_add(sl, a.stat_slots, cout0 + c, tile_n, s);
        slot_add(sl, a.stat_slots, a.Cout + cout0 + c, tile_n, q);
        continue;
      }
      a.stats[(size_t)(cout0 + c) * a.ntile_n + tile_n] = s;
      a.stats[((size_t)a.Cout + cout0 + c) * a.ntile_n + tile_n] = q;
    }
  }
}

// 16x16 MFMA fragments: row i = channels cw + i*16 + (lane>>4)*4 .. +3, the wave index along pixels selects the reduction slot
template <typename T, int TM, int TN, int BM, int WNN, int NT, bool PRE, typename PixF>
__device__ __forceinline__ void epilogue_px(const ConvArgs& a, int cout0, int cw, int wn, int lr, int lg, f32x4 (&acc)[TM][TN], char* smem,
                                            int tile_n, const u32x2 (*pre)[TN], PixF pixf, const u32x2 (*prea)[TN] = nullptr) {
  epilogue_cf<T, TM, TN, BM, WNN, NT, PRE>(a, cout0, wn, lr, acc, smem, tile_n, pre, pixf, [cw, lg](int i) { return cw + i * 16 + lg * 4; }, prea);
}

// linear pixel tiles (conv_igemm.hip): the BN pixels of a tile are consecutive flattened pixels
template <typename T, int BM, int BN, int WM, int WN, bool PRE = false>
__device__ __forceinline__ void epilogue(const ConvArgs& a, int cout0, int pix0, int wm, int wn, int lr, int lg,
                                         f32x4 (&acc)[BM / WM / 16][BN / WN / 16], char* smem, int tile_n,
                                         const u32x2 (*pre)[BN / WN / 16] = nullptr) {
  const int pb = pix0 + wn * (BN / WN) + lr, P = a.P;
  epilogue_px<T, BM / WM / 16, BN / WN / 16, BM, WN, 256, PRE>(a, cout0, wm * (BM / WM), wn, lr, lg, acc, smem, tile_n, pre,
                                                               [pb, P](int j) { const int pm = pb + j * 16; return pm < P ? pm : -1; });
}

// ------------------------------------------------------------------------------------------------------------------------
// ROW-MAJOR epilogue for the linear-pixel tiles of conv_igemm.hip (16-bit storage, 256 threads, 16 x 16 MFMA fragments).
// In the MFMA C layout a lane owns 4 channels of one pixel: residual / accumulate / BatchNormalization-input loads and the stores
// are 8-byte accesses scattered over 16 pixel rows per instruction, one dependent round trip per fragment, and the per-channel sums
// cost a DPP reduction per fragment (scratch/igemm_ep_bench.py: 1x1 64 -> 256 @ 4 x 256 x 256 plain 51 us, with a residual 98 us, with
// the BatchNormalization-backward sums 126 us).  Here - as in conv_halo.hip's epilogue_rm - the workgroup's fp32 tile goes through
// LDS ([pixel][BM channels], rows padded by 16 bytes) and a thread owns 8 CONSECUTIVE channels of a pixel: 16-byte coalesced
// operand loads issued up front, 16-byte stores, sums as plain register adds + one fixed-order reduction per workgroup.
// Needs Cout, Cd0 and Cout - Cd0 multiples of 8 (otherwise the caller keeps the fragment-order epilogue above).
typedef float f32x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2v unpack_bf16x2(uint32_t w) { return f32x2v{h16lo_to_f32(w), h16hi_to_f32(w)}; }

template <int BM, int BN, int WM, int WN, typename PixF>
__device__ __forceinline__ void epilogue_rm_lin(const ConvArgs& a, f32x4 (&acc)[BM / WM / 16][BN / WN / 16], char* smem, int cout0, int tile_n,
                                                int wm, int wn, int lr, int lg, int tid, PixF pixf) {
  typedef bf16_t T;
  constexpr int NT = 256, TM = BM / WM / 16, TN = BN / WN / 16;
  constexpr int RS = BM * 4 + 16;                            // staged row: BM floats + 16 bytes of padding
  constexpr int CG = BM / 8, PP = NT / CG, NP = BN / PP;     // channel groups of 8, pixels per pass, passes
  static_assert(BM % 8 == 0 && NT % CG == 0 && BN % PP == 0, "tile shape");
  const int c8 = tid % CG, p0 = tid / CG;
  const int co = cout0 + c8 * 8;
  const bool cok = co < a.Cout;                               // Cout % 8 == 0: a group is valid as a whole
  const bool first = co < a.Cd0;
  T* const dbase = first ? reinterpret_cast<T*>(a.dst0) + co : reinterpret_cast<T*>(a.dst1) + (co - a.Cd0);
  const int dC = first ? a.Cd0 : a.Cd1;
  const bool accum = first ? a.acc0 : a.acc1;
  const bool bnb = a.bnb.x != nullptr, st = a.stats != nullptr;
  const T* res = bnb ? reinterpret_cast<const T*>(a.bnb.x) : reinterpret_cast<const T*>(a.residual);

  // global operands of this thread's NP pixels: issued before the staging so their latency hides under it
  int pm[NP];
  u32x4 opr[NP], opa[NP];
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    pm[k] = pixf(p0 + k * PP);
    opr[k] = opa[k] = u32x4{0u, 0u, 0u, 0u};
  }
  if (cok) {
    if (res) {
#pragma unroll
      for (int k = 0; k < NP; ++k)
        if (pm[k] >= 0) opr[k] = *reinterpret_cast<const u32x4*>(res + (size_t)pm[k] * a.Cout + co);
    }
    if (accum) {
#pragma unroll
      for (int k = 0; k < NP; ++k)
        if (pm[k] >= 0) opa[k] = *reinterpret_cast<const u32x4*>(dbase + (size_t)pm[k] * dC);
    }
  }
  lds_barrier();                                              // every wave has left the K loop: the operand ring is dead
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int px = wn * (BN / WN) + j * 16 + lr, c = wm * (BM / WM) + i * 16 + lg * 4;
      *reinterpret_cast<f32x4*>(smem + px * RS + c * 4) = acc[i][j];
    }
  lds_barrier();

  f32x2v ss[4], qq[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) ss[e] = qq[e] = f32x2v{0.f, 0.f};
  if (cok) {
    f32x2v bias2[4], ksc[4], ksh[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) bias2[e] = ksc[e] = ksh[e] = f32x2v{0.f, 0.f};
    if (!bnb && a.bias) {
#pragma unroll
      for (int e = 0; e < 4; ++e) bias2[e] = *reinterpret_cast<const f32x2v*>(a.bias + co + 2 * e);
    }
    if (bnb) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float r = a.bnb.rstd[co + e], sc = a.bnb.gamma ? r * a.bnb.gamma[co + e] : r;
        ksc[e >> 1][e & 1] = sc;
        ksh[e >> 1][e & 1] = (a.bnb.beta ? a.bnb.beta[co + e] : 0.f) - a.bnb.mean[co + e] * sc;
      }
    }
    const bool relu = a.relu != 0;
    const float alo = a.bnb.relu ? __uint_as_float(1u) : -__builtin_inff();
    const float ahi = a.bnb.relu == 2 ? __uint_as_float(0x40bfffffu) : __builtin_inff();      // largest float below 6
    // The operands of a launch (BatchNormalization-backward form, statistics, bias, residual, accumulate, ReLU) are launch-uniform; as
    // run-time conditions inside the unrolled element loop they were if-converted into selects, and EVERY launch paid the clamp /
    // compare / select / fma of the backward form, the sums of the statistics form and the adds of all optional operands (the same
    // finding as conv_halo.hip's epilogue, DESIGN 3.11: the epilogue is VALU-issue bound).  The loop is instantiated per combination;
    // the ones the networks use get their own copy, the rest takes the all-operands copy (STP_EPILOGUE_SPECIAL=0: always).
    // B_: backward form; S_: statistics; HB / HR / HA: bias / residual / accumulate; HL: ReLU (backward form: the fused activation is a
    // plain ReLU - the gradient passes iff the re-derived output is > 0, the clamp window of ReLU being [smallest positive, inf]).
    auto passes = [&](auto b_, auto s_, auto hb_, auto hr_, auto ha_, auto hl_) {
      constexpr bool B_ = decltype(b_)::value, S_ = decltype(s_)::value, HB = decltype(hb_)::value, HR = decltype(hr_)::value,
                     HA = decltype(ha_)::value, HL = decltype(hl_)::value;
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        if (pm[k] < 0) continue;
        const int px = p0 + k * PP;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(smem + px * RS + c8 * 32), v1 = *reinterpret_cast<const f32x4*>(smem + px * RS + c8 * 32 + 16);
        f32x2v v[4] = {f32x2v{v0.x, v0.y}, f32x2v{v0.z, v0.w}, f32x2v{v1.x, v1.y}, f32x2v{v1.z, v1.w}};
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (!B_) {
            if (HB) v[e] += bias2[e];
            if (HR) v[e] += unpack_bf16x2(opr[k][e]);
          }
          if (HA) v[e] += unpack_bf16x2(opa[k][e]);
          if (!B_ && HL) v[e] = f32x2v{fmaxf(v[e].x, 0.f), fmaxf(v[e].y, 0.f)};
          if (!B_) o[e] = pack_bf16x2(v[e].x, v[e].y);
          if (S_ && !B_) {
            const f32x2v sv = unpack_bf16x2(o[e]);
            ss[e] += sv;
            qq[e] += sv * sv;
          }
          if (B_) {   // masked gradient (the activation mask re-derived from the BatchNormalization input with the forward's fma), rounded ONCE
            const f32x2v xv = unpack_bf16x2(opr[k][e]);
            const f32x2v tt = xv * ksc[e] + ksh[e];
            const bool on0 = HL ? tt.x > 0.f : __builtin_amdgcn_fmed3f(tt.x, alo, ahi) == tt.x;
            const bool on1 = HL ? tt.y > 0.f : __builtin_amdgcn_fmed3f(tt.y, alo, ahi) == tt.y;
            o[e] = pack_bf16x2(on0 ? v[e].x : 0.f, on1 ? v[e].y : 0.f);
            const f32x2v g = unpack_bf16x2(o[e]);
            ss[e] += g;
            qq[e] += g * xv;
          }
        }
        *reinterpret_cast<u32x4*>(dbase + (size_t)pm[k] * dC) = o;
      }
    };
    // every operand behind its own launch-uniform test, as the one loop of the earlier rounds had it
    auto passes_all = [&]() {
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        if (pm[k] < 0) continue;
        const int px = p0 + k * PP;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(smem + px * RS + c8 * 32), v1 = *reinterpret_cast<const f32x4*>(smem + px * RS + c8 * 32 + 16);
        f32x2v v[4] = {f32x2v{v0.x, v0.y}, f32x2v{v0.z, v0.w}, f32x2v{v1.x, v1.y}, f32x2v{v1.z, v1.w}};
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (!bnb) {
            v[e] += bias2[e];
            if (res) v[e] += unpack_bf16x2(opr[k][e]);
          }
          if (accum) v[e] += unpack_bf16x2(opa[k][e]);
          if (!bnb && relu) v[e] = f32x2v{fmaxf(v[e].x, 0.f), fmaxf(v[e].y, 0.f)};
          o[e] = pack_bf16x2(v[e].x, v[e].y);
          if (st && !bnb) {
            const f32x2v sv = unpack_bf16x2(o[e]);
            ss[e] += sv;
            qq[e] += sv * sv;
          }
          if (bnb) {   // dY as stored -> masked gradient (the activation mask re-derived from the BatchNormalization input with the forward's fma)
            const f32x2v xv = unpack_bf16x2(opr[k][e]), dy = unpack_bf16x2(o[e]);
            const f32x2v tt = xv * ksc[e] + ksh[e];
            const f32x2v g = f32x2v{__builtin_amdgcn_fmed3f(tt.x, alo, ahi) == tt.x ? dy.x : 0.f, __builtin_amdgcn_fmed3f(tt.y, alo, ahi) == tt.y ? dy.y : 0.f};
            ss[e] += g;
            qq[e] += g * xv;
            o[e] = pack_bf16x2(g.x, g.y);
          }
        }
        *reinterpret_cast<u32x4*>(dbase + (size_t)pm[k] * dC) = o;
      }
    };
    typedef std::true_type Y_;
    typedef std::false_type N_;
    const bool hb = !bnb && a.bias != nullptr, hr = !bnb && res != nullptr, hl = !bnb && relu;
    if (a.ep_generic) passes_all();
    else if (bnb) {
      const bool r1 = a.bnb.relu == 1;
      if (accum) { if (r1) passes(Y_(), Y_(), N_(), N_(), Y_(), Y_()); else passes(Y_(), Y_(), N_(), N_(), Y_(), N_()); }
      else { if (r1) passes(Y_(), Y_(), N_(), N_(), N_(), Y_()); else passes(Y_(), Y_(), N_(), N_(), N_(), N_()); }
    } else if (st) {
      if (!hb && !hr && !accum && !hl) passes(N_(), Y_(), N_(), N_(), N_(), N_());
      else if (!hb && hr && !accum && !hl) passes(N_(), Y_(), N_(), Y_(), N_(), N_());
      else if (!hb && !hr && accum && !hl) passes(N_(), Y_(), N_(), N_(), Y_(), N_());
      else passes_all();
    } else {
      if (!hb && !hr && !accum && !hl) passes(N_(), N_(), N_(), N_(), N_(), N_());
      else if (!hb && hr && !accum && !hl) passes(N_(), N_(), N_(), Y_(), N_(), N_());
      else if (!hb && !hr && accum && !hl) passes(N_(), N_(), N_(), N_(), Y_(), N_());
      else if (hb && !hr && !accum) { if (hl) passes(N_(), N_(), Y_(), N_(), N_(), Y_()); else passes(N_(), N_(), Y_(), N_(), N_(), N_()); }
      else passes_all();
    }
  }
  if (st) {     // (workgroup-uniform)
    if (bnb && cok) {   // sum g * xhat = rstd * (sum g * x - mean * sum g), per thread (linear, so the partition does not matter)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float mu = a.bnb.mean[co + e], rs = a.bnb.rstd[co + e];
        qq[e >> 1][e & 1] = rs * (qq[e >> 1][e & 1] - mu * ss[e >> 1][e & 1]);
      }
    }
    // fixed-order reduction over the NT / CG threads that share a channel group (the shape of conv_halo.hip's, for 256 threads)
    constexpr int J = NT / CG, NOUT = (CG * 16 < NT) ? CG * 16 : NT, NPART = NT / NOUT, JP = J / NPART, KS = NT + CG;
    static_assert(NPART >= 1 && J % NPART == 0 && CG * 16 <= NT * 16, "reduction shape");
    lds_barrier();                                          // the staged tile is dead
    float* r1 = reinterpret_cast<float*>(smem);               // [16][KS]
    float* r2 = r1 + 16 * KS;                                 // [NPART][CG * 16]
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      r1[(e * 4 + 0) * KS + tid] = ss[e].x; r1[(e * 4 + 1) * KS + tid] = ss[e].y;
      r1[(e * 4 + 2) * KS + tid] = qq[e].x; r1[(e * 4 + 3) * KS + tid] = qq[e].y;
    }
    lds_barrier();
    // CG * 16 outputs (k = 0..15 per channel group); each is the sum of J threads' partials: NPART threads take JP of them each
    for (int o = tid; o < CG * 16 * NPART; o += NT) {
      const int oo = o % (CG * 16), q = o / (CG * 16);
      const float* src = r1 + (oo / CG) * KS + (oo % CG) + CG * (q * JP);
      float acc_ = 0.f;
#pragma unroll
      for (int j = 0; j < JP; ++j) acc_ += src[CG * j];
      r2[q * (CG * 16) + oo] = acc_;
    }
    lds_barrier();
    if (tid < CG * 16) {
      float tot = 0.f;
#pragma unroll
      for (int q = 0; q < NPART; ++q) tot += r2[q * (CG * 16) + tid];
      const int k = tid / CG, ch = (tid % CG) * 8 + (k >> 2) * 2 + (k & 1), stat = (k >> 1) & 1;
      if (cout0 + ch < a.Cout) stats_store(a, ((size_t)stat * a.Cout + cout0 + ch) * a.ntile_n + tile_n, tot);
    }
    stats_group_finish<NT>(a, a.Cout, cout0, BM, cout0 / BM, tile_n, a.ntile_n, tid, reinterpret_cast<unsigned*>(smem));
  }
}
__host__ __device__ __forceinline__ bool epilogue_rm_ok(const ConvArgs& a) {
  return !a.no_rm && !(a.Cout & 7) && !(a.Cd0 & 7) && !(a.Cd1 & 7) && !a.stat_slots;
}

// stp_conv_params -> ConvArgs (validation shared by every MFMA convolution kernel)
static inline int fill_args(const stp_conv_params* p, ConvArgs& a, bool* c4_out, int* ut_out) {
  if (!p || !p->src0 || !p->weight || !p->dst0) return STP_E_BADARG;
  if (p->dtype != STP_F32 && p->dtype != STP_H16) return STP_E_BADARG;
  const int vec = p->dtype == STP_H16 ? 8 : 4;
  const int sz = p->dtype == STP_H16 ? 2 : 4;
  const int ke = 128 / sz;
  const bool c4 = (p->dtype == STP_H16) && p->C0 == 4 && p->C1 == 0;
  if (c4) {
    if ((p->KW & 1) || p->src0_mode != STP_SRC_DIRECT) return STP_E_BADARG;
  } else if ((p->C0 % vec) || (p->C1 % vec)) {
    return STP_E_BADARG;
  }
  if (p->C1 > 0 && !p->src1) return STP_E_BADARG;
  if (p->Cd0 <= 0 || p->Cd0 > p->Cout || (p->Cd0 < p->Cout && (!p->dst1 || (p->Cd0 & 3) || ((p->Cout - p->Cd0) & 3))))
    return STP_E_BADARG;
  if (p->N <= 0 || p->Ho <= 0 || p->Wo <= 0 || p->Cout <= 0 || p->KH <= 0 || p->KW <= 0 || p->stride <= 0) return STP_E_BADARG;
  a.src0 = (const char*)p->src0; a.src1 = (const char*)p->src1; a.weight = (const char*)p->weight;
  a.residual = (const char*)p->residual; a.bias = p->bias; a.dst0 = (char*)p->dst0; a.dst1 = (char*)p->dst1;
  a.N = p->N; a.Hs0 = p->Hs0; a.Ws0 = p->Ws0; a.Hv = p->Hv; a.Wv = p->Wv; a.C0 = p->C0; a.C1 = p->C1;
  a.Ctot = p->C0 + p->C1; a.mode = p->src0_mode;
  a.KH = p->KH; a.KW = p->KW; a.stride = p->stride; a.pad = p->pad; a.Ho = p->Ho; a.Wo = p->Wo;
  a.Cout = p->Cout; a.Cd0 = p->Cd0; a.Cd1 = p->Cout - p->Cd0;
  a.acc0 = p->accumulate0; a.acc1 = p->accumulate1; a.relu = p->relu;
  a.K = p->KH * p->KW * a.Ctot;
  const int64_t P = (int64_t)p->N * p->Ho * p->Wo;
  if (P >= (1ll << 31)) return STP_E_BADARG;
  a.P = (int)P; a.HoWo = p->Ho * p->Wo; a.wrows = round_up(p->Cout, 16);
  a.divC = make_fastdiv((uint32_t)a.Ctot); a.divKW = make_fastdiv((uint32_t)a.KW);
  a.divHoWo = make_fastdiv((uint32_t)(p->Ho * p->Wo)); a.divWo = make_fastdiv((uint32_t)p->Wo);
  const int64_t lim = 1ll << 31;
  const int64_t b0 = (int64_t)p->N * p->Hs0 * p->Ws0 * p->C0 * sz;
  const int64_t b1 = (int64_t)p->N * p->Hv * p->Wv * p->C1 * sz;
  const int64_t bw = (int64_t)a.wrows * a.K * sz;
  a.bytes0 = (uint32_t)(b0 < lim ? b0 : 0); a.bytes1 = (uint32_t)(b1 < lim ? b1 : 0); a.bytesw = (uint32_t)(bw < lim ? bw : 0);
  a.ntile_m = a.ntile_n = 0;
  a.zperm = 0;
  a.weight_up = (const char*)p->weight_up; a.byteswu = 0; a.upc = 0; a.ucpt0 = a.ucpt1 = 1;
  a.fold_src = a.fold_weight = nullptr; a.fold_cpt = 0; a.bytes_fold = a.bytesw_fold = 0;
  {
    static const int no_rm = (getenv("STP_IGEMM_RM") && atoi(getenv("STP_IGEMM_RM")) == 0) ? 1 : 0;
    a.no_rm = no_rm;
    static const int ep_generic = (getenv("STP_EPILOGUE_SPECIAL") && atoi(getenv("STP_EPILOGUE_SPECIAL")) == 0) ? 1 : 0;
    a.ep_generic = ep_generic;
  }
  a.stats = p->stats_partial;
  a.stat_slots = p->stats_slots;
  if (a.stats && ((p->Cout & 3) || (p->Cd0 != p->Cout && !p->dst_sum2x2))) return STP_E_BADARG;   // (two destinations: only the 2 x 2-summed form has sums)
  a.sum2x2 = p->dst_sum2x2 ? 1 : 0;
  a.sg_out = nullptr; a.sg_cnt = nullptr; a.sg_G = 0;
  a.d2s = p->s2d_dgrad ? 1 : 0;
  if (p->stats_group > 1) {        // (the launcher checks that its kernel has the grouped epilogue and that G is the one it would choose)
    if (!p->stats_group_out || !p->stats_group_counters || !a.stats || a.stat_slots || p->stats_group > 16 || (p->stats_group & (p->stats_group - 1)))
      return STP_E_BADARG;
    a.sg_out = p->stats_group_out; a.sg_cnt = p->stats_group_counters; a.sg_G = p->stats_group;
  }
  if (a.stat_slots && (!a.stats || (a.stat_slots & (a.stat_slots - 1)) || a.stat_slots > 64)) return STP_E_BADARG;
  a.bnb.x = (const char*)p->bnb_x; a.bnb.mean = p->bnb_mean; a.bnb.rstd = p->bnb_rstd; a.bnb.gamma = p->bnb_gamma;
  a.bnb.beta = p->bnb_beta; a.bnb.relu = p->bnb_relu;
  a.pbn.x = nullptr; a.pbn.mean = p->src_bn_mean; a.pbn.rstd = p->src_bn_rstd; a.pbn.gamma = p->src_bn_gamma; a.pbn.beta = p->src_bn_beta;
  a.pbn.relu = p->src_bn_relu;
  *c4_out = c4;
  *ut_out = 0;
  if (!c4 && b0 < lim && b1 < lim && bw < lim) {
    if ((a.Ctot % ke == 0) && (a.C0 % ke == 0)) *ut_out = 1;
    else if (a.C1 == 0 && (ke % a.Ctot == 0)) *ut_out = 2;
  }
  return STP_OK;
}

