// 3x3 / stride-1 convolution on MFMA with HALO-RESIDENT activation slabs (gfx950, bf16) - forward and data-gradient
// of the ResNet / decoder 3x3 layers with 128..512 channels.
//
// conv_igemm.hip stages one [pixels][64 channels] operand tile PER FILTER TAP: every activation byte travels L2 -> LDS
// nine times, and with the weights that is (BM + BN) * 128 bytes of LDS-DMA per K-step - the launches of the 256/512-channel
// layers are bound by that path (what-if build without MFMA work: 23-35 us of a 40-46 us launch, ~18 TB/s of L2 -> LDS).
// Here a workgroup owns a TH x 16 block of output pixels of ONE image and keeps, per 64-channel chunk ("slab"), the
// (TH+2) x 18 input pixels that the nine taps of the block touch resident in LDS: a tap is just a different LDS row
// offset of the same slab.  Only the weights stream: BM x 64 per K-step.
//
// K order: slab-major, tap-minor (k = slab * 9 + tap; the matching weight column is tap * Cin + slab * 64), so the first
// nine K-steps need only slab 0, and slab s+1 streams in (one 64-pixel pass per K-step) under the taps of slab s.
// 512 threads = 8 waves (2 per SIMD) as WM (channels) x WN (pixel rows); a wave computes CW = BM / WM channels x RW = TH / WN
// rows of 16 pixels with v_mfma_f32_32x32x16_bf16 (measured 17 % more FLOP per cycle than the 16x16x32 form): the 32
// columns of an MFMA tile are TWO image rows of 16 pixels.
//
// What the measurements of the first versions said (scratch/halo_timing.py, scratch/pmc_halo.sh; 128->128 @ 16x64x64):
//  * a K-step cost ~650 cycles + 5.2 cycles per 16x16x32 MFMA regardless of the barrier schedule; the 650 were ISSUE
//    overhead: 120 VALU + 106 SALU instructions per 32 MFMAs (fragment addresses, tap decomposition, sink selection).
//    -> the nine taps are unrolled: tap offsets are instruction immediates, the swizzle key depends on the pixel COLUMN only
//       (18 pixels per halo row is even, so the bank half of a row is the parity of its column), 16 lane-constant address
//       registers serve every fragment read, LDS-DMA offsets are (per-thread constant) + (scalar soffset);
//  * with one barrier per K-step all 8 waves read fragments, then all issue MFMAs: nothing overlaps.
//    -> PING-PONG: waves 0-3 and 4-7 (one of each per SIMD) run the same loop half a K-step apart, one group reads its 16
//       fragments while the other issues its MFMAs, swapping at a barrier;
//  * LDS-DMA instructions cost 100-180 issue cycles when they open a phase and ~60 between MFMAs -> they are issued
//    inside the MFMA phase.
//
//   MEM(k): read the fragments of K-step k; lgkmcnt(0); vmcnt(youngest group) (own pieces of every older group have landed)
//   MMA(k): MFMAs, interleaved with group(k) = { W(k+3) -> stage (k+3)%4, pass t_k of slab s_k + 1 }
//   A:  MEM(0) | MMA(0) | MEM(1) | MMA(1) | ...          B:  -- | MEM(0) | MMA(0) | MEM(1) | ...      ( | = s_barrier )
// RAW: W(k+1) is group(k-2), covered by the vmcnt that ends MEM(k) of BOTH groups, and a barrier separates those from
//      either group's MEM(k+1).  WAR: group(k) overwrites W(k-1) / slab s_k - 1, last read in MEM(k-1) of both groups, which
//      ended (lgkmcnt(0)) at least one barrier before either group's MMA(k).
// LDS image: rows of 128 bytes (64 channels), 16-byte slots XOR-swizzled by ((row or column) >> 1) & 7: with the row / column
// parity selecting the 128-byte half of a bank line this is conflict-free for the 16-lane groups of ds_read_b128 when a 32-row
// MFMA operand is 32 consecutive weight rows, or 2 x 16 consecutive pixels (any start).
// Epilogue (bias, residual, ReLU, dual destination, accumulate, BatchNormalization statistics, BatchNormalization-backward
// sums): conv_common.h, shared with conv_igemm.hip.
#include "conv_common.h"
#include <cstdlib>
#include <type_traits>

#define STP_OOB 0x80000000u
#define HALO_NWST 4
#if defined(STP_TIMING_REAL)   // scratch builds: the 100 MHz counter every CU shares (phase relations between workgroups) instead of the per-engine shader clock
#define STP_CLOCK() __builtin_amdgcn_s_memrealtime()
#else
#define STP_CLOCK() __builtin_amdgcn_s_memtime()
#endif

// (f32x16 / mfma16_32x32x16: common.h)

// counted wait with a count that is a constant only after loop unrolling
__device__ __forceinline__ void wait_vmcnt_n(int n) {
  switch (n) {
    case 0: wait_vmcnt<0>(); break;
    case 1: wait_vmcnt<1>(); break;
    case 2: wait_vmcnt<2>(); break;
    case 3: wait_vmcnt<3>(); break;
    case 4: wait_vmcnt<4>(); break;
    case 5: wait_vmcnt<5>(); break;
    case 6: wait_vmcnt<6>(); break;
    case 7: wait_vmcnt<7>(); break;
    case 8: wait_vmcnt<8>(); break;
    case 9: wait_vmcnt<9>(); break;
    case 10: wait_vmcnt<10>(); break;
    case 11: wait_vmcnt<11>(); break;
    default: wait_vmcnt<12>(); break;
  }
}

// Row-major epilogue.  In the MFMA C layout a lane owns 4 channels of one pixel: its 8-byte stores / residual loads touch 64
// different 256-byte pixel rows per instruction, and the per-channel BatchNormalization sums need a DPP row reduction per
// fragment (measured on 128->128 @ 16x64x64: 5.4 us of a 24.6 us launch plain, 8.8 us with the statistics, 14 us with the
// BatchNormalization-backward sums).  Instead the workgroup's fp32 tile is written to LDS as [pixel][BM channels] (row stride
// padded by 16 bytes: conflict-free 16-byte writes) and read back so that a thread owns 8 CONSECUTIVE channels of a pixel:
// 16-byte coalesced loads (residual / accumulate / BatchNormalization input) and stores, 16 lanes per 256-byte pixel row, and
// the per-channel sums are plain register adds over the thread's pixels followed by ONE fixed-order reduction per workgroup.
// The tile has 32K outputs per CU, so the epilogue is VALU-bound once the layout is right: EP selects the variant at compile
// time (no per-element branches) and the arithmetic is written on float pairs (v_pk_* instructions).
//   EP 0: bias / residual / accumulate / ReLU                      EP 1: EP 0 + sum, sum of squares of the stored values
//   EP 2: (accumulate) + BatchNormalization-backward: store g = dY under the activation mask, reduce sum g and sum g * xhat
//   EP 3: two destinations (data gradient of conv3x3(concat(UpSampling2D(2)(x), skip))): channel tiles below Cd0 sum every 2 x 2 pixel
//         block of the staged fp32 tile and treat the LOW-RESOLUTION result as EP 2 does (accumulate, mask, sums over [2][Cd0][tiles]) -
//         the gradient of the upsampled tensor is never written, stp_upsample2x_bwd_bn disappears; tiles at or above Cd0: EP 0 into dst1
// Same arithmetic per element as conv_common.h's epilogue (sum g * xhat is accumulated as sum g * x and centred once per channel).
// (f32x2v / unpack_bf16x2: conv_common.h)

// NTH = threads of the workgroup (512; 256 in the persistent 64 -> 64 form)
template <int TH, int BM, int WM, int WN, int EP, int NTH = 512>
__device__ __forceinline__ void epilogue_rm(const ConvArgs& a, f32x16 (&acc)[BM / WM / 32][TH / WN / 2], char* smem, int n, int y0, int x0,
                                            int cout0, int tile_n, int wm, int wn, int lane, int tid, int wave) {
  typedef bf16_t T;
  constexpr int CW = BM / WM, RW = TH / WN, TM = CW / 32, TN = RW / 2;
  constexpr int NPX = TH * 16, RS = BM * 4 + 16;          // staged row: BM floats + 16 bytes of padding
  constexpr int CG = BM / 8, PP = NTH / CG, NP = NPX / PP;   // channel groups of 8, pixels per pass, passes
  static_assert(NPX % PP == 0, "tile pixels per pass");
  const int l15 = lane & 15, l4b = (lane >> 4) & 1, l5 = lane >> 5;
  const int c8 = tid % CG, p0 = tid / CG;
  const int co = cout0 + c8 * 8;
  const bool cok = co < a.Cout;                               // Cout % 8 == 0: a group is valid as a whole
  const bool first = co < a.Cd0;
  // D2S (a.d2s, the space-to-depth data gradient of a stride-2 convolution): the Cout = 4 x Cq channels are parity-class major - channel
  // co = cls * Cq + pc of tile pixel (a, b) is channel pc of the destination pixel (2a + (cls >> 1), 2b + (cls & 1)) of [N][2 Ho][2 Wo][Cq];
  // the per-channel operands (BatchNormalization constants, the sums) belong to pc.  Cq % 8 == 0: a thread's 8 channels share a class.
  const bool d2s = (EP == 0 || EP == 2) && a.d2s;
  const int Cq = d2s ? a.Cout >> 2 : a.Cout;
  const int cls = d2s ? co / Cq : 0, pc = co - cls * Cq;
  T* const dbase = d2s ? reinterpret_cast<T*>(a.dst0) + pc : first ? reinterpret_cast<T*>(a.dst0) + co : reinterpret_cast<T*>(a.dst1) + (co - a.Cd0);
  const int dC = d2s ? Cq : first ? a.Cd0 : a.Cd1;
  const bool accum = first ? a.acc0 : a.acc1;
  const T* res = EP == 2 ? reinterpret_cast<const T*>(a.bnb.x) : reinterpret_cast<const T*>(a.residual);      // (EP 3: no residual)
  const int rC = d2s ? Cq : a.Cout;                           // row stride of `res` (same pixel grid as the destination)

  // EP 3, channel tile of the UPSAMPLED source (cout0 < Cd0, the whole tile: Cd0 % BM == 0): see below
  const bool summed = EP == 3 && cout0 < a.Cd0;
  // global operands of this thread's NP pixels: issued before the staging so their latency hides under it
  int pm[NP];
  u32x4 opr[NP], opa[NP];
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const int px = p0 + k * PP;
    pm[k] = d2s ? (n * a.Ho * 2 + (y0 + (px >> 4)) * 2 + (cls >> 1)) * (a.Wo * 2) + (x0 + (px & 15)) * 2 + (cls & 1)
                : n * a.HoWo + (y0 + (px >> 4)) * a.Wo + x0 + (px & 15);
  }
  if (cok && !summed) {
    if (res) {
#pragma unroll
      for (int k = 0; k < NP; ++k) opr[k] = *reinterpret_cast<const u32x4*>(res + (size_t)pm[k] * rC + pc);
    }
    if (accum) {
#pragma unroll
      for (int k = 0; k < NP; ++k) opa[k] = *reinterpret_cast<const u32x4*>(dbase + (size_t)pm[k] * dC);
    }
  }
  lds_barrier();                                            // every wave has left the K loop: the ring and the slabs are dead
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int px = (wn * RW + 2 * j + l4b) * 16 + l15;
        const int c = wm * CW + i * 32 + 8 * g + 4 * l5;
        *reinterpret_cast<f32x4*>(smem + px * RS + c * 4) = f32x4{acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
      }
  lds_barrier();

  f32x2v ss[4], qq[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) ss[e] = qq[e] = f32x2v{0.f, 0.f};
  // EP 3, channel tile of the UPSAMPLED source (cout0 < Cd0, whole tile: Cd0 % BM == 0): 2 x 2 block sums of the staged tile -> the
  // low-resolution gradient [N][Ho/2][Wo/2][Cd0], accumulated / masked / summed as EP 2 does (a.bnb refers to that tensor)
  if (EP == 3 && summed && cok) {
    constexpr int LNPX = NPX / 4, NLP = (LNPX + PP - 1) / PP;
    const int Hl = a.Ho >> 1, Wl = a.Wo >> 1;
    const T* const xlow = reinterpret_cast<const T*>(a.bnb.x);
    T* const dlow = reinterpret_cast<T*>(a.dst0);
    size_t plow[NLP];
    u32x4 oxl[NLP], oal[NLP];
#pragma unroll
    for (int k = 0; k < NLP; ++k) {
      const int lp = p0 + k * PP;
      plow[k] = ((size_t)n * Hl * Wl + (size_t)((y0 >> 1) + (lp >> 3)) * Wl + (x0 >> 1) + (lp & 7)) * a.Cd0 + co;
      if (lp < LNPX) {
        oxl[k] = *reinterpret_cast<const u32x4*>(xlow + plow[k]);
        if (a.acc0) oal[k] = *reinterpret_cast<const u32x4*>(dlow + plow[k]);
      }
    }
    f32x2v ksc[4], ksh[4];
    {
      const f32x4 r0 = *reinterpret_cast<const f32x4*>(a.bnb.rstd + co), r1 = *reinterpret_cast<const f32x4*>(a.bnb.rstd + co + 4);
      const f32x4 m0 = *reinterpret_cast<const f32x4*>(a.bnb.mean + co), m1 = *reinterpret_cast<const f32x4*>(a.bnb.mean + co + 4);
      f32x4 g0 = {1.f, 1.f, 1.f, 1.f}, g1 = g0, b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
      if (a.bnb.gamma) { g0 = *reinterpret_cast<const f32x4*>(a.bnb.gamma + co); g1 = *reinterpret_cast<const f32x4*>(a.bnb.gamma + co + 4); }
      if (a.bnb.beta) { b0 = *reinterpret_cast<const f32x4*>(a.bnb.beta + co); b1 = *reinterpret_cast<const f32x4*>(a.bnb.beta + co + 4); }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float r = e < 4 ? r0[e & 3] : r1[e & 3], mu = e < 4 ? m0[e & 3] : m1[e & 3];
        const float sc = a.bnb.gamma ? r * (e < 4 ? g0[e & 3] : g1[e & 3]) : r;
        ksc[e >> 1][e & 1] = sc;
        ksh[e >> 1][e & 1] = (e < 4 ? b0[e & 3] : b1[e & 3]) - mu * sc;
      }
    }
    const float alo = a.bnb.relu ? __uint_as_float(1u) : -__builtin_inff();
    const float ahi = a.bnb.relu == 2 ? __uint_as_float(0x40bfffffu) : __builtin_inff();
#pragma unroll
    for (int k = 0; k < NLP; ++k) {
      const int lp = p0 + k * PP;
      if (lp >= LNPX) continue;
      const int hp = ((lp >> 3) * 2) * 16 + (lp & 7) * 2;       // top-left pixel of the 2 x 2 block in the staged tile
      f32x2v v[4] = {f32x2v{0.f, 0.f}, f32x2v{0.f, 0.f}, f32x2v{0.f, 0.f}, f32x2v{0.f, 0.f}};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int px = hp + (q >> 1) * 16 + (q & 1);
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(smem + px * RS + c8 * 32), v1 = *reinterpret_cast<const f32x4*>(smem + px * RS + c8 * 32 + 16);
        v[0] += f32x2v{v0.x, v0.y}; v[1] += f32x2v{v0.z, v0.w}; v[2] += f32x2v{v1.x, v1.y}; v[3] += f32x2v{v1.z, v1.w};
      }
      u32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (a.acc0) v[e] += unpack_bf16x2(oal[k][e]);
        const uint32_t st = pack_bf16x2(v[e].x, v[e].y);          // dY of the low-resolution tensor as it would be stored
        const f32x2v xv = unpack_bf16x2(oxl[k][e]), dy = unpack_bf16x2(st);
        const f32x2v tt = xv * ksc[e] + ksh[e];
        const f32x2v g = f32x2v{__builtin_amdgcn_fmed3f(tt.x, alo, ahi) == tt.x ? dy.x : 0.f, __builtin_amdgcn_fmed3f(tt.y, alo, ahi) == tt.y ? dy.y : 0.f};
        ss[e] += g;
        qq[e] += g * xv;
        o[e] = pack_bf16x2(g.x, g.y);
      }
      *reinterpret_cast<u32x4*>(dlow + plow[k]) = o;
    }
  }
  if (cok && !summed) {
    // per-channel constants of the thread's 8 channels, as pairs
    f32x2v bias2[4], ksc[4], ksh[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) bias2[e] = ksc[e] = ksh[e] = f32x2v{0.f, 0.f};
    if (EP != 2 && a.bias) {
#pragma unroll
      for (int e = 0; e < 4; ++e) bias2[e] = *reinterpret_cast<const f32x2v*>(a.bias + pc + 2 * e);
    }
    if (EP == 2) {
      const f32x4 r0 = *reinterpret_cast<const f32x4*>(a.bnb.rstd + pc), r1 = *reinterpret_cast<const f32x4*>(a.bnb.rstd + pc + 4);
      const f32x4 m0 = *reinterpret_cast<const f32x4*>(a.bnb.mean + pc), m1 = *reinterpret_cast<const f32x4*>(a.bnb.mean + pc + 4);
      f32x4 g0 = {1.f, 1.f, 1.f, 1.f}, g1 = g0, b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
      if (a.bnb.gamma) { g0 = *reinterpret_cast<const f32x4*>(a.bnb.gamma + pc); g1 = *reinterpret_cast<const f32x4*>(a.bnb.gamma + pc + 4); }
      if (a.bnb.beta) { b0 = *reinterpret_cast<const f32x4*>(a.bnb.beta + pc); b1 = *reinterpret_cast<const f32x4*>(a.bnb.beta + pc + 4); }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float r = e < 4 ? r0[e & 3] : r1[e & 3], mu = e < 4 ? m0[e & 3] : m1[e & 3];
        const float sc = a.bnb.gamma ? r * (e < 4 ? g0[e & 3] : g1[e & 3]) : r;
        ksc[e >> 1][e & 1] = sc;
        ksh[e >> 1][e & 1] = (e < 4 ? b0[e & 3] : b1[e & 3]) - mu * sc;
      }
    }
    const bool relu = a.relu != 0;
    const bool hb = EP != 2 && a.bias != nullptr, hr = EP != 2 && res != nullptr, hl = EP != 2 && relu;
    // activation window of the fused BatchNormalization (the gradient passes strictly inside it): t is "on" iff it equals its
    // clamp to [smallest positive number, below the upper bound] - one v_med3 + one compare instead of two compares + mask logic
    const float alo = a.bnb.relu ? __uint_as_float(1u) : -__builtin_inff();
    const float ahi = a.bnb.relu == 2 ? __uint_as_float(0x40bfffffu) : __builtin_inff();      // largest float below 6
    // The optional operands (bias, residual, accumulate, ReLU) are LAUNCH-uniform.  As run-time conditions inside the element loop the
    // compiler turned them into selects - every launch paid the adds, the unpacks and ~100 v_cndmask of all of them (the epilogue is
    // VALU-issue bound: 560 - 1020 instructions per thread for 32 outputs, profiles/r05q_epilogue_instruction_counts.txt).  The loop is
    // therefore instantiated per combination; the common ones (nothing / residual (+ ReLU) / accumulate) get their own copy.
    auto passes = [&](auto hb_, auto hr_, auto ha_, auto hl_) {
      // (EP 2: HL = "the fused activation is a plain ReLU" - the gradient passes iff the re-derived output is > 0: one compare instead of
      //  clamp + compare; identical decisions: the clamp window of ReLU is [smallest positive number, inf])
      constexpr bool HB = decltype(hb_)::value, HR = decltype(hr_)::value, HA = decltype(ha_)::value, HL = decltype(hl_)::value;
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        const int px = p0 + k * PP;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(smem + px * RS + c8 * 32), v1 = *reinterpret_cast<const f32x4*>(smem + px * RS + c8 * 32 + 16);
        f32x2v v[4] = {f32x2v{v0.x, v0.y}, f32x2v{v0.z, v0.w}, f32x2v{v1.x, v1.y}, f32x2v{v1.z, v1.w}};
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (EP != 2) {
            if (HB) v[e] += bias2[e];
            if (HR) v[e] += unpack_bf16x2(opr[k][e]);
          }
          if (HA) v[e] += unpack_bf16x2(opa[k][e]);
          if (EP != 2 && HL) v[e] = f32x2v{fmaxf(v[e].x, 0.f), fmaxf(v[e].y, 0.f)};
          if (EP != 2) o[e] = pack_bf16x2(v[e].x, v[e].y);
          if (EP == 1) {
            const f32x2v sv = unpack_bf16x2(o[e]);
            ss[e] += sv;
            qq[e] += sv * sv;
          }
          if (EP == 2) {
            // masked gradient (the activation mask re-derived from the BatchNormalization input with the forward's fma), rounded ONCE:
            // round(on ? dY : 0) = on ? round(dY) : 0; the sums take the values as stored
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
    // the rest: every operand behind its own (launch-uniform) test, as the one loop of the earlier rounds had it
    auto passes_all = [&]() {
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        const int px = p0 + k * PP;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(smem + px * RS + c8 * 32), v1 = *reinterpret_cast<const f32x4*>(smem + px * RS + c8 * 32 + 16);
        f32x2v v[4] = {f32x2v{v0.x, v0.y}, f32x2v{v0.z, v0.w}, f32x2v{v1.x, v1.y}, f32x2v{v1.z, v1.w}};
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] += bias2[e];
          if (hr) v[e] += unpack_bf16x2(opr[k][e]);
          if (accum) v[e] += unpack_bf16x2(opa[k][e]);
          if (hl) v[e] = f32x2v{fmaxf(v[e].x, 0.f), fmaxf(v[e].y, 0.f)};
          o[e] = pack_bf16x2(v[e].x, v[e].y);
          if (EP == 1) {
            const f32x2v sv = unpack_bf16x2(o[e]);
            ss[e] += sv;
            qq[e] += sv * sv;
          }
        }
        *reinterpret_cast<u32x4*>(dbase + (size_t)pm[k] * dC) = o;
      }
    };
    typedef std::true_type Y_;
    typedef std::false_type N_;
    if (EP == 2) {                                  // (res = the BatchNormalization input: always read)
      const bool r1 = a.bnb.relu == 1 && !a.ep_generic;
      if (accum) { if (r1) passes(N_(), N_(), Y_(), Y_()); else passes(N_(), N_(), Y_(), N_()); }
      else { if (r1) passes(N_(), N_(), N_(), Y_()); else passes(N_(), N_(), N_(), N_()); }
    } else if (a.ep_generic) passes_all();
    else if (!hb && !hr && !accum && !hl) passes(N_(), N_(), N_(), N_());
    else if (!hb && hr && !accum) { if (hl) passes(N_(), Y_(), N_(), Y_()); else passes(N_(), Y_(), N_(), N_()); }
    else if (!hb && !hr && accum && !hl) passes(N_(), N_(), Y_(), N_());
    else if (hb && !hr && !accum) { if (hl) passes(Y_(), N_(), N_(), Y_()); else passes(Y_(), N_(), N_(), N_()); }
    else passes_all();
  }
#if defined(STP_EXP) && STP_EXP == 31   // what-if: no cross-thread reduction / partial-sum stores (values kept live)
  if (EP >= 1 && a.P < 0) {
#else
  if (EP == 3 && !summed) return;                                // (workgroup-uniform: the skip tensor's tiles have no sums)
  if (EP >= 1) {
#endif
    if ((EP == 2 || EP == 3) && cok) {   // sum g * xhat = rstd * (sum g * x - mean * sum g), per thread (linear, so the partition does not matter)
      const f32x4 m0 = *reinterpret_cast<const f32x4*>(a.bnb.mean + pc), m1 = *reinterpret_cast<const f32x4*>(a.bnb.mean + pc + 4);
      const f32x4 r0 = *reinterpret_cast<const f32x4*>(a.bnb.rstd + pc), r1 = *reinterpret_cast<const f32x4*>(a.bnb.rstd + pc + 4);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float mu = e < 4 ? m0[e & 3] : m1[e & 3], rs = e < 4 ? r0[e & 3] : r1[e & 3];
        qq[e >> 1][e & 1] = rs * (qq[e >> 1][e & 1] - mu * ss[e >> 1][e & 1]);
      }
    }
    // Fixed-order reduction over the 512 / CG threads that share a channel group, through LDS only (the ds_bpermute butterfly +
    // per-wave table this replaces cost 4500-5600 cycles per workgroup, scratch/halo_timing.py): every thread writes its 16
    // partials [k][thread] (k = channel pair, sum / sum of squares), 512 threads sum 512 / NPART / CG of them each, NOUT threads
    // combine the NPART parts and store the tile's column of the [stat][channel][tile] partial sums.
    constexpr int J = NTH / CG, NOUT = CG * 16, NPART = NTH / NOUT, JP = J / NPART, KS = NTH + CG;   // KS: conflict-free k stride
    static_assert(NPART >= 1 && J % NPART == 0, "reduction shape");
    lds_barrier();                                          // the staged tile is dead
    float* r1 = reinterpret_cast<float*>(smem);               // [16][KS]
    float* r2 = r1 + 16 * KS;                                 // [NPART][NOUT]
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      r1[(e * 4 + 0) * KS + tid] = ss[e].x; r1[(e * 4 + 1) * KS + tid] = ss[e].y;
      r1[(e * 4 + 2) * KS + tid] = qq[e].x; r1[(e * 4 + 3) * KS + tid] = qq[e].y;
    }
    lds_barrier();
    {
      const int o = tid % NOUT, q = tid / NOUT;
      const float* src = r1 + (o / CG) * KS + (o % CG) + CG * (q * JP);
      float acc_ = 0.f;
#pragma unroll
      for (int j = 0; j < JP; ++j) acc_ += src[CG * j];
      r2[q * NOUT + o] = acc_;
    }
    lds_barrier();
    if (tid < NOUT) {
      float tot = 0.f;
#pragma unroll
      for (int q = 0; q < NPART; ++q) tot += r2[q * NOUT + tid];
      const int k = tid / CG, ch = (tid % CG) * 8 + (k >> 2) * 2 + (k & 1), stat = (k >> 1) & 1;
      const int Cs = EP == 3 ? a.Cd0 : a.Cout;                  // EP 3: the table covers the summed destination's channels
      if (d2s) {
        // the four parity classes of a destination channel are four column blocks of ITS row: [stat][Cq][4 x tiles]
        const int v = cout0 + ch, vc = v / Cq, vp = v - vc * Cq;
        if (v < a.Cout) a.stats[(((size_t)stat * Cq + vp) * 4 + vc) * a.ntile_n + tile_n] = tot;
      } else if (cout0 + ch < Cs) stats_store(a, ((size_t)stat * Cs + cout0 + ch) * a.ntile_n + tile_n, tot);
    }
    if (d2s) return;                                            // (launch-uniform; no group-level pre-reduction in this form)
    // group-level pre-reduction (conv_common.h): the last workgroup of every G tiles brings the table under 128 columns
    stats_group_finish<NTH>(a, EP == 3 ? a.Cd0 : a.Cout, cout0, BM, cout0 / BM, tile_n, a.ntile_n, tid, reinterpret_cast<unsigned*>(smem));
  }
}

// PBN: the fused producer BatchNormalization (below) is a compile-time parameter of the body - as a run-time branch its 14-34 registers
// tax every launch that does not use it (halo 8x64: 117 -> 102 VGPRs = one more wave per SIMD); the two __global__ entry points keep
// the plain kernel's name (profiles, bench.py keys) and give the fused variant its own.
// SRC2 (round 5): the input is concat(src0, src1) along the channels, src0 optionally behind UpSampling2D(2) (nearest) - the forward of the
// U-Net decoder's conv3x3(concat(UpSampling2D(2)(x), skip)).  A slab (64 channels) lies in ONE source: slabs below C0 / 64 are staged from
// src0 - with the halo pixel (y, x) fetched from the low-resolution pixel (y >> 1, x >> 1): the 2 x 2 replicas come out of L2, the slab in
// LDS is the upsampled one and the K loop does not know - the others from src1; the weight column of slab s is s * 64 of the
// concatenated channel order either way.  A compile-time parameter: a second set of per-pass offsets would tax the one-source kernels.
// NT (round 5) = taps per slab: 9 = the 3 x 3 window; 4 = the 2 x 2 window at offsets (+0 / +1, +0 / +1) - the SPACE-TO-DEPTH form of
// the data gradient of a 3x3 / stride-2 / pad-1 convolution (stp_conv_params.s2d_dgrad): per output parity class (py, px) the gradient
// at (2a + py, 2b + px) reads dY at (a, b), (a, b + 1), (a + 1, b), (a + 1, b + 1) under 1 / 2 / 2 / 4 of the nine kernel taps, so all
// four classes are ONE dense 2 x 2-tap convolution of dY into 4 x Cin channels (class-major; weights zero where a class has no tap:
// 9 of the 16 (tap, class) blocks are live) whose store is a depth-to-space (epilogue_rm, D2S).  A tap is an LDS offset of the slab, as
// before; only the prefetch schedule depends on NT: the passes of the next slab must have been issued three K-steps before its first
// use, i.e. in the taps 0 .. NT - 3 of the current one (PPS passes per K-step).
// FOLD1 (round 5, NT == 9 with a second source): the second source is dY of a SIBLING 1x1 / stride-1 convolution that reads the same tensor
// (the projection shortcut of a ResNet basic block's first unit): its data gradient is the centre tap of this launch over
// a.fold_weight = [Cout^16][C1] - the slabs of the second source run one live K-step (tap 4) and eight that keep only their barriers.
// RING (round 5) = stages of the weight ring; W(k + RING - 1) is requested during MMA(k).  What the space-to-depth launches showed
// (profiles/r05h_s2d_layer_bench.txt): a K-step that issues NO MFMAs still costs 0.45 us - an LDS-DMA needs ~1 us from issue to landing
// under load, and with 4 stages a piece has 1.5 - 2 K-steps.  RING = 6 gives it 3.5 - 4: the wait that ends MEM(k) then leaves the
// RING - 3 youngest groups in flight (their sizes are kept in scalar registers), the passes of the next slab must be issued in the
// taps 0 .. NT - RING + 1.  LDS allows it for the 8-row tiles (111 -> 143 KB) and the one-slab 64-channel tiles.
template <int TH, int BM, int WM, int WN, int EP, bool PBN, bool SRC2, int NT = 9, bool FOLD1 = false, int RING = HALO_NWST>
__device__ __forceinline__ void conv_halo_body(const ConvArgs& a_in) {
  static_assert(RING == 4 || (RING == 6 && NT == 9), "weight ring depth");
  static_assert(!FOLD1 || (SRC2 && NT == 9 && !PBN), "the folded 1x1 sibling is a second source of the 3 x 3 window");
  static_assert(WM * WN == 8 && BM % (WM * 32) == 0 && TH % (WN * 2) == 0, "config");
  static_assert(NT == 9 || NT == 4, "taps per slab");
#if defined(__HIP_DEVICE_COMPILE__)
#if defined(STP_TIMING)   // scratch build: `bias` carries a u64[4 * workgroups] buffer of shader-clock stamps (scratch/halo_timing.py)
  ConvArgs a = a_in;
  unsigned long long* const tdbg = reinterpret_cast<unsigned long long*>(const_cast<float*>(a_in.bias));
  a.bias = nullptr;
  unsigned long long tstamp[4];
  tstamp[0] = STP_CLOCK();
#define STP_STAMP(i) tstamp[i] = STP_CLOCK()
#else
  const ConvArgs& a = a_in;
#define STP_STAMP(i)
#endif
  typedef bf16_t T;
  constexpr int TW = 16, HWD = TW + 2, HH = TH + 2, NHP = HH * HWD;
  constexpr int NPASS = (NHP + 63) / 64;         // 64 halo pixels (8 per wave) per pass of the 512 threads
  static_assert(NPASS <= 6 || TH == 32, "the next slab must have landed (own pieces) when the K-step of tap 8 begins (TH = 32: one-slab inputs only)");
  constexpr int SROWS = (NHP + 7) / 8 * 8;       // slab rows kept in LDS (8 per wave instruction; rows >= NHP are never read)
  constexpr int SLAB = SROWS * 128;
  constexpr int NWST = RING;                     // weight ring stages
  constexpr int PD = RING - 1;                   // prefetch distance of the weights (K-steps)
  constexpr int WSTAGE = BM * 128;
  constexpr int LW = BM / 64;                    // weight LDS-DMA instructions per thread per K-step
  constexpr int CW = BM / WM, RW = TH / WN;      // channels / pixel rows per wave
  constexpr int TM = CW / 32, TN = RW / 2;       // 32x32 MFMA tiles per wave
  constexpr int PPS = (NPASS + (NT - PD + 1) - 1) / (NT - PD + 1);   // slab passes per K-step: all issued in taps 0 .. NT - PD
  constexpr int NPT = (NPASS + PPS - 1) / PPS;              // K-steps that carry passes
  static_assert(LW + PPS <= TM * TN * 4, "one LDS-DMA instruction per MFMA at most");
  static_assert(NPT <= NT - PD + 1 || TH == 32, "the next slab is complete PD K-steps before its first use");
  static_assert((PD - 2) * (LW + PPS) <= 12, "wait_vmcnt_n covers 0 .. 12");
  // LDS: [slab 0][slab 1 unless Cin == 64][NWST weight stages][scale/shift table of a fused producer BatchNormalization]

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int l15 = lane & 15, l31 = lane & 31, l4b = (lane >> 4) & 1, l5 = lane >> 5;

  const int bid = xcd_remap(blockIdx.x, a.ntile_m * a.ntile_n);
  const int tile_n = bid / a.ntile_m, tile_m = bid - tile_n * a.ntile_m;   // channel tiles innermost: they share the slabs in L2
  const int tx_n = a.Wo / TW, ty_n = a.Ho / TH;
  const int n = tile_n / (tx_n * ty_n), trem = tile_n - n * (tx_n * ty_n);
  const int ty = trem / tx_n, tx = trem - ty * tx_n;
  const int y0 = ty * TH, x0 = tx * TW, cout0 = tile_m * BM;

  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)a.weight, 0, a.bytesw, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsw2 = __builtin_amdgcn_make_buffer_rsrc((void*)((NT == 4 || FOLD1) && a.fold_weight ? a.fold_weight : a.weight), 0,
                                                                         (NT == 4 || FOLD1) && a.fold_weight ? a.bytesw_fold : a.bytesw, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)a.src0, 0, a.bytes0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)(SRC2 && a.src1 ? a.src1 : a.src0), 0, SRC2 && a.src1 ? a.bytes1 : a.bytes0, 0x00020000);

  // ---- per-thread LDS-DMA source offsets ------------------------------------------------------------------------------
  // slab pass i: halo pixel hp = i*64 + tid/8 (column hx) at physical slot tid&7, which holds logical slot (tid&7) ^ ((hx>>1)&7)
  const int prow = tid >> 3, pslot = tid & 7;
  uint32_t soff[NPASS], soff1[SRC2 ? NPASS : 1];
  const int up = (SRC2 && a.mode == STP_SRC_NEAREST2X) ? 1 : 0;      // src0 behind UpSampling2D(2): [N][Hv / 2][Wv / 2][C0]
#pragma unroll
  for (int i = 0; i < NPASS; ++i) {
    const int hp = i * 64 + prow;
    const int hy = hp / HWD, hx = hp - hy * HWD;
    const int y = y0 - 1 + hy, x = x0 - 1 + hx;
    const bool ok = hp < NHP && (unsigned)y < (unsigned)a.Hv && (unsigned)x < (unsigned)a.Wv;
    const uint32_t ch = (uint32_t)((pslot ^ ((hx >> 1) & 7)) * 8);
    if (SRC2) {
      soff[i] = ok ? (((uint32_t)(n * a.Hs0 + (y >> up)) * (uint32_t)a.Ws0 + (uint32_t)(x >> up)) * (uint32_t)a.C0 + ch) * 2u : STP_OOB;
      soff1[i] = ok ? (((uint32_t)(n * a.Hv + y) * (uint32_t)a.Wv + (uint32_t)x) * (uint32_t)a.C1 + ch) * 2u : STP_OOB;
    } else {
      soff[i] = ok ? (((uint32_t)(n * a.Hv + y) * (uint32_t)a.Wv + (uint32_t)x) * (uint32_t)a.C0 + ch) * 2u : STP_OOB;
    }
  }
  // weight instruction i: row i*64 + tid/8 of the channel tile, logical slot (tid&7) ^ ((row>>1)&7)
  uint32_t woff[LW];
#pragma unroll
  for (int i = 0; i < LW; ++i)
    woff[i] = ((uint32_t)(cout0 + i * 64 + prow) * (uint32_t)a.K + (uint32_t)((pslot ^ ((prow >> 1) & 7)) * 8)) * 2u;
  // NT == 4 (space-to-depth data gradient): NO weight copy of its own - row r = cls * Cq + ci of the dense 2 x 2-tap matrix is row ci of the
  // 3x3 layer's ordinary data-gradient copy a.weight = [Cq^16][3][3][C0] (bwd[ci][kh'][kw'][co] = W[co][2 - kh'][2 - kw'][ci]) at the kernel
  // tap that class (py, px) meets under tap (da, db): kh = (py, da): (0,0) 1, (0,1) none, (1,0) 2, (1,1) 0; kw alike - or nothing: an
  // out-of-range offset, the LDS-DMA then writes zeros and moves no bytes.  The slabs of the second source (the sibling 1x1 / stride-2
  // shortcut's dY) take a.fold_weight = [Cq^16][C1], live for class 0 / tap 0 only.
  uint32_t wtap[NT == 4 ? LW : 1][4], wsc[(NT == 4 || FOLD1) ? LW : 1];
  unsigned live_main = 0xfu, live_sc = 0x1u;      // taps with live weights for THIS WAVE's channels (dead K-steps skip their MFMAs)
  if constexpr (FOLD1) {
    live_main = 0x1ffu; live_sc = 0x10u;          // nine taps of the 3 x 3 layer; the centre tap of the folded 1x1 sibling
#pragma unroll
    for (int i = 0; i < LW; ++i) {
      const int r = cout0 + i * 64 + prow;
      wsc[i] = r < a.Cout ? ((uint32_t)r * (uint32_t)a.C1 + (uint32_t)((pslot ^ ((prow >> 1) & 7)) * 8)) * 2u : STP_OOB;
    }
  }
  if constexpr (NT == 4) {
    const int Cq = a.Cout >> 2;
#pragma unroll
    for (int i = 0; i < LW; ++i) {
      const int r = cout0 + i * 64 + prow, cls = r / Cq, ci = r - cls * Cq;
      const uint32_t slot = (uint32_t)((pslot ^ ((prow >> 1) & 7)) * 8);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int da = t >> 1, db = t & 1;
        const int kh = (cls >> 1) == 0 ? (da == 0 ? 1 : -1) : (da == 0 ? 2 : 0), kw = (cls & 1) == 0 ? (db == 0 ? 1 : -1) : (db == 0 ? 2 : 0);
        wtap[i][t] = (r < a.Cout && kh >= 0 && kw >= 0) ? ((uint32_t)((ci * 3 + (2 - kh)) * 3 + (2 - kw)) * (uint32_t)a.C0 + slot) * 2u : STP_OOB;
      }
      wsc[i] = (r < a.Cout && cls == 0 && a.C1) ? ((uint32_t)ci * (uint32_t)a.C1 + slot) * 2u : STP_OOB;
    }
    if (CW <= Cq && (Cq % CW) == 0) {             // the wave's channels lie in one class
      const int wc = (cout0 + wm * CW) / Cq;
      live_main = wc == 0 ? 0x1u : wc == 1 ? 0x3u : wc == 2 ? 0x5u : 0xfu;
      live_sc = wc == 0 ? 0x1u : 0x0u;
    }
  }

  const int nslab = (SRC2 ? a.Ctot : a.C0) >> 6, nslab0 = a.C0 >> 6;
  const int off_w = (nslab > 1 ? 2 : 1) * SLAB;   // one slab buffer is enough for a 64-channel input: two workgroups share a CU
  const uint32_t tapb = (uint32_t)((SRC2 && !FOLD1) ? a.Ctot : a.C0) * 2u;     // bytes between the weight columns of consecutive taps

  // piece i of the weights of K-step (slab s, tap t) -> ring stage st; the column offset travels in the scalar soffset
  auto issue_weight_piece = [&](int i, int st, int s, int t) {
    if constexpr (NT == 4) {
      if (s >= nslab0)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw2, (__attribute__((address_space(3))) void*)(smem + off_w + st * WSTAGE + (i * 64 + wave * 8) * 128), 16,
                                                 t == 0 ? wsc[i] : STP_OOB, (uint32_t)(s - nslab0) * 128u, 0, 0);
      else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (__attribute__((address_space(3))) void*)(smem + off_w + st * WSTAGE + (i * 64 + wave * 8) * 128), 16,
                                                 wtap[i][t], (uint32_t)s * 128u, 0, 0);
    } else {
      if (FOLD1 && s >= nslab0)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw2, (__attribute__((address_space(3))) void*)(smem + off_w + st * WSTAGE + (i * 64 + wave * 8) * 128), 16,
                                                 t == 4 ? wsc[i] : STP_OOB, (uint32_t)(s - nslab0) * 128u, 0, 0);
      else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (__attribute__((address_space(3))) void*)(smem + off_w + st * WSTAGE + (i * 64 + wave * 8) * 128), 16,
                                                 woff[i], (uint32_t)t * tapb + (uint32_t)s * 128u, 0, 0);
    }
  };
  // pass p (compile-time) of slab s.  A wave whose 8 rows lie past the slab issues nothing: pass_on() enters the vmcnt counts
  auto pass_on = [&](int p) { return (p * 64 + wave * 8) < SROWS; };   // wave-uniform
  auto issue_slab_pass = [&](int s, int p) {
    if (!pass_on(p)) return;
    if (SRC2 && s >= nslab0)       // (s is a loop counter: wave-uniform) a slab of the second source
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (__attribute__((address_space(3))) void*)(smem + (s & 1) * SLAB + (p * 64 + wave * 8) * 128), 16,
                                               soff1[SRC2 ? p : 0], (uint32_t)(s - nslab0) * 128u, 0, 0);
    else
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, (__attribute__((address_space(3))) void*)(smem + (s & 1) * SLAB + (p * 64 + wave * 8) * 128), 16, soff[p],
                                               (uint32_t)s * 128u, 0, 0);
  };

  // ---- fused PRODUCER BatchNormalization (+activation): src0 holds the convolution output x that the BatchNormalization reads;
  // y = act(fma(x, scale, shift)) is computed IN LDS on the slab, once per slab (9 K-steps), by the thread that DMA'd the 16 bytes
  // (own data: only its own vmcnt orders the read-modify-write, the K-step barrier publishes it).  Same fma / activation / bf16
  // rounding as stp_bn_apply -> bit-identical operands; pixels outside the image stay 0 (the padding applies to y, not x).
  static_assert(!(PBN && SRC2), "the fused producer BatchNormalization is a one-source feature");
  const bool fuse_bn = PBN && a.pbn.mean != nullptr;
  float* const tab = reinterpret_cast<float*>(smem + off_w + NWST * WSTAGE);      // scale[C0], shift[C0]
  auto transform_slab = [&](int s_) {
    char* sb = smem + (s_ & 1) * SLAB;
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
      if ((i * 64 + wave * 8) >= SROWS) continue;                    // wave-uniform: rows past the slab were never fetched
      const int hp = i * 64 + prow;
      if (soff[i] == STP_OOB) continue;
      const int hx = hp % HWD;
      u32x4* vp = reinterpret_cast<u32x4*>(sb + hp * 128 + pslot * 16);
      const int ch = s_ * 64 + ((pslot ^ ((hx >> 1) & 7)) << 3);
      const f32x4 sc0 = *reinterpret_cast<const f32x4*>(tab + ch), sc1 = *reinterpret_cast<const f32x4*>(tab + ch + 4);
      const f32x4 sh0 = *reinterpret_cast<const f32x4*>(tab + a.C0 + ch), sh1 = *reinterpret_cast<const f32x4*>(tab + a.C0 + ch + 4);
      const u32x4 v = *vp;
      u32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float lo = bn_act(bn_affine(h16lo_to_f32(v[e]), e < 2 ? sc0[2 * e] : sc1[2 * e - 4], e < 2 ? sh0[2 * e] : sh1[2 * e - 4]), a.pbn.relu);
        const float hi = bn_act(bn_affine(h16hi_to_f32(v[e]), e < 2 ? sc0[2 * e + 1] : sc1[2 * e - 3], e < 2 ? sh0[2 * e + 1] : sh1[2 * e - 3]), a.pbn.relu);
        o[e] = pack_bf16x2(lo, hi);
      }
      *vp = o;
    }
  };

  // ---- lane-constant fragment addresses ---------------------------------------------------------------------------------
  // A (weights): row wm*CW + I*32 + l31 of the stage, 16-byte slot kc*2 + l5 of the K-step's 8 -> + stage offset + I*4096
  // B (pixels) : halo pixel (wn*RW + 2J + l4b + dy) * 18 + l15 + dx of the slab           -> + (2J + dy) * 2304 (immediate)
  uint32_t a_lane[4], b_lane[3][4];
#pragma unroll
  for (int kc = 0; kc < 4; ++kc) {
    a_lane[kc] = (uint32_t)(off_w + (wm * CW + l31) * 128 + (((kc * 2 + l5) ^ ((l31 >> 1) & 7)) << 4));
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int hx = l15 + dx;
      b_lane[dx][kc] = (uint32_t)(((wn * RW + l4b) * HWD + hx) * 128 + (((kc * 2 + l5) ^ ((hx >> 1) & 7)) << 4));
    }
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // ---- prologue: slab 0, then W(0), W(1), W(2) --------------------------------------------------------------------------
#pragma unroll
  for (int p = 0; p < NPASS; ++p) issue_slab_pass(0, p);
#pragma unroll
  for (int t = 0; t < PD; ++t)             // W(0) .. W(PD - 1)  (NT >= PD: all of slab 0)
#pragma unroll
    for (int i = 0; i < LW; ++i) issue_weight_piece(i, t, 0, t);
  static_assert(NT >= PD || RING == 4, "the prologue's weight tiles lie in slab 0");

  if (fuse_bn) {
    for (int c = tid; c < a.C0; c += 512) {
      const float r = a.pbn.rstd[c], sc = a.pbn.gamma ? r * a.pbn.gamma[c] : r;
      tab[c] = sc;
      tab[a.C0 + c] = (a.pbn.beta ? a.pbn.beta[c] : 0.f) - a.pbn.mean[c] * sc;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();          // the table is visible (every wave executes this: fuse_bn is uniform)
  }
  STP_STAMP(1);

  {
    const bool grp_b = wave >= 4;          // wave-uniform
    wait_vmcnt<(PD - 1) * LW>();           // slab 0 and W(0): everything but W(1) .. W(PD - 1)
    if (fuse_bn) {
      transform_slab(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (grp_b) __builtin_amdgcn_s_barrier();

    u32x4 fa[4][TM], fb[4][TN];
    int wst = 0;                           // ring stage of the current K-step = k % RING
    int gq0 = LW, gq1 = LW, gq2 = LW;      // RING = 6: LDS-DMA instructions of this wave in the three youngest groups (the prologue's W(2 ..) count)
    for (int s = 0; s < nslab; ++s) {
      const bool last = s + 1 == nslab;    // no slab to prefetch, and the weights of the next "slab" do not exist
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int dy = NT == 9 ? t / 3 : 1 + t / 2, dx = NT == 9 ? t - (t / 3) * 3 : 1 + (t & 1);             // compile-time after unrolling
        // ---------------- MEM(k)
        if (fuse_bn && t == NT - 1 && !last) transform_slab(s + 1);   // its passes landed (own pieces) before MEM(9s+7) ended
        // NT == 4: a K-step whose weights are all zero for this wave's channels keeps its barriers and its LDS-DMA issue, nothing else
        const bool live = (NT != 4 && !FOLD1) || (((s < nslab0 ? live_main : live_sc) >> t) & 1u);      // (wave-uniform)
        if (live) {
          const uint32_t so = (uint32_t)(wst * WSTAGE);
#pragma unroll
          for (int kc = 0; kc < 4; ++kc) {
            const uint32_t ab = a_lane[kc] + so;
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[kc][j] = *reinterpret_cast<const u32x4*>(smem + b_lane[dx][kc] + (2 * j + dy) * (HWD * 128));
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[kc][i] = *reinterpret_cast<const u32x4*>(smem + ab + i * 4096);
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // leave the youngest group (issued in MMA(k-1)) in flight: LW weight pieces unless that K-step had none left to
        // prefetch, plus its slab pass if there is a next slab and this wave has rows in that pass
        if constexpr (RING != 4) {
          wait_vmcnt_n(gq0 + gq1 + gq2);                    // the three youngest groups stay in flight
        } else if (t == 0) {
          wait_vmcnt_n(LW);                                 // previous K-step = the last tap of a slab that is not the last: weights, no pass
        } else {
          int np = 0;                                       // passes issued in MMA(k - 1) by this wave
          if ((t - 1) < NPT && !last) {
#pragma unroll
            for (int i = 0; i < PPS; ++i)
              if ((t - 1) * PPS + i < NPASS && pass_on((t - 1) * PPS + i)) ++np;
          }
          wait_vmcnt_n(((last && t - 1 + 3 >= NT) ? 0 : LW) + np);
        }
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---------------- MMA(k) + group(k)
        __builtin_amdgcn_s_setprio(1);
        {
          const int st3 = RING == 4 ? ((wst + 3) & 3) : (wst == 0 ? RING - 1 : wst - 1);
          const int t3 = (t + PD) % NT, s3 = s + (t + PD) / NT;
          const bool wlive = RING == 4 ? !(last && t + 3 >= NT) : (s3 < nslab);
          if constexpr (RING != 4) {          // sizes of the youngest groups: this K-step's joins, the oldest leaves
            int np = 0;
            if (t < NPT && !last) {
#pragma unroll
              for (int i = 0; i < PPS; ++i)
                if (t * PPS + i < NPASS && pass_on(t * PPS + i)) ++np;
            }
            gq2 = gq1; gq1 = gq0; gq0 = (wlive ? LW : 0) + np;
          }
          int piece = 0;
          if (!live) {
#pragma unroll
            for (int pz = 0; pz < LW + PPS; ++pz) {
              if (pz < LW) {
                if (wlive) issue_weight_piece(pz, st3, s3, t3);
              } else if (t < NPT && t * PPS + (pz - LW) < NPASS) {
                if (!last) issue_slab_pass(s + 1, t * PPS + (pz - LW));
              }
            }
          } else
#pragma unroll
          for (int kc = 0; kc < 4; ++kc)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
              for (int j = 0; j < TN; ++j) {
                acc[i][j] = mfma16_32x32x16(fa[kc][i], fb[kc][j], acc[i][j]);
                __builtin_amdgcn_sched_barrier(0);
                if (piece < LW) {
                  if (wlive) issue_weight_piece(piece, st3, s3, t3);
                } else if (piece < LW + PPS && t < NPT && t * PPS + (piece - LW) < NPASS) {
                  if (!last) issue_slab_pass(s + 1, t * PPS + (piece - LW));
                }
                __builtin_amdgcn_sched_barrier(0);
                ++piece;
              }
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        wst = RING == 4 ? ((wst + 1) & 3) : (wst + 1 == RING ? 0 : wst + 1);
      }
      // the fragment addresses follow the slab: slot (s+1) & 1
      const uint32_t d = (s & 1) ? (uint32_t)(-SLAB) : (uint32_t)SLAB;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) b_lane[dx][kc] += d;
    }
    if (!grp_b) __builtin_amdgcn_s_barrier();
  }
  STP_STAMP(2);

  // ---- epilogue: the fp32 tile goes through LDS into [pixel][channel] order (see epilogue_rm) ----------------------------
  epilogue_rm<TH, BM, WM, WN, EP>(a, acc, smem, n, y0, x0, cout0, tile_n, wm, wn, lane, tid, wave);
#if defined(STP_TIMING)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  STP_STAMP(3);
  if (tid == 0)
    for (int i = 0; i < 4; ++i) tdbg[(size_t)blockIdx.x * 4 + i] = tstamp[i];
#endif
#endif
}

// 64-channel tiles of 16 x 16 pixels: one slab + the ring = 76 KB, so TWO workgroups fit a CU - if the kernel stays within 128 registers
// (4 waves per SIMD).  Left to itself the allocator took 139 and every CU ran one workgroup at a time (stage 1: 1024 tiles in four
// rounds instead of two).
#define HALO_MIN_WAVES(TH, BM) ((TH) == 16 && (BM) == 64 ? 4 : 1)
template <int TH, int BM, int WM, int WN, int EP>
__global__ __launch_bounds__(512, HALO_MIN_WAVES(TH, BM)) void conv_halo_kernel(const ConvArgs a) { conv_halo_body<TH, BM, WM, WN, EP, false, false>(a); }
template <int TH, int BM, int WM, int WN, int EP>
__global__ __launch_bounds__(512, HALO_MIN_WAVES(TH, BM)) void conv_halo_pbn_kernel(const ConvArgs a) { conv_halo_body<TH, BM, WM, WN, EP, true, false>(a); }
// six-stage weight ring (8-row tiles: the LDS has room), one source
template <int TH, int BM, int WM, int WN, int EP>
__global__ __launch_bounds__(512, 1) void conv_halo_r6_kernel(const ConvArgs a) { conv_halo_body<TH, BM, WM, WN, EP, false, false, 9, false, 6>(a); }
// two sources / upsampled first source (forward only: EP 0 / 1)
template <int TH, int BM, int WM, int WN, int EP>
__global__ __launch_bounds__(512, HALO_MIN_WAVES(TH, BM)) void conv_halo2_kernel(const ConvArgs a) { conv_halo_body<TH, BM, WM, WN, EP, false, true>(a); }
// data gradient of a 3x3 / stride-1 layer with its sibling 1x1 / stride-1 shortcut folded in as a second source (EP 0 / 2)
template <int TH, int BM, int WM, int WN, int EP>
__global__ __launch_bounds__(512, HALO_MIN_WAVES(TH, BM)) void conv_halo_fold1_kernel(const ConvArgs a) { conv_halo_body<TH, BM, WM, WN, EP, false, true, 9, true>(a); }
// space-to-depth data gradient of a stride-2 convolution: 2 x 2 taps, one or two sources, depth-to-space store (EP 0 / 2)
template <int TH, int BM, int WM, int WN, int EP>
__global__ __launch_bounds__(512, HALO_MIN_WAVES(TH, BM)) void conv_halo_s2d_kernel(const ConvArgs a) { conv_halo_body<TH, BM, WM, WN, EP, false, true, 4>(a); }

// ================================================================================================ persistent 64 -> 64 form (P64)
// The 64 -> 64-channel layers at 128 x 128 (ResNet34 stage 1 forward + data gradient: 13 launches of the U-Net step) are the worst-placed
// launches of the family: 39 us each against an HBM floor of 13 (67 MB) and an MFMA floor of 8.  The shared-clock timeline of the 16 x 16 x 64
// tiles (scratch/r05/halo_phase.py, profiles/r05p_halo64_phase_timeline.txt) shows why: the two co-resident workgroups of a CU start together
// and stay in LOCKSTEP - 512 workgroups fetch their halos (2.3 us, HBM busy, MFMA idle), 512 run their nine K-steps (7.3 us, HBM idle), 512
// store (4.7 us, MFMA idle), and the second round repeats it.  Nothing of one phase ever runs under another; and with K = 576 the epilogue of
// a tile (staging, rounding, sums: VALU + LDS) costs about what its 72 MFMAs per wave do.
// Here TWO 4-wave workgroups per CU (one wave per SIMD each) walk 8 x 16-pixel tiles (tile = round * workgroups + workgroup), half a tile apart:
//  * the layer's whole weight matrix is 64 x 576 x 2 = 72 KB: a wave's 32 channels x 576 are 144 registers of MFMA A fragments, fetched ONCE per
//    workgroup (LDS-DMA into the not yet used slab / staging space, then ds_read) - no weight ring, no LDS-DMA of weights and no barrier inside
//    the K loop, and a K-step reads only its B fragments from LDS;
//  * two slab buffers: the halo of tile i + 2 is requested (LDS-DMA) when the epilogue of tile i ends and has the K loop of tile i + 1 to land;
//  * the epilogue stages through its OWN region of the LDS (2 x 23 + 34 = 80 KB per workgroup, 160 KB per CU), its stores drain under the next
//    K loop - and while one workgroup of the CU is in its epilogue (VALU, LDS, HBM) the other has the MFMA pipes to itself: the workgroups of
//    the second slot start STP_P64_STAGGER x ~0.4 us late and the two stay out of phase from there.
// Per tile a wave runs: K loop (72 MFMAs, reads PF steps ahead) | vmcnt(0): the next slab has landed | epilogue_rm (barriers) | request slab i + 2.
// Shape: one directly read source, C0 == Cout == 64, 3 x 3 / stride 1 / pad 1, EP 0 - 2 (plain, statistics, BatchNormalization backward).
#if !defined(STP_P64_STAGGER)
#define STP_P64_STAGGER 4
#endif
template <int TH, int BM, int WM, int WN, int EP>
__global__ __launch_bounds__(256, 2) void conv_halo_p64_kernel(const ConvArgs a_in) {
  static_assert(TH == 8 && BM == 64 && WM == 2 && WN == 2 && EP <= 2, "the 64 -> 64 form");
#if defined(__HIP_DEVICE_COMPILE__)
#if defined(STP_TIMING)   // scratch build: `bias` carries u64[4 * workgroups]: prologue, sum of the K loops, of the slab waits, of the epilogues (scratch/r05/p64_phase.py)
  ConvArgs a = a_in;
  unsigned long long* const tdbg = reinterpret_cast<unsigned long long*>(const_cast<float*>(a_in.bias));
  a.bias = nullptr;
  unsigned long long tsum[4] = {0, 0, 0, 0}, tlast = STP_CLOCK();
  const unsigned long long tfirst = tlast;
#define P64_LAP(i) { const unsigned long long now_ = STP_CLOCK(); tsum[i] += now_ - tlast; tlast = now_; }
#else
  const ConvArgs& a = a_in;
#define P64_LAP(i)
#endif
  typedef bf16_t T;
  constexpr int NTH = 256;
  constexpr int TW = 16, HWD = TW + 2, HH = TH + 2, NHP = HH * HWD;
  constexpr int SROWS = (NHP + 7) / 8 * 8, SLAB = SROWS * 128;
  constexpr int NPASS = (SROWS + 31) / 32;         // 32 halo pixels (8 per wave) per pass of the 256 threads
  constexpr int RW = TH / WN, TN = RW / 2;         // a wave: 32 channels x RW rows of 16 pixels = TN MFMA tiles
  constexpr int PF = 2;                            // B fragments are read PF (tap, k-chunk) steps ahead of their MFMAs
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const stage = smem + 2 * SLAB;             // the epilogue's staged fp32 tile (+ its reduction scratch)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int l15 = lane & 15, l31 = lane & 31, l4b = (lane >> 4) & 1, l5 = lane >> 5;
  const int ntile = a.ntile_n, nwg = gridDim.x;
  const int wg = xcd_remap(blockIdx.x, nwg);       // neighbouring tiles of a round share an L2
  const int tx_n = a.Wo / TW, ty_n = a.Ho / TH, tpi = tx_n * ty_n;
  const int prow = tid >> 3, pslot = tid & 7;

  if (STP_P64_STAGGER > 0 && (int)blockIdx.x >= (nwg >> 1)) {          // second slot of a CU: half a tile behind the first
    for (int i = 0; i < STP_P64_STAGGER; ++i) __builtin_amdgcn_s_sleep(16);
  }
  // ---- the weights, once: [tap][64 rows][128 bytes] (the main kernel's swizzled stage image) over the slab + staging space -> A fragments
  {
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)a.weight, 0, a.bytesw, 0x00020000);
#pragma unroll
    for (int i = 0; i < 18; ++i) {                  // pass i: rows (i & 1) * 32 + tid / 8 of tap i / 2
      const int r = (i & 1) * 32 + prow;
      const uint32_t off = ((uint32_t)r * (uint32_t)a.K + (uint32_t)((pslot ^ ((r >> 1) & 7)) * 8)) * 2u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (__attribute__((address_space(3))) void*)(smem + (i * 32 + wave * 8) * 128), 16, off, (uint32_t)(i >> 1) * 128u, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  u32x4 fa[9][4];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int kc = 0; kc < 4; ++kc)
      fa[t][kc] = *reinterpret_cast<const u32x4*>(smem + (t * 64 + wm * 32 + l31) * 128 + (((kc * 2 + l5) ^ ((l31 >> 1) & 7)) << 4));
  lds_barrier();                                   // every wave holds its fragments: the space is free for the slabs

  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)a.src0, 0, a.bytes0, 0x00020000);
  // slab pass p: halo pixel hp = p * 32 + tid / 8 at physical 16-byte slot tid & 7 = logical slot (tid & 7) ^ ((hx >> 1) & 7) (the main kernel's image)
  auto request_slab = [&](int tile, int buf) {
    const int n = tile / tpi, trem = tile - n * tpi, ty = trem / tx_n, tx = trem - ty * tx_n;
    const int y0 = ty * TH, x0 = tx * TW;
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
      if ((p * 32 + wave * 8) >= SROWS) continue;                      // wave-uniform: rows past the slab
      const int hp = p * 32 + prow;
      const int hy = hp / HWD, hx = hp - hy * HWD;
      const int y = y0 - 1 + hy, x = x0 - 1 + hx;
      const bool ok = hp < NHP && (unsigned)y < (unsigned)a.Hv && (unsigned)x < (unsigned)a.Wv;
      const uint32_t off = ok ? (((uint32_t)(n * a.Hv + y) * (uint32_t)a.Wv + (uint32_t)x) * 64u + (uint32_t)((pslot ^ ((hx >> 1) & 7)) * 8)) * 2u : STP_OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, (__attribute__((address_space(3))) void*)(smem + buf * SLAB + (p * 32 + wave * 8) * 128), 16, off, 0, 0, 0);
    }
  };
  if (wg < ntile) request_slab(wg, 0);
  if (wg + nwg < ntile) request_slab(wg + nwg, 1);

  // B (pixels): halo pixel (wn * RW + 2 j + l4b + dy) * 18 + l15 + dx of the slab -> + (2 j + dy) * 2304 (immediate) + the buffer
  uint32_t b_lane[3][4];
#pragma unroll
  for (int kc = 0; kc < 4; ++kc)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int hx = l15 + dx;
      b_lane[dx][kc] = (uint32_t)(((wn * RW + l4b) * HWD + hx) * 128 + (((kc * 2 + l5) ^ ((hx >> 1) & 7)) << 4));
    }

  int buf = 0;
  for (int tile = wg; tile < ntile; tile += nwg, buf ^= 1) {
    const int n = tile / tpi, trem = tile - n * tpi, ty = trem / tx_n, tx = trem - ty * tx_n;
    const int y0 = ty * TH, x0 = tx * TW;
    if (tile == wg) {                              // first tile: its slab has to be here; later ones were awaited below
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      P64_LAP(0);
    }
    f32x16 acc[1][TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[0][j][e] = 0.f;
    const char* const sb = smem + buf * SLAB;
    u32x4 fb[PF + 1][TN];
#pragma unroll
    for (int q = 0; q < PF; ++q)
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[q][j] = *reinterpret_cast<const u32x4*>(sb + b_lane[(q >> 2) % 3][q & 3] + (2 * j + (q >> 2) / 3) * (HWD * 128));
#pragma unroll
    for (int q = 0; q < 36; ++q) {
      if (q + PF < 36) {
        const int t = (q + PF) >> 2, kc = (q + PF) & 3, dy = t / 3, dx = t - dy * 3;
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[(q + PF) % (PF + 1)][j] = *reinterpret_cast<const u32x4*>(sb + b_lane[dx][kc] + (2 * j + dy) * (HWD * 128));
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[0][j] = mfma16_32x32x16(fa[q >> 2][q & 3], fb[q % (PF + 1)][j], acc[0][j]);
      __builtin_amdgcn_sched_barrier(0);           // (keeps the scheduler from hoisting the tile's 72 reads over the 144 resident A registers)
    }
    P64_LAP(1);
    // the next tile's slab (requested one K loop ago) has landed: own pieces here, everyone's behind the epilogue's first barrier
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    P64_LAP(2);
    epilogue_rm<TH, BM, WM, WN, EP, NTH>(a, acc, stage, n, y0, x0, 0, tile, wm, wn, lane, tid, wave);
    // every wave has left this tile's K loop (the epilogue's barriers): its slab buffer takes the tile after the next
    if (tile + 2 * nwg < ntile) request_slab(tile + 2 * nwg, buf);
    P64_LAP(3);
  }
#if defined(STP_TIMING)
  if (tid == 0) {
    for (int i = 0; i < 4; ++i) tdbg[(size_t)blockIdx.x * 8 + i] = tsum[i];
    tdbg[(size_t)blockIdx.x * 8 + 4] = tfirst;
    tdbg[(size_t)blockIdx.x * 8 + 5] = tlast;
    tdbg[(size_t)blockIdx.x * 8 + 6] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));      // HW_ID: wave / SIMD / CU / SE / XCC ids
  }
#endif
#endif
}

// ================================================================================================ host side
struct HaloCfg { int th, bm; };
// variant ids (stp_conv_params.tile = STP_TILE_HALO + id)
static const HaloCfg HALO_CFGS[] = {{16, 128}, {8, 128}, {16, 64}, {8, 64}, {32, 64}, {8, 64}};       // 5: the persistent 64 -> 64 form (8 x 16-pixel tiles, two 4-wave workgroups per CU)
#define STP_TILE_HALO 1024
#define HALO_NCFG 6

template <int TH, int BM, int WM, int WN, int EP>
static int launch_halo_ep(ConvArgs& a, hipStream_t s) {
  static bool attr_set = false;
  constexpr int NHP = (TH + 2) * 18, SROWS = (NHP + 7) / 8 * 8;
  const bool src2 = a.C1 > 0 || a.mode == STP_SRC_NEAREST2X || a.d2s;
  // STP_HALO_RING=6: the 8-row tiles with one source take the six-stage weight ring (experiment switch; default below)
  static const int ring_env = getenv("STP_HALO_RING") ? atoi(getenv("STP_HALO_RING")) : 4;
  const bool ring6 = ring_env == 6 && TH == 8 && !src2 && !a.pbn.mean && !a.fold_weight;
  size_t lds = (size_t)((src2 ? a.Ctot : a.C0) > 64 ? 2 : 1) * SROWS * 128 + (ring6 ? 6 : HALO_NWST) * BM * 128 + (a.pbn.mean ? (size_t)8 * a.C0 : 0);
  const size_t lds_ep = (size_t)TH * 16 * (BM * 4 + 16);   // the epilogue's staged fp32 tile
  if (lds < lds_ep) lds = lds_ep;
  a.ntile_m = ceil_div(a.Cout, BM);
  a.ntile_n = a.N * (a.Ho / TH) * (a.Wo / 16);
  static bool attr_set_pbn = false, attr_set_src2 = false;
  const bool pbn = a.pbn.mean != nullptr;
  static bool attr_set_s2d = false, attr_set_fold1 = false;
  const bool fold1 = !a.d2s && a.fold_weight != nullptr;      // (set by stp_conv2d_halo: fold_src / fold_weight / fold_C on a stride-1 launch)
  if (a.d2s) {
    if (pbn || BM != 128 || (EP != 0 && EP != 2)) return STP_E_BADARG;
  } else if (fold1) {
    if (pbn || (EP != 0 && EP != 2)) return STP_E_BADARG;
  } else if (src2 && (pbn || EP > 1)) return STP_E_BADARG;
  auto kern = pbn ? conv_halo_pbn_kernel<TH, BM, WM, WN, EP> : conv_halo_kernel<TH, BM, WM, WN, EP>;
  if constexpr (EP <= 1) {
    if (src2 && !a.d2s && !fold1) kern = conv_halo2_kernel<TH, BM, WM, WN, EP>;
  }
  if constexpr (EP == 0 || EP == 2) {
    if (fold1) kern = conv_halo_fold1_kernel<TH, BM, WM, WN, EP>;
  }
  if constexpr (BM == 128 && (EP == 0 || EP == 2)) {
    if (a.d2s) kern = conv_halo_s2d_kernel<TH, BM, WM, WN, EP>;
  }
  static bool attr_set_r6 = false;
  if constexpr (TH == 8) {
    if (ring6) kern = conv_halo_r6_kernel<TH, BM, WM, WN, EP>;
  }
  bool& done = ring6 ? attr_set_r6 : a.d2s ? attr_set_s2d : fold1 ? attr_set_fold1 : src2 ? attr_set_src2 : pbn ? attr_set_pbn : attr_set;
  if (!done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return STP_E_LAUNCH;
    done = true;
  }
  hipLaunchKernelGGL(kern, dim3(a.ntile_m * a.ntile_n), dim3(512), lds, s, a);
  STP_LAUNCH_CHECK();
  return STP_OK;
}

template <int TH, int BM, int WM, int WN>
static int launch_halo(ConvArgs& a, hipStream_t s) {
  if (a.sum2x2) return launch_halo_ep<TH, BM, WM, WN, 3>(a, s);
  if (a.bnb.x) return launch_halo_ep<TH, BM, WM, WN, 2>(a, s);
  if (a.stats) return launch_halo_ep<TH, BM, WM, WN, 1>(a, s);
  return launch_halo_ep<TH, BM, WM, WN, 0>(a, s);
}

// two sources / an upsampled first source (round 5, forward launches: no BatchNormalization-backward epilogue, no fused producer
// BatchNormalization, one destination).  STP_HALO2=0 sends those launches back to the per-tap kernel (A/B).
static bool halo2_enabled() {
  static const bool on = !(getenv("STP_HALO2") && atoi(getenv("STP_HALO2")) == 0);
  return on;
}
static bool halo_src_ok(const stp_conv_params* p) {
  const bool direct = p->src0_mode == STP_SRC_DIRECT && p->Hs0 == p->Hv && p->Ws0 == p->Wv;
  if (direct && p->C1 == 0) return true;
  const bool up2 = p->src0_mode == STP_SRC_NEAREST2X && p->Hs0 * 2 == p->Hv && p->Ws0 * 2 == p->Wv;
  return halo2_enabled() && (direct || up2) && (p->C1 == 0 || (p->src1 && (p->C1 % 64) == 0)) && !p->src_bn_mean && !p->bnb_x && !p->dst_sum2x2 &&
         p->Cd0 == p->Cout;
}
// space-to-depth data gradient (stp_conv_params.s2d_dgrad; STP_S2D=0 refuses it - the plan then keeps the parity-class launch)
static bool halo_s2d_ok(const stp_conv_params* p) {
  static const bool on = !(getenv("STP_S2D") && atoi(getenv("STP_S2D")) == 0);
  return on && p && p->s2d_dgrad && p->dtype == STP_H16 && p->KH == 2 && p->KW == 2 && p->stride == 1 && p->pad == 0 && p->src0_mode == STP_SRC_DIRECT &&
         p->Hs0 == p->Hv && p->Ws0 == p->Wv && p->Ho == p->Hv && p->Wo == p->Wv && (p->Wo % 16) == 0 && (p->Ho % 8) == 0 && p->C0 >= 64 &&
         (p->C0 % 64) == 0 && (p->C1 == 0 || (p->src1 && p->fold_weight && (p->C1 % 64) == 0)) && (p->Cout % 128) == 0 && p->Cd0 == p->Cout && !p->dst1 &&
         !p->dst_sum2x2 && !p->stats_slots && !p->src_bn_mean && !p->bias && !p->relu && !p->residual && !p->accumulate1;
}
// fold_src / fold_weight / fold_C on a 3x3 / stride-1 / pad-1 launch: the dY of a sibling 1x1 / stride-1 convolution over the same input
// (STP_FOLD1=0 refuses it)
static bool halo_fold1_ok(const stp_conv_params* p) {
  static const bool on = !(getenv("STP_FOLD1") && atoi(getenv("STP_FOLD1")) == 0);
  return on && p->fold_src && p->fold_weight && p->fold_C >= 64 && (p->fold_C % 64) == 0 && p->C1 == 0 && p->src0_mode == STP_SRC_DIRECT &&
         !p->src_bn_mean && !p->dst_sum2x2 && !p->dst1 && p->Cd0 == p->Cout && !p->bias && !p->relu && !p->residual;
}
// persistent 64 -> 64 form (variant 5, conv_halo_p64_kernel; STP_HALO_P64=0 refuses it): one directly read source, C0 == Cout == 64, one
// destination, no fused producer BatchNormalization, no 2 x 2-summed destination
static bool halo_p64_ok(const stp_conv_params* p) {
  static const bool on = !(getenv("STP_HALO_P64") && atoi(getenv("STP_HALO_P64")) == 0);
  return on && p->C0 == 64 && p->C1 == 0 && p->Cout == 64 && p->Cd0 == 64 && !p->dst1 && p->src0_mode == STP_SRC_DIRECT && !p->fold_src && !p->s2d_dgrad &&
         !p->src_bn_mean && !p->dst_sum2x2 && (p->Ho % 8) == 0;
}
static int halo_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    n = v;
  }
  return n;
}
template <int EP>
static int launch_halo_p64_ep(ConvArgs& a, hipStream_t s) {
  constexpr int TH = 8, NHP = (TH + 2) * 18, SROWS = (NHP + 7) / 8 * 8;
  const size_t lds = (size_t)2 * SROWS * 128 + (size_t)TH * 16 * (64 * 4 + 16);      // two slabs + the epilogue's staged tile: 80 KB, two workgroups per CU
  static_assert(2 * SROWS * 128 + TH * 16 * (64 * 4 + 16) >= 9 * 64 * 128, "the weight image fits the workgroup's LDS");
  a.ntile_m = 1;
  a.ntile_n = a.N * (a.Ho / TH) * (a.Wo / 16);
  auto kern = conv_halo_p64_kernel<8, 64, 2, 2, EP>;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return STP_E_LAUNCH;
    attr_set = true;
  }
  const int slots = 2 * halo_cus();
  const int nwg = a.ntile_n < slots ? a.ntile_n : slots;       // two workgroups per CU walk tile = round * nwg + workgroup
  hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), lds, s, a);
  STP_LAUNCH_CHECK();
  return STP_OK;
}
static int launch_halo_p64(ConvArgs& a, hipStream_t s) {
  if (a.sum2x2 || a.pbn.mean) return STP_E_BADARG;
  if (a.bnb.x) return launch_halo_p64_ep<2>(a, s);
  if (a.stats) return launch_halo_p64_ep<1>(a, s);
  return launch_halo_p64_ep<0>(a, s);
}

static bool halo_shape_ok(const stp_conv_params* p) {
  if (p && p->s2d_dgrad) return halo_s2d_ok(p);
  if (p && p->fold_src && !halo_fold1_ok(p)) return false;
  return p && p->dtype == STP_H16 && p->KH == 3 && p->KW == 3 && p->stride == 1 && p->pad == 1 && halo_src_ok(p) &&
         p->C0 >= 64 && (p->C0 % 64) == 0 && p->Ho == p->Hv && p->Wo == p->Wv &&
         (p->Wo % 16) == 0 && (p->Ho % 8) == 0 && (p->Cout % 16) == 0 && p->Cout >= 64 && !p->stats_slots &&
         // 2 x 2-summed first destination (EP 3): two destinations, fused BatchNormalization backward of the summed one, whole 64-channel tiles
         // (or ONE destination, every channel tile summed: Cd0 == Cout, the data gradient of conv3x3(UpSampling2D(2)(x)))
         (!p->dst_sum2x2 || (p->bnb_x && (p->Cd0 % 64) == 0 && !p->bias && !p->relu && !p->residual &&
                             (p->Cd0 == p->Cout || (p->dst1 && p->Cd0 < p->Cout)))) &&
         (p->Cd0 % 8) == 0 &&
         (!p->src_bn_mean || (p->src_bn_rstd && p->C0 <= 512));
}

// variant for a shape: -1 = not eligible / not faster.  Measured on MI355X against the per-tap DMA kernel (scratch/halo_bench.py,
// bs16 U-Net/ResNet34 shapes).  One 8-wave workgroup per CU (100-150 KB of LDS) for inputs of 128+ channels, so the grid should
// cover the CUs.
static int halo_auto(const stp_conv_params* p) {
  if (!halo_shape_ok(p)) return -1;
  {   // experiment switch: STP_HALO_FORCE=<variant> uses that variant wherever its shape rules allow (co-residency studies)
    static const int force = getenv("STP_HALO_FORCE") ? atoi(getenv("STP_HALO_FORCE")) : -1;
    if (force >= 0 && force < 4 && (p->Ho % (force == 0 || force == 2 ? 16 : 8)) == 0) return force;
  }
  const int64_t px16 = (p->Ho % 16) == 0 ? (int64_t)p->N * (p->Ho / 16) * (p->Wo / 16) : 0;
  const int64_t px8 = (int64_t)p->N * (p->Ho / 8) * (p->Wo / 16);
  if (p->s2d_dgrad) {     // 128-channel tiles only (a channel tile of 4 x Cq outputs never straddles... it may: classes are handled per thread)
    const int ct2 = ceil_div(p->Cout, 128);
    if (px16 * ct2 >= 256) return 0;
    return px8 * ct2 >= 192 ? 1 : -1;
  }
  if (p->C1 > 0 || p->src0_mode != STP_SRC_DIRECT) {
    // two sources / upsampled source, 64-channel outputs (decoder_stage2_conv1: 128 upsampled + 64 skip -> 64 @ 16 x 128 x 128): 69 us on the
    // 8 x 16 x 64 tiles against 89 us on the per-tap kernel (profiles/r05c_*); STP_HALO2_64=0 sends them back
    static const bool c64 = !(getenv("STP_HALO2_64") && atoi(getenv("STP_HALO2_64")) == 0);
    if (p->Cout <= 64 && !c64) return -1;
  }
  if (p->fold_src && p->C0 == 64) {
    // a folded 1x1 sibling is a second slab: the 16 x 16 tiles would lose their second workgroup per CU (116 KB) - 8 x 16 x 64 keeps it (79 KB)
    return px8 * ceil_div(p->Cout, 64) >= 512 ? 3 : -1;
  }
  if (p->C0 == 64 && p->C1 == 0) {   // one slab, nine K-steps: the 64-channel tiles keep one slab + the ring under 80 KB, two workgroups per CU
    // several channel tiles over the same pixels (a data gradient into concatenated sources: 64 -> 192): 32 x 16 pixel tiles halve
    // the weight stream per pixel (scratch/halo_bench.py: 89 -> 81 us); one channel tile: no gain (35 us either way)
    if (p->Cout > 64 && (p->Ho % 32) == 0 && (int64_t)p->N * (p->Ho / 32) * (p->Wo / 16) * ceil_div(p->Cout, 64) >= 512) return 4;
    static const int p64 = getenv("STP_HALO_P64") ? atoi(getenv("STP_HALO_P64")) : 0;      // (experiment: 1 = the persistent form wherever a CU gets two or more tiles)
    if (p64 == 1 && halo_p64_ok(p) && px8 >= 4 * halo_cus()) return 5;      // (two or more tiles per workgroup)
    static const int v4 = getenv("STP_HALO_64V4") ? atoi(getenv("STP_HALO_64V4")) : 0;      // (experiment: 32-row tiles for one channel tile, too)
    if (v4 == 1 && (p->Ho % 32) == 0 && (int64_t)p->N * (p->Ho / 32) * (p->Wo / 16) >= 512) return 4;
    if (px16 * ceil_div(p->Cout, 64) >= 512) return 2;
    if (px8 * ceil_div(p->Cout, 64) >= 512) return 3;
    return -1;
  }
  const int ct = ceil_div(p->Cout, 128);
  if (p->Cout > 64 && px16 * ct >= 256) return 0;
  if (p->Cout > 64 && px8 * ct >= 192) return 1;
  if (px8 * ceil_div(p->Cout, 64) >= 192) return 3;
  return -1;
}

extern "C" int stp_conv2d_halo_variant(const stp_conv_params* p) {
  if (!p) return -1;
  int v = -1;
  if (p->tile >= STP_TILE_HALO && p->tile < STP_TILE_HALO + HALO_NCFG) {
    v = p->tile - STP_TILE_HALO;
    if (!halo_shape_ok(p) || (p->Ho % HALO_CFGS[v].th) || (v == 4 && (p->C0 != 64 || p->C1)) || (p->s2d_dgrad && v > 1) || (v == 5 && !halo_p64_ok(p))) return -1;
  } else if (p->tile == 0) {
    v = halo_auto(p);
  }
  if (v >= 0 && p->dst_sum2x2 && (p->Cd0 % HALO_CFGS[v].bm)) return -1;      // (a channel tile lies in ONE destination)
  return v;
}

// number of pixel tiles (BatchNormalization partial-sum columns) of a variant
extern "C" int stp_conv2d_halo_tiles(const stp_conv_params* p, int variant) {
  return p->N * (p->Ho / HALO_CFGS[variant].th) * (p->Wo / 16);
}

extern "C" int stp_conv2d_halo(const stp_conv_params* p, int variant, void* stream) {
  if (variant < 0 || variant >= HALO_NCFG || !halo_shape_ok(p) || (p->Ho % HALO_CFGS[variant].th) || (variant == 4 && (p->C0 != 64 || p->C1)) ||
      (p->s2d_dgrad && variant > 1) || (variant == 5 && !halo_p64_ok(p))) return STP_E_BADARG;
  ConvArgs a;
  bool c4;
  int ut;
  const int rc = fill_args(p, a, &c4, &ut);
  if (rc != STP_OK) return rc;
  if (ut != 1) return STP_E_BADARG;   // 32-bit buffer offsets, 64-channel K-steps
  if (a.bnb.x && (!a.stats || !a.bnb.mean || !a.bnb.rstd || p->relu || p->residual)) return STP_E_BADARG;
  if (a.sum2x2 && (!a.bnb.x || (p->Cd0 % HALO_CFGS[variant].bm))) return STP_E_BADARG;
  if (!p->s2d_dgrad && p->fold_src) {      // stride-1 sibling 1x1 folded in: its dY is the second source, its weights ride in fold_weight
    if (!halo_fold1_ok(p)) return STP_E_BADARG;
    const int64_t bf = (int64_t)p->N * p->Hv * p->Wv * p->fold_C * 2, bwf = (int64_t)round_up(p->Cout, 16) * p->fold_C * 2;
    if (bf >= (1ll << 31) || bwf >= (1ll << 31)) return STP_E_BADARG;
    a.src1 = (const char*)p->fold_src; a.C1 = p->fold_C; a.Ctot = a.C0 + a.C1; a.bytes1 = (uint32_t)bf;
    a.fold_weight = (const char*)p->fold_weight; a.bytesw_fold = (uint32_t)bwf;
    // (a.K stays 9 x C0: the row pitch of the 3 x 3 layer's own weight copy)
  }
  if (p->s2d_dgrad) {      // weights: the ordinary data-gradient copies of the 3x3 layer (and of the shortcut), addressed per parity class
    const int64_t rows = round_up(p->Cout / 4, 16), bw = rows * 9 * p->C0 * 2, bf = rows * (int64_t)p->C1 * 2;
    if (bw >= (1ll << 31) || bf >= (1ll << 31) || (p->C1 > 0 && !p->fold_weight)) return STP_E_BADARG;
    a.bytesw = (uint32_t)bw;
    a.fold_weight = p->C1 > 0 ? (const char*)p->fold_weight : nullptr;
    a.bytesw_fold = (uint32_t)bf;
  }
  const_cast<stp_conv_params*>(p)->stats_tiles = stp_conv2d_halo_tiles(p, variant) * (p->s2d_dgrad ? 4 : 1);   // (s2d: four column blocks per channel)
  hipStream_t s = (hipStream_t)stream;
  switch (variant) {
    case 0: return launch_halo<16, 128, 2, 4>(a, s);   // wave: 64 channels x 4 rows
    case 1: return launch_halo<8, 128, 2, 4>(a, s);    // wave: 64 channels x 2 rows
    case 2: return launch_halo<16, 64, 1, 8>(a, s);    // wave: 64 channels x 2 rows
    case 3: return launch_halo<8, 64, 2, 4>(a, s);     // wave: 32 channels x 2 rows
    case 4: return launch_halo<32, 64, 1, 8>(a, s);    // wave: 64 channels x 4 rows; 64-channel inputs: the weights stream once per 512 pixels
    case 5: return launch_halo_p64(a, s);              // wave: 32 channels x 4 rows, weights in registers, one workgroup per CU over its tiles
    default: return STP_E_BADARG;
  }
}
