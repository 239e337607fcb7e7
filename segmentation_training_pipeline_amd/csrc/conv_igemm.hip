// Implicit-GEMM convolution on MFMA for gfx950 (forward and data-gradient).
//
// GEMM view:  C[cout][pixel] = sum_k  W[cout][k] * V[pixel][k],   k = (kh*KW + kw)*Ctot + c
//   A operand (MFMA rows  i) = weight rows  (K-contiguous in HBM, OHWI layout)
//   B operand (MFMA cols  j) = im2col rows gathered on the fly from NHWC activations, with
//     nearest-2x upsampling / zero-insertion of src0 and channel-concat of src1 folded into the
//     gather (the upsampled / concatenated / zero-inserted tensors are never materialised).
// With the 16x16 MFMA C layout (col = lane&15, row = (lane>>4)*4 + r) every lane ends up with 4
// consecutive output channels of one pixel, i.e. one 8-byte (bf16) or 16-byte (fp32) NHWC store.
//
// Tile bytes are identical for both dtypes: every LDS row holds 128 bytes of K (64 bf16 / 32
// fp32); a lane's 16-byte fragment read feeds one v_mfma_f32_16x16x32_bf16 or four
// v_mfma_f32_16x16x4_f32 (exact fp32, used by the parity mode).  LDS rows are XOR-swizzled in
// 16-byte slots (slot ^= row & 7): conflict-free for the ds_read_b128 lane groups of gfx950.
//
// Two kernels share the MFMA loop and the epilogue:
//  * conv_igemm_ut_kernel ("uniform tap"): when Ctot and C0 are multiples of the K-step, all 8 slots
//    of a K-tile belong to ONE filter tap and ONE source, so the tap decomposition is scalar (SALU) and
//    every 16-byte vector is fetched with `buffer_load_dwordx4 ... offen lds` straight into LDS: 32-bit
//    offsets, hardware out-of-range -> 0 for padding, no VGPR staging, no ds_write.  The LDS image of
//    an LDS-DMA is lane-linear, so the swizzle is applied to the SOURCE (the thread owning physical
//    slot s of row r fetches logical slot s ^ (r&7)).  STAGES-deep LDS ring, counted s_waitcnt vmcnt(N)
//    + bare s_barrier (never a drain inside the loop).
//  * conv_igemm_kernel (general): per-lane tap decomposition, global -> VGPR -> ds_write double buffer;
//    handles the 4-channel stem (C4), 16/32-channel layers and ragged shapes.
#include "conv_common.h"
#include <cstdlib>

// ================================================================================================
// uniform-tap buffer-DMA kernel
// ================================================================================================
#define STP_OOB 0x80000000u  // voffset beyond any descriptor: the buffer load returns 0 (conv padding)

// UNI = true : Ctot % KE == 0 and C0 % KE == 0 -> one tap and one source per K-tile (scalar decomposition)
// UNI = false: KE % Ctot == 0 and C1 == 0 (16/32-channel layers): a K-tile spans KE/Ctot taps, the tap is a
//              per-lane constant offset from a scalar base, still no per-load division
// UPC = true : the instance the class-collapsed upsample + concat layers run on (ConvArgs::upc).  A compile-time parameter: as a run-
//              time branch its registers and address arithmetic cost every OTHER user of the kernel - measured +9 % on the FPN/ResNet50
//              and +11 % on the PSPNet/ResNet101 step (bottleneck 1x1 convolutions), for -0.5 % on the U-Net.
// ZP = true  : the instance the parity-class (zperm) launches run on - stride-2 data gradients and the UPC layers; same reasoning.
template <typename T, int BM, int BN, int WM, int WN, int STAGES, bool UNI, bool UPC = false, bool ZP = false>
__global__ __launch_bounds__(256) void conv_igemm_ut_kernel(const ConvArgs a) {
  static_assert(!UPC || ZP, "the class-collapsed layers use the parity-class pixel order");
  static_assert(WM * WN == 4 && BM % 16 == 0 && BN % 32 == 0 && STAGES >= 2, "config");
#if defined(__HIP_DEVICE_COMPILE__)  // the buffer-resource builtins exist in the device pass only; the host pass needs just the stub
  constexpr int SZ = (int)sizeof(T);
  constexpr int VEC = Elem<T>::VEC;
  constexpr int KE = 128 / SZ;
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
  constexpr int RA = (BM + 31) / 32, RB = BN / 32;
  constexpr int STAGE = (BM + BN) * 128;
  constexpr int L = RA + RB;  // VMEM instructions per wave per K-tile
  constexpr int DUMP = STAGES * STAGE;  // 4 KiB sink: keeps every wave at L loads per tile when BM < 32

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int r0 = tid >> 3;
  const int lslot = (tid & 7) ^ (r0 & 7);  // logical 16-byte slot fetched by this thread (row&7 == r0&7 for all passes)

#if defined(STP_EXP) && STP_EXP == 5   // what-if: empty workgroups (launch + dispatch cost of the grid)
  if (a.P >= 0) return;
#endif
  const int bid = xcd_remap(blockIdx.x, a.ntile_m * a.ntile_n);
  const int tile_n = (int)fdiv((uint32_t)bid, a.divNtm), tile_m = bid - tile_n * a.ntile_m;
  const int cout0 = tile_m * BM, pix0 = tile_n * BN;

  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)a.weight, 0, a.bytesw, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)a.src0, 0, a.bytes0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.src1 ? a.src1 : a.src0), 0, a.src1 ? a.bytes1 : 0u, 0x00020000);

  // zperm: this tile's parity class fixes the taps that meet real samples of the zero-inserted source (all scalar)
  const bool zp = UNI && ZP && a.zperm;
  // upc: parity-class pixel order too; the class fixes which 2 x 2 low-resolution pixels the taps over the upsampled src0 read
  const bool up = UNI && UPC && a.upc;
  const __amdgpu_buffer_rsrc_t rswu = __builtin_amdgcn_make_buffer_rsrc((void*)(up ? a.weight_up : a.weight), 0, up ? a.byteswu : 0u, 0x00020000);
  const int upy = up ? ((int)fdiv((uint32_t)pix0, a.divPc) >> 1) : 0, upx = up ? ((int)fdiv((uint32_t)pix0, a.divPc) & 1) : 0;
  int zkh0 = 0, zkw0 = 0, znkw = 1, znk = 0, zfn = 0;
  if (zp && !up) {
    const int zc = (int)fdiv((uint32_t)pix0, a.divPc);
    zkh0 = ((zc >> 1) + a.pad) & 1;                  // ho - pad + kh even  <=>  kh = zkh0 (mod 2)
    zkw0 = ((zc & 1) + a.pad) & 1;
    const int znkh = (a.KH - zkh0 + 1) >> 1;
    znkw = (a.KW - zkw0 + 1) >> 1;
    znk = znkh * znkw * a.zcpt;
    // folded shortcut gradient: the 1x1 / stride-2 sibling reaches the (even, even) pixels only, as one more (centre) tap
    if (a.fold_src && zc == 0) zfn = a.fold_cpt;
  }
  const __amdgpu_buffer_rsrc_t rsf = __builtin_amdgcn_make_buffer_rsrc((void*)(zfn ? a.fold_src : a.src0), 0, zfn ? a.bytes_fold : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rswf = __builtin_amdgcn_make_buffer_rsrc((void*)(zfn ? a.fold_weight : a.weight), 0, zfn ? a.bytesw_fold : 0u, 0x00020000);

  // per-thread pixel rows: image offsets (elements) in both sources and the top-left tap coordinate
  int hb[RB], wb[RB];
  uint32_t nof0[RB], nof1[RB];
#pragma unroll
  for (int i = 0; i < RB; ++i) {
    const int pm = pix0 + r0 + 32 * i;
    if (pm < a.P) {
      int n, ho, wo;
      if (zp) {
        zperm_decode(a, pm, n, ho, wo);
      } else {
        n = (int)fdiv((uint32_t)pm, a.divHoWo);
        const int rem = pm - n * a.HoWo;
        ho = (int)fdiv((uint32_t)rem, a.divWo);
        wo = rem - ho * a.Wo;
      }
      hb[i] = ho * a.stride - a.pad;
      wb[i] = wo * a.stride - a.pad;
      nof0[i] = (uint32_t)n * (uint32_t)(a.Hs0 * a.Ws0 * a.C0);
      nof1[i] = (uint32_t)n * (uint32_t)(a.Hv * a.Wv * a.C1);
    } else {
      hb[i] = wb[i] = -(1 << 24);
      nof0[i] = nof1[i] = 0;
    }
  }
  // !UNI: this lane's tap offset inside a K-tile and its channel offset inside the tap
  const uint32_t lane_dpos = UNI ? 0u : fdiv((uint32_t)(lslot * VEC), a.divC);
  const uint32_t lane_ci = UNI ? 0u : (uint32_t)(lslot * VEC) - lane_dpos * (uint32_t)a.Ctot;
  // weight rows: byte offset of (row, logical slot); rows past the allocation fall out of the descriptor
  uint32_t wof[RA];
#pragma unroll
  for (int i = 0; i < RA; ++i) wof[i] = ((uint32_t)(cout0 + r0 + 32 * i) * (uint32_t)a.K + (uint32_t)lslot * VEC) * SZ;
  uint32_t wofu[RA];   // rows of weight_up: [Cout_pad][16 * C0]
#pragma unroll
  for (int i = 0; i < RA; ++i) wofu[i] = ((uint32_t)(cout0 + r0 + 32 * i) * (uint32_t)(16 * a.C0) + (uint32_t)lslot * VEC) * SZ;

  auto issue_tile = [&](int kt, int buf) {
    char* sa = smem + buf * STAGE;
    char* sb = sa + BM * 128;
    uint32_t k0 = (uint32_t)kt * KE;  // wave-uniform from here: scalar tap decomposition
    if (up) {
      const int nk0 = 4 * a.ucpt0;
      if (kt < nk0) {   // collapsed tap t = ty * 2 + tx of this class over the LOW-RESOLUTION src0, summed weights
        const int t = (int)fdiv((uint32_t)kt, a.divU0), ch = kt - t * a.ucpt0;
        const uint32_t ku = (uint32_t)((((upy * 2 + upx) * 4 + t) * a.C0 + ch * KE) * SZ);
#pragma unroll
        for (int i = 0; i < RA; ++i) {
          const bool act = (i * 32 + wave * 8) < BM;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(
              rswu, (__attribute__((address_space(3))) void*)(act ? sa + (i * 32 + wave * 8) * 128 : smem + DUMP + wave * 1024), 16,
              act ? wofu[i] + ku : STP_OOB, 0, 0, 0);
        }
        const int dy = (t >> 1) - 1 + upy, dx = (t & 1) - 1 + upx;
        const uint32_t ci = (uint32_t)(ch * KE + lslot * VEC);
#pragma unroll
        for (int i = 0; i < RB; ++i) {
          // hb = ho - 1 with ho = 2 * i_lo + upy (pad 1, stride 1): i_lo = (hb + 1) >> 1
          const int hv = ((hb[i] + 1) >> 1) + dy, wv = ((wb[i] + 1) >> 1) + dx;
          const bool ok = (unsigned)hv < (unsigned)a.Hs0 && (unsigned)wv < (unsigned)a.Ws0;
          const uint32_t off = nof0[i] + (uint32_t)(hv * a.Ws0 + wv) * (uint32_t)a.C0 + ci;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, (__attribute__((address_space(3))) void*)(sb + (i * 32 + wave * 8) * 128), 16,
                                                   ok ? off * SZ : STP_OOB, 0, 0, 0);
        }
        return;
      }
      // the nine taps of src1 with the original weights
      const int k1 = kt - nk0, tap = (int)fdiv((uint32_t)k1, a.divU1), ch = k1 - tap * a.ucpt1;
      k0 = (uint32_t)(tap * a.Ctot + a.C0 + ch * KE);
    } else if (zp) {   // kt-th K-tile of the class: (live tap, 64-channel chunk)
      if (ZP && kt >= znk) {   // folded shortcut: chunk kt - znk of its dY at the centre tap (pixel (ho, wo) / 2), its own weight rows [.][C0]
        const uint32_t cf = (uint32_t)((kt - znk) * KE + lslot * VEC);
#pragma unroll
        for (int i = 0; i < RA; ++i) {
          const bool act = (i * 32 + wave * 8) < BM;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(
              rswf, (__attribute__((address_space(3))) void*)(act ? sa + (i * 32 + wave * 8) * 128 : smem + DUMP + wave * 1024), 16,
              act ? ((uint32_t)(cout0 + r0 + 32 * i) * (uint32_t)a.C0 + cf) * SZ : STP_OOB, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) {
          const int hv = hb[i] + a.pad, wv = wb[i] + a.pad;        // = (ho, wo): both even in this class
          const bool ok = (unsigned)hv < (unsigned)a.Hv && (unsigned)wv < (unsigned)a.Wv;
          const uint32_t off = nof0[i] + (uint32_t)((hv >> 1) * a.Ws0 + (wv >> 1)) * (uint32_t)a.C0 + cf;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsf, (__attribute__((address_space(3))) void*)(sb + (i * 32 + wave * 8) * 128), 16,
                                                   ok ? off * SZ : STP_OOB, 0, 0, 0);
        }
        return;
      }
      const int tix = (int)fdiv((uint32_t)kt, a.divCpt), ch = kt - tix * a.zcpt;
      const int ta = znkw == 2 ? tix >> 1 : tix, tb = znkw == 2 ? tix & 1 : 0;
      k0 = (uint32_t)(((zkh0 + 2 * ta) * a.KW + zkw0 + 2 * tb) * a.Ctot + ch * KE);
    }
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      const bool act = (i * 32 + wave * 8) < BM;  // wave-uniform
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          rsw, (__attribute__((address_space(3))) void*)(act ? sa + (i * 32 + wave * 8) * 128 : smem + DUMP + wave * 1024), 16,
          (act && (UNI || (int)(k0 + lslot * VEC) < a.K)) ? wof[i] + k0 * SZ : STP_OOB, 0, 0, 0);
    }
    if constexpr (!UNI) {
      // several taps per K-tile: tap = k0/Ctot + (this lane's constant tap offset); single source
      const uint32_t pos = fdiv(k0, a.divC) + lane_dpos;
      const int kh = (int)fdiv(pos, a.divKW);
      const int kw = (int)pos - kh * a.KW;
      const bool kok = pos < (uint32_t)(a.KH * a.KW);
      const int sh = a.mode ? 1 : 0;
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        const int hv = hb[i] + kh, wv = wb[i] + kw;
        bool ok = kok && (unsigned)hv < (unsigned)a.Hv && (unsigned)wv < (unsigned)a.Wv;
        if (a.mode == STP_SRC_ZEROINS2X) ok = ok && (((hv | wv) & 1) == 0);
        const uint32_t off = nof0[i] + (uint32_t)((hv >> sh) * a.Ws0 + (wv >> sh)) * (uint32_t)a.C0 + lane_ci;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, (__attribute__((address_space(3))) void*)(sb + (i * 32 + wave * 8) * 128), 16,
                                                 ok ? off * SZ : STP_OOB, 0, 0, 0);
      }
      return;
    }
    const uint32_t pos = fdiv(k0, a.divC);
    const int cb = (int)(k0 - pos * (uint32_t)a.Ctot);
    const int kh = (int)fdiv(pos, a.divKW);
    const int kw = (int)pos - kh * a.KW;
    if (cb < a.C0) {
      const int sh = a.mode ? 1 : 0;
      const uint32_t ci = (uint32_t)(cb + lslot * VEC);
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        const int hv = hb[i] + kh, wv = wb[i] + kw;
        bool ok = (unsigned)hv < (unsigned)a.Hv && (unsigned)wv < (unsigned)a.Wv;
        if (a.mode == STP_SRC_ZEROINS2X) ok = ok && (((hv | wv) & 1) == 0);
        const uint32_t off = nof0[i] + (uint32_t)((hv >> sh) * a.Ws0 + (wv >> sh)) * (uint32_t)a.C0 + ci;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, (__attribute__((address_space(3))) void*)(sb + (i * 32 + wave * 8) * 128), 16,
                                                 ok ? off * SZ : STP_OOB, 0, 0, 0);
      }
    } else {
      const uint32_t ci = (uint32_t)(cb - a.C0 + lslot * VEC);
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        const int hv = hb[i] + kh, wv = wb[i] + kw;
        const bool ok = (unsigned)hv < (unsigned)a.Hv && (unsigned)wv < (unsigned)a.Wv;
        const uint32_t off = nof1[i] + (uint32_t)(hv * a.Wv + wv) * (uint32_t)a.C1 + ci;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (__attribute__((address_space(3))) void*)(sb + (i * 32 + wave * 8) * 128), 16,
                                                 ok ? off * SZ : STP_OOB, 0, 0, 0);
      }
    }
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int lr = lane & 15, lg = lane >> 4;

  const int nk = up ? 4 * a.ucpt0 + 9 * a.ucpt1 : zp ? znk + zfn : (a.K + KE - 1) / KE;  // !UNI: the tail taps of the last tile are out of range -> zeros on both operands
  // epilogue operands (residual or the BatchNormalization-backward x; never both) fetched now: the loads are OLDER than every
  // tile load, so the counted vmcnt waits of the K loop also cover them, and their latency hides under the whole loop
  constexpr bool PRE = SZ == 2;
  // row-major epilogue (conv_common.h: epilogue_rm_lin) where the channel counts allow: it fetches its own operands, 16 bytes a lane
  bool rm = false;
  if constexpr (SZ == 2) {
    static_assert((size_t)BN * (BM * 4 + 16) <= (size_t)STAGES * STAGE + 4096, "the staged fp32 tile fits the operand ring");
    rm = epilogue_rm_ok(a);
  }
  u32x2 pre[PRE ? TM : 1][TN];
  if constexpr (PRE) {
    const T* ps = a.residual ? reinterpret_cast<const T*>(a.residual) : reinterpret_cast<const T*>(a.bnb.x);
    if (ps && !rm) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int co = cout0 + wm * (BM / WM) + i * 16 + lg * 4;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          int pm = pix0 + wn * (BN / WN) + j * 16 + lr;
          if (zp) pm = zperm_pixel(a, pm);                                      // (P is a multiple of the tile in this mode)
          const bool ok = co + 3 < a.Cout && pm < a.P;                       // others are never read; clamp keeps the load in bounds
          pre[i][j] = *reinterpret_cast<const u32x2*>(ps + (ok ? (size_t)pm * a.Cout + co : (size_t)0));
        }
      }
    }
  }
  // parity-class launches (ZP) that accumulate into dst0: its current contents are fetched here as well (see epilogue_cf)
  constexpr bool PREA = PRE && ZP && !UPC;
  u32x2 prea[PREA ? TM : 1][TN];
  const bool use_prea = PREA && zp && a.acc0 && a.Cd0 == a.Cout && !rm;
  if constexpr (PREA) {
    if (use_prea) {
      const T* pd = reinterpret_cast<const T*>(a.dst0);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int co = cout0 + wm * (BM / WM) + i * 16 + lg * 4;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int pm = zperm_pixel(a, pix0 + wn * (BN / WN) + j * 16 + lr);
          const bool ok = co + 3 < a.Cout && pm < a.P;
          prea[i][j] = *reinterpret_cast<const u32x2*>(pd + (ok ? (size_t)pm * a.Cout + co : (size_t)0));
        }
      }
    }
  }
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s)
    if (s < nk) issue_tile(s, s);
  int buf = 0, nbuf = STAGES - 1;
  for (int kt = 0; kt < nk; ++kt) {
    // tile kt must have landed; up to STAGES-2 younger tiles stay in flight across the barrier
    const int ahead = nk - 1 - kt;
    if (STAGES >= 3 && ahead >= STAGES - 2) wait_vmcnt<(STAGES >= 3 ? (STAGES - 2) : 0) * L>();
    else if (STAGES >= 4 && ahead == 1) wait_vmcnt<L>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();  // tile kt visible to every wave; the buffer of tile kt-1 is free
#if !defined(STP_EXP) || (STP_EXP != 2 && STP_EXP != 3)  // (what-if builds of scratch/exp_build.sh: 1 = no MFMA work, 2 = no loads in the loop, 3 = 2 + no LDS reads)
    if (kt + STAGES - 1 < nk) issue_tile(kt + STAGES - 1, nbuf);
#endif
#if !defined(STP_EXP) || STP_EXP != 1
    compute_tile<T, BM, BN, WM, WN>(smem + buf * STAGE, wm, wn, lr, lg, acc);
#endif
    buf = (buf + 1 == STAGES) ? 0 : buf + 1;
    nbuf = (nbuf + 1 == STAGES) ? 0 : nbuf + 1;
  }
  if constexpr (SZ == 2) {
    if (rm) {
      epilogue_rm_lin<BM, BN, WM, WN>(a, acc, smem, cout0, tile_n, wm, wn, lr, lg, tid, [&a, pix0, zp](int p) {
        const int pl = pix0 + p;
        return pl >= a.P ? -1 : (zp ? zperm_pixel(a, pl) : pl);
      });
      return;
    }
  }
  if (zp) {
    const int pb = pix0 + wn * (BN / WN) + lr;
    epilogue_px<T, TM, TN, BM, WN, 256, PRE>(a, cout0, wm * (BM / WM), wn, lr, lg, acc, smem, tile_n, pre,
                                              [pb, &a](int j) { return zperm_pixel(a, pb + j * 16); }, use_prea ? prea : nullptr);
  } else {
    epilogue<T, BM, BN, WM, WN, PRE>(a, cout0, pix0, wm, wn, lr, lg, acc, smem, tile_n, pre);
  }
#endif
}

// ================================================================================================
// general register-staged kernel
// ================================================================================================
template <typename T, int BM, int BN, int WM, int WN, bool C4>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvArgs a) {
  static_assert(WM * WN == 4, "4 waves");
  constexpr int VEC = Elem<T>::VEC;
  constexpr int KE = 128 / (int)sizeof(T);  // K elements per LDS row / K-step
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
  constexpr int RA = (BM + 31) / 32, RB = (BN + 31) / 32;
  constexpr int STAGE = (BM + BN) * 128;

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int r0 = tid >> 3, slot = tid & 7;

  const int bid = xcd_remap(blockIdx.x, a.ntile_m * a.ntile_n);
  const int tile_n = (int)fdiv((uint32_t)bid, a.divNtm);
  const int tile_m = bid - tile_n * a.ntile_m;  // cout tiles innermost: they share the same pixels
  const int cout0 = tile_m * BM;
  const int pix0 = tile_n * BN;

  // ---- per-thread gather metadata for its pixel rows -------------------------------------
  int hb[RB], wb[RB], nb[RB];
#pragma unroll
  for (int i = 0; i < RB; ++i) {
    const int pm = pix0 + r0 + 32 * i;
    if (pm < a.P && (r0 + 32 * i) < BN) {
      const int n = (int)fdiv((uint32_t)pm, a.divHoWo);
      const int rem = pm - n * a.HoWo;
      const int ho = (int)fdiv((uint32_t)rem, a.divWo);
      const int wo = rem - ho * a.Wo;
      nb[i] = n;
      hb[i] = ho * a.stride - a.pad;
      wb[i] = wo * a.stride - a.pad;
    } else {
      nb[i] = 0;
      hb[i] = -(1 << 24);
      wb[i] = -(1 << 24);
    }
  }

  u32x4 ra[RA], rb[RB];

  auto load_tile = [&](int kt) {
    const int k = kt * KE + slot * VEC;
    const bool kok = k < a.K;
    // weights: plain rows
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      const int row = cout0 + r0 + 32 * i;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (kok && row < a.wrows && (r0 + 32 * i) < BM)
        v = *reinterpret_cast<const u32x4*>(a.weight + ((size_t)row * a.K + k) * sizeof(T));
      ra[i] = v;
    }
    // activations: im2col gather
    if constexpr (C4) {
      // stem: 4 (padded) channels per pixel; one 16-byte vector = two horizontally adjacent taps
      const uint32_t pos = (uint32_t)k >> 2;
      const int kh = (int)fdiv(pos, a.divKW);
      const int kw = (int)pos - kh * a.KW;
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        const int hv = hb[i] + kh, wv = wb[i] + kw;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (kok && (unsigned)hv < (unsigned)a.Hv) {
          const char* rowp = a.src0 + ((size_t)(nb[i] * a.Hs0 + hv) * a.Ws0) * (4 * sizeof(T));
          if ((unsigned)wv < (unsigned)a.Wv) {
            u32x2 t = *reinterpret_cast<const u32x2*>(rowp + (size_t)wv * (4 * sizeof(T)));
            v.x = t.x; v.y = t.y;
          }
          if ((unsigned)(wv + 1) < (unsigned)a.Wv) {
            u32x2 t = *reinterpret_cast<const u32x2*>(rowp + (size_t)(wv + 1) * (4 * sizeof(T)));
            v.z = t.x; v.w = t.y;
          }
        }
        rb[i] = v;
      }
    } else {
      const uint32_t pos = fdiv((uint32_t)k, a.divC);
      int ci = k - (int)pos * a.Ctot;
      const int kh = (int)fdiv(pos, a.divKW);
      const int kw = (int)pos - kh * a.KW;
      const bool first = ci < a.C0;
      const char* base = first ? a.src0 : a.src1;
      const int cs = first ? a.C0 : a.C1;
      const int mode = first ? a.mode : 0;
      const int Hs = first ? a.Hs0 : a.Hv, Ws = first ? a.Ws0 : a.Wv;
      if (!first) ci -= a.C0;
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        const int hv = hb[i] + kh, wv = wb[i] + kw;
        bool ok = kok && (unsigned)hv < (unsigned)a.Hv && (unsigned)wv < (unsigned)a.Wv;
        if (mode == STP_SRC_ZEROINS2X) ok = ok && (((hv | wv) & 1) == 0);
        const int hs = mode ? (hv >> 1) : hv, ws = mode ? (wv >> 1) : wv;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (ok) v = *reinterpret_cast<const u32x4*>(base + (((size_t)(nb[i] * Hs + hs) * Ws + ws) * cs + ci) * sizeof(T));
        rb[i] = v;
      }
    }
  };

  auto store_tile = [&](int buf) {
    char* sa = smem + buf * STAGE;
    char* sb = sa + BM * 128;
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      const int row = r0 + 32 * i;
      if (row < BM) *reinterpret_cast<u32x4*>(sa + row * 128 + ((slot ^ (row & 7)) << 4)) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const int row = r0 + 32 * i;
      if (row < BN) *reinterpret_cast<u32x4*>(sb + row * 128 + ((slot ^ (row & 7)) << 4)) = rb[i];
    }
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int lr = lane & 15, lg = lane >> 4;
  const int nk = (a.K + KE - 1) / KE;
  load_tile(0);
  store_tile(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) load_tile(kt + 1);
    compute_tile<T, BM, BN, WM, WN>(smem + cur * STAGE, wm, wn, lr, lg, acc);
    if (kt + 1 < nk) store_tile(cur ^ 1);
    __syncthreads();
  }
  epilogue<T, BM, BN, WM, WN>(a, cout0, pix0, wm, wn, lr, lg, acc, smem, tile_n);
}

// ================================================================================================
// host side
// ================================================================================================
template <typename K>
static int launch_kernel(K kern, ConvArgs& a, size_t lds, bool& attr_set, hipStream_t s) {
  if (lds > 64 * 1024 && !attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return STP_E_LAUNCH;
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(a.ntile_m * a.ntile_n), dim3(256), lds, s, a);
  STP_LAUNCH_CHECK();
  return STP_OK;
}

template <typename T, int BM, int BN, int WM, int WN, bool C4>
static int launch_gen(ConvArgs& a, hipStream_t s) {
  static bool attr_set = false;  // one per template instantiation
  a.ntile_m = ceil_div(a.Cout, BM);
  a.ntile_n = ceil_div(a.P, BN);
  a.divNtm = make_fastdiv((uint32_t)a.ntile_m);
  return launch_kernel(conv_igemm_kernel<T, BM, BN, WM, WN, C4>, a, (size_t)2 * (BM + BN) * 128, attr_set, s);
}

template <typename T, int BM, int BN, int WM, int WN, int STAGES, bool UNI = true>
static int launch_ut(ConvArgs& a, hipStream_t s) {
  static bool attr_set = false;
  a.ntile_m = ceil_div(a.Cout, BM);
  a.ntile_n = ceil_div(a.P, BN);
  a.divNtm = make_fastdiv((uint32_t)a.ntile_m);
  constexpr size_t LDS = (size_t)STAGES * (BM + BN) * 128 + 4096;
  if constexpr (UNI && (STAGES == 2 || (STAGES == 4 && BM == 64 && BN == 64))) {      // (the tiles auto_tile() picks: stp_conv2d sets upc for these only)
    static bool attr_set_upc = false;
    if (a.upc) return launch_kernel(conv_igemm_ut_kernel<T, BM, BN, WM, WN, STAGES, UNI, true, true>, a, LDS, attr_set_upc, s);
  }
  if (a.upc) return STP_E_BADARG;
  if constexpr (UNI) {
    static bool attr_set_zp = false;
    if (a.zperm) return launch_kernel(conv_igemm_ut_kernel<T, BM, BN, WM, WN, STAGES, UNI, false, true>, a, LDS, attr_set_zp, s);
  }
  if (a.zperm) return STP_E_BADARG;
  return launch_kernel(conv_igemm_ut_kernel<T, BM, BN, WM, WN, STAGES, UNI>, a, LDS, attr_set, s);
}

// Tile ids (output channels x pixels per workgroup):
//   general kernel   1: 128x128   2: 64x256   3: 32x256   4: 16x256   5: 64x64   6: 128x64
//   uniform-tap DMA  32*STAGES + {1, 2, 5, 6} (STAGES = 2, 3; 4 for the two small tiles)
// ut_ok: 0 = general kernel only, 1 = uniform-tap DMA eligible, 2 = per-lane-tap DMA eligible (small Ctot)
template <typename T, bool C4>
static int launch_tile(ConvArgs& a, int tile, int ut_ok, hipStream_t s) {
  if constexpr (C4) {
    switch (tile) {
      case 2: return launch_gen<T, 64, 256, 1, 4, true>(a, s);
      case 5: return launch_gen<T, 64, 64, 2, 2, true>(a, s);
      default: return STP_E_BADARG;
    }
  } else {
    if (tile >= 256 ? ut_ok != 2 : (tile >= 64 && ut_ok != 1)) return STP_E_BADARG;
    switch (tile) {
      case 1: return launch_gen<T, 128, 128, 2, 2, false>(a, s);
      case 2: return launch_gen<T, 64, 256, 1, 4, false>(a, s);
      case 3: return launch_gen<T, 32, 256, 1, 4, false>(a, s);
      case 4: return launch_gen<T, 16, 256, 1, 4, false>(a, s);
      case 5: return launch_gen<T, 64, 64, 2, 2, false>(a, s);
      case 6: return launch_gen<T, 128, 64, 4, 1, false>(a, s);
      case 7: return launch_gen<T, 64, 128, 2, 2, false>(a, s);
      case 64 + 1: return launch_ut<T, 128, 128, 2, 2, 2>(a, s);
      case 64 + 2: return launch_ut<T, 64, 256, 1, 4, 2>(a, s);
      case 64 + 5: return launch_ut<T, 64, 64, 2, 2, 2>(a, s);
      case 64 + 6: return launch_ut<T, 128, 64, 4, 1, 2>(a, s);
      case 64 + 7: return launch_ut<T, 64, 128, 2, 2, 2>(a, s);
      case 96 + 7: return launch_ut<T, 64, 128, 2, 2, 3>(a, s);
      case 96 + 1: return launch_ut<T, 128, 128, 2, 2, 3>(a, s);
      case 96 + 2: return launch_ut<T, 64, 256, 1, 4, 3>(a, s);
      case 96 + 5: return launch_ut<T, 64, 64, 2, 2, 3>(a, s);
      case 96 + 6: return launch_ut<T, 128, 64, 4, 1, 3>(a, s);
      case 128 + 5: return launch_ut<T, 64, 64, 2, 2, 4>(a, s);
      case 128 + 6: return launch_ut<T, 128, 64, 4, 1, 4>(a, s);
      case 64 + 3: return launch_ut<T, 32, 256, 1, 4, 2>(a, s);
      case 64 + 4: return launch_ut<T, 16, 256, 1, 4, 2>(a, s);
      // per-lane-tap variants for 8/16/32-channel inputs (ut_ok == 2)
      case 256 + 2: return launch_ut<T, 64, 256, 1, 4, 2, false>(a, s);
      case 256 + 3: return launch_ut<T, 32, 256, 1, 4, 2, false>(a, s);
      case 256 + 4: return launch_ut<T, 16, 256, 1, 4, 2, false>(a, s);
      case 256 + 1: return launch_ut<T, 128, 128, 2, 2, 2, false>(a, s);
      case 256 + 5: return launch_ut<T, 64, 64, 2, 2, 2, false>(a, s);
      case 256 + 6: return launch_ut<T, 128, 64, 4, 1, 2, false>(a, s);
      default: return STP_E_BADARG;
    }
  }
}

// Heuristics from scratch/conv_bench.py on MI355X (bf16): the kernel is latency- rather than MFMA-bound, so
// 2-stage rings that let 2+ workgroups share a CU beat deeper rings at 1 workgroup/CU, and tiles are
// shrunk until the grid covers the 256 CUs at least ~1.5x.
static int auto_tile(const ConvArgs& a, int ut_ok) {
  const int co = a.Cout;
  if (co <= 16) return ut_ok == 2 ? 256 + 4 : ut_ok == 1 ? 64 + 4 : 4;
  if (co <= 32) return ut_ok == 2 ? 256 + 3 : ut_ok == 1 ? 64 + 3 : 3;
  if (co <= 64) {
    if (ut_ok == 2) return 256 + 2;
    if (ut_ok == 1) return ceil_div(a.P, 128) >= 384 ? 64 + 7 : 64 + 5;
    return ((int64_t)a.P >= 256 * 256) ? 2 : 5;
  }
  const int64_t big = (int64_t)ceil_div(co, 128) * ceil_div(a.P, 128);
  const int64_t mid = (int64_t)ceil_div(co, 128) * ceil_div(a.P, 64);
  if (ut_ok == 1) {
    // the last 128-channel tile at most half full (Cout = 192: a data gradient into two concatenated sources): 64-channel tiles
    // waste nothing (scratch/tile_sweep.py: decoder_stage2_conv1 dgrad 120 -> 104 us)
    const int tail = co % 128;
    if (tail > 0 && tail <= 64 && ceil_div(a.P, 128) >= 384) return 64 + 7;
    // one or two K steps (1x1 bottleneck convolutions): nothing to pipeline, the narrower pixel tile's extra workgroups hide
    // more of the per-workgroup latency (scratch/conv1x1_bench.py: 64->256 @ 4x256^2 65 -> 57 us)
    // ... with the BatchNormalization-backward epilogue (three operand tensors per output tile) the 64-channel x 128-pixel tile is the
    // faster one (scratch/r05/dgrad1x1_bench.py: 128 -> 512 @ 8 x 96^2 73.6 -> 57.3 us, 64 -> 256 @ 8 x 192^2 137.5 -> 113.5 us)
    if (a.K <= 128 && mid >= 384) return a.bnb.x ? 64 + 7 : 64 + 6;
    // ... and up to four K steps under that epilogue (the class head's tap channels, 192 -> 512 @ 8 x 96^2: 73.1 -> 64.3 us;
    // STP_DGRAD1X1_BNB_K=<largest K> for A/Bs)
    static const int k71 = getenv("STP_DGRAD1X1_BNB_K") ? atoi(getenv("STP_DGRAD1X1_BNB_K")) : 256;
    if (a.bnb.x && a.KH == 1 && a.KW == 1 && a.K <= k71 && mid >= 384) return 64 + 7;
    // deep-K 1x1 GEMMs (PSPNet's psp_final: 2560 -> 512 and its data gradient at 8 x 96 x 96, 40 / 8 K steps of 64): STP_1X1_DEEPK_TILE
    // = the tile id for 1x1 launches with K >= 512 (experiments: 97 / 129 = the 128 x 128 tile with a 3 / 4-stage ring)
    static const int deepk = getenv("STP_1X1_DEEPK_TILE") ? atoi(getenv("STP_1X1_DEEPK_TILE")) : 0;
    if (deepk > 0 && a.KH == 1 && a.KW == 1 && a.K >= 512 && big >= 384) return deepk;
    // ... and for the small-M ones (the 1x1 layers of ResNet50's stages 3 - 4 at 4 x 64 x 64 / 4 x 32 x 32: 4 - 16 K pixels, K = 256 - 2048,
    // 25 - 40 us at 3 - 7 x their floors): STP_1X1_SMALLM_TILE = the tile id when fewer than 384 128 x 128 tiles exist (experiments)
    static const int smallm = getenv("STP_1X1_SMALLM_TILE") ? atoi(getenv("STP_1X1_SMALLM_TILE")) : 0;
    if (smallm > 0 && a.KH == 1 && a.KW == 1 && a.K >= 256 && big < 384) return smallm;
    if (big >= 384) return 64 + 1;
    if (mid >= 384) return 64 + 6;
    return 128 + 5;
  }
  if (ut_ok == 2) {
    if (big >= 384) return 256 + 1;
    if (mid >= 384) return 256 + 6;
    return 256 + 5;
  }
  if (big >= 384) return 1;
  if (mid >= 384) return 6;
  return 5;
}

static int tile_pixels(int tile) {
  if (tile == 512) return 256;
  const int t = tile >= 256 ? tile - 256 : tile % 32;
  return (t == 2 || t == 3 || t == 4) ? 256 : ((t == 1 || t == 7) ? 128 : 64);
}

extern "C" int stp_conv2d_sc_eligible(const stp_conv_params* p);
extern "C" int stp_conv2d_sc(const stp_conv_params* p, void* stream);
extern "C" int stp_conv2d_sc_stats_tiles(const stp_conv_params* p);
extern "C" int stp_conv2d_scw_eligible(const stp_conv_params* p);
extern "C" int stp_conv2d_scw(const stp_conv_params* p, void* stream);
extern "C" int stp_conv2d_scw_stats_tiles(const stp_conv_params* p);
extern "C" int stp_conv2d_scn_eligible(const stp_conv_params* p);
extern "C" int stp_conv2d_scn(const stp_conv_params* p, void* stream);
extern "C" int stp_conv2d_scn_stats_tiles(const stp_conv_params* p);
extern "C" int stp_conv2d_s64_eligible(const stp_conv_params* p);
extern "C" int stp_conv2d_s64(const stp_conv_params* p, void* stream);
extern "C" int stp_conv2d_s64_stats_tiles(const stp_conv_params* p);
extern "C" int stp_conv2d_stem_eligible(const stp_conv_params* p);
extern "C" int stp_conv2d_stem(const stp_conv_params* p, void* stream);
extern "C" int stp_conv2d_stem_stats_tiles(const stp_conv_params* p);
extern "C" int stp_conv2d_halo_variant(const stp_conv_params* p);
extern "C" int stp_conv2d_halo_tiles(const stp_conv_params* p, int variant);
extern "C" int stp_conv2d_halo(const stp_conv_params* p, int variant, void* stream);
#define STP_TILE_HALO 1024  // + variant: the halo-resident 3x3 kernel of conv_halo.hip
// automatic use of the halo kernel (an explicit tile id always works): STP_HALO=0 switches it off for A/B runs
static bool halo_auto_enabled() {
  static const bool on = !(getenv("STP_HALO") && atoi(getenv("STP_HALO")) == 0);
  return on;
}
static int halo_variant_for(const stp_conv_params* p) {
  if (!p || (p->tile == 0 && !halo_auto_enabled())) return -1;
  return stp_conv2d_halo_variant(p);
}
#define STP_TILE_SC 512  // the small-channel halo-tile kernel of conv_sc.hip
#define STP_TILE_STEM 768  // the 7x7 / stride-2 stem kernel of conv_sc.hip
#define STP_TILE_SCW 640  // the wide-output (two-destination, 2x2-summed) data-gradient kernel of conv_sc.hip
#define STP_TILE_SCN 704  // the narrow-output (upsample + concat -> 32 channels) forward kernel of conv_sc.hip
#define STP_TILE_S64 736  // the 64 -> 64 channel weights-in-registers kernel of conv_sc.hip (opt-in: STP_S64=1, or this tile id)
#define STP_TILE_PW 800   // the pointwise (1x1 / stride 1) pixel-streaming kernel of conv_pw.hip (round 6)
extern "C" int stp_conv2d_pw_eligible(const stp_conv_params* p);
extern "C" int stp_conv2d_pw_cols(const stp_conv_params* p);
extern "C" int stp_conv2d_pw(const stp_conv_params* p, void* stream);
static bool s64_auto() {
  static const bool on = getenv("STP_S64") && atoi(getenv("STP_S64")) == 1;
  return on;
}

// Which tile configuration stp_conv2d would launch (profiling / roofline bookkeeping).
extern "C" int stp_conv2d_tile_for(const stp_conv_params* p) {
  if (p && (p->tile == 0 || p->tile == STP_TILE_SC) && stp_conv2d_sc_eligible(p)) return STP_TILE_SC;
  if (p && (p->tile == 0 || p->tile == STP_TILE_SCW) && stp_conv2d_scw_eligible(p)) return STP_TILE_SCW;
  if (p && (p->tile == 0 || p->tile == STP_TILE_SCN) && stp_conv2d_scn_eligible(p)) return STP_TILE_SCN;
  if (p && ((p->tile == 0 && s64_auto()) || p->tile == STP_TILE_S64) && stp_conv2d_s64_eligible(p)) return STP_TILE_S64;
  if (p && (p->tile == 0 || p->tile == STP_TILE_STEM) && stp_conv2d_stem_eligible(p)) return STP_TILE_STEM;
  if (p && (p->tile == 0 || p->tile == STP_TILE_PW) && stp_conv2d_pw_eligible(p)) return STP_TILE_PW;
  if (p && p->tile == STP_TILE_PW) return STP_E_BADARG;
  {
    const int hv = halo_variant_for(p);
    if (hv >= 0) return STP_TILE_HALO + hv;
    if (p && p->tile >= STP_TILE_HALO) return STP_E_BADARG;
  }
  ConvArgs a;
  bool c4;
  int ut;
  const int rc = fill_args(p, a, &c4, &ut);
  if (rc != STP_OK) return rc;
  int tile = p->tile ? p->tile : auto_tile(a, ut);
  if (c4 && tile != 2 && tile != 5) tile = 2;
  return tile;
}

extern "C" size_t stp_conv2d_stats_floats(const stp_conv_params* p) {
  const int tile = stp_conv2d_tile_for(p);
  if (tile < 0) return 0;
  if (tile == STP_TILE_SC) return (size_t)stp_conv2d_sc_stats_tiles(p) * 2 * p->Cout;
  if (tile == STP_TILE_SCW) return (size_t)stp_conv2d_scw_stats_tiles(p) * 2 * p->Cd0;
  if (tile == STP_TILE_SCN) return (size_t)stp_conv2d_scn_stats_tiles(p) * 2 * p->Cout;
  if (tile == STP_TILE_S64) return (size_t)stp_conv2d_s64_stats_tiles(p) * 2 * p->Cout;
  if (tile == STP_TILE_STEM) return (size_t)stp_conv2d_stem_stats_tiles(p) * 2 * p->Cout;
  if (tile == STP_TILE_PW) return (size_t)stp_conv2d_pw_cols(p) * 2 * p->Cout;
  if (tile >= STP_TILE_HALO) return (size_t)stp_conv2d_halo_tiles(p, tile - STP_TILE_HALO) * 2 * (p->dst_sum2x2 ? p->Cd0 : p->Cout);
  return (size_t)ceil_div((int64_t)p->N * p->Ho * p->Wo, tile_pixels(tile)) * 2 * p->Cout;
}

// Group-level pre-reduction of the statistic columns (conv_common.h: stats_group_finish): the kernels whose epilogue has it are the halo
// kernel and the buffer-DMA kernel with its row-major epilogue (16-bit storage).  STP_STATS_GROUP=0 switches the feature off (A/B).
extern "C" int stp_conv2d_stats_group_for(const stp_conv_params* p) {
  // OPT-IN (STP_STATS_GROUP=1).  Measured on the headline step, same box (profiles/r05e_stats_group_ab.txt): 40 of the 50 finalize launches
  // disappear (BatchNormalization launches -118 us in the eager table) but every ROUND of convolution workgroups pays ~2.2 us for the
  // protocol - a drained store queue before the ticket, the ticket's round trip before the workgroup may end (halo 16 x 16 x 128, one
  // round: +2.2 us; 16 x 16 x 64, two rounds: +4.3; 8 x 16 x 64 at 128 x 128, four rounds: +12) = +121 us; in the captured graph, where a
  // finalize launch costs ~3 us rather than the 5 of the eager table, the step is 0.02 (G <= 2) to 0.10 ms (all groups) SLOWER.
  static const bool on = getenv("STP_STATS_GROUP") && atoi(getenv("STP_STATS_GROUP")) == 1;
  if (!on || !p || p->stats_slots || p->s2d_dgrad) return 0;
  const int tile = stp_conv2d_tile_for(p);
  if (tile < 0) return 0;
  // STP_STATS_GROUP_MAXG: largest group taken (experiments: the protocol costs every workgroup a drained store queue + one ticket,
  // which a kernel of several ROUNDS of workgroups pays once per round)
  static const int maxg = getenv("STP_STATS_GROUP_MAXG") ? atoi(getenv("STP_STATS_GROUP_MAXG")) : 16;
  if (tile >= STP_TILE_HALO) {
    const int G = stats_group_size(stp_conv2d_halo_tiles(p, tile - STP_TILE_HALO));
    return G <= maxg ? G : 0;
  }
  if (tile >= 64 && tile < 512 && p->dtype == STP_H16) {
    ConvArgs a;
    bool c4;
    int ut;
    stp_conv_params q = *p;
    q.stats_group = 0;
    if (fill_args(&q, a, &c4, &ut) != STP_OK || c4 || !epilogue_rm_ok(a)) return 0;
    const int G = stats_group_size(ceil_div(a.P, tile_pixels(tile)));
    static const bool igemm_on = !(getenv("STP_STATS_GROUP_IGEMM") && atoi(getenv("STP_STATS_GROUP_IGEMM")) == 0);
    return (igemm_on && G <= maxg) ? G : 0;
  }
  return 0;
}
extern "C" size_t stp_conv2d_stats_group_counters(const stp_conv_params* p, int32_t G) {
  if (!p || G < 2) return 0;
  const int Cs = p->dst_sum2x2 ? p->Cd0 : p->Cout;
  const size_t cols = stp_conv2d_stats_floats(p) / (2 * (size_t)(Cs > 0 ? Cs : 1));
  return (size_t)ceil_div(Cs, 16) * ceil_div((int64_t)cols, G);         // (channel tiles are 16 channels or wider)
}

static bool zperm_applies(const ConvArgs& a, int tile, int ut) {
  static const bool zperm_on = !(getenv("STP_ZPERM") && atoi(getenv("STP_ZPERM")) == 0);
  const bool uni_tile = tile >= 64 && tile < 256;
  return zperm_on && a.mode == STP_SRC_ZEROINS2X && ut == 1 && uni_tile && a.stride == 1 && a.KH <= 3 && a.KW <= 3 && !(a.Ho & 1) && !(a.Wo & 1) &&
         a.C1 == 0 && ((a.P / 4) % tile_pixels(tile)) == 0;
}
static bool fold_geometry_ok(const ConvArgs& a) { return a.KH == 3 && a.KW == 3 && a.pad == 1 && a.Hv == 2 * a.Hs0 - 1 && a.Wv == 2 * a.Ws0 - 1; }

extern "C" int stp_conv2d_fold_ok(const stp_conv_params* p) {
  static const bool on = !(getenv("STP_FOLD_SHORTCUT") && atoi(getenv("STP_FOLD_SHORTCUT")) == 0);
  if (!on || !p || p->tile != 0 || stp_conv2d_sc_eligible(p) || stp_conv2d_scw_eligible(p) || stp_conv2d_scn_eligible(p) || (s64_auto() && stp_conv2d_s64_eligible(p)) || stp_conv2d_stem_eligible(p) || stp_conv2d_pw_eligible(p) || halo_variant_for(p) >= 0) return 0;
  ConvArgs a;
  bool c4;
  int ut;
  if (fill_args(p, a, &c4, &ut) != STP_OK || c4) return 0;
  return zperm_applies(a, auto_tile(a, ut), ut) && fold_geometry_ok(a) ? 1 : 0;
}

extern "C" int stp_conv2d(const stp_conv_params* p, void* stream) {
  if (p && p->stats_group > 1 && p->stats_group != stp_conv2d_stats_group_for(p)) return STP_E_BADARG;   // (only the kernels that have the epilogue)
  if (p && (p->tile == 0 || p->tile == STP_TILE_SC)) {
    if (stp_conv2d_sc_eligible(p)) return stp_conv2d_sc(p, stream);
    if (p->tile == STP_TILE_SC) return STP_E_BADARG;
  }
  if (p && (p->tile == 0 || p->tile == STP_TILE_SCW)) {
    if (stp_conv2d_scw_eligible(p)) return stp_conv2d_scw(p, stream);
    if (p->tile == STP_TILE_SCW) return STP_E_BADARG;
  }
  if (p && (p->tile == 0 || p->tile == STP_TILE_SCN)) {
    if (stp_conv2d_scn_eligible(p)) return stp_conv2d_scn(p, stream);
    if (p->tile == STP_TILE_SCN) return STP_E_BADARG;
  }
  if (p && ((p->tile == 0 && s64_auto()) || p->tile == STP_TILE_S64)) {
    if (stp_conv2d_s64_eligible(p)) return stp_conv2d_s64(p, stream);
    if (p->tile == STP_TILE_S64) return STP_E_BADARG;
  }
  if (p && (p->tile == 0 || p->tile == STP_TILE_STEM)) {
    if (stp_conv2d_stem_eligible(p)) return stp_conv2d_stem(p, stream);
    if (p->tile == STP_TILE_STEM) return STP_E_BADARG;
  }
  if (p && (p->tile == 0 || p->tile == STP_TILE_PW)) {
    if (stp_conv2d_pw_eligible(p)) return stp_conv2d_pw(p, stream);
    if (p->tile == STP_TILE_PW) return STP_E_BADARG;
  }
  {
    const int hv = halo_variant_for(p);
    if (hv >= 0) return stp_conv2d_halo(p, hv, stream);
    if (p && p->tile >= STP_TILE_HALO) return STP_E_BADARG;
    // forms that ONLY the halo kernel implements: the space-to-depth data gradient (its KH = KW = 2 / Cout = 4 x C0 geometry means
    // something else to the per-tap kernel) and a folded shortcut on a stride-1 launch - a plan / dispatcher mismatch (STP_HALO=0
    // against a plan built without it) must fail, never run the generic kernel on these parameters
    if (p && (p->s2d_dgrad || (p->fold_src && p->src0_mode != STP_SRC_ZEROINS2X))) return STP_E_BADARG;
  }
  ConvArgs a;
  bool c4;
  int ut;
  const int rc = fill_args(p, a, &c4, &ut);
  if (rc != STP_OK) return rc;
  if (p->dst_sum2x2 || p->src_bn_mean) return STP_E_BADARG;  // folded upsample gradient / fused producer BN: small-channel kernel only
  // (checked here, not in fill_args: the sizing queries run before stats_partial is allocated)
  if (a.bnb.x && (!a.stats || !a.bnb.mean || !a.bnb.rstd || p->relu || p->residual)) return STP_E_BADARG;
  int tile = p->tile ? p->tile : auto_tile(a, ut);
  if (c4 && tile != 2 && tile != 5) tile = 2;
  hipStream_t s = (hipStream_t)stream;
  const_cast<stp_conv_params*>(p)->stats_tiles = ceil_div(a.P, tile_pixels(tile));
  {
    // data gradient of a stride-2 convolution through the uniform-tap kernel: parity-class pixel order (see ConvArgs::zperm)
    const int ke = p->dtype == STP_H16 ? 64 : 32;
    if (zperm_applies(a, tile, ut)) {
      a.zperm = 1;
      a.zPc = a.P / 4; a.zH2W2 = (a.Ho / 2) * (a.Wo / 2); a.zW2 = a.Wo / 2; a.zcpt = a.Ctot / ke;
      a.divPc = make_fastdiv((uint32_t)a.zPc); a.divH2W2 = make_fastdiv((uint32_t)a.zH2W2); a.divW2 = make_fastdiv((uint32_t)a.zW2);
      a.divCpt = make_fastdiv((uint32_t)a.zcpt);
    }
    if (p->fold_src) {      // folded shortcut data gradient: only on this path, 3x3 / pad 1 geometry, equal channel counts
      if (!a.zperm || !fold_geometry_ok(a) || !p->fold_weight || p->fold_C != a.C0) return STP_E_BADARG;
      const int64_t bf = (int64_t)a.N * a.Hs0 * a.Ws0 * a.C0 * (p->dtype == STP_H16 ? 2 : 4), bwf = (int64_t)a.wrows * a.C0 * (p->dtype == STP_H16 ? 2 : 4);
      if (bf >= (1ll << 31) || bwf >= (1ll << 31)) return STP_E_BADARG;
      a.fold_src = (const char*)p->fold_src; a.fold_weight = (const char*)p->fold_weight; a.fold_cpt = a.C0 / ke;
      a.bytes_fold = (uint32_t)bf; a.bytesw_fold = (uint32_t)bwf;
    }
  }
  {
    // forward convolution over UpSampling2D(2) + concat with class-collapsed weights (see ConvArgs::upc)
    static const bool upc_on = !(getenv("STP_UPCOLLAPSE") && atoi(getenv("STP_UPCOLLAPSE")) == 0);
    const int ke = p->dtype == STP_H16 ? 64 : 32;
    const bool uni_tile = (tile >= 64 && tile < 96) || tile == 128 + 5;       // the uniform-tap tiles auto_tile() picks: the UPC instances of the kernel
    if (upc_on && p->weight_up && a.mode == STP_SRC_NEAREST2X && ut == 1 && uni_tile && a.stride == 1 && a.KH == 3 && a.KW == 3 && a.pad == 1 &&
        !(a.Ho & 1) && !(a.Wo & 1) && a.Ho == a.Hv && a.Wo == a.Wv && a.Hs0 * 2 == a.Hv && a.Ws0 * 2 == a.Wv && a.C1 > 0 && (a.C1 % ke) == 0 &&
        ((a.P / 4) % tile_pixels(tile)) == 0) {
      const int64_t bwu = (int64_t)a.wrows * 16 * a.C0 * (p->dtype == STP_H16 ? 2 : 4);
      if (bwu < (1ll << 31)) {
        a.upc = 1; a.zperm = 1; a.byteswu = (uint32_t)bwu;
        a.zPc = a.P / 4; a.zH2W2 = (a.Ho / 2) * (a.Wo / 2); a.zW2 = a.Wo / 2; a.zcpt = a.Ctot / ke;
        a.divPc = make_fastdiv((uint32_t)a.zPc); a.divH2W2 = make_fastdiv((uint32_t)a.zH2W2); a.divW2 = make_fastdiv((uint32_t)a.zW2);
        a.divCpt = make_fastdiv((uint32_t)a.zcpt);
        a.ucpt0 = a.C0 / ke; a.ucpt1 = a.C1 / ke;
        a.divU0 = make_fastdiv((uint32_t)a.ucpt0); a.divU1 = make_fastdiv((uint32_t)a.ucpt1);
      }
    }
  }
  if (p->dtype == STP_H16) return c4 ? launch_tile<bf16_t, true>(a, tile, ut, s) : launch_tile<bf16_t, false>(a, tile, ut, s);
  return launch_tile<float, false>(a, tile, ut, s);
}
