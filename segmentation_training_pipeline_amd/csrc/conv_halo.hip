// 3x3 / stride-1 convolution on MFMA with HALO-RESIDENT activation slabs (gfx950, bf16) - forward and data-gradient
// of the ResNet / decoder 3x3 layers with 64..512 channels.
//
// conv_igemm.hip stages one [pixels][64 channels] operand tile PER FILTER TAP: every activation byte travels L2 -> LDS
// nine times, and with the weights that is (BM + BN) * 128 bytes of LDS-DMA per K-step - the launches of the 256/512-channel
// layers are bound by that path (what-if build without MFMA work: 23-35 us of a 40-46 us launch, ~18 TB/s of L2 -> LDS).
// Here a workgroup owns a TH x 16 block of output pixels of ONE image and keeps, per 64-channel chunk ("slab"), the
// (TH+2) x 18 input pixels that the nine taps of the block touch resident in LDS: a tap is just a different LDS row
// offset of the same slab (pixel rows are 128 bytes = 64 channels, 16-byte slots XOR-swizzled by row & 7 exactly as in
// conv_igemm.hip; the swizzle is conflict-free for ANY starting row, so tap-shifted fragment reads stay conflict-free).
// Only the weights stream: BM x 64 per K-step.  L2 -> LDS bytes per workgroup K-step drop from (BM + BN) * 128 to
// BM * 128 + slab / 9, LDS-DMA instructions per wave from 8 to 3.
//
// K order: slab-major, tap-minor (k = slab * 9 + tap; the matching weight column is tap * Cin + slab * 64), so the first
// nine K-steps need only slab 0 and slab s+1 streams in (one 64-pixel pass per K-step) under the taps of slab s.
// 512 threads = 8 waves (2 per SIMD) as WM (channels) x WN (pixel rows); a wave computes 64 channels x RW rows of 16 pixels.
// Ring: 2 slabs + 3 weight stages; one counted s_waitcnt vmcnt + one s_barrier per K-step, every K-step issues the same
// number of LDS-DMA instructions per wave (missing ones go to a sink) so the count is a compile-time constant.
// Epilogue (bias, residual, ReLU, dual destination, accumulate, BatchNormalization statistics, BatchNormalization-backward
// sums): conv_common.h, shared with conv_igemm.hip.
#include "conv_common.h"

#define STP_OOB 0x80000000u
#if defined(STP_HALO_SCHED) && (STP_HALO_SCHED == 1 || STP_HALO_SCHED == 2)
#define HALO_NWST 3
#else
#define HALO_NWST 4
#endif

template <int TH, int BM, int WM, int WN>
__global__ __launch_bounds__(512) void conv_halo_kernel(const ConvArgs a_in) {
  static_assert(WM * WN == 8 && BM % (WM * 64) == 0 && BM / WM == 64 && TH % WN == 0, "config");
#if defined(__HIP_DEVICE_COMPILE__)
#if defined(STP_TIMING)   // scratch build: `bias` carries a u64[4 * workgroups] buffer of shader-clock stamps (scratch/halo_timing.py)
  ConvArgs a = a_in;
  unsigned long long* const tdbg = reinterpret_cast<unsigned long long*>(const_cast<float*>(a_in.bias));
  a.bias = nullptr;
  unsigned long long tstamp[4];
  tstamp[0] = __builtin_amdgcn_s_memtime();
#define STP_STAMP(i) tstamp[i] = __builtin_amdgcn_s_memtime()
#else
  const ConvArgs& a = a_in;
#define STP_STAMP(i)
#endif
  typedef bf16_t T;
  constexpr int TW = 16, HWD = TW + 2, HH = TH + 2, NHP = HH * HWD;
  constexpr int NPASS = (NHP + 63) / 64;         // 64 halo pixels (8 per wave) per pass of the 512 threads
  static_assert(NPASS <= 6, "the next slab must have landed (own pieces) when the K-step of tap 8 begins");
  constexpr int SROWS = (NHP + 7) / 8 * 8;       // slab rows kept in LDS (8 per wave instruction; rows >= NHP are never read)
  constexpr int SLAB = SROWS * 128;
  constexpr int NWST = HALO_NWST;                // weight ring stages
  constexpr int WSTAGE = BM * 128;
  constexpr int LW = BM / 64;                    // weight LDS-DMA instructions per thread per K-step
  constexpr int L = LW + 1;                      // + one slab pass (or sink)
  constexpr int TM = 4, RW = TH / WN, TN = RW;
  constexpr int OFF_W = 2 * SLAB, OFF_DUMP = OFF_W + NWST * WSTAGE, OFF_TAB = OFF_DUMP + 8 * 1024;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int lr = lane & 15, lg = lane >> 4;

  const int bid = xcd_remap(blockIdx.x, a.ntile_m * a.ntile_n);
  const int tile_n = bid / a.ntile_m, tile_m = bid - tile_n * a.ntile_m;   // channel tiles innermost: they share the slabs in L2
  const int tx_n = a.Wo / TW, ty_n = a.Ho / TH;
  const int n = tile_n / (tx_n * ty_n), trem = tile_n - n * (tx_n * ty_n);
  const int ty = trem / tx_n, tx = trem - ty * tx_n;
  const int y0 = ty * TH, x0 = tx * TW, cout0 = tile_m * BM;

  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)a.weight, 0, a.bytesw, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)a.src0, 0, a.bytes0, 0x00020000);

  // ---- per-thread source offsets -----------------------------------------------------------------------------------
  // slab pass i: halo pixel hp = i*64 + tid/8 at physical slot tid&7, which holds logical slot (tid&7) ^ (hp&7)
  const int prow = tid >> 3, pslot = tid & 7;
  uint32_t soff[NPASS];
#pragma unroll
  for (int i = 0; i < NPASS; ++i) {
    const int hp = i * 64 + prow;
    const int hy = hp / HWD, hx = hp - hy * HWD;
    const int y = y0 - 1 + hy, x = x0 - 1 + hx;
    const bool ok = hp < NHP && (unsigned)y < (unsigned)a.Hv && (unsigned)x < (unsigned)a.Wv;
    soff[i] = ok ? (((uint32_t)(n * a.Hv + y) * (uint32_t)a.Wv + (uint32_t)x) * (uint32_t)a.C0 + (uint32_t)((pslot ^ (hp & 7)) * 8)) * 2u : STP_OOB;
  }
  // weight instruction i: row i*64 + tid/8 of the channel tile, logical slot (tid&7) ^ (row&7)   (row&7 == prow&7)
  uint32_t woff[LW];
#pragma unroll
  for (int i = 0; i < LW; ++i)
    woff[i] = ((uint32_t)(cout0 + i * 64 + prow) * (uint32_t)a.K + (uint32_t)((pslot ^ (prow & 7)) * 8)) * 2u;

  const int nslab = a.C0 >> 6;
  const int nk = nslab * 9;
  char* const sink = smem + OFF_DUMP + wave * 1024;

  auto issue_weight_piece = [&](int i, int k, int s, int t) {   // piece i of K-step k = (slab s, tap t) -> ring stage k % NWST
    char* dst = smem + OFF_W + (k % NWST) * WSTAGE;
    const uint32_t kb = (uint32_t)(t * a.C0 + s * 64) * 2u;
    const bool on = k < nk;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (__attribute__((address_space(3))) void*)(on ? dst + (i * 64 + wave * 8) * 128 : sink), 16,
                                             on ? woff[i] + kb : STP_OOB, 0, 0, 0);
  };
  auto issue_weights = [&](int k, int s, int t) {
#pragma unroll
    for (int i = 0; i < LW; ++i) issue_weight_piece(i, k, s, t);
  };
  auto issue_slab_pass = [&](int s, int p) {        // pass p of slab s (no-op -> sink when out of range)
    const bool on = s < nslab && p < NPASS && (p * 64 + wave * 8) < SROWS;
    uint32_t so = STP_OOB;
#pragma unroll
    for (int i = 0; i < NPASS; ++i)
      if (i == p) so = soff[i];
    if (on && so != STP_OOB) so += (uint32_t)s * 128u;
    else so = STP_OOB;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, (__attribute__((address_space(3))) void*)(on ? smem + (s & 1) * SLAB + (p * 64 + wave * 8) * 128 : sink),
                                             16, so, 0, 0, 0);
  };

  // ---- fused PRODUCER BatchNormalization (+activation): src0 holds the convolution output x that the BatchNormalization reads;
  // y = act(fma(x, scale, shift)) is computed IN LDS on the slab, once per slab (9 K-steps), by the thread that DMA'd the 16 bytes
  // (own data: only its own vmcnt orders the read-modify-write, the K-step barrier publishes it).  Same fma / activation / bf16
  // rounding as stp_bn_apply -> bit-identical operands; pixels outside the image stay 0 (the padding applies to y, not x).
  const bool fuse_bn = a.pbn.mean != nullptr;
  float* const tab = reinterpret_cast<float*>(smem + OFF_TAB);      // scale[C0], shift[C0]
  auto transform_slab = [&](int s_) {
    char* sb = smem + (s_ & 1) * SLAB;
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
      if ((i * 64 + wave * 8) >= SROWS) continue;                    // wave-uniform: rows that went to the sink
      const int hp = i * 64 + prow;
      if (soff[i] == STP_OOB) continue;
      u32x4* vp = reinterpret_cast<u32x4*>(sb + hp * 128 + pslot * 16);
      const int ch = s_ * 64 + ((pslot ^ (hp & 7)) << 3);
      const f32x4 sc0 = *reinterpret_cast<const f32x4*>(tab + ch), sc1 = *reinterpret_cast<const f32x4*>(tab + ch + 4);
      const f32x4 sh0 = *reinterpret_cast<const f32x4*>(tab + a.C0 + ch), sh1 = *reinterpret_cast<const f32x4*>(tab + a.C0 + ch + 4);
      const u32x4 v = *vp;
      u32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float lo = bn_act(bn_affine(__uint_as_float(v[e] << 16), e < 2 ? sc0[2 * e] : sc1[2 * e - 4], e < 2 ? sh0[2 * e] : sh1[2 * e - 4]), a.pbn.relu);
        const float hi = bn_act(bn_affine(__uint_as_float(v[e] & 0xffff0000u), e < 2 ? sc0[2 * e + 1] : sc1[2 * e - 3], e < 2 ? sh0[2 * e + 1] : sh1[2 * e - 3]), a.pbn.relu);
        o[e] = pack_bf16x2(lo, hi);
      }
      *vp = o;
    }
  };

  // ---- epilogue operands (residual or BatchNormalization-backward x) first: older than every tile load ---------------
  const int pixb = n * a.HoWo + (y0 + wn * RW) * a.Wo + x0 + lr;     // + j * Wo for fragment column j
  u32x2 pre[TM][TN];
  {
    const T* ps = a.residual ? reinterpret_cast<const T*>(a.residual) : reinterpret_cast<const T*>(a.bnb.x);
    if (ps) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int co = cout0 + wm * 64 + i * 16 + lg * 4;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int pm = pixb + j * a.Wo;
          pre[i][j] = *reinterpret_cast<const u32x2*>(ps + (co + 3 < a.Cout ? (size_t)pm * a.Cout + co : (size_t)0));
        }
      }
    }
  }

  f32x4 acc[TM][TN];
  // ---- fragment reads of one 32-deep half ("chunk" c) of K-step (slab s_, tap t_), weight ring stage st_ ------------
  auto read_chunk = [&](int c, int s_, int t_, int st_, u32x4 (&fa)[TM], u32x4 (&fb)[TN]) {
    const int dy = t_ / 3, dx = t_ - dy * 3;
    const char* wa = smem + OFF_W + st_ * WSTAGE + (wm * 64) * 128;
    const char* sb = smem + (s_ & 1) * SLAB;
    const int hpb = (wn * RW + dy) * HWD + lr + dx;
#if defined(STP_HX) && STP_HX == 2   // what-if: no LDS fragment reads
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[j] = u32x4{(uint32_t)hpb, (uint32_t)j, (uint32_t)c, (uint32_t)lg};
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[i] = u32x4{(uint32_t)st_, (uint32_t)i, (uint32_t)c, (uint32_t)lr};
    asm volatile("" ::: "memory");
    return;
#endif
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int hp = hpb + j * HWD;
      fb[j] = *reinterpret_cast<const u32x4*>(sb + hp * 128 + (((c * 4 + lg) ^ (hp & 7)) << 4));
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int row = i * 16 + lr;
      fa[i] = *reinterpret_cast<const u32x4*>(wa + row * 128 + (((c * 4 + lg) ^ (row & 7)) << 4));
    }
  };
  auto mma_chunk = [&](const u32x4 (&fa)[TM], const u32x4 (&fb)[TN]) {
#if defined(STP_HX) && STP_HX == 1   // what-if: no MFMA work (the fragments stay live)
#pragma unroll
    for (int i = 0; i < TM; ++i) asm volatile("" ::"v"(fa[i]));
#pragma unroll
    for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(fb[j]));
    return;
#endif
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) Mma<T>::run(fa[i], fb[j], acc[i][j]);
  };

  // ---- prologue: slab 0, then the groups { W(0), sink } and { W(1), sink } --------------------------------------------
#pragma unroll
  for (int p = 0; p < NPASS; ++p) issue_slab_pass(0, p);
  issue_weights(0, 0, 0);
  issue_slab_pass(nslab, 0);
  issue_weights(1, 0, 1);
  issue_slab_pass(nslab, 0);
#if HALO_NWST == 4
  issue_weights(2, 0, 2);
  issue_slab_pass(nslab, 0);
#endif

#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

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
#if defined(STP_HALO_SCHED) && STP_HALO_SCHED == 2
  // Software pipeline over the two 32-deep halves of a K-step: the fragment reads of half h+1 are in flight under the MFMAs
  // of half h; the K-step boundary (counted vmcnt + barrier + the next LDS-DMA group) sits BETWEEN the two MFMA halves, so
  // after the barrier every wave has 16 MFMAs queued behind its 8 reads and the LDS burst of the 8 waves hides under them.
  //   group(j), issued right after barrier(j): { W(j+2) -> stage (j+2)%3, pass t_j of slab s_j + 1 }
  //   barrier(j) follows vmcnt(L): only group(j-1) is in flight, so W(j) and every pass of slab s_j have landed for all waves;
  //   before the barrier lgkmcnt(0): this wave's reads of stage j%3... (j-1)%3 and of the previous slab are complete, so the
  //   DMA of group(j) may overwrite them.
  u32x4 fa0[TM], fb0[TN], fa1[TM], fb1[TN];
  wait_vmcnt<L>();
  __builtin_amdgcn_s_barrier();
  issue_weights(2, 0, 2);
  issue_slab_pass(1, 0);
  read_chunk(0, 0, 0, 0, fa0, fb0);
  int s = 0, t = 0;          // K-step k
  int s1 = 0, t1 = 1;        // K-step k + 1
  int s3 = 0, t3 = 3;        // K-step k + 3
  for (int k = 0; k < nk; ++k) {
    read_chunk(1, s, t, k % NWST, fa1, fb1);
    mma_chunk(fa0, fb0);
    if (k + 1 < nk) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      wait_vmcnt<L>();
      __builtin_amdgcn_s_barrier();
      issue_weights(k + 3, s3, t3);
      issue_slab_pass(s1 + 1, t1);
      read_chunk(0, s1, t1, (k + 1) % NWST, fa0, fb0);
    }
    mma_chunk(fa1, fb1);
    s = s1; t = t1;
    if (++t1 == 9) { t1 = 0; ++s1; }
    if (++t3 == 9) { t3 = 0; ++s3; }
  }
#elif defined(STP_HALO_SCHED) && STP_HALO_SCHED == 1
  // One counted wait + one barrier per K-step, then the LDS-DMA group of K-step k+2, the 16 fragment reads and the 32 MFMAs.
  //   group(j), issued right after barrier(j): { W(j+2) -> stage (j+2)%3, pass t_j of slab s_j + 1 }
  //   barrier(j) follows vmcnt(L): only group(j-1) is in flight, so W(j) and every pass of slab s_j have landed for all waves,
  //   and every wave has consumed (MFMA operands) the fragments of K-step j-1: stage (j+2)%3 and slab slot (s_j+1)&1 are free.
  {
    u32x4 fa0[TM], fb0[TN], fa1[TM], fb1[TN];
    int s = 0, t = 0;          // K-step k
    int s2 = 0, t2 = 2;        // K-step k + 2
    for (int k = 0; k < nk; ++k) {
      wait_vmcnt<L>();
#if !defined(STP_HX) || STP_HX != 4   // what-if 4: no barrier in the loop
      __builtin_amdgcn_s_barrier();
#endif
#if !defined(STP_HX) || STP_HX != 3   // what-if 3: no LDS-DMA in the loop
      issue_weights(k + 2, s2, t2);
      issue_slab_pass(s + 1, t);
#endif
      read_chunk(0, s, t, k % NWST, fa0, fb0);
      read_chunk(1, s, t, k % NWST, fa1, fb1);
      __builtin_amdgcn_sched_barrier(0);
      mma_chunk(fa0, fb0);
      mma_chunk(fa1, fb1);
      if (++t == 9) { t = 0; ++s; }
      if (++t2 == 9) { t2 = 0; ++s2; }
    }
  }
#else
  // PING-PONG schedule.  The 8 waves form two groups (waves 0-3 = A, 4-7 = B: one wave of each per SIMD) that run the same
  // loop half a K-step apart: while one group issues its 32 MFMAs the other reads its 16 fragments, then they swap at a
  // barrier - the matrix pipe of every SIMD always has a wave feeding it.  Measured (scratch/halo_timing.py): in the lock-step
  // schedule the memory part of a K-step is a ~1000-cycle LATENCY chain per wave (3 LDS-DMA instructions cost 100-180 issue
  // cycles each when they open a phase, then 16 ds_read_b128 and their wait), which nothing overlapped.  So the LDS-DMA
  // instructions are issued BETWEEN the MFMAs of the compute phase (where they cost ~60 cycles) and the memory phase is
  // fragment reads only.
  //   MEM(k): read the fragments of K-step k; lgkmcnt(0); vmcnt(L) (own pieces of every group but the youngest have landed)
  //   MMA(k): 32 MFMAs, interleaved with group(k) = { W(k+3) -> stage (k+3)%4, pass t_k of slab s_k + 1 }
  //   A:  MEM(0) | MMA(0) | MEM(1) | MMA(1) | ...          B:  -- | MEM(0) | MMA(0) | MEM(1) | ...      ( | = s_barrier )
  // RAW: W(k+1) is group(k-2), covered by the vmcnt(L) that ends MEM(k) of BOTH groups, and a barrier separates those from
  //      either group's MEM(k+1).  WAR: group(k) overwrites W(k-1) / slab s_k - 1, last read in MEM(k-1) of both groups, which
  //      ended (lgkmcnt(0)) at least one barrier before either group's MMA(k).
  {
    u32x4 fa0[TM], fb0[TN], fa1[TM], fb1[TN];
    const bool grp_b = wave >= 4;          // wave-uniform
    wait_vmcnt<2 * L>();                   // slab 0 and W(0)
    if (fuse_bn) {
      transform_slab(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (grp_b) __builtin_amdgcn_s_barrier();
    int s = 0, t = 0;          // K-step k
    int s3 = 0, t3 = 3;        // K-step k + 3
    for (int k = 0; k < nk; ++k) {
      // slab s+1: its passes were issued in MMA(9s .. 9s+NPASS-1) and this wave's own pieces landed before MEM(9s+7) ended
      if (fuse_bn && t == 8 && s + 1 < nslab) transform_slab(s + 1);
      read_chunk(0, s, t, k % NWST, fa0, fb0);
      read_chunk(1, s, t, k % NWST, fa1, fb1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      wait_vmcnt<L>();
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) Mma<T>::run(fa0[i], fb0[j], acc[i][j]);
        __builtin_amdgcn_sched_barrier(0);
        if (i < LW) issue_weight_piece(i, k + 3, s3, t3);
        else if (i == LW) issue_slab_pass(s + 1, t);
        __builtin_amdgcn_sched_barrier(0);
      }
      mma_chunk(fa1, fb1);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      if (++t == 9) { t = 0; ++s; }
      if (++t3 == 9) { t3 = 0; ++s3; }
    }
    if (!grp_b) __builtin_amdgcn_s_barrier();
  }
#endif
  wait_vmcnt<0>();                         // the sink loads of the last two groups
  STP_STAMP(2);
  const int Wo = a.Wo;
  epilogue_px<T, TM, TN, BM, WN, 512, true>(a, cout0, wm * 64, wn, lr, lg, acc, smem, tile_n, pre,
                                            [pixb, Wo](int j) { return pixb + j * Wo; });
#if defined(STP_TIMING)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  STP_STAMP(3);
  if (tid == 0)
    for (int i = 0; i < 4; ++i) tdbg[(size_t)blockIdx.x * 4 + i] = tstamp[i];
#endif
#endif
}

// ================================================================================================ host side
struct HaloCfg { int th, bm; };
// variant ids (stp_conv_params.tile = STP_TILE_HALO + id)
static const HaloCfg HALO_CFGS[] = {{16, 128}, {8, 128}, {16, 64}, {8, 64}};
#define STP_TILE_HALO 1024
#define HALO_NCFG 4

template <int TH, int BM, int WM, int WN>
static int launch_halo(ConvArgs& a, hipStream_t s) {
  static bool attr_set = false;
  constexpr int NHP = (TH + 2) * 18, SROWS = (NHP + 7) / 8 * 8;
  const size_t lds = (size_t)2 * SROWS * 128 + HALO_NWST * BM * 128 + 8 * 1024 + (a.pbn.mean ? (size_t)8 * a.C0 : 0);
  a.ntile_m = ceil_div(a.Cout, BM);
  a.ntile_n = a.N * (a.Ho / TH) * (a.Wo / 16);
  auto kern = conv_halo_kernel<TH, BM, WM, WN>;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return STP_E_LAUNCH;
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(a.ntile_m * a.ntile_n), dim3(512), lds, s, a);
  STP_LAUNCH_CHECK();
  return STP_OK;
}

static bool halo_shape_ok(const stp_conv_params* p) {
  return p && p->dtype == STP_BF16 && p->KH == 3 && p->KW == 3 && p->stride == 1 && p->pad == 1 && p->src0_mode == STP_SRC_DIRECT &&
         p->C1 == 0 && p->C0 >= 64 && (p->C0 % 64) == 0 && p->Ho == p->Hv && p->Wo == p->Wv && p->Hs0 == p->Hv && p->Ws0 == p->Wv &&
         (p->Wo % 16) == 0 && (p->Ho % 8) == 0 && (p->Cout % 16) == 0 && p->Cout >= 64 && !p->dst_sum2x2 && !p->stats_slots &&
         (!p->src_bn_mean || (p->src_bn_rstd && p->C0 <= 512));
}

// variant for a shape: -1 = not eligible / not faster.  Measured on MI355X against the per-tap DMA kernel (scratch/halo_bench.py,
// bs16 U-Net/ResNet34 shapes): 128ch@64^2 30.2 -> 26.9 us (variant 0), 256ch@32^2 32.5 -> 27.4 (1), 512ch@16^2 42.7 -> 36.4 (3),
// 128->384@64^2 78.0 -> 70.8 (0); 64-channel INPUTS (one slab, nine K-steps: nothing to amortise the slab over) are slower
// (42 -> 47 us) and stay on the per-tap kernel.  One 8-wave workgroup per CU (100-158 KB of LDS), so the grid should cover the CUs.
static int halo_auto(const stp_conv_params* p) {
  if (!halo_shape_ok(p) || p->C0 < 128) return -1;
  const int64_t px16 = (p->Ho % 16) == 0 ? (int64_t)p->N * (p->Ho / 16) * (p->Wo / 16) : 0;
  const int64_t px8 = (int64_t)p->N * (p->Ho / 8) * (p->Wo / 16);
  const int ct = ceil_div(p->Cout, 128);
  if (p->Cout > 64 && px16 * ct >= 256) return 0;
  if (p->Cout > 64 && px8 * ct >= 192) return 1;
  if (px8 * ceil_div(p->Cout, 64) >= 192) return 3;
  return -1;
}

extern "C" int stp_conv2d_halo_variant(const stp_conv_params* p) {
  if (!p) return -1;
  if (p->tile >= STP_TILE_HALO && p->tile < STP_TILE_HALO + HALO_NCFG) {
    const int v = p->tile - STP_TILE_HALO;
    if (!halo_shape_ok(p) || (p->Ho % HALO_CFGS[v].th)) return -1;
    return v;
  }
  return p->tile == 0 ? halo_auto(p) : -1;
}

// number of pixel tiles (BatchNormalization partial-sum columns) of a variant
extern "C" int stp_conv2d_halo_tiles(const stp_conv_params* p, int variant) {
  return p->N * (p->Ho / HALO_CFGS[variant].th) * (p->Wo / 16);
}

extern "C" int stp_conv2d_halo(const stp_conv_params* p, int variant, void* stream) {
  if (variant < 0 || variant >= HALO_NCFG || !halo_shape_ok(p) || (p->Ho % HALO_CFGS[variant].th)) return STP_E_BADARG;
  ConvArgs a;
  bool c4;
  int ut;
  const int rc = fill_args(p, a, &c4, &ut);
  if (rc != STP_OK) return rc;
  if (ut != 1) return STP_E_BADARG;   // 32-bit buffer offsets, 64-channel K-steps
  if (a.bnb.x && (!a.stats || !a.bnb.mean || !a.bnb.rstd || p->relu || p->residual)) return STP_E_BADARG;
  const_cast<stp_conv_params*>(p)->stats_tiles = stp_conv2d_halo_tiles(p, variant);
  hipStream_t s = (hipStream_t)stream;
  switch (variant) {
    case 0: return launch_halo<16, 128, 2, 4>(a, s);
    case 1: return launch_halo<8, 128, 2, 4>(a, s);
    case 2: return launch_halo<16, 64, 1, 8>(a, s);
    case 3: return launch_halo<8, 64, 1, 8>(a, s);
    default: return STP_E_BADARG;
  }
}
