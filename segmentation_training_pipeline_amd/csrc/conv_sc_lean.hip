// LEAN form of the small-channel streaming kernel (round 4) - 16-bit storage, the four configurations the training step uses.
//
// conv_sc_stream_kernel (conv_sc.hip) serves every combination of bias / ReLU / accumulate / statistics / BatchNormalization-backward /
// summed upsampling gradient / producer BatchNormalization with run-time flags.  Counters (scratch/r04/pmc_sc_feat.sh, 16 -> 16 channels at
// 16 x 512 x 512, per wave and 8 x 32 tile): 220 VALU + 204 SALU instructions plain, 290 + 210 with the producer BatchNormalization,
// 479 + 378 with the BatchNormalization-backward epilogue - against 20 MFMAs and 20 LDS reads.  With four waves per SIMD that is 58-75 %
// of the vector-issue slots of the launch: these kernels were INSTRUCTION-bound at 2.3-4.4 TB/s, not memory-bound (what-if builds of
// round 2: no stores -7 %, no loads -15 %).  Where the instructions went: per-tile address arithmetic in 64 bits per fragment and per
// staging pass, bounds checks of every fragment, per-ELEMENT branches on the run-time activation mode, one branch per feature flag per
// fragment, (x - mean) * rstd per element, selects for the lanes past K.
// Here:
//   * the configuration is a template parameter (EPI, PBN): no feature branches in the tile loop;
//   * a tile whose halo lies inside the image ("interior": all but the image border) takes a path without bounds checks: the LDS-DMA
//     offsets are (tile base, scalar) + (per-lane constant), the stores / epilogue loads are (tile base, scalar) + (per-lane constant)
//     + (per-fragment scalar) - no per-tile 64-bit vector arithmetic; border tiles take the checked path of the generic kernel;
//   * the activation mask of the fused BatchNormalization backward is one v_med3 + compare against bounds that depend on the mode only
//     (as in conv_halo.hip), sum g * xhat is accumulated as sum g * x and centred once per workgroup, packed fp32 arithmetic on pairs;
//   * everything the generic streaming kernel learned stays: persistent workgroups, weights as register-resident A fragments, the next
//     tile's halo by LDS-DMA into the other half of a double buffer (EVERY wave issues the same number of DMA instructions for every
//     tile, so the compiler's counted wait for the epilogue operands leaves them in flight), a ring of LDS fragment reads in flight
//     (inline asm, counted waits), per-workgroup statistic columns.
// Same MFMA order as the generic kernel: outputs are bit-identical to it; the statistic sums differ in the last bits (other order).
#include "conv_sc.h"

enum { SCL_STATS = 0, SCL_BNB = 1, SCL_BNB_SUM2 = 2, SCL_HEAD = 3 };

struct TileC { int n, y0, x0; bool interior; };      // a tile of the walk: image, first pixel, halo inside the image

__device__ __forceinline__ f32x2 scl_unpack(uint32_t w) { return f32x2{h16lo_to_f32(w), h16hi_to_f32(w)}; }

// one store per fragment: 8 bytes (TM == 1) or the lane's two channel blocks as 16 bytes (TM == 2)
template <int TM>
__device__ __forceinline__ void scl_store(char* p, const u32x2 (&o)[TM]) {
  if constexpr (TM == 2) *reinterpret_cast<u32x4*>(p) = u32x4{o[0].x, o[0].y, o[1].x, o[1].y};
  else *reinterpret_cast<u32x2*>(p) = o[0];
}

template <int CIN, int TM, int EPI, bool PBN, bool UP>
__global__ __launch_bounds__(256, (TM == 1 ? (CIN <= 16 ? SC_WPE_SMALL : 3) : 2)) void conv_sc_lean_kernel(const ScArgs a) {
  typedef bf16_t T;
  constexpr int SZ = 2, VEC = 8, KC = 32, K = 9 * CIN, NCH = (K + KC - 1) / KC, VPP = CIN / VEC, PIXB = CIN * SZ;
  // UP (nearest-2x upsampled source): the tile is staged at the source's LOW resolution - 6 x 18 instead of 10 x 34 pixels (a third of
  // the LDS-DMA bytes, of the LDS and of the producer-BatchNormalization work); the fragment reads resolve hi-res pixel -> low-res pixel
  constexpr int HH = UP ? SC_TH / 2 + 2 : SC_HH, HW = UP ? SC_TW / 2 + 2 : SC_HW;      // staged tile, in source pixels
  constexpr int NV = HH * HW * VPP, NPASS = (NV + 255) / 256, BUF = NPASS * 4096;
  constexpr bool PART = (K % KC) != 0;
  constexpr int COUT = EPI == SCL_HEAD ? 1 : TM * 16;      // (the launcher checks a.Cout == COUT)
  constexpr int CB = COUT * SZ;                              // bytes of an output pixel
  static_assert(EPI != SCL_HEAD || TM == 1, "head: one output channel");
  static_assert(!PBN || EPI == SCL_STATS || EPI == SCL_HEAD, "producer BatchNormalization: forward only");

  // LDS: [2][BUF] halo tiles | statistics scratch [4][TM*16][2] | producer-BN table [2][32]
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* red = reinterpret_cast<float*>(smem + 2 * BUF);
  float* ptab = red + 4 * TM * 16 * 2;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  constexpr int sh = UP ? 1 : 0;                              // (the launcher checks a.up == UP)

  // ---- tiles of this workgroup (XCD by XCD, as the generic kernel) -----------------------------------
  const int ntiles = a.N * a.tiles_x * a.tiles_y;
  int t_first, t_step, t_end;
  if ((gridDim.x & 7) == 0) {
    const int q = ntiles >> 3, r = ntiles & 7, x = blockIdx.x & 7;
    const int start = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    t_first = start + (int)(blockIdx.x >> 3); t_step = (int)(gridDim.x >> 3); t_end = start + q + (x < r ? 1 : 0);
  } else {
    t_first = (int)blockIdx.x; t_step = (int)gridDim.x; t_end = ntiles;
  }
  if (t_first >= t_end) return;

  // ---- staging constants: halo coordinates (border tiles) and the byte offset relative to the tile's first pixel (interior tiles) ----
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.src, 0, a.src_bytes, 0x00020000);
  int hyx[NPASS];            // hy << 16 | hx ; -1 past the tile
  uint32_t lo[NPASS];        // ((hy - 1) * Ws + (hx - 1)) * PIXB + channel vector (source pixels); past the tile: 2^31 (out of range for any tile)
  const int cvb = (tid % VPP) * 16;
#pragma unroll
  for (int p = 0; p < NPASS; ++p) {
    const int v = p * 256 + tid, pix = v / VPP;
    const int hy = pix / HW, hx = pix - hy * HW;
    hyx[p] = v < NV ? (hy << 16 | hx) : -1;
    lo[p] = v < NV ? (uint32_t)(((hy - 1) * a.Ws + (hx - 1)) * PIXB + cvb) : 0x80000000u;
  }

  auto decode = [&](int tile) -> TileC {
    const int bq = (int)fdiv((uint32_t)tile, a.divTx);
    const int tx = tile - bq * a.tiles_x;
    const int n = (int)fdiv((uint32_t)bq, a.divTy);
    const int ty = bq - n * a.tiles_y;
    TileC t;
    t.n = n; t.y0 = ty * SC_TH; t.x0 = tx * SC_TW;
    t.interior = ty > 0 && tx > 0 && t.y0 + SC_TH < a.H && t.x0 + SC_TW < a.W;      // the halo lies inside the image
    return t;
  };
  // LDS-DMA of a tile into buffer half b (NPASS instructions on every path); returns the mask of passes whose vector lies inside the image
  auto issue_tile = [&](const TileC& t, int b, bool live) -> uint32_t {
    uint32_t inside = 0;
    if (live && t.interior) {
      const uint32_t tb = (uint32_t)((t.n * a.Hs + (t.y0 >> sh)) * a.Ws + (t.x0 >> sh)) * (uint32_t)PIXB;
#pragma unroll
      for (int p = 0; p < NPASS; ++p)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + b * BUF + p * 4096 + wave * 1024), 16, (int)(tb + lo[p]), 0, 0, 0);
      inside = (1u << NPASS) - 1u;
    } else {
#pragma unroll
      for (int p = 0; p < NPASS; ++p) {
        const int gy = (t.y0 >> sh) - 1 + (hyx[p] >> 16), gx = (t.x0 >> sh) - 1 + (hyx[p] & 0xffff);      // source pixel
        const bool ok = live && hyx[p] >= 0 && (unsigned)gy < (unsigned)a.Hs && (unsigned)gx < (unsigned)a.Ws;
        const uint32_t off = ok ? (uint32_t)((t.n * a.Hs + gy) * a.Ws + gx) * (uint32_t)PIXB + (uint32_t)cvb : 0x80000000u;
        inside |= ok ? (1u << p) : 0u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + b * BUF + p * 4096 + wave * 1024), 16, (int)off, 0, 0, 0);
      }
    }
    return inside;
  };

  TileC tc = decode(t_first);
  uint32_t inside_cur = issue_tile(tc, 0, true);

  // ---- once per workgroup: weights -> registers (A fragments), lane addresses of the B fragments, constants --------------------
  u32x4 fa[TM][NCH];
  // LDS address of fragment chunk c for tile row 2*wave (+ a under UP: the low-res row of a tile row depends on the tap row), column lr
  // (buffer half 0); lane groups past K: address 0
  constexpr int NA = UP ? 2 : 1;
  uint32_t bla[NA][NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int k0 = c * KC + lg * VEC;
    const int tap = k0 / CIN, ch = k0 - tap * CIN;
    const int kh = tap / 3, kw = tap - kh * 3;
#pragma unroll
    for (int ar = 0; ar < NA; ++ar) {
      // UP: hi-res halo pixel (2 wave + ar + kh, lr + kw) -> low-res tile pixel ((hy + 1) >> 1, (hx + 1) >> 1)
      const int prow = UP ? wave + ((ar + kh + 1) >> 1) : wave * 2 + kh, pcol = UP ? ((lr + kw + 1) >> 1) : kw + lr;
      bla[ar][c] = (uint32_t)(uintptr_t)smem + ((k0 < K) ? (uint32_t)(((prow * HW + pcol) * CIN + ch) * SZ) : 0u);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      u32x4 w = {0u, 0u, 0u, 0u};
      // (TM == 2: row slot (i, m) holds output channel (m >> 2) * 8 + i * 4 + (m & 3) - the lane's two blocks are 8 consecutive channels,
      //  ONE 16-byte store / operand load per fragment, 64 contiguous bytes per pixel and instruction instead of 32)
      const int wrow = TM == 2 ? (lr >> 2) * 8 + i * 4 + (lr & 3) : i * 16 + lr;
      if (k0 < K) w = *reinterpret_cast<const u32x4*>(a.weight + ((size_t)wrow * K + k0) * SZ);
      fa[i][c] = w;
    }
  }
  const bool last_ok = (NCH - 1) * KC + lg * VEC < K;
  if (PBN && tid < CIN) {
    const float r = a.pbn.rstd[tid], sc = a.pbn.gamma ? r * a.pbn.gamma[tid] : r;
    ptab[tid] = sc;
    ptab[32 + tid] = (a.pbn.beta ? a.pbn.beta[tid] : 0.f) - a.pbn.mean[tid] * sc;
  }
  // epilogue constants
  f32x2 ksc[TM][2], ksh[TM][2];      // fused BatchNormalization backward: scale / shift of the lane's 4 channels per 16-channel block, as pairs
  float alo = 0.f, ahi = 0.f;
  if constexpr (EPI == SCL_BNB || EPI == SCL_BNB_SUM2) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const BnBackCh k = bnback_load(a.bnb, TM == 2 ? lg * 8 + i * 4 : i * 16 + lg * 4);
      ksc[i][0] = f32x2{k.sc[0], k.sc[1]}; ksc[i][1] = f32x2{k.sc[2], k.sc[3]};
      ksh[i][0] = f32x2{k.sh[0], k.sh[1]}; ksh[i][1] = f32x2{k.sh[2], k.sh[3]};
    }
    // activation window of the fused BatchNormalization (the gradient passes strictly inside it): t is "on" iff it equals its clamp
    // to [smallest positive number, largest float below the upper bound]
    alo = a.bnb.relu ? __uint_as_float(1u) : -__builtin_inff();
    ahi = a.bnb.relu == 2 ? __uint_as_float(0x40bfffffu) : __builtin_inff();
  }
  float bias0 = 0.f;
  if constexpr (EPI == SCL_HEAD) bias0 = a.bias ? a.bias[0] : 0.f;
  // per-lane byte offset of the lane's first output inside a tile (pixel (2 * wave, lr), channels lg * 4 ..); fragment f adds
  // ((f >> 1) * W + (f & 1) * 16) * CB, channel block i adds 32 bytes
  constexpr int LB = TM == 2 ? 16 : 8;      // bytes of a pixel owned by one lane
  const uint32_t so = EPI == SCL_BNB_SUM2 ? (uint32_t)((wave * (a.W >> 1) + (lr >> 1)) * CB + lg * LB)
                      : EPI == SCL_HEAD   ? (uint32_t)(((wave * 2) * a.W + lr) * CB)
                                          : (uint32_t)(((wave * 2) * a.W + lr) * CB + lg * LB);
  f32x4 ss[TM], qq[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) { ss[i] = f32x4{0.f, 0.f, 0.f, 0.f}; qq[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  lds_barrier();               // the table is visible

  // epilogue operands (the BatchNormalization input at the lane's output pixels) of tile pt -> xo
  // (border tiles: the lane's pixel of fragment f may lie outside the image; interior tiles carry no checks at all)
  const bool own = !(lr & 1);        // summed upsampling gradient: even lanes own the low-resolution pixel
  auto prefetch = [&](const TileC& pt, const bool EDGE, u32x2 (&xo)[TM][4]) __attribute__((always_inline)) {
    if constexpr (EPI == SCL_BNB) {
      const char* xb = a.bnb.x + (((size_t)pt.n * a.H + pt.y0) * a.W + pt.x0) * CB;
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        const char* xf = xb + (size_t)(((f >> 1) * a.W + (f & 1) * 16) * CB);
        const bool ok = !EDGE || (pt.y0 + wave * 2 + (f >> 1) < a.H && pt.x0 + (f & 1) * 16 + lr < a.W);
        if constexpr (TM == 2) {
          const u32x4 w4 = *reinterpret_cast<const u32x4*>(ok ? xf + so : a.bnb.x);
          xo[0][f] = u32x2{w4.x, w4.y}; xo[TM - 1][f] = u32x2{w4.z, w4.w};
        } else {
          xo[0][f] = *reinterpret_cast<const u32x2*>(ok ? xf + so : a.bnb.x);
        }
      }
    }
    if constexpr (EPI == SCL_BNB_SUM2) {
      const char* xb = a.bnb.x + (((size_t)pt.n * (a.H >> 1) + (pt.y0 >> 1)) * (a.W >> 1) + (pt.x0 >> 1)) * CB;
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        const char* xf = xb + (size_t)(h2 * 8 * CB);
        const bool ok = own && (!EDGE || (pt.y0 + wave * 2 < a.H && pt.x0 + h2 * 16 + lr < a.W));
        if constexpr (TM == 2) {
          const u32x4 w4 = *reinterpret_cast<const u32x4*>(ok ? xf + so : a.bnb.x);
          xo[0][h2] = u32x2{w4.x, w4.y}; xo[TM - 1][h2] = u32x2{w4.z, w4.w};
        } else {
          xo[0][h2] = *reinterpret_cast<const u32x2*>(ok ? xf + so : a.bnb.x);
        }
      }
    }
  };
  u32x2 xp[TM][4];                   // ... of the CURRENT tile
  // (32 -> 32 channels: the second operand set does not fit 256 registers - measured with 64 bytes of spills - so that instance fetches
  //  them at the top of their own tile)
  constexpr bool AHEAD = (EPI == SCL_BNB || EPI == SCL_BNB_SUM2) && !(CIN == 32 && TM == 2 && EPI == SCL_BNB);
  if constexpr (AHEAD) {
    if (tc.interior) prefetch(tc, false, xp); else prefetch(tc, true, xp);
  }

  // (one copy of the tile body: the buffer half is a run-time scalar - the per-lane fragment addresses are re-based once per tile, NCH
  //  additions, and the code is half the size of the generic kernel's two specialised copies)
  int cur = 0;
  for (int tile = t_first; tile < t_end; tile += t_step, cur ^= 1) {
    const TileC t = tc;
    ScStageBn<T> sbn;
    if (PBN) sbn.load_tab(ptab, ptab + 32, (tid % VPP) * VEC);      // (table reads: requested before the wait)
    // this tile's pieces (own) have landed; the previous tile's stores are out.  (The builtin, not inline asm: the compiler's
    // wait-count bookkeeping restarts from zero here.)
    __builtin_amdgcn_s_waitcnt(0x0f70);                   // vmcnt(0)
    asm volatile("" ::: "memory");
    if (PBN) {
      // padding applies to the NORMALISED tensor: out-of-image vectors stay zero (border tiles; the slack past the tile does not matter)
#pragma unroll
      for (int p0 = 0; p0 < NPASS; p0 += 3) {
        u32x4 v[3];
#pragma unroll
        for (int q = 0; q < 3; ++q)
          if (p0 + q < NPASS) v[q] = *reinterpret_cast<const u32x4*>(smem + cur * BUF + ((p0 + q) * 256 + tid) * 16);
#pragma unroll
        for (int q = 0; q < 3; ++q)
          if (p0 + q < NPASS && (inside_cur & (1u << (p0 + q))))
            *reinterpret_cast<u32x4*>(smem + cur * BUF + ((p0 + q) * 256 + tid) * 16) = sbn.apply(v[q], a.pbn.relu);
      }
    }
    lds_barrier();

    // ---- the next tile's halo, then the epilogue operands of the NEXT tile (BatchNormalization input): requested a whole tile ahead -
    // fetched at the top of their own tile they came back after the (short) MFMA phase: 540 of 2200 wave-cycles per tile waiting ------
    const bool edge = !t.interior;
    if constexpr (!AHEAD) {             // (this tile's operands, at the top of the tile)
      if (edge) prefetch(t, true, xp); else prefetch(t, false, xp);
    }
    const int next = tile + t_step;
    const bool live = next < t_end;
    if (live) tc = decode(next);
    inside_cur = issue_tile(tc, cur ^ 1, live);
    u32x2 xn[TM][4];
    if (AHEAD && live) {
      if (tc.interior) prefetch(tc, false, xn); else prefetch(tc, true, xn);
    }

    // ---- MFMAs: wave w owns tile rows 2w, 2w+1; 4 fragments of 16 pixels; a ring of fragment reads in flight ----------
    f32x4 acc[TM][4];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int f = 0; f < 4; ++f) acc[i][f] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
      constexpr int R = 4 * NCH;                            // reads of a tile, fragment-major: k = f * NCH + c
      // (32 -> 32 channels with the BatchNormalization-backward operands of two tiles live: a shorter ring instead of spills)
      constexpr int RING = (CIN == 32 && TM == 2 && EPI == SCL_BNB_SUM2) ? 4 : SC_RING;
      constexpr int D = RING < R ? RING : R;
      u32x4 ring[D];
      auto issue = [&ring, &bla](auto kc) {                  // (bla points into this tile's buffer half: re-based at the end of every tile)
        constexpr int k = decltype(kc)::value, F = k / NCH, c = k % NCH;
        // the fragment's tile row / column half: an instruction offset (UP: the row is part of the address register, the half = 8 low-res pixels)
        constexpr int OFF = UP ? (F & 1) * 8 * PIXB : ((F >> 1) * SC_HW + (F & 1) * 16) * PIXB;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ring[k % D]) : "v"(bla[UP ? (F >> 1) : 0][c]), "n"(OFF));
      };
      sc_unroll<D>(issue);
      sc_unroll<R>([&ring, &fa, &acc, &issue, last_ok](auto kc) {
        constexpr int k = decltype(kc)::value, F = k / NCH, c = k % NCH;
        constexpr int young = (R - 1 - k) < (D - 1) ? (R - 1 - k) : (D - 1);      // reads issued after read k at this point
        asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(ring[k % D]) : "n"(young));
        u32x4 v = ring[k % D];
        if (PART && c == NCH - 1 && !last_ok) v = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[i][F] = mfma16_16x16x32(fa[i][c], v, acc[i][F]);
        if constexpr (k + D < R) issue(std::integral_constant<int, k + D>{});
      });
    }

    // ---- epilogue -----------------------------------------------------------------------------------
    auto epilogue = [&](const bool EDGE) __attribute__((always_inline)) {
      bool okf[4];
#pragma unroll
      for (int f = 0; f < 4; ++f) okf[f] = !EDGE || (t.y0 + wave * 2 + (f >> 1) < a.H && t.x0 + (f & 1) * 16 + lr < a.W);
      if constexpr (EPI == SCL_STATS) {
        char* ob = a.dst + (((size_t)t.n * a.H + t.y0) * a.W + t.x0) * CB;
        const bool st = a.stats != nullptr;
#pragma unroll
        for (int f = 0; f < 4; ++f) {
          char* of = ob + (size_t)(((f >> 1) * a.W + (f & 1) * 16) * CB);
          u32x2 o[TM];
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            const f32x4 v = acc[i][f];
            o[i] = u32x2{pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w)};
            if (st && okf[f]) {
              const f32x2 s0 = scl_unpack(o[i].x), s1 = scl_unpack(o[i].y);
              const f32x4 sv = {s0.x, s0.y, s1.x, s1.y};
              ss[i] += sv;
              qq[i] += sv * sv;
            }
          }
          if (okf[f]) scl_store<TM>(of + so, o);
        }
      }
      if constexpr (EPI == SCL_HEAD) {
        char* ob = a.dst + (((size_t)t.n * a.H + t.y0) * a.W + t.x0) * CB;
#pragma unroll
        for (int f = 0; f < 4; ++f) {
          char* of = ob + (size_t)(((f >> 1) * a.W + (f & 1) * 16) * CB);
          const float x = acc[0][f][0] + bias0;
          if (lg == 0 && okf[f]) *reinterpret_cast<T*>(of + so) = (T)(pack_bf16x2(x, 0.f) & 0xffffu);
        }
      }
      if constexpr (EPI == SCL_BNB) {
        char* ob = a.dst + (((size_t)t.n * a.H + t.y0) * a.W + t.x0) * CB;
#pragma unroll
        for (int f = 0; f < 4; ++f) {
          char* of = ob + (size_t)(((f >> 1) * a.W + (f & 1) * 16) * CB);
          u32x2 oo[TM];
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            const f32x4 v = acc[i][f];
            u32x2& o = oo[i];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              const uint32_t stw = pack_bf16x2(v[2 * e], v[2 * e + 1]);        // dY as it would be stored
              const f32x2 dy = scl_unpack(stw), xv = scl_unpack(xp[i][f][e]);
              const f32x2 tt = __builtin_elementwise_fma(xv, ksc[i][e], ksh[i][e]);
              const f32x2 g = {okf[f] && __builtin_amdgcn_fmed3f(tt.x, alo, ahi) == tt.x ? dy.x : 0.f,
                               okf[f] && __builtin_amdgcn_fmed3f(tt.y, alo, ahi) == tt.y ? dy.y : 0.f};
              ss[i][2 * e] += g.x; ss[i][2 * e + 1] += g.y;
              qq[i][2 * e] = fmaf(g.x, xv.x, qq[i][2 * e]); qq[i][2 * e + 1] = fmaf(g.y, xv.y, qq[i][2 * e + 1]);
              o[e] = pack_bf16x2(g.x, g.y);
            }
          }
          if (okf[f]) scl_store<TM>(of + so, oo);
        }
      }
      if constexpr (EPI == SCL_BNB_SUM2) {
        // gradient of UpSampling2D(2): the wave's two tile rows are one output row (fragments h2 and h2 + 2, same lane), lanes lr and
        // lr ^ 1 one output column (quad_perm DPP); even lanes own the low-resolution pixel
        char* ob = a.dst + (((size_t)t.n * (a.H >> 1) + (t.y0 >> 1)) * (a.W >> 1) + (t.x0 >> 1)) * CB;
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          char* of = ob + (size_t)(h2 * 8 * CB);
          const bool mine = own && okf[h2];
          u32x2 oo[TM];
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            f32x4 v = acc[i][h2] + acc[i][h2 + 2];
#pragma unroll
            for (int e = 0; e < 4; ++e)
              v[e] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[e]), 0xB1, 0xf, 0xf, true));  // quad_perm [1,0,3,2]
            u32x2& o = oo[i];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              const uint32_t stw = pack_bf16x2(v[2 * e], v[2 * e + 1]);
              const f32x2 dy = scl_unpack(stw), xv = scl_unpack(xp[i][h2][e]);
              const f32x2 tt = __builtin_elementwise_fma(xv, ksc[i][e], ksh[i][e]);
              const f32x2 g = {mine && __builtin_amdgcn_fmed3f(tt.x, alo, ahi) == tt.x ? dy.x : 0.f,
                               mine && __builtin_amdgcn_fmed3f(tt.y, alo, ahi) == tt.y ? dy.y : 0.f};
              ss[i][2 * e] += g.x; ss[i][2 * e + 1] += g.y;
              qq[i][2 * e] = fmaf(g.x, xv.x, qq[i][2 * e]); qq[i][2 * e + 1] = fmaf(g.y, xv.y, qq[i][2 * e + 1]);
              o[e] = pack_bf16x2(g.x, g.y);
            }
          }
          if (mine) scl_store<TM>(of + so, oo);
        }
      }
    };
    if (edge) epilogue(true); else epilogue(false);
    {
      const uint32_t flip = cur ? (uint32_t)(-BUF) : (uint32_t)BUF;      // the fragment addresses move to the other buffer half
#pragma unroll
      for (int ar = 0; ar < NA; ++ar)
#pragma unroll
        for (int c = 0; c < NCH; ++c) bla[ar][c] += flip;
    }
    if constexpr (AHEAD) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int f = 0; f < 4; ++f) xp[i][f] = xn[i][f];
    }
  }

  // ---- one column of [stat][channel][workgroups] per workgroup --------------------------------------------
  if (EPI != SCL_HEAD && a.stats) {
    if constexpr (EPI == SCL_BNB || EPI == SCL_BNB_SUM2) {
      // sum g * xhat = rstd * (sum g * x - mean * sum g), per lane (linear, so the partition does not matter)
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int cb = TM == 2 ? lg * 8 + i * 4 : i * 16 + lg * 4;
        const f32x4 mu = *reinterpret_cast<const f32x4*>(a.bnb.mean + cb), rsd = *reinterpret_cast<const f32x4*>(a.bnb.rstd + cb);
#pragma unroll
        for (int e = 0; e < 4; ++e) qq[i][e] = rsd[e] * (qq[i][e] - mu[e] * ss[i][e]);
      }
    }
    // butterfly over the 16 pixel lanes, then the 4 waves (same channels, different rows) through LDS
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float sv = row_sum16_to_lane15(ss[i][e]), qv = row_sum16_to_lane15(qq[i][e]);
        if (lr == 15) {
          const int cl = (TM == 2 ? lg * 8 + i * 4 : i * 16 + lg * 4) + e;
          red[(wave * TM * 16 + cl) * 2] = sv;
          red[(wave * TM * 16 + cl) * 2 + 1] = qv;
        }
      }
    lds_barrier();
    if (tid < TM * 16) {
      float sv = 0.f, qv = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) { sv += red[(w * TM * 16 + tid) * 2]; qv += red[(w * TM * 16 + tid) * 2 + 1]; }
      a.stats[(size_t)tid * gridDim.x + blockIdx.x] = sv;                     // [stat][channel][workgroup]
      a.stats[((size_t)COUT + tid) * gridDim.x + blockIdx.x] = qv;
    }
  }
}

template <int CIN, int TM, int EPI, bool PBN, bool UP>
static int launch_scl(const ScArgs& a, hipStream_t s) {
  const int ntiles = a.N * a.tiles_x * a.tiles_y;
  constexpr int NV = (UP ? (SC_TH / 2 + 2) * (SC_TW / 2 + 2) : SC_HH * SC_HW) * (CIN / 8), NPASS = (NV + 255) / 256;
  const size_t lds = (size_t)2 * NPASS * 4096 + (4 * TM * 16 * 2 + 64) * sizeof(float);
  const int blocks = sc_stream_blocks(STP_H16, CIN, TM * 16, ntiles);
  static bool attr_set = false;
  if (lds > 64 * 1024 && !attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_sc_lean_kernel<CIN, TM, EPI, PBN, UP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return STP_E_LAUNCH;
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_sc_lean_kernel<CIN, TM, EPI, PBN, UP>), dim3(blocks), dim3(256), lds, s, a);
  STP_LAUNCH_CHECK();
  return STP_OK;
}

template <int EPI, bool PBN, bool UP>
static int dispatch_scl2(const ScArgs& a, int cin, int tm, hipStream_t s) {
  switch (cin * 4 + tm) {
    case 8 * 4 + 1: return launch_scl<8, 1, EPI, PBN, UP>(a, s);
    case 16 * 4 + 1: return launch_scl<16, 1, EPI, PBN, UP>(a, s);
    case 32 * 4 + 1: return launch_scl<32, 1, EPI, PBN, UP>(a, s);
    case 8 * 4 + 2: if constexpr (EPI != SCL_HEAD) return launch_scl<8, 2, EPI, PBN, UP>(a, s); else return 1;
    case 16 * 4 + 2: if constexpr (EPI != SCL_HEAD) return launch_scl<16, 2, EPI, PBN, UP>(a, s); else return 1;
    case 32 * 4 + 2: if constexpr (EPI != SCL_HEAD) return launch_scl<32, 2, EPI, PBN, UP>(a, s); else return 1;
    default: return 1;
  }
}
// (the upsampled-source form exists for the forward with statistics only: decoder conv3x3(UpSampling2D(2)(x)))
template <int EPI, bool PBN>
static int dispatch_scl(const ScArgs& a, int cin, int tm, hipStream_t s) {
  if (a.up) {
    if constexpr (EPI == SCL_STATS) return dispatch_scl2<EPI, PBN, true>(a, cin, tm, s); else return 1;
  }
  return dispatch_scl2<EPI, PBN, false>(a, cin, tm, s);
}

static bool sc_lean_on() {
  static const bool on = !(getenv("STP_SC_LEAN") && atoi(getenv("STP_SC_LEAN")) == 0);
  return on;
}

// Does the lean kernel serve this configuration?  1 = no (the caller falls back to the generic streaming kernel).
int sc_lean_launch(const ScArgs& a, int cin, int dtype, hipStream_t s) {
  if (!sc_lean_on() || !sc_stream_on() || dtype != STP_H16 || a.stat_slots || a.accumulate || a.relu) return 1;
  const bool pbn = a.pbn.mean != nullptr;
  const int tm = a.Cout <= 16 ? 1 : 2;
  if (a.bnb.x) {
    if (pbn || a.bias || !a.stats || a.Cout != tm * 16) return 1;
    return a.sum2 ? dispatch_scl<SCL_BNB_SUM2, false>(a, cin, tm, s) : dispatch_scl<SCL_BNB, false>(a, cin, tm, s);
  }
  if (a.sum2) return 1;
  if (a.Cout == 1) {
    if (a.stats) return 1;
    return pbn ? dispatch_scl<SCL_HEAD, true>(a, cin, 1, s) : dispatch_scl<SCL_HEAD, false>(a, cin, 1, s);
  }
  if (a.bias || a.Cout != tm * 16) return 1;
  return pbn ? dispatch_scl<SCL_STATS, true>(a, cin, tm, s) : dispatch_scl<SCL_STATS, false>(a, cin, tm, s);
}

// =================================================================================================
// LEAN small-channel WEIGHT GRADIENT:  dW[co][tap * CIN + ci] = sum over pixels of dY[p][co] * X[p + tap][ci]
//
// conv_sc_wgrad_stream_kernel (conv_sc.hip) gives wave w the taps w, w + 4, w + 8 of every tile row: wave 0 does 3/2 of the others' work,
// every wave re-reads the dY fragment of every row, the row loop keeps one transpose read in flight, and the compiler puts an
// s_waitcnt vmcnt(0) - i.e. the NEXT tile's LDS-DMA - in front of the first LDS read of every tile (it cannot tell the two apart).
// 56-130 us per launch for layers whose operands take 17-34 us at HBM speed.  Here:
//   * a wave owns two of the tile's eight rows and ALL nine taps: per halo row (four per wave) three B fragments per 16-channel block
//     serve the taps (kh, 0..2) of both rows that see it - 28 / 52 transpose reads and 18 / 36 MFMAs per tile and wave (16 / 32 input
//     channels) instead of 64 / 112 and 24 / 48 on the critical wave; the four waves' accumulators meet once per workgroup, through LDS,
//     in a fixed order;
//   * the reads of the next halo row are in flight while the MFMAs of this one issue (inline asm, counted waits: the compiler never
//     sees an LDS read next to the DMA);
//   * interior tiles: LDS-DMA offsets = (tile base, scalar) + (per-lane constant); every wave issues the same number of DMA
//     instructions for every tile; tiles walked XCD by XCD.
// Slab contract unchanged (one fp32 slab per workgroup, summed by the fixed-order reduce kernel); the sums are taken in another
// order than the generic kernel's, so the two agree to fp32 rounding, not bit for bit.
// =================================================================================================
template <int CIN, int COUT, bool PBN, bool UP>
__global__ __launch_bounds__(256) void conv_sc_wgrad_lean_kernel(const ScWgArgs a) {
  typedef bf16_t T;
  constexpr int SZ = 2, VEC = 8, TMo = (COUT + 15) / 16, TNi = CIN / 16, PIXB = CIN * SZ, DYB = COUT * SZ;
  constexpr int VPPX = CIN / VEC, VPPD = COUT / VEC;
  constexpr int HH = UP ? SC_TH / 2 + 2 : SC_HH, HW = UP ? SC_TW / 2 + 2 : SC_HW;      // staged X tile in SOURCE pixels (UP: low resolution, see above)
  constexpr int NVX = HH * HW * VPPX, NPX = (NVX + 255) / 256;      // halo vectors / passes
  constexpr int NVD = SC_TH * SC_TW * VPPD, NPD = (NVD + 255) / 256;      // dY vectors / passes
  constexpr int XBUF = NPX * 4096, DBUF = NPD * 4096, BUF = XBUF + DBUF;  // whole 1 KB wave pieces
  constexpr int F3 = 3 * TMo * TNi;                                        // accumulator fragments of one filter row
  static_assert(4 * F3 * 1024 <= 2 * BUF, "the final reduction of a filter row fits the (dead) staging buffers");

  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][halo tile | dY tile]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  constexpr int sh = UP ? 1 : 0;                              // (the launcher checks a.up == UP)

  int t_first, t_step, t_end;
  if ((gridDim.x & 7) == 0) {
    const int q = a.ntiles >> 3, r = a.ntiles & 7, x = blockIdx.x & 7;
    const int start = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    t_first = start + (int)(blockIdx.x >> 3); t_step = (int)(gridDim.x >> 3); t_end = start + q + (x < r ? 1 : 0);
  } else {
    t_first = (int)blockIdx.x; t_step = (int)gridDim.x; t_end = a.ntiles;
  }

  f32x4 acc[9][TMo][TNi];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < TMo; ++i)
#pragma unroll
      for (int j = 0; j < TNi; ++j) acc[t][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (t_first < t_end) {
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)a.src, 0, a.src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, a.dy_bytes, 0x00020000);
    int hyx[NPX], pyx[NPD];
    uint32_t lox[NPX], lod[NPD];     // byte offset relative to the tile's first pixel (interior tiles); past the tile: 2^31
    const uint32_t cvx = (uint32_t)(tid % VPPX) * 16u, cvd = (uint32_t)(tid % VPPD) * 16u;
#pragma unroll
    for (int p = 0; p < NPX; ++p) {
      const int v = p * 256 + tid, pix = v / VPPX;
      const int hy = pix / HW, hx = pix - hy * HW;
      hyx[p] = v < NVX ? (hy << 16 | hx) : -1;
      lox[p] = v < NVX ? (uint32_t)(((hy - 1) * a.Ws + (hx - 1)) * PIXB) + cvx : 0x80000000u;
    }
#pragma unroll
    for (int p = 0; p < NPD; ++p) {
      const int v = p * 256 + tid, pix = v / VPPD;
      const int py = pix / SC_TW, px = pix % SC_TW;
      pyx[p] = v < NVD ? (py << 16 | px) : -1;
      lod[p] = v < NVD ? (uint32_t)((py * a.W + px) * DYB) + cvd : 0x80000000u;
    }
    auto decode = [&](int tile) -> TileC {
      int b = tile;
      const int tx = b % a.tiles_x; b /= a.tiles_x;
      const int ty = b % a.tiles_y;
      TileC t;
      t.n = b / a.tiles_y; t.y0 = ty * SC_TH; t.x0 = tx * SC_TW;
      t.interior = ty > 0 && tx > 0 && t.y0 + SC_TH < a.H && t.x0 + SC_TW < a.W;
      return t;
    };
    // NPX + NPD LDS-DMA instructions on every path; returns the mask of halo passes whose vector lies inside the image
    auto issue_tile = [&](const TileC& t, int bsel, bool live) -> uint32_t {
      uint32_t inside = 0;
      char* xb = smem + bsel * BUF;
      if (live && t.interior) {
        const uint32_t tbx = (uint32_t)((t.n * a.Hs + (t.y0 >> sh)) * a.Ws + (t.x0 >> sh)) * (uint32_t)PIXB;
        const uint32_t tbd = (uint32_t)((t.n * a.H + t.y0) * a.W + t.x0) * (uint32_t)DYB;
#pragma unroll
        for (int p = 0; p < NPX; ++p)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (__attribute__((address_space(3))) void*)(xb + p * 4096 + wave * 1024), 16, (int)(tbx + lox[p]), 0, 0, 0);
#pragma unroll
        for (int p = 0; p < NPD; ++p)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsd, (__attribute__((address_space(3))) void*)(xb + XBUF + p * 4096 + wave * 1024), 16, (int)(tbd + lod[p]), 0, 0, 0);
        inside = (1u << NPX) - 1u;
      } else {
#pragma unroll
        for (int p = 0; p < NPX; ++p) {
          const int gy = (t.y0 >> sh) - 1 + (hyx[p] >> 16), gx = (t.x0 >> sh) - 1 + (hyx[p] & 0xffff);      // source pixel
          const bool ok = live && hyx[p] >= 0 && (unsigned)gy < (unsigned)a.Hs && (unsigned)gx < (unsigned)a.Ws;
          const uint32_t off = ok ? (uint32_t)((t.n * a.Hs + gy) * a.Ws + gx) * (uint32_t)PIXB + cvx : 0x80000000u;
          inside |= ok ? (1u << p) : 0u;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (__attribute__((address_space(3))) void*)(xb + p * 4096 + wave * 1024), 16, (int)off, 0, 0, 0);
        }
#pragma unroll
        for (int p = 0; p < NPD; ++p) {
          const int gy = t.y0 + (pyx[p] >> 16), gx = t.x0 + (pyx[p] & 0xffff);
          const bool ok = live && pyx[p] >= 0 && gy < a.H && gx < a.W;
          const uint32_t off = ok ? (uint32_t)((t.n * a.H + gy) * a.W + gx) * (uint32_t)DYB + cvd : 0x80000000u;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsd, (__attribute__((address_space(3))) void*)(xb + XBUF + p * 4096 + wave * 1024), 16, (int)off, 0, 0, 0);
        }
      }
      return inside;
    };

    TileC tc = decode(t_first);
    uint32_t inside_cur = issue_tile(tc, 0, true);
    ScStageBn<T> sbn;
    if (PBN) sbn.load(a.pbn, (tid % VPPX) * VEC);
    // lane group g owns pixels x = 4g..4g+3 (lo) and 16+4g..16+4g+3 (hi) of a 32-pixel row; lane i of a group addresses pixel (i>>2), quad (i&3)
    const int xl = lg * 4 + (lr >> 2), qb = (lr & 3) * 8;
    const uint32_t aA0 = (uint32_t)(uintptr_t)smem + (uint32_t)(XBUF + ((2 * wave * SC_TW + xl) * COUT) * SZ + qb);      // dY row 2 * wave
    // X fragments: halo row 2 * wave (+ h), pixel xl + kw (+ 16); UP: hi-res halo pixel (hy, hx) -> low-res tile pixel ((hy + 1) >> 1, (hx + 1) >> 1),
    // one address register per kw (the column mapping is not additive in kw)
    uint32_t aB0[3];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
      aB0[kw] = (uint32_t)(uintptr_t)smem + (uint32_t)((UP ? (wave * HW + ((xl + kw + 1) >> 1)) : ((2 * wave) * HW + xl + kw)) * CIN * SZ + qb);

    int cur = 0;
    for (int tile = t_first; tile < t_end; tile += t_step, cur ^= 1) {
      __builtin_amdgcn_s_waitcnt(0x0f70);                   // vmcnt(0): own pieces of this tile have landed
      asm volatile("" ::: "memory");
      if (PBN) {
        // padding applies to the NORMALISED tensor: out-of-image vectors stay zero
#pragma unroll
        for (int p0 = 0; p0 < NPX; p0 += 3) {
          u32x4 v[3];
#pragma unroll
          for (int q = 0; q < 3; ++q)
            if (p0 + q < NPX) v[q] = *reinterpret_cast<const u32x4*>(smem + cur * BUF + ((p0 + q) * 256 + tid) * 16);
#pragma unroll
          for (int q = 0; q < 3; ++q)
            if (p0 + q < NPX && (inside_cur & (1u << (p0 + q))))
              *reinterpret_cast<u32x4*>(smem + cur * BUF + ((p0 + q) * 256 + tid) * 16) = sbn.apply(v[q], a.pbn.relu);
        }
      }
      lds_barrier();                                          // tile visible; every wave has left the other buffer
      const int next = tile + t_step;
      const bool live = next < t_end;
      if (live) tc = decode(next);
      inside_cur = issue_tile(tc, cur ^ 1, live);

      const uint32_t aA = aA0 + (uint32_t)(cur * BUF);
      uint32_t aB[3];
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) aB[kw] = aB0[kw] + (uint32_t)(cur * BUF);
      u32x2 fa[2][TMo][2];          // dY fragments of the wave's two rows: [row][16-channel block][lo / hi pixels]
      u32x2 fb[2][3][TNi][2];       // X fragments of one halo row, double-buffered: [buffer][kw][16-channel block][lo / hi]
      sc_unroll<2 * TMo * 2>([&fa, aA](auto kc) {
        constexpr int k = decltype(kc)::value, r = k / (TMo * 2), i = (k / 2) % TMo, hl = k & 1;
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(fa[r][i][hl]) : "v"(aA), "n"((r * SC_TW * COUT + i * 16 + hl * 16 * COUT) * SZ));
      });
      auto issue_b = [&fb, &aB](auto hc) {
        constexpr int h = decltype(hc)::value;
        sc_unroll<3 * TNi * 2>([&fb, &aB](auto kc) {
          constexpr int k = decltype(kc)::value, kw = k / (TNi * 2), j = (k / 2) % TNi, hl = k & 1;
          // row h of the wave's four halo rows, pixel half hl (UP: low-res row (h + 1) >> 1, half = 8 low-res pixels)
          constexpr int OFF = UP ? ((((h + 1) >> 1) * HW + hl * 8) * CIN + j * 16) * SZ : ((h * HW + hl * 16) * CIN + j * 16) * SZ;
          asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(fb[h & 1][kw][j][hl]) : "v"(aB[kw]), "n"(OFF));
        });
      };
      issue_b(std::integral_constant<int, 0>{});
      sc_unroll<4>([&](auto hc) {
        constexpr int h = decltype(hc)::value;
        if constexpr (h < 3) issue_b(std::integral_constant<int, h + 1>{});
        // all but the reads of the next halo row have returned
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(h < 3 ? 3 * TNi * 2 : 0) : "memory");
        if constexpr (h == 0) {
#pragma unroll
          for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int i = 0; i < TMo; ++i) { asm volatile("" : "+v"(fa[r][i][0])); asm volatile("" : "+v"(fa[r][i][1])); }
        }
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
          for (int j = 0; j < TNi; ++j) { asm volatile("" : "+v"(fb[h & 1][kw][j][0])); asm volatile("" : "+v"(fb[h & 1][kw][j][1])); }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const int kh = h - r;           // halo row h of the wave = tile row 2 * wave + r under filter row kh
          if (kh < 0 || kh > 2) continue;
#pragma unroll
          for (int kw = 0; kw < 3; ++kw)
#pragma unroll
            for (int j = 0; j < TNi; ++j) {
              const u32x4 vb = {fb[h & 1][kw][j][0].x, fb[h & 1][kw][j][0].y, fb[h & 1][kw][j][1].x, fb[h & 1][kw][j][1].y};
#pragma unroll
              for (int i = 0; i < TMo; ++i) {
                const u32x4 va = {fa[r][i][0].x, fa[r][i][0].y, fa[r][i][1].x, fa[r][i][1].y};
                acc[kh * 3 + kw][i][j] = mfma16_16x16x32(va, vb, acc[kh * 3 + kw][i][j]);
              }
            }
        }
      });
    }
  }

  // ---- the four waves' partial sums meet in LDS, one filter row per round (fixed order: wave 0..3); C layout row (co) = lg*4 + r, col (ci) = lr ----
  float* out = a.slabs + (size_t)blockIdx.x * a.Cout * (9 * a.ctot);
  f32x4* rbuf = reinterpret_cast<f32x4*>(smem);              // [4 waves][F3][64 lanes]
  __builtin_amdgcn_s_waitcnt(0x0f70);                         // vmcnt(0): the last (idle) LDS-DMA has written its zeros - rbuf reuses that memory
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    lds_barrier();                                            // staging buffers / the previous round are dead
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
      for (int i = 0; i < TMo; ++i)
#pragma unroll
        for (int j = 0; j < TNi; ++j) rbuf[(wave * F3 + (kw * TMo + i) * TNi + j) * 64 + lane] = acc[kh * 3 + kw][i][j];
    lds_barrier();
    for (int f = wave; f < F3; f += 4) {
      const f32x4 s01 = rbuf[(0 * F3 + f) * 64 + lane] + rbuf[(1 * F3 + f) * 64 + lane];
      const f32x4 s23 = rbuf[(2 * F3 + f) * 64 + lane] + rbuf[(3 * F3 + f) * 64 + lane];
      const f32x4 v = s01 + s23;
      const int kw = f / (TMo * TNi), i = (f / TNi) % TMo, j = f % TNi;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = i * 16 + lg * 4 + r;
        if (co < a.Cout) out[(size_t)co * (9 * a.ctot) + (kh * 3 + kw) * a.ctot + a.coff + j * 16 + lr] = v[r];
      }
    }
  }
}

template <int CIN, int COUT, bool PBN, bool UP>
static int launch_scwl(const ScWgArgs& a, int blocks, hipStream_t s) {
  constexpr int NPX = ((UP ? (SC_TH / 2 + 2) * (SC_TW / 2 + 2) : SC_HH * SC_HW) * (CIN / 8) + 255) / 256, NPD = (SC_TH * SC_TW * (COUT / 8) + 255) / 256;
  const size_t lds = (size_t)2 * (NPX + NPD) * 4096;
  static bool attr_set = false;
  if (lds > 64 * 1024 && !attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_sc_wgrad_lean_kernel<CIN, COUT, PBN, UP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return STP_E_LAUNCH;
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_sc_wgrad_lean_kernel<CIN, COUT, PBN, UP>), dim3(blocks), dim3(256), lds, s, a);
  STP_LAUNCH_CHECK();
  return STP_OK;
}

static bool sc_wg_lean_3232() {
  static const bool on = getenv("STP_SC_WG_LEAN_3232") && atoi(getenv("STP_SC_WG_LEAN_3232")) != 0;
  return on;
}

template <bool PBN, bool UP>
static int dispatch_scwl(const ScWgArgs& a, int cin, int cout, int blocks, hipStream_t s) {
  switch (cin * 64 + cout) {
    case 16 * 64 + 8: return launch_scwl<16, 8, PBN, UP>(a, blocks, s);
    case 16 * 64 + 16: return launch_scwl<16, 16, PBN, UP>(a, blocks, s);
    case 16 * 64 + 32: return launch_scwl<16, 32, PBN, UP>(a, blocks, s);
    case 32 * 64 + 8: return launch_scwl<32, 8, PBN, UP>(a, blocks, s);
    case 32 * 64 + 16: return launch_scwl<32, 16, PBN, UP>(a, blocks, s);
    // (32 -> 32 at full resolution stays with the generic kernel: 80 KB of staging = two workgroups per CU, measured 54 us against 58 us here)
    case 32 * 64 + 32: if (UP || sc_wg_lean_3232()) return launch_scwl<32, 32, PBN, UP>(a, blocks, s); else return 1;
    default: return 1;
  }
}

// 1 = not served (the caller falls back to the generic streaming kernel)
int sc_wg_lean_launch(const ScWgArgs& a, int cin, int cout, int dtype, int blocks, hipStream_t s) {
  if (!sc_lean_on() || !sc_stream_on() || dtype != STP_H16 || !a.src_bytes || !a.dy_bytes) return 1;
  if (a.up) return a.pbn.mean ? dispatch_scwl<true, true>(a, cin, cout, blocks, s) : dispatch_scwl<false, true>(a, cin, cout, blocks, s);
  return a.pbn.mean ? dispatch_scwl<true, false>(a, cin, cout, blocks, s) : dispatch_scwl<false, false>(a, cin, cout, blocks, s);
}

// =================================================================================================
// PERSISTENT form of the stem kernel (7x7 / stride 2 / pad 3, 4 padded input channels -> 64; conv_stem_kernel in conv_sc.hip is the
// single-shot form): there every one of the 4096 workgroups of the headline launch copied the 29 KB weight matrix and its 11.8 KB input
// patch through registers into LDS, synchronised, multiplied and retired - 79 us for a layer that moves 168 MB (25 us at HBM speed).
// Here workgroups are persistent (two per CU): the weights go to LDS once per workgroup, the input patch of the NEXT 8 x 32 output
// tile is written into the other half of a double buffer by LDS-DMA while this one is multiplied, interior tiles take their DMA
// offsets as (tile base) + (per-lane constant), all LDS reads of the tile loop are inline asm (the compiler would put the next
// tile's DMA in front of the first one), the fused statistics run over all tiles of the workgroup in registers (one column per
// workgroup instead of one per tile: 512 instead of 4096 columns for the finalize).
// Patch geometry: 16-byte DMA vectors = two 8-byte pixels, so the patch starts at the EVEN column 2 x0 - 4 (the single-shot form starts
// at 2 x0 - 3): 21 rows x 35 vectors; a B fragment (two horizontally adjacent taps x 4 channels) then sits at an odd pixel, 8-byte
// aligned: two ds_read_b64.  Same MFMA order as the single-shot kernel: bit-identical outputs.
// =================================================================================================
constexpr int STL_VPR = 35, STL_NV = ST_HH * STL_VPR, STL_NPASS = (STL_NV + 255) / 256, STL_BUF = STL_NPASS * 4096, STL_ROW = STL_VPR * 16;
constexpr int STL_LDS = ST_WBYTES + 2 * STL_BUF + 4 * 64 * 2 * 4;

__global__ __launch_bounds__(256, 2) void conv_stem_lean_kernel(const StemArgs a) {
  constexpr int TM = 4, NCH = 7, K = 224, COUT = 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];      // [weights 64 x 464][patch 0][patch 1][statistics scratch]
  char* const wl = smem;
  char* const hb = smem + ST_WBYTES;
  float* const red = reinterpret_cast<float*>(smem + ST_WBYTES + 2 * STL_BUF);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lg = lane >> 4;

  const int ntiles = a.N * a.tiles_x * a.tiles_y;
  int t_first, t_step, t_end;
  if ((gridDim.x & 7) == 0) {
    const int q = ntiles >> 3, r = ntiles & 7, x = blockIdx.x & 7;
    const int start = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    t_first = start + (int)(blockIdx.x >> 3); t_step = (int)(gridDim.x >> 3); t_end = start + q + (x < r ? 1 : 0);
  } else {
    t_first = (int)blockIdx.x; t_step = (int)gridDim.x; t_end = ntiles;
  }
  f32x4 ss[TM], qq[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) { ss[i] = f32x4{0.f, 0.f, 0.f, 0.f}; qq[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  if (t_first < t_end) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.src, 0, a.src_bytes, 0x00020000);
    int hyx[STL_NPASS];
    uint32_t lo[STL_NPASS];      // (hy * W + 2 vx) * 8: byte offset relative to the patch's first pixel; past the patch: 2^31
#pragma unroll
    for (int p = 0; p < STL_NPASS; ++p) {
      const int v = p * 256 + tid, hy = v / STL_VPR, vx = v - hy * STL_VPR;
      hyx[p] = v < STL_NV ? (hy << 16 | vx) : -1;
      lo[p] = v < STL_NV ? (uint32_t)((hy * a.W + 2 * vx) * 8) : 0x80000000u;
    }
    auto decode = [&](int tile) -> TileC {
      const int bq = (int)fdiv((uint32_t)tile, a.divTx);
      const int tx = tile - bq * a.tiles_x;
      const int n = (int)fdiv((uint32_t)bq, a.divTy);
      const int ty = bq - n * a.tiles_y;
      TileC t;
      t.n = n; t.y0 = ty * SC_TH; t.x0 = tx * SC_TW;
      // the whole patch (rows 2 y0 - 3 .. + 20, columns 2 x0 - 4 .. + 69) and the whole output tile lie inside the images
      t.interior = 2 * t.y0 >= 3 && 2 * t.x0 >= 4 && 2 * t.y0 + 17 < a.H && 2 * t.x0 + 65 < a.W && t.y0 + SC_TH <= a.Ho && t.x0 + SC_TW <= a.Wo;
      return t;
    };
    auto issue_tile = [&](const TileC& t, int b, bool live) {
      if (live && t.interior) {
        const uint32_t tb = (uint32_t)(((t.n * a.H + 2 * t.y0 - 3) * a.W + 2 * t.x0 - 4) * 8);
#pragma unroll
        for (int p = 0; p < STL_NPASS; ++p)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(hb + b * STL_BUF + p * 4096 + wave * 1024), 16, (int)(tb + lo[p]), 0, 0, 0);
      } else {
#pragma unroll
        for (int p = 0; p < STL_NPASS; ++p) {
          const int gy = 2 * t.y0 - 3 + (hyx[p] >> 16), gx = 2 * t.x0 - 4 + 2 * (hyx[p] & 0xffff);      // (W is even: a vector is inside or outside as a whole)
          const bool ok = live && hyx[p] >= 0 && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
          const uint32_t off = ok ? (uint32_t)(((t.n * a.H + gy) * a.W + gx) * 8) : 0x80000000u;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(hb + b * STL_BUF + p * 4096 + wave * 1024), 16, (int)off, 0, 0, 0);
        }
      }
    };
    TileC tc = decode(t_first);
    issue_tile(tc, 0, true);
    // once per workgroup: the weights [64][7][8][4] -> LDS, FRAGMENT-major: vector ((c * 4 + g) * 64 + co) = weight[co][chunk c][lane group g]
    // (an A fragment read = 64 consecutive vectors: conflict-free for the lane groups of ds_read_b128; rows of 464 bytes, the single-shot
    // kernel's layout, cost one extra LDS cycle in every group: SQ_LDS_BANK_CONFLICT = 48 % of SQ_LDS_IDX_ACTIVE)
    // Row slot of output channel co: MFMA block i = (co >> 5) * 2 + ((co >> 2) & 1), row m = ((co >> 3) & 3) * 4 + (co & 3) - a lane (pixel lr,
    // lane group lg) then holds channels (i >> 1) * 32 + lg * 8 + (i & 1) * 4 + r: blocks 2h, 2h + 1 are EIGHT consecutive channels, one
    // 16-byte store, and the four lanes of a pixel write 64 contiguous bytes per instruction (32 with the natural order)
    for (int v = tid; v < COUT * (K / 8); v += 256) {
      const int co = v / (K / 8), q = v - co * (K / 8);      // q = c * 4 + g
      const int slot = ((co >> 5) * 2 + ((co >> 2) & 1)) * 16 + ((co >> 3) & 3) * 4 + (co & 3);
      *reinterpret_cast<u32x4*>(wl + (q * COUT + slot) * 16) = *reinterpret_cast<const u32x4*>(a.weight + ((size_t)co * K + q * 8) * 2);
    }
    // lane addresses: A fragment (channel block i, chunk c) = wa + (c * 256 + i * 16) vectors; B fragment (tile row 2 wave + (f >> 1), pixel
    // (f & 1) * 16 + lr, chunk c = kh, taps kw = 2 lg, 2 lg + 1) = patch row 2 py + c, pixel 2 px + 2 lg + 1 (the patch starts at column 2 x0 - 4)
    const uint32_t wa = (uint32_t)(uintptr_t)wl + (uint32_t)((lg * COUT + lr) * 16);
    uint32_t ba = (uint32_t)(uintptr_t)hb + (uint32_t)((4 * wave) * STL_ROW + (2 * lr + 2 * lg + 1) * 8);
    // per-lane byte offset of the lane's first output inside a tile: pixel (2 wave, lr), channels lg * 4 ..
    const uint32_t so = (uint32_t)(((wave * 2) * a.Wo + lr) * (COUT * 2) + lg * 16);
#if defined(STP_STEM_WAIT_LATE)
    __builtin_amdgcn_s_waitcnt(0x0f70);
    asm volatile("" ::: "memory");
#endif

    int cur = 0;
    bool prev_full = false;
    for (int tile = t_first; tile < t_end; tile += t_step, cur ^= 1) {
      const TileC t = tc;
#if defined(STP_STEM_COUNTED)
      // (what-if) the previous tile's 16 stores - issued AFTER this tile's LDS-DMA - stay in flight when that tile was an interior one
      if (prev_full) __builtin_amdgcn_s_waitcnt(0x4f70);    // vmcnt(16)
      else __builtin_amdgcn_s_waitcnt(0x0f70);
      asm volatile("" ::: "memory");
      prev_full = t.interior;
#elif !defined(STP_STEM_WAIT_LATE)
      __builtin_amdgcn_s_waitcnt(0x0f70);                   // vmcnt(0): this tile's pieces (own) and the weights have landed
      asm volatile("" ::: "memory");
#endif
      lds_barrier();
      const int next = tile + t_step;
      const bool live = next < t_end;
      if (live) tc = decode(next);
      issue_tile(tc, cur ^ 1, live);

      f32x4 acc[TM][4];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int f = 0; f < 4; ++f) acc[i][f] = f32x4{0.f, 0.f, 0.f, 0.f};
      // chunk c + 1's twelve reads are in flight while chunk c's sixteen MFMAs issue
      u32x4 fa[2][TM];
      u32x2 fb[2][4][2];
      auto reads = [&fa, &fb, wa, ba](auto cc) {
        constexpr int c = decltype(cc)::value;
        sc_unroll<TM>([&fa, wa](auto ic) {
          constexpr int i = decltype(ic)::value;
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[c & 1][i]) : "v"(wa), "n"((c * 4 * 64 + i * 16) * 16));
        });
        sc_unroll<8>([&fb, ba](auto kc) {
          constexpr int k = decltype(kc)::value, f = k >> 1, hl = k & 1;
          asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(fb[c & 1][f][hl]) : "v"(ba), "n"((2 * (f >> 1) + c) * STL_ROW + (f & 1) * 32 * 8 + hl * 8));
        });
      };
      reads(std::integral_constant<int, 0>{});
      sc_unroll<NCH>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        if constexpr (c + 1 < NCH) reads(std::integral_constant<int, c + 1>{});
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(c + 1 < NCH ? 12 : 0) : "memory");
#pragma unroll
        for (int i = 0; i < TM; ++i) asm volatile("" : "+v"(fa[c & 1][i]));
#pragma unroll
        for (int f = 0; f < 4; ++f) { asm volatile("" : "+v"(fb[c & 1][f][0])); asm volatile("" : "+v"(fb[c & 1][f][1])); }
#pragma unroll
        for (int f = 0; f < 4; ++f) {
          const u32x4 vb = {fb[c & 1][f][0].x, fb[c & 1][f][0].y, fb[c & 1][f][1].x, fb[c & 1][f][1].y};
#pragma unroll
          for (int i = 0; i < TM; ++i) acc[i][f] = mfma16_16x16x32(fa[c & 1][i], vb, acc[i][f]);
        }
      });
      ba += cur ? (uint32_t)(-STL_BUF) : (uint32_t)STL_BUF;      // the B addresses move to the other buffer half

#if defined(STP_STEM_WAIT_LATE)
      // (what-if) the next tile's pieces are waited for HERE, a whole MFMA phase after their issue and BEFORE this tile's stores: the
      // stores then drain under the next tile's MFMAs instead of in front of its barrier
      __builtin_amdgcn_s_waitcnt(0x0f70);
      asm volatile("" ::: "memory");
#endif
      // epilogue: store, statistics of the stored values
      char* ob = a.dst + (((size_t)t.n * a.Ho + t.y0) * a.Wo + t.x0) * (COUT * 2);
      const bool st = a.stats != nullptr;
      auto epilogue = [&](const bool EDGE) __attribute__((always_inline)) {
#pragma unroll
        for (int f = 0; f < 4; ++f) {
          char* of = ob + (size_t)(((f >> 1) * a.Wo + (f & 1) * 16) * (COUT * 2));
          const bool ok = !EDGE || (t.y0 + wave * 2 + (f >> 1) < a.Ho && t.x0 + (f & 1) * 16 + lr < a.Wo);
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const f32x4 v0 = acc[2 * h][f], v1 = acc[2 * h + 1][f];
            const u32x4 o = {pack_bf16x2(v0.x, v0.y), pack_bf16x2(v0.z, v0.w), pack_bf16x2(v1.x, v1.y), pack_bf16x2(v1.z, v1.w)};
            if (ok) {
              *reinterpret_cast<u32x4*>(of + so + h * 64) = o;
              if (st) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                  const f32x2 s0 = scl_unpack(o[2 * u]), s1 = scl_unpack(o[2 * u + 1]);
                  const f32x4 sv = {s0.x, s0.y, s1.x, s1.y};
                  ss[2 * h + u] += sv;
                  qq[2 * h + u] += sv * sv;
                }
              }
            }
          }
        }
      };
      if (t.interior) epilogue(false); else epilogue(true);
    }
  }
  // one column of [stat][channel][workgroups] per workgroup (a workgroup without tiles contributes zeros)
  if (a.stats) {
    __builtin_amdgcn_s_waitcnt(0x0f70);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float sv = row_sum16_to_lane15(ss[i][e]), qv = row_sum16_to_lane15(qq[i][e]);
        if (lr == 15) {
          const int cl = (i >> 1) * 32 + lg * 8 + (i & 1) * 4 + e;      // (the channel order of the row slots, see the weight staging)
          red[(wave * COUT + cl) * 2] = sv;
          red[(wave * COUT + cl) * 2 + 1] = qv;
        }
      }
    lds_barrier();
    if (tid < COUT) {
      float sv = 0.f, qv = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) { sv += red[(w * COUT + tid) * 2]; qv += red[(w * COUT + tid) * 2 + 1]; }
      a.stats[(size_t)tid * gridDim.x + blockIdx.x] = sv;
      a.stats[((size_t)COUT + tid) * gridDim.x + blockIdx.x] = qv;
    }
  }
}

bool stem_lean_serves(int N, int H, int W) {
  static const bool on = !(getenv("STP_STEM_LEAN") && atoi(getenv("STP_STEM_LEAN")) == 0);
  return on && !(W & 1) && (uint64_t)N * H * W * 8 < 0x80000000ull;
}
int stem_lean_blocks(int ntiles) {
  const int64_t b = (int64_t)sc_cu_count() * 2;
  return (int)(b < ntiles ? b : ntiles);
}
int stem_lean_launch(const StemArgs& a, hipStream_t s) {
  if (!stem_lean_serves(a.N, a.H, a.W) || !a.src_bytes) return 1;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_stem_lean_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)STL_LDS) != hipSuccess)
      return STP_E_LAUNCH;
    attr_set = true;
  }
  hipLaunchKernelGGL(conv_stem_lean_kernel, dim3(stem_lean_blocks(a.N * a.tiles_x * a.tiles_y)), dim3(256), STL_LDS, s, a);
  STP_LAUNCH_CHECK();
  return STP_OK;
}

// =================================================================================================
// Stem WEIGHT GRADIENT (7x7 / stride 2 / pad 3, input 4 padded channels, 64 output channels):
//   dW[co][kh][kw (8)][c (4)] = sum over output pixels of dY[p][co] * X[2 py - 3 + kh][2 px - 3 + kw][c]
// The generic pixel-reduction GEMM gathers its 7 x 8 x 4 patch columns pixel by pixel through the vector-memory path: 96-100 us for a
// layer that reads 168 MB (25 us at HBM speed; SQ_ACTIVE_INST_ANY / SQ_BUSY_CYCLES 9.8, MFMA 14 % busy).  Here, with the recipe of the
// kernels above: persistent workgroups walk 4 x 32-pixel output tiles; the 13 x 70-pixel input patch (16-byte vectors = two pixels,
// starting at the even column 2 x0 - 4 as in conv_stem_lean_kernel) and the dY tile go to a double buffer by LDS-DMA.  Reduction index =
// the 32 pixels of a tile row, fragments by transpose reads: A = dY^T (wave w owns output channels 16 w .. 16 w + 15: no cross-wave
// reduction), B = 32 pixels x 16 columns, the 16 columns being FOUR taps kw x 4 channels = 32 contiguous bytes of the patch.  A B
// fragment of patch row r serves every (tile row, kh) pair with 2 * row + kh = r: 52 + 8 transpose reads for 56 MFMAs per tile and wave.
// One fp32 slab [64][224] per workgroup, summed by the generic fixed-order reduce kernel (the small-channel plan of conv_wgrad.hip).
// =================================================================================================
constexpr int SWG_TH = 4, SWG_PH = 2 * (SWG_TH - 1) + 7;                      // tile rows, patch rows (13)
constexpr int SWG_NVX = SWG_PH * STL_VPR, SWG_NPX = (SWG_NVX + 255) / 256;    // patch vectors / passes
constexpr int SWG_NVD = SWG_TH * SC_TW * 8, SWG_NPD = (SWG_NVD + 255) / 256;  // dY vectors (64 channels = 8 per pixel) / passes
constexpr int SWG_XBUF = SWG_NPX * 4096, SWG_BUF = SWG_XBUF + SWG_NPD * 4096;

struct StemWgArgs {
  const char* src;   // [N,H,W,4]
  const char* dy;    // [N,Ho,Wo,64]
  float* slabs;      // [workgroups][64][224]
  int N, H, W, Ho, Wo, tiles_x, tiles_y, ntiles;
  uint32_t src_bytes, dy_bytes;
};

__global__ __launch_bounds__(256, 2) void conv_stem_wgrad_lean_kernel(const StemWgArgs a) {
  constexpr int COUT = 64, K = 224, NF = 14;             // accumulator fragments of a wave: (kh, kw half)
  extern __shared__ __attribute__((aligned(16))) char smem[];      // [2][patch | dY tile]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lg = lane >> 4;

  int t_first, t_step, t_end;
  if ((gridDim.x & 7) == 0) {
    const int q = a.ntiles >> 3, r = a.ntiles & 7, x = blockIdx.x & 7;
    const int start = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    t_first = start + (int)(blockIdx.x >> 3); t_step = (int)(gridDim.x >> 3); t_end = start + q + (x < r ? 1 : 0);
  } else {
    t_first = (int)blockIdx.x; t_step = (int)gridDim.x; t_end = a.ntiles;
  }
  f32x4 acc[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) acc[f] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (t_first < t_end) {
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)a.src, 0, a.src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, a.dy_bytes, 0x00020000);
    int hyx[SWG_NPX], pyx[SWG_NPD];
    uint32_t lox[SWG_NPX], lod[SWG_NPD];
#pragma unroll
    for (int p = 0; p < SWG_NPX; ++p) {
      const int v = p * 256 + tid, hy = v / STL_VPR, vx = v - hy * STL_VPR;
      hyx[p] = v < SWG_NVX ? (hy << 16 | vx) : -1;
      lox[p] = v < SWG_NVX ? (uint32_t)((hy * a.W + 2 * vx) * 8) : 0x80000000u;
    }
#pragma unroll
    for (int p = 0; p < SWG_NPD; ++p) {
      const int v = p * 256 + tid, pix = v >> 3, py = pix / SC_TW, px = pix % SC_TW;
      pyx[p] = v < SWG_NVD ? (py << 16 | px) : -1;
      lod[p] = v < SWG_NVD ? (uint32_t)((py * a.Wo + px) * (COUT * 2) + (v & 7) * 16) : 0x80000000u;
    }
    const uint32_t cvd = (uint32_t)(tid & 7) * 16u;
    auto decode = [&](int tile) -> TileC {
      int b = tile;
      const int tx = b % a.tiles_x; b /= a.tiles_x;
      const int ty = b % a.tiles_y;
      TileC t;
      t.n = b / a.tiles_y; t.y0 = ty * SWG_TH; t.x0 = tx * SC_TW;
      t.interior = 2 * t.y0 >= 3 && 2 * t.x0 >= 4 && 2 * t.y0 - 3 + SWG_PH <= a.H && 2 * t.x0 + 65 < a.W && t.y0 + SWG_TH <= a.Ho && t.x0 + SC_TW <= a.Wo;
      return t;
    };
    auto issue_tile = [&](const TileC& t, int bsel, bool live) {
      char* xb = smem + bsel * SWG_BUF;
      if (live && t.interior) {
        const uint32_t tbx = (uint32_t)(((t.n * a.H + 2 * t.y0 - 3) * a.W + 2 * t.x0 - 4) * 8);
        const uint32_t tbd = (uint32_t)(((t.n * a.Ho + t.y0) * a.Wo + t.x0) * (COUT * 2));
#pragma unroll
        for (int p = 0; p < SWG_NPX; ++p)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (__attribute__((address_space(3))) void*)(xb + p * 4096 + wave * 1024), 16, (int)(tbx + lox[p]), 0, 0, 0);
#pragma unroll
        for (int p = 0; p < SWG_NPD; ++p)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsd, (__attribute__((address_space(3))) void*)(xb + SWG_XBUF + p * 4096 + wave * 1024), 16, (int)(tbd + lod[p]), 0, 0, 0);
      } else {
#pragma unroll
        for (int p = 0; p < SWG_NPX; ++p) {
          const int gy = 2 * t.y0 - 3 + (hyx[p] >> 16), gx = 2 * t.x0 - 4 + 2 * (hyx[p] & 0xffff);      // (W is even: a vector is inside or outside as a whole)
          const bool ok = live && hyx[p] >= 0 && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
          const uint32_t off = ok ? (uint32_t)(((t.n * a.H + gy) * a.W + gx) * 8) : 0x80000000u;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (__attribute__((address_space(3))) void*)(xb + p * 4096 + wave * 1024), 16, (int)off, 0, 0, 0);
        }
#pragma unroll
        for (int p = 0; p < SWG_NPD; ++p) {
          const int gy = t.y0 + (pyx[p] >> 16), gx = t.x0 + (pyx[p] & 0xffff);
          const bool ok = live && pyx[p] >= 0 && gy < a.Ho && gx < a.Wo;
          const uint32_t off = ok ? (uint32_t)(((t.n * a.Ho + gy) * a.Wo + gx) * (COUT * 2)) + cvd : 0x80000000u;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsd, (__attribute__((address_space(3))) void*)(xb + SWG_XBUF + p * 4096 + wave * 1024), 16, (int)off, 0, 0, 0);
        }
      }
    };
    TileC tc = decode(t_first);
    issue_tile(tc, 0, true);
    // lane group g owns pixels x = 4g..4g+3 (lo) and 16+4g..16+4g+3 (hi) of a 32-pixel tile row; lane i of a group addresses pixel (i>>2), quad (i&3).
    // A (dY^T, channels 16 wave ..): quad = 4 channels.  B: quad = tap kw (4 channels = 8 bytes); output pixel x, tap kw (of the fragment's
    // four) sit at patch pixel 2 x + 1 + kw (the patch starts at column 2 x0 - 4), the upper four taps 32 bytes further
    const int xl = lg * 4 + (lr >> 2), qd = lr & 3;
    uint32_t aA = (uint32_t)(uintptr_t)smem + (uint32_t)(SWG_XBUF + (xl * COUT + wave * 16) * 2 + qd * 8);
    uint32_t aB = (uint32_t)(uintptr_t)smem + (uint32_t)((2 * xl + 1 + qd) * 8);

    int cur = 0;
    for (int tile = t_first; tile < t_end; tile += t_step, cur ^= 1) {
      __builtin_amdgcn_s_waitcnt(0x0f70);                   // vmcnt(0): own pieces of this tile have landed
      asm volatile("" ::: "memory");
      lds_barrier();
      const int next = tile + t_step;
      const bool live = next < t_end;
      if (live) tc = decode(next);
      issue_tile(tc, cur ^ 1, live);

      u32x2 fa[SWG_TH][2];          // dY fragments of the four tile rows: [row][lo / hi pixels]
      u32x2 fb[2][2][2];            // X fragments of one patch row, double-buffered: [buffer][kw half][lo / hi pixels]
      sc_unroll<SWG_TH * 2>([&fa, aA](auto kc) {
        constexpr int k = decltype(kc)::value, r = k >> 1, hl = k & 1;
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(fa[r][hl]) : "v"(aA), "n"((r * SC_TW + hl * 16) * COUT * 2));
      });
      auto issue_b = [&fb, aB](auto rc) {
        constexpr int pr = decltype(rc)::value;
        sc_unroll<4>([&fb, aB](auto kc) {
          constexpr int k = decltype(kc)::value, kwh = k >> 1, hl = k & 1;
          asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(fb[pr & 1][kwh][hl]) : "v"(aB), "n"(pr * STL_ROW + kwh * 32 + hl * 32 * 8));
        });
      };
      issue_b(std::integral_constant<int, 0>{});
      sc_unroll<SWG_PH>([&](auto rc) {
        constexpr int pr = decltype(rc)::value;           // patch row: serves (tile row r, kh) with 2 r + kh == pr
        if constexpr (pr + 1 < SWG_PH) issue_b(std::integral_constant<int, pr + 1>{});
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(pr + 1 < SWG_PH ? 4 : 0) : "memory");
        if constexpr (pr == 0) {
#pragma unroll
          for (int r = 0; r < SWG_TH; ++r) { asm volatile("" : "+v"(fa[r][0])); asm volatile("" : "+v"(fa[r][1])); }
        }
#pragma unroll
        for (int kwh = 0; kwh < 2; ++kwh) { asm volatile("" : "+v"(fb[pr & 1][kwh][0])); asm volatile("" : "+v"(fb[pr & 1][kwh][1])); }
#pragma unroll
        for (int r = 0; r < SWG_TH; ++r) {
          const int kh = pr - 2 * r;
          if (kh < 0 || kh > 6) continue;
          const u32x4 va = {fa[r][0].x, fa[r][0].y, fa[r][1].x, fa[r][1].y};
#pragma unroll
          for (int kwh = 0; kwh < 2; ++kwh) {
            const u32x4 vb = {fb[pr & 1][kwh][0].x, fb[pr & 1][kwh][0].y, fb[pr & 1][kwh][1].x, fb[pr & 1][kwh][1].y};
            acc[kh * 2 + kwh] = mfma16_16x16x32(va, vb, acc[kh * 2 + kwh]);
          }
        }
      });
      const uint32_t flip = cur ? (uint32_t)(-SWG_BUF) : (uint32_t)SWG_BUF;
      aA += flip; aB += flip;
    }
  }
  // slab: C layout row (channel) = lg*4 + r, column = lr = (tap of the half, input channel): k = kh * 32 + kw half * 16 + lr
  float* out = a.slabs + (size_t)blockIdx.x * COUT * K;
#pragma unroll
  for (int f = 0; f < NF; ++f)
#pragma unroll
    for (int r = 0; r < 4; ++r) out[(size_t)(wave * 16 + lg * 4 + r) * K + (f >> 1) * 32 + (f & 1) * 16 + lr] = acc[f][r];
}

bool stem_wg_lean_serves(int N, int H, int W, int Ho, int Wo) {
  static const bool on = !(getenv("STP_STEM_WG_LEAN") && atoi(getenv("STP_STEM_WG_LEAN")) == 0);
  return on && !(W & 1) && (uint64_t)N * H * W * 8 < 0x80000000ull && (uint64_t)N * Ho * Wo * 128 < 0x80000000ull;
}
int stem_wg_lean_blocks(int N, int Ho, int Wo) {
  const int64_t tiles = (int64_t)N * ((Ho + SWG_TH - 1) / SWG_TH) * ((Wo + SC_TW - 1) / SC_TW), b = (int64_t)sc_cu_count() * 2;
  return (int)(b < tiles ? b : tiles);
}
int stem_wg_lean_launch(const void* src, const void* dy, float* slabs, int N, int H, int W, int Ho, int Wo, hipStream_t s) {
  StemWgArgs a;
  a.src = (const char*)src; a.dy = (const char*)dy; a.slabs = slabs;
  a.N = N; a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo;
  a.tiles_x = (Wo + SC_TW - 1) / SC_TW; a.tiles_y = (Ho + SWG_TH - 1) / SWG_TH; a.ntiles = N * a.tiles_x * a.tiles_y;
  a.src_bytes = (uint32_t)((uint64_t)N * H * W * 8); a.dy_bytes = (uint32_t)((uint64_t)N * Ho * Wo * 128);
  hipLaunchKernelGGL(conv_stem_wgrad_lean_kernel, dim3(stem_wg_lean_blocks(N, Ho, Wo)), dim3(256), 2 * SWG_BUF, s, a);
  STP_LAUNCH_CHECK();
  return STP_OK;
}
