// "Small-channel" 3x3 / stride-1 / pad-1 convolution for the full-resolution decoder tail and the head
// (Cin <= 32, Cout <= 32): these layers are HBM-bound (AI 8..96 FLOP/B), and the generic
// implicit-GEMM gather re-reads every input pixel 9 times through the vector-memory path.
//
// Here a workgroup owns an 8 x 32 output tile: the (8+2) x (32+2) x Cin input halo tile is staged ONCE in
// LDS (coalesced 16-byte vectors, padding and nearest-2x upsampling resolved while staging), the whole
// weight matrix lives in registers as MFMA A fragments, and every B fragment is a single ds_read_b128
// from the halo tile at (pixel + tap offset): no im2col copy exists anywhere.  Per 16 output pixels:
// ceil(9*Cin/32) LDS reads and as many MFMAs per 16 output channels.  With ~11-22 KB of LDS per
// workgroup 7+ workgroups share a CU, which is what hides the load latency of this streaming kernel.
//
//   C[cout][pixel] = sum_k W[cout][k] * halo[pixel + tap(k)][ch(k)],   k = tap*Cin + ch, tap = kh*3 + kw
// The k -> (lane group, vector slot) assignment is the same for A and B (all a contraction needs).
// Used for forward and (with the flipped/transposed weight copy) for the data gradient.
#include "conv_sc.h"

__device__ __forceinline__ f32x4 sc_stored(f32x4 v, const float*) { return v; }
__device__ __forceinline__ f32x4 sc_stored(f32x4 v, const bf16_t*) {
  const uint32_t a = pack_bf16x2(v.x, v.y), b = pack_bf16x2(v.z, v.w);
  return f32x4{h16lo_to_f32(a), h16hi_to_f32(a), h16lo_to_f32(b), h16hi_to_f32(b)};
}

template <typename T> struct ScMma;
template <> struct ScMma<bf16_t> {
  __device__ static __forceinline__ void run(const u32x4& a, const u32x4& b, f32x4& c) {
    c = mfma16_16x16x32(a, b, c);
  }
};
template <> struct ScMma<float> {
  __device__ static __forceinline__ void run(const u32x4& a, const u32x4& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
  }
};

// 4 channels as fetched (8 bytes of bf16 stay packed until they are used: prefetched epilogue operands cost half the registers)
template <typename T> struct ScRaw4;
template <> struct ScRaw4<float> {
  f32x4 v;
  __device__ __forceinline__ void load(const float* p) { v = *reinterpret_cast<const f32x4*>(p); }
  __device__ __forceinline__ f32x4 get() const { return v; }
};
template <> struct ScRaw4<bf16_t> {
  u32x2 v;
  __device__ __forceinline__ void load(const bf16_t* p) { v = *reinterpret_cast<const u32x2*>(p); }
  __device__ __forceinline__ f32x4 get() const {
    return f32x4{h16lo_to_f32(v.x), h16hi_to_f32(v.x), h16lo_to_f32(v.y), h16hi_to_f32(v.y)};
  }
};


// Epilogue of the small-channel kernels, shared by the single-shot and the streaming form.  prefetch() issues the global
// operands of the fused BatchNormalization backward (the BN input x of every output this lane owns: clamped addresses, no
// control flow between the loads, one latency) - the streaming kernel calls it BEFORE it starts the next tile's LDS-DMA, so
// the vmcnt wait that releases these registers does not also wait for that tile; finish() does the arithmetic and the stores.
// LDSK: the per-channel constants of the fused BatchNormalization backward live in an LDS table [4][32] (scale, shift, mean,
// rstd; filled once per persistent workgroup) and are read where they are used instead of occupying 16 registers per 16 channels.
template <typename T, int TM, bool LDSK = false>
struct ScEpilogue {
  ScRaw4<T> xpre[TM][4];
  BnBackCh bks[LDSK ? 1 : TM];
  const float* ktab;
  // LDSK (persistent workgroup): the fused sums run over ALL tiles of the workgroup in these registers and are reduced once
  // (flush_stats) into column `workgroup` of [stat][channel][workgroups]; the slot form still adds per tile
  f32x4 ssp[LDSK ? TM : 1], qqp[LDSK ? TM : 1];
  // LDSK: the bias of the lane's channels, fetched ONCE per persistent workgroup (fill_table) - as a global load in finish() it was a
  // memory round trip (and a vmcnt(0), which also waits for the next tile's LDS-DMA) per tile
  f32x4 biasr[LDSK ? TM : 1];
  __device__ __forceinline__ void reset_stats() {
#pragma unroll
    for (int i = 0; i < (LDSK ? TM : 1); ++i) { ssp[i] = f32x4{0.f, 0.f, 0.f, 0.f}; qqp[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  }

  __device__ __forceinline__ void load_constants(const ScArgs& a, int lg) {
    if (a.bnb.x) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
        if (i * 16 + lg * 4 < a.Cout) bks[i] = bnback_load(a.bnb, i * 16 + lg * 4);
    }
  }
  // (all 256 threads) fill the LDS table; the caller's next barrier publishes it
  __device__ __forceinline__ void fill_table(const ScArgs& a, float* tab, int tid) {
    ktab = tab;
    if constexpr (LDSK) {
      const int lg = (tid & 63) >> 4;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = i * 16 + lg * 4 + r;
          biasr[i][r] = (a.bias && co < a.Cout) ? a.bias[co] : 0.f;
        }
    }
    if (a.bnb.x && tid < 32) {
      float sc = 0.f, sh = 0.f, mu = 0.f, rs = 0.f;
      if (tid < a.Cout) {
        mu = a.bnb.mean[tid]; rs = a.bnb.rstd[tid];
        sc = a.bnb.gamma ? rs * a.bnb.gamma[tid] : rs;
        sh = (a.bnb.beta ? a.bnb.beta[tid] : 0.f) - mu * sc;
      }
      tab[tid] = sc; tab[32 + tid] = sh; tab[64 + tid] = mu; tab[96 + tid] = rs;
    }
  }
  __device__ __forceinline__ BnBackCh bk(int i, int lg) const {
    if constexpr (LDSK) {
      // Inline asm: a ds_read the compiler knows about, issued while the next tile's LDS-DMA is in flight, gets an s_waitcnt vmcnt(0)
      // in front of it (the wait-count pass assumes the two may alias) - that put the whole DMA latency into every tile's epilogue.
      BnBackCh k;
      const uint32_t ad = (uint32_t)(uintptr_t)ktab + (uint32_t)(i * 16 + lg * 4) * 4u;
      asm volatile("ds_read_b128 %0, %1" : "=v"(k.sc) : "v"(ad));
      asm volatile("ds_read_b128 %0, %1 offset:128" : "=v"(k.sh) : "v"(ad));
      asm volatile("ds_read_b128 %0, %1 offset:256" : "=v"(k.mu) : "v"(ad));
      asm volatile("ds_read_b128 %0, %1 offset:384" : "=v"(k.rs) : "v"(ad));
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(k.sc), "+v"(k.sh), "+v"(k.mu), "+v"(k.rs));
      return k;
    } else {
      return bks[i];
    }
  }

  __device__ __forceinline__ void prefetch(const ScArgs& a, int n, int y0, int x0, int wave, int lr, int lg) {
    if (!a.bnb.x) return;
    if (a.sum2) {
      const int H2 = a.H >> 1, W2 = a.W >> 1;
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        const int gy = y0 + wave * 2, gx = x0 + h2 * 16 + lr;
        const size_t pm = ((size_t)n * H2 + (gy >> 1)) * W2 + (gx >> 1);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int co = i * 16 + lg * 4;
          const bool ok = !(lr & 1) && gy < a.H && gx < a.W && co < a.Cout;
          xpre[i][h2].load(reinterpret_cast<const T*>(a.bnb.x) + (ok ? pm * a.Cout + co : (size_t)0));
        }
      }
    } else {
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        const int gy = y0 + wave * 2 + (f >> 1), gx = x0 + (f & 1) * 16 + lr;
        const size_t pm = ((size_t)n * a.H + gy) * a.W + gx;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int co = i * 16 + lg * 4;
          const bool ok = gy < a.H && gx < a.W && co + 3 < a.Cout;
          xpre[i][f].load(reinterpret_cast<const T*>(a.bnb.x) + (ok ? pm * a.Cout + co : (size_t)0));
        }
      }
    }
  }

  // tile / ntiles: column of this tile in the [stat][channel][tile] partial sums
  __device__ __forceinline__ void finish(const ScArgs& a, f32x4 (&acc)[TM][4], float* red, int n, int y0, int x0, int tile, int ntiles, int tid,
                                         int wave, int lr, int lg) {
    T* out = reinterpret_cast<T*>(a.dst);
    f32x4 ss[TM], qq[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) { ss[i] = f32x4{0.f, 0.f, 0.f, 0.f}; qq[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    if (a.sum2) {
      // gradient of UpSampling2D(2): the wave's two tile rows are one output row (fragments f and f+2, same lane), lanes
      // lr and lr^1 one output column (quad_perm DPP); even lanes own the low-resolution pixel
      const int H2 = a.H >> 1, W2 = a.W >> 1;
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        const int gy = y0 + wave * 2, gx = x0 + h2 * 16 + lr;
        const bool own = !(lr & 1) && gy < a.H && gx < a.W;
        const size_t pm = ((size_t)n * H2 + (gy >> 1)) * W2 + (gx >> 1);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int co = i * 16 + lg * 4;
          f32x4 v = acc[i][h2] + acc[i][h2 + 2];
#pragma unroll
          for (int e = 0; e < 4; ++e)
            v[e] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[e]), 0xB1, 0xf, 0xf, true));  // quad_perm [1,0,3,2]
          if (!own || co >= a.Cout) continue;
          T* d = out + pm * a.Cout + co;
          if (a.accumulate) v += load4(d);
          if (a.bnb.x) v = bnback_apply(bk(i, lg), a.bnb.relu, xpre[i][h2].get(), sc_stored(v, (const T*)nullptr), ss[i], qq[i]);
#if defined(STP_EXP) && STP_EXP == 11
          if (a.N < 0)
#endif
          store4(d, v);
        }
      }
    } else {
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        const int gy = y0 + wave * 2 + (f >> 1), gx = x0 + (f & 1) * 16 + lr;
        if (gy >= a.H || gx >= a.W) continue;
        const size_t pm = ((size_t)n * a.H + gy) * a.W + gx;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int co = i * 16 + lg * 4;
          if (co >= a.Cout) continue;
          f32x4 v = acc[i][f];
          if (co + 3 < a.Cout) {
            if (a.bias) {
              if constexpr (LDSK) v += biasr[i]; else v += *reinterpret_cast<const f32x4*>(a.bias + co);
            }
            T* d = out + pm * a.Cout + co;
            if (a.accumulate) v += load4(d);
            if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            if (a.bnb.x) v = bnback_apply(bk(i, lg), a.bnb.relu, xpre[i][f].get(), sc_stored(v, (const T*)nullptr), ss[i], qq[i]);
#if defined(STP_EXP) && STP_EXP == 11
            if (a.N < 0)
#endif
            store4(d, v);
            if (a.stats && !a.bnb.x) {
              const f32x4 sv = sc_stored(v, (const T*)nullptr);
              ss[i] += sv;
              qq[i] += sv * sv;
            }
          } else {
            for (int r = 0; r < 4 && co + r < a.Cout; ++r) {
              float x = v[r];
              if (a.bias) {
                if constexpr (LDSK) x += biasr[i][r]; else x += a.bias[co + r];
              }
              T* d = out + pm * a.Cout + co + r;
              if (a.accumulate) x += Elem<T>::load(d);
              if (a.relu) x = fmaxf(x, 0.f);
#if defined(STP_EXP) && STP_EXP == 11
              if (a.N < 0)
#endif
              Elem<T>::store(d, x);
            }
          }
        }
      }
    }
    if constexpr (LDSK) {
      if (a.stats && !a.stat_slots) {     // keep summing; flush_stats() reduces once per workgroup
#pragma unroll
        for (int i = 0; i < TM; ++i) { ssp[i] += ss[i]; qqp[i] += qq[i]; }
        return;
      }
    }
    reduce_stats(a, ss, qq, red, tile, ntiles, tid, wave, lr, lg);
  }

  __device__ __forceinline__ void flush_stats(const ScArgs& a, float* red, int column, int columns, int tid, int wave, int lr, int lg) {
    if constexpr (LDSK) {
      if (a.stats && !a.stat_slots) reduce_stats(a, ssp, qqp, red, column, columns, tid, wave, lr, lg);
    }
  }

  template <int NS>
  __device__ __forceinline__ void reduce_stats(const ScArgs& a, f32x4 (&ss)[NS], f32x4 (&qq)[NS], float* red, int tile, int ntiles, int tid,
                                               int wave, int lr, int lg) {
    if (a.stats) {
      // butterfly over the 16 pixel lanes, then the 4 waves (same channels, different rows) through LDS
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float sv = row_sum16_to_lane15(ss[i][e]), qv = row_sum16_to_lane15(qq[i][e]);
          if (lr == 15) {
            const int cl = i * 16 + lg * 4 + e;
            red[(wave * TM * 16 + cl) * 2] = sv;
            red[(wave * TM * 16 + cl) * 2 + 1] = qv;
          }
        }
      lds_barrier();
      if (tid < TM * 16 && tid < a.Cout) {
        float sv = 0.f, qv = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) { sv += red[(w * TM * 16 + tid) * 2]; qv += red[(w * TM * 16 + tid) * 2 + 1]; }
        if (a.stat_slots) {
          long long* sl = reinterpret_cast<long long*>(a.stats);
          slot_add(sl, a.stat_slots, tid, tile, sv);
          slot_add(sl, a.stat_slots, a.Cout + tid, tile, qv);
        } else {
          a.stats[(size_t)tid * ntiles + tile] = sv;                     // [stat][channel][tile]
          a.stats[((size_t)a.Cout + tid) * ntiles + tile] = qv;
        }
      }
    }
  }
};

template <typename T, int CIN, int TM>
__global__ __launch_bounds__(256) void conv_sc_kernel(const ScArgs a) {
  constexpr int SZ = (int)sizeof(T);
  constexpr int VEC = Elem<T>::VEC;
  constexpr int KC = 4 * VEC;                    // k values per MFMA chunk (32 bf16 / 16 fp32)
  constexpr int K = 9 * CIN;
  constexpr int NCH = (K + KC - 1) / KC;
  constexpr int VPP = CIN / VEC;                 // 16-byte vectors per pixel
  constexpr int PIXB = CIN * SZ;                 // bytes per pixel in the halo tile
  static_assert(CIN % VEC == 0, "CIN must be a multiple of the 16-byte vector");

  extern __shared__ __attribute__((aligned(16))) char halo[];  // [SC_HH][SC_HW][CIN]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;

  int b = blockIdx.x;
  const int bq = (int)fdiv((uint32_t)b, a.divTx);
  const int tx = b - bq * a.tiles_x;
  const int n = (int)fdiv((uint32_t)bq, a.divTy);
  const int ty = bq - n * a.tiles_y;
  const int y0 = ty * SC_TH, x0 = tx * SC_TW;

  // ---- stage the halo tile -----------------------------------------------------------------
  const int sh = a.up ? 1 : 0;
  const char* img = a.src + (size_t)n * a.Hs * a.Ws * PIXB;
  static_assert(256 % VPP == 0, "a thread stages the same channel vector in every pass");
  const bool pbn = a.pbn.mean != nullptr;
  ScStageBn<T> sbn;
  if (pbn) sbn.load(a.pbn, (tid % VPP) * VEC);
  for (int v = tid; v < SC_HH * SC_HW * VPP; v += 256) {
    const int pix = v / VPP, cv = v - pix * VPP;
    const int hy = pix / SC_HW, hx = pix - hy * SC_HW;
    const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
    u32x4 val = {0u, 0u, 0u, 0u};
    if ((unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W) {
      val = *reinterpret_cast<const u32x4*>(img + ((size_t)(gy >> sh) * a.Ws + (gx >> sh)) * PIXB + cv * 16);
      if (pbn) val = sbn.apply(val, a.pbn.relu);
    }
    *reinterpret_cast<u32x4*>(halo + v * 16) = val;
  }

  // ---- weights -> registers (A fragments), per-lane tap offsets of the B fragments -----------
  u32x4 fa[TM][NCH];
  int boff[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int k0 = c * KC + lg * VEC;
    const int tap = k0 / CIN, ch = k0 - tap * CIN;
    const int kh = tap / 3, kw = tap - kh * 3;
    boff[c] = (k0 < K) ? ((kh * SC_HW + kw) * CIN + ch) * SZ : -1;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      u32x4 w = {0u, 0u, 0u, 0u};
      if (k0 < K) w = *reinterpret_cast<const u32x4*>(a.weight + ((size_t)(i * 16 + lr) * K + k0) * SZ);
      fa[i][c] = w;
    }
  }
  __syncthreads();
  float* red = reinterpret_cast<float*>(halo + SC_HH * SC_HW * PIXB);  // [4][TM*16][2] behind the halo tile
  // (A variant that looped over 16-channel passes of the same halo tile to cut registers - 5 instead of 3 waves/SIMD for
  //  Cout = 32 - measured slower: the weights then load after the barrier instead of under the halo staging.)

  // ---- MFMAs: wave w owns tile rows 2w, 2w+1; 4 fragments of 16 pixels -------------------------
  f32x4 acc[TM][4];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int f = 0; f < 4; ++f) acc[i][f] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    const int py = wave * 2 + (f >> 1), px = (f & 1) * 16 + lr;
    const char* pbase = halo + (py * SC_HW + px) * PIXB;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      u32x4 fb = {0u, 0u, 0u, 0u};
      if (boff[c] >= 0) fb = *reinterpret_cast<const u32x4*>(pbase + boff[c]);
#pragma unroll
      for (int i = 0; i < TM; ++i) ScMma<T>::run(fa[i][c], fb, acc[i][f]);
    }
  }

  // ---- epilogue -----------------------------------------------------------------------------------
  ScEpilogue<T, TM> ep;
  ep.load_constants(a, lg);
  ep.prefetch(a, n, y0, x0, wave, lr, lg);
  ep.finish(a, acc, red, n, y0, x0, (int)blockIdx.x, (int)gridDim.x, tid, wave, lr, lg);
}

// =================================================================================================
// STREAMING form of the kernel above (the one stp_conv2d_sc launches; the single-shot form stays for A/B runs, STP_SC_STREAM=0).
// The single-shot workgroup loads its halo tile through registers (2-6 dependent global-load round trips, the trip count is
// not a compile-time constant), synchronises, computes, stores and retires: measured 85-150 us on the 16x512x512 layers whose
// tensors take 30-55 us at HBM speed.  Here workgroups are PERSISTENT (a few per CU), the weights' A fragments and the fused
// BatchNormalization constants are fetched once per workgroup, and the halo tile of the NEXT tile is written straight into the
// other half of a double buffer by LDS-DMA (buffer_load ... lds, 16 bytes per lane, out-of-image pixels = out-of-range offset =
// zeros) while the current tile is multiplied and stored.  Per tile:
//   vmcnt(0) (own pieces of this tile, issued one tile ago) | producer BatchNormalization of the own pieces, in LDS | barrier |
//   epilogue operand prefetch | LDS-DMA of the next tile | MFMAs | epilogue (vmcnt leaves the DMA in flight) | stores
// Tiles are walked XCD by XCD: workgroup b (XCD b & 7) owns every (gridDim/8)-th tile of the b&7-th eighth of the tile list, so
// the halo rows shared by neighbouring tiles meet in one L2.  Same MFMA order, same epilogue arithmetic: results are bit-identical
// to the single-shot kernel.
// =================================================================================================
// (second launch bound = waves per SIMD = workgroups per CU the register allocation must leave room for)
template <typename T, int CIN, int TM>
__global__ __launch_bounds__(256, (TM == 1 ? (sizeof(T) == 2 && CIN <= 16 ? SC_WPE_SMALL : 3) : 2)) void conv_sc_stream_kernel(const ScArgs a) {
  constexpr int SZ = (int)sizeof(T);
  constexpr int VEC = Elem<T>::VEC;
  constexpr int KC = 4 * VEC;
  constexpr int K = 9 * CIN;
  constexpr int NCH = (K + KC - 1) / KC;
  constexpr int VPP = CIN / VEC;
  constexpr int PIXB = CIN * SZ;
  constexpr int NV = SC_HH * SC_HW * VPP;          // 16-byte vectors of a halo tile
  constexpr int NPASS = (NV + 255) / 256;          // LDS-DMA instructions per thread per tile
  constexpr int BUF = NPASS * 4096;                // one half of the double buffer (whole 1 KB wave pieces)
  static_assert(CIN % VEC == 0 && 256 % VPP == 0, "a thread stages the same channel vector in every pass");

  // LDS: [2][BUF] halo tiles | statistics scratch [4][TM*16][2] | BN-backward table [4][32] | producer-BN table [2][32]
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* red = reinterpret_cast<float*>(smem + 2 * BUF);
  float* ktab = red + 4 * TM * 16 * 2;
  float* ptab = ktab + 128;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  const int sh = a.up ? 1 : 0;

  // ---- tiles of this workgroup ------------------------------------------------------------------
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

  // ---- per-thread constants of the staging: halo coordinates of the vector of every pass ------------
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.src, 0, a.src_bytes, 0x00020000);
  int hyx[NPASS];            // hy << 16 | hx ; -1 past the tile
  const int cvb = (tid % VPP) * 16;
#pragma unroll
  for (int p = 0; p < NPASS; ++p) {
    const int v = p * 256 + tid, pix = v / VPP;
    const int hy = pix / SC_HW, hx = pix - hy * SC_HW;
    hyx[p] = v < NV ? (hy << 16 | hx) : -1;
  }
  const bool pbn = a.pbn.mean != nullptr;

  auto decode = [&](int tile, int& n, int& y0, int& x0) {
    const int bq = (int)fdiv((uint32_t)tile, a.divTx);
    const int tx = tile - bq * a.tiles_x;
    n = (int)fdiv((uint32_t)bq, a.divTy);
    const int ty = bq - n * a.tiles_y;
    y0 = ty * SC_TH; x0 = tx * SC_TW;
  };
  // LDS-DMA of a tile into buffer half b; returns the mask of passes whose vector lies inside the image
  // (EVERY wave issues NPASS instructions for EVERY tile slot - a piece that lies past the tile, or a tile past the end of the list
  //  (live == false), is requested out of range: zeros into the buffer's slack / the idle half, no memory traffic.  With a
  //  conditional issue the compiler cannot count the instructions that follow the epilogue's operand prefetch and waits for
  //  (nearly) all of them, i.e. for the NEXT tile, in every tile's epilogue.)
  auto issue_tile = [&](int tile, int b, bool live) -> uint32_t {
    int n, y0, x0;
    decode(tile, n, y0, x0);
    uint32_t inside = 0;
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
      const int gy = y0 - 1 + (hyx[p] >> 16), gx = x0 - 1 + (hyx[p] & 0xffff);
      const bool ok = live && hyx[p] >= 0 && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
      const uint32_t off = ok ? (uint32_t)((n * a.Hs + (gy >> sh)) * a.Ws + (gx >> sh)) * (uint32_t)PIXB + (uint32_t)cvb : 0x80000000u;
      inside |= ok ? (1u << p) : 0u;
#if !defined(STP_EXP) || STP_EXP != 12   // (what-if builds of scratch/sc_exp_build.sh: 11 = no output stores, 12 = no halo loads, 13 = no LDS reads / MFMAs)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + b * BUF + p * 4096 + wave * 1024), 16, off, 0, 0, 0);
#endif
    }
    return inside;
  };

  uint32_t inside_cur = issue_tile(t_first, 0, true);

  // ---- once per workgroup: weights -> registers (A fragments), lane addresses of the B fragments, constant tables -> LDS ----
  u32x4 fa[TM][NCH];
  uint32_t bl[NCH];            // LDS address of fragment chunk c for tile row 2*wave, column lr (buffer half 0); ~0 = chunk past K
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int k0 = c * KC + lg * VEC;
    const int tap = k0 / CIN, ch = k0 - tap * CIN;
    const int kh = tap / 3, kw = tap - kh * 3;
    bl[c] = (k0 < K) ? (uint32_t)((((wave * 2 + kh) * SC_HW + kw + lr) * CIN + ch) * SZ) : 0xffffffffu;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      u32x4 w = {0u, 0u, 0u, 0u};
      if (k0 < K) w = *reinterpret_cast<const u32x4*>(a.weight + ((size_t)(i * 16 + lr) * K + k0) * SZ);
      fa[i][c] = w;
    }
  }
  // (asm read path) the same addresses with the lane groups past K parked on address 0; last_ok: this lane's slot of the last chunk is below K
  uint32_t bla[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) bla[c] = (uint32_t)(uintptr_t)smem + (bl[c] == 0xffffffffu ? 0u : bl[c]);
  const bool last_ok = (NCH - 1) * KC + lg * VEC < K;
  ScEpilogue<T, TM, true> ep;
  ep.fill_table(a, ktab, tid);
  if (pbn && tid < CIN) {
    const float r = a.pbn.rstd[tid], sc = a.pbn.gamma ? r * a.pbn.gamma[tid] : r;
    ptab[tid] = sc;
    ptab[32 + tid] = (a.pbn.beta ? a.pbn.beta[tid] : 0.f) - a.pbn.mean[tid] * sc;
  }
  lds_barrier();               // the tables are visible

  auto body = [&](int tile, auto curc) {
    constexpr int CUR = decltype(curc)::value;
    int n, y0, x0;
    decode(tile, n, y0, x0);
    // this tile's pieces (own) have landed; the previous tile's stores are out.  (The builtin, not inline asm: the compiler's
    // wait-count bookkeeping restarts from zero here instead of carrying the prologue's pending loads around the loop.)
    ScStageBn<T> sbn;
    if (pbn) sbn.load_tab(ptab, ptab + 32, (tid % VPP) * VEC);      // (table reads: requested before the wait)
    __builtin_amdgcn_s_waitcnt(0x0f70);                   // vmcnt(0)
    asm volatile("" ::: "memory");
    if (pbn) {
      // padding applies to the NORMALISED tensor: out-of-image vectors stay zero.  Reads in batches of three (one LDS round trip per
      // batch instead of one per vector).
#pragma unroll
      for (int p0 = 0; p0 < NPASS; p0 += 3) {
        u32x4 v[3];
#pragma unroll
        for (int q = 0; q < 3; ++q)
          if (p0 + q < NPASS) v[q] = *reinterpret_cast<const u32x4*>(smem + CUR * BUF + ((p0 + q) * 256 + tid) * 16);
#pragma unroll
        for (int q = 0; q < 3; ++q)
          if (p0 + q < NPASS && (inside_cur & (1u << (p0 + q))))
            *reinterpret_cast<u32x4*>(smem + CUR * BUF + ((p0 + q) * 256 + tid) * 16) = sbn.apply(v[q], a.pbn.relu);
      }
    }
    lds_barrier();
    ep.prefetch(a, n, y0, x0, wave, lr, lg);
    const int next = tile + t_step;
    inside_cur = issue_tile(next < t_end ? next : tile, CUR ^ 1, next < t_end);

    f32x4 acc[TM][4];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int f = 0; f < 4; ++f) acc[i][f] = f32x4{0.f, 0.f, 0.f, 0.f};
#if !defined(STP_SC_PLAIN_READS)
    if constexpr (SZ == 2) {
      // SC_RING reads are kept IN FLIGHT ahead of the MFMAs that consume them (a ring of register quads, refilled after every
      // MFMA) and released one by one with counted waits.  As plain C++ the compiler kept one or two reads in flight (register budget of 4 waves per SIMD)
      // and paired every ds_read_b128 with an lgkmcnt(0): 12-36 exposed LDS round trips per tile in every wave, the largest item
      // of the per-tile serial chain (what-if without the reads / MFMAs: -25..-40 %).  Inline asm, so the order is ours; a counted
      // wait only ever over-waits when the compiler adds LDS / scalar-memory operations of its own (they are older or younger than
      // the whole group it releases).  The fragment's tile row / column half and the buffer half are instruction offsets.
      constexpr bool PART = (K % KC) != 0;                 // last chunk: lane groups past K read address 0 and are zeroed below
      constexpr int R = 4 * NCH;                            // reads of a tile, fragment-major: k = f * NCH + c
      constexpr int D = SC_RING < R ? SC_RING : R;          // reads in flight (a ring of D register quads)
      u32x4 ring[D];
      (void)bla; (void)last_ok;
      auto issue = [&ring, &bla](auto kc) {
        constexpr int k = decltype(kc)::value, F = k / NCH, c = k % NCH;
        constexpr int OFF = CUR * BUF + ((F >> 1) * SC_HW + (F & 1) * 16) * PIXB;
        static_assert(OFF + 2 * SC_HW * PIXB < 65536, "ds_read offset field");
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ring[k % D]) : "v"(bla[c]), "n"(OFF));
      };
      sc_unroll<D>(issue);
      sc_unroll<R>([&ring, &fa, &acc, &issue, last_ok](auto kc) {
        constexpr int k = decltype(kc)::value, F = k / NCH, c = k % NCH;
        constexpr int young = (R - 1 - k) < (D - 1) ? (R - 1 - k) : (D - 1);      // reads issued after read k at this point
        asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(ring[k % D]) : "n"(young));
        u32x4 v = ring[k % D];
        if (PART && c == NCH - 1 && !last_ok) v = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < TM; ++i) ScMma<T>::run(fa[i][c], v, acc[i][F]);
        if constexpr (k + D < R) issue(std::integral_constant<int, k + D>{});
      });
    } else
#endif
    {
#pragma unroll
    for (int f = 0; f < 4; ++f) {
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        u32x4 fb = {0u, 0u, 0u, 0u};
        // the fragment's tile row / column half and the buffer half are instruction offsets
#if !defined(STP_EXP) || STP_EXP != 13
        if (bl[c] != 0xffffffffu) fb = *reinterpret_cast<const u32x4*>(smem + bl[c] + (CUR * BUF + ((f >> 1) * SC_HW + (f & 1) * 16) * PIXB));
#pragma unroll
        for (int i = 0; i < TM; ++i) ScMma<T>::run(fa[i][c], fb, acc[i][f]);
#else
        if (a.N < 0) acc[0][f][0] += __uint_as_float(fa[0][c][0] + bl[c]);
#endif
      }
    }
    }
    ep.finish(a, acc, red, n, y0, x0, tile, ntiles, tid, wave, lr, lg);
  };

  ep.reset_stats();
  for (int tile = t_first; tile < t_end; tile += 2 * t_step) {
    body(tile, std::integral_constant<int, 0>{});
    if (tile + t_step < t_end) body(tile + t_step, std::integral_constant<int, 1>{});
  }
  ep.flush_stats(a, red, (int)blockIdx.x, (int)gridDim.x, tid, wave, lr, lg);
}

template <typename T, int CIN, int TM>
static int launch_sc(const ScArgs& a, hipStream_t s) {
  const int ntiles = a.N * a.tiles_x * a.tiles_y;
  if (sc_stream_on()) {
    constexpr int NV = SC_HH * SC_HW * (CIN / Elem<T>::VEC), NPASS = (NV + 255) / 256;
    const size_t lds = (size_t)2 * NPASS * 4096 + (4 * TM * 16 * 2 + 128 + 64) * sizeof(float);
    const int blocks = sc_stream_blocks(Elem<T>::DTYPE, CIN, TM * 16, ntiles);      // a multiple of 8 on this machine: XCD-contiguous tile walk
    static bool attr_set = false;
    if (lds > 64 * 1024 && !attr_set) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_sc_stream_kernel<T, CIN, TM>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return STP_E_LAUNCH;
      attr_set = true;
    }
    hipLaunchKernelGGL((conv_sc_stream_kernel<T, CIN, TM>), dim3(blocks), dim3(256), lds, s, a);
    STP_LAUNCH_CHECK();
    return STP_OK;
  }
  const size_t lds = (size_t)SC_HH * SC_HW * CIN * sizeof(T) + 4 * TM * 16 * 2 * sizeof(float);
  hipLaunchKernelGGL((conv_sc_kernel<T, CIN, TM>), dim3(ntiles), dim3(256), lds, s, a);
  STP_LAUNCH_CHECK();
  return STP_OK;
}

// columns of the [stat][channel][column] partial sums this kernel writes for p (one per tile; one per workgroup in streaming form)
extern "C" int stp_conv2d_sc_stats_tiles(const stp_conv_params* p) {
  const int ntiles = p->N * ceil_div(p->Hv, SC_TH) * ceil_div(p->Wv, SC_TW);
  if (!sc_stream_on() || p->stats_slots) return ntiles;
  return sc_stream_blocks(p->dtype, p->C0, p->Cout <= 16 ? 16 : 32, ntiles);
}

template <typename T>
static int dispatch_sc(const ScArgs& a, int cin, hipStream_t s) {
  const int tm = a.Cout <= 16 ? 1 : 2;
  switch (cin * 4 + tm) {
    case 4 * 4 + 1: if constexpr (sizeof(T) == 4) return launch_sc<T, 4, 1>(a, s); else return STP_E_BADARG;
    case 4 * 4 + 2: if constexpr (sizeof(T) == 4) return launch_sc<T, 4, 2>(a, s); else return STP_E_BADARG;
    case 8 * 4 + 1: return launch_sc<T, 8, 1>(a, s);
    case 8 * 4 + 2: return launch_sc<T, 8, 2>(a, s);
    case 16 * 4 + 1: return launch_sc<T, 16, 1>(a, s);
    case 16 * 4 + 2: return launch_sc<T, 16, 2>(a, s);
    case 32 * 4 + 1: return launch_sc<T, 32, 1>(a, s);
    case 32 * 4 + 2: return launch_sc<T, 32, 2>(a, s);
    default: return STP_E_BADARG;
  }
}

// Is this convolution served by the small-channel kernel?  (stp_conv2d consults this before the GEMM path.)
extern "C" int stp_conv2d_sc_eligible(const stp_conv_params* p) {
  if (!p || !stp_dtype_ok(p->dtype)) return 0;
  const int vec = p->dtype == STP_H16 ? 8 : 4;
  const int cin = p->C0;
  // fp32 holds half as many k per 16-byte vector: cap Cin at 16 there so the A fragments stay in registers
  const bool cin_ok = p->dtype == STP_F32 ? (cin == 4 || cin == 8 || cin == 16) : (cin == 8 || cin == 16 || cin == 32);
  return p->KH == 3 && p->KW == 3 && p->stride == 1 && p->pad == 1 && p->C1 == 0 && cin_ok && (cin % vec) == 0 && p->Cout <= 32 &&
         p->Cd0 == p->Cout && !p->residual && p->Ho == p->Hv && p->Wo == p->Wv &&
         (p->src0_mode == STP_SRC_DIRECT || (p->src0_mode == STP_SRC_NEAREST2X && p->Hv == 2 * p->Hs0 && p->Wv == 2 * p->Ws0));
}

extern "C" int stp_conv2d_sc(const stp_conv_params* p, void* stream) {
  if (!stp_conv2d_sc_eligible(p) || !p->src0 || !p->weight || !p->dst0) return STP_E_BADARG;
  ScArgs a;
  a.src = (const char*)p->src0; a.weight = (const char*)p->weight; a.bias = p->bias; a.dst = (char*)p->dst0;
  a.N = p->N; a.H = p->Hv; a.W = p->Wv; a.Hs = p->Hs0; a.Ws = p->Ws0; a.Cout = p->Cout;
  a.up = p->src0_mode == STP_SRC_NEAREST2X; a.accumulate = p->accumulate0; a.relu = p->relu;
  {
    const uint64_t sb = (uint64_t)p->N * p->Hs0 * p->Ws0 * p->C0 * (p->dtype == STP_H16 ? 2 : 4);
    if (sb >= 0x80000000ull) return STP_E_BADARG;   // 32-bit LDS-DMA offsets
    a.src_bytes = (uint32_t)sb;
  }
  a.tiles_x = ceil_div(a.W, SC_TW); a.tiles_y = ceil_div(a.H, SC_TH);
  a.divTx = make_fastdiv((uint32_t)a.tiles_x); a.divTy = make_fastdiv((uint32_t)a.tiles_y);
  a.stats = p->stats_partial;
  a.stat_slots = p->stats_slots;
  if (a.stats && (p->Cout & 3)) return STP_E_BADARG;
  if (a.stat_slots && (!a.stats || (a.stat_slots & (a.stat_slots - 1)) || a.stat_slots > 64)) return STP_E_BADARG;
  a.bnb.x = (const char*)p->bnb_x; a.bnb.mean = p->bnb_mean; a.bnb.rstd = p->bnb_rstd; a.bnb.gamma = p->bnb_gamma;
  a.bnb.beta = p->bnb_beta; a.bnb.relu = p->bnb_relu;
  a.sum2 = p->dst_sum2x2;
  a.pbn.x = nullptr; a.pbn.mean = p->src_bn_mean; a.pbn.rstd = p->src_bn_rstd; a.pbn.gamma = p->src_bn_gamma; a.pbn.beta = p->src_bn_beta;
  a.pbn.relu = p->src_bn_relu;
  if (a.pbn.mean && !a.pbn.rstd) return STP_E_BADARG;
  if (a.bnb.x && (!a.stats || !a.bnb.mean || !a.bnb.rstd || p->relu)) return STP_E_BADARG;
  if (a.sum2 && ((p->Cout & 3) || (a.H & 1) || (a.W & 1) || p->bias || p->relu || (a.stats && !a.bnb.x))) return STP_E_BADARG;
  const_cast<stp_conv_params*>(p)->stats_tiles = stp_conv2d_sc_stats_tiles(p);
  hipStream_t s = (hipStream_t)stream;
  {
    const int r = sc_lean_launch(a, p->C0, p->dtype, s);      // the lean kernel (conv_sc_lean.hip) where it serves the configuration
    if (r != 1) return r;
  }
  return p->dtype == STP_H16 ? dispatch_sc<bf16_t>(a, p->C0, s) : dispatch_sc<float>(a, p->C0, s);
}

// =================================================================================================
// WIDE-OUTPUT form: data gradient of the first full-resolution decoder convolution,
//   conv3x3(concat(UpSampling2D(2)(x), skip)) -> 32 channels   (U-Net decoder_stage3_conv1 at 16 x 256 x 256)
// i.e. a 3x3 convolution of dY (32 channels) into 128 channels that belong to TWO tensors: the first Cd0 are the gradient of the
// upsampled tensor - its 2x2 blocks are summed here (dst_sum2x2) and land on the low-resolution gradient directly, optionally
// with the fused BatchNormalization-backward sums - the rest the gradient of the skip tensor.  The generic per-tap kernel spent
// 216 us on it (K = 32 per tap: a ring stage per MFMA step) and wrote a full-resolution gradient of the upsampled tensor that
// stp_upsample2x_bwd read back; the layer moves 67 MB in and 168 MB out, ~45 us at HBM speed.
// Same streaming scheme as conv_sc_stream_kernel (persistent workgroups, 8 x 32 tiles, the next tile's dY halo written into the
// other half of a double buffer by LDS-DMA); what differs is the split: the 4 waves share every B fragment (pixels) and own 32
// OUTPUT CHANNELS each (A fragments = 32 x 288 weights = 72 registers), a tile is walked in two passes of 4 rows (accumulators:
// 2 x 8 fragments).  Waves below Cd0 / 32 take the summed epilogue, the others store the skip gradient.
// =================================================================================================
struct ScwArgs {
  const char* src;      // dY [N,H,W,32]
  const char* weight;   // [128][9*32]  (the data-gradient weight copy: rows = input channels of the forward convolution)
  char* dst_up;         // [N,H/2,W/2,C0]
  char* dst_sk;         // [N,H,W,C1]
  int N, H, W, C0, C1, acc_up, acc_sk;
  int tiles_x, tiles_y;
  FastDiv divTx, divTy;
  float* stats;         // [2][C0][workgroups] (fused BatchNormalization-backward sums of the first C0 channels)
  BnBack bnb;
  uint32_t src_bytes, x_bytes;   // sizes of src and bnb.x (buffer descriptors of the LDS-DMA)
  uint32_t up_bytes, sk_bytes;   // sizes of dst_up and dst_sk (buffer descriptors of the stores)
};

template <typename T>
__global__ __launch_bounds__(256, 2) void conv_scw_stream_kernel(const ScwArgs a) {
  constexpr int CIN = 32, SZ = (int)sizeof(T), K = 9 * CIN, NCH = 9, VPP = 4, PIXB = CIN * SZ;
  constexpr int NV = SC_HH * SC_HW * VPP, NPASS = (NV + 255) / 256, BUF = NPASS * 4096;
  static_assert(SZ == 2, "16-bit storage");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* ktab = reinterpret_cast<float*>(smem + 2 * BUF);     // [4][128]: scale, shift, mean, rstd of the first C0 channels
  // [2][XBUF]: the BatchNormalization input of the tile's 4 x 16 low-resolution pixels ([pixel][C0]), staged by LDS-DMA with the
  // halo tile - a tile ahead of its use (fetched where it is used, each pass of the summed epilogue stalled on an HBM round trip:
  // 135 vs 86 us without the fused sums)
  constexpr int XBUF = 3 * 4096;
  char* const xlds = smem + 2 * BUF + 2048;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  // this wave's two 16-channel groups.  With as many summed as skip channels (the U-Net shape) every wave takes one group of each:
  // the summed epilogue carries the BatchNormalization-backward arithmetic, and two waves doing all of it while the other two
  // wait at the tile barrier cost 40 us of 125
  const int cbi[2] = {a.C0 == 64 ? wave * 16 : wave * 32, a.C0 == 64 ? 64 + wave * 16 : wave * 32 + 16};
  const bool upi[2] = {cbi[0] < a.C0, cbi[1] < a.C0};          // (wave-uniform) summed epilogue / skip epilogue

  const int ntiles = a.N * a.tiles_x * a.tiles_y;
  int t_first, t_step, t_end;
  if ((gridDim.x & 7) == 0) {
    const int q = ntiles >> 3, r = ntiles & 7, x = blockIdx.x & 7;
    const int start = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    t_first = start + (int)(blockIdx.x >> 3); t_step = (int)(gridDim.x >> 3); t_end = start + q + (x < r ? 1 : 0);
  } else {
    t_first = (int)blockIdx.x; t_step = (int)gridDim.x; t_end = ntiles;
  }
  if (t_first >= t_end) {    // (no tile: its statistics column must still be defined)
    if (a.stats && tid < a.C0) {
      a.stats[(size_t)tid * gridDim.x + blockIdx.x] = 0.f;
      a.stats[((size_t)a.C0 + tid) * gridDim.x + blockIdx.x] = 0.f;
    }
    return;
  }

  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.src, 0, a.src_bytes, 0x00020000);
  const int H2 = a.H >> 1, W2 = a.W >> 1;
  const bool bnb = a.bnb.x != nullptr;
  const bool xrs_on = bnb;
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)(bnb ? a.bnb.x : a.src), 0, bnb ? a.x_bytes : 0u, 0x00020000);
  int hyx[NPASS];
  const int cvb = (tid % VPP) * 16;
#pragma unroll
  for (int p = 0; p < NPASS; ++p) {
    const int v = p * 256 + tid, pix = v / VPP;
    const int hy = pix / SC_HW, hx = pix - hy * SC_HW;
    hyx[p] = v < NV ? (hy << 16 | hx) : -1;
  }
  int xq[3];                 // vector p * 256 + tid of the x tile: low-res row << 16 | column << 8 | 16-byte vector of the pixel; -1 past it
  {
    const int vpp = a.C0 >> 3;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const int v = p * 256 + tid, px = v / vpp;
      xq[p] = (xrs_on && px < 64) ? ((px >> 4) << 16 | (px & 15) << 8 | (v - px * vpp)) : -1;
    }
  }
  auto decode = [&](int tile, int& n, int& y0, int& x0) {
    const int bq = (int)fdiv((uint32_t)tile, a.divTx);
    const int tx = tile - bq * a.tiles_x;
    n = (int)fdiv((uint32_t)bq, a.divTy);
    const int ty = bq - n * a.tiles_y;
    y0 = ty * SC_TH; x0 = tx * SC_TW;
  };
  // (instruction count matters here: the waves of these streaming kernels spend as long in address arithmetic as in MFMAs.  A tile
  //  whose halo lies inside the image - all but the border tiles - takes its LDS-DMA offsets as (per-thread constant) + (tile base))
  uint32_t hrel[NPASS];
#pragma unroll
  for (int p = 0; p < NPASS; ++p) hrel[p] = hyx[p] >= 0 ? (uint32_t)(((hyx[p] >> 16) * a.W + (hyx[p] & 0xffff)) * PIXB + cvb) : 0x80000000u;
  auto issue_tile = [&](int tile, int b) {
    int n, y0, x0;
    decode(tile, n, y0, x0);
    const bool inner = y0 >= 1 && x0 >= 1 && y0 + SC_TH + 1 <= a.H && x0 + SC_TW + 1 <= a.W;      // (uniform)
    const uint32_t tbase = (uint32_t)(((n * a.H + y0 - 1) * a.W + x0 - 1) * PIXB);
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
      uint32_t off;
      if (inner) {
        off = hyx[p] >= 0 ? tbase + hrel[p] : 0x80000000u;
      } else {
        const int gy = y0 - 1 + (hyx[p] >> 16), gx = x0 - 1 + (hyx[p] & 0xffff);
        const bool ok = hyx[p] >= 0 && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
        off = ok ? (uint32_t)((n * a.H + gy) * a.W + gx) * (uint32_t)PIXB + (uint32_t)cvb : 0x80000000u;
      }
#if !defined(STP_EXP) || STP_EXP != 32   // (what-if builds, scratch/sc_exp_build.sh: 31 = no output stores, 32 = no halo loads, 33 = no LDS reads / MFMAs)
      if (p * 256 + wave * 64 < NV)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + b * BUF + p * 4096 + wave * 1024), 16, off, 0, 0, 0);
#endif
    }
    if (xrs_on) {
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        const int ly = (y0 >> 1) + (xq[p] >> 16), lx = (x0 >> 1) + ((xq[p] >> 8) & 0xff);
        const bool ok = xq[p] >= 0 && ly < H2 && lx < W2;
        const uint32_t off = ok ? (uint32_t)(((n * H2 + ly) * W2 + lx) * a.C0 + (xq[p] & 0xff) * 8) * 2u : 0x80000000u;
        if (p * 256 + wave * 64 < 8 * a.C0)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void*)(xlds + b * XBUF + p * 4096 + wave * 1024), 16, off, 0, 0, 0);
      }
    }
  };
  issue_tile(t_first, 0);

  // once per workgroup: this wave's weights -> registers; chunk c of K = tap c (32 channels: lane group lg holds channels 8 lg ..)
  u32x4 fa[2][NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int i = 0; i < 2; ++i)
      fa[i][c] = *reinterpret_cast<const u32x4*>(a.weight + ((size_t)(cbi[i] + lr) * K + c * CIN + lg * 8) * SZ);
  const uint32_t lbase = (uint32_t)((lr * CIN + lg * 8) * SZ);     // lane part of every B-fragment address
  if (bnb && tid < 128) {
    float sc = 0.f, sh = 0.f, mu = 0.f, rsd = 0.f;
    if (tid < a.C0) {
      mu = a.bnb.mean[tid]; rsd = a.bnb.rstd[tid];
      sc = a.bnb.gamma ? rsd * a.bnb.gamma[tid] : rsd;
      sh = (a.bnb.beta ? a.bnb.beta[tid] : 0.f) - mu * sc;
    }
    ktab[tid] = sc; ktab[128 + tid] = sh; ktab[256 + tid] = mu; ktab[384 + tid] = rsd;
  }
  f32x2 ssp[2], qqp[2];        // fused sums of the lane's two channels (see the summed epilogue)
#pragma unroll
  for (int i = 0; i < 2; ++i) { ssp[i] = f32x2{0.f, 0.f}; qqp[i] = f32x2{0.f, 0.f}; }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the first tile's own pieces have landed
  lds_barrier();               // ... everybody's; the table is visible
  const __amdgpu_buffer_rsrc_t rup = __builtin_amdgcn_make_buffer_rsrc((void*)a.dst_up, 0, a.up_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsk = __builtin_amdgcn_make_buffer_rsrc((void*)a.dst_sk, 0, a.sk_bytes, 0x00020000);
  // per-lane byte offsets inside an output row segment (see the epilogues): summed half = low-resolution pixel lr / 2, channels
  // cbi + 4 lg + 2 (lr & 1); skip half = pixel lr, channels cbi - C0 + 4 lg
  uint32_t lup[2], lsk[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    lup[i] = (uint32_t)((lr >> 1) * a.C0 + cbi[i] + lg * 4 + 2 * (lr & 1)) * 2u;
    lsk[i] = (uint32_t)(lr * a.C1 + (cbi[i] - a.C0) + lg * 4) * 2u;
  }

  auto body = [&](int tile, auto curc) {
    constexpr int CUR = decltype(curc)::value;
    int n, y0, x0;
    decode(tile, n, y0, x0);
    // (this tile's halo is complete and visible, and every wave has left the other half: the wait + barrier at the end of the
    //  previous tile - there the wait sits BEFORE the last pass's stores, so it drains the stores of a pass ago instead of the
    //  ones just issued: at the top of the tile it exposed a store round trip per tile)
    const int next = tile + t_step;
    const bool full = y0 + SC_TH <= a.H && x0 + SC_TW <= a.W;      // (uniform) no pixel of the tile lies outside the image
    if (next < t_end) issue_tile(next, CUR ^ 1);

#pragma unroll
    for (int h = 0; h < 2; ++h) {
      f32x4 acc[2][8];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int f = 0; f < 8; ++f) acc[i][f] = f32x4{0.f, 0.f, 0.f, 0.f};
      // a B fragment = 16 pixels of halo row r shifted by kw: it serves the taps (kh, kw) of the output rows r - kh, so each of the
      // 6 x 3 x 2 fragments of a pass is read ONCE and used by up to three output rows (36 LDS reads for 72 fragment uses).
      // Software pipeline, three reads ahead: left alone the compiler keeps ONE fragment register (ds_read -> lgkmcnt(0) -> 2-6
      // MFMAs: 36 exposed LDS latencies a pass); the scheduling barriers pin read s + 3 in front of the MFMAs of read s
      auto bfrag = [&](int sidx) -> u32x4 {
        const int r = sidx / 6, kw = (sidx >> 1) % 3, h2 = sidx & 1;
        return *reinterpret_cast<const u32x4*>(smem + lbase + (CUR * BUF + ((h * 4 + r) * SC_HW + h2 * 16 + kw) * PIXB));
      };
      u32x4 fbq[4];
#if defined(STP_EXP) && STP_EXP == 33
      if (a.N < 0) acc[0][0][0] += __uint_as_float(fa[0][h][0] + fa[1][h][1] + lbase);
#else
      fbq[0] = bfrag(0); fbq[1] = bfrag(1); fbq[2] = bfrag(2);
#pragma unroll
      for (int sidx = 0; sidx < 36; ++sidx) {
#if !defined(STP_EXP) || STP_EXP != 34
        __builtin_amdgcn_sched_barrier(0);
#endif
        if (sidx + 3 < 36) fbq[(sidx + 3) & 3] = bfrag(sidx + 3);
        const int r = sidx / 6, kw = (sidx >> 1) % 3, h2 = sidx & 1;
        const u32x4 fb = fbq[sidx & 3];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          const int orow = r - kh;
          if (orow < 0 || orow > 3) continue;
          acc[0][orow * 2 + h2] = mfma16_16x16x32(fa[0][kh * 3 + kw], fb, acc[0][orow * 2 + h2]);
          acc[1][orow * 2 + h2] = mfma16_16x16x32(fa[1][kh * 3 + kw], fb, acc[1][orow * 2 + h2]);
        }
      }
#if !defined(STP_EXP) || STP_EXP != 34
      __builtin_amdgcn_sched_barrier(0);
#endif
#endif
      if (h == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the next tile's own pieces have landed (issued two passes ago)
      auto epilogue = [&](auto fullc) {
        constexpr bool FULL = decltype(fullc)::value;      // no pixel of the tile lies outside the image: no per-lane bounds work
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (upi[i]) {
          // gradient of UpSampling2D(2): fragments f and f + 2 are the two rows of an output row, lanes lr and lr ^ 1 its columns.
          // After the quad_perm add both lanes of a pair hold the 4 channel sums: the even lane finishes channels 0, 1, the odd lane
          // channels 2, 3 (half the per-lane arithmetic of the fused BatchNormalization backward, 4-byte accesses, no idle lanes)
          const int par = lr & 1;
          const int co = cbi[i] + lg * 4 + 2 * par;
#pragma unroll
          for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
              asm volatile("" ::: "memory");       // (keeps the table / x reads of the iterations from being hoisted together: registers)
              const int gy = y0 + h * 4 + q * 2, gx = x0 + h2 * 16 + lr;
              f32x4 v = acc[i][q * 4 + h2] + acc[i][q * 4 + h2 + 2];
#pragma unroll
              for (int e = 0; e < 4; ++e)
                v[e] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[e]), 0xB1, 0xf, 0xf, true));  // quad_perm [1,0,3,2]
              f32x2 v2 = par ? f32x2{v[2], v[3]} : f32x2{v[0], v[1]};
              // buffer addressing: (per-lane constant voffset) + (scalar soffset of the output row / column half) - no per-store VALU
              const uint32_t so = (uint32_t)(((n * H2 + (y0 >> 1) + h * 2 + q) * W2 + (x0 >> 1) + h2 * 8) * a.C0) * 2u;
              const uint32_t vo = (FULL || (gy < a.H && gx < a.W)) ? lup[i] : 0x80000000u;
              if (a.acc_up) {
                const uint32_t w0 = __builtin_amdgcn_raw_buffer_load_b32(rup, vo, so, 0);
                v2 += f32x2{h16lo_to_f32(w0), h16hi_to_f32(w0)};
              }
              uint32_t w = pack_bf16x2(v2.x, v2.y);
              if (bnb && (FULL || vo != 0x80000000u)) {
                const f32x2 ksc = *reinterpret_cast<const f32x2*>(ktab + co), ksh = *reinterpret_cast<const f32x2*>(ktab + 128 + co);
                const f32x2 kmu = *reinterpret_cast<const f32x2*>(ktab + 256 + co), krs = *reinterpret_cast<const f32x2*>(ktab + 384 + co);
                const uint32_t xw = *reinterpret_cast<const uint32_t*>(xlds + CUR * XBUF + (((h * 2 + q) * 16 + h2 * 8 + (lr >> 1)) * a.C0 + co) * 2);
                const f32x2 xv = {h16lo_to_f32(xw), h16hi_to_f32(xw)}, dy = {h16lo_to_f32(w), h16hi_to_f32(w)};     // dY as stored
                f32x2 g;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                  g[e] = bn_act_on(bn_affine(xv[e], ksc[e], ksh[e]), a.bnb.relu) ? dy[e] : 0.f;
                  ssp[i][e] += g[e];
                  qqp[i][e] += g[e] * ((xv[e] - kmu[e]) * krs[e]);
                }
                w = pack_bf16x2(g.x, g.y);
              }
#if defined(STP_EXP) && STP_EXP == 31
              if (a.N < 0)
#endif
              __builtin_amdgcn_raw_buffer_store_b32(w, rup, vo, so, 0);
            }
        } else {
#pragma unroll
          for (int f = 0; f < 8; ++f) {
            const int gy = y0 + h * 4 + (f >> 1), gx = x0 + (f & 1) * 16 + lr;
            const uint32_t so = (uint32_t)(((n * a.H + y0 + h * 4 + (f >> 1)) * a.W + x0 + (f & 1) * 16) * a.C1) * 2u;
            const uint32_t vo = (FULL || (gy < a.H && gx < a.W)) ? lsk[i] : 0x80000000u;
            f32x4 v = acc[i][f];
            if (a.acc_sk) {
              const u32x2 w0 = __builtin_amdgcn_raw_buffer_load_b64(rsk, vo, so, 0);
              v += f32x4{h16lo_to_f32(w0.x), h16hi_to_f32(w0.x), h16lo_to_f32(w0.y), h16hi_to_f32(w0.y)};
            }
#if defined(STP_EXP) && STP_EXP == 31
            if (a.N < 0)
#endif
            __builtin_amdgcn_raw_buffer_store_b64(u32x2{pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w)}, rsk, vo, so, 0);
          }
        }
      }
      };
      if (full) epilogue(std::integral_constant<bool, true>{}); else epilogue(std::integral_constant<bool, false>{});
    }
    lds_barrier();        // the next tile is visible to everybody, and everybody has left this one
  };

  for (int tile = t_first; tile < t_end; tile += 2 * t_step) {
    body(tile, std::integral_constant<int, 0>{});
    if (tile + t_step < t_end) body(tile + t_step, std::integral_constant<int, 1>{});
  }
  // fused sums: one column per workgroup; a 16-channel group belongs to one wave outright (no cross-wave step).  Even lanes hold
  // the sums of channels 0, 1 of their 4-channel block, odd lanes of channels 2, 3: row_shr 2, 4, 8 leave them in lanes 14 and 15
  if (a.stats) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (!upi[i]) continue;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        float sv = ssp[i][e], qv = qqp[i][e];
#pragma unroll
        for (int sh = 0; sh < 3; ++sh) {
          sv += __builtin_bit_cast(float, sh == 0 ? __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sv), 0x112, 0xf, 0xf, true)
                                          : sh == 1 ? __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sv), 0x114, 0xf, 0xf, true)
                                                    : __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sv), 0x118, 0xf, 0xf, true));
          qv += __builtin_bit_cast(float, sh == 0 ? __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, qv), 0x112, 0xf, 0xf, true)
                                          : sh == 1 ? __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, qv), 0x114, 0xf, 0xf, true)
                                                    : __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, qv), 0x118, 0xf, 0xf, true));
        }
        if (lr >= 14) {
          const int ch = cbi[i] + lg * 4 + 2 * (lr & 1) + e;
          a.stats[(size_t)ch * gridDim.x + blockIdx.x] = sv;
          a.stats[((size_t)a.C0 + ch) * gridDim.x + blockIdx.x] = qv;
        }
      }
    }
  }
}

static int scw_blocks(int ntiles) {
  const int64_t b = (int64_t)sc_cu_count() * 2;
  return (int)(b < ntiles ? b : ntiles);
}

// Is this convolution the two-destination data gradient served by conv_scw_stream_kernel?  (stp_conv2d consults this first.)
extern "C" int stp_conv2d_scw_eligible(const stp_conv_params* p) {
  static const bool on = !(getenv("STP_SCW") && atoi(getenv("STP_SCW")) == 0);
  if (!on || !p || p->dtype != STP_H16) return 0;
  return p->KH == 3 && p->KW == 3 && p->stride == 1 && p->pad == 1 && p->C0 == 32 && p->C1 == 0 && p->Cout == 128 && p->dst1 != nullptr &&
         p->Cd0 > 0 && p->Cd0 < 128 && (p->Cd0 % 32) == 0 && p->dst_sum2x2 == 1 && p->src0_mode == STP_SRC_DIRECT && p->Hs0 == p->Hv &&
         p->Ws0 == p->Wv && p->Ho == p->Hv && p->Wo == p->Wv && !(p->Ho & 1) && !(p->Wo & 1) && !p->bias && !p->relu && !p->residual &&
         !p->src_bn_mean && !p->stats_slots && !p->fold_src && !p->weight_up;
}
extern "C" int stp_conv2d_scw_stats_tiles(const stp_conv_params* p) {
  return scw_blocks(p->N * ceil_div(p->Hv, SC_TH) * ceil_div(p->Wv, SC_TW));
}

extern "C" int stp_conv2d_scw(const stp_conv_params* p, void* stream) {
  if (!stp_conv2d_scw_eligible(p) || !p->src0 || !p->weight || !p->dst0) return STP_E_BADARG;
  ScwArgs a;
  a.src = (const char*)p->src0; a.weight = (const char*)p->weight; a.dst_up = (char*)p->dst0; a.dst_sk = (char*)p->dst1;
  a.N = p->N; a.H = p->Hv; a.W = p->Wv; a.C0 = p->Cd0; a.C1 = p->Cout - p->Cd0; a.acc_up = p->accumulate0; a.acc_sk = p->accumulate1;
  {
    const uint64_t sb = (uint64_t)p->N * p->Hs0 * p->Ws0 * p->C0 * 2;
    if (sb >= 0x80000000ull) return STP_E_BADARG;   // 32-bit LDS-DMA offsets
    a.src_bytes = (uint32_t)sb;
    a.x_bytes = (uint32_t)((uint64_t)p->N * (p->Hv / 2) * (p->Wv / 2) * p->Cd0 * 2);    // (a quarter of the pixels, < 128 channels: smaller)
    a.up_bytes = a.x_bytes;
    const uint64_t kb = (uint64_t)p->N * p->Hv * p->Wv * (p->Cout - p->Cd0) * 2;
    if (kb >= 0x80000000ull) return STP_E_BADARG;
    a.sk_bytes = (uint32_t)kb;
  }
  a.tiles_x = ceil_div(a.W, SC_TW); a.tiles_y = ceil_div(a.H, SC_TH);
  a.divTx = make_fastdiv((uint32_t)a.tiles_x); a.divTy = make_fastdiv((uint32_t)a.tiles_y);
  a.stats = p->stats_partial;
  a.bnb.x = (const char*)p->bnb_x; a.bnb.mean = p->bnb_mean; a.bnb.rstd = p->bnb_rstd; a.bnb.gamma = p->bnb_gamma;
  a.bnb.beta = p->bnb_beta; a.bnb.relu = p->bnb_relu;
  if (a.bnb.x && (!a.stats || !a.bnb.mean || !a.bnb.rstd)) return STP_E_BADARG;
  if (a.stats && !a.bnb.x) return STP_E_BADARG;      // the only sums this kernel fuses are the BatchNormalization-backward ones
  const int ntiles = a.N * a.tiles_x * a.tiles_y;
  const int blocks = scw_blocks(ntiles);
  const_cast<stp_conv_params*>(p)->stats_tiles = blocks;
  constexpr int NPASS = (SC_HH * SC_HW * 4 + 255) / 256;
  const size_t lds = (size_t)2 * NPASS * 4096 + 512 * sizeof(float) + 2 * 3 * 4096;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_scw_stream_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return STP_E_LAUNCH;
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_scw_stream_kernel<bf16_t>), dim3(blocks), dim3(256), lds, (hipStream_t)stream, a);
  STP_LAUNCH_CHECK();
  return STP_OK;
}

// =================================================================================================
// NARROW-OUTPUT form: forward of the first full-resolution decoder convolution,
//   conv3x3(concat(UpSampling2D(2)(x), skip)) : 64 + 64 -> 32 channels   (U-Net decoder_stage3_conv1 at 16 x 256 x 256)
// The generic per-tap kernel stages every pixel once per tap and channel block: 1.9 GB through L2 -> LDS per launch (8.5 TB/s, 221 us;
// ring depth and tile size change nothing).  Halo residency stages each input byte once (the upsampled half as its LOW-RESOLUTION
// pixels: 6 x 18 instead of 10 x 34), 220 MB.  With 32 output channels a staged byte feeds one MFMA column block only, so nothing is
// shared across channel tiles: every wave keeps ALL the weights as A fragments - 32 x 1152 = 288 registers, which is why this kernel
// runs ONE wave per SIMD (512 registers a lane: 256 VGPRs + the accumulators and the rest of the weights in AGPRs) - and owns two rows
// of the 8 x 32 tile like conv_sc_stream_kernel, whose epilogue (bias, ReLU, fused BatchNormalization statistics) it reuses.
// LDS per buffer: the halo as four 32-channel planes ([skip 0-31][skip 32-63][up 0-31][up 32-63]: 64-byte pixels, conflict-free
// ds_read_b128 fragments) = 56 KB, double-buffered by LDS-DMA one tile ahead.  A fragment of halo row r shifted by kw serves the taps
// (kh, kw) of both output rows; with one wave per SIMD the LDS latency is hidden by hand: the scheduling barriers keep three fragment
// reads in flight ahead of the MFMAs that consume them.
// =================================================================================================
struct ScnArgs {         // (next to a ScArgs = the epilogue's view: dst, Cout, N, H, W, bias, relu, accumulate, stats; src = the upsampled tensor [N,H/2,W/2,64])
  const char* skip;     // [N,H,W,64]
  uint32_t up_bytes, skip_bytes, dst_bytes;
};

constexpr int SCN_SKV = SC_HH * SC_HW * 4, SCN_UH = SC_TH / 2 + 2, SCN_UW = SC_TW / 2 + 2, SCN_UPV = SCN_UH * SCN_UW * 4;   // 16-byte vectors of a plane
constexpr int SCN_UP0 = (2 * SCN_SKV + 63) / 64 * 64;        // the upsampled planes start at a wave boundary of the staging (one descriptor per wave and pass)
constexpr int SCN_NV = SCN_UP0 + 2 * SCN_UPV, SCN_NPASS = (SCN_NV + 255) / 256, SCN_BUF = SCN_NPASS * 4096;

template <typename T>
__global__ __launch_bounds__(256, 1) void conv_scn_stream_kernel(const ScArgs a, const ScnArgs aa) {
  constexpr int SZ = (int)sizeof(T), K = 9 * 128, PIXB = 64;
  static_assert(SZ == 2, "16-bit storage");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // LDS: [halo buffer 0][statistics scratch [4][32][2]][halo buffer 1 = the weight staging of the prologue (73.7 KB)]
  float* red = reinterpret_cast<float*>(smem + SCN_BUF);
  constexpr int SCN_B1 = SCN_BUF + 4 * 32 * 2 * 4;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  const int Hs = a.H >> 1, Ws = a.W >> 1;

  const int ntiles = a.N * a.tiles_x * a.tiles_y;
  int t_first, t_step, t_end;
  if ((gridDim.x & 7) == 0) {
    const int q = ntiles >> 3, r = ntiles & 7, x = blockIdx.x & 7;
    const int start = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    t_first = start + (int)(blockIdx.x >> 3); t_step = (int)(gridDim.x >> 3); t_end = start + q + (x < r ? 1 : 0);
  } else {
    t_first = (int)blockIdx.x; t_step = (int)gridDim.x; t_end = ntiles;
  }
  // fused BatchNormalization statistics: sums of the STORED values over all tiles of the workgroup, one column per workgroup
  f32x4 ssp[2], qqp[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) { ssp[i] = f32x4{0.f, 0.f, 0.f, 0.f}; qqp[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  auto flush_stats = [&]() __attribute__((always_inline)) {
    if (!a.stats) return;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e_ = 0; e_ < 4; ++e_) {
        const float sv = row_sum16_to_lane15(ssp[i][e_]), qv = row_sum16_to_lane15(qqp[i][e_]);
        if (lr == 15) {
          const int cl = i * 16 + lg * 4 + e_;
          red[(wave * 32 + cl) * 2] = sv;
          red[(wave * 32 + cl) * 2 + 1] = qv;
        }
      }
    lds_barrier();
    if (tid < 32) {
      float sv = 0.f, qv = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) { sv += red[(w * 32 + tid) * 2]; qv += red[(w * 32 + tid) * 2 + 1]; }
      a.stats[(size_t)tid * gridDim.x + blockIdx.x] = sv;                     // [stat][channel][workgroup]
      a.stats[((size_t)32 + tid) * gridDim.x + blockIdx.x] = qv;
    }
  };
  if (t_first >= t_end) {      // (no tile: the statistics column of this workgroup must still be defined)
    flush_stats();
    return;
  }

  const __amdgpu_buffer_rsrc_t rsk = __builtin_amdgcn_make_buffer_rsrc((void*)aa.skip, 0, aa.skip_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rup = __builtin_amdgcn_make_buffer_rsrc((void*)a.src, 0, aa.up_bytes, 0x00020000);
  // staging pass p of this thread: vector v = p * 256 + tid of the buffer image.  Per-thread constants: which tensor, the halo
  // coordinates (border tiles) and the byte offset relative to the tile's first halo pixel (interior tiles: offset = tile base + rel)
  // vector v -> (upsampled tensor?, halo row, halo column, byte offset relative to the first halo pixel); evaluated once per thread
  // for hrel (interior tiles) and again per border tile for the bounds (registers are the scarce resource of this kernel)
  auto vcoord = [&](int v, bool& isup, int& cy, int& cx) __attribute__((always_inline)) -> uint32_t {
    cy = cx = 0; isup = v >= SCN_UP0;
    if (v < 2 * SCN_SKV) {
      const int plane = v / SCN_SKV, r = v - plane * SCN_SKV, px = r >> 2, vec = r & 3;
      cy = px / SC_HW; cx = px - cy * SC_HW;
      return (uint32_t)((cy * a.W + cx) * 128 + plane * 64 + vec * 16);
    }
    const int u = v - SCN_UP0;
    if (u < 0 || u >= 2 * SCN_UPV) return 0x80000000u;      // padding between / behind the planes
    const int plane = u / SCN_UPV, r = u - plane * SCN_UPV, px = r >> 2, vec = r & 3;
    cy = px / SCN_UW; cx = px - cy * SCN_UW;
    return (uint32_t)((cy * Ws + cx) * 128 + plane * 64 + vec * 16);
  };
  uint32_t hrel[SCN_NPASS];
#pragma unroll
  for (int p = 0; p < SCN_NPASS; ++p) {
    bool u_; int cy_, cx_;
    hrel[p] = vcoord(p * 256 + tid, u_, cy_, cx_);
  }
  auto decode = [&](int tile, int& n, int& y0, int& x0) __attribute__((always_inline)) {
    const int bq = (int)fdiv((uint32_t)tile, a.divTx);
    const int tx = tile - bq * a.tiles_x;
    n = (int)fdiv((uint32_t)bq, a.divTy);
    const int ty = bq - n * a.tiles_y;
    y0 = ty * SC_TH; x0 = tx * SC_TW;
  };
  auto issue_tile = [&](int tile, int b) __attribute__((always_inline)) {
    int n, y0, x0;
    decode(tile, n, y0, x0);
    const bool inner = y0 >= 2 && x0 >= 2 && y0 + SC_TH + 2 <= a.H && x0 + SC_TW + 2 <= a.W;      // (uniform) both halos inside
    const uint32_t bsk = (uint32_t)(((n * a.H + y0 - 1) * a.W + x0 - 1) * 128);
    const uint32_t bup = (uint32_t)(((n * Hs + (y0 >> 1) - 1) * Ws + (x0 >> 1) - 1) * 128);
#pragma unroll
    for (int p = 0; p < SCN_NPASS; ++p) {
      const bool wup = (p * 256 + wave * 64) >= SCN_UP0;       // (wave-uniform) this wave's 64 vectors of the pass belong to the upsampled tensor
      uint32_t off = hrel[p] == 0x80000000u ? 0x80000000u : (wup ? bup : bsk) + hrel[p];
      if (!inner && off != 0x80000000u) {
        bool u_; int cy, cx;
        vcoord(p * 256 + tid, u_, cy, cx);
        const int gy = u_ ? (y0 >> 1) - 1 + cy : y0 - 1 + cy, gx = u_ ? (x0 >> 1) - 1 + cx : x0 - 1 + cx;
        if (!((unsigned)gy < (unsigned)(u_ ? Hs : a.H) && (unsigned)gx < (unsigned)(u_ ? Ws : a.W))) off = 0x80000000u;
      }
      if (p * 256 + wave * 64 < SCN_NV) {
        char* dst = smem + (b ? SCN_B1 : 0) + p * 4096 + wave * 1024;
        if (wup) __builtin_amdgcn_raw_ptr_buffer_load_lds(rup, (__attribute__((address_space(3))) void*)dst, 16, off, 0, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsk, (__attribute__((address_space(3))) void*)dst, 16, off, 0, 0, 0);
      }
    }
  };
  issue_tile(t_first, 0);

  // once per workgroup: the weight matrix (73.7 KB) -> LDS by LDS-DMA, fragment-major ([chunk][lane group][32 output channels] 16-byte
  // vectors: conflict-free fragment reads), then ALL of it -> the registers of every wave.  (Each wave fetching its 72 fragments from L2
  // itself: 75 MB of L2 reads per launch.)  chunk c = slice * 9 + tap; slice s = 32 input channels (0, 1: upsampled, 2, 3: skip)
  {
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.weight, 0, 32 * K * SZ, 0x00020000);
#pragma unroll
    for (int p = 0; p < 18; ++p) {      // 4608 vectors = 72 wave instructions: vector ((c * 4 + g) * 32 + co) <- weight[co][k0(c) + 8 g]
      const int wi = p * 4 + wave, grp = wi * 2 + (lane >> 5), c = grp >> 2, g = grp & 3, s_ = c / 9, t = c - s_ * 9;
      const uint32_t off = (uint32_t)(((lane & 31) * K + t * 128 + s_ * 32 + g * 8) * SZ);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(smem + SCN_B1 + wi * 1024), 16, off, 0, 0, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the weights' and the first tile's own pieces have landed
  lds_barrier();
  u32x4 fa[2][36];
#pragma unroll
  for (int c = 0; c < 36; ++c)
#pragma unroll
    for (int i = 0; i < 2; ++i)
      fa[i][c] = *reinterpret_cast<const u32x4*>(smem + SCN_B1 + ((c * 4 + lg) * 32 + i * 16 + lr) * 16);
  // (the wave's rows are part of the lane constant: halo rows 2 wave .., low-resolution rows wave ..)
  const uint32_t lsk = (uint32_t)(lr * PIXB + lg * 16 + wave * 2 * SC_HW * PIXB);
  uint32_t lup[3];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) lup[kw] = (uint32_t)((((lr + kw - 1) >> 1) + 1) * PIXB + lg * 16 + wave * SCN_UW * PIXB);
  const __amdgpu_buffer_rsrc_t rdst = __builtin_amdgcn_make_buffer_rsrc((void*)a.dst, 0, aa.dst_bytes, 0x00020000);
  const uint32_t lvo = (uint32_t)((lr * 32 + lg * 4) * SZ);      // lane part of the output offsets: pixel lr of a 16-pixel row segment, channels 4 lg ..
  f32x4 bias4[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  if (a.bias) { bias4[0] = *reinterpret_cast<const f32x4*>(a.bias + lg * 4); bias4[1] = *reinterpret_cast<const f32x4*>(a.bias + 16 + lg * 4); }
  lds_barrier();               // every wave holds the weights: halo buffer 1 is free

  auto body = [&](int tile, auto curc) {
    constexpr int CUR = decltype(curc)::value;
    int n, y0, x0;
    decode(tile, n, y0, x0);
    const int next = tile + t_step;
    if (next < t_end) issue_tile(next, CUR ^ 1);

    f32x4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int f = 0; f < 4; ++f) acc[i][f] = f32x4{0.f, 0.f, 0.f, 0.f};
    // read sidx = ((slice * 4 + r) * 3 + kw) * 2 + h2: halo row 2 wave + r (r = 0..3), shifted by kw, column half h2
    auto bfrag = [&](int sidx) __attribute__((always_inline)) -> u32x4 {
      const int h2 = sidx & 1, kw = (sidx >> 1) % 3, r = (sidx / 6) & 3, sl = sidx / 24;
      if (sl >= 2) {
        return *reinterpret_cast<const u32x4*>(smem + lsk + ((CUR ? SCN_B1 : 0) + (sl - 2) * SCN_SKV * 16 + (r * SC_HW + h2 * 16 + kw) * PIXB));
      } else {
        // hi-res halo row hyy = 2 wave + r -> low-res row ((hyy - 1) >> 1) + 1 = wave + ((r - 1) >> 1) + 1; column half: + 8 low-res pixels
        const int lrow = ((r + 1) >> 1);      // ((r - 1) >> 1) + 1 for r = 0..3: 0, 1, 1, 2
        return *reinterpret_cast<const u32x4*>(smem + lup[kw] + ((CUR ? SCN_B1 : 0) + SCN_UP0 * 16 + sl * SCN_UPV * 16 + (lrow * SCN_UW + h2 * 8) * PIXB));
      }
    };
    u32x4 fbq[4];
    fbq[0] = bfrag(0); fbq[1] = bfrag(1); fbq[2] = bfrag(2);
#pragma unroll
    for (int sidx = 0; sidx < 96; ++sidx) {
      __builtin_amdgcn_sched_barrier(0);
      if (sidx + 3 < 96) fbq[(sidx + 3) & 3] = bfrag(sidx + 3);
      const int h2 = sidx & 1, kw = (sidx >> 1) % 3, r = (sidx / 6) & 3, sl = sidx / 24;
      const u32x4 fb = fbq[sidx & 3];
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int orow = r - kh;
        if (orow < 0 || orow > 1) continue;
        acc[0][orow * 2 + h2] = mfma16_16x16x32(fa[0][sl * 9 + kh * 3 + kw], fb, acc[0][orow * 2 + h2]);
        acc[1][orow * 2 + h2] = mfma16_16x16x32(fa[1][sl * 9 + kh * 3 + kw], fb, acc[1][orow * 2 + h2]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the next tile's own pieces have landed (before this tile's stores are issued)
    // epilogue: bias, accumulate, ReLU, store (buffer addressing: lane constant + scalar row-segment offset), statistics of the stored values
    const bool full = y0 + SC_TH <= a.H && x0 + SC_TW <= a.W;      // (uniform)
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const int gy = y0 + wave * 2 + (f >> 1), gx = x0 + (f & 1) * 16 + lr;
      const uint32_t so = (uint32_t)(((n * a.H + gy) * a.W + x0 + (f & 1) * 16) * 32) * (uint32_t)SZ;
      const bool ok = full || (gy < a.H && gx < a.W);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const uint32_t vo = ok ? lvo + (uint32_t)(i * 16 * SZ) : 0x80000000u;
        f32x4 v = acc[i][f] + bias4[i];
        if (a.accumulate) {
          const u32x2 w0 = __builtin_amdgcn_raw_buffer_load_b64(rdst, vo, so, 0);
          v += f32x4{h16lo_to_f32(w0.x), h16hi_to_f32(w0.x), h16lo_to_f32(w0.y), h16hi_to_f32(w0.y)};
        }
        if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        const u32x2 o = {pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w)};
        __builtin_amdgcn_raw_buffer_store_b64(o, rdst, vo, so, 0);
        if (a.stats && ok) {
          const f32x4 sv = {h16lo_to_f32(o.x), h16hi_to_f32(o.x), h16lo_to_f32(o.y), h16hi_to_f32(o.y)};
          ssp[i] += sv;
          qqp[i] += sv * sv;
        }
      }
    }
    lds_barrier();        // the next tile is visible to everybody, and everybody has left this one
  };

  for (int tile = t_first; tile < t_end; tile += 2 * t_step) {
    body(tile, std::integral_constant<int, 0>{});
    if (tile + t_step < t_end) body(tile + t_step, std::integral_constant<int, 1>{});
  }
  flush_stats();
}

static int scn_blocks(int ntiles) {
  const int b = sc_cu_count();
  return b < ntiles ? b : ntiles;
}

// Is this convolution the forward served by conv_scn_stream_kernel?  (stp_conv2d consults this before the generic kernels.)
extern "C" int stp_conv2d_scn_eligible(const stp_conv_params* p) {
  static const bool on = !(getenv("STP_SCN") && atoi(getenv("STP_SCN")) == 0);
  if (!on || !p || p->dtype != STP_H16) return 0;
  return p->KH == 3 && p->KW == 3 && p->stride == 1 && p->pad == 1 && p->C0 == 64 && p->C1 == 64 && p->src1 != nullptr && p->Cout == 32 &&
         p->Cd0 == 32 && p->src0_mode == STP_SRC_NEAREST2X && p->Hv == 2 * p->Hs0 && p->Wv == 2 * p->Ws0 && p->Ho == p->Hv && p->Wo == p->Wv &&
         !p->residual && !p->dst_sum2x2 && !p->src_bn_mean && !p->stats_slots && !p->fold_src && !p->bnb_x;
}
extern "C" int stp_conv2d_scn_stats_tiles(const stp_conv_params* p) {
  return scn_blocks(p->N * ceil_div(p->Hv, SC_TH) * ceil_div(p->Wv, SC_TW));
}

extern "C" int stp_conv2d_scn(const stp_conv_params* p, void* stream) {
  if (!stp_conv2d_scn_eligible(p) || !p->src0 || !p->weight || !p->dst0) return STP_E_BADARG;
  ScnArgs aa;
  ScArgs a;
  a.src = (const char*)p->src0; a.weight = (const char*)p->weight; a.bias = p->bias; a.dst = (char*)p->dst0;
  a.N = p->N; a.H = p->Hv; a.W = p->Wv; a.Hs = p->Hs0; a.Ws = p->Ws0; a.Cout = p->Cout;
  a.up = 1; a.accumulate = p->accumulate0; a.relu = p->relu;
  aa.skip = (const char*)p->src1;
  {
    const uint64_t ub = (uint64_t)p->N * p->Hs0 * p->Ws0 * 64 * 2, kb = (uint64_t)p->N * p->Hv * p->Wv * 64 * 2;
    if (kb >= 0x80000000ull) return STP_E_BADARG;   // 32-bit LDS-DMA offsets
    aa.up_bytes = (uint32_t)ub; aa.skip_bytes = (uint32_t)kb; aa.dst_bytes = (uint32_t)((uint64_t)p->N * p->Hv * p->Wv * 32 * 2);
    a.src_bytes = aa.up_bytes;
  }
  a.tiles_x = ceil_div(a.W, SC_TW); a.tiles_y = ceil_div(a.H, SC_TH);
  a.divTx = make_fastdiv((uint32_t)a.tiles_x); a.divTy = make_fastdiv((uint32_t)a.tiles_y);
  a.stats = p->stats_partial;
  a.stat_slots = 0;
  a.bnb.x = nullptr; a.bnb.mean = nullptr; a.bnb.rstd = nullptr; a.bnb.gamma = nullptr; a.bnb.beta = nullptr; a.bnb.relu = 0;
  a.sum2 = 0;
  a.pbn.x = nullptr; a.pbn.mean = nullptr; a.pbn.rstd = nullptr; a.pbn.gamma = nullptr; a.pbn.beta = nullptr; a.pbn.relu = 0;
  const int ntiles = a.N * a.tiles_x * a.tiles_y;
  const int blocks = scn_blocks(ntiles);
  const_cast<stp_conv_params*>(p)->stats_tiles = blocks;
  const size_t lds = (size_t)SCN_BUF + 4 * 32 * 2 * 4 + 32 * 9 * 128 * 2;      // the weight staging (73.7 KB) covers halo buffer 1 (60 KB)
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_scn_stream_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return STP_E_LAUNCH;
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_scn_stream_kernel<bf16_t>), dim3(blocks), dim3(256), lds, (hipStream_t)stream, a, aa);
  STP_LAUNCH_CHECK();
  return STP_OK;
}

// =================================================================================================
// 64 -> 64 channels, 3x3 / stride 1 (ResNet stage 1 and decoder_stage2_conv2 at 16 x 128 x 128: 14 launches of the U-Net step, forward
// and data gradient): the weight matrix lives in REGISTERS - every wave keeps 32 output channels x 576 = 144 registers of A fragments
// (8 waves: 2 channel halves x 4 row pairs of the 8 x 32 tile, two waves per SIMD) - so nothing but the halo moves through LDS
// (2 x 32-channel planes of 10 x 34 pixels, double-buffered by LDS-DMA one tile ahead) and a pixel fragment is read once for up to 6
// MFMAs (2 channel tiles x the taps (kh, kw) of the output rows r - kh).  conv_halo_kernel<16, 64, 1, 8> streams the weights through a
// ring for every 256-pixel tile (73 KB per tile = 64 % of its L2 -> LDS bytes) and spends 43 % of a workgroup's life outside its K
// loop: 31-39 us per layer, 500-600 TFLOP/s; the layer moves 80-115 MB (16-23 us at 5 TB/s) and needs 10 us of MFMA time.
// (First version: ALL 288 weight registers per wave at one wave per SIMD, as in the narrow-output kernel above: 47 us - with a single
//  wave per SIMD the MFMA phase ran at 39 % and the LDS-DMA of the next tile cost 4 us per tile on top; what-if builds in DESIGN 3.2b.)
// Epilogue from the accumulators (8-byte buffer stores): residual | fused BatchNormalization statistics of the stored values | fused
// BatchNormalization-backward mask + sums (stp_conv_params.bnb_x); the sums run over all tiles of the persistent workgroup (per-thread
// accumulators in LDS) and are written once, one column per workgroup.
// =================================================================================================
struct S64Args {        // (next to a ScArgs: src, weight [64][576], dst, N, H, W, stats, bnb, tiles)
  const char* residual;
  uint32_t io_bytes;    // size of src = dst = residual = bnb.x
};
constexpr int S64_NT = 512, S64_PLV = SC_HH * SC_HW * 4, S64_NV = 2 * S64_PLV, S64_NPASS = (S64_NV + S64_NT - 1) / S64_NT, S64_BUF = S64_NPASS * S64_NT * 16;

template <typename T>
__global__ __launch_bounds__(S64_NT, 2) void conv_s64_stream_kernel(const ScArgs a, const S64Args aa) {
  constexpr int SZ = (int)sizeof(T), K = 9 * 64, PIXB = 64;
  static_assert(SZ == 2, "16-bit storage");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // LDS: [halo buffer 0][tables][halo buffer 1 = the weight staging of the prologue (73.7 KB)]
  float* wsum = reinterpret_cast<float*>(smem + S64_BUF);          // [8 waves][32 channels][2]: running sums of the fused statistics
  float* ktab = wsum + 8 * 32 * 2;                                 // [4][64]: scale, shift, mean, rstd of the fused BatchNormalization backward
  uint32_t* hrel = reinterpret_cast<uint32_t*>(ktab + 256);        // [S64_NPASS][512 threads]: LDS-DMA offsets relative to the first halo pixel
  constexpr int S64_B1 = S64_BUF + (8 * 32 * 2 + 256) * 4 + S64_NPASS * S64_NT * 4;      // byte offset of halo buffer 1
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave = wv & 3, cg = wv >> 2;                           // row pair of the tile, channel half
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
  const bool bnb = a.bnb.x != nullptr;
  if (tid < 8 * 32 * 2) wsum[tid] = 0.f;
  auto flush_stats = [&]() __attribute__((always_inline)) {
    if (!a.stats) return;
    lds_barrier();
    if (tid < 64) {
      float sv = 0.f, qv = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) { sv += wsum[(((tid >> 5) * 4 + w) * 32 + (tid & 31)) * 2]; qv += wsum[(((tid >> 5) * 4 + w) * 32 + (tid & 31)) * 2 + 1]; }
      a.stats[(size_t)tid * gridDim.x + blockIdx.x] = sv;                     // [stat][channel][workgroup]
      a.stats[((size_t)64 + tid) * gridDim.x + blockIdx.x] = qv;
    }
  };
  if (t_first >= t_end) {      // (no tile: the statistics column of this workgroup must still be defined)
    flush_stats();
    return;
  }

  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.src, 0, aa.io_bytes, 0x00020000);
  // staging: vector v = p * 512 + tid of the buffer image [plane][10 x 34 pixels][4 vectors]; byte offset relative to the first halo pixel
  auto vcoord = [&](int v, int& cy, int& cx) __attribute__((always_inline)) -> uint32_t {
    cy = cx = 0;
    if (v >= S64_NV) return 0x80000000u;
    const int plane = v / S64_PLV, r = v - plane * S64_PLV, px = r >> 2, vec = r & 3;
    cy = px / SC_HW; cx = px - cy * SC_HW;
    return (uint32_t)((cy * a.W + cx) * 128 + plane * 64 + vec * 16);
  };
#pragma unroll
  for (int p = 0; p < S64_NPASS; ++p) {
    int cy_, cx_;
    hrel[p * S64_NT + tid] = vcoord(p * S64_NT + tid, cy_, cx_);
  }
  auto decode = [&](int tile, int& n, int& y0, int& x0) __attribute__((always_inline)) {
    const int bq = (int)fdiv((uint32_t)tile, a.divTx);
    const int tx = tile - bq * a.tiles_x;
    n = (int)fdiv((uint32_t)bq, a.divTy);
    const int ty = bq - n * a.tiles_y;
    y0 = ty * SC_TH; x0 = tx * SC_TW;
  };
  auto issue_tile = [&](int tile, int b) __attribute__((always_inline)) {
    int n, y0, x0;
    decode(tile, n, y0, x0);
    const bool inner = y0 >= 1 && x0 >= 1 && y0 + SC_TH + 1 <= a.H && x0 + SC_TW + 1 <= a.W;      // (uniform) the halo lies inside the image
    const uint32_t base = (uint32_t)(((n * a.H + y0 - 1) * a.W + x0 - 1) * 128);
#pragma unroll
    for (int p = 0; p < S64_NPASS; ++p) {
      const uint32_t hr = hrel[p * S64_NT + tid];
      uint32_t off = hr == 0x80000000u ? 0x80000000u : base + hr;
      if (!inner && off != 0x80000000u) {
        int cy, cx;
        vcoord(p * S64_NT + tid, cy, cx);
        if (!((unsigned)(y0 - 1 + cy) < (unsigned)a.H && (unsigned)(x0 - 1 + cx) < (unsigned)a.W)) off = 0x80000000u;
      }
      if (p * S64_NT + wv * 64 < S64_NV)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + (b ? S64_B1 : 0) + p * (S64_NT * 16) + wv * 1024), 16, off, 0, 0, 0);
    }
  };
  issue_tile(t_first, 0);

  // once per workgroup: the weight matrix (73.7 KB) -> LDS by LDS-DMA, fragment-major ([chunk][lane group][64 output channels] 16-byte
  // vectors: conflict-free fragment reads), then this wave's 32 output channels -> registers.  (Every wave fetching its fragments from
  // L2 itself = 75 MB of L2 reads per launch for 4 tiles per workgroup: 9 us of a 35 us launch.)  chunk c = slice * 9 + tap
  {
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.weight, 0, 64 * K * SZ, 0x00020000);
#pragma unroll
    for (int p = 0; p < 9; ++p) {      // 4608 vectors = 72 wave instructions: vector (c * 4 + g) * 64 + co <- weight[co][k0(c) + 8 g]
      const int wi = p * 8 + wv, c = wi >> 2, g = wi & 3, s_ = c / 9, t = c - s_ * 9;
      const uint32_t off = (uint32_t)((lane * K + t * 64 + s_ * 32 + g * 8) * SZ);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(smem + S64_B1 + wi * 1024), 16, off, 0, 0, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the weights' and the first tile's own pieces have landed
  lds_barrier();
  u32x4 fa[2][18];
#pragma unroll
  for (int c = 0; c < 18; ++c)
#pragma unroll
    for (int i = 0; i < 2; ++i)
      fa[i][c] = *reinterpret_cast<const u32x4*>(smem + S64_B1 + ((c * 4 + lg) * 64 + cg * 32 + i * 16 + lr) * 16);
  const uint32_t lsk = (uint32_t)(lr * PIXB + lg * 16 + wave * 2 * SC_HW * PIXB);       // lane part of every fragment address (halo rows 2 wave ..)
  const __amdgpu_buffer_rsrc_t rdst = __builtin_amdgcn_make_buffer_rsrc((void*)a.dst, 0, aa.io_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rop = __builtin_amdgcn_make_buffer_rsrc((void*)(bnb ? a.bnb.x : (aa.residual ? aa.residual : a.dst)), 0, aa.io_bytes, 0x00020000);
  const bool hasop = bnb || aa.residual != nullptr;
  const uint32_t lvo = (uint32_t)((lr * 64 + cg * 32 + lg * 4) * SZ);      // lane part of the output offsets: pixel lr of a 16-pixel row segment, channels 32 cg + 4 lg ..
  if (bnb && tid < 64) {
    const float mu = a.bnb.mean[tid], rsd = a.bnb.rstd[tid];
    const float sc = a.bnb.gamma ? rsd * a.bnb.gamma[tid] : rsd;
    ktab[tid] = sc; ktab[64 + tid] = (a.bnb.beta ? a.bnb.beta[tid] : 0.f) - mu * sc; ktab[128 + tid] = mu; ktab[192 + tid] = rsd;
  }
  lds_barrier();               // every wave holds its weights: halo buffer 1 is free; the tables are visible

  auto body = [&](int tile, auto curc) {
    constexpr int CUR = decltype(curc)::value;
    int n, y0, x0;
    decode(tile, n, y0, x0);
    const bool full = y0 + SC_TH <= a.H && x0 + SC_TW <= a.W;      // (uniform)
    const int next = tile + t_step;
    if (next < t_end) issue_tile(next, CUR ^ 1);

    f32x4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int f = 0; f < 4; ++f) acc[i][f] = f32x4{0.f, 0.f, 0.f, 0.f};
    // read sidx = ((slice * 4 + r) * 3 + kw) * 2 + h2: halo row 2 wave + r (r = 0..3), shifted by kw, column half h2
    auto bfrag = [&](int sidx) __attribute__((always_inline)) -> u32x4 {
      const int h2 = sidx & 1, kw = (sidx >> 1) % 3, r = (sidx / 6) & 3, sl = sidx / 24;
      return *reinterpret_cast<const u32x4*>(smem + lsk + ((CUR ? S64_B1 : 0) + sl * S64_PLV * 16 + (r * SC_HW + h2 * 16 + kw) * PIXB));
    };
    u32x4 fbq[4];
    fbq[0] = bfrag(0); fbq[1] = bfrag(1); fbq[2] = bfrag(2);
#pragma unroll
    for (int sidx = 0; sidx < 48; ++sidx) {
      __builtin_amdgcn_sched_barrier(0);
      if (sidx + 3 < 48) fbq[(sidx + 3) & 3] = bfrag(sidx + 3);
      const int h2 = sidx & 1, kw = (sidx >> 1) % 3, r = (sidx / 6) & 3, sl = sidx / 24;
      const u32x4 fb = fbq[sidx & 3];
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int orow = r - kh;
        if (orow < 0 || orow > 1) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i][orow * 2 + h2] = mfma16_16x16x32(fa[i][sl * 9 + kh * 3 + kw], fb, acc[i][orow * 2 + h2]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the next tile's own pieces (and this tile's operands) have landed
    // epilogue.  Operands (residual, or the BatchNormalization input of the fused backward) are fetched one fragment ahead
    auto opload = [&](int f, u32x2 (&o)[2]) __attribute__((always_inline)) {
      const int gy = y0 + wave * 2 + (f >> 1), gx = x0 + (f & 1) * 16 + lr;
      const uint32_t so = (uint32_t)(((n * a.H + gy) * a.W + x0 + (f & 1) * 16) * 64) * (uint32_t)SZ;
      const bool ok = full || (gy < a.H && gx < a.W);
#pragma unroll
      for (int i = 0; i < 2; ++i) o[i] = __builtin_amdgcn_raw_buffer_load_b64(rop, ok ? lvo + (uint32_t)(i * 16 * SZ) : 0x80000000u, so, 0);
    };
    u32x2 opq[2][2];
    if (hasop) opload(0, opq[0]);
    f32x4 ssl[2], qql[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) { ssl[i] = f32x4{0.f, 0.f, 0.f, 0.f}; qql[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      if (hasop && f + 1 < 4) opload(f + 1, opq[(f + 1) & 1]);
      const int gy = y0 + wave * 2 + (f >> 1), gx = x0 + (f & 1) * 16 + lr;
      const uint32_t so = (uint32_t)(((n * a.H + gy) * a.W + x0 + (f & 1) * 16) * 64) * (uint32_t)SZ;
      const bool ok = full || (gy < a.H && gx < a.W);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const uint32_t vo = ok ? lvo + (uint32_t)(i * 16 * SZ) : 0x80000000u;
        f32x4 v = acc[i][f];
        const u32x2 ow = opq[f & 1][i];
        const f32x4 op = {h16lo_to_f32(ow.x), h16hi_to_f32(ow.x), h16lo_to_f32(ow.y), h16hi_to_f32(ow.y)};
        if (!bnb && hasop) v += op;
        u32x2 o = {pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w)};
        f32x4 sv = {h16lo_to_f32(o.x), h16hi_to_f32(o.x), h16lo_to_f32(o.y), h16hi_to_f32(o.y)};      // as stored
        if (bnb) {
          const int co = cg * 32 + i * 16 + lg * 4;
          BnBackCh kk;
          kk.sc = *reinterpret_cast<const f32x4*>(ktab + co); kk.sh = *reinterpret_cast<const f32x4*>(ktab + 64 + co);
          kk.mu = *reinterpret_cast<const f32x4*>(ktab + 128 + co); kk.rs = *reinterpret_cast<const f32x4*>(ktab + 192 + co);
          f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, q0 = s0;
          const f32x4 g = bnback_apply(kk, a.bnb.relu, op, sv, s0, q0);
          if (ok) { ssl[i] += s0; qql[i] += q0; }
          o = u32x2{pack_bf16x2(g.x, g.y), pack_bf16x2(g.z, g.w)};
        } else if (a.stats && ok) {
          ssl[i] += sv;
          qql[i] += sv * sv;
        }
        __builtin_amdgcn_raw_buffer_store_b64(o, rdst, vo, so, 0);
      }
    }
    if (a.stats) {      // this wave's 32 channels: row reduction, lane 15 of each row adds into the wave's running sums
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e_ = 0; e_ < 4; ++e_) {
          const float sv = row_sum16_to_lane15(ssl[i][e_]), qv = row_sum16_to_lane15(qql[i][e_]);
          if (lr == 15) {
            float* ws = wsum + ((wv * 32) + i * 16 + lg * 4 + e_) * 2;
            ws[0] += sv; ws[1] += qv;
          }
        }
    }
    lds_barrier();        // the next tile is visible to everybody, and everybody has left this one
  };

  for (int tile = t_first; tile < t_end; tile += 2 * t_step) {
    body(tile, std::integral_constant<int, 0>{});
    if (tile + t_step < t_end) body(tile + t_step, std::integral_constant<int, 1>{});
  }
  flush_stats();
}

static int s64_blocks(int ntiles) {
  const int b = sc_cu_count();
  return b < ntiles ? b : ntiles;
}

// Can conv_s64_stream_kernel serve this convolution?  NOT used automatically (STP_S64=1 makes stp_conv2d prefer it to the halo kernel; an
// explicit tile id 736 always works): measured on 16 x 128 x 128 against conv_halo_kernel<16, 64, 1, 8> - plain 30.1 vs 33.3 us, with the
// fused statistics 32.6 vs 34.6, residual + statistics 35.5 vs 35.7, BatchNormalization-backward sums 42.4 vs 37.1 (its epilogue works
// from the accumulator layout: 8-byte accesses); at batch 64 (16 tiles per workgroup instead of 4) 99.8 vs 128.8 us = 775 vs 600 TFLOP/s:
// the prologue (weights -> LDS -> registers, 7 us) is what four tiles per workgroup cannot amortise.
extern "C" int stp_conv2d_s64_eligible(const stp_conv_params* p) {
  if (!p || p->dtype != STP_H16) return 0;
  return p->KH == 3 && p->KW == 3 && p->stride == 1 && p->pad == 1 && p->C0 == 64 && p->C1 == 0 && p->Cout == 64 && p->Cd0 == 64 && !p->dst1 &&
         p->src0_mode == STP_SRC_DIRECT && p->Hs0 == p->Hv && p->Ws0 == p->Wv && p->Ho == p->Hv && p->Wo == p->Wv && !p->bias && !p->relu &&
         !p->accumulate0 && !p->dst_sum2x2 && !p->src_bn_mean && !p->stats_slots && !p->fold_src && !(p->bnb_x && p->residual) &&
         (int64_t)p->N * ceil_div(p->Hv, SC_TH) * ceil_div(p->Wv, SC_TW) >= 2 * sc_cu_count();      // two tiles per workgroup at least: the weights load once per workgroup
}
extern "C" int stp_conv2d_s64_stats_tiles(const stp_conv_params* p) {
  return s64_blocks(p->N * ceil_div(p->Hv, SC_TH) * ceil_div(p->Wv, SC_TW));
}

extern "C" int stp_conv2d_s64(const stp_conv_params* p, void* stream) {
  if (!stp_conv2d_s64_eligible(p) || !p->src0 || !p->weight || !p->dst0) return STP_E_BADARG;
  S64Args aa;
  ScArgs a;
  a.src = (const char*)p->src0; a.weight = (const char*)p->weight; a.bias = nullptr; a.dst = (char*)p->dst0;
  a.N = p->N; a.H = p->Hv; a.W = p->Wv; a.Hs = p->Hs0; a.Ws = p->Ws0; a.Cout = 64;
  a.up = 0; a.accumulate = 0; a.relu = 0;
  aa.residual = (const char*)p->residual;
  {
    const uint64_t ib = (uint64_t)p->N * p->Hv * p->Wv * 64 * 2;
    if (ib >= 0x80000000ull) return STP_E_BADARG;   // 32-bit buffer offsets
    aa.io_bytes = (uint32_t)ib; a.src_bytes = aa.io_bytes;
  }
  a.tiles_x = ceil_div(a.W, SC_TW); a.tiles_y = ceil_div(a.H, SC_TH);
  a.divTx = make_fastdiv((uint32_t)a.tiles_x); a.divTy = make_fastdiv((uint32_t)a.tiles_y);
  a.stats = p->stats_partial;
  a.stat_slots = 0;
  a.bnb.x = (const char*)p->bnb_x; a.bnb.mean = p->bnb_mean; a.bnb.rstd = p->bnb_rstd; a.bnb.gamma = p->bnb_gamma;
  a.bnb.beta = p->bnb_beta; a.bnb.relu = p->bnb_relu;
  if (a.bnb.x && (!a.stats || !a.bnb.mean || !a.bnb.rstd)) return STP_E_BADARG;
  a.sum2 = 0;
  a.pbn.x = nullptr; a.pbn.mean = nullptr; a.pbn.rstd = nullptr; a.pbn.gamma = nullptr; a.pbn.beta = nullptr; a.pbn.relu = 0;
  const int ntiles = a.N * a.tiles_x * a.tiles_y;
  const int blocks = s64_blocks(ntiles);
  const_cast<stp_conv_params*>(p)->stats_tiles = blocks;
  const size_t lds = (size_t)S64_BUF + (8 * 32 * 2 + 256) * 4 + S64_NPASS * S64_NT * 4 + 64 * 9 * 64 * 2;      // the weight staging (73.7 KB) covers halo buffer 1
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_s64_stream_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return STP_E_LAUNCH;
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_s64_stream_kernel<bf16_t>), dim3(blocks), dim3(S64_NT), lds, (hipStream_t)stream, a, aa);
  STP_LAUNCH_CHECK();
  return STP_OK;
}

// =================================================================================================
// Stem: 7x7 / stride 2 / pad 3 convolution of the 4-channel (3 image channels + 1) bf16 input to 64 channels (ResNet conv0).
// The generic implicit GEMM gathers 49 taps of 8 bytes per output pixel through the vector-memory path; this layer is
// HBM-bound (33 MB in, 134 MB out at 16x512x512), so the same halo-tile scheme as above is used: the (2*8+5) x (2*32+5)
// input patch of an 8 x 32 output tile is staged once (rows padded to 70 pixels = a 16-byte multiple) next to the weights
// [64][7][8][4] (kw padded to 8 by the prepare kernel: K = 224 = 7 MFMA chunks, chunk = kh; LDS rows padded to 464 bytes =
// an odd number of 16-byte units, conflict-free for the 16 rows of an A fragment), and a B fragment is ONE ds_read_b128:
// lane group g of chunk kh reads taps kw = 2g, 2g+1 (2 pixels x 4 channels, contiguous).  (Weights in registers - 112 VGPRs -
// leave one wave per SIMD; from LDS the kernel runs 4.)
// =================================================================================================
__global__ __launch_bounds__(256) void conv_stem_kernel(const StemArgs a) {
  constexpr int TM = 4, NCH = 7, K = 224, COUT = 64;
  extern __shared__ __attribute__((aligned(16))) char halo[];  // [ST_HH][ST_HW] pixels of 8 bytes, then the stats scratch
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  int b = blockIdx.x;
  const int bq = (int)fdiv((uint32_t)b, a.divTx);
  const int tx = b - bq * a.tiles_x;
  const int n = (int)fdiv((uint32_t)bq, a.divTy);
  const int ty = bq - n * a.tiles_y;
  const int y0 = ty * SC_TH, x0 = tx * SC_TW;
  const int iy0 = 2 * y0 - 3, ix0 = 2 * x0 - 3;

  const char* img = a.src + (size_t)n * a.H * a.W * 8;
  for (int v = tid; v < ST_HH * ST_HW; v += 256) {
    const int hy = v / ST_HW, hx = v - hy * ST_HW;
    const int gy = iy0 + hy, gx = ix0 + hx;
    u32x2 val = {0u, 0u};
    if ((unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W) val = *reinterpret_cast<const u32x2*>(img + ((size_t)gy * a.W + gx) * 8);
    *reinterpret_cast<u32x2*>(halo + v * 8) = val;
  }
  char* wl = halo + ST_HALO;  // [64][ST_WROW]
  for (int v = tid; v < COUT * (K / 8); v += 256) {
    const int row = v / (K / 8), q = v - row * (K / 8);
    *reinterpret_cast<u32x4*>(wl + row * ST_WROW + q * 16) = *reinterpret_cast<const u32x4*>(a.weight + ((size_t)row * K + q * 8) * 2);
  }
  __syncthreads();

  f32x4 acc[TM][4];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int f = 0; f < 4; ++f) acc[i][f] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1   // (unrolled, the compiler hoists all 28 A fragments: 112 VGPRs and one wave per SIMD)
  for (int c = 0; c < NCH; ++c) {
    u32x4 fa[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const u32x4*>(wl + (i * 16 + lr) * ST_WROW + c * 64 + lg * 16);
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const int py = wave * 2 + (f >> 1), px = (f & 1) * 16 + lr;
      const u32x4 fb = *reinterpret_cast<const u32x4*>(halo + ((2 * py + c) * ST_HW + 2 * px + 2 * lg) * 8);
#pragma unroll
      for (int i = 0; i < TM; ++i) ScMma<bf16_t>::run(fa[i], fb, acc[i][f]);
    }
  }
  __syncthreads();   // the stats scratch below reuses the weight region

  // channel tile OUTER: one pair of statistics accumulators live at a time (register pressure decides the occupancy here)
  bf16_t* out = reinterpret_cast<bf16_t*>(a.dst);
  float* red = reinterpret_cast<float*>(halo + ST_HALO);  // [4][64][2], over the (dead) weights
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    f32x4 ss = {0.f, 0.f, 0.f, 0.f}, qq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const int gy = y0 + wave * 2 + (f >> 1), gx = x0 + (f & 1) * 16 + lr;
      if (gy >= a.Ho || gx >= a.Wo) continue;
      const size_t pm = ((size_t)n * a.Ho + gy) * a.Wo + gx;
      const f32x4 v = acc[i][f];
      store4(out + pm * COUT + i * 16 + lg * 4, v);
      const f32x4 sv = sc_stored(v, (const bf16_t*)nullptr);
      ss += sv;
      qq += sv * sv;
    }
    if (a.stats) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float sv = row_sum16_to_lane15(ss[e]), qv = row_sum16_to_lane15(qq[e]);
        if (lr == 15) {
          const int cl = i * 16 + lg * 4 + e;
          red[(wave * COUT + cl) * 2] = sv;
          red[(wave * COUT + cl) * 2 + 1] = qv;
        }
      }
    }
  }
  if (a.stats) {
    __syncthreads();
    if (tid < COUT) {
      float sv = 0.f, qv = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) { sv += red[(w * COUT + tid) * 2]; qv += red[(w * COUT + tid) * 2 + 1]; }
      a.stats[(size_t)tid * gridDim.x + blockIdx.x] = sv;                     // [stat][channel][tile]
      a.stats[((size_t)COUT + tid) * gridDim.x + blockIdx.x] = qv;
    }
  }
}

extern "C" int stp_conv2d_stem_eligible(const stp_conv_params* p) {
  if (!p) return 0;
  return p->dtype == STP_H16 && p->C0 == 4 && p->C1 == 0 && p->KH == 7 && p->KW == 8 && p->stride == 2 && p->pad == 3 && p->Cout == 64 &&
         p->Cd0 == 64 && p->src0_mode == STP_SRC_DIRECT && p->Hs0 == p->Hv && p->Ws0 == p->Wv && p->Ho == (p->Hv - 1) / 2 + 1 &&
         p->Wo == (p->Wv - 1) / 2 + 1 && !p->bias && !p->residual && !p->relu && !p->accumulate0 && !p->bnb_x && !p->dst_sum2x2 &&
         !p->stats_slots && !p->src_bn_mean;
}

// columns of the [stat][channel][column] partial sums: one per workgroup of the persistent kernel, one per tile of the single-shot one
extern "C" int stp_conv2d_stem_stats_tiles(const stp_conv_params* p) {
  const int ntiles = p->N * ceil_div(p->Ho, SC_TH) * ceil_div(p->Wo, SC_TW);
  return stem_lean_serves(p->N, p->Hv, p->Wv) ? stem_lean_blocks(ntiles) : ntiles;
}

extern "C" int stp_conv2d_stem(const stp_conv_params* p, void* stream) {
  if (!stp_conv2d_stem_eligible(p) || !p->src0 || !p->weight || !p->dst0) return STP_E_BADARG;
  StemArgs a;
  a.src = (const char*)p->src0; a.weight = (const char*)p->weight; a.dst = (char*)p->dst0;
  a.N = p->N; a.H = p->Hv; a.W = p->Wv; a.Ho = p->Ho; a.Wo = p->Wo;
  a.tiles_x = ceil_div(a.Wo, SC_TW); a.tiles_y = ceil_div(a.Ho, SC_TH);
  a.divTx = make_fastdiv((uint32_t)a.tiles_x); a.divTy = make_fastdiv((uint32_t)a.tiles_y);
  a.stats = p->stats_partial;
  const_cast<stp_conv_params*>(p)->stats_tiles = stp_conv2d_stem_stats_tiles(p);
  {
    const uint64_t sb = (uint64_t)p->N * p->Hv * p->Wv * 8;
    a.src_bytes = sb < 0x80000000ull ? (uint32_t)sb : 0u;
    const int r = stem_lean_launch(a, (hipStream_t)stream);      // the persistent form (conv_sc_lean.hip) where it serves the shape
    if (r != 1) return r;
  }
  const size_t lds = (size_t)ST_HALO + ST_WBYTES;
  hipLaunchKernelGGL(conv_stem_kernel, dim3(a.N * a.tiles_x * a.tiles_y), dim3(256), lds, (hipStream_t)stream, a);
  STP_LAUNCH_CHECK();
  return STP_OK;
}

// =================================================================================================
// Small-channel weight gradient:  dW[co][tap*CIN + ci] = sum_pixels dY[p][co] * X[p + tap][ci]
// Persistent workgroups walk 8x32 tiles; per tile the X halo tile and the dY tile are staged in LDS.
// The reduction index (pixels) is the slow dimension of both tiles, so bf16 fragments come from the
// transpose read ds_read_b64_tr_b16 ([4 pixels][16 channels] per 16-lane group); fp32 uses one element
// per lane.  A 32-pixel MFMA chunk is one tile row.  The 9 taps are split over the 4 waves (wave w owns
// taps w, w+4, w+8), so no cross-wave reduction is needed and accumulators stay in registers across
// tiles.  Each workgroup writes one fp32 slab; the generic fixed-order reduce kernel sums them.
// =================================================================================================
typedef short s16x4_t __attribute__((ext_vector_type(4)));


// STREAMING form of conv_sc_wgrad_kernel below (the one stp_wgrad_sc_partial launches; the register-staged form stays for A/B runs,
// STP_SC_STREAM=0): the X halo tile and the dY tile of the NEXT tile are written into the other half of a double buffer by LDS-DMA
// (out-of-image pixels = out-of-range offset = zeros) while the current tile is multiplied; the fused producer BatchNormalization
// normalises the staged halo vectors in LDS (same fma / activation / rounding).  Same MFMA order: bit-identical slabs.
template <typename T, int CIN, int COUT>
__global__ __launch_bounds__(256) void conv_sc_wgrad_stream_kernel(const ScWgArgs a) {
  constexpr int SZ = (int)sizeof(T);
  constexpr int VEC = Elem<T>::VEC;
  constexpr int TMo = (COUT + 15) / 16, TNi = CIN / 16;
  constexpr int PIXB = CIN * SZ, DYB = COUT * SZ;
  constexpr int HALO_BYTES = SC_HH * SC_HW * PIXB;
  static_assert(CIN % 16 == 0 && COUT % VEC == 0, "channel granularity");

  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][halo tile | dY tile], each padded to whole 1 KB wave pieces
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  const int sh = a.up ? 1 : 0;

  static_assert(256 % (CIN / VEC) == 0, "a thread stages the same channel vector in every pass");
  const bool pbn = a.pbn.mean != nullptr;
  ScStageBn<T> sbn;
  if (pbn) sbn.load(a.pbn, (tid % (CIN / VEC)) * VEC);

  f32x4 acc[3][TMo][TNi];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int i = 0; i < TMo; ++i)
#pragma unroll
      for (int j = 0; j < TNi; ++j) acc[t][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- staging: LDS-DMA into a double buffer; the tile of the NEXT iteration streams in while this one is multiplied --------
  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)a.src, 0, a.src_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, a.dy_bytes, 0x00020000);
  constexpr int VPPX = CIN / VEC, VPPD = COUT / VEC;
  constexpr int NVX = SC_HH * SC_HW * VPPX, NPX = (NVX + 255) / 256;      // halo vectors / passes
  constexpr int NVD = SC_TH * SC_TW * VPPD, NPD = (NVD + 255) / 256;      // dY vectors / passes
  constexpr int XBUF = NPX * 4096, DBUF = NPD * 4096, BUF = XBUF + DBUF;  // whole 1 KB wave pieces
  int hyx[NPX], pyx[NPD];
#pragma unroll
  for (int p = 0; p < NPX; ++p) {
    const int v = p * 256 + tid, pix = v / VPPX;
    const int hy = pix / SC_HW, hx = pix - hy * SC_HW;
    hyx[p] = v < NVX ? (hy << 16 | hx) : -1;
  }
#pragma unroll
  for (int p = 0; p < NPD; ++p) {
    const int v = p * 256 + tid, pix = v / VPPD;
    pyx[p] = v < NVD ? ((pix / SC_TW) << 16 | (pix % SC_TW)) : -1;
  }
  const uint32_t cvx = (uint32_t)(tid % VPPX) * 16u, cvd = (uint32_t)(tid % VPPD) * 16u;
  auto decode = [&](int tile, int& n, int& y0, int& x0) {
    int b = tile;
    const int tx = b % a.tiles_x; b /= a.tiles_x;
    const int ty = b % a.tiles_y;
    n = b / a.tiles_y;
    y0 = ty * SC_TH; x0 = tx * SC_TW;
  };
  auto issue_tile = [&](int tile, int bsel) -> uint32_t {
    int n, y0, x0;
    decode(tile, n, y0, x0);
    uint32_t inside = 0;
    char* xb = smem + bsel * BUF;
#pragma unroll
    for (int p = 0; p < NPX; ++p) {
      const int gy = y0 - 1 + (hyx[p] >> 16), gx = x0 - 1 + (hyx[p] & 0xffff);
      const bool ok = hyx[p] >= 0 && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
      const uint32_t off = ok ? (uint32_t)((n * a.Hs + (gy >> sh)) * a.Ws + (gx >> sh)) * (uint32_t)PIXB + cvx : 0x80000000u;
      inside |= ok ? (1u << p) : 0u;
      if (p * 256 + wave * 64 < NVX)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (__attribute__((address_space(3))) void*)(xb + p * 4096 + wave * 1024), 16, off, 0, 0, 0);
    }
#pragma unroll
    for (int p = 0; p < NPD; ++p) {
      const int gy = y0 + (pyx[p] >> 16), gx = x0 + (pyx[p] & 0xffff);
      const bool ok = pyx[p] >= 0 && gy < a.H && gx < a.W;
      const uint32_t off = ok ? (uint32_t)((n * a.H + gy) * a.W + gx) * (uint32_t)DYB + cvd : 0x80000000u;
      if (p * 256 + wave * 64 < NVD)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsd, (__attribute__((address_space(3))) void*)(xb + XBUF + p * 4096 + wave * 1024), 16, off, 0, 0, 0);
    }
    return inside;
  };

  int cur = 0;
  uint32_t inside_cur = (int)blockIdx.x < a.ntiles ? issue_tile((int)blockIdx.x, 0) : 0u;
  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    char* halo = smem + cur * BUF;
    char* dyt = halo + XBUF;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // own pieces of this tile have landed
    if (pbn) {
#pragma unroll
      for (int p = 0; p < NPX; ++p)
        if (inside_cur & (1u << p)) {                      // padding applies to the NORMALISED tensor
          u32x4* vp = reinterpret_cast<u32x4*>(halo + (p * 256 + tid) * 16);
          *vp = sbn.apply(*vp, a.pbn.relu);
        }
    }
    lds_barrier();                                          // tile visible; every wave has left the previous tile's buffer
    if (tile + (int)gridDim.x < a.ntiles) inside_cur = issue_tile(tile + (int)gridDim.x, cur ^ 1);
    cur ^= 1;

#pragma unroll 1
    for (int row = 0; row < SC_TH; ++row) {  // one 32-pixel chunk = one tile row
      if constexpr (sizeof(T) == 2) {
        // lane group g owns pixels x = 4g..4g+3 (lo) and 16+4g..16+4g+3 (hi); lane i of a group addresses pixel (i>>2), quad (i&3)
        const int xl = lg * 4 + (lr >> 2), qb = (lr & 3) * 8;
        u32x4 fa[TMo];
#pragma unroll
        for (int i = 0; i < TMo; ++i) {
          const char* p = dyt + ((row * SC_TW + xl) * COUT + i * 16) * SZ + qb;
          const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p);
          const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p + 16 * DYB));
          const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
          fa[i] = u32x4{l2.x, l2.y, h2.x, h2.y};
        }
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          const int tap = wave + 4 * t;
          if (tap < 9) {
            const int kh = tap / 3, kw = tap - kh * 3;
#pragma unroll
            for (int j = 0; j < TNi; ++j) {
              const char* p = halo + (((row + kh) * SC_HW + xl + kw) * CIN + j * 16) * SZ + qb;
              const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p);
              const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p + 16 * PIXB));
              const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
              const u32x4 fb = u32x4{l2.x, l2.y, h2.x, h2.y};
#pragma unroll
              for (int i = 0; i < TMo; ++i)
                acc[t][i][j] = mfma16_16x16x32(fa[i], fb, acc[t][i][j]);
            }
          }
        }
      } else {
        // fp32: 8 k-steps of 4 pixels; lane group g owns pixel 4s + g
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          const int x = s * 4 + lg;
          float fa[TMo];
#pragma unroll
          for (int i = 0; i < TMo; ++i) fa[i] = *reinterpret_cast<const float*>(dyt + ((row * SC_TW + x) * COUT + i * 16 + lr) * SZ);
#pragma unroll
          for (int t = 0; t < 3; ++t) {
            const int tap = wave + 4 * t;
            if (tap < 9) {
              const int kh = tap / 3, kw = tap - kh * 3;
#pragma unroll
              for (int j = 0; j < TNi; ++j) {
                const float fb = *reinterpret_cast<const float*>(halo + (((row + kh) * SC_HW + x + kw) * CIN + j * 16 + lr) * SZ);
#pragma unroll
                for (int i = 0; i < TMo; ++i) acc[t][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i], fb, acc[t][i][j], 0, 0, 0);
              }
            }
          }
        }
      }
    }
  }

  // ---- slab: C layout row (co) = lg*4 + r, col (ci) = lr ------------------------------------------
  float* out = a.slabs + (size_t)blockIdx.x * a.Cout * (9 * a.ctot);
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const int tap = wave + 4 * t;
    if (tap >= 9) continue;
#pragma unroll
    for (int i = 0; i < TMo; ++i)
#pragma unroll
      for (int j = 0; j < TNi; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = i * 16 + lg * 4 + r;
          if (co < a.Cout) out[(size_t)co * (9 * a.ctot) + tap * a.ctot + a.coff + j * 16 + lr] = acc[t][i][j][r];
        }
  }
}


template <typename T, int CIN, int COUT>
__global__ __launch_bounds__(256) void conv_sc_wgrad_kernel(const ScWgArgs a) {
  constexpr int SZ = (int)sizeof(T);
  constexpr int VEC = Elem<T>::VEC;
  constexpr int TMo = (COUT + 15) / 16, TNi = CIN / 16;
  constexpr int PIXB = CIN * SZ, DYB = COUT * SZ;
  constexpr int HALO_BYTES = SC_HH * SC_HW * PIXB;
  static_assert(CIN % 16 == 0 && COUT % VEC == 0, "channel granularity");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* halo = smem;
  char* dyt = smem + HALO_BYTES;  // [SC_TH][SC_TW][COUT]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  const int sh = a.up ? 1 : 0;

  static_assert(256 % (CIN / VEC) == 0, "a thread stages the same channel vector in every pass");
  const bool pbn = a.pbn.mean != nullptr;
  ScStageBn<T> sbn;
  if (pbn) sbn.load(a.pbn, (tid % (CIN / VEC)) * VEC);

  f32x4 acc[3][TMo][TNi];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int i = 0; i < TMo; ++i)
#pragma unroll
      for (int j = 0; j < TNi; ++j) acc[t][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    int b = tile;
    const int tx = b % a.tiles_x; b /= a.tiles_x;
    const int ty = b % a.tiles_y;
    const int n = b / a.tiles_y;
    const int y0 = ty * SC_TH, x0 = tx * SC_TW;
    __syncthreads();  // previous tile fully consumed
    const char* img = a.src + (size_t)n * a.Hs * a.Ws * PIXB;
    for (int v = tid; v < SC_HH * SC_HW * (CIN / VEC); v += 256) {
      const int pix = v / (CIN / VEC), cv = v - pix * (CIN / VEC);
      const int hy = pix / SC_HW, hx = pix - hy * SC_HW;
      const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
      u32x4 val = {0u, 0u, 0u, 0u};
      if ((unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W) {
        val = *reinterpret_cast<const u32x4*>(img + ((size_t)(gy >> sh) * a.Ws + (gx >> sh)) * PIXB + cv * 16);
        if (pbn) val = sbn.apply(val, a.pbn.relu);
      }
      *reinterpret_cast<u32x4*>(halo + v * 16) = val;
    }
    const char* dimg = a.dy + (size_t)n * a.H * a.W * DYB;
    for (int v = tid; v < SC_TH * SC_TW * (COUT / VEC); v += 256) {
      const int pix = v / (COUT / VEC), cv = v - pix * (COUT / VEC);
      const int py = pix / SC_TW, px = pix - py * SC_TW;
      const int gy = y0 + py, gx = x0 + px;
      u32x4 val = {0u, 0u, 0u, 0u};
      if (gy < a.H && gx < a.W) val = *reinterpret_cast<const u32x4*>(dimg + ((size_t)gy * a.W + gx) * DYB + cv * 16);
      *reinterpret_cast<u32x4*>(dyt + v * 16) = val;
    }
    __syncthreads();

#pragma unroll 1
    for (int row = 0; row < SC_TH; ++row) {  // one 32-pixel chunk = one tile row
      if constexpr (sizeof(T) == 2) {
        // lane group g owns pixels x = 4g..4g+3 (lo) and 16+4g..16+4g+3 (hi); lane i of a group addresses pixel (i>>2), quad (i&3)
        const int xl = lg * 4 + (lr >> 2), qb = (lr & 3) * 8;
        u32x4 fa[TMo];
#pragma unroll
        for (int i = 0; i < TMo; ++i) {
          const char* p = dyt + ((row * SC_TW + xl) * COUT + i * 16) * SZ + qb;
          const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p);
          const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p + 16 * DYB));
          const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
          fa[i] = u32x4{l2.x, l2.y, h2.x, h2.y};
        }
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          const int tap = wave + 4 * t;
          if (tap < 9) {
            const int kh = tap / 3, kw = tap - kh * 3;
#pragma unroll
            for (int j = 0; j < TNi; ++j) {
              const char* p = halo + (((row + kh) * SC_HW + xl + kw) * CIN + j * 16) * SZ + qb;
              const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p);
              const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p + 16 * PIXB));
              const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
              const u32x4 fb = u32x4{l2.x, l2.y, h2.x, h2.y};
#pragma unroll
              for (int i = 0; i < TMo; ++i)
                acc[t][i][j] = mfma16_16x16x32(fa[i], fb, acc[t][i][j]);
            }
          }
        }
      } else {
        // fp32: 8 k-steps of 4 pixels; lane group g owns pixel 4s + g
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          const int x = s * 4 + lg;
          float fa[TMo];
#pragma unroll
          for (int i = 0; i < TMo; ++i) fa[i] = *reinterpret_cast<const float*>(dyt + ((row * SC_TW + x) * COUT + i * 16 + lr) * SZ);
#pragma unroll
          for (int t = 0; t < 3; ++t) {
            const int tap = wave + 4 * t;
            if (tap < 9) {
              const int kh = tap / 3, kw = tap - kh * 3;
#pragma unroll
              for (int j = 0; j < TNi; ++j) {
                const float fb = *reinterpret_cast<const float*>(halo + (((row + kh) * SC_HW + x + kw) * CIN + j * 16 + lr) * SZ);
#pragma unroll
                for (int i = 0; i < TMo; ++i) acc[t][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i], fb, acc[t][i][j], 0, 0, 0);
              }
            }
          }
        }
      }
    }
  }

  // ---- slab: C layout row (co) = lg*4 + r, col (ci) = lr ------------------------------------------
  float* out = a.slabs + (size_t)blockIdx.x * a.Cout * (9 * a.ctot);
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const int tap = wave + 4 * t;
    if (tap >= 9) continue;
#pragma unroll
    for (int i = 0; i < TMo; ++i)
#pragma unroll
      for (int j = 0; j < TNi; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = i * 16 + lg * 4 + r;
          if (co < a.Cout) out[(size_t)co * (9 * a.ctot) + tap * a.ctot + a.coff + j * 16 + lr] = acc[t][i][j][r];
        }
  }
}

#define SC_WG_MAX_BLOCKS 1024

// the stem's weight gradient (7x7 / stride 2, 4 padded input channels -> 64) through conv_stem_wgrad_lean_kernel (conv_sc_lean.hip)
static bool wgrad_stem_shape(const stp_wgrad_params* p) {
  return p->dtype == STP_H16 && p->C0 == 4 && p->C1 == 0 && p->KH == 7 && p->KW == 8 && p->stride == 2 && p->pad == 3 && p->Cout == 64 &&
         p->src0_mode == STP_SRC_DIRECT && p->Hs0 == p->Hv && p->Ws0 == p->Wv && p->Ho == (p->Hv - 1) / 2 + 1 && p->Wo == (p->Wv - 1) / 2 + 1 &&
         !p->src_bn_mean && stem_wg_lean_serves(p->N, p->Hv, p->Wv, p->Ho, p->Wo);
}

extern "C" int stp_wgrad_sc_eligible(const stp_wgrad_params* p) {
  if (!p || !stp_dtype_ok(p->dtype)) return 0;
  if (wgrad_stem_shape(p)) return 1;
  const int vec = p->dtype == STP_H16 ? 8 : 4;
  const bool c0ok = p->C0 == 16 || p->C0 == 32 || (p->C0 == 64 && vec == 8);
  const bool c1ok = p->C1 == 0 || p->C1 == 16 || p->C1 == 32 || (p->C1 == 64 && vec == 8);   // (shape-only: no pointers here)
  return p->KH == 3 && p->KW == 3 && p->stride == 1 && p->pad == 1 && c0ok && c1ok &&
         (p->Cout == 8 || p->Cout == 16 || p->Cout == 32 || (p->Cout == 4 && vec == 4)) && (p->Cout % vec) == 0 &&
         p->Ho == p->Hv && p->Wo == p->Wv &&
         (p->src0_mode == STP_SRC_DIRECT || (p->src0_mode == STP_SRC_NEAREST2X && p->Hv == 2 * p->Hs0 && p->Wv == 2 * p->Ws0));
}

// number of slabs (= workgroups) the small-channel weight gradient writes
extern "C" int stp_wgrad_sc_slabs(const stp_wgrad_params* p) {
  if (wgrad_stem_shape(p)) return stem_wg_lean_blocks(p->N, p->Ho, p->Wo);
  const int64_t tiles = (int64_t)p->N * ceil_div(p->Hv, SC_TH) * ceil_div(p->Wv, SC_TW);
  static const int max_blocks = getenv("STP_SC_WG_BLOCKS") ? atoi(getenv("STP_SC_WG_BLOCKS")) : SC_WG_MAX_BLOCKS;
  // one workgroup per slot the double-buffered staging leaves on a CU (round 4: 1024 workgroups of an 80 KB kernel were two rounds of
  // 512, each with its own prologue, slab and share of the reduce kernel's input)
  const int vec = p->dtype == STP_H16 ? 8 : 4, cin = p->C0 > p->C1 ? p->C0 : p->C1;
  const int64_t npx = ((int64_t)SC_HH * SC_HW * (cin / vec) + 255) / 256, npd = ((int64_t)SC_TH * SC_TW * ((p->Cout + vec - 1) / vec) + 255) / 256;
  const int64_t lds = 2 * (npx + npd) * 4096;
  int per_cu = (int)((160 * 1024) / (lds > 0 ? lds : 1));
  per_cu = per_cu < 1 ? 1 : per_cu > 4 ? 4 : per_cu;
  int64_t blocks = (int64_t)sc_cu_count() * per_cu;
  if (blocks > max_blocks) blocks = max_blocks;
  return (int)(tiles < blocks ? tiles : blocks);
}

template <typename T, int CIN, int COUT>
static int launch_sc_wg(const ScWgArgs& a, int blocks, hipStream_t s) {
  if (sc_stream_on() && a.src_bytes && a.dy_bytes) {
    constexpr int VEC = Elem<T>::VEC;
    constexpr int NPX = (SC_HH * SC_HW * (CIN / VEC) + 255) / 256, NPD = (SC_TH * SC_TW * (COUT / VEC) + 255) / 256;
    const size_t lds = (size_t)2 * (NPX + NPD) * 4096;
    static bool attr_set = false;
    if (lds > 64 * 1024 && !attr_set) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_sc_wgrad_stream_kernel<T, CIN, COUT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return STP_E_LAUNCH;
      attr_set = true;
    }
    hipLaunchKernelGGL((conv_sc_wgrad_stream_kernel<T, CIN, COUT>), dim3(blocks), dim3(256), lds, s, a);
    STP_LAUNCH_CHECK();
    return STP_OK;
  }
  const size_t lds = (size_t)SC_HH * SC_HW * CIN * sizeof(T) + (size_t)SC_TH * SC_TW * COUT * sizeof(T) + 64;
  hipLaunchKernelGGL((conv_sc_wgrad_kernel<T, CIN, COUT>), dim3(blocks), dim3(256), lds, s, a);
  STP_LAUNCH_CHECK();
  return STP_OK;
}

template <typename T>
static int sc_wg_dispatch(const ScWgArgs& a, int cin, int cout, int blocks, hipStream_t s) {
  {
    const int r = sc_wg_lean_launch(a, cin, cout, Elem<T>::DTYPE, blocks, s);      // the lean kernel (conv_sc_lean.hip) where it serves the configuration
    if (r != 1) return r;
  }
  const int key = cin * 64 + cout;
  if constexpr (sizeof(T) == 2) {
    switch (key) {
      case 16 * 64 + 8: return launch_sc_wg<T, 16, 8>(a, blocks, s);
      case 16 * 64 + 16: return launch_sc_wg<T, 16, 16>(a, blocks, s);
      case 16 * 64 + 32: return launch_sc_wg<T, 16, 32>(a, blocks, s);
      case 32 * 64 + 8: return launch_sc_wg<T, 32, 8>(a, blocks, s);
      case 32 * 64 + 16: return launch_sc_wg<T, 32, 16>(a, blocks, s);
      case 32 * 64 + 32: return launch_sc_wg<T, 32, 32>(a, blocks, s);
      case 64 * 64 + 8: return launch_sc_wg<T, 64, 8>(a, blocks, s);
      case 64 * 64 + 16: return launch_sc_wg<T, 64, 16>(a, blocks, s);
      case 64 * 64 + 32: return launch_sc_wg<T, 64, 32>(a, blocks, s);
      default: return STP_E_BADARG;
    }
  } else {
    switch (key) {
      case 16 * 64 + 4: return launch_sc_wg<T, 16, 4>(a, blocks, s);
      case 16 * 64 + 8: return launch_sc_wg<T, 16, 8>(a, blocks, s);
      case 16 * 64 + 16: return launch_sc_wg<T, 16, 16>(a, blocks, s);
      case 16 * 64 + 32: return launch_sc_wg<T, 16, 32>(a, blocks, s);
      case 32 * 64 + 4: return launch_sc_wg<T, 32, 4>(a, blocks, s);
      case 32 * 64 + 8: return launch_sc_wg<T, 32, 8>(a, blocks, s);
      case 32 * 64 + 16: return launch_sc_wg<T, 32, 16>(a, blocks, s);
      case 32 * 64 + 32: return launch_sc_wg<T, 32, 32>(a, blocks, s);
      default: return STP_E_BADARG;
    }
  }
}

// One launch per source of the concatenated input; both write disjoint column ranges of the same slabs.
extern "C" int stp_wgrad_sc_partial(const stp_wgrad_params* p, void* workspace, void* stream) {
  if (!stp_wgrad_sc_eligible(p) || !p->src0 || !p->dy || !workspace || (p->C1 > 0 && !p->src1)) return STP_E_BADARG;
  if (wgrad_stem_shape(p)) return stem_wg_lean_launch(p->src0, p->dy, (float*)workspace, p->N, p->Hv, p->Wv, p->Ho, p->Wo, (hipStream_t)stream);
  ScWgArgs a;
  a.dy = (const char*)p->dy; a.slabs = (float*)workspace;
  a.N = p->N; a.H = p->Hv; a.W = p->Wv; a.Cout = p->Cout;
  a.tiles_x = ceil_div(a.W, SC_TW); a.tiles_y = ceil_div(a.H, SC_TH); a.ntiles = a.N * a.tiles_x * a.tiles_y;
  a.ctot = p->C0 + p->C1;
  a.pbn.x = nullptr; a.pbn.mean = p->src_bn_mean; a.pbn.rstd = p->src_bn_rstd; a.pbn.gamma = p->src_bn_gamma; a.pbn.beta = p->src_bn_beta;
  a.pbn.relu = p->src_bn_relu;
  if (a.pbn.mean && (!a.pbn.rstd || p->C1 > 0)) return STP_E_BADARG;
  const int blocks = stp_wgrad_sc_slabs(p);
  hipStream_t s = (hipStream_t)stream;
  const uint64_t szb = p->dtype == STP_H16 ? 2 : 4, lim = 0x80000000ull;
  const uint64_t dyb = (uint64_t)p->N * p->Hv * p->Wv * p->Cout * szb, b0 = (uint64_t)p->N * p->Hs0 * p->Ws0 * p->C0 * szb;
  const uint64_t b1 = (uint64_t)p->N * p->Hv * p->Wv * p->C1 * szb;
  a.dy_bytes = dyb < lim ? (uint32_t)dyb : 0u;      // (0 = beyond 32-bit LDS-DMA offsets: the register-staged kernel runs)
  a.src_bytes = b0 < lim ? (uint32_t)b0 : 0u;
  a.src = (const char*)p->src0; a.Hs = p->Hs0; a.Ws = p->Ws0; a.up = p->src0_mode == STP_SRC_NEAREST2X; a.coff = 0;
  int rc = p->dtype == STP_H16 ? sc_wg_dispatch<bf16_t>(a, p->C0, p->Cout, blocks, s) : sc_wg_dispatch<float>(a, p->C0, p->Cout, blocks, s);
  if (rc != STP_OK || p->C1 == 0) return rc;
  a.pbn.mean = nullptr;
  a.src = (const char*)p->src1; a.Hs = p->Hv; a.Ws = p->Wv; a.up = 0; a.coff = p->C0;
  a.src_bytes = b1 < lim ? (uint32_t)b1 : 0u;
  return p->dtype == STP_H16 ? sc_wg_dispatch<bf16_t>(a, p->C1, p->Cout, blocks, s) : sc_wg_dispatch<float>(a, p->C1, p->Cout, blocks, s);
}
