// BatchNormalization (training / inference phase), max-pool and upsample-gradient kernels.
// All are HBM-bound streaming kernels: 16-byte vector access, channel constants staged in LDS,
// two-stage fixed-order reductions (per-block partials -> finalize) so results are
// deterministic run to run.
#include "common.h"
#include <algorithm>
#include <cstdlib>

#define BN_MAX_BLOCKS 1024
#ifndef POOL_ROWS
#define POOL_ROWS 4  // image rows per workgroup of the max-pool kernels (one row per workgroup = 8-32 K tiny workgroups: dispatch-bound)
#endif


template <typename T> __device__ __forceinline__ f32x4 ld4(const T* p);
template <> __device__ __forceinline__ f32x4 ld4<float>(const float* p) { return load4(p); }
template <> __device__ __forceinline__ f32x4 ld4<bf16_t>(const bf16_t* p) { return load4(p); }

// V consecutive channels <-> float[V]; V*sizeof(T) is 8 or 16 bytes
template <typename T, int V> __device__ __forceinline__ void ldv(const T* p, float (&o)[V]) {
  if constexpr (V == 4) {
    const f32x4 v = ld4<T>(p);
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
  } else {  // 8 bf16 = one 16-byte load
    const u32x4 r = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[2 * e] = h16lo_to_f32(r[e]); o[2 * e + 1] = h16hi_to_f32(r[e]); }
  }
}
template <typename T, int V> __device__ __forceinline__ void stv(T* p, const float (&o)[V]) {
  if constexpr (V == 4) {
    store4(p, f32x4{o[0], o[1], o[2], o[3]});
  } else {
    u32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = pack_bf16x2(o[2 * e], o[2 * e + 1]);
    *reinterpret_cast<u32x4*>(p) = r;
  }
}

// ------------------------------------------------------------------------------------------
// statistics: partial[block][0][c] = sum x, partial[block][1][c] = sum x^2 over the block's rows
// Each thread owns V consecutive channels (16 bytes for bf16 V=8 / fp32 V=4) and strides over rows with 4
// independent loads in flight; row lanes are then combined through LDS in a fixed order.
template <typename T, int V>
__global__ __launch_bounds__(256) void bn_partial_kernel(const T* __restrict__ x, int64_t rows, int C, float* partial) {
  extern __shared__ float red[];  // [256][2V]
  const int cg = C / V;
  const int colsPerPass = cg < 256 ? cg : 256;
  const int rowLanes = 256 / colsPerPass;
  const int tcol = threadIdx.x % colsPerPass, trow = threadIdx.x / colsPerPass;
  const int64_t rowsPerBlock = (rows + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = (int64_t)blockIdx.x * rowsPerBlock;
  const int64_t r1 = r0 + rowsPerBlock < rows ? r0 + rowsPerBlock : rows;
  for (int c0 = 0; c0 < cg; c0 += colsPerPass) {
    const int c = c0 + tcol;
    float s[V], q[V];
#pragma unroll
    for (int e = 0; e < V; ++e) s[e] = q[e] = 0.f;
    if (c < cg && trow < rowLanes) {
      const T* base = x + (size_t)c * V;
      int64_t r = r0 + trow;
      for (; r + 3 * (int64_t)rowLanes < r1; r += 4 * (int64_t)rowLanes) {
        float v0[V], v1[V], v2[V], v3[V];
        ldv<T, V>(base + r * C, v0);
        ldv<T, V>(base + (r + rowLanes) * C, v1);
        ldv<T, V>(base + (r + 2 * (int64_t)rowLanes) * C, v2);
        ldv<T, V>(base + (r + 3 * (int64_t)rowLanes) * C, v3);
#pragma unroll
        for (int e = 0; e < V; ++e) {
          s[e] += (v0[e] + v1[e]) + (v2[e] + v3[e]);
          q[e] += (v0[e] * v0[e] + v1[e] * v1[e]) + (v2[e] * v2[e] + v3[e] * v3[e]);
        }
      }
      for (; r < r1; r += rowLanes) {
        float v0[V];
        ldv<T, V>(base + r * C, v0);
#pragma unroll
        for (int e = 0; e < V; ++e) { s[e] += v0[e]; q[e] += v0[e] * v0[e]; }
      }
    }
    const int slot = (trow * colsPerPass + tcol) * 2 * V;
#pragma unroll
    for (int e = 0; e < V; ++e) { red[slot + e] = s[e]; red[slot + V + e] = q[e]; }
    __syncthreads();
    if (trow == 0 && c < cg) {
      float a[2 * V];
#pragma unroll
      for (int e = 0; e < 2 * V; ++e) a[e] = 0.f;
      for (int t = 0; t < rowLanes; ++t)
#pragma unroll
        for (int e = 0; e < 2 * V; ++e) a[e] += red[(t * colsPerPass + tcol) * 2 * V + e];
      float* p = partial + (size_t)blockIdx.x * 2 * C;
#pragma unroll
      for (int e = 0; e < V; ++e) { p[c * V + e] = a[e]; p[C + c * V + e] = a[V + e]; }
    }
    __syncthreads();
  }
}

// raw image input: uint8, any channel count (C = 3); one thread per (row-lane, channel)
__global__ __launch_bounds__(256) void bn_partial_u8_kernel(const uint8_t* __restrict__ x, int64_t rows, int C, float* partial) {
  __shared__ float rs[256 * 2];
  const int rowLanes = 256 / C;
  const int tcol = threadIdx.x % C, trow = threadIdx.x / C;
  const int64_t rowsPerBlock = (rows + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = (int64_t)blockIdx.x * rowsPerBlock;
  const int64_t r1 = r0 + rowsPerBlock < rows ? r0 + rowsPerBlock : rows;
  float s = 0.f, q = 0.f;
  if (trow < rowLanes)
    for (int64_t r = r0 + trow; r < r1; r += rowLanes) {
      const float v = (float)x[r * C + tcol];
      s += v;
      q += v * v;
    }
  rs[threadIdx.x * 2] = s;
  rs[threadIdx.x * 2 + 1] = q;
  __syncthreads();
  if (trow == 0 && threadIdx.x < C) {
    float a = 0.f, b = 0.f;
    for (int t = 0; t < rowLanes; ++t) { a += rs[(t * C + tcol) * 2]; b += rs[(t * C + tcol) * 2 + 1]; }
    partial[(size_t)blockIdx.x * 2 * C + tcol] = a;
    partial[(size_t)blockIdx.x * 2 * C + C + tcol] = b;
  }
}

// The same statistics for 1..8 interleaved uint8 channels (the raw image), vectorised: a thread takes 16 rows = C 16-byte loads
// per iteration, so byte k of the group belongs to channel k % C at compile time; sums are exact integers (order-free, hence
// deterministic), converted once per workgroup.  rows per workgroup are a multiple of 16; the ragged end goes byte by byte.
template <int C>
__global__ __launch_bounds__(256) void bn_partial_u8v_kernel(const uint8_t* __restrict__ x, int64_t rows, float* partial) {
  __shared__ unsigned long long red[4][2 * C];
  const int64_t rowsPerBlock = ((rows + gridDim.x - 1) / gridDim.x + 15) & ~(int64_t)15;
  const int64_t r0 = (int64_t)blockIdx.x * rowsPerBlock;
  const int64_t r1 = r0 + rowsPerBlock < rows ? r0 + rowsPerBlock : rows;
  const int64_t rv = r0 + ((r1 > r0 ? r1 - r0 : 0) & ~(int64_t)15);            // end of the whole 16-row groups
  uint32_t sm[C], sq[C];
#pragma unroll
  for (int c = 0; c < C; ++c) sm[c] = sq[c] = 0u;
  for (int64_t r = r0 + (int64_t)threadIdx.x * 16; r < rv; r += 256 * 16) {
    uint4 v[C];
    const uint4* p = reinterpret_cast<const uint4*>(x + r * C);
#pragma unroll
    for (int j = 0; j < C; ++j) v[j] = p[j];
#pragma unroll
    for (int j = 0; j < C; ++j) {
      const uint32_t w[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const uint32_t b = (w[k >> 2] >> (8 * (k & 3))) & 0xffu;
        sm[(j * 16 + k) % C] += b;
        sq[(j * 16 + k) % C] += b * b;
      }
    }
  }
  for (int64_t i = rv * C + threadIdx.x; i < r1 * C; i += 256) {                 // (fewer than 16 rows)
    const uint32_t b = x[i];
    const int ch = (int)(i % C);
#pragma unroll
    for (int c = 0; c < C; ++c)
      if (c == ch) { sm[c] += b; sq[c] += b * b; }
  }
#pragma unroll
  for (int c = 0; c < C; ++c) {
    unsigned long long a = sm[c], b = sq[c];
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][c] = a; red[threadIdx.x >> 6][C + c] = b; }
  }
  __syncthreads();
  if (threadIdx.x < 2 * C)
    partial[(size_t)blockIdx.x * 2 * C + threadIdx.x] =
        (float)(double)(red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// Sums partial[b][which][c] over b for 4 consecutive channels per workgroup: 64 strided lanes per
// channel (independent loads in flight), then a fixed-shape LDS tree -> deterministic.
// Returns the totals (valid on lane 0 of each channel column) for `which` = 0 and 1.
struct Sum2 { double s, q; };
__device__ __forceinline__ Sum2 reduce_partials(const float* partial, int blocks, int C, int c, bool want_q) {
  __shared__ double sh[2][64][4];
  const int cc = threadIdx.x & 3, lane = threadIdx.x >> 2;
  double s = 0.0, q = 0.0;
  if (c < C)
    for (int b = lane; b < blocks; b += 64) {
      s += (double)partial[(size_t)b * 2 * C + c];
      if (want_q) q += (double)partial[(size_t)b * 2 * C + C + c];
    }
  sh[0][lane][cc] = s;
  sh[1][lane][cc] = q;
  __syncthreads();
  for (int w = 32; w > 0; w >>= 1) {
    if (lane < w) {
      sh[0][lane][cc] += sh[0][lane + w][cc];
      sh[1][lane][cc] += sh[1][lane + w][cc];
    }
    __syncthreads();
  }
  Sum2 r;
  r.s = sh[0][0][cc];
  r.q = sh[1][0][cc];
  return r;
}

// launch: grid = ceil(C/4), block = 256
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* partial, int blocks, int C, double inv_rows, double unbias,
                                                          float eps, float momentum, float* mean, float* rstd, float* mm,
                                                          float* mv) {
  const int c = blockIdx.x * 4 + (threadIdx.x & 3);
  const Sum2 t = reduce_partials(partial, blocks, C, c, true);
  if ((threadIdx.x >> 2) != 0 || c >= C) return;
  const double m = t.s * inv_rows;
  double var = t.q * inv_rows - m * m;
  if (var < 0.0) var = 0.0;
  mean[c] = (float)m;
  rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (mm) mm[c] = mm[c] * momentum + (float)m * (1.f - momentum);
  if (mv) mv[c] = mv[c] * momentum + (float)(var * unbias) * (1.f - momentum);
}

extern "C" size_t stp_bn_workspace_bytes(int32_t C) { return (size_t)BN_MAX_BLOCKS * 2 * C * sizeof(float); }

static int bn_blocks(int64_t rows) {
  int64_t b = rows / 64;
  if (b < 1) b = 1;
  if (b > BN_MAX_BLOCKS) b = BN_MAX_BLOCKS;
  return (int)b;
}

extern "C" int stp_bn_stats(const void* x, int32_t xdtype, int64_t rows, int32_t C, float eps, float momentum,
                            float* mean, float* rstd, float* moving_mean, float* moving_var, void* workspace,
                            size_t workspace_bytes, void* stream) {
  if (!x || !mean || !rstd || !workspace || rows <= 0 || C <= 0) return STP_E_BADARG;
  if (workspace_bytes < stp_bn_workspace_bytes(C)) return STP_E_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const int blocks = bn_blocks(rows);
  float* partial = (float*)workspace;
  if (xdtype == STP_U8) {
    if (C > 256) return STP_E_BADARG;
    const bool vec = ((uintptr_t)x & 15) == 0;
#define STP_U8V(CC) hipLaunchKernelGGL(bn_partial_u8v_kernel<CC>, dim3(blocks), dim3(256), 0, s, (const uint8_t*)x, rows, partial)
    switch (vec ? C : 0) {
      case 1: STP_U8V(1); break;
      case 2: STP_U8V(2); break;
      case 3: STP_U8V(3); break;
      case 4: STP_U8V(4); break;
      case 5: STP_U8V(5); break;
      case 6: STP_U8V(6); break;
      case 7: STP_U8V(7); break;
      case 8: STP_U8V(8); break;
      default: hipLaunchKernelGGL(bn_partial_u8_kernel, dim3(blocks), dim3(256), 0, s, (const uint8_t*)x, rows, C, partial);
    }
#undef STP_U8V
  } else {
    if (C & 3) return STP_E_BADARG;
    if (xdtype == STP_H16 && (C & 7) == 0)
      hipLaunchKernelGGL((bn_partial_kernel<bf16_t, 8>), dim3(blocks), dim3(256), 256 * 16 * sizeof(float), s, (const bf16_t*)x, rows, C, partial);
    else if (xdtype == STP_H16)
      hipLaunchKernelGGL((bn_partial_kernel<bf16_t, 4>), dim3(blocks), dim3(256), 256 * 8 * sizeof(float), s, (const bf16_t*)x, rows, C, partial);
    else if (xdtype == STP_F32)
      hipLaunchKernelGGL((bn_partial_kernel<float, 4>), dim3(blocks), dim3(256), 256 * 8 * sizeof(float), s, (const float*)x, rows, C, partial);
    else
      return STP_E_BADARG;
  }
  STP_LAUNCH_CHECK();
  const double unbias = rows > 1 ? (double)rows / (double)(rows - 1) : 1.0;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(ceil_div(C, 4)), dim3(256), 0, s, partial, blocks, C, 1.0 / (double)rows,
                     unbias, eps, momentum, mean, rstd, moving_mean, moving_var);
  STP_LAUNCH_CHECK();
  return STP_OK;
}

// Sum of the [2][tiles] partial columns of one channel by 256 threads (one workgroup per channel), fp64, fixed order: up to four
// columns per thread with all loads issued first, a DPP row reduction on the two halves of the doubles (no LDS round trips: the
// ds_bpermute butterfly this replaces was 24 dependent LDS operations of a 5 us launch), the four row totals of a wave by
// v_readlane, the four waves through LDS (one barrier).  Valid in thread 0.
__device__ __forceinline__ double bnf_row_shr_add(double v, int ctl_id) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  int lo = (int)(u & 0xffffffffull), hi = (int)(u >> 32);
  int mlo, mhi;
  switch (ctl_id) {      // bound_ctrl: lanes shifted in from outside the row read 0.0
    case 1: mlo = __builtin_amdgcn_update_dpp(0, lo, 0x111, 0xf, 0xf, true); mhi = __builtin_amdgcn_update_dpp(0, hi, 0x111, 0xf, 0xf, true); break;
    case 2: mlo = __builtin_amdgcn_update_dpp(0, lo, 0x112, 0xf, 0xf, true); mhi = __builtin_amdgcn_update_dpp(0, hi, 0x112, 0xf, 0xf, true); break;
    case 4: mlo = __builtin_amdgcn_update_dpp(0, lo, 0x114, 0xf, 0xf, true); mhi = __builtin_amdgcn_update_dpp(0, hi, 0x114, 0xf, 0xf, true); break;
    default: mlo = __builtin_amdgcn_update_dpp(0, lo, 0x118, 0xf, 0xf, true); mhi = __builtin_amdgcn_update_dpp(0, hi, 0x118, 0xf, 0xf, true); break;
  }
  return v + __builtin_bit_cast(double, ((unsigned long long)(unsigned)mhi << 32) | (unsigned long long)(unsigned)mlo);
}
__device__ __forceinline__ double bnf_readlane_f64(double v, int lane) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(u & 0xffffffffull), lane), hi = (unsigned)__builtin_amdgcn_readlane((int)(u >> 32), lane);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ void bnf_channel_sums(const float* __restrict__ ps, const float* __restrict__ pq, int tiles, double (*sh)[4], double& S, double& Q) {
  double s = 0.0, q = 0.0;
  if (tiles <= 1024) {
    float a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int t = (int)threadIdx.x + 256 * u;
      a[u] = b[u] = 0.f;
      if (t < tiles) { a[u] = ps[t]; b[u] = pq[t]; }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if ((int)threadIdx.x + 256 * u < tiles) { s += (double)a[u]; q += (double)b[u]; }
  } else {
    for (int t = threadIdx.x; t < tiles; t += 256) { s += (double)ps[t]; q += (double)pq[t]; }
  }
  s = bnf_row_shr_add(s, 1); q = bnf_row_shr_add(q, 1);
  s = bnf_row_shr_add(s, 2); q = bnf_row_shr_add(q, 2);
  s = bnf_row_shr_add(s, 4); q = bnf_row_shr_add(q, 4);
  s = bnf_row_shr_add(s, 8); q = bnf_row_shr_add(q, 8);      // lane 15 of every 16-lane row: the row's total
  const double ws = (bnf_readlane_f64(s, 15) + bnf_readlane_f64(s, 31)) + (bnf_readlane_f64(s, 47) + bnf_readlane_f64(s, 63));
  const double wq = (bnf_readlane_f64(q, 15) + bnf_readlane_f64(q, 31)) + (bnf_readlane_f64(q, 47) + bnf_readlane_f64(q, 63));
  if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = ws; sh[1][threadIdx.x >> 6] = wq; }
  __syncthreads();
  S = (sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3]);
  Q = (sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3]);
}

// Finalize for statistics produced by a convolution epilogue: partial is [2][C][tiles] (tile contiguous).
// One wave per channel: coalesced strided walk + wave reduction, fixed order.
__global__ __launch_bounds__(256) void bn_finalize_tiles_kernel(const float* __restrict__ partial, int tiles, int C, double inv_rows,
                                                                double unbias, float eps, float momentum, float* mean, float* rstd,
                                                                float* mm, float* mv) {
  __shared__ double sh[2][4];
  const int c = blockIdx.x;
  double S, Q;
  bnf_channel_sums(partial + (size_t)c * tiles, partial + ((size_t)C + c) * tiles, tiles, sh, S, Q);
  if (threadIdx.x != 0) return;
  const double m = S * inv_rows;
  double var = Q * inv_rows - m * m;
  if (var < 0.0) var = 0.0;
  mean[c] = (float)m;
  rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (mm) mm[c] = mm[c] * momentum + (float)m * (1.f - momentum);
  if (mv) mv[c] = mv[c] * momentum + (float)(var * unbias) * (1.f - momentum);
}

extern "C" int stp_bn_finalize(const float* partial, int32_t tiles, int64_t rows, int32_t C, float eps, float momentum, float* mean,
                               float* rstd, float* moving_mean, float* moving_var, void* stream) {
  if (!partial || !mean || !rstd || tiles <= 0 || rows <= 0 || C <= 0) return STP_E_BADARG;
  const double unbias = rows > 1 ? (double)rows / (double)(rows - 1) : 1.0;
  hipLaunchKernelGGL(bn_finalize_tiles_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, partial, tiles, C, 1.0 / (double)rows,
                     unbias, eps, momentum, mean, rstd, moving_mean, moving_var);
  STP_LAUNCH_CHECK();
  return STP_OK;
}

// ------------------------------------------------------------------------------------------
// apply: y = relu?(x*scale + shift); scale/shift staged in LDS once per block
template <typename TX, typename TY>
__global__ __launch_bounds__(256) void bn_apply_kernel(const TX* __restrict__ x, TY* __restrict__ y, int64_t rows, int C,
                                                       const float* mean, const float* rstd, const float* gamma,
                                                       const float* beta, const float* mvar, float eps, int relu,
                                                       const long long* slots = nullptr, int nslots = 0, double inv_rows = 0.0,
                                                       double unbias = 0.0, float momentum = 0.f, float* mean_out = nullptr,
                                                       float* rstd_out = nullptr, float* mm = nullptr, float* mv = nullptr) {
  extern __shared__ float ss[];  // scale[C], shift[C]
  for (int c = threadIdx.x; c < C; c += 256) {
    float m, r;
    if (slots) {
      // batch statistics straight from the producing convolution's fixed-point slots (every workgroup derives the same
      // numbers; workgroup 0 publishes them for the backward pass and updates the moving statistics)
      const double md = slot_sum(slots, nslots, c) * inv_rows;
      double var = slot_sum(slots, nslots, C + c) * inv_rows - md * md;
      if (var < 0.0) var = 0.0;
      m = (float)md;
      r = (float)(1.0 / sqrt(var + (double)eps));
      if (blockIdx.x == 0) {
        mean_out[c] = m;
        rstd_out[c] = r;
        if (mm) mm[c] = mm[c] * momentum + m * (1.f - momentum);
        if (mv) mv[c] = mv[c] * momentum + (float)(var * unbias) * (1.f - momentum);
      }
    } else {
      m = mean[c];
      r = rstd ? rstd[c] : rsqrtf(mvar[c] + eps);
    }
    const float sc = gamma ? r * gamma[c] : r;
    ss[c] = sc;
    ss[C + c] = (beta ? beta[c] : 0.f) - m * sc;
  }
  __syncthreads();
  const int cg = C >> 2;
  const int64_t total = rows * cg;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % cg) * 4;
    f32x4 v = ld4<TX>(x + i * 4);
    v.x = bn_affine(v.x, ss[c], ss[C + c]);
    v.y = bn_affine(v.y, ss[c + 1], ss[C + c + 1]);
    v.z = bn_affine(v.z, ss[c + 2], ss[C + c + 2]);
    v.w = bn_affine(v.w, ss[c + 3], ss[C + c + 3]);
    if (relu) { v.x = bn_act(v.x, relu); v.y = bn_act(v.y, relu); v.z = bn_act(v.z, relu); v.w = bn_act(v.w, relu); }
    store4(y + i * 4, v);
  }
}

// bf16 fast path: a thread owns 8 CONSECUTIVE channels (one 16-byte vector per row) for the whole launch - the grid stride is a
// multiple of the C/8 channel groups, so scale / shift live in 16 registers instead of being re-read from LDS per element
// (rocprofv3: the LDS-table kernel spent 27 % of its time in LDS with 64 % bank conflicts) - and walks rows with 4 vectors in flight.
__global__ __launch_bounds__(256) void bn_apply_v8_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int64_t rows, int C,
                                                          const float* mean, const float* rstd, const float* gamma, const float* beta,
                                                          int relu) {
  // the thread fetches the constants of ITS 8 channels itself (eight 16-byte loads that hit L2, issued with the first data vectors):
  // no LDS table, no barrier - on the 4-17 MB tensors a thread handles a few vectors and the table prologue was most of its life
  const int cg = C >> 3;
  const int64_t t0 = (int64_t)blockIdx.x * 256 + threadIdx.x, stride = (int64_t)gridDim.x * 256;   // stride % cg == 0 (host)
  const int c = (int)(t0 % cg) * 8;
  float sc[8], sh[8];
  {
    const f32x4 m0 = *reinterpret_cast<const f32x4*>(mean + c), m1 = *reinterpret_cast<const f32x4*>(mean + c + 4);
    const f32x4 r0 = *reinterpret_cast<const f32x4*>(rstd + c), r1 = *reinterpret_cast<const f32x4*>(rstd + c + 4);
    f32x4 g0 = {1.f, 1.f, 1.f, 1.f}, g1 = g0, b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
    if (gamma) { g0 = *reinterpret_cast<const f32x4*>(gamma + c); g1 = *reinterpret_cast<const f32x4*>(gamma + c + 4); }
    if (beta) { b0 = *reinterpret_cast<const f32x4*>(beta + c); b1 = *reinterpret_cast<const f32x4*>(beta + c + 4); }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float r = e < 4 ? r0[e & 3] : r1[e & 3], k = gamma ? r * (e < 4 ? g0[e & 3] : g1[e & 3]) : r;
      sc[e] = k;
      sh[e] = (e < 4 ? b0[e & 3] : b1[e & 3]) - (e < 4 ? m0[e & 3] : m1[e & 3]) * k;
    }
  }
  const int64_t total = rows * cg;
  auto one = [&](const u32x4 r) {
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float lo = bn_act(bn_affine(h16lo_to_f32(r[e]), sc[2 * e], sh[2 * e]), relu);
      const float hi = bn_act(bn_affine(h16hi_to_f32(r[e]), sc[2 * e + 1], sh[2 * e + 1]), relu);
      o[e] = pack_bf16x2(lo, hi);
    }
    return o;
  };
  int64_t i = t0;
  for (; i + 3 * stride < total; i += 4 * stride) {
    const u32x4 r0 = *reinterpret_cast<const u32x4*>(x + i * 8), r1 = *reinterpret_cast<const u32x4*>(x + (i + stride) * 8);
    const u32x4 r2 = *reinterpret_cast<const u32x4*>(x + (i + 2 * stride) * 8), r3 = *reinterpret_cast<const u32x4*>(x + (i + 3 * stride) * 8);
    *reinterpret_cast<u32x4*>(y + i * 8) = one(r0);
    *reinterpret_cast<u32x4*>(y + (i + stride) * 8) = one(r1);
    *reinterpret_cast<u32x4*>(y + (i + 2 * stride) * 8) = one(r2);
    *reinterpret_cast<u32x4*>(y + (i + 3 * stride) * 8) = one(r3);
  }
  for (; i < total; i += stride) *reinterpret_cast<u32x4*>(y + i * 8) = one(*reinterpret_cast<const u32x4*>(x + i * 8));
}

// a grid whose stride (grid x 256) is a multiple of `groups`, close to (and not above) `want` blocks; 0 if none
static int grid_multiple_of(int want, int groups);
static int grid_fixed_channels(int want, int groups) {   // prefer a grid that keeps every thread on one channel group
  const int g = grid_multiple_of(want, groups);
  return g > 0 ? g : want;
}
static int grid_multiple_of(int want, int groups) {
  // grid * 256 % groups == 0  <=>  grid % (groups / gcd(groups, 256)) == 0
  int a = groups, b = 256;
  while (b) { const int t = a % b; a = b; b = t; }
  const int q = groups / a;
  const int g = want / q * q;
  return g >= 1 ? g : 0;
}

// uint8 [rows][C <= CY] -> TY [rows][CY] (CY = 4, or 8 for images of 4..7 channels); padded channels get pad_value
template <typename TY, int CY>
__global__ __launch_bounds__(256) void bn_apply_u8_kernel(const uint8_t* __restrict__ x, TY* __restrict__ y, int64_t rows,
                                                          int C, const float* mean, const float* rstd, const float* gamma,
                                                          const float* beta, const float* mvar, float eps, int relu,
                                                          float pad_value) {
  float sc[CY], sh[CY];
#pragma unroll
  for (int c = 0; c < CY; ++c) {
    if (c < C) {
      const float r = rstd ? rstd[c] : rsqrtf(mvar[c] + eps);
      sc[c] = gamma ? r * gamma[c] : r;
      sh[c] = (beta ? beta[c] : 0.f) - mean[c] * sc[c];
    } else { sc[c] = 0.f; sh[c] = pad_value; }
  }
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < rows; r += (int64_t)gridDim.x * 256) {
#pragma unroll
    for (int h = 0; h < CY / 4; ++h) {
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = h * 4 + e;
        float t = c < C ? bn_affine((float)x[r * C + c], sc[c], sh[c]) : pad_value;
        if (relu && c < C) t = bn_act(t, relu);
        v[e] = t;
      }
      store4(y + r * CY + h * 4, v);
    }
  }
}

static int grid_for(int64_t items) {
  int64_t b = (items + 255) / 256;
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (int)b;
}
// elementwise BatchNormalization passes with per-thread channel constants: at least `per_thread` vectors per thread once every CU
// has two workgroups, so the constant set-up is amortised (a thread with one vector spends most of its life fetching constants)
static int grid_for_amortised(int64_t items, int per_thread) {
  int64_t b = (items + 255) / 256;
  const int64_t lean = (items + 256 * (int64_t)per_thread - 1) / (256 * (int64_t)per_thread);
  if (b > 512) b = lean > 512 ? lean : 512;
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (int)b;
}

static int bn_apply_dispatch(const void* x, int xdt, void* y, int ydt, int64_t rows, int C, int Cy, const float* mean,
                             const float* rstd, const float* gamma, const float* beta, const float* mvar, float eps,
                             int relu, float pad_value, hipStream_t s) {
  if (!x || !y || !mean || rows <= 0) return STP_E_BADARG;
  if (xdt == STP_U8) {
    if ((Cy != 4 && Cy != 8) || C > Cy) return STP_E_BADARG;
    const int g = grid_for(rows);
#define STP_APPLY_U8(TY_, CY_) \
    hipLaunchKernelGGL((bn_apply_u8_kernel<TY_, CY_>), dim3(g), dim3(256), 0, s, (const uint8_t*)x, (TY_*)y, rows, C, mean, rstd, gamma, beta, mvar, eps, \
                       relu, pad_value)
    if (ydt == STP_H16) { if (Cy == 4) STP_APPLY_U8(bf16_t, 4); else STP_APPLY_U8(bf16_t, 8); }
    else if (ydt == STP_F32) { if (Cy == 4) STP_APPLY_U8(float, 4); else STP_APPLY_U8(float, 8); }
    else return STP_E_BADARG;
#undef STP_APPLY_U8
    STP_LAUNCH_CHECK();
    return STP_OK;
  }
  if ((C & 3) || Cy != C || xdt != ydt) return STP_E_BADARG;
  if (xdt == STP_H16 && (C & 7) == 0 && rstd && !mvar) {
    const int gv = grid_multiple_of(grid_for_amortised(rows * (C >> 3), 8), C >> 3);   // 8 vectors per thread once the CUs are covered
    if (gv > 0) {
      hipLaunchKernelGGL(bn_apply_v8_kernel, dim3(gv), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, rows, C, mean, rstd, gamma, beta, relu);
      STP_LAUNCH_CHECK();
      return STP_OK;
    }
  }
  const size_t lds = 2 * (size_t)C * sizeof(float);
  const int g = grid_for(rows * (C >> 2));
  if (xdt == STP_H16)
    hipLaunchKernelGGL((bn_apply_kernel<bf16_t, bf16_t>), dim3(g), dim3(256), lds, s, (const bf16_t*)x, (bf16_t*)y, rows, C,
                       mean, rstd, gamma, beta, mvar, eps, relu);
  else if (xdt == STP_F32)
    hipLaunchKernelGGL((bn_apply_kernel<float, float>), dim3(g), dim3(256), lds, s, (const float*)x, (float*)y, rows, C, mean,
                       rstd, gamma, beta, mvar, eps, relu);
  else
    return STP_E_BADARG;
  STP_LAUNCH_CHECK();
  return STP_OK;
}

extern "C" int stp_bn_apply(const void* x, int32_t xdtype, void* y, int32_t ydtype, int64_t rows, int32_t C, int32_t Cy,
                            const float* mean, const float* rstd, const float* gamma, const float* beta, int32_t relu,
                            float pad_value, void* stream) {
  if (!rstd) return STP_E_BADARG;
  return bn_apply_dispatch(x, xdtype, y, ydtype, rows, C, Cy, mean, rstd, gamma, beta, nullptr, 0.f, relu, pad_value,
                           (hipStream_t)stream);
}

extern "C" int stp_bn_apply_slots(const void* x, void* y, int32_t dtype, int64_t rows, int32_t C, const int64_t* slots, int32_t nslots,
                                  float eps, float momentum, float* mean, float* rstd, float* moving_mean, float* moving_var,
                                  const float* gamma, const float* beta, int32_t relu, void* stream) {
  if (!stp_dtype_ok(dtype)) return STP_E_BADARG;      // (the other build's 16-bit code, or garbage)
  if (!x || !y || !slots || !mean || !rstd || rows <= 0 || C <= 0 || (C & 3) || nslots < 1 || (nslots & (nslots - 1))) return STP_E_BADARG;
  const size_t lds = 2 * (size_t)C * sizeof(float);
  const int g = grid_for(rows * (C >> 2));
  const double inv_rows = 1.0 / (double)rows, unbias = rows > 1 ? (double)rows / (double)(rows - 1) : 1.0;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == STP_H16)
    hipLaunchKernelGGL((bn_apply_kernel<bf16_t, bf16_t>), dim3(g), dim3(256), lds, s, (const bf16_t*)x, (bf16_t*)y, rows, C, (const float*)nullptr,
                       (const float*)nullptr, gamma, beta, (const float*)nullptr, eps, relu, (const long long*)slots, nslots, inv_rows, unbias,
                       momentum, mean, rstd, moving_mean, moving_var);
  else if (dtype == STP_F32)
    hipLaunchKernelGGL((bn_apply_kernel<float, float>), dim3(g), dim3(256), lds, s, (const float*)x, (float*)y, rows, C, (const float*)nullptr,
                       (const float*)nullptr, gamma, beta, (const float*)nullptr, eps, relu, (const long long*)slots, nslots, inv_rows, unbias,
                       momentum, mean, rstd, moving_mean, moving_var);
  else
    return STP_E_BADARG;
  STP_LAUNCH_CHECK();
  return STP_OK;
}

__global__ __launch_bounds__(256) void zero_kernel(uint4* p, int64_t n16) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) p[i] = uint4{0u, 0u, 0u, 0u};
}
extern "C" int stp_zero_bytes(void* p, int64_t bytes, void* stream) {
  if (!p || bytes <= 0 || (bytes & 15) || ((uintptr_t)p & 15)) return STP_E_BADARG;
  hipLaunchKernelGGL(zero_kernel, dim3(grid_for(bytes >> 4)), dim3(256), 0, (hipStream_t)stream, (uint4*)p, bytes >> 4);
  STP_LAUNCH_CHECK();
  return STP_OK;
}

extern "C" int stp_bn_inference(const void* x, int32_t xdtype, void* y, int32_t ydtype, int64_t rows, int32_t C, int32_t Cy,
                                const float* moving_mean, const float* moving_var, float eps, const float* gamma,
                                const float* beta, int32_t relu, float pad_value, void* stream) {
  if (!moving_var) return STP_E_BADARG;
  return bn_apply_dispatch(x, xdtype, y, ydtype, rows, C, Cy, moving_mean, nullptr, gamma, beta, moving_var, eps, relu,
                           pad_value, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------
// backward.  g = dy * [relu ? (x*scale+shift > 0) : 1];  partial sums of g and g*xhat
template <typename T, int V>
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const T* __restrict__ x, const T* __restrict__ dy, int64_t rows,
                                                             int C, const float* mean, const float* rstd,
                                                             const float* gamma, const float* beta, int relu,
                                                             float* partial) {
  extern __shared__ float red[];
  const int cg = C / V;
  const int colsPerPass = cg < 256 ? cg : 256;
  const int rowLanes = 256 / colsPerPass;
  const int tcol = threadIdx.x % colsPerPass, trow = threadIdx.x / colsPerPass;
  const int64_t rowsPerBlock = (rows + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = (int64_t)blockIdx.x * rowsPerBlock;
  const int64_t r1 = r0 + rowsPerBlock < rows ? r0 + rowsPerBlock : rows;
  for (int c0 = 0; c0 < cg; c0 += colsPerPass) {
    const int c = c0 + tcol;
    float s[V], q[V];
#pragma unroll
    for (int e = 0; e < V; ++e) s[e] = q[e] = 0.f;
    if (c < cg && trow < rowLanes) {
      float mu[V], rs[V], sc[V], sh[V];
#pragma unroll
      for (int e = 0; e < V; ++e) {
        mu[e] = mean[c * V + e];
        rs[e] = rstd[c * V + e];
        sc[e] = gamma ? rs[e] * gamma[c * V + e] : rs[e];
        sh[e] = (beta ? beta[c * V + e] : 0.f) - mu[e] * sc[e];
      }
      const size_t cb = (size_t)c * V;
      int64_t r = r0 + trow;
      for (; r + (int64_t)rowLanes < r1; r += 2 * (int64_t)rowLanes) {
        float x0[V], g0[V], x1[V], g1[V];
        ldv<T, V>(x + r * C + cb, x0);
        ldv<T, V>(dy + r * C + cb, g0);
        ldv<T, V>(x + (r + rowLanes) * C + cb, x1);
        ldv<T, V>(dy + (r + rowLanes) * C + cb, g1);
#pragma unroll
        for (int e = 0; e < V; ++e) {
          if (!bn_act_on(bn_affine(x0[e], sc[e], sh[e]), relu)) g0[e] = 0.f;
          if (!bn_act_on(bn_affine(x1[e], sc[e], sh[e]), relu)) g1[e] = 0.f;
          s[e] += g0[e] + g1[e];
          q[e] += g0[e] * ((x0[e] - mu[e]) * rs[e]) + g1[e] * ((x1[e] - mu[e]) * rs[e]);
        }
      }
      for (; r < r1; r += rowLanes) {
        float x0[V], g0[V];
        ldv<T, V>(x + r * C + cb, x0);
        ldv<T, V>(dy + r * C + cb, g0);
#pragma unroll
        for (int e = 0; e < V; ++e) {
          if (!bn_act_on(bn_affine(x0[e], sc[e], sh[e]), relu)) g0[e] = 0.f;
          s[e] += g0[e];
          q[e] += g0[e] * ((x0[e] - mu[e]) * rs[e]);
        }
      }
    }
    const int slot = (trow * colsPerPass + tcol) * 2 * V;
#pragma unroll
    for (int e = 0; e < V; ++e) { red[slot + e] = s[e]; red[slot + V + e] = q[e]; }
    __syncthreads();
    if (trow == 0 && c < cg) {
      float a[2 * V];
#pragma unroll
      for (int e = 0; e < 2 * V; ++e) a[e] = 0.f;
      for (int t = 0; t < rowLanes; ++t)
#pragma unroll
        for (int e = 0; e < 2 * V; ++e) a[e] += red[(t * colsPerPass + tcol) * 2 * V + e];
      float* p = partial + (size_t)blockIdx.x * 2 * C;
#pragma unroll
      for (int e = 0; e < V; ++e) { p[c * V + e] = a[e]; p[C + c * V + e] = a[V + e]; }
    }
    __syncthreads();
  }
}

// sums[0][c] = dbeta, sums[1][c] = dgamma (raw sums, also written to the grad buffers)
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* partial, int blocks, int C, float* sums,
                                                              float* dgamma, float* dbeta) {
  const int c = blockIdx.x * 4 + (threadIdx.x & 3);
  const Sum2 t = reduce_partials(partial, blocks, C, c, true);
  if ((threadIdx.x >> 2) != 0 || c >= C) return;
  sums[c] = (float)t.s;
  sums[C + c] = (float)t.q;
  if (dbeta) dbeta[c] = (float)t.s;
  if (dgamma) dgamma[c] = (float)t.q;
}

template <typename T, int V>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* dx,
                                                           int64_t rows, int C, const float* mean, const float* rstd,
                                                           const float* gamma, const float* beta, const float* sums,
                                                           float inv_rows, int relu, int accumulate, const long long* slots = nullptr,
                                                           int nslots = 0, float* dgamma = nullptr, float* dbeta = nullptr, const T* dadd = nullptr) {
  // accumulate: dx = result + dadd (dadd == nullptr: in place, dadd = dx; otherwise the addend lives in ANOTHER buffer and stays intact -
  // a residual gradient that a pending grouped weight gradient still reads as its dY)
  if (!dadd) dadd = dx;
  extern __shared__ float ss[];  // mean, rstd, scale, shift, k1 = dbeta/M, k2 = dgamma/M   [6][C]
  if (!slots && V == 8 && ((int64_t)gridDim.x * 256) % (C / V) == 0) {
    // The thread keeps its 8 channels for the whole launch and fetches their constants ITSELF (twelve 16-byte loads that hit L2, one
    // latency, issued together with the first data vectors): no LDS table, no barrier.  The table prologue - C x 6 values through
    // LDS per workgroup, then 48 conflicting LDS reads per thread - was most of the launch on the 4-17 MB tensors of stages 2-4,
    // where a thread handles one or two vectors (rocprofv3: 22 us for an 8 MB tensor; launch_table: 0.7-1.1 TB/s).
    const int cg8 = C / 8;
    const int c = (int)(((int64_t)blockIdx.x * 256 + threadIdx.x) % cg8) * 8;
    float mu[8], rs[8], sc[8], sh[8], k1[8], k2[8];
    {
      const f32x4 m0 = *reinterpret_cast<const f32x4*>(mean + c), m1 = *reinterpret_cast<const f32x4*>(mean + c + 4);
      const f32x4 r0 = *reinterpret_cast<const f32x4*>(rstd + c), r1 = *reinterpret_cast<const f32x4*>(rstd + c + 4);
      const f32x4 s0 = *reinterpret_cast<const f32x4*>(sums + c), s1 = *reinterpret_cast<const f32x4*>(sums + c + 4);
      const f32x4 q0 = *reinterpret_cast<const f32x4*>(sums + C + c), q1 = *reinterpret_cast<const f32x4*>(sums + C + c + 4);
      f32x4 g0 = {1.f, 1.f, 1.f, 1.f}, g1 = g0, b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
      if (gamma) { g0 = *reinterpret_cast<const f32x4*>(gamma + c); g1 = *reinterpret_cast<const f32x4*>(gamma + c + 4); }
      if (beta) { b0 = *reinterpret_cast<const f32x4*>(beta + c); b1 = *reinterpret_cast<const f32x4*>(beta + c + 4); }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float m = e < 4 ? m0[e & 3] : m1[e & 3], r = e < 4 ? r0[e & 3] : r1[e & 3];
        const float k = gamma ? r * (e < 4 ? g0[e & 3] : g1[e & 3]) : r;
        mu[e] = m; rs[e] = r; sc[e] = k;
        sh[e] = (e < 4 ? b0[e & 3] : b1[e & 3]) - m * k;
        k1[e] = (e < 4 ? s0[e & 3] : s1[e & 3]) * inv_rows;
        k2[e] = (e < 4 ? q0[e & 3] : q1[e & 3]) * inv_rows;
      }
    }
    const int64_t total8 = rows * cg8, stride8 = (int64_t)gridDim.x * 256;
    auto one8 = [&](int64_t i, const float (&xv)[V], float (&g)[V], const float (&d)[V]) {
      float o[V];
#pragma unroll
      for (int e = 0; e < V; ++e) {
        if (!bn_act_on(bn_affine(xv[e], sc[e], sh[e]), relu)) g[e] = 0.f;
        const float xh = (xv[e] - mu[e]) * rs[e];
        o[e] = sc[e] * (g[e] - k1[e] - xh * k2[e]);
        if (accumulate) o[e] += d[e];
      }
      stv<T, V>(dx + i * V, o);
    };
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + stride8 < total8; i += 2 * stride8) {          // two rows in flight per thread
      float x0[V], g0[V], d0[V], x1[V], g1[V], d1[V];
      ldv<T, V>(x + i * V, x0);
      ldv<T, V>(x + (i + stride8) * V, x1);
      ldv<T, V>(dy + i * V, g0);
      ldv<T, V>(dy + (i + stride8) * V, g1);
      if (accumulate) { ldv<T, V>(dadd + i * V, d0); ldv<T, V>(dadd + (i + stride8) * V, d1); }
      one8(i, x0, g0, d0);
      one8(i + stride8, x1, g1, d1);
    }
    for (; i < total8; i += stride8) {
      float x0[V], g0[V], d0[V];
      ldv<T, V>(x + i * V, x0);
      ldv<T, V>(dy + i * V, g0);
      if (accumulate) ldv<T, V>(dadd + i * V, d0);
      one8(i, x0, g0, d0);
    }
    return;
  }
  for (int c = threadIdx.x; c < C; c += 256) {
    const float r = rstd[c], sc = gamma ? r * gamma[c] : r;
    ss[c] = mean[c];
    ss[C + c] = r;
    ss[2 * C + c] = sc;
    ss[3 * C + c] = (beta ? beta[c] : 0.f) - mean[c] * sc;
    float sb, sg;
    if (slots) {    // sums from the data-gradient convolution's fixed-point slots; workgroup 0 publishes the parameter gradients
      sb = (float)slot_sum(slots, nslots, c);
      sg = (float)slot_sum(slots, nslots, C + c);
      if (blockIdx.x == 0) {
        if (dbeta) dbeta[c] = sb;
        if (dgamma) dgamma[c] = sg;
      }
    } else {
      sb = sums[c];
      sg = sums[C + c];
    }
    ss[4 * C + c] = sb * inv_rows;
    ss[5 * C + c] = sg * inv_rows;
  }
  __syncthreads();
  const int cg = C / V;
  const int64_t total = rows * cg;
  const int64_t stride = (int64_t)gridDim.x * 256;
  if (stride % cg == 0) {
    // the thread keeps its V channels for the whole launch: the six per-channel constants live in registers instead of being
    // re-read from LDS per element (rocprofv3: 44 % LDS time, 62 % bank conflicts in the table-driven loop below)
    const int c = (int)(((int64_t)blockIdx.x * 256 + threadIdx.x) % cg) * V;
    float mu[V], rs[V], sc[V], sh[V], k1[V], k2[V];
#pragma unroll
    for (int e = 0; e < V; ++e) {
      mu[e] = ss[c + e]; rs[e] = ss[C + c + e]; sc[e] = ss[2 * C + c + e]; sh[e] = ss[3 * C + c + e];
      k1[e] = ss[4 * C + c + e]; k2[e] = ss[5 * C + c + e];
    }
    auto one = [&](int64_t i, const float (&xv)[V], float (&g)[V], const float (&d)[V]) {
      float o[V];
#pragma unroll
      for (int e = 0; e < V; ++e) {
        if (!bn_act_on(bn_affine(xv[e], sc[e], sh[e]), relu)) g[e] = 0.f;
        const float xh = (xv[e] - mu[e]) * rs[e];
        o[e] = sc[e] * (g[e] - k1[e] - xh * k2[e]);
        if (accumulate) o[e] += d[e];
      }
      stv<T, V>(dx + i * V, o);
    };
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + stride < total; i += 2 * stride) {          // two rows in flight per thread
      float x0[V], g0[V], d0[V], x1[V], g1[V], d1[V];
      ldv<T, V>(x + i * V, x0);
      ldv<T, V>(x + (i + stride) * V, x1);
      ldv<T, V>(dy + i * V, g0);
      ldv<T, V>(dy + (i + stride) * V, g1);
      if (accumulate) { ldv<T, V>(dadd + i * V, d0); ldv<T, V>(dadd + (i + stride) * V, d1); }
      one(i, x0, g0, d0);
      one(i + stride, x1, g1, d1);
    }
    for (; i < total; i += stride) {
      float x0[V], g0[V], d0[V];
      ldv<T, V>(x + i * V, x0);
      ldv<T, V>(dy + i * V, g0);
      if (accumulate) ldv<T, V>(dadd + i * V, d0);
      one(i, x0, g0, d0);
    }
    return;
  }
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    const int c = (int)(i % cg) * V;
    float xv[V], g[V], o[V];
    ldv<T, V>(x + i * V, xv);
    ldv<T, V>(dy + i * V, g);
#pragma unroll
    for (int e = 0; e < V; ++e) {
      if (!bn_act_on(bn_affine(xv[e], ss[2 * C + c + e], ss[3 * C + c + e]), relu)) g[e] = 0.f;
      const float xh = (xv[e] - ss[c + e]) * ss[C + c + e];
      o[e] = ss[2 * C + c + e] * (g[e] - ss[4 * C + c + e] - xh * ss[5 * C + c + e]);
    }
    if (accumulate) {
      float d[V];
      ldv<T, V>(dadd + i * V, d);
#pragma unroll
      for (int e = 0; e < V; ++e) o[e] += d[e];
    }
    stv<T, V>(dx + i * V, o);
  }
}

extern "C" int stp_bn_backward(const void* x, const void* dy, void* dx, int32_t dtype, int64_t rows, int32_t C,
                               const float* mean, const float* rstd, const float* gamma, const float* beta, float* dgamma,
                               float* dbeta, int32_t relu, int32_t accumulate_dx, void* workspace, size_t workspace_bytes,
                               void* stream) {
  if (!stp_dtype_ok(dtype)) return STP_E_BADARG;      // (the other build's 16-bit code, or garbage)
  if (!x || !dy || !dx || !mean || !rstd || !workspace || rows <= 0 || C <= 0 || (C & 3)) return STP_E_BADARG;
  if (workspace_bytes < stp_bn_workspace_bytes(C)) return STP_E_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const int blocks = bn_blocks(rows) > BN_MAX_BLOCKS - 1 ? BN_MAX_BLOCKS - 1 : bn_blocks(rows);
  float* partial = (float*)workspace;
  float* sums = partial + (size_t)(BN_MAX_BLOCKS - 1) * 2 * C;  // last slab holds the finalized sums
  const bool v8 = dtype == STP_H16 && (C & 7) == 0;
  if (v8)
    hipLaunchKernelGGL((bn_bwd_partial_kernel<bf16_t, 8>), dim3(blocks), dim3(256), 256 * 16 * sizeof(float), s, (const bf16_t*)x,
                       (const bf16_t*)dy, rows, C, mean, rstd, gamma, beta, relu, partial);
  else if (dtype == STP_H16)
    hipLaunchKernelGGL((bn_bwd_partial_kernel<bf16_t, 4>), dim3(blocks), dim3(256), 256 * 8 * sizeof(float), s, (const bf16_t*)x,
                       (const bf16_t*)dy, rows, C, mean, rstd, gamma, beta, relu, partial);
  else if (dtype == STP_F32)
    hipLaunchKernelGGL((bn_bwd_partial_kernel<float, 4>), dim3(blocks), dim3(256), 256 * 8 * sizeof(float), s, (const float*)x,
                       (const float*)dy, rows, C, mean, rstd, gamma, beta, relu, partial);
  else
    return STP_E_BADARG;
  STP_LAUNCH_CHECK();
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(ceil_div(C, 4)), dim3(256), 0, s, partial, blocks, C, sums, dgamma, dbeta);
  STP_LAUNCH_CHECK();
  const size_t lds2 = 6 * (size_t)C * sizeof(float);
  const int g = grid_fixed_channels(grid_for_amortised(rows * (C / (v8 ? 8 : 4)), 8), C / (v8 ? 8 : 4));
  const float inv_rows = (float)(1.0 / (double)rows);
  if (v8)
    hipLaunchKernelGGL((bn_bwd_apply_kernel<bf16_t, 8>), dim3(g), dim3(256), lds2, s, (const bf16_t*)x, (const bf16_t*)dy,
                       (bf16_t*)dx, rows, C, mean, rstd, gamma, beta, sums, inv_rows, relu, accumulate_dx);
  else if (dtype == STP_H16)
    hipLaunchKernelGGL((bn_bwd_apply_kernel<bf16_t, 4>), dim3(g), dim3(256), lds2, s, (const bf16_t*)x, (const bf16_t*)dy,
                       (bf16_t*)dx, rows, C, mean, rstd, gamma, beta, sums, inv_rows, relu, accumulate_dx);
  else
    hipLaunchKernelGGL((bn_bwd_apply_kernel<float, 4>), dim3(g), dim3(256), lds2, s, (const float*)x, (const float*)dy, (float*)dx,
                       rows, C, mean, rstd, gamma, beta, sums, inv_rows, relu, accumulate_dx);
  STP_LAUNCH_CHECK();
  return STP_OK;
}

// ------------------------------------------------------------------------------------------
// FUSED finalize + apply (round 3).  The finalize launches (one workgroup per channel, ~5 us each: 86 per U-Net/ResNet34 step, pure
// launch latency on the critical chain conv -> finalize -> apply -> conv) disappear where the partial table is small: a workgroup
// of the apply pass owns a SLAB of 64 channels (128 bytes of every row: whole cache lines) x a chunk of rows, and first reduces the
// [2][64][tiles] partial sums of its own slab - fp64, fixed order, the same for every workgroup of the slab, so all of them
// normalise with bit-identical constants; the chunk-0 workgroup of a slab also publishes mean / rstd / moving statistics
// (dgamma / dbeta in the backward form).  Each workgroup re-reads 512 x tiles bytes that sit in L2 - eligible while that is <= 64 KB.
#define BNF_SLAB 64
static bool bn_fused_ok(int dtype, int64_t rows, int C, int tiles) {
  static const bool on = !(getenv("STP_BN_FUSE_FINALIZE") && atoi(getenv("STP_BN_FUSE_FINALIZE")) == 0);
  // measured per shape (U-Net/ResNet34 bs16, eager launches, profiles/r03j_bn_fused_vs_separate.txt): 32 columns (stage 4) 8.1 vs 6.7 + 7.0 us
  // forward and 7.7 vs 11.4 backward, 128 columns (stage 3) 10.4 vs 13.7 and 10.9 vs 12.5, 256 columns (stage 2: 128 KB of partial sums
  // per workgroup, twice its own rows) 14.7 vs 16.0 and 17.8 vs 15.8 - the fused form stops at 128 columns
  // (STP_BN_FUSE_MAXCOLS: A/B switch for the column limit; round 4 re-measured 256 columns with the rows prefetched before the sums)
  static const int maxcols = getenv("STP_BN_FUSE_MAXCOLS") ? atoi(getenv("STP_BN_FUSE_MAXCOLS")) : 128;
  return on && dtype == STP_H16 && (C % BNF_SLAB) == 0 && tiles >= 1 && tiles <= maxcols && rows >= 64;
}
static int bn_fused_chunks(int64_t rows, int C) {
  // ~8 rows per thread once the CUs are covered (a thread owns 8 channels of a row: 32 rows per workgroup pass)
  const int slabs = C / BNF_SLAB;
  int64_t want = (rows + 255) / 256;              // 8 passes of 32 rows
  const int64_t lo = (2 * 256 + slabs - 1) / slabs;
  if (want < lo) want = lo;
  const int64_t hi = (rows + 31) / 32;
  if (want > hi) want = hi;
  return (int)(want < 1 ? 1 : want);
}

// sum of partial[stat][c][0..tiles) for the 64 channels of the slab: 4 lanes per channel, fp64, fixed order; result in lanes j == 0
__device__ __forceinline__ void bnf_slab_sums(const float* __restrict__ partial, int tiles, int C, int c0, double& s, double& q) {
  const int cl = threadIdx.x >> 2, j = threadIdx.x & 3;
  const float* ps = partial + (size_t)(c0 + cl) * tiles;
  const float* pq = partial + ((size_t)C + c0 + cl) * tiles;
  s = 0.0; q = 0.0;
  if ((tiles & 3) == 0 && tiles > 128 && tiles <= 256) {
    // two batches of the 128-column form below (same order of additions as the generic loop: t ascending per lane)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      f32x4 a[8], b[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int t = 4 * j + 16 * (u + 8 * h);
        a[u] = b[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (t < tiles) { a[u] = *reinterpret_cast<const f32x4*>(ps + t); b[u] = *reinterpret_cast<const f32x4*>(pq + t); }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (4 * j + 16 * (u + 8 * h) < tiles) {
          s += ((double)a[u].x + (double)a[u].y) + ((double)a[u].z + (double)a[u].w);
          q += ((double)b[u].x + (double)b[u].y) + ((double)b[u].z + (double)b[u].w);
        }
    }
  } else if ((tiles & 3) == 0 && tiles <= 128) {
    // (the fused kernels' case.)  All loads first - as a loop with a run-time trip count every one of the up to 8 iterations exposed
    // a memory round trip at the start of EVERY workgroup of the apply pass; then the sums in the same order
    f32x4 a[8], b[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int t = 4 * j + 16 * u;
      a[u] = b[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (t < tiles) { a[u] = *reinterpret_cast<const f32x4*>(ps + t); b[u] = *reinterpret_cast<const f32x4*>(pq + t); }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (4 * j + 16 * u < tiles) {
        s += ((double)a[u].x + (double)a[u].y) + ((double)a[u].z + (double)a[u].w);
        q += ((double)b[u].x + (double)b[u].y) + ((double)b[u].z + (double)b[u].w);
      }
  } else if ((tiles & 3) == 0) {
    for (int t = 4 * j; t < tiles; t += 16) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(ps + t), b = *reinterpret_cast<const f32x4*>(pq + t);
      s += ((double)a.x + (double)a.y) + ((double)a.z + (double)a.w);
      q += ((double)b.x + (double)b.y) + ((double)b.z + (double)b.w);
    }
  } else {
    for (int t = j; t < tiles; t += 4) { s += (double)ps[t]; q += (double)pq[t]; }
  }
  s += __shfl_xor(s, 1, 64); q += __shfl_xor(q, 1, 64);
  s += __shfl_xor(s, 2, 64); q += __shfl_xor(q, 2, 64);
}

__global__ __launch_bounds__(256) void bn_finalize_apply_kernel(const float* __restrict__ partial, int tiles, const bf16_t* __restrict__ x,
                                                                bf16_t* __restrict__ y, int64_t rows, int C, int chunks, double inv_rows,
                                                                double unbias, float eps, float momentum, float* mean, float* rstd,
                                                                float* mm, float* mv, const float* gamma, const float* beta, int relu) {
  __shared__ float tab[2][BNF_SLAB];
  const int slab = blockIdx.x / chunks, chunk = blockIdx.x - slab * chunks, c0 = slab * BNF_SLAB;
  const int cg = threadIdx.x & 7, r0 = threadIdx.x >> 3;
  const int64_t per = (rows + chunks - 1) / chunks, ra = (int64_t)chunk * per, rb = ra + per < rows ? ra + per : rows;
  const size_t col = (size_t)c0 + cg * 8;
  // The first four rows of this thread are requested BEFORE the partial sums are reduced (round 4): they do not depend on the
  // statistics, and with ~8 rows per thread the launch is two dependent memory round trips long (sums, then rows) - this makes it one.
  int64_t r = ra + r0;
  const bool pre = r + 96 < rb;
  u32x4 p0 = {0u, 0u, 0u, 0u}, p1 = p0, p2 = p0, p3 = p0;
  if (pre) {
    p0 = *reinterpret_cast<const u32x4*>(x + (size_t)r * C + col); p1 = *reinterpret_cast<const u32x4*>(x + (size_t)(r + 32) * C + col);
    p2 = *reinterpret_cast<const u32x4*>(x + (size_t)(r + 64) * C + col); p3 = *reinterpret_cast<const u32x4*>(x + (size_t)(r + 96) * C + col);
  }
  {
    const int cc = c0 + (threadIdx.x >> 2);
    const float gam = gamma ? gamma[cc] : 1.f, bet = beta ? beta[cc] : 0.f;     // (requested with the rows: independent of the sums)
    double s, q;
    bnf_slab_sums(partial, tiles, C, c0, s, q);
    if ((threadIdx.x & 3) == 0) {
      const int c = cc;
      const double m = s * inv_rows;
      double var = q * inv_rows - m * m;
      if (var < 0.0) var = 0.0;
      const float mf = (float)m, rf = (float)(1.0 / sqrt(var + (double)eps));
      const float k = gamma ? rf * gam : rf;
      tab[0][threadIdx.x >> 2] = k;
      tab[1][threadIdx.x >> 2] = bet - mf * k;
      if (chunk == 0) {
        mean[c] = mf;
        rstd[c] = rf;
        if (mm) mm[c] = mm[c] * momentum + mf * (1.f - momentum);
        if (mv) mv[c] = mv[c] * momentum + (float)(var * unbias) * (1.f - momentum);
      }
    }
  }
  __syncthreads();
  float sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { sc[e] = tab[0][cg * 8 + e]; sh[e] = tab[1][cg * 8 + e]; }
  auto one = [&](const u32x4 rr) {
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float lo = bn_act(bn_affine(h16lo_to_f32(rr[e]), sc[2 * e], sh[2 * e]), relu);
      const float hi = bn_act(bn_affine(h16hi_to_f32(rr[e]), sc[2 * e + 1], sh[2 * e + 1]), relu);
      o[e] = pack_bf16x2(lo, hi);
    }
    return o;
  };
  if (pre) {
    *reinterpret_cast<u32x4*>(y + (size_t)r * C + col) = one(p0);
    *reinterpret_cast<u32x4*>(y + (size_t)(r + 32) * C + col) = one(p1);
    *reinterpret_cast<u32x4*>(y + (size_t)(r + 64) * C + col) = one(p2);
    *reinterpret_cast<u32x4*>(y + (size_t)(r + 96) * C + col) = one(p3);
    r += 128;
  }
  for (; r + 96 < rb; r += 128) {       // four rows in flight per thread
    const u32x4 v0 = *reinterpret_cast<const u32x4*>(x + (size_t)r * C + col), v1 = *reinterpret_cast<const u32x4*>(x + (size_t)(r + 32) * C + col);
    const u32x4 v2 = *reinterpret_cast<const u32x4*>(x + (size_t)(r + 64) * C + col), v3 = *reinterpret_cast<const u32x4*>(x + (size_t)(r + 96) * C + col);
    *reinterpret_cast<u32x4*>(y + (size_t)r * C + col) = one(v0);
    *reinterpret_cast<u32x4*>(y + (size_t)(r + 32) * C + col) = one(v1);
    *reinterpret_cast<u32x4*>(y + (size_t)(r + 64) * C + col) = one(v2);
    *reinterpret_cast<u32x4*>(y + (size_t)(r + 96) * C + col) = one(v3);
  }
  for (; r < rb; r += 32) *reinterpret_cast<u32x4*>(y + (size_t)r * C + col) = one(*reinterpret_cast<const u32x4*>(x + (size_t)r * C + col));
}

extern "C" int stp_bn_finalize_apply_ok(int32_t dtype, int64_t rows, int32_t C, int32_t tiles) { return bn_fused_ok(dtype, rows, C, tiles) ? 1 : 0; }

extern "C" int stp_bn_finalize_apply(const float* partial, int32_t tiles, const void* x, void* y, int32_t dtype, int64_t rows, int32_t C,
                                     float eps, float momentum, float* mean, float* rstd, float* moving_mean, float* moving_var,
                                     const float* gamma, const float* beta, int32_t relu, void* stream) {
  if (!stp_dtype_ok(dtype)) return STP_E_BADARG;
  if (!partial || !x || !y || !mean || !rstd || rows <= 0 || C <= 0 || !bn_fused_ok(dtype, rows, C, tiles)) return STP_E_BADARG;
  const int chunks = bn_fused_chunks(rows, C);
  const double unbias = rows > 1 ? (double)rows / (double)(rows - 1) : 1.0;
  hipLaunchKernelGGL(bn_finalize_apply_kernel, dim3((C / BNF_SLAB) * chunks), dim3(256), 0, (hipStream_t)stream, partial, tiles, (const bf16_t*)x,
                     (bf16_t*)y, rows, C, chunks, 1.0 / (double)rows, unbias, eps, momentum, mean, rstd, moving_mean, moving_var, gamma, beta, relu);
  STP_LAUNCH_CHECK();
  return STP_OK;
}

// backward form: dx = scale * (g - sum(g)/M - xhat * sum(g xhat)/M) (+ dadd); g already carries the activation mask
__global__ __launch_bounds__(256) void bn_bwd_finalize_apply_kernel(const float* __restrict__ partial, int tiles, const bf16_t* __restrict__ x,
                                                                    const bf16_t* __restrict__ g, bf16_t* dx, const bf16_t* dadd, int64_t rows, int C,
                                                                    int chunks, float inv_rows, const float* __restrict__ mean,
                                                                    const float* __restrict__ rstd, const float* __restrict__ gamma, float* dgamma,
                                                                    float* dbeta) {
  __shared__ float tab[5][BNF_SLAB];      // sum g / M, sum g xhat / M, mean, rstd, rstd * gamma
  const int slab = blockIdx.x / chunks, chunk = blockIdx.x - slab * chunks, c0 = slab * BNF_SLAB;
  const int cg = threadIdx.x & 7, r0 = threadIdx.x >> 3;
  const int64_t per = (rows + chunks - 1) / chunks, ra = (int64_t)chunk * per, rb = ra + per < rows ? ra + per : rows;
  const size_t col = (size_t)c0 + cg * 8;
  // (the first FOUR rows of x / g / dadd are requested before the partial sums are reduced: see bn_finalize_apply_kernel; a thread
  //  owns ~8 rows, so the launch is two batches of loads deep instead of four)
  int64_t r = ra + r0;
  const bool pre = r + 96 < rb;
  const u32x4 z4 = {0u, 0u, 0u, 0u};
  u32x4 px[4] = {z4, z4, z4, z4}, pg[4] = {z4, z4, z4, z4}, pd[4] = {z4, z4, z4, z4};
  if (pre) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const size_t o = (size_t)(r + 32 * k) * C + col;
      px[k] = *reinterpret_cast<const u32x4*>(x + o);
      pg[k] = *reinterpret_cast<const u32x4*>(g + o);
      if (dadd) pd[k] = *reinterpret_cast<const u32x4*>(dadd + o);
    }
  }
  {
    const int cl = threadIdx.x >> 2, c = c0 + cl;
    // per-channel constants requested with the rows (independent of the sums); scale = rstd * gamma goes through the table as well -
    // read from global memory after the barrier, the 8 gamma values of a thread were 8 dependent round trips in every workgroup
    const float mc = mean[c], rc = rstd[c], gc = gamma ? gamma[c] : 1.f;
    double s, q;
    bnf_slab_sums(partial, tiles, C, c0, s, q);
    if ((threadIdx.x & 3) == 0) {
      const float sf = (float)s, qf = (float)q;
      tab[0][cl] = sf * inv_rows;
      tab[1][cl] = qf * inv_rows;
      tab[2][cl] = mc;
      tab[3][cl] = rc;
      tab[4][cl] = gamma ? rc * gc : rc;
      if (chunk == 0) {
        if (dbeta) dbeta[c] = sf;
        if (dgamma) dgamma[c] = qf;
      }
    }
  }
  __syncthreads();
  float k1[8], k2[8], mu[8], rs[8], sc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    k1[e] = tab[0][cg * 8 + e]; k2[e] = tab[1][cg * 8 + e]; mu[e] = tab[2][cg * 8 + e]; rs[e] = tab[3][cg * 8 + e];
    sc[e] = tab[4][cg * 8 + e];
  }
  // same arithmetic per element as before (fp32, same order): dx = scale * (g - k1 - xhat * k2) (+ dadd)
  auto one = [&](size_t off, const u32x4 xr, const u32x4 gr, const u32x4 dr) {
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float xl = h16lo_to_f32(xr[e]), xh_ = h16hi_to_f32(xr[e]);
      const float gl = h16lo_to_f32(gr[e]), gh = h16hi_to_f32(gr[e]);
      const float xhl = (xl - mu[2 * e]) * rs[2 * e], xhh = (xh_ - mu[2 * e + 1]) * rs[2 * e + 1];
      float ol = sc[2 * e] * (gl - k1[2 * e] - xhl * k2[2 * e]), oh = sc[2 * e + 1] * (gh - k1[2 * e + 1] - xhh * k2[2 * e + 1]);
      if (dadd) { ol += h16lo_to_f32(dr[e]); oh += h16hi_to_f32(dr[e]); }
      o[e] = pack_bf16x2(ol, oh);
    }
    *reinterpret_cast<u32x4*>(dx + off) = o;
  };
  if (pre) {
#pragma unroll
    for (int k = 0; k < 4; ++k) one((size_t)(r + 32 * k) * C + col, px[k], pg[k], pd[k]);
    r += 128;
  }
  for (; r + 96 < rb; r += 128) {          // four rows in flight per thread
    u32x4 vx[4], vg[4], vd[4] = {z4, z4, z4, z4};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const size_t o = (size_t)(r + 32 * k) * C + col;
      vx[k] = *reinterpret_cast<const u32x4*>(x + o);
      vg[k] = *reinterpret_cast<const u32x4*>(g + o);
      if (dadd) vd[k] = *reinterpret_cast<const u32x4*>(dadd + o);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) one((size_t)(r + 32 * k) * C + col, vx[k], vg[k], vd[k]);
  }
  for (; r < rb; r += 32) {
    const size_t o = (size_t)r * C + col;
    const u32x4 vx = *reinterpret_cast<const u32x4*>(x + o), vg = *reinterpret_cast<const u32x4*>(g + o);
    const u32x4 vd = dadd ? *reinterpret_cast<const u32x4*>(dadd + o) : z4;
    one(o, vx, vg, vd);
  }
}

static int bn_bwd_fused_launch(const void* x, const void* g, void* dx, const void* dadd, int64_t rows, int C, const float* mean, const float* rstd,
                               const float* gamma, const float* partial, int tiles, float* dgamma, float* dbeta, int accumulate_dx, hipStream_t s) {
  const int chunks = bn_fused_chunks(rows, C);
  hipLaunchKernelGGL(bn_bwd_finalize_apply_kernel, dim3((C / BNF_SLAB) * chunks), dim3(256), 0, s, partial, tiles, (const bf16_t*)x, (const bf16_t*)g,
                     (bf16_t*)dx, accumulate_dx ? (const bf16_t*)dadd : (const bf16_t*)nullptr, rows, C, chunks, (float)(1.0 / (double)rows), mean, rstd,
                     gamma, dgamma, dbeta);
  STP_LAUNCH_CHECK();
  return STP_OK;
}

// [2][C][tiles] epilogue partials -> sums (same contract as bn_bwd_finalize_kernel); one workgroup per channel
__global__ __launch_bounds__(256) void bn_bwd_finalize_tiles_kernel(const float* __restrict__ partial, int tiles, int C, float* sums,
                                                                    float* dgamma, float* dbeta) {
  __shared__ double sh[2][4];
  const int c = blockIdx.x;
  double S, Q;
  bnf_channel_sums(partial + (size_t)c * tiles, partial + ((size_t)C + c) * tiles, tiles, sh, S, Q);
  if (threadIdx.x != 0) return;
  sums[c] = (float)S;
  sums[C + c] = (float)Q;
  if (dbeta) dbeta[c] = (float)S;
  if (dgamma) dgamma[c] = (float)Q;
}

extern "C" int stp_bn_backward_slots(const void* x, const void* g, void* dx, int32_t dtype, int64_t rows, int32_t C, const float* mean,
                                     const float* rstd, const float* gamma, const int64_t* slots, int32_t nslots, float* dgamma,
                                     float* dbeta, int32_t accumulate_dx, void* stream) {
  if (!stp_dtype_ok(dtype)) return STP_E_BADARG;      // (the other build's 16-bit code, or garbage)
  if (!x || !g || !dx || !mean || !rstd || !slots || rows <= 0 || C <= 0 || (C & 3) || nslots < 1 || (nslots & (nslots - 1))) return STP_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  const bool v8 = dtype == STP_H16 && (C & 7) == 0;
  const size_t lds2 = 6 * (size_t)C * sizeof(float);
  const int gr = grid_fixed_channels(grid_for_amortised(rows * (C / (v8 ? 8 : 4)), 8), C / (v8 ? 8 : 4));
  const float inv_rows = (float)(1.0 / (double)rows);
  const long long* sl = (const long long*)slots;
  if (v8)
    hipLaunchKernelGGL((bn_bwd_apply_kernel<bf16_t, 8>), dim3(gr), dim3(256), lds2, s, (const bf16_t*)x, (const bf16_t*)g, (bf16_t*)dx, rows, C,
                       mean, rstd, gamma, (const float*)nullptr, (const float*)nullptr, inv_rows, 0, accumulate_dx, sl, nslots, dgamma, dbeta);
  else if (dtype == STP_H16)
    hipLaunchKernelGGL((bn_bwd_apply_kernel<bf16_t, 4>), dim3(gr), dim3(256), lds2, s, (const bf16_t*)x, (const bf16_t*)g, (bf16_t*)dx, rows, C,
                       mean, rstd, gamma, (const float*)nullptr, (const float*)nullptr, inv_rows, 0, accumulate_dx, sl, nslots, dgamma, dbeta);
  else if (dtype == STP_F32)
    hipLaunchKernelGGL((bn_bwd_apply_kernel<float, 4>), dim3(gr), dim3(256), lds2, s, (const float*)x, (const float*)g, (float*)dx, rows, C,
                       mean, rstd, gamma, (const float*)nullptr, (const float*)nullptr, inv_rows, 0, accumulate_dx, sl, nslots, dgamma, dbeta);
  else
    return STP_E_BADARG;
  STP_LAUNCH_CHECK();
  return STP_OK;
}

extern "C" int stp_bn_backward_fused_add(const void* x, const void* g, void* dx, const void* dadd, int32_t dtype, int64_t rows, int32_t C,
                                     const float* mean, const float* rstd, const float* gamma, const float* partial,
                                     int32_t tiles, float* dgamma, float* dbeta, int32_t accumulate_dx, void* workspace,
                                     size_t workspace_bytes, void* stream) {
  if (!stp_dtype_ok(dtype)) return STP_E_BADARG;      // (the other build's 16-bit code, or garbage)
  if (!x || !g || !dx || (accumulate_dx && !dadd) || !mean || !rstd || !partial || !workspace || rows <= 0 || C <= 0 || (C & 3) || tiles <= 0) return STP_E_BADARG;
  if (workspace_bytes < stp_bn_workspace_bytes(C)) return STP_E_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  if (bn_fused_ok(dtype, rows, C, tiles))      // one launch: every workgroup reduces the partial sums of its own 64-channel slab
    return bn_bwd_fused_launch(x, g, dx, dadd, rows, C, mean, rstd, gamma, partial, tiles, dgamma, dbeta, accumulate_dx, s);
  float* sums = (float*)workspace + (size_t)(BN_MAX_BLOCKS - 1) * 2 * C;
  hipLaunchKernelGGL(bn_bwd_finalize_tiles_kernel, dim3(C), dim3(256), 0, s, partial, tiles, C, sums, dgamma, dbeta);
  STP_LAUNCH_CHECK();
  const bool v8 = dtype == STP_H16 && (C & 7) == 0;
  const size_t lds2 = 6 * (size_t)C * sizeof(float);
  const int gr = grid_fixed_channels(grid_for_amortised(rows * (C / (v8 ? 8 : 4)), 8), C / (v8 ? 8 : 4));
  const float inv_rows = (float)(1.0 / (double)rows);
  // the ReLU mask is already folded into g
  if (v8)
    hipLaunchKernelGGL((bn_bwd_apply_kernel<bf16_t, 8>), dim3(gr), dim3(256), lds2, s, (const bf16_t*)x, (const bf16_t*)g,
                       (bf16_t*)dx, rows, C, mean, rstd, gamma, (const float*)nullptr, sums, inv_rows, 0, accumulate_dx, (const long long*)nullptr, 0, (float*)nullptr, (float*)nullptr,
                       (const bf16_t*)dadd);
  else if (dtype == STP_H16)
    hipLaunchKernelGGL((bn_bwd_apply_kernel<bf16_t, 4>), dim3(gr), dim3(256), lds2, s, (const bf16_t*)x, (const bf16_t*)g,
                       (bf16_t*)dx, rows, C, mean, rstd, gamma, (const float*)nullptr, sums, inv_rows, 0, accumulate_dx, (const long long*)nullptr, 0, (float*)nullptr, (float*)nullptr,
                       (const bf16_t*)dadd);
  else if (dtype == STP_F32)
    hipLaunchKernelGGL((bn_bwd_apply_kernel<float, 4>), dim3(gr), dim3(256), lds2, s, (const float*)x, (const float*)g, (float*)dx,
                       rows, C, mean, rstd, gamma, (const float*)nullptr, sums, inv_rows, 0, accumulate_dx, (const long long*)nullptr, 0, (float*)nullptr, (float*)nullptr,
                       (const float*)dadd);
  else
    return STP_E_BADARG;
  STP_LAUNCH_CHECK();
  return STP_OK;
}

extern "C" int stp_bn_backward_fused(const void* x, const void* g, void* dx, int32_t dtype, int64_t rows, int32_t C,
                                     const float* mean, const float* rstd, const float* gamma, const float* partial,
                                     int32_t tiles, float* dgamma, float* dbeta, int32_t accumulate_dx, void* workspace,
                                     size_t workspace_bytes, void* stream) {
  return stp_bn_backward_fused_add(x, g, dx, dx, dtype, rows, C, mean, rstd, gamma, partial, tiles, dgamma, dbeta, accumulate_dx, workspace,
                                   workspace_bytes, stream);
}

// ------------------------------------------------------------------------------------------
// ZeroPadding2D(1) + MaxPooling2D(3, 2, valid).  idx[n,ho,wo,c] = kh*3+kw of the first maximum
// (padded taps take part with value 0, as in the Keras graph).
template <typename T, int V>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, uint8_t* __restrict__ idx,
                                                          int N, int H, int W, int C, int Ho, int Wo) {
  const int cg = C / V;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= Wo * cg) return;
  const int wo = t / cg, c = (t - wo * cg) * V;
  for (int rr = 0; rr < POOL_ROWS; ++rr) {
  const int orow = blockIdx.y * POOL_ROWS + rr;
  if (orow >= N * Ho) break;
  const int n = orow / Ho, ho = orow - n * Ho;
  float best[V];
  uint8_t bi[V];
  // all nine taps are fetched BEFORE the first comparison (clamped addresses, out-of-image taps replaced by 0 afterwards): one
  // memory round trip per thread instead of nine dependent ones (the predicated form waited for every tap before the next)
  float v[9][V];
#pragma unroll
  for (int kh = 0; kh < 3; ++kh)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int h = min(max(2 * ho - 1 + kh, 0), H - 1), w = min(max(2 * wo - 1 + kw, 0), W - 1);
      ldv<T, V>(x + (((size_t)n * H + h) * W + w) * C + c, v[kh * 3 + kw]);
    }
#pragma unroll
  for (int kh = 0; kh < 3; ++kh)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int h = 2 * ho - 1 + kh, w = 2 * wo - 1 + kw;
      const bool in = (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W;      // padded taps take part with value 0
#pragma unroll
      for (int e = 0; e < V; ++e) {
        const float tv = in ? v[kh * 3 + kw][e] : 0.f;
        if ((kh | kw) == 0 || tv > best[e]) { best[e] = tv; bi[e] = (uint8_t)(kh * 3 + kw); }
      }
    }
  const size_t o = (((size_t)n * Ho + ho) * Wo + wo) * C + c;
  stv<T, V>(y + o, best);
  if (idx) {
    if constexpr (V == 8) *reinterpret_cast<uint2*>(idx + o) = *reinterpret_cast<const uint2*>(bi);
    else *reinterpret_cast<uint32_t*>(idx + o) = *reinterpret_cast<const uint32_t*>(bi);
  }
  }
}

// BatchNormalization apply (+ activation) FUSED with the max-pooling that follows it (round 5: the stem's bn0 -> relu0 -> pooling0).
// The thread of pooled pixel (ho, wo) fetches the nine taps of the PRE-normalisation tensor, normalises them with the arithmetic and
// the rounding of stp_bn_apply (the maximum is taken over the values AS STORED), writes the pooled value + index, and stores the
// normalised tensor for the 2 x 2 block of input pixels (2ho, 2wo) .. (2ho + 1, 2wo + 1) - its taps (1..2, 1..2): every input pixel
// of an even-sized map belongs to exactly one block.  One read of the 134 MB tensor instead of two (apply: read + write, pool: read).
__global__ __launch_bounds__(256) void bn_apply_maxpool_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ yb, bf16_t* __restrict__ y,
                                                               uint8_t* __restrict__ idx, int N, int H, int W, int C, int Ho, int Wo,
                                                               const float* mean, const float* rstd, const float* gamma, const float* beta, int relu) {
  const int cg = C >> 3;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= Wo * cg) return;
  const int wo = t / cg, c = (t - wo * cg) * 8;
  float sc[8], sh[8];
  {
    const f32x4 m0 = *reinterpret_cast<const f32x4*>(mean + c), m1 = *reinterpret_cast<const f32x4*>(mean + c + 4);
    const f32x4 r0 = *reinterpret_cast<const f32x4*>(rstd + c), r1 = *reinterpret_cast<const f32x4*>(rstd + c + 4);
    f32x4 g0 = {1.f, 1.f, 1.f, 1.f}, g1 = g0, b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
    if (gamma) { g0 = *reinterpret_cast<const f32x4*>(gamma + c); g1 = *reinterpret_cast<const f32x4*>(gamma + c + 4); }
    if (beta) { b0 = *reinterpret_cast<const f32x4*>(beta + c); b1 = *reinterpret_cast<const f32x4*>(beta + c + 4); }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float r = e < 4 ? r0[e & 3] : r1[e & 3], k = gamma ? r * (e < 4 ? g0[e & 3] : g1[e & 3]) : r;
      sc[e] = k;
      sh[e] = (e < 4 ? b0[e & 3] : b1[e & 3]) - (e < 4 ? m0[e & 3] : m1[e & 3]) * k;
    }
  }
  for (int rr = 0; rr < POOL_ROWS; ++rr) {
    const int orow = blockIdx.y * POOL_ROWS + rr;
    if (orow >= N * Ho) break;
    const int n = orow / Ho, ho = orow - n * Ho;
    u32x4 v[9];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int h = min(max(2 * ho - 1 + kh, 0), H - 1), w = min(max(2 * wo - 1 + kw, 0), W - 1);
        v[kh * 3 + kw] = *reinterpret_cast<const u32x4*>(x + (((size_t)n * H + h) * W + w) * C + c);
      }
    float best[8];
    uint8_t bi[8];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int h = 2 * ho - 1 + kh, w = 2 * wo - 1 + kw;
        const bool in = (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W;      // padded taps take part with value 0
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float lo = bn_act(bn_affine(h16lo_to_f32(v[kh * 3 + kw][e]), sc[2 * e], sh[2 * e]), relu);
          const float hi = bn_act(bn_affine(h16hi_to_f32(v[kh * 3 + kw][e]), sc[2 * e + 1], sh[2 * e + 1]), relu);
          o[e] = pack_bf16x2(lo, hi);
          const float tl = in ? h16lo_to_f32(o[e]) : 0.f, th = in ? h16hi_to_f32(o[e]) : 0.f;      // the values as stored
          if ((kh | kw) == 0 || tl > best[2 * e]) { best[2 * e] = tl; bi[2 * e] = (uint8_t)(kh * 3 + kw); }
          if ((kh | kw) == 0 || th > best[2 * e + 1]) { best[2 * e + 1] = th; bi[2 * e + 1] = (uint8_t)(kh * 3 + kw); }
        }
        if (kh >= 1 && kw >= 1 && in) *reinterpret_cast<u32x4*>(yb + (((size_t)n * H + h) * W + w) * C + c) = o;   // this thread's 2 x 2 block
      }
    const size_t oo = (((size_t)n * Ho + ho) * Wo + wo) * C + c;
    *reinterpret_cast<u32x4*>(y + oo) = u32x4{pack_bf16x2(best[0], best[1]), pack_bf16x2(best[2], best[3]), pack_bf16x2(best[4], best[5]),
                                             pack_bf16x2(best[6], best[7])};
    if (idx) *reinterpret_cast<uint2*>(idx + oo) = *reinterpret_cast<const uint2*>(bi);
  }
}
// H, W even, C % 8 == 0, 16-bit storage.  x = the tensor BEFORE the BatchNormalization, yb = the normalised (+ activated) tensor
// [N,H,W,C] (stp_bn_apply's output, bit for bit), y / idx = stp_maxpool3x3s2's outputs for it.
extern "C" int stp_bn_apply_maxpool3x3s2(const void* x, void* yb, void* y, uint8_t* idx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t dtype,
                                         const float* mean, const float* rstd, const float* gamma, const float* beta, int32_t relu, void* stream) {
  if (!stp_dtype_ok(dtype)) return STP_E_BADARG;      // (the other build's 16-bit code, or garbage)
  if (dtype != STP_H16 || !x || !yb || !y || !mean || !rstd || N <= 0 || (C & 7) || (H & 1) || (W & 1)) return STP_E_BADARG;
  const int Ho = H / 2, Wo = W / 2;
  if ((int64_t)N * Ho > 65535 * POOL_ROWS) return STP_E_BADARG;  // gridDim.y
  const dim3 grid(ceil_div(Wo * (C / 8), 256), ceil_div(N * Ho, POOL_ROWS));
  hipLaunchKernelGGL(bn_apply_maxpool_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)yb, (bf16_t*)y, idx, N, H, W, C, Ho, Wo,
                     mean, rstd, gamma, beta, relu);
  STP_LAUNCH_CHECK();
  return STP_OK;
}

// ------------------------------------------------------------------------------------------
// Shared tail of the gradient kernels that COMPLETE the gradient of a BatchNormalization(+activation) output (the 2x2 fold of
// UpSampling2D, the max-pool gather): the thread's V values are masked with the activation re-derived from the BN input x,
// stored, and the workgroup reduces sum(g) / sum(g * xhat) per channel into partial[2][C][workgroups] - the layout
// stp_bn_backward_fused consumes, exactly what the convolution epilogues write (stp_conv_params.bnb_x).
// Thread layout of the callers: t = blockIdx.x * 256 + threadIdx.x = pixel * cg + channel group, cg = C / V dividing 256.
__device__ __forceinline__ f32x4 stored4(f32x4 v, const float*) { return v; }   // the value as the destination dtype holds it
__device__ __forceinline__ f32x4 stored4(f32x4 v, const bf16_t*) {
  const uint32_t a = pack_bf16x2(v.x, v.y), b = pack_bf16x2(v.z, v.w);
  return f32x4{h16lo_to_f32(a), h16hi_to_f32(a), h16lo_to_f32(b), h16hi_to_f32(b)};
}

#define BNB_ROWS 8   // image rows per workgroup of the *_bn gradient kernels: fewer, larger partial-sum tiles


template <typename T, int V>
__device__ __forceinline__ void bnb_mask_store(float (&g)[V], size_t o, int c, const BnBack& bnb, T* __restrict__ dx, float (&sg)[V],
                                               float (&sq)[V]) {
  float xv[V];
  ldv<T, V>(reinterpret_cast<const T*>(bnb.x) + o, xv);
#pragma unroll
  for (int q = 0; q < V / 4; ++q) {
    const BnBackCh k = bnback_load(bnb, c + 4 * q);
    const f32x4 st = stored4(f32x4{g[4 * q], g[4 * q + 1], g[4 * q + 2], g[4 * q + 3]}, (const T*)nullptr);
    f32x4 s4 = {0.f, 0.f, 0.f, 0.f}, q4 = {0.f, 0.f, 0.f, 0.f};
    const f32x4 m = bnback_apply(k, bnb.relu, f32x4{xv[4 * q], xv[4 * q + 1], xv[4 * q + 2], xv[4 * q + 3]}, st, s4, q4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { g[4 * q + e] = m[e]; sg[4 * q + e] += s4[e]; sq[4 * q + e] += q4[e]; }
  }
  stv<T, V>(dx + o, g);
}

// threads t and t + cg own the same channels: fixed-order combine over the workgroup's 256 / cg pixel columns
template <int V>
__device__ __forceinline__ void bnb_reduce(const float (&sg)[V], const float (&sq)[V], int cg, int C, float* __restrict__ partial) {
  extern __shared__ float red[];  // [256][2V]
#pragma unroll
  for (int e = 0; e < V; ++e) { red[threadIdx.x * 2 * V + e] = sg[e]; red[threadIdx.x * 2 * V + V + e] = sq[e]; }
  __syncthreads();
  const int cl = cg < 256 ? cg : 256;               // channel groups present in this workgroup
  for (int idx = threadIdx.x; idx < cl * V; idx += 256) {
    const int gq = idx / V, e = idx - gq * V;
    const int cg0 = (blockIdx.x * 256 + gq) % cg;   // channel group of local thread gq
    float s = 0.f, q2 = 0.f;
    for (int l = gq; l < 256; l += cl) { s += red[l * 2 * V + e]; q2 += red[l * 2 * V + V + e]; }
    const size_t nblk = (size_t)gridDim.x * gridDim.y, blk = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    const int ch = cg0 * V + e;
    partial[(size_t)ch * nblk + blk] = s;
    partial[((size_t)C + ch) * nblk + blk] = q2;
  }
}

// One workgroup row per input row (blockIdx.y = n*H + h): 32-bit index arithmetic only, V channels per thread
// (16 bytes of bf16), each input pixel gathers from the <= 4 windows that contain it.
template <typename T, int V, bool BNB>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const uint8_t* __restrict__ idx, const T* __restrict__ dy,
                                                          T* __restrict__ dx, int N, int H, int W, int C, int Ho, int Wo,
                                                          int accumulate, BnBack bnb, float* __restrict__ partial) {
  const int cg = C / V;
  const int t = blockIdx.x * 256 + threadIdx.x;
  const bool on = t < W * cg;
  if (!BNB && !on) return;
  const int w = on ? t / cg : 0, c = on ? (t - w * cg) * V : 0;
  constexpr int ROWS = BNB ? BNB_ROWS : POOL_ROWS;
  float sg[V], sq[V];
#pragma unroll
  for (int e = 0; e < V; ++e) { sg[e] = 0.f; sq[e] = 0.f; }
  for (int rr = 0; rr < ROWS; ++rr) {
    const int row = blockIdx.y * ROWS + rr;
    if (row >= N * H || !on) break;
    const int n = row / H, h = row - n * H;
    float g[V];
#pragma unroll
    for (int e = 0; e < V; ++e) g[e] = 0.f;
    const size_t oo = (((size_t)n * H + h) * W + w) * C + c;
    // windows (ho,wo) with 2*ho-1+kh == h  ->  kh = h+1-2*ho in [0,2]: an even row has ONE (kh = 1), an odd row TWO (kh = 0
    // of ho = (h+1)/2 and kh = 2 of ho = (h-1)/2); the same along w.  The <= 4 (index, gradient) pairs are fetched up front
    // with clamped addresses - one memory round trip - and the invalid ones masked afterwards.
    int hos[2], khs[2], wos[2], kws[2];
    bool hv[2], wv[2];
    if (h & 1) { hos[0] = (h + 1) >> 1; khs[0] = 0; hos[1] = (h - 1) >> 1; khs[1] = 2; hv[0] = hos[0] < Ho; hv[1] = true; }
    else { hos[0] = h >> 1; khs[0] = 1; hos[1] = 0; khs[1] = 1; hv[0] = hos[0] < Ho; hv[1] = false; }
    if (w & 1) { wos[0] = (w + 1) >> 1; kws[0] = 0; wos[1] = (w - 1) >> 1; kws[1] = 2; wv[0] = wos[0] < Wo; wv[1] = true; }
    else { wos[0] = w >> 1; kws[0] = 1; wos[1] = 0; kws[1] = 1; wv[0] = wos[0] < Wo; wv[1] = false; }
    uint8_t id[4][V];
    float d[4][V];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const size_t o = (((size_t)n * Ho + min(hos[a], Ho - 1)) * Wo + min(wos[b], Wo - 1)) * C + c;
        if constexpr (V == 8) *reinterpret_cast<uint2*>(id[a * 2 + b]) = *reinterpret_cast<const uint2*>(idx + o);
        else *reinterpret_cast<uint32_t*>(id[a * 2 + b]) = *reinterpret_cast<const uint32_t*>(idx + o);
        ldv<T, V>(dy + o, d[a * 2 + b]);
      }
    // (summation order = the former kh-major / kw-minor walk: kh 0 before kh 2, kw 0 before kw 2)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const bool ok = hv[a] && wv[b];
        const uint8_t me = (uint8_t)(khs[a] * 3 + kws[b]);
#pragma unroll
        for (int e = 0; e < V; ++e)
          if (ok && id[a * 2 + b][e] == me) g[e] += d[a * 2 + b][e];
      }
    if (accumulate) {
      float o[V];
      ldv<T, V>(dx + oo, o);
#pragma unroll
      for (int e = 0; e < V; ++e) g[e] += o[e];
    }
    if constexpr (BNB) bnb_mask_store<T, V>(g, oo, c, bnb, dx, sg, sq);
    else stv<T, V>(dx + oo, g);
  }
  if constexpr (BNB) bnb_reduce<V>(sg, sq, cg, C, partial);
}

extern "C" int stp_maxpool3x3s2(const void* x, void* y, uint8_t* idx, int32_t N, int32_t H, int32_t W, int32_t C,
                                int32_t dtype, void* stream) {
  if (!stp_dtype_ok(dtype)) return STP_E_BADARG;      // (the other build's 16-bit code, or garbage)
  if (!x || !y || (C & 3) || N <= 0) return STP_E_BADARG;
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  if ((int64_t)N * Ho > 65535) return STP_E_BADARG;  // gridDim.y
  hipStream_t s = (hipStream_t)stream;
  const bool v8 = dtype == STP_H16 && (C & 7) == 0;
  const dim3 grid(ceil_div(Wo * (C / (v8 ? 8 : 4)), 256), ceil_div(N * Ho, POOL_ROWS));
  if (v8)
    hipLaunchKernelGGL((maxpool_fwd_kernel<bf16_t, 8>), grid, dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, idx, N, H, W, C, Ho, Wo);
  else if (dtype == STP_H16)
    hipLaunchKernelGGL((maxpool_fwd_kernel<bf16_t, 4>), grid, dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, idx, N, H, W, C, Ho, Wo);
  else if (dtype == STP_F32)
    hipLaunchKernelGGL((maxpool_fwd_kernel<float, 4>), grid, dim3(256), 0, s, (const float*)x, (float*)y, idx, N, H, W, C, Ho, Wo);
  else
    return STP_E_BADARG;
  STP_LAUNCH_CHECK();
  return STP_OK;
}

static int maxpool_bwd_launch(const uint8_t* idx, const void* dy, void* dx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t dtype,
                              int32_t accumulate, const BnBack* bnb, float* partial, hipStream_t s) {
  if (!idx || !dy || !dx || (C & 3) || N <= 0) return STP_E_BADARG;
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  if ((int64_t)N * H > 65535) return STP_E_BADARG;  // gridDim.y
  const bool v8 = dtype == STP_H16 && (C & 7) == 0;
  const int V = v8 ? 8 : 4;
  const dim3 grid(ceil_div(W * (C / V), 256), ceil_div(N * H, POOL_ROWS));
  BnBack none;
  none.x = nullptr; none.mean = none.rstd = none.gamma = none.beta = nullptr; none.relu = 0;
  if (bnb) {
    if (256 % (C / V) != 0 || !partial) return STP_E_BADARG;
    const size_t lds = 256 * 2 * V * sizeof(float);
    const dim3 gridb(grid.x, ceil_div(N * H, BNB_ROWS));
    if (v8) hipLaunchKernelGGL((maxpool_bwd_kernel<bf16_t, 8, true>), gridb, dim3(256), lds, s, idx, (const bf16_t*)dy, (bf16_t*)dx, N, H, W, C, Ho, Wo, accumulate, *bnb, partial);
    else if (dtype == STP_H16) hipLaunchKernelGGL((maxpool_bwd_kernel<bf16_t, 4, true>), gridb, dim3(256), lds, s, idx, (const bf16_t*)dy, (bf16_t*)dx, N, H, W, C, Ho, Wo, accumulate, *bnb, partial);
    else if (dtype == STP_F32) hipLaunchKernelGGL((maxpool_bwd_kernel<float, 4, true>), gridb, dim3(256), lds, s, idx, (const float*)dy, (float*)dx, N, H, W, C, Ho, Wo, accumulate, *bnb, partial);
    else return STP_E_BADARG;
  } else {
    if (v8) hipLaunchKernelGGL((maxpool_bwd_kernel<bf16_t, 8, false>), grid, dim3(256), 0, s, idx, (const bf16_t*)dy, (bf16_t*)dx, N, H, W, C, Ho, Wo, accumulate, none, nullptr);
    else if (dtype == STP_H16) hipLaunchKernelGGL((maxpool_bwd_kernel<bf16_t, 4, false>), grid, dim3(256), 0, s, idx, (const bf16_t*)dy, (bf16_t*)dx, N, H, W, C, Ho, Wo, accumulate, none, nullptr);
    else if (dtype == STP_F32) hipLaunchKernelGGL((maxpool_bwd_kernel<float, 4, false>), grid, dim3(256), 0, s, idx, (const float*)dy, (float*)dx, N, H, W, C, Ho, Wo, accumulate, none, nullptr);
    else return STP_E_BADARG;
  }
  STP_LAUNCH_CHECK();
  return STP_OK;
}

extern "C" int stp_maxpool3x3s2_bwd(const uint8_t* idx, const void* dy, void* dx, int32_t N, int32_t H, int32_t W, int32_t C,
                                    int32_t dtype, int32_t accumulate, void* stream) {
  if (!stp_dtype_ok(dtype)) return STP_E_BADARG;      // (the other build's 16-bit code, or garbage)
  return maxpool_bwd_launch(idx, dy, dx, N, H, W, C, dtype, accumulate, nullptr, nullptr, (hipStream_t)stream);
}

// Same tile count rule as stp_upsample2x_bwd_bn_tiles (H, W = the pool INPUT size): workgroups of the launch, 0 = unsupported C.
extern "C" int stp_maxpool3x3s2_bwd_bn_tiles(int32_t N, int32_t H, int32_t W, int32_t C, int32_t dtype) {
  if (!stp_dtype_ok(dtype)) return STP_E_BADARG;      // (the other build's 16-bit code, or garbage)
  const int V = (dtype == STP_H16 && (C & 7) == 0) ? 8 : 4;
  if (C <= 0 || (C & 3) || 256 % (C / V) != 0) return 0;
  return ceil_div(W * (C / V), 256) * ceil_div(N * H, BNB_ROWS);
}

extern "C" int stp_maxpool3x3s2_bwd_bn(const uint8_t* idx, const void* dy, void* dx, int32_t N, int32_t H, int32_t W, int32_t C,
                                       int32_t dtype, int32_t accumulate, const void* bn_x, const float* mean, const float* rstd,
                                       const float* gamma, const float* beta, int32_t relu, float* partial, void* stream) {
  if (!stp_dtype_ok(dtype)) return STP_E_BADARG;      // (the other build's 16-bit code, or garbage)
  if (!bn_x || !mean || !rstd || !partial) return STP_E_BADARG;
  BnBack b;
  b.x = (const char*)bn_x; b.mean = mean; b.rstd = rstd; b.gamma = gamma; b.beta = beta; b.relu = relu;
  return maxpool_bwd_launch(idx, dy, dx, N, H, W, C, dtype, accumulate, &b, partial, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------
// Data gradient of a 1x1 / stride-2 convolution (the projection shortcut of a bottleneck ResNet's first unit), second half: the GEMM
// t = W^T dY runs at LOW resolution ([N, Ho, Wo, C], a plain 1x1 / stride-1 launch); this pass puts t[n, a, b] at (2a, 2b) of the
// [N, H, W, C] gradient - zeros elsewhere, or on top of what the tensor's other consumers wrote (accumulate) - and, when it completes the
// gradient of a BatchNormalization(+activation) output, masks it and reduces the backward sums (bnb_mask_store / bnb_reduce, as the
// max-pool gather above).  The zero-inserted form it replaces ran the GEMM over all four parity classes of the high-resolution grid:
// 229 us for 256 <- 512 channels at 4 x 256 x 256 (FPN/ResNet50), 3.4 x its memory floor.
template <typename T, int V, bool BNB>
__global__ __launch_bounds__(256) void scatter2x_bwd_kernel(const T* __restrict__ t, T* __restrict__ dx, int N, int H, int W, int C, int Ho, int Wo,
                                                            int accumulate, BnBack bnb, float* __restrict__ partial) {
  const int cg = C / V;
  const int tt = blockIdx.x * 256 + threadIdx.x;
  const bool on = tt < W * cg;
  if (!BNB && !on) return;
  const int w = on ? tt / cg : 0, c = on ? (tt - w * cg) * V : 0;
  constexpr int ROWS = BNB ? BNB_ROWS : POOL_ROWS;
  constexpr int RB = 4;                                 // rows requested together: one memory round trip per RB rows, not per row
  static_assert(ROWS % RB == 0, "row batches");
  float sg[V], sq[V];
#pragma unroll
  for (int e = 0; e < V; ++e) { sg[e] = 0.f; sq[e] = 0.f; }
  BnBackCh kc[V / 4];
  if constexpr (BNB) {
#pragma unroll
    for (int q = 0; q < V / 4; ++q) kc[q] = bnback_load(bnb, c + 4 * q);
  }
  const bool wev = !(w & 1) && (w >> 1) < Wo;
  for (int r0 = 0; r0 < ROWS; r0 += RB) {
    float g[RB][V], o[RB][V], xv[RB][V];
    bool live[RB];
#pragma unroll
    for (int j = 0; j < RB; ++j) {
      const int row = blockIdx.y * ROWS + r0 + j;
      live[j] = on && row < N * H;
      const int n = live[j] ? row / H : 0, h = live[j] ? row - n * H : 0;
      const size_t oo = (((size_t)n * H + h) * W + w) * C + c;
      const bool hit = live[j] && wev && !(h & 1) && (h >> 1) < Ho;
      if (!BNB && accumulate && !hit) live[j] = false;   // nothing to add there, nothing to mask: the position keeps what it holds
#pragma unroll
      for (int e = 0; e < V; ++e) g[j][e] = o[j][e] = xv[j][e] = 0.f;
      if (hit) ldv<T, V>(t + (((size_t)n * Ho + (h >> 1)) * Wo + (w >> 1)) * C + c, g[j]);
      if (live[j] && accumulate) ldv<T, V>(dx + oo, o[j]);
      if (BNB && live[j]) ldv<T, V>(reinterpret_cast<const T*>(bnb.x) + oo, xv[j]);
    }
#pragma unroll
    for (int j = 0; j < RB; ++j) {
      if (!live[j]) continue;
      const int row = blockIdx.y * ROWS + r0 + j;
      const int n = row / H, h = row - n * H;
      const size_t oo = (((size_t)n * H + h) * W + w) * C + c;
#pragma unroll
      for (int e = 0; e < V; ++e) g[j][e] += o[j][e];
      if constexpr (BNB) {
#pragma unroll
        for (int q = 0; q < V / 4; ++q) {
          const f32x4 st = stored4(f32x4{g[j][4 * q], g[j][4 * q + 1], g[j][4 * q + 2], g[j][4 * q + 3]}, (const T*)nullptr);
          f32x4 s4 = {0.f, 0.f, 0.f, 0.f}, q4 = {0.f, 0.f, 0.f, 0.f};
          const f32x4 m = bnback_apply(kc[q], bnb.relu, f32x4{xv[j][4 * q], xv[j][4 * q + 1], xv[j][4 * q + 2], xv[j][4 * q + 3]}, st, s4, q4);
#pragma unroll
          for (int e = 0; e < 4; ++e) { g[j][4 * q + e] = m[e]; sg[4 * q + e] += s4[e]; sq[4 * q + e] += q4[e]; }
        }
      }
      stv<T, V>(dx + oo, g[j]);
    }
  }
  if constexpr (BNB) bnb_reduce<V>(sg, sq, cg, C, partial);
}

static int scatter2x_bwd_launch(const void* t, void* dx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t dtype, int32_t accumulate,
                                const BnBack* bnb, float* partial, hipStream_t s) {
  if (!t || !dx || (C & 3) || N <= 0 || H <= 0 || W <= 0) return STP_E_BADARG;
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  if ((int64_t)N * H > 65535 * (int64_t)(bnb ? BNB_ROWS : POOL_ROWS)) return STP_E_BADARG;      // gridDim.y
  const bool v8 = dtype == STP_H16 && (C & 7) == 0;
  const int V = v8 ? 8 : 4;
  BnBack none;
  none.x = nullptr; none.mean = none.rstd = none.gamma = none.beta = nullptr; none.relu = 0;
  if (bnb) {
    if (256 % (C / V) != 0 || !partial) return STP_E_BADARG;
    const size_t lds = 256 * 2 * V * sizeof(float);
    const dim3 grid(ceil_div(W * (C / V), 256), ceil_div(N * H, BNB_ROWS));
    if (v8) hipLaunchKernelGGL((scatter2x_bwd_kernel<bf16_t, 8, true>), grid, dim3(256), lds, s, (const bf16_t*)t, (bf16_t*)dx, N, H, W, C, Ho, Wo, accumulate, *bnb, partial);
    else if (dtype == STP_H16) hipLaunchKernelGGL((scatter2x_bwd_kernel<bf16_t, 4, true>), grid, dim3(256), lds, s, (const bf16_t*)t, (bf16_t*)dx, N, H, W, C, Ho, Wo, accumulate, *bnb, partial);
    else if (dtype == STP_F32) hipLaunchKernelGGL((scatter2x_bwd_kernel<float, 4, true>), grid, dim3(256), lds, s, (const float*)t, (float*)dx, N, H, W, C, Ho, Wo, accumulate, *bnb, partial);
    else return STP_E_BADARG;
  } else {
    const dim3 grid(ceil_div(W * (C / V), 256), ceil_div(N * H, POOL_ROWS));
    if (v8) hipLaunchKernelGGL((scatter2x_bwd_kernel<bf16_t, 8, false>), grid, dim3(256), 0, s, (const bf16_t*)t, (bf16_t*)dx, N, H, W, C, Ho, Wo, accumulate, none, nullptr);
    else if (dtype == STP_H16) hipLaunchKernelGGL((scatter2x_bwd_kernel<bf16_t, 4, false>), grid, dim3(256), 0, s, (const bf16_t*)t, (bf16_t*)dx, N, H, W, C, Ho, Wo, accumulate, none, nullptr);
    else if (dtype == STP_F32) hipLaunchKernelGGL((scatter2x_bwd_kernel<float, 4, false>), grid, dim3(256), 0, s, (const float*)t, (float*)dx, N, H, W, C, Ho, Wo, accumulate, none, nullptr);
    else return STP_E_BADARG;
  }
  STP_LAUNCH_CHECK();
  return STP_OK;
}

// t: [N, (H - 1) / 2 + 1, (W - 1) / 2 + 1, C]; dx: [N, H, W, C].  accumulate != 0: dx += the scattered t (odd positions untouched).
extern "C" int stp_scatter2x_bwd(const void* t, void* dx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t dtype, int32_t accumulate, void* stream) {
  if (!stp_dtype_ok(dtype)) return STP_E_BADARG;      // (the other build's 16-bit code, or garbage)
  return scatter2x_bwd_launch(t, dx, N, H, W, C, dtype, accumulate, nullptr, nullptr, (hipStream_t)stream);
}

// columns of the [2][C][tiles] partial table of stp_scatter2x_bwd_bn (H, W = the HIGH-resolution size); 0 = unsupported C
extern "C" int stp_scatter2x_bwd_bn_tiles(int32_t N, int32_t H, int32_t W, int32_t C, int32_t dtype) {
  if (!stp_dtype_ok(dtype)) return STP_E_BADARG;      // (the other build's 16-bit code, or garbage)
  const int V = (dtype == STP_H16 && (C & 7) == 0) ? 8 : 4;
  if (C <= 0 || (C & 3) || 256 % (C / V) != 0) return 0;
  return ceil_div(W * (C / V), 256) * ceil_div(N * H, BNB_ROWS);
}

// ... and the completed gradient masked with the activation of the BatchNormalization whose output the tensor is (bn_x: its input),
// stored in place of dY, sum g / sum g * xhat reduced into partial[2][C][tiles] for stp_bn_backward_fused
extern "C" int stp_scatter2x_bwd_bn(const void* t, void* dx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t dtype, int32_t accumulate,
                                    const void* bn_x, const float* mean, const float* rstd, const float* gamma, const float* beta, int32_t relu,
                                    float* partial, void* stream) {
  if (!stp_dtype_ok(dtype)) return STP_E_BADARG;      // (the other build's 16-bit code, or garbage)
  if (!bn_x || !mean || !rstd || !partial) return STP_E_BADARG;
  BnBack b;
  b.x = (const char*)bn_x; b.mean = mean; b.rstd = rstd; b.gamma = gamma; b.beta = beta; b.relu = relu;
  return scatter2x_bwd_launch(t, dx, N, H, W, C, dtype, accumulate, &b, partial, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------
// Conv2D(3x3, padding 1) with FEW output channels over MANY input channels (the class heads of FPN and PSPNet: 512 -> 3 / 20) as a 1x1
// convolution into 9 x Cout TAP CHANNELS + a tap sum.  Channel mixing is pointwise, so W_t . shift_t(x) = shift_t(W_t . x): the 3x3 kernel
// [Cout][3][3][Cin] IS the matrix [9 Cout][Cin] of a 1x1 convolution z = W x (row o * 9 + t: no copy, the master parameter's own bytes) and
//   y[n, h, w, o] = bias[o] + sum_t z[n, h + t / 3 - 1, w + t % 3 - 1, o * 9 + t]        (taps outside the image: zero, as the padding)
// The per-tap kernel read the 512-channel input nine times from L2 for 3 useful output channels (340 us forward, 308 us data gradient,
// 183 us weight gradient on FPN/ResNet50 1024 x 1024 batch 4); the 1x1 form reads it once per pass and everything else is 27 channels wide.
// Backward: dz[n, h, w, o * 9 + t] = dy[n, h - (t / 3 - 1), w - (t % 3 - 1), o], then the 1x1 convolution's ordinary gradients - its weight
// gradient is the 3x3 kernel's gradient in place.
template <typename T>
__global__ __launch_bounds__(256) void tapsum_fwd_kernel(const T* __restrict__ z, T* __restrict__ y, const float* __restrict__ bias, int N, int H, int W,
                                                         int Cout, int Zc, int Cy) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)N * H * W * Cout) return;
  const int o = (int)(idx % Cout);
  const long long p = idx / Cout;
  const int w = (int)(p % W), h = (int)((p / W) % H), n = (int)(p / ((long long)W * H));
  float acc = bias ? bias[o] : 0.f;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int hh = h + t / 3 - 1, ww = w + t % 3 - 1;
    if ((unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W) acc += Elem<T>::load(z + (((size_t)n * H + hh) * W + ww) * Zc + o * 9 + t);
  }
  Elem<T>::store(y + (size_t)p * Cy + o, acc);
}
template <typename T>
__global__ __launch_bounds__(256) void tapsum_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dz, int N, int H, int W, int Cout, int Cdy, int Cdz) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)N * H * W * Cdz) return;
  const int j = (int)(idx % Cdz);
  const long long p = idx / Cdz;
  const int w = (int)(p % W), h = (int)((p / W) % H), n = (int)(p / ((long long)W * H));
  float v = 0.f;
  if (j < 9 * Cout) {
    const int o = j / 9, t = j - o * 9;
    const int hh = h - (t / 3 - 1), ww = w - (t % 3 - 1);
    if ((unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W) v = Elem<T>::load(dy + (((size_t)n * H + hh) * W + ww) * Cdy + o);
  }
  Elem<T>::store(dz + (size_t)idx, v);                      // (the padded tap channels j >= 9 Cout: zero)
}

// z: [N, H, W, Zc] with the tap channel (o, t) at o * 9 + t (Zc >= 9 Cout); y: [N, H, W, Cy] (channels >= Cout untouched); bias: [Cout] or NULL
extern "C" int stp_tapsum_fwd(const void* z, void* y, const float* bias, int32_t N, int32_t H, int32_t W, int32_t Cout, int32_t Zc, int32_t Cy,
                              int32_t dtype, void* stream) {
  if (!stp_dtype_ok(dtype)) return STP_E_BADARG;      // (the other build's 16-bit code, or garbage)
  if (!z || !y || N <= 0 || H <= 0 || W <= 0 || Cout <= 0 || Zc < 9 * Cout || Cy < Cout) return STP_E_BADARG;
  const long long total = (long long)N * H * W * Cout;
  if (total > 0x7fffffffll * 128) return STP_E_BADARG;
  const dim3 grid((unsigned)((total + 255) / 256));
  if (dtype == STP_H16) hipLaunchKernelGGL(tapsum_fwd_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)z, (bf16_t*)y, bias, N, H, W, Cout, Zc, Cy);
  else if (dtype == STP_F32) hipLaunchKernelGGL(tapsum_fwd_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)z, (float*)y, bias, N, H, W, Cout, Zc, Cy);
  else return STP_E_BADARG;
  STP_LAUNCH_CHECK();
  return STP_OK;
}
// dy: [N, H, W, Cdy] (the first Cout channels are read); dz: [N, H, W, Cdz], written whole (Cdz >= 9 Cout)
extern "C" int stp_tapsum_bwd(const void* dy, void* dz, int32_t N, int32_t H, int32_t W, int32_t Cout, int32_t Cdy, int32_t Cdz, int32_t dtype,
                              void* stream) {
  if (!stp_dtype_ok(dtype)) return STP_E_BADARG;      // (the other build's 16-bit code, or garbage)
  if (!dy || !dz || N <= 0 || H <= 0 || W <= 0 || Cout <= 0 || Cdz < 9 * Cout || Cdy < Cout) return STP_E_BADARG;
  const long long total = (long long)N * H * W * Cdz;
  if (total > 0x7fffffffll * 128) return STP_E_BADARG;
  const dim3 grid((unsigned)((total + 255) / 256));
  if (dtype == STP_H16) hipLaunchKernelGGL(tapsum_bwd_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, (bf16_t*)dz, N, H, W, Cout, Cdy, Cdz);
  else if (dtype == STP_F32) hipLaunchKernelGGL(tapsum_bwd_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)dy, (float*)dz, N, H, W, Cout, Cdy, Cdz);
  else return STP_E_BADARG;
  STP_LAUNCH_CHECK();
  return STP_OK;
}

// ------------------------------------------------------------------------------------------
// MaxPooling2D(2, 2) without padding (keras.applications VGG blocks).  idx[n,ho,wo,c] = 2*dy+dx of the first maximum;
// every input pixel belongs to exactly one window, so the gradient is a masked copy.  H and W even.
template <typename T, int V>
__global__ __launch_bounds__(256) void maxpool2_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, uint8_t* __restrict__ idx,
                                                           int N, int H, int W, int C) {
  const int cg = C / V, Ho = H >> 1, Wo = W >> 1;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= Wo * cg) return;
  const int wo = t / cg, c = (t - wo * cg) * V;
  for (int rr = 0; rr < POOL_ROWS; ++rr) {
  const int orow = blockIdx.y * POOL_ROWS + rr;
  if (orow >= N * Ho) break;
  const int n = orow / Ho, ho = orow - n * Ho;
  float best[V];
  uint8_t bi[V];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float v[V];
    ldv<T, V>(x + (((size_t)n * H + 2 * ho + (k >> 1)) * W + 2 * wo + (k & 1)) * C + c, v);
#pragma unroll
    for (int e = 0; e < V; ++e)
      if (k == 0 || v[e] > best[e]) { best[e] = v[e]; bi[e] = (uint8_t)k; }
  }
  const size_t o = (((size_t)n * Ho + ho) * Wo + wo) * C + c;
  stv<T, V>(y + o, best);
  if (idx) {
    if constexpr (V == 8) *reinterpret_cast<uint2*>(idx + o) = *reinterpret_cast<const uint2*>(bi);
    else *reinterpret_cast<uint32_t*>(idx + o) = *reinterpret_cast<const uint32_t*>(bi);
  }
  }
}

template <typename T, int V>
__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const uint8_t* __restrict__ idx, const T* __restrict__ dy, T* __restrict__ dx,
                                                           int N, int H, int W, int C, int accumulate) {
  const int cg = C / V, Ho = H >> 1, Wo = W >> 1;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= W * cg) return;
  const int w = t / cg, c = (t - w * cg) * V;
  const int n = blockIdx.y / H, h = blockIdx.y - n * H;
  const size_t o = (((size_t)n * Ho + (h >> 1)) * Wo + (w >> 1)) * C + c;
  uint8_t id[V];
  if constexpr (V == 8) *reinterpret_cast<uint2*>(id) = *reinterpret_cast<const uint2*>(idx + o);
  else *reinterpret_cast<uint32_t*>(id) = *reinterpret_cast<const uint32_t*>(idx + o);
  float d[V], g[V];
  ldv<T, V>(dy + o, d);
  const uint8_t me = (uint8_t)(((h & 1) << 1) | (w & 1));
#pragma unroll
  for (int e = 0; e < V; ++e) g[e] = id[e] == me ? d[e] : 0.f;
  T* out = dx + (((size_t)n * H + h) * W + w) * C + c;
  if (accumulate) {
    float a[V];
    ldv<T, V>(out, a);
#pragma unroll
    for (int e = 0; e < V; ++e) g[e] += a[e];
  }
  stv<T, V>(out, g);
}

extern "C" int stp_maxpool2x2(const void* x, void* y, uint8_t* idx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t dtype,
                              void* stream) {
  if (!stp_dtype_ok(dtype)) return STP_E_BADARG;      // (the other build's 16-bit code, or garbage)
  if (!x || !y || (C & 3) || N <= 0 || (H & 1) || (W & 1) || (int64_t)N * (H >> 1) > 65535) return STP_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  const bool v8 = dtype == STP_H16 && (C & 7) == 0;
  const dim3 grid(ceil_div((W >> 1) * (C / (v8 ? 8 : 4)), 256), N * (H >> 1));
  if (v8) hipLaunchKernelGGL((maxpool2_fwd_kernel<bf16_t, 8>), grid, dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, idx, N, H, W, C);
  else if (dtype == STP_H16) hipLaunchKernelGGL((maxpool2_fwd_kernel<bf16_t, 4>), grid, dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, idx, N, H, W, C);
  else if (dtype == STP_F32) hipLaunchKernelGGL((maxpool2_fwd_kernel<float, 4>), grid, dim3(256), 0, s, (const float*)x, (float*)y, idx, N, H, W, C);
  else return STP_E_BADARG;
  STP_LAUNCH_CHECK();
  return STP_OK;
}

extern "C" int stp_maxpool2x2_bwd(const uint8_t* idx, const void* dy, void* dx, int32_t N, int32_t H, int32_t W, int32_t C,
                                  int32_t dtype, int32_t accumulate, void* stream) {
  if (!stp_dtype_ok(dtype)) return STP_E_BADARG;      // (the other build's 16-bit code, or garbage)
  if (!idx || !dy || !dx || (C & 3) || N <= 0 || (H & 1) || (W & 1) || (int64_t)N * H > 65535) return STP_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  const bool v8 = dtype == STP_H16 && (C & 7) == 0;
  const dim3 grid(ceil_div(W * (C / (v8 ? 8 : 4)), 256), N * H);
  if (v8) hipLaunchKernelGGL((maxpool2_bwd_kernel<bf16_t, 8>), grid, dim3(256), 0, s, idx, (const bf16_t*)dy, (bf16_t*)dx, N, H, W, C, accumulate);
  else if (dtype == STP_H16) hipLaunchKernelGGL((maxpool2_bwd_kernel<bf16_t, 4>), grid, dim3(256), 0, s, idx, (const bf16_t*)dy, (bf16_t*)dx, N, H, W, C, accumulate);
  else if (dtype == STP_F32) hipLaunchKernelGGL((maxpool2_bwd_kernel<float, 4>), grid, dim3(256), 0, s, idx, (const float*)dy, (float*)dx, N, H, W, C, accumulate);
  else return STP_E_BADARG;
  STP_LAUNCH_CHECK();
  return STP_OK;
}

// dY <- dY * [y > 0] in place: the gradient of a ReLU fused into a convolution epilogue (VGG: Conv2D(activation='relu'))
template <typename T>
__global__ __launch_bounds__(256) void relu_bwd_kernel(const T* __restrict__ y, T* __restrict__ dy, int64_t count4) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count4; i += (int64_t)gridDim.x * 256) {
    const f32x4 a = ld4<T>(y + i * 4);
    f32x4 g = ld4<T>(dy + i * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) g[e] = a[e] > 0.f ? g[e] : 0.f;
    store4(dy + i * 4, g);
  }
}

extern "C" int stp_relu_bwd(const void* y, void* dy, int64_t count, int32_t dtype, void* stream) {
  if (!stp_dtype_ok(dtype)) return STP_E_BADARG;      // (the other build's 16-bit code, or garbage)
  if (!y || !dy || count <= 0 || (count & 3)) return STP_E_BADARG;
  const int g = grid_for(count >> 2);
  if (dtype == STP_H16) hipLaunchKernelGGL(relu_bwd_kernel<bf16_t>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)y, (bf16_t*)dy, count >> 2);
  else if (dtype == STP_F32) hipLaunchKernelGGL(relu_bwd_kernel<float>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const float*)y, (float*)dy, count >> 2);
  else return STP_E_BADARG;
  STP_LAUNCH_CHECK();
  return STP_OK;
}

// ------------------------------------------------------------------------------------------
// gradient of UpSampling2D(2): dy is [N,2H,2W,ldy] (first C channels used), dx [N,H,W,C].
// BNB: dx is the gradient of a BatchNormalization(+activation) output that this launch completes - the value is masked with the
// activation re-derived from the BN input x and the workgroup writes its partial sums of g and g * xhat ([2][C][workgroups],
// the layout stp_bn_backward_fused reduces), exactly as the convolution epilogues do (stp_conv_params.bnb_x).
template <typename T, int V, bool BNB>
__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int N, int H, int W,
                                                             int C, int ldy, int accumulate, BnBack bnb, float* __restrict__ partial) {
  const int cg = C / V;
  const int t = blockIdx.x * 256 + threadIdx.x;
  const bool on = t < W * cg;
  if (!BNB && !on) return;
  const int w = on ? t / cg : 0, c = on ? (t - w * cg) * V : 0;
  constexpr int ROWS = BNB ? BNB_ROWS : 1;
  float sg[V], sq[V];
#pragma unroll
  for (int e = 0; e < V; ++e) { sg[e] = 0.f; sq[e] = 0.f; }
  for (int rr = 0; rr < ROWS; ++rr) {
    const int row = blockIdx.y * ROWS + rr;
    if (row >= N * H || !on) break;
    const int n = row / H, h = row - n * H;
    const size_t o = (((size_t)n * H + h) * W + w) * C + c;
    const T* b = dy + (((size_t)n * 2 * H + 2 * h) * 2 * W + 2 * w) * ldy + c;
    float g[V], a1[V], a2[V], a3[V];
    ldv<T, V>(b, g);
    ldv<T, V>(b + ldy, a1);
    ldv<T, V>(b + (size_t)2 * W * ldy, a2);
    ldv<T, V>(b + (size_t)2 * W * ldy + ldy, a3);
#pragma unroll
    for (int e = 0; e < V; ++e) g[e] = ((g[e] + a1[e]) + a2[e]) + a3[e];
    if (accumulate) {
      float p[V];
      ldv<T, V>(dx + o, p);
#pragma unroll
      for (int e = 0; e < V; ++e) g[e] += p[e];
    }
    if constexpr (BNB) bnb_mask_store<T, V>(g, o, c, bnb, dx, sg, sq);
    else stv<T, V>(dx + o, g);
  }
  if constexpr (BNB) bnb_reduce<V>(sg, sq, cg, C, partial);
}

static int upsample2x_bwd_launch(const void* dy, void* dx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t ldy, int32_t dtype,
                                 int32_t accumulate, const BnBack* bnb, float* partial, hipStream_t s) {
  if (!dy || !dx || (C & 3) || (ldy & 3) || ldy < C || (int64_t)N * H > 65535) return STP_E_BADARG;
  const bool v8 = dtype == STP_H16 && (C & 7) == 0 && (ldy & 7) == 0;
  const int V = v8 ? 8 : 4;
  const dim3 grid(ceil_div(W * (C / V), 256), N * H);
  BnBack none;
  none.x = nullptr; none.mean = none.rstd = none.gamma = none.beta = nullptr; none.relu = 0;
  if (bnb) {
    if (256 % (C / V) != 0 || !partial) return STP_E_BADARG;       // each workgroup must hold whole pixels of all its channel groups
    const size_t lds = 256 * 2 * V * sizeof(float);
    const dim3 gridb(grid.x, ceil_div(N * H, BNB_ROWS));
    if (v8) hipLaunchKernelGGL((upsample2x_bwd_kernel<bf16_t, 8, true>), gridb, dim3(256), lds, s, (const bf16_t*)dy, (bf16_t*)dx, N, H, W, C, ldy, accumulate, *bnb, partial);
    else if (dtype == STP_H16) hipLaunchKernelGGL((upsample2x_bwd_kernel<bf16_t, 4, true>), gridb, dim3(256), lds, s, (const bf16_t*)dy, (bf16_t*)dx, N, H, W, C, ldy, accumulate, *bnb, partial);
    else if (dtype == STP_F32) hipLaunchKernelGGL((upsample2x_bwd_kernel<float, 4, true>), gridb, dim3(256), lds, s, (const float*)dy, (float*)dx, N, H, W, C, ldy, accumulate, *bnb, partial);
    else return STP_E_BADARG;
  } else {
    if (v8) hipLaunchKernelGGL((upsample2x_bwd_kernel<bf16_t, 8, false>), grid, dim3(256), 0, s, (const bf16_t*)dy, (bf16_t*)dx, N, H, W, C, ldy, accumulate, none, nullptr);
    else if (dtype == STP_H16) hipLaunchKernelGGL((upsample2x_bwd_kernel<bf16_t, 4, false>), grid, dim3(256), 0, s, (const bf16_t*)dy, (bf16_t*)dx, N, H, W, C, ldy, accumulate, none, nullptr);
    else if (dtype == STP_F32) hipLaunchKernelGGL((upsample2x_bwd_kernel<float, 4, false>), grid, dim3(256), 0, s, (const float*)dy, (float*)dx, N, H, W, C, ldy, accumulate, none, nullptr);
    else return STP_E_BADARG;
  }
  STP_LAUNCH_CHECK();
  return STP_OK;
}

extern "C" int stp_upsample2x_bwd(const void* dy, void* dx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t ldy,
                                  int32_t dtype, int32_t accumulate, void* stream) {
  if (!stp_dtype_ok(dtype)) return STP_E_BADARG;      // (the other build's 16-bit code, or garbage)
  return upsample2x_bwd_launch(dy, dx, N, H, W, C, ldy, dtype, accumulate, nullptr, nullptr, (hipStream_t)stream);
}

// number of workgroups (= partial-sum tiles per channel) of stp_upsample2x_bwd_bn
extern "C" int stp_upsample2x_bwd_bn_tiles(int32_t N, int32_t H, int32_t W, int32_t C, int32_t ldy, int32_t dtype) {
  if (!stp_dtype_ok(dtype)) return STP_E_BADARG;      // (the other build's 16-bit code, or garbage)
  const bool v8 = dtype == STP_H16 && (C & 7) == 0 && (ldy & 7) == 0;
  const int V = v8 ? 8 : 4;
  if (C <= 0 || (C & 3) || 256 % (C / V) != 0) return 0;          // 0: not supported for this channel count
  return ceil_div(W * (C / V), 256) * ceil_div(N * H, BNB_ROWS);
}

extern "C" int stp_upsample2x_bwd_bn(const void* dy, void* dx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t ldy, int32_t dtype,
                                     int32_t accumulate, const void* bn_x, const float* mean, const float* rstd, const float* gamma,
                                     const float* beta, int32_t relu, float* partial, void* stream) {
  if (!stp_dtype_ok(dtype)) return STP_E_BADARG;      // (the other build's 16-bit code, or garbage)
  if (!bn_x || !mean || !rstd || !partial) return STP_E_BADARG;
  BnBack b;
  b.x = (const char*)bn_x; b.mean = mean; b.rstd = rstd; b.gamma = gamma; b.beta = beta; b.relu = relu;
  return upsample2x_bwd_launch(dy, dx, N, H, W, C, ldy, dtype, accumulate, &b, partial, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------
// AveragePooling2D(pool = strides = k) with H % k == W % k == 0 (PSPNet's pyramid pooling levels) and its gradient
// (every input pixel belongs to exactly one window: dx = dy / k^2).
template <typename T>
__global__ __launch_bounds__(256) void avgpool_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int C, int k) {
  const int Ho = H / k, Wo = W / k;
  const int64_t total = (int64_t)N * Ho * Wo * C;
  const float inv = 1.f / (float)(k * k);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    const int wo = (int)((i / C) % Wo);
    const int ho = (int)((i / ((int64_t)C * Wo)) % Ho);
    const int n = (int)(i / ((int64_t)C * Wo * Ho));
    const T* b = x + (((int64_t)n * H + (int64_t)ho * k) * W + (int64_t)wo * k) * C + c;
    float acc = 0.f;
    for (int dy = 0; dy < k; ++dy)
      for (int dx = 0; dx < k; ++dx) acc += Elem<T>::load(b + ((int64_t)dy * W + dx) * C);
    Elem<T>::store(y + i, acc * inv);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int N, int H, int W, int C, int k,
                                                          int accumulate) {
  const int Ho = H / k, Wo = W / k;
  const int64_t total = (int64_t)N * H * W * C;
  const float inv = 1.f / (float)(k * k);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    const int w = (int)((i / C) % W);
    const int h = (int)((i / ((int64_t)C * W)) % H);
    const int n = (int)(i / ((int64_t)C * W * H));
    float g = Elem<T>::load(dy + (((int64_t)n * Ho + h / k) * Wo + w / k) * C + c) * inv;
    if (accumulate) g += Elem<T>::load(dx + i);
    Elem<T>::store(dx + i, g);
  }
}

// Large windows (PSPNet: 96x96 .. 16x16 pixels per output): one workgroup per (window, channel chunk).  VL vector lanes x
// PL pixel lanes stride over the window with 16-byte loads; the pixel lanes are combined through LDS in a fixed order.
template <typename T, int V>
__global__ __launch_bounds__(256) void avgpool_win_kernel(const T* __restrict__ x, T* __restrict__ y, int H, int W, int C, int k, int VL,
                                                          int rps, float* __restrict__ part) {
  extern __shared__ float red[];  // [PL][VL*V]
  const int PL = 256 / VL;
  const int vl = threadIdx.x % VL, pl = threadIdx.x / VL;
  const int Ho = H / k, Wo = W / k;
  int b = blockIdx.x;
  const int wo = b % Wo; b /= Wo;
  const int ho = b % Ho;
  const int n = b / Ho;
  const int c = (blockIdx.y * VL + vl) * V;
  float acc[V];
#pragma unroll
  for (int e = 0; e < V; ++e) acc[e] = 0.f;
  if (c < C && pl < PL) {
    const T* base = x + (((int64_t)n * H + (int64_t)ho * k) * W + (int64_t)wo * k) * C + c;
    const int r0 = blockIdx.z * rps, r1 = min(k, r0 + rps);      // window rows of this split (all of them without a workspace)
    for (int p = r0 * k + pl; p < r1 * k; p += PL) {
      const int dy = p / k, dx = p - dy * k;
      float v[V];
      ldv<T, V>(base + ((int64_t)dy * W + dx) * C, v);
#pragma unroll
      for (int e = 0; e < V; ++e) acc[e] += v[e];
    }
  }
  if (pl < PL) {
#pragma unroll
    for (int e = 0; e < V; ++e) red[(pl * VL + vl) * V + e] = acc[e];
  }
  __syncthreads();
  const int t = threadIdx.x, cc = blockIdx.y * VL * V + t;
  if (t < VL * V && cc < C) {
    float sum = 0.f;
    for (int l = 0; l < PL; ++l) sum += red[l * VL * V + t];
    if (part) part[((int64_t)blockIdx.z * gridDim.x + blockIdx.x) * C + cc] = sum;
    else Elem<T>::store(y + (((int64_t)n * Ho + ho) * Wo + wo) * C + cc, sum / (float)(k * k));
  }
}

// second stage of a split reduction: dst[i] (+)= scale * sum_s part[s][i], s in ascending order
template <typename T>
__global__ __launch_bounds__(256) void split_combine_kernel(const float* __restrict__ part, int S, int64_t n, float scale, T* __restrict__ dst,
                                                            int accumulate) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float sum = 0.f;
  for (int q = 0; q < S; ++q) sum += part[(int64_t)q * n + i];
  sum *= scale;
  if (accumulate) sum += Elem<T>::load(dst + i);
  Elem<T>::store(dst + i, sum);
}

// rows-per-split of a reduction over `rows` x `cols` taps done by `blocks` workgroups: split until >= 1024 workgroups exist
static int split_rows(int64_t blocks, int rows, int cols) {
  if (blocks >= 1024 || (int64_t)rows * cols < 1024) return rows;
  const int want = (int)ceil_div((int64_t)1024, blocks);
  const int rps = (int)ceil_div(rows, want < rows ? want : rows);
  return rps < 1 ? 1 : rps;
}

template <typename T, int V>
__global__ __launch_bounds__(256) void avgpool_bwd_vec_kernel(const T* __restrict__ dy, T* __restrict__ dx, int H, int W, int C, int k,
                                                              int accumulate) {
  const int cg = C / V;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= W * cg) return;
  const int w = t / cg, c = (t - w * cg) * V;
  const int n = blockIdx.y / H, h = blockIdx.y - n * H;
  const float inv = 1.f / (float)(k * k);
  float g[V], o[V];
  ldv<T, V>(dy + (((int64_t)n * (H / k) + h / k) * (W / k) + w / k) * C + c, g);
  T* d = dx + (((int64_t)n * H + h) * W + w) * C + c;
#pragma unroll
  for (int e = 0; e < V; ++e) o[e] = g[e] * inv;
  if (accumulate) {
    ldv<T, V>(d, g);
#pragma unroll
    for (int e = 0; e < V; ++e) o[e] += g[e];
  }
  stv<T, V>(d, o);
}

// vector width shared by the pooling / resize kernels: 16-byte (bf16 x8, fp32 x4) or 8-byte (bf16 x4) channel groups
static int vec_for(int dtype, int a, int b, int c) {
  const int m = a | b | c;
  if (dtype == STP_H16 && !(m & 7)) return 8;
  return (m & 3) ? 1 : 4;
}

static int pool_vl(int cg) { return cg < 8 ? cg : 8; }   // 8 vector lanes = 128 contiguous bytes per pixel and workgroup

extern "C" size_t stp_avgpool_workspace_bytes(int32_t N, int32_t H, int32_t W, int32_t C, int32_t k) {
  if (N <= 0 || C <= 0 || k < 1 || H % k || W % k) return 0;
  const int64_t nwin = (int64_t)N * (H / k) * (W / k);
  size_t need = 0;
  for (int V = 4; V <= 8; V += 4) {                      // either vector width the launch may pick
    const int cg = ceil_div(C, V);
    const int rps = split_rows(nwin * ceil_div(cg, pool_vl(cg)), k, k);
    if (rps < k) need = std::max(need, (size_t)ceil_div(k, rps) * nwin * C * sizeof(float));
  }
  return need;
}

extern "C" int stp_avgpool(const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t k, int32_t dtype, void* workspace,
                           size_t workspace_bytes, void* stream) {
  if (!stp_dtype_ok(dtype)) return STP_E_BADARG;      // (the other build's 16-bit code, or garbage)
  if (!x || !y || N <= 0 || C <= 0 || k < 1 || H % k || W % k) return STP_E_BADARG;
  if (dtype != STP_H16 && dtype != STP_F32) return STP_E_BADARG;
  const int V = vec_for(dtype, C, 0, 0);
  if (V > 1 && k * k >= 32) {
    const int cg = C / V, VL = pool_vl(cg);
    const int64_t nwin = (int64_t)N * (H / k) * (W / k);
    int rps = split_rows(nwin * ceil_div(cg, VL), k, k);
    int S = ceil_div(k, rps);
    if (S > 1 && (!workspace || workspace_bytes < (size_t)S * nwin * C * sizeof(float))) { rps = k; S = 1; }   // no workspace: one pass
    float* part = S > 1 ? (float*)workspace : nullptr;
    const dim3 grid((unsigned)nwin, ceil_div(cg, VL), S);
    const size_t lds = (size_t)(256 / VL) * VL * V * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
    if (V == 8) hipLaunchKernelGGL((avgpool_win_kernel<bf16_t, 8>), grid, dim3(256), lds, s, (const bf16_t*)x, (bf16_t*)y, H, W, C, k, VL, rps, part);
    else if (dtype == STP_H16) hipLaunchKernelGGL((avgpool_win_kernel<bf16_t, 4>), grid, dim3(256), lds, s, (const bf16_t*)x, (bf16_t*)y, H, W, C, k, VL, rps, part);
    else hipLaunchKernelGGL((avgpool_win_kernel<float, 4>), grid, dim3(256), lds, s, (const float*)x, (float*)y, H, W, C, k, VL, rps, part);
    STP_LAUNCH_CHECK();
    if (S > 1) {
      const int64_t ne = nwin * C;
      const float sc = 1.f / (float)(k * k);
      if (dtype == STP_H16) hipLaunchKernelGGL(split_combine_kernel<bf16_t>, dim3((unsigned)ceil_div(ne, (int64_t)256)), dim3(256), 0, s, part, S, ne, sc, (bf16_t*)y, 0);
      else hipLaunchKernelGGL(split_combine_kernel<float>, dim3((unsigned)ceil_div(ne, (int64_t)256)), dim3(256), 0, s, part, S, ne, sc, (float*)y, 0);
      STP_LAUNCH_CHECK();
    }
    return STP_OK;
  }
  const int g = grid_for((int64_t)N * (H / k) * (W / k) * C);
  if (dtype == STP_H16) hipLaunchKernelGGL(avgpool_kernel<bf16_t>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, N, H, W, C, k);
  else if (dtype == STP_F32) hipLaunchKernelGGL(avgpool_kernel<float>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)y, N, H, W, C, k);
  else return STP_E_BADARG;
  STP_LAUNCH_CHECK();
  return STP_OK;
}

extern "C" int stp_avgpool_bwd(const void* dy, void* dx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t k, int32_t dtype,
                               int32_t accumulate, void* stream) {
  if (!stp_dtype_ok(dtype)) return STP_E_BADARG;      // (the other build's 16-bit code, or garbage)
  if (!dy || !dx || N <= 0 || C <= 0 || k < 1 || H % k || W % k) return STP_E_BADARG;
  if (dtype != STP_H16 && dtype != STP_F32) return STP_E_BADARG;
  const int V = vec_for(dtype, C, 0, 0);
  if (V > 1 && (int64_t)N * H <= 65535) {
    const dim3 grid(ceil_div(W * (C / V), 256), N * H);
    hipStream_t s = (hipStream_t)stream;
    if (V == 8) hipLaunchKernelGGL((avgpool_bwd_vec_kernel<bf16_t, 8>), grid, dim3(256), 0, s, (const bf16_t*)dy, (bf16_t*)dx, H, W, C, k, accumulate);
    else if (dtype == STP_H16) hipLaunchKernelGGL((avgpool_bwd_vec_kernel<bf16_t, 4>), grid, dim3(256), 0, s, (const bf16_t*)dy, (bf16_t*)dx, H, W, C, k, accumulate);
    else hipLaunchKernelGGL((avgpool_bwd_vec_kernel<float, 4>), grid, dim3(256), 0, s, (const float*)dy, (float*)dx, H, W, C, k, accumulate);
    STP_LAUNCH_CHECK();
    return STP_OK;
  }
  const int g = grid_for((int64_t)N * H * W * C);
  if (dtype == STP_H16) hipLaunchKernelGGL(avgpool_bwd_kernel<bf16_t>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, (bf16_t*)dx, N, H, W, C, k, accumulate);
  else if (dtype == STP_F32) hipLaunchKernelGGL(avgpool_bwd_kernel<float>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const float*)dy, (float*)dx, N, H, W, C, k, accumulate);
  else return STP_E_BADARG;
  STP_LAUNCH_CHECK();
  return STP_OK;
}

// ------------------------------------------------------------------------------------------
// FPN pieces.  (a) x[n,2h+i,2w+j,c] += m[n,h,w,c]: Add()([lateral, UpSampling2D(2)(m)]) in place on the lateral tensor.
template <typename T, int V>
__global__ __launch_bounds__(256) void upsample2x_add_kernel(T* __restrict__ x, const T* __restrict__ m, int N, int H, int W, int C) {
  const int cg = C / V;                       // H, W: size of x (even)
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= W * cg) return;
  const int w = t / cg, c = (t - w * cg) * V;
  const int n = blockIdx.y / H, h = blockIdx.y - n * H;
  float a[V], b[V];
  T* px = x + (((size_t)n * H + h) * W + w) * C + c;
  ldv<T, V>(px, a);
  ldv<T, V>(m + (((size_t)n * (H >> 1) + (h >> 1)) * (W >> 1) + (w >> 1)) * C + c, b);
#pragma unroll
  for (int e = 0; e < V; ++e) a[e] += b[e];
  stv<T, V>(px, a);
}

extern "C" int stp_upsample2x_add(void* x, const void* m, int32_t N, int32_t H, int32_t W, int32_t C, int32_t dtype, void* stream) {
  if (!stp_dtype_ok(dtype)) return STP_E_BADARG;      // (the other build's 16-bit code, or garbage)
  if (!x || !m || (C & 3) || N <= 0 || (H & 1) || (W & 1) || (int64_t)N * H > 65535) return STP_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  const bool v8 = dtype == STP_H16 && (C & 7) == 0;
  const dim3 grid(ceil_div(W * (C / (v8 ? 8 : 4)), 256), N * H);
  if (v8) hipLaunchKernelGGL((upsample2x_add_kernel<bf16_t, 8>), grid, dim3(256), 0, s, (bf16_t*)x, (const bf16_t*)m, N, H, W, C);
  else if (dtype == STP_H16) hipLaunchKernelGGL((upsample2x_add_kernel<bf16_t, 4>), grid, dim3(256), 0, s, (bf16_t*)x, (const bf16_t*)m, N, H, W, C);
  else if (dtype == STP_F32) hipLaunchKernelGGL((upsample2x_add_kernel<float, 4>), grid, dim3(256), 0, s, (float*)x, (const float*)m, N, H, W, C);
  else return STP_E_BADARG;
  STP_LAUNCH_CHECK();
  return STP_OK;
}

// (b) tf.image.resize_bilinear(align_corners=False) of TF 1.x (what Keras 2.2.4 K.resize_images(interpolation='bilinear')
// calls) by an integer factor f: src = dst / f, x0 = floor(src), x1 = min(x0 + 1, in - 1), no half-pixel offset.
// The output may be a channel slice of a wider tensor (ldo / coff: Concatenate of the resized pyramid levels); f = 1 copies.
template <typename T>
__global__ __launch_bounds__(256) void resize_bilinear_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int C,
                                                              int f, int ldo, int coff) {
  const int Ho = H * f, Wo = W * f;
  const int64_t total = (int64_t)N * Ho * Wo * C;
  const float inv = 1.f / (float)f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    const int xo = (int)((i / C) % Wo);
    const int yo = (int)((i / ((int64_t)C * Wo)) % Ho);
    const int n = (int)(i / ((int64_t)C * Wo * Ho));
    const int x0 = xo / f, y0 = yo / f;
    const float fx = (float)(xo - x0 * f) * inv, fy = (float)(yo - y0 * f) * inv;
    const int x1 = min(x0 + 1, W - 1), y1 = min(y0 + 1, H - 1);
    const T* b = x + (int64_t)n * H * W * C + c;
    const float v00 = Elem<T>::load(b + ((int64_t)y0 * W + x0) * C), v01 = Elem<T>::load(b + ((int64_t)y0 * W + x1) * C);
    const float v10 = Elem<T>::load(b + ((int64_t)y1 * W + x0) * C), v11 = Elem<T>::load(b + ((int64_t)y1 * W + x1) * C);
    const float top = v00 + (v01 - v00) * fx, bot = v10 + (v11 - v10) * fx;      // TF's lerp order
    Elem<T>::store(y + (((int64_t)n * Ho + yo) * Wo + xo) * ldo + coff + c, top + (bot - top) * fy);
  }
}

// gradient: dx[n,h,w,c] (+)= sum over the outputs that read this input pixel, in a fixed order (deterministic gather)
template <typename T>
__global__ __launch_bounds__(256) void resize_bilinear_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int N, int H, int W,
                                                                  int C, int f, int ldo, int coff, int accumulate) {
  const int Ho = H * f, Wo = W * f;
  const int64_t total = (int64_t)N * H * W * C;
  const float inv = 1.f / (float)f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    const int w = (int)((i / C) % W);
    const int h = (int)((i / ((int64_t)C * W)) % H);
    const int n = (int)(i / ((int64_t)C * W * H));
    const T* b = dy + (int64_t)n * Ho * Wo * ldo + coff + c;
    float acc = 0.f;
    // rows yo with y0 == h contribute (1 - fy) [all of it when h is the last row: y1 clamps to h]; rows with y0 == h - 1 contribute fy
    for (int yo = max((h - 1) * f, 0); yo < (h + 1) * f; ++yo) {
      const int y0 = yo / f;
      const float fy = (float)(yo - y0 * f) * inv;
      float wy = 0.f;
      if (y0 == h) wy += 1.f - fy;
      if (min(y0 + 1, H - 1) == h) wy += fy;
      if (wy == 0.f) continue;
      for (int xo = max((w - 1) * f, 0); xo < (w + 1) * f; ++xo) {
        const int x0 = xo / f;
        const float fx = (float)(xo - x0 * f) * inv;
        float wx = 0.f;
        if (x0 == w) wx += 1.f - fx;
        if (min(x0 + 1, W - 1) == w) wx += fx;
        if (wx != 0.f) acc += wy * wx * Elem<T>::load(b + ((int64_t)yo * Wo + xo) * ldo);
      }
    }
    if (accumulate) acc += Elem<T>::load(dx + i);
    Elem<T>::store(dx + i, acc);
  }
}

// 16-byte channel groups: one thread per (output pixel, channel group), the same lerp order as the scalar kernel
template <typename T, int V>
__global__ __launch_bounds__(256) void resize_bilinear_vec_kernel(const T* __restrict__ x, T* __restrict__ y, int H, int W, int C, int f,
                                                                  int ldo, int coff) {
  const int cg = C / V, Ho = H * f, Wo = W * f;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= Wo * cg) return;
  const int xo = t / cg, c = (t - xo * cg) * V;
  const int n = blockIdx.y / Ho, yo = blockIdx.y - n * Ho;
  const float inv = 1.f / (float)f;
  const int x0 = xo / f, y0 = yo / f;
  const float fx = (float)(xo - x0 * f) * inv, fy = (float)(yo - y0 * f) * inv;
  const int x1 = min(x0 + 1, W - 1), y1 = min(y0 + 1, H - 1);
  const T* b = x + (int64_t)n * H * W * C + c;
  float v00[V], v01[V], v10[V], v11[V], o[V];
  ldv<T, V>(b + ((int64_t)y0 * W + x0) * C, v00);
  ldv<T, V>(b + ((int64_t)y0 * W + x1) * C, v01);
  ldv<T, V>(b + ((int64_t)y1 * W + x0) * C, v10);
  ldv<T, V>(b + ((int64_t)y1 * W + x1) * C, v11);
#pragma unroll
  for (int e = 0; e < V; ++e) {
    const float top = v00[e] + (v01[e] - v00[e]) * fx, bot = v10[e] + (v11[e] - v10[e]) * fx;
    o[e] = top + (bot - top) * fy;
  }
  stv<T, V>(y + (((int64_t)n * Ho + yo) * Wo + xo) * ldo + coff + c, o);
}

// gradient, ROW-PAIR TILE form for wide channel slices (round 6: FPN's pyramid levels resized x2 / x4 / x8 into the 512-channel
// concatenation - 67 MB of dY per level; the gather kernel below re-reads every output pixel four times from L2 with a division per tap:
// 84 us per level).  One workgroup = (image, two input rows, WS input columns, CC channels):
//   phase 1 (vertical): the <= 3 f - 1 output rows that touch the two input rows are streamed ONCE; a thread owns <= K (column, 16-byte
//            channel group) items of the tile's <= (WS + 1) f - 1 output columns and keeps two fp32 sums per element (one per input row,
//            row weights are workgroup-uniform) - K independent 16-byte loads per row, no barrier in the loop;
//   phase 2 (horizontal): the sums go to LDS once, then every (input row, input column, 4 channels) item adds its <= 2 f - 1 columns
//            with the column weights, in a fixed order - deterministic.
// Every dY byte is read (3 f - 1) / (2 f) x ((WS + 1) f - 1) / (WS f) ~ 1.6 times (from L2 where neighbouring workgroups overlap).
#define RBT_K 3
template <typename T>
__global__ __launch_bounds__(256) void resize_bilinear_bwd_tile_kernel(const T* __restrict__ dy, T* __restrict__ dx, int H, int W, int C, int f,
                                                                       int ldo, int coff, int accumulate, int WS, int CC) {
  extern __shared__ __attribute__((aligned(16))) float rbt_smem[];      // [2][cols][CC]
  const int HP = (H + 1) >> 1;
  const int n = blockIdx.y / HP, h0 = (blockIdx.y - n * HP) * 2, h1 = h0 + 1;
  const int w0 = blockIdx.x * WS, w1 = min(w0 + WS, W);
  const int c0 = blockIdx.z * CC;
  const int Ho = H * f, Wo = W * f;
  const int xlo = max((w0 - 1) * f + 1, 0), xhi = min(w1 * f, Wo), cols = xhi - xlo;
  const int ylo = max((h0 - 1) * f + 1, 0), yhi = min((h1 + 1) * f, Ho);
  const int cgc = CC >> 3, items = cols * cgc;
  const float inv = 1.f / (float)f;
  float a0[RBT_K][8], a1[RBT_K][8];
  int64_t off[RBT_K];
#pragma unroll
  for (int k = 0; k < RBT_K; ++k) {
#pragma unroll
    for (int e = 0; e < 8; ++e) a0[k][e] = a1[k][e] = 0.f;
    const int it = threadIdx.x + k * 256;
    const int col = it / cgc, cv = it - col * cgc;
    off[k] = it < items ? (int64_t)(xlo + col) * ldo + coff + c0 + cv * 8 : -1;
  }
  const T* base = dy + (int64_t)n * Ho * Wo * ldo;
#pragma unroll 2
  for (int yo = ylo; yo < yhi; ++yo) {
    const int y0 = yo / f, y1 = min(y0 + 1, H - 1);
    const float fy = (float)(yo - y0 * f) * inv;
    const float wa = (y0 == h0 ? 1.f - fy : 0.f) + (y1 == h0 ? fy : 0.f), wb = (y0 == h1 ? 1.f - fy : 0.f) + (y1 == h1 ? fy : 0.f);
    const T* row = base + (int64_t)yo * Wo * ldo;
    u32x4 v[RBT_K];
#pragma unroll
    for (int k = 0; k < RBT_K; ++k)
      if (off[k] >= 0) v[k] = *reinterpret_cast<const u32x4*>(row + off[k]);
#pragma unroll
    for (int k = 0; k < RBT_K; ++k) {
      if (off[k] < 0) continue;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float lo = h16lo_to_f32(v[k][e]), hi = h16hi_to_f32(v[k][e]);
        a0[k][2 * e] += wa * lo; a0[k][2 * e + 1] += wa * hi;
        a1[k][2 * e] += wb * lo; a1[k][2 * e + 1] += wb * hi;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < RBT_K; ++k) {
    const int it = threadIdx.x + k * 256;
    if (it >= items) continue;
    float* d0 = rbt_smem + (size_t)it * 8, *d1 = d0 + (size_t)cols * CC;      // item order = [col][channel group]: the [cols][CC] image
    *reinterpret_cast<f32x4*>(d0) = f32x4{a0[k][0], a0[k][1], a0[k][2], a0[k][3]};
    *reinterpret_cast<f32x4*>(d0 + 4) = f32x4{a0[k][4], a0[k][5], a0[k][6], a0[k][7]};
    *reinterpret_cast<f32x4*>(d1) = f32x4{a1[k][0], a1[k][1], a1[k][2], a1[k][3]};
    *reinterpret_cast<f32x4*>(d1 + 4) = f32x4{a1[k][4], a1[k][5], a1[k][6], a1[k][7]};
  }
  __syncthreads();
  const int c4n = CC >> 2, per_row = (w1 - w0) * c4n;
  for (int idx = threadIdx.x; idx < 2 * per_row; idx += 256) {
    const int r = idx >= per_row ? 1 : 0, rem = idx - r * per_row;
    const int wl = rem / c4n, c4 = rem - wl * c4n, w = w0 + wl;
    const int h = r ? h1 : h0;
    if (h >= H) continue;
    const float* src = rbt_smem + (size_t)r * cols * CC + c4 * 4;
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    const int x_lo = max((w - 1) * f + 1, 0), x_hi = min((w + 1) * f, Wo);
    for (int xo = x_lo; xo < x_hi; ++xo) {
      const int x0 = xo / f, x1 = min(x0 + 1, W - 1);
      const float fx = (float)(xo - x0 * f) * inv;
      const float wx = (x0 == w ? 1.f - fx : 0.f) + (x1 == w ? fx : 0.f);
      const f32x4 t = *reinterpret_cast<const f32x4*>(src + (size_t)(xo - xlo) * CC);
      sum.x += wx * t.x; sum.y += wx * t.y; sum.z += wx * t.z; sum.w += wx * t.w;
    }
    T* d = dx + (((int64_t)n * H + h) * W + w) * C + c0 + c4 * 4;
    if (accumulate) {
      const f32x4 o = load4(d);
      sum.x += o.x; sum.y += o.y; sum.z += o.z; sum.w += o.w;
    }
    store4(d, sum);
  }
}

// gradient, one workgroup per (input pixel, channel chunk): the (2f)^2 outputs that may read the pixel are spread over TL tap
// lanes (VL vector lanes each), tap-lane partials are combined through LDS in a fixed order - deterministic, and parallel
// even when a whole 96x96 map is the gradient of a single pooled pixel (PSPNet level 1)
template <typename T, int V>
__global__ __launch_bounds__(256) void resize_bilinear_bwd_blk_kernel(const T* __restrict__ dy, T* __restrict__ dx, int H, int W, int C,
                                                                      int f, int ldo, int coff, int accumulate, int VL, int rps,
                                                                      float* __restrict__ part, int TLp, int PPB, int npix) {
  // 256 threads = PPB input pixels x TLp tap lanes x VL vector lanes (few taps and few channels - the x4 resize of 3-class
  // logits - would leave most of a one-pixel workgroup idle)
  extern __shared__ float red[];  // [PPB][TLp][VL*V]
  const int vl = threadIdx.x % VL, q = threadIdx.x / VL;
  const int pl = q / TLp, tl = q - pl * TLp;
  const int Ho = H * f, Wo = W * f;
  const int pix = blockIdx.x * PPB + pl;
  const bool pon = pl < PPB && pix < npix;
  int b = pon ? pix : 0;
  const int w = b % W; b /= W;
  const int h = b % H;
  const int n = b / H;
  const int c = (blockIdx.y * VL + vl) * V;
  const float inv = 1.f / (float)f;
  float acc[V];
#pragma unroll
  for (int e = 0; e < V; ++e) acc[e] = 0.f;
  if (c < C && pon) {
    const T* src = dy + (int64_t)n * Ho * Wo * ldo + coff + c;
    const int span = 2 * f;
    const int r0 = blockIdx.z * rps, r1 = min(span, r0 + rps);   // tap rows of this split
    for (int t = r0 * span + tl; t < r1 * span; t += TLp) {
      const int ty = t / span, tx = t - ty * span;
      const int yo = (h - 1) * f + ty, xo = (w - 1) * f + tx;
      if (yo < 0 || xo < 0) continue;
      const int y0 = yo / f, x0 = xo / f;
      const float fy = (float)(yo - y0 * f) * inv, fx = (float)(xo - x0 * f) * inv;
      float wy = 0.f, wx = 0.f;
      if (y0 == h) wy += 1.f - fy;
      if (min(y0 + 1, H - 1) == h) wy += fy;
      if (x0 == w) wx += 1.f - fx;
      if (min(x0 + 1, W - 1) == w) wx += fx;
      const float wgt = wy * wx;
      if (wgt == 0.f) continue;
      float v[V];
      ldv<T, V>(src + ((int64_t)yo * Wo + xo) * ldo, v);
#pragma unroll
      for (int e = 0; e < V; ++e) acc[e] += wgt * v[e];
    }
  }
  if (pl < PPB) {
#pragma unroll
    for (int e = 0; e < V; ++e) red[((pl * TLp + tl) * VL + vl) * V + e] = acc[e];
  }
  __syncthreads();
  const int CW = VL * V;                               // channels of this workgroup's chunk
  for (int t = threadIdx.x; t < PPB * CW; t += 256) {
    const int p2 = t / CW, ch = t - p2 * CW;
    const int cc = blockIdx.y * CW + ch, px = blockIdx.x * PPB + p2;
    if (cc >= C || px >= npix) continue;
    float sum = 0.f;
    for (int l = 0; l < TLp; ++l) sum += red[(p2 * TLp + l) * CW + ch];
    if (part) {
      part[((int64_t)blockIdx.z * gridDim.x + blockIdx.x) * C + cc] = sum;      // (splits: PPB == 1)
    } else {
      T* d = dx + (int64_t)px * C + cc;
      if (accumulate) sum += Elem<T>::load(d);
      Elem<T>::store(d, sum);
    }
  }
}

// gradient, ROW-STREAMING form for the resize of CLASS LOGITS (few channels, x4 / x8: PSPNet's final_interpolation, FPN's last_upsample -
// round 5): one workgroup per input row (n, h).  The <= 2f output rows that touch it are streamed whole through a double-buffered LDS
// row (coalesced 16-byte loads; the row of the next iteration sits in registers while this one is reduced); a thread owns up to two
// (w, 16-byte channel group) items and takes the 2f taps of its item from the staged row in a fixed order - deterministic.  The gather
// kernel below gives such shapes ONE input pixel per workgroup (256 tap lanes, an LDS reduction per 48 output bytes): 224 us for PSPNet's
// x8 gradient (226 MB), 128 us for FPN's x4.
template <typename T>
__global__ __launch_bounds__(256) void resize_bilinear_bwd_rows_kernel(const T* __restrict__ dy, T* __restrict__ dx, int H, int W, int C, int f,
                                                                       int ldo, int coff, int accumulate, int nvec) {
  extern __shared__ __attribute__((aligned(16))) char rb_smem[];
  constexpr int EV = 16 / (int)sizeof(T);
  const int n = blockIdx.x / H, h = blockIdx.x - n * H;
  const int Ho = H * f, Wo = W * f;
  const int rowV = Wo * ldo / EV, rowB = rowV * 16;                   // vectors / bytes of one output row (host: ldo % EV == 0)
  const int yo0 = max((h - 1) * f, 0), yo1 = min((h + 1) * f, Ho);    // output rows that may read input row h
  const float inv = 1.f / (float)f;
  const int cgv = C / EV, nitem = W * cgv;
  float acc[2][EV];
  int iw[2], ic[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
#pragma unroll
    for (int e = 0; e < EV; ++e) acc[k][e] = 0.f;
    const int it = threadIdx.x + k * 256;
    iw[k] = it < nitem ? it / cgv : -1;
    ic[k] = it < nitem ? (it - (it / cgv) * cgv) * EV : 0;
  }
  constexpr int MAXV = 10;                                            // 16-byte vectors of a row per thread (host: nvec <= MAXV)
  u32x4 pre[MAXV];
  auto fetch = [&](int yo) {
    const char* src = reinterpret_cast<const char*>(dy + ((int64_t)(n * Ho + yo) * Wo) * ldo);
#pragma unroll
    for (int v = 0; v < MAXV; ++v) {
      const int o = threadIdx.x + v * 256;
      if (v < nvec && o < rowV) pre[v] = *reinterpret_cast<const u32x4*>(src + (size_t)o * 16);
    }
  };
  auto park = [&](int buf) {
#pragma unroll
    for (int v = 0; v < MAXV; ++v) {
      const int o = threadIdx.x + v * 256;
      if (v < nvec && o < rowV) *reinterpret_cast<u32x4*>(rb_smem + buf * rowB + o * 16) = pre[v];
    }
  };
  fetch(yo0);
  int buf = 0;
  for (int yo = yo0; yo < yo1; ++yo) {
    park(buf);
    __syncthreads();                        // row yo is staged; the buffer of row yo - 1 is free
    if (yo + 1 < yo1) fetch(yo + 1);
    const int y0 = yo / f;
    const float fy = (float)(yo - y0 * f) * inv;
    float wy = 0.f;
    if (y0 == h) wy += 1.f - fy;
    if (min(y0 + 1, H - 1) == h) wy += fy;
    if (wy != 0.f) {
      const T* row = reinterpret_cast<const T*>(rb_smem + buf * rowB) + coff;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int w = iw[k];
        if (w < 0) continue;
        float r[EV];
#pragma unroll
        for (int e = 0; e < EV; ++e) r[e] = 0.f;
        for (int tx = 0; tx < 2 * f; ++tx) {
          const int xo = (w - 1) * f + tx;
          if (xo < 0) continue;
          const int x0 = xo / f;
          const float fx = (float)(xo - x0 * f) * inv;
          float wx = 0.f;
          if (x0 == w) wx += 1.f - fx;
          if (min(x0 + 1, W - 1) == w) wx += fx;
          float v[EV];
          ldv<T, EV>(row + xo * ldo + ic[k], v);
#pragma unroll
          for (int e = 0; e < EV; ++e) r[e] += wx * v[e];
        }
#pragma unroll
        for (int e = 0; e < EV; ++e) acc[k][e] += wy * r[e];
      }
    }
    buf ^= 1;
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    if (iw[k] < 0) continue;
    T* d = dx + ((int64_t)(n * H + h) * W + iw[k]) * C + ic[k];
    if (accumulate) {
      float a[EV];
      ldv<T, EV>(d, a);
#pragma unroll
      for (int e = 0; e < EV; ++e) acc[k][e] += a[e];
    }
    stv<T, EV>(d, acc[k]);
  }
}

// gradient, TINY input maps resized by a large factor (PSPNet's pyramid levels: 1 x 1 ... 6 x 6 maps of 512 channels, factors 96 ... 16,
// written into a slice of the concatenation - round 5): the reduction is separable.  Pass 1, one workgroup per OUTPUT row (n, yo): a
// thread owns one 16-byte channel group and every G-th pixel of the row (a wave reads one pixel's whole channel slice: 1 KB contiguous),
// keeps W <= 8 column accumulators, the G pixel groups meet through LDS in a fixed order -> part[n][yo][w][c] (fp32).  Pass 2: input
// row h sums its <= 2f rows of `part` with the row weights.  Every byte of dY is read once, coalesced; the gather kernel below took 51-80 us
// per level for 75 MB (one scattered 16-byte load per tap, an LDS reduction per input pixel).
#define RBT_MAXW 8
template <typename T>
__global__ __launch_bounds__(256) void resize_bilinear_bwd_tiny_rows_kernel(const T* __restrict__ dy, float* __restrict__ part, int H, int W, int C, int f,
                                                                            int ldo, int coff) {
  extern __shared__ float rbt_red[];                    // [G - 1][W][C]
  constexpr int EV = 16 / (int)sizeof(T);
  const int cgv = C / EV, G = 256 / cgv;                // pixel groups of the workgroup (host: 256 % cgv == 0)
  const int cv = threadIdx.x % cgv, g = threadIdx.x / cgv;
  const int Wo = W * f;
  const int64_t row = blockIdx.x;                       // n * Ho + yo
  const T* src = dy + row * Wo * ldo + coff + cv * EV;
  const float inv = 1.f / (float)f;
  float acc[RBT_MAXW][EV];
#pragma unroll
  for (int w = 0; w < RBT_MAXW; ++w)
#pragma unroll
    for (int e = 0; e < EV; ++e) acc[w][e] = 0.f;
  // four pixels per iteration, their loads issued before the first use (a run-time trip count keeps ONE load in flight otherwise: §3.5a)
  for (int xb = g; xb < Wo; xb += 4 * G) {
    float v[4][EV];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int xo = xb + u * G;
      if (xo < Wo) ldv<T, EV>(src + (int64_t)xo * ldo, v[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int xo = xb + u * G;
      if (xo >= Wo) break;
      const int x0 = xo / f;
      const float fx = (float)(xo - x0 * f) * inv;
      const int x1 = min(x0 + 1, W - 1);
#pragma unroll
      for (int w = 0; w < RBT_MAXW; ++w) {
        float wx = 0.f;
        if (x0 == w) wx += 1.f - fx;
        if (x1 == w) wx += fx;
        if (wx != 0.f) {
#pragma unroll
          for (int e = 0; e < EV; ++e) acc[w][e] += wx * v[u][e];
        }
      }
    }
  }
  if (g > 0) {
#pragma unroll
    for (int w = 0; w < RBT_MAXW; ++w)
      if (w < W)
#pragma unroll
        for (int e = 0; e < EV; ++e) rbt_red[((size_t)(g - 1) * W + w) * C + cv * EV + e] = acc[w][e];
  }
  __syncthreads();
  if (g == 0) {
#pragma unroll
    for (int w = 0; w < RBT_MAXW; ++w) {
      if (w >= W) break;
#pragma unroll
      for (int e = 0; e < EV; ++e) {
        float sum = acc[w][e];
        for (int q = 1; q < G; ++q) sum += rbt_red[((size_t)(q - 1) * W + w) * C + cv * EV + e];
        part[(row * W + w) * C + cv * EV + e] = sum;
      }
    }
  }
}
// pass 2: one WAVE per (n, h, w, 4 channels): its 64 lanes take the <= 2f rows of `part` in slices (all loads in flight), a fixed butterfly
// sums them (an output summed its rows one load at a time in the first version: 192 dependent-latency steps for PSPNet's level 1)
template <typename T>
__global__ __launch_bounds__(256) void resize_bilinear_bwd_tiny_cols_kernel(const float* __restrict__ part, T* __restrict__ dx, int N, int H, int W, int C,
                                                                            int f, int accumulate) {
  const int Ho = H * f, c4n = C >> 2;
  const int64_t nvec = (int64_t)N * H * W * c4n;
  const int lane = threadIdx.x & 63;
  const int64_t ov = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ov >= nvec) return;                                 // (wave-uniform)
  const int c4 = (int)(ov % c4n);
  const int w = (int)((ov / c4n) % W), h = (int)((ov / ((int64_t)c4n * W)) % H), n = (int)(ov / ((int64_t)c4n * W * H));
  const float inv = 1.f / (float)f;
  const int yo0 = max((h - 1) * f, 0), yo1 = min((h + 1) * f, Ho);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int yo = yo0 + lane; yo < yo1; yo += 64) {
    const int y0 = yo / f;
    const float fy = (float)(yo - y0 * f) * inv;
    float wy = 0.f;
    if (y0 == h) wy += 1.f - fy;
    if (min(y0 + 1, H - 1) == h) wy += fy;
    const f32x4 v = *reinterpret_cast<const f32x4*>(part + (((int64_t)n * Ho + yo) * W + w) * C + c4 * 4);
    acc += wy * v;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) acc[e] = wave_sum(acc[e]);
  if (lane == 0) {
    T* d = dx + (((int64_t)n * H + h) * W + w) * C + c4 * 4;
    float o[4] = {acc[0], acc[1], acc[2], acc[3]};
    if (accumulate) {
      float a[4];
      ldv<T, 4>(d, a);
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] += a[e];
    }
    stv<T, 4>(d, o);
  }
}
// eligibility + workspace of the two-pass form
static bool resize_bwd_tiny_ok(int N, int H, int W, int C, int factor, int ldo, int coff, int dtype) {
  static const bool on = !(getenv("STP_RESIZE_BWD_TINY") && atoi(getenv("STP_RESIZE_BWD_TINY")) == 0);
  const int ev = dtype == STP_H16 ? 8 : 4;
  if (!on || factor < 8 || W > RBT_MAXW || H > 16 || (C % ev) || (ldo % ev) || (coff % ev)) return false;
  const int cgv = C / ev;
  return cgv >= 1 && cgv <= 256 && (256 % cgv) == 0 && (size_t)(256 / cgv - 1) * W * C * sizeof(float) <= 64 * 1024 && (int64_t)N * H * factor < (1ll << 31);
}

// factor 1 (the level of a concatenation that keeps its resolution: PSPNet's feature map, FPN's stride-4 branch): the gradient is the
// channel slice [coff, coff + C) of dY copied (or added) back - 16-byte vectors, four in flight.  The generic gather kernel ran its 2 x 2
// tap loops with integer divisions and 2-byte loads on it: 244 us for 151 MB (PSPNet 768 x 768 batch 8), 236 us for FPN's (round 5).
template <typename T, int V>
__global__ __launch_bounds__(256) void slice_copy_kernel(const T* __restrict__ dy, T* __restrict__ dx, int64_t pixels, int C, int ldo, int coff, int accumulate) {
  const int cg = C / V;
  const int64_t total = pixels * cg;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t p = i / cg;
    const int c = (int)(i - p * cg) * V;
    float v[V];
    ldv<T, V>(dy + p * ldo + coff + c, v);
    if (accumulate) {
      float a[V];
      ldv<T, V>(dx + p * C + c, a);
#pragma unroll
      for (int e = 0; e < V; ++e) v[e] += a[e];
    }
    stv<T, V>(dx + p * C + c, v);
  }
}

extern "C" int stp_resize_bilinear(const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t factor, int32_t ldo,
                                   int32_t coff, int32_t dtype, void* stream) {
  if (!stp_dtype_ok(dtype)) return STP_E_BADARG;      // (the other build's 16-bit code, or garbage)
  if (!x || !y || N <= 0 || H <= 0 || W <= 0 || C <= 0 || factor < 1 || ldo < coff + C || coff < 0) return STP_E_BADARG;
  if (dtype != STP_H16 && dtype != STP_F32) return STP_E_BADARG;
  const int V = vec_for(dtype, C, ldo, coff);
  if (V > 1 && (int64_t)N * H * factor <= 65535) {
    const dim3 grid(ceil_div(W * factor * (C / V), 256), N * H * factor);
    hipStream_t s = (hipStream_t)stream;
    if (V == 8) hipLaunchKernelGGL((resize_bilinear_vec_kernel<bf16_t, 8>), grid, dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, H, W, C, factor, ldo, coff);
    else if (dtype == STP_H16) hipLaunchKernelGGL((resize_bilinear_vec_kernel<bf16_t, 4>), grid, dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, H, W, C, factor, ldo, coff);
    else hipLaunchKernelGGL((resize_bilinear_vec_kernel<float, 4>), grid, dim3(256), 0, s, (const float*)x, (float*)y, H, W, C, factor, ldo, coff);
    STP_LAUNCH_CHECK();
    return STP_OK;
  }
  const int g = grid_for((int64_t)N * H * factor * W * factor * C);
  if (dtype == STP_H16) hipLaunchKernelGGL(resize_bilinear_kernel<bf16_t>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, N, H, W, C, factor, ldo, coff);
  else if (dtype == STP_F32) hipLaunchKernelGGL(resize_bilinear_kernel<float>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)y, N, H, W, C, factor, ldo, coff);
  else return STP_E_BADARG;
  STP_LAUNCH_CHECK();
  return STP_OK;
}

// ---- PSPNet head (round 6): Conv2D(1x1) over Concatenate([feature, resized pyramid levels]) = Conv2D(1x1) over the feature with ITS columns
// of the kernel + the sum over the levels of resize(Conv2D(1x1) of the TINY level map with the level's columns) - a 1x1 convolution is a
// per-pixel linear map, so it commutes with the (per-channel, linear) bilinear resize.  The 2560-channel concatenation (377 MB at 8 x 96 x
// 96), its gradient and 80 % of the convolution's FLOP disappear.  This kernel is the sum of the resized level terms: up to four sources
// [N][h_i][w_i][C], h_i * f_i = Ho, one fp32 sum, ONE rounding; same lerp order as resize_bilinear_vec_kernel (TF 1.x, align_corners False).
struct UpSum4 { const void* x[4]; int h[4], w[4], f[4], col0[4]; FastDiv df[4], dcg; int n, cols; };
// One workgroup per OUTPUT ROW (n, yo): the two source rows every level contributes to this row (sum of the widths <= 12 columns x 2 rows x C
// channels, 16-bit: 24 KB for 512 channels) are staged in LDS once, then a thread walks (pixel, 16-byte channel group) items of the row.
// (First form: one thread per output vector with sixteen 16-byte loads from the tiny maps in flight - 128 registers, three waves per SIMD,
//  69-74 us for 75 MB = latency x occupancy, not bytes.)  Same lerp order as resize_bilinear_vec_kernel: along x first (top, bottom), then y.
template <typename T, int V>
__global__ __launch_bounds__(256) void upsample_sum_kernel(const UpSum4 a, T* __restrict__ y, int Ho, int Wo, int C) {
  extern __shared__ __attribute__((aligned(16))) char ups_rows[];          // [2][cols][C] of T
  T* rows = reinterpret_cast<T*>(ups_rows);
  const int cg = C / V;
  const int n = blockIdx.x / Ho, yo = blockIdx.x - n * Ho;                 // (workgroup-uniform)
  float fy[4];
  for (int i = 0; i < a.n; ++i) {
    const int H = a.h[i], W = a.w[i], f = a.f[i];
    const int y0 = (int)fdiv((uint32_t)yo, a.df[i]), y1 = min(y0 + 1, H - 1);
    fy[i] = (float)(yo - y0 * f) * (1.f / (float)f);
    const T* b = reinterpret_cast<const T*>(a.x[i]) + (int64_t)n * H * W * C;
    for (int t = threadIdx.x; t < 2 * W * cg; t += 256) {
      const int r = t / (W * cg), rem = t - r * (W * cg);                    // rem = w * cg + channel group: contiguous in the source row
      float v[V];
      ldv<T, V>(b + ((int64_t)(r ? y1 : y0) * W) * C + (int64_t)rem * V, v);
      stv<T, V>(rows + ((size_t)(r * a.cols + a.col0[i]) * C) + (size_t)rem * V, v);
    }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < Wo * cg; t += 256) {
    const int xo = (int)fdiv((uint32_t)t, a.dcg), c = (t - xo * cg) * V;
    float o[V];
#pragma unroll
    for (int e = 0; e < V; ++e) o[e] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i >= a.n) break;
      const int f = a.f[i];
      const int x0 = (int)fdiv((uint32_t)xo, a.df[i]), x1 = min(x0 + 1, a.w[i] - 1);
      const float fx = (float)(xo - x0 * f) * (1.f / (float)f);
      const T* r0 = rows + (size_t)a.col0[i] * C + c;
      const T* r1 = r0 + (size_t)a.cols * C;
      float v00[V], v01[V], v10[V], v11[V];
      ldv<T, V>(r0 + (size_t)x0 * C, v00);
      ldv<T, V>(r0 + (size_t)x1 * C, v01);
      ldv<T, V>(r1 + (size_t)x0 * C, v10);
      ldv<T, V>(r1 + (size_t)x1 * C, v11);
#pragma unroll
      for (int e = 0; e < V; ++e) {
        const float top = v00[e] + (v01[e] - v00[e]) * fx, bot = v10[e] + (v11[e] - v10[e]) * fx;
        o[e] += top + (bot - top) * fy[i];
      }
    }
    stv<T, V>(y + (((int64_t)n * Ho + yo) * Wo + xo) * C + c, o);
  }
}

extern "C" int stp_upsample_sum(const void* x0, const void* x1, const void* x2, const void* x3, int32_t h0, int32_t h1, int32_t h2, int32_t h3,
                                void* y, int32_t N, int32_t Ho, int32_t Wo, int32_t C, int32_t dtype, void* stream) {
  if (!stp_dtype_ok(dtype) || !y || N <= 0 || Ho <= 0 || Wo <= 0 || C <= 0 || Ho != Wo) return STP_E_BADARG;
  const void* xs[4] = {x0, x1, x2, x3};
  const int hs[4] = {h0, h1, h2, h3};
  UpSum4 a;
  a.n = 0; a.cols = 0;
  for (int i = 0; i < 4; ++i) {
    if (!xs[i]) break;
    if (hs[i] <= 0 || Ho % hs[i]) return STP_E_BADARG;
    a.x[a.n] = xs[i]; a.h[a.n] = a.w[a.n] = hs[i]; a.f[a.n] = Ho / hs[i]; a.df[a.n] = make_fastdiv((uint32_t)(Ho / hs[i]));
    a.col0[a.n] = a.cols; a.cols += hs[i];
    ++a.n;
  }
  for (int i = a.n; i < 4; ++i) { a.x[i] = nullptr; a.h[i] = a.w[i] = a.f[i] = 1; a.col0[i] = 0; a.df[i] = make_fastdiv(1u); }
  const int V = dtype == STP_H16 ? (C % 8 == 0 ? 8 : C % 4 == 0 ? 4 : 0) : (C % 4 == 0 ? 4 : 0);
  const size_t lds = (size_t)2 * a.cols * C * (dtype == STP_H16 ? 2 : 4);
  if (!a.n || !V || (int64_t)N * Ho >= (1ll << 31) || lds > 64 * 1024) return STP_E_BADARG;      // (PSPNet: 12 columns x 512 channels = 24 KB)
  a.dcg = make_fastdiv((uint32_t)(C / V));
  const dim3 grid((unsigned)(N * Ho));
  hipStream_t s = (hipStream_t)stream;
  if (dtype == STP_H16 && V == 8) hipLaunchKernelGGL((upsample_sum_kernel<bf16_t, 8>), grid, dim3(256), lds, s, a, (bf16_t*)y, Ho, Wo, C);
  else if (dtype == STP_H16) hipLaunchKernelGGL((upsample_sum_kernel<bf16_t, 4>), grid, dim3(256), lds, s, a, (bf16_t*)y, Ho, Wo, C);
  else hipLaunchKernelGGL((upsample_sum_kernel<float, 4>), grid, dim3(256), lds, s, a, (float*)y, Ho, Wo, C);
  STP_LAUNCH_CHECK();
  return STP_OK;
}

// fp32 column slice of a row-major matrix: dst[r][0 .. cols) (+)= src[r][0 .. cols) with separate row pitches - the 1x1 convolutions
// that multiply by a COLUMN RANGE of a shared kernel (stp_upsample_sum above) read their range as a dense matrix and hand their weight
// gradient back into it.  cols, both pitches and both pointers multiples of 4 floats.
__global__ __launch_bounds__(256) void copy_cols_f32_kernel(float* __restrict__ dst, int ldd, const float* __restrict__ src, int lds, int rows,
                                                            int cols4, int accumulate) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)rows * cols4) return;
  const int r = (int)(i / cols4), c = (int)(i - (int64_t)r * cols4) * 4;
  f32x4 v = *reinterpret_cast<const f32x4*>(src + (int64_t)r * lds + c);
  float* d = dst + (int64_t)r * ldd + c;
  if (accumulate) v += *reinterpret_cast<const f32x4*>(d);
  *reinterpret_cast<f32x4*>(d) = v;
}
extern "C" int stp_copy_cols_f32(float* dst, int32_t ld_dst, const float* src, int32_t ld_src, int32_t rows, int32_t cols, int32_t accumulate, void* stream) {
  if (!dst || !src || rows <= 0 || cols <= 0 || (cols & 3) || (ld_dst & 3) || (ld_src & 3) || ld_dst < cols || ld_src < cols ||
      (reinterpret_cast<uintptr_t>(dst) & 15) || (reinterpret_cast<uintptr_t>(src) & 15))
    return STP_E_BADARG;
  hipLaunchKernelGGL(copy_cols_f32_kernel, dim3(ceil_div((int64_t)rows * (cols / 4), 256)), dim3(256), 0, (hipStream_t)stream, dst, ld_dst, src, ld_src,
                     rows, cols / 4, accumulate);
  STP_LAUNCH_CHECK();
  return STP_OK;
}

extern "C" size_t stp_resize_bilinear_bwd_workspace_bytes(int32_t N, int32_t H, int32_t W, int32_t C, int32_t factor) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || factor < 2) return 0;
  const int64_t npix = (int64_t)N * H * W;
  const int span = 2 * factor;
  size_t need = 0;
  // the two-pass form for tiny maps (part[N][H * factor][W][C] fp32); sized for either 16-bit or fp32 vectors, any slice alignment
  if (factor >= 8 && W <= RBT_MAXW && H <= 16) need = (size_t)N * H * factor * W * C * sizeof(float);
  for (int V = 4; V <= 8; V += 4) {
    const int cg = ceil_div(C, V);
    const int rps = split_rows(npix * ceil_div(cg, pool_vl(cg)), span, span);
    if (rps < span) need = std::max(need, (size_t)ceil_div(span, rps) * npix * C * sizeof(float));
  }
  return need;
}

extern "C" int stp_resize_bilinear_bwd(const void* dy, void* dx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t factor,
                                       int32_t ldo, int32_t coff, int32_t dtype, int32_t accumulate, void* workspace,
                                       size_t workspace_bytes, void* stream) {
  if (!stp_dtype_ok(dtype)) return STP_E_BADARG;      // (the other build's 16-bit code, or garbage)
  if (!dy || !dx || N <= 0 || H <= 0 || W <= 0 || C <= 0 || factor < 1 || ldo < coff + C || coff < 0) return STP_E_BADARG;
  if (dtype != STP_H16 && dtype != STP_F32) return STP_E_BADARG;
  const int V = vec_for(dtype, C, ldo, coff);
  if (V > 1 && factor == 1) {      // a channel slice copied (or added) back
    const int64_t pixels = (int64_t)N * H * W;
    const int g = grid_for_amortised(pixels * (C / V), 4);
    hipStream_t s = (hipStream_t)stream;
    if (V == 8) hipLaunchKernelGGL((slice_copy_kernel<bf16_t, 8>), dim3(g), dim3(256), 0, s, (const bf16_t*)dy, (bf16_t*)dx, pixels, C, ldo, coff, accumulate);
    else if (dtype == STP_H16) hipLaunchKernelGGL((slice_copy_kernel<bf16_t, 4>), dim3(g), dim3(256), 0, s, (const bf16_t*)dy, (bf16_t*)dx, pixels, C, ldo, coff, accumulate);
    else hipLaunchKernelGGL((slice_copy_kernel<float, 4>), dim3(g), dim3(256), 0, s, (const float*)dy, (float*)dx, pixels, C, ldo, coff, accumulate);
    STP_LAUNCH_CHECK();
    return STP_OK;
  }
  if (resize_bwd_tiny_ok(N, H, W, C, factor, ldo, coff, dtype) && workspace &&
      workspace_bytes >= (size_t)N * H * factor * W * C * sizeof(float) && !(reinterpret_cast<uintptr_t>(dy) & 15)) {
    // tiny map, large factor (pyramid levels): separable two-pass reduction, every byte of dY read once
    const int ev = dtype == STP_H16 ? 8 : 4, cgv = C / ev, G = 256 / cgv;
    const size_t lds = (size_t)(G - 1) * W * C * sizeof(float);
    float* part = (float*)workspace;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)(N * H * factor));
    if (dtype == STP_H16) hipLaunchKernelGGL(resize_bilinear_bwd_tiny_rows_kernel<bf16_t>, grid, dim3(256), lds, s, (const bf16_t*)dy, part, H, W, C, factor, ldo, coff);
    else hipLaunchKernelGGL(resize_bilinear_bwd_tiny_rows_kernel<float>, grid, dim3(256), lds, s, (const float*)dy, part, H, W, C, factor, ldo, coff);
    STP_LAUNCH_CHECK();
    const int g2 = (int)(((int64_t)N * H * W * (C / 4) + 3) / 4);
    if (dtype == STP_H16) hipLaunchKernelGGL(resize_bilinear_bwd_tiny_cols_kernel<bf16_t>, dim3(g2), dim3(256), 0, s, part, (bf16_t*)dx, N, H, W, C, factor, accumulate);
    else hipLaunchKernelGGL(resize_bilinear_bwd_tiny_cols_kernel<float>, dim3(g2), dim3(256), 0, s, part, (float*)dx, N, H, W, C, factor, accumulate);
    STP_LAUNCH_CHECK();
    return STP_OK;
  }
  {
    // class logits (few channels, whole output rows fit an LDS buffer): the row-streaming kernel.  STP_RESIZE_BWD_ROWS=0: the gather kernel.
    static const bool rows_on = !(getenv("STP_RESIZE_BWD_ROWS") && atoi(getenv("STP_RESIZE_BWD_ROWS")) == 0);
    const int esz = dtype == STP_H16 ? 2 : 4, ev = 16 / esz;
    const int64_t rowB = (int64_t)W * factor * ldo * esz;
    const int nvec = (int)((rowB / 16 + 255) / 256);
    if (rows_on && factor >= 2 && !(C % ev) && !(ldo % ev) && !(coff % ev) && (int64_t)W * (C / ev) <= 512 && rowB <= 40960 && nvec <= 10 &&
        !(reinterpret_cast<uintptr_t>(dy) & 15) && !(reinterpret_cast<uintptr_t>(dx) & 15) && (int64_t)N * H < (1ll << 31)) {
      const dim3 grid((unsigned)(N * H));
      hipStream_t s = (hipStream_t)stream;
      if (dtype == STP_H16)
        hipLaunchKernelGGL(resize_bilinear_bwd_rows_kernel<bf16_t>, grid, dim3(256), (size_t)(2 * rowB), s, (const bf16_t*)dy, (bf16_t*)dx, H, W, C, factor, ldo,
                           coff, accumulate, nvec);
      else
        hipLaunchKernelGGL(resize_bilinear_bwd_rows_kernel<float>, grid, dim3(256), (size_t)(2 * rowB), s, (const float*)dy, (float*)dx, H, W, C, factor, ldo, coff,
                           accumulate, nvec);
      STP_LAUNCH_CHECK();
      return STP_OK;
    }
  }
  {
    // wide 16-bit channel slices (FPN's pyramid levels): the row-pair tile kernel.  STP_RESIZE_BWD_TILE=0: the gather kernel.
    static const bool tile_on = !(getenv("STP_RESIZE_BWD_TILE") && atoi(getenv("STP_RESIZE_BWD_TILE")) == 0);
    const int WS = factor >= 64 ? 1 : 64 / (factor > 0 ? factor : 1), CC = C < 64 ? C : 64;
    const int cols = (WS + 1) * factor - 1;
    const size_t lds = (size_t)2 * cols * CC * sizeof(float);
    if (tile_on && V == 8 && factor >= 2 && C >= 32 && C % CC == 0 && cols * (CC / 8) <= 256 * RBT_K && lds <= 96 * 1024 &&
        (int64_t)N * ((H + 1) / 2) <= 65535 && C / CC <= 65535 && !(reinterpret_cast<uintptr_t>(dy) & 15) && !(reinterpret_cast<uintptr_t>(dx) & 7)) {
      static bool attr_set = false;
      if (lds > 64 * 1024 && !attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(resize_bilinear_bwd_tile_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                96 * 1024) != hipSuccess)
          return STP_E_LAUNCH;
        attr_set = true;
      }
      const dim3 grid((unsigned)ceil_div(W, WS), (unsigned)(N * ((H + 1) / 2)), (unsigned)(C / CC));
      hipLaunchKernelGGL(resize_bilinear_bwd_tile_kernel<bf16_t>, grid, dim3(256), lds, (hipStream_t)stream, (const bf16_t*)dy, (bf16_t*)dx, H, W, C,
                         factor, ldo, coff, accumulate, WS, CC);
      STP_LAUNCH_CHECK();
      return STP_OK;
    }
  }
  if (V > 1 && factor >= 2) {
    const int cg = C / V, span = 2 * factor;
    const int64_t npix = (int64_t)N * H * W;
    // few taps per pixel: wide channel chunks keep the tap lanes busy; many taps: 128-byte chunks and more workgroups
    const int VL = span * span >= 256 ? pool_vl(cg) : (cg < 32 ? cg : 32);
    int rps = VL == pool_vl(cg) ? split_rows(npix * ceil_div(cg, VL), span, span) : span;
    int S = ceil_div(span, rps);
    if (S > 1 && (!workspace || workspace_bytes < (size_t)S * npix * C * sizeof(float))) { rps = span; S = 1; }
    float* part = S > 1 ? (float*)workspace : nullptr;
    // tap lanes per pixel: the largest power of two that the taps can keep busy; the rest of the workgroup takes more pixels
    int TLp = 1;
    while (TLp * 2 <= 256 / VL && TLp * 2 <= span * span) TLp *= 2;
    int PPB = (256 / VL) / TLp;
    if (S > 1 || PPB < 1) { PPB = 1; TLp = 256 / VL; }
    const dim3 grid((unsigned)ceil_div(npix, PPB), ceil_div(cg, VL), S);
    const size_t lds = (size_t)(256 / VL) * VL * V * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
    if (V == 8) hipLaunchKernelGGL((resize_bilinear_bwd_blk_kernel<bf16_t, 8>), grid, dim3(256), lds, s, (const bf16_t*)dy, (bf16_t*)dx, H, W, C, factor, ldo, coff, accumulate, VL, rps, part, TLp, PPB, (int)npix);
    else if (dtype == STP_H16) hipLaunchKernelGGL((resize_bilinear_bwd_blk_kernel<bf16_t, 4>), grid, dim3(256), lds, s, (const bf16_t*)dy, (bf16_t*)dx, H, W, C, factor, ldo, coff, accumulate, VL, rps, part, TLp, PPB, (int)npix);
    else hipLaunchKernelGGL((resize_bilinear_bwd_blk_kernel<float, 4>), grid, dim3(256), lds, s, (const float*)dy, (float*)dx, H, W, C, factor, ldo, coff, accumulate, VL, rps, part, TLp, PPB, (int)npix);
    STP_LAUNCH_CHECK();
    if (S > 1) {
      const int64_t ne = npix * C;
      if (dtype == STP_H16) hipLaunchKernelGGL(split_combine_kernel<bf16_t>, dim3((unsigned)ceil_div(ne, (int64_t)256)), dim3(256), 0, s, part, S, ne, 1.f, (bf16_t*)dx, accumulate);
      else hipLaunchKernelGGL(split_combine_kernel<float>, dim3((unsigned)ceil_div(ne, (int64_t)256)), dim3(256), 0, s, part, S, ne, 1.f, (float*)dx, accumulate);
      STP_LAUNCH_CHECK();
    }
    return STP_OK;
  }
  const int g = grid_for((int64_t)N * H * W * C);
  if (dtype == STP_H16) hipLaunchKernelGGL(resize_bilinear_bwd_kernel<bf16_t>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, (bf16_t*)dx, N, H, W, C, factor, ldo, coff, accumulate);
  else if (dtype == STP_F32) hipLaunchKernelGGL(resize_bilinear_bwd_kernel<float>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const float*)dy, (float*)dx, N, H, W, C, factor, ldo, coff, accumulate);
  else return STP_E_BADARG;
  STP_LAUNCH_CHECK();
  return STP_OK;
}

// ---- PSPNet's pyramid pooling in ONE pass over the feature map (round 6).  AveragePooling2D at the four pyramid levels (windows 96 / 48 / 32 /
// 16 pixels at 8 x 96 x 96 x 512) read the same 75 MB four times (4 x 20 us) and their gradients rewrite its gradient four times (108 us).
// Every coarser window is a union of windows of the finest level, so: the fp32 sums of the FINEST windows (avgpool_win_kernel with its
// partial-sum table), then one small launch that forms every level's mean from those sums in a fixed order (one rounding per output, as
// the separate launches); backward: one pass over the feature gradient that adds the <= 4 levels' dY / k^2 of its pixel.
struct PoolPyr { void* y[4]; int k[4]; int64_t first[5]; int64_t nwin; int levels; };
template <typename T>
__global__ __launch_bounds__(256) void avgpool_pyramid_combine_kernel(const float* __restrict__ part, int S, PoolPyr p, int H, int W, int C) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= p.first[p.levels]) return;
  int L = 0;
#pragma unroll
  for (int q = 1; q < 4; ++q) if (q < p.levels && i >= p.first[q]) L = q;
  const int64_t j = i - p.first[L];
  const int k0 = p.k[0], k = p.k[L], r = k / k0, Ho = H / k, Wo = W / k, H0 = H / k0, W0 = W / k0;
  const int c = (int)(j % C);
  int64_t b = j / C;
  const int wo = (int)(b % Wo); b /= Wo;
  const int ho = (int)(b % Ho);
  const int64_t n = b / Ho;
  float sum = 0.f;
  for (int a = 0; a < r; ++a)
    for (int d = 0; d < r; ++d) {
      const int64_t win = (n * H0 + (int64_t)ho * r + a) * W0 + (int64_t)wo * r + d;
      for (int q = 0; q < S; ++q) sum += part[((int64_t)q * p.nwin + win) * C + c];
    }
  Elem<T>::store((T*)p.y[L] + j, sum / (float)(k * k));
}

struct PoolPyrB { const void* dy[4]; int k[4]; float inv[4]; int levels; };
template <typename T, int V>
__global__ __launch_bounds__(256) void avgpool_pyramid_bwd_kernel(PoolPyrB p, T* __restrict__ dx, int H, int W, int C, int accumulate) {
  const int cg = C / V;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= W * cg) return;
  const int w = t / cg, c = (t - w * cg) * V;
  const int n = blockIdx.y / H, h = blockIdx.y - n * H;
  float o[V], g[V];
#pragma unroll
  for (int e = 0; e < V; ++e) o[e] = 0.f;
#pragma unroll
  for (int L = 0; L < 4; ++L) {
    if (L < p.levels) {
      const int k = p.k[L];
      ldv<T, V>((const T*)p.dy[L] + (((int64_t)n * (H / k) + h / k) * (W / k) + w / k) * C + c, g);
#pragma unroll
      for (int e = 0; e < V; ++e) o[e] += g[e] * p.inv[L];
    }
  }
  T* d = dx + (((int64_t)n * H + h) * W + w) * C + c;
  if (accumulate) {
    ldv<T, V>(d, g);
#pragma unroll
    for (int e = 0; e < V; ++e) o[e] += g[e];
  }
  stv<T, V>(d, o);
}

static int pool_pyramid_levels(const int* k) {
  int levels = 0;
  while (levels < 4 && k[levels] > 0) ++levels;
  return levels;
}
static bool pool_pyramid_ok(int N, int H, int W, int C, const int* k, int levels, int dtype) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || levels < 1 || levels > 4 || (dtype != STP_H16 && dtype != STP_F32)) return false;
  if (vec_for(dtype, C, 0, 0) <= 1 || (int64_t)N * H > 65535) return false;
  for (int i = 0; i < levels; ++i)
    if (k[i] < 1 || H % k[i] || W % k[i] || k[i] % k[0]) return false;
  return k[0] * k[0] >= 32;      // (the window kernel serves the finest level)
}
// 1 if the four AveragePooling2D(k_i) of one tensor can run as a pyramid: k0 = the finest window, every other k a multiple of it (k = 0: level unused)
extern "C" int stp_avgpool_pyramid_ok(int32_t N, int32_t H, int32_t W, int32_t C, int32_t k0, int32_t k1, int32_t k2, int32_t k3, int32_t dtype) {
  const bool on = !(getenv("STP_POOL_PYRAMID") && atoi(getenv("STP_POOL_PYRAMID")) == 0);      // (a plan-time query: read at every call)
  const int k[4] = {k0, k1, k2, k3};
  return on && stp_dtype_ok(dtype) && pool_pyramid_ok(N, H, W, C, k, pool_pyramid_levels(k), dtype) ? 1 : 0;
}
// rows-per-split of the finest level's window launch (the table of fp32 sums is always written, also with one split)
static int pool_pyramid_rps(int N, int H, int W, int C, int k0, int dtype) {
  const int V = vec_for(dtype, C, 0, 0), cg = C / V;
  const int64_t nwin = (int64_t)N * (H / k0) * (W / k0);
  return split_rows(nwin * ceil_div(cg, pool_vl(cg)), k0, k0);
}
extern "C" size_t stp_avgpool_pyramid_workspace_bytes(int32_t N, int32_t H, int32_t W, int32_t C, int32_t k0, int32_t dtype) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || k0 < 1 || H % k0 || W % k0 || (dtype != STP_H16 && dtype != STP_F32) || vec_for(dtype, C, 0, 0) <= 1) return 0;
  const int S = ceil_div(k0, pool_pyramid_rps(N, H, W, C, k0, dtype));
  return (size_t)S * N * (H / k0) * (W / k0) * C * sizeof(float);
}
extern "C" int stp_avgpool_pyramid(const void* x, void* y0, void* y1, void* y2, void* y3, int32_t k0, int32_t k1, int32_t k2, int32_t k3, int32_t N,
                                   int32_t H, int32_t W, int32_t C, int32_t dtype, void* workspace, size_t workspace_bytes, void* stream) {
  if (!stp_dtype_ok(dtype)) return STP_E_BADARG;      // (the other build's 16-bit code, or garbage)
  const int k[4] = {k0, k1, k2, k3};
  void* y[4] = {y0, y1, y2, y3};
  PoolPyr p;
  p.levels = pool_pyramid_levels(k);
  if (!x || !workspace || !pool_pyramid_ok(N, H, W, C, k, p.levels, dtype)) return STP_E_BADARG;
  if (workspace_bytes < stp_avgpool_pyramid_workspace_bytes(N, H, W, C, k0, dtype)) return STP_E_WORKSPACE;
  p.first[0] = 0;
  for (int i = 0; i < 4; ++i) {
    const bool on = i < p.levels;
    if (on && !y[i]) return STP_E_BADARG;
    p.y[i] = on ? y[i] : nullptr;
    p.k[i] = on ? k[i] : 0;
    p.first[i + 1] = p.first[i] + (on ? (int64_t)N * (H / k[i]) * (W / k[i]) * C : 0);
  }
  const int V = vec_for(dtype, C, 0, 0), cg = C / V, VL = pool_vl(cg);
  p.nwin = (int64_t)N * (H / k0) * (W / k0);
  const int rps = pool_pyramid_rps(N, H, W, C, k0, dtype), S = ceil_div(k0, rps);
  float* part = (float*)workspace;
  const dim3 grid((unsigned)p.nwin, ceil_div(cg, VL), S);
  const size_t lds = (size_t)(256 / VL) * VL * V * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
  if (V == 8) hipLaunchKernelGGL((avgpool_win_kernel<bf16_t, 8>), grid, dim3(256), lds, s, (const bf16_t*)x, (bf16_t*)nullptr, H, W, C, k0, VL, rps, part);
  else if (dtype == STP_H16) hipLaunchKernelGGL((avgpool_win_kernel<bf16_t, 4>), grid, dim3(256), lds, s, (const bf16_t*)x, (bf16_t*)nullptr, H, W, C, k0, VL, rps, part);
  else hipLaunchKernelGGL((avgpool_win_kernel<float, 4>), grid, dim3(256), lds, s, (const float*)x, (float*)nullptr, H, W, C, k0, VL, rps, part);
  STP_LAUNCH_CHECK();
  const unsigned g = (unsigned)ceil_div(p.first[p.levels], (int64_t)256);
  if (dtype == STP_H16) hipLaunchKernelGGL(avgpool_pyramid_combine_kernel<bf16_t>, dim3(g), dim3(256), 0, s, part, S, p, H, W, C);
  else hipLaunchKernelGGL(avgpool_pyramid_combine_kernel<float>, dim3(g), dim3(256), 0, s, part, S, p, H, W, C);
  STP_LAUNCH_CHECK();
  return STP_OK;
}
extern "C" int stp_avgpool_pyramid_bwd(const void* dy0, const void* dy1, const void* dy2, const void* dy3, int32_t k0, int32_t k1, int32_t k2, int32_t k3,
                                       void* dx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t dtype, int32_t accumulate, void* stream) {
  if (!stp_dtype_ok(dtype)) return STP_E_BADARG;      // (the other build's 16-bit code, or garbage)
  const int k[4] = {k0, k1, k2, k3};
  const void* dy[4] = {dy0, dy1, dy2, dy3};
  PoolPyrB p;
  p.levels = pool_pyramid_levels(k);
  if (!dx || !pool_pyramid_ok(N, H, W, C, k, p.levels, dtype)) return STP_E_BADARG;
  for (int i = 0; i < 4; ++i) {
    const bool on = i < p.levels;
    if (on && !dy[i]) return STP_E_BADARG;
    p.dy[i] = on ? dy[i] : nullptr;
    p.k[i] = on ? k[i] : 1;
    p.inv[i] = on ? 1.f / (float)(k[i] * k[i]) : 0.f;
  }
  const int V = vec_for(dtype, C, 0, 0);
  const dim3 grid(ceil_div(W * (C / V), 256), N * H);
  hipStream_t s = (hipStream_t)stream;
  if (V == 8) hipLaunchKernelGGL((avgpool_pyramid_bwd_kernel<bf16_t, 8>), grid, dim3(256), 0, s, p, (bf16_t*)dx, H, W, C, accumulate);
  else if (dtype == STP_H16) hipLaunchKernelGGL((avgpool_pyramid_bwd_kernel<bf16_t, 4>), grid, dim3(256), 0, s, p, (bf16_t*)dx, H, W, C, accumulate);
  else hipLaunchKernelGGL((avgpool_pyramid_bwd_kernel<float, 4>), grid, dim3(256), 0, s, p, (float*)dx, H, W, C, accumulate);
  STP_LAUNCH_CHECK();
  return STP_OK;
}

// ------------------------------------------------------------------------------------------
// MaxPooling2D(pool_size = strides = k) for any k (PSPNet `psp_pooling_type: max`, schemas/segmentation.raml:233-236): one thread
// per (output pixel, channel), channels fastest (coalesced); idx = position kh * k + kw of the FIRST maximum (int32, k up to the
// whole map); the gradient goes to that position only (windows do not overlap: one thread per input element, no atomics)
template <typename T>
__global__ __launch_bounds__(256) void maxpool_k_kernel(const T* __restrict__ x, T* __restrict__ y, int32_t* __restrict__ idx, int N, int H, int W,
                                                        int C, int k) {
  const int Ho = H / k, Wo = W / k;
  const int64_t total = (int64_t)N * Ho * Wo * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    const int64_t p = i / C;
    const int xo = (int)(p % Wo), yo = (int)((p / Wo) % Ho), n = (int)(p / ((int64_t)Wo * Ho));
    const T* b = x + (((int64_t)n * H + yo * k) * W + xo * k) * C + c;
    float best = -3.4e38f;
    int bi = 0;
    for (int a = 0; a < k; ++a)
      for (int q = 0; q < k; ++q) {
        const float v = Elem<T>::load(b + ((int64_t)a * W + q) * C);
        if (v > best) { best = v; bi = a * k + q; }
      }
    Elem<T>::store(y + i, best);
    if (idx) idx[i] = bi;
  }
}
template <typename T>
__global__ __launch_bounds__(256) void maxpool_k_bwd_kernel(const int32_t* __restrict__ idx, const T* __restrict__ dy, T* __restrict__ dx, int N,
                                                            int H, int W, int C, int k, int accumulate) {
  const int Ho = H / k, Wo = W / k;
  const int64_t total = (int64_t)N * H * W * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    const int64_t p = i / C;
    const int xi = (int)(p % W), yi = (int)((p / W) % H), n = (int)(p / ((int64_t)W * H));
    const int64_t o = (((int64_t)n * Ho + yi / k) * Wo + xi / k) * C + c;
    float g = (idx[o] == (yi % k) * k + (xi % k)) ? Elem<T>::load(dy + o) : 0.f;
    if (accumulate) g += Elem<T>::load(dx + i);
    Elem<T>::store(dx + i, g);
  }
}
extern "C" int stp_maxpool_k(const void* x, void* y, int32_t* idx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t k, int32_t dtype,
                             void* stream) {
  if (!stp_dtype_ok(dtype)) return STP_E_BADARG;      // (the other build's 16-bit code, or garbage)
  if (!x || !y || N <= 0 || H <= 0 || W <= 0 || C <= 0 || k < 1 || H % k || W % k) return STP_E_BADARG;
  const int g = grid_for((int64_t)N * (H / k) * (W / k) * C);
  if (dtype == STP_H16) hipLaunchKernelGGL(maxpool_k_kernel<bf16_t>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, idx, N, H, W, C, k);
  else if (dtype == STP_F32) hipLaunchKernelGGL(maxpool_k_kernel<float>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)y, idx, N, H, W, C, k);
  else return STP_E_BADARG;
  STP_LAUNCH_CHECK();
  return STP_OK;
}
extern "C" int stp_maxpool_k_bwd(const int32_t* idx, const void* dy, void* dx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t k,
                                 int32_t dtype, int32_t accumulate, void* stream) {
  if (!stp_dtype_ok(dtype)) return STP_E_BADARG;      // (the other build's 16-bit code, or garbage)
  if (!idx || !dy || !dx || N <= 0 || H <= 0 || W <= 0 || C <= 0 || k < 1 || H % k || W % k) return STP_E_BADARG;
  const int g = grid_for((int64_t)N * H * W * C);
  if (dtype == STP_H16) hipLaunchKernelGGL(maxpool_k_bwd_kernel<bf16_t>, dim3(g), dim3(256), 0, (hipStream_t)stream, idx, (const bf16_t*)dy, (bf16_t*)dx, N, H, W, C, k, accumulate);
  else if (dtype == STP_F32) hipLaunchKernelGGL(maxpool_k_bwd_kernel<float>, dim3(g), dim3(256), 0, (hipStream_t)stream, idx, (const float*)dy, (float*)dx, N, H, W, C, k, accumulate);
  else return STP_E_BADARG;
  STP_LAUNCH_CHECK();
  return STP_OK;
}

// ------------------------------------------------------------------------------------------
// ResizeImage(factor, interpolation='nearest') = UpSampling2D(factor): y[n, yo, xo, coff + c] = x[n, yo / f, xo / f, c]; the
// gradient sums the f x f outputs of an input pixel in row-major order (segmentation_models' FPN `interpolation: nearest`, PSPNet
// `final_interpolation: nearest`, schemas/segmentation.raml:196-199, 245-248)
template <typename T>
__global__ __launch_bounds__(256) void resize_nearest_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int C, int f,
                                                             int ldo, int coff) {
  const int Ho = H * f, Wo = W * f;
  const int64_t total = (int64_t)N * Ho * Wo * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    const int64_t p = i / C;
    const int xo = (int)(p % Wo), yo = (int)((p / Wo) % Ho), n = (int)(p / ((int64_t)Wo * Ho));
    y[p * ldo + coff + c] = x[(((int64_t)n * H + yo / f) * W + xo / f) * C + c];
  }
}
template <typename T>
__global__ __launch_bounds__(256) void resize_nearest_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int N, int H, int W, int C,
                                                                 int f, int ldo, int coff, int accumulate) {
  const int Wo = W * f, Ho = H * f;
  const int64_t total = (int64_t)N * H * W * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    const int64_t p = i / C;
    const int xi = (int)(p % W), yi = (int)((p / W) % H), n = (int)(p / ((int64_t)W * H));
    float s = 0.f;
    for (int a = 0; a < f; ++a)
      for (int b = 0; b < f; ++b) s += Elem<T>::load(dy + (((int64_t)n * Ho + yi * f + a) * Wo + xi * f + b) * ldo + coff + c);
    if (accumulate) s += Elem<T>::load(dx + i);
    Elem<T>::store(dx + i, s);
  }
}
extern "C" int stp_resize_nearest(const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t factor, int32_t ldo, int32_t coff,
                                  int32_t dtype, void* stream) {
  if (!stp_dtype_ok(dtype)) return STP_E_BADARG;      // (the other build's 16-bit code, or garbage)
  if (!x || !y || N <= 0 || H <= 0 || W <= 0 || C <= 0 || factor < 1 || ldo < coff + C || coff < 0) return STP_E_BADARG;
  const int g = grid_for((int64_t)N * H * factor * W * factor * C);
  if (dtype == STP_H16) hipLaunchKernelGGL(resize_nearest_kernel<bf16_t>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, N, H, W, C, factor, ldo, coff);
  else if (dtype == STP_F32) hipLaunchKernelGGL(resize_nearest_kernel<float>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)y, N, H, W, C, factor, ldo, coff);
  else return STP_E_BADARG;
  STP_LAUNCH_CHECK();
  return STP_OK;
}
extern "C" int stp_resize_nearest_bwd(const void* dy, void* dx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t factor, int32_t ldo,
                                      int32_t coff, int32_t dtype, int32_t accumulate, void* stream) {
  if (!stp_dtype_ok(dtype)) return STP_E_BADARG;      // (the other build's 16-bit code, or garbage)
  if (!dy || !dx || N <= 0 || H <= 0 || W <= 0 || C <= 0 || factor < 1 || ldo < coff + C || coff < 0) return STP_E_BADARG;
  const int g = grid_for((int64_t)N * H * W * C);
  if (dtype == STP_H16) hipLaunchKernelGGL(resize_nearest_bwd_kernel<bf16_t>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, (bf16_t*)dx, N, H, W, C, factor, ldo, coff, accumulate);
  else if (dtype == STP_F32) hipLaunchKernelGGL(resize_nearest_bwd_kernel<float>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const float*)dy, (float*)dx, N, H, W, C, factor, ldo, coff, accumulate);
  else return STP_E_BADARG;
  STP_LAUNCH_CHECK();
  return STP_OK;
}

// ------------------------------------------------------------------------------------------
// per-channel sums of a [rows][C] tensor (bias gradient) and dst += src
__global__ __launch_bounds__(256) void colsum_finalize_kernel(const float* partial, int blocks, int C, float* out,
                                                              int accumulate) {
  const int c = blockIdx.x * 4 + (threadIdx.x & 3);
  const Sum2 t = reduce_partials(partial, blocks, C, c, false);
  if ((threadIdx.x >> 2) != 0 || c >= C) return;
  out[c] = accumulate ? out[c] + (float)t.s : (float)t.s;
}

extern "C" int stp_channel_sum(const void* x, int32_t dtype, int64_t rows, int32_t C, float* out, int32_t accumulate,
                               void* workspace, size_t workspace_bytes, void* stream) {
  if (!stp_dtype_ok(dtype)) return STP_E_BADARG;      // (the other build's 16-bit code, or garbage)
  if (!x || !out || !workspace || rows <= 0 || C <= 0 || (C & 3)) return STP_E_BADARG;
  if (workspace_bytes < stp_bn_workspace_bytes(C)) return STP_E_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const int blocks = bn_blocks(rows);
  float* partial = (float*)workspace;
  if (dtype == STP_H16 && (C & 7) == 0)
    hipLaunchKernelGGL((bn_partial_kernel<bf16_t, 8>), dim3(blocks), dim3(256), 256 * 16 * sizeof(float), s, (const bf16_t*)x, rows, C, partial);
  else if (dtype == STP_H16)
    hipLaunchKernelGGL((bn_partial_kernel<bf16_t, 4>), dim3(blocks), dim3(256), 256 * 8 * sizeof(float), s, (const bf16_t*)x, rows, C, partial);
  else if (dtype == STP_F32)
    hipLaunchKernelGGL((bn_partial_kernel<float, 4>), dim3(blocks), dim3(256), 256 * 8 * sizeof(float), s, (const float*)x, rows, C, partial);
  else
    return STP_E_BADARG;
  STP_LAUNCH_CHECK();
  hipLaunchKernelGGL(colsum_finalize_kernel, dim3(ceil_div(C, 4)), dim3(256), 0, s, partial, blocks, C, out, accumulate);
  STP_LAUNCH_CHECK();
  return STP_OK;
}

template <typename T>
__global__ __launch_bounds__(256) void add_inplace_kernel(T* __restrict__ dst, const T* __restrict__ src, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256)
    store4(dst + i * 4, ld4<T>(dst + i * 4) + ld4<T>(src + i * 4));
}

extern "C" int stp_add_inplace(void* dst, const void* src, int64_t count, int32_t dtype, void* stream) {
  if (!stp_dtype_ok(dtype)) return STP_E_BADARG;      // (the other build's 16-bit code, or garbage)
  if (!dst || !src || count <= 0 || (count & 3)) return STP_E_BADARG;
  const int g = grid_for(count >> 2);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == STP_H16)
    hipLaunchKernelGGL(add_inplace_kernel<bf16_t>, dim3(g), dim3(256), 0, s, (bf16_t*)dst, (const bf16_t*)src, count >> 2);
  else if (dtype == STP_F32)
    hipLaunchKernelGGL(add_inplace_kernel<float>, dim3(g), dim3(256), 0, s, (float*)dst, (const float*)src, count >> 2);
  else
    return STP_E_BADARG;
  STP_LAUNCH_CHECK();
  return STP_OK;
}
