// POINTWISE (1x1 / stride 1) convolution as a pixel-STREAMING kernel (round 6) - 16-bit storage.
//
// The bottleneck ResNets (BASELINE.json configs[3] FPN/ResNet50 and configs[4] PSPNet/ResNet101) spend a third of their step in 1x1
// convolutions with 64 ... 512 channels on 75 K ... 295 K pixels: 32 - 100 FLOP per byte, i.e. HBM-bound by a factor of 3 - 10 against
// the MFMA pipe.  On the per-tap buffer-DMA kernel (conv_igemm.hip) they ran 1.6 - 3.5 x above their HBM floor
// (profiles/r06a_floor_config{3,4}.txt): one K-step per workgroup (K = 64 ... 512 = 1 ... 8 steps), so every 64 x 128 tile pays the
// whole prologue (operand ring fill), an epilogue with three workgroup barriers and a partial-sum column of its own, and nothing of
// tile i overlaps tile i + 1 inside a workgroup.  With the BatchNormalization-backward epilogue the launch is epilogue-bound
// (64 -> 256 @ 8 x 192 x 192: 148 us against a 54 us floor).
//
// Here (the recipe of the small-channel streaming kernels, conv_sc_lean.hip, for a 1x1 window):
//   * PERSISTENT workgroups walk TP-pixel tiles of the flattened [pixels][CIN] tensor (a 1x1 window has no geometry);
//   * the WHOLE weight matrix lives in registers as MFMA A fragments, distributed over the waves (wave (wm, wn) owns COUT / WM output
//     channels x TP / WN pixels): fetched once per workgroup - no operand ring, no barrier inside a tile;
//   * the next tile's pixels travel by LDS-DMA into the other half of a double buffer while this tile is multiplied and stored;
//     the rows are XOR-swizzled by WHICH 16-byte piece a lane requests (the LDS side of a DMA instruction is linear), so the
//     ds_read_b128 fragment reads are conflict-free;
//   * the epilogue works straight from the accumulators: the assignment of output channels to MFMA row slots is free (it is only
//     which weight row a lane loads), so two 16-row blocks are interleaved such that a lane owns EIGHT CONSECUTIVE channels of a pixel -
//     one 16-byte store / operand load, four lanes cover 64 contiguous bytes of a pixel row; no LDS staging, no barrier;
//   * residual / BatchNormalization-input / accumulate operands of a tile are requested at the top of the tile, before the wait for
//     its pixels (buffer loads: lane constant + scalar tile offset), and consumed after the MFMA phase;
//   * fused statistics / BatchNormalization-backward sums are accumulated in registers ACROSS the tiles of a workgroup and reduced
//     ONCE: the table has one column per workgroup (<= 512) instead of one per 128-pixel tile (2304 at 8 x 192 x 192).
// Counted waits: loads retire in order AMONG LOADS (stores may overtake them), so every count is stated on loads only: at the top of
// tile t at most the NOPL operand loads may be outstanding (then the older LDS-DMA of this tile has landed); after the barrier the
// pixels of t + 1 are requested, and the wait in front of the epilogue leaves exactly those NPASS instructions in flight.  The LDS
// fragment reads are inline asm: a C++ LDS read after an LDS-DMA instruction gets an s_waitcnt vmcnt(0) from the wait-count pass
// (DESIGN.md 3.1a), which would drain the prefetch.
// Deterministic: static tile -> workgroup assignment, fixed reduction order; a replay is bit-identical.
#include "conv_common.h"
#include "conv_sc.h"
#include "../../include/stp_hip.h"

struct PwArgs {
  const char* x;        // [P][CIN]
  const char* w;        // [COUT][CIN] (the forward copy of a forward launch, the flipped + transposed copy of a data gradient)
  char* dst;            // [P][COUT]
  const char* opr;      // residual [P][COUT] (EP 0 / 1) or the BatchNormalization input (EP 2); may be NULL (EP 0 / 1)
  const float* bias;    // EP 0 only
  float* stats;         // [2][COUT][gridDim.x]
  const float *mean, *rstd, *gamma, *beta;
  int bnb_relu;
  int ntiles;
  uint32_t x_bytes, o_bytes;
};

enum { PW_PLAIN = 0, PW_STATS = 1, PW_BNB = 2 };

// NW waves (4 or 8) as WM (channels) x WN (pixels).  OPA: accumulate into dst; OPR: a residual is added (EP 0 / 1; EP 2 always reads opr)
template <int CIN, int COUT, int NW, int TP, int EP, bool OPA, bool OPR>
__global__ __launch_bounds__(NW * 64, (NW == 4 ? 2 : 1)) void conv_pw_kernel(const PwArgs a) {
  constexpr int WM = (COUT / 32 < NW) ? COUT / 32 : NW, WN = NW / WM;
  constexpr int CW = COUT / WM, MP = CW / 32, PWV = TP / WN, NB = PWV / 16, KS = CIN / 32;
  constexpr int SLOTS = CIN / 8, TILE = TP * CIN * 2, PASSB = NW * 1024, NPASS = TILE / PASSB;
  constexpr int KG = KS < 4 ? KS : 4, KH = KS / KG, NSTEP = NB * KH;      // (KG fragments of 4 registers in flight per step, twice)
  constexpr bool HASR = EP == PW_BNB || OPR;
  constexpr int NST = MP * NB, NOPL = NST * ((HASR ? 1 : 0) + (OPA ? 1 : 0));
  static_assert(WM * WN == NW && CW % 32 == 0 && PWV % 16 == 0 && CIN % 32 == 0 && TILE % PASSB == 0 && KS % KG == 0, "config");
  static_assert(EP != PW_STATS || !OPA, "statistics of an accumulated tensor: not a training-graph case");
  static_assert((NB - 1) * 16 * CIN * 2 < 65536, "fragment offsets are instruction immediates");
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave - wm * WN;
  const int n16 = lane & 15, q = lane >> 4;
  const int G = (int)gridDim.x;
  int t = (int)blockIdx.x;

  // ---- LDS-DMA: per-pass lane constants.  LDS slot S (16 bytes) of the tile = pixel S / SLOTS, position S % SLOTS; the piece stored
  // there is channel chunk position ^ key(pixel): key = (pixel >> 1) & 7 for 128-byte rows (the pixel's parity picks the half of a
  // 256-byte bank line), pixel & 15 for rows of whole bank lines
  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  uint32_t lo[NPASS];
#pragma unroll
  for (int ps = 0; ps < NPASS; ++ps) {
    const int S = (ps * NW + wave) * 64 + lane, p = S / SLOTS, j = S - p * SLOTS;
    const int key = CIN == 64 ? ((p >> 1) & 7) : (p & 15);
    lo[ps] = (uint32_t)((p * CIN + (j ^ key) * 8) * 2);
  }
  auto issue = [&](int tile, int buf, bool live) __attribute__((always_inline)) {
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (__attribute__((address_space(3))) void*)(smem + buf * TILE + ps * PASSB + wave * 1024), 16,
                                               live ? (int)lo[ps] : (int)0x80000000u, live ? tile * TILE : 0, 0, 0);
  };
  issue(t, 0, true);

  // ---- once per workgroup: the weights as A fragments.  Row slot r of block b of pair mp <-> channel wm * CW + mp * 32 + 8 * (r >> 2)
  // + 4 * b + (r & 3): the lane of C-layout row group q (rows 4q .. 4q + 3 of both blocks) then owns channels 8q .. 8q + 7 of the pair
  u32x4 fa[MP][2][KS];
#pragma unroll
  for (int mp = 0; mp < MP; ++mp)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int ch = wm * CW + mp * 32 + 8 * (n16 >> 2) + 4 * b + (n16 & 3);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) fa[mp][b][ks] = *reinterpret_cast<const u32x4*>(a.w + ((size_t)ch * CIN + ks * 32 + 8 * q) * 2);
    }
  // B fragment of K-step ks for pixel block nb: pixel wn * PWV + nb * 16 + n16, channel chunk ks * 4 + q (the key depends on n16 only)
  uint32_t badr[KS];
  {
    const int p = wn * PWV + n16, key = CIN == 64 ? ((n16 >> 1) & 7) : n16;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) badr[ks] = (uint32_t)(uintptr_t)smem + (uint32_t)((p * SLOTS + ((ks * 4 + q) ^ key)) * 16);
  }
  // epilogue addressing: byte offset of the lane's 8 channels of pixel (wn * PWV + n16) in a [TP][COUT] tile
  const uint32_t lvo = (uint32_t)(((wn * PWV + n16) * COUT + wm * CW + 8 * q) * 2);
  const __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc((void*)a.dst, 0, a.o_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsr = __builtin_amdgcn_make_buffer_rsrc((void*)(HASR ? a.opr : a.dst), 0, a.o_bytes, 0x00020000);
  const int ch0 = wm * CW + 8 * q;                 // + mp * 32: the lane's first channel of pair mp

  f32x4 binit[MP][2];
#pragma unroll
  for (int mp = 0; mp < MP; ++mp)
#pragma unroll
    for (int b = 0; b < 2; ++b) binit[mp][b] = (EP == PW_PLAIN && a.bias) ? *reinterpret_cast<const f32x4*>(a.bias + ch0 + mp * 32 + 4 * b) : f32x4{0.f, 0.f, 0.f, 0.f};
  f32x2v ksc[MP][4], ksh[MP][4];
  if (EP == PW_BNB) {
#pragma unroll
    for (int mp = 0; mp < MP; ++mp) {
      const int c = ch0 + mp * 32;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float r = a.rstd[c + e], mu = a.mean[c + e];
        const float sc = a.gamma ? r * a.gamma[c + e] : r;
        ksc[mp][e >> 1][e & 1] = sc;
        ksh[mp][e >> 1][e & 1] = (a.beta ? a.beta[c + e] : 0.f) - mu * sc;
      }
    }
  }
  const float alo = a.bnb_relu ? __uint_as_float(1u) : -__builtin_inff();
  const float ahi = a.bnb_relu == 2 ? __uint_as_float(0x40bfffffu) : __builtin_inff();
  const bool relu1 = a.bnb_relu == 1;
  f32x2v ss[MP][4], qq[MP][4];
#pragma unroll
  for (int mp = 0; mp < MP; ++mp)
#pragma unroll
    for (int e = 0; e < 4; ++e) ss[mp][e] = qq[mp][e] = f32x2v{0.f, 0.f};

  for (int it = 0; t < a.ntiles; t += G, ++it) {
    const int buf = it & 1;
    const uint32_t so = (uint32_t)t * (uint32_t)(TP * COUT * 2);
    // operands of this tile (consumed after the MFMA phase)
    u32x4 opr[MP][NB], opa[MP][NB];
    if (HASR) {
#pragma unroll
      for (int mp = 0; mp < MP; ++mp)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) opr[mp][nb] = __builtin_amdgcn_raw_buffer_load_b128(rsr, lvo + (uint32_t)((nb * 16 * COUT + mp * 32) * 2), so, 0);
    }
    if (OPA) {
#pragma unroll
      for (int mp = 0; mp < MP; ++mp)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) opa[mp][nb] = __builtin_amdgcn_raw_buffer_load_b128(rsd, lvo + (uint32_t)((nb * 16 * COUT + mp * 32) * 2), so, 0);
    }
    // this tile's pixels have landed (every wave waits for its own pieces, the barrier makes that all of them); the other buffer half
    // was last read in the MFMA phase of tile t - G, which every wave has left.  The count: LOADS retire in order among loads (so with
    // at most NOPL operations outstanding the LDS-DMA of this tile - older than the operand loads - is done), but a STORE may retire
    // before an older load: counting the previous tile's stores as "still in the queue behind the DMA" (vmcnt(NST + NOPL), first build)
    // read tiles before they had landed - 2.6 % of a 64 -> 64 launch wrong.  The wait therefore also drains the stores of tile t - G.
    wait_vmcnt<NOPL>();
    lds_barrier();
    issue(t + G, buf ^ 1, t + G < a.ntiles);

    // ---- MFMA phase: NSTEP = NB x KH steps of KG K-steps; the fragments of step s + 1 are in flight under the MFMAs of step s
    f32x4 acc[MP][2][NB];
#pragma unroll
    for (int mp = 0; mp < MP; ++mp)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[mp][b][nb] = binit[mp][b];
    u32x4 fb[2][KG];
    const uint32_t bsel = (uint32_t)(buf * TILE);
#pragma unroll
    for (int k = 0; k < KG; ++k) asm volatile("ds_read_b128 %0, %1 offset:0" : "=v"(fb[0][k]) : "v"(badr[k] + bsel));
    sc_unroll<NSTEP>([&](auto s_) {
      constexpr int s = decltype(s_)::value, nb = s / KH, kh = s % KH;
      if constexpr (s + 1 < NSTEP) {
        constexpr int nb1 = (s + 1) / KH, kh1 = (s + 1) % KH;
#pragma unroll
        for (int k = 0; k < KG; ++k)
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[(s + 1) & 1][k]) : "v"(badr[kh1 * KG + k] + bsel), "n"(nb1 * 16 * CIN * 2));
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(KG) : "memory");
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
#pragma unroll
      for (int k = 0; k < KG; ++k) asm volatile("" : "+v"(fb[s & 1][k]));
#pragma unroll
      for (int k = 0; k < KG; ++k)
#pragma unroll
        for (int mp = 0; mp < MP; ++mp)
#pragma unroll
          for (int b = 0; b < 2; ++b) acc[mp][b][nb] = mfma16_16x16x32(fa[mp][b][kh * KG + k], fb[s & 1][k], acc[mp][b][nb]);
    });

    // ---- epilogue from the accumulators: the operands have landed, the next tile's pixels stay in flight
    wait_vmcnt<NPASS>();
#pragma unroll
    for (int mp = 0; mp < MP; ++mp)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const f32x4 c0 = acc[mp][0][nb], c1 = acc[mp][1][nb];
        f32x2v v[4] = {f32x2v{c0.x, c0.y}, f32x2v{c0.z, c0.w}, f32x2v{c1.x, c1.y}, f32x2v{c1.z, c1.w}};
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (EP != PW_BNB && OPR) v[e] += unpack_bf16x2(opr[mp][nb][e]);
          if (OPA) v[e] += unpack_bf16x2(opa[mp][nb][e]);
          if (EP != PW_BNB) o[e] = pack_bf16x2(v[e].x, v[e].y);
          if (EP == PW_STATS) {
            const f32x2v sv = unpack_bf16x2(o[e]);
            ss[mp][e] += sv;
            qq[mp][e] += sv * sv;
          }
          if (EP == PW_BNB) {
            // masked gradient (the activation mask re-derived from the BatchNormalization input with the forward's fma), rounded once;
            // the sums take the values as stored (same arithmetic as the halo kernel's EP 2)
            const f32x2v xv = unpack_bf16x2(opr[mp][nb][e]);
            const f32x2v tt = xv * ksc[mp][e] + ksh[mp][e];
            const bool on0 = relu1 ? tt.x > 0.f : __builtin_amdgcn_fmed3f(tt.x, alo, ahi) == tt.x;
            const bool on1 = relu1 ? tt.y > 0.f : __builtin_amdgcn_fmed3f(tt.y, alo, ahi) == tt.y;
            o[e] = pack_bf16x2(on0 ? v[e].x : 0.f, on1 ? v[e].y : 0.f);
            const f32x2v g = unpack_bf16x2(o[e]);
            ss[mp][e] += g;
            qq[mp][e] += g * xv;
          }
        }
        __builtin_amdgcn_sched_barrier(0);      // (the store stays BEHIND the arithmetic that reads o: see the hazard note below)
        __builtin_amdgcn_raw_buffer_store_b128(o, rsd, lvo + (uint32_t)((nb * 16 * COUT + mp * 32) * 2), so, 0);
        // STORE-DATA HAZARD (found on MI355X, round 6): a 128-bit buffer store WITH AN SGPR soffset still reads its data registers
        // during the cycles after issue - a VALU instruction that overwrites the first data register in the very next slot (here: the
        // statistics arithmetic recycling o) corrupted that dword in the last four lanes of every 16-lane row (0.3 - 1.4 % of the
        // outputs of every statistics / BatchNormalization-backward launch; the plain launches, which do not reuse the registers at
        // once, were clean).  hipcc's hazard recognizer inserts the wait state only for stores WITHOUT a register soffset.  The store
        // is fenced on both sides (the scheduler had hoisted it above the sums) and followed by two idle slots
        // (tests/test_isa_hazards.py checks the generated code):
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 1");
        __builtin_amdgcn_sched_barrier(0);
      }
  }

  // ---- once per workgroup: the column of fused sums.  The 16 lanes of a DPP row hold 16 pixels of the same 8 channels; the WN waves
  // that share the channels meet through LDS in a fixed order
  if (EP != PW_PLAIN) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the dummy LDS-DMA of the last tile writes zeros into the buffer reused below)
    lds_barrier();
    float* red = reinterpret_cast<float*>(smem);           // [WN][2][COUT]
#pragma unroll
    for (int mp = 0; mp < MP; ++mp)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const float s1 = row_sum16_to_lane15(ss[mp][e][h]), s2 = row_sum16_to_lane15(qq[mp][e][h]);
          if (n16 == 15) {
            const int c = ch0 + mp * 32 + 2 * e + h;
            red[(wn * 2 + 0) * COUT + c] = s1;
            red[(wn * 2 + 1) * COUT + c] = s2;
          }
        }
    lds_barrier();
    for (int c = tid; c < COUT; c += NW * 64) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int w_ = 0; w_ < WN; ++w_) { s1 += red[(w_ * 2 + 0) * COUT + c]; s2 += red[(w_ * 2 + 1) * COUT + c]; }
      if (EP == PW_BNB) s2 = a.rstd[c] * (s2 - a.mean[c] * s1);      // sum g * xhat = rstd * (sum g * x - mean * sum g)
      a.stats[(size_t)c * G + blockIdx.x] = s1;
      a.stats[((size_t)COUT + c) * G + blockIdx.x] = s2;
    }
  }
#endif
}

// ---- host side ---------------------------------------------------------------------------------------------------------------------
struct PwConfig { int cin, cout, nw, tp; };
// BatchNormalization-backward epilogue per shape: 2 = with and without accumulate, 1 = without only, 0 = not served (the A fragments of
// the 128 K-element matrices leave no room for the epilogue's constants and operands: those launches stay on the per-tap kernel)
constexpr int pw_bnb_level(int cin, int cout) { return (cin == 256 && cout == 512) ? 0 : ((cin == 128 && cout == 512) || (cin == 512 && cout == 128)) ? 1 : 2; }
// the shapes of the bottleneck ResNets' stages 1 - 2, the FPN laterals and their data gradients (weights <= 128 K elements: they fit the
// register budget of one workgroup as A fragments).  NW / TP: 4-wave workgroups (two or more per CU, out of phase) where the registers
// allow, 8 waves for the wide layers
static const PwConfig PW_CONFIGS[] = {
    {64, 64, 4, 64},   {64, 256, 4, 64},  {256, 64, 4, 64},  {256, 128, 4, 64}, {128, 256, 8, 128},
    {128, 512, 8, 64}, {512, 128, 4, 32}, {256, 256, 8, 64}, {512, 256, 8, 32}, {256, 512, 8, 32},
    {64, 512, 8, 64},      // (the data gradient of FPN's class head in its tap-channel form: 64 padded tap channels -> 512 at 4 x 256 x 256)
};
static const PwConfig* pw_config(int cin, int cout) {
  for (const PwConfig& c : PW_CONFIGS)
    if (c.cin == cin && c.cout == cout) return &c;
  return nullptr;
}
static bool pw_auto() {
  static const bool on = !(getenv("STP_PW") && atoi(getenv("STP_PW")) == 0);      // STP_PW=0: the per-tap kernel again (A/B runs)
  return on;
}

extern "C" int stp_conv2d_pw_eligible(const stp_conv_params* p) {
  if (!p || p->dtype != STP_H16 || p->KH != 1 || p->KW != 1 || p->stride != 1 || p->pad != 0 || p->src0_mode != STP_SRC_DIRECT) return 0;
  if (p->C1 || p->src1 || p->dst1 || (p->Cd0 && p->Cd0 != p->Cout) || p->relu || p->dst_sum2x2 || p->stats_slots || p->src_bn_mean || p->weight_up ||
      p->fold_src || p->stats_group > 1 || p->s2d_dgrad || p->accumulate1)
    return 0;
  if (p->Ho != p->Hs0 || p->Wo != p->Ws0 || p->Hv != p->Hs0 || p->Wv != p->Ws0) return 0;
  if (p->tile == 0 && !pw_auto()) return 0;
  const PwConfig* c = pw_config(p->C0, p->Cout);
  if (!c) return 0;
  const int64_t P = (int64_t)p->N * p->Ho * p->Wo;
  if (P % c->tp || P * p->C0 * 2 >= (1ll << 31) || P * p->Cout * 2 >= (1ll << 31)) return 0;
  // (stats_partial may still be NULL here: the sizing queries - stp_conv2d_stats_floats - run before the table is allocated, and must
  //  give the answer of the launch that follows; a BatchNormalization-backward launch without its table is refused by stp_conv2d_pw)
  const bool bnb = p->bnb_x != nullptr, stats = p->stats_partial != nullptr || bnb;
  if (bnb && (!p->bnb_mean || !p->bnb_rstd || p->residual || p->bias)) return 0;
  if (bnb && pw_bnb_level(c->cin, c->cout) < (p->accumulate0 ? 2 : 1)) return 0;
  if (stats && !bnb && (p->bias || p->accumulate0)) return 0;
  if (!stats && !bnb && p->residual && p->accumulate0) return 0;      // (no training-graph case; keeps the instance list short)
  return 1;
}

static int pw_cu_count() {
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
    n = 256;   // MI355X (also the answer on a build host without a GPU: plan sizes must not depend on where they are computed)
  return n;
}
static int pw_grid(const PwConfig* c, int64_t P) {
  static const int cus = pw_cu_count();
  const int lds = 2 * c->tp * c->cin * 2;
  int per_cu = c->nw == 4 ? 2 : 1;                                     // <= 256 registers per lane: 8 waves per CU
  if (lds * per_cu > 160 * 1024) per_cu = 1;
  const int64_t ntiles = P / c->tp;
  const int64_t g = (int64_t)cus * per_cu;
  return (int)(ntiles < g ? ntiles : g);
}

// columns of the [2][Cout][columns] table of fused sums = workgroups of the launch
extern "C" int stp_conv2d_pw_cols(const stp_conv_params* p) {
  const PwConfig* c = p ? pw_config(p->C0, p->Cout) : nullptr;
  return c ? pw_grid(c, (int64_t)p->N * p->Ho * p->Wo) : 0;
}

template <int CIN, int COUT, int NW, int TP, int EP, bool OPA, bool OPR>
static int pw_launch1(const PwArgs& a, int grid, hipStream_t s) {
  constexpr int lds = 2 * TP * CIN * 2;
  static bool attr_set = false;
  if (lds > 64 * 1024 && !attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_pw_kernel<CIN, COUT, NW, TP, EP, OPA, OPR>), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      return STP_E_LAUNCH;
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_pw_kernel<CIN, COUT, NW, TP, EP, OPA, OPR>), dim3(grid), dim3(NW * 64), lds, s, a);
  STP_LAUNCH_CHECK();
  return STP_OK;
}

template <int CIN, int COUT, int NW, int TP>
static int pw_launch(const stp_conv_params* p, const PwArgs& a, int grid, hipStream_t s) {
  const bool stats = p->stats_partial != nullptr, bnb = p->bnb_x != nullptr, acc = p->accumulate0 != 0, res = p->residual != nullptr;
  if (bnb) {
    if constexpr (pw_bnb_level(CIN, COUT) >= 2) { if (acc) return pw_launch1<CIN, COUT, NW, TP, PW_BNB, true, false>(a, grid, s); }
    if constexpr (pw_bnb_level(CIN, COUT) >= 1) { if (!acc) return pw_launch1<CIN, COUT, NW, TP, PW_BNB, false, false>(a, grid, s); }
    return STP_E_BADARG;
  }
  if (stats) return res ? pw_launch1<CIN, COUT, NW, TP, PW_STATS, false, true>(a, grid, s) : pw_launch1<CIN, COUT, NW, TP, PW_STATS, false, false>(a, grid, s);
  if (acc) return pw_launch1<CIN, COUT, NW, TP, PW_PLAIN, true, false>(a, grid, s);
  return res ? pw_launch1<CIN, COUT, NW, TP, PW_PLAIN, false, true>(a, grid, s) : pw_launch1<CIN, COUT, NW, TP, PW_PLAIN, false, false>(a, grid, s);
}

extern "C" int stp_conv2d_pw(const stp_conv_params* p, void* stream) {
  if (!p) return STP_E_BADARG;
  {
    stp_conv_params forced = *p;
    forced.tile = 800;                                   // (by tile id the kernel also serves when STP_PW=0 switched the automatic use off)
    if (!stp_conv2d_pw_eligible(&forced)) return STP_E_BADARG;
  }
  if (p->bnb_x && !p->stats_partial) return STP_E_BADARG;
  const PwConfig* c = pw_config(p->C0, p->Cout);
  const int64_t P = (int64_t)p->N * p->Ho * p->Wo;
  PwArgs a;
  a.x = (const char*)p->src0; a.w = (const char*)p->weight; a.dst = (char*)p->dst0;
  a.opr = (const char*)(p->bnb_x ? p->bnb_x : p->residual);
  a.bias = p->bias; a.stats = p->stats_partial;
  a.mean = p->bnb_mean; a.rstd = p->bnb_rstd; a.gamma = p->bnb_gamma; a.beta = p->bnb_beta; a.bnb_relu = p->bnb_relu;
  a.ntiles = (int)(P / c->tp);
  a.x_bytes = (uint32_t)(P * p->C0 * 2); a.o_bytes = (uint32_t)(P * p->Cout * 2);
  const int grid = pw_grid(c, P);
  const_cast<stp_conv_params*>(p)->stats_tiles = grid;
  hipStream_t s = (hipStream_t)stream;
#define PW_CASE(CI, CO, NW_, TP_) \
  if (c->cin == CI && c->cout == CO) return pw_launch<CI, CO, NW_, TP_>(p, a, grid, s);
  PW_CASE(64, 64, 4, 64) PW_CASE(64, 256, 4, 64) PW_CASE(256, 64, 4, 64) PW_CASE(256, 128, 4, 64) PW_CASE(128, 256, 8, 128)
  PW_CASE(128, 512, 8, 64) PW_CASE(512, 128, 4, 32) PW_CASE(256, 256, 8, 64) PW_CASE(512, 256, 8, 32) PW_CASE(256, 512, 8, 32)
  PW_CASE(64, 512, 8, 64)
#undef PW_CASE
  return STP_E_BADARG;
}
