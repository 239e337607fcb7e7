/*
 * stp_hip.h - C-ABI of the MI355X-native segmentation training hot path (libstp_hip.so).
 *
 * Boundary contract (last row): the reference has no native layer at all - its
 * hot loop is Keras `fit_generator` over TF/cuDNN ops reached from
 * segmentation_pipeline/segmentation.py:35 (GenericImageTaskConfig.fit) and :155
 * (`clazz(**cleaned)` -> segmentation_models.Unet graph).  Each entry point below replaces the
 * TF op family named in its comment; the Python host mirror that binds them with ctypes lives
 * in segmentation_training_pipeline_amd/_lib.py, and INTEGRATION.md shows the reference-side
 * stub.
 *
 * Rules for every entry point:
 *   - plain pointers / ints / floats / hipStream_t only (passed as void*); no C++ or torch types;
 *   - the caller owns every buffer (device memory, 16-byte aligned); nothing here allocates;
 *   - stream-ordered, never synchronises, safe to capture into a hipGraph;
 *   - returns 0 on success, <0 on error (STP_E_*); never throws;
 *   - re-entrant on distinct streams/workspaces.
 *
 * Tensors: activations NHWC dense; dtype 0 = fp32, 1 = bf16 (storage of activations and of the
 * compute copies of weights; accumulation is always fp32; master weights/grads/optimizer
 * state are always fp32).
 */
#ifndef STP_HIP_H
#define STP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STP_OK 0
#define STP_E_BADARG (-1)
#define STP_E_LAUNCH (-2)
#define STP_E_WORKSPACE (-3)

#define STP_F32 0
#define STP_BF16 1
/* STP_U8 (2) is defined with the BatchNormalization entry points below. */
#define STP_F16 3   /* IEEE half storage + v_mfma_*_f16 (BASELINE.json configs[3] "fp16 MFMA").  The 16-bit storage format is a
                       BUILD parameter of the kernel set: libstp_hip.so serves STP_F32 and STP_BF16, libstp_hip_f16.so - the
                       same sources compiled with -DSTP_STORAGE_F16=1, same symbols - serves STP_F32 and STP_F16.  Every
                       "dtype" argument documented as bf16 below means "the library's 16-bit format".  fp16 needs loss scaling:
                       the loss entry points' grad_scale and the optimizers' gscale carry it (host: HipSegModel loss_scale). */

/* How src0 is resampled into the virtual input plane of the implicit GEMM gather. */
#define STP_SRC_DIRECT 0     /* virtual (h,w) = src (h,w)                                        */
#define STP_SRC_NEAREST2X 1  /* UpSampling2D(2): virtual (h,w) reads src (h/2,w/2)               */
#define STP_SRC_ZEROINS2X 2  /* zero-insertion: virtual (h,w) reads src (h/2,w/2) iff both even  */
                             /* (data-gradient of a stride-2 convolution)                        */

int stp_abi_version(void);
/* The 16-bit dtype code this library was built for: STP_BF16 (libstp_hip.so) or STP_F16 (libstp_hip_f16.so). */
int stp_storage_dtype(void);

/* ----------------------------------------------------------------------------------------------
 * Convolution as implicit GEMM on MFMA.  Replaces keras.layers.Conv2D forward and the
 * Conv2DBackpropInput op of its gradient (reference call site segmentation.py:155 builds the
 * graph; schemas/segmentation.raml:158-178 names the decoder it produces).
 *
 * virtual input  V[n,h,w,c] = concat_c( resample(src0)[.., C0], src1[.., C1] ), h<Hv, w<Wv
 * out[n,ho,wo,co] = sum_{kh,kw,c} V[n, ho*stride - pad + kh, wo*stride - pad + kw, c]
 *                                 * weight[co][(kh*KW + kw)*(C0+C1) + c]
 *                   (+ bias[co]) (+ residual[n,ho,wo,co]) (ReLU if relu)
 * Output channels [0,Cd0) go to dst0 (row stride Cd0), [Cd0,Cout) to dst1 (row stride
 * Cout-Cd0); each destination either overwrites or accumulates (+=).
 * Constraints: C0, C1 multiples of 8 (bf16) / 4 (fp32), or C0 == 4 && C1 == 0 (stem, KW even);
 * Cd0 and Cout-Cd0 multiples of 4 unless Cout < 4 is handled by padding the weight rows
 * (Cout is then the number of valid rows; only those are stored).
 */
typedef struct stp_conv_params {
  const void* src0;     /* [N,Hs0,Ws0,C0]                       */
  const void* src1;     /* [N,Hv,Wv,C1] or NULL                 */
  const void* weight;   /* [Cout_pad][KH*KW*(C0+C1)] dtype      */
  const float* bias;    /* [Cout] or NULL                       */
  const void* residual; /* [N,Ho,Wo,Cout] dtype or NULL         */
  void* dst0;           /* [N,Ho,Wo,Cd0]                        */
  void* dst1;           /* [N,Ho,Wo,Cout-Cd0] or NULL           */
  int32_t N, Hs0, Ws0, Hv, Wv, C0, C1;
  int32_t src0_mode;    /* STP_SRC_*                            */
  int32_t KH, KW, stride, pad;
  int32_t Ho, Wo, Cout, Cd0;
  int32_t accumulate0, accumulate1, relu;
  int32_t dtype;        /* STP_F32 / STP_BF16                   */
  int32_t tile;         /* 0 = auto; else forces a tile config (testing/tuning) */
  int32_t stats_tiles;  /* OUT (written by stp_conv2d when stats_partial != NULL): number of pixel tiles */
  float* stats_partial; /* optional [2][Cout][pixel tiles] fp32: per-tile sum / sum of squares of the stored
                           output per channel (fused BatchNormalization statistics, finalize with
                           stp_bn_finalize); capacity >= stp_conv2d_stats_floats(p) floats */
  /* BatchNormalization-backward fusion for a data-gradient convolution whose destination is dY of a BN(+ReLU)
   * output and which COMPLETES that gradient - the only consumer of the BN output, or the last one to contribute
   * (accumulate0 = 1 on top of what the other consumers wrote): with bnb_x != NULL (the BN INPUT, [N,Ho,Wo,Cout] dtype)
   * the epilogue stores g = dY * [bn(x) > 0] (g = dY when bnb_relu == 0) instead of dY and writes
   * stats_partial = per-tile sum(g) / sum(g * xhat) per channel; finish with stp_bn_backward_fused.  Requires
   * stats_partial, Cd0 == Cout, Cout % 4 == 0, no relu / residual. */
  const void* bnb_x;
  const float* bnb_mean;
  const float* bnb_rstd;
  const float* bnb_gamma; /* NULL: scale=False */
  const float* bnb_beta;  /* NULL: center=False */
  int32_t bnb_relu;
  /* Gradient of UpSampling2D(2) folded into the epilogue of the data-gradient convolution whose (virtual) destination
   * is the upsampled tensor: with dst_sum2x2 != 0, dst0 is [N,Ho/2,Wo/2,Cout] and receives the sum of each 2x2 block
   * (the hi-res gradient is never written); bnb_x / stats_partial / accumulate0 then refer to that low-resolution
   * tensor.  Small-channel kernel (stp_conv2d_sc_eligible), Ho and Wo even, no bias / relu - and, for 64+ channels, the halo kernel
   * in its fused form (bnb_x set, Cd0 a multiple of the variant's channel tile: conv_halo.hip's EP 3, with or without dst1).
   * With a second destination (dst1 != NULL: the data gradient of conv3x3(concat(UpSampling2D(2)(x), skip)), served by the
   * wide-output kernel, stp_conv2d_scw_eligible, or by the halo kernel) only dst0 - the first Cd0 channels - is summed, and bnb_x / stats_partial
   * refer to those Cd0 channels; dst1 stays [N,Ho,Wo,Cout-Cd0]. */
  int32_t dst_sum2x2;
  /* stats_slots > 0 (a power of two <= 64): stats_partial points to PRE-ZEROED int64 fixed-point slots [2][Cout][stats_slots]
   * (2^-24 units) and every pixel tile ADDS its sums atomically into slot (tile % stats_slots) - integer addition, hence
   * order-independent and deterministic - instead of writing a per-tile partial: no finalize kernel; the consumer
   * (stp_bn_apply_slots / stp_bn_backward_slots) sums the few slots of each channel in its prologue. */
  int32_t stats_slots;
  /* Fused PRODUCER BatchNormalization (small-channel kernel and halo kernel only): src_bn_mean != NULL means src0 holds the tensor BEFORE a
   * training-phase BatchNormalization (+ activation src_bn_relu: 0 none, 1 ReLU, 2 ReLU6).  The kernel normalises it while
   * staging its halo tile, with the arithmetic and rounding of stp_bn_apply, so the result equals the unfused pair bit for
   * bit and the normalised tensor is never written to HBM.  Padding stays zero (it pads the NORMALISED tensor). */
  const float* src_bn_mean;
  const float* src_bn_rstd;
  const float* src_bn_gamma; /* may be NULL */
  const float* src_bn_beta;  /* may be NULL */
  int32_t src_bn_relu;
  /* Optional class-collapsed weights for a NEAREST-2x source (uniform-tap kernel, forward of the decoder convolutions over
   * UpSampling2D(2) + concat): for output parity class (py, px) the nine taps over the upsampled src0 read only 2 x 2 distinct
   * low-resolution pixels, so with weight_up = [Cout_pad][4 classes][2][2][C0] (stp_weight_prepare_upcollapse: the 3x3 taps that
   * share a pixel summed) the K loop takes 4 x C0 + 9 x C1 instead of 9 x (C0 + C1) steps.  NULL: the plain gather. */
  const void* weight_up;
  /* Data gradient of a 1x1 / stride-2 SHORTCUT folded into the data gradient of its sibling 3x3 / stride-2 / pad-1 convolution (both
   * read the same tensor, ResNet basic block with a projection shortcut): in the parity-class order of that launch the shortcut
   * touches only the (even, even) class, where it is one more tap - fold_src = dY of the shortcut ([N,Hs0,Ws0,fold_C], fold_C == C0),
   * fold_weight = its data-gradient weight copy [round_up(Cout,16)][fold_C] (rows = channels of the shared input).  The K loop of
   * that class runs fold_C / 64 more tiles; no separate launch reads and rewrites the whole gradient tensor.  Honoured only where
   * stp_conv2d_fold_ok(p) != 0 (STP_E_BADARG otherwise). */
  const void* fold_src;
  const void* fold_weight;
  int32_t fold_C;
  /* GROUP-LEVEL PRE-REDUCTION of the fused sums (round 5).  stats_group = G > 1: the pixel tiles of a channel tile are taken in groups
   * of G consecutive tiles; every workgroup publishes its column of stats_partial write-through and draws an arrival ticket from
   * stats_group_counters[channel tile x groups] (uint32, ZERO before the first launch; the launch leaves them at zero); the workgroup
   * that draws the last ticket of a group sums the group's columns IN TILE ORDER (fixed order: deterministic, replay bit-identical)
   * into stats_group_out = [2][C][ceil(tiles / G)].  The table the BatchNormalization reads then has <= 128 columns, which
   * stp_bn_finalize_apply / the one-launch form of stp_bn_backward_fused reduce in their prologue: no finalize launch on the chain
   * conv -> finalize -> apply -> conv.  G must be stp_conv2d_stats_group_for(p) (0 = this shape's kernel has no such epilogue). */
  float* stats_group_out;
  uint32_t* stats_group_counters;
  int32_t stats_group;
  /* SPACE-TO-DEPTH data gradient of a 3x3 / stride-2 / pad-1 convolution (+ its sibling 1x1 / stride-2 projection shortcut), round 5.
   * s2d_dgrad = 1 describes ONE dense launch for the four output parity classes: src0 = dY of the 3x3 layer [N,Ho,Wo,C0], src1 = dY of
   * the shortcut [N,Ho,Wo,C1] or NULL, KH = KW = 2, stride 1, pad 0 (tap (da, db) reads (a + da, b + db); past the bottom / right
   * edge = zero), Cout = 4 x Cq output channels in parity-class-major order (class cls = py * 2 + px), dst0 = [N,2Ho,2Wo,Cq]: channel
   * cls * Cq + c of pixel (a, b) is
   * stored as channel c of pixel (2a + (cls >> 1), 2b + (cls & 1)).  accumulate0, bnb_x (+ stats_partial = [2][Cq][4 x tiles]: four
   * column blocks per channel, stp_conv2d_stats_floats counts them) refer to that destination.  NO weight copy of its own: weight = the
   * 3x3 layer's ordinary data-gradient copy [Cq^16][3][3][C0] (stp_weight_prepare's `bwd`), fold_weight = the shortcut's [Cq^16][C1]
   * (required when C1 > 0) - the kernel addresses, per output row, the kernel tap its class meets under each of the 2 x 2 taps
   * (9 of the 16 (tap, class) blocks are live; the dead ones move no bytes and issue no MFMAs).  Served by the halo kernel only:
   * stp_conv2d_halo_variant(p) >= 0 tells. */
  int32_t s2d_dgrad;
} stp_conv_params;
/* Group size G (a power of two, >= 2) for which stp_conv2d(p) can pre-reduce its stats_partial columns to <= 128 (see stats_group), or 0
 * when the table is small enough already / the kernel that serves p has no such epilogue.  stp_conv2d_stats_group_counters(p, G) =
 * uint32 words stats_group_counters must hold. */
int stp_conv2d_stats_group_for(const stp_conv_params* p);
size_t stp_conv2d_stats_group_counters(const stp_conv_params* p, int32_t G);
/* 1: stp_conv2d(p) takes the parity-class path that honours fold_src / fold_weight / fold_C for this shape (p->fold_* need not be set). */
int stp_conv2d_fold_ok(const stp_conv_params* p);

int stp_conv2d(const stp_conv_params* p, void* stream);
/* weight_up of stp_conv_params from the fp32 master [Cout][3][3][C0 + C1]: rows = Cout rounded up to 16 (zero rows behind Cout),
 * class c = py * 2 + px, tap t = ty * 2 + tx; row taps of (py, ty): (0,0) {0}, (0,1) {1,2}, (1,0) {0,1}, (1,1) {2}; columns alike. */
int stp_weight_prepare_upcollapse(const float* master, void* weight_up, int32_t Cout, int32_t C0, int32_t C1, int32_t dtype,
                                  void* stream);
/* The same for several layers in ONE launch: desc_dev = nlayers records {const float* master; void* weight_up; int32 Cout, rows
 * (= Cout rounded up to 16), C0, C0 + C1} (stp_weight_prepare_upcollapse_desc_bytes() = 32 bytes each) in device memory. */
size_t stp_weight_prepare_upcollapse_desc_bytes(void);
int stp_weight_prepare_upcollapse_batched(const void* desc_dev, int32_t nlayers, int32_t dtype, void* stream);
/* Data gradient w.r.t. the low-resolution source of such a layer = a 4x4 / stride 2 / pad 1 convolution of dY (stp_conv2d, DIRECT
 * source): this builds its weights [round_up(C0, 16)][4][4][CoutB] from the master - row taps {2}, {1,2}, {0,1}, {0} of the 3x3
 * kernel summed, columns alike.  desc_dev: nlayers records {const float* master; void* out; int32 Cout, CoutB, C0, C0 + C1}. */
int stp_weight_prepare_upcollapse_bwd_batched(const void* desc_dev, int32_t nlayers, int32_t dtype, void* stream);
/* floats needed by stats_partial for this shape (tile choice included) */
size_t stp_conv2d_stats_floats(const stp_conv_params* p);
/* tile configuration stp_conv2d would pick for p (see conv_igemm.hip: 1..6 register-staged tiles, 32*STAGES+t
 * uniform-tap DMA tiles, 256+t per-lane-tap DMA tiles, 512 small-channel kernel, 768 stem kernel, 1024+v halo kernel); used by bench.py. */
int stp_conv2d_tile_for(const stp_conv_params* p);
/* Small-channel path (Cin <= 32, Cout <= 32, 3x3 stride 1 pad 1, single source, optional nearest-2x):
 * an 8x32 output tile per workgroup with the input halo tile staged once in LDS (conv_sc.hip).
 * stp_conv2d uses it automatically when stp_conv2d_sc_eligible(p) != 0 (tile id 512). */
int stp_conv2d_sc_eligible(const stp_conv_params* p);
/* Columns of the [stat][channel][column] partial sums the small-channel kernel writes for p (what stats_tiles will report):
 * one per 8x32 tile, or one per persistent workgroup in the streaming form. */
int stp_conv2d_sc_stats_tiles(const stp_conv_params* p);
int stp_conv2d_sc(const stp_conv_params* p, void* stream);
/* Wide-output form of the small-channel kernel (conv_sc.hip: conv_scw_stream_kernel): 3x3 / stride 1 / pad 1, 32 input channels,
 * 128 output channels split over dst0 (Cd0 channels, a multiple of 32, dst_sum2x2 = 1: low-resolution, optional bnb_x + stats_partial)
 * and dst1 (full resolution); 16-bit storage.  Replaces the generic per-tap data gradient of decoder_stage3_conv1 + stp_upsample2x_bwd
 * (TF: Conv2DBackpropInput + ResizeNearestNeighborGrad + FusedBatchNormGrad's reductions).  stp_conv2d uses it automatically
 * (tile id 640; STP_SCW=0 in the environment switches it off).  stats_tiles = stp_conv2d_scw_stats_tiles(p) columns of [2][Cd0][columns]. */
int stp_conv2d_scw_eligible(const stp_conv_params* p);
int stp_conv2d_scw_stats_tiles(const stp_conv_params* p);
int stp_conv2d_scw(const stp_conv_params* p, void* stream);
/* Narrow-output form (conv_sc.hip: conv_scn_stream_kernel): the FORWARD of Conv2D(32, 3x3)(Concatenate([UpSampling2D(2)(x), skip])) with 64 + 64
 * input channels (src0 = x at half resolution, src0_mode = STP_SRC_NEAREST2X, src1 = skip), 16-bit storage: both halos resident in LDS (x as its
 * low-resolution pixels), all weights in registers, one wave per SIMD.  Takes `weight` (the plain forward copy [32][3][3][128]); weight_up is
 * ignored.  bias / relu / accumulate0 / stats_partial as for stp_conv2d.  stp_conv2d uses it automatically (tile id 704; STP_SCN=0 switches it
 * off).  stats_tiles = stp_conv2d_scn_stats_tiles(p) columns. */
int stp_conv2d_scn_eligible(const stp_conv_params* p);
int stp_conv2d_scn_stats_tiles(const stp_conv_params* p);
int stp_conv2d_scn(const stp_conv_params* p, void* stream);
/* 64 -> 64 channels, 3x3 / stride 1 / pad 1, 16-bit storage, single source and destination (conv_sc.hip: conv_s64_stream_kernel; ResNet stage 1
 * forward and data gradient): the whole weight matrix in the registers of every wave, the halo double-buffered in LDS.  residual, or stats_partial,
 * or bnb_x + stats_partial; no bias / relu / accumulate.  OPT-IN (tile id 736, or STP_S64=1 in the environment for stp_conv2d's automatic choice
 * when the launch has at least two tiles of 8 x 32 pixels per CU): at the U-Net's batch the halo kernel is as fast (conv_sc.hip has the numbers).
 * stats_tiles = stp_conv2d_s64_stats_tiles(p) columns (one per workgroup). */
int stp_conv2d_s64_eligible(const stp_conv_params* p);
int stp_conv2d_s64_stats_tiles(const stp_conv_params* p);
int stp_conv2d_s64(const stp_conv_params* p, void* stream);
/* The ResNet stem (classification_models conv0: 7x7 / stride 2 / pad 3, 3+1 input channels -> 64, bf16): halo-tile kernel,
 * used by stp_conv2d automatically when eligible (tile id 768); optional fused BatchNormalization sums (stats_partial). */
int stp_conv2d_stem_eligible(const stp_conv_params* p);
int stp_conv2d_stem(const stp_conv_params* p, void* stream);
/* Halo-resident 3x3 kernel (conv_halo.hip; bf16, stride 1, pad 1, one direct source, Cin % 64 == 0, Cout % 16 == 0, Wo % 16 == 0,
 * Ho % 8 == 0): a workgroup keeps the (TH+2) x 18 input pixels of its TH x 16 output block resident in LDS per 64-channel slab and
 * streams only the weights.  Tile ids 1024 + variant (0: 16x16 px x 128 channels, 1: 8x16 x 128, 2: 16x16 x 64, 3: 8x16 x 64);
 * stp_conv2d picks it automatically where it measured faster (stp_conv2d_halo_variant(p) >= 0 with p->tile == 0; the environment
 * variable STP_HALO=0 disables the automatic choice).  It is the one GEMM-class kernel that honours src_bn_*: the producer
 * BatchNormalization (+activation) is applied in LDS, bit-identical to stp_bn_apply.  stp_conv2d_halo_tiles = pixel tiles
 * (= stats_partial columns) of a variant. */
int stp_conv2d_halo_variant(const stp_conv_params* p);
int stp_conv2d_halo_tiles(const stp_conv_params* p, int variant);
int stp_conv2d_halo(const stp_conv_params* p, int variant, void* stream);

/* ----------------------------------------------------------------------------------------------
 * Weight gradient (Conv2DBackpropFilter).  dW[co][(kh*KW+kw)*(C0+C1)+c] =
 *   sum_{n,ho,wo} dY[n,ho,wo,co] * V[n, ho*stride-pad+kh, wo*stride-pad+kw, c]
 * with V gathered exactly as in stp_conv2d.  Pixels are split over `splits` slabs written to
 * `workspace` ([splits][Cout][K] fp32) and reduced in fixed order (deterministic).
 * dW is fp32 [Cout][K]; accumulate != 0 adds to it.
 */
typedef struct stp_wgrad_params {
  const void* src0;
  const void* src1;
  const void* dy;       /* [N,Ho,Wo,Cout] dtype */
  float* dw;            /* [Cout][K] fp32       */
  int32_t N, Hs0, Ws0, Hv, Wv, C0, C1, src0_mode;
  int32_t KH, KW, stride, pad;
  int32_t Ho, Wo, Cout;
  int32_t accumulate;
  int32_t dtype;
  int32_t splits;       /* 0 = auto */
  /* as stp_conv_params.src_bn_*: src0 is the pre-normalisation tensor (small-channel and row-of-taps weight gradients - single-layer
   * and grouped launches -, C1 == 0 only) */
  const float* src_bn_mean;
  const float* src_bn_rstd;
  const float* src_bn_gamma;
  const float* src_bn_beta;
  int32_t src_bn_relu;
} stp_wgrad_params;

size_t stp_conv2d_wgrad_workspace_bytes(const stp_wgrad_params* p);
int stp_conv2d_wgrad(const stp_wgrad_params* p, void* workspace, size_t workspace_bytes, void* stream);
/* The two phases of stp_conv2d_wgrad as separate launches (so that a profiler / the plan can time them
 * individually).  variant: 0 = auto, 1 = register-staged kernel, 2 / 3 = buffer-DMA ring with 2 / 3 stages,
 * 4 = row-of-taps kernel (bf16, 3x3 / stride 1 / pad 1, C0 % 64 == 0 and C1 % 64 == 0, src0 direct or nearest-2x, Wo a multiple
 * of 64 or a power of two >= 16 with Ho * Wo % 64 == 0; STP_E_BADARG otherwise). */
int stp_conv2d_wgrad_partial(const stp_wgrad_params* p, void* workspace, size_t workspace_bytes, int32_t variant, void* stream);
int stp_conv2d_wgrad_reduce(const stp_wgrad_params* p, const void* workspace, int32_t variant, void* stream);
/* Small-channel weight gradient (conv_sc.hip): chosen automatically by variant 0 when eligible. */
int stp_wgrad_sc_eligible(const stp_wgrad_params* p);
int stp_wgrad_sc_slabs(const stp_wgrad_params* p);
int stp_wgrad_sc_partial(const stp_wgrad_params* p, void* workspace, void* stream);
/* Which kernel family variant 0 launches for p: 0 = pixel-reduction GEMM (conv_wgrad_kernel / conv_wgrad_dma_kernel),
 * 1 = small-channel halo kernel, 2 / 3 = row-of-taps kernel with 128 / 64 output channels per workgroup. */
int stp_conv2d_wgrad_kernel_id(const stp_wgrad_params* p);

/* Grouped weight gradient (row-of-taps kernel): ONE partial launch + ONE reduce launch for several layers (the 3x3 / stride-1
 * layers of a network stage).  A layer launched alone offers 6-96 output tiles to 512-768 workgroup slots, so its pixel reduction
 * is split 5-85 ways (~12 steps per workgroup, one 96 KB partial slab each); the group's (layer, tile, step) space is cut into one
 * contiguous chunk per workgroup slot instead (fixed partition -> deterministic), a tile is covered by 1-3 partial slabs.
 *   stp_wgrad_group_class        : 0 = the layer cannot join a group; otherwise the class (32 / 64 / 128 output channels per tile) -
 *                                  the layers of one group must share it
 *   stp_wgrad_group_table_bytes  : size of the descriptor table of a group (0: invalid group)
 *   stp_wgrad_group_workspace_bytes : size of the partial slabs
 *   stp_wgrad_group_build        : fills the HOST copy of the table (it holds the layers' device pointers: build it when src0 / src1 /
 *                                  dy / dw are final); the caller copies it to device memory
 *   stp_wgrad_group_partial / _reduce : the two launches; host_table (header read on the host) and its device copy
 * Layers may carry src_bn_* (C1 == 0, directly read src0): the group then runs the kernel instance that normalises those layers' halo
 * tiles in LDS (per-layer switch; header word 14 of the table says which instance the group uses). */
int stp_wgrad_group_class(const stp_wgrad_params* p);
size_t stp_wgrad_group_table_bytes(const stp_wgrad_params* const* layers, int32_t n);
size_t stp_wgrad_group_workspace_bytes(const stp_wgrad_params* const* layers, int32_t n);
int stp_wgrad_group_build(const stp_wgrad_params* const* layers, int32_t n, void* host_table, size_t table_bytes);
int stp_wgrad_group_partial(const void* host_table, const void* dev_table, void* workspace, size_t workspace_bytes, void* stream);
int stp_wgrad_group_reduce(const void* host_table, const void* dev_table, const void* workspace, void* stream);

/* ----------------------------------------------------------------------------------------------
 * Compute copies of a convolution kernel from the fp32 master (layout [Cout][KH][KW][Cin]):
 *   fwd : [Cout_pad16][KH][KWp][Cinp] dtype, zero padded (operand of stp_conv2d / layout of dW)
 *   bwd : [Cin_pad16][KH][KW][CoutB]  dtype, spatially flipped and transposed, zero padded
 *         (operand of the data-gradient GEMM: stp_conv2d over dY with pad' = K-1-pad)
 * Either pointer may be NULL.  stp_weight_grad_unpad copies a padded gradient
 * [Cout..][KH][KWp][Cinp] back to the master layout.  stp_stem_beta_grad: see loss_optim.hip.
 */
int stp_weight_prepare(const float* master, void* fwd, void* bwd, int32_t Cout, int32_t KH, int32_t KW,
                       int32_t Cin, int32_t KWp, int32_t Cinp, int32_t CoutB, int32_t dtype, void* stream);
/* All layers of a network in ONE launch: the caller fills a host array of opaque descriptors
 * (stp_weight_prepare_desc_bytes() each) with stp_weight_prepare_desc_fill - `start` = running sum of the
 * returned element counts - uploads it, and replays stp_weight_prepare_batched every step. */
size_t stp_weight_prepare_desc_bytes(void);
int64_t stp_weight_prepare_desc_fill(void* desc_host, int32_t index, int64_t start, const float* master, void* fwd, void* bwd,
                                     int32_t Cout, int32_t KH, int32_t KW, int32_t Cin, int32_t KWp, int32_t Cinp, int32_t CoutB);
int stp_weight_prepare_batched(const void* desc_dev, int32_t nlayers, int64_t total, int32_t dtype, void* stream);
int stp_weight_grad_unpad(const float* padded, float* grad, int32_t Cout, int32_t KH, int32_t KW,
                          int32_t Cin, int32_t KWp, int32_t Cinp, int32_t accumulate, void* stream);
int stp_stem_beta_grad(const float* padded_dw, const float* master, float* dbeta, int32_t Cout, int32_t KH,
                       int32_t KW, int32_t Cin, int32_t KWp, int32_t Cinp, int32_t one_ch, void* stream);

/* ----------------------------------------------------------------------------------------------
 * BatchNormalization (Keras semantics: biased batch variance for the normalisation, unbiased
 * for the moving variance, momentum on the moving stats).  Replaces keras BatchNormalization /
 * tf.nn.fused_batch_norm + ReLU, forward and gradient.
 *   stp_bn_stats     : per-channel mean and rstd (= 1/sqrt(var+eps)) over `rows` = N*H*W, plus the
 *                      moving-stat update (either may be NULL).  workspace: stp_bn_workspace_bytes.
 *   stp_bn_apply     : y = (x-mean)*rstd*gamma + beta, optional ReLU.  For uint8 input
 *                      (xdtype = STP_U8, C <= 4, the raw image) Cy must be 4 and the padded
 *                      channels are written as pad_value; otherwise Cy == C and xdtype == ydtype.
 *   stp_bn_inference : same with rstd derived from the moving variance.
 *   stp_bn_backward  : dy is the gradient w.r.t. the (post-ReLU) output; the ReLU mask is
 *                      re-derived from x (same single-fma affine as the forward).  Produces dx
 *                      (optionally accumulating) and the raw sums dgamma / dbeta (may be NULL).
 * gamma and beta may be NULL (scale=False / center=False).
 */
#define STP_U8 2
size_t stp_bn_workspace_bytes(int32_t C);
/* The slot forms (stp_conv_params.stats_slots): statistics -> mean / rstd (published for the backward pass), moving statistics
 * and the normalisation + activation in ONE kernel; and the BatchNormalization backward from bnb slots in ONE kernel. */
int stp_bn_apply_slots(const void* x, void* y, int32_t dtype, int64_t rows, int32_t C, const int64_t* slots, int32_t nslots,
                       float eps, float momentum, float* mean, float* rstd, float* moving_mean, float* moving_var,
                       const float* gamma, const float* beta, int32_t relu, void* stream);
int stp_bn_backward_slots(const void* x, const void* g, void* dx, int32_t dtype, int64_t rows, int32_t C, const float* mean,
                          const float* rstd, const float* gamma, const int64_t* slots, int32_t nslots, float* dgamma,
                          float* dbeta, int32_t accumulate_dx, void* stream);
/* zero-fill (the slot arena, once per step); p and bytes 16-byte aligned */
int stp_zero_bytes(void* p, int64_t bytes, void* stream);
/* second half of stp_bn_backward when the partial sums came from a convolution epilogue (stp_conv_params.bnb_x):
 * g is the masked gradient that epilogue stored, partial its [2][C][tiles] sums. */
int stp_bn_backward_fused(const void* x, const void* g, void* dx, int32_t dtype, int64_t rows, int32_t C,
                          const float* mean, const float* rstd, const float* gamma, const float* partial, int32_t tiles,
                          float* dgamma, float* dbeta, int32_t accumulate_dx, void* workspace, size_t workspace_bytes,
                          void* stream);
/* stp_bn_finalize + stp_bn_apply as ONE launch (16-bit dtype, C % 64 == 0, tiles <= 128; stp_bn_finalize_apply_ok tells): every
 * workgroup of the apply pass owns a 64-channel slab x a chunk of rows and reduces the [2][64][tiles] partial sums of its slab
 * itself (fp64, fixed order - all workgroups of a slab normalise with bit-identical constants), the first chunk of a slab also
 * writes mean / rstd / the moving statistics.  stp_bn_backward_fused(_add) takes the same one-launch form where eligible. */
int stp_bn_finalize_apply_ok(int32_t dtype, int64_t rows, int32_t C, int32_t tiles);
int stp_bn_finalize_apply(const float* partial, int32_t tiles, const void* x, void* y, int32_t dtype, int64_t rows, int32_t C,
                          float eps, float momentum, float* mean, float* rstd, float* moving_mean, float* moving_var,
                          const float* gamma, const float* beta, int32_t relu, void* stream);
/* the same with the accumulated addend in ANOTHER buffer: dx = result + dadd when accumulate_dx (dadd is left intact - a residual
 * gradient that a grouped weight gradient, stp_wgrad_group_partial, still reads as its dY); dadd == dx is the in-place form above */
int stp_bn_backward_fused_add(const void* x, const void* g, void* dx, const void* dadd, int32_t dtype, int64_t rows, int32_t C,
                              const float* mean, const float* rstd, const float* gamma, const float* partial, int32_t tiles,
                              float* dgamma, float* dbeta, int32_t accumulate_dx, void* workspace, size_t workspace_bytes,
                              void* stream);
int stp_bn_stats(const void* x, int32_t xdtype, int64_t rows, int32_t C, float eps, float momentum,
                 float* mean, float* rstd, float* moving_mean, float* moving_var,
                 void* workspace, size_t workspace_bytes, void* stream);
/* second half of stp_bn_stats for statistics produced by a convolution epilogue (stats_partial) */
int stp_bn_finalize(const float* partial, int32_t tiles, int64_t rows, int32_t C, float eps, float momentum,
                    float* mean, float* rstd, float* moving_mean, float* moving_var, void* stream);
int stp_bn_apply(const void* x, int32_t xdtype, void* y, int32_t ydtype, int64_t rows, int32_t C, int32_t Cy,
                 const float* mean, const float* rstd, const float* gamma, const float* beta,
                 int32_t relu, float pad_value, void* stream);
int stp_bn_inference(const void* x, int32_t xdtype, void* y, int32_t ydtype, int64_t rows, int32_t C, int32_t Cy,
                     const float* moving_mean, const float* moving_var, float eps,
                     const float* gamma, const float* beta, int32_t relu, float pad_value, void* stream);
int stp_bn_backward(const void* x, const void* dy, void* dx, int32_t dtype, int64_t rows, int32_t C,
                    const float* mean, const float* rstd, const float* gamma, const float* beta,
                    float* dgamma, float* dbeta, int32_t relu, int32_t accumulate_dx,
                    void* workspace, size_t workspace_bytes, void* stream);

/* ZeroPadding2D(1) + MaxPooling2D(3, strides 2, 'valid') (classification_models ResNet stem).
 * idx [N,Ho,Wo,C] uint8 records kh*3+kw of the first maximum (padded taps compete with value 0);
 * the gradient routes dy through it (TF MaxPoolGrad behaviour). */
int stp_maxpool3x3s2(const void* x, void* y, uint8_t* idx, int32_t N, int32_t H, int32_t W, int32_t C,
                     int32_t dtype, void* stream);
int stp_maxpool3x3s2_bwd(const uint8_t* idx, const void* dy, void* dx, int32_t N, int32_t H, int32_t W, int32_t C,
                         int32_t dtype, int32_t accumulate, void* stream);
/* stp_bn_apply (+ activation) and the stp_maxpool3x3s2 that follows it as ONE launch (the ResNet stem: bn0 -> relu0 -> pooling0): x = the
 * tensor before the BatchNormalization, yb = the normalised tensor (written, bit-identical to stp_bn_apply's - other layers read it),
 * y / idx = the pooling outputs.  16-bit storage, H and W even, C % 8 == 0. */
int stp_bn_apply_maxpool3x3s2(const void* x, void* yb, void* y, uint8_t* idx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t dtype,
                              const float* mean, const float* rstd, const float* gamma, const float* beta, int32_t relu, void* stream);
/* The same when dx is the gradient of a BatchNormalization(+activation) output that this launch completes (bn0 of the ResNet
 * stem: pooled, and read by a decoder skip): mask + partial sums for stp_bn_backward_fused, as stp_upsample2x_bwd_bn. */
int stp_maxpool3x3s2_bwd_bn_tiles(int32_t N, int32_t H, int32_t W, int32_t C, int32_t dtype);
int stp_maxpool3x3s2_bwd_bn(const uint8_t* idx, const void* dy, void* dx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t dtype,
                            int32_t accumulate, const void* bn_x, const float* mean, const float* rstd, const float* gamma,
                            const float* beta, int32_t relu, float* partial, void* stream);
/* MaxPooling2D(2, 2), no padding (keras.applications VGG blocks; H and W even).  idx = 2*dy+dx of the first maximum. */
int stp_maxpool2x2(const void* x, void* y, uint8_t* idx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t dtype, void* stream);
int stp_maxpool2x2_bwd(const uint8_t* idx, const void* dy, void* dx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t dtype,
                       int32_t accumulate, void* stream);
/* AveragePooling2D(pool = strides = k), H % k == W % k == 0 (PSPNet pyramid pooling) and its gradient dx = dy / k^2.
 * Large windows over few outputs are reduced in two fixed-order stages through `workspace` (fp32 partials; size from the
 * query, 0 = not needed); without it the single-stage kernel runs. */
size_t stp_avgpool_workspace_bytes(int32_t N, int32_t H, int32_t W, int32_t C, int32_t k);
int stp_avgpool(const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t k, int32_t dtype, void* workspace,
                size_t workspace_bytes, void* stream);
int stp_avgpool_bwd(const void* dy, void* dx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t k, int32_t dtype,
                    int32_t accumulate, void* stream);
/* PSPNet's pyramid pooling (segmentation_models 0.2.1 psp builder: AveragePooling2D(size / level) of ONE feature map for levels 1, 2, 3, 6;
 * schemas/segmentation.raml:225-249) in one pass over the feature map: up to four levels y_i [N][H / k_i][W / k_i][C], k0 the FINEST window
 * (smallest k), every other k_i a multiple of k0 (k_i = 0: level unused) - the means are formed from the fp32 sums of the finest windows, one
 * rounding per output as stp_avgpool.  _bwd: dx (+)= sum_i dy_i / k_i^2 in one pass.  _ok: 1 if the pyramid form serves the shape
 * (STP_POOL_PYRAMID=0: never); workspace >= stp_avgpool_pyramid_workspace_bytes(). */
int stp_avgpool_pyramid_ok(int32_t N, int32_t H, int32_t W, int32_t C, int32_t k0, int32_t k1, int32_t k2, int32_t k3, int32_t dtype);
size_t stp_avgpool_pyramid_workspace_bytes(int32_t N, int32_t H, int32_t W, int32_t C, int32_t k0, int32_t dtype);
int stp_avgpool_pyramid(const void* x, void* y0, void* y1, void* y2, void* y3, int32_t k0, int32_t k1, int32_t k2, int32_t k3, int32_t N,
                        int32_t H, int32_t W, int32_t C, int32_t dtype, void* workspace, size_t workspace_bytes, void* stream);
int stp_avgpool_pyramid_bwd(const void* dy0, const void* dy1, const void* dy2, const void* dy3, int32_t k0, int32_t k1, int32_t k2, int32_t k3,
                            void* dx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t dtype, int32_t accumulate, void* stream);
/* MaxPooling2D(pool_size = strides = k), any k dividing H and W (PSPNet `psp_pooling_type: max`): idx[N,H/k,W/k,C] int32 = position of the
 * first maximum inside its window (NULL = not wanted); the gradient goes to that position. */
int stp_maxpool_k(const void* x, void* y, int32_t* idx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t k, int32_t dtype, void* stream);
int stp_maxpool_k_bwd(const int32_t* idx, const void* dy, void* dx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t k, int32_t dtype,
                      int32_t accumulate, void* stream);
/* dy <- dy * [y > 0] in place (gradient of a ReLU fused into a convolution epilogue); count % 4 == 0 */
int stp_relu_bwd(const void* y, void* dy, int64_t count, int32_t dtype, void* stream);

/* Gradient of UpSampling2D(2): dx[n,h,w,c] (+)= sum of the 2x2 block of dy ([N,2H,2W,ldy]). */
int stp_upsample2x_bwd(const void* dy, void* dx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t ldy,
                       int32_t dtype, int32_t accumulate, void* stream);
/* The same when dx is the gradient of a BatchNormalization(+activation) output that this launch completes: the stored value
 * is masked with the activation re-derived from the BN input bn_x and `partial` ([2][C][tiles], tiles from the query; 0 =
 * channel count not supported) receives the sums stp_bn_backward_fused consumes - see stp_conv_params.bnb_x. */
/* Second half of the data gradient of a 1x1 / stride-2 convolution (reference: the projection shortcut `sc` of
 * classification_models' bottleneck ResNets, reached through segmentation.py:109-118): t = W^T dY is computed at LOW resolution by a plain
 * 1x1 / stride-1 stp_conv2d; stp_scatter2x_bwd puts t[n, a, b] at (2a, 2b) of dx [N, H, W, C] (zeros elsewhere; accumulate != 0: dx +=).
 * The _bn form completes the gradient of a BatchNormalization(+activation) output as stp_maxpool3x3s2_bwd_bn does: masked in place,
 * partial[2][C][stp_scatter2x_bwd_bn_tiles] for stp_bn_backward_fused. */
/* Conv2D(3x3, padding 1) with few output channels over many input channels (segmentation_models' class heads `final_conv` of FPN / PSPNet,
 * reached through segmentation.py:109-118) as a 1x1 stp_conv2d into 9 x Cout tap channels (the 3x3 kernel [Cout][3][3][Cin] read as the
 * matrix [9 Cout][Cin]: tap channel (o, t) = o * 9 + t) followed by stp_tapsum_fwd:
 *   y[n, h, w, o] = bias[o] + sum_t z[n, h + t / 3 - 1, w + t % 3 - 1, o * 9 + t]   (taps outside the image contribute zero)
 * stp_tapsum_bwd is its adjoint: dz[n, h, w, o * 9 + t] = dy[n, h - (t / 3 - 1), w - (t % 3 - 1), o]; the padded channels of dz are zeroed. */
int stp_tapsum_fwd(const void* z, void* y, const float* bias, int32_t N, int32_t H, int32_t W, int32_t Cout, int32_t Zc, int32_t Cy, int32_t dtype,
                   void* stream);
int stp_tapsum_bwd(const void* dy, void* dz, int32_t N, int32_t H, int32_t W, int32_t Cout, int32_t Cdy, int32_t Cdz, int32_t dtype, void* stream);
int stp_scatter2x_bwd(const void* t, void* dx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t dtype, int32_t accumulate, void* stream);
int stp_scatter2x_bwd_bn_tiles(int32_t N, int32_t H, int32_t W, int32_t C, int32_t dtype);
int stp_scatter2x_bwd_bn(const void* t, void* dx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t dtype, int32_t accumulate,
                         const void* bn_x, const float* mean, const float* rstd, const float* gamma, const float* beta, int32_t relu,
                         float* partial, void* stream);
int stp_upsample2x_bwd_bn_tiles(int32_t N, int32_t H, int32_t W, int32_t C, int32_t ldy, int32_t dtype);
int stp_upsample2x_bwd_bn(const void* dy, void* dx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t ldy, int32_t dtype,
                          int32_t accumulate, const void* bn_x, const float* mean, const float* rstd, const float* gamma,
                          const float* beta, int32_t relu, float* partial, void* stream);
/* x[n,2h+i,2w+j,c] += m[n,h,w,c] in place (FPN: Add()([lateral, UpSampling2D(2)(m)])); H, W = size of x, even. */
int stp_upsample2x_add(void* x, const void* m, int32_t N, int32_t H, int32_t W, int32_t C, int32_t dtype, void* stream);
/* tf.image.resize_bilinear(align_corners=False) of TF 1.x by an integer factor (src = dst / factor, no half-pixel offset) -
 * Keras 2.2.4's K.resize_images(interpolation='bilinear').  x [N,H,W,C] -> channels [coff, coff+C) of y [N,H*f,W*f,ldo]
 * (Concatenate of resized maps without a copy; factor 1 = strided copy).  The gradient is a fixed-order gather: one
 * workgroup per input pixel over the (2f)^2 outputs that may read it, split further through `workspace` (fp32 partials,
 * size from the query, 0 = not needed) when a few pixels collect a whole map (PSPNet level 1: 1x1 -> 96x96). */
int stp_resize_bilinear(const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t factor, int32_t ldo,
                        int32_t coff, int32_t dtype, void* stream);
size_t stp_resize_bilinear_bwd_workspace_bytes(int32_t N, int32_t H, int32_t W, int32_t C, int32_t factor);
int stp_resize_bilinear_bwd(const void* dy, void* dx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t factor, int32_t ldo,
                            int32_t coff, int32_t dtype, int32_t accumulate, void* workspace, size_t workspace_bytes, void* stream);
/* ResizeImage(factor, interpolation='nearest') (= UpSampling2D(factor)) with the same slice addressing, and its gradient (the sum of
 * the factor x factor outputs of every input pixel). */
int stp_resize_nearest(const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t factor, int32_t ldo, int32_t coff,
                       int32_t dtype, void* stream);
int stp_resize_nearest_bwd(const void* dy, void* dx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t factor, int32_t ldo, int32_t coff,
                           int32_t dtype, int32_t accumulate, void* stream);

/* Bias gradient: out[c] (+)= sum over rows of x[rows][C]; and the in-place tensor add used where two
 * gradient paths meet outside a GEMM epilogue.  workspace as for stp_bn_stats. */
int stp_channel_sum(const void* x, int32_t dtype, int64_t rows, int32_t C, float* out, int32_t accumulate,
                    void* workspace, size_t workspace_bytes, void* stream);
int stp_add_inplace(void* dst, const void* src, int64_t count, int32_t dtype, void* stream);

/* ----------------------------------------------------------------------------------------------
 * Segmentation head + loss.  logits -> sigmoid -> w_bce*binary_crossentropy + w_dice*dice_loss
 * (Keras TF-backend BCE with the 1e-7 probability clip; musket dice_coef_loss, smooth 1), the
 * metrics (dice at 0.5, binary_accuracy) and dL/dlogits (channel 0 of a [count][dl_channels]
 * tensor, other channels zero, so it can feed the 8-channel-granular gather of the GEMMs).
 * scalars (fp32[10]): loss, bce, dice_loss, dice_metric, binary_accuracy, sum_p, sum_y, sum_py, iou, iot
 *   (iou = (sum_py+1)/(sum_y+sum_p-sum_py+1); iot = the same with p thresholded at 0.5).
 * Two-stage, fixed-order reduction (deterministic).
 */
size_t stp_loss_workspace_bytes(void);
int stp_sigmoid_bce_dice(const void* logits, const uint8_t* target, int64_t count, int32_t dtype,
                         float w_bce, float w_dice, float* scalars, void* dlogits, int32_t dl_channels,
                         float grad_scale, void* workspace, size_t workspace_bytes, void* stream);
/* The rest of the loss registry of segmentation.py:15-22 for the sigmoid head: weights5 (HOST pointer, read at launch)
 * = weights of {binary_crossentropy, dice_loss, iou_loss, jaccard_loss, focal_loss} in the composite `a+w*b` spec
 * (README.md:210-214).  iou_loss = 1 - iou_coef (smooth 1); jaccard_loss = jaccard_distance_loss (smooth 100, per pixel,
 * mean); focal_loss = binary focal loss (gamma 2, alpha 0.25, 1e-7 clip, mean).  scalars (fp32[12]) = the ten of
 * stp_sigmoid_bce_dice + [10] jaccard_loss, [11] focal_loss. */
int stp_sigmoid_loss_ex(const void* logits, const uint8_t* target, int64_t count, int32_t dtype, const float* weights5,
                        float* scalars, void* dlogits, int32_t dl_channels, float grad_scale, void* workspace,
                        size_t workspace_bytes, void* stream);
/* lovasz_loss (segmentation.py:18; README.md:379,411): binary Lovasz hinge per image on the Keras-recovered logits, mean over
 * images.  Runs AFTER stp_sigmoid_bce_dice / stp_sigmoid_loss_ex on the same scalars / dlogits: scalars[12] = lovasz_loss,
 * scalars[0] += weight * lovasz_loss, dlogits channel 0 += weight * gradient (scalars is fp32[16] here).  The workspace holds the
 * sort keys and rocPRIM's temporary storage: stp_lovasz_workspace_bytes(images * per_image, images) (0 = not available). */
size_t stp_lovasz_workspace_bytes(int64_t count, int32_t images);
int stp_lovasz_hinge(const void* logits, const uint8_t* target, int32_t images, int64_t per_image, int32_t dtype, float weight,
                     float* scalars, void* dlogits, int32_t dl_channels, void* workspace, size_t workspace_bytes, void* stream);
/* Bias gradient of the class convolution (= sum of dL/dlogit): stp_sigmoid_bce_dice / stp_sigmoid_loss_ex leave one partial sum
 * per gradient workgroup in their workspace; this adds them up in a fixed order into dbias[0] (instead of a stp_channel_sum pass
 * over the padded gradient tensor).  Same workspace and count as the loss call that wrote dlogits. */
int stp_sigmoid_loss_bias_grad(const void* workspace, int64_t count, float* dbias, int32_t accumulate, void* stream);
int stp_sigmoid(const void* logits, float* probs, int64_t count, int32_t dtype, void* stream);
/* Multi-class head (activation: softmax, loss: categorical_crossentropy[+w*dice_loss]; schemas/segmentation.raml:12-21,
 * 62-63): channel softmax over the first `classes` (2..32) channels of logits [pixels][ldc], target = uint8 class index
 * per pixel.  scalars as stp_sigmoid_bce_dice with [1] = categorical_crossentropy and the sums taken over every
 * (pixel, class) element of the one-hot target; dlogits [pixels][dl_channels] gets classes gradients + zero padding. */
int stp_softmax_cce_dice(const void* logits, const uint8_t* target, int64_t pixels, int32_t classes, int32_t ldc,
                         int32_t dtype, float w_cce, float w_dice, float* scalars, void* dlogits, int32_t dl_channels,
                         float grad_scale, void* workspace, size_t workspace_bytes, void* stream);
/* The same loss on class logits the network produces at 1 / factor of the mask's resolution and resizes bilinearly (segmentation_models
 * 0.2.1: PSPNet `final_interpolation: bilinear` x downsample_factor, schemas/segmentation.raml:225-249; FPN's last UpSampling2D(4,
 * 'bilinear')) - replaces, in the training step, stp_resize_bilinear of the logits + stp_softmax_cce_dice + stp_scale_by_device +
 * stp_resize_bilinear_bwd of their gradient WITHOUT the resized tensors: low [N][H][W][ldc], target [N][H factor][W factor] (4-byte
 * aligned), dlow [N][H][W][dl_channels] = d loss / d low (classes gradients + zero padding; NULL: scalars only), rounded to the storage
 * type where the unfused chain rounds (the resized logit, the per-pixel gradient).  dev_scale (may be NULL): device multiplier applied to
 * dlow (the dynamic loss scale, stp_scale_by_device), recorded into dev_record[0].  factor 2 / 4 / 8 / 16; workspace >=
 * stp_loss_workspace_bytes(); corners >= stp_softmax_cce_dice_up_corner_bytes() (16-byte aligned; only with dlow).  _ok: 1 if the
 * fused form serves (factor, classes) (STP_UP_LOSS=0: never). */
int stp_softmax_cce_dice_up_ok(int32_t factor, int32_t classes, int32_t dtype);
size_t stp_softmax_cce_dice_up_corner_bytes(int32_t N, int32_t H, int32_t W, int32_t classes);
int stp_softmax_cce_dice_up(const void* low, const uint8_t* target, int32_t N, int32_t H, int32_t W, int32_t factor, int32_t classes,
                            int32_t ldc, int32_t dtype, float w_cce, float w_dice, float* scalars, void* dlow, int32_t dl_channels,
                            float grad_scale, const float* dev_scale, float* dev_record, void* workspace, size_t workspace_bytes,
                            void* corners, size_t corner_bytes, void* stream);
/* probs [pixels][classes] fp32 = softmax of the first `classes` channels */
int stp_softmax(const void* logits, float* probs, int64_t pixels, int32_t classes, int32_t ldc, int32_t dtype, void* stream);

/* ----------------------------------------------------------------------------------------------
 * Optimizers over a flat fp32 arena (Keras 2.2.4 update rules, schemas/segmentation.raml:77-89).
 * `state` is a device int32[2]: [0] = iteration counter (incremented by the launch, so a captured
 * graph advances), [1] = scratch for the bias-corrected step size.  count must be a multiple of 4.
 * lr lives in device memory (lr[0]) so callbacks can change it without re-capturing.
 * mask (uint8, optional) freezes elements where mask[i] == 0 (freeze_encoder).
 * gscale (device float, optional) multiplies the gradient (clipnorm / 1/world_size).
 */
int stp_adam(float* param, const float* grad, float* m, float* v, int64_t count, const float* lr,
             float beta1, float beta2, float eps, int32_t* state, const uint8_t* mask, const float* gscale,
             float clipvalue, void* stream);
int stp_sgd(float* param, const float* grad, float* vel, int64_t count, const float* lr, float momentum,
            int32_t nesterov, const uint8_t* mask, const float* gscale, float clipvalue, void* stream);
/* RMSprop (Keras 2.2.4): a <- rho a + (1-rho) g^2 ; p <- p - lr g / (sqrt(a) + eps).  Same mask / gscale / clipvalue
 * conventions as stp_adam. */
int stp_rmsprop(float* param, const float* grad, float* acc, int64_t count, const float* lr, float rho, float eps,
                const uint8_t* mask, const float* gscale, float clipvalue, void* stream);
/* Nadam (Keras 2.2.4, schedule_decay form; lr default 0.002).  state: int32[2] as stp_adam (state[0] = iteration);
 * fstate: float[8] device scratch whose element 0 is m_schedule and MUST be initialised to 1.0. */
int stp_nadam(float* param, const float* grad, float* m, float* v, int64_t count, const float* lr, float beta1, float beta2,
              float eps, float schedule_decay, int32_t* state, float* fstate, const uint8_t* mask, const float* gscale,
              float clipvalue, void* stream);
/* gscale[0] = min(1, clipnorm / ||base*grad||_2) * base  (base = 1/world_size for summed data-parallel
 * gradients, times 1/loss_scale in fp16 mode; clipnorm <= 0: no clipping; deterministic two-stage reduction).  workspace >= 4 KiB.
 * OVERFLOW GUARD: gscale is float[2]; when the squared norm is not finite (an inf / NaN anywhere in the arena - fp16 overflow
 * under loss scaling) gscale[0] = -1 and gscale[1] is incremented, and every optimizer entry point (stp_adam, stp_sgd,
 * stp_rmsprop, stp_nadam) given that gscale returns without touching parameters, moments or its step counter: the step is skipped. */
int stp_grad_global_scale(const float* grad, int64_t count, float clipnorm, float base, float* gscale,
                          void* workspace, size_t workspace_bytes, void* stream);
/* DYNAMIC LOSS SCALING (fp16 storage; replaces a Keras LossScaleOptimizer / torch GradScaler around the reference's fit()).
 * dls = float[8] on the device: [0] multiplier m of the next backward pass (on top of the static scale of the loss kernels),
 * [1] clean steps, [2] growth interval, [3] smallest m, [4] m of the gradients now in the arena, [5] largest m.
 * stp_scale_by_device: x *= scalar[0] (the loss gradient right after the loss kernel seeded it; record = dls + 4).  stp_grad_global_scale_dls: as
 * stp_grad_global_scale with base / m; a skipped step halves m, `interval` clean steps double it - all on the device (graph-safe). */
int stp_scale_by_device(void* x, int64_t count, int32_t dtype, const float* scalar, float* record /* may be NULL: receives scalar[0] */, void* stream);
int stp_grad_global_scale_dls(const float* grad, int64_t count, float clipnorm, float base, float* gscale, float* dls,
                              void* workspace, size_t workspace_bytes, void* stream);

/* ----------------------------------------------------------------------------------------------
 * DeepLabV3+ (the in-tree model, segmentation_pipeline/impl/deeplab/model.py) - the ops nothing else needs.
 *
 * Depthwise convolution (DepthwiseConv2D, model.py:136, 255-259): x [N,H,W,C] dtype, w fp32 [k][k][C] (Keras'
 * (kh,kw,C,1) kernel as stored), explicit top/left padding (TF 'same' is bottom/right heavy), dilation; C % 4 == 0.
 * The weight gradient needs stp_dwconv_wgrad_workspace_bytes(C, k) of workspace (two-stage fixed-order reduction). */
int stp_dwconv(const void* x, const float* w, void* y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t k, int32_t stride,
               int32_t pad_t, int32_t pad_l, int32_t dilation, int32_t Ho, int32_t Wo, int32_t dtype, void* stream);
int stp_dwconv_dgrad(const void* dy, const float* w, void* dx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t k,
                     int32_t stride, int32_t pad_t, int32_t pad_l, int32_t dilation, int32_t Ho, int32_t Wo, int32_t dtype,
                     int32_t accumulate, void* stream);
size_t stp_dwconv_wgrad_workspace_bytes(int32_t C, int32_t k);
int stp_dwconv_wgrad(const void* x, const void* dy, float* dw, int32_t N, int32_t H, int32_t W, int32_t C, int32_t k,
                     int32_t stride, int32_t pad_t, int32_t pad_l, int32_t dilation, int32_t Ho, int32_t Wo, int32_t dtype,
                     int32_t accumulate, void* workspace, size_t workspace_bytes, void* stream);
/* tf.image.resize_bilinear(align_corners=True) to any output size (BilinearUpsampling, model.py:94-100) and its gradient
 * (fixed-order gather).  Dense [N,H,W,C] tensors, any C. */
int stp_resize_bilinear_ac(const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t Ho, int32_t Wo,
                           int32_t dtype, void* stream);
int stp_resize_bilinear_ac_bwd(const void* dy, void* dx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t Ho, int32_t Wo,
                               int32_t dtype, int32_t accumulate, void* stream);
/* Inverted dropout (model.py:461): y = mask * x / (1 - rate), mask = counter-based hash of (state[0], salt, index) - the same
 * call with dy gives the gradient; stp_counter_tick advances state[0] (int32 on the device) once per step, so a hipGraph
 * replay draws a fresh mask.  x == y allowed. */
int stp_counter_tick(int32_t* state, void* stream);
int stp_dropout(const void* x, void* y, int64_t count, float rate, const int32_t* state, uint32_t salt, int32_t dtype, void* stream);
/* SpatialDropout2D(rate) on [N][HW][C] (segmentation_models' FPN / PSPNet `dropout`, schemas/segmentation.raml:201-203, 241-243):
 * one keep / drop decision per (sample, channel), mask index n * C + c, same hash, scaling and step counter as stp_dropout. */
int stp_dropout_spatial(const void* x, void* y, int32_t N, int64_t HW, int32_t C, float rate, const int32_t* state, uint32_t salt,
                        int32_t dtype, void* stream);
/* Activation('sigmoid') carried by the last convolution (model.py:485) as a tensor op on the first `channels` columns of
 * [rows][ld] tensors, and dz = dp * p * (1 - p) over [rows][ldg] gradients (padding columns written as 0). */
int stp_sigmoid_act(const void* z, void* p, int64_t rows, int32_t channels, int32_t ldz, int32_t ldp, int32_t dtype, void* stream);
int stp_sigmoid_act_bwd(const void* p, const void* dp, void* dz, int64_t rows, int32_t channels, int32_t ldp, int32_t ldg,
                        int32_t dtype, void* stream);
/* Loss on PROBABILITIES (the model upsamples sigmoid outputs, model.py:485-486): same scalars as stp_sigmoid_bce_dice,
 * gradient w.r.t. the probabilities into column 0 of dprobs [count][dl_channels].  workspace >= stp_loss_workspace_bytes(). */
int stp_prob_bce_dice(const void* probs, const uint8_t* target, int64_t count, int32_t dtype, float w_bce, float w_dice,
                      float* scalars, void* dprobs, int32_t dl_channels, void* workspace, size_t workspace_bytes, void* stream);
/* The multi-class form of the same head (classes 2..32, activation softmax inside the class convolution, model.py:485):
 * Activation('softmax') over the first `classes` columns as a tensor op with its gradient dz_c = p_c (dp_c - sum_k p_k dp_k),
 * and Keras categorical_crossentropy (p / sum p, 1e-7 clip) + w_dice * dice_loss on the RESIZED probabilities
 * [pixels][ldc] against class-index targets; scalars as stp_softmax_cce_dice, dprobs [pixels][dl_channels >= classes]. */
int stp_softmax_act(const void* z, void* p, int64_t rows, int32_t classes, int32_t ldz, int32_t ldp, int32_t dtype, void* stream);
int stp_softmax_act_bwd(const void* p, const void* dp, void* dz, int64_t rows, int32_t classes, int32_t ldp, int32_t ldg,
                        int32_t dtype, void* stream);
int stp_prob_cce_dice(const void* probs, const uint8_t* target, int64_t pixels, int32_t classes, int32_t ldc, int32_t dtype,
                      float w_cce, float w_dice, float* scalars, void* dprobs, int32_t dl_channels, void* workspace,
                      size_t workspace_bytes, void* stream);

/* ----------------------------------------------------------------------------------------------
 * On-device augmentation (replaces the imgaug worker processes, schemas/augmenters.raml:43-133).
 * One fused pass per batch: inverse affine warp (flips, Affine, crop / pad augmenters and the final Resize composed on
 * the host into a 2x3 matrix per sample), bilinear for the image (constant 0 border, result rounded to uint8 like
 * imgaug), nearest for the mask, then the point operations in the fixed order Add, Multiply, MultiplyElementwise,
 * AddElementwise, AdditiveGaussianNoise, Dropout, Grayscale, Invert (integer / single-rounded float arithmetic: the
 * numpy oracle reproduces them bit for bit; per-pixel randomness is a counter-based hash of seed, pixel, channel, op).
 * params per sample: float[STP_AUG_RECORD] (integers stored as exactly representable floats)
 *   0-5 m00 m01 m02 m10 m11 m12 (output->input pixel map)   6-8 Add per channel   9-11 Multiply per channel
 *   12 flags (1 Invert, 2 noise per channel, 4 dropout per channel, 8 AddElementwise per channel,
 *             16 MultiplyElementwise per channel, 32 AddElementwise on, 64 MultiplyElementwise on)
 *   13 Grayscale alpha * 256   14 noise k = rint(sigma * 65536 / 147.8)   15 dropout threshold rint(p * 2^24)
 *   16-17 AddElementwise [lo, hi] (int)   18-19 MultiplyElementwise [lo, hi]   20 seed (integer < 2^24)   21-23 zero
 */
#define STP_AUG_RECORD 24
int stp_augment_u8(const uint8_t* img, const uint8_t* mask, uint8_t* img_out, uint8_t* mask_out,
                   const float* params, int32_t N, int32_t Hin, int32_t Win, int32_t Hout, int32_t Wout,
                   int32_t C, void* stream);
/* Neighbourhood filters of the augmenter catalogue on a uint8 batch [N,H,W,C] (src != dst): per image an
 * int32[STP_FILTER_RECORD] record = K (odd <= 13; 0 = copy), mode (0 = linear K x K filter, 1 = median), 0, 0, K*K weights
 * in 1/16384 units.  Reflect-101 border, integer arithmetic (bit-exact against the oracle).  GaussianBlur, AverageBlur,
 * Sharpen, Emboss and EdgeDetect are linear filters whose weights the host derives; MedianBlur is mode 1. */
#define STP_FILTER_RECORD 173
int stp_filter_u8(const uint8_t* src, uint8_t* dst, const int32_t* params, int32_t N, int32_t H, int32_t W, int32_t C,
                  void* stream);
/* The non-affine geometric augmenters (schemas/augmenters.raml:126-133) as a per-pixel displacement field [N][Hout][Wout]
 * (int32 = two int16 (dx, dy) in 1/64 pixel) that stp_augment_field_u8 adds to the output pixel position before its matrix:
 * out(p) = in(M (p + D(p))), image bilinear / mask nearest as in stp_augment_u8 (field NULL = stp_augment_u8).
 * stp_field_piecewise: PiecewiseAffine - grid int32 [N][rows][cols][2] = control-point jitter in 1/64 pixel at
 * linspace(0, H, rows) x linspace(0, W, cols), affine per triangle.  stp_field_elastic: ElasticTransformation - per image
 * int32[STP_ELASTIC_RECORD] = seed, alpha (1/64 pixel), blur radius r <= 64, 0, one-sided Gaussian weights w[0..r] in 1/32768
 * (full kernel sums to 32768); tmp is a scratch buffer of the field's size.  Integer arithmetic, bit-exact against the oracle. */
#define STP_ELASTIC_RECORD 69
int stp_augment_field_u8(const uint8_t* img, const uint8_t* mask, uint8_t* img_out, uint8_t* mask_out, const float* params,
                         const int32_t* field, int32_t N, int32_t Hin, int32_t Win, int32_t Hout, int32_t Wout, int32_t C,
                         void* stream);
int stp_field_piecewise(int32_t* field, const int32_t* grid, int32_t N, int32_t H, int32_t W, int32_t rows, int32_t cols,
                        void* stream);
int stp_field_elastic(int32_t* field, int32_t* tmp, const int32_t* params, int32_t N, int32_t H, int32_t W, void* stream);
/* BackgroundReplacer (README.md:270-278, FAQ.md:24-38): out = img inside the mask eroded by a (2*erosion+1)^2 minimum, bg
 * (a background image already resized to H x W, same layout as img) elsewhere; out != img; the mask is not changed. */
int stp_background_replace_u8(const uint8_t* img, const uint8_t* mask, const uint8_t* bg, uint8_t* out, int32_t N, int32_t H,
                              int32_t W, int32_t C, int32_t erosion, void* stream);

/* Gradient-bucket helpers for the RCCL all-reduce (fp32 <-> bf16 wire format). */
int stp_cast_f32_to_bf16(const float* src, void* dst, int64_t count, void* stream);
int stp_cast_bf16_to_f32(const void* src, float* dst, int64_t count, float scale, void* stream);

/* segmentation_models' PSPNet head (schemas/segmentation.raml:226-249: Conv2D(512, 1x1) over Concatenate([feature, the four resized
 * pyramid levels])) WITHOUT the concatenation: a 1x1 convolution commutes with the bilinear resize, so the head is Conv2D(1x1) of the
 * feature with its columns of the kernel + stp_upsample_sum of the 1x1 convolutions of the TINY level maps with theirs (the sum enters
 * the feature convolution as its residual operand).  stp_upsample_sum: y[N][Ho][Wo][C] = sum over the non-NULL sources i of the TF-1.x
 * bilinear resize (align_corners False) of x_i [N][h_i][h_i][C] by Ho / h_i (Ho == Wo, Ho % h_i == 0), one rounding; its gradient is
 * stp_resize_bilinear_bwd per source.  stp_copy_cols_f32: dst[r][0..cols) (+)= src[r][0..cols), fp32 row-major with row pitches ld_* -
 * the column range of the shared kernel as a dense matrix for stp_weight_prepare, and the weight gradient back into the range. */
int stp_upsample_sum(const void* x0, const void* x1, const void* x2, const void* x3, int32_t h0, int32_t h1, int32_t h2, int32_t h3,
                     void* y, int32_t N, int32_t Ho, int32_t Wo, int32_t C, int32_t dtype, void* stream);
int stp_copy_cols_f32(float* dst, int32_t ld_dst, const float* src, int32_t ld_src, int32_t rows, int32_t cols, int32_t accumulate, void* stream);

/* Batched reduce of lone weight gradients (round 6; the weight gradient of Keras' backward feeds only the optimizer, so its split-K
 * reduction can wait): a layer whose stp_conv2d_wgrad_partial launch (variant 0) wrote plain [splits][Cout * KH * KW * C] slabs into a
 * workspace OF ITS OWN adds a descriptor to a host table (stp_wgrad_reduce_desc_fill: returns the layer's element count, 0 = this layer
 * keeps its own stp_conv2d_wgrad_reduce); the table, copied to the device, is reduced by ONE launch (n layers, max_count = the largest
 * element count).  Deterministic (fixed walk, fixed tree). */
size_t stp_wgrad_reduce_desc_bytes(void);
int64_t stp_wgrad_reduce_desc_fill(void* host_table, int32_t index, const stp_wgrad_params* p, const void* workspace);
int stp_wgrad_reduce_batched(const void* table_dev, int32_t n, int64_t max_count, void* stream);

/* Conv2D(1x1, strides 1) of the bottleneck ResNets (classification_models residual_bottleneck_block conv1 / conv3 / the stride-1 shortcut,
 * segmentation_models' FPN laterals; reached through segmentation.py:109-118) and its data gradient, 16-bit storage, 64 ... 512 channels
 * (stp_conv2d_pw_eligible: the served (C0, Cout) pairs, N * Ho * Wo a multiple of the tile): a pixel-STREAMING kernel - persistent
 * workgroups, the whole weight matrix as register-resident MFMA fragments, pixels double-buffered by LDS-DMA, epilogue (bias | statistics
 * (+ residual) | BatchNormalization backward (+ accumulate)) from the accumulators with 16-byte accesses.  stp_conv2d dispatches to it
 * (STP_PW=0 switches the automatic use off; tile = 800 forces it).  The [2][Cout][columns] table of fused sums has ONE column per
 * workgroup: stp_conv2d_pw_cols (= stp_conv2d_stats_floats / (2 Cout)). */
int stp_conv2d_pw_eligible(const stp_conv_params* p);
int stp_conv2d_pw_cols(const stp_conv_params* p);
int stp_conv2d_pw(const stp_conv_params* p, void* stream);

/* Box calibration probes (bench.py `box_calibration`; no counterpart in the reference - the boxes of one MI355X pool differ by +-2.5 % in
 * sustained clocks, more than one round's gain on the step, so the bench line states what the box it ran on sustains):
 * stp_calib_mfma - `blocks` workgroups of 8 waves, each wave `iters` x 16 back-to-back v_mfma_f32_32x32x16 of the build's 16-bit
 * format on register operands (stp_calib_mfma_flops(blocks, iters) = the FLOP of one launch); out: >= blocks * 512 floats of scratch.
 * stp_calib_copy - device copy with 16-byte accesses (bytes a multiple of 16, both pointers 16-byte aligned). */
int64_t stp_calib_mfma_flops(int32_t blocks, int32_t iters);
int stp_calib_mfma(float* out, int32_t blocks, int32_t iters, void* stream);
int stp_calib_copy(void* dst, const void* src, int64_t bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* STP_HIP_H */
