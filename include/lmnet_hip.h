/*
 * lmnet_hip.h -- C-ABI of liblmnet_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for the
 * forward/backward hot path of Asunatan/LM-Net (core/LM_Net.py:95-123 and the live classes of
 * core/modules.py).
 *
 * The reference has NO FFI/plugin layer for this path: its "operator API" is the Python
 * nn.Module `core.LM_Net.LM_Net` (core/LM_Net.py:5-123), built from torch.nn ops plus the
 * third-party CUDA ops natten2dqkrpb / natten2dav (natten, reached at core/modules.py:509,517).
 * This header is therefore the boundary the reference WOULD bind if it had one: one entry per
 * fused row of SURVEY.md section 8a, plain pointers + sizes + a stream, no torch types.
 * Each entry cites the reference code whose arithmetic it replaces.
 *
 * Conventions
 *   - parameters, statistics, gradients of parameters and workspaces are fp32 device pointers.  ACTIVATION tensors
 *     (`const void*` / `void*` arguments, lmn_src_t.ptr, aux / residual / out / dy of the conv family) are stored
 *     as fp32 or bf16: every activation tensor of ONE call has the storage type named by its `act_dtype` argument
 *     (LMN_F32 | LMN_BF16; all arithmetic, accumulation and statistics are fp32 either way).  Strides and offsets
 *     are in ELEMENTS.  The network input and the logits are fp32 NCHW; lmn_nchw_to_nhwc / lmn_nhwc_to_nchw are
 *     the converters at that boundary.
 *   - activations are NHWC ("pixel-major"): element
 *     (b,y,x,c) of a tensor with pixel stride `cstride` lives at ((b*H+y)*W+x)*cstride + c.
 *     A pointer may address a channel slice of a wider buffer (cstride > C) -- this is how
 *     torch.cat along channels (core/modules.py:104,139,497) is expressed without a copy.
 *   - ROW-PLANAR tensors ("RP4", round 4): the E-wide tensors INSIDE a ReparamConv block (core/modules.py:587-599: the expand
 *     conv's output z / x1, the GELU input `pre`, and in the backward u, dpre, dh / dx1, dz) are stored as
 *         element (b,y,x,c)  at  ((b*H + y) * (C/4) + (c >> 2)) * 4*W + 4*x + (c & 3)
 *     i.e. one contiguous plane of W pixels x 4 channels per image row and channel quad (the reference's own NCHW
 *     with the channel quad folded next to x).  The depthwise stencils (lmn_dw_*) take ONLY this layout: their lanes run
 *     along x with a wave-uniform channel, so a wave's row is one coalesced load and needs no transpose.  The 1x1 convs
 *     and weight gradients either side of them read / write it when told so: lmn_src_t.rp_w, lmn_conv_args_t.out_rp_w /
 *     aux_rp_w, lmn_wgrad_args_t.dy_rp_w = W (0: NHWC); every row-planar operand of one call has the same W, is a whole
 *     tensor (cstride == C) and belongs to a 1x1 stride-1 call.
 *   - the caller owns every buffer (incl. workspace); kernels never allocate, free or synchronise;
 *     everything is enqueued on `stream` (hipStream_t) and is stream-ordered and re-entrant.
 *   - return value: 0 on success, <0 = LMN_E_* argument error, >0 = hipError_t from the launch.
 *   - reductions into `stats`/gradient buffers use atomicAdd: the caller zeroes them first
 *     (lmn_fill) unless stated otherwise.
 */
#ifndef LMNET_HIP_H
#define LMNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* lmn_stream_t; /* hipStream_t */

#define LMN_ABI_VERSION 14
/* arithmetic type of the matrix-core operands of a dense contraction (accumulators, epilogues, statistics: fp32) */
#define LMN_F32 0  /* v_mfma_f32_16x16x4_f32: exact fp32 (k-ordered fma chain)                                  */
#define LMN_BF16 1 /* v_mfma_f32_16x16x16_bf16: operands rounded to bf16 (RNE) when staged / packed -- the mixed- */
                   /* precision path (reference: torch.cuda.amp.autocast, utils/train_eval_utils.py:130-138)      */
#define LMN_E_BADARG (-1)
#define LMN_E_UNSUPPORTED (-2)

int lmn_abi_version(void);
/* sizeof() of the argument structs, so a binding can verify its mirror of the layout */
int lmn_sizeof_conv_args(void);
int lmn_sizeof_src(void);
const char* lmn_last_error(void);

/* ------------------------------------------------------------------------------------------
 * Dense convolution family on the matrix cores (v_mfma_f32_16x16x4_f32, exact fp32).
 * Replaces every nn.Conv2d(k=1|3, stride=1|2) and nn.Linear of the path:
 *   ReparamConv.expand_conv / pointwise_conv / shortcut   core/modules.py:537-539,576-584
 *   down1..4, up1..4 convs, output of M2Skip/M3Skip convs   core/LM_Net.py:14-39,58-74; modules.py:83-143
 *   OverlapPatchEmbed                                       core/modules.py:22-40
 *   Mlp.fc1/fc2, GlobalAttention.qkv/proj, natten qkv/proj  core/modules.py:42-56,235-279,509
 *   GFT.conv                                                core/modules.py:337
 * and, with `transposed=1` / lmn_conv_wgrad, their autograd (ATen conv/linear backward).
 * ------------------------------------------------------------------------------------------ */

/* source (input operand) transforms applied on load, in this order */
#define LMN_SRC_GELU 1 /* x <- gelu(x)                (exact erf form, nn.GELU default)          */
#define LMN_SRC_DROP 2 /* x <- x * keep(seed,idx)/(1-p) (same mask as the forward epilogue)      */
/* (round 5) fused producers: the tensor that the reference materialises between two modules is formed where the conv stages it.
 * LMN_SRC_LN  x <- LayerNorm_C(x) * ln_gamma + ln_beta over the source's C channels of each pixel (biased variance, ln_eps):
 *             `norm1` -> `qkv` and `norm2` -> `fc1` of the transformer blocks as ONE row each (core/modules.py:516-518, 343-344,
 *             50-56; SURVEY 8a "LN1+qkv") -- n1 / n2 never cross HBM.  1x1 stride-1 single-source calls only, applied before
 *             the other transforms.  lmn_conv_fwd computes (mean, rstd) per pixel in its staging and, when ln_stats != NULL,
 *             stores them as [pixels][2]; lmn_conv_wgrad READS them there (ln_stats required) and applies the same transform.
 * LMN_SRC_UP2 the source tensor is [B][Hin/2][Win/2][C] and x is its bilinear x2 upsampling with align_corners=True
 *             (F.interpolate / nn.Upsample, core/LM_Net.py:59-72, core/modules.py:94,129; ATen index arithmetic:
 *             src = dst * (in - 1) / (out - 1) in fp32) sampled where the 3x3 window is staged -- `up` never crosses HBM
 *             (SURVEY row A10 "fuse upsample into the conv's tile load").  3x3 stride-1 calls (forward and weight gradient). */
#define LMN_SRC_LN 4
#define LMN_SRC_UP2 8

typedef struct {
  const void* ptr;    /* NHWC base (act_dtype of the call), already offset to the slice's first channel */
  const float* scale; /* optional [B][C] per-(image,channel) multiplier (SE gate) or NULL         */
  int32_t C;          /* channels contributed by this source (multiple of 4)                      */
  int32_t cstride;    /* floats between consecutive pixels                                        */
  int32_t flags;      /* LMN_SRC_*                                                                */
  uint32_t drop_seed; /* dropout stream id for LMN_SRC_DROP                                       */
  float drop_p;       /* dropout probability for LMN_SRC_DROP                                     */
  int32_t rp_w;       /* 0: NHWC; W > 0: the tensor is row-planar (RP4, see Conventions) with image width W */
  const float* ln_gamma; /* LMN_SRC_LN: [C] affine of the LayerNorm                                  */
  const float* ln_beta;
  float* ln_stats;       /* LMN_SRC_LN: [pixels][2] = (mean, rstd); conv_fwd writes (or NULL), conv_wgrad reads */
  float ln_eps;          /* LMN_SRC_LN: nn.LayerNorm eps (1e-5)                                      */
  int32_t _pad0;
} lmn_src_t;

/* epilogues; v = accumulator + bias[co] */
#define LMN_EP_LINEAR 0     /* o = v                                                              */
#define LMN_EP_AFFINE_ACT 1 /* o = act(v*p0[co] + p1[co])           (BN folded to scale/shift)    */
#define LMN_EP_DGELU 2      /* o = v * gelu'(aux)                    (Mlp backward through GELU)  */
#define LMN_EP_BN_BWD1 3    /* zh=(v-p0)*p1; h=p2*zh+p3; o=aux*act'(h); stats+=(o, o*zh); out may be NULL (statistics only) */
#define LMN_EP_BN_BWD2 4    /* zh=(v-p0)*p1; dh = p5 ? aux*act'(p5*zh+p6) : aux; o=p2*dh - p3 - zh*p4 */
#define LMN_EP_SE_BWD 5     /* o = v; stats[b][co] += v*gelu(aux)    (d gate of the SE block)     */
#define LMN_EP_LN_BWD 6     /* LayerNorm backward behind the data gradient of the Linear it feeds (1x1, every cout in one block: Cout <= 48):   */
                            /* g = v*p0 (gamma); zh = (aux - mean)*rstd with (mean, rstd) = p6[pixel][2], the table the LMN_SRC_LN forward left;    */
                            /* o = rstd*(g - mean_c(g) - zh*mean_c(g*zh)) (+ residual); stats[2][Cout] += (sum_px v, sum_px v*zh) = (d beta, d gamma), */
                            /* rows swapped when act != 0.  aux = the LayerNorm INPUT.  core/modules.py:516-518 backward, one launch instead of two  */

#define LMN_ACT_NONE 0
#define LMN_ACT_HSWISH 1 /* nn.Hardswish  core/modules.py:539 */
#define LMN_ACT_GELU 2   /* nn.GELU       core/modules.py:574,98,134 */

/* what `stats` accumulates (atomicAdd) */
#define LMN_STATS_NONE 0
#define LMN_STATS_SUM_SQ 1 /* [2][Cout]: sum d, sum d*d over all pixels, d = v - p4[co] (p4 NULL: d = v): BatchNorm batch statistics */
#define LMN_STATS_EP 2     /* defined by the epilogue (BN_BWD1: [2][Cout]; SE_BWD: [B][Cout])        */

/* In-kernel BatchNorm bookkeeping of lmn_conv_fwd: the conv that CONSUMES a batch statistic forms its per-channel
 * coefficients itself -- every block for its own cout range, blocks with blockIdx.x == 0 also write the side outputs --
 * which saves the lmn_bn_finalize / lmn_bn_bwd_coef launch between the statistics pass and the consuming pass (two
 * dependent ~10 us launches per BatchNorm and step on the critical path).                                            */
#define LMN_FIN_NONE 0
#define LMN_FIN_BN 1     /* EP_AFFINE_ACT: p0 / p1 (A, shift) are formed from `sums` exactly as lmn_bn_finalize would  */
#define LMN_FIN_BN_BWD 2 /* EP_BN_BWD2: p2 / p3 / p4 (c1, c2, c3) are formed from `sums` exactly as lmn_bn_bwd_coef would */
typedef struct {
  int32_t mode;          /* LMN_FIN_*                                                                              */
  int32_t nrep;          /* slices of `sums` (1..16)                                                               */
  float count, eps, momentum;
  int32_t batch_stats;   /* LMN_FIN_BN_BWD: as lmn_bn_bwd_coef                                                     */
  const float* sums;     /* [nrep][2][Cout]                                                                        */
  const float* gamma;    /* LMN_FIN_BN                                                                             */
  const float* beta;
  const float* about;    /* LMN_FIN_BN: the per-channel shift the sums were taken about, or NULL.  It must NOT alias  */
                         /* `rmean` (block 0 updates the running mean while other blocks still read `about`): pass    */
                         /* the snapshot the statistics pass left behind its slices (lmn_conv_args_t.stats_snap)      */
  float* mean;           /* LMN_FIN_BN outputs, each may be NULL                                                   */
  float* rstd;
  float* A;
  float* shift;
  float* rmean;          /* running statistics, updated in place (momentum, unbiased variance)                      */
  float* rvar;
  const float* Ain;      /* LMN_FIN_BN_BWD: A = gamma * rstd of the forward                                         */
  float* dgamma;         /* LMN_FIN_BN_BWD: += sum dh*zhat, += sum dh (each may be NULL)                             */
  float* dbeta;
} lmn_bn_fin_t;

/* (ABI 13) A SECOND 1x1 conv chained behind a 1x1 lmn_conv_fwd call: out2 = W2 . out + bias2 per pixel, computed from the first
 * conv's output tile while it is still in registers (the accumulator fragment of a 16x16x4 MFMA is the B fragment of the next one) --
 * `out` is written as before, but never re-read, and one launch disappears.  Two uses inside a stage `Sequential(ReparamConv,
 * ReparamConv)` (core/LM_Net.py:11-57):
 *   forward : pointwise + shortcut of block A (core/modules.py:597-599) -> expand conv of block B (core/modules.py:537, 586):
 *             epilogue LMN_EP_LINEAR, stats_mode LMN_STATS_SUM_SQ (about `shift`, slices, snapshot: as lmn_conv_args_t);
 *   backward: folded data gradient of block B (lmn_reparam_fold) -> data gradient of block A's pointwise conv with the squeeze-excite
 *             sums (epilogue LMN_EP_SE_BWD, aux = A's `pre`, stats [B][Cout], stats_mode LMN_STATS_EP).
 * wpack: lmn_conv_pack / lmn_conv_pack_t of the second conv over ONE source of the first conv's Cout rounded up to 4 channels.
 * Served by the LDS-DMA streaming kernel only (fp32, levels 0-1 of LM-Net at its instance table): lmn_conv_chain_ok says beforehand
 * whether lmn_conv_fwd takes the arguments; it does not in deterministic mode (two statistics destinations, one slot scratch).    */
typedef struct {
  const float* wpack;    /* NULL: no chained conv                                                                                */
  const float* bias;     /* [Cout] or NULL                                                                                       */
  const float* shift;    /* SUM_SQ: statistics about shift[co] (NULL: about 0), as p4 of the first conv                          */
  const void* aux;       /* SE_BWD: activation tensor sampled at the output element                                              */
  void* out;             /* [pixels][Cout] activation tensor of the second conv (NHWC or row-planar)                             */
  float* stats;          /* SUM_SQ: [stats_rep (+1 with stats_snap)][2][Cout]; SE_BWD: [B][Cout]                                */
  int32_t Cout, out_cstride, out_rp_w, aux_cstride, aux_rp_w;
  int32_t epilogue, stats_mode, stats_rep, stats_snap;
} lmn_conv_chain_t;

typedef struct {
  int32_t B, Hout, Wout, Hin, Win;
  int32_t ksize;      /* 1 or 3; padding = ksize/2                                                */
  int32_t stride;     /* 1 or 2                                                                   */
  int32_t transposed; /* 0: out(y,x) reads in(y*s + t - pad)                                      */
                      /* 1: data-gradient form: out(y,x) reads in((y + pad - t)/s) when divisible  */
  int32_t nsrc;       /* 1..3 sources, concatenated along the reduction (input-channel) axis      */
  int32_t Cout;
  lmn_src_t src[3];
  const float* wpack; /* weights in MFMA fragment order, see lmn_conv_pack                        */
  const float* bias;  /* [Cout] or NULL                                                           */
  const float* p0;
  const float* p1;
  const float* p2;
  const float* p3;
  const float* p4;       /* per-Cout epilogue vectors                                             */
  const void* aux;       /* epilogue ACTIVATION tensor sampled at the output element, or NULL     */
  const void* residual;  /* activation tensor added after everything else, or NULL                */
  void* out;             /* activation tensor; NULL: nothing is written (statistics-only pass)    */
  float* stats;          /* see stats_mode                                                        */
  int32_t aux_cstride, res_cstride, out_cstride;
  int32_t epilogue, act, stats_mode;
  float drop_p;       /* >0: epilogue dropout  o <- o*keep/(1-p) before the residual add          */
  uint32_t drop_seed; /* (nn.Dropout(0.1), core/modules.py:48,53,55)                              */
  const uint32_t* seed_ctr; /* optional DEVICE word added to every dropout seed of this call: lets a captured */
                            /* hipGraph draw a new mask per replay (the host bumps the word once per step)    */
  const float* bias2;       /* second [Cout] bias or NULL: one conv over two sources stands for two biased    */
                            /* convs (pointwise_conv + shortcut, core/modules.py:597-599)                     */
  int32_t stats_rep;        /* > 1 (SUM_SQ and BN_BWD1 statistics only): `stats` holds stats_rep slices of     */
  int32_t mma_dtype;        /* [2][Cout]; block b adds into slice b % stats_rep (1280 blocks adding to the same */
                            /* 2 cache lines cost 15-20 us of same-address atomics); the consumers              */
                            /* (lmn_bn_finalize / lmn_bn_bwd_coef, argument nrep) sum the slices                */
                            /* mma_dtype: LMN_F32 | LMN_BF16 -- `wpack` must have been packed with the same dtype */
  int32_t act_dtype;        /* storage of src[].ptr / aux / residual / out: LMN_F32, or LMN_BF16 (needs mma_dtype   */
  int32_t out_rp_w;         /* LMN_BF16).  out_rp_w / aux_rp_w: 0 = NHWC, W > 0 = `out` / `aux` is row-planar (RP4)      */
  const float* p5;          /* BN_BWD2 only: BatchNorm gamma / beta when `aux` is the gradient w.r.t. the ACTIVATED   */
  const float* p6;          /* output (the activation derivative is then applied here and dh is never written)        */
  lmn_bn_fin_t fin;         /* optional in-kernel BatchNorm bookkeeping (fin.mode = LMN_FIN_NONE: off)                   */
  int32_t stats_snap;       /* LMN_STATS_SUM_SQ: also copy the shift vector p4 (zeros if NULL) to stats[stats_rep*2*Cout..]: */
  int32_t aux_rp_w;         /* a snapshot that stays valid when p4 is a running mean updated by the consuming pass        */
  lmn_conv_chain_t chain;   /* (ABI 13) optional second 1x1 conv applied to the output tile of this one (chain.wpack NULL: none) */
} lmn_conv_args_t;

/* number of floats lmn_conv_pack writes for (ksize, Cout, src channel counts c[nsrc]) */
int64_t lmn_conv_pack_size(int ksize, int Cout, int nsrc, const int32_t* c);
/* w: torch layout [Cout][Cin][k][k] (Cin = sum c[]; Linear weights are k=1).
 * transposed=0 packs for lmn_conv_fwd(args.transposed=0): rows = Cout, reduction = Cin.
 * transposed=1 packs the data-gradient operator: rows = Cin (single source of Cout channels);
 *   `c` then lists ONE entry = Cout of the forward conv, `row_off`/`rows` select the slice of
 *   forward input channels whose gradient is produced (one call per forward source).        */
int lmn_conv_pack(const float* w, float* wpack, int ksize, int Cout, int Cin, int nsrc, const int32_t* c,
                  int transposed, int row_off, int rows, int dtype /* LMN_F32 | LMN_BF16: half the bytes */,
                  lmn_stream_t stream);
/* The same packing for MANY weights in one launch (weights change every optimizer step, so a training step
 * re-packs every dense weight of the net: one launch instead of ~190).  `jobs_dev` is a DEVICE array of njobs
 * descriptors sorted by first_block, where job j owns blocks [first_block_j, first_block_j + ceil(total_j/1024))
 * and total_j = lmn_conv_pack_size(...); total_blocks = the sum.  Field meaning as in lmn_conv_pack.            */
typedef struct {
  const float* w;
  float* wpack;
  int64_t total;       /* floats written = lmn_conv_pack_size(ksize, rows, nsrc, c) */
  int64_t first_block;
  int32_t ksize, Cout, Cin, nsrc;
  int32_t c[3];
  int32_t transposed, row_off, rows;
  int32_t dtype; /* LMN_F32 | LMN_BF16 */
  int32_t _pad;
} lmn_pack_job_t;
int lmn_sizeof_pack_job(void);
int lmn_conv_pack_batch(const lmn_pack_job_t* jobs_dev, int njobs, int64_t total_blocks, lmn_stream_t stream);
int lmn_conv_fwd(const lmn_conv_args_t* args, lmn_stream_t stream);
/* (ABI 13) Kernel-form switch of lmn_conv_fwd: the LDS-DMA double-buffered kernels or the LDS-tiled one, for
 *   bit 0 (1): the small-channel 3x3 stride-1 calls (one plain NHWC fp32 source of 12 / 24 channels, Cout <= 32, bias / residual /
 *              SUM_SQ statistics; conv_dma3.hip) -- skip-fuser convs, `nat` patch embedding and their data gradients at levels 0-1
 *              (core/modules.py:22-39,83-143, core/LM_Net.py:58-74);
 *   bit 1 (2): the 1x1 convs of ReparamConv at levels 0-1 (conv_dma1.hip: expand conv, pointwise + shortcut, SE-gradient conv,
 *              folded data gradient; core/modules.py:537-539,576-599).
 * Same arithmetic either way (k-ordered fp32 MFMA chains; the ORDER of the channels inside a tap differs, so results agree to fp32
 * rounding, not bit for bit).  mode: bit mask, 0 = off, -1 keep (default 3); min_tiles: calls with fewer pixel tiles keep the
 * LDS-tiled kernel (-1 keep; default 512).  Returns the previous mode (-1: not yet read from LMN_CONV_DMA).  Process-wide.        */
int lmn_conv_dma_config(int mode, int min_tiles);
/* (ABI 13) 1 when lmn_conv_fwd accepts these arguments WITH their `chain` (see lmn_conv_chain_t), 0 when the caller launches the two
 * convs separately.  Host arithmetic only, nothing is launched, lmn_last_error is left untouched.                                */
int lmn_conv_chain_ok(const lmn_conv_args_t* args);

/* Weight/bias gradient:  dW[co][ci][ty][tx] += sum_pixels dy[p][co] * src(p*s + t - pad)[ci]
 * (sources and their on-load transforms exactly as in the forward call), db[co] += sum dy.
 * dW is the torch-layout gradient [Cout][Cin][k][k] of the forward weight; accumulates
 * (atomicAdd) -- zero it first.  `dy` has pixel stride dy_cstride; dy_flags/dy_seed/dy_p allow
 * LMN_SRC_DROP on dy (gradient through the epilogue dropout).                                 */
typedef struct {
  int32_t B, Hout, Wout, Hin, Win, ksize, stride, nsrc, Cout;
  lmn_src_t src[3];
  const void* dy;   /* activation tensor (act_dtype) */
  int32_t dy_cstride;
  int32_t dy_flags;
  uint32_t dy_seed;
  float dy_p;
  float* dW;
  float* db;        /* or NULL */
  float* workspace; /* optional scratch for the deterministic two-stage K-split reduction (no atomics);   */
  int64_t workspace_floats; /* size lmn_conv_wgrad_workspace() asks for; NULL/0 => LDS-reduced atomics   */
  const uint32_t* seed_ctr; /* as in lmn_conv_args_t: device word added to the dropout seeds of this call   */
  float* dW_src[3]; /* optional: gradient of the weight slice of source s as its OWN tensor [Cout][C_s][k][k]    */
                    /* (one pass over dy for convs that were fused over several sources in the forward, e.g.   */
                    /* pointwise_conv + shortcut); NULL entries fall back to the columns of dW (may be NULL if */
                    /* every source has its own tensor)                                                        */
  float* db2;       /* optional second bias gradient receiving the same sum as db                              */
  int32_t mma_dtype; /* LMN_F32 | LMN_BF16: operand type of the pixel-reduction MFMAs (accumulators fp32)                */
  int32_t act_dtype; /* storage of src[].ptr and dy                                                                      */
  int32_t defer_reduce; /* 1: with a workspace, the call only writes its K-split block partials; the caller sums them    */
  int32_t dy_rp_w;      /*    later with lmn_wgrad_reduce_batch (job description: lmn_conv_wgrad_job).  The workspace    */
                        /*    must then be the call's own until that launch (not a scratch shared with other calls)      */
                        /* dy_rp_w: 0 = NHWC, W > 0 = `dy` is row-planar (RP4)                                           */
} lmn_wgrad_args_t;
int lmn_sizeof_wgrad_args(void);
/* floats of workspace that make lmn_conv_wgrad use the two-stage reduction for this problem (0: not useful) */
int64_t lmn_conv_wgrad_workspace(const lmn_wgrad_args_t* args);
int lmn_conv_wgrad(const lmn_wgrad_args_t* args, lmn_stream_t stream);
/* (ABI 13) 1 when lmn_conv_wgrad takes these arguments with LMN_SRC_UP2 on its source -- the 3x3 weight gradient of
 * `Upsample(x2, bilinear, align_corners) -> Conv2d` (core/LM_Net.py:58-74, core/modules.py:94,129) sampling the upsampling
 * where it stages its window; 0: the caller materialises the upsampled tensor (lmn_up2_fwd) and passes it as a plain source.
 * Host arithmetic only (the launch's own geometry predicate), nothing is launched, lmn_last_error is left untouched.       */
int lmn_conv_wgrad_up2_ok(const lmn_wgrad_args_t* args);
/* Deferred second stage of the K-split reduction.  Every weight gradient of a backward pass used to be followed by its own
 * reduction launch (82 launches of 8-11 us per LM-Net training step); with defer_reduce the partials stay in per-call
 * workspaces and ONE launch per gradient bucket sums all of them (same fixed order: deterministic).
 * lmn_conv_wgrad_job: host arithmetic only -- describes the reduction the call with these arguments leaves behind
 * (out->nblk == 0: the call reduces by itself, nothing to defer).  The caller fills first_block (running sum of
 * gy * blocks_per_set over the jobs of a batch), copies the jobs to device memory and launches them together.          */
typedef struct {
  const float* partial;   /* [gy][nblk][per] block partials */
  int64_t first_block;
  int32_t nblk, per, gy, nsets_n;
  int32_t taps, NMT, NNT, nsrc;
  int32_t Cout, Cin, NMTT, NNTT;
  int32_t srcC[3], ntile_off[3], cbase[3];
  int32_t ksl, blocks_per_set, _pad;
  float* dW;
  float* dW_src[3];
  float* db;
  float* db2;
} lmn_reduce_job_t;
int lmn_sizeof_reduce_job(void);
/* ReparamConv backward on the z-path (lmn_dw_pre_t): the BatchNorm backward of the expand conv (core/modules.py:537-539, 587)
 * folded into weights.  From hstats [2][E] = (sum dh, sum dh * z) (lmn_dw_bwd_bn) and the forward's mean / rstd / A it forms
 *   dz = a * dh + b * z + c   (a = A, b = -A T rstd / N, c = -A S0 / N - b mean, T = (S1 - mean S0) rstd; eval-mode BN: b = c = 0)
 * writes coef [3][E] = (a, b, c) for the weight-gradient side (lmn_affine2), adds dgamma += T, dbeta += S0, and -- since
 * z = W_e x + b_e -- packs the block's whole data gradient as ONE three-source 1x1 conv for lmn_conv_fwd:
 *   dx = [W_e^T diag(a)] dh + [W_e^T diag(b) W_e] x + W_sc^T dy + kbias,   kbias = W_e^T (b * b_e + c)
 * wpack: lmn_conv_pack_size(1, rows, {E, rows, cred}) floats; rows / cred = channels of x / dy (>= the weights' cin_w / cout_w,
 * zero-padded operators).  w_expand [E][cin_w], w_shortcut [cout_w][cin_w] in torch layout.                                    */
int lmn_reparam_fold(const float* hstats, const float* mean, const float* rstd, const float* A, float count, int batch_stats,
                     const float* w_expand, const float* b_expand, const float* w_shortcut, int E, int rows, int cin_w, int cred,
                     int cout_w, float* wpack, float* kbias, float* coef, float* dgamma, float* dbeta, int dtype,
                     lmn_stream_t stream);
int lmn_conv_wgrad_job(const lmn_wgrad_args_t* args, lmn_reduce_job_t* out);
int lmn_wgrad_reduce_batch(const lmn_reduce_job_t* jobs_dev, int njobs, int64_t total_blocks, lmn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Multi-branch depthwise stencil of ReparamConv (row A2): four depthwise convs 5x5, 3x3, 3x1,
 * 1x3 (zero pad k/2, no bias), each followed by its own BatchNorm, summed, then GELU
 * (core/modules.py:548-574, 592-597).  Weight pointers use torch layouts [E][1][kh][kw].
 * Every ACTIVATION tensor of this family (x1 / z, pre, u, dpre, dx1 / dh) is ROW-PLANAR (RP4, see Conventions) with
 * C = E and the W of the call.
 * ------------------------------------------------------------------------------------------ */
/* z-path of ReparamConv (core/modules.py:587, 592-597): the tensor handed to the depthwise kernels is z, the output of the
 * expand conv BEFORE its BatchNorm and Hardswish; x1 = Hardswish(A * z + shift) is formed when the kernels stage their rows
 * (once per staged element, zero padding applied after it).  The training forward then needs no statistics-only conv over the
 * expand conv's input -- its batch sums come out of the one pass that writes z (lmn_conv_fwd: LMN_STATS_SUM_SQ with `out`) and
 * lmn_dw_stats finalises them (fin.mode = LMN_FIN_BN: A / shift formed per block, the first block of a channel chunk writes
 * mean / rstd / A / shift and blends the running statistics) -- and the backward forms dh = dx1 * Hardswish'(A z + shift) and
 * its BatchNorm-backward sums where dx1 leaves lmn_dw_bwd_bn.  NULL / A == NULL and fin.mode == NONE: the tensor holds x1.   */
typedef struct {
  const float* A;
  const float* shift;
  lmn_bn_fin_t fin;   /* lmn_dw_stats only */
} lmn_dw_pre_t;
/* batch statistics of the four branch outputs: stats[4][2][E] += (sum y_b, sum y_b^2), b=0..3
 * in the order large(5x5), square(3x3), ver(3x1), hor(1x3); each [2][E] row pair feeds lmn_bn_finalize. */
int lmn_dw_stats(const void* x1, int B, int H, int W, int E, const float* w5, const float* w3, const float* wv,
                 const float* wh, float* stats, const lmn_dw_pre_t* zpre, int act_dtype, lmn_stream_t stream);
/* pre = sum_b A_b * conv_b(x1) + bias  expressed as ONE merged 5x5 stencil keff[E][25] + beff[E]
 * (training: A_b = gamma_b*rstd_b from lmn_dw_stats; eval/deploy: running stats, i.e. exactly
 * ReparamConv.get_equivalent_kernel_bias, core/modules.py:622-642).  Writes `pre` (the GELU
 * input) and accumulates gsum[B][E] += sum_hw gelu(pre) for the SE squeeze (modules.py:1030). */
/* Squeeze-excite gate inside the depthwise forward (core/modules.py:1030-1036: s = Hardsigmoid(fc2(ReLU(fc1(mean_HW g))))):
 * the block that completes an image's GELU sums (an arrival counter per image, `ticket` [B] zeroed by the caller) computes
 * s[B][E] and hidden[B][R] there -- no lmn_se_fwd launch between the depthwise pass and the pointwise conv.  NULL: off.  */
typedef struct {
  uint32_t* ticket;
  const float* w1; const float* b1; const float* w2; const float* b2;   /* fc1 [R][E], [R]; fc2 [E][R], [E] */
  float* s; float* hidden;
  float inv_hw;
  int32_t R;
} lmn_se_fuse_t;
/* Squeeze-excite backward inside lmn_dw_bwd_stats: every block forms dm[b][its channels] = d(loss)/d(mean_HW g) from
 * ds[B][E] (lmn_conv_fwd, LMN_EP_SE_BWD), the saved gate and hidden vectors (lmn_se_bwd_dm arithmetic; E*R MACs per block),
 * the first block of an image writes dvec[B][E+R] for lmn_se_bwd_params.  NULL: dm comes from the caller.                */
typedef struct {
  const float* ds; const float* w1; const float* w2; const float* hidden;
  float* dvec;
  float inv_hw;
  int32_t R;
} lmn_se_bwd_t;
int lmn_dw_fwd(const void* x1, void* pre, float* gsum, int B, int H, int W, int E, const float* keff,
               const float* beff, const lmn_se_fuse_t* se, const lmn_dw_pre_t* zpre, int act_dtype, lmn_stream_t stream);
/* lmn_dw_finalize_merge + lmn_dw_fwd in ONE launch (training): every wave forms the merged stencil of its channel pair from
 * the batch sums `stats` [4][2][E] and the four branch weights, the first block of a channel chunk writes mean / rstd / A
 * [4][E] and updates the running statistics.  Arguments as lmn_dw_finalize_merge; no keff / beff tensors exist.        */
int lmn_dw_fwd_bn(const void* x1, void* pre, float* gsum, int B, int H, int W, int E, const float* stats, float count,
                  const float* const* gamma, const float* const* beta, float* const* running_mean, float* const* running_var,
                  const float* eps, const float* momentum, const float* w5, const float* w3, const float* wv, const float* wh,
                  float* mean, float* rstd, float* A, const lmn_se_fuse_t* se, const lmn_dw_pre_t* zpre, int act_dtype,
                  lmn_stream_t stream);
/* builds keff/beff on the device from the four branch weights and per-branch affine (A_b, shift_b) */
/* Training forward between lmn_dw_stats and lmn_dw_fwd, one launch: finalise the four branch BatchNorms from the
 * batch sums `stats` [4][2][E] (mean/rstd/A [4][E] out, running statistics updated with `momentum`, unbiased variance
 * -- nn.BatchNorm2d in training mode, core/modules.py:548-572) and merge the branches into keff [E][25], beff [E]
 * (get_equivalent_kernel_bias, core/modules.py:622-642).  gamma/beta/running_* are host arrays of 4 device pointers
 * (branch order 5x5, 3x3, 3x1, 1x3); eps/momentum host arrays of 4 floats.                                        */
int lmn_dw_finalize_merge(const float* stats, float count, const float* const* gamma, const float* const* beta,
                          float* const* running_mean, float* const* running_var, const float* eps, const float* momentum,
                          const float* w5, const float* w3, const float* wv, const float* wh, float* mean, float* rstd,
                          float* A, float* keff, float* beff, int E, lmn_stream_t stream);
int lmn_dw_merge(const float* w5, const float* w3, const float* wv, const float* wh, const float* A /*[4][E]*/,
                 const float* shift /*[4][E]*/, float* keff, float* beff, int E, lmn_stream_t stream);
/* backward, pass 1: dpre = (u*s[b,e] + dm[b,e]) * gelu'(pre); writes dpre and accumulates
 * bstats[5][E] += (sum dpre, sum dpre*y_b for the 4 branches).                               */
int lmn_dw_bwd_stats(const void* x1, const void* pre, const void* u, const float* s, const float* dm,
                     void* dpre, int B, int H, int W, int E, const float* w5, const float* w3, const float* wv,
                     const float* wh, float* bstats, const lmn_se_bwd_t* seb, const lmn_dw_pre_t* zpre, int act_dtype,
                     lmn_stream_t stream);
/* per-branch BatchNorm-backward coefficients from bstats (pass 1) and the forward statistics
 * mean/rstd/A ([4][E] each): dgamma_b += T_b, dbeta_b += S0, and f_b = cA*dpre + cC*y_b + cD
 * (batch_stats=0, i.e. eval-mode BN: cC = cD = 0).                                            */
int lmn_dw_bwd_coef(const float* bstats, const float* mean, const float* rstd, const float* A, float count,
                    int batch_stats, float* cA, float* cC, float* cD, float* dg0, float* dg1, float* dg2, float* dg3,
                    float* db0, float* db1, float* db2, float* db3, int E, lmn_stream_t stream);
/* backward, pass 2: f_b = cA[b]*dpre + cC[b]*y_b + cD[b] (inside the image), dx1 = sum_b w_b^T * f_b,
 * dW_b[e][t] += sum_p f_b[p] * x1[p+t]  into the four torch-layout weight gradients.          */
int lmn_dw_bwd(const void* x1, const void* dpre, void* dx1, int B, int H, int W, int E, const float* w5,
               const float* w3, const float* wv, const float* wh, const float* cA, const float* cC,
               const float* cD, float* dw5, float* dw3, float* dwv, float* dwh, int act_dtype, lmn_stream_t stream);

/* lmn_dw_bwd_coef + lmn_dw_bwd in ONE launch: the coefficients of f_b are formed per wave from bstats / mean / rstd / A,
 * the first block of a channel chunk adds the gamma / beta gradients (dgamma / dbeta: host arrays of 4 device pointers). */
int lmn_dw_bwd_bn(const void* x1, const void* dpre, void* dx1, int B, int H, int W, int E, const float* w5, const float* w3,
                  const float* wv, const float* wh, const float* bstats, const float* mean, const float* rstd, const float* A,
                  float count, int batch_stats, float* const* dgamma, float* const* dbeta, float* dw5, float* dw3, float* dwv,
                  float* dwh, int part /* 0: everything; 1: dx1 (+ gamma / beta gradients) only; 2: the four weight gradients only */,
                  const lmn_dw_pre_t* zpre /* z-path: x1 is z, `dx1` receives dh = dx1 * Hardswish'(A z + shift) and           */,
                  float* hstats /* [2][E] += (sum dh, sum dh * z) over all pixels: the expand conv's BatchNorm-backward sums */,
                  int act_dtype, lmn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * SE gate (core/modules.py:1020-1036): s = hardsigmoid(W2 relu(W1 m + b1) + b2), m = gsum/HW.
 * ------------------------------------------------------------------------------------------ */
int lmn_se_fwd(const float* gsum, float inv_hw, const float* w1, const float* b1, const float* w2, const float* b2,
               float* s, float* hidden, int B, int E, int R, lmn_stream_t stream);
/* ds[B][E] (= sum_hw u*g) -> dm[B][E] (already divided by HW) and parameter gradients (+=). */
int lmn_se_bwd(const float* ds, const float* gsum, float inv_hw, const float* w1, const float* b1, const float* w2,
               const float* b2, const float* hidden, float* dm, float* dw1, float* db1, float* dw2, float* db2,
               int B, int E, int R, lmn_stream_t stream);
/* The same backward in two launches (core/modules.py:150-153, autograd of fc1 / ReLU / fc2 / Hardsigmoid):
 * lmn_se_bwd_dm     ds[B][E], the saved gate s[B][E] (hardsigmoid' = 1/6 where 0 < s < 1), hidden[B][R] ->
 *                   dm[B][E] (divided by HW) and dvec[B][E+R] = (d pre-gate | d pre-ReLU) per image; no atomics
 * lmn_se_bwd_params dvec, gsum, hidden -> dw1[R][E], db1[R], dw2[E][R], db2[E] (+=), a batch reduction without atomics;
 *                   independent of everything downstream of dm, i.e. it can run on a side stream. */
int lmn_se_bwd_dm(const float* ds, const float* s, float inv_hw, const float* w1, const float* w2, const float* hidden,
                  float* dm, float* dvec, int B, int E, int R, lmn_stream_t stream);
int lmn_se_bwd_params(const float* dvec, const float* gsum, float inv_hw, const float* hidden, float* dw1, float* db1,
                      float* dw2, float* db2, int B, int E, int R, lmn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Neighborhood attention core (row A7; natten2dqkrpb + softmax + natten2dav fused).
 * qkv: [B,H,W,3C] with channel = which*C + head*hd + d (the layout natten's module gives:
 * reshape(B,H,W,3,heads,hd)); out: [B,H,W,C] channel = head*hd + d; rpb: [heads][2K-1][2K-1].
 * K: natten's kernel_size, odd, 3..13 (window start = clamp(i - K/2, 0, L - K), bias index = neighbour - query + K - 1).
 * The reference constructs K = 3 (core/modules.py:509; the LDS-tiled kernels), its LM_Net signature also carries [3, 5]
 * (core/LM_Net.py:81-84): any other K runs the direct form, and so does K given NEGATIVE (-3: the direct form at K = 3,
 * which the tests compare with the tiled kernels).  scale = hd^-0.5.
 * ------------------------------------------------------------------------------------------ */
int lmn_na_fwd(const void* qkv, const float* rpb, void* out, int B, int H, int W, int heads, int hd, int K, float scale,
               int act_dtype, lmn_stream_t stream);
/* dqkv is fully overwritten (two gather passes, no atomics, deterministic); drpb +=;
 * stat: caller workspace of 2*heads floats per pixel ([B*H*W][2][heads]: log-sum-exp and sum_n p_n dp_n) of the two-pass forms;
 *       the one-pass kernel of head_dim <= 2 (K = 3, C = 12 / 24, maps >= 16 x 16) keeps these in LDS and does not touch it */
int lmn_na_bwd(const void* qkv, const float* rpb, const void* dout, void* dqkv, float* drpb, float* stat, int B,
               int H, int W, int heads, int hd, int K, float scale, int act_dtype, lmn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Dense global attention of GFT (core/modules.py:267-279): qkv [B,N,3C] (channel = which*C +
 * head*hd + d), out [B,N,C].  Any N (streamed over key tiles; exercised up to 16384 tokens = 2048x2048 inputs), hd <= 32.
 * ------------------------------------------------------------------------------------------ */
int lmn_gattn_fwd(const void* qkv, void* out, float* lse, int B, int N, int heads, int hd, float scale, int act_dtype,
                  lmn_stream_t stream);
/* delta: caller workspace [B*heads*N]; dqkv is fully overwritten (no atomics) */
int lmn_gattn_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* delta,
                  int B, int N, int heads, int hd, float scale, int act_dtype, lmn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * LayerNorm over the channel axis of NHWC rows (core/modules.py:330,333,508,511); eps = 1e-5.
 * ------------------------------------------------------------------------------------------ */
int lmn_ln_fwd(const void* x, const float* gamma, const float* beta, void* y, int64_t rows, int C, int act_dtype,
               lmn_stream_t stream);
/* dx = LN backward (+ dres if not NULL: the residual branch's gradient); dgamma/dbeta += */
int lmn_ln_bwd(const void* x, const float* gamma, const void* dy, const void* dres, void* dx, float* dgamma,
               float* dbeta, int64_t rows, int C, int act_dtype, lmn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * BatchNorm(batch stats)+GELU tail of M2Skip/M3Skip.fuse_conv (core/modules.py:96-99,121-123,131-134)
 * on a stored conv output z:  y = gelu(z*a[c] + b[c]).
 * ------------------------------------------------------------------------------------------ */
int lmn_bnact_fwd(const void* z, const float* a, const float* b, void* y, int64_t rows, int C, int act, int act_dtype,
                  lmn_stream_t stream);
/* pass 1: dh = dy*act'(h); stats[2][C] += (sum dh, sum dh*zhat) ; pass 2: dz = c1*dh - c2 - zhat*c3 */
int lmn_bnact_bwd_stats(const void* z, const void* dy, const float* mean, const float* rstd, const float* gamma,
                        const float* beta, float* stats, int64_t rows, int C, int act, int act_dtype, lmn_stream_t stream);
int lmn_bnact_bwd(const void* z, const void* dy, const float* mean, const float* rstd, const float* gamma,
                  const float* beta, const float* c1, const float* c2, const float* c3, void* dz, int64_t rows,
                  int C, int act, int act_dtype, lmn_stream_t stream);
/* The two tails with the tiny launch between statistics and application folded in (lmn_bn_fin_t):
 * lmn_bnact_fwd_fin  fin.mode = LMN_FIN_BN: A / shift are formed from fin.sums by every block exactly as lmn_bn_finalize would (block 0
 *                    stores mean / rstd / A / shift and blends the running statistics; fin.about must not alias fin.rmean);
 * lmn_bnact_bwd_fin  fin.mode = LMN_FIN_BN_BWD: c1 / c2 / c3 are formed from fin.sums (the output of lmn_bnact_bwd_stats) and fin.Ain as
 *                    lmn_bn_bwd_coef would, block 0 adds the gamma / beta gradients.  core/modules.py:83-143 (fuse_conv tails). */
int lmn_bnact_fwd_fin(const void* z, const lmn_bn_fin_t* fin, void* y, int64_t rows, int C, int act, int act_dtype, lmn_stream_t stream);
int lmn_bnact_bwd_fin(const void* z, const void* dy, const float* mean, const float* rstd, const float* gamma,
                      const float* beta, const lmn_bn_fin_t* fin, void* dz, int64_t rows, int C, int act, int act_dtype,
                      lmn_stream_t stream);

/* BatchNorm bookkeeping on [C]-vectors (momentum 0.1, unbiased running var; torch semantics).
 * sums = [2][C] (sum, sumsq) over `count` elements.  Writes mean, rstd (biased var + eps),
 * A = gamma*rstd, shift = beta - mean*A, and updates running_mean/var in place if not NULL.   */
int lmn_bn_finalize(const float* sums, int nrep, float count, const float* gamma, const float* beta, float eps,
                    float momentum, float* mean, float* rstd, float* A, float* shift, float* running_mean,
                    float* running_var, const float* about, int C, lmn_stream_t stream);
/* sums: nrep slices of [2][C] (nrep >= 1), added in double.  about (or NULL): the sums were taken ABOUT about[c] -- sum(v - about),
 * sum((v - about)^2), which is what lmn_conv_fwd(stats_mode=LMN_STATS_SUM_SQ, p4=about) accumulates -- so that the single-pass
 * variance E[d^2] - E[d]^2 cancels numbers of the size of the variance, not of the squared mean (torch: Welford).  It may alias
 * running_mean (read before the in-place update).                                                                              */
/* eval-mode BatchNorm (running statistics): mean, rstd, A = gamma*rstd, shift = beta - mean*A */
int lmn_bn_fold(const float* running_mean, const float* running_var, const float* gamma, const float* beta, float eps,
                float* mean, float* rstd, float* A, float* shift, int C, lmn_stream_t stream);
/* BN backward coefficients from bstats = [2][C] (S0 = sum dh, S1 = sum dh*zhat):
 * dgamma += S1, dbeta += S0, c1 = A, c2 = A*S0/N, c3 = A*S1/N  (batch_stats=0: c2 = c3 = 0)    */
int lmn_bn_bwd_coef(const float* bstats, int nrep, float count, int batch_stats, const float* A, float* dgamma,
                    float* dbeta, float* c1, float* c2, float* c3, int C, lmn_stream_t stream);  /* bstats: nrep slices */

/* ------------------------------------------------------------------------------------------
 * Resampling rows: bilinear x2 upsample with align_corners=True (core/LM_Net.py:59-72,
 * modules.py:94,129) and the exact f x f mean pool of PyramidPool (modules.py:496).
 * ------------------------------------------------------------------------------------------ */
int lmn_up2_fwd(const void* x, void* y, int B, int Hin, int Win, int C, int x_cstride, int y_cstride, int act_dtype,
                lmn_stream_t stream);
int lmn_up2_bwd(const void* dy, void* dx, int B, int Hin, int Win, int C, int dy_cstride, int dx_cstride, int act_dtype,
                lmn_stream_t stream);
int lmn_avgpool_fwd(const void* x, void* y, int B, int Hout, int Wout, int f, int C, int x_cstride,
                    int y_cstride, int act_dtype, lmn_stream_t stream);
/* dx = (accumulate ? dx : 0) + dy/(f*f) broadcast over each f x f window */
int lmn_avgpool_bwd(const void* dy, void* dx, int B, int Hout, int Wout, int f, int C, int dy_cstride,
                    int dx_cstride, int accumulate, int act_dtype, lmn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Boundary layout rows.  The network input arrives NCHW [B,channel,H,W] (core/LM_Net.py:95) and the
 * logits leave NCHW [B,n_classes,H,W] (core/LM_Net.py:122-123); inside, everything is NHWC.  The
 * segmentation head output_layer (1x1 conv, core/LM_Net.py:87) runs on the generic conv with its
 * rows padded to a multiple of 4, then lmn_nhwc_to_nchw keeps the first n_classes channels.
 * ------------------------------------------------------------------------------------------ */
/* y[b,h,w,0:C] = x[b,0:C,h,w]; y[b,h,w,C:y_cstride] = 0 */
int lmn_nchw_to_nhwc(const float* x, void* y, int B, int C, int H, int W, int y_cstride, int act_dtype, lmn_stream_t stream);
/* y[b,0:C,h,w] = x[b,h,w,0:C]  (x has pixel stride x_cstride) */
int lmn_nhwc_to_nchw(const void* x, float* y, int B, int C, int H, int W, int x_cstride, int act_dtype, lmn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Small utilities
 * ------------------------------------------------------------------------------------------ */
/* Segmentation loss around the path on NCHW logits [B][C][HW], int64 labels [B][HW]:
 *   loss = CrossEntropyLoss(weight=w_ce, label_smoothing)(logits, target)            (train.py:157)
 *        + DiceLoss(C)(logits, target, weight=w_dice)  -- softmax, one-hot, per class
 *          1 - (2*sum(p*t) + smooth)/(sum(p^2) + sum(t^2) + smooth), weighted, / C   (utils/loss.py:170-206),
 * as called in utils/train_eval_utils.py:141.  `sums` [3+3C] and `coef` [3+2C] are device workspaces the
 * backward re-uses; `loss` [1] stays on the device (no host sync).  C in {2,3,4,8}.
 * lmn_segloss_bwd writes dlogits = gscale[0] * dloss/dlogits (gscale NULL = 1).                                 */
int lmn_segloss_fwd(const float* logits, const int64_t* target, const float* w_ce, const float* w_dice, int B, int C,
                    int64_t HW, float label_smoothing, float smooth, float* sums, float* coef, float* loss,
                    lmn_stream_t stream);
int lmn_segloss_bwd(const float* logits, const int64_t* target, const float* w_ce, const float* coef, const float* gscale,
                    int B, int C, int64_t HW, float* dlogits, lmn_stream_t stream);
/* On-device confusion matrix (row N2): counts[t*C + p] += #pixels with label t and argmax(logits) = p.  Dice and IoU
 * follow as 2TP/(2TP+FP+FN), TP/(TP+FP+FN) (utils/train_eval_utils.py:78-95).  C in {2,3,4}.                   */
int lmn_confusion(const float* logits, const int64_t* target, int B, int C, int64_t HW, float* counts, lmn_stream_t stream);
/* Device-side input pipeline (row N4): uint8 HWC images [B,Hs,Ws,3] (channel order untouched, as cv2.imread hands it
 * over) and grayscale masks [B,Hs,Ws] -> normalised fp32 NCHW images [B,3,H,W] and int64 labels [B,H,W].
 * Replaces, for data already in HBM, A.Resize + A.Normalize + ToTensorV2 (dataset/data_loading.py:203-206), the mask
 * threshold (:237) and the two flips of the training transform (:213-214): cv2.resize INTER_LINEAR semantics on uint8
 * (11-bit fixed-point coefficients, rounded back to uint8), INTER_NEAREST for masks, (v - mean*255)/(std*255).
 * flips (device, [B], may be NULL): bit 0 horizontal, bit 1 vertical.  mean / std: HOST arrays of 3 doubles (A.Normalize
 * arguments, max_pixel_value 255).  Either images/out or masks/labels may be NULL.                                   */
int lmn_preprocess_u8(const uint8_t* images, const uint8_t* masks, const uint8_t* flips, int B, int Hs, int Ws, int H,
                      int W, const double* mean, const double* std, float* out, int64_t* labels, lmn_stream_t stream);
/* One AdamW step over flat buffers of n floats (n % 4 == 0): replaces torch.optim.AdamW.step() of
 * train.py:156 when parameters and gradients live in the flat layout of lm_net_amd.LM_Net.
 * bias_corr1 = 1 - beta1^t, bias_corr2 = 1 - beta2^t (t = step count, from the host). */
int lmn_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                   float eps, float weight_decay, float bias_corr1, float bias_corr2, lmn_stream_t stream);
int lmn_fill(float* p, float v, int64_t n, lmn_stream_t stream);
/* y = a + b (+ c) (+ d); any of c,d may be NULL; y may alias a */
/* Weight / bias gradient of the expand conv on the z-path WITHOUT materialising dz (no lmn_affine2 pass, no second read of z):
 *   dW_e += diag(a) R + diag(b) (W_e M + b_e m^T) + c m^T ;  db_e += a S0 + b (W_e m + N b_e) + c N
 * R [E][rows] = sum dh x^T (lmn_conv_wgrad over (x, dh)); M [rows][rows] = sum x x^T, m [rows] = sum x (lmn_conv_wgrad over (x, x),
 * issued once in the forward); coef [3][E] and hstats from lmn_reparam_fold / lmn_dw_bwd_bn.  dW [E][cin_w], db [E] (or NULL). */
/* `se` (or NULL): the arguments of lmn_se_bwd_params of the same block -- extra thread blocks of this launch then form the squeeze-excite
 * parameter gradients (both are small batch reductions beside the critical path: one launch instead of two per ReparamConv block). */
typedef struct {
  const float* dvec; const float* gsum; const float* hidden;
  float* dw1; float* db1; float* dw2; float* db2;
  float inv_hw;
  int32_t B, E, R;
} lmn_se_params_t;
int lmn_reparam_wfin(const float* R, const float* M, const float* m, const float* coef, const float* hstats, const float* w_expand,
                     const float* b_expand, float count, int E, int rows, int cin_w, float* dW, float* db, const lmn_se_params_t* se,
                     lmn_stream_t stream);
/* y[p][c] = coef[0][c] * u[p][c] + coef[1][c] * v[p][c] + coef[2][c]  over `rows` pixels of C channels (activation tensors):
 * dz = a * dh + b * z + c of the z-path (lmn_reparam_fold), materialised beside the critical path for the weight gradient.
 * rp_w: 0 = NHWC tensors, W > 0 = all three are row-planar (RP4) with image width W. */
int lmn_affine2(const void* u, const void* v, const float* coef, void* y, int64_t rows, int C, int rp_w, int act_dtype, lmn_stream_t stream);
int lmn_add(const void* a, const void* b, const void* c, const void* d, void* y, int64_t n, int act_dtype,
            lmn_stream_t stream);
/* out[C] += column sums of x[rows][cstride] (bias gradients) */
int lmn_colsum(const void* x, float* out, int64_t rows, int C, int cstride, int act_dtype, lmn_stream_t stream);
/* copy a channel slice: y[rows][y_cstride][0:C] = x[rows][x_cstride][0:C] */
int lmn_copy_slice(const void* x, void* y, int64_t rows, int C, int x_cstride, int y_cstride, int act_dtype,
                   lmn_stream_t stream);

/* y[r][0:cols] = x[r][0:cols] for r < rows, any cols >= 1 (row strides in floats): pads / un-pads the few weights whose
 * channel count is not a multiple of 4 (the RGB input is carried as NHWC4, the 2-class head on 4 rows).            */
int lmn_copy2d(const float* x, float* y, int64_t rows, int cols, int x_stride, int y_stride, lmn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Single-crossing schedule (lmn_plan_create / lmn_forward / lmn_backward).
 * The hot loop this serves is utils/train_eval_utils.py:140-145 (output = model(images); loss.backward()):
 * per input shape, a pass of the network is a FIXED sequence of the entries above over FIXED buffers.  A plan
 * records that sequence once -- while recording, every entry called on this thread runs AND is remembered with its
 * arguments by value (argument structs are copied) -- and lmn_plan_run re-issues it: one FFI crossing per pass (or per
 * data-parallel gradient bucket, see `lo`/`hi`) instead of one per kernel.  Cross-stream dependencies are part of the
 * schedule: lmn_stream_wait(waiter, waited) makes `waiter` wait (hipEventRecord + hipStreamWaitEvent, no host sync) for
 * everything enqueued on `waited` so far, and is recorded like any other entry.  The caller guarantees that every
 * pointer a recorded entry received stays valid and means the same buffer for the life of the plan (lm_net_amd
 * allocates them from one arena per plan).  Thread model: a plan is recorded and run by one thread at a time.
 * ------------------------------------------------------------------------------------------ */
typedef void* lmn_plan_t;
int lmn_stream_wait(lmn_stream_t waiter, lmn_stream_t waited);
/* numbered events (64 slots): mark a point on one stream, make another stream wait for exactly that point (a fork
 * that is joined later, after more work has been queued behind the point); recorded by plans like any entry        */
int lmn_event_record(int slot, lmn_stream_t stream);
int lmn_event_wait(int slot, lmn_stream_t stream);
/* Wave priority of the kernels launched on a stream (process-wide table of 8 streams -- a ninth replaces the oldest entry; level 0
 * clears the entry, 3 is the highest).  The step is as long as the dependent chain of kernels on the caller's stream; the weight-gradient and branch streams have
 * slack.  The kernels launched on a listed stream raise their waves' issue priority (s_setprio level) against the waves of the other
 * streams' kernels on the same CU.  HIP stream priorities are no substitute: they starve the lower queue (INTEGRATION, switches).
 * Not recorded by plans: set it before recording.   (ABI 14)                                                                        */
int lmn_set_priority_stream(lmn_stream_t stream, int level);
/* Deterministic mode (process-wide; default off): every cross-block float reduction (BatchNorm / SE / LayerNorm statistics, bias /
 * gamma / beta / bias-table gradients, depthwise weight gradients, K-split weight-gradient partials) is summed in a FIXED order --
 * per-block partials in private slots of a per-stream scratch, folded by a sum kernel right after the producer -- instead of by
 * float atomics in arrival order: two runs of a step on the same inputs give bit-identical results.  Slower (one or more extra
 * launches per reducing kernel); the scratch is the one buffer the library allocates itself (hipMalloc, per stream, grown on demand). */
int lmn_set_deterministic(int on);
int lmn_get_deterministic(void);
lmn_plan_t lmn_plan_create(void);
int lmn_plan_destroy(lmn_plan_t plan);
int lmn_plan_record_begin(lmn_plan_t plan);              /* start / resume recording on this thread                  */
int64_t lmn_plan_record_end(lmn_plan_t plan, int seal);  /* pause (seal=0) or finish (seal=1); returns #ops recorded  */
int64_t lmn_plan_size(lmn_plan_t plan);
int lmn_plan_run(lmn_plan_t plan, int64_t lo, int64_t hi); /* re-issue ops [lo, hi); hi < 0 = to the end              */
/* diagnostics: run the plan once timing every entry on the HOST clock; text lines "entry \t ops \t total_us" (returns the bytes
 * needed incl. the terminator; -1 on error) */
int64_t lmn_plan_host_profile(lmn_plan_t plan, char* out, int64_t cap);

/* ------------------------------------------------------------------------------------------
 * In-library kernel timer (measurement): HIP events on the launch stream around every kernel launch whose
 * name contains one of the '|'-separated substrings of `filter` (NULL / "" = all kernels).  lmn_prof_end stops the timer,
 * synchronises the device and writes one line per kernel name: "name\tlaunches\ttotal_us\tflops\tbytes\n", where
 * flops / bytes are the ALGORITHMIC costs (layer shapes; SURVEY 8d convention) the conv / depthwise / attention entries
 * declare per launch.  Returns the number of bytes needed for the full report.
 * A filter that starts with '@' reports one line per (kernel, declared cost) -- i.e. per layer shape ("name#flops/bytes");
 * one that starts with '!' reports one line per LAUNCH: "name\tstream\tstart_us\tend_us\t0\n" on the device clock of the
 * first timed launch (events of different streams are comparable: the overlapped timeline of a step).               */
int lmn_prof_begin(const char* filter);
int64_t lmn_prof_end(char* out, int64_t cap);

#ifdef __cplusplus
}
#endif
#endif /* LMNET_HIP_H */
