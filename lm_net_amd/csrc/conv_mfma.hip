// Dense convolution family on the CDNA4 matrix cores -- forward, data-gradient and weight-gradient.
//
// One kernel family serves every nn.Conv2d(k=1|3, stride 1|2) and nn.Linear on the LM-Net path
// (reference call sites listed in include/lmnet_hip.h).  Design (MI355X-first, not a translation):
//
//  * implicit GEMM on v_mfma_f32_16x16x4_f32 (exact fp32, k-ordered fma chain): M = output
//    channels, N = 16 output pixels ("pixel group"), K = (tap, input channel).  With M on the MFMA
//    row index each lane ends up holding 4 CONSECUTIVE output channels of ONE pixel, so the epilogue
//    (bias / BN affine / activation / dropout / residual) and the NHWC store are float4-wide.
//  * no LDS and no barriers: the B operand (activations) is loaded straight from HBM/L2 as float4
//    per lane -- lane (q = lane>>4, n = lane&15) loads channels 16t+4q..+3 of pixel n; element j of
//    that float4 feeds MFMA j of the K16 block, whose k index q therefore means channel 16t+4q+j.
//    The weights are pre-packed (lmn_conv_pack) in exactly that fragment order, so the A operand is
//    one coalesced 1 KiB float4 load per (tap, K16 block, cout tile), served by L1/L2.
//  * a wave owns NPG pixel groups x NCT cout tiles (accumulators in registers, 4 VGPR each) and a
//    contiguous range of group sets, so per-channel statistics (BatchNorm batch stats, SE gradient)
//    are reduced in registers first and hit global atomics once per block.
//  * pixel groups are flattened over one image (never straddle images: the SE scale and the
//    per-image statistics are wave-uniform).  For the data gradient of a stride-2 conv the groups
//    enumerate one parity class of the output at a time, so the valid taps are uniform per group.
#include "common.h"
#include <stdlib.h>

namespace {

struct ConvParams {
  lmn_conv_args_t a;
  int nkb[3];      // K16 blocks per source
  int kb_off[3];   // first K16 block of each source
  int NKB;         // total K16 blocks
  int NCTT;        // total cout tiles
  int ncls;        // 1 or 4 parity classes
  int gpi;         // groups per image
  int ng_c[4];     // groups per class
  int Hc[4], Wc[4];
  int total_sets;
  float inv_keep_ep;      // 1/(1-p) of the epilogue dropout
  float inv_keep_src[3];
  // LDS-tiled kernel (conv_tile_kernel): output tile TH x TW (TP pixels, NG pixel groups), input window XH x XW,
  // LDS pixel stride CS floats, CKB K16-blocks per staged chunk
  int TH, TW, TP, NG, XH, XW, CS, CKB, tiles_x, tiles_y, total_tiles;
  uint32_t mTW, mXW;
  int strided;
  float* det_stats;       // deterministic mode: slot copies of the statistics destination (one slot per block / per wave), else NULL
  LmnLay lay_src[3], lay_out, lay_aux;   // address forms of the operands (common.h: NHWC or row-planar)
  int32_t rpw;                           // image width of the row-planar operands of the call (0: none)
  uint32_t rp_magic;                     // floor(2^32 / rpw)
};

// precision mode of a conv-family kernel instance: 0 = fp32 storage + fp32 MFMA, 1 = fp32 storage + bf16 MFMA operands,
// 2 = bf16 storage + bf16 MFMA operands (accumulators / epilogues / statistics are fp32 in every mode)
template <int PM> struct ActT { typedef float type; };
template <> struct ActT<2> { typedef lmn_bf16 type; };

// ---- bf16 operand forms (mixed-precision path: bf16 MFMA operands, fp32 accumulators / epilogues / statistics)
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_bf16(float a, float b) {  // one v_cvt_pk_bf16_f32 (round to nearest even, NaN kept)
  const f32x2_t t = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(t, bf16x2_t));
}
__device__ __forceinline__ uint2 pk4_bf16(f32x4 v) { return uint2{pk_bf16(v[0], v[1]), pk_bf16(v[2], v[3])}; }
// MFMA operand fragment of one K16 block: fp32 path = 4 floats (one per v_mfma_f32_16x16x4_f32, blocked channel order),
// bf16 path = 4 bf16 (channels 4q..4q+3 of the lane's row, ONE v_mfma_f32_16x16x16_bf16)
template <bool BF> struct Frag { typedef f32x4 type; };
template <> struct Frag<true> { typedef uint2 type; };
template <bool BF> __device__ __forceinline__ typename Frag<BF>::type ldfrag(const float* p) {
  if constexpr (BF) return *reinterpret_cast<const uint2*>(p);
  else return *reinterpret_cast<const f32x4*>(p);
}
__device__ __forceinline__ f32x4 mfma_bf16(uint2 a, uint2 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), c, 0, 0, 0);
}
// Two K16 blocks (or two 16-pixel K steps of a weight gradient) per call: TWO v_mfma_f32_16x16x16_bf16.
// gfx950 has a shape of its own for this, v_mfma_f32_16x16x32_bf16 (twice the K per instruction at the cycles of the 16x16x16 form;
// its k index of lane group q, element j is 8q + j, the fragments carry block 0: channels 4q..4q+3 | block 1: channels 16+4q..16+4q+3
// -- a different bijection k <-> channel, which is fine as long as A and B use the same one), and rounds 2-3 used it (+1.6 % at
// bf16 batch 64).  NOT USED ANY MORE (round 4): kernels issuing it corrupt the results of OTHER kernels running beside them on
// another stream.  Found with the deterministic-mode schedule check in bf16: the neighborhood-attention backward of level 1 came
// out with wrong sum_n p_n dp_n terms for ~0.4 % of the queries (errors of 0.2 on values of order 1: wrong dq / dk, correct dv)
// whenever a bf16 3x3 conv or weight gradient ran on a second stream; reproduced stand-alone (tools/gpu_na_stress2.py: 27 of 30
// runs wrong against a quiet re-run and against the fp32 oracle; the fp32 attention kernel beside a bf16 conv likewise; inputs,
// canaries around the buffers and the kernels' register allocations all intact); 0 of 80 with this function as two 16x16x16
// MFMAs, 0 with fp32 conv kernels, 0 with the bf16 instances that never reach the 32-wide form (one K16 block; the M-split and
// 1x1 instances were not seen to do it either).  -DLMN_MFMA_X2 restores the instruction for experiments.
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x4 mfma_bf16x2(uint2 a0, uint2 a1, uint2 b0, uint2 b1, f32x4 c) {
#ifdef LMN_MFMA_X2
  const uint4 a = uint4{a0.x, a0.y, a1.x, a1.y}, b = uint4{b0.x, b0.y, b1.x, b1.y};
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
#else
  return mfma_bf16(a1, b1, mfma_bf16(a0, b0, c));
#endif
}

// ------------------------------------------------------------------------------------ LDS-tiled forward / data-gradient
// Implicit-GEMM convolution on v_mfma_f32_16x16x4_f32 (M = cout tile: a lane holds 4 consecutive output channels of one
// pixel), structured for latency and registers:
//   * a block walks a contiguous range of output tiles (<= 128 pixels = 8 pixel groups); per tile and per chunk of
//     <= 32 input channels the input WINDOW (with its zero-padded halo) is staged once into LDS by coalesced,
//     unconditional float4 loads with the on-load transforms applied there (once per element, not once per tap);
//   * each wave owns two pixel groups x NCT cout tiles: the B operand is a conflict-free ds_read_b128 (no bounds
//     checks in the MFMA loop: padding is already in LDS), the A operand the packed weight fragment from L1/L2;
//   * ~100 VGPRs instead of 140-256: 4-5 waves per SIMD hide the remaining latency.
// S2T: data gradient of a stride-2 3x3 conv (pad 1).  Output pixels of parity class (py, px) = blockIdx.z form a
// stride-1 problem over dy with a 1x1 / 1x2 / 2x1 / 2x2 sub-kernel: in = out_c + ((parity + pad - t) >> 1) for the taps of
// matching parity.  The tile walks CLASS coordinates, the window is TH+1 x TW+1 pixels of dy, outputs land at
// (2*yc + py, 2*xc + px).
// Epilogue parameter vectors (bias, bias2, p0..p6) of a block's cout range [ct0*16, +NCT*16) -> s_par[9][NCT*16], staged ONCE
// per block: inside the tile loop every one of them was an L2 round trip in a tile's serial chain (window load -> MFMA ->
// parameters -> math -> store), and the run-time branches around them serialised those round trips (LINEAR 49 us,
// AFFINE_ACT 59 us, BN_BWD2 123 us for the same 1x1 conv at level 0).  Absent vectors and channels past Cout read 0.
// fin (lmn_bn_fin_t): the BatchNorm coefficients this conv consumes are formed HERE from the batch sums instead of by a
// separate lmn_bn_finalize / lmn_bn_bwd_coef launch (same arithmetic: slices summed in double / float); `first` blocks
// (blockIdx.x == 0: one per cout range) also write the side outputs -- saved mean / rstd / A / shift and the running
// statistics, or the gamma / beta gradients.  stats_snap: the shift vector p4 of a SUM_SQ statistics pass is copied behind
// the slices, a copy that stays valid while the consuming pass updates the running mean it came from.
template <int NCT>
__device__ __forceinline__ void conv_stage_params(const lmn_conv_args_t& A, float* s_par, int ct0, int tid, bool first) {
  const float* const pv[9] = {A.bias, A.bias2, A.p0, A.p1, A.p2, A.p3, A.p4, A.p5, A.p6};
  const lmn_bn_fin_t& F = A.fin;
  for (int i = tid; i < NCT * 16; i += 256) {   // (NCT * 16 <= 256: one trip; all loads of it are in flight together)
    const int co = ct0 * 16 + i;
    const bool cok = co < A.Cout;
    const int cs = cok ? co : 0;
    float t[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) t[k] = (pv[k] ? pv[k] : A.wpack)[pv[k] ? cs : 0];  // absent vector: any valid address
#pragma unroll
    for (int k = 0; k < 9; ++k) t[k] = (pv[k] && cok) ? t[k] : 0.f;
    if (F.mode != LMN_FIN_NONE) {  // block-uniform
      float u0[16], u1[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rs = r < F.nrep ? r : 0;
        u0[r] = F.sums[(int64_t)rs * 2 * A.Cout + cs];
        u1[r] = F.sums[(int64_t)rs * 2 * A.Cout + A.Cout + cs];
      }
      if (F.mode == LMN_FIN_BN) {
        const float ab = F.about ? F.about[cs] : 0.f, ga = F.gamma[cs], be = F.beta[cs];
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (r < F.nrep) { s0 += (double)u0[r]; s1 += (double)u1[r]; }
        const double md = s0 / (double)F.count;
        float var = (float)(s1 / (double)F.count - md * md);  // biased
        var = var > 0.f ? var : 0.f;
        const float m = (float)md + ab;
        const float rs = rsqrtf(var + F.eps);
        const float a = ga * rs, sh = be - m * a;
        t[2] = cok ? a : 0.f;
        t[3] = cok ? sh : 0.f;
        if (first && cok) {
          if (F.mean) F.mean[co] = m;
          if (F.rstd) F.rstd[co] = rs;
          if (F.A) F.A[co] = a;
          if (F.shift) F.shift[co] = sh;
          if (F.rmean) F.rmean[co] = (1.f - F.momentum) * F.rmean[co] + F.momentum * m;
          if (F.rvar) F.rvar[co] = (1.f - F.momentum) * F.rvar[co] + F.momentum * var * (F.count > 1.f ? F.count / (F.count - 1.f) : 1.f);
        }
      } else {  // LMN_FIN_BN_BWD
        float S0 = 0.f, S1 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (r < F.nrep) { S0 += u0[r]; S1 += u1[r]; }
        const float a = F.Ain[cs];
        t[4] = cok ? a : 0.f;
        t[5] = (cok && F.batch_stats) ? a * S0 / F.count : 0.f;
        t[6] = (cok && F.batch_stats) ? a * S1 / F.count : 0.f;
        if (first && cok) {
          if (F.dgamma) F.dgamma[co] += S1;
          if (F.dbeta) F.dbeta[co] += S0;
        }
      }
    }
    if (A.stats_snap && first && cok && A.stats) A.stats[(int64_t)A.stats_rep * 2 * A.Cout + co] = t[6];
    t[0] += t[1];  // slot 0 = bias + bias2: the accumulators of a tile START from it (no zeroing, no bias add in the epilogue)
#pragma unroll
    for (int k = 0; k < 9; ++k) s_par[k * NCT * 16 + i] = t[k];
  }
}

#ifdef LMN_CT_TIMING
// phase clocks of conv_tile_kernel (debug builds): per block {barrier 1, staging, barrier 2, MFMA, epilogue, life, start, end}
__device__ unsigned long long g_ct_timing[4096 * 8];
#endif
#define LMN_SLOT c
// WL (3x3): the packed weight fragments of a chunk (9 taps x <= 2 K16 blocks x NCT cout tiles) are staged in LDS -- once per
// block when the layer is a single chunk -- and the MFMA loop reads its A operands with ds_read_b128.  Fetched from L2 one
// step ahead, a fragment had 6-16 MFMAs (200-500 cycles) to arrive in: every step of the small-channel layers stalled on it
// (phase clocks: the MFMA loop was 52-61 % of a block's life at 35 % MFMA-pipe use).
template <int TAPS, int NCT, int EPI, bool S2T = false, int PM = 0, bool WL = false>
__global__ __launch_bounds__(256) void conv_tile_kernel(const ConvParams P) {
  static_assert(!WL || (TAPS == 9 && !S2T), "LDS-staged weights: 3x3 stride-1 windows");
  // PM bit 2 (RP): some operand of this 1x1 call is ROW-PLANAR (common.h LmnLay) -- a separate instantiation, so that the NHWC
  // instances keep their registers (the layout arithmetic in every instance cost the SE-gradient conv its fifth wave per SIMD)
  constexpr int PMB = PM & 3;
  constexpr bool RP = (PM & 4) != 0;
  static_assert(!RP || TAPS == 1, "row-planar operands: 1x1 convs");
  constexpr bool BF = PMB >= 1;
  typedef typename ActT<PMB>::type TA;   // activation storage type
  // BF: operands rounded to bf16 when they are staged / packed, v_mfma_f32_16x16x16_bf16 (8x the fp32 MFMA rate), the LDS
  // window holds 4-bf16 fragments (pixel stride P.CS dwords = 8 per K16 block + 4: conflict-free ds_read_b64)
  typedef typename Frag<BF>::type wfrag;
  constexpr int WT = BF ? 128 : 256;   // floats per packed weight fragment tile (64 lanes x 8 or 16 bytes)
  constexpr int KD = BF ? 8 : 16;      // LDS dwords per K16 block of a pixel
  const lmn_conv_args_t& A = P.a;
  const uint32_t soff = A.seed_ctr ? *A.seed_ctr : 0u;  // device-side dropout stream offset (graph replays: one bump per step)
  // EPI: 0 plain (LINEAR / AFFINE_ACT, no statistics), 1 generic, 2 LINEAR + SUM_SQ statistics, 3 BN_BWD1, 4 BN_BWD2,
  // 5 SE_BWD.  For EPI >= 2 the epilogue kind is a compile-time constant: each instance carries only its own code
  // (the generic instance keeps every variant resident: 125-160 VGPRs + spills, and measured 20-40 us over its
  // memory time at level 0).
  const int ep_kind = EPI == 2 ? LMN_EP_LINEAR : EPI == 3 ? LMN_EP_BN_BWD1 : EPI == 4 ? LMN_EP_BN_BWD2 : EPI == 5 ? LMN_EP_SE_BWD : A.epilogue;
  const int st_mode = EPI == 2 ? LMN_STATS_SUM_SQ : (EPI == 3 || EPI == 5) ? LMN_STATS_EP : EPI == 4 ? LMN_STATS_NONE : A.stats_mode;
  const bool has_drop = EPI <= 1 && A.drop_p > 0.f;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* XS = smem;                       // [XH*XW][CS]
  float* s_stats = smem + P.XH * P.XW * P.CS;  // [2][NCT*16]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave index as an SGPR: branches on it stay scalar
  const int q = lane >> 4, n = lane & 15;
  const int ct0 = blockIdx.y * NCT;
  const int pad = A.ksize >> 1;
  constexpr int KS = TAPS == 9 ? 3 : 1;
  // S2T: this class's taps (weight tap index, window offset in pixels)
  const int cpy = S2T ? (int)(blockIdx.z >> 1) : 0, cpx = S2T ? (int)(blockIdx.z & 1) : 0;
  int s2_wt[4], s2_off[4], s2_n = 0;
  if constexpr (S2T) {
#pragma unroll
    for (int k = 0; k < 4; ++k) s2_wt[k] = s2_off[k] = 0;
    for (int ty = 0; ty < 3; ++ty)
      for (int tx = 0; tx < 3; ++tx) {
        const int ey = cpy + 1 - ty, ex = cpx + 1 - tx;
        if ((ey | ex) & 1) continue;
        s2_wt[s2_n] = ty * 3 + tx;
        s2_off[s2_n] = (ey >> 1) * P.XW + (ex >> 1);
        ++s2_n;
      }
  }
  const float* wlane = A.wpack + lane * (BF ? 2 : 4);
  int wtile[NCT];  // cout tiles past the end re-read the last real tile (results dropped in the epilogue)
#pragma unroll
  for (int c = 0; c < NCT; ++c) wtile[c] = min(ct0 + c, P.NCTT - 1) * WT;

  for (int i = tid; i < 2 * NCT * 16; i += 256) s_stats[i] = 0.f;
  float* s_par = s_stats + 2 * NCT * 16;  // [9][NCT*16]
  conv_stage_params<NCT>(A, s_par, ct0, tid, blockIdx.x == 0 && blockIdx.z == 0);  // (one writer per cout range, also with parity classes in grid.z)
  __syncthreads();  // the tiles read s_par from their first instruction on (accumulators start from the bias)
  float* s_w = s_par + 9 * NCT * 16;  // WL: [tap * nkbc + kk][NCT][WT]
  auto stage_w = [&](int s, int kb0, int nkbc) __attribute__((always_inline)) {
    constexpr int Q = WT / 4;  // float4 per fragment tile (64 lanes x 16 or 8 bytes)
    const int ksh = nkbc - 1;
    for (int i = tid; i < TAPS * nkbc * NCT * Q; i += 256) {
      const int t = i / Q, l = i - t * Q;
      const int tk = t / NCT, c = t - tk * NCT;
      const int tap = tk >> ksh, kk = tk & ksh;
      const float* src = A.wpack + (((int64_t)tap * P.NKB + P.kb_off[s] + kb0 + kk) * P.NCTT + min(ct0 + c, P.NCTT - 1)) * WT + l * 4;
      *reinterpret_cast<f32x4*>(&s_w[t * WT + l * 4]) = *reinterpret_cast<const f32x4*>(src);
    }
  };
  const bool wonce = WL && A.nsrc == 1 && P.nkb[0] <= P.CKB;  // single-chunk layer: the block's weights are staged once
  if (wonce) stage_w(0, 0, P.nkb[0]);
  float st0[NCT][4], st1[NCT][4];
#pragma unroll
  for (int c = 0; c < NCT; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) st0[c][r] = st1[c][r] = 0.f;
  int cur_b = -1;  // image whose SE_BWD sums are in st0

  // tiles of a block: contiguous range [t_begin, t_end) (tstep 1), or every gridDim.x-th tile (P.strided: all blocks
  // get floor or ceil of the average and the surplus lands on the first-dispatched blocks, one per CU)
  const int t_begin = P.strided ? (int)blockIdx.x : (int)(((int64_t)blockIdx.x * P.total_tiles) / gridDim.x);
  const int t_end = P.strided ? P.total_tiles : (int)(((int64_t)(blockIdx.x + 1) * P.total_tiles) / gridDim.x);
  const int tstep = P.strided ? (int)gridDim.x : 1;
#ifdef LMN_CT_TIMING
  unsigned long long tk0 = __builtin_amdgcn_s_memtime(), tk_s[5] = {0, 0, 0, 0, 0}, tk_a = tk0, tk_b;
#define LMN_TK(i) do { tk_b = __builtin_amdgcn_s_memtime(); tk_s[i] += tk_b - tk_a; tk_a = tk_b; } while (0)
#define LMN_TK_DRAIN() __builtin_amdgcn_s_waitcnt(0)
#else
#define LMN_TK(i) do { } while (0)
#define LMN_TK_DRAIN() do { } while (0)
#endif
  for (int tile = t_begin; tile < t_end; tile += tstep) {
#ifdef LMN_CT_TIMING
    tk_a = __builtin_amdgcn_s_memtime();
#endif
    // (1x1: the host flattens the image to ONE row of H*W pixels -- tiles_y == 1, TH == 1, no padding: the row / window terms of
    //  the general index math are dropped at compile time; these kernels are instruction-issue bound, PMC: some instruction of the
    //  SIMD's five waves active 69 % of the time, 190 VALU + 215 SALU per wave and 128-pixel tile before this)
    const int b = TAPS == 1 ? tile / P.tiles_x : tile / (P.tiles_x * P.tiles_y);
    const int tt = TAPS == 1 ? tile - b * P.tiles_x : tile - b * P.tiles_x * P.tiles_y;
    const int oy0 = TAPS == 1 ? 0 : (tt / P.tiles_x) * P.TH, ox0 = TAPS == 1 ? tt * P.TW : (tt % P.tiles_x) * P.TW;
    // window origin in input coordinates (forward: out*s - pad; data gradient, stride 1: out - pad, taps flipped)
    const int wy0 = S2T ? oy0 : (A.transposed ? oy0 - pad : oy0 * A.stride - pad);
    const int wx0 = S2T ? ox0 : (A.transposed ? ox0 - pad : ox0 * A.stride - pad);

    if (EPI && ep_kind == LMN_EP_SE_BWD && b != cur_b) {  // block-uniform: flush the previous image's sums
      if (cur_b >= 0) {
#pragma unroll
        for (int c = 0; c < NCT; ++c)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float t = st0[c][r];
            t += __shfl_xor(t, 1, 64); t += __shfl_xor(t, 2, 64); t += __shfl_xor(t, 4, 64); t += __shfl_xor(t, 8, 64);
            const int co = (ct0 + c) * 16 + q * 4 + r;
            if (n == 0 && (ct0 + c < P.NCTT) && co < A.Cout)   // (deterministic mode: slot of this wave, [B][Cout] per slot)
              lmn_red_add((P.det_stats ? P.det_stats + (int64_t)((blockIdx.x + gridDim.x * blockIdx.z) * 4 + wv) * A.B * A.Cout : A.stats) + cur_b * A.Cout + co, t, P.det_stats != nullptr);
            st0[c][r] = 0.f;
          }
      }
      cur_b = b;
    }

    // this wave's two pixel groups: tile pixel -> (row, col), LDS base address, validity
    int pbase[2], opix[2];
    bool pvalid[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int i = (wv + 4 * g) * 16 + n;
      const bool in_t = i < P.TP;
      const int is = in_t ? i : 0;
      const int r = TAPS == 1 ? 0 : (int)__umulhi((uint32_t)is, P.mTW), c = is - r * P.TW;
      const int oy = S2T ? 2 * (oy0 + r) + cpy : oy0 + r, ox = S2T ? 2 * (ox0 + c) + cpx : ox0 + c;
      pvalid[g] = in_t && oy < A.Hout && ox < A.Wout;
      opix[g] = pvalid[g] ? (b * A.Hout + oy) * A.Wout + ox : 0;
      const int sr = (S2T || A.transposed) ? r : r * A.stride, sc = (S2T || A.transposed) ? c : c * A.stride;
      pbase[g] = (sr * P.XW + sc) * P.CS + q * (BF ? 2 : 4);
    }
    const bool g1 = (wv + 4) < P.NG;  // wave-uniform: second group exists

    f32x4 acc[2][NCT];  // start from bias (+ bias2): 4 channels q*4.. of cout tile c, the same for both pixel groups
#pragma unroll
    for (int c = 0; c < NCT; ++c) acc[0][c] = acc[1][c] = *reinterpret_cast<const f32x4*>(s_par + c * 16 + q * 4);
    // BN_BWD instances: the epilogue's second operand (aux: an output-shaped tensor) is requested HERE, ahead of the staging
    // barrier -- its HBM latency runs beside that of the window instead of after the MFMAs (two exposed round trips per
    // tile -> one; level 0, cold operands: BN_BWD1 102 -> 88 us, BN_BWD2 122 -> 105 us; SE_BWD measured slower with it)
    constexpr bool AUXP = (EPI == 3 && NCT <= 2) || EPI == 4;  // (three cout tiles: the 12 prefetch registers cost BN_BWD1 a wave per SIMD -- 137 -> 122 VGPRs, 38.9 -> 32.4 us at level 1)
    f32x4 axp[AUXP ? 2 : 1][AUXP ? NCT : 1];
    if constexpr (AUXP) {
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        uint32_t oa = 0;
        if constexpr (RP) {   // pixel part of the aux offset (common.h LmnLay)
          oa = (uint32_t)opix[g] * (uint32_t)P.lay_aux.cs;
          if (P.lay_aux.rf) oa += lmn_div_row((uint32_t)opix[g], (uint32_t)P.rpw, P.rp_magic) * (uint32_t)P.lay_aux.rf;
        }
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
          const int co = (ct0 + c) * 16 + q * 4;
          const int cos = ((ct0 + c < P.NCTT) && co < A.Cout) ? co : 0;
          if constexpr (RP) axp[g][c] = A.aux ? ld4((const TA*)A.aux + (oa + (uint32_t)(cos >> 2) * (uint32_t)P.lay_aux.qs)) : f32x4{0.f, 0.f, 0.f, 0.f};
          else axp[g][c] = A.aux ? ld4((const TA*)A.aux + (uint32_t)opix[g] * A.aux_cstride + cos) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
    }

    for (int s = 0; s < A.nsrc; ++s) {
      const lmn_src_t& S = A.src[s];
      const LmnLay LS = P.lay_src[s];   // (used by the row-planar instances only)
      (void)LS;
      for (int kb0 = 0; kb0 < P.nkb[s]; kb0 += P.CKB) {
        const int nkbc = P.nkb[s] - kb0 < P.CKB ? P.nkb[s] - kb0 : P.CKB;
        const int ksh = nkbc - 1, niter = (S2T ? s2_n : TAPS) * nkbc;  // step it = (tap, kk): tap = it >> ksh, kk = it & ksh (nkbc is 1 or 2)
        wfrag wcur[NCT];
        if constexpr (!WL) {
          const float* wp = wlane + (((int64_t)(S2T ? s2_wt[0] : 0) * P.NKB + P.kb_off[s] + kb0) * P.NCTT) * WT;
#pragma unroll
          for (int c = 0; c < NCT; ++c) wcur[c] = ldfrag<BF>(wp + wtile[c]);
        }
        __syncthreads();  // previous chunk / tile fully consumed
        LMN_TK(0);
        if constexpr (WL) {
          if (!wonce) stage_w(s, kb0, nkbc);
        }
        // ---- stage the window chunk: unconditional float4 loads from clamped addresses, transforms, zero padding
        //      Rounds of SU x 256 items: ALL loads of a round are issued before the first is consumed (the plain loop compiled
        //      to load -> s_waitcnt vmcnt(0) -> write per item, i.e. 3-6 serial HBM round trips per 3x3 chunk).
        const int psh = nkbc == 2 ? 3 : 2, per_px = 1 << psh;   // quads per pixel: 4 (one K16 block) or 8
        const int nitems = P.XH * P.XW * per_px;
        constexpr int SU = TAPS == 1 ? 2 : 4;  // items per thread and round.  1x1: four in flight cost the epilogue-heavy instances a wave per
                                               // SIMD; two (+6 VGPRs, same occupancy bracket for all but <1,3,0>) halve the 4-6 serial round
                                               // trips of the 24-48 channel layers: +0.3 % fp32 batch 8, +1.6 % bf16 batch 64
        for (int i0 = 0; i0 < nitems; i0 += SU * 256) {
          f32x4 sv[SU];
          int sgp[SU];  // global pixel index, -1 = outside the image / past the channels / past the window
#pragma unroll
          for (int u = 0; u < SU; ++u) {
            const int i = i0 + u * 256 + tid;
            const int f = i & (per_px - 1), pix = i >> psh;
            const int r = TAPS == 1 ? 0 : (int)__umulhi((uint32_t)pix, P.mXW), c = pix - r * P.XW;
            const int iy = wy0 + r, ix = wx0 + c;
            const int ch = kb0 * 16 + f * 4;
            const bool ok = i < nitems && ch < S.C && (TAPS == 1 || (unsigned)iy < (unsigned)A.Hin) && (unsigned)ix < (unsigned)A.Win;
            const int gp = ok ? (TAPS == 1 ? b * A.Win + ix : (b * A.Hin + iy) * A.Win + ix) : 0;
            sgp[u] = ok ? gp : -1;
            if constexpr (RP) {
              uint32_t so_ = (uint32_t)(gp * LS.cs) + (uint32_t)((ok ? ch : 0) >> 2) * (uint32_t)LS.qs;
              if (LS.rf) so_ += lmn_div_row((uint32_t)gp, (uint32_t)P.rpw, P.rp_magic) * (uint32_t)LS.rf;   // (wave-uniform: row-planar source)
              sv[u] = ld4((const TA*)S.ptr + so_);
            } else {
              sv[u] = ld4((const TA*)S.ptr + (uint32_t)(gp * S.cstride + (ok ? ch : 0)));
            }
          }
#pragma unroll
          for (int u = 0; u < SU; ++u) {
            const int i = i0 + u * 256 + tid;
            if (i >= nitems) continue;
            const int f = i & (per_px - 1), pix = i >> psh;
            const bool ok = sgp[u] >= 0;
            const int gp = ok ? sgp[u] : 0, chs = ok ? kb0 * 16 + f * 4 : 0;
            f32x4 v = sv[u];
            if (S.flags & LMN_SRC_GELU) {
#pragma unroll
              for (int k = 0; k < 4; ++k) v[k] = lmn_gelu(v[k]);
            }
            if (S.flags & LMN_SRC_DROP) {
#pragma unroll
              for (int k = 0; k < 4; ++k) v[k] *= lmn_drop_scale(S.drop_seed + soff, (uint32_t)(gp * S.C + chs + k), S.drop_p, P.inv_keep_src[s]);
            }
            if (S.scale) v *= ld4(S.scale + b * S.C + chs);
            if (!ok) v = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (BF) {  // natural channel order, 4 bf16 = one 8-byte fragment slot
              *reinterpret_cast<uint2*>(&XS[pix * P.CS + f * 2]) = pk4_bf16(v);
            } else {  // blocked K layout: channel 4*fl + r of its K16 block sits at position 4*r + fl (MFMA j reads channel 4j + q)
              float* d = &XS[pix * P.CS + (f >> 2) * 16 + (f & 3)];
              d[0] = v[0]; d[4] = v[1]; d[8] = v[2]; d[12] = v[3];
            }
          }
        }
        LMN_TK_DRAIN();
        LMN_TK(1);
        __syncthreads();
        LMN_TK(2);
        // ---- MFMA: taps x K16 blocks of the chunk; the packed weights of step it+1 are fetched while step it runs
        //      (the first fetch was issued before the staging loop), so no L2 latency is exposed inside the loop
        // 3x3: the pixel operands of step it+1 are read from LDS while the MFMAs of step it run (cold operands, 24 -> 24 at level 1:
        // 51 -> 46 us, (24,24,24) -> 24: 134 -> 120 us; the one-cout-tile layers with LDS-resident weights lose 6 % to the 8
        // extra registers and keep the plain order)
        constexpr bool XPF = TAPS == 9 && (!WL || NCT >= 2);
        wfrag xq0, xq1;
        if constexpr (XPF) {
          const int fy0 = A.transposed ? KS - 1 : 0, fx0 = A.transposed ? KS - 1 : 0;
          const int toff0 = S2T ? s2_off[0] * P.CS : (fy0 * P.XW + fx0) * P.CS;
          xq0 = ldfrag<BF>(&XS[pbase[0] + toff0]);
          xq1 = ldfrag<BF>(&XS[pbase[1] + toff0]);
        }
        if constexpr (BF) {
          if (nkbc == 2) {   // wave-uniform: both K16 blocks of a tap in ONE v_mfma_f32_16x16x32_bf16 (see mfma_bf16x2)
            const int ntap = S2T ? s2_n : TAPS;
            uint2 wa[NCT], wb[NCT];
            if constexpr (!WL) {
              const float* wp1 = wlane + (((int64_t)(S2T ? s2_wt[0] : 0) * P.NKB + P.kb_off[s] + kb0 + 1) * P.NCTT) * WT;
#pragma unroll
              for (int c = 0; c < NCT; ++c) { wa[c] = wcur[c]; wb[c] = ldfrag<true>(wp1 + wtile[c]); }
            }
            for (int t = 0; t < ntap; ++t) {
              const int tn = t + 1 < ntap ? t + 1 : t;
              uint2 wna[NCT], wnb[NCT];
              if constexpr (WL) {
#pragma unroll
                for (int c = 0; c < NCT; ++c) {
                  wa[c] = ldfrag<true>(&s_w[((2 * t) * NCT + c) * WT + lane * 2]);
                  wb[c] = ldfrag<true>(&s_w[((2 * t + 1) * NCT + c) * WT + lane * 2]);
                }
              } else {
                const int tapn = S2T ? s2_wt[tn & 3] : tn;
                const float* wp = wlane + (((int64_t)tapn * P.NKB + P.kb_off[s] + kb0) * P.NCTT) * WT;
#pragma unroll
                for (int c = 0; c < NCT; ++c) { wna[c] = ldfrag<true>(wp + wtile[c]); wnb[c] = ldfrag<true>(wp + P.NCTT * WT + wtile[c]); }
              }
              const int ty = t / KS, tx = t - ty * KS;
              const int fy = A.transposed ? KS - 1 - ty : ty, fx = A.transposed ? KS - 1 - tx : tx;
              const int toff = S2T ? s2_off[t & 3] * P.CS : (fy * P.XW + fx) * P.CS;
              const uint2 xa0 = ldfrag<true>(&XS[pbase[0] + toff]), xb0 = ldfrag<true>(&XS[pbase[0] + toff + KD]);
              const uint2 xa1 = ldfrag<true>(&XS[pbase[1] + toff]), xb1 = ldfrag<true>(&XS[pbase[1] + toff + KD]);
#pragma unroll
              for (int c = 0; c < NCT; ++c) {
                acc[0][c] = mfma_bf16x2(wa[c], wb[c], xa0, xb0, acc[0][c]);
                acc[1][c] = mfma_bf16x2(wa[c], wb[c], xa1, xb1, acc[1][c]);
              }
              if constexpr (!WL) {
#pragma unroll
                for (int c = 0; c < NCT; ++c) { wa[c] = wna[c]; wb[c] = wnb[c]; }
              }
            }
            continue;
          }
        }
        for (int it = 0; it < niter; ++it) {
          const int itn = it + 1 < niter ? it + 1 : it;
          wfrag wnext[NCT];
          if constexpr (WL) {
#pragma unroll
            for (int c = 0; c < NCT; ++c) wcur[c] = ldfrag<BF>(&s_w[(it * NCT + c) * WT + lane * (BF ? 2 : 4)]);
          } else {
            const int tapi = itn >> ksh, kkn = itn & ksh;
            const int tapn = S2T ? s2_wt[tapi & 3] : tapi;
            const float* wp = wlane + (((int64_t)tapn * P.NKB + P.kb_off[s] + kb0 + kkn) * P.NCTT) * WT;
#pragma unroll
            for (int c = 0; c < NCT; ++c) wnext[c] = ldfrag<BF>(wp + wtile[c]);
          }
          const int tap = it >> ksh, kk = it & ksh;
          const int ty = tap / KS, tx = tap - ty * KS;
          const int fy = A.transposed ? KS - 1 - ty : ty, fx = A.transposed ? KS - 1 - tx : tx;
          const int toff = S2T ? s2_off[tap & 3] * P.CS : (fy * P.XW + fx) * P.CS;
          wfrag x0, x1;
          if constexpr (XPF) {
            x0 = xq0; x1 = xq1;
            const int tapq = itn >> ksh, kkq = itn & ksh;
            const int tyq = tapq / KS, txq = tapq - tyq * KS;
            const int fyq = A.transposed ? KS - 1 - tyq : tyq, fxq = A.transposed ? KS - 1 - txq : txq;
            const int toffq = S2T ? s2_off[tapq & 3] * P.CS : (fyq * P.XW + fxq) * P.CS;
            xq0 = ldfrag<BF>(&XS[pbase[0] + toffq + kkq * KD]);
            xq1 = ldfrag<BF>(&XS[pbase[1] + toffq + kkq * KD]);
          } else {
            x0 = ldfrag<BF>(&XS[pbase[0] + toff + kk * KD]);
            x1 = ldfrag<BF>(&XS[pbase[1] + toff + kk * KD]);
          }
          if constexpr (BF) {
#pragma unroll
            for (int c = 0; c < NCT; ++c) {
              acc[0][c] = mfma_bf16(wcur[c], x0, acc[0][c]);
              acc[1][c] = mfma_bf16(wcur[c], x1, acc[1][c]);
            }
          } else {
          const int nj = (S.C - (kb0 + kk) * 16 + 3) >> 2;  // K slices of this block that hold channels (wave-uniform)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (j == 0 || j < nj) {
#pragma unroll
              for (int c = 0; c < NCT; ++c) {
                // both groups unconditionally (a missing second group reads pixel 0 and is dropped in the epilogue)
                acc[0][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wcur[c][j], x0[j], acc[0][c], 0, 0, 0);
                acc[1][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wcur[c][j], x1[j], acc[1][c], 0, 0, 0);
              }
            }
          }
          }
          if constexpr (!WL) {
#pragma unroll
            for (int c = 0; c < NCT; ++c) wcur[c] = wnext[c];
          }
        }
      }
    }

    LMN_TK_DRAIN();
    LMN_TK(3);
    // ---- epilogue (lane holds channels co..co+3 of its pixel)
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const uint32_t opx = (uint32_t)opix[g];
      uint32_t oout_g = 0, oaux_g = 0;   // row-planar instances: pixel part of the out / aux offsets (LmnLay)
      if constexpr (RP) {
        const uint32_t rowp = lmn_div_row(opx, (uint32_t)P.rpw, P.rp_magic);
        oout_g = opx * (uint32_t)P.lay_out.cs + rowp * (uint32_t)P.lay_out.rf;
        oaux_g = opx * (uint32_t)P.lay_aux.cs + rowp * (uint32_t)P.lay_aux.rf;
      }
      (void)oout_g; (void)oaux_g;
#pragma unroll
      for (int c = 0; c < NCT; ++c) {
        const int co = (ct0 + c) * 16 + q * 4;
        const bool cok = (ct0 + c < P.NCTT) && co < A.Cout;
        const bool live = pvalid[g] && cok && (g == 0 || g1);
        const int cos = cok ? co : 0;
        f32x4 v = acc[g][c];
        const float* sp = s_par + LMN_SLOT * 16 + q * 4;
#define LMN_PAR(k) (*reinterpret_cast<const f32x4*>(sp + (k) * NCT * 16))
        f32x4 o = v;
        if (EPI && st_mode == LMN_STATS_SUM_SQ && live) {
          // sums about p4[co] when given (the BatchNorm's running mean): E[x^2] - E[x]^2 then subtracts numbers of the
          // size of the variance, not of the squared mean
          const f32x4 sh = LMN_PAR(6);
#pragma unroll
          for (int r = 0; r < 4; ++r) { const float d = v[r] - sh[r]; st0[c][r] += d; st1[c][r] += d * d; }
        }
        if (ep_kind == LMN_EP_AFFINE_ACT) {
          const f32x4 s0 = LMN_PAR(2), s1 = LMN_PAR(3);
          o = lmn_act4(v * s0 + s1, A.act);
        }
        if (EPI) {
          f32x4 ax = f32x4{0.f, 0.f, 0.f, 0.f};
          if constexpr (AUXP) ax = axp[g][c];
          else if (A.aux) { if constexpr (RP) ax = ld4((const TA*)A.aux + (oaux_g + (uint32_t)(cos >> 2) * (uint32_t)P.lay_aux.qs)); else ax = ld4((const TA*)A.aux + opx * A.aux_cstride + cos); }
          switch (ep_kind) {
            case LMN_EP_DGELU: {
#pragma unroll
              for (int r = 0; r < 4; ++r) o[r] = v[r] * lmn_dgelu(ax[r]);
            } break;
            case LMN_EP_BN_BWD1: {
              const f32x4 mu = LMN_PAR(2), rs = LMN_PAR(3), ga = LMN_PAR(4), be = LMN_PAR(5);
              const f32x4 zh = (v - mu) * rs;
              o = ax * lmn_dact4(ga * zh + be, A.act);
              if (live) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { st0[c][r] += o[r]; st1[c][r] += o[r] * zh[r]; }
              }
            } break;
            case LMN_EP_BN_BWD2: {
              const f32x4 mu = LMN_PAR(2), rs = LMN_PAR(3), c1 = LMN_PAR(4), c2 = LMN_PAR(5), c3 = LMN_PAR(6);
              if (A.p5) {  // aux is the gradient w.r.t. the ACTIVATED output: dh = aux * act'(gamma*zh + beta) formed here
                const f32x4 ga = LMN_PAR(7), be = LMN_PAR(8);
                const f32x4 zh = (v - mu) * rs;
                o = c1 * (ax * lmn_dact4(ga * zh + be, A.act)) - c2 - zh * c3;
              } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = c1[r] * ax[r] - c2[r] - (v[r] - mu[r]) * rs[r] * c3[r];
              }
            } break;
            case LMN_EP_SE_BWD: {
              if (live) {
#pragma unroll
                for (int r = 0; r < 4; ++r) st0[c][r] += v[r] * lmn_gelu(ax[r]);
              }
            } break;
            default: break;
          }
        }
        if (has_drop) {
          const uint32_t idx = opx * A.Cout + cos;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] *= lmn_drop_scale(A.drop_seed + soff, idx + r, A.drop_p, P.inv_keep_ep);
        }
        if (A.residual) o += ld4((const TA*)A.residual + opx * A.res_cstride + cos);
        if (A.out && live) { if constexpr (RP) st4((TA*)A.out + (oout_g + (uint32_t)(cos >> 2) * (uint32_t)P.lay_out.qs), o); else st4((TA*)A.out + opx * A.out_cstride + cos, o); }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    LMN_TK_DRAIN();
    LMN_TK(4);
  }
#ifdef LMN_CT_TIMING
  {
    const unsigned long long tk_e = __builtin_amdgcn_s_memtime();
    const int bid = blockIdx.y * gridDim.x + blockIdx.x;
    if (tid == 0 && bid < 4096) {
      for (int k = 0; k < 5; ++k) g_ct_timing[bid * 8 + k] = tk_s[k];
      g_ct_timing[bid * 8 + 5] = tk_e - tk0;
      g_ct_timing[bid * 8 + 6] = tk0;
      g_ct_timing[bid * 8 + 7] = tk_e;
    }
  }
#endif

  // ---- statistics: wave shuffle over the 16 pixels -> LDS -> one global atomic per channel per block
  const bool se = EPI && ep_kind == LMN_EP_SE_BWD;
  const bool chan_stats = EPI && ((st_mode == LMN_STATS_SUM_SQ) || (ep_kind == LMN_EP_BN_BWD1) || se);
  if (chan_stats) {
    __syncthreads();   // (block-uniform) every wave is through with the window of the last tile: it now parks the wave sums
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float a = st0[c][r], bb = st1[c][r];
#pragma unroll
        for (int m = 1; m <= 8; m <<= 1) {
          a += __shfl_xor(a, m, 64);
          bb += __shfl_xor(bb, m, 64);
        }
        if (n == 0) {   // the four waves' sums side by side in the (now free) window, added in wave order below: no LDS atomics
          XS[wv * 2 * NCT * 16 + c * 16 + q * 4 + r] = a;
          XS[wv * 2 * NCT * 16 + NCT * 16 + c * 16 + q * 4 + r] = bb;
        }
      }
    __syncthreads();
    const bool det = P.det_stats != nullptr;
    for (int i = tid; i < (se ? 1 : 2) * NCT * 16; i += 256) {
      const int which = i / (NCT * 16), cc = i - which * NCT * 16;
      const int co = ct0 * 16 + cc;
      const float v = ((XS[i] + XS[2 * NCT * 16 + i]) + XS[4 * NCT * 16 + i]) + XS[6 * NCT * 16 + i];
      if (co < A.Cout) {
        if (se) {
          if (cur_b >= 0) lmn_red_add((det ? P.det_stats + (int64_t)((blockIdx.x + gridDim.x * blockIdx.z) * 4) * A.B * A.Cout : A.stats) + cur_b * A.Cout + co, v, det);
        } else if (det) {
          P.det_stats[(int64_t)(blockIdx.x + gridDim.x * blockIdx.z) * 2 * A.Cout + (int64_t)which * A.Cout + co] = v;   // slot of this block: [2][Cout]
        } else {
          atomicAdd(A.stats + (A.stats_rep > 1 ? (int64_t)(blockIdx.x % A.stats_rep) * 2 * A.Cout : 0) + (int64_t)which * A.Cout + co, v);
        }
      }
    }
  }
}

// M-split variant for wide layers (Cout > 80): the four waves of a block own DIFFERENT cout tiles (wave w: tiles
// ct0 + w + 4i, i < NCW) and ALL pixel groups of the tile, instead of different pixel groups and all cout tiles.
// Each weight fragment is then fetched by exactly one wave of the block (the N-split form pulls every fragment
// through L1 four times; at Cout = 372 that stream, not the MFMAs, set the pace) and feeds 8 x 4 MFMAs; the pixel
// operand comes from LDS, where re-reading it per wave is cheap.
#undef LMN_SLOT
#define LMN_SLOT (wv + 4 * c)
template <int TAPS, int NCW, int EPI, int PM = 0>
__global__ __launch_bounds__(256) void conv_tileM_kernel(const ConvParams P) {
  constexpr int PMB = PM & 3;
  constexpr bool RP = (PM & 4) != 0;      // (row-planar operands: see conv_tile_kernel)
  static_assert(!RP || TAPS == 1, "row-planar operands: 1x1 convs");
  constexpr bool BF = PMB >= 1;
  typedef typename ActT<PMB>::type TA;
  typedef typename Frag<BF>::type wfrag;  // (bf16 operand form: see conv_tile_kernel)
  constexpr int WT = BF ? 128 : 256;
  constexpr int KD = BF ? 8 : 16;
  constexpr int NCT = 4 * NCW;  // cout tiles per block
  const lmn_conv_args_t& A = P.a;
  const uint32_t soff = A.seed_ctr ? *A.seed_ctr : 0u;  // device-side dropout stream offset (graph replays: one bump per step)
  // EPI: 0 plain (LINEAR / AFFINE_ACT, no statistics), 1 generic, 2 LINEAR + SUM_SQ statistics, 3 BN_BWD1, 4 BN_BWD2,
  // 5 SE_BWD.  For EPI >= 2 the epilogue kind is a compile-time constant: each instance carries only its own code
  // (the generic instance keeps every variant resident: 125-160 VGPRs + spills, and measured 20-40 us over its
  // memory time at level 0).
  const int ep_kind = EPI == 2 ? LMN_EP_LINEAR : EPI == 3 ? LMN_EP_BN_BWD1 : EPI == 4 ? LMN_EP_BN_BWD2 : EPI == 5 ? LMN_EP_SE_BWD : A.epilogue;
  const int st_mode = EPI == 2 ? LMN_STATS_SUM_SQ : (EPI == 3 || EPI == 5) ? LMN_STATS_EP : EPI == 4 ? LMN_STATS_NONE : A.stats_mode;
  const bool has_drop = EPI <= 1 && A.drop_p > 0.f;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* XS = smem;                       // [XH*XW][CS]
  float* s_stats = smem + P.XH * P.XW * P.CS;  // [2][NCT*16]
  constexpr int NGM = 8;                       // pixel groups per tile (all owned by every wave)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave index as an SGPR: branches on it stay scalar
  const int q = lane >> 4, n = lane & 15;
  const int ct0 = blockIdx.y * NCT;
  const int pad = A.ksize >> 1;
  constexpr int KS = TAPS == 9 ? 3 : 1;
  const float* wlane = A.wpack + lane * (BF ? 2 : 4);
  int wtile[NCW];  // cout tiles past the end (last chunk) re-read the last real tile: loads stay unconditional, in bounds
#pragma unroll
  for (int c = 0; c < NCW; ++c) wtile[c] = min(ct0 + wv + 4 * c, P.NCTT - 1) * WT;

  for (int i = tid; i < 2 * NCT * 16; i += 256) s_stats[i] = 0.f;
  float* s_par = s_stats + 2 * NCT * 16;  // [9][NCT*16]
  conv_stage_params<NCT>(A, s_par, ct0, tid, blockIdx.x == 0);
  __syncthreads();  // the tiles read s_par from their first instruction on (accumulators start from the bias)
  float st0[NCW][4], st1[NCW][4];
#pragma unroll
  for (int c = 0; c < NCW; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) st0[c][r] = st1[c][r] = 0.f;
  int cur_b = -1;  // image whose SE_BWD sums are in st0

  // tiles of a block: contiguous range [t_begin, t_end) (tstep 1), or every gridDim.x-th tile (P.strided: all blocks
  // get floor or ceil of the average and the surplus lands on the first-dispatched blocks, one per CU)
  const int t_begin = P.strided ? (int)blockIdx.x : (int)(((int64_t)blockIdx.x * P.total_tiles) / gridDim.x);
  const int t_end = P.strided ? P.total_tiles : (int)(((int64_t)(blockIdx.x + 1) * P.total_tiles) / gridDim.x);
  const int tstep = P.strided ? (int)gridDim.x : 1;
#ifdef LMN_CT_TIMING
  unsigned long long tk0 = __builtin_amdgcn_s_memtime(), tk_s[5] = {0, 0, 0, 0, 0}, tk_a = tk0, tk_b;
#endif
  for (int tile = t_begin; tile < t_end; tile += tstep) {
#ifdef LMN_CT_TIMING
    tk_a = __builtin_amdgcn_s_memtime();
#endif
    const int b = tile / (P.tiles_x * P.tiles_y), tt = tile - b * P.tiles_x * P.tiles_y;
    const int oy0 = (tt / P.tiles_x) * P.TH, ox0 = (tt % P.tiles_x) * P.TW;
    // window origin in input coordinates (forward: out*s - pad; data gradient, stride 1: out - pad, taps flipped)
    const int wy0 = A.transposed ? oy0 - pad : oy0 * A.stride - pad;
    const int wx0 = A.transposed ? ox0 - pad : ox0 * A.stride - pad;

    if (EPI && ep_kind == LMN_EP_SE_BWD && b != cur_b) {  // block-uniform: flush the previous image's sums
      if (cur_b >= 0) {
#pragma unroll
        for (int c = 0; c < NCW; ++c)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float t = st0[c][r];
            t += __shfl_xor(t, 1, 64); t += __shfl_xor(t, 2, 64); t += __shfl_xor(t, 4, 64); t += __shfl_xor(t, 8, 64);
            const int ctc = ct0 + wv + 4 * c;
            const int co = ctc * 16 + q * 4 + r;
            if (n == 0 && ctc < P.NCTT && co < A.Cout)   // (the waves own different channels: one slot per block)
              lmn_red_add((P.det_stats ? P.det_stats + (int64_t)(blockIdx.x * 4) * A.B * A.Cout : A.stats) + cur_b * A.Cout + co, t, P.det_stats != nullptr);
            st0[c][r] = 0.f;
          }
      }
      cur_b = b;
    }

    // this wave's two pixel groups: tile pixel -> (row, col), LDS base address, validity
    int pbase[NGM], opix[NGM];
    bool pvalid[NGM];
#pragma unroll
    for (int g = 0; g < NGM; ++g) {
      const int i = g * 16 + n;
      const bool in_t = i < P.TP;
      const int is = in_t ? i : 0;
      const int r = (int)__umulhi((uint32_t)is, P.mTW), c = is - r * P.TW;
      const int oy = oy0 + r, ox = ox0 + c;
      pvalid[g] = in_t && oy < A.Hout && ox < A.Wout;
      opix[g] = pvalid[g] ? (b * A.Hout + oy) * A.Wout + ox : 0;
      const int sr = A.transposed ? r : r * A.stride, sc = A.transposed ? c : c * A.stride;
      pbase[g] = (sr * P.XW + sc) * P.CS + q * (BF ? 2 : 4);
    }
    f32x4 acc[NGM][NCW];  // start from bias (+ bias2)
#pragma unroll
    for (int c = 0; c < NCW; ++c) {
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(s_par + (wv + 4 * c) * 16 + q * 4);
#pragma unroll
      for (int g = 0; g < NGM; ++g) acc[g][c] = b4;
    }

    for (int s = 0; s < A.nsrc; ++s) {
      const lmn_src_t& S = A.src[s];
      const LmnLay LS = P.lay_src[s];   // (used by the row-planar instances only)
      (void)LS;
      for (int kb0 = 0; kb0 < P.nkb[s]; kb0 += P.CKB) {
        const int nkbc = P.nkb[s] - kb0 < P.CKB ? P.nkb[s] - kb0 : P.CKB;
        // step it = (tap, kk).  3x3: nkbc is 1 or 2, tap = it >> ksh, kk = it & ksh; 1x1: kk = it, chunks of up to 8 K16 blocks (the
        // whole K of most wide layers: ONE staging round trip + barrier pair per tile instead of one per 32 channels -- on the small
        // maps a block's life was that chain, phase clocks: staging + barriers 40-50 %, MFMA 36 %)
        const int ksh = nkbc - 1, niter = TAPS * nkbc;
        wfrag wcur[NCW];
        {
          const float* wp = wlane + ((int64_t)(P.kb_off[s] + kb0) * P.NCTT) * WT;
#pragma unroll
          for (int c = 0; c < NCW; ++c) wcur[c] = ldfrag<BF>(wp + wtile[c]);
        }
        __syncthreads();  // previous chunk / tile fully consumed
        LMN_TK(0);
        // ---- stage the window chunk: unconditional float4 loads from clamped addresses, transforms, zero padding
        //      Rounds of SU x 256 items: ALL loads of a round are issued before the first is consumed (the plain loop compiled
        //      to load -> s_waitcnt vmcnt(0) -> write per item, i.e. 3-6 serial HBM round trips per 3x3 chunk).
        const int psh = nkbc <= 1 ? 2 : nkbc == 2 ? 3 : nkbc <= 4 ? 4 : 5, per_px = 1 << psh;   // quad slots per pixel: 4 per K16 block, rounded up to a power of two
        const int nq = nkbc * 4;                                                                // quads per pixel in this chunk
        const int nitems = P.XH * P.XW * per_px;
        constexpr int SU = 4;  // items per thread and round
        for (int i0 = 0; i0 < nitems; i0 += SU * 256) {
          f32x4 sv[SU];
          int sgp[SU];  // global pixel index, -1 = outside the image / past the channels / past the window
#pragma unroll
          for (int u = 0; u < SU; ++u) {
            const int i = i0 + u * 256 + tid;
            const int f = i & (per_px - 1), pix = i >> psh;
            const int r = (int)__umulhi((uint32_t)pix, P.mXW), c = pix - r * P.XW;
            const int iy = wy0 + r, ix = wx0 + c;
            const int ch = kb0 * 16 + f * 4;
            const bool ok = i < nitems && f < nq && ch < S.C && (unsigned)iy < (unsigned)A.Hin && (unsigned)ix < (unsigned)A.Win;
            const int gp = ok ? (b * A.Hin + iy) * A.Win + ix : 0;
            sgp[u] = ok ? gp : -1;
            if constexpr (RP) {
              uint32_t so_ = (uint32_t)(gp * LS.cs) + (uint32_t)((ok ? ch : 0) >> 2) * (uint32_t)LS.qs;
              if (LS.rf) so_ += lmn_div_row((uint32_t)gp, (uint32_t)P.rpw, P.rp_magic) * (uint32_t)LS.rf;   // (wave-uniform: row-planar source)
              sv[u] = ld4((const TA*)S.ptr + so_);
            } else {
              sv[u] = ld4((const TA*)S.ptr + (uint32_t)(gp * S.cstride + (ok ? ch : 0)));
            }
          }
#pragma unroll
          for (int u = 0; u < SU; ++u) {
            const int i = i0 + u * 256 + tid;
            if (i >= nitems) continue;
            const int f = i & (per_px - 1), pix = i >> psh;
            if (f >= nq) continue;
            const bool ok = sgp[u] >= 0;
            const int gp = ok ? sgp[u] : 0, chs = ok ? kb0 * 16 + f * 4 : 0;
            f32x4 v = sv[u];
            if (S.flags & LMN_SRC_GELU) {
#pragma unroll
              for (int k = 0; k < 4; ++k) v[k] = lmn_gelu(v[k]);
            }
            if (S.flags & LMN_SRC_DROP) {
#pragma unroll
              for (int k = 0; k < 4; ++k) v[k] *= lmn_drop_scale(S.drop_seed + soff, (uint32_t)(gp * S.C + chs + k), S.drop_p, P.inv_keep_src[s]);
            }
            if (S.scale) v *= ld4(S.scale + b * S.C + chs);
            if (!ok) v = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (BF) {  // natural channel order, 4 bf16 = one 8-byte fragment slot
              *reinterpret_cast<uint2*>(&XS[pix * P.CS + f * 2]) = pk4_bf16(v);
            } else {  // blocked K layout: channel 4*fl + r of its K16 block sits at position 4*r + fl (MFMA j reads channel 4j + q)
              float* d = &XS[pix * P.CS + (f >> 2) * 16 + (f & 3)];
              d[0] = v[0]; d[4] = v[1]; d[8] = v[2]; d[12] = v[3];
            }
          }
        }
        LMN_TK_DRAIN();
        LMN_TK(1);
        __syncthreads();
        LMN_TK(2);
        // ---- MFMA: taps x K16 blocks of the chunk; weights of step it+1 are fetched while step it runs (the first
        //      fetch was issued before the staging loop)
        // two half-sets of four groups in flight -- the second half of step it is requested before the MFMAs of its first half,
        // the first half of step it+1 before the MFMAs of the second (the loop then never waits on LDS with one wave per SIMD;
        // 1x1: +16 VGPRs put the two-tile instances at two waves per SIMD, and the step still gains 0.6 %)
        if constexpr (BF) {
          if (nkbc >= 2) {   // wave-uniform: pairs of K16 blocks in ONE v_mfma_f32_16x16x32_bf16 (mfma_bf16x2); an odd last block alone
            const int npair = nkbc >> 1;
            const int nstep = TAPS * npair;
            uint2 wa[NCW], wb[NCW];
            {
              const float* wp1 = wlane + ((int64_t)(P.kb_off[s] + kb0 + 1) * P.NCTT) * WT;
#pragma unroll
              for (int c = 0; c < NCW; ++c) { wa[c] = wcur[c]; wb[c] = ldfrag<true>(wp1 + wtile[c]); }
            }
            for (int st = 0; st < nstep; ++st) {
              const int sn = st + 1 < nstep ? st + 1 : st;
              const int tap = st / npair, pp = st - tap * npair;
              const int tapn = sn / npair, ppn = sn - tapn * npair;
              uint2 wna[NCW], wnb[NCW];
              {
                const float* wp = wlane + (((int64_t)tapn * P.NKB + P.kb_off[s] + kb0 + 2 * ppn) * P.NCTT) * WT;
#pragma unroll
                for (int c = 0; c < NCW; ++c) { wna[c] = ldfrag<true>(wp + wtile[c]); wnb[c] = ldfrag<true>(wp + P.NCTT * WT + wtile[c]); }
              }
              const int ty = tap / KS, tx = tap - ty * KS;
              const int fy = A.transposed ? KS - 1 - ty : ty, fx = A.transposed ? KS - 1 - tx : tx;
              const int toff = (fy * P.XW + fx) * P.CS + 2 * pp * KD;
#pragma unroll
              for (int h = 0; h < NGM; h += 4) {
                uint2 x0[4], x1[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { x0[u] = ldfrag<true>(&XS[pbase[h + u] + toff]); x1[u] = ldfrag<true>(&XS[pbase[h + u] + toff + KD]); }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                  if (h + u < P.NG) {
#pragma unroll
                    for (int c = 0; c < NCW; ++c) acc[h + u][c] = mfma_bf16x2(wa[c], wb[c], x0[u], x1[u], acc[h + u][c]);
                  }
                }
              }
#pragma unroll
              for (int c = 0; c < NCW; ++c) { wa[c] = wna[c]; wb[c] = wnb[c]; }
            }
            if (nkbc & 1) {   // the odd last K16 block of the chunk: the 16x16x16 form, per tap
              const int kk = nkbc - 1;
              for (int tap = 0; tap < TAPS; ++tap) {
                const float* wp = wlane + (((int64_t)tap * P.NKB + P.kb_off[s] + kb0 + kk) * P.NCTT) * WT;
                uint2 w1[NCW];
#pragma unroll
                for (int c = 0; c < NCW; ++c) w1[c] = ldfrag<true>(wp + wtile[c]);
                const int ty = tap / KS, tx = tap - ty * KS;
                const int fy = A.transposed ? KS - 1 - ty : ty, fx = A.transposed ? KS - 1 - tx : tx;
                const int toff = (fy * P.XW + fx) * P.CS + kk * KD;
#pragma unroll
                for (int g = 0; g < NGM; ++g) {
                  if (g < P.NG) {
                    const uint2 xg = ldfrag<true>(&XS[pbase[g] + toff]);
#pragma unroll
                    for (int c = 0; c < NCW; ++c) acc[g][c] = mfma_bf16(w1[c], xg, acc[g][c]);
                  }
                }
              }
            }
            LMN_TK(3);
            continue;
          }
        }
        constexpr bool XPF = true;
        wfrag xa[4];
        if constexpr (XPF) {
          const int fy0 = A.transposed ? KS - 1 : 0, fx0 = A.transposed ? KS - 1 : 0;
          const int toff0 = (fy0 * P.XW + fx0) * P.CS;
#pragma unroll
          for (int u = 0; u < 4; ++u) xa[u] = ldfrag<BF>(&XS[pbase[u] + toff0]);
        }
        for (int it = 0; it < niter; ++it) {
          const int itn = it + 1 < niter ? it + 1 : it;
          wfrag wnext[NCW];
          {
            const int tapn = TAPS == 1 ? 0 : itn >> ksh, kkn = TAPS == 1 ? itn : itn & ksh;
            const float* wp = wlane + (((int64_t)tapn * P.NKB + P.kb_off[s] + kb0 + kkn) * P.NCTT) * WT;
#pragma unroll
            for (int c = 0; c < NCW; ++c) wnext[c] = ldfrag<BF>(wp + wtile[c]);
          }
          const int tap = TAPS == 1 ? 0 : it >> ksh, kk = TAPS == 1 ? it : it & ksh;
          const int ty = tap / KS, tx = tap - ty * KS;
          const int fy = A.transposed ? KS - 1 - ty : ty, fx = A.transposed ? KS - 1 - tx : tx;
          const int toff = (fy * P.XW + fx) * P.CS;
          // (no zero-slice skipping here: wide layers have few partial K16 blocks, and the branch cost this loop its schedule)
          // the pixel operands of four groups are requested together, ahead of the (wave-uniform) tests for missing groups: with a
          // read -> wait -> 4 MFMAs chain per group, a block that has a SIMD to itself (the small maps) kept the matrix core
          // waiting on LDS half of the time
          int toffq = 0, kkq = 0;
          if constexpr (XPF) {
            const int tapq = TAPS == 1 ? 0 : itn >> ksh, tyq = tapq / KS, txq = tapq - tyq * KS;
            const int fyq = A.transposed ? KS - 1 - tyq : tyq, fxq = A.transposed ? KS - 1 - txq : txq;
            toffq = (fyq * P.XW + fxq) * P.CS; kkq = TAPS == 1 ? itn : itn & ksh;
          }
#pragma unroll
          for (int h = 0; h < NGM; h += 4) {
            wfrag xg[4];
            if constexpr (XPF) {
              if (h == 0) {
#pragma unroll
                for (int u = 0; u < 4; ++u) { xg[u] = xa[u]; xa[u] = ldfrag<BF>(&XS[pbase[4 + u] + toff + kk * KD]); }   // second half of this step
              } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) { xg[u] = xa[u]; xa[u] = ldfrag<BF>(&XS[pbase[u] + toffq + kkq * KD]); }     // first half of the next step
              }
            } else {
#pragma unroll
              for (int u = 0; u < 4; ++u) xg[u] = ldfrag<BF>(&XS[pbase[h + u] + toff + kk * KD]);   // (groups past NG read pixel 0)
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int g = h + u;
              if (g < P.NG) {
                if constexpr (BF) {
#pragma unroll
                  for (int c = 0; c < NCW; ++c) acc[g][c] = mfma_bf16(wcur[c], xg[u], acc[g][c]);
                } else {
#pragma unroll
                  for (int c = 0; c < NCW; ++c)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[g][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wcur[c][j], xg[u][j], acc[g][c], 0, 0, 0);
                }
              }
            }
          }
#pragma unroll
          for (int c = 0; c < NCW; ++c) wcur[c] = wnext[c];
        }
        LMN_TK(3);
      }
    }

    // ---- epilogue (lane holds channels co..co+3 of its pixel)
#pragma unroll
    for (int g = 0; g < NGM; ++g) {
      const uint32_t opx = (uint32_t)opix[g];
      uint32_t oout_g = 0, oaux_g = 0;   // row-planar instances: pixel part of the out / aux offsets (LmnLay)
      if constexpr (RP) {
        const uint32_t rowp = lmn_div_row(opx, (uint32_t)P.rpw, P.rp_magic);
        oout_g = opx * (uint32_t)P.lay_out.cs + rowp * (uint32_t)P.lay_out.rf;
        oaux_g = opx * (uint32_t)P.lay_aux.cs + rowp * (uint32_t)P.lay_aux.rf;
      }
      (void)oout_g; (void)oaux_g;
#pragma unroll
      for (int c = 0; c < NCW; ++c) {
        const int ctc = ct0 + wv + 4 * c;
        const int co = ctc * 16 + q * 4;
        const bool cok = ctc < P.NCTT && co < A.Cout;
        const bool live = pvalid[g] && cok && g < P.NG;
        const int cos = cok ? co : 0;
        f32x4 v = acc[g][c];
        const float* sp = s_par + LMN_SLOT * 16 + q * 4;
#define LMN_PAR(k) (*reinterpret_cast<const f32x4*>(sp + (k) * NCT * 16))
        f32x4 o = v;
        if (EPI && st_mode == LMN_STATS_SUM_SQ && live) {
          // sums about p4[co] when given (the BatchNorm's running mean): E[x^2] - E[x]^2 then subtracts numbers of the
          // size of the variance, not of the squared mean
          const f32x4 sh = LMN_PAR(6);
#pragma unroll
          for (int r = 0; r < 4; ++r) { const float d = v[r] - sh[r]; st0[c][r] += d; st1[c][r] += d * d; }
        }
        if (ep_kind == LMN_EP_AFFINE_ACT) {
          const f32x4 s0 = LMN_PAR(2), s1 = LMN_PAR(3);
          o = lmn_act4(v * s0 + s1, A.act);
        }
        if (EPI) {
          f32x4 ax = f32x4{0.f, 0.f, 0.f, 0.f};
          if (A.aux) { if constexpr (RP) ax = ld4((const TA*)A.aux + (oaux_g + (uint32_t)(cos >> 2) * (uint32_t)P.lay_aux.qs)); else ax = ld4((const TA*)A.aux + opx * A.aux_cstride + cos); }
          switch (ep_kind) {
            case LMN_EP_DGELU: {
#pragma unroll
              for (int r = 0; r < 4; ++r) o[r] = v[r] * lmn_dgelu(ax[r]);
            } break;
            case LMN_EP_BN_BWD1: {
              const f32x4 mu = LMN_PAR(2), rs = LMN_PAR(3), ga = LMN_PAR(4), be = LMN_PAR(5);
              const f32x4 zh = (v - mu) * rs;
              o = ax * lmn_dact4(ga * zh + be, A.act);
              if (live) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { st0[c][r] += o[r]; st1[c][r] += o[r] * zh[r]; }
              }
            } break;
            case LMN_EP_BN_BWD2: {
              const f32x4 mu = LMN_PAR(2), rs = LMN_PAR(3), c1 = LMN_PAR(4), c2 = LMN_PAR(5), c3 = LMN_PAR(6);
              if (A.p5) {  // aux is the gradient w.r.t. the ACTIVATED output: dh = aux * act'(gamma*zh + beta) formed here
                const f32x4 ga = LMN_PAR(7), be = LMN_PAR(8);
                const f32x4 zh = (v - mu) * rs;
                o = c1 * (ax * lmn_dact4(ga * zh + be, A.act)) - c2 - zh * c3;
              } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = c1[r] * ax[r] - c2[r] - (v[r] - mu[r]) * rs[r] * c3[r];
              }
            } break;
            case LMN_EP_SE_BWD: {
              if (live) {
#pragma unroll
                for (int r = 0; r < 4; ++r) st0[c][r] += v[r] * lmn_gelu(ax[r]);
              }
            } break;
            default: break;
          }
        }
        if (has_drop) {
          const uint32_t idx = opx * A.Cout + cos;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] *= lmn_drop_scale(A.drop_seed + soff, idx + r, A.drop_p, P.inv_keep_ep);
        }
        if (A.residual) o += ld4((const TA*)A.residual + opx * A.res_cstride + cos);
        if (A.out && live) { if constexpr (RP) st4((TA*)A.out + (oout_g + (uint32_t)(cos >> 2) * (uint32_t)P.lay_out.qs), o); else st4((TA*)A.out + opx * A.out_cstride + cos, o); }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    LMN_TK_DRAIN();
    LMN_TK(4);
  }
#ifdef LMN_CT_TIMING
  {
    const unsigned long long tk_e = __builtin_amdgcn_s_memtime();
    const int bid = blockIdx.y * gridDim.x + blockIdx.x;
    if (tid == 0 && bid < 4096) {
      for (int k = 0; k < 5; ++k) g_ct_timing[bid * 8 + k] = tk_s[k];
      g_ct_timing[bid * 8 + 5] = tk_e - tk0;
      g_ct_timing[bid * 8 + 6] = tk0;
      g_ct_timing[bid * 8 + 7] = tk_e;
    }
  }
#endif

  // ---- statistics: wave shuffle over the 16 pixels -> LDS -> one global atomic per channel per block
  const bool se = EPI && ep_kind == LMN_EP_SE_BWD;
  const bool chan_stats = EPI && ((st_mode == LMN_STATS_SUM_SQ) || (ep_kind == LMN_EP_BN_BWD1) || se);
  if (chan_stats) {
#pragma unroll
    for (int c = 0; c < NCW; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float a = st0[c][r], bb = st1[c][r];
#pragma unroll
        for (int m = 1; m <= 8; m <<= 1) {
          a += __shfl_xor(a, m, 64);
          bb += __shfl_xor(bb, m, 64);
        }
        if (n == 0) {  // each (tile, channel) is owned by exactly one wave: plain stores
          s_stats[(wv + 4 * c) * 16 + q * 4 + r] = a;
          s_stats[NCT * 16 + (wv + 4 * c) * 16 + q * 4 + r] = bb;
        }
      }
    __syncthreads();
    for (int i = tid; i < (se ? 1 : 2) * NCT * 16; i += 256) {
      const int which = i / (NCT * 16), cc = i - which * NCT * 16;
      const int co = ct0 * 16 + cc;
      const bool det = P.det_stats != nullptr;
      if (co < A.Cout) {
        if (se) {
          if (cur_b >= 0) lmn_red_add((det ? P.det_stats + (int64_t)(blockIdx.x * 4) * A.B * A.Cout : A.stats) + cur_b * A.Cout + co, s_stats[i], det);
        } else if (det) {
          P.det_stats[(int64_t)blockIdx.x * 2 * A.Cout + (int64_t)which * A.Cout + co] = s_stats[i];
        } else {
          atomicAdd(A.stats + (A.stats_rep > 1 ? (int64_t)(blockIdx.x % A.stats_rep) * 2 * A.Cout : 0) + (int64_t)which * A.Cout + co, s_stats[i]);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------ weight packing
struct PackGeom {
  int taps, Cout, Cin, nsrc, cs[3], transposed, row_off, rows;
  int bf16;  // 1: fragment order of v_mfma_f32_16x16x16_bf16 (lane (q, n) holds channels 4q..4q+3), elements stored as bf16
};

// One element i of the packed stream [tap][K16 block][cout tile][64 lanes][4].
__device__ __forceinline__ float pack_element(const float* __restrict__ w, const PackGeom& g, int64_t i) {
  int nkb[3], kboff[3], cbase[3], NKB = 0, cb = 0;
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    nkb[s] = s < g.nsrc ? (g.cs[s] + 15) / 16 : 0;
    kboff[s] = NKB;
    cbase[s] = cb;
    NKB += nkb[s];
    cb += s < g.nsrc ? g.cs[s] : 0;
  }
  const int nrows = g.transposed ? g.rows : g.Cout;
  const int NCTT = (nrows + 15) / 16;
  const int j = (int)(i & 3);
  const int lane = (int)((i >> 2) & 63);
  uint32_t t = (uint32_t)(i >> 8);   // (a packed weight has < 2^31 elements: 32-bit divisions -- the 64-bit ones were most of this kernel's time)
  const int ct = (int)(t % (uint32_t)NCTT);
  t /= (uint32_t)NCTT;
  const int kb = (int)(t % (uint32_t)NKB);
  const int tap = (int)(t / (uint32_t)NKB);
  int s = 0;
  while (s + 1 < g.nsrc && kb >= kboff[s + 1]) ++s;
  // MFMA j of a K16 block covers channels 4j .. 4j+3 (k slot = lane >> 4): a source whose last block holds fewer than
  // 16 channels leaves whole MFMAs zero, which the kernels skip (12 channels: 3 of 4)
  const int kk = (kb - kboff[s]) * 16 + (g.bf16 ? (lane >> 4) * 4 + j : j * 4 + (lane >> 4));  // reduction index inside the source
  const int row = ct * 16 + (lane & 15);
  float v = 0.f;
  // (a source / row count may be the weight's padded to a multiple of 4 -- the RGB input travels as NHWC4, the 2-class
  //  head is computed on 4 rows: elements past the real weight are zeros)
  if (row < nrows && kk < g.cs[s]) {
    if (!g.transposed) {
      if (row < g.Cout && cbase[s] + kk < g.Cin) v = w[((int64_t)row * g.Cin + cbase[s] + kk) * g.taps + tap];
    } else {  // rows = forward input channels, reduction = forward output channels
      if (kk < g.Cout && g.row_off + row < g.Cin) v = w[((int64_t)kk * g.Cin + g.row_off + row) * g.taps + tap];
    }
  }
  return v;
}

__device__ __forceinline__ void pack_store(float* __restrict__ wp, int64_t i, float v, int bf16) {
  if (bf16) reinterpret_cast<__bf16*>(wp)[i] = (__bf16)v;
  else wp[i] = v;
}

__global__ void conv_pack_kernel(const float* __restrict__ w, float* __restrict__ wp, const PackGeom g, int64_t total) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
    pack_store(wp, i, pack_element(w, g, i), g.bf16);
}

// All weights of a pass in ONE launch: block b finds its job by bisection over the jobs' first-block table, then
// packs 1024 consecutive elements of it.  (192 separate pack launches cost ~1 ms per training step.)
__global__ __launch_bounds__(256) void conv_pack_batch_kernel(const lmn_pack_job_t* __restrict__ jobs, int njobs) {
  int lo = 0, hi = njobs - 1;
  const int64_t b = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].first_block <= b) lo = mid; else hi = mid - 1;
  }
  const lmn_pack_job_t J = jobs[lo];
  PackGeom g;
  g.taps = J.ksize * J.ksize; g.Cout = J.Cout; g.Cin = J.Cin; g.nsrc = J.nsrc;
  g.cs[0] = J.c[0]; g.cs[1] = J.c[1]; g.cs[2] = J.c[2];
  g.transposed = J.transposed; g.row_off = J.row_off; g.rows = J.rows;
  g.bf16 = J.dtype == LMN_BF16;
  const int64_t base = (b - J.first_block) * 1024;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t i = base + r * 256 + threadIdx.x;
    if (i < J.total) pack_store(J.wpack, i, pack_element(J.w, g, i), g.bf16);
  }
}

// ------------------------------------------------------------------------------------ ReparamConv backward, folded
// z-path (lmn_dw_pre_t): BatchNorm backward of the expand conv folded into WEIGHTS.  With dh = dL/d(BN output) and the sums
// S0 = sum dh, S1 = sum dh * z (lmn_dw_bwd_bn), the gradient w.r.t. the conv output is affine in two stored tensors,
//     dz = a * dh + b * z + c,   a = A,  b = -A * T * rstd / N,  c = -A * S0 / N - b * mean,  T = (S1 - mean * S0) * rstd
// (eval-mode BatchNorm: b = c = 0), and z = W_e x + bias_e, so the gradient w.r.t. the block input is ONE conv over three sources
//     dx = [W_e^T diag(a)] dh + [W_e^T diag(b) W_e] x + W_sc^T dy + W_e^T (b * bias_e + c)
// -- no statistics conv, no second conv that writes dz, no separate shortcut gradient on the critical path.  This kernel forms
// a / b / c (every block, in LDS; block 0 stores them for the weight-gradient side and adds the gamma / beta gradients) and
// writes the three operators straight into the packed fragment order of lmn_conv_fwd (sources dh [E], x [Cin], dy [Cout]).
struct FoldParams {
  const float* hstats; const float* mean; const float* rstd; const float* A;   // [2][E], [E] x 3
  const float* we; const float* be; const float* wsc;                          // [E][cinw], [E], [coutw][cinw]  (torch layouts, k = 1)
  float* wpack; float* kbias; float* coef; float* dgamma; float* dbeta;        // packed operators, [rows], [3][E], [E] +=, [E] +=
  float count; int batch_stats;
  int E, rows, cinw, cred, coutw, bf16;   // rows: channels of x / dx (a multiple of 4 >= cinw); cred: channels of dy (>= coutw)
};
__global__ __launch_bounds__(256) void reparam_fold_kernel(const FoldParams P) {
  // one block per packed fragment tile (K16 block kb, row tile ct): 64 lanes x 4 elements.  The Q tiles need E-long dot
  // products of two 16-column panels of W_e: both panels are staged in LDS with coalesced 64-byte rows (read straight from
  // global memory, the loop was a chain of 2 * E dependent L2 round trips per thread: 24 us for a 2 MFLOP problem).
  extern __shared__ float sm[];   // a[E], b[E], c[E], panel R [E][16], panel K [E][16]
  float* sa = sm; float* sb = sm + P.E; float* sc = sm + 2 * P.E;
  float* pr = sm + 3 * P.E; float* pk = pr + P.E * 16;
  const int tid = threadIdx.x;
  for (int e = tid; e < P.E; e += 256) {
    const float S0 = P.hstats[e], S1 = P.hstats[P.E + e];
    const float mu = P.mean[e], rs = P.rstd[e], A = P.A[e];
    const float T = (S1 - mu * S0) * rs;   // sum dh * zhat
    const float b = P.batch_stats ? -A * T * rs / P.count : 0.f;
    const float c = P.batch_stats ? (-A * S0 / P.count - b * mu) : 0.f;
    sa[e] = A; sb[e] = b; sc[e] = c;
    if (blockIdx.x == 0) {
      P.coef[e] = A; P.coef[P.E + e] = b; P.coef[2 * P.E + e] = c;
      if (P.dgamma) P.dgamma[e] += T;
      if (P.dbeta) P.dbeta[e] += S0;
    }
  }
  const int nkb0 = (P.E + 15) / 16, nkb1 = (P.rows + 15) / 16, nkb2 = (P.cred + 15) / 16;
  const int NCTT = (P.rows + 15) / 16;
  const int ntiles = (nkb0 + nkb1 + nkb2) * NCTT;
  if ((int)blockIdx.x < ntiles) {
    const int ct = blockIdx.x % NCTT, kb = blockIdx.x / NCTT;
    const int s = kb < nkb0 ? 0 : (kb < nkb0 + nkb1 ? 1 : 2);
    const int kbl = kb - (s == 0 ? 0 : (s == 1 ? nkb0 : nkb0 + nkb1));
    if (s == 1) {   // panels W_e[:, ct*16 .. +16) and W_e[:, kbl*16 .. +16)
      for (int i = tid; i < P.E * 16; i += 256) {
        const int e = i >> 4, c = i & 15;
        const int r0 = ct * 16 + c, k0 = kbl * 16 + c;
        pr[i] = r0 < P.cinw ? P.we[(int64_t)e * P.cinw + r0] : 0.f;
        pk[i] = k0 < P.cinw ? P.we[(int64_t)e * P.cinw + k0] : 0.f;
      }
    }
    __syncthreads();
    const int lane = tid >> 2, j = tid & 3;
    const int kq = P.bf16 ? (lane >> 4) * 4 + j : j * 4 + (lane >> 4);   // reduction index inside the K16 block
    const int kk = kbl * 16 + kq, rl = lane & 15, row = ct * 16 + rl;
    float v = 0.f;
    if (row < P.cinw) {
      if (s == 0) {
        if (kk < P.E) v = P.we[(int64_t)kk * P.cinw + row] * sa[kk];
      } else if (s == 1) {
        if (kk < P.cinw) {
          float q = 0.f;
          for (int e = 0; e < P.E; ++e) q += pr[e * 16 + rl] * sb[e] * pk[e * 16 + kq];
          v = q;
        }
      } else {
        if (kk < P.coutw) v = P.wsc[(int64_t)kk * P.cinw + row];
      }
    }
    pack_store(P.wpack, (int64_t)blockIdx.x * 256 + tid, v, P.bf16);
  } else {   // the last block: kbias[row] = sum_e W_e[e][row] * (b[e] * bias_e[e] + c[e])  (16 lanes per row over e)
    __syncthreads();
    const int sub = tid & 15;
    for (int row = tid >> 4; row < P.rows; row += 16) {
      float k = 0.f;
      if (row < P.cinw)
        for (int e = sub; e < P.E; e += 16) k += P.we[(int64_t)e * P.cinw + row] * (sb[e] * P.be[e] + sc[e]);
      k += __shfl_xor(k, 1, 64); k += __shfl_xor(k, 2, 64); k += __shfl_xor(k, 4, 64); k += __shfl_xor(k, 8, 64);
      if (sub == 0) P.kbias[row] = k;
    }
  }
}

// Weight / bias gradient of the expand conv on the z-path without materialising dz = a dh + b z + c:
//   dW_e = sum_p dz x^T = diag(a) R + diag(b) (W_e M + b_e m^T) + c m^T,   R = sum dh x^T (the raw weight gradient),
//   M = sum x x^T, m = sum x (one weight-gradient launch over x alone, in the forward), since z = W_e x + b_e;
//   db_e = a S0 + b (W_e m + N b_e) + c N   (zero up to rounding under batch statistics, as in the reference's autograd).
__global__ __launch_bounds__(256) void reparam_wfin_kernel(const float* __restrict__ R, const float* __restrict__ M,
                                                           const float* __restrict__ m, const float* __restrict__ coef,
                                                           const float* __restrict__ hstats, const float* __restrict__ we,
                                                           const float* __restrict__ be, float count, int E, int rows, int cinw,
                                                           float* __restrict__ dW, float* __restrict__ db) {
  // 16 lanes per output element split the cinw-long dot product (one thread per element was a chain of up to 192 dependent
  // L2 round trips: 24 us per launch, 16 launches per step)
  const int i = (blockIdx.x * 256 + threadIdx.x) >> 4, sub = threadIdx.x & 15;
  const bool live = i < E * (cinw + 1);
  const int e = live ? i / (cinw + 1) : 0, c = live ? i - e * (cinw + 1) : 0;
  float acc = 0.f;
  if (live) {
    if (c < cinw) {
      for (int j = sub; j < cinw; j += 16) acc += we[(int64_t)e * cinw + j] * M[(int64_t)j * rows + c];
    } else {
      for (int j = sub; j < cinw; j += 16) acc += we[(int64_t)e * cinw + j] * m[j];
    }
  }
  acc += __shfl_xor(acc, 1, 64); acc += __shfl_xor(acc, 2, 64); acc += __shfl_xor(acc, 4, 64); acc += __shfl_xor(acc, 8, 64);
  if (!live || sub != 0) return;
  const float a = coef[e], b = coef[E + e], k = coef[2 * E + e];
  if (c < cinw) {
    const float zx = be[e] * m[c] + acc;
    dW[(int64_t)e * cinw + c] += a * R[(int64_t)e * rows + c] + b * zx + k * m[c];
  } else if (db) {
    const float zs = count * be[e] + acc;
    db[e] += a * hstats[e] + b * zs + k * count;
  }
}

// ------------------------------------------------------------------------------------ weight gradient
struct WgradParams {
  LmnLay lay_src[3], lay_dy;   // address forms of the operands (common.h: NHWC or row-planar; 1x1 kernels only)
  int32_t rpw;                 // image width of the row-planar operands (0: none)
  uint32_t rp_magic;
  lmn_wgrad_args_t a;
  int ntile_src[3];  // 16-channel tiles per source
  int ntile_off[3];
  int cbase[3];
  int NNTT, NMTT;    // total cin tiles (sum over sources), total cout tiles
  int Cin;
  int nsets_n;       // number of cin tile sets
  int steps_per_img;
  int total_steps;
  float inv_keep_src[3];
  float inv_keep_dy;
  float* partial;    // two-stage reduction: [gridDim.y][gridDim.x][NT*256 + NMT*16] block partials, or NULL (atomics)
  // LDS-staged kernel: output-pixel tile TH x TW, its input window XH x XW, LDS pixel strides, tile counts
  int TH, TW, XH, XW, CSx, CSy, tiles_x, tiles_y, total_tiles;
  uint32_t mXW, mTW;  // magic multipliers: n / d == (n * m) >> 32 for n, d < 2^16
  int dbg;
};

// first element (tap 0) of dW[co][ch of source sidx]: the source's own gradient tensor if the caller gave one,
// else its column range of the concatenated dW
__device__ __forceinline__ float* wgrad_dst(const WgradParams& P, int co, int sidx, int ch, int taps) {
  float* own = P.a.dW_src[sidx];
  return own ? own + ((int64_t)co * P.a.src[sidx].C + ch) * taps
             : P.a.dW + ((int64_t)co * P.Cin + P.cbase[sidx] + ch) * taps;
}

// ------------------------------------------------------------------------------------ weight gradient, LDS-staged
// dW[co][ci][tap] = sum_p dy[p][co] * x[p*s + tap - pad][ci]   as an MFMA GEMM with K = pixels.
// A block walks a contiguous range of output-pixel tiles.  Per tile the input window (with halo) and the dy
// tile are staged ONCE into LDS by coalesced float4 loads (on-load transforms -- GELU, SE scale, dropout mask
// -- applied here, once per element); the four waves then split the tile's 4-pixel K steps and read both MFMA
// operands from LDS as conflict-free ds_read_b32 (lanes along channels; pixel stride = 16 mod 32 banks).
// One global load per element instead of one per (element, tap): the direct form was bound by the texture
// addresser (one 4-segment dword load per MFMA).  Accumulators stay in registers across all tiles of the block.
// BF: operands rounded to bf16 when staged ([tile][pixel][16 bf16] planes, pixel stride CS dwords), a K step is 16 pixels
// = ONE v_mfma_f32_16x16x16_bf16 per (tap, cout tile, cin tile): lane (q, n) gathers pixels 4q..4q+3 of channel n with
// four ds_read_u16 per operand (the same LDS instruction count per pixel as the fp32 form, an eighth of its MFMA time).
template <int TAPS, int NMT, int NNT, int PM = 0>
__global__ __launch_bounds__(256, 2) void wgrad_lds_kernel(const WgradParams P) {
  constexpr bool BF = PM >= 1;
  typedef typename ActT<PM>::type TA;
  const lmn_wgrad_args_t& A = P.a;
  const uint32_t soff = A.seed_ctr ? *A.seed_ctr : 0u;  // device-side dropout stream offset (graph replays: one bump per step)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // one 16-channel PLANE per cin / cout tile: [tile][pixel][16] -- a K step reads 4 pixels x 16 channels = 64
  // consecutive floats of one plane (conflict-free ds_read_b32 at stride 1), and no padding is needed
  const int XP = P.XH * P.XW;
  float* XS = smem;                                   // [NNT][XH*XW][16]
  float* YS = smem + NNT * XP * P.CSx;                // [NMT][TH*TW][16]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave index as an SGPR: branches on it stay scalar
  const int q = lane >> 4, n = lane & 15;
  const int mset = blockIdx.y / P.nsets_n, nset = blockIdx.y - mset * P.nsets_n;
  const int mt0 = mset * NMT, nt0 = nset * NNT;
  const int pad = A.ksize >> 1;
  const int NP = P.TH * P.TW;

  // per cin tile of this block: source and channel base (block-uniform, hoisted out of the tile loop)
  const TA* tptr[NNT];
  const float* tscale[NNT];
  int tC[NNT], tcs[NNT], tflags[NNT], tch0[NNT];
  uint32_t tseed[NNT];
  float tp_[NNT], tik[NNT];
#pragma unroll
  for (int t = 0; t < NNT; ++t) {
    const int nt = nt0 + t;
    int sidx = 0;
    while (sidx + 1 < A.nsrc && nt >= P.ntile_off[sidx + 1]) ++sidx;
    tptr[t] = (const TA*)A.src[sidx].ptr;
    tscale[t] = A.src[sidx].scale;
    tC[t] = nt < P.NNTT ? A.src[sidx].C : 0;
    tcs[t] = A.src[sidx].cstride;
    tflags[t] = A.src[sidx].flags;
    tseed[t] = A.src[sidx].drop_seed + soff;
    tp_[t] = A.src[sidx].drop_p;
    tik[t] = P.inv_keep_src[sidx];
    tch0[t] = (nt - P.ntile_off[sidx]) * 16;
  }

  f32x4 acc[TAPS][NMT][NNT];
  float bsum[NMT];  // bias gradient: running sum of this lane's dy values (pixel q of every K step, channel n)
#pragma unroll
  for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
    for (int m = 0; m < NMT; ++m)
#pragma unroll
      for (int t = 0; t < NNT; ++t) acc[tp][m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int m = 0; m < NMT; ++m) bsum[m] = 0.f;

  const int t_begin = (int)(((int64_t)blockIdx.x * P.total_tiles) / gridDim.x);
  const int t_end = (int)(((int64_t)(blockIdx.x + 1) * P.total_tiles) / gridDim.x);
  for (int tile = t_begin; tile < t_end; ++tile) {
    const int b = tile / (P.tiles_x * P.tiles_y), tt = tile - b * P.tiles_x * P.tiles_y;
    const int oy0 = (tt / P.tiles_x) * P.TH, ox0 = (tt % P.tiles_x) * P.TW;
    const int iy0 = oy0 * A.stride - pad, ix0 = ox0 * A.stride - pad;
    __syncthreads();  // previous tile's reads are done
    // ---- stage the input window and the dy tile: float4 items (pixel, quad).  ALL global loads of a round (up to
    //      UX x 256 window items and UY x 256 dy items -- normally the whole tile) are issued before the first one is
    //      consumed: one exposed memory latency per tile instead of one per 256 items (the K loop of a tile is
    //      shorter than two such latencies).
    constexpr int UX = (TAPS == 9 && NMT * NNT == 4) ? 3 : 4, UY = 2;
    const int NXI = P.XH * P.XW * 4, NYI = NP * 4;
    for (int rd = 0; rd * (UX * 256) < NXI || rd * (UY * 256) < NYI; ++rd) {
      f32x4 vx[UX][NNT], vy[UY][NMT];
      int gpx[UX], gpy[UY];  // global pixel index, -1 = outside the image
#pragma unroll
      for (int u = 0; u < UX; ++u) {
        const int i = rd * (UX * 256) + u * 256 + tid;
        const int j = i & 3, pix = i >> 2;
        bool inb;
        int gp;
        if constexpr (TAPS == 1) {  // 1x1: the image is one flat row (host), the window is the tile itself
          inb = i < NXI && ox0 + pix < A.Wout;
          gp = inb ? b * A.Wout + ox0 + pix : 0;
        } else {
          const int r = (int)__umulhi((uint32_t)pix, P.mXW), c = pix - r * P.XW;
          const int iy = iy0 + r, ix = ix0 + c;
          inb = i < NXI && (unsigned)iy < (unsigned)A.Hin && (unsigned)ix < (unsigned)A.Win;
          gp = inb ? (b * A.Hin + iy) * A.Win + ix : 0;
        }
        gpx[u] = inb ? gp : -1;
#pragma unroll
        for (int t = 0; t < NNT; ++t) {
          const int ch = tch0[t] + j * 4;
          vx[u][t] = ld4(tptr[t] + (uint32_t)(gp * tcs[t] + (ch < tC[t] ? ch : 0)));
        }
      }
#pragma unroll
      for (int u = 0; u < UY; ++u) {
        const int i = rd * (UY * 256) + u * 256 + tid;
        const int j = i & 3, pix = i >> 2;
        bool inb;
        int gp;
        if constexpr (TAPS == 1) {
          inb = i < NYI && ox0 + pix < A.Wout;
          gp = inb ? b * A.Wout + ox0 + pix : 0;
        } else {
          const int r = (int)__umulhi((uint32_t)pix, P.mTW), c = pix - r * P.TW;
          const int oy = oy0 + r, ox = ox0 + c;
          inb = i < NYI && oy < A.Hout && ox < A.Wout;
          gp = inb ? (b * A.Hout + oy) * A.Wout + ox : 0;
        }
        gpy[u] = inb ? gp : -1;
#pragma unroll
        for (int m = 0; m < NMT; ++m) {
          const int co = (mt0 + m) * 16 + j * 4;
          const bool cok = (mt0 + m) < P.NMTT && co < A.Cout;
          vy[u][m] = ld4((const TA*)A.dy + (uint32_t)(gp * A.dy_cstride + (cok ? co : 0)));
        }
      }
#pragma unroll
      for (int u = 0; u < UX; ++u) {
        const int i = rd * (UX * 256) + u * 256 + tid;
        if (i < NXI) {
          const int j = i & 3, pix = i >> 2;
          const bool inb = gpx[u] >= 0;
          const int gp = inb ? gpx[u] : 0;
#pragma unroll
          for (int t = 0; t < NNT; ++t) {
            const int ch = tch0[t] + j * 4;
            const bool ok = inb && ch < tC[t];
            const int chs = ch < tC[t] ? ch : 0;
            f32x4 w = vx[u][t];
            if (tflags[t] & LMN_SRC_GELU) {
#pragma unroll
              for (int k = 0; k < 4; ++k) w[k] = lmn_gelu(w[k]);
            }
            if (tflags[t] & LMN_SRC_DROP) {
#pragma unroll
              for (int k = 0; k < 4; ++k) w[k] *= lmn_drop_scale(tseed[t], (uint32_t)(gp * tC[t] + chs + k), tp_[t], tik[t]);
            }
            if (tscale[t]) w *= ld4(tscale[t] + (inb ? b : 0) * tC[t] + chs);
            if (!ok) w = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (BF) *reinterpret_cast<uint2*>(&XS[(t * XP + pix) * P.CSx + j * 2]) = pk4_bf16(w);
            else *reinterpret_cast<f32x4*>(&XS[(t * XP + pix) * P.CSx + j * 4]) = w;
          }
        }
      }
#pragma unroll
      for (int u = 0; u < UY; ++u) {
        const int i = rd * (UY * 256) + u * 256 + tid;
        if (i < NYI) {
          const int j = i & 3, pix = i >> 2;
          const bool inb = gpy[u] >= 0;
          const int gp = inb ? gpy[u] : 0;
#pragma unroll
          for (int m = 0; m < NMT; ++m) {
            const int co = (mt0 + m) * 16 + j * 4;
            const bool cok = (mt0 + m) < P.NMTT && co < A.Cout;
            const int cos = cok ? co : 0;
            f32x4 w = vy[u][m];
            if (A.dy_flags & LMN_SRC_DROP) {
#pragma unroll
              for (int k = 0; k < 4; ++k) w[k] *= lmn_drop_scale(A.dy_seed + soff, (uint32_t)(gp * A.Cout + cos + k), A.dy_p, P.inv_keep_dy);
            }
            if (!(inb && cok)) w = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (BF) *reinterpret_cast<uint2*>(&YS[(m * NP + pix) * P.CSy + j * 2]) = pk4_bf16(w);
            else *reinterpret_cast<f32x4*>(&YS[(m * NP + pix) * P.CSy + j * 4]) = w;
          }
        }
      }
    }
    __syncthreads();
    if constexpr (BF) {
      // ---- bf16: K steps of 16 consecutive tile pixels; lane (q, n) owns pixels 4q..4q+3 of the step, channel n
      const uint16_t* XH16 = reinterpret_cast<const uint16_t*>(XS);
      const uint16_t* YH16 = reinterpret_cast<const uint16_t*>(YS);
      for (int ks = wv; ks * 16 < NP; ks += 4) {
        int xb[4];
        bool pin[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int pix = ks * 16 + q * 4 + j;
          pin[j] = pix < NP;
          const int pixs = pin[j] ? pix : 0;
          if constexpr (TAPS == 1) {
            xb[j] = pixs * P.CSx * 2 + n;
          } else {
            const int pr = (int)__umulhi((uint32_t)pixs, P.mTW), pc = pixs - pr * P.TW;
            xb[j] = ((pr * A.stride) * P.XW + pc * A.stride) * P.CSx * 2 + n;
          }
        }
        uint2 av[NMT], bvv[TAPS][NNT];
#pragma unroll
        for (int m = 0; m < NMT; ++m) {
          uint32_t h[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int pixs = pin[j] ? ks * 16 + q * 4 + j : 0;
            h[j] = pin[j] ? (uint32_t)YH16[(m * NP + pixs) * P.CSy * 2 + n] : 0u;
            bsum[m] += __builtin_bit_cast(float, h[j] << 16);
          }
          av[m] = uint2{h[0] | (h[1] << 16), h[2] | (h[3] << 16)};
        }
#pragma unroll
        for (int tp = 0; tp < TAPS; ++tp) {
          const int ty = (TAPS == 9) ? tp / 3 : 0, tx = (TAPS == 9) ? tp % 3 : 0;
#pragma unroll
          for (int t = 0; t < NNT; ++t) {
            const int toff = (t * XP + ty * P.XW + tx) * P.CSx * 2;
            const uint32_t h0 = XH16[xb[0] + toff], h1 = XH16[xb[1] + toff], h2 = XH16[xb[2] + toff], h3 = XH16[xb[3] + toff];
            bvv[tp][t] = uint2{h0 | (h1 << 16), h2 | (h3 << 16)};
          }
        }
#pragma unroll
        for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
          for (int t = 0; t < NNT; ++t)
#pragma unroll
            for (int m = 0; m < NMT; ++m) acc[tp][m][t] = mfma_bf16(av[m], bvv[tp][t], acc[tp][m][t]);
      }
    } else
    // ---- MFMA over this wave's K steps (4 consecutive tile pixels each); all LDS reads of a step are issued
    //      before its MFMAs so their latency overlaps
    for (int ks = wv; ks * 4 < NP; ks += 4) {
      const int pix = ks * 4 + q;
      const bool pin = pix < NP;
      const int pixs = pin ? pix : 0;
      float av[NMT], bvv[TAPS][NNT];
#pragma unroll
      for (int m = 0; m < NMT; ++m) av[m] = YS[(m * NP + pixs) * P.CSy + n];
      int xb;
      if constexpr (TAPS == 1) {
        xb = pixs * P.CSx + n;
      } else {
        const int pr = (int)__umulhi((uint32_t)pixs, P.mTW), pc = pixs - pr * P.TW;
        xb = ((pr * A.stride) * P.XW + pc * A.stride) * P.CSx + n;
      }
#pragma unroll
      for (int tp = 0; tp < TAPS; ++tp) {
        const int ty = (TAPS == 9) ? tp / 3 : 0, tx = (TAPS == 9) ? tp % 3 : 0;
#pragma unroll
        for (int t = 0; t < NNT; ++t) bvv[tp][t] = XS[xb + (t * XP + ty * P.XW + tx) * P.CSx];
      }
      if (!pin) {
#pragma unroll
        for (int m = 0; m < NMT; ++m) av[m] = 0.f;
      }
#pragma unroll
      for (int m = 0; m < NMT; ++m) bsum[m] += av[m];
#pragma unroll
      for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
        for (int t = 0; t < NNT; ++t)
#pragma unroll
          for (int m = 0; m < NMT; ++m) acc[tp][m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], bvv[tp][t], acc[tp][m][t], 0, 0, 0);
    }
  }

  // ---- block-level reduction in LDS (4 waves -> 1): plain stores / read-add-stores in four wave rounds.
  //      (LDS float atomics cost ~3 cycles per LANE on gfx950 -- measured 180 us for this tail with ds_add_f32.)
  constexpr int NT = TAPS * NMT * NNT;
  __syncthreads();
  float* s_acc = smem;  // reuse the staging area (>= NT*256 + NMT*16 floats, checked on the host)
  for (int w = 0; w < 4; ++w) {
    if (wv == w) {
#pragma unroll
      for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
        for (int m = 0; m < NMT; ++m)
#pragma unroll
          for (int t = 0; t < NNT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float* d = &s_acc[(((tp * NMT + m) * NNT + t) * 4 + r) * 64 + lane];
              *d = (w == 0) ? acc[tp][m][t][r] : *d + acc[tp][m][t][r];
            }
#pragma unroll
      for (int m = 0; m < NMT; ++m) {  // bias: sum the 4 pixel lanes (q) of channel n
        float v = bsum[m];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (q == 0) {
          float* d = &s_acc[NT * 256 + m * 16 + n];
          *d = (w == 0) ? v : *d + v;
        }
      }
    }
    __syncthreads();
  }
  if (P.partial) {
    float* dst = P.partial + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * (NT * 256 + NMT * 16);
    for (int i = tid; i < NT * 256 + NMT * 16; i += 256) dst[i] = s_acc[i];
    return;
  }
  for (int i = tid; i < NT * 256; i += 256) {
    const int ln = i & 63, r = (i >> 6) & 3, tl = i >> 8;
    const int t = tl % NNT, m = (tl / NNT) % NMT, tp = tl / (NNT * NMT);
    const int qq = ln >> 4, nn = ln & 15;
    const int co = (mt0 + m) * 16 + qq * 4 + r;
    const int nt = nt0 + t;
    if ((mt0 + m) >= P.NMTT || co >= A.Cout || nt >= P.NNTT) continue;
    int sidx = 0;
    while (sidx + 1 < A.nsrc && nt >= P.ntile_off[sidx + 1]) ++sidx;
    const int ch = (nt - P.ntile_off[sidx]) * 16 + nn;
    if (ch >= A.src[sidx].C) continue;
    atomicAdd(wgrad_dst(P, co, sidx, ch, TAPS) + tp, s_acc[i]);
  }
  if (A.db && nset == 0) {
    for (int i = tid; i < NMT * 16; i += 256) {
      const int co = mt0 * 16 + i;
      if (co < A.Cout && (mt0 + i / 16) < P.NMTT) {
        atomicAdd(A.db + co, s_acc[NT * 256 + i]);
        if (A.db2) atomicAdd(A.db2 + co, s_acc[NT * 256 + i]);
      }
    }
  }
}

// ------------------------------------------------------------------------------------ weight gradient, 3x3 stride 1, maps >= 32 wide
// The LDS-staged form above specialised for the layers that carry most of the weight-gradient time (tile 32 pixels wide,
// window 34 wide, one-tile or 2 x 2-tile blocks).  Against the general kernel:
//  * 2 x 2-tile blocks give each wave ONE output tile (wave w: cout tile w / 2, cin tile w % 2) with all 9 taps over ALL K steps
//    of the tile, instead of all four tiles over every fourth step: 36 accumulator VGPRs instead of 144 (room for a third
//    resident block per CU and the prefetch registers below) and no cross-wave reduction at the end;
//  * the global loads of tile i+1 are issued BEFORE the K loop of tile i and committed to LDS after it, so a tile's memory
//    latency hides behind this block's own MFMAs, not only behind those of the CU's other blocks;
//  * index math is hoisted out of the per-tile / per-step paths (it ran at ~35 VALU per K step and ~25 per staged item, which
//    on 3-wave SIMDs competes with the MFMA issue slots -- measured: MFMA busy 47 % with the K phase at 70 % of a block's
//    life).  A K step lies in ONE tile row (32 | tile width): row / column are scalar shifts of the step index and the nine tap
//    reads are immediate offsets from one lane address.  Staged items are addressed by per-thread offsets computed once per
//    kernel plus a scalar tile base; raw buffer loads return 0 for the halo outside the image (offset forced out of range),
//    so the commit is a plain register -> LDS copy when the source has no on-load transform (bf16 storage: the loaded words
//    go to LDS untouched).
// LDS planes are padded to whole 64-pixel item rounds ([tile][UX * 64 px][CS]): every plane / round offset is an immediate.
template <int ESZ> struct RawOf { typedef f32x4 type; };
template <> struct RawOf<2> { typedef u32x2 type; };
__device__ __forceinline__ f32x4 raw_f32(f32x4 v) { return v; }
__device__ __forceinline__ f32x4 raw_f32(u32x2 v) { return f32x4{lmn_bf16_lo(v.x), lmn_bf16_hi(v.x), lmn_bf16_lo(v.y), lmn_bf16_hi(v.y)}; }
template <typename RawT> __device__ __forceinline__ RawT raw_load(BufRsrc r, uint32_t off) {
  if constexpr (sizeof(RawT) == 16) return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0));
  else return __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, 0);
}
#ifdef LMN_WG_TIMING
// phase clocks (debug builds only): per block {tile-top barrier, commit, second barrier + issue, rest, start, end}
__device__ unsigned long long g_wg_timing[4096 * 6];
#define LMN_TCLK() __builtin_amdgcn_s_memtime()
#endif

template <int NMT, int NNT, int PM>
__global__ __launch_bounds__(256, 3) void wgrad3_kernel(const WgradParams P) {
  static_assert(NMT * NNT == 4 || NMT * NNT == 1, "one-tile or 2 x 2-tile blocks");
  constexpr bool TS = NMT * NNT == 4;  // one output tile per wave
  constexpr bool BF = PM >= 1;
  typedef typename ActT<PM>::type TA;
  constexpr int ESZ = sizeof(TA);
  typedef typename RawOf<ESZ>::type RawT;
  constexpr int XWC = 34, CS = BF ? 12 : 16;  // window width; dwords per LDS pixel (bf16: 8 + 4 pad)
  constexpr int QW = BF ? 2 : 4;              // dwords of one staged (pixel, channel quad) item in LDS
  constexpr int UX = TS ? 4 : 6, UY = TS ? 2 : 4;  // staged items per thread: window, dy tile
  constexpr int XPA = UX * 64, NPA = UY * 64;      // padded plane sizes in pixels
  const lmn_wgrad_args_t& A = P.a;
  const uint32_t soff = A.seed_ctr ? *A.seed_ctr : 0u;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* XS = smem;                     // [NNT][XPA][CS]
  float* YS = smem + NNT * XPA * CS;    // [NMT][NPA][CS]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, n = lane & 15;
  const int m_w = TS ? wv / NNT : 0, t_w = TS ? wv % NNT : 0;
  const int mset = blockIdx.y / P.nsets_n, nset = blockIdx.y - mset * P.nsets_n;
  const int mt0 = mset * NMT, nt0 = nset * NNT;
  const int NP = P.TH * 32;
  const int j = tid & 3, pl = tid >> 2;  // this thread's channel quad and pixel-in-round of every staged item

  // per cin tile of this block: source, channel base, buffer descriptor
  const float* tscale[NNT];
  int tC[NNT], tcs[NNT], tflags[NNT], tch0[NNT];
  uint32_t tseed[NNT];
  float tp_[NNT], tik[NNT];
  BufRsrc rx[NNT];
  bool cx[NNT];
#pragma unroll
  for (int t = 0; t < NNT; ++t) {
    const int nt = nt0 + t;
    int sidx = 0;
    while (sidx + 1 < A.nsrc && nt >= P.ntile_off[sidx + 1]) ++sidx;
    tscale[t] = A.src[sidx].scale;
    tC[t] = nt < P.NNTT ? A.src[sidx].C : 0;
    tcs[t] = A.src[sidx].cstride;
    tflags[t] = A.src[sidx].flags;
    tseed[t] = A.src[sidx].drop_seed + soff;
    tp_[t] = A.src[sidx].drop_p;
    tik[t] = P.inv_keep_src[sidx];
    tch0[t] = (nt - P.ntile_off[sidx]) * 16;
    rx[t] = make_rsrc(A.src[sidx].ptr, (unsigned)((int64_t)A.B * A.Hin * A.Win * tcs[t] * ESZ));
    cx[t] = tch0[t] + j * 4 < tC[t];
  }
  const BufRsrc ry = make_rsrc(A.dy, (unsigned)((int64_t)A.B * A.Hout * A.Wout * A.dy_cstride * ESZ));

  // tile-invariant item descriptors: window items (row, column) and byte offsets from the window's first pixel
  uint32_t xrc[UX], xo[UX][NNT];
#pragma unroll
  for (int u = 0; u < UX; ++u) {
    const int pix = u * 64 + pl, r = pix / XWC, c = pix - r * XWC;
    xrc[u] = r < P.XH ? (uint32_t)(r << 16 | c) : 0x7fff0000u;  // rows past the window never pass the bounds test
#pragma unroll
    for (int t = 0; t < NNT; ++t) xo[u][t] = (uint32_t)(((r * A.Win + c) * tcs[t] + tch0[t] + j * 4) * ESZ);
  }
  // dy items: pixel u * 64 + pl of the 32-wide tile = row 2u + (pl >> 5), column pl & 31
  const int yc = pl & 31, yr0 = pl >> 5;
  uint32_t yo0[NMT];
  bool cy[NMT];
#pragma unroll
  for (int m = 0; m < NMT; ++m) {
    const int co = (mt0 + m) * 16 + j * 4;
    cy[m] = (mt0 + m) < P.NMTT && co < A.Cout;
    yo0[m] = (uint32_t)(((yr0 * A.Wout + yc) * A.dy_cstride + (cy[m] ? co : 0)) * ESZ);
  }
  const uint32_t ystep = (uint32_t)(2 * A.Wout * A.dy_cstride * ESZ);  // two tile rows

  f32x4 acc[9];
  float bsum = 0.f;  // bias gradient: this lane's dy values (pixel q of every K step, channel n)
#pragma unroll
  for (int tp = 0; tp < 9; ++tp) acc[tp] = f32x4{0.f, 0.f, 0.f, 0.f};

  RawT vx[UX][NNT], vy[UY][NMT];
  const int tpi = P.tiles_x * P.tiles_y;
  auto issue = [&](int tile) __attribute__((always_inline)) {
    const int b = tile / tpi, tt = tile - b * tpi;
    const int ty_ = tt / P.tiles_x;
    const int oy0 = ty_ * P.TH, ox0 = (tt - ty_ * P.tiles_x) * 32;
    const int iy0 = oy0 - 1, ix0 = ox0 - 1;
    const int wpix = (b * A.Hin + iy0) * A.Win + ix0;  // window's first pixel (may lie before the tensor: wraps back below)
#pragma unroll
    for (int u = 0; u < UX; ++u) {
      const int r = (int)(xrc[u] >> 16), c = (int)(xrc[u] & 0xffffu);
      const bool inb = (unsigned)(iy0 + r) < (unsigned)A.Hin && (unsigned)(ix0 + c) < (unsigned)A.Win;
#pragma unroll
      for (int t = 0; t < NNT; ++t) {
        const uint32_t off = (uint32_t)(wpix * tcs[t] * ESZ) + xo[u][t];
        vx[u][t] = raw_load<RawT>(rx[t], (inb && cx[t]) ? off : 0xffffffffu);
      }
    }
    const uint32_t ybase = (uint32_t)(((b * A.Hout + oy0) * A.Wout + ox0) * A.dy_cstride * ESZ);
    const bool cok = ox0 + yc < A.Wout;
    const int rlim = (A.Hout - oy0 < P.TH ? A.Hout - oy0 : P.TH) - yr0;  // rows of this tile inside the image
#pragma unroll
    for (int u = 0; u < UY; ++u) {
      const bool inb = cok && 2 * u < rlim;
#pragma unroll
      for (int m = 0; m < NMT; ++m)
        vy[u][m] = raw_load<RawT>(ry, (inb && cy[m]) ? ybase + (uint32_t)u * ystep + yo0[m] : 0xffffffffu);
    }
  };
  float* xl = XS + pl * CS + j * QW;
  float* yl = YS + pl * CS + j * QW;
  auto commit = [&](int tile) __attribute__((always_inline)) {
    const int b = tile / tpi, tt = tile - b * tpi;
    const int ty_ = tt / P.tiles_x;
    const int oy0 = ty_ * P.TH, ox0 = (tt - ty_ * P.tiles_x) * 32;
#pragma unroll
    for (int t = 0; t < NNT; ++t) {
      const bool tf = tflags[t] != 0 || tscale[t] != nullptr;  // block-uniform
#pragma unroll
      for (int u = 0; u < UX; ++u) {
        float* dst = xl + (t * XPA + u * 64) * CS;
        if (!tf) {
          if constexpr (ESZ == 2) *reinterpret_cast<u32x2*>(dst) = vx[u][t];
          else if constexpr (BF) *reinterpret_cast<uint2*>(dst) = pk4_bf16(raw_f32(vx[u][t]));
          else *reinterpret_cast<f32x4*>(dst) = raw_f32(vx[u][t]);
          continue;
        }
        f32x4 w = raw_f32(vx[u][t]);  // lanes outside the image / past the channels hold 0 and stay 0 under every transform
        const int chs = cx[t] ? tch0[t] + j * 4 : 0;
        if (tflags[t] & LMN_SRC_GELU) {
#pragma unroll
          for (int k = 0; k < 4; ++k) w[k] = lmn_gelu(w[k]);
        }
        if (tflags[t] & LMN_SRC_DROP) {
          const int r = (int)(xrc[u] >> 16), c = (int)(xrc[u] & 0xffffu);
          const int gp = (b * A.Hin + oy0 - 1 + r) * A.Win + ox0 - 1 + c;
#pragma unroll
          for (int k = 0; k < 4; ++k) w[k] *= lmn_drop_scale(tseed[t], (uint32_t)(gp * tC[t] + chs + k), tp_[t], tik[t]);
        }
        if (tscale[t]) w *= ld4(tscale[t] + b * tC[t] + chs);
        if constexpr (BF) *reinterpret_cast<uint2*>(dst) = pk4_bf16(w);
        else *reinterpret_cast<f32x4*>(dst) = w;
      }
    }
#pragma unroll
    for (int m = 0; m < NMT; ++m) {
#pragma unroll
      for (int u = 0; u < UY; ++u) {
        float* dst = yl + (m * NPA + u * 64) * CS;
        if (!(A.dy_flags & LMN_SRC_DROP)) {
          if constexpr (ESZ == 2) *reinterpret_cast<u32x2*>(dst) = vy[u][m];
          else if constexpr (BF) *reinterpret_cast<uint2*>(dst) = pk4_bf16(raw_f32(vy[u][m]));
          else *reinterpret_cast<f32x4*>(dst) = raw_f32(vy[u][m]);
          continue;
        }
        f32x4 w = raw_f32(vy[u][m]);
        const int cos = cy[m] ? (mt0 + m) * 16 + j * 4 : 0;
        const int gp = (b * A.Hout + oy0 + 2 * u + yr0) * A.Wout + ox0 + yc;
#pragma unroll
        for (int k = 0; k < 4; ++k) w[k] *= lmn_drop_scale(A.dy_seed + soff, (uint32_t)(gp * A.Cout + cos + k), A.dy_p, P.inv_keep_dy);
        if constexpr (BF) *reinterpret_cast<uint2*>(dst) = pk4_bf16(w);
        else *reinterpret_cast<f32x4*>(dst) = w;
      }
    }
  };

  const int t_begin = (int)(((int64_t)blockIdx.x * P.total_tiles) / gridDim.x);
  const int t_end = (int)(((int64_t)(blockIdx.x + 1) * P.total_tiles) / gridDim.x);
#ifdef LMN_WG_TIMING
  unsigned long long tk0 = LMN_TCLK(), tk_s1 = 0, tk_s2 = 0, tk_s3 = 0, tk_a, tk_b;
#endif
  if (t_begin < t_end) issue(t_begin);
  const int K0 = TS ? 0 : wv, KSTEP = TS ? 1 : 4;
  for (int tile = t_begin; tile < t_end; ++tile) {
#ifdef LMN_WG_TIMING
    tk_a = LMN_TCLK();
#endif
    __syncthreads();  // previous tile's reads are done
#ifdef LMN_WG_TIMING
    tk_b = LMN_TCLK(); tk_s1 += tk_b - tk_a; tk_a = tk_b;
#endif
    commit(tile);
#ifdef LMN_WG_TIMING
    __builtin_amdgcn_s_waitcnt(0);
    tk_b = LMN_TCLK(); tk_s2 += tk_b - tk_a; tk_a = tk_b;
#endif
    __syncthreads();
    if (tile + 1 < t_end) issue(tile + 1);
#ifdef LMN_WG_TIMING
    tk_b = LMN_TCLK(); tk_s3 += tk_b - tk_a;
#endif
    if constexpr (BF) {
      // K steps of 16 consecutive tile pixels; lane (q, n) owns pixels 4q..4q+3 of the step, channel n
      const uint16_t* xw = reinterpret_cast<const uint16_t*>(XS) + (t_w * XPA + q * 4) * (CS * 2) + n;
      const uint16_t* yw = reinterpret_cast<const uint16_t*>(YS) + (m_w * NPA + q * 4) * (CS * 2) + n;
      auto kfrag = [&](int ks, uint2& av, uint2 (&bv)[9]) __attribute__((always_inline)) {   // operands of the 16-pixel K step ks
        const int pix0 = ks * 16, pr = pix0 >> 5, pc0 = pix0 & 31;
        const uint16_t* xp = xw + (pr * XWC + pc0) * (CS * 2);
        const uint16_t* yp = yw + pix0 * (CS * 2);
        uint32_t h[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          h[jj] = yp[jj * CS * 2];
          bsum += __builtin_bit_cast(float, h[jj] << 16);
        }
        av = uint2{h[0] | (h[1] << 16), h[2] | (h[3] << 16)};
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) {
          const uint16_t* xt = xp + ((tp / 3) * XWC + tp % 3) * (CS * 2);
          const uint32_t h0 = xt[0], h1 = xt[CS * 2], h2 = xt[2 * CS * 2], h3 = xt[3 * CS * 2];
          bv[tp] = uint2{h0 | (h1 << 16), h2 | (h3 << 16)};
        }
      };
      // two 16-pixel K steps per v_mfma_f32_16x16x32_bf16 (mfma_bf16x2: K = 32 pixels); an odd last step in the 16x16x16 form
      int ks = K0;
      for (; (ks + KSTEP) * 16 < NP; ks += 2 * KSTEP) {
        uint2 av0, av1, bv0[9], bv1[9];
        kfrag(ks, av0, bv0);
        kfrag(ks + KSTEP, av1, bv1);
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) acc[tp] = mfma_bf16x2(av0, av1, bv0[tp], bv1[tp], acc[tp]);
      }
      if (ks * 16 < NP) {
        uint2 av, bv[9];
        kfrag(ks, av, bv);
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) acc[tp] = mfma_bf16(av, bv[tp], acc[tp]);
      }
    } else {
      const float* xw = XS + (t_w * XPA + q) * CS + n;
      const float* yw = YS + (m_w * NPA + q) * CS + n;
      for (int ks = K0; ks * 4 < NP; ks += KSTEP) {
        const int pix0 = ks * 4, pr = pix0 >> 5, pc0 = pix0 & 31;
        const float* xp = xw + (pr * XWC + pc0) * CS;
        const float av = yw[pix0 * CS];
        float bv[9];
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) bv[tp] = xp[((tp / 3) * XWC + tp % 3) * CS];
        bsum += av;
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) acc[tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[tp], acc[tp], 0, 0, 0);
      }
    }
  }
#ifdef LMN_WG_TIMING
  {
    const unsigned long long tk_e = LMN_TCLK();
    const int bid = blockIdx.y * gridDim.x + blockIdx.x;
    if (tid == 0 && bid < 4096) {
      g_wg_timing[bid * 6 + 0] = tk_s1;
      g_wg_timing[bid * 6 + 1] = tk_s2;
      g_wg_timing[bid * 6 + 2] = tk_s3;
      g_wg_timing[bid * 6 + 3] = tk_e - tk0 - tk_s1 - tk_s2 - tk_s3;
      g_wg_timing[bid * 6 + 4] = tk0;
      g_wg_timing[bid * 6 + 5] = tk_e;
    }
  }
#endif

  constexpr int NT = 9 * NMT * NNT;
  float* pdst = P.partial ? P.partial + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * (NT * 256 + NMT * 16) : nullptr;
  if constexpr (TS) {
    // every wave holds the finished sums of ITS tile: straight to the block partial (the layout of the LDS-reduced form:
    // [tap][m][t][r][lane], then the bias sums) or to dW
    const int nt = nt0 + t_w;
    int sidx = 0;
    while (sidx + 1 < A.nsrc && nt >= P.ntile_off[sidx + 1]) ++sidx;
    const int ch = (nt - P.ntile_off[sidx]) * 16 + n;
#pragma unroll
    for (int tp = 0; tp < 9; ++tp)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (pdst) {
          pdst[(((tp * NMT + m_w) * NNT + t_w) * 4 + r) * 64 + lane] = acc[tp][r];
        } else {
          const int co = (mt0 + m_w) * 16 + q * 4 + r;
          if ((mt0 + m_w) < P.NMTT && co < A.Cout && nt < P.NNTT && ch < A.src[sidx].C)
            atomicAdd(wgrad_dst(P, co, sidx, ch, 9) + tp, acc[tp][r]);
        }
      }
    if (t_w == 0) {
      float v = bsum;
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (q == 0) {
        const int co = (mt0 + m_w) * 16 + n;
        if (pdst) pdst[NT * 256 + m_w * 16 + n] = v;
        else if (A.db && nset == 0 && co < A.Cout && (mt0 + m_w) < P.NMTT) {
          atomicAdd(A.db + co, v);
          if (A.db2) atomicAdd(A.db2 + co, v);
        }
      }
    }
    return;
  }
  // one-tile blocks: the four waves hold K-split partial sums of the same tile -- reduce in LDS in four wave rounds
  __syncthreads();
  float* s_acc = smem;
  for (int w = 0; w < 4; ++w) {
    if (wv == w) {
#pragma unroll
      for (int tp = 0; tp < 9; ++tp)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float* d = &s_acc[(tp * 4 + r) * 64 + lane];
          *d = (w == 0) ? acc[tp][r] : *d + acc[tp][r];
        }
      float v = bsum;
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (q == 0) {
        float* d = &s_acc[NT * 256 + n];
        *d = (w == 0) ? v : *d + v;
      }
    }
    __syncthreads();
  }
  if (pdst) {
    for (int i = tid; i < NT * 256 + 16; i += 256) pdst[i] = s_acc[i];
    return;
  }
  for (int i = tid; i < NT * 256; i += 256) {
    const int ln = i & 63, r = (i >> 6) & 3, tp = i >> 8;
    const int co = mt0 * 16 + (ln >> 4) * 4 + r;
    if (mt0 >= P.NMTT || co >= A.Cout || nt0 >= P.NNTT) continue;
    int sidx = 0;
    while (sidx + 1 < A.nsrc && nt0 >= P.ntile_off[sidx + 1]) ++sidx;
    const int ch = (nt0 - P.ntile_off[sidx]) * 16 + (ln & 15);
    if (ch >= A.src[sidx].C) continue;
    atomicAdd(wgrad_dst(P, co, sidx, ch, 9) + tp, s_acc[i]);
  }
  if (A.db && nset == 0 && tid < 16) {
    const int co = mt0 * 16 + tid;
    if (co < A.Cout && mt0 < P.NMTT) {
      atomicAdd(A.db + co, s_acc[NT * 256 + tid]);
      if (A.db2) atomicAdd(A.db2 + co, s_acc[NT * 256 + tid]);
    }
  }
}

// ------------------------------------------------------------------------------------ weight gradient, 1x1 direct
// 1x1 / stride 1: both operands of a K step (4 pixels x 16 channels) are 4 x 64 contiguous bytes in NHWC, so
// a lane's MFMA operand IS one dword of global memory -- no LDS staging, no barriers in the main loop.  Every wave
// owns a contiguous range of K steps and keeps U steps (U * (NMT + NNT) loads) in flight.  The pixel index is flat
// over the batch; the image index (needed only for the per-image SE scale) is tracked per batch of steps.
// BF: the U = 8 four-pixel steps of a batch form two 16-pixel K steps; a lane packs its four loaded values (pixels
// 4q..4q+3 of one K16 step, i.e. step 4h + q, pixel j) into one bf16 fragment: 2 instead of 8 MFMAs per tile and batch.
template <int NMT, int NNT, int PM = 0>
__global__ __launch_bounds__(256) void wgrad_1x1_kernel(const WgradParams P) {
  constexpr int PMB = PM & 3;
  constexpr bool RP = (PM & 4) != 0;   // some operand is row-planar (LmnLay): its own instantiation, as in conv_tile_kernel
  constexpr bool BF = PMB >= 1;
  typedef typename ActT<PMB>::type TA;
  constexpr int U = 8;
  const lmn_wgrad_args_t& A = P.a;
  const uint32_t soff = A.seed_ctr ? *A.seed_ctr : 0u;  // device-side dropout stream offset (graph replays: one bump per step)
  const int lane = threadIdx.x & 63;
  const int q = lane >> 4, n = lane & 15;
  const int wvb = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wave = blockIdx.x * 4 + wvb, nwaves = gridDim.x * 4;
  const int mset = blockIdx.y / P.nsets_n, nset = blockIdx.y - mset * P.nsets_n;
  const int mt0 = mset * NMT, nt0 = nset * NNT;
  const int HW = A.Hout * A.Wout;
  const int NPX = A.B * HW;

  const TA* sptr[NNT];
  const float* sscale[NNT];
  int sC[NNT], scs[NNT], sflags[NNT], sch[NNT];
  uint32_t scho[NNT], srf[NNT];   // channel part of the offset, row term (LmnLay)
  uint32_t sseed[NNT];
  float sp[NNT], sik[NNT];
  bool any_tf = (A.dy_flags & LMN_SRC_DROP) != 0;
#pragma unroll
  for (int t = 0; t < NNT; ++t) {
    const int nt = nt0 + t;
    int s = 0;
    while (s + 1 < A.nsrc && nt >= P.ntile_off[s + 1]) ++s;
    const int ch = (nt - P.ntile_off[s]) * 16 + n;
    sptr[t] = (const TA*)A.src[s].ptr;
    sscale[t] = A.src[s].scale;
    sC[t] = A.src[s].C;
    scs[t] = RP ? P.lay_src[s].cs : A.src[s].cstride;
    srf[t] = RP ? (uint32_t)P.lay_src[s].rf : 0u;
    sflags[t] = A.src[s].flags;
    sseed[t] = A.src[s].drop_seed + soff;
    sp[t] = A.src[s].drop_p;
    sik[t] = P.inv_keep_src[s];
    sch[t] = (nt < P.NNTT && ch < A.src[s].C) ? ch : -1;
    { const int chc = sch[t] >= 0 ? sch[t] : 0; scho[t] = (uint32_t)(chc >> 2) * (uint32_t)P.lay_src[s].qs + (uint32_t)(chc & 3); }
    any_tf = any_tf || sflags[t] != 0 || sscale[t] != nullptr;
  }
  int mco[NMT];  // this lane's cout per tile, or -1
#pragma unroll
  for (int m = 0; m < NMT; ++m) mco[m] = ((mt0 + m) < P.NMTT && (mt0 + m) * 16 + n < A.Cout) ? (mt0 + m) * 16 + n : -1;
  uint32_t mcho[NMT];
#pragma unroll
  for (int m = 0; m < NMT; ++m) { const int cc = mco[m] >= 0 ? mco[m] : 0; mcho[m] = (uint32_t)(cc >> 2) * (uint32_t)P.lay_dy.qs + (uint32_t)(cc & 3); }

  f32x4 acc[NMT][NNT];
  float bsum[NMT];
#pragma unroll
  for (int m = 0; m < NMT; ++m) {
    bsum[m] = 0.f;
#pragma unroll
    for (int t = 0; t < NNT; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  const int total_steps = (NPX + 3) >> 2;
  const int sb = (int)(((int64_t)wave * total_steps) / nwaves), se = (int)(((int64_t)(wave + 1) * total_steps) / nwaves);
  // per-(image, channel) source scale (the SE gate): a lane's channel is fixed and a batch spans at most two images, so the
  // two candidate values live in registers and are refetched only when the image changes (loading them per element put
  // a second, dependent memory round trip into every batch: 85 us against 29 us for the untransformed layer at level 0)
  float sc0[NNT], sc1[NNT];
  int sc_b = -1;
#pragma unroll
  for (int t = 0; t < NNT; ++t) sc0[t] = sc1[t] = 1.f;
  for (int step0 = sb; step0 < se; step0 += U) {
    float av[U][NMT], bv[U][NNT];
    // ---- all loads of the batch, straight-line from clamped addresses
#pragma unroll
    for (int u = 0; u < U; ++u) {
      // fp32: lane (q, n) holds pixel q of four-pixel step u; bf16: pixel (u & 3) of lane group q in 16-pixel step u >> 2
      const int px = BF ? step0 * 4 + (u >> 2) * 16 + q * 4 + (u & 3) : (step0 + u) * 4 + q;
      const bool ok = (BF ? px < se * 4 : step0 + u < se) && px < NPX;
      const int ps = ok ? px : 0;
      if constexpr (RP) {
        const uint32_t rowp = lmn_div_row((uint32_t)ps, (uint32_t)P.rpw, P.rp_magic);   // (row term of the row-planar operands)
#pragma unroll
        for (int m = 0; m < NMT; ++m) av[u][m] = ld1((const TA*)A.dy + ((uint32_t)(ps * P.lay_dy.cs) + mcho[m] + rowp * (uint32_t)P.lay_dy.rf));
#pragma unroll
        for (int t = 0; t < NNT; ++t) bv[u][t] = ld1(sptr[t] + ((uint32_t)(ps * scs[t]) + scho[t] + rowp * srf[t]));
      } else {
#pragma unroll
        for (int m = 0; m < NMT; ++m) av[u][m] = ld1((const TA*)A.dy + (uint32_t)(ps * A.dy_cstride + (mco[m] >= 0 ? mco[m] : 0)));
#pragma unroll
        for (int t = 0; t < NNT; ++t) bv[u][t] = ld1(sptr[t] + (uint32_t)(ps * scs[t] + (sch[t] >= 0 ? sch[t] : 0)));
      }
    }
    // ---- on-load transforms (wave-uniform flags) and masking
    int b0 = 0;
    if (any_tf) {
      b0 = (step0 * 4) / HW;  // image of the batch's first pixel (scalar); a batch spans <= 2 images (HW >= 4U)
      if (b0 != sc_b) {       // wave-uniform, a handful of times per wave
        sc_b = b0;
        const int b1 = b0 + 1 < A.B ? b0 + 1 : b0;
#pragma unroll
        for (int t = 0; t < NNT; ++t) {
          if (sscale[t]) {
            const int chs = sch[t] >= 0 ? sch[t] : 0;
            sc0[t] = sscale[t][b0 * sC[t] + chs];
            sc1[t] = sscale[t][b1 * sC[t] + chs];
          }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int px = BF ? step0 * 4 + (u >> 2) * 16 + q * 4 + (u & 3) : (step0 + u) * 4 + q;
      const bool ok = (BF ? px < se * 4 : step0 + u < se) && px < NPX;
      const int ps = ok ? px : 0;
#pragma unroll
      for (int m = 0; m < NMT; ++m) {
        float v = av[u][m];
        if (A.dy_flags & LMN_SRC_DROP) v *= lmn_drop_scale(A.dy_seed + soff, (uint32_t)(ps * A.Cout + (mco[m] >= 0 ? mco[m] : 0)), A.dy_p, P.inv_keep_dy);
        av[u][m] = (ok && mco[m] >= 0) ? v : 0.f;
      }
#pragma unroll
      for (int t = 0; t < NNT; ++t) {
        float v = bv[u][t];
        if (any_tf) {
          const int chs = sch[t] >= 0 ? sch[t] : 0;
          if (sflags[t] & LMN_SRC_GELU) v = lmn_gelu(v);
          if (sflags[t] & LMN_SRC_DROP) v *= lmn_drop_scale(sseed[t], (uint32_t)(ps * sC[t] + chs), sp[t], sik[t]);
          if (sscale[t]) v *= (ps >= (b0 + 1) * HW) ? sc1[t] : sc0[t];
        }
        bv[u][t] = (ok && sch[t] >= 0) ? v : 0.f;
      }
    }
    if constexpr (BF) {
#pragma unroll
      for (int h = 0; h < U / 4; ++h) {
        uint2 af[NMT], bf[NNT];
#pragma unroll
        for (int m = 0; m < NMT; ++m) {
          bsum[m] += (av[4 * h][m] + av[4 * h + 1][m]) + (av[4 * h + 2][m] + av[4 * h + 3][m]);
          af[m] = uint2{pk_bf16(av[4 * h][m], av[4 * h + 1][m]), pk_bf16(av[4 * h + 2][m], av[4 * h + 3][m])};
        }
#pragma unroll
        for (int t = 0; t < NNT; ++t) bf[t] = uint2{pk_bf16(bv[4 * h][t], bv[4 * h + 1][t]), pk_bf16(bv[4 * h + 2][t], bv[4 * h + 3][t])};
#pragma unroll
        for (int t = 0; t < NNT; ++t)
#pragma unroll
          for (int m = 0; m < NMT; ++m) acc[m][t] = mfma_bf16(af[m], bf[t], acc[m][t]);
      }
    } else {
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int m = 0; m < NMT; ++m) bsum[m] += av[u][m];
#pragma unroll
      for (int t = 0; t < NNT; ++t)
#pragma unroll
        for (int m = 0; m < NMT; ++m) acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][m], bv[u][t], acc[m][t], 0, 0, 0);
    }
    }
  }

  // ---- block-level reduction in LDS (4 waves -> 1), plain stores in four wave rounds; layout as wgrad_lds_kernel
  constexpr int NT = NMT * NNT;
  __shared__ float s_acc[NT * 256 + NMT * 16];
  for (int w = 0; w < 4; ++w) {
    if (wvb == w) {
#pragma unroll
      for (int m = 0; m < NMT; ++m)
#pragma unroll
        for (int t = 0; t < NNT; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float* d = &s_acc[((m * NNT + t) * 4 + r) * 64 + lane];
            *d = (w == 0) ? acc[m][t][r] : *d + acc[m][t][r];
          }
#pragma unroll
      for (int m = 0; m < NMT; ++m) {
        float v = bsum[m];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (q == 0) {
          float* d = &s_acc[NT * 256 + m * 16 + n];
          *d = (w == 0) ? v : *d + v;
        }
      }
    }
    __syncthreads();
  }
  const int tid = threadIdx.x;
  if (P.partial) {
    float* dst = P.partial + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * (NT * 256 + NMT * 16);
    for (int i = tid; i < NT * 256 + NMT * 16; i += 256) dst[i] = s_acc[i];
    return;
  }
  for (int i = tid; i < NT * 256; i += 256) {
    const int ln = i & 63, r = (i >> 6) & 3, tl = i >> 8;
    const int t = tl % NNT, m = tl / NNT;
    const int qq = ln >> 4, nn = ln & 15;
    const int co = (mt0 + m) * 16 + qq * 4 + r;
    const int nt = nt0 + t;
    if ((mt0 + m) >= P.NMTT || co >= A.Cout || nt >= P.NNTT) continue;
    int sidx = 0;
    while (sidx + 1 < A.nsrc && nt >= P.ntile_off[sidx + 1]) ++sidx;
    const int ch = (nt - P.ntile_off[sidx]) * 16 + nn;
    if (ch >= A.src[sidx].C) continue;
    atomicAdd(wgrad_dst(P, co, sidx, ch, 1), s_acc[i]);
  }
  if (A.db && nset == 0) {
    for (int i = tid; i < NMT * 16; i += 256) {
      const int co = mt0 * 16 + i;
      if (co < A.Cout && (mt0 + i / 16) < P.NMTT) {
        atomicAdd(A.db + co, s_acc[NT * 256 + i]);
        if (A.db2) atomicAdd(A.db2 + co, s_acc[NT * 256 + i]);
      }
    }
  }
}

// ------------------------------------------------------------------------------------ weight gradient, 1x1, wave-staged
// Same job and same block / K-split geometry as wgrad_1x1_kernel, different data path.  The direct form feeds every MFMA
// operand with ONE dword (or one bf16) per lane straight from global memory: 256 B (128 B) per wave instruction, and the
// kernel runs at the pace of its vector-memory instructions (1.1 TB/s over the step's 1x1 weight gradients).  Here a wave
// moves a CHUNK of 16 pixels x 16 channels per tile with ONE 16-byte (8-byte) load per lane -- 1 KiB per wave instruction,
// whole 64-byte runs per pixel -- applies the on-load transforms once per element, parks the chunk in a WAVE-PRIVATE LDS
// slice ([pixel][16 channels] fp32: conflict-free ds_write_b128 / ds_read_b32) and reads the MFMA operands back in the
// (pixel = k, channel = row / column) form.  No block barrier in the main loop; the loads of chunk c+1 are in flight while
// chunk c is multiplied.
template <int NMT, int NNT, int PM>
__global__ __launch_bounds__(256) void wgrad_1x1w_kernel(const WgradParams P) {
  constexpr int PMB = PM & 3;
  constexpr bool RP = (PM & 4) != 0;   // some operand is row-planar (LmnLay): its own instantiation, as in conv_tile_kernel
  constexpr bool BF = PMB >= 1;
  typedef typename ActT<PMB>::type TA;
  constexpr int NTT = NMT + NNT;              // staged tiles per chunk: dy tiles first, then source tiles
  const lmn_wgrad_args_t& A = P.a;
  const uint32_t soff = A.seed_ctr ? *A.seed_ctr : 0u;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63;
  const int q = lane >> 4, n = lane & 15;
  const int wvb = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* WS = smem + wvb * (NTT * 256);       // this wave's slice: NTT tiles x [16 px][16 ch]
  const int wave = blockIdx.x * 4 + wvb, nwaves = gridDim.x * 4;
  const int mset = blockIdx.y / P.nsets_n, nset = blockIdx.y - mset * P.nsets_n;
  const int mt0 = mset * NMT, nt0 = nset * NNT;
  const int HW = A.Hout * A.Wout;
  const int NPX = A.B * HW;
  // staging role of this lane: pixel lp of the chunk, channel quad lq of every tile
  const int lp = lane >> 2, lq = lane & 3;

  const TA* sptr[NNT];
  const float* sscale[NNT];
  int sC[NNT], scs[NNT], sflags[NNT], sch4[NNT];
  uint32_t scho[NNT], srf[NNT];   // channel part of the offset, row term (LmnLay)
  uint32_t sseed[NNT];
  float sp[NNT], sik[NNT];
#pragma unroll
  for (int t = 0; t < NNT; ++t) {
    const int nt = nt0 + t;
    int s = 0;
    while (s + 1 < A.nsrc && nt >= P.ntile_off[s + 1]) ++s;
    const int ch = (nt - P.ntile_off[s]) * 16 + lq * 4;
    sptr[t] = (const TA*)A.src[s].ptr;
    sscale[t] = A.src[s].scale;
    sC[t] = A.src[s].C;
    scs[t] = RP ? P.lay_src[s].cs : A.src[s].cstride;
    srf[t] = RP ? (uint32_t)P.lay_src[s].rf : 0u;
    sflags[t] = A.src[s].flags;
    sseed[t] = A.src[s].drop_seed + soff;
    sp[t] = A.src[s].drop_p;
    sik[t] = P.inv_keep_src[s];
    sch4[t] = (nt < P.NNTT && ch < A.src[s].C) ? ch : -1;   // first channel of this lane's quad, or -1
    scho[t] = (uint32_t)((sch4[t] >= 0 ? sch4[t] : 0) >> 2) * (uint32_t)P.lay_src[s].qs;
  }
  int mco4[NMT];
#pragma unroll
  for (int m = 0; m < NMT; ++m) mco4[m] = ((mt0 + m) < P.NMTT && (mt0 + m) * 16 + lq * 4 < A.Cout) ? (mt0 + m) * 16 + lq * 4 : -1;
  uint32_t mcho[NMT];
#pragma unroll
  for (int m = 0; m < NMT; ++m) mcho[m] = (uint32_t)((mco4[m] >= 0 ? mco4[m] : 0) >> 2) * (uint32_t)P.lay_dy.qs;

  f32x4 acc[NMT][NNT];
  float bsum[NMT];
#pragma unroll
  for (int m = 0; m < NMT; ++m) {
    bsum[m] = 0.f;
#pragma unroll
    for (int t = 0; t < NNT; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // this wave's pixels: whole chunks of 16
  const int total_chunks = (NPX + 15) >> 4;
  const int cb = (int)(((int64_t)wave * total_chunks) / nwaves), ce = (int)(((int64_t)(wave + 1) * total_chunks) / nwaves);
  f32x4 stg[NTT];
  auto chunk_load = [&](int c) {
    const int px = c * 16 + lp;
    const int ps = px < NPX ? px : 0;
    if constexpr (RP) {
      const uint32_t rowp = lmn_div_row((uint32_t)ps, (uint32_t)P.rpw, P.rp_magic);   // (row term of the row-planar operands)
#pragma unroll
      for (int m = 0; m < NMT; ++m) stg[m] = ld4((const TA*)A.dy + ((uint32_t)(ps * P.lay_dy.cs) + mcho[m] + rowp * (uint32_t)P.lay_dy.rf));
#pragma unroll
      for (int t = 0; t < NNT; ++t) stg[NMT + t] = ld4(sptr[t] + ((uint32_t)(ps * scs[t]) + scho[t] + rowp * srf[t]));
    } else {
#pragma unroll
      for (int m = 0; m < NMT; ++m) stg[m] = ld4((const TA*)A.dy + (uint32_t)(ps * A.dy_cstride + (mco4[m] >= 0 ? mco4[m] : 0)));
#pragma unroll
      for (int t = 0; t < NNT; ++t) stg[NMT + t] = ld4(sptr[t] + (uint32_t)(ps * scs[t] + (sch4[t] >= 0 ? sch4[t] : 0)));
    }
  };
  auto chunk_put = [&](int c) {   // transforms, masking, LDS
    const int px = c * 16 + lp;
    const bool ok = px < NPX;
    const int ps = ok ? px : 0;
#pragma unroll
    for (int m = 0; m < NMT; ++m) {
      f32x4 v = stg[m];
      if (A.dy_flags & LMN_SRC_DROP) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] *= lmn_drop_scale(A.dy_seed + soff, (uint32_t)(ps * A.Cout + (mco4[m] >= 0 ? mco4[m] : 0) + k), A.dy_p, P.inv_keep_dy);
      }
      if (!(ok && mco4[m] >= 0)) v = f32x4{0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(&WS[m * 256 + lp * 16 + lq * 4]) = v;
    }
#pragma unroll
    for (int t = 0; t < NNT; ++t) {
      f32x4 v = stg[NMT + t];
      const int chs = sch4[t] >= 0 ? sch4[t] : 0;
      if (sflags[t] & LMN_SRC_GELU) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = lmn_gelu(v[k]);
      }
      if (sflags[t] & LMN_SRC_DROP) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] *= lmn_drop_scale(sseed[t], (uint32_t)(ps * sC[t] + chs + k), sp[t], sik[t]);
      }
      if (sscale[t]) v *= ld4(sscale[t] + (ps / HW) * sC[t] + chs);
      if (!(ok && sch4[t] >= 0)) v = f32x4{0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(&WS[(NMT + t) * 256 + lp * 16 + lq * 4]) = v;
    }
  };
  if (cb < ce) chunk_load(cb);
  for (int c = cb; c < ce; ++c) {
    chunk_put(c);                       // (LDS operations of one wave execute in order: the previous chunk's reads are done)
    if (c + 1 < ce) chunk_load(c + 1);  // in flight during the MFMAs below
    if constexpr (BF) {                 // one K16 step: lane (q, n) owns pixels 4q..4q+3, channel n
      uint2 af[NMT], bf_[NNT];
#pragma unroll
      for (int m = 0; m < NMT; ++m) {
        const float a0 = WS[m * 256 + (4 * q) * 16 + n], a1 = WS[m * 256 + (4 * q + 1) * 16 + n];
        const float a2 = WS[m * 256 + (4 * q + 2) * 16 + n], a3 = WS[m * 256 + (4 * q + 3) * 16 + n];
        bsum[m] += (a0 + a1) + (a2 + a3);
        af[m] = uint2{pk_bf16(a0, a1), pk_bf16(a2, a3)};
      }
#pragma unroll
      for (int t = 0; t < NNT; ++t) {
        const float* b = &WS[(NMT + t) * 256 + (4 * q) * 16 + n];
        bf_[t] = uint2{pk_bf16(b[0], b[16]), pk_bf16(b[32], b[48])};
      }
#pragma unroll
      for (int t = 0; t < NNT; ++t)
#pragma unroll
        for (int m = 0; m < NMT; ++m) acc[m][t] = mfma_bf16(af[m], bf_[t], acc[m][t]);
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {     // four K4 steps: lane (q, n) owns pixel 4k + q, channel n
        float av[NMT], bv[NNT];
#pragma unroll
        for (int m = 0; m < NMT; ++m) av[m] = WS[m * 256 + (4 * k + q) * 16 + n];
#pragma unroll
        for (int t = 0; t < NNT; ++t) bv[t] = WS[(NMT + t) * 256 + (4 * k + q) * 16 + n];
#pragma unroll
        for (int m = 0; m < NMT; ++m) bsum[m] += av[m];
#pragma unroll
        for (int t = 0; t < NNT; ++t)
#pragma unroll
          for (int m = 0; m < NMT; ++m) acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], bv[t], acc[m][t], 0, 0, 0);
      }
    }
  }

  // ---- block-level reduction in LDS (4 waves -> 1), plain stores in four wave rounds; layout as wgrad_lds_kernel
  constexpr int NT = NMT * NNT;
  __syncthreads();                      // every wave is done with its staging slice
  float* s_acc = smem;                  // reuse (host sizes the allocation to max(staging, NT*256 + NMT*16) floats)
  for (int w = 0; w < 4; ++w) {
    if (wvb == w) {
#pragma unroll
      for (int m = 0; m < NMT; ++m)
#pragma unroll
        for (int t = 0; t < NNT; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float* d = &s_acc[((m * NNT + t) * 4 + r) * 64 + lane];
            *d = (w == 0) ? acc[m][t][r] : *d + acc[m][t][r];
          }
#pragma unroll
      for (int m = 0; m < NMT; ++m) {
        float v = bsum[m];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (q == 0) {
          float* d = &s_acc[NT * 256 + m * 16 + n];
          *d = (w == 0) ? v : *d + v;
        }
      }
    }
    __syncthreads();
  }
  const int tid = threadIdx.x;
  if (P.partial) {
    float* dst = P.partial + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * (NT * 256 + NMT * 16);
    for (int i = tid; i < NT * 256 + NMT * 16; i += 256) dst[i] = s_acc[i];
    return;
  }
  for (int i = tid; i < NT * 256; i += 256) {
    const int ln = i & 63, r = (i >> 6) & 3, tl = i >> 8;
    const int t = tl % NNT, m = tl / NNT;
    const int qq = ln >> 4, nn = ln & 15;
    const int co = (mt0 + m) * 16 + qq * 4 + r;
    const int nt = nt0 + t;
    if ((mt0 + m) >= P.NMTT || co >= A.Cout || nt >= P.NNTT) continue;
    int sidx = 0;
    while (sidx + 1 < A.nsrc && nt >= P.ntile_off[sidx + 1]) ++sidx;
    const int ch = (nt - P.ntile_off[sidx]) * 16 + nn;
    if (ch >= A.src[sidx].C) continue;
    atomicAdd(wgrad_dst(P, co, sidx, ch, 1), s_acc[i]);
  }
  if (A.db && nset == 0) {
    for (int i = tid; i < NMT * 16; i += 256) {
      const int co = mt0 * 16 + i;
      if (co < A.Cout && (mt0 + i / 16) < P.NMTT) {
        atomicAdd(A.db + co, s_acc[NT * 256 + i]);
        if (A.db2) atomicAdd(A.db2 + co, s_acc[NT * 256 + i]);
      }
    }
  }
}

// second stage: dW[co][ci][tap] += sum over the K-split blocks of their partial tiles (fixed order => deterministic).
// 1024 threads = epq groups of 4 consecutive elements (one 16 B load each) x ksl K-slices, 4 loads in flight per thread: the
// 768 partials of a 3x3 layer are 28 MB, and with 4 B loads the 145 blocks of that launch had 2.4 MB in flight (1 TB/s).
template <int TAPS, int NMT, int NNT>
__global__ __launch_bounds__(1024) void wgrad_reduce_kernel(const WgradParams P, int nblk, int ksl) {
  // ksl = 1 .. 64, a power of two sized to nblk on the host (with 2-3 partials, many slices leave most waves idle)
  const lmn_wgrad_args_t& A = P.a;
  constexpr int NT = TAPS * NMT * NNT, PER = NT * 256 + NMT * 16;  // PER % 4 == 0
  __shared__ __attribute__((aligned(16))) float red[4096];
  const int mset = blockIdx.y / P.nsets_n, nset = blockIdx.y - mset * P.nsets_n;
  const int mt0 = mset * NMT, nt0 = nset * NNT;
  const float* src = P.partial + (int64_t)blockIdx.y * nblk * PER;
  const int epq = 1024 / ksl, epb = epq * 4;
  const int q = threadIdx.x & (epq - 1), ks = threadIdx.x / epq;
  const int i4 = (blockIdx.x * epq + q) * 4;
  f32x4 s4 = f32x4{0.f, 0.f, 0.f, 0.f};
  if (i4 < PER) {
    const float* sp = src + i4;
#pragma unroll 4
    for (int k = ks; k < nblk; k += ksl) s4 += *reinterpret_cast<const f32x4*>(sp + (int64_t)k * PER);
  }
  *reinterpret_cast<f32x4*>(&red[threadIdx.x * 4]) = s4;
  __syncthreads();
  for (int e = threadIdx.x; e < epb; e += 1024) {  // (epb > 1024 for ksl < 4)
  const int i = blockIdx.x * epb + e;
  if (i >= PER) return;
  float sum = 0.f;
  for (int k = 0; k < ksl; ++k) sum += red[k * epb + e];
  if (i < NT * 256) {
    const int ln = i & 63, r = (i >> 6) & 3, tile = i >> 8;
    const int t = tile % NNT, m = (tile / NNT) % NMT, tp = tile / (NNT * NMT);
    const int qq = ln >> 4, nn = ln & 15;
    const int co = (mt0 + m) * 16 + qq * 4 + r;
    const int nt = nt0 + t;
    if ((mt0 + m) >= P.NMTT || co >= A.Cout || nt >= P.NNTT) continue;
    int sidx = 0;
    while (sidx + 1 < A.nsrc && nt >= P.ntile_off[sidx + 1]) ++sidx;
    const int ch = (nt - P.ntile_off[sidx]) * 16 + nn;
    if (ch >= A.src[sidx].C) continue;
    wgrad_dst(P, co, sidx, ch, TAPS)[tp] += sum;
  } else if (A.db && nset == 0) {
    const int j = i - NT * 256, co = mt0 * 16 + j;
    if (co < A.Cout && (mt0 + j / 16) < P.NMTT) {
      A.db[co] += sum;
      if (A.db2) A.db2[co] += sum;
    }
  }
  }
}

// All deferred reductions of a pass (lmn_wgrad_args_t.defer_reduce) in ONE launch: block b finds its job by bisection over the
// jobs' first-block table, then sums one slice of one tile set exactly as wgrad_reduce_kernel does (same fixed order =>
// deterministic).  82 reduction launches of 8-11 us each per training step become one or a few.
__global__ __launch_bounds__(1024) void wgrad_reduce_batch_kernel(const lmn_reduce_job_t* __restrict__ jobs, int njobs) {
  __shared__ __attribute__((aligned(16))) float red[4096];
  int lo = 0, hi = njobs - 1;
  const int64_t b = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].first_block <= b) lo = mid; else hi = mid - 1;
  }
  const lmn_reduce_job_t& J = jobs[lo];
  const int rel = (int)(b - J.first_block);
  const int set = rel / J.blocks_per_set, bx = rel - set * J.blocks_per_set;
  const int NMT = J.NMT, NNT = J.NNT, TAPS = J.taps, PER = J.per, nblk = J.nblk, ksl = J.ksl;
  const int NT = TAPS * NMT * NNT;
  const int mset = set / J.nsets_n, nset = set - mset * J.nsets_n;
  const int mt0 = mset * NMT, nt0 = nset * NNT;
  const float* src = J.partial + (int64_t)set * nblk * PER;
  const int epq = 1024 / ksl, epb = epq * 4;
  const int q = threadIdx.x & (epq - 1), ks = threadIdx.x / epq;
  const int i4 = (bx * epq + q) * 4;
  f32x4 s4 = f32x4{0.f, 0.f, 0.f, 0.f};
  if (i4 < PER) {
    const float* sp = src + i4;
#pragma unroll 4
    for (int k = ks; k < nblk; k += ksl) s4 += *reinterpret_cast<const f32x4*>(sp + (int64_t)k * PER);
  }
  *reinterpret_cast<f32x4*>(&red[threadIdx.x * 4]) = s4;
  __syncthreads();
  for (int e = threadIdx.x; e < epb; e += 1024) {  // (epb > 1024 for ksl < 4)
    const int i = bx * epb + e;
    if (i >= PER) return;
    float sum = 0.f;
    for (int k = 0; k < ksl; ++k) sum += red[k * epb + e];
    if (i < NT * 256) {
      const int ln = i & 63, r = (i >> 6) & 3, tile = i >> 8;
      const int t = tile % NNT, m = (tile / NNT) % NMT, tp = tile / (NNT * NMT);
      const int qq = ln >> 4, nn = ln & 15;
      const int co = (mt0 + m) * 16 + qq * 4 + r;
      const int nt = nt0 + t;
      if ((mt0 + m) >= J.NMTT || co >= J.Cout || nt >= J.NNTT) continue;
      int sidx = 0;
      while (sidx + 1 < J.nsrc && nt >= J.ntile_off[sidx + 1]) ++sidx;
      const int ch = (nt - J.ntile_off[sidx]) * 16 + nn;
      if (ch >= J.srcC[sidx]) continue;
      float* own = J.dW_src[sidx];
      float* d = own ? own + ((int64_t)co * J.srcC[sidx] + ch) * TAPS : J.dW + ((int64_t)co * J.Cin + J.cbase[sidx] + ch) * TAPS;
      d[tp] += sum;
    } else if (J.db && nset == 0) {
      const int j = i - NT * 256, co = mt0 * 16 + j;
      if (co < J.Cout && (mt0 + j / 16) < J.NMTT) {
        J.db[co] += sum;
        if (J.db2) J.db2[co] += sum;
      }
    }
  }
}

}  // namespace

thread_local char g_lmn_err[256] = {0};

// (a function template of its own: only there is the `if constexpr` branch of the tile shapes wgrad3_kernel does not have discarded
// without being instantiated)
template <int M, int N, int PM>
static void wgrad3_launch(dim3 grid, size_t shmem, hipStream_t st, const WgradParams& P) {
  if constexpr (M * N == 4 || M * N == 1) LMN_LAUNCH((wgrad3_kernel<M, N, PM>), grid, dim3(256), shmem, st, P);
}

extern "C" {

int lmn_abi_version(void) { return LMN_ABI_VERSION; }
int lmn_sizeof_conv_args(void) { return (int)sizeof(lmn_conv_args_t); }
int lmn_sizeof_src(void) { return (int)sizeof(lmn_src_t); }
int lmn_sizeof_wgrad_args(void) { return (int)sizeof(lmn_wgrad_args_t); }
int lmn_sizeof_pack_job(void) { return (int)sizeof(lmn_pack_job_t); }

// cout x cin 16-channel tiles per block.  The direct 1x1 kernel holds up to 4 accumulator tiles in any shape (a
// one-tile-wide side is read exactly once); the LDS-staged kernel is instantiated for 1 or 2 tiles per side.
static void wgrad_tile_shape(const lmn_wgrad_args_t& a, int nmtt, int nntt, int* NMT, int* NNT) {
  bool rp_any = a.dy_rp_w != 0;
  for (int s = 0; s < a.nsrc && s < 3; ++s) rp_any = rp_any || a.src[s].rp_w != 0;
  const bool direct = a.ksize == 1 && a.stride == 1 && ((int64_t)a.Hout * a.Wout >= 32 || rp_any);   // (row-planar operands: direct kernels only)
  if (direct) {
    // 1x1: a block of (M x N) tiles reads its M dy tiles and N source tiles once per pixel; the grid's tile sets re-read
    // dy ceil(nntt/N) times and the sources ceil(nmtt/M) times: take the instantiated shape with the fewest bytes per
    // pixel (ties: fewer accumulator tiles)
    static const int shapes[][2] = {{1, 1}, {1, 2}, {1, 3}, {1, 4}, {2, 1}, {3, 1}, {4, 1}, {2, 2}, {2, 3}, {3, 2}, {2, 4}, {4, 2}};
    int cin = 0;
    for (int s = 0; s < a.nsrc && s < 3; ++s) cin += a.src[s].C;
    long best = -1;
    for (const auto& sh : shapes) {
      const int M = sh[0], N = sh[1];
      if (M > nmtt && M > 1) continue;
      if (N > nntt && N > 1) continue;
      const long cost = ((long)((nmtt + M - 1) / M) * cin + (long)((nntt + N - 1) / N) * a.Cout) * 64 + M * N;
      if (best < 0 || cost < best) { best = cost; *NMT = M; *NNT = N; }
    }
    return;
  }
  const bool small = nmtt == 1 || nntt == 1;
  *NMT = small ? 1 : 2; *NNT = small ? 1 : 2;  // LDS-staged kernel: (1,2)/(2,1) measured slower here
}

// 3x3 stride-1 layers on maps >= 32 wide run wgrad3_kernel: <= 154 VGPRs and <= 48 KB of LDS per block, so THREE blocks are
// resident per CU and the K-split aims at 768 blocks instead of 512.  LMN_WGRAD_V1=0 selects the
// general kernel (A/B runs), LMN_WGRAD_CAP overrides the block count.
static int wgrad_v1_env() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("LMN_WGRAD_V1"); v = e ? atoi(e) : 1; }
  return v;
}
static bool wgrad_v1(const lmn_wgrad_args_t& a) { return a.ksize == 3 && a.stride == 1 && wgrad_v1_env() != 0; }
static int wgrad_blocks_total(const lmn_wgrad_args_t& a) {
  static int c = -1;
  if (c < 0) { const char* e = getenv("LMN_WGRAD_CAP"); c = e ? atoi(e) : 0; }
  if (!wgrad_v1(a)) return 512;
  // 512 = two resident blocks per CU.  Alone the kernel is fastest at 768 (three per CU), but it runs on the weight-gradient stream
  // beside the backward chain: inside the step 512 blocks measured 14.59 ms against 14.68 (768), 14.63 (384), 14.65 (640), 14.71 (256)
  // -- fewer partials to write and re-read (28 -> 19 MB per 24-channel layer) and more CU time left to the chain (round 4, three
  // runs each on one box).  (bf16 storage would fit 4 per CU: 1024 blocks measured 3 % slower.)
  return c > 0 ? c : 512;
}

int64_t lmn_conv_wgrad_workspace(const lmn_wgrad_args_t* a) {
  if (!a) return 0;
  int nntt = 0;
  for (int s = 0; s < a->nsrc && s < 3; ++s) nntt += (a->src[s].C + 15) / 16;
  const int nmtt = (a->Cout + 15) / 16;
  int NMT, NNT;
  wgrad_tile_shape(*a, nmtt, nntt, &NMT, &NNT);
  const int gy = ((nmtt + NMT - 1) / NMT) * ((nntt + NNT - 1) / NNT);
  const int64_t per = (int64_t)a->ksize * a->ksize * NMT * NNT * 256 + NMT * 16;
  const int btot = wgrad_blocks_total(*a);
  const int64_t cap = btot / gy > 2 ? btot / gy : 2;  // upper bound of the K-split block count (see lmn_conv_wgrad)
  const int64_t need = gy * cap * per;
  return need <= (int64_t)(16 << 20) ? need : 0;  // at most 64 MB of partials; larger problems use atomics
}
const char* lmn_last_error(void) { return g_lmn_err; }
#ifdef LMN_CT_TIMING
int lmn_ct_timing(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ct_timing), sizeof(unsigned long long) * n);
}
#endif
#ifdef LMN_WG_TIMING
int lmn_wg_timing(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wg_timing), sizeof(unsigned long long) * n);
}
#endif

int64_t lmn_conv_pack_size(int ksize, int Cout, int nsrc, const int32_t* c) {
  int64_t nkb = 0;
  for (int s = 0; s < nsrc; ++s) nkb += (c[s] + 15) / 16;
  return (int64_t)ksize * ksize * nkb * ((Cout + 15) / 16) * 256;
}

int lmn_conv_pack(const float* w, float* wpack, int ksize, int Cout, int Cin, int nsrc, const int32_t* c,
                  int transposed, int row_off, int rows, int dtype, lmn_stream_t stream) {
  LMN_REQUIRE(w && wpack && c, "conv_pack: null pointer");
  LMN_REQUIRE(dtype == LMN_F32 || dtype == LMN_BF16, "conv_pack: dtype %d", dtype);
  if (g_lmn_rec && nsrc >= 1 && nsrc <= 3) {
    int32_t cc[3] = {c[0], nsrc > 1 ? c[1] : 0, nsrc > 2 ? c[2] : 0};
    const int32_t c0 = cc[0], c1 = cc[1], c2 = cc[2];
    lmn_rec_push([=]() -> int { const int32_t ca[3] = {c0, c1, c2}; return lmn_conv_pack(w, wpack, ksize, Cout, Cin, nsrc, ca, transposed, row_off, rows, dtype, stream); }, "lmn_conv_pack(");
  }
  LMN_REQUIRE(ksize == 1 || ksize == 3, "conv_pack: ksize %d", ksize);
  LMN_REQUIRE(nsrc >= 1 && nsrc <= 3, "conv_pack: nsrc %d", nsrc);
  int csum = 0;
  for (int s = 0; s < nsrc; ++s) csum += c[s];
  if (!transposed) {
    LMN_REQUIRE(csum == Cin || (nsrc == 1 && csum > Cin && csum < Cin + 4),
                "conv_pack: sources sum to %d channels, weight has %d", csum, Cin);
  } else {
    LMN_REQUIRE(nsrc == 1 && c[0] >= Cout && c[0] < Cout + 4, "conv_pack(transposed): one source of Cout=%d channels expected", Cout);
    LMN_REQUIRE(row_off >= 0 && rows > 0 && row_off + rows < Cin + 4, "conv_pack(transposed): rows [%d,+%d) of %d", row_off, rows, Cin);
  }
  const int nrows = transposed ? rows : (Cout + 3) / 4 * 4;
  const int64_t total = lmn_conv_pack_size(ksize, nrows, nsrc, c);
  LMN_REQUIRE(total < (1LL << 31), "conv_pack: %lld packed elements (32-bit index arithmetic)", (long long)total);
  const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  PackGeom g;
  g.taps = ksize * ksize; g.Cout = Cout; g.Cin = Cin; g.nsrc = nsrc;
  g.cs[0] = c[0]; g.cs[1] = nsrc > 1 ? c[1] : 0; g.cs[2] = nsrc > 2 ? c[2] : 0;
  g.transposed = transposed; g.row_off = row_off; g.rows = rows;
  g.bf16 = dtype == LMN_BF16;
  LMN_LAUNCH(conv_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, wpack, g, total);
  return lmn_launch_status("conv_pack");
}

int lmn_conv_pack_batch(const lmn_pack_job_t* jobs_dev, int njobs, int64_t total_blocks, lmn_stream_t stream) {
  LMN_REC(lmn_conv_pack_batch(jobs_dev, njobs, total_blocks, stream));
  LMN_REQUIRE(jobs_dev && njobs > 0 && total_blocks > 0 && total_blocks < (1LL << 31), "conv_pack_batch: bad job table");
  LMN_LAUNCH(conv_pack_batch_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, jobs_dev, njobs);
  return lmn_launch_status("conv_pack_batch");
}

int lmn_conv_fwd(const lmn_conv_args_t* args, lmn_stream_t stream) {
  LMN_REQUIRE(args, "conv_fwd: null args");
  if (g_lmn_rec) {
    const lmn_conv_args_t copy = *args;
    lmn_rec_push([copy, stream]() -> int { return lmn_conv_fwd(&copy, stream); }, "lmn_conv_fwd(");
  }
  const lmn_conv_args_t& A = *args;
  LMN_REQUIRE(A.ksize == 1 || A.ksize == 3, "conv_fwd: ksize %d", A.ksize);
  LMN_REQUIRE(A.stride == 1 || A.stride == 2, "conv_fwd: stride %d", A.stride);
  LMN_REQUIRE(A.nsrc >= 1 && A.nsrc <= 3, "conv_fwd: nsrc %d", A.nsrc);
  LMN_REQUIRE(A.Cout > 0 && A.Cout % 4 == 0, "conv_fwd: Cout %d must be a positive multiple of 4", A.Cout);
  LMN_REQUIRE(A.mma_dtype == LMN_F32 || A.mma_dtype == LMN_BF16, "conv_fwd: mma_dtype %d", A.mma_dtype);
  LMN_REQUIRE(A.act_dtype == LMN_F32 || (A.act_dtype == LMN_BF16 && A.mma_dtype == LMN_BF16),
              "conv_fwd: act_dtype %d with mma_dtype %d (bf16 storage needs bf16 matrix-core operands)", A.act_dtype, A.mma_dtype);
  LMN_REQUIRE(A.B > 0 && A.Hout > 0 && A.Wout > 0 && A.Hin > 0 && A.Win > 0, "conv_fwd: empty tensor");
  LMN_REQUIRE(A.wpack, "conv_fwd: null packed weights");
  LMN_REQUIRE(A.out || A.stats, "conv_fwd: neither out nor stats requested");
  LMN_REQUIRE(!(A.transposed && A.nsrc != 1), "conv_fwd: transposed form takes one source");
  ConvParams P;
  P.det_stats = nullptr;
  P.a = A;
  P.NKB = 0;
  for (int s = 0; s < 3; ++s) {
    P.nkb[s] = P.kb_off[s] = 0;
    P.inv_keep_src[s] = 1.f;
  }
  for (int s = 0; s < A.nsrc; ++s) {
    LMN_REQUIRE(A.src[s].ptr && A.src[s].C > 0 && A.src[s].C % 4 == 0 && A.src[s].cstride >= A.src[s].C && A.src[s].cstride % 4 == 0,
                "conv_fwd: source %d: C=%d cstride=%d (need multiples of 4, cstride>=C)", s, A.src[s].C, A.src[s].cstride);
    P.nkb[s] = (A.src[s].C + 15) / 16;
    P.kb_off[s] = P.NKB;
    P.NKB += P.nkb[s];
    if (A.src[s].flags & LMN_SRC_DROP) {
      LMN_REQUIRE(A.src[s].drop_p >= 0.f && A.src[s].drop_p < 1.f, "conv_fwd: source dropout p");
      P.inv_keep_src[s] = 1.f / (1.f - A.src[s].drop_p);
    }
  }
  {  // row-planar operands (RP4): whole tensors of a 1x1 stride-1 call, one image width for all of them
    int rw = 0;
    for (int s = 0; s < 3; ++s) {
      const int w = s < A.nsrc ? A.src[s].rp_w : 0;
      P.lay_src[s] = lmn_lay_make(w, s < A.nsrc ? A.src[s].C : 4, s < A.nsrc ? A.src[s].cstride : 4);
      if (w) {
        LMN_REQUIRE(w > 0 && A.src[s].cstride == A.src[s].C && (rw == 0 || rw == w), "conv_fwd: row-planar source %d must be a whole tensor (cstride == C) of the call's width", s);
        rw = w;
      }
    }
    const int ow = A.out ? A.out_rp_w : 0, aw = A.aux ? A.aux_rp_w : 0;
    P.lay_out = lmn_lay_make(ow, A.Cout, A.out_cstride);
    P.lay_aux = lmn_lay_make(aw, A.Cout, A.aux_cstride);
    if (ow) { LMN_REQUIRE(ow > 0 && A.out_cstride == A.Cout && (rw == 0 || rw == ow), "conv_fwd: row-planar out must be a whole tensor of the call's width"); rw = ow; }
    if (aw) { LMN_REQUIRE(aw > 0 && A.aux_cstride == A.Cout && (rw == 0 || rw == aw), "conv_fwd: row-planar aux must be a whole tensor of the call's width"); rw = aw; }
    if (rw) {
      LMN_REQUIRE(A.ksize == 1 && A.stride == 1 && ((int64_t)A.Hin * A.Win) % rw == 0 && ((int64_t)A.Hout * A.Wout) % rw == 0,
                  "conv_fwd: row-planar operands belong to 1x1 stride-1 calls whose image is a whole number of rows of width %d", rw);
    }
    P.rpw = rw;
    P.rp_magic = lmn_div_magic(rw);
  }
  {  // kernels index with 32-bit element offsets
    const int64_t lim = (1LL << 31) - 1;
    for (int s = 0; s < A.nsrc; ++s)
      LMN_REQUIRE((int64_t)A.B * A.Hin * A.Win * A.src[s].cstride <= lim, "conv_fwd: source %d larger than 2^31 elements", s);
    const int64_t ocs = A.out_cstride > A.aux_cstride ? A.out_cstride : A.aux_cstride;
    LMN_REQUIRE((int64_t)A.B * A.Hout * A.Wout * (ocs > A.res_cstride ? ocs : A.res_cstride) <= lim && (int64_t)A.B * A.Hout * A.Wout * A.Cout <= lim,
                "conv_fwd: output larger than 2^31 elements");
  }
  LMN_REQUIRE(A.drop_p >= 0.f && A.drop_p < 1.f, "conv_fwd: dropout p %f", A.drop_p);
  P.inv_keep_ep = 1.f / (1.f - A.drop_p);
  if (A.out) LMN_REQUIRE(A.out_cstride >= A.Cout && A.out_cstride % 4 == 0, "conv_fwd: out_cstride %d", A.out_cstride);
  if (A.residual) LMN_REQUIRE(A.res_cstride >= A.Cout && A.res_cstride % 4 == 0, "conv_fwd: res_cstride %d", A.res_cstride);
  if (A.aux) LMN_REQUIRE(A.aux_cstride >= A.Cout && A.aux_cstride % 4 == 0, "conv_fwd: aux_cstride %d", A.aux_cstride);
  switch (A.epilogue) {
    case LMN_EP_LINEAR: break;
    case LMN_EP_AFFINE_ACT: LMN_REQUIRE((A.p0 && A.p1) || A.fin.mode == LMN_FIN_BN, "conv_fwd: AFFINE_ACT needs p0,p1 (or fin)"); break;
    case LMN_EP_DGELU: LMN_REQUIRE(A.aux, "conv_fwd: DGELU needs aux"); break;
    case LMN_EP_BN_BWD1: LMN_REQUIRE(A.aux && A.p0 && A.p1 && A.p2 && A.p3 && A.stats, "conv_fwd: BN_BWD1 operands"); break;
    case LMN_EP_BN_BWD2:
      LMN_REQUIRE(A.aux && A.p0 && A.p1 && ((A.p2 && A.p3 && A.p4) || A.fin.mode == LMN_FIN_BN_BWD) && (!A.p5 == !A.p6), "conv_fwd: BN_BWD2 operands");
      break;
    case LMN_EP_SE_BWD: LMN_REQUIRE(A.aux && A.stats, "conv_fwd: SE_BWD operands"); break;
    default: LMN_REQUIRE(false, "conv_fwd: epilogue %d", A.epilogue);
  }
  if (A.stats_mode == LMN_STATS_SUM_SQ) LMN_REQUIRE(A.stats, "conv_fwd: SUM_SQ needs stats");
  if (!A.transposed) {
    LMN_REQUIRE(A.Hout == (A.Hin + 2 * (A.ksize / 2) - A.ksize) / A.stride + 1 && A.Wout == (A.Win + 2 * (A.ksize / 2) - A.ksize) / A.stride + 1,
                "conv_fwd: out %dx%d inconsistent with in %dx%d k%d s%d", A.Hout, A.Wout, A.Hin, A.Win, A.ksize, A.stride);
  } else {
    LMN_REQUIRE(A.Hin == (A.Hout + 2 * (A.ksize / 2) - A.ksize) / A.stride + 1 && A.Win == (A.Wout + 2 * (A.ksize / 2) - A.ksize) / A.stride + 1,
                "conv_fwd(T): dy %dx%d inconsistent with dx %dx%d k%d s%d", A.Hin, A.Win, A.Hout, A.Wout, A.ksize, A.stride);
  }
  LMN_REQUIRE(A.fin.mode == LMN_FIN_NONE || A.fin.mode == LMN_FIN_BN || A.fin.mode == LMN_FIN_BN_BWD, "conv_fwd: fin.mode %d", A.fin.mode);
  if (A.fin.mode != LMN_FIN_NONE) {
    LMN_REQUIRE(A.fin.sums && A.fin.nrep >= 1 && A.fin.nrep <= 16 && A.fin.count > 0.f, "conv_fwd: fin needs sums, 1..16 slices, count > 0");
    if (A.fin.mode == LMN_FIN_BN)
      LMN_REQUIRE(A.epilogue == LMN_EP_AFFINE_ACT && A.fin.gamma && A.fin.beta && (!A.fin.about || A.fin.about != A.fin.rmean),
                  "conv_fwd: LMN_FIN_BN belongs to EP_AFFINE_ACT, needs gamma / beta, and `about` must not alias rmean");
    else
      LMN_REQUIRE(A.epilogue == LMN_EP_BN_BWD2 && A.fin.Ain, "conv_fwd: LMN_FIN_BN_BWD belongs to EP_BN_BWD2 and needs Ain");
  }
  LMN_REQUIRE(!A.stats_snap || (A.stats && A.stats_mode == LMN_STATS_SUM_SQ), "conv_fwd: stats_snap belongs to a SUM_SQ statistics pass");
  if (g_lmn_prof_on) {  // algorithmic cost of this launch (each HBM tensor once, MACs of the layer shape)
    int64_t cin = 0;
    for (int s = 0; s < A.nsrc; ++s) cin += A.src[s].C;
    const double opix = (double)A.B * A.Hout * A.Wout, ipix = (double)A.B * A.Hin * A.Win;
    const double macs = (A.transposed ? ipix : opix) * (double)cin * A.Cout * A.ksize * A.ksize;
    double by = ipix * cin + (A.out ? opix * A.Cout : 0.0);
    if (A.aux) by += opix * A.Cout;
    if (A.residual) by += opix * A.Cout;
    lmn_prof_cost(2.0 * macs, (A.act_dtype == LMN_BF16 ? 2.0 : 4.0) * by);
  }
  P.NCTT = (A.Cout + 15) / 16;
  P.ncls = (A.transposed && A.stride == 2) ? 4 : 1;
  P.gpi = 0;
  for (int c = 0; c < 4; ++c) {
    P.ng_c[c] = 0;
    P.Hc[c] = P.Wc[c] = 1;
  }
  for (int c = 0; c < P.ncls; ++c) {
    if (P.ncls == 4) {
      P.Hc[c] = (A.Hout - (c >> 1) + 1) >> 1;
      P.Wc[c] = (A.Wout - (c & 1) + 1) >> 1;
    } else {
      P.Hc[c] = A.Hout;
      P.Wc[c] = A.Wout;
    }
    if (P.Wc[c] < 1) P.Wc[c] = 1;  // empty class (Hout or Wout == 1): keep divisors sane
    const int64_t px = (int64_t)((P.ncls == 4) ? ((A.Hout - (c >> 1) + 1) >> 1) * (int64_t)((A.Wout - (c & 1) + 1) >> 1) : (int64_t)A.Hout * A.Wout);
    P.ng_c[c] = (int)((px + 15) / 16);
    P.gpi += P.ng_c[c];
  }
  const int64_t total_groups = (int64_t)A.B * P.gpi;
  LMN_REQUIRE(total_groups < (1LL << 30), "conv_fwd: too many pixel groups");
  int nct = P.NCTT >= 6 ? 6 : (P.NCTT == 5 ? 6 : P.NCTT);
  const int chunks = (P.NCTT + nct - 1) / nct;
  hipStream_t st = (hipStream_t)stream;
  // data gradient of a stride-2 3x3 conv: four parity classes in grid.z of the tile kernel (S2T)
  const bool s2t = A.transposed && A.stride == 2 && A.ksize == 3 && !(A.epilogue == LMN_EP_SE_BWD);
  if (!(A.transposed && A.stride == 2) || s2t) {
    // ---- LDS-tiled kernel.  1x1: the image is a flat row of H*W pixels.
    ConvParams T = P;
    lmn_conv_args_t& a = T.a;
    if (a.ksize == 1) {
      a.Wout *= a.Hout; a.Win *= a.Hin; a.Hout = a.Hin = 1;
    }
    const int tpmax = (a.stride == 2 && !s2t) ? 64 : 128;
    const int gW = s2t ? (a.Wout + 1) / 2 : a.Wout, gH = s2t ? (a.Hout + 1) / 2 : a.Hout;  // tiled grid (S2T: class coordinates)
    int ncw = 0;  // > 0: M-split kernel with ncw cout tiles per wave
    // M-split from 6 cout tiles: 5 tiles split 2+1+1+1 over the four waves (24->72 3x3 at 176x176: 242 us, N-split 185 us)
    static int msplit_env = -1;   // LMN_MSPLIT_MIN: A/B runs
    if (msplit_env < 0) { const char* e = getenv("LMN_MSPLIT_MIN"); msplit_env = e ? atoi(e) : 6; }
    const int msplit_min = msplit_env;
    int tnct = nct, tchunks = chunks;  // N-split form: cout tiles per block (<= 3), cout chunks (grid.y)
    if (P.NCTT > 3) { tnct = 3; tchunks = (P.NCTT + 2) / 3; }
    if (a.ksize == 1) { T.TW = a.Wout < tpmax ? a.Wout : tpmax; T.TH = 1; }
    else {
      T.TW = gW <= 32 ? gW : 16;
      T.TH = tpmax / T.TW;
      if (T.TH > gH) T.TH = gH;
      if (T.TH < 1) T.TH = 1;
    }
    if (P.NCTT >= msplit_min && !s2t) {
      // Wide layer -> M-split kernel.  Wide layers sit on the small feature maps, where a 128-pixel tile times a
      // few cout chunks can leave most of the 256 CUs idle: pick (tile pixels, cout tiles per wave) by a cost model
      // -- rounds of 256 blocks x per-block MFMA work (+ staging, inflated by the halo for short 3x3 tiles).
      float best = 1e30f;
      int bTW = T.TW, bTH = T.TH;
      for (int cand = 0; cand < 16; ++cand) {
        int tw, th;
        if (a.ksize == 1) {
          static int tp_env = -1;   // LMN_CONVM_TP: force the 1x1 tile (pixels), A/B runs
          if (tp_env < 0) { const char* e = getenv("LMN_CONVM_TP"); tp_env = e ? atoi(e) : 0; }
          if (cand > 2) break;
          if (tp_env > 0 && (tpmax >> cand) != tp_env && tp_env <= a.Wout) continue;
          tw = tpmax >> cand; th = 1;
          if (tw > a.Wout) { if (cand) continue; tw = a.Wout; }
          if (cand && tw < 16) continue;
        } else {
          static int th_env = -1;   // LMN_CONVM_TH: force the 3x3 tile height (A/B runs)
          if (th_env < 0) { const char* e = getenv("LMN_CONVM_TH"); th_env = e ? atoi(e) : 0; }
          tw = T.TW; th = T.TH - cand;
          if (th < 1) break;
          if (th_env > 0 && th_env <= T.TH && th != th_env) continue;
        }
        const int ng = (tw * th + 15) / 16;
        const long tiles = (long)a.B * ((a.Wout + tw - 1) / tw) * ((a.Hout + th - 1) / th);
        for (int w = 1; w <= 2; ++w) {
          const long blk = tiles * ((P.NCTT + 4 * w - 1) / (4 * w));
          const float halo = a.ksize == 1 ? 1.f : (float)(th + 2) / th;
          const float cost = (float)((blk + 255) / 256) * (ng * (w + 0.3f * halo) + 1.f);
          if (cost < best * 0.999f) { best = cost; bTW = tw; bTH = th; ncw = w; }
        }
      }
      T.TW = bTW; T.TH = bTH;
      LMN_REQUIRE(ncw > 0, "conv_fwd: no tile candidate");
    }
    T.TP = T.TH * T.TW;
    T.NG = (T.TP + 15) / 16;
    LMN_REQUIRE(T.NG <= 8, "conv_fwd: tile of %d pixels", T.TP);
    const int st_in = a.transposed ? 1 : a.stride;
    T.XH = s2t ? T.TH + 1 : (T.TH - 1) * st_in + a.ksize;
    T.XW = s2t ? T.TW + 1 : (T.TW - 1) * st_in + a.ksize;
    int maxkb = 0;
    for (int s = 0; s < a.nsrc; ++s) maxkb = P.nkb[s] > maxkb ? P.nkb[s] : maxkb;
    T.CKB = maxkb < 2 ? maxkb : 2;
    if (ncw && a.ksize == 1) {  // M-split 1x1: K chunks of up to 8 K16 blocks while the window stays within ~36 KB of LDS
      static int ckb_cap = -1;
      if (ckb_cap < 0) { const char* e = getenv("LMN_CONVM_CKB"); ckb_cap = e ? atoi(e) : 8; }
      int c = maxkb < ckb_cap ? maxkb : ckb_cap;
      while (c > 2 && (size_t)T.TP * (c * 16 + 8) * sizeof(float) > 37 * 1024) --c;
      if (c > T.CKB) T.CKB = c;
    }
    // LDS pixel stride: conflict-free ds_read_b128 for 16 pixels st_in apart (brute-forced over the b128 lane groups):
    // +8 floats at unit stride, +4 at stride 2 (PMC: 0.5 conflict cycles per LDS cycle with +4 at unit stride)
    T.CS = T.CKB * 16 + (st_in == 1 ? 8 : 4);
    const bool bf = a.mma_dtype == LMN_BF16;
    const int pm = bf ? (a.act_dtype == LMN_BF16 ? 2 : 1) : 0;
    if (bf) T.CS = T.CKB * 8 + 4;  // dwords: 8 per K16 block of 4-bf16 fragments, +4: conflict-free ds_read_b64 (brute-forced)
    T.tiles_x = (gW + T.TW - 1) / T.TW;
    T.tiles_y = (gH + T.TH - 1) / T.TH;
    T.total_tiles = a.B * T.tiles_x * T.tiles_y;
    // strided tile assignment: -3 % over the conv family at batch 8 (contiguous ranges leave some blocks with twice the
    // tiles of others: 1936 tiles on 1280 blocks); SE_BWD keeps contiguous ranges (per-image sums live in registers)
    T.strided = a.epilogue == LMN_EP_SE_BWD ? 0 : 1;
    T.mTW = (uint32_t)((1ull << 32) / (uint32_t)T.TW + 1);
    T.mXW = (uint32_t)((1ull << 32) / (uint32_t)T.XW + 1);
    LMN_REQUIRE(T.XH * T.XW < 65536, "conv_fwd: window too large");
    size_t shmem = ((size_t)T.XH * T.XW * T.CS + (2 + 9) * tnct * 16) * sizeof(float);  // window, statistics, epilogue parameters
    // LDS-staged weights (3x3 tile kernel) while the block stays within LMN_CONV_WL_KB of LDS (default 32: 5 blocks per CU; level-0
    // 12 -> 12: 62 -> 52 us on cold operands; at 48 KB the 24-channel layers lose more to occupancy than they gain)
    static int wl_kb = -1;
    if (wl_kb < 0) { const char* e = getenv("LMN_CONV_WL_KB"); wl_kb = e ? atoi(e) : 32; }
    const size_t wl_bytes = (size_t)9 * T.CKB * tnct * (a.mma_dtype == LMN_BF16 ? 128 : 256) * sizeof(float);
    const bool wlk = a.ksize == 3 && !s2t && shmem + wl_bytes <= (size_t)wl_kb * 1024;
    if (wlk) shmem += wl_bytes;
    LMN_REQUIRE(shmem <= 64 * 1024, "conv_fwd: LDS window %zu B", shmem);
    int blocks = T.total_tiles;
    int maxb = 1280 / tchunks > 256 ? 1280 / tchunks : 256;  // ~5 resident blocks per CU: one round of persistent blocks
    if (s2t) maxb = 320 / tchunks > 64 ? 320 / tchunks : 64;          // x 4 classes in grid.z
    if (blocks > maxb) blocks = maxb;
    // epilogue instance (see the kernel): 0 plain, 2 LINEAR+SUM_SQ, 3 BN_BWD1, 4 BN_BWD2, 5 SE_BWD, 1 everything else
    int ek = 1;
    if (a.epilogue <= LMN_EP_AFFINE_ACT && a.stats_mode == LMN_STATS_NONE) ek = 0;
    else if (a.drop_p > 0.f) ek = 1;
    else if (a.epilogue == LMN_EP_LINEAR && a.stats_mode == LMN_STATS_SUM_SQ) ek = 2;
    else if (a.epilogue == LMN_EP_BN_BWD1 && a.stats_mode == LMN_STATS_EP) ek = 3;
    else if (a.epilogue == LMN_EP_BN_BWD2 && a.stats_mode == LMN_STATS_NONE) ek = 4;
    else if (a.epilogue == LMN_EP_SE_BWD && a.stats_mode == LMN_STATS_EP) ek = 5;
    // deterministic mode: the statistics go to one slot per block (SE_BWD: per wave) of the stream's scratch, summed in fixed order
    // by lmn_det_sum right after the launch into slice 0 of the caller's buffer
    int det_ns = 0;
    int64_t det_sz = 0;
    auto det_prep = [&](int nbx) -> int {
      if (!g_lmn_det || !a.stats || a.stats_mode == LMN_STATS_NONE) return 0;
      const bool sebwd = a.epilogue == LMN_EP_SE_BWD;
      det_ns = sebwd ? nbx * 4 : nbx;
      det_sz = sebwd ? (int64_t)a.B * a.Cout : 2 * (int64_t)a.Cout;
      lmn_det_begin(st);
      T.det_stats = lmn_det_slots(st, (size_t)det_ns * det_sz);
      LMN_REQUIRE(T.det_stats, "conv_fwd: deterministic mode: no scratch for %d x %lld statistics slots", det_ns, (long long)det_sz);
      return 0;
    };
    auto det_finish = [&]() { if (T.det_stats) lmn_det_sum(st, T.det_stats, det_ns, det_sz, a.stats); };
    if (ncw) {
      const int mchunks = (P.NCTT + 4 * ncw - 1) / (4 * ncw);
      int mblocks = T.total_tiles;
      const int mmax = 2048 / mchunks > 256 ? 2048 / mchunks : 256;
      if (mblocks > mmax) mblocks = mmax;
      const dim3 mgrid(mblocks, mchunks);
      if (int rc = det_prep(mblocks)) return rc;
      const size_t msh = ((size_t)T.XH * T.XW * T.CS + (2 + 9) * 4 * ncw * 16) * sizeof(float);
#define LMN_CM(TT, NN, BFV)                                                                              \
  do {                                                                                                   \
    switch ((TT) == 1 ? ek : (ek > 2 ? 1 : ek)) {                                                        \
      case 0: LMN_LAUNCH((conv_tileM_kernel<TT, NN, 0, BFV>), mgrid, dim3(256), msh, st, T); break;   \
      case 2: LMN_LAUNCH((conv_tileM_kernel<TT, NN, 2, BFV>), mgrid, dim3(256), msh, st, T); break;   \
      case 3: LMN_LAUNCH((conv_tileM_kernel<1, NN, 3, BFV>), mgrid, dim3(256), msh, st, T); break;    \
      case 4: LMN_LAUNCH((conv_tileM_kernel<1, NN, 4, BFV>), mgrid, dim3(256), msh, st, T); break;    \
      case 5: LMN_LAUNCH((conv_tileM_kernel<1, NN, 5, BFV>), mgrid, dim3(256), msh, st, T); break;    \
      default: LMN_LAUNCH((conv_tileM_kernel<TT, NN, 1, BFV>), mgrid, dim3(256), msh, st, T); break;  \
    }                                                                                                    \
  } while (0)
#define LMN_CMB(TT, NN) do { if (pm == 2) LMN_CM(TT, NN, 2); else if (pm == 1) LMN_CM(TT, NN, 1); else LMN_CM(TT, NN, 0); } while (0)
#define LMN_CMR(NN) do { if (pm == 2) LMN_CM(1, NN, 6); else if (pm == 1) LMN_CM(1, NN, 5); else LMN_CM(1, NN, 4); } while (0)
      if (a.ksize == 1 && T.rpw) { if (ncw == 2) LMN_CMR(2); else LMN_CMR(1); }   // (row-planar operands: their own instances)
      else if (a.ksize == 1) { if (ncw == 2) LMN_CMB(1, 2); else LMN_CMB(1, 1); }
      else { if (ncw == 2) LMN_CMB(9, 2); else LMN_CMB(9, 1); }
#undef LMN_CMR
#undef LMN_CMB
#undef LMN_CM
      det_finish();
      return lmn_launch_status("conv_fwd(tileM)");
    }
    if (s2t) {
      const dim3 zgrid(blocks, tchunks, 4);
      if (int rc = det_prep(blocks * 4)) return rc;
#define LMN_CZ(NN, BFV)                                                                                       \
  do {                                                                                                        \
    if (ek == 0) LMN_LAUNCH((conv_tile_kernel<9, NN, 0, true, BFV>), zgrid, dim3(256), shmem, st, T); \
    else LMN_LAUNCH((conv_tile_kernel<9, NN, 1, true, BFV>), zgrid, dim3(256), shmem, st, T);         \
  } while (0)
#define LMN_CZB(NN) do { if (pm == 2) LMN_CZ(NN, 2); else if (pm == 1) LMN_CZ(NN, 1); else LMN_CZ(NN, 0); } while (0)
      switch (tnct) {
        case 1: LMN_CZB(1); break;
        case 2: LMN_CZB(2); break;
        default: LMN_CZB(3); break;
      }
#undef LMN_CZB
#undef LMN_CZ
      det_finish();
      return lmn_launch_status("conv_fwd(tile, stride-2 data gradient)");
    }
    const dim3 grid(blocks, tchunks);
    if (int rc = det_prep(blocks)) return rc;
#define LMN_CT(TT, NN, BFV)                                                                              \
  do {                                                                                                   \
    switch ((TT) == 1 ? ek : (ek > 2 ? 1 : ek)) {                                                        \
      case 0: if ((TT) == 9 && wlk) LMN_LAUNCH((conv_tile_kernel<9, NN, 0, false, (BFV) & 3, true>), grid, dim3(256), shmem, st, T); \
              else LMN_LAUNCH((conv_tile_kernel<TT, NN, 0, false, BFV>), grid, dim3(256), shmem, st, T); break;   \
      case 2: if ((TT) == 9 && wlk) LMN_LAUNCH((conv_tile_kernel<9, NN, 2, false, (BFV) & 3, true>), grid, dim3(256), shmem, st, T); \
              else LMN_LAUNCH((conv_tile_kernel<TT, NN, 2, false, BFV>), grid, dim3(256), shmem, st, T); break;   \
      case 3: LMN_LAUNCH((conv_tile_kernel<1, NN, 3, false, BFV>), grid, dim3(256), shmem, st, T); break;    \
      case 4: LMN_LAUNCH((conv_tile_kernel<1, NN, 4, false, BFV>), grid, dim3(256), shmem, st, T); break;    \
      case 5: LMN_LAUNCH((conv_tile_kernel<1, NN, 5, false, BFV>), grid, dim3(256), shmem, st, T); break;    \
      default: if ((TT) == 9 && wlk) LMN_LAUNCH((conv_tile_kernel<9, NN, 1, false, (BFV) & 3, true>), grid, dim3(256), shmem, st, T); \
               else LMN_LAUNCH((conv_tile_kernel<TT, NN, 1, false, BFV>), grid, dim3(256), shmem, st, T); break;  \
    }                                                                                                    \
  } while (0)
#define LMN_CTN(TT)                                                                                      \
  switch ((tnct > 3 ? 3 : tnct) * 4 + pm) {                                                              \
    case 4: LMN_CT(TT, 1, 0); break;                                                                     \
    case 5: LMN_CT(TT, 1, 1); break;                                                                     \
    case 6: LMN_CT(TT, 1, 2); break;                                                                     \
    case 8: LMN_CT(TT, 2, 0); break;                                                                     \
    case 9: LMN_CT(TT, 2, 1); break;                                                                     \
    case 10: LMN_CT(TT, 2, 2); break;                                                                    \
    case 13: LMN_CT(TT, 3, 1); break;                                                                    \
    case 14: LMN_CT(TT, 3, 2); break;                                                                    \
    default: LMN_CT(TT, 3, 0); break;                                                                    \
  }
    if (a.ksize == 1 && T.rpw) {   // (row-planar operands: their own instances, precision mode | 4)
      switch ((tnct > 3 ? 3 : tnct) * 4 + pm) {
        case 4: LMN_CT(1, 1, 4); break;
        case 5: LMN_CT(1, 1, 5); break;
        case 6: LMN_CT(1, 1, 6); break;
        case 8: LMN_CT(1, 2, 4); break;
        case 9: LMN_CT(1, 2, 5); break;
        case 10: LMN_CT(1, 2, 6); break;
        case 13: LMN_CT(1, 3, 5); break;
        case 14: LMN_CT(1, 3, 6); break;
        default: LMN_CT(1, 3, 4); break;
      }
    } else if (a.ksize == 1) { LMN_CTN(1) } else { LMN_CTN(9) }
#undef LMN_CTN
#undef LMN_CT
    det_finish();
    return lmn_launch_status("conv_fwd(tile)");
  }
  LMN_REQUIRE(false, "conv_fwd: the data gradient of a stride-2 conv is implemented for 3x3 kernels (got ksize %d, epilogue %d)", A.ksize, A.epilogue);
  return -1;
}

static int reduce_slices(int nblk) {  // k-slices of the reduction kernel: ~4 partials per slice, at most 64
  int k = 1;
  while (k < 64 && k * 4 < nblk) k <<= 1;
  return k;
}

// K-split blocks either add their LDS-reduced tile straight into dW with atomics or write it to the workspace for a
// second (reduction) launch.  Scattered float atomics retire at ~40 per nanosecond chip-wide (measured: 4.6 M of
// them made a 114 us tail on the 192->96 3x3 layer), the reduction launch costs ~6-10 us: atomics only for small totals.
static bool wgrad_two_stage(int64_t gy, int64_t blocks, int64_t per) {
  return blocks > 48 || gy * blocks * per > 256 * 1024;
}

// Geometry of one weight-gradient call: tile shape, K-split block count, reduction form.  ONE source of truth for the launch
// (lmn_conv_wgrad) and for the description of its deferred reduction (lmn_conv_wgrad_job).
struct WgGeom {
  int NMT, NNT, gy, taps, blocks;   // cout x cin tiles per block, tile sets (grid.y), K-split blocks (grid.x)
  int64_t per;                      // floats of one block partial
  bool direct, v1, wave_staged;
  size_t shmem;
  int pm;
};

static int wgrad_setup(const lmn_wgrad_args_t& A, WgradParams& P, WgGeom& G) {
  LMN_REQUIRE(A.ksize == 1 || A.ksize == 3, "conv_wgrad: ksize %d", A.ksize);
  LMN_REQUIRE(A.stride == 1 || A.stride == 2, "conv_wgrad: stride %d", A.stride);
  LMN_REQUIRE(A.nsrc >= 1 && A.nsrc <= 3, "conv_wgrad: nsrc %d", A.nsrc);
  LMN_REQUIRE(A.dy && A.Cout > 0 && A.dy_cstride >= A.Cout, "conv_wgrad: dy/Cout");
  for (int s = 0; s < A.nsrc && s < 3; ++s) LMN_REQUIRE(A.dW || A.dW_src[s], "conv_wgrad: no gradient tensor for source %d", s);
  LMN_REQUIRE(!A.db2 || A.db, "conv_wgrad: db2 without db");
  LMN_REQUIRE(A.B > 0 && A.Hout > 0 && A.Wout > 0, "conv_wgrad: empty tensor");
  LMN_REQUIRE(A.mma_dtype == LMN_F32 || A.mma_dtype == LMN_BF16, "conv_wgrad: mma_dtype %d", A.mma_dtype);
  LMN_REQUIRE(A.act_dtype == LMN_F32 || (A.act_dtype == LMN_BF16 && A.mma_dtype == LMN_BF16),
              "conv_wgrad: act_dtype %d with mma_dtype %d", A.act_dtype, A.mma_dtype);
  P.a = A;
  P.NNTT = 0;
  P.Cin = 0;
  for (int s = 0; s < 3; ++s) {
    P.ntile_src[s] = P.ntile_off[s] = P.cbase[s] = 0;
    P.inv_keep_src[s] = 1.f;
  }
  for (int s = 0; s < A.nsrc; ++s) {
    LMN_REQUIRE(A.src[s].ptr && A.src[s].C > 0 && A.src[s].cstride >= A.src[s].C, "conv_wgrad: source %d", s);
    P.ntile_src[s] = (A.src[s].C + 15) / 16;
    P.ntile_off[s] = P.NNTT;
    P.cbase[s] = P.Cin;
    P.NNTT += P.ntile_src[s];
    P.Cin += A.src[s].C;
    if (A.src[s].flags & LMN_SRC_DROP) P.inv_keep_src[s] = 1.f / (1.f - A.src[s].drop_p);
  }
  P.inv_keep_dy = (A.dy_flags & LMN_SRC_DROP) ? 1.f / (1.f - A.dy_p) : 1.f;
  {  // row-planar operands (RP4): whole tensors of a 1x1 stride-1 call, one image width for all of them
    int rw = 0;
    for (int s = 0; s < 3; ++s) {
      const int w = s < A.nsrc ? A.src[s].rp_w : 0;
      P.lay_src[s] = lmn_lay_make(w, s < A.nsrc ? A.src[s].C : 4, s < A.nsrc ? A.src[s].cstride : 4);
      if (w) {
        LMN_REQUIRE(w > 0 && A.src[s].cstride == A.src[s].C && (rw == 0 || rw == w), "conv_wgrad: row-planar source %d must be a whole tensor (cstride == C) of the call's width", s);
        rw = w;
      }
    }
    P.lay_dy = lmn_lay_make(A.dy_rp_w, A.Cout, A.dy_cstride);
    if (A.dy_rp_w) { LMN_REQUIRE(A.dy_rp_w > 0 && A.dy_cstride == A.Cout && (rw == 0 || rw == A.dy_rp_w), "conv_wgrad: row-planar dy must be a whole tensor of the call's width"); rw = A.dy_rp_w; }
    if (rw) LMN_REQUIRE(A.ksize == 1 && A.stride == 1 && ((int64_t)A.Hout * A.Wout) % rw == 0, "conv_wgrad: row-planar operands belong to 1x1 stride-1 calls");
    P.rpw = rw;
    P.rp_magic = lmn_div_magic(rw);
  }
  P.NMTT = (A.Cout + 15) / 16;
  P.steps_per_img = (A.Hout * A.Wout + 3) / 4;
  LMN_REQUIRE((int64_t)A.B * P.steps_per_img < (1LL << 31), "conv_wgrad: too many pixels");
  for (int s = 0; s < A.nsrc; ++s)
    LMN_REQUIRE((int64_t)A.B * A.Hin * A.Win * A.src[s].cstride < (1LL << 31), "conv_wgrad: source %d larger than 2^31 elements", s);
  LMN_REQUIRE((int64_t)A.B * A.Hout * A.Wout * A.dy_cstride < (1LL << 31), "conv_wgrad: dy larger than 2^31 elements");
  P.total_steps = A.B * P.steps_per_img;
  if (A.ksize == 1 && A.stride == 1) {  // 1x1: every image is one flat row of H*W pixels (all tiles full)
    P.a.Wout = A.Hout * A.Wout; P.a.Hout = 1;
    P.a.Win = A.Hin * A.Win; P.a.Hin = 1;
  }
  int NMT, NNT;  // cout x cin tiles per block
  wgrad_tile_shape(A, P.NMTT, P.NNTT, &NMT, &NNT);
  const int msets = (P.NMTT + NMT - 1) / NMT;
  P.nsets_n = (P.NNTT + NNT - 1) / NNT;
  const int gy = msets * P.nsets_n;
  const int taps = A.ksize * A.ksize;
  const int64_t per = (int64_t)taps * NMT * NNT * 256 + NMT * 16;
  // ---- tile geometry of the LDS-staged kernel
  const lmn_wgrad_args_t& Gm = P.a;  // (flattened) geometry
  // tile pixels: 256 for the 1x1 form, for one-tile blocks and on the small maps (fewer barriers and less halo per pixel:
  // -5..10 %), 128 where a 2x2-tile block would need 76 KB of LDS for it (+3 % there)
  const bool v1c = wgrad_v1(A) && Gm.Wout >= 32 && (NMT * NNT == 4 || NMT * NNT == 1);  // V1 candidate (item counts checked below)
  const int npmax = Gm.stride == 2 ? 64 : ((Gm.ksize == 1 || NMT * NNT == 1 || (Gm.Wout <= 32 && !v1c)) ? 256 : 128);
  P.TW = Gm.Wout < (Gm.ksize == 1 ? npmax : 32) ? Gm.Wout : (Gm.ksize == 1 ? npmax : 32);
  if (Gm.stride == 2 && P.TW > 16) P.TW = 16;
  P.TH = npmax / P.TW;
  if (P.TH > Gm.Hout) P.TH = Gm.Hout;
  if (P.TH < 1) P.TH = 1;
  P.XH = (P.TH - 1) * Gm.stride + Gm.ksize;
  P.XW = (P.TW - 1) * Gm.stride + Gm.ksize;
  // LDS planes [tile][pixel][CS]: the 4 pixels of a K step are CS*stride floats apart; 16 (stride 1) and 24
  // (stride 2: 48 = 16 mod 32 banks) keep the two 16-lane halves of a ds_read_b32 group on disjoint banks
  P.CSy = 16;
  P.CSx = A.stride == 1 ? 16 : 24;
  const bool bf = A.mma_dtype == LMN_BF16;
  const int pm = bf ? (A.act_dtype == LMN_BF16 ? 2 : 1) : 0;
  // bf16 planes: 16 bf16 = 8 dwords per pixel, +4: the lane groups q (pixels 4 apart) read disjoint bank ranges
  if (bf) P.CSy = P.CSx = 12;
  P.tiles_x = (Gm.Wout + P.TW - 1) / P.TW;
  P.tiles_y = (Gm.Hout + P.TH - 1) / P.TH;
  P.total_tiles = A.B * P.tiles_x * P.tiles_y;
  P.dbg = 0;
  LMN_REQUIRE(P.XH * P.XW < 65536 && P.TH * P.TW < 65536, "conv_wgrad: tile too large");
  P.mXW = (uint32_t)((1ull << 32) / (uint32_t)P.XW + 1);
  P.mTW = (uint32_t)((1ull << 32) / (uint32_t)P.TW + 1);
  int64_t lds_floats = (int64_t)NNT * P.XH * P.XW * P.CSx + (int64_t)NMT * P.TH * P.TW * P.CSy;
  if (lds_floats < per) lds_floats = per;
  LMN_REQUIRE(lds_floats * 4 <= 160 * 1024, "conv_wgrad: LDS tile too large (%lld B)", (long long)lds_floats * 4);
  // K-split: enough blocks to fill the chip (~1024 in total), each walking a contiguous range of tiles
  int64_t blocks64 = P.total_tiles;
  // wgrad3_kernel: the whole tile's items in the prefetch registers (6 + 4 per thread for one-tile, 4 + 2 for 2 x 2-tile
  // blocks), byte offsets in 32 bits (raw buffer descriptors)
  const int64_t esz = A.act_dtype == LMN_BF16 ? 2 : 4;
  bool v1 = v1c && P.TW == 32 && P.XW == 34 &&
            P.XH * P.XW * 4 <= (NMT * NNT == 1 ? 6 : 4) * 256 && P.TH * P.TW * 4 <= (NMT * NNT == 1 ? 4 : 2) * 256 &&
            (int64_t)A.B * A.Hout * A.Wout * A.dy_cstride * esz < 0xfffffff0LL;
  for (int s = 0; s < A.nsrc; ++s) v1 = v1 && (int64_t)A.B * A.Hin * A.Win * A.src[s].cstride * esz < 0xfffffff0LL;
  if (v1) {  // padded planes: [tile][UX * 64 pixels][CS dwords]
    const int ux = NMT * NNT == 1 ? 6 : 4, uy = NMT * NNT == 1 ? 4 : 2, cs = bf ? 12 : 16;
    lds_floats = (int64_t)(NNT * ux + NMT * uy) * 64 * cs;
    if (NMT * NNT == 1 && lds_floats < per) lds_floats = per;
  }
  static int cap_other = -1;   // LMN_WGRAD_CAP_OTHER: K-split blocks of the other weight-gradient kernels (1x1, LDS-staged 3x3), A/B runs
  if (cap_other < 0) { const char* e = getenv("LMN_WGRAD_CAP_OTHER"); cap_other = e && atoi(e) > 0 ? atoi(e) : 512; }
  const int btot = v1 ? wgrad_blocks_total(A) : cap_other;
  const int64_t cap = btot / gy > 2 ? btot / gy : 2;
  if (blocks64 > cap) blocks64 = cap;
  if (blocks64 < 1) blocks64 = 1;
  bool rp_any = A.dy_rp_w != 0;
  for (int s = 0; s < A.nsrc; ++s) rp_any = rp_any || A.src[s].rp_w != 0;
  G.direct = A.ksize == 1 && A.stride == 1 && ((int64_t)Gm.Hout * Gm.Wout >= 32 || rp_any);
  if (G.direct) {  // direct (no LDS) kernels: K steps split over ~1024*4/gy waves, at least 16 steps per wave
    const int64_t steps = ((int64_t)A.B * Gm.Hout * Gm.Wout + 3) / 4;
    blocks64 = (steps + 63) / 64;
    if (blocks64 > cap) blocks64 = cap;
    if (blocks64 < 1) blocks64 = 1;
  }
  // two-stage reduction when the caller's workspace holds every block partial; else LDS-reduced atomics with fewer blocks
  P.partial = nullptr;
  if (A.workspace && (wgrad_two_stage(gy, blocks64, per) || (g_lmn_det && blocks64 > 1)) && (int64_t)gy * blocks64 * per <= A.workspace_floats) {
    P.partial = A.workspace;
  } else if (g_lmn_det) {
    blocks64 = 1;         // deterministic mode without room for the partials: no K split (one block per tile set adds its sums alone)
  } else if (blocks64 > 512 / gy && 512 / gy >= 2) {
    blocks64 = 512 / gy;  // atomics: fewer blocks
  }
  G.NMT = NMT; G.NNT = NNT; G.gy = gy; G.taps = taps; G.per = per; G.blocks = (int)blocks64;
  G.v1 = v1; G.pm = pm; G.shmem = (size_t)lds_floats * 4;
  // 1x1 data path: wave-staged chunks (wgrad_1x1w_kernel) except for bf16-stored operands without on-load transforms in
  // one of the direct kernel's shapes, where one 2-byte load per lane and MFMA operand is faster (measured, level 0:
  // 28 vs 40 us; with transforms 70 vs 63 us; fp32 storage: wave-staged -19 % over the 13 probe layers)
  bool any_tf = (A.dy_flags & LMN_SRC_DROP) != 0;
  for (int s = 0; s < A.nsrc; ++s) any_tf = any_tf || A.src[s].flags != 0 || A.src[s].scale != nullptr;
  const bool old_shape = (NMT == 1 && NNT <= 4) || (NNT == 1 && NMT <= 4) || (NMT == 2 && NNT == 2);
  G.wave_staged = !(pm == 2 && !any_tf && old_shape) || (rp_any && (int64_t)Gm.Hout * Gm.Wout < 32);   // (tiny maps: the staged kernel divides per pixel)
  return 0;
}

int lmn_sizeof_reduce_job(void) { return (int)sizeof(lmn_reduce_job_t); }

// Description of the reduction a deferred weight-gradient call (args->defer_reduce) leaves behind: `out->nblk == 0` when the call
// reduces on its own (atomics: small totals, or no workspace).  Pure host arithmetic, nothing is launched.
int lmn_conv_wgrad_job(const lmn_wgrad_args_t* args, lmn_reduce_job_t* out) {
  LMN_REQUIRE(args && out, "conv_wgrad_job: null pointer");
  WgradParams P;
  WgGeom G;
  const int rc = wgrad_setup(*args, P, G);
  if (rc) return rc;
  memset(out, 0, sizeof(*out));
  if (!P.partial) return 0;
  const lmn_wgrad_args_t& A = *args;
  out->partial = P.partial;
  out->nblk = G.blocks; out->per = (int32_t)G.per; out->gy = G.gy; out->nsets_n = P.nsets_n;
  out->taps = G.taps; out->NMT = G.NMT; out->NNT = G.NNT; out->nsrc = A.nsrc;
  out->Cout = A.Cout; out->Cin = P.Cin; out->NMTT = P.NMTT; out->NNTT = P.NNTT;
  for (int s = 0; s < 3; ++s) {
    out->srcC[s] = s < A.nsrc ? A.src[s].C : 0;
    out->ntile_off[s] = P.ntile_off[s];
    out->cbase[s] = P.cbase[s];
    out->dW_src[s] = A.dW_src[s];
  }
  out->dW = A.dW; out->db = A.db; out->db2 = A.db2;
  out->ksl = reduce_slices(G.blocks);
  out->blocks_per_set = (int32_t)((G.per + 4096 / out->ksl - 1) / (4096 / out->ksl));
  return 0;
}

int lmn_reparam_fold(const float* hstats, const float* mean, const float* rstd, const float* A, float count, int batch_stats,
                     const float* w_expand, const float* b_expand, const float* w_shortcut, int E, int rows, int cin_w, int cred,
                     int cout_w, float* wpack, float* kbias, float* coef, float* dgamma, float* dbeta, int dtype,
                     lmn_stream_t stream) {
  LMN_REC(lmn_reparam_fold(hstats, mean, rstd, A, count, batch_stats, w_expand, b_expand, w_shortcut, E, rows, cin_w, cred, cout_w,
                           wpack, kbias, coef, dgamma, dbeta, dtype, stream));
  LMN_REQUIRE(hstats && mean && rstd && A && w_expand && b_expand && w_shortcut && wpack && kbias && coef && count > 0.f,
              "reparam_fold: null pointer");
  LMN_REQUIRE(E > 0 && rows > 0 && rows % 4 == 0 && cin_w > 0 && cin_w <= rows && cout_w > 0 && cout_w <= cred && cred % 4 == 0,
              "reparam_fold: E=%d rows=%d cin=%d cred=%d cout=%d", E, rows, cin_w, cred, cout_w);
  LMN_REQUIRE(dtype == LMN_F32 || dtype == LMN_BF16, "reparam_fold: dtype %d", dtype);
  FoldParams P;
  P.hstats = hstats; P.mean = mean; P.rstd = rstd; P.A = A; P.we = w_expand; P.be = b_expand; P.wsc = w_shortcut;
  P.wpack = wpack; P.kbias = kbias; P.coef = coef; P.dgamma = dgamma; P.dbeta = dbeta; P.count = count; P.batch_stats = batch_stats;
  P.E = E; P.rows = rows; P.cinw = cin_w; P.cred = cred; P.coutw = cout_w; P.bf16 = dtype == LMN_BF16;
  const int ntiles = ((E + 15) / 16 + (rows + 15) / 16 + (cred + 15) / 16) * ((rows + 15) / 16);
  LMN_REQUIRE((size_t)35 * E * sizeof(float) <= 64 * 1024, "reparam_fold: E = %d too wide for the block's scratch", E);
  LMN_LAUNCH(reparam_fold_kernel, dim3(ntiles + 1), dim3(256), (size_t)35 * E * sizeof(float), (hipStream_t)stream, P);
  return lmn_launch_status("reparam_fold");
}

int lmn_reparam_wfin(const float* R, const float* M, const float* m, const float* coef, const float* hstats, const float* w_expand,
                     const float* b_expand, float count, int E, int rows, int cin_w, float* dW, float* db, lmn_stream_t stream) {
  LMN_REC(lmn_reparam_wfin(R, M, m, coef, hstats, w_expand, b_expand, count, E, rows, cin_w, dW, db, stream));
  LMN_REQUIRE(R && M && m && coef && hstats && w_expand && b_expand && dW && E > 0 && rows >= cin_w && cin_w > 0 && count > 0.f,
              "reparam_wfin: bad argument");
  LMN_LAUNCH(reparam_wfin_kernel, dim3(lmn_cdiv((int64_t)E * (cin_w + 1), 16)), dim3(256), 0, (hipStream_t)stream, R, M, m, coef,
             hstats, w_expand, b_expand, count, E, rows, cin_w, dW, db);
  return lmn_launch_status("reparam_wfin");
}

int lmn_wgrad_reduce_batch(const lmn_reduce_job_t* jobs_dev, int njobs, int64_t total_blocks, lmn_stream_t stream) {
  LMN_REC(lmn_wgrad_reduce_batch(jobs_dev, njobs, total_blocks, stream));
  LMN_REQUIRE(jobs_dev && njobs > 0 && total_blocks > 0 && total_blocks < (1LL << 31), "wgrad_reduce_batch: bad job table");
  LMN_LAUNCH(wgrad_reduce_batch_kernel, dim3((unsigned)total_blocks), dim3(1024), 0, (hipStream_t)stream, jobs_dev, njobs);
  return lmn_launch_status("wgrad_reduce_batch");
}

int lmn_conv_wgrad(const lmn_wgrad_args_t* args, lmn_stream_t stream) {
  if (args && g_lmn_rec) {
    const lmn_wgrad_args_t copy = *args;
    lmn_rec_push([copy, stream]() -> int { return lmn_conv_wgrad(&copy, stream); }, "lmn_conv_wgrad(");
  }
  LMN_REQUIRE(args, "conv_wgrad: null args");
  const lmn_wgrad_args_t& A = *args;
  WgradParams P;
  WgGeom G;
  {
    const int rc = wgrad_setup(A, P, G);
    if (rc) return rc;
  }
  if (g_lmn_prof_on) {  // algorithmic cost: MACs of the layer shape; every source and dy read once
    const double opix = (double)A.B * A.Hout * A.Wout, ipix = (double)A.B * A.Hin * A.Win;
    lmn_prof_cost(2.0 * opix * (double)P.Cin * A.Cout * A.ksize * A.ksize, (A.act_dtype == LMN_BF16 ? 2.0 : 4.0) * (ipix * P.Cin + opix * A.Cout));
  }
  const int NMT = G.NMT, NNT = G.NNT, gy = G.gy, pm = G.pm, blocks = G.blocks;
  const int64_t per = G.per;
  const bool v1 = G.v1;
  const bool reduce_now = P.partial && !A.defer_reduce;   // deferred: the caller batches the reductions (lmn_wgrad_reduce_batch)
  const size_t shmem = G.shmem;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(blocks, gy);
#define LMN_WG(T, M, N)                                                                                             \
  do {                                                                                                              \
    if (shmem > 64 * 1024) {                                                                                        \
      (void)hipFuncSetAttribute((const void*)wgrad_lds_kernel<T, M, N, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem); \
      (void)hipFuncSetAttribute((const void*)wgrad_lds_kernel<T, M, N, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem); \
      (void)hipFuncSetAttribute((const void*)wgrad_lds_kernel<T, M, N, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem); \
    }                                                                                                               \
    if constexpr (T == 9 && (M * N == 4 || M * N == 1)) {                                                           \
      if (v1) {                                                                                                     \
        if (pm == 2) wgrad3_launch<M, N, 2>(grid, shmem, st, P);                                                    \
        else if (pm == 1) wgrad3_launch<M, N, 1>(grid, shmem, st, P);                                               \
        else wgrad3_launch<M, N, 0>(grid, shmem, st, P);                                                            \
        goto wg_reduce_##T##M##N;                                                                                   \
      }                                                                                                             \
    }                                                                                                               \
    if (pm == 2) LMN_LAUNCH((wgrad_lds_kernel<T, M, N, 2>), grid, dim3(256), shmem, st, P);                 \
    else if (pm == 1) LMN_LAUNCH((wgrad_lds_kernel<T, M, N, 1>), grid, dim3(256), shmem, st, P);            \
    else LMN_LAUNCH((wgrad_lds_kernel<T, M, N, 0>), grid, dim3(256), shmem, st, P);                         \
  wg_reduce_##T##M##N:                                                                                              \
    if (reduce_now) {                                                                                               \
      const int ksl = reduce_slices(blocks);                                                                        \
      const int rb = (int)((per + 4096 / ksl - 1) / (4096 / ksl));                                                    \
      LMN_LAUNCH((wgrad_reduce_kernel<T, M, N>), dim3(rb, gy), dim3(1024), 0, st, P, blocks, ksl);          \
    }                                                                                                               \
  } while (0)
  if (G.direct) {
    const int64_t nb = blocks;
    const dim3 dgrid((unsigned)nb, gy);
    const bool wave_staged = G.wave_staged;
#define LMN_WDK(KERN, M, N, SH)                                                                                    \
  do {                                                                                                              \
    if (P.rpw) {   /* row-planar operands: their own instances (precision mode | 4) */                              \
      if (pm == 2) LMN_LAUNCH((KERN<M, N, 6>), dgrid, dim3(256), SH, st, P);                                        \
      else if (pm == 1) LMN_LAUNCH((KERN<M, N, 5>), dgrid, dim3(256), SH, st, P);                                   \
      else LMN_LAUNCH((KERN<M, N, 4>), dgrid, dim3(256), SH, st, P);                                                \
    } else if (pm == 2) LMN_LAUNCH((KERN<M, N, 2>), dgrid, dim3(256), SH, st, P);                                   \
    else if (pm == 1) LMN_LAUNCH((KERN<M, N, 1>), dgrid, dim3(256), SH, st, P);                                     \
    else LMN_LAUNCH((KERN<M, N, 0>), dgrid, dim3(256), SH, st, P);                                                  \
  } while (0)
#define LMN_WD(M, N)                                                                                               \
  do {                                                                                                              \
    if (wave_staged) {                                                                                              \
      const int64_t stf = 4 * (M + N) * 256, rdf = (int64_t)M * N * 256 + M * 16;                                   \
      const size_t wsh = (size_t)(stf > rdf ? stf : rdf) * 4;                                                       \
      LMN_WDK(wgrad_1x1w_kernel, M, N, wsh);                                                                        \
    } else LMN_WDK(wgrad_1x1_kernel, M, N, 0);                                                                      \
    if (reduce_now) {                                                                                               \
      const int ksl = reduce_slices((int)nb);                                                                       \
      const int rb = (int)((per + 4096 / ksl - 1) / (4096 / ksl));                                                    \
      LMN_LAUNCH((wgrad_reduce_kernel<1, M, N>), dim3(rb, gy), dim3(1024), 0, st, P, (int)nb, ksl);         \
    }                                                                                                               \
  } while (0)
#define LMN_WW(M, N)                                                                                               \
  do {                                                                                                              \
    const int64_t stf = 4 * (M + N) * 256, rdf = (int64_t)M * N * 256 + M * 16;                                     \
    const size_t wsh = (size_t)(stf > rdf ? stf : rdf) * 4;                                                         \
    LMN_WDK(wgrad_1x1w_kernel, M, N, wsh);                                                                          \
    if (reduce_now) {                                                                                               \
      const int ksl = reduce_slices((int)nb);                                                                       \
      const int rb = (int)((per + 4096 / ksl - 1) / (4096 / ksl));                                                    \
      LMN_LAUNCH((wgrad_reduce_kernel<1, M, N>), dim3(rb, gy), dim3(1024), 0, st, P, (int)nb, ksl);         \
    }                                                                                                               \
  } while (0)
    switch (NMT * 8 + NNT) {
      case 1 * 8 + 1: LMN_WD(1, 1); break;
      case 1 * 8 + 2: LMN_WD(1, 2); break;
      case 1 * 8 + 3: LMN_WD(1, 3); break;
      case 1 * 8 + 4: LMN_WD(1, 4); break;
      case 2 * 8 + 1: LMN_WD(2, 1); break;
      case 3 * 8 + 1: LMN_WD(3, 1); break;
      case 4 * 8 + 1: LMN_WD(4, 1); break;
      case 2 * 8 + 2: LMN_WD(2, 2); break;
      case 2 * 8 + 3: LMN_WW(2, 3); break;
      case 3 * 8 + 2: LMN_WW(3, 2); break;
      case 2 * 8 + 4: LMN_WW(2, 4); break;
      default: LMN_WW(4, 2); break;
    }
#undef LMN_WW
#undef LMN_WD
#undef LMN_WDK
    return lmn_launch_status("conv_wgrad(1x1)");
  }
#define LMN_WGS(T)                                                   \
  do {                                                               \
    if (NMT == 1 && NNT == 1) LMN_WG(T, 1, 1);                       \
    else if (NMT == 1) LMN_WG(T, 1, 2);                              \
    else if (NNT == 1) LMN_WG(T, 2, 1);                              \
    else LMN_WG(T, 2, 2);                                            \
  } while (0)
  if (A.ksize == 1) LMN_WGS(1); else LMN_WGS(9);
#undef LMN_WGS
#undef LMN_WG
  return lmn_launch_status("conv_wgrad");
}

}  // extern "C"
