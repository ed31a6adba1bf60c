// conv_tile_kernel: the LDS-tiled forward / data-gradient kernel of the dense-convolution family (N-split form).
#pragma once
#include "conv_common.h"

namespace {

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

#define LMN_SLOT c
// WL (3x3): the packed weight fragments of a chunk (9 taps x <= 2 K16 blocks x NCT cout tiles) are staged in LDS -- once per
// block when the layer is a single chunk -- and the MFMA loop reads its A operands with ds_read_b128.  Fetched from L2 one
// step ahead, a fragment had 6-16 MFMAs (200-500 cycles) to arrive in: every step of the small-channel layers stalled on it
// (phase clocks: the MFMA loop was 52-61 % of a block's life at 35 % MFMA-pipe use).
// LN (1x1, one source, plain epilogue): LMN_SRC_LN -- the source is layer-normalised over its channels where it is staged.  The quads
// of a pixel sit in ADJACENT lanes of the staging loop (item = pixel * per_px + quad, per_px = 4 or 8, the source is a single chunk
// of <= 32 channels: the host sends wider sources to the M-split kernel), so mean and variance are two DPP reductions over 4 / 8
// lanes on the loaded registers (two-pass form: deviations from the mean) -- no extra memory access, no barrier.  The blocks of
// cout chunk 0 store (mean, rstd) per pixel for the weight gradient.
// UP (3x3 stride-1 forward, one source): LMN_SRC_UP2 -- the source tensor is the HALF-resolution map and the window is sampled from its
// bilinear x2 upsampling (align_corners=True, lmn_up_coord: the arithmetic of lmn_up2_fwd) where it is staged: four float4 loads per
// (window pixel, channel quad) from the quarter-size tensor (neighbouring window pixels share their taps: L1 / L2 hits) instead of
// one from a materialised `up` tensor that a separate kernel wrote.  Window pixels outside the UPSAMPLED image are the conv's zero
// padding.  Two items per thread and round (8 loads in flight).
// NPG: pixel groups (16 pixels) per wave.  2 = tiles of up to 128 pixels; 4 = up to 256 (3x3 only).  A tile costs a block the same
// ~11 K cycles of window round trips, barriers and LDS latencies whatever its size (tile-height sweep, DESIGN 5h: 8 x 16 pixels 53 us,
// 4 x 16 pixels 95 us for the same layer), so the larger tile halves that cost per pixel and doubles the MFMAs per LDS round trip.
template <int TAPS, int NCT, int EPI, bool S2T = false, int PM = 0, bool WL = false, bool LN = false, bool UP = false, int NPG = 2>
__global__ __launch_bounds__(256) void conv_tile_kernel(const ConvParams P) {
  static_assert(NPG == 2 || (NPG == 4 && TAPS == 9 && !S2T), "four pixel groups per wave: 3x3 stride-1 windows");
  static_assert(!UP || (TAPS == 9 && !S2T && (PM & 4) == 0), "bilinear x2 on load: NHWC 3x3 stride-1 forward calls");
  static_assert(!WL || (TAPS == 9 && !S2T), "LDS-staged weights: 3x3 stride-1 windows");
  static_assert(!LN || (TAPS == 1 && EPI == 0 && (PM & 4) == 0), "LayerNorm on load: NHWC 1x1 calls with the plain epilogue");
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
  if (P.prio > 0 && P.prio < 100) lmn_wave_prio(P.prio);   // (uniform)
  const uint32_t soff = A.seed_ctr ? *A.seed_ctr : 0u;  // device-side dropout stream offset (graph replays: one bump per step)
  // EPI: 0 plain (LINEAR / AFFINE_ACT, no statistics), 1 generic, 2 LINEAR + SUM_SQ statistics, 3 BN_BWD1, 4 BN_BWD2,
  // 5 SE_BWD, 6 LN_BWD (1x1 NHWC only).  For EPI >= 2 the epilogue kind is a compile-time constant: each instance carries only its own code
  // (the generic instance keeps every variant resident: 125-160 VGPRs + spills, and measured 20-40 us over its
  // memory time at level 0).
  const int ep_kind = EPI == 2 ? LMN_EP_LINEAR : EPI == 3 ? LMN_EP_BN_BWD1 : EPI == 4 ? LMN_EP_BN_BWD2 : EPI == 5 ? LMN_EP_SE_BWD : EPI == 6 ? LMN_EP_LN_BWD : A.epilogue;
  const int st_mode = EPI == 2 ? LMN_STATS_SUM_SQ : (EPI == 3 || EPI == 5 || EPI == 6) ? LMN_STATS_EP : EPI == 4 ? LMN_STATS_NONE : A.stats_mode;
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
  if (P.prio >= 100) lmn_wave_stagger(P.prio - 100);   // (uniform)
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
  // LN: a thread stages the same channel quad of every item (item index = tid mod per_px): its gamma / beta once per kernel
  f32x4 ln_g = f32x4{0.f, 0.f, 0.f, 0.f}, ln_b = ln_g;
  float ln_invC = 0.f;
  if constexpr (LN) {
    const lmn_src_t& S0 = A.src[0];
    const int ppx = P.nkb[0] >= 2 ? 8 : 4;
    const int ch = (tid & (ppx - 1)) * 4;
    if (ch < S0.C) { ln_g = ld4(S0.ln_gamma + ch); ln_b = ld4(S0.ln_beta + ch); }
    ln_invC = 1.f / (float)S0.C;
  }

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

    // this wave's NPG pixel groups (wv, wv + 4, ...): tile pixel -> (row, col), LDS base address, validity
    int pbase[NPG], opix[NPG];
    bool pvalid[NPG];
#pragma unroll
    for (int g = 0; g < NPG; ++g) {
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
    bool gl[NPG];  // wave-uniform: group g of this wave exists in the tile
#pragma unroll
    for (int g = 0; g < NPG; ++g) gl[g] = (wv + 4 * g) < P.NG;

    f32x4 acc[NPG][NCT];  // start from bias (+ bias2): 4 channels q*4.. of cout tile c, the same for every pixel group
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
      for (int g = 0; g < NPG; ++g) acc[g][c] = *reinterpret_cast<const f32x4*>(s_par + c * 16 + q * 4);
    // BN_BWD instances: the epilogue's second operand (aux: an output-shaped tensor) is requested HERE, ahead of the staging
    // barrier -- its HBM latency runs beside that of the window instead of after the MFMAs (two exposed round trips per
    // tile -> one; level 0, cold operands: BN_BWD1 102 -> 88 us, BN_BWD2 122 -> 105 us; SE_BWD measured slower with it)
    constexpr bool AUXP = (EPI == 3 && NCT <= 2) || EPI == 4;  // (three cout tiles: the 12 prefetch registers cost BN_BWD1 a wave per SIMD -- 137 -> 122 VGPRs, 38.9 -> 32.4 us at level 1)
    f32x4 axp[AUXP ? NPG : 1][AUXP ? NCT : 1];
    if constexpr (AUXP) {
#pragma unroll
      for (int g = 0; g < NPG; ++g) {
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
        if constexpr (UP) {
          const int hs = A.Hin >> 1, ws = A.Win >> 1;   // the source map (A.Hin x A.Win is the upsampled image the conv sees)
          const float sh = (float)(hs - 1) / (float)(A.Hin - 1), sw = (float)(ws - 1) / (float)(A.Win - 1);
          constexpr int SUU = 2;
          for (int i0 = 0; i0 < nitems; i0 += SUU * 256) {
            f32x4 t4[SUU][4];
            bool tok[SUU];
#pragma unroll
            for (int u = 0; u < SUU; ++u) {
              const int i = i0 + u * 256 + tid;
              const int f = i & (per_px - 1), pix = i >> psh;
              const int r = (int)__umulhi((uint32_t)pix, P.mXW), c = pix - r * P.XW;
              const int iy = wy0 + r, ix = wx0 + c;
              const int ch = kb0 * 16 + f * 4;
              const bool ok = i < nitems && ch < S.C && (unsigned)iy < (unsigned)A.Hin && (unsigned)ix < (unsigned)A.Win;
              tok[u] = ok;
              int y0, yp, x0, xp;
              float l0, l1;
              lmn_up_coord(ok ? iy : 0, hs, sh, y0, yp, l0, l1);
              lmn_up_coord(ok ? ix : 0, ws, sw, x0, xp, l0, l1);
              const TA* p00 = (const TA*)S.ptr + (uint32_t)(((b * hs + y0) * ws + x0) * S.cstride + (ok ? ch : 0));
              const TA* p10 = p00 + yp * ws * S.cstride;
              t4[u][0] = ld4(p00); t4[u][1] = ld4(p00 + xp * S.cstride);
              t4[u][2] = ld4(p10); t4[u][3] = ld4(p10 + xp * S.cstride);
            }
#pragma unroll
            for (int u = 0; u < SUU; ++u) {
              const int i = i0 + u * 256 + tid;
              if (i >= nitems) continue;
              const int f = i & (per_px - 1), pix = i >> psh;
              const int r = (int)__umulhi((uint32_t)pix, P.mXW), c = pix - r * P.XW;
              int y0, yp, x0, xp;
              float ly0, ly1, lx0, lx1;
              lmn_up_coord(tok[u] ? wy0 + r : 0, hs, sh, y0, yp, ly0, ly1);
              lmn_up_coord(tok[u] ? wx0 + c : 0, ws, sw, x0, xp, lx0, lx1);
              f32x4 v = ly0 * (lx0 * t4[u][0] + lx1 * t4[u][1]) + ly1 * (lx0 * t4[u][2] + lx1 * t4[u][3]);   // (the expression of up2_fwd_kernel)
              if (!tok[u]) v = f32x4{0.f, 0.f, 0.f, 0.f};
              if constexpr (BF) {
                *reinterpret_cast<uint2*>(&XS[pix * P.CS + f * 2]) = pk4_bf16(v);
              } else {
                float* d = &XS[pix * P.CS + (f >> 2) * 16 + (f & 3)];
                d[0] = v[0]; d[4] = v[1]; d[8] = v[2]; d[12] = v[3];
              }
            }
          }
        } else {
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
            if constexpr (LN) {
              // (items past the window end in whole pixels: the lanes of a pixel are all active here or all skipped above)
              auto px_sum = [&](float t) -> float {
                t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
                t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
                if (per_px == 8) t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0x141, 0xF, 0xF, true));   // row_half_mirror: the other quad of the 8
                return t;
              };
              const bool cok = chs + 0 < S.C && ok;   // (a quad past the channels loaded channel 0..3: it must not enter the sums)
              const float mean = px_sum(cok ? (v[0] + v[1]) + (v[2] + v[3]) : 0.f) * ln_invC;
              const f32x4 dv = v - mean;
              const float var = px_sum(cok ? (dv[0] * dv[0] + dv[1] * dv[1]) + (dv[2] * dv[2] + dv[3] * dv[3]) : 0.f) * ln_invC;
              const float rstd = rsqrtf(var + S.ln_eps);
              v = dv * rstd * ln_g + ln_b;
              if (f == 0 && ok && blockIdx.y == 0 && S.ln_stats) *reinterpret_cast<float2*>(S.ln_stats + 2 * (int64_t)gp) = float2{mean, rstd};
            }
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
        }   // (!UP)
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
        wfrag xq[NPG];
        if constexpr (XPF) {
          const int fy0 = A.transposed ? KS - 1 : 0, fx0 = A.transposed ? KS - 1 : 0;
          const int toff0 = S2T ? s2_off[0] * P.CS : (fy0 * P.XW + fx0) * P.CS;
#pragma unroll
          for (int g = 0; g < NPG; ++g) xq[g] = ldfrag<BF>(&XS[pbase[g] + toff0]);
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
              uint2 xa[NPG], xb[NPG];
#pragma unroll
              for (int g = 0; g < NPG; ++g) { xa[g] = ldfrag<true>(&XS[pbase[g] + toff]); xb[g] = ldfrag<true>(&XS[pbase[g] + toff + KD]); }
#pragma unroll
              for (int c = 0; c < NCT; ++c)
#pragma unroll
                for (int g = 0; g < NPG; ++g) acc[g][c] = mfma_bf16x2(wa[c], wb[c], xa[g], xb[g], acc[g][c]);
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
          wfrag xg[NPG];
          if constexpr (XPF) {
#pragma unroll
            for (int g = 0; g < NPG; ++g) xg[g] = xq[g];
            const int tapq = itn >> ksh, kkq = itn & ksh;
            const int tyq = tapq / KS, txq = tapq - tyq * KS;
            const int fyq = A.transposed ? KS - 1 - tyq : tyq, fxq = A.transposed ? KS - 1 - txq : txq;
            const int toffq = S2T ? s2_off[tapq & 3] * P.CS : (fyq * P.XW + fxq) * P.CS;
#pragma unroll
            for (int g = 0; g < NPG; ++g) xq[g] = ldfrag<BF>(&XS[pbase[g] + toffq + kkq * KD]);
          } else {
#pragma unroll
            for (int g = 0; g < NPG; ++g) xg[g] = ldfrag<BF>(&XS[pbase[g] + toff + kk * KD]);
          }
          if constexpr (BF) {
#pragma unroll
            for (int c = 0; c < NCT; ++c)
#pragma unroll
              for (int g = 0; g < NPG; ++g) acc[g][c] = mfma_bf16(wcur[c], xg[g], acc[g][c]);
          } else {
          const int nj = (S.C - (kb0 + kk) * 16 + 3) >> 2;  // K slices of this block that hold channels (wave-uniform)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (j == 0 || j < nj) {
#pragma unroll
              for (int c = 0; c < NCT; ++c)
                // every group unconditionally (a missing group reads pixel 0 and is dropped in the epilogue)
#pragma unroll
                for (int g = 0; g < NPG; ++g) acc[g][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wcur[c][j], xg[g][j], acc[g][c], 0, 0, 0);
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
    if constexpr (EPI == 6) {
      // LayerNorm backward (LMN_EP_LN_BWD): the block holds every channel of its pixels (grid.y == 1) -- the channel means are sums over a
      // lane's quads of the NCT cout tiles and over the four lane groups q of the pixel (two shuffles)
      const float invC = 1.f / (float)A.Cout;
#pragma unroll
      for (int g = 0; g < NPG; ++g) {
        const uint32_t opx = (uint32_t)opix[g];
        const float2 mr = *reinterpret_cast<const float2*>(A.p6 + 2 * (size_t)opx);
        f32x4 gx[NCT], zh[NCT];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
          const int co = (ct0 + c) * 16 + q * 4;
          const bool cok = (ct0 + c < P.NCTT) && co < A.Cout;
          const int cos = cok ? co : 0;
          const f32x4 ga = *reinterpret_cast<const f32x4*>(s_par + c * 16 + q * 4 + 2 * NCT * 16);   // p0 = gamma (0 past Cout)
          const f32x4 ax = ld4((const TA*)A.aux + opx * A.aux_cstride + cos);
          const f32x4 v = acc[g][c];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            zh[c][r] = cok ? (ax[r] - mr.x) * mr.y : 0.f;
            gx[c][r] = v[r] * ga[r];
            s1 += gx[c][r];
            s2 += gx[c][r] * zh[c][r];
          }
          if (pvalid[g] && cok && gl[g]) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { st0[c][r] += v[r]; st1[c][r] += v[r] * zh[c][r]; }
          }
        }
        s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
        const float m1 = s1 * invC, m2 = s2 * invC;
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
          const int co = (ct0 + c) * 16 + q * 4;
          const bool cok = (ct0 + c < P.NCTT) && co < A.Cout;
          const int cos = cok ? co : 0;
          f32x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = mr.y * (gx[c][r] - m1 - zh[c][r] * m2);
          if (A.residual) o += ld4((const TA*)A.residual + opx * A.res_cstride + cos);
          if (A.out && pvalid[g] && cok && gl[g]) st4((TA*)A.out + opx * A.out_cstride + cos, o);
        }
      }
    } else
#pragma unroll
    for (int g = 0; g < NPG; ++g) {
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
        const bool live = pvalid[g] && cok && gl[g];
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
  const bool chan_stats = EPI && ((st_mode == LMN_STATS_SUM_SQ) || (ep_kind == LMN_EP_BN_BWD1) || se || EPI == 6);
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
    const int wsw = (EPI == 6 && A.act != 0) ? 1 : 0;   // LN_BWD: rows (d beta, d gamma) or, swapped, (d gamma, d beta)
    for (int i = tid; i < (se ? 1 : 2) * NCT * 16; i += 256) {
      const int which = i / (NCT * 16), cc = i - which * NCT * 16;
      const int co = ct0 * 16 + cc;
      const float v = ((XS[i] + XS[2 * NCT * 16 + i]) + XS[4 * NCT * 16 + i]) + XS[6 * NCT * 16 + i];
      if (co < A.Cout) {
        if (se) {
          if (cur_b >= 0) lmn_red_add((det ? P.det_stats + (int64_t)((blockIdx.x + gridDim.x * blockIdx.z) * 4) * A.B * A.Cout : A.stats) + cur_b * A.Cout + co, v, det);
        } else if (det) {
          P.det_stats[(int64_t)(blockIdx.x + gridDim.x * blockIdx.z) * 2 * A.Cout + (int64_t)(which ^ wsw) * A.Cout + co] = v;   // slot of this block: [2][Cout]
        } else {
          atomicAdd(A.stats + (A.stats_rep > 1 ? (int64_t)(blockIdx.x % A.stats_rep) * 2 * A.Cout : 0) + (int64_t)(which ^ wsw) * A.Cout + co, v);
        }
      }
    }
  }
}
#undef LMN_PAR
#undef LMN_SLOT

}  // namespace
