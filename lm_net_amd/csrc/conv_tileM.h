// conv_tileM_kernel: the M-split form of the LDS-tiled conv kernel for wide layers.
#pragma once
#include "conv_tile.h"   // conv_stage_params

namespace {

// M-split variant for wide layers (Cout > 80): the four waves of a block own DIFFERENT cout tiles (wave w: tiles
// ct0 + w + 4i, i < NCW) and ALL pixel groups of the tile, instead of different pixel groups and all cout tiles.
// Each weight fragment is then fetched by exactly one wave of the block (the N-split form pulls every fragment
// through L1 four times; at Cout = 372 that stream, not the MFMAs, set the pace) and feeds 8 x 4 MFMAs; the pixel
// operand comes from LDS, where re-reading it per wave is cheap.
#define LMN_SLOT (wv + 4 * c)
// LN (1x1, one source, plain epilogue): LMN_SRC_LN on a source of ANY width (K chunks of <= 128 channels: the statistics of a pixel
// are not in one staging round).  A pre-pass per tile -- two threads per pixel walk the pixel's channels in global memory (the lines
// the staging reads right after: L2 / L1 hits) with sums about the pixel's first channel value (no cancellation against a large mean)
// -- leaves (mean, rstd) of the tile's pixels in LDS; the staging applies (x - mean) * rstd * gamma + beta.  Wide sources sit on the
// small feature maps (C = 48 / 96 / 372 at 88^2 / 44^2 / 22^2): the pre-pass is a few microseconds per launch.
// UP (3x3 forward, one source): LMN_SRC_UP2, as in conv_tile_kernel.
template <int TAPS, int NCW, int EPI, int PM = 0, bool LN = false, bool UP = false>
__global__ __launch_bounds__(256) void conv_tileM_kernel(const ConvParams P) {
  static_assert(!UP || (TAPS == 9 && (PM & 4) == 0), "bilinear x2 on load: NHWC 3x3 stride-1 forward calls");
  static_assert(!LN || (TAPS == 1 && EPI == 0 && (PM & 4) == 0), "LayerNorm on load: NHWC 1x1 calls with the plain epilogue");
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
  if (P.prio > 0 && P.prio < 100) lmn_wave_prio(P.prio);   // (uniform)
  if (P.prio >= 100) lmn_wave_stagger(P.prio - 100);
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
  float* s_ln = s_par + 9 * NCT * 16;     // LN: [XH*XW][2] = (mean, rstd) of the tile's pixels
  (void)s_ln;
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

    if constexpr (LN) {
      // statistics of the tile's pixels (the previous tile's staging -- the last reader of s_ln -- finished before its MFMA
      // barrier; the chunk loop's leading barrier below orders these writes before the staging reads)
      const lmn_src_t& S0 = A.src[0];
      const int C4 = S0.C >> 2;
      const float invC = 1.f / (float)S0.C;
      // tpp threads per pixel (a power of two: 256 / the tile's pixel count rounded up to one, at most 64): a thread walks every
      // tpp-th channel quad -- at C = 372 on 32-pixel tiles 8 threads x 12 loads (two threads per pixel were 47 dependent-issue
      // loads each: the fused GFT linears ran 24-42 us over the unfused pair)
      const int npix = P.XH * P.XW;
      int np2 = 4;
      while (np2 < npix) np2 <<= 1;
      const int tpp = 256 / np2;                       // npix <= 128: 2 .. 64
      const int p = tid / tpp, h = tid - p * tpp;
      {
        const bool inp = p < npix;
        const int ps = inp ? p : 0;
        const int r = (int)__umulhi((uint32_t)ps, P.mXW), c = ps - r * P.XW;
        const int iy = wy0 + r, ix = wx0 + c;
        const bool okp = inp && (unsigned)iy < (unsigned)A.Hin && (unsigned)ix < (unsigned)A.Win;
        const int gp = okp ? (b * A.Hin + iy) * A.Win + ix : 0;
        const TA* px = (const TA*)S0.ptr + (uint32_t)(gp * S0.cstride);
        const float x0 = ld4(px)[0];
        float sm = 0.f, sq = 0.f;
        for (int f = h; f < C4; f += tpp) {
          const f32x4 d = ld4(px + f * 4) - x0;
          sm += (d[0] + d[1]) + (d[2] + d[3]);
          sq += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
        }
        for (int m = tpp >> 1; m >= 1; m >>= 1) {
          sm += __shfl_xor(sm, m, 64);
          sq += __shfl_xor(sq, m, 64);
        }
        const float md = sm * invC;
        float var = sq * invC - md * md;
        var = var > 0.f ? var : 0.f;
        float mean = x0 + md;
        // The sums are about channel 0.  When that channel sits far from the row mean (md^2 >> var) both x0 + md and the subtraction
        // above cancel digits: such rows (uniform over a pixel's tpp threads -- the values are the reduced ones) take a second pass
        // about the first estimate of the mean, as the N-split kernel's in-register two-pass form does.  Its loads hit L1 / L2.
        if (md * md > 4.f * var) {
          float s1 = 0.f, s2 = 0.f;
          for (int f = h; f < C4; f += tpp) {
            const f32x4 d = ld4(px + f * 4) - mean;
            s1 += (d[0] + d[1]) + (d[2] + d[3]);
            s2 += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
          }
          for (int m = tpp >> 1; m >= 1; m >>= 1) { s1 += __shfl_xor(s1, m, 64); s2 += __shfl_xor(s2, m, 64); }
          const float dm = s1 * invC;
          mean += dm;
          var = s2 * invC - dm * dm;
          var = var > 0.f ? var : 0.f;
        }
        const float rstd = rsqrtf(var + S0.ln_eps);
        if (h == 0 && inp) {
          s_ln[2 * p] = mean;
          s_ln[2 * p + 1] = rstd;
          if (okp && blockIdx.y == 0 && S0.ln_stats) *reinterpret_cast<float2*>(S0.ln_stats + 2 * (int64_t)gp) = float2{mean, rstd};
        }
      }
    }
    for (int s = 0; s < A.nsrc; ++s) {
      const lmn_src_t& S = A.src[s];
      const LmnLay LS = P.lay_src[s];   // (used by the row-planar instances only)
      (void)LS;
      for (int kb0 = 0, nkbc = 0; kb0 < P.nkb[s]; kb0 += nkbc) {
        nkbc = P.nkb[s] - kb0 < P.CKB ? P.nkb[s] - kb0 : P.CKB;
        if (TAPS == 9) nkbc = nkbc >= 8 ? 8 : nkbc >= 4 ? 4 : nkbc >= 2 ? 2 : 1;   // 3x3: chunks of 1 / 2 / 4 / 8 K16 blocks (the step index splits by shift / mask)
        // step it = (tap, kk).  3x3: nkbc is 1 or 2, tap = it >> ksh, kk = it & kmsk; 1x1: kk = it, chunks of up to 8 K16 blocks (the
        // whole K of most wide layers: ONE staging round trip + barrier pair per tile instead of one per 32 channels -- on the small
        // maps a block's life was that chain, phase clocks: staging + barriers 40-50 %, MFMA 36 %)
        const int ksh = nkbc >= 8 ? 3 : nkbc >= 4 ? 2 : nkbc - 1, kmsk = nkbc - 1, niter = TAPS * nkbc;   // (3x3) tap = it >> ksh, kk = it & kmsk
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
              const bool ok = i < nitems && f < nq && ch < S.C && (unsigned)iy < (unsigned)A.Hin && (unsigned)ix < (unsigned)A.Win;
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
              if (f >= nq) continue;
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
            if constexpr (LN) {
              const float2 mr = *reinterpret_cast<const float2*>(&s_ln[2 * pix]);
              v = (v - mr.x) * mr.y * ld4(S.ln_gamma + chs) + ld4(S.ln_beta + chs);
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
            const int tapn = TAPS == 1 ? 0 : itn >> ksh, kkn = TAPS == 1 ? itn : itn & kmsk;
            const float* wp = wlane + (((int64_t)tapn * P.NKB + P.kb_off[s] + kb0 + kkn) * P.NCTT) * WT;
#pragma unroll
            for (int c = 0; c < NCW; ++c) wnext[c] = ldfrag<BF>(wp + wtile[c]);
          }
          const int tap = TAPS == 1 ? 0 : it >> ksh, kk = TAPS == 1 ? it : it & kmsk;
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
            toffq = (fyq * P.XW + fxq) * P.CS; kkq = TAPS == 1 ? itn : itn & kmsk;
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
#undef LMN_PAR
#undef LMN_SLOT

}  // namespace
