// Multi-branch depthwise stencil of ReparamConv (core/modules.py:548-574, 592-597).
//
//   pre = sum_b BN_b( dw_b(x1) ),  b in {5x5, 3x3, 3x1, 1x3};   g = GELU(pre)
//
// HBM-bound (arithmetic intensity ~1-2 FLOP/B): the job of these kernels is to move each activation
// byte once.  In NHWC the 4 BatchNorm'd branches collapse to ONE 5x5 per-channel stencil once the
// per-branch scale A_b = gamma_b * rstd_b is known (lmn_dw_merge) -- in eval/deploy mode that is the
// reference's own re-parameterisation (modules.py:622-642), in training it needs the batch statistics
// of every branch first (lmn_dw_stats, a read-only pass).
//
// Forward (lmn_dw_fwd, the flagship kernel), backward pass 2 (lmn_dw_bwd): STRIP-WALKING form -- a wave owns one
// channel pair, its 64 lanes are 64 adjacent columns, the pair's stencil weights live in SGPRs and feed
// v_pk_fma_f32 directly; the wave walks down the rows of a segment: one x1 row (5 ds_read_b64: columns x-2..x+2)
// feeds rotating row accumulators (register blocking along y); rows are staged / drained through LDS in batches by
// the whole block with coalesced 16 B accesses (LDS pixel stride 10 floats = conflict-free for 64 columns).
// Both statistics passes (lmn_dw_stats, lmn_dw_bwd_stats) use the same strip form.
#include "common.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

// XCD-aware block order (guide T1, bijective form): consecutive LOGICAL tiles run on the same XCD, so the 2-pixel
// halos shared by neighbouring tiles hit that XCD's L2 instead of being re-fetched from HBM by another XCD.
__device__ __forceinline__ int xcd_swizzle(int bid, int nwg) {
  const int qd = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  return (xcd < r ? xcd * (qd + 1) : r * (qd + 1) + (xcd - r) * qd) + (bid >> 3);
}


// ---------------------------------------------------------------------------------------------------------------
// z-path (lmn_dw_pre_t): the tensor handed to the depthwise kernels is z, the expand conv's output BEFORE its BatchNorm and
// Hardswish; x1 = Hardswish(A * z + shift) is formed when a row batch is committed to LDS (once per staged element).  The
// training forward then needs no statistics-only conv and no second read of the expand conv's input, and the backward can
// form dh = dx1 * Hardswish'(A * z + shift) and its BatchNorm-backward sums where dx1 leaves the depthwise backward.
// Zero padding: rows / columns / channels outside the tensor must stay 0 AFTER the transform -> an explicit 0 / 1 factor.
typedef lmn_dw_pre_t DwPreK;   // A / shift [E] (NULL: the tensor holds x1, no transform); fin.mode LMN_FIN_BN: formed in the kernel
struct DwPreS { const float* A; const float* shift; };   // the same without the finalisation (kernel arguments of the other passes)
__device__ __forceinline__ f32x4 dw_pre4(f32x4 z, f32x4 a, f32x4 sh, float ok) {
  f32x4 x = z * a + sh;
  f32x4 t = x * (1.f / 6.f) + 0.5f;
#pragma unroll
  for (int r = 0; r < 4; ++r) t[r] = __builtin_amdgcn_fmed3f(t[r], 0.f, 1.f);
  return x * t * ok;
}
// A / shift of the block's SW_CH = 8 channels -> pre_s[0..7] = A, pre_s[8..15] = shift (threads 0..7), formed from the batch sums
// of the expand conv when fin.mode says so (lmn_bn_fin_t arithmetic: slices summed in double about `about`); `writer` blocks
// also store mean / rstd / A / shift and blend the running statistics.
__device__ __forceinline__ void dw_pre_setup(const DwPreS& P, float* pre_s, int ch0, int E, int tid) {
  if (tid < 8) {
    const int e = ch0 + tid;
    const bool cok = e < E;
    pre_s[tid] = cok ? P.A[e] : 0.f;
    pre_s[8 + tid] = cok ? P.shift[e] : 0.f;
  }
}
__device__ __forceinline__ void dw_pre_setup(const DwPreK& P, float* pre_s, int ch0, int E, int tid, bool writer) {
  if (tid < 8) {
    const int e = ch0 + tid;
    const bool cok = e < E;
    const int es = cok ? e : 0;
    float a = 0.f, sh = 0.f;
    if (P.fin.mode == LMN_FIN_BN) {
      const lmn_bn_fin_t& F = P.fin;
      double s0 = 0.0, s1 = 0.0;
      for (int r = 0; r < F.nrep; ++r) {
        s0 += (double)F.sums[(int64_t)r * 2 * E + es];
        s1 += (double)F.sums[(int64_t)r * 2 * E + E + es];
      }
      const double md = s0 / (double)F.count;
      float var = (float)(s1 / (double)F.count - md * md);  // biased
      var = var > 0.f ? var : 0.f;
      const float m = (float)md + (F.about ? F.about[es] : 0.f);
      const float rs = rsqrtf(var + F.eps);
      a = F.gamma[es] * rs;
      sh = F.beta[es] - m * a;
      if (writer && cok) {
        if (F.mean) F.mean[e] = m;
        if (F.rstd) F.rstd[e] = rs;
        if (F.A) F.A[e] = a;
        if (F.shift) F.shift[e] = sh;
        if (F.rmean) F.rmean[e] = (1.f - F.momentum) * F.rmean[e] + F.momentum * m;
        if (F.rvar) F.rvar[e] = (1.f - F.momentum) * F.rvar[e] + F.momentum * var * (F.count > 1.f ? F.count / (F.count - 1.f) : 1.f);
      }
    } else if (P.A) {
      a = P.A[es];
      sh = P.shift[es];
    }
    pre_s[tid] = cok ? a : 0.f;
    pre_s[8 + tid] = cok ? sh : 0.f;
  }
}

// In-kernel BatchNorm bookkeeping of the depthwise block (by value in the kernel arguments; stats / bstats == NULL: off).
// Forward: the merged 5x5 stencil of a wave's channel pair is formed from the batch sums and the four branch weights
// (lmn_dw_finalize_merge arithmetic); backward: the coefficients cA / cC / cD of f_b (lmn_dw_bwd_coef arithmetic).  The
// first block of every channel chunk also writes the side outputs (saved mean / rstd / A and the running statistics; the
// gamma / beta gradients).
struct DwFin {
  const float* stats;  // [4][2][E] batch sums
  float count;
  const float* gamma[4];
  const float* beta[4];
  float* rmean[4];
  float* rvar[4];
  float eps[4], mom[4];
  const float* w5; const float* w3; const float* wv; const float* wh;
  float* mean; float* rstd; float* A;  // [4][E] out
};
struct DwCoef {
  const float* bstats;  // [5][E]
  const float* mean; const float* rstd; const float* A;  // [4][E] of the forward
  float count;
  int batch_stats;
  float* dg[4];
  float* db[4];
};

struct BranchW {  // this thread's channel pair of the four branch kernels
  f32x2 w5[25], w3[9], wv[3], wh[3];
};

__device__ __forceinline__ void load_branch_w(BranchW& bw, const float* w5, const float* w3, const float* wv,
                                              const float* wh, int ch, int E) {
  const f32x2 z = f32x2{0.f, 0.f};
  const bool ok = ch < E;  // partial last chunk: invalid pairs carry zero weights
#pragma unroll
  for (int t = 0; t < 25; ++t) bw.w5[t] = ok ? f32x2{w5[(int64_t)ch * 25 + t], w5[(int64_t)(ch + 1) * 25 + t]} : z;
#pragma unroll
  for (int t = 0; t < 9; ++t) bw.w3[t] = ok ? f32x2{w3[(int64_t)ch * 9 + t], w3[(int64_t)(ch + 1) * 9 + t]} : z;
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    bw.wv[t] = ok ? f32x2{wv[(int64_t)ch * 3 + t], wv[(int64_t)(ch + 1) * 3 + t]} : z;
    bw.wh[t] = ok ? f32x2{wh[(int64_t)ch * 3 + t], wh[(int64_t)(ch + 1) * 3 + t]} : z;
  }
}

__global__ void dw_merge_kernel(const float* __restrict__ w5, const float* __restrict__ w3, const float* __restrict__ wv,
                                const float* __restrict__ wh, const float* __restrict__ A, const float* __restrict__ shift,
                                float* __restrict__ keff, float* __restrict__ beff, int E) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= E * 25) return;
  const int e = i / 25, t = i - e * 25, ky = t / 5, kx = t - ky * 5;
  float v = A[e] * w5[i];
  if (ky >= 1 && ky <= 3 && kx >= 1 && kx <= 3) v += A[E + e] * w3[e * 9 + (ky - 1) * 3 + (kx - 1)];
  if (kx == 2 && ky >= 1 && ky <= 3) v += A[2 * E + e] * wv[e * 3 + (ky - 1)];
  if (ky == 2 && kx >= 1 && kx <= 3) v += A[3 * E + e] * wh[e * 3 + (kx - 1)];
  keff[i] = v;
  if (t == 0) beff[e] = shift[e] + shift[E + e] + shift[2 * E + e] + shift[3 * E + e];
}

// ---------------------------------------------------------------------------------------------------------------
// backward pass 2, strip-walking form:  f_b = cA_b*dpre + cC_b*y_b + cD_b inside the image (0 outside);
//   dx1 = sum_b corr^T(f_b, w_b);   dW_b[t] += sum_p f_b[p] * x1[p+t]
//
// VALU-bound work (per pixel and channel: 40 FMAs for the four y_b, 40 for dx1, 40 for the weight gradients), so
// the layout is chosen to make every FMA a packed v_pk_fma_f32 with NO per-FMA LDS operand:
//   * a WAVE owns one channel pair; its 64 LANES are 64 adjacent columns [xs-2, xs+62) of a strip.  All weights
//     and BN coefficients of the pair are wave-uniform, i.e. live in SGPRs and feed v_pk_fma_f32 directly;
//   * the wave walks down the rows of a segment.  One x1 row (5 ds_read_b64: columns x-2..x+2) feeds the rotating
//     accumulators of the four branch outputs y_b; two steps later row q = r-2 is complete, f_b(q) = a_b*dpre +
//     c_b*y_b + d_b follows, its +-1/+-2 column neighbours come from DPP wave shifts (no LDS), and it is scattered
//     into 5 rotating dx1-row accumulators; the weight gradients pair the thread's own f history (registers) with a
//     re-read of x1 row r-4;
//   * x1 / dpre rows are staged (and dx1 rows drained) through LDS by the whole block in batches of 5 rows with
//     coalesced 16 B accesses; LDS pixel stride 10 floats (2*odd) makes the 64-column b64 reads conflict-free.
// Valid dx1 columns are lanes 2..61 (60 per strip); f on lanes 0,1,62,63 is halo.  Rows: x1 rows [ys-4, ye+4).
constexpr int SW_NW = 4, SW_CH = 2 * SW_NW, SW_CS = SW_CH + 2;
constexpr int SW_XC = 68, SW_FC = 64, SW_R = 5, SW_XR = 10, SW_OC = 60;

// (scalar temporaries on purpose: __builtin_bit_cast on a vector ELEMENT made hipcc 7.2 shift element 0 only and
//  broadcast it -- checked in the ISA)
__device__ __forceinline__ float dpp_wave_shr1(float a) {   // lane l gets lane l-1, lane 0 gets 0
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x138, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_wave_shl1(float a) {   // lane l gets lane l+1, lane 63 gets 0
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x130, 0xF, 0xF, true));
}
__device__ __forceinline__ f32x2 lane_from_left(f32x2 v) {   // result[l] = v[l-1]
  const float a0 = v.x, a1 = v.y;
  f32x2 r;
  r.x = dpp_wave_shr1(a0);
  r.y = dpp_wave_shr1(a1);
  return r;
}
__device__ __forceinline__ f32x2 lane_from_right(f32x2 v) {  // result[l] = v[l+1]
  const float a0 = v.x, a1 = v.y;
  f32x2 r;
  r.x = dpp_wave_shl1(a0);
  r.y = dpp_wave_shl1(a1);
  return r;
}

struct SwState {
  f32x2 a5[5], a3[5], av[5], ah[5];          // y_b accumulators, slot = row index mod 5
  f32x2 h5[5], h3[5], hv[5], hh[5];          // own-pixel f_b history (masked), slot = row index mod 5
  f32x2 dxa[5];                              // dx1 row accumulators, slot = row index mod 5
  f32x2 g5[25], g3[9], gv[3], gh[3];         // weight-gradient accumulators
};

// One row step.  P = step index mod 5 (compile time, so every slot below is a fixed register).
// PART: 0 = dx1 and the weight gradients in one pass, 1 = dx1 only, 2 = weight gradients only (the two halves can then run on
// different streams: dx1 is on the critical path of the backward, the weight gradients are not)
// FAST: a step in the interior of the segment (f row owned, dx row stored, weight-gradient products on): no row flags at all.
// Column flags never appear here: columns outside the image carry zero coefficients (ca / cc / cd are per-lane registers), the
// halo lanes of the strip are dropped from the weight-gradient sums at the end of the kernel.  Accumulator slots are not
// zeroed: the first contribution to a row (kernel row 0 for y_b, kernel row 4 for dx1) is a plain product.
// CL: the coefficients of f_b come from LDS (CFS; z-path instances, whose drain needs the registers: 26 spilled VGPRs -> 1,
// -3..4 % there; the plain instances keep them in registers: +12 % with the LDS form)
template <int P, int PART, bool FAST, bool CL>
__device__ __forceinline__ void sw_step(SwState& S, const BranchW& bw, const f32x2 (&ca)[4], const f32x2 (&cc)[4],
                                        const f32x2 (&cd)[4], const float* CFS, f32x2 cm, const float* XS,
                                        const float* DPS, float* OUT, int j,
                                        int lane, int wv, bool frow_in, bool own, bool dx_row, bool dw_ok) {
  // ---- x1 row j: columns x-2 .. x+2
  const float* xr = XS + ((j % SW_XR) * SW_XC + lane) * SW_CS + wv * 2;
  f32x2 in[5];
#pragma unroll
  for (int d = 0; d < 5; ++d) in[d] = *reinterpret_cast<const f32x2*>(xr + d * SW_CS);
  // (column-major order: consecutive FMAs go to DIFFERENT row accumulators.  Row-major, each accumulator took its five products
  //  back to back -- a dependent chain the two waves of a SIMD cannot hide: the pass ran at a third of its VALU issue bound)
#pragma unroll
  for (int d = 0; d < 5; ++d) {
    if (d == 0) S.a5[(P + 2) % 5] = bw.w5[0] * in[0];
    else S.a5[(P + 2) % 5] += bw.w5[d] * in[d];
#pragma unroll
    for (int ky = 1; ky < 5; ++ky) S.a5[(P - ky + 7) % 5] += bw.w5[ky * 5 + d] * in[d];
    if (d >= 1 && d <= 3) {
      if (d == 1) { S.a3[(P + 1) % 5] = bw.w3[0] * in[1]; S.ah[P] = bw.wh[0] * in[1]; }
      else { S.a3[(P + 1) % 5] += bw.w3[d - 1] * in[d]; S.ah[P] += bw.wh[d - 1] * in[d]; }
#pragma unroll
      for (int ky = 1; ky < 3; ++ky) S.a3[(P - ky + 6) % 5] += bw.w3[ky * 3 + d - 1] * in[d];
    }
    if (d == 2) {
      S.av[(P + 1) % 5] = bw.wv[0] * in[2];
#pragma unroll
      for (int ky = 1; ky < 3; ++ky) S.av[(P - ky + 6) % 5] += bw.wv[ky] * in[2];
    }
  }
  // ---- row q = j-2 is complete: f_b
  constexpr int Q = (P + 3) % 5;
  const f32x2 dp = *reinterpret_cast<const f32x2*>(DPS + (P * SW_XC + lane) * SW_CS + wv * 2);
  f32x2 f5, f3, fv, fh;
  if constexpr (CL) {
    // coefficients of the wave's channel pair from LDS (broadcast reads: CFS[(k * 3 + {a, c, d}) * 2]); cm = 0 in columns outside
    // the image (f_b = 0 there), 1 elsewhere
    const float* cfp = CFS;
    asm volatile("" : "+v"(cfp));   // (re-read per step: hoisted out of the loop the 24 values are 24 registers again)
    const f32x2* cf = reinterpret_cast<const f32x2*>(cfp);
    f5 = (cf[1] * S.a5[Q] + (cf[0] * dp + cf[2])) * cm;
    f3 = (cf[4] * S.a3[Q] + (cf[3] * dp + cf[5])) * cm;
    fv = (cf[7] * S.av[Q] + (cf[6] * dp + cf[8])) * cm;
    fh = (cf[10] * S.ah[Q] + (cf[9] * dp + cf[11])) * cm;
  } else {   // per-lane registers (zero in columns outside the image)
    f5 = cc[0] * S.a5[Q] + (ca[0] * dp + cd[0]);
    f3 = cc[1] * S.a3[Q] + (ca[1] * dp + cd[1]);
    fv = cc[2] * S.av[Q] + (ca[2] * dp + cd[2]);
    fh = cc[3] * S.ah[Q] + (ca[3] * dp + cd[3]);
  }
  if constexpr (FAST) {
    S.h5[Q] = f5; S.h3[Q] = f3; S.hv[Q] = fv; S.hh[Q] = fh;
  } else {
    // row flags as wave-uniform 0 / 1 factors (one SGPR each) instead of lane masks (an SGPR pair each, next to the 80 weight
    // SGPRs: the selects made this form reload 50 spilled scalars per step); the accumulators are zero-initialised, so the
    // products of the warm-up steps are finite
    const float mf = frow_in ? 1.f : 0.f, mo = own ? mf : 0.f;
    f5 *= mf; f3 *= mf; fv *= mf; fh *= mf;
    S.h5[Q] = f5 * mo; S.h3[Q] = f3 * mo; S.hv[Q] = fv * mo; S.hh[Q] = fh * mo;
  }
  // ---- dx1: f row q feeds dx rows q-2..q+2; column neighbours by DPP.  sh[k][l] = f[l + 2 - k]
  if constexpr (PART != 2) {
    f32x2 sh[5];
    sh[2] = f5;
    sh[1] = lane_from_right(f5);
    sh[0] = lane_from_right(sh[1]);
    sh[3] = lane_from_left(f5);
    sh[4] = lane_from_left(sh[3]);
    f32x2 s3[3];
    s3[1] = f3;
    s3[0] = lane_from_right(f3);
    s3[2] = lane_from_left(f3);
    const f32x2 hr = lane_from_right(fh), hl = lane_from_left(fh);
    // column-major again: the five dx rows take their products in turn (dx row j gets its first contribution from kernel row 4)
#pragma unroll
    for (int kx = 0; kx < 5; ++kx) {
      if (kx == 0) S.dxa[P] = bw.w5[20] * sh[0];
      else S.dxa[P] += bw.w5[20 + kx] * sh[kx];
#pragma unroll
      for (int ky = 0; ky < 4; ++ky) S.dxa[(P + ky + 1) % 5] += bw.w5[ky * 5 + kx] * sh[kx];
    }
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) S.dxa[(P + ky + 2) % 5] += bw.w3[ky * 3 + kx] * s3[kx];
    }
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) S.dxa[(P + ky + 2) % 5] += bw.wv[ky] * fv;
    S.dxa[Q] += bw.wh[0] * hr;
    S.dxa[Q] += bw.wh[1] * fh;
    S.dxa[Q] += bw.wh[2] * hl;
  }
  // ---- dx row j-4 is complete
  constexpr int D = (P + 1) % 5;
  if constexpr (PART != 2) {
    if (FAST || dx_row) *reinterpret_cast<f32x2*>(OUT + (P * SW_XC + lane) * SW_CS + wv * 2) = S.dxa[D];
  }
  // ---- weight gradients: x1 row j-4 (re-read) against the own-pixel f history
  if (PART != 1 && (FAST || dw_ok)) {
    const float* x2 = XS + (((j + SW_XR - 4) % SW_XR) * SW_XC + lane) * SW_CS + wv * 2;
    f32x2 i2[5];
#pragma unroll
    for (int d = 0; d < 5; ++d) i2[d] = *reinterpret_cast<const f32x2*>(x2 + d * SW_CS);
#pragma unroll
    for (int ky = 0; ky < 5; ++ky)
#pragma unroll
      for (int kx = 0; kx < 5; ++kx) S.g5[ky * 5 + kx] += S.h5[(P + 8 - ky) % 5] * i2[kx];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) S.g3[ky * 3 + kx] += S.h3[(P + 7 - ky) % 5] * i2[1 + kx];
      S.gv[ky] += S.hv[(P + 7 - ky) % 5] * i2[2];
    }
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) S.gh[kx] += S.hh[D] * i2[1 + kx];
  }
}

#ifdef LMN_DW_TIMING
// phase clocks of dw_bwd_strip_kernel (debug builds, tools/gpu_dw_phases.py): per block {staging incl. barriers, row steps, life}
__device__ unsigned long long g_dw_timing[4096 * 4];
#define LMN_DTK(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); dtk[i] += t_ - dta; dta = t_; } while (0)
#else
#define LMN_DTK(i) do { } while (0)
#endif
template <typename TA, int PART = 0, bool ZT = false>
__global__ __launch_bounds__(256, 2) void dw_bwd_strip_kernel(
    const TA* __restrict__ x1, const TA* __restrict__ dpre, TA* __restrict__ dx1, int B, int H, int W, int E,
    const float* __restrict__ w5, const float* __restrict__ w3, const float* __restrict__ wvv,
    const float* __restrict__ whh, const float* __restrict__ cA, const float* __restrict__ cC,
    const float* __restrict__ cD, const DwCoef CF, float* __restrict__ dw5, float* __restrict__ dw3, float* __restrict__ dwv,
    float* __restrict__ dwh, const DwPreS PRE, float* __restrict__ hstats, int strips, int segs, int seg_rows, int chunks, int det) {
  __shared__ __attribute__((aligned(16))) float XS[SW_XR * SW_XC * SW_CS];
  __shared__ __attribute__((aligned(16))) float DPS[SW_R * SW_XC * SW_CS];  // (rows of 68 like XS: one LDS index per staged item)
  __shared__ __attribute__((aligned(16))) float OUT[SW_R * SW_XC * SW_CS];
  // z-path (DwPreK): x1 is formed from z when a batch is staged; the drain turns dx1 into dh = dx1 * Hardswish'(A z + shift) and
  // sums dh, dh * z per channel (hstats [2][E]: the BatchNorm-backward statistics of the expand conv).  The per-thread sums
  // live in LDS (hacc[j][tid]: conflict-free, no accumulator registers in a kernel that has none to spare)
  __shared__ __attribute__((aligned(16))) float pre_s[16];
  __shared__ __attribute__((aligned(16))) float coef_s[ZT ? SW_NW * 24 : 4];   // z-path, per wave: (cA, cC, cD) x 4 branches x channel pair
  __shared__ float hacc[(PART == 2 || !ZT) ? 1 : 8 * 256];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  // logical block id: channel chunk fastest (siblings share the x1 / dpre cache lines -> same XCD, same L2)
  int lid = xcd_swizzle(blockIdx.x, gridDim.x);
  const int chunk = lid % chunks; lid /= chunks;
  const int strip = lid % strips; lid /= strips;
  const int seg = lid % segs;
  const int b = lid / segs;
  const int ch0 = chunk * SW_CH;
  const int ch = ch0 + wv * 2;
  const bool cok = ch < E;                      // wave-uniform: pairs past E (partial last chunk) compute on zeros
  const int chs = cok ? ch : 0;
  constexpr bool zt = ZT;                       // the input tensor is z (see DwPreK)
  if constexpr (ZT) {
    dw_pre_setup(PRE, pre_s, ch0, E, tid);
    if (PART != 2) {
#pragma unroll
      for (int j = 0; j < 8; ++j) hacc[j * 256 + tid] = 0.f;
    }
  }
  BranchW bw;
  load_branch_w(bw, w5, w3, wvv, whh, chs, E);
  f32x2 ca[4], cc[4], cd[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (CF.bstats) {  // block-uniform: coefficients formed here (lmn_dw_bwd_coef arithmetic)
      float a2[2], c2[2], d2[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int e = chs + h, i = k * E + e;
        const float S0 = CF.bstats[e], S1 = CF.bstats[(1 + k) * E + e];
        const float mu = CF.mean[i], rs = CF.rstd[i], a = CF.A[i];
        const float T = (S1 - mu * S0) * rs;  // sum dpre * yhat_b
        const float c = CF.batch_stats ? -a * T * rs / CF.count : 0.f;
        a2[h] = a; c2[h] = c;
        d2[h] = CF.batch_stats ? (-a * S0 / CF.count - c * mu) : 0.f;
        if (PART != 2 && cok && lane == 0 && strip == 0 && seg == 0 && b == 0) {  // one wave per channel pair
          CF.dg[k][e] += T;
          CF.db[k][e] += S0;
        }
      }
      ca[k] = f32x2{a2[0], a2[1]}; cc[k] = f32x2{c2[0], c2[1]}; cd[k] = f32x2{d2[0], d2[1]};
    } else {
    ca[k] = f32x2{cA[k * E + chs], cA[k * E + chs + 1]};
    cc[k] = f32x2{cC[k * E + chs], cC[k * E + chs + 1]};
    cd[k] = f32x2{cD[k * E + chs], cD[k * E + chs + 1]};
    }
    if constexpr (ZT) {
      // the 80 weight SGPRs already fill the scalar file and, with the z-path drain, the vector file is full as well: the 24
      // coefficient floats of the wave live in LDS (12 broadcast ds_read_b64 per row step)
      if (lane == 0) {
        float* c = coef_s + wv * 24 + k * 6;
        c[0] = ca[k].x; c[1] = ca[k].y; c[2] = cc[k].x; c[3] = cc[k].y; c[4] = cd[k].x; c[5] = cd[k].y;
      }
    } else {
      // the 80 weight SGPRs already fill the scalar file: keep the 24 coefficient floats in VGPRs (otherwise the
      // allocator spills weights to VGPR lanes and every use costs v_readlane x2 + s_nop)
      asm volatile("" : "+v"(ca[k].x), "+v"(ca[k].y), "+v"(cc[k].x), "+v"(cc[k].y), "+v"(cd[k].x), "+v"(cd[k].y));
    }
  }
  const float* CFS = coef_s + wv * 24;
  const int ys = seg * seg_rows, ye = min(ys + seg_rows, H);
  const int xs = strip * SW_OC;
  const int cx = xs - 2 + lane;
  const bool col_in = cx >= 0 && cx < W;
  const bool own_col = lane >= 2 && lane < 2 + SW_OC;
  const f32x2 cm = col_in ? f32x2{1.f, 1.f} : f32x2{0.f, 0.f};   // f_b = 0 in columns outside the image
  if (!ZT && !col_in) {
#pragma unroll
    for (int k = 0; k < 4; ++k) ca[k] = cc[k] = cd[k] = f32x2{0.f, 0.f};
  }
  const TA* xb = x1 + (int64_t)b * H * W * E;
  const TA* db = dpre + (int64_t)b * H * W * E;
  TA* ob = dx1 + (int64_t)b * H * W * E;

  SwState S;
  const f32x2 z2 = f32x2{0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 5; ++k) S.a5[k] = S.a3[k] = S.av[k] = S.ah[k] = S.h5[k] = S.h3[k] = S.hv[k] = S.hh[k] = S.dxa[k] = z2;
#pragma unroll
  for (int k = 0; k < 25; ++k) S.g5[k] = z2;
#pragma unroll
  for (int k = 0; k < 9; ++k) S.g3[k] = z2;
#pragma unroll
  for (int k = 0; k < 3; ++k) S.gv[k] = S.gh[k] = z2;

  // step j: x1 row (ys-4+j) enters; f row (ys-6+j); finished dx row (ys-8+j); weight-gradient products of x1 row
  // (ys-8+j).  The last own f row (ye-1) still meets x1 row ye+1 => rows + 10 steps (the last two only feed dW).
  const int nsteps = (ye - ys) + 10;
  const int ndx = (ye - ys) + 8;  // steps that finish a dx row: 8 <= j < ndx
  // Staging: bounds are the buffer descriptors' job (columns / channels outside the tensor get an offset beyond
  // num_records, rows above the image wrap negative, rows below it exceed num_records: loads return 0, stores are
  // dropped), offsets are 32-bit, and ALL loads of a batch are issued before the first LDS write (one exposed
  // memory latency per batch instead of six).
  // Geometry is fixed per thread: item i = tid + k*256 is (row rr, column c, channel half k4) of a 5-row x 68-column batch, and
  // x1 (columns xs-4+c), dpre (c in [2, 66)) and dx1 (c in [4, 64)) all use THE SAME item -> image column map, so one byte
  // offset fo[k] and one LDS index li[k] per item serve the three tensors (a batch adds its wave-uniform row offset).
  constexpr unsigned OOB = 0x80000000u;
  constexpr int NXB = (SW_R * SW_XC * 2 + 255) / 256;
  constexpr unsigned ES = sizeof(TA);
  const int rowb = W * E * (int)ES;  // bytes per image row (host: (H+16)*W*E*4 < 2^30)
  const BufRsrc rx = make_rsrc(xb, (unsigned)H * rowb);
  const BufRsrc rd = make_rsrc(db, (unsigned)H * rowb);
  const BufRsrc ro = make_rsrc(ob + (int64_t)ys * W * E, (unsigned)(ye - ys) * rowb);
  unsigned fo[NXB];
  int li[NXB];  // LDS float index of the item in a 5 x 68 x SW_CS batch | bit 16: dpre item | bit 17: dx1 item
                // | bit 18: inside the tensor (column, channel) | bits 20..22: its row in the batch   (the last two: z-path)
#pragma unroll
  for (int k = 0; k < NXB; ++k) {
    const int i = tid + k * 256;
    const int k4 = i & 1, pc = i >> 1;
    const int rr = pc / SW_XC, c = pc - rr * SW_XC;
    const int gx = xs - 4 + c;
    const bool ok = i < SW_R * SW_XC * 2 && gx >= 0 && gx < W && ch0 + k4 * 4 < E;
    fo[k] = ok ? (unsigned)((rr * W + gx) * E + ch0 + k4 * 4) * ES : OOB;
    const bool in = i < SW_R * SW_XC * 2;
    li[k] = ((rr * SW_XC + c) * SW_CS + k4 * 4) | (in && c >= 2 && c < 2 + SW_FC ? 0x10000 : 0) | (in && c >= 4 && c < 4 + SW_OC ? 0x20000 : 0) |
            (ok ? 0x40000 : 0) | (rr & 7) << 20;
  }
  if (zt) __syncthreads();   // coef_s, pre_s, hacc
  // z-path: the z rows a drain needs (the rows of the dx batch; this block staged the same lines one or two batches ago, so they
  // come back from L2) are requested at the END of the batch's row steps -- the step temporaries are dead there -- and are in
  // registers when the next staging phase drains: no load latency inside the drain, no overlap with the x1 / dpre loads' registers
  f32x4 vz[ZT ? NXB : 1];
  auto zfetch = [&](int jb) {
    if constexpr (ZT) {
      const unsigned zb = (unsigned)((ys + jb - 8) * rowb);
#pragma unroll
      for (int k = 0; k < NXB; ++k) {
        int l = li[k];
        asm volatile("" : "+v"(l));
        vz[k] = buf_load4<TA>(rx, (l & 0x20000) ? fo[k] + zb : OOB);
      }
    }
  };
  auto drain = [&](int jb) {  // dx rows of the batch that started at step jb: segment rows jb-8+rr
    const unsigned base = (unsigned)((jb - 8) * rowb);
    if constexpr (ZT) {
      // dh = dx1 * Hardswish'(A z + shift); sum dh, sum dh * z over the rows of this segment inside the image (vz: zfetch)
      const f32x4 tA = *reinterpret_cast<const f32x4*>(&pre_s[(tid & 1) * 4]), tS = *reinterpret_cast<const f32x4*>(&pre_s[8 + (tid & 1) * 4]);
      int l[NXB];
#pragma unroll
      for (int k = 0; k < NXB; ++k) {
        l[k] = li[k];
        asm volatile("" : "+v"(l[k]));
      }
      f32x4 h0 = f32x4{0.f, 0.f, 0.f, 0.f}, h1 = h0;
#pragma unroll
      for (int k = 0; k < NXB; ++k) {
        if (l[k] & 0x20000) {
          const float* o = &OUT[(l[k] & 0xFFFF) - 2 * SW_CS];
          const f32x2 a = *reinterpret_cast<const f32x2*>(o), d = *reinterpret_cast<const f32x2*>(o + 2);
          const f32x4 hh = vz[k] * tA + tS;
          f32x4 dh = f32x4{a[0] * lmn_dhswish(hh[0]), a[1] * lmn_dhswish(hh[1]), d[0] * lmn_dhswish(hh[2]), d[1] * lmn_dhswish(hh[3])};
          buf_store4<TA>(ro, fo[k] + base, dh);
          const bool in = (l[k] & 0x40000) && (unsigned)(jb - 8 + ((l[k] >> 20) & 7)) < (unsigned)(ye - ys);
          if (!in) dh = f32x4{0.f, 0.f, 0.f, 0.f};   // (a select: rows past the segment hold whatever the LDS held)
          h0 += dh;
          h1 += dh * vz[k];
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        hacc[j * 256 + tid] += h0[j];
        hacc[(4 + j) * 256 + tid] += h1[j];
      }
      return;
    }
#pragma unroll
    for (int k = 0; k < NXB; ++k) {
      int l = li[k];
      asm volatile("" : "+v"(l));  // (derived indices / masks stay inside the batch: hoisted, they cost accumulator registers)
      if (l & 0x20000) {
        const float* o = &OUT[(l & 0xFFFF) - 2 * SW_CS];  // dx column c-4 sits at lane c-2
        const f32x2 a = *reinterpret_cast<const f32x2*>(o), d = *reinterpret_cast<const f32x2*>(o + 2);
        buf_store4<TA>(ro, fo[k] + base, f32x4{a[0], a[1], d[0], d[1]});
      }
    }
  };
  auto stage = [&](int j0) {
    __syncthreads();  // previous batch computed: OUT holds its dx rows, XS/DPS slots are free
    if constexpr (ZT) {
      if (PART != 2 && j0 >= 8 + SW_R - 4) drain(j0 - SW_R);  // (before the loads below are requested: their 24 registers and the drain's do not add up)
    }
    // ---- stage x1 rows (image rows ys-4+j, columns xs-4 .. xs+63) and dpre rows (rows ys-6+j, columns xs-2 .. xs+61)
    {
      const unsigned bx = (unsigned)((ys - 4 + j0) * rowb), bd = (unsigned)((ys - 6 + j0) * rowb);
      f32x4 vx[NXB], vd[NXB];
      int l[NXB];
#pragma unroll
      for (int k = 0; k < NXB; ++k) {
        l[k] = li[k];
        asm volatile("" : "+v"(l[k]));
      }
#pragma unroll
      for (int k = 0; k < NXB; ++k) vx[k] = buf_load4<TA>(rx, fo[k] + bx);  // (OOB + base stays out of range)
#pragma unroll
      for (int k = 0; k < NXB; ++k) vd[k] = buf_load4<TA>(rd, (l[k] & 0x10000) ? fo[k] + bd : OOB);
      if constexpr (!ZT) {
        if (PART != 2 && j0 >= 8 + SW_R - 4) drain(j0 - SW_R);  // dx rows of the previous batch leave while the loads are in flight
      }
      float* ring = XS + (j0 % SW_XR) * (SW_XC * SW_CS);
      if constexpr (ZT) {   // x1 = Hardswish(A z + shift) where the batch is committed; rows / columns outside the image stay 0
        const f32x4 tA = *reinterpret_cast<const f32x4*>(&pre_s[(tid & 1) * 4]), tS = *reinterpret_cast<const f32x4*>(&pre_s[8 + (tid & 1) * 4]);
        const int yb = ys - 4 + j0;
        const bool inner = yb >= 0 && yb + SW_R <= H;
#pragma unroll
        for (int k = 0; k < NXB; ++k) {
          bool ok = (l[k] & 0x40000) != 0;
          if (!inner) ok = ok && (unsigned)(yb + ((l[k] >> 20) & 7)) < (unsigned)H;
          vx[k] = dw_pre4(vx[k], tA, tS, ok ? 1.f : 0.f);
        }
      }
#pragma unroll
      for (int k = 0; k < NXB; ++k) {
        if (k * 256 + 255 < SW_R * SW_XC * 2 || tid + k * 256 < SW_R * SW_XC * 2) {
          float* d = ring + (l[k] & 0xFFFF);
          *reinterpret_cast<f32x2*>(d) = f32x2{vx[k][0], vx[k][1]};
          *reinterpret_cast<f32x2*>(d + 2) = f32x2{vx[k][2], vx[k][3]};
        }
      }
#pragma unroll
      for (int k = 0; k < NXB; ++k) {
        if (l[k] & 0x10000) {
          float* d = &DPS[(l[k] & 0xFFFF) - 2 * SW_CS];  // dpre column c-2 sits at lane c-2
          *reinterpret_cast<f32x2*>(d) = f32x2{vd[k][0], vd[k][1]};
          *reinterpret_cast<f32x2*>(d + 2) = f32x2{vd[k][2], vd[k][3]};
        }
      }
    }
    __syncthreads();
  };
  // ---- five row steps per batch (fixed register slots per phase).  Interior batches (every step: f row owned by the segment, dx
  // row stored, weight-gradient products on) take the flag-free form of the step; the three loops run one after the other (both
  // forms inside ONE loop made the allocator spill 200 registers to scratch)
#define LMN_SW_STEP(PH)                                                                                       \
    {                                                                                                         \
      const int j = j0 + PH;                                                                                  \
      if (j < nsteps) {                                                                                       \
        const int fy = ys - 6 + j;                                                                            \
        const bool frow_in = j >= 4 && fy >= 0 && fy < H;                                                     \
        const bool own = fy >= ys && fy < ye;                                                                 \
        sw_step<PH, PART, false, ZT>(S, bw, ca, cc, cd, CFS, cm, XS, DPS, OUT, j, lane, wv, frow_in, own, j >= 8 && j < ndx, j >= 4); \
      }                                                                                                       \
    }
#define LMN_SW_FAST(PH) sw_step<PH, PART, true, ZT>(S, bw, ca, cc, cd, CFS, cm, XS, DPS, OUT, j0 + PH, lane, wv, true, true, true, true);
#ifdef LMN_DW_TIMING
  unsigned long long dtk[2] = {0, 0}, dta = __builtin_amdgcn_s_memtime();
  const unsigned long long dt0 = dta;
#endif
  int j0 = 0;
  for (; j0 < nsteps && j0 < 8; j0 += SW_R) {
    stage(j0);
    LMN_DTK(0);
    LMN_SW_STEP(0) LMN_SW_STEP(1) LMN_SW_STEP(2) LMN_SW_STEP(3) LMN_SW_STEP(4)
    if (PART != 2) zfetch(j0);
    LMN_DTK(1);
  }
  for (; j0 + 4 < (ye - ys) + 6; j0 += SW_R) {
    stage(j0);
    LMN_DTK(0);
    LMN_SW_FAST(0) LMN_SW_FAST(1) LMN_SW_FAST(2) LMN_SW_FAST(3) LMN_SW_FAST(4)
    if (PART != 2) zfetch(j0);
    LMN_DTK(1);
  }
  for (; j0 < nsteps; j0 += SW_R) {
    stage(j0);
    LMN_DTK(0);
    LMN_SW_STEP(0) LMN_SW_STEP(1) LMN_SW_STEP(2) LMN_SW_STEP(3) LMN_SW_STEP(4)
    if (PART != 2) zfetch(j0);
    LMN_DTK(1);
  }
#ifdef LMN_DW_TIMING
  if (tid == 0 && blockIdx.x < 4096) {
    g_dw_timing[blockIdx.x * 4] = dtk[0]; g_dw_timing[blockIdx.x * 4 + 1] = dtk[1];
    g_dw_timing[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_memtime() - dt0; g_dw_timing[blockIdx.x * 4 + 3] = nsteps;
  }
#endif
#undef LMN_SW_STEP
#undef LMN_SW_FAST
  __syncthreads();
  if (PART != 2) drain(((nsteps + SW_R - 1) / SW_R) * SW_R - SW_R);  // dx rows of the last batch
  if (PART != 2 && zt) {   // hstats: 128 threads per channel quad -> 16 sums per block -> one atomic instruction
    __syncthreads();
    if (tid < 16) {
      const int k4 = (tid >> 2) & 1, r = tid & 3, which = tid >> 3;   // tid = which * 8 + k4 * 4 + r
      float a = 0.f;
      for (int t = k4; t < 256; t += 2) a += hacc[(which * 4 + r) * 256 + t];
      const int e = ch0 + k4 * 4 + r;
      // (deterministic mode: hstats / dw5 .. dwh address slot copies of the destinations, one slot per (image, segment, strip))
      if (e < E) lmn_red_add(hstats + (det ? (int64_t)((b * segs + seg) * strips + strip) * 2 * E : 0) + (int64_t)which * E + e, a, det);
    }
  }
  if (PART == 1) return;
  // ---- weight gradients: butterfly over the 64 columns, one LDS row per wave, then one atomic per (tap, channel)
  __syncthreads();
  float* red = XS;  // [4 waves][40 taps][2]
  auto wave_sum_store = [&](f32x2 v, int t) {  // DPP reduction: total lands in lane 63 (no LDS round trips)
    float a = own_col ? v.x : 0.f, c = own_col ? v.y : 0.f;  // (the halo lanes of the strip belong to its neighbours)
#define LMN_DPP_ADD(CTRL)                                                                               \
    a += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), CTRL, 0xF, 0xF, true));       \
    c += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(c), CTRL, 0xF, 0xF, true));
    LMN_DPP_ADD(0x111) LMN_DPP_ADD(0x112) LMN_DPP_ADD(0x114) LMN_DPP_ADD(0x118)  // row_shr 1, 2, 4, 8
    LMN_DPP_ADD(0x142) LMN_DPP_ADD(0x143)                                        // row_bcast 15, 31
#undef LMN_DPP_ADD
    if (lane == 63) *reinterpret_cast<f32x2*>(&red[(wv * 40 + t) * 2]) = f32x2{a, c};
  };
#pragma unroll
  for (int t = 0; t < 25; ++t) wave_sum_store(S.g5[t], t);
#pragma unroll
  for (int t = 0; t < 9; ++t) wave_sum_store(S.g3[t], 25 + t);
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    wave_sum_store(S.gv[t], 34 + t);
    wave_sum_store(S.gh[t], 37 + t);
  }
  __syncthreads();
  for (int i = tid; i < SW_NW * 40 * 2; i += 256) {
    const int k = i & 1, t = (i >> 1) % 40, w = i / 80;
    const int e = ch0 + w * 2 + k;
    if (e >= E) continue;
    const float v = red[i];
    const int64_t ds = det ? (int64_t)((b * segs + seg) * strips + strip) * E : 0;   // slot offset in channels
    if (t < 25) lmn_red_add(dw5 + (ds + e) * 25 + t, v, det);
    else if (t < 34) lmn_red_add(dw3 + (ds + e) * 9 + t - 25, v, det);
    else if (t < 37) lmn_red_add(dwv + (ds + e) * 3 + t - 34, v, det);
    else lmn_red_add(dwh + (ds + e) * 3 + t - 37, v, det);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Forward in the same strip-walking form (flagship HBM-bound kernel of row A2): wave = channel pair, lanes = 64
// adjacent output columns, the merged 5x5 kernel of the pair in SGPRs.  Per row step: 5 ds_read_b64 + 25 packed
// FMAs into 5 rotating row accumulators; the finished row gets bias + GELU-sum and leaves through LDS as coalesced
// 16 B stores.  No halo recompute along x (68 staged columns for 64 outputs), 4 halo rows per row segment.
constexpr int FS_XR = 2 * SW_R;  // x1 ring: the batch being consumed + the batch being committed

template <int P, bool FAST = false>
__device__ __forceinline__ void fs_step(f32x2 (&acc)[5], f32x2& gs, const f32x2 (&w)[25], f32x2 bias, const float* XS,
                                        float* OUT, int j, int lane, int wv, bool row_out) {
  const float* xr = XS + ((j % FS_XR) * SW_XC + lane) * SW_CS + wv * 2;
  f32x2 in[5];
#pragma unroll
  for (int d = 0; d < 5; ++d) in[d] = *reinterpret_cast<const f32x2*>(xr + d * SW_CS);
  // column-major: consecutive FMAs go to different row accumulators (back-to-back dependent v_pk_fma_f32 cost an s_nop each);
  // output row j gets its first contribution as a plain product (no zeroed accumulators)
#pragma unroll
  for (int d = 0; d < 5; ++d) {
    if (d == 0) acc[P] = w[0] * in[0];
    else acc[P] += w[d] * in[d];
#pragma unroll
    for (int ky = 1; ky < 5; ++ky) acc[(P - ky + 5) % 5] += w[ky * 5 + d] * in[d];
  }
  constexpr int D = (P + 1) % 5;  // output row j-4 is complete
  if (FAST || row_out) {
    const f32x2 pv = acc[D] + bias;
    *reinterpret_cast<f32x2*>(OUT + (P * SW_FC + lane) * SW_CS + wv * 2) = pv;
    gs += f32x2{lmn_gelu(pv[0]), lmn_gelu(pv[1])};  // (columns outside the image are dropped from the lane sums at the end)
  }
}

template <typename TA>
__global__ __launch_bounds__(256) void dw_fwd_strip_kernel(const TA* __restrict__ x1, TA* __restrict__ pre,
                                                            float* __restrict__ gsum, int H, int W, int E,
                                                            const float* __restrict__ keff,
                                                            const float* __restrict__ beff, const DwFin FN,
                                                            const lmn_se_fuse_t SE, const DwPreS PRE, int strips,
                                                            int segs, int seg_rows, int chunks, int det) {
  __shared__ __attribute__((aligned(16))) float XS[FS_XR * SW_XC * SW_CS];
  __shared__ __attribute__((aligned(16))) float OUT[SW_R * SW_FC * SW_CS];
  __shared__ float gs_s[SW_CH];
  __shared__ int s_last;
  __shared__ __attribute__((aligned(16))) float pre_s[16];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  int lid = xcd_swizzle(blockIdx.x, gridDim.x);  // channel chunk fastest: siblings share cache lines and an L2
  const int chunk = lid % chunks; lid /= chunks;
  const int strip = lid % strips; lid /= strips;
  const int seg = lid % segs;
  const int b = lid / segs;
  const int ch0 = chunk * SW_CH, ch = ch0 + wv * 2;
  const bool cok = ch < E;
  const int chs = cok ? ch : 0;
  const bool zt = PRE.A != nullptr;   // block-uniform: the input tensor is z (see DwPreK)
  if (zt) dw_pre_setup(PRE, pre_s, ch0, E, tid);
  f32x2 w[25];
  f32x2 bias;
  if (FN.stats) {  // block-uniform: the four branch BatchNorms are finalised and merged here (lmn_dw_finalize_merge arithmetic)
    float wm[2][25], bs[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int e = chs + h;
      float a4[4];
      float sh = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float m = FN.stats[(t * 2) * E + e] / FN.count;
        float var = FN.stats[(t * 2 + 1) * E + e] / FN.count - m * m;  // biased
        var = var > 0.f ? var : 0.f;
        const float rs = rsqrtf(var + FN.eps[t]);
        const float a = FN.gamma[t][e] * rs;
        a4[t] = a;
        sh += FN.beta[t][e] - m * a;
        if (cok && lane == 0 && strip == 0 && seg == 0 && b == 0) {  // one wave per channel pair writes the side outputs
          FN.mean[t * E + e] = m;
          FN.rstd[t * E + e] = rs;
          FN.A[t * E + e] = a;
          FN.rmean[t][e] = (1.f - FN.mom[t]) * FN.rmean[t][e] + FN.mom[t] * m;
          FN.rvar[t][e] = (1.f - FN.mom[t]) * FN.rvar[t][e] + FN.mom[t] * var * (FN.count > 1.f ? FN.count / (FN.count - 1.f) : 1.f);
        }
      }
      bs[h] = sh;
#pragma unroll
      for (int t = 0; t < 25; ++t) {
        const int ky = t / 5, kx = t - ky * 5;
        float v = a4[0] * FN.w5[e * 25 + t];
        if (ky >= 1 && ky <= 3 && kx >= 1 && kx <= 3) v += a4[1] * FN.w3[e * 9 + (ky - 1) * 3 + (kx - 1)];
        if (kx == 2 && ky >= 1 && ky <= 3) v += a4[2] * FN.wv[e * 3 + (ky - 1)];
        if (ky == 2 && kx >= 1 && kx <= 3) v += a4[3] * FN.wh[e * 3 + (kx - 1)];
        wm[h][t] = v;
      }
    }
#pragma unroll
    for (int t = 0; t < 25; ++t) w[t] = f32x2{wm[0][t], wm[1][t]};
    bias = f32x2{bs[0], bs[1]};
  } else {
#pragma unroll
    for (int t = 0; t < 25; ++t) w[t] = f32x2{keff[(int64_t)chs * 25 + t], keff[(int64_t)(chs + 1) * 25 + t]};
    bias = f32x2{beff[chs], beff[chs + 1]};
  }
  const int ys = seg * seg_rows, ye = min(ys + seg_rows, H);
  const int xs = strip * SW_FC;
  const bool cvalid = cok && xs + lane < W;
  const TA* xb = x1 + (int64_t)b * H * W * E;
  TA* ob = pre + (int64_t)b * H * W * E;
  const int nsteps = (ye - ys) + 4;  // x1 row (ys-2+j) enters at step j; output row (ys+j-4) completes

  // Staging geometry is fixed per thread: byte offsets into the image (loads) / the row segment (stores) and LDS
  // indices are computed ONCE; a batch only adds its wave-uniform row offset.  Bounds are the buffer descriptor's job:
  // columns / channels outside the tensor carry an offset beyond num_records for good, rows above the image wrap
  // negative, rows below it (or below the segment, for stores) exceed num_records -- loads return 0, stores are dropped.
  constexpr int NX = (SW_R * SW_XC * 2 + 255) / 256, ND = (SW_R * SW_FC * 2 + 255) / 256;
  constexpr unsigned OOB = 0x80000000u;
  constexpr unsigned ES = sizeof(TA);
  const int rowb = W * E * (int)ES;  // bytes per image row (host: (H+8)*W*E*4 < 2^30)
  const BufRsrc rin = make_rsrc(xb, (unsigned)H * rowb);
  const BufRsrc rout = make_rsrc(ob + (int64_t)ys * W * E, (unsigned)(ye - ys) * rowb);
  unsigned fo[NX], so[ND];
  int li[NX], lo[ND];
  int okrr = 0;  // z-path: per staged item k, bit k = inside the tensor (column, channel), bits 8+4k.. = its row in the batch
#pragma unroll
  for (int k = 0; k < NX; ++k) {
    const int i = tid + k * 256;
    const int k4 = i & 1, pc = i >> 1;
    const int rr = pc / SW_XC, c = pc - rr * SW_XC;
    const int gx = xs - 2 + c;
    const bool ok = i < SW_R * SW_XC * 2 && gx >= 0 && gx < W && ch0 + k4 * 4 < E;
    fo[k] = ok ? (unsigned)(((rr - 2) * W + gx) * E + ch0 + k4 * 4) * ES : OOB;
    li[k] = (rr * SW_XC + c) * SW_CS + k4 * 4;
    okrr |= (ok ? 1 : 0) << k | (rr & 15) << (8 + 4 * k);
  }
#pragma unroll
  for (int k = 0; k < ND; ++k) {
    const int i = tid + k * 256;
    const int k4 = i & 1, pc = i >> 1;
    const int rr = pc / SW_FC, c = pc - rr * SW_FC;
    const bool ok = i < SW_R * SW_FC * 2 && xs + c < W && ch0 + k4 * 4 < E;
    so[k] = ok ? (unsigned)(((rr - 4) * W + xs + c) * E + ch0 + k4 * 4) * ES : OOB;
    lo[k] = (rr * SW_FC + c) * SW_CS + k4 * 4;
  }
  if (zt) __syncthreads();   // pre_s
  f32x4 px[NX];
  auto fetch = [&](int j0) {  // x1 rows ys-2+j0 .. +4
    const unsigned base = (unsigned)((ys + j0) * rowb);
#pragma unroll
    for (int k = 0; k < NX; ++k) px[k] = buf_load4<TA>(rin, fo[k] + base);  // OOB + base stays >= 2^31
  };
  auto commit = [&](int j0) {
    float* ring = XS + (j0 % FS_XR) * (SW_XC * SW_CS);
    const int yb = ys - 2 + j0;                       // image row of the batch's first row
    const bool inner = yb >= 0 && yb + SW_R <= H;     // block-uniform: every row of the batch lies inside the image
#pragma unroll
    for (int k = 0; k < NX; ++k) {
      if (k * 256 + 255 < SW_R * SW_XC * 2 || tid + k * 256 < SW_R * SW_XC * 2) {
        float* d = ring + li[k];
        f32x4 v = px[k];
        if (zt) {
          const f32x4 tA = *reinterpret_cast<const f32x4*>(&pre_s[(tid & 1) * 4]), tS = *reinterpret_cast<const f32x4*>(&pre_s[8 + (tid & 1) * 4]);
          bool ok = (okrr >> k) & 1;
          if (!inner) ok = ok && (unsigned)(yb + ((okrr >> (8 + 4 * k)) & 15)) < (unsigned)H;
          v = dw_pre4(v, tA, tS, ok ? 1.f : 0.f);
        }
        *reinterpret_cast<f32x2*>(d) = f32x2{v[0], v[1]};
        *reinterpret_cast<f32x2*>(d + 2) = f32x2{v[2], v[3]};
      }
    }
  };
  auto drain = [&](int j0) {  // output rows ys+j0-4 .. ys+j0 (rows outside the segment fall off the descriptor)
    const unsigned base = (unsigned)(j0 * rowb);
#pragma unroll
    for (int k = 0; k < ND; ++k) {
      if (k * 256 + 255 < SW_R * SW_FC * 2 || tid + k * 256 < SW_R * SW_FC * 2) {
        const float* o = &OUT[lo[k]];
        const f32x2 a = *reinterpret_cast<const f32x2*>(o), c = *reinterpret_cast<const f32x2*>(o + 2);
        buf_store4<TA>(rout, so[k] + base, f32x4{a[0], a[1], c[0], c[1]});
      }
    }
  };

  f32x2 acc[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) acc[k] = f32x2{0.f, 0.f};
  f32x2 gs = f32x2{0.f, 0.f};
  fetch(0);
  commit(0);
  __syncthreads();
  for (int j0 = 0; j0 < nsteps; j0 += SW_R) {
    if (j0 + SW_R < nsteps) fetch(j0 + SW_R);  // next batch in flight during the five steps
#define LMN_FS_STEP(PH)                                                                         \
    {                                                                                           \
      const int j = j0 + PH;                                                                    \
      if (j < nsteps) fs_step<PH>(acc, gs, w, bias, XS, OUT, j, lane, wv, j >= 4);              \
    }
    LMN_FS_STEP(0) LMN_FS_STEP(1) LMN_FS_STEP(2) LMN_FS_STEP(3) LMN_FS_STEP(4)
#undef LMN_FS_STEP
    __syncthreads();  // OUT complete; the other half of the ring is free
    drain(j0);
    if (j0 + SW_R < nsteps) commit(j0 + SW_R);
    __syncthreads();
  }
  // SE squeeze: wave total by DPP (lands in lane 63), parked in LDS
  {
    float a = cvalid ? gs.x : 0.f, c = cvalid ? gs.y : 0.f;
#define LMN_DPP_ADD(CTRL)                                                                               \
    a += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), CTRL, 0xF, 0xF, true));       \
    c += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(c), CTRL, 0xF, 0xF, true));
    LMN_DPP_ADD(0x111) LMN_DPP_ADD(0x112) LMN_DPP_ADD(0x114) LMN_DPP_ADD(0x118) LMN_DPP_ADD(0x142) LMN_DPP_ADD(0x143)
#undef LMN_DPP_ADD
    if (lane == 63) { gs_s[wv * 2] = a; gs_s[wv * 2 + 1] = c; }
  }
  __syncthreads();  // the block's 8 channel sums leave as ONE atomic instruction (single-lane atomics per wave queue up in L2)
  if (SE.ticket == nullptr) {
    // (deterministic mode: gsum addresses slot copies of [B][E], one slot per (segment, strip))
    if (tid < SW_CH && ch0 + tid < E) lmn_red_add(gsum + (det ? (int64_t)(seg * strips + strip) * (int)(gridDim.x / (unsigned)(strips * segs * chunks)) * E : 0) + (int64_t)b * E + ch0 + tid, gs_s[tid], det);
    return;
  }
  // ---- squeeze-excite gate of image b by the block that completes its sums (lmn_se_fuse_t).  Hand-off: wave 0 adds this
  // block's sums with RETURNING atomics -- a no-return add is acknowledged (vmcnt) before it is performed at the memory side,
  // and under load a later reader saw the sum without it (one image's gate off by 5e-4, once in ~50 steps); the returned value
  // arrives only after the add has been performed.  Then its lane 0 draws a ticket; the block that draws the last one reads
  // the sums back with returning atomics too (performed where the adds were: no cache can hold an older value).
  if (tid < 64) {
    if (tid < SW_CH && ch0 + tid < E) {
      const float old = __hip_atomic_fetch_add(gsum + (int64_t)b * E + ch0 + tid, gs_s[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("" :: "v"(old) : "memory");   // the add has returned: it is in the sum every later reader sees
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0) s_last = atomicAdd(SE.ticket + b, 1u) == (unsigned)(strips * segs * chunks - 1) ? 1 : 0;
  }
  __syncthreads();
  if (!s_last) return;
  float* m = OUT;        // [E]
  float* h = OUT + E;    // [R]   (E + R <= SW_R * SW_FC * SW_CS: checked on the host)
  const int R = SE.R;
  for (int e = tid; e < E; e += 256)
    m[e] = __hip_atomic_fetch_add(gsum + (int64_t)b * E + e, 0.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * SE.inv_hw;
  __syncthreads();
  for (int r = tid; r < R; r += 256) {
    float a = SE.b1[r];
    for (int e = 0; e < E; ++e) a += SE.w1[(int64_t)r * E + e] * m[e];
    a = a > 0.f ? a : 0.f;
    h[r] = a;
    SE.hidden[(int64_t)b * R + r] = a;
  }
  __syncthreads();
  for (int e = tid; e < E; e += 256) {
    float a = SE.b2[e];
    for (int r = 0; r < R; ++r) a += SE.w2[(int64_t)e * R + r] * h[r];
    SE.s[(int64_t)b * E + e] = lmn_hsigmoid(a);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Statistics passes in the strip-walking form (layout of dw_fwd_strip_kernel; the 5x5 kernel of the channel pair in
// SGPRs, the three small kernels in VGPRs -- 80 weight floats do not fit the scalar file):
//   MODE 0: forward batch statistics   stats[4][2][E] += (sum y_b, sum y_b^2)
//   MODE 1: backward pass 1            dpre = (u*s + dm)*gelu'(pre) -> store; stats[5][E] += (sum dpre, sum dpre*y_b)
// Sums stay in registers for the whole segment; wave totals by DPP, parked in LDS, ONE atomic instruction per block
// (a first version issued 16 single-lane atomics per wave: 73 k atomic instructions on 192 addresses = 170 us).
template <int P, int MODE, bool FAST = false>
__device__ __forceinline__ void ss_step(f32x2 (&a5)[5], f32x2 (&a3)[5], f32x2 (&av)[5], f32x2 (&ah)[5],
                                        f32x2 (&sum)[MODE == 0 ? 8 : 5], const BranchW& bw, const float* XS, const float* PS,
                                        const float* US, float* OUT, int j, int lane, int wv, bool row_out, bool valid,
                                        f32x2 sv, f32x2 dv) {
  const float* xr = XS + ((j % FS_XR) * SW_XC + lane) * SW_CS + wv * 2;
  f32x2 in[5];
#pragma unroll
  for (int d = 0; d < 5; ++d) in[d] = *reinterpret_cast<const f32x2*>(xr + d * SW_CS);
  // the first contribution to an output row is a plain product (no zeroed accumulators, no register rotation); column-major:
  // consecutive FMAs go to different accumulators (back-to-back dependent v_pk_fma_f32 cost an s_nop each)
#pragma unroll
  for (int d = 0; d < 5; ++d) {
    if (d == 0) a5[P] = bw.w5[0] * in[0];
    else a5[P] += bw.w5[d] * in[d];
#pragma unroll
    for (int ky = 1; ky < 5; ++ky) a5[(P - ky + 5) % 5] += bw.w5[ky * 5 + d] * in[d];
    if (d >= 1 && d <= 3) {
      if (d == 1) { a3[(P + 4) % 5] = bw.w3[0] * in[1]; ah[(P + 3) % 5] = bw.wh[0] * in[1]; }
      else { a3[(P + 4) % 5] += bw.w3[d - 1] * in[d]; ah[(P + 3) % 5] += bw.wh[d - 1] * in[d]; }
#pragma unroll
      for (int ky = 1; ky < 3; ++ky) a3[(P + 4 - ky) % 5] += bw.w3[ky * 3 + d - 1] * in[d];
    }
    if (d == 2) {
      av[(P + 4) % 5] = bw.wv[0] * in[2];
#pragma unroll
      for (int ky = 1; ky < 3; ++ky) av[(P + 4 - ky) % 5] += bw.wv[ky] * in[2];
    }
  }
  constexpr int D = (P + 1) % 5;  // output row j-4 is complete in every branch
  if (FAST || row_out) {
    const f32x2 y5 = a5[D], y3 = a3[D], yv = av[D], yh = ah[D];
    if (MODE == 0) {  // (columns outside the image are dropped from the lane sums at the end)
      sum[0] += y5; sum[1] += y3; sum[2] += yv; sum[3] += yh;
      sum[4] += y5 * y5; sum[5] += y3 * y3; sum[6] += yv * yv; sum[7] += yh * yh;
    } else {
      const float m = valid ? 1.f : 0.f;
      const int o = (P * SW_FC + lane) * SW_CS + wv * 2;
      const f32x2 pv = *reinterpret_cast<const f32x2*>(PS + o), uv = *reinterpret_cast<const f32x2*>(US + o);
      f32x2 d;
      d[0] = (uv[0] * sv[0] + dv[0]) * lmn_dgelu(pv[0]);
      d[1] = (uv[1] * sv[1] + dv[1]) * lmn_dgelu(pv[1]);
      *reinterpret_cast<f32x2*>(OUT + o) = d;
      d *= m;
      sum[0] += d; sum[1] += d * y5; sum[2] += d * y3; sum[3] += d * yv; sum[4] += d * yh;
    }
  }
}

// MODE 1 without the branch outputs: sum_p dpre[p] * y_b[p] = sum_t w_b[t] * G[t] with G[t] = sum_p dpre[p] * x1[p + t] (the 5x5
// correlation of dpre with x1, of which the 3x3 / 3x1 / 1x3 taps are subsets), so the pass accumulates the 25 G products per
// pixel (25 packed FMAs) instead of the four y_b (40) and their products with dpre (4), and contracts G with the four kernels
// once per block.  dpre depends on pre / u only, so its row o is formed at step o -- when x1 row o+2 (ring row o) arrives -- and
// pairs with x1 ring rows o .. o+4 over the next five steps: G[ky][kx] += dpre[j - ky] * x1ring[j][x + kx - 2] at step j.
template <int P>
__device__ __forceinline__ void ss_step_g(f32x2 (&G)[25], f32x2 (&hist)[5], f32x2& sum0, const float* XS, const float* PS,
                                          const float* US, float* OUT, int j, int lane, int wv, bool row_ok, bool valid,
                                          f32x2 sv, f32x2 dv) {
  const float* xr = XS + ((j % FS_XR) * SW_XC + lane) * SW_CS + wv * 2;
  f32x2 in[5];
#pragma unroll
  for (int d = 0; d < 5; ++d) in[d] = *reinterpret_cast<const f32x2*>(xr + d * SW_CS);
  const int o = (P * SW_FC + lane) * SW_CS + wv * 2;
  const f32x2 pv = *reinterpret_cast<const f32x2*>(PS + o), uv = *reinterpret_cast<const f32x2*>(US + o);
  f32x2 d;
  d[0] = (uv[0] * sv[0] + dv[0]) * lmn_dgelu(pv[0]);
  d[1] = (uv[1] * sv[1] + dv[1]) * lmn_dgelu(pv[1]);
  *reinterpret_cast<f32x2*>(OUT + o) = d;            // rows past the segment are dropped by the store descriptor
  d *= (valid && row_ok) ? 1.f : 0.f;
  sum0 += d;
  hist[P] = d;
#pragma unroll
  for (int ky = 0; ky < 5; ++ky)
#pragma unroll
    for (int kx = 0; kx < 5; ++kx) G[ky * 5 + kx] += hist[(P - ky + 5) % 5] * in[kx];
}

template <int MODE, typename TA>
__global__ __launch_bounds__(256) void dw_stats_strip_kernel(
    const TA* __restrict__ x1, const TA* __restrict__ pre, const TA* __restrict__ u,
    const float* __restrict__ sgate, const float* __restrict__ dm, TA* __restrict__ dpre, int H, int W, int E,
    const float* __restrict__ w5, const float* __restrict__ w3, const float* __restrict__ wvv,
    const float* __restrict__ whh, float* __restrict__ stats, const lmn_se_bwd_t SB, const DwPreK PRE, int strips, int segs,
    int seg_rows, int chunks, int det) {
  constexpr int NS = MODE == 0 ? 8 : 5;
  __shared__ __attribute__((aligned(16))) float pre_s[16];
  constexpr int NAUX = MODE == 1 ? SW_R * SW_FC * SW_CS : 4;
  __shared__ __attribute__((aligned(16))) float XS[FS_XR * SW_XC * SW_CS];
  __shared__ __attribute__((aligned(16))) float PS[NAUX], US[NAUX];
  // dpre leaves through the slot its `pre` element came in by: a row step reads PS[o] before it writes OUT[o], and the drain reads an
  // item before the commit of the next batch overwrites it (same thread, same index both times) -- 12.8 KB less LDS: three blocks
  // per CU instead of two
  float* const OUT = PS;
  __shared__ float red[NS * SW_CH];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  int lid = xcd_swizzle(blockIdx.x, gridDim.x);
  const int chunk = lid % chunks; lid /= chunks;
  const int strip = lid % strips; lid /= strips;
  const int seg = lid % segs;
  const int b = lid / segs;
  const int ch0 = chunk * SW_CH, ch = ch0 + wv * 2;
  const bool cok = ch < E;
  const int chs = cok ? ch : 0;
  const bool zt = PRE.A != nullptr || PRE.fin.mode == LMN_FIN_BN;   // block-uniform: the input tensor is z (see DwPreK)
  if (zt) dw_pre_setup(PRE, pre_s, ch0, E, tid, strip == 0 && seg == 0 && b == 0);
  BranchW bw;
  if constexpr (MODE == 0) {   // (MODE 1 needs the kernels only for the final contraction: loaded there)
    load_branch_w(bw, w5, w3, wvv, whh, chs, E);
#pragma unroll
    for (int k = 0; k < 9; ++k) asm volatile("" : "+v"(bw.w3[k].x), "+v"(bw.w3[k].y));
#pragma unroll
    for (int k = 0; k < 3; ++k) asm volatile("" : "+v"(bw.wv[k].x), "+v"(bw.wv[k].y), "+v"(bw.wh[k].x), "+v"(bw.wh[k].y));
  }
  f32x2 sv = f32x2{0.f, 0.f}, dv = f32x2{0.f, 0.f};
  if (MODE == 1) {
    sv = f32x2{sgate[(int64_t)b * E + chs], sgate[(int64_t)b * E + chs + 1]};
    if (SB.ds) {
      // squeeze-excite backward of image b (lmn_se_bwd_t, lmn_se_bwd_dm arithmetic), formed by every block for its own channels:
      //   dt[e] = ds[b][e] * hardsigmoid'(.) (1/6 where 0 < s < 1);  da[r] = relu'(h[r]) * sum_e w2[e][r] * dt[e];
      //   dm[e] = inv_hw * sum_r w1[r][e] * da[r]
      const int R = SB.R;
      float* dt = XS;            // [E]      (the staging ring is not in use yet)
      float* da = XS + E;        // [R]
      float* part = XS + E + R;  // [256]
      const bool writer = chunk == 0 && strip == 0 && seg == 0;   // one block per image keeps dt / da for lmn_se_bwd_params
      for (int e = tid; e < E; e += 256) {
        const float g = sgate[(int64_t)b * E + e];
        const float d = (g > 0.f && g < 1.f) ? SB.ds[(int64_t)b * E + e] * (1.f / 6.f) : 0.f;
        dt[e] = d;
        if (writer) SB.dvec[(int64_t)b * (E + R) + e] = d;
      }
      __syncthreads();
      const int groups = R <= 256 ? 256 / R : 1;
      for (int r0 = 0; r0 < R; r0 += 256) {
        const int g = tid / R, r = r0 + (R <= 256 ? tid - g * R : tid);
        float a = 0.f;
        if (g < groups && r < R)
          for (int e = g; e < E; e += groups) a += SB.w2[(int64_t)e * R + r] * dt[e];
        part[tid] = a;
        __syncthreads();
        if (tid < R - r0 && tid < 256) {
          float v = 0.f;
          if (R <= 256) { for (int k = 0; k < groups; ++k) v += part[k * R + tid]; }
          else v = part[tid];
          const int rr = r0 + tid;
          v = SB.hidden[(int64_t)b * R + rr] > 0.f ? v : 0.f;
          da[rr] = v;
          if (writer) SB.dvec[(int64_t)b * (E + R) + E + rr] = v;
        }
        __syncthreads();
      }
      float d2[2] = {0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        float a = 0.f;
        for (int r = 0; r < R; ++r) a += SB.w1[(int64_t)r * E + chs + k] * da[r];
        d2[k] = a * SB.inv_hw;
      }
      dv = f32x2{d2[0], d2[1]};
      __syncthreads();           // dt / da are consumed: the ring may be staged
    } else {
      dv = f32x2{dm[(int64_t)b * E + chs], dm[(int64_t)b * E + chs + 1]};
    }
  }
  const int ys = seg * seg_rows, ye = min(ys + seg_rows, H);
  const int xs = strip * SW_FC;
  const bool cvalid = cok && xs + lane < W;
  const int64_t ib = (int64_t)b * H * W * E;
  const TA* xb = x1 + ib;
  const int nsteps = (ye - ys) + 4;  // x1 row (ys-2+j) enters at step j; output row (ys+j-4) completes

  // staging geometry fixed per thread, bounds by buffer descriptors (see dw_fwd_strip_kernel)
  constexpr int NX = (SW_R * SW_XC * 2 + 255) / 256, ND = (SW_R * SW_FC * 2 + 255) / 256;
  constexpr unsigned OOB = 0x80000000u;
  constexpr unsigned ES = sizeof(TA);
  const int rowb = W * E * (int)ES;
  const int64_t sb = ib + (int64_t)ys * W * E;  // first element of the row segment
  const BufRsrc rin = make_rsrc(xb, (unsigned)H * rowb);
  const BufRsrc rpre = make_rsrc(MODE == 1 ? pre + sb : xb, MODE == 1 ? (unsigned)(ye - ys) * rowb : 0u);
  const BufRsrc ru = make_rsrc(MODE == 1 ? u + sb : xb, MODE == 1 ? (unsigned)(ye - ys) * rowb : 0u);
  const BufRsrc rout = make_rsrc(MODE == 1 ? dpre + sb : xb, MODE == 1 ? (unsigned)(ye - ys) * rowb : 0u);
  unsigned fo[NX], so[MODE == 1 ? ND : 1];
  int li[NX], lo[MODE == 1 ? ND : 1];
  int okrr = 0;  // z-path: per staged item k, bit k = inside the tensor (column, channel), bits 8+4k.. = its row in the batch
#pragma unroll
  for (int k = 0; k < NX; ++k) {
    const int i = tid + k * 256;
    const int k4 = i & 1, pc = i >> 1;
    const int rr = pc / SW_XC, c = pc - rr * SW_XC;
    const int gx = xs - 2 + c;
    const bool ok = i < SW_R * SW_XC * 2 && gx >= 0 && gx < W && ch0 + k4 * 4 < E;
    fo[k] = ok ? (unsigned)(((rr - 2) * W + gx) * E + ch0 + k4 * 4) * ES : OOB;
    li[k] = (rr * SW_XC + c) * SW_CS + k4 * 4;
    okrr |= (ok ? 1 : 0) << k | (rr & 15) << (8 + 4 * k);
  }
  if (zt) __syncthreads();   // pre_s
  if (MODE == 1) {
#pragma unroll
    for (int k = 0; k < ND; ++k) {
      const int i = tid + k * 256;
      const int k4 = i & 1, pc = i >> 1;
      const int rr = pc / SW_FC, c = pc - rr * SW_FC;
      const bool ok = i < SW_R * SW_FC * 2 && xs + c < W && ch0 + k4 * 4 < E;
      so[k] = ok ? (unsigned)((rr * W + xs + c) * E + ch0 + k4 * 4) * ES : OOB;   // output row j0 + rr of the batch at step j0
      lo[k] = (rr * SW_FC + c) * SW_CS + k4 * 4;
    }
  }
  f32x4 px[NX], pp[MODE == 1 ? ND : 1], pu[MODE == 1 ? ND : 1];
  auto fetch = [&](int j0) {
    const unsigned base = (unsigned)((ys + j0) * rowb);
#pragma unroll
    for (int k = 0; k < NX; ++k) px[k] = buf_load4<TA>(rin, fo[k] + base);
    if (MODE == 1) {  // pre / u rows of the batch: segment rows j0+rr (past the segment: 0 by the descriptor)
      const unsigned sbase = (unsigned)(j0 * rowb);
#pragma unroll
      for (int k = 0; k < ND; ++k) {
        pp[k] = buf_load4<TA>(rpre, so[k] + sbase);
        pu[k] = buf_load4<TA>(ru, so[k] + sbase);
      }
    }
  };
  auto commit = [&](int j0) {
    float* ring = XS + (j0 % FS_XR) * (SW_XC * SW_CS);
    const int yb = ys - 2 + j0;                       // image row of the batch's first row
    const bool inner = yb >= 0 && yb + SW_R <= H;     // block-uniform: every row of the batch lies inside the image
#pragma unroll
    for (int k = 0; k < NX; ++k) {
      if (k * 256 + 255 < SW_R * SW_XC * 2 || tid + k * 256 < SW_R * SW_XC * 2) {
        float* d = ring + li[k];
        f32x4 v = px[k];
        if (zt) {
          const f32x4 tA = *reinterpret_cast<const f32x4*>(&pre_s[(tid & 1) * 4]), tS = *reinterpret_cast<const f32x4*>(&pre_s[8 + (tid & 1) * 4]);
          bool ok = (okrr >> k) & 1;
          if (!inner) ok = ok && (unsigned)(yb + ((okrr >> (8 + 4 * k)) & 15)) < (unsigned)H;
          v = dw_pre4(v, tA, tS, ok ? 1.f : 0.f);
        }
        *reinterpret_cast<f32x2*>(d) = f32x2{v[0], v[1]};
        *reinterpret_cast<f32x2*>(d + 2) = f32x2{v[2], v[3]};
      }
    }
    if (MODE == 1) {
#pragma unroll
      for (int k = 0; k < ND; ++k) {
        if (k * 256 + 255 < SW_R * SW_FC * 2 || tid + k * 256 < SW_R * SW_FC * 2) {
          const int o = lo[k];
          *reinterpret_cast<f32x2*>(&PS[o]) = f32x2{pp[k][0], pp[k][1]};
          *reinterpret_cast<f32x2*>(&PS[o + 2]) = f32x2{pp[k][2], pp[k][3]};
          *reinterpret_cast<f32x2*>(&US[o]) = f32x2{pu[k][0], pu[k][1]};
          *reinterpret_cast<f32x2*>(&US[o + 2]) = f32x2{pu[k][2], pu[k][3]};
        }
      }
    }
  };
  auto drain = [&](int j0) {  // MODE 1: dpre rows formed by steps j0 .. j0+4 (segment rows j0+rr)
    const unsigned sbase = (unsigned)(j0 * rowb);
#pragma unroll
    for (int k = 0; k < (MODE == 1 ? ND : 0); ++k) {
      if (k * 256 + 255 < SW_R * SW_FC * 2 || tid + k * 256 < SW_R * SW_FC * 2) {
        const float* o = &OUT[lo[k]];
        const f32x2 a = *reinterpret_cast<const f32x2*>(o), c = *reinterpret_cast<const f32x2*>(o + 2);
        buf_store4<TA>(rout, so[k] + sbase, f32x4{a[0], a[1], c[0], c[1]});
      }
    }
  };

  const f32x2 z2 = f32x2{0.f, 0.f};
  f32x2 a5[5], a3[5], av[5], ah[5], sum[NS];
  f32x2 G[MODE == 1 ? 25 : 1], hist[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) a5[k] = a3[k] = av[k] = ah[k] = hist[k] = z2;
#pragma unroll
  for (int k = 0; k < (MODE == 1 ? 25 : 1); ++k) G[k] = z2;
#pragma unroll
  for (int k = 0; k < NS; ++k) sum[k] = z2;
  fetch(0);
  commit(0);
  __syncthreads();
  for (int j0 = 0; j0 < nsteps; j0 += SW_R) {
    if (j0 + SW_R < nsteps) fetch(j0 + SW_R);
#define LMN_SS_STEP(PH)                                                                                          \
    {                                                                                                            \
      const int j = j0 + PH;                                                                                     \
      if (j < nsteps) {                                                                                          \
        if constexpr (MODE == 1) ss_step_g<PH>(G, hist, sum[0], XS, PS, US, OUT, j, lane, wv, j < ye - ys, cvalid, sv, dv); \
        else ss_step<PH, MODE>(a5, a3, av, ah, sum, bw, XS, PS, US, OUT, j, lane, wv, j >= 4, cvalid, sv, dv);   \
      }                                                                                                          \
    }
    LMN_SS_STEP(0) LMN_SS_STEP(1) LMN_SS_STEP(2) LMN_SS_STEP(3) LMN_SS_STEP(4)
#undef LMN_SS_STEP
    __syncthreads();
    if (MODE == 1) drain(j0);
    if (j0 + SW_R < nsteps) commit(j0 + SW_R);
    __syncthreads();
  }
  if constexpr (MODE == 1) {  // sum dpre * y_b = <w_b, G> (taps of the small kernels embedded in the 5x5 window as in dw_merge)
    load_branch_w(bw, w5, w3, wvv, whh, chs, E);
#pragma unroll
    for (int t = 0; t < 25; ++t) sum[1] += bw.w5[t] * G[t];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) sum[2] += bw.w3[ky * 3 + kx] * G[(ky + 1) * 5 + kx + 1];
      sum[3] += bw.wv[ky] * G[(ky + 1) * 5 + 2];
      sum[4] += bw.wh[ky] * G[2 * 5 + ky + 1];
    }
  }
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    float a = sum[k].x, c = sum[k].y;
    if (MODE == 0 && !cvalid) a = c = 0.f;
#define LMN_DPP_ADD(CTRL)                                                                               \
    a += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), CTRL, 0xF, 0xF, true));       \
    c += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(c), CTRL, 0xF, 0xF, true));
    LMN_DPP_ADD(0x111) LMN_DPP_ADD(0x112) LMN_DPP_ADD(0x114) LMN_DPP_ADD(0x118) LMN_DPP_ADD(0x142) LMN_DPP_ADD(0x143)
#undef LMN_DPP_ADD
    if (lane == 63) { red[k * SW_CH + wv * 2] = a; red[k * SW_CH + wv * 2 + 1] = c; }
  }
  __syncthreads();
  if (tid < NS * SW_CH) {
    const int k = tid / SW_CH, cc = tid - k * SW_CH;
    const int row = MODE == 0 ? ((k & 3) * 2 + (k >> 2)) : k;  // MODE 0: [branch][sum|sumsq][E]
    // (deterministic mode: stats addresses slot copies of [NS][E], one slot per (image, segment, strip))
    if (ch0 + cc < E) lmn_red_add(stats + (det ? (int64_t)((b * segs + seg) * strips + strip) * NS * E : 0) + (int64_t)row * E + ch0 + cc, red[tid], det);
  }
}

// Training forward, between the statistics pass and lmn_dw_fwd: finalise the four branch BatchNorms from their
// batch sums (mean, rstd, A = gamma*rstd, running statistics with momentum -- nn.BatchNorm2d semantics, unbiased
// running variance) AND merge the branches into the effective 5x5 kernel, in ONE launch instead of 4 + 1.
// One block per channel: threads 0..3 finalise branch b, then 25 threads write the taps.
struct DwBnPtrs {
  const float* gamma[4];
  const float* beta[4];
  float* rmean[4];
  float* rvar[4];
};

__global__ __launch_bounds__(64) void dw_finalize_merge_kernel(
    const float* __restrict__ stats, float count, const DwBnPtrs bn, float eps0, float eps1, float eps2, float eps3,
    float mom0, float mom1, float mom2, float mom3, const float* __restrict__ w5, const float* __restrict__ w3,
    const float* __restrict__ wv, const float* __restrict__ wh, float* __restrict__ mean, float* __restrict__ rstd,
    float* __restrict__ A, float* __restrict__ keff, float* __restrict__ beff, int E) {
  __shared__ float sA[4], sS[4];
  const int e = blockIdx.x, t = threadIdx.x;
  if (t < 4) {
    const float eps = t == 0 ? eps0 : t == 1 ? eps1 : t == 2 ? eps2 : eps3;
    const float mom = t == 0 ? mom0 : t == 1 ? mom1 : t == 2 ? mom2 : mom3;
    const float m = stats[(t * 2) * E + e] / count;
    float var = stats[(t * 2 + 1) * E + e] / count - m * m;  // biased
    var = var > 0.f ? var : 0.f;
    const float rs = rsqrtf(var + eps);
    const float a = bn.gamma[t][e] * rs;
    mean[t * E + e] = m;
    rstd[t * E + e] = rs;
    A[t * E + e] = a;
    sA[t] = a;
    sS[t] = bn.beta[t][e] - m * a;
    bn.rmean[t][e] = (1.f - mom) * bn.rmean[t][e] + mom * m;
    bn.rvar[t][e] = (1.f - mom) * bn.rvar[t][e] + mom * var * (count > 1.f ? count / (count - 1.f) : 1.f);
  }
  __syncthreads();
  if (t < 25) {
    const int ky = t / 5, kx = t - ky * 5;
    float v = sA[0] * w5[e * 25 + t];
    if (ky >= 1 && ky <= 3 && kx >= 1 && kx <= 3) v += sA[1] * w3[e * 9 + (ky - 1) * 3 + (kx - 1)];
    if (kx == 2 && ky >= 1 && ky <= 3) v += sA[2] * wv[e * 3 + (ky - 1)];
    if (ky == 2 && kx >= 1 && kx <= 3) v += sA[3] * wh[e * 3 + (kx - 1)];
    keff[e * 25 + t] = v;
    if (t == 0) beff[e] = sS[0] + sS[1] + sS[2] + sS[3];
  }
}

// per-branch BN-backward coefficients from bst[5][E] = (S0 = sum dpre, S1_b = sum dpre*y_b)
__global__ void dw_bwd_coef_kernel(const float* __restrict__ bst, const float* __restrict__ mean,
                                   const float* __restrict__ rstd, const float* __restrict__ A, float count,
                                   int batch_stats, float* __restrict__ cA, float* __restrict__ cC,
                                   float* __restrict__ cD, float* dg0, float* dg1, float* dg2, float* dg3, float* db0,
                                   float* db1, float* db2, float* db3, int E) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 4 * E) return;
  const int b = i / E, e = i - b * E;
  float* dgs[4] = {dg0, dg1, dg2, dg3};
  float* dbs[4] = {db0, db1, db2, db3};
  const float S0 = bst[e], S1 = bst[(1 + b) * E + e];
  const float T = (S1 - mean[i] * S0) * rstd[i];  // sum dpre * yhat_b
  float* dg = b == 0 ? dgs[0] : (b == 1 ? dgs[1] : (b == 2 ? dgs[2] : dgs[3]));
  float* db = b == 0 ? dbs[0] : (b == 1 ? dbs[1] : (b == 2 ? dbs[2] : dbs[3]));
  dg[e] += T;
  db[e] += S0;
  const float a = A[i];
  const float c = batch_stats ? -a * T * rstd[i] / count : 0.f;
  cA[i] = a;
  cC[i] = c;
  cD[i] = batch_stats ? (-a * S0 / count - c * mean[i]) : 0.f;
}

}  // namespace

// Row segments of the strip kernels: every segment re-walks `halo` extra rows, and the grid runs in rounds of
// occ blocks per CU x 256 CUs -- pick the segment count that minimises rounds x (rows + halo) (e.g. level 0 of the
// backward kernel: 8 segments = 3 rounds x 54 row steps, 7 segments = 2 rounds x 61).
static int strip_segments(int64_t blocks_per_seg, int H, int halo, int occ, int* seg_rows) {
  int best = 1;
  int64_t best_cost = -1;
  for (int sg = 1; sg <= H; ++sg) {
    const int rows = lmn_cdiv(H, sg);
    if (sg > 1 && rows < 8) break;
    const int nseg = lmn_cdiv(H, rows);
    if (nseg != sg) continue;  // same partition as a smaller count
    const int64_t rounds = (blocks_per_seg * nseg + (int64_t)occ * 256 - 1) / ((int64_t)occ * 256);
    const int64_t cost = rounds * (rows + halo);
    if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = sg; }
  }
  *seg_rows = lmn_cdiv(H, best);
  return lmn_cdiv(H, *seg_rows);
}

template <int MODE>
static int launch_dw_strip_stats(const void* x1, const void* pre, const void* u, const float* s, const float* dm,
                                 void* dpre, int B, int H, int W, int E, const float* w5, const float* w3,
                                 const float* wv, const float* wh, float* stats, const lmn_se_bwd_t& sb, const lmn_dw_pre_t& zp,
                                 int act_dtype, hipStream_t st) {
  LMN_REQUIRE((int64_t)(H + 8) * W * E * 4 < (1LL << 30), "dw statistics: one image (%d x %d x %d) must stay below 1 GiB", H, W, E);
  const int strips = lmn_cdiv(W, SW_FC), chunks = lmn_cdiv(E, SW_CH);
  int seg_rows;
  const int segs = strip_segments((int64_t)B * strips * chunks, H, 4, 3, &seg_rows);   // (both modes: three blocks per CU)
  const int64_t nblk = (int64_t)B * strips * chunks * segs;
  if (nblk >= (1LL << 31)) return -1;
  if (g_lmn_prof_on) lmn_prof_cost(2.0 * 42 * (double)B * H * W * E, (act_dtype == LMN_BF16 ? 2.0 : 4.0) * (MODE == 0 ? 1 : 4) * (double)B * H * W * E);
  float* sdst = stats;
  const int nslots = B * segs * strips, NS = MODE == 0 ? 8 : 5;
  if (g_lmn_det) {   // per-block sums into slot copies of [NS][E], folded in fixed order below
    lmn_det_begin(st);
    sdst = lmn_det_slots(st, (size_t)nslots * NS * E);
    LMN_REQUIRE(sdst, "dw statistics: deterministic mode: no scratch");
  }
  LMN_ACT_DISPATCH(act_dtype, LMN_LAUNCH((dw_stats_strip_kernel<MODE, T>), dim3((unsigned)nblk), dim3(256), 0, st, (const T*)x1, (const T*)pre, (const T*)u, s, dm, (T*)dpre, H, W, E,
                     w5, w3, wv, wh, sdst, sb, zp, strips, segs, seg_rows, chunks, g_lmn_det));
  if (g_lmn_det) lmn_det_sum(st, sdst, nslots, (int64_t)NS * E, stats);
  return 0;
}


extern "C" {

#ifdef LMN_DW_TIMING
int lmn_dw_timing(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dw_timing), sizeof(unsigned long long) * n);
}
#endif

static int dw_pre_check(const lmn_dw_pre_t* pre, lmn_dw_pre_t* out, bool allow_fin, const char* what) {
  memset(out, 0, sizeof(*out));
  if (!pre || (!pre->A && pre->fin.mode == LMN_FIN_NONE)) return 0;
  if (pre->fin.mode != LMN_FIN_NONE) {
    const lmn_bn_fin_t& F = pre->fin;
    LMN_REQUIRE(allow_fin && F.mode == LMN_FIN_BN, "%s: the expand conv's BatchNorm can only be finalised by lmn_dw_stats", what);
    LMN_REQUIRE(F.sums && F.nrep >= 1 && F.nrep <= 64 && F.count > 0.f && F.gamma && F.beta && (!F.about || F.about != F.rmean),
                "%s: fin needs sums, slices, count, gamma / beta (and `about` must not alias rmean)", what);
  } else {
    LMN_REQUIRE(pre->A && pre->shift, "%s: z-path needs A and shift", what);
  }
  *out = *pre;
  return 0;
}

static int se_fuse_check(const lmn_se_fuse_t* se, lmn_se_fuse_t* out, int E, const char* what) {
  memset(out, 0, sizeof(*out));
  if (!se || !se->ticket) return 0;
  LMN_REQUIRE(se->w1 && se->b1 && se->w2 && se->b2 && se->s && se->hidden && se->R > 0 && se->inv_hw > 0.f, "%s: squeeze-excite operands", what);
  LMN_REQUIRE(E + se->R <= SW_R * SW_FC * SW_CS, "%s: E + R = %d exceeds the block's scratch", what, E + se->R);
  *out = *se;
  return 0;
}

int lmn_dw_fwd(const void* x1, void* pre, float* gsum, int B, int H, int W, int E, const float* keff,
               const float* beff, const lmn_se_fuse_t* se, const lmn_dw_pre_t* zpre, int act_dtype, lmn_stream_t stream) {
  lmn_se_fuse_t sf;
  { const int rc = se_fuse_check(se, &sf, E, "dw_fwd"); if (rc) return rc; }
  lmn_dw_pre_t zp;
  { const int rc = dw_pre_check(zpre, &zp, false, "dw_fwd"); if (rc) return rc; }
  if (g_lmn_rec) lmn_rec_push([=]() -> int { return lmn_dw_fwd(x1, pre, gsum, B, H, W, E, keff, beff, &sf, &zp, act_dtype, stream); }, "lmn_dw_fwd(");
  LMN_REQUIRE_DT(act_dtype, "dw_fwd");
  LMN_REQUIRE(x1 && pre && gsum && keff && beff, "dw_fwd: null pointer");
  LMN_REQUIRE(B > 0 && H > 0 && W > 0 && E > 0 && E % 4 == 0, "dw_fwd: E=%d must be a multiple of 4", E);
  LMN_REQUIRE((int64_t)(H + 8) * W * E * 4 < (1LL << 30), "dw_fwd: one image (%d x %d x %d) must stay below 1 GiB", H, W, E);
  // strip-walking kernel: blocks = B x strips(64 columns) x row segments x 8-channel chunks
  const int strips = lmn_cdiv(W, SW_FC), chunks = lmn_cdiv(E, SW_CH);
  int seg_rows;
  const int segs = strip_segments((int64_t)B * strips * chunks, H, 4, 4, &seg_rows);   // 40 KB LDS: 4 blocks per CU
  const int64_t nblk = (int64_t)B * strips * chunks * segs;
  LMN_REQUIRE(nblk < (1LL << 31), "dw_fwd: grid too large");
  if (g_lmn_prof_on) lmn_prof_cost(2.0 * 25 * (double)B * H * W * E, (act_dtype == LMN_BF16 ? 2.0 : 4.0) * 2 * (double)B * H * W * E);
  LMN_REQUIRE(!g_lmn_det || !sf.ticket, "dw_fwd: deterministic mode takes the squeeze-excite sums without the fused gate (lmn_se_fuse_t NULL)");
  float* gdst = gsum;
  if (g_lmn_det) {   // per-block sums into slot copies of [B][E]
    lmn_det_begin((hipStream_t)stream);
    gdst = lmn_det_slots((hipStream_t)stream, (size_t)segs * strips * B * E);
    LMN_REQUIRE(gdst, "dw_fwd: deterministic mode: no scratch");
  }
  LMN_ACT_DISPATCH(act_dtype, LMN_LAUNCH((dw_fwd_strip_kernel<T>), dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, (const T*)x1, (T*)pre, gdst, H, W, E, keff,
                     beff, DwFin{}, sf, DwPreS{zp.A, zp.shift}, strips, segs, seg_rows, chunks, g_lmn_det));
  if (g_lmn_det) lmn_det_sum((hipStream_t)stream, gdst, segs * strips, (int64_t)B * E, gsum);
  return lmn_launch_status("dw_fwd");
}

int lmn_dw_fwd_bn(const void* x1, void* pre, float* gsum, int B, int H, int W, int E, const float* stats, float count,
                  const float* const* gamma, const float* const* beta, float* const* running_mean, float* const* running_var,
                  const float* eps, const float* momentum, const float* w5, const float* w3, const float* wv, const float* wh,
                  float* mean, float* rstd, float* A, const lmn_se_fuse_t* se, const lmn_dw_pre_t* zpre, int act_dtype,
                  lmn_stream_t stream) {
  LMN_REQUIRE_DT(act_dtype, "dw_fwd_bn");
  lmn_se_fuse_t sf;
  { const int rc = se_fuse_check(se, &sf, E, "dw_fwd_bn"); if (rc) return rc; }
  lmn_dw_pre_t zp;
  { const int rc = dw_pre_check(zpre, &zp, false, "dw_fwd_bn"); if (rc) return rc; }
  LMN_REQUIRE(x1 && pre && gsum && stats && gamma && beta && running_mean && running_var && eps && momentum && w5 && w3 && wv && wh &&
                  mean && rstd && A && count > 0.f, "dw_fwd_bn: bad argument");
  LMN_REQUIRE(B > 0 && H > 0 && W > 0 && E > 0 && E % 4 == 0, "dw_fwd_bn: E=%d must be a multiple of 4", E);
  LMN_REQUIRE((int64_t)(H + 8) * W * E * 4 < (1LL << 30), "dw_fwd_bn: one image (%d x %d x %d) must stay below 1 GiB", H, W, E);
  DwFin fn;
  fn.stats = stats; fn.count = count;
  for (int b = 0; b < 4; ++b) {
    LMN_REQUIRE(gamma[b] && beta[b] && running_mean[b] && running_var[b], "dw_fwd_bn: null BatchNorm tensor %d", b);
    fn.gamma[b] = gamma[b]; fn.beta[b] = beta[b]; fn.rmean[b] = running_mean[b]; fn.rvar[b] = running_var[b];
    fn.eps[b] = eps[b]; fn.mom[b] = momentum[b];
  }
  fn.w5 = w5; fn.w3 = w3; fn.wv = wv; fn.wh = wh; fn.mean = mean; fn.rstd = rstd; fn.A = A;
  const int strips = lmn_cdiv(W, SW_FC), chunks = lmn_cdiv(E, SW_CH);
  int seg_rows;
  const int segs = strip_segments((int64_t)B * strips * chunks, H, 4, 4, &seg_rows);
  const int64_t nblk = (int64_t)B * strips * chunks * segs;
  LMN_REQUIRE(nblk < (1LL << 31), "dw_fwd_bn: grid too large");
  auto launch = [=]() -> int {
    if (g_lmn_prof_on) lmn_prof_cost(2.0 * 25 * (double)B * H * W * E, (act_dtype == LMN_BF16 ? 2.0 : 4.0) * 2 * (double)B * H * W * E);
    LMN_REQUIRE(!g_lmn_det || !sf.ticket, "dw_fwd_bn: deterministic mode takes the squeeze-excite sums without the fused gate (lmn_se_fuse_t NULL)");
    float* gdst = gsum;
    if (g_lmn_det) {   // per-block sums into slot copies of [B][E]
      lmn_det_begin((hipStream_t)stream);
      gdst = lmn_det_slots((hipStream_t)stream, (size_t)segs * strips * B * E);
      LMN_REQUIRE(gdst, "dw_fwd_bn: deterministic mode: no scratch");
    }
    LMN_ACT_DISPATCH(act_dtype, LMN_LAUNCH((dw_fwd_strip_kernel<T>), dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, (const T*)x1, (T*)pre, gdst, H, W, E,
                       (const float*)nullptr, (const float*)nullptr, fn, sf, DwPreS{zp.A, zp.shift}, strips, segs, seg_rows, chunks, g_lmn_det));
    if (g_lmn_det) lmn_det_sum((hipStream_t)stream, gdst, segs * strips, (int64_t)B * E, gsum);
    return lmn_launch_status("dw_fwd_bn");
  };
  if (g_lmn_rec) lmn_rec_push(launch, "lmn_dw_fwd_bn(");
  return launch();
}

int lmn_dw_finalize_merge(const float* stats, float count, const float* const* gamma, const float* const* beta,
                          float* const* running_mean, float* const* running_var, const float* eps, const float* momentum,
                          const float* w5, const float* w3, const float* wv, const float* wh, float* mean, float* rstd,
                          float* A, float* keff, float* beff, int E, lmn_stream_t stream) {
  LMN_REQUIRE(stats && gamma && beta && running_mean && running_var && eps && momentum && w5 && w3 && wv && wh && mean &&
                  rstd && A && keff && beff && E > 0 && count > 0.f,
              "dw_finalize_merge: bad argument");
  if (g_lmn_rec) {
    struct Arr { const float* g[4]; const float* b[4]; float* rm[4]; float* rv[4]; float eps[4]; float mom[4]; } a;
    for (int b = 0; b < 4; ++b) {
      a.g[b] = gamma[b]; a.b[b] = beta[b]; a.rm[b] = running_mean[b]; a.rv[b] = running_var[b];
      a.eps[b] = eps[b]; a.mom[b] = momentum[b];
    }
    lmn_rec_push([=]() -> int {
      return lmn_dw_finalize_merge(stats, count, a.g, a.b, a.rm, a.rv, a.eps, a.mom, w5, w3, wv, wh, mean, rstd, A, keff, beff, E, stream);
    }, "lmn_dw_finalize_merge(");
  }
  DwBnPtrs bn;
  for (int b = 0; b < 4; ++b) {
    LMN_REQUIRE(gamma[b] && beta[b] && running_mean[b] && running_var[b], "dw_finalize_merge: null BatchNorm tensor %d", b);
    bn.gamma[b] = gamma[b]; bn.beta[b] = beta[b]; bn.rmean[b] = running_mean[b]; bn.rvar[b] = running_var[b];
  }
  LMN_LAUNCH(dw_finalize_merge_kernel, dim3(E), dim3(64), 0, (hipStream_t)stream, stats, count, bn, eps[0], eps[1],
                     eps[2], eps[3], momentum[0], momentum[1], momentum[2], momentum[3], w5, w3, wv, wh, mean, rstd, A, keff,
                     beff, E);
  return lmn_launch_status("dw_finalize_merge");
}

int lmn_dw_merge(const float* w5, const float* w3, const float* wv, const float* wh, const float* A, const float* shift,
                 float* keff, float* beff, int E, lmn_stream_t stream) {
  LMN_REC(lmn_dw_merge(w5, w3, wv, wh, A, shift, keff, beff, E, stream));
  LMN_REQUIRE(w5 && w3 && wv && wh && A && shift && keff && beff && E > 0, "dw_merge: bad argument");
  LMN_LAUNCH(dw_merge_kernel, dim3(lmn_cdiv(E * 25, 256)), dim3(256), 0, (hipStream_t)stream, w5, w3, wv, wh,
                     A, shift, keff, beff, E);
  return lmn_launch_status("dw_merge");
}

int lmn_dw_stats(const void* x1, int B, int H, int W, int E, const float* w5, const float* w3, const float* wv,
                 const float* wh, float* stats, const lmn_dw_pre_t* zpre, int act_dtype, lmn_stream_t stream) {
  lmn_dw_pre_t zp;
  { const int rc = dw_pre_check(zpre, &zp, true, "dw_stats"); if (rc) return rc; }
  if (g_lmn_rec) lmn_rec_push([=]() -> int { return lmn_dw_stats(x1, B, H, W, E, w5, w3, wv, wh, stats, &zp, act_dtype, stream); }, "lmn_dw_stats(");
  LMN_REQUIRE_DT(act_dtype, "dw_stats");
  LMN_REQUIRE(x1 && w5 && w3 && wv && wh && stats, "dw_stats: null pointer");
  LMN_REQUIRE(B > 0 && H > 0 && W > 0 && E > 0 && E % 4 == 0, "dw_stats: E=%d must be a multiple of 4", E);
  launch_dw_strip_stats<0>(x1, nullptr, nullptr, nullptr, nullptr, nullptr, B, H, W, E, w5, w3, wv, wh, stats, lmn_se_bwd_t{}, zp, act_dtype, (hipStream_t)stream);
  return lmn_launch_status("dw_stats");
}

int lmn_dw_bwd_stats(const void* x1, const void* pre, const void* u, const float* s, const float* dm, void* dpre,
                     int B, int H, int W, int E, const float* w5, const float* w3, const float* wv, const float* wh,
                     float* bstats, const lmn_se_bwd_t* seb, const lmn_dw_pre_t* zpre, int act_dtype, lmn_stream_t stream) {
  lmn_dw_pre_t zp;
  { const int rc = dw_pre_check(zpre, &zp, false, "dw_bwd_stats"); if (rc) return rc; }
  lmn_se_bwd_t sb;
  memset(&sb, 0, sizeof(sb));
  if (seb && seb->ds) {
    LMN_REQUIRE(seb->w1 && seb->w2 && seb->hidden && seb->dvec && seb->R > 0 && seb->inv_hw > 0.f, "dw_bwd_stats: squeeze-excite operands");
    LMN_REQUIRE(E + seb->R + 256 <= FS_XR * SW_XC * SW_CS, "dw_bwd_stats: E + R = %d exceeds the block's scratch", E + seb->R);
    sb = *seb;
  }
  if (g_lmn_rec) lmn_rec_push([=]() -> int { return lmn_dw_bwd_stats(x1, pre, u, s, dm, dpre, B, H, W, E, w5, w3, wv, wh, bstats, &sb, &zp, act_dtype, stream); }, "lmn_dw_bwd_stats(");
  LMN_REQUIRE_DT(act_dtype, "dw_bwd_stats");
  LMN_REQUIRE(x1 && pre && u && s && (dm || sb.ds) && dpre && w5 && w3 && wv && wh && bstats, "dw_bwd_stats: null pointer");
  LMN_REQUIRE(B > 0 && H > 0 && W > 0 && E > 0 && E % 4 == 0, "dw_bwd_stats: E=%d must be a multiple of 4", E);
  launch_dw_strip_stats<1>(x1, pre, u, s, dm, dpre, B, H, W, E, w5, w3, wv, wh, bstats, sb, zp, act_dtype, (hipStream_t)stream);
  return lmn_launch_status("dw_bwd_stats");
}

int lmn_dw_bwd_coef(const float* bstats, const float* mean, const float* rstd, const float* A, float count,
                    int batch_stats, float* cA, float* cC, float* cD, float* dg0, float* dg1, float* dg2, float* dg3,
                    float* db0, float* db1, float* db2, float* db3, int E, lmn_stream_t stream) {
  LMN_REC(lmn_dw_bwd_coef(bstats, mean, rstd, A, count, batch_stats, cA, cC, cD, dg0, dg1, dg2, dg3, db0, db1, db2, db3, E, stream));
  LMN_REQUIRE(bstats && mean && rstd && A && cA && cC && cD && dg0 && dg1 && dg2 && dg3 && db0 && db1 && db2 && db3 && E > 0 && count > 0.f,
              "dw_bwd_coef: bad argument");
  LMN_LAUNCH(dw_bwd_coef_kernel, dim3(lmn_cdiv(4 * E, 256)), dim3(256), 0, (hipStream_t)stream, bstats, mean, rstd,
                     A, count, batch_stats, cA, cC, cD, dg0, dg1, dg2, dg3, db0, db1, db2, db3, E);
  return lmn_launch_status("dw_bwd_coef");
}

int lmn_dw_bwd(const void* x1, const void* dpre, void* dx1, int B, int H, int W, int E, const float* w5,
               const float* w3, const float* wv, const float* wh, const float* cA, const float* cC, const float* cD,
               float* dw5, float* dw3, float* dwv, float* dwh, int act_dtype, lmn_stream_t stream) {
  LMN_REC(lmn_dw_bwd(x1, dpre, dx1, B, H, W, E, w5, w3, wv, wh, cA, cC, cD, dw5, dw3, dwv, dwh, act_dtype, stream));
  LMN_REQUIRE_DT(act_dtype, "dw_bwd");
  LMN_REQUIRE(x1 && dpre && dx1 && w5 && w3 && wv && wh && cA && cC && cD && dw5 && dw3 && dwv && dwh, "dw_bwd: null pointer");
  LMN_REQUIRE(B > 0 && H > 0 && W > 0 && E > 0 && E % 4 == 0, "dw_bwd: E=%d must be a multiple of 4", E);
  LMN_REQUIRE((int64_t)(H + 16) * W * E * 4 < (1LL << 30), "dw_bwd: one image (%d x %d x %d) must stay below 1 GiB", H, W, E);
  // strip-walking kernel: blocks = B x strips(60 columns) x row segments x 8-channel chunks (segments: strip_segments)
  const int strips = lmn_cdiv(W, SW_OC), chunks = lmn_cdiv(E, SW_CH);
  int seg_rows;
  const int segs = strip_segments((int64_t)B * strips * chunks, H, 10, 2, &seg_rows);  // 242 VGPRs: 2 blocks per CU
  const int64_t nblk = (int64_t)B * strips * chunks * segs;
  LMN_REQUIRE(nblk < (1LL << 31), "dw_bwd: grid too large");
  if (g_lmn_prof_on) lmn_prof_cost(2.0 * 2 * 42 * (double)B * H * W * E, (act_dtype == LMN_BF16 ? 2.0 : 4.0) * 3 * (double)B * H * W * E);
  float *g5 = dw5, *g3 = dw3, *gv = dwv, *gh = dwh;
  const int nslots = B * segs * strips;
  if (g_lmn_det) {   // per-block weight-gradient sums into slot copies of the four tensors
    hipStream_t st = (hipStream_t)stream;
    lmn_det_begin(st);
    float* base = lmn_det_slots(st, (size_t)nslots * 40 * E);
    LMN_REQUIRE(base, "dw_bwd: deterministic mode: no scratch");
    g5 = base; g3 = g5 + (size_t)nslots * 25 * E; gv = g3 + (size_t)nslots * 9 * E; gh = gv + (size_t)nslots * 3 * E;
  }
  LMN_ACT_DISPATCH(act_dtype, LMN_LAUNCH((dw_bwd_strip_kernel<T>), dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, (const T*)x1, (const T*)dpre, (T*)dx1, B, H, W, E, w5,
                     w3, wv, wh, cA, cC, cD, DwCoef{}, g5, g3, gv, gh, DwPreS{nullptr, nullptr}, (float*)nullptr, strips, segs, seg_rows, chunks, g_lmn_det));
  if (g_lmn_det) {
    hipStream_t st = (hipStream_t)stream;
    lmn_det_sum(st, g5, nslots, (int64_t)25 * E, dw5); lmn_det_sum(st, g3, nslots, (int64_t)9 * E, dw3);
    lmn_det_sum(st, gv, nslots, (int64_t)3 * E, dwv); lmn_det_sum(st, gh, nslots, (int64_t)3 * E, dwh);
  }
  return lmn_launch_status("dw_bwd");
}

int lmn_dw_bwd_bn(const void* x1, const void* dpre, void* dx1, int B, int H, int W, int E, const float* w5, const float* w3,
                  const float* wv, const float* wh, const float* bstats, const float* mean, const float* rstd, const float* A,
                  float count, int batch_stats, float* const* dgamma, float* const* dbeta, float* dw5, float* dw3, float* dwv,
                  float* dwh, int part, const lmn_dw_pre_t* zpre, float* hstats, int act_dtype, lmn_stream_t stream) {
  LMN_REQUIRE_DT(act_dtype, "dw_bwd_bn");
  lmn_dw_pre_t zp;
  { const int rc = dw_pre_check(zpre, &zp, false, "dw_bwd_bn"); if (rc) return rc; }
  LMN_REQUIRE(!zp.A || (hstats && part != 2), "dw_bwd_bn: the z-path writes dh and needs hstats [2][E] (part 0 or 1)");
  LMN_REQUIRE(part >= 0 && part <= 2, "dw_bwd_bn: part %d", part);
  LMN_REQUIRE(x1 && dpre && dx1 && w5 && w3 && wv && wh && bstats && mean && rstd && A && dgamma && dbeta && dw5 && dw3 && dwv && dwh &&
                  count > 0.f, "dw_bwd_bn: bad argument");
  LMN_REQUIRE(B > 0 && H > 0 && W > 0 && E > 0 && E % 4 == 0, "dw_bwd_bn: E=%d must be a multiple of 4", E);
  LMN_REQUIRE((int64_t)(H + 16) * W * E * 4 < (1LL << 30), "dw_bwd_bn: one image (%d x %d x %d) must stay below 1 GiB", H, W, E);
  DwCoef cf;
  cf.bstats = bstats; cf.mean = mean; cf.rstd = rstd; cf.A = A; cf.count = count; cf.batch_stats = batch_stats;
  for (int b = 0; b < 4; ++b) {
    LMN_REQUIRE(dgamma[b] && dbeta[b], "dw_bwd_bn: null gradient tensor %d", b);
    cf.dg[b] = dgamma[b]; cf.db[b] = dbeta[b];
  }
  const int strips = lmn_cdiv(W, SW_OC), chunks = lmn_cdiv(E, SW_CH);
  int seg_rows;
  const int segs = strip_segments((int64_t)B * strips * chunks, H, 10, part == 1 ? 3 : 2, &seg_rows);  // dx1 only: 151 VGPRs, 3 blocks per CU
  const int64_t nblk = (int64_t)B * strips * chunks * segs;
  LMN_REQUIRE(nblk < (1LL << 31), "dw_bwd_bn: grid too large");
  auto launch = [=]() -> int {
    if (g_lmn_prof_on) lmn_prof_cost(2.0 * 2 * 42 * (double)B * H * W * E, (act_dtype == LMN_BF16 ? 2.0 : 4.0) * 3 * (double)B * H * W * E);
    float *g5 = dw5, *g3 = dw3, *gv = dwv, *gh = dwh, *hs = hstats;
    const int nslots = B * segs * strips;
    hipStream_t st = (hipStream_t)stream;
    if (g_lmn_det) {   // per-block sums into slot copies of the four weight-gradient tensors and of hstats [2][E]
      lmn_det_begin(st);
      float* base = lmn_det_slots(st, (size_t)nslots * 42 * E);
      LMN_REQUIRE(base, "dw_bwd_bn: deterministic mode: no scratch");
      g5 = base; g3 = g5 + (size_t)nslots * 25 * E; gv = g3 + (size_t)nslots * 9 * E; gh = gv + (size_t)nslots * 3 * E;
      if (hstats) hs = gh + (size_t)nslots * 3 * E;
    }
#define LMN_DWB2(PT, Z) LMN_ACT_DISPATCH(act_dtype, LMN_LAUNCH((dw_bwd_strip_kernel<T, PT, Z>), dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, (const T*)x1, (const T*)dpre, (T*)dx1, B, H, W, E, w5, \
                       w3, wv, wh, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, cf, g5, g3, gv, gh, DwPreS{zp.A, zp.shift}, hs, strips, segs, seg_rows, chunks, g_lmn_det))
#define LMN_DWB(PT) do { if (zp.A) LMN_DWB2(PT, true); else LMN_DWB2(PT, false); } while (0)
    if (part == 1) { LMN_DWB(1); } else if (part == 2) { LMN_DWB(2); } else { LMN_DWB(0); }
#undef LMN_DWB2
#undef LMN_DWB
    if (g_lmn_det) {
      if (part != 1) {
        lmn_det_sum(st, g5, nslots, (int64_t)25 * E, dw5); lmn_det_sum(st, g3, nslots, (int64_t)9 * E, dw3);
        lmn_det_sum(st, gv, nslots, (int64_t)3 * E, dwv); lmn_det_sum(st, gh, nslots, (int64_t)3 * E, dwh);
      }
      if (hstats && part != 2) lmn_det_sum(st, hs, nslots, (int64_t)2 * E, hstats);
    }
    return lmn_launch_status("dw_bwd_bn");
  };
  if (g_lmn_rec) lmn_rec_push(launch, "lmn_dw_bwd_bn(");
  return launch();
}

}  // extern "C"
