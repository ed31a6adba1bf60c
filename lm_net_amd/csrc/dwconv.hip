// Multi-branch depthwise stencil of ReparamConv (core/modules.py:548-574, 592-597).
//
//   pre = sum_b BN_b( dw_b(x1) ),  b in {5x5, 3x3, 3x1, 1x3};   g = GELU(pre)
//
// Round 4: the E-wide tensors of a ReparamConv block (z / x1, pre, u, dpre, dh) live in the ROW-PLANAR layout "RP4"
// (include/lmnet_hip.h): element (image row r = b*H + y, column x, channel c) at (r * (E/4) + (c >> 2)) * 4*W + 4*x + (c & 3),
// i.e. one contiguous plane of W pixels x 4 channels per (row, channel quad).  A depthwise stencil wants lanes along x and a
// wave-uniform channel (weights in SGPRs): in RP4 the 64 columns of a wave's channel pair are ONE 8-bytes-per-lane load at a
// 16-byte lane stride -- no LDS transpose, no block-shared staging, no barriers.  A WAVE owns (image, row segment, strip of 60
// output columns, channel pair) and walks down the rows alone: global loads a few rows ahead in registers, one LDS row per
// wave to exchange the +-1 / +-2 column neighbours (ds_write_b64 + four ds_read_b64, conflict-free), rotating row accumulators.
// The conv kernels read / write RP4 through lmn_src_t.rp_w / out_rp_w / aux_rp_w (same 16-byte accesses, other addresses).
//
// What bounds the kernels now (tools/micro/dwq_proto.hip, MI355X): the fp32 vector ALU (32 lanes per SIMD and cycle: a packed FMA
// costs 4 cycles, packing saves issue slots only).  Forward statistics 48 + 8 FMAs per element: 34 us at level 0 for a 32 us
// ALU floor; forward / backward statistics stream at 3.9 / 4.1 TB/s; the backward (120 FMAs per element) runs at ~70 % of its
// ALU time inside the row loop.
#include "common.h"
// the launch's deterministic-mode argument carries the wave priority in bits 8-9 (lmn_set_priority_stream: kernels on the compute
// chain's stream raise their waves' issue priority against the waves of the other streams' kernels on the same CU)
#define LMN_DW_SETPRIO do { lmn_setprio_level(det >> 8); det &= 0xff; } while (0)
#define DW_DET(st) (g_lmn_det | (lmn_prio_level(st) << 8))

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr unsigned OOB = 0x80000000u;     // voffset of a lane outside the tensor: beyond num_records, loads return 0, stores drop
constexpr unsigned NREC = 0x7FFFFFFFu;    // every in-tensor voffset is smaller than one plane; the row offset travels in soffset
constexpr int QW = 60;                    // output columns per strip: lanes 2..61 of a wave (lane l = column xs - 2 + l)

// ---------------------------------------------------------------------------------------------------------------- helpers
__device__ __forceinline__ float dpp_wave_shr1(float a) {   // lane l gets lane l-1, lane 0 gets 0
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x138, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_wave_shl1(float a) {   // lane l gets lane l+1, lane 63 gets 0
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x130, 0xF, 0xF, true));
}
// (scalar temporaries on purpose: __builtin_bit_cast on a vector ELEMENT made hipcc 7.2 shift element 0 only and broadcast it)
__device__ __forceinline__ f32x2 lane_from_left(f32x2 v) { const float a0 = v.x, a1 = v.y; f32x2 r; r.x = dpp_wave_shr1(a0); r.y = dpp_wave_shr1(a1); return r; }
__device__ __forceinline__ f32x2 lane_from_right(f32x2 v) { const float a0 = v.x, a1 = v.y; f32x2 r; r.x = dpp_wave_shl1(a0); r.y = dpp_wave_shl1(a1); return r; }
__device__ __forceinline__ float wave_total(float a) {   // DPP reduction: the total lands in lane 63
#define LMN_DPP_ADD(CTRL) a += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), CTRL, 0xF, 0xF, true));
  LMN_DPP_ADD(0x111) LMN_DPP_ADD(0x112) LMN_DPP_ADD(0x114) LMN_DPP_ADD(0x118) LMN_DPP_ADD(0x142) LMN_DPP_ADD(0x143)
#undef LMN_DPP_ADD
  return a;
}
__device__ __forceinline__ f32x2 wave_total(f32x2 v) { return f32x2{wave_total(v.x), wave_total(v.y)}; }

// the lanes of ONE wave exchange data through LDS: hipcc sees no per-thread alias between a lane's store and its reads of the
// neighbours' slots (it hoisted the reads above the store), so both directions get a code-free wave-level fence
#ifndef LMN_DWF_OCC
#define LMN_DWF_OCC 3
#endif
#define LMN_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
#define LMN_SB() __builtin_amdgcn_sched_barrier(0)

// activation loads / stores with the row offset in soffset: two consecutive channels (a pair) or one channel
template <typename TA> __device__ __forceinline__ f32x2 ld_pair(BufRsrc r, unsigned voff, unsigned so) {
  if constexpr (sizeof(TA) == 4) return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)so, 0));
  else { const unsigned u = __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)so, 0); return f32x2{lmn_bf16_lo(u), lmn_bf16_hi(u)}; }
}
template <typename TA> __device__ __forceinline__ void st_pair(BufRsrc r, unsigned voff, unsigned so, f32x2 v) {
  if constexpr (sizeof(TA) == 4) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, (int)voff, (int)so, 0);
  else __builtin_amdgcn_raw_buffer_store_b32(lmn_pk_bf16(v.x, v.y), r, (int)voff, (int)so, 0);
}
template <typename TA> __device__ __forceinline__ float ld_one(BufRsrc r, unsigned voff, unsigned so) {
  if constexpr (sizeof(TA) == 4) return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)so, 0));
  else return lmn_bf16_lo((unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r, (int)voff, (int)so, 0));
}
template <typename TA> __device__ __forceinline__ void st_one(BufRsrc r, unsigned voff, unsigned so, float v) {
  if constexpr (sizeof(TA) == 4) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)voff, (int)so, 0);
  else __builtin_amdgcn_raw_buffer_store_b16((short)(lmn_pk_bf16(v, 0.f) & 0xFFFFu), r, (int)voff, (int)so, 0);
}

// ---------------------------------------------------------------------------------------------------------------
// z-path (lmn_dw_pre_t): the tensor handed to the depthwise kernels is z, the expand conv's output BEFORE its BatchNorm and
// Hardswish; x1 = Hardswish(A * z + shift) is formed where a row enters a wave (once per element).  The training forward then
// needs no statistics-only conv and no second read of the expand conv's input, and the backward forms dh = dx1 *
// Hardswish'(A * z + shift) and its BatchNorm-backward sums where dx1 leaves the depthwise backward.
// Zero padding: rows / columns / channels outside the tensor must stay 0 AFTER the transform -> explicit 0 / 1 factors.
typedef lmn_dw_pre_t DwPreK;   // A / shift [E] (NULL: the tensor holds x1, no transform); fin.mode LMN_FIN_BN: formed in the kernel
struct DwPreS { const float* A; const float* shift; };   // the same without the finalisation (kernel arguments of the other passes)
// ---- nn.GELU() (erf form) and its derivative on a channel pair, in packed arithmetic.
// erf(u) = u A P(t) / Q(t), t = u^2, |u| <= 4 (erf(4) = 1 - 1.5e-8): an [11/10] rational form fitted here (weighted least squares
// on Chebyshev nodes, fit error 2.3e-8), P and Q monic so that each Horner chain starts with an add.  Evaluated in fp32:
// |gelu error| <= 2.6e-7 max(|x|, 1), |gelu' error| <= 3.2e-7 (checked against scipy over [-12, 12] in steps of 1e-5).
// The scalar lmn_erf (7.1.26: exp + rcp + a 5-term polynomial per element: 17 VALU instructions, two of them quarter-rate, plus
// their wait states) was as much issue time in the forward as the 25-tap stencil; here a pair costs 17 packed instructions + 2 rcp.
// The constants travel as halves of SGPR pairs selected by op_sel (hipcc builds {c, c} VGPR pairs with two v_mov per use otherwise).
#define LMN_PK_K(NAME, OP, ARGS, LO, HI)                                                                        \
  __device__ __forceinline__ f32x2 NAME(f32x2 a, f32x2 b, f32x2 kp, int hi) {                                   \
    f32x2 r;                                                                                                    \
    if (hi) asm(OP " %0, " ARGS " " HI : "=v"(r) : "v"(a), "v"(b), "s"(kp));                                   \
    else asm(OP " %0, " ARGS " " LO : "=v"(r) : "v"(a), "v"(b), "s"(kp));                                      \
    return r;                                                                                                   \
  }
LMN_PK_K(pk_fma_k, "v_pk_fma_f32", "%1, %2, %3", "op_sel_hi:[1,1,0]", "op_sel:[0,0,1] op_sel_hi:[1,1,1]")   // a * b + k
LMN_PK_K(pk_add_k, "v_pk_add_f32", "%1, %3", "op_sel_hi:[1,0]", "op_sel:[0,1] op_sel_hi:[1,1]")              // a + k   (b unused)
LMN_PK_K(pk_mul_k, "v_pk_mul_f32", "%1, %3", "op_sel_hi:[1,0]", "op_sel:[0,1] op_sel_hi:[1,1]")              // a * k   (b unused)
LMN_PK_K(pk_mad_k, "v_pk_fma_f32", "%1, %3, %2", "op_sel_hi:[1,0,1]", "op_sel:[0,1,0] op_sel_hi:[1,1,1]")   // a * k + b
#undef LMN_PK_K
struct ErfQ { f32x2 u, pr; };   // clamped argument, P(t) / Q(t)
__device__ __forceinline__ ErfQ erf_parts(f32x2 x) {
  const f32x2 K0 = {0.70710678118654752440f, 1.443648930e+02f}, K1 = {1.985670226e+03f, 2.685849656e+04f},
              K2 = {9.820033276e+04f, 5.689660988e+05f}, K3 = {3.201790747e+01f, 4.086303418e+02f},
              K4 = {3.091390772e+03f, 1.349988240e+04f}, K5 = {2.668350604e+04f, 0.f};
  f32x2 u = pk_mul_k(x, x, K0, 0);
  u.x = __builtin_amdgcn_fmed3f(u.x, -4.f, 4.f);
  u.y = __builtin_amdgcn_fmed3f(u.y, -4.f, 4.f);
  const f32x2 t = u * u;
  f32x2 p = pk_add_k(t, t, K0, 1);
  p = pk_fma_k(p, t, K1, 0);
  p = pk_fma_k(p, t, K1, 1);
  p = pk_fma_k(p, t, K2, 0);
  p = pk_fma_k(p, t, K2, 1);
  f32x2 q = pk_add_k(t, t, K3, 0);
  q = pk_fma_k(q, t, K3, 1);
  q = pk_fma_k(q, t, K4, 0);
  q = pk_fma_k(q, t, K4, 1);
  q = pk_fma_k(q, t, K5, 0);
  return ErfQ{u, p * f32x2{__builtin_amdgcn_rcpf(q.x), __builtin_amdgcn_rcpf(q.y)}};
}
__device__ __forceinline__ f32x2 gelu2(f32x2 x) {   // x / 2 + (x A / 2) u P / Q
  const f32x2 KA = {2.645949311e-02f, 0.39894228040143267794f};
  const ErfQ e = erf_parts(x);
  const f32x2 m = (pk_mul_k(x, x, KA, 0) * e.u) * e.pr;
  return x * 0.5f + m;
}
__device__ __forceinline__ f32x2 dgelu2(f32x2 x) {   // 1/2 + (A / 2) u P / Q + x exp(-x^2 / 2) / sqrt(2 pi)
  const f32x2 KA = {2.645949311e-02f, 0.39894228040143267794f}, KE = {-0.72134752044448170368f, 0.f};
  const ErfQ e = erf_parts(x);
  const f32x2 m = pk_mul_k(e.u, e.u, KA, 0) * e.pr + 0.5f;
  const f32x2 a = pk_mul_k(x * x, x, KE, 0);   // -x^2 / 2 in base 2
  const f32x2 xe = x * f32x2{__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};
  return pk_mad_k(xe, m, KA, 1);
}
__device__ __forceinline__ f32x2 hswish2(f32x2 z, f32x2 a, f32x2 s) {
  const f32x2 x = z * a + s;
  f32x2 t = x * (1.f / 6.f) + 0.5f;
  t.x = __builtin_amdgcn_fmed3f(t.x, 0.f, 1.f);
  t.y = __builtin_amdgcn_fmed3f(t.y, 0.f, 1.f);
  return x * t;
}
// (experiment, LMN_DWF_ASM) Stencil FMAs of the forward-type passes as asm statements: they keep the source's order (round-robin over the five row accumulators),
// so no packed result is read by the NEXT instruction.  Left to the scheduler, hipcc grouped the FMAs by accumulator and the hazard
// recogniser then put an `s_nop 0` between every dependent pair (a packed fp32 result needs one wait state before a VALU read): ~20 of
// ~85 issue slots per row step in kernels that the counters show issue-bound.
__device__ __forceinline__ void fpk_fma(f32x2& acc, f32x2 x, f32x2 y) { asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(y)); }
__device__ __forceinline__ void fpk_mul(f32x2& acc, f32x2 x, f32x2 y) { asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(acc) : "v"(x), "v"(y)); }
__device__ __forceinline__ void fpk_fma_s(f32x2& acc, f32x2 x, f32x2 ws) { asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(x), "s"(ws)); }   // weight pair in SGPRs
__device__ __forceinline__ void fpk_mul_s(f32x2& acc, f32x2 x, f32x2 ws) { asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(acc) : "v"(x), "s"(ws)); }

// A / shift of the block's 8 channels -> pre_s[0..7] = A, pre_s[8..15] = shift (threads 0..7), formed from the batch sums of the
// expand conv when fin.mode says so (lmn_bn_fin_t arithmetic: slices summed in double about `about`); `writer` blocks also
// store mean / rstd / A / shift and blend the running statistics.
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
// (lmn_dw_finalize_merge arithmetic); backward: the coefficients cA / cC / cD of f_b (lmn_dw_bwd_coef arithmetic).  One wave per
// channel (pair) also writes the side outputs (saved mean / rstd / A and the running statistics; the gamma / beta gradients).
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

struct BranchW {  // a wave's channel pair of the four branch kernels
  f32x2 w5[25], w3[9], wv[3], wh[3];
};
__device__ __forceinline__ void load_branch_w(BranchW& bw, const float* w5, const float* w3, const float* wv, const float* wh, int ch, bool ok) {
  const f32x2 z = f32x2{0.f, 0.f};   // pairs past E (partial last chunk) carry zero weights
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
// Geometry of the forward-type passes (statistics, forward, backward statistics): block = 4 waves = 4 channel pairs (8 channels,
// two quads: the pair siblings of a quad read the same cache lines) of one (image, row segment, strip).
struct GeoP {
  int b, seg, strip, ch0, ch, ys, ye, xs, r0;
  bool cok;
  unsigned rowb, qoff;   // bytes per image row of the tensor (all quads); byte offset of this wave's quad plane inside a row
};
template <int ES>
__device__ __forceinline__ GeoP decode_pair(int E, int H, int W, int strips, int segs, int seg_rows, int chunks, int wv) {
  GeoP g;
  int lid = blockIdx.x;
  const int chunk = lid % chunks; lid /= chunks;
  g.strip = lid % strips; lid /= strips;
  g.seg = lid % segs;
  g.b = lid / segs;
  g.ch0 = chunk * 8;
  g.ch = g.ch0 + wv * 2;
  g.cok = g.ch < E;
  if (!g.cok) g.ch = 0;
  g.ys = g.seg * seg_rows;
  g.ye = min(g.ys + seg_rows, H);
  g.xs = g.strip * QW;
  g.rowb = (unsigned)(E * W) * (unsigned)ES;
  g.qoff = (unsigned)((g.ch >> 2) * 4 * W) * (unsigned)ES;
  g.r0 = g.b * H;
  return g;
}
__device__ __forceinline__ unsigned row_off(const GeoP& g, int iy, int H) {   // rows outside the image read the segment's first row (callers mask the value)
  const int y = (unsigned)iy < (unsigned)H ? iy : g.ys;
  return (unsigned)(g.r0 + y) * g.rowb + g.qoff;
}

// ---------------------------------------------------------------------------------------------------------------
// Forward batch statistics: stats[4][2][E] += (sum y_b, sum y_b^2).  z row (ys-2+j) is exchanged during step j-1, its FMAs run at
// step j (the exchange of the next row is software-pipelined under them); output row (ys+j-4) completes at step j.
template <typename TA, bool ZT>
__global__ __launch_bounds__(256, 4) void dw_stats0_kernel(const TA* __restrict__ x1, int H, int W, int E, const float* __restrict__ w5,
                                                            const float* __restrict__ w3, const float* __restrict__ wvv,
                                                            const float* __restrict__ whh, float* __restrict__ stats, const DwPreK PRE,
                                                            int strips, int segs, int seg_rows, int chunks, int det) {
  constexpr int ES = sizeof(TA), D = 3;
  __shared__ f32x2 XSa[4][2][68];
  __shared__ float red[8 * 8];
  __shared__ __attribute__((aligned(16))) float pre_s[16];
  LMN_DW_SETPRIO;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const GeoP g = decode_pair<ES>(E, H, W, strips, segs, seg_rows, chunks, wv);
  constexpr bool zt = ZT;   // the input tensor is z (see DwPreK): a template argument, so that the row steps carry no branch on it
  if (zt) dw_pre_setup(PRE, pre_s, g.ch0, E, tid, g.strip == 0 && g.seg == 0 && g.b == 0);
  f32x2* XS0 = XSa[wv][0];
  if (lane < 8) XS0[(lane >> 2) * 68 + ((lane & 3) < 2 ? (lane & 3) : 64 + (lane & 3))] = f32x2{0.f, 0.f};
  BranchW bw;
  load_branch_w(bw, w5, w3, wvv, whh, g.ch, g.cok);
  // (the 5x5 kernel of the pair in SGPRs, the three small kernels in VGPRs: 80 weight floats do not fit the scalar file)
#pragma unroll
  for (int k = 0; k < 9; ++k) asm volatile("" : "+v"(bw.w3[k]));
#pragma unroll
  for (int k = 0; k < 3; ++k) asm volatile("" : "+v"(bw.wv[k]), "+v"(bw.wh[k]));
  if (zt) __syncthreads();
  const f32x2 pa = zt ? f32x2{pre_s[wv * 2], pre_s[wv * 2 + 1]} : f32x2{0.f, 0.f};
  const f32x2 ps = zt ? f32x2{pre_s[8 + wv * 2], pre_s[8 + wv * 2 + 1]} : f32x2{0.f, 0.f};
  const int cx = g.xs - 2 + lane;
  const bool col_in = cx >= 0 && cx < W && g.cok;
  const float cm = col_in ? 1.f : 0.f;
  const bool ovalid = lane >= 2 && lane < 62 && cx < W && g.cok;
  const unsigned voff = col_in ? (unsigned)(cx * 4 + (g.ch & 3)) * (unsigned)ES : OOB;
  const BufRsrc rz = make_rsrc(x1, NREC);
  const int rows = g.ye - g.ys, nsteps = rows + 4;
  f32x2 pf[5];
#pragma unroll
  for (int d = 0; d < D; ++d) pf[d] = ld_pair<TA>(rz, voff, row_off(g, g.ys - 2 + d, H));
  const f32x2 z2 = f32x2{0.f, 0.f};
  f32x2 a5[5], a3[5], av[5], ah[5], sum[8];
#pragma unroll
  for (int k = 0; k < 5; ++k) a5[k] = a3[k] = av[k] = ah[k] = z2;
#pragma unroll
  for (int k = 0; k < 8; ++k) sum[k] = z2;
  f32x2 in[5], inn[5];
  {  // row 0
    const int iy = g.ys - 2;
    const float rm = (iy >= 0 && iy < H) ? cm : 0.f;
    const f32x2 x1v = (zt ? hswish2(pf[0], pa, ps) : pf[0]) * rm;
    pf[D % 5] = ld_pair<TA>(rz, voff, row_off(g, g.ys - 2 + D, H));
    XS0[lane + 2] = x1v;
    LMN_WAVE_SYNC();
    in[0] = XS0[lane]; in[1] = XS0[lane + 1]; in[2] = x1v; in[3] = XS0[lane + 3]; in[4] = XS0[lane + 4];
  }
#define LMN_STEP(P)                                                                                                \
  {                                                                                                                \
    const int j = j0 + P;                                                                                          \
    const int iyn = g.ys - 1 + j;                                                                                  \
    f32x2 x1n = (zt ? hswish2(pf[(P + 1) % 5], pa, ps) : pf[(P + 1) % 5]) * ((iyn >= 0 && iyn < H) ? cm : 0.f);    \
    pf[(P + 1 + D) % 5] = ld_pair<TA>(rz, voff, row_off(g, g.ys - 1 + j + D, H));                                   \
    f32x2* XS = XS0 + ((P + 1) & 1) * 68;                                                                          \
    LMN_WAVE_SYNC();                                                                                               \
    XS[lane + 2] = x1n;                                                                                            \
    LMN_WAVE_SYNC();                                                                                               \
    inn[0] = XS[lane]; inn[1] = XS[lane + 1]; inn[2] = x1n; inn[3] = XS[lane + 3]; inn[4] = XS[lane + 4];          \
    _Pragma("unroll") for (int d = 0; d < 5; ++d) {                                                                \
      if (d == 0) a5[P] = bw.w5[0] * in[0];                                                                        \
      else a5[P] += bw.w5[d] * in[d];                                                                              \
      _Pragma("unroll") for (int ky = 1; ky < 5; ++ky) a5[(P - ky + 5) % 5] += bw.w5[ky * 5 + d] * in[d];          \
      if (d >= 1 && d <= 3) {                                                                                      \
        if (d == 1) { a3[(P + 4) % 5] = bw.w3[0] * in[1]; ah[(P + 3) % 5] = bw.wh[0] * in[1]; }                    \
        else { a3[(P + 4) % 5] += bw.w3[d - 1] * in[d]; ah[(P + 3) % 5] += bw.wh[d - 1] * in[d]; }                 \
        _Pragma("unroll") for (int ky = 1; ky < 3; ++ky) a3[(P + 4 - ky) % 5] += bw.w3[ky * 3 + d - 1] * in[d];    \
      }                                                                                                            \
      if (d == 2) {                                                                                                \
        av[(P + 4) % 5] = bw.wv[0] * in[2];                                                                        \
        _Pragma("unroll") for (int ky = 1; ky < 3; ++ky) av[(P + 4 - ky) % 5] += bw.wv[ky] * in[2];                \
      }                                                                                                            \
    }                                                                                                              \
    constexpr int DD = (P + 1) % 5;   /* output row j-4 is complete in every branch */                             \
    if (j >= 4 && j < nsteps) {                                                                                    \
      const f32x2 y5 = a5[DD], y3 = a3[DD], yv = av[DD], yh = ah[DD];                                              \
      sum[0] += y5; sum[1] += y3; sum[2] += yv; sum[3] += yh;                                                      \
      sum[4] += y5 * y5; sum[5] += y3 * y3; sum[6] += yv * yv; sum[7] += yh * yh;                                  \
    }                                                                                                              \
    _Pragma("unroll") for (int d = 0; d < 5; ++d) in[d] = inn[d];                                                  \
    LMN_SB();                                                                                                      \
  }
  for (int j0 = 0; j0 < nsteps; j0 += 5) { LMN_STEP(0) LMN_STEP(1) LMN_STEP(2) LMN_STEP(3) LMN_STEP(4) }
#undef LMN_STEP
#pragma unroll
  for (int k = 0; k < 8; ++k) {   // (columns outside the image / the strip's halo lanes are dropped from the lane sums here)
    const f32x2 v = wave_total(ovalid ? sum[k] : z2);
    if (lane == 63) { red[k * 8 + wv * 2] = v.x; red[k * 8 + wv * 2 + 1] = v.y; }
  }
  __syncthreads();
  if (tid < 64) {   // the block's 64 sums leave as ONE atomic instruction
    const int k = tid / 8, c8 = tid - k * 8;
    const int row = (k & 3) * 2 + (k >> 2);   // [branch][sum | sumsq][E]
    // (deterministic mode: stats addresses slot copies of [8][E], one slot per (image, segment, strip))
    if (g.ch0 + c8 < E) lmn_red_add(stats + (det ? (int64_t)((g.b * segs + g.seg) * strips + g.strip) * 8 * E : 0) + (int64_t)row * E + g.ch0 + c8, red[tid], det);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Forward (flagship HBM-bound kernel of row A2): pre = merged 5x5 (+ bias) -> store; gsum[b][e] += sum GELU(pre) (SE squeeze), and
// optionally the squeeze-excite gate of an image by the block that completes its sums (lmn_se_fuse_t).
#ifndef LMN_DWF_ASM
#define LMN_DWF_ASM 0   // experiment (-DLMN_DWF_ASM=1): stencil FMAs of dw_fwd as ordered asm statements -- 135 -> 51 s_nop, 135 -> 111 VGPRs, same time (59.2 -> 61.2 us at level 0): the pass is not issue-bound after all, DESIGN 5h
#endif
template <typename TA, bool ZT>
__global__ __launch_bounds__(256, LMN_DWF_OCC) void dw_fwd_kernel(const TA* __restrict__ x1, TA* __restrict__ pre, float* __restrict__ gsum, int H, int W,
                                                         int E, const float* __restrict__ keff, const float* __restrict__ beff, const DwFin FN,
                                                         const lmn_se_fuse_t SE, const DwPreS PRE, int strips, int segs, int seg_rows, int chunks,
                                                         int det, int nimg) {
  constexpr int ES = sizeof(TA), D = 5;
  __shared__ f32x2 XSa[4][2][68];
  __shared__ float gs_s[8];
  __shared__ int s_last;
  __shared__ __attribute__((aligned(16))) float pre_s[16];
  __shared__ float scr[2048];   // squeeze-excite gate: mean [E], hidden [R]
  LMN_DW_SETPRIO;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const GeoP g = decode_pair<ES>(E, H, W, strips, segs, seg_rows, chunks, wv);
  constexpr bool zt = ZT;   // the input tensor is z (see DwPreK)
  if (zt) dw_pre_setup(PRE, pre_s, g.ch0, E, tid);
  f32x2* XS0 = XSa[wv][0];
  if (lane < 8) XS0[(lane >> 2) * 68 + ((lane & 3) < 2 ? (lane & 3) : 64 + (lane & 3))] = f32x2{0.f, 0.f};
  f32x2 w[25];
  f32x2 bias;
  if (FN.stats) {  // block-uniform: the four branch BatchNorms are finalised and merged here (lmn_dw_finalize_merge arithmetic)
    float wm[2][25], bs[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int e = g.ch + h;
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
        if (g.cok && lane == 0 && g.strip == 0 && g.seg == 0 && g.b == 0) {  // one wave per channel pair writes the side outputs
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
        wm[h][t] = g.cok ? v : 0.f;
      }
    }
#pragma unroll
    for (int t = 0; t < 25; ++t) w[t] = f32x2{wm[0][t], wm[1][t]};
    bias = g.cok ? f32x2{bs[0], bs[1]} : f32x2{0.f, 0.f};
  } else {
#pragma unroll
    for (int t = 0; t < 25; ++t) w[t] = g.cok ? f32x2{keff[(int64_t)g.ch * 25 + t], keff[(int64_t)(g.ch + 1) * 25 + t]} : f32x2{0.f, 0.f};
    bias = g.cok ? f32x2{beff[g.ch], beff[g.ch + 1]} : f32x2{0.f, 0.f};
  }
  if (zt) __syncthreads();
  const f32x2 pa = zt ? f32x2{pre_s[wv * 2], pre_s[wv * 2 + 1]} : f32x2{0.f, 0.f};
  const f32x2 ps = zt ? f32x2{pre_s[8 + wv * 2], pre_s[8 + wv * 2 + 1]} : f32x2{0.f, 0.f};
  const int cx = g.xs - 2 + lane;
  const bool col_in = cx >= 0 && cx < W && g.cok;
  const float cm = col_in ? 1.f : 0.f;
  const bool ovalid = lane >= 2 && lane < 62 && cx < W && g.cok;
  const unsigned voff = col_in ? (unsigned)(cx * 4 + (g.ch & 3)) * (unsigned)ES : OOB;
  const unsigned vst = ovalid ? voff : OOB;
  const BufRsrc rz = make_rsrc(x1, NREC), ro = make_rsrc(pre, NREC);
  const int rows = g.ye - g.ys, nsteps = rows + 4;
  f32x2 pf[5];
#pragma unroll
  for (int d = 0; d < D; ++d) pf[d] = ld_pair<TA>(rz, voff, row_off(g, g.ys - 2 + d, H));
  f32x2 acc[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) acc[k] = f32x2{0.f, 0.f};
  f32x2 gs = f32x2{0.f, 0.f};
  f32x2 in[5], inn[5];
  {
    const int iy = g.ys - 2;
    const float rm = (iy >= 0 && iy < H) ? cm : 0.f;
    const f32x2 x1v = (zt ? hswish2(pf[0], pa, ps) : pf[0]) * rm;
    pf[D % 5] = ld_pair<TA>(rz, voff, row_off(g, g.ys - 2 + D, H));
    XS0[lane + 2] = x1v;
    LMN_WAVE_SYNC();
    in[0] = XS0[lane]; in[1] = XS0[lane + 1]; in[2] = x1v; in[3] = XS0[lane + 3]; in[4] = XS0[lane + 4];
  }
#define LMN_STEP(P)                                                                                                \
  {                                                                                                                \
    const int j = j0 + P;                                                                                          \
    const int iyn = g.ys - 1 + j;                                                                                  \
    f32x2 x1n = (zt ? hswish2(pf[(P + 1) % 5], pa, ps) : pf[(P + 1) % 5]) * ((iyn >= 0 && iyn < H) ? cm : 0.f);    \
    pf[(P + 1 + D) % 5] = ld_pair<TA>(rz, voff, row_off(g, g.ys - 1 + j + D, H));                                   \
    f32x2* XS = XS0 + ((P + 1) & 1) * 68;                                                                          \
    LMN_WAVE_SYNC();                                                                                               \
    XS[lane + 2] = x1n;                                                                                            \
    LMN_WAVE_SYNC();                                                                                               \
    inn[0] = XS[lane]; inn[1] = XS[lane + 1]; inn[2] = x1n; inn[3] = XS[lane + 3]; inn[4] = XS[lane + 4];          \
    _Pragma("unroll") for (int d = 0; d < 5; ++d) {                                                                \
      if (LMN_DWF_ASM) {                                                                                           \
        if (d == 0) fpk_mul(acc[P], in[0], w[0]);                                                                  \
        else fpk_fma(acc[P], in[d], w[d]);                                                                         \
        _Pragma("unroll") for (int ky = 1; ky < 5; ++ky) fpk_fma(acc[(P - ky + 5) % 5], in[d], w[ky * 5 + d]);      \
      } else {                                                                                                     \
        if (d == 0) acc[P] = w[0] * in[0];                                                                         \
        else acc[P] += w[d] * in[d];                                                                               \
        _Pragma("unroll") for (int ky = 1; ky < 5; ++ky) acc[(P - ky + 5) % 5] += w[ky * 5 + d] * in[d];           \
      }                                                                                                            \
    }                                                                                                              \
    constexpr int DD = (P + 1) % 5;                                                                                \
    if (j >= 4 && j < nsteps) {                                                                                    \
      const f32x2 pv = acc[DD] + bias;                                                                             \
      st_pair<TA>(ro, vst, row_off(g, g.ys + j - 4, H), pv);                                                       \
      gs += gelu2(pv);                                                                                                \
    }                                                                                                              \
    _Pragma("unroll") for (int d = 0; d < 5; ++d) in[d] = inn[d];                                                  \
    LMN_SB();                                                                                                      \
  }
  for (int j0 = 0; j0 < nsteps; j0 += 5) { LMN_STEP(0) LMN_STEP(1) LMN_STEP(2) LMN_STEP(3) LMN_STEP(4) }
#undef LMN_STEP
  {
    const f32x2 v = wave_total(ovalid ? gs : f32x2{0.f, 0.f});
    if (lane == 63) { gs_s[wv * 2] = v.x; gs_s[wv * 2 + 1] = v.y; }
  }
  __syncthreads();  // the block's 8 channel sums leave as ONE atomic instruction (single-lane atomics per wave queue up in L2)
  if (SE.ticket == nullptr) {
    // (deterministic mode: gsum addresses slot copies of [B][E], one slot per (segment, strip))
    if (tid < 8 && g.ch0 + tid < E) lmn_red_add(gsum + (det ? (int64_t)(g.seg * strips + g.strip) * nimg * E : 0) + (int64_t)g.b * E + g.ch0 + tid, gs_s[tid], det);
    return;
  }
  // ---- squeeze-excite gate of image b by the block that completes its sums (lmn_se_fuse_t).  Hand-off: wave 0 adds this
  // block's sums with RETURNING atomics -- a no-return add is acknowledged (vmcnt) before it is performed at the memory side,
  // and under load a later reader saw the sum without it (one image's gate off by 5e-4, once in ~50 steps); the returned value
  // arrives only after the add has been performed.  Then its lane 0 draws a ticket; the block that draws the last one reads
  // the sums back with returning atomics too (performed where the adds were: no cache can hold an older value).
  if (tid < 64) {
    if (tid < 8 && g.ch0 + tid < E) {
      const float old = __hip_atomic_fetch_add(gsum + (int64_t)g.b * E + g.ch0 + tid, gs_s[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("" :: "v"(old) : "memory");   // the add has returned: it is in the sum every later reader sees
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0) s_last = atomicAdd(SE.ticket + g.b, 1u) == (unsigned)(strips * segs * chunks - 1) ? 1 : 0;
  }
  __syncthreads();
  if (!s_last) return;
  float* m = scr;        // [E]
  float* h = scr + E;    // [R]   (E + R <= 2048: checked on the host)
  const int R = SE.R;
  for (int e = tid; e < E; e += 256)
    m[e] = __hip_atomic_fetch_add(gsum + (int64_t)g.b * E + e, 0.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * SE.inv_hw;
  __syncthreads();
  // (this tail is serial latency at the end of the kernel: one thread per hidden unit walking E loads cost 13 us at E = 192.
  // G = 256 / R threads per hidden unit, each a strided part of the row with 8 loads in flight, partials through LDS.)
  float* part = scr + E + R;   // [256]  (E + R + 256 <= 2048: checked on the host)
  for (int r0 = 0; r0 < R; r0 += 256) {
    const int G = R - r0 >= 256 ? 1 : 256 / (R - r0);
    const int r = r0 + tid / G, q = tid - (tid / G) * G;
    float a = 0.f;
    if (r < R) {
#pragma unroll 8
      for (int e = q; e < E; e += G) a += SE.w1[(int64_t)r * E + e] * m[e];
    }
    part[tid] = a;
    __syncthreads();
    if (q == 0 && r < R) {
      float v = SE.b1[r];
      for (int k = 0; k < G; ++k) v += part[tid + k];
      v = v > 0.f ? v : 0.f;
      h[r] = v;
      SE.hidden[(int64_t)g.b * R + r] = v;
    }
    __syncthreads();
  }
  for (int e = tid; e < E; e += 256) {
    float a = SE.b2[e];
#pragma unroll 8
    for (int r = 0; r < R; ++r) a += SE.w2[(int64_t)e * R + r] * h[r];
    SE.s[(int64_t)g.b * E + e] = lmn_hsigmoid(a);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Backward pass 1: dpre = (u*s + dm)*gelu'(pre) -> store; bst[5][E] += (sum dpre, sum dpre*y_b).  The branch outputs are not
// recomputed: sum_p dpre[p] * y_b[p] = sum_t w_b[t] * G[t] with G[t] = sum_p dpre[p] * x1[p + t] (the 5x5 correlation of dpre with x1,
// of which the 3x3 / 3x1 / 1x3 taps are subsets): 25 packed FMAs per pixel pair, contracted with the four kernels once per wave.
// dpre row o (segment row) is formed at step o -- when x1 row o+2 (ring row o) is in `in` -- and pairs with x1 ring rows o .. o+4
// over the next five steps: G[ky][kx] += dpre[j - ky] * x1ring[j][x + kx - 2] at step j.
template <typename TA, bool ZT>
__global__ __launch_bounds__(256, 3) void dw_stats1_kernel(const TA* __restrict__ x1, const TA* __restrict__ pre, const TA* __restrict__ u,
                                                            const float* __restrict__ sgate, const float* __restrict__ dm, TA* __restrict__ dpre,
                                                            int H, int W, int E, const float* __restrict__ w5, const float* __restrict__ w3,
                                                            const float* __restrict__ wvv, const float* __restrict__ whh, float* __restrict__ bst,
                                                            const lmn_se_bwd_t SB, const DwPreK PRE, int strips, int segs, int seg_rows,
                                                            int chunks, int det) {
  constexpr int ES = sizeof(TA), D = 3;
  __shared__ f32x2 XSa[4][2][68];
  __shared__ float red[5 * 8];
  __shared__ __attribute__((aligned(16))) float pre_s[16];
  __shared__ float scr[4096];   // squeeze-excite backward: dt [E], da [R], partial sums [256]
  LMN_DW_SETPRIO;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const GeoP g = decode_pair<ES>(E, H, W, strips, segs, seg_rows, chunks, wv);
  constexpr bool zt = ZT;
  if (zt) dw_pre_setup(PRE, pre_s, g.ch0, E, tid, false);
  f32x2* XS0 = XSa[wv][0];
  if (lane < 8) XS0[(lane >> 2) * 68 + ((lane & 3) < 2 ? (lane & 3) : 64 + (lane & 3))] = f32x2{0.f, 0.f};
  f32x2 sv = g.cok ? f32x2{sgate[(int64_t)g.b * E + g.ch], sgate[(int64_t)g.b * E + g.ch + 1]} : f32x2{0.f, 0.f};
  if (zt) __syncthreads();
  const f32x2 pa = zt ? f32x2{pre_s[wv * 2], pre_s[wv * 2 + 1]} : f32x2{0.f, 0.f};
  const f32x2 ps = zt ? f32x2{pre_s[8 + wv * 2], pre_s[8 + wv * 2 + 1]} : f32x2{0.f, 0.f};
  const int cx = g.xs - 2 + lane;
  const bool col_in = cx >= 0 && cx < W && g.cok;
  const float cm = col_in ? 1.f : 0.f;
  const bool ovalid = lane >= 2 && lane < 62 && cx < W && g.cok;
  const float om = ovalid ? 1.f : 0.f;
  const unsigned voff = col_in ? (unsigned)(cx * 4 + (g.ch & 3)) * (unsigned)ES : OOB;
  const unsigned vst = ovalid ? voff : OOB;
  const BufRsrc rz = make_rsrc(x1, NREC), rp = make_rsrc(pre, NREC), ru = make_rsrc(u, NREC), ro = make_rsrc(dpre, NREC);
  const int rows = g.ye - g.ys, nsteps = rows + 4;
  f32x2 pf[5], pp[5], pu[5];
#pragma unroll
  for (int d = 0; d < D; ++d) {
    pf[d] = ld_pair<TA>(rz, voff, row_off(g, g.ys - 2 + d, H));
    const unsigned so = row_off(g, g.ys + d, H);
    pp[d] = ld_pair<TA>(rp, vst, so);
    pu[d] = ld_pair<TA>(ru, vst, so);
  }
  // (the first rows are in flight while the squeeze-excite vector below is formed)
  f32x2 dv = f32x2{0.f, 0.f};
  if (SB.ds) {
    // squeeze-excite backward of image b (lmn_se_bwd_t, lmn_se_bwd_dm arithmetic), formed by every block for its own channels:
    //   dt[e] = ds[b][e] * hardsigmoid'(.) (1/6 where 0 < s < 1);  da[r] = relu'(h[r]) * sum_e w2[e][r] * dt[e];
    //   dm[e] = inv_hw * sum_r w1[r][e] * da[r]
    const int R = SB.R;
    float* dt = scr;            // [E]
    float* da = scr + E;        // [R]
    float* part = scr + E + R;  // [256]
    const bool writer = g.ch0 == 0 && g.strip == 0 && g.seg == 0;   // one block per image keeps dt / da for lmn_se_bwd_params
    for (int e = tid; e < E; e += 256) {
      const float gt = sgate[(int64_t)g.b * E + e];
      const float d = (gt > 0.f && gt < 1.f) ? SB.ds[(int64_t)g.b * E + e] * (1.f / 6.f) : 0.f;
      dt[e] = d;
      if (writer) SB.dvec[(int64_t)g.b * (E + R) + e] = d;
    }
    __syncthreads();
    const int groups = R <= 256 ? 256 / R : 1;
    for (int r0 = 0; r0 < R; r0 += 256) {
      const int gq = tid / R, r = r0 + (R <= 256 ? tid - gq * R : tid);
      float a = 0.f;
      if (gq < groups && r < R) {
#pragma unroll 8
        for (int e = gq; e < E; e += groups) a += SB.w2[(int64_t)e * R + r] * dt[e];
      }
      part[tid] = a;
      __syncthreads();
      if (tid < R - r0 && tid < 256) {
        float v = 0.f;
        if (R <= 256) { for (int k = 0; k < groups; ++k) v += part[k * R + tid]; }
        else v = part[tid];
        const int rr = r0 + tid;
        v = SB.hidden[(int64_t)g.b * R + rr] > 0.f ? v : 0.f;
        da[rr] = v;
        if (writer) SB.dvec[(int64_t)g.b * (E + R) + E + rr] = v;
      }
      __syncthreads();
    }
    // dm of the wave's channel pair: lanes over r (a chain of 2 R dependent loads before: this prologue delays every block's first row)
    float d2[2] = {0.f, 0.f};
    if (g.cok) {
      for (int r = lane; r < R; r += 64) {
        const float d = da[r];
        d2[0] += SB.w1[(int64_t)r * E + g.ch] * d;
        d2[1] += SB.w1[(int64_t)r * E + g.ch + 1] * d;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { d2[0] += __shfl_xor(d2[0], o, 64); d2[1] += __shfl_xor(d2[1], o, 64); }
    dv = g.cok ? f32x2{d2[0] * SB.inv_hw, d2[1] * SB.inv_hw} : f32x2{0.f, 0.f};
  } else {
    dv = g.cok ? f32x2{dm[(int64_t)g.b * E + g.ch], dm[(int64_t)g.b * E + g.ch + 1]} : f32x2{0.f, 0.f};
  }
  const f32x2 z2 = f32x2{0.f, 0.f};
  f32x2 G[25], hist[5], sum0 = z2;
#pragma unroll
  for (int k = 0; k < 25; ++k) G[k] = z2;
#pragma unroll
  for (int k = 0; k < 5; ++k) hist[k] = z2;
  f32x2 in[5], inn[5];
  {
    const int iy = g.ys - 2;
    const float rm = (iy >= 0 && iy < H) ? cm : 0.f;
    const f32x2 x1v = (zt ? hswish2(pf[0], pa, ps) : pf[0]) * rm;
    pf[D % 5] = ld_pair<TA>(rz, voff, row_off(g, g.ys - 2 + D, H));
    XS0[lane + 2] = x1v;
    LMN_WAVE_SYNC();
    in[0] = XS0[lane]; in[1] = XS0[lane + 1]; in[2] = x1v; in[3] = XS0[lane + 3]; in[4] = XS0[lane + 4];
  }
#define LMN_STEP(P)                                                                                                \
  {                                                                                                                \
    const int j = j0 + P;                                                                                          \
    const int iyn = g.ys - 1 + j;                                                                                  \
    f32x2 x1n = (zt ? hswish2(pf[(P + 1) % 5], pa, ps) : pf[(P + 1) % 5]) * ((iyn >= 0 && iyn < H) ? cm : 0.f);    \
    const f32x2 pv = pp[P], uv = pu[P];                                                                            \
    pf[(P + 1 + D) % 5] = ld_pair<TA>(rz, voff, row_off(g, g.ys - 1 + j + D, H));                                   \
    { const unsigned so = row_off(g, g.ys + j + D, H); pp[(P + D) % 5] = ld_pair<TA>(rp, vst, so); pu[(P + D) % 5] = ld_pair<TA>(ru, vst, so); } \
    f32x2* XS = XS0 + ((P + 1) & 1) * 68;                                                                          \
    LMN_WAVE_SYNC();                                                                                               \
    XS[lane + 2] = x1n;                                                                                            \
    LMN_WAVE_SYNC();                                                                                               \
    inn[0] = XS[lane]; inn[1] = XS[lane + 1]; inn[2] = x1n; inn[3] = XS[lane + 3]; inn[4] = XS[lane + 4];          \
    const bool rok = j < rows;                                                                                     \
    f32x2 d = (uv * sv + dv) * dgelu2(pv) * (rok ? om : 0.f);                                                   \
    st_pair<TA>(ro, rok ? vst : OOB, row_off(g, g.ys + j, H), d);                                                  \
    sum0 += d;                                                                                                     \
    hist[P] = d;                                                                                                   \
    _Pragma("unroll") for (int ky = 0; ky < 5; ++ky)                                                               \
      _Pragma("unroll") for (int kx = 0; kx < 5; ++kx) G[ky * 5 + kx] += hist[(P - ky + 5) % 5] * in[kx];          \
    _Pragma("unroll") for (int dd = 0; dd < 5; ++dd) in[dd] = inn[dd];                                             \
    LMN_SB();                                                                                                      \
  }
  for (int j0 = 0; j0 < nsteps; j0 += 5) { LMN_STEP(0) LMN_STEP(1) LMN_STEP(2) LMN_STEP(3) LMN_STEP(4) }
#undef LMN_STEP
  f32x2 sum[5];
  sum[0] = sum0;
  sum[1] = sum[2] = sum[3] = sum[4] = z2;
  {  // sum dpre * y_b = <w_b, G> (taps of the small kernels embedded in the 5x5 window as in dw_merge)
    BranchW bw;
    load_branch_w(bw, w5, w3, wvv, whh, g.ch, g.cok);
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
  for (int k = 0; k < 5; ++k) {
    const f32x2 v = wave_total(sum[k]);   // (lanes outside the strip / image carry dpre = 0)
    if (lane == 63) { red[k * 8 + wv * 2] = v.x; red[k * 8 + wv * 2 + 1] = v.y; }
  }
  __syncthreads();
  if (tid < 40) {
    const int k = tid / 8, c8 = tid - k * 8;
    // (deterministic mode: bst addresses slot copies of [5][E], one slot per (image, segment, strip))
    if (g.ch0 + c8 < E) lmn_red_add(bst + (det ? (int64_t)((g.b * segs + g.seg) * strips + g.strip) * 5 * E : 0) + (int64_t)k * E + g.ch0 + c8, red[tid], det);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Backward pass 2:  f_b = cA_b*dpre + cC_b*y_b + cD_b inside the image (0 outside);
//   dx1 = sum_b corr^T(f_b, w_b);   dW_b[t] += sum_p f_b[p] * x1[p+t];   z-path: dh = dx1 * Hardswish'(A z + shift), hstats += (sum dh, sum dh z)
// 120 FMAs per element and 138 accumulator values per channel.  The two halves of every packed register are TWO ROW SEGMENTS
// (sa, sa + segs/2) of ONE channel and strip, so the 40 weights and 12 BatchNorm coefficients are plain scalars -- and the packed
// FMAs are written as asm, because hipcc never folds a scalar splat into op_sel of an SGPR operand: it builds an {w, w} SGPR pair
// per weight (104 SGPRs, spilled to VGPR lanes and read back by v_readlane before every use: 57 per row step in the channel-pair
// form of this kernel).  Two weights share one SGPR pair; op_sel / op_sel_hi broadcast the chosen half into both lanes of the
// packed operation (44 + 12 SGPRs).  Packed results need one wait state before a dependent VALU read: consecutive statements never
// chain (different accumulators), the few that would are separated by s_nop 0.
// Per step j (P = j mod 5): z rows (ys-4+j) of both segments enter; f rows (ys-6+j); dx rows (ys-8+j) complete; dW products of x1 rows
// (ys-8+j) against the f history.  lane l = column xs-2+l (HALO: lanes 0..3 also load the four extra x1 columns xs-4, xs-3, xs+62,
// xs+63 of the 68-entry LDS row, so a strip yields 60 outputs) or xs-4+l (56 outputs, no extra load: chosen when it costs no strip).
// block = 4 waves = the 4 channels of one quad (they share every cache line of the row-planar layout).
__device__ __forceinline__ void pkfma(f32x2& acc, f32x2 x, f32x2 wp, int hi) {   // acc += x * wp[hi]  (hi folds after unrolling)
  if (hi) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(x), "s"(wp));
  else asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(x), "s"(wp));
}
__device__ __forceinline__ void pkmul(f32x2& acc, f32x2 x, f32x2 wp, int hi) {   // acc = x * wp[hi]
  if (hi) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(acc) : "v"(x), "s"(wp));
  else asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(acc) : "v"(x), "s"(wp));
}
// the same with the weight pair in a VGPR pair (the 3x3 / 3x1 / 1x3 kernels: 18 SGPRs fewer -- the scalar file was over-subscribed
// by ~70 values, every spilled one a v_readlane in the loop)
#ifndef LMN_DW_WSGPR
#define LMN_DW_WSGPR 0
#endif
#if LMN_DW_WSGPR
#define LMN_WC "s"
#else
#define LMN_WC "v"
#endif
__device__ __forceinline__ void pkfmaV(f32x2& acc, f32x2 x, f32x2 wp, int hi) {
  if (hi) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(x), LMN_WC(wp));
  else asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(x), LMN_WC(wp));
}
__device__ __forceinline__ void pkmulV(f32x2& acc, f32x2 x, f32x2 wp, int hi) {
  if (hi) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(acc) : "v"(x), LMN_WC(wp));
  else asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(acc) : "v"(x), LMN_WC(wp));
}
__device__ __forceinline__ void pkfma_vv(f32x2& acc, f32x2 x, f32x2 y) { asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(y)); }
#define LMN_NOP0() asm volatile("s_nop 0")
struct W2 { f32x2 w5[13], w3[5], wv[2], wh[2]; };   // tap t of a kernel: pair t >> 1, half t & 1
#define LMN_W5(t) bw.w5[(t) >> 1], (t) & 1
#define LMN_W3(t) bw.w3[(t) >> 1], (t) & 1
#define LMN_WV(t) bw.wv[(t) >> 1], (t) & 1
#define LMN_WH(t) bw.wh[(t) >> 1], (t) & 1
struct SwState {
  f32x2 a5[5], a3[5], av[5], ah[5];          // y_b accumulators, slot = row index mod 5
  f32x2 dxa[5];                              // dx1 row accumulators, slot = row index mod 5
  f32x2 g5[25], g3[9], gv[3], gh[3];         // weight-gradient accumulators
};

// PART: 0 = dx1 and the weight gradients in one pass, 1 = dx1 only, 2 = weight gradients only
// Round 5: the step in a form that issues fewer non-FMA instructions (ISA of the round-4 loop: 645 v_pk_fma of 1045 VALU per five
// steps -- 130 v_pk_mul, 80 v_mov_dpp, 76 v_readlane / v_writelane of spilled scalars, 54 v_cndmask, 61 s_nop):
//   * f row q = j - 3 (one step later than before) is formed FIRST, from branch accumulators that were completed by the previous
//     step, as two packed FMAs per branch -- dp * cA' + cD', then + a_b * cC' -- with the BatchNorm-backward coefficients as per-lane
//     VGPR pairs that carry the column mask (cA' = cA * [column inside the image], ...): no mask multiplies, 12 SGPRs fewer;
//   * its column neighbours (f5 +-1, +-2; f3 +-1; fh +-1) travel through ds_bpermute_b32 -- the LDS crossbar, no VALU issue slot
//     (v_pk_fma_f32 takes no DPP operand: the 16 v_mov_dpp per step were pure issue overhead) -- requested right after f is formed
//     and consumed by the dx1 FMAs at the END of the step, behind the 40 branch-output and 40 weight-gradient FMAs;
//   * the row masks (rows outside the image / outside the wave's own segment) are wave-uniform: interior steps skip them by ONE
//     scalar branch (msk_on), only the own-row mask of the weight-gradient history (h = f * mo) stays a multiply.
// Rows by step j (P = j mod 5; slots are row mod 5): x1 row j enters (ring slot P, whose previous content -- row j - 5 -- was read
// as i2 before); branch outputs accumulate into rows j - 2 .. j + 2; f row j - 3; weight-gradient products of x1 row j - 5 against
// f rows j - 3 .. j - 7; dx rows j - 5 .. j - 1, of which row j - 5 is complete after this step.
// f = dp * sA[hi] + vD[hi]: the scalar coefficient cA_b from an SGPR pair (two branches per pair), cD_b' = cD_b * column mask from a
// per-lane VGPR pair (two branches per pair)
__device__ __forceinline__ void pkfma_cA(f32x2& f, f32x2 dp, f32x2 sA, f32x2 vD, int hi) {
  if (hi) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,1] op_sel_hi:[1,1,1]" : "=v"(f) : "v"(dp), "s"(sA), "v"(vD));
  else asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(f) : "v"(dp), "s"(sA), "v"(vD));
}
__device__ __forceinline__ void pkfma_cC(f32x2& f, f32x2 a, f32x2 vC, int hi) {   // f += a * vC[hi]  (cC_b' = cC_b * column mask, per lane)
  if (hi) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(f) : "v"(a), "v"(vC));
  else asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(f) : "v"(a), "v"(vC));
}
__device__ __forceinline__ f32x2 pk_hsw_affine(f32x2 z, f32x2 as) {   // z * as.lo + as.hi  (A' / shift' of the lane in ONE VGPR pair)
  f32x2 x;
  asm("v_pk_fma_f32 %0, %1, %2, %2 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "=v"(x) : "v"(z), "v"(as));
  return x;
}
__device__ __forceinline__ f32x2 hswish2_as(f32x2 z, f32x2 as) {
  const f32x2 x = pk_hsw_affine(z, as);
  f32x2 t = x * (1.f / 6.f) + 0.5f;
  t.x = __builtin_amdgcn_fmed3f(t.x, 0.f, 1.f);
  t.y = __builtin_amdgcn_fmed3f(t.y, 0.f, 1.f);
  return x * t;
}
__device__ __forceinline__ f32x2 bperm2(int addr, f32x2 v) {   // lane l <- lane (addr / 4) mod 64 (both halves)
  f32x2 r;
  r.x = __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(v.x)));
  r.y = __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(v.y)));
  return r;
}
struct SwCoef { f32x2 sA[2], vC[2], vD[2]; };   // {cA_0, cA_1}, {cA_2, cA_3} (scalars); cC / cD likewise, times the lane's column mask
struct SwF { f32x2 sh[5], s3[3], hr, hl, fv, fh; };   // f row q and its column neighbours: sh[k][l] = f5[l + 2 - k], s3[k][l] = f3[l + 1 - k]

// first half of a step: f row q = j - 3, its neighbours (requested), the weight-gradient products of x1 row j - 5 (i2)
__device__ __forceinline__ void pkmul_s2(f32x2& d, f32x2 x, f32x2 sp) {   // d = x * {sp.lo, sp.hi}: a per-half scalar mask from an SGPR pair
  asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(x), "s"(sp));
}
// msk_on (wave-uniform): a step that touches rows outside the image (mf: the f row of each half); mo: the f row lies in the wave's own
// segment (weight-gradient history; always applied: it is the one mask whose off / on forms would need two registers per value)
// f row q (slot QS = q mod 5 of the branch accumulators, all four complete) and its column neighbours, requested through the LDS
// crossbar: formed at the END of a step for the NEXT one, so that the round trip hides behind the loop edge and the next step's 40
// branch-output FMAs.  mf (wave-uniform, boundary steps only): the f row of each half lies inside the image.
template <int QS>
__device__ __forceinline__ void sw_form_f(const SwState& S, const SwCoef& CF, SwF& F, f32x2 dp, f32x2 mf, bool msk_on, int bp0) {
  typedef f32x2 V;
  V f5, f3, fv, fh;
  pkfma_cA(f5, dp, CF.sA[0], CF.vD[0], 0); pkfma_cA(f3, dp, CF.sA[0], CF.vD[0], 1);
  pkfma_cA(fv, dp, CF.sA[1], CF.vD[1], 0); pkfma_cA(fh, dp, CF.sA[1], CF.vD[1], 1);
  pkfma_cC(f5, S.a5[QS], CF.vC[0], 0); pkfma_cC(f3, S.a3[QS], CF.vC[0], 1); pkfma_cC(fv, S.av[QS], CF.vC[1], 0); pkfma_cC(fh, S.ah[QS], CF.vC[1], 1);
  LMN_NOP0();
  if (msk_on) { pkmul_s2(f5, f5, mf); pkmul_s2(f3, f3, mf); pkmul_s2(fv, fv, mf); pkmul_s2(fh, fh, mf); LMN_NOP0(); }   // (rows outside the image)
#ifndef LMN_DW_ALWAYS_MSK
#define LMN_DW_ALWAYS_MSK 0   // A/B builds: 1 = every step applies the row masks (no wave-uniform branches in the loop)
#endif
#ifndef LMN_DW_NB
#define LMN_DW_NB 1   // how the column neighbours of f travel: 1 = DPP wave shifts (16 VALU slots per step; level 0 alone 203 us), 0 = ds_bpermute_b32 (LDS crossbar, no VALU slot: 209-218 us -- the kernel is not issue-bound at two waves per SIMD)
#endif
#if LMN_DW_NB == 1
  F.sh[1] = lane_from_right(f5); F.sh[0] = lane_from_right(F.sh[1]); F.sh[3] = lane_from_left(f5); F.sh[4] = lane_from_left(F.sh[3]);
  F.s3[0] = lane_from_right(f3); F.s3[2] = lane_from_left(f3);
  F.hr = lane_from_right(fh); F.hl = lane_from_left(fh);
  (void)bp0;
#else
  // bp0 = 4 * ((lane - 2) & 63): lanes -2, -1, +1, +2 at +0, +4, +12, +16 (byte address + instruction offset, modulo the wave)
  F.sh[4] = bperm2(bp0, f5); F.sh[3] = bperm2(bp0 + 4, f5); F.sh[1] = bperm2(bp0 + 12, f5); F.sh[0] = bperm2(bp0 + 16, f5);
  F.s3[2] = bperm2(bp0 + 4, f3); F.s3[0] = bperm2(bp0 + 12, f3);
  F.hl = bperm2(bp0 + 4, fh); F.hr = bperm2(bp0 + 12, fh);
#endif
  F.sh[2] = f5; F.s3[1] = f3; F.fv = fv; F.fh = fh;
}
// weight gradients of f row q = j - 3 against ALL its x1 rows q - 2 .. q + 2 = j - 5 .. j - 1:
//   dW_b[ky][kx] = sum_x f_b[q][x] x1[q + ky - 2][x + kx - 2] = sum_x' f_b[q][x' + 2 - kx] x1[q + ky - 2][x']
// -- the SHIFTED copies of f that the dx1 FMAs need anyway (F.sh / F.s3 / F.hr / F.hl) against the wave's OWN column of the last five
// x1 rows (xo[ky] = row j - 5 + ky: four plain LDS reads of the ring, requested at the top of the step, and the row the ring has just
// lost, kept in a register).  No f history (40 VGPRs in the round-4 form), no shifted LDS read; lane l then holds the products of f
// column l + 2 - kx, and the lanes whose f column is not one of the strip's own are dropped per kx in the final reduction.  mo (boundary steps): the f row lies inside the wave's own segment.
template <int P>
__device__ __forceinline__ void sw_step_dw(SwState& S, const SwF& F, const f32x2 (&xo)[5], f32x2 mo, bool msk_on) {
  typedef f32x2 V;
#pragma unroll
  for (int ky = 0; ky < 5; ++ky) {
    V X = xo[ky];
    if (msk_on) { pkmul_s2(X, X, mo); LMN_NOP0(); }
#pragma unroll
    for (int kx = 0; kx < 5; ++kx) pkfma_vv(S.g5[ky * 5 + kx], F.sh[kx], X);
    if (ky >= 1 && ky <= 3) {
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) pkfma_vv(S.g3[(ky - 1) * 3 + kx], F.s3[kx], X);
      pkfma_vv(S.gv[ky - 1], F.fv, X);
    }
    if (ky == 2) { pkfma_vv(S.gh[0], F.hr, X); pkfma_vv(S.gh[1], F.fh, X); pkfma_vv(S.gh[2], F.hl, X); }
  }
}
// dx1: f row q = j - 3 feeds dx rows q - 2 .. q + 2 = j - 5 .. j - 1 (slots P .. P + 4; row j - 1 starts here, row j - 5 completes)
template <int P>
__device__ __forceinline__ void sw_step_dx(SwState& S, const W2& bw, const SwF& F) {
  constexpr int Q = (P + 2) % 5;
#pragma unroll
  for (int kx = 0; kx < 5; ++kx) {
    if (kx == 0) pkmul(S.dxa[(P + 4) % 5], F.sh[0], LMN_W5(20));
    else pkfma(S.dxa[(P + 4) % 5], F.sh[kx], LMN_W5(20 + kx));
#pragma unroll
    for (int ky = 0; ky < 4; ++ky) pkfma(S.dxa[(P + ky) % 5], F.sh[kx], LMN_W5(ky * 5 + kx));
  }
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) pkfmaV(S.dxa[(P + ky + 1) % 5], F.s3[kx], LMN_W3(ky * 3 + kx));
  }
  // 3x1 and 1x3: interleaved so that no statement reads the accumulator its predecessor wrote
  pkfmaV(S.dxa[(P + 2) % 5], F.fv, LMN_WV(1));
  pkfmaV(S.dxa[(P + 1) % 5], F.fv, LMN_WV(0));
  pkfmaV(S.dxa[Q], F.hr, LMN_WH(0));
  pkfmaV(S.dxa[(P + 3) % 5], F.fv, LMN_WV(2));
  pkfmaV(S.dxa[Q], F.fh, LMN_WH(1));
  LMN_NOP0();   // (the one chain of the step: two consecutive statements into the same accumulator)
  pkfmaV(S.dxa[Q], F.hl, LMN_WH(2));
}
// x1 row j: the four branch outputs y_b (column-major: consecutive FMAs go to DIFFERENT row accumulators; the first contribution to
// a row is a plain product: no zeroed accumulators)
template <int P>
__device__ __forceinline__ void sw_step_y(SwState& S, const W2& bw, const f32x2 (&in)[5]) {
#pragma unroll
  for (int d = 0; d < 5; ++d) {
    if (d == 0) pkmul(S.a5[(P + 2) % 5], in[0], LMN_W5(0));
    else pkfma(S.a5[(P + 2) % 5], in[d], LMN_W5(d));
#pragma unroll
    for (int ky = 1; ky < 5; ++ky) pkfma(S.a5[(P - ky + 7) % 5], in[d], LMN_W5(ky * 5 + d));
    if (d >= 1 && d <= 3) {
      if (d == 1) { pkmulV(S.a3[(P + 1) % 5], in[1], LMN_W3(0)); pkmulV(S.ah[P], in[1], LMN_WH(0)); }
      else { pkfmaV(S.a3[(P + 1) % 5], in[d], LMN_W3(d - 1)); pkfmaV(S.ah[P], in[d], LMN_WH(d - 1)); }
#pragma unroll
      for (int ky = 1; ky < 3; ++ky) pkfmaV(S.a3[(P - ky + 6) % 5], in[d], LMN_W3(ky * 3 + d - 1));
    }
    if (d == 2) {
      pkmulV(S.av[(P + 1) % 5], in[2], LMN_WV(0));
#pragma unroll
      for (int ky = 1; ky < 3; ++ky) pkfmaV(S.av[(P - ky + 6) % 5], in[2], LMN_WV(ky));
    }
  }
}

template <typename TA, int PART, bool HALO, bool ZT, int WPS>
__global__ __launch_bounds__(256, WPS) void dw_bwd_kernel(
    const TA* __restrict__ x1, const TA* __restrict__ dpre, TA* __restrict__ dx1, int B, int H, int W, int E,
    const float* __restrict__ w5, const float* __restrict__ w3, const float* __restrict__ wvv, const float* __restrict__ whh,
    const float* __restrict__ cA, const float* __restrict__ cC, const float* __restrict__ cD, const DwCoef CF, float* __restrict__ dw5,
    float* __restrict__ dw3, float* __restrict__ dwv, float* __restrict__ dwh, const DwPreS PRE, float* __restrict__ hstats, int strips,
    int segs /* even */, int seg_rows, int chunks /* quads */, int det) {
  typedef f32x2 V;
  constexpr int ES = sizeof(TA), D = 3, D2 = 2;   // rows in flight (z / dpre) beyond the next step's; 3 costs four more VGPRs (the prefetch rings), and the kernel sits at the 256 of two waves per SIMD
  constexpr int QS = HALO ? 60 : 56;
  __shared__ V XSa[4][5 * 68];
  __shared__ V ZSa[4][ZT && PART != 2 ? 5 * 64 : 1];
  __shared__ float red[4 * 44];
  LMN_DW_SETPRIO;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  // decode: quad fastest, then strip, segment PAIR, image
  int lid = blockIdx.x;
  const int quad = lid % chunks; lid /= chunks;
  const int strip = lid % strips; lid /= strips;
  const int hs = segs >> 1;
  const int sa = lid % hs;
  const int b = lid / hs;
  const int ch = quad * 4 + wv;   // (E is a multiple of 4: every channel of the quad exists)
  const int ysA = sa * seg_rows, yeA = min(ysA + seg_rows, H);
  const int ysB = min((sa + hs) * seg_rows, H), yeB = min(ysB + seg_rows, H);   // (may be empty: rowsB = 0)
  const int rowsA = yeA - ysA, rowsB = yeB - ysB;
  const int xs = strip * QS;
  const unsigned rowb = (unsigned)(E * W) * (unsigned)ES, qoff = (unsigned)(quad * 4 * W) * (unsigned)ES;
  const int r0 = b * H;
  // row offsets (soffset of the buffer accesses): rows outside the image read a SAFE row of the half instead (one scalar select:
  // no clamps; their values are masked by rm / mf below), stores of rows outside the segment are skipped by a scalar branch
  const int HB = rowsB > 0 ? H : 0;
  const unsigned safeA = (unsigned)(r0 + ysA) * rowb + qoff, safeB = rowsB > 0 ? (unsigned)(r0 + ysB) * rowb + qoff : safeA;
  auto soA = [&](int iy) -> unsigned { return (unsigned)iy < (unsigned)H ? (unsigned)(r0 + iy) * rowb + qoff : safeA; };
  auto soB = [&](int iy) -> unsigned { return (unsigned)iy < (unsigned)HB ? (unsigned)(r0 + iy) * rowb + qoff : safeB; };
  V* XS = XSa[wv];
  V* ZS = ZSa[wv];
  for (int i = lane; i < 5 * 68; i += 64) XS[i] = V{0.f, 0.f};
  if (ZT && PART != 2) {   // (the drain multiplies masked rows by what the ring holds: it must be finite from the first step on)
    for (int i = lane; i < 5 * 64; i += 64) ZS[i] = V{0.f, 0.f};
  }
  W2 bw;
  auto wl = [&](const float* w, int NT, int t) -> float { return t < NT ? w[(int64_t)ch * NT + t] : 0.f; };
#pragma unroll
  for (int k = 0; k < 13; ++k) bw.w5[k] = V{wl(w5, 25, 2 * k), wl(w5, 25, 2 * k + 1)};
#pragma unroll
  for (int k = 0; k < 5; ++k) bw.w3[k] = V{wl(w3, 9, 2 * k), wl(w3, 9, 2 * k + 1)};
#pragma unroll
  for (int k = 0; k < 2; ++k) { bw.wv[k] = V{wl(wvv, 3, 2 * k), wl(wvv, 3, 2 * k + 1)}; bw.wh[k] = V{wl(whh, 3, 2 * k), wl(whh, 3, 2 * k + 1)}; }
  // (the 5x5 kernel in SGPR pairs, the three small kernels in VGPR pairs)
#if !LMN_DW_WSGPR
#pragma unroll
  for (int k = 0; k < 5; ++k) asm volatile("" : "+v"(bw.w3[k]));
#pragma unroll
  for (int k = 0; k < 2; ++k) asm volatile("" : "+v"(bw.wv[k]), "+v"(bw.wh[k]));
#endif
  float a4[4], c4[4], d4[4];   // BatchNorm-backward coefficients of f_b = cA dpre + cC y_b + cD (per channel: wave-uniform)
  {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (CF.bstats) {  // block-uniform: coefficients formed here (lmn_dw_bwd_coef arithmetic)
        const int i = k * E + ch;
        const float S0 = CF.bstats[ch], S1 = CF.bstats[(1 + k) * E + ch];
        const float mu = CF.mean[i], rs = CF.rstd[i], a = CF.A[i];
        const float T = (S1 - mu * S0) * rs;  // sum dpre * yhat_b
        const float c = CF.batch_stats ? -a * T * rs / CF.count : 0.f;
        a4[k] = a; c4[k] = c;
        d4[k] = CF.batch_stats ? (-a * S0 / CF.count - c * mu) : 0.f;
        if (PART != 2 && lane == 0 && strip == 0 && sa == 0 && b == 0) {  // one wave per channel
          CF.dg[k][ch] += T;
          CF.db[k][ch] += S0;
        }
      } else {
        a4[k] = cA[k * E + ch]; c4[k] = cC[k * E + ch]; d4[k] = cD[k * E + ch];
      }
    }
  }
  const float pa = ZT ? PRE.A[ch] : 0.f, ps = ZT ? PRE.shift[ch] : 0.f;
  const int cx = xs - (HALO ? 2 : 4) + lane;
  const bool col_in = cx >= 0 && cx < W;
  const float cm = col_in ? 1.f : 0.f;
  const int hx = lane < 2 ? xs - 4 + lane : xs + 60 + lane;
  const bool hcol_in = HALO && lane < 4 && hx >= 0 && hx < W;
  const float hm = hcol_in ? 1.f : 0.f;
  const int hidx = lane < 2 ? lane : 64 + lane;
  const bool own_col = HALO ? (lane >= 2 && lane < 62) : (lane >= 4 && lane < 60);
  const bool ovalid = own_col && cx < W;
  const unsigned voff = col_in ? (unsigned)(cx * 4 + (ch & 3)) * (unsigned)ES : OOB;
  const unsigned vhalo = hcol_in ? (unsigned)(hx * 4 + (ch & 3)) * (unsigned)ES : OOB;
  const unsigned vst = ovalid ? voff : OOB;
  const BufRsrc rz = make_rsrc(x1, NREC), rd = make_rsrc(dpre, NREC), ro = make_rsrc(dx1, NREC);
  // per-lane coefficient pairs with the column mask folded in (f_b = 0 in columns outside the image): VGPR operands of the two
  // packed FMAs that form f_b (sw_step), instead of 12 SGPRs + mask multiplies
  SwCoef CFv;
  {
    // (provably wave-uniform for the "s" operands of the asm statements)
    auto uni = [](float v) -> float { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); };
    CFv.sA[0] = V{uni(a4[0]), uni(a4[1])}; CFv.sA[1] = V{uni(a4[2]), uni(a4[3])};
    CFv.vC[0] = V{c4[0] * cm, c4[1] * cm}; CFv.vC[1] = V{c4[2] * cm, c4[3] * cm};
    CFv.vD[0] = V{d4[0] * cm, d4[1] * cm}; CFv.vD[1] = V{d4[2] * cm, d4[3] * cm};
    asm volatile("" : "+v"(CFv.vC[0]), "+v"(CFv.vC[1]), "+v"(CFv.vD[0]), "+v"(CFv.vD[1]));
  }
  // Hardswish(A z + shift) with the column mask folded into A / shift ({A', shift'} of the lane in one VGPR pair): x1 = 0 in columns
  // outside the image without a multiply
  V pas = V{pa * cm, ps * cm}, pash = V{pa * hm, ps * hm};
  asm volatile("" : "+v"(pas), "+v"(pash));
  // (without the transform the loaded tensor IS x1, and lanes outside the image load 0 through the out-of-range voffset)
  // column neighbours of f through ds_bpermute_b32 (byte address = 4 * source lane + instruction offset, modulo the wave; the
  // wrap-around lanes are halo lanes whose dx is never stored)
  const int bp0 = ((lane - 2) & 63) * 4;
  const int nsteps = rowsA + 9;   // rowsA >= rowsB; the last dx row (ys + rows - 1) completes at step rows + 8
  V pfz[5], pfh[5], pfd[5];
  auto ldz = [&](int j, V& zz, V& zh_) {
    const unsigned a = soA(ysA - 4 + j), bb = soB(ysB - 4 + j);
    zz = V{ld_one<TA>(rz, voff, a), ld_one<TA>(rz, voff, bb)};
    if (HALO) zh_ = V{ld_one<TA>(rz, vhalo, a), ld_one<TA>(rz, vhalo, bb)};
  };
  auto ldd = [&](int j, V& dd) { dd = V{ld_one<TA>(rd, voff, soA(ysA - 7 + j)), ld_one<TA>(rd, voff, soB(ysB - 7 + j))}; };
  // f rows of half h live at image rows [fl_h, H): inside the image and not above the first row the segment needs (ys - 2)
  const int flA = max(ysA - 2, 0), flB = max(ysB - 2, 0);
#pragma unroll
  for (int d = 0; d <= D; ++d) ldz(d, pfz[d], pfh[d]);       // z rows 0 .. D (row 0 is exchanged before the loop)
#pragma unroll
  for (int d = 1; d <= D2; ++d) ldd(d, pfd[d]);              // dpre rows of f rows formed at the end of steps 0 .. D2 - 1
  SwState S;
  const V z2 = V{0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 5; ++k) S.a5[k] = S.a3[k] = S.av[k] = S.ah[k] = S.dxa[k] = z2;
#pragma unroll
  for (int k = 0; k < 25; ++k) S.g5[k] = z2;
#pragma unroll
  for (int k = 0; k < 9; ++k) S.g3[k] = z2;
#pragma unroll
  for (int k = 0; k < 3; ++k) S.gv[k] = S.gh[k] = z2;
  V hs0 = z2, hs1 = z2;
  // interior steps: every row the step touches lies inside the image AND inside the wave's own segments, for both halves -- x1 row
  // ys-4+j, f row ys-7+j, dx row ys-9+j: j in [9, jhi).  They skip the row-mask blocks by ONE scalar compare; the conditions of the
  // boundary steps are evaluated inside those blocks only (their scalars stay out of the interior path).
  int jhi = rowsB + 7;
  if (H + 4 - ysA < jhi) jhi = H + 4 - ysA;
  if (H + 4 - ysB < jhi) jhi = H + 4 - ysB;
  if (rowsB <= 0) jhi = 0;
  const unsigned jn = jhi > 9 ? (unsigned)(jhi - 9) : 0u;
  auto mskf = [](bool a) -> float { return a ? 1.f : 0.f; };
  // ---- the step is software-pipelined over the loop edge: what a step needs through the LDS crossbar -- the shifted copies of f row
  //      j - 3 (F) and of x1 row j (inn) -- is requested at the END of the previous step (LMN_NEXT), so that both round trips hide
  //      behind the loop edge and the 40 branch-output FMAs that open a step.
  SwF F;
  F.sh[0] = F.sh[1] = F.sh[2] = F.sh[3] = F.sh[4] = F.s3[0] = F.s3[1] = F.s3[2] = F.hr = F.hl = F.fv = F.fh = z2;
  V inn[5];
  V xold = z2, xold2 = z2;   // own column of the x1 rows that left the ring last (at step j: rows j - 5 and, after X, j - 4)
  // x1 row JN (ring slot PN) from its prefetched z row: transform, row mask on boundary steps, exchange of the +-1 / +-2 columns.  The
  // row that leaves the ring (JN - 5) is read first (LDS operations of a wave execute in order).
#define LMN_X1ROW(JN, PN, MSKN)                                                                                    \
  {                                                                                                                \
    const V zv = pfz[PN];                                                                                          \
    V x1v = ZT ? hswish2_as(zv, pas) : zv;                                                                         \
    V x1h = z2;                                                                                                    \
    if (HALO) x1h = ZT ? hswish2_as(pfh[PN], pash) : pfh[PN];                                                      \
    if (MSKN) {   /* x1 row ys-4+JN inside the image */                                                            \
      const V rm = V{mskf((unsigned)(ysA - 4 + (JN)) < (unsigned)H), mskf((unsigned)(ysB - 4 + (JN)) < (unsigned)HB)}; \
      pkmul_s2(x1v, x1v, rm);                                                                                      \
      if (HALO) pkmul_s2(x1h, x1h, rm);                                                                            \
      LMN_NOP0();                                                                                                  \
    }                                                                                                              \
    xold2 = XS[(PN) * 68 + lane + 2];   /* (same address as the write below: program order holds without a fence) */ \
    XS[(PN) * 68 + lane + 2] = x1v;                                                                                \
    if (HALO) { if (lane < 4) XS[(PN) * 68 + hidx] = x1h; }                                                        \
    LMN_WAVE_SYNC();                                                                                               \
    { const V* xr = XS + (PN) * 68 + lane; inn[0] = xr[0]; inn[1] = xr[1]; inn[2] = x1v; inn[3] = xr[3]; inn[4] = xr[4]; } \
  }
  { const bool m0 = !((unsigned)(0 - 9) < jn); LMN_X1ROW(0, 0, m0) }   // (row 0: step 0 is a boundary step)
  // Order inside a step (A -> X -> B -> C -> D): the branch outputs of x1 row j (inn: requested a whole step ago); THEN the exchange of
  // row j+1 is requested (its round trip hides behind B .. D); the weight gradients and dx1 of f row j-3 (F: requested at the end of the
  // previous step, behind A); the drain of the completed dx row; f row j-2 for the next step.
#define LMN_STEP(P)                                                                                                \
  {                                                                                                                \
    const int j = j0 + P;                                                                                          \
    const bool msk_on = LMN_DW_ALWAYS_MSK || !((unsigned)(j - 9) < jn), msk_nx = LMN_DW_ALWAYS_MSK || !((unsigned)(j - 8) < jn);   /* this step's / the next step's rows */ \
    /* f row ys-7+j inside the wave's own segment; dx row ys-9+j inside it (per half; wave-uniform) */             \
    const bool oA = (unsigned)(j - 7) < (unsigned)rowsA, oB = (unsigned)(j - 7) < (unsigned)rowsB;                 \
    const bool okA = (unsigned)(j - 9) < (unsigned)rowsA, okB = (unsigned)(j - 9) < (unsigned)rowsB;               \
    ldz(j + 1 + D, pfz[(P + 1 + D) % 5], pfh[(P + 1 + D) % 5]);                                                    \
    ldd(j + 1 + D2, pfd[(P + 1 + D2) % 5]);                                                                        \
    /* own column of x1 rows j-5 .. j-1 for the weight gradients: rows j-5 / j-4 from registers (they left / leave the ring), */ \
    /* rows j-3 .. j-1 from the ring                                                                                        */ \
    V xo[5];                                                                                                       \
    xo[0] = xold;                                                                                                  \
    if (PART != 1) { _Pragma("unroll") for (int ky = 2; ky < 5; ++ky) xo[ky] = XS[((P + ky) % 5) * 68 + lane + 2]; } \
    /* A: x1 row j (inn) -> the branch outputs; row j-2 of every branch is complete after it */                    \
    sw_step_y<P>(S, bw, inn);                                                                                      \
    /* X: x1 row j+1 into ring slot P+1 (whose row j-4 is kept: xold2); its shifted copies are requested for the next step */ \
    LMN_X1ROW(j + 1, (P + 1) % 5, msk_nx)                                                                          \
    xo[1] = xold2;                                                                                                 \
    xold = xold2;                                                                                                  \
    /* B: f row j-3 (F): weight gradients against x1 rows j-5 .. j-1 (xo), dx rows j-5 .. j-1 */                   \
    if (PART != 1) sw_step_dw<P>(S, F, xo, V{mskf(oA), mskf(oB)}, msk_on);                                         \
    if (PART != 2) sw_step_dx<P>(S, bw, F);                                                                        \
    /* C: dx row j-5 (image row ys-9+j) is complete */                                                             \
    if (PART != 2) {                                                                                               \
      V dv = S.dxa[P];                                                                                             \
      if (ZT) {   /* dh = dx1 * Hardswish'(A z + shift); sum dh, sum dh * z over the rows of the two segments */   \
        const V zr = ZS[P * 64 + lane];   /* z row j-5; then z row j takes the slot (same address: program order holds) */ \
        ZS[P * 64 + lane] = pfz[P];                                                                                \
        const V hh = zr * pa + ps;                                                                                 \
        dv = dv * V{lmn_dhswish(hh.x), lmn_dhswish(hh.y)};                                                         \
        if (msk_on) { pkmul_s2(dv, dv, V{mskf(okA), mskf(okB)}); LMN_NOP0(); }                                     \
        hs0 += dv;                                                                                                 \
        hs1 += dv * zr;                                                                                            \
      }                                                                                                            \
      if (okA) st_one<TA>(ro, vst, (unsigned)(r0 + ysA - 9 + j) * rowb + qoff, dv.x);                              \
      if (okB) st_one<TA>(ro, vst, (unsigned)(r0 + ysB - 9 + j) * rowb + qoff, dv.y);                              \
    }                                                                                                              \
    /* D: f row j-2 for step j+1 (its branch outputs were completed by A); the neighbours are requested behind the loop edge */ \
    {                                                                                                              \
      V mf = z2;                                                                                                   \
      if (msk_nx) {   /* f row ys-6+j inside the image and not above the first row the segment needs */            \
        const int fyA = ysA - 6 + j, fyB = ysB - 6 + j;                                                            \
        mf = V{mskf((unsigned)(fyA - flA) < (unsigned)(H - flA)), mskf((unsigned)(fyB - flB) < (unsigned)(HB - flB) && rowsB > 0)}; \
      }                                                                                                            \
      sw_form_f<(P + 3) % 5>(S, CFv, F, pfd[(P + 1) % 5], mf, msk_nx, bp0);                                        \
    }                                                                                                              \
    LMN_SB();                                                                                                      \
  }
  for (int j0 = 0; j0 < nsteps; j0 += 5) { LMN_STEP(0) LMN_STEP(1) LMN_STEP(2) LMN_STEP(3) LMN_STEP(4) }
#undef LMN_STEP
#undef LMN_X1ROW
  // ---- reductions: both halves add up; wave totals by DPP (the total lands in lane 63), then one atomic per (tap, channel)
  float* rw = red + wv * 44;
  auto put = [&](V v, int t, bool ok) {
    const float mt = wave_total(ok ? v.x + v.y : 0.f);   // (the halo lanes of the strip belong to its neighbours)
    if (lane == 63) rw[t] = mt;
  };
  if (ZT && PART != 2) { put(hs0, 40, ovalid); put(hs1, 41, ovalid); }
  if (PART != 1) {
    // lane l of g[..][kx] holds the products of f column l + 2 - kx (5x5) / l + 1 - kx (3x3, 1x3) / l (3x1): kept where that column is
    // one of the strip's own
    const int lo = HALO ? 2 : 4, hi = HALO ? 61 : 59;
    auto ownf = [&](int l) -> bool { return l >= lo && l <= hi; };
#pragma unroll
    for (int t = 0; t < 25; ++t) put(S.g5[t], t, ownf(lane + 2 - t % 5));
#pragma unroll
    for (int t = 0; t < 9; ++t) put(S.g3[t], 25 + t, ownf(lane + 1 - t % 3));
#pragma unroll
    for (int t = 0; t < 3; ++t) { put(S.gv[t], 34 + t, ownf(lane)); put(S.gh[t], 37 + t, ownf(lane + 1 - t)); }
  }
  __syncthreads();
  // (deterministic mode: hstats / dw5 .. dwh address slot copies of the destinations, one slot per (image, segment pair, strip))
  const int64_t slot = det ? (int64_t)((b * hs + sa) * strips + strip) : 0;
  for (int i = tid; i < 4 * 44; i += 256) {
    const int w = i / 44, t = i - w * 44;
    const int e = quad * 4 + w;
    if (t >= 42) continue;
    const float v = red[i];
    if (t >= 40) { if (ZT && PART != 2) lmn_red_add(hstats + slot * 2 * E + (int64_t)(t - 40) * E + e, v, det); }
    else if (PART == 1) continue;
    else if (t < 25) lmn_red_add(dw5 + (slot * E + e) * 25 + t, v, det);
    else if (t < 34) lmn_red_add(dw3 + (slot * E + e) * 9 + t - 25, v, det);
    else if (t < 37) lmn_red_add(dwv + (slot * E + e) * 3 + t - 34, v, det);
    else lmn_red_add(dwh + (slot * E + e) * 3 + t - 37, v, det);
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

// Row segments of a pass.  A wave walks (rows + halo) row steps (padded to batches of 5); `per_seg` waves exist per segment (pair); a
// SIMD holds `wps` waves and the machine 1024 SIMDs.  A row step of ONE wave is a dependent chain (load -> transform -> LDS
// exchange -> FMAs) that takes about `lat` times its issue time, so up to `lat` waves per SIMD run for free; beyond that the pass
// is issue-bound: time ~ steps x max(waves per SIMD, lat) while all waves are resident, waves / 1024 x steps plus a tail past
// that.  (Level 3, 44 x 44 x 192 channels: one segment = 768 waves of 50 steps took 31 us, four segments of 15 steps 12 us; level 0
// of the backward: 2 segment pairs = 2304 waves on 2048 slots ran as two rounds.)
static int dw_segments(int64_t per_seg, int H, int halo, int wps, int mult, double lat, int* seg_rows) {
  static double lat_scale = -1.0;   // LMN_DW_LAT: scales `lat` (A/B runs of the segment counts inside the step)
  if (lat_scale < 0.0) { const char* e = getenv("LMN_DW_LAT"); lat_scale = e ? atof(e) : 1.0; }
  lat *= lat_scale;
  int best = mult;
  double best_cost = -1.0;
  for (int sg = mult; sg <= H; sg += mult) {
    const int rows = lmn_cdiv(H, sg);
    if (sg > mult && rows < 6) break;
    const int nseg = lmn_cdiv(H, rows);
    if (lmn_cdiv(nseg, mult) * mult != sg) continue;  // same partition as a smaller count
    const int steps = lmn_cdiv(rows + halo, 5) * 5;
    const double waves = (double)per_seg * (sg / mult);
    const double per_simd = waves / 1024.0;
    double occ = waves <= 1024.0 * wps ? (double)lmn_cdiv((int64_t)waves, 1024) : per_simd + 0.5 * wps;
    if (occ < lat) occ = lat;
    const double cost = occ * steps;
    if (best_cost < 0.0 || cost < best_cost) { best_cost = cost; best = sg; }
  }
  *seg_rows = lmn_cdiv(H, best);
  return best;
}

template <int MODE>
static int launch_dw_stats(const void* x1, const void* pre, const void* u, const float* s, const float* dm,
                           void* dpre, int B, int H, int W, int E, const float* w5, const float* w3,
                           const float* wv, const float* wh, float* stats, const lmn_se_bwd_t& sb, const lmn_dw_pre_t& zp,
                           int act_dtype, hipStream_t st) {
  LMN_REQUIRE((int64_t)B * H * W * E * (act_dtype == LMN_BF16 ? 2 : 4) < (1LL << 32), "dw statistics: the tensor (%d x %d x %d x %d) must stay below 4 GiB (row offsets travel in the 32-bit soffset)", B, H, W, E);
  const int strips = lmn_cdiv(W, QW), chunks = lmn_cdiv(E, 8);
  int seg_rows;
  const int segs = dw_segments((int64_t)B * strips * chunks * 4, H, 4, MODE == 0 ? 4 : 3, 1, 3.0, &seg_rows);
  const int nseg = lmn_cdiv(H, seg_rows);
  const int64_t nblk = (int64_t)B * strips * chunks * nseg;
  (void)segs;
  if (nblk >= (1LL << 31)) return -1;
  // row A2 is priced at 2 tensor passes forward + 3 backward (lmn_dw_fwd / lmn_dw_bwd declare them).  The two BatchNorm
  // statistics passes are extra passes of this implementation: they declare ZERO algorithmic bytes (their time lowers row A2's
  // achieved fraction; what they physically move -- 1 resp. 4 tensor passes -- is the PMC `traffic` of the bench line)
  if (g_lmn_prof_on) lmn_prof_cost(2.0 * 42 * (double)B * H * W * E, 0.0);
  float* sdst = stats;
  const int nslots = B * nseg * strips, NS = MODE == 0 ? 8 : 5;
  if (g_lmn_det) {   // per-block sums into slot copies of [NS][E], folded in fixed order below
    lmn_det_begin(st);
    sdst = lmn_det_slots(st, (size_t)nslots * NS * E);
    LMN_REQUIRE(sdst, "dw statistics: deterministic mode: no scratch");
  }
  if (MODE == 0) {
    const bool zt = zp.A != nullptr || zp.fin.mode == LMN_FIN_BN;
#define LMN_DS0(Z) LMN_ACT_DISPATCH(act_dtype, LMN_LAUNCH((dw_stats0_kernel<T, Z>), dim3((unsigned)nblk), dim3(256), 0, st, (const T*)x1, H, W, E, w5, w3, wv, wh, sdst, zp, strips, nseg, \
                                          seg_rows, chunks, DW_DET(st)))
    if (zt) LMN_DS0(true); else LMN_DS0(false);
#undef LMN_DS0
  } else {
    const bool zt = zp.A != nullptr;
#define LMN_DS1(Z) LMN_ACT_DISPATCH(act_dtype, LMN_LAUNCH((dw_stats1_kernel<T, Z>), dim3((unsigned)nblk), dim3(256), 0, st, (const T*)x1, (const T*)pre, (const T*)u, s, dm, (T*)dpre, H, W, E, \
                                          w5, w3, wv, wh, sdst, sb, zp, strips, nseg, seg_rows, chunks, DW_DET(st)))
    if (zt) LMN_DS1(true); else LMN_DS1(false);
#undef LMN_DS1
  }
  if (g_lmn_det) lmn_det_sum(st, sdst, nslots, (int64_t)NS * E, stats);
  return 0;
}


extern "C" {

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
  LMN_REQUIRE(E + se->R + 256 <= 2048, "%s: E + R = %d exceeds the block's scratch", what, E + se->R);
  *out = *se;
  return 0;
}

static int dw_fwd_launch(const void* x1, void* pre, float* gsum, int B, int H, int W, int E, const float* keff, const float* beff,
                         const DwFin& fn, const lmn_se_fuse_t& sf, const lmn_dw_pre_t& zp, int act_dtype, hipStream_t st, const char* what) {
  LMN_REQUIRE((int64_t)B * H * W * E * (act_dtype == LMN_BF16 ? 2 : 4) < (1LL << 32), "%s: the tensor (%d x %d x %d x %d) must stay below 4 GiB (row offsets travel in the 32-bit soffset)", what, B, H, W, E);
  // blocks = B x row segments x strips (60 output columns) x 8-channel chunks
  const int strips = lmn_cdiv(W, QW), chunks = lmn_cdiv(E, 8);
  int seg_rows;
  dw_segments((int64_t)B * strips * chunks * 4, H, 4, LMN_DWF_OCC, 1, 3.0, &seg_rows);
  const int segs = lmn_cdiv(H, seg_rows);
  const int64_t nblk = (int64_t)B * strips * chunks * segs;
  LMN_REQUIRE(nblk < (1LL << 31), "%s: grid too large", what);
  if (g_lmn_prof_on) lmn_prof_cost(2.0 * 25 * (double)B * H * W * E, (act_dtype == LMN_BF16 ? 2.0 : 4.0) * 2 * (double)B * H * W * E);
  LMN_REQUIRE(!g_lmn_det || !sf.ticket, "%s: deterministic mode takes the squeeze-excite sums without the fused gate (lmn_se_fuse_t NULL)", what);
  float* gdst = gsum;
  if (g_lmn_det) {   // per-block sums into slot copies of [B][E]
    lmn_det_begin(st);
    gdst = lmn_det_slots(st, (size_t)segs * strips * B * E);
    LMN_REQUIRE(gdst, "%s: deterministic mode: no scratch", what);
  }
#define LMN_DF(Z) LMN_ACT_DISPATCH(act_dtype, LMN_LAUNCH((dw_fwd_kernel<T, Z>), dim3((unsigned)nblk), dim3(256), 0, st, (const T*)x1, (T*)pre, gdst, H, W, E, keff, beff, fn, sf, \
                                        DwPreS{zp.A, zp.shift}, strips, segs, seg_rows, chunks, DW_DET(st), B))
  if (zp.A) LMN_DF(true); else LMN_DF(false);
#undef LMN_DF
  if (g_lmn_det) lmn_det_sum(st, gdst, segs * strips, (int64_t)B * E, gsum);
  return lmn_launch_status(what);
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
  DwFin fn;
  memset(&fn, 0, sizeof(fn));
  return dw_fwd_launch(x1, pre, gsum, B, H, W, E, keff, beff, fn, sf, zp, act_dtype, (hipStream_t)stream, "dw_fwd");
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
  DwFin fn;
  fn.stats = stats; fn.count = count;
  for (int b = 0; b < 4; ++b) {
    LMN_REQUIRE(gamma[b] && beta[b] && running_mean[b] && running_var[b], "dw_fwd_bn: null BatchNorm tensor %d", b);
    fn.gamma[b] = gamma[b]; fn.beta[b] = beta[b]; fn.rmean[b] = running_mean[b]; fn.rvar[b] = running_var[b];
    fn.eps[b] = eps[b]; fn.mom[b] = momentum[b];
  }
  fn.w5 = w5; fn.w3 = w3; fn.wv = wv; fn.wh = wh; fn.mean = mean; fn.rstd = rstd; fn.A = A;
  auto launch = [=]() -> int {
    return dw_fwd_launch(x1, pre, gsum, B, H, W, E, nullptr, nullptr, fn, sf, zp, act_dtype, (hipStream_t)stream, "dw_fwd_bn");
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
  const int rc = launch_dw_stats<0>(x1, nullptr, nullptr, nullptr, nullptr, nullptr, B, H, W, E, w5, w3, wv, wh, stats, lmn_se_bwd_t{}, zp, act_dtype, (hipStream_t)stream);
  if (rc) return rc;
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
    LMN_REQUIRE(E + seb->R + 256 <= 4096, "dw_bwd_stats: E + R = %d exceeds the block's scratch", E + seb->R);
    sb = *seb;
  }
  if (g_lmn_rec) lmn_rec_push([=]() -> int { return lmn_dw_bwd_stats(x1, pre, u, s, dm, dpre, B, H, W, E, w5, w3, wv, wh, bstats, &sb, &zp, act_dtype, stream); }, "lmn_dw_bwd_stats(");
  LMN_REQUIRE_DT(act_dtype, "dw_bwd_stats");
  LMN_REQUIRE(x1 && pre && u && s && (dm || sb.ds) && dpre && w5 && w3 && wv && wh && bstats, "dw_bwd_stats: null pointer");
  LMN_REQUIRE(B > 0 && H > 0 && W > 0 && E > 0 && E % 4 == 0, "dw_bwd_stats: E=%d must be a multiple of 4", E);
  const int rc = launch_dw_stats<1>(x1, pre, u, s, dm, dpre, B, H, W, E, w5, w3, wv, wh, bstats, sb, zp, act_dtype, (hipStream_t)stream);
  if (rc) return rc;
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

// the backward launch: strips of 60 (HALO) or 56 output columns, segment PAIRS, quads
static int dw_bwd_launch(const void* x1, const void* dpre, void* dx1, int B, int H, int W, int E, const float* w5, const float* w3, const float* wv,
                         const float* wh, const float* cA, const float* cC, const float* cD, const DwCoef& cf, float* dw5, float* dw3, float* dwv,
                         float* dwh, int part, const lmn_dw_pre_t& zp, float* hstats, int act_dtype, hipStream_t st, const char* what) {
  LMN_REQUIRE((int64_t)B * H * W * E * (act_dtype == LMN_BF16 ? 2 : 4) < (1LL << 32), "%s: the tensor (%d x %d x %d x %d) must stay below 4 GiB (row offsets travel in the 32-bit soffset)", what, B, H, W, E);
  const bool halo = lmn_cdiv(W, 60) < lmn_cdiv(W, 56);   // the four extra x1 columns only where they save a strip
  const int strips = halo ? lmn_cdiv(W, 60) : lmn_cdiv(W, 56), chunks = E / 4;
  const int wps = part == 1 ? 3 : 2;   // (dx1 alone: three waves per SIMD -- round 5: at the 128 VGPRs of four it spills)
  int seg_rows;
  const int segs = dw_segments((int64_t)B * strips * chunks * 4, H, 10, wps, 2, part == 1 ? 3.0 : 2.0, &seg_rows);   // even: a wave walks the pair (sa, sa + segs/2)
  const int hs = segs / 2;
  const int64_t nblk = (int64_t)B * strips * chunks * hs;
  LMN_REQUIRE(nblk < (1LL << 31), "%s: grid too large", what);
  if (g_lmn_prof_on) lmn_prof_cost(2.0 * 2 * 42 * (double)B * H * W * E, (act_dtype == LMN_BF16 ? 2.0 : 4.0) * 3 * (double)B * H * W * E);
  float *g5 = dw5, *g3 = dw3, *gv = dwv, *gh = dwh, *hs_ = hstats;
  const int nslots = B * hs * strips;
  if (g_lmn_det) {   // per-block sums into slot copies of the four weight-gradient tensors and of hstats [2][E]
    lmn_det_begin(st);
    float* base = lmn_det_slots(st, (size_t)nslots * 42 * E);
    LMN_REQUIRE(base, "%s: deterministic mode: no scratch", what);
    g5 = base; g3 = g5 + (size_t)nslots * 25 * E; gv = g3 + (size_t)nslots * 9 * E; gh = gv + (size_t)nslots * 3 * E;
    if (hstats) hs_ = gh + (size_t)nslots * 3 * E;
  }
#define LMN_DWB4(PT, HL, Z, WP) LMN_ACT_DISPATCH(act_dtype, LMN_LAUNCH((dw_bwd_kernel<T, PT, HL, Z, WP>), dim3((unsigned)nblk), dim3(256), 0, st, (const T*)x1, (const T*)dpre, \
                       (T*)dx1, B, H, W, E, w5, w3, wv, wh, cA, cC, cD, cf, g5, g3, gv, gh, DwPreS{zp.A, zp.shift}, hs_, strips, segs, seg_rows, chunks, DW_DET(st)))
#define LMN_DWB3(PT, HL, WP) do { if (zp.A) LMN_DWB4(PT, HL, true, WP); else LMN_DWB4(PT, HL, false, WP); } while (0)
#define LMN_DWB2(PT, WP) do { if (halo) LMN_DWB3(PT, true, WP); else LMN_DWB3(PT, false, WP); } while (0)
  if (part == 1) { LMN_DWB2(1, 3); } else if (part == 2) { LMN_DWB2(2, 2); } else { LMN_DWB2(0, 2); }
#undef LMN_DWB4
#undef LMN_DWB3
#undef LMN_DWB2
  if (g_lmn_det) {
    if (part != 1) {
      lmn_det_sum(st, g5, nslots, (int64_t)25 * E, dw5); lmn_det_sum(st, g3, nslots, (int64_t)9 * E, dw3);
      lmn_det_sum(st, gv, nslots, (int64_t)3 * E, dwv); lmn_det_sum(st, gh, nslots, (int64_t)3 * E, dwh);
    }
    if (hstats && part != 2) lmn_det_sum(st, hs_, nslots, (int64_t)2 * E, hstats);
  }
  return lmn_launch_status(what);
}

int lmn_dw_bwd(const void* x1, const void* dpre, void* dx1, int B, int H, int W, int E, const float* w5,
               const float* w3, const float* wv, const float* wh, const float* cA, const float* cC, const float* cD,
               float* dw5, float* dw3, float* dwv, float* dwh, int act_dtype, lmn_stream_t stream) {
  LMN_REC(lmn_dw_bwd(x1, dpre, dx1, B, H, W, E, w5, w3, wv, wh, cA, cC, cD, dw5, dw3, dwv, dwh, act_dtype, stream));
  LMN_REQUIRE_DT(act_dtype, "dw_bwd");
  LMN_REQUIRE(x1 && dpre && dx1 && w5 && w3 && wv && wh && cA && cC && cD && dw5 && dw3 && dwv && dwh, "dw_bwd: null pointer");
  LMN_REQUIRE(B > 0 && H > 0 && W > 0 && E > 0 && E % 4 == 0, "dw_bwd: E=%d must be a multiple of 4", E);
  DwCoef cf;
  memset(&cf, 0, sizeof(cf));
  lmn_dw_pre_t zp;
  memset(&zp, 0, sizeof(zp));
  return dw_bwd_launch(x1, dpre, dx1, B, H, W, E, w5, w3, wv, wh, cA, cC, cD, cf, dw5, dw3, dwv, dwh, 0, zp, nullptr, act_dtype, (hipStream_t)stream, "dw_bwd");
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
  DwCoef cf;
  cf.bstats = bstats; cf.mean = mean; cf.rstd = rstd; cf.A = A; cf.count = count; cf.batch_stats = batch_stats;
  for (int b = 0; b < 4; ++b) {
    LMN_REQUIRE(dgamma[b] && dbeta[b], "dw_bwd_bn: null gradient tensor %d", b);
    cf.dg[b] = dgamma[b]; cf.db[b] = dbeta[b];
  }
  auto launch = [=]() -> int {
    return dw_bwd_launch(x1, dpre, dx1, B, H, W, E, w5, w3, wv, wh, nullptr, nullptr, nullptr, cf, dw5, dw3, dwv, dwh, part, zp, hstats, act_dtype,
                         (hipStream_t)stream, "dw_bwd_bn");
  };
  if (g_lmn_rec) lmn_rec_push(launch, "lmn_dw_bwd_bn(");
  return launch();
}

}  // extern "C"
