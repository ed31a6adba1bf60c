// Row-wise / element-wise HBM-bound kernels of the LM-Net path: LayerNorm, BatchNorm(+GELU) tails,
// BatchNorm bookkeeping, SE gate, bilinear x2 resampling, pyramid mean-pool, layout converters and
// small utilities.  All are float4-vectorised along the NHWC channel axis (every channel count on the
// path is a multiple of 4) with lanes walking the channel axis first so a wave touches contiguous bytes.
// Per-channel reductions keep a fixed channel quad per thread, reduce in registers over a grid-stride
// loop, then LDS atomics, then one global atomic per channel per block (guide: G12).
#include "common.h"

namespace {


// ------------------------------------------------------------------------------------ LayerNorm
// G lanes cooperate on one row; lane j owns float4 slots j, j+G (C/4 <= 2G).
template <int G, typename T>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, T* __restrict__ y,
                                                     int64_t rows, int C) {
  const int C4 = C >> 2;
  const int j = threadIdx.x % G;
  const int rpb = 256 / G;
  const bool has0 = j < C4, has1 = j + G < C4;
  f32x4 g0 = f32x4{0, 0, 0, 0}, g1 = g0, b0 = g0, b1 = g0;
  if (has0) { g0 = ld4(gamma + j * 4); b0 = ld4(beta + j * 4); }
  if (has1) { g1 = ld4(gamma + (j + G) * 4); b1 = ld4(beta + (j + G) * 4); }
  const float invC = 1.0f / (float)C;
  const int64_t nit = (rows + (int64_t)gridDim.x * rpb - 1) / ((int64_t)gridDim.x * rpb);
  for (int64_t it = 0; it < nit; ++it) {
    const int64_t row = (it * gridDim.x + blockIdx.x) * rpb + threadIdx.x / G;
    const bool rok = row < rows;  // keep all lanes in the shuffles
    const T* xr = x + row * C;
    f32x4 v0 = f32x4{0, 0, 0, 0}, v1 = v0;
    if (rok && has0) v0 = ld4(xr + j * 4);
    if (rok && has1) v1 = ld4(xr + (j + G) * 4);
    float s = v0[0] + v0[1] + v0[2] + v0[3] + v1[0] + v1[1] + v1[2] + v1[3];
#pragma unroll
    for (int m = G >> 1; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
    const float mean = s * invC;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (has0) q += (v0[k] - mean) * (v0[k] - mean);
      if (has1) q += (v1[k] - mean) * (v1[k] - mean);
    }
#pragma unroll
    for (int m = G >> 1; m >= 1; m >>= 1) q += __shfl_xor(q, m, 64);
    const float rstd = rsqrtf(q * invC + 1e-5f);
    if (rok && has0) st4(y + row * C + j * 4, (v0 - mean) * rstd * g0 + b0);
    if (rok && has1) st4(y + row * C + (j + G) * 4, (v1 - mean) * rstd * g1 + b1);
  }
}

template <int G, typename T>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                     const T* __restrict__ dy, const T* __restrict__ dres,
                                                     T* __restrict__ dx, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta, int64_t rows, int C, int det) {
  extern __shared__ float red[];  // [4 waves][2][C]
  const int C4 = C >> 2;
  const int j = threadIdx.x % G;
  const int rpb = 256 / G;
  const bool has0 = j < C4, has1 = j + G < C4;
  f32x4 g0 = f32x4{0, 0, 0, 0}, g1 = g0;
  if (has0) g0 = ld4(gamma + j * 4);
  if (has1) g1 = ld4(gamma + (j + G) * 4);
  f32x4 ag0 = f32x4{0, 0, 0, 0}, ag1 = ag0, ab0 = ag0, ab1 = ag0;
  const float invC = 1.0f / (float)C;
  const int64_t nit = (rows + (int64_t)gridDim.x * rpb - 1) / ((int64_t)gridDim.x * rpb);
  for (int64_t it = 0; it < nit; ++it) {
    const int64_t row = (it * gridDim.x + blockIdx.x) * rpb + threadIdx.x / G;
    const bool rok = row < rows;
    f32x4 v0 = f32x4{0, 0, 0, 0}, v1 = v0, d0 = v0, d1 = v0;
    if (rok && has0) { v0 = ld4(x + row * C + j * 4); d0 = ld4(dy + row * C + j * 4); }
    if (rok && has1) { v1 = ld4(x + row * C + (j + G) * 4); d1 = ld4(dy + row * C + (j + G) * 4); }
    float s = v0[0] + v0[1] + v0[2] + v0[3] + v1[0] + v1[1] + v1[2] + v1[3];
#pragma unroll
    for (int m = G >> 1; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
    const float mean = s * invC;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (has0) q += (v0[k] - mean) * (v0[k] - mean);
      if (has1) q += (v1[k] - mean) * (v1[k] - mean);
    }
#pragma unroll
    for (int m = G >> 1; m >= 1; m >>= 1) q += __shfl_xor(q, m, 64);
    const float rstd = rsqrtf(q * invC + 1e-5f);
    const f32x4 h0 = (v0 - mean) * rstd, h1 = (v1 - mean) * rstd;
    const f32x4 t0 = d0 * g0, t1 = d1 * g1;
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (has0) { m1 += t0[k]; m2 += t0[k] * h0[k]; }
      if (has1) { m1 += t1[k]; m2 += t1[k] * h1[k]; }
    }
#pragma unroll
    for (int m = G >> 1; m >= 1; m >>= 1) {
      m1 += __shfl_xor(m1, m, 64);
      m2 += __shfl_xor(m2, m, 64);
    }
    m1 *= invC;
    m2 *= invC;
    if (rok && has0) {
      f32x4 o = (t0 - m1 - h0 * m2) * rstd;
      if (dres) o += ld4(dres + row * C + j * 4);
      st4(dx + row * C + j * 4, o);
      ag0 += d0 * h0;
      ab0 += d0;
    }
    if (rok && has1) {
      f32x4 o = (t1 - m1 - h1 * m2) * rstd;
      if (dres) o += ld4(dres + row * C + (j + G) * 4);
      st4(dx + row * C + (j + G) * 4, o);
      ag1 += d1 * h1;
      ab1 += d1;
    }
  }
  // dgamma / dbeta: the row slots of a wave (lanes with equal j differ in the lane bits >= log2 G) fold by shuffles,
  // the four waves through plain LDS stores, then ONE global atomic per channel per block.  (Per-thread LDS float
  // atomics on 2C shared addresses plus ~1000 same-address global atomics per channel put a ~40 us floor under
  // every call, whatever the size of the tensor.)
#pragma unroll
  for (int m = G; m < 64; m <<= 1) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      ag0[k] += __shfl_xor(ag0[k], m, 64); ab0[k] += __shfl_xor(ab0[k], m, 64);
      ag1[k] += __shfl_xor(ag1[k], m, 64); ab1[k] += __shfl_xor(ab1[k], m, 64);
    }
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float* wred = red + wv * 2 * C;  // [4 waves][2][C]
  if (lane < G) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (has0) { wred[j * 4 + k] = ag0[k]; wred[C + j * 4 + k] = ab0[k]; }
      if (has1) { wred[(j + G) * 4 + k] = ag1[k]; wred[C + (j + G) * 4 + k] = ab1[k]; }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    const float v = red[i] + red[2 * C + i] + red[4 * C + i] + red[6 * C + i];
    // (deterministic mode: dgamma / dbeta address slot copies [blocks][C] of the two vectors)
    lmn_red_add((i < C ? dgamma : dbeta - C) + (det ? (int64_t)blockIdx.x * C : 0) + i, v, det);
  }
}

// ------------------------------------------------------------------------------------ BN(+act) tails
template <typename T>
__global__ __launch_bounds__(256) void bnact_fwd_kernel(const T* __restrict__ z, const float* __restrict__ a,
                                                        const float* __restrict__ b, T* __restrict__ y,
                                                        int64_t n4, int C4, int act) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C4) * 4;
    const f32x4 v = ld4(z + i * 4), aa = ld4(a + c), bb = ld4(b + c);
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = lmn_act(v[k] * aa[k] + bb[k], act);
    st4(y + i * 4, o);
  }
}

// The same with the BatchNorm finalize inside (lmn_bn_fin_t, LMN_FIN_BN: lmn_bn_finalize arithmetic): every block forms A / shift of
// all C channels in LDS before its loop; block 0 also stores mean / rstd / A / shift and blends the running statistics (F.about must
// not alias F.rmean: other blocks still read it).  One launch instead of two behind the statistics-writing conv of the skip fusers.
template <typename T>
__global__ __launch_bounds__(256) void bnact_fwd_fin_kernel(const T* __restrict__ z, const lmn_bn_fin_t F, T* __restrict__ y, int64_t n4,
                                                            int C4, int act) {
  extern __shared__ __attribute__((aligned(16))) float ab[];   // [2][C]
  const int C = C4 * 4;
  for (int c = threadIdx.x; c < C; c += 256) {
    double s0 = 0.0, s1 = 0.0;
    for (int r = 0; r < F.nrep; ++r) { s0 += (double)F.sums[r * 2 * C + c]; s1 += (double)F.sums[r * 2 * C + C + c]; }
    const double md = s0 / (double)F.count;
    float var = (float)(s1 / (double)F.count - md * md);  // biased
    var = var > 0.f ? var : 0.f;
    const float m = (float)md + (F.about ? F.about[c] : 0.f);
    const float rs = rsqrtf(var + F.eps);
    const float a = F.gamma[c] * rs, sh = F.beta[c] - m * a;
    ab[c] = a; ab[C + c] = sh;
    if (blockIdx.x == 0) {
      if (F.mean) F.mean[c] = m;
      if (F.rstd) F.rstd[c] = rs;
      if (F.A) F.A[c] = a;
      if (F.shift) F.shift[c] = sh;
      if (F.rmean) F.rmean[c] = (1.f - F.momentum) * F.rmean[c] + F.momentum * m;
      if (F.rvar) F.rvar[c] = (1.f - F.momentum) * F.rvar[c] + F.momentum * var * (F.count > 1.f ? F.count / (F.count - 1.f) : 1.f);
    }
  }
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C4) * 4;
    const f32x4 v = ld4(z + i * 4), aa = *reinterpret_cast<const f32x4*>(&ab[c]), bb = *reinterpret_cast<const f32x4*>(&ab[C + c]);
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = lmn_act(v[k] * aa[k] + bb[k], act);
    st4(y + i * 4, o);
  }
}

// Block-level sum of per-thread partials whose owner pattern is "thread t holds channel quad t % C4 of pixel slot t / C4"
// (C4 need not be a power of two, so no DPP tree): the NV float4 partials of every thread are parked in LDS ([NV*4][256]) and
// summed in two short chains (G groups of slots, then the G group sums) -- LDS atomics on the 4*C4 addresses serialised the
// whole block at its tail (2040 atomics on 24 addresses at C = 12: ~20 us of a 42 us kernel over its four block rounds).
// Result: red[v * C + c] for v < NV, c < C.  `park` holds NV*4*256 + NV*C*G floats.
template <int NV>
__device__ __forceinline__ void chan_block_reduce(const f32x4 (&s)[NV], float* park, float* red, int C, int C4, int NTH, int rpb, int tid) {
#pragma unroll
  for (int v = 0; v < NV; ++v)
#pragma unroll
    for (int k = 0; k < 4; ++k) park[(v * 4 + k) * 256 + tid] = tid < NTH ? s[v][k] : 0.f;
  __syncthreads();
  const int nred = NV * C;
  int G = 256 / nred;
  if (G < 1) G = 1;
  if (G > rpb) G = rpb;
  float* p2 = park + NV * 4 * 256;
  for (int i = tid; i < nred * G; i += 256) {
    const int o = i / G, g = i - o * G;
    const int v = o / C, ch = o - v * C, q = ch >> 2, k = ch & 3;
    float a = 0.f;
    for (int r = g; r < rpb; r += G) a += park[(v * 4 + k) * 256 + r * C4 + q];
    p2[i] = a;
  }
  __syncthreads();
  for (int o = tid; o < nred; o += 256) {
    float a = 0.f;
    for (int g = 0; g < G; ++g) a += p2[o * G + g];
    red[o] = a;
  }
  __syncthreads();
}

// MODE 0: stats[2][C] += (sum dh, sum dh*zhat);  MODE 1: dz = c1*dh - c2 - zhat*c3
// MODE 2: out[C] += column sums of x (bias gradients); x has pixel stride cstride
template <int MODE, typename T>
__global__ __launch_bounds__(256) void chan_kernel(const T* __restrict__ z, const T* __restrict__ dy,
                                                   const float* __restrict__ mean, const float* __restrict__ rstd,
                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                   const float* __restrict__ c1, const float* __restrict__ c2,
                                                   const float* __restrict__ c3, void* __restrict__ out_,
                                                   int64_t rows, int C, int cstride, int act, int det, const lmn_bn_fin_t F) {
  float* out = reinterpret_cast<float*>(out_);   // MODE 0 / 2: fp32 statistics; MODE 1: the activation tensor dz
  T* out_act = reinterpret_cast<T*>(out_);
  extern __shared__ float red[];  // [2][C] (MODE 0), [C] (MODE 2)
  const int C4 = C >> 2;
  const int NTH = (256 / C4) * C4;  // active threads: each keeps a fixed channel quad
  const int rpb = NTH / C4;
  const int tid = threadIdx.x;
  f32x4 s0 = f32x4{0, 0, 0, 0}, s1 = s0;
  if (tid < NTH) {
    const int c = (tid % C4) * 4;
    f32x4 mu = s0, rs = s0, ga = s0, be = s0, k1 = s0, k2 = s0, k3 = s0;
    if (MODE != 2) { mu = ld4(mean + c); rs = ld4(rstd + c); ga = ld4(gamma + c); be = ld4(beta + c); }
    if (MODE == 1) {
      if (F.mode == LMN_FIN_BN_BWD) {   // (uniform) c1 / c2 / c3 formed here from the statistics pass's sums (lmn_bn_bwd_coef arithmetic)
        f32x4 S0 = s0, S1 = s0;
        for (int r = 0; r < F.nrep; ++r) { S0 += ld4(F.sums + r * 2 * C + c); S1 += ld4(F.sums + r * 2 * C + C + c); }
        k1 = ld4(F.Ain + c);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          k2[k] = F.batch_stats ? k1[k] * S0[k] / F.count : 0.f;
          k3[k] = F.batch_stats ? k1[k] * S1[k] / F.count : 0.f;
        }
        if (blockIdx.x == 0 && tid < C4) {   // one writer per channel quad
          if (F.dgamma) { f32x4 g = ld4(F.dgamma + c); g += S1; *reinterpret_cast<f32x4*>(F.dgamma + c) = g; }
          if (F.dbeta) { f32x4 g = ld4(F.dbeta + c); g += S0; *reinterpret_cast<f32x4*>(F.dbeta + c) = g; }
        }
      } else { k1 = ld4(c1 + c); k2 = ld4(c2 + c); k3 = ld4(c3 + c); }
    }
    for (int64_t row = (int64_t)blockIdx.x * rpb + tid / C4; row < rows; row += (int64_t)gridDim.x * rpb) {
      if (MODE == 2) {
        s0 += ld4(z + row * cstride + c);
      } else {
        const f32x4 v = ld4(z + row * C + c), d = ld4(dy + row * C + c);
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float zh = (v[k] - mu[k]) * rs[k];
          const float dh = d[k] * lmn_dact(ga[k] * zh + be[k], act);
          if (MODE == 0) { s0[k] += dh; s1[k] += dh * zh; }
          else o[k] = k1[k] * dh - k2[k] - zh * k3[k];
        }
        if (MODE == 1) st4(out_act + row * C + c, o);
      }
    }
  }
  if (MODE != 1) {
    float* park = red + 2 * C;
    if (MODE == 0) {
      const f32x4 sv[2] = {s0, s1};
      chan_block_reduce<2>(sv, park, red, C, C4, NTH, rpb, tid);
    } else {
      const f32x4 sv[1] = {s0};
      chan_block_reduce<1>(sv, park, red, C, C4, NTH, rpb, tid);
    }
    const int nred = MODE == 0 ? 2 * C : C;
    // (deterministic mode: out addresses slot copies [blocks][nred])
    for (int i = tid; i < nred; i += 256) lmn_red_add(out + (det ? (int64_t)blockIdx.x * nred : 0) + i, red[i], det);
  }
}

__global__ void bn_finalize_kernel(const float* __restrict__ sums, int nrep, float count, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float eps, float momentum, float* mean, float* rstd,
                                   float* A, float* shift, float* rmean, float* rvar, const float* about, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  // the slices are summed in double; the sums were taken about about[c] (NULL: about 0), so the subtraction below
  // cancels numbers of the size of the variance instead of the squared mean (torch uses Welford for the same reason)
  double s0 = 0.0, s1 = 0.0;
  for (int r = 0; r < nrep; ++r) { s0 += (double)sums[r * 2 * C + c]; s1 += (double)sums[r * 2 * C + C + c]; }
  const double md = s0 / (double)count;
  float var = (float)(s1 / (double)count - md * md);  // biased
  var = var > 0.f ? var : 0.f;
  const float m = (float)md + (about ? about[c] : 0.f);
  const float rs = rsqrtf(var + eps);
  const float a = gamma[c] * rs;
  if (mean) mean[c] = m;
  if (rstd) rstd[c] = rs;
  if (A) A[c] = a;
  if (shift) shift[c] = beta[c] - m * a;
  if (rmean) rmean[c] = (1.f - momentum) * rmean[c] + momentum * m;
  if (rvar) rvar[c] = (1.f - momentum) * rvar[c] + momentum * var * (count > 1.f ? count / (count - 1.f) : 1.f);
}

__global__ void bn_fold_kernel(const float* __restrict__ rmean, const float* __restrict__ rvar,
                               const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float* mean,
                               float* rstd, float* A, float* shift, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float rs = rsqrtf(rvar[c] + eps), a = gamma[c] * rs;
  if (mean) mean[c] = rmean[c];
  if (rstd) rstd[c] = rs;
  if (A) A[c] = a;
  if (shift) shift[c] = beta[c] - rmean[c] * a;
}

__global__ void bn_bwd_coef_kernel(const float* __restrict__ bstats, int nrep, float count, int batch_stats,
                                   const float* __restrict__ A, float* dgamma, float* dbeta, float* c1, float* c2,
                                   float* c3, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float S0 = 0.f, S1 = 0.f;
  for (int r = 0; r < nrep; ++r) { S0 += bstats[r * 2 * C + c]; S1 += bstats[r * 2 * C + C + c]; }
  const float a = A[c];
  if (dgamma) dgamma[c] += S1;
  if (dbeta) dbeta[c] += S0;
  c1[c] = a;
  c2[c] = batch_stats ? a * S0 / count : 0.f;
  c3[c] = batch_stats ? a * S1 / count : 0.f;
}

// ------------------------------------------------------------------------------------ SE gate (one block per image)
__global__ __launch_bounds__(256) void se_fwd_kernel(const float* __restrict__ gsum, float inv_hw,
                                                     const float* __restrict__ w1, const float* __restrict__ b1,
                                                     const float* __restrict__ w2, const float* __restrict__ b2,
                                                     float* __restrict__ s, float* __restrict__ hidden, int E, int R) {
  extern __shared__ float sm[];  // m[E], h[R]
  float* m = sm;
  float* h = sm + E;
  const int b = blockIdx.x;
  for (int e = threadIdx.x; e < E; e += 256) m[e] = gsum[(int64_t)b * E + e] * inv_hw;
  __syncthreads();
  for (int r = threadIdx.x; r < R; r += 256) {
    float a = b1[r];
    for (int e = 0; e < E; ++e) a += w1[(int64_t)r * E + e] * m[e];
    a = a > 0.f ? a : 0.f;
    h[r] = a;
    hidden[(int64_t)b * R + r] = a;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < E; e += 256) {
    float a = b2[e];
    for (int r = 0; r < R; ++r) a += w2[(int64_t)e * R + r] * h[r];
    s[(int64_t)b * E + e] = lmn_hsigmoid(a);
  }
}

__global__ __launch_bounds__(256) void se_bwd_kernel(const float* __restrict__ ds, const float* __restrict__ gsum,
                                                     float inv_hw, const float* __restrict__ w1,
                                                     const float* __restrict__ w2, const float* __restrict__ b2,
                                                     const float* __restrict__ hidden, float* __restrict__ dm,
                                                     float* dw1, float* db1, float* dw2, float* db2, int E, int R) {
  extern __shared__ float sm[];  // m[E], dt[E], h[R], da[R]
  float* m = sm;
  float* dt = sm + E;
  float* h = sm + 2 * E;
  float* da = sm + 2 * E + R;
  const int b = blockIdx.x;
  for (int r = threadIdx.x; r < R; r += 256) h[r] = hidden[(int64_t)b * R + r];
  for (int e = threadIdx.x; e < E; e += 256) m[e] = gsum[(int64_t)b * E + e] * inv_hw;
  __syncthreads();
  for (int e = threadIdx.x; e < E; e += 256) {
    float a = b2[e];
    for (int r = 0; r < R; ++r) a += w2[(int64_t)e * R + r] * h[r];
    const float d = ds[(int64_t)b * E + e] * lmn_dhsigmoid(a);
    dt[e] = d;
    atomicAdd(db2 + e, d);
  }
  __syncthreads();
  for (int r = threadIdx.x; r < R; r += 256) {
    float a = 0.f;
    for (int e = 0; e < E; ++e) a += w2[(int64_t)e * R + r] * dt[e];
    a = h[r] > 0.f ? a : 0.f;
    da[r] = a;
    atomicAdd(db1 + r, a);
  }
  for (int i = threadIdx.x; i < E * R; i += 256) {
    const int e = i / R, r = i - e * R;
    atomicAdd(dw2 + i, dt[e] * h[r]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < E * R; i += 256) {
    const int r = i / E, e = i - r * E;
    atomicAdd(dw1 + i, da[r] * m[e]);
  }
  for (int e = threadIdx.x; e < E; e += 256) {
    float a = 0.f;
    for (int r = 0; r < R; ++r) a += w1[(int64_t)r * E + e] * da[r];
    dm[(int64_t)b * E + e] = a * inv_hw;
  }
}

// SE backward in two launches: the per-image vectors (dm feeds the depthwise backward: critical path) and the parameter
// gradients (a batch reduction without atomics; the engine puts it on the weight-gradient stream).  The one-launch form above
// adds 2*E*R float atomics per image (590 k per call at E = 384) on the critical path.
// hardsigmoid'(a) is recovered from the saved gate s = clamp(a/6 + 1/2, 0, 1): 1/6 where 0 < s < 1.
__global__ __launch_bounds__(256) void se_bwd_dm_kernel(const float* __restrict__ ds, const float* __restrict__ sgate,
                                                        float inv_hw, const float* __restrict__ w1,
                                                        const float* __restrict__ w2, const float* __restrict__ hidden,
                                                        float* __restrict__ dm, float* __restrict__ dvec, int E, int R) {
  extern __shared__ float sm[];  // dt[E], da[R], part[256]
  float* dt = sm;
  float* da = sm + E;
  float* part = sm + E + R;
  const int b = blockIdx.x, t = threadIdx.x;
  for (int e = t; e < E; e += 256) {
    const float sv = sgate[(int64_t)b * E + e];
    const float d = (sv > 0.f && sv < 1.f) ? ds[(int64_t)b * E + e] * (1.f / 6.f) : 0.f;
    dt[e] = d;
    dvec[(int64_t)b * (E + R) + e] = d;
  }
  __syncthreads();
  // da[r] = relu'(h[r]) * sum_e w2[e][r] * dt[e]: groups of R threads (coalesced rows of w2) split the e range
  const int groups = R <= 256 ? 256 / R : 1;
  for (int r0 = 0; r0 < R; r0 += 256) {  // (one pass unless R > 256)
    const int g = t / R, r = r0 + (R <= 256 ? t - g * R : t);
    float a = 0.f;
    if (g < groups && r < R)
      for (int e = g; e < E; e += groups) a += w2[(int64_t)e * R + r] * dt[e];
    part[t] = a;
    __syncthreads();
    if (t < R - r0 && t < 256) {
      float v = 0.f;
      if (R <= 256) { for (int k = 0; k < groups; ++k) v += part[k * R + t]; }
      else v = part[t];
      const int rr = r0 + t;
      v = hidden[(int64_t)b * R + rr] > 0.f ? v : 0.f;
      da[rr] = v;
      dvec[(int64_t)b * (E + R) + E + rr] = v;
    }
    __syncthreads();
  }
  for (int e = t; e < E; e += 256) {
    float a = 0.f;
    for (int r = 0; r < R; ++r) a += w1[(int64_t)r * E + e] * da[r];
    dm[(int64_t)b * E + e] = a * inv_hw;
  }
}

__global__ __launch_bounds__(256) void se_bwd_params_kernel(const float* __restrict__ dvec, const float* __restrict__ gsum,
                                                            float inv_hw, const float* __restrict__ hidden, float* dw1,
                                                            float* db1, float* dw2, float* db2, int B, int E, int R) {
  lmn_se_bwd_params_item(blockIdx.x * 256 + threadIdx.x, dvec, gsum, inv_hw, hidden, dw1, db1, dw2, db2, B, E, R);
}

// ------------------------------------------------------------------------------------ bilinear x2, align_corners=True
// index arithmetic mirrors ATen's upsample_bilinear2d (fp32): src = dst * (in-1)/(out-1)  (lmn_up_coord, common.h: shared with the conv
// family's LMN_SRC_UP2 staging)
#define up_coord lmn_up_coord

// One block row (blockIdx.y, grid-strided) = one OUTPUT row of one image, so the vertical coordinate arithmetic is block-uniform;
// blockIdx.x covers the (column, channel quad) items of that row, decoded with a magic multiplier.  (The flat-index form spent three
// 64-bit divisions per element on the decode: several hundred instructions around four loads.)
template <typename T>
__global__ __launch_bounds__(256) void up2_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int Hin,
                                                      int Win, int C4, int xcs, int ycs, uint32_t mC4) {
  const int Hout = 2 * Hin, Wout = 2 * Win;
  const float sh = (float)(Hin - 1) / (float)(Hout - 1), sw = (float)(Win - 1) / (float)(Wout - 1);
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= Wout * C4) return;
  const int ox = (int)lmn_div_row((uint32_t)j, (uint32_t)C4, mC4), c = (j - ox * C4) * 4;
  int x0, xp;
  float lx0, lx1;
  up_coord(ox, Win, sw, x0, xp, lx0, lx1);
  for (int row = blockIdx.y; row < B * Hout; row += gridDim.y) {
    const int b = row / Hout, oy = row - b * Hout;
    int y0, yp;
    float ly0, ly1;
    up_coord(oy, Hin, sh, y0, yp, ly0, ly1);
    const T* r0 = x + ((int64_t)(b * Hin + y0) * Win) * xcs + c;
    const T* r1 = r0 + (int64_t)yp * Win * xcs;
    const f32x4 v00 = ld4(r0 + x0 * xcs), v01 = ld4(r0 + (x0 + xp) * xcs);
    const f32x4 v10 = ld4(r1 + x0 * xcs), v11 = ld4(r1 + (x0 + xp) * xcs);
    const f32x4 o = ly0 * (lx0 * v00 + lx1 * v01) + ly1 * (lx0 * v10 + lx1 * v11);
    st4(y + ((int64_t)row * Wout + ox) * ycs + c, o);
  }
}

// Adjoint of the above, gather form: source pixel (iy, ix) collects from the output pixels whose two taps include it.  Output
// index o reads sources i0(o) = (int)(o s) and i0 + ip; with s = (n - 1) / (2n - 1) the outputs that can reach source i lie in
// [2i - 2, 2i + 3], so six candidates per axis with the weight  w(o, i) = [i0(o) == i] l0(o) + [i0(o) + ip(o) == i] l1(o)  (the
// forward's own arithmetic, selects instead of branches).  The vertical candidates and weights are block-uniform (one block row =
// one source row): rows with weight zero are skipped by a scalar branch; the horizontal six are per thread.  (Before: eight
// candidates per axis with per-lane branches around 64 loads and three 64-bit divisions per element: ~1000 instructions per
// element, 62 us per launch inside the step against 27 alone.)
__device__ __forceinline__ float up_adj_weight(int o, int i, int n_in, int n_out, float scale) {
  int i0, ip;
  float l0, l1;
  up_coord(o, n_in, scale, i0, ip, l0, l1);
  const float w = (i0 == i ? l0 : 0.f) + (i0 + ip == i ? l1 : 0.f);
  return (unsigned)o < (unsigned)n_out ? w : 0.f;
}
template <typename T>
__global__ __launch_bounds__(256) void up2_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int B,
                                                      int Hin, int Win, int C4, int dycs, int dxcs, uint32_t mC4, int prio) {
  lmn_setprio_level(prio);   // (lmn_set_priority_stream)
  const int Hout = 2 * Hin, Wout = 2 * Win;
  const float sh = (float)(Hin - 1) / (float)(Hout - 1), sw = (float)(Win - 1) / (float)(Wout - 1);
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= Win * C4) return;
  const int ix = (int)lmn_div_row((uint32_t)j, (uint32_t)C4, mC4), c = (j - ix * C4) * 4;
  float wx[6];
  int oxo[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const int ox = 2 * ix - 2 + k;
    wx[k] = up_adj_weight(ox, ix, Win, Wout, sw);
    oxo[k] = min(max(ox, 0), Wout - 1) * dycs;
  }
  for (int row = blockIdx.y; row < B * Hin; row += gridDim.y) {
    const int b = row / Hin, iy = row - b * Hin;
    f32x4 acc = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int ky = 0; ky < 6; ++ky) {
      const int oy = 2 * iy - 2 + ky;
      const float wy = up_adj_weight(oy, iy, Hin, Hout, sh);   // block-uniform
      if (wy == 0.f) continue;
      const T* rp = dy + ((int64_t)(b * Hout + oy) * Wout) * dycs + c;
#pragma unroll
      for (int kx = 0; kx < 6; ++kx) acc += (wy * wx[kx]) * ld4(rp + oxo[kx]);
    }
    st4(dx + ((int64_t)row * Win + ix) * dxcs + c, acc);
  }
}

// ------------------------------------------------------------------------------------ f x f mean pool
// Wave-per-output form (C <= 256): a wave owns one output pixel -- lane = (window slot, channel quad), a lane sums its slots of the
// f x f window, the slots meet in a wave-private LDS tile (wave-level fence, no block barrier) and lane ch writes channel ch.  The
// block-per-output form below (three block barriers and a divided reduction per output pixel) took 59 us per launch inside the
// step against 12 alone: every one of its 3872 blocks is a chain of barriers that waits for CU slots next to the other streams.
template <typename T>
__global__ __launch_bounds__(256) void avgpool_fwd_wave_kernel(const T* __restrict__ x, T* __restrict__ y, int Hout, int Wout, int f,
                                                               int C, int xcs, int ycs, int total, uint32_t mW, uint32_t mHW, uint32_t mC4, int prio) {
  __shared__ __attribute__((aligned(16))) float park[4][64 * 4];
  lmn_setprio_level(prio);   // (lmn_set_priority_stream)
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int op = blockIdx.x * 4 + wv;
  if (op >= total) return;                      // (wave-uniform; no block barrier below)
  const int C4 = C >> 2, nsl = 64 / C4;         // window slots per pass
  const uint32_t b = lmn_div_row((uint32_t)op, (uint32_t)(Hout * Wout), mHW), r = (uint32_t)op - b * (uint32_t)(Hout * Wout);
  const int oy = (int)lmn_div_row(r, (uint32_t)Wout, mW), ox = (int)r - oy * Wout;
  const int slot = (int)lmn_div_row((uint32_t)lane, (uint32_t)C4, mC4), q = lane - slot * C4;
  const int Win = Wout * f;
  const T* base = x + (((int64_t)((int)b * Hout + oy) * f) * Win + (int64_t)ox * f) * xcs + q * 4;
  const bool pow2 = (f & (f - 1)) == 0;
  const int lf = 31 - __clz(f);
  f32x4 s = f32x4{0, 0, 0, 0};
  if (slot < nsl) {
    for (int w = slot; w < f * f; w += nsl) {
      const int wy = pow2 ? (w >> lf) : w / f, wx = w - wy * f;
      s += ld4(base + ((int64_t)wy * Win + wx) * xcs);
    }
  }
  *reinterpret_cast<f32x4*>(&park[wv][lane * 4]) = s;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const float inv = 1.0f / (float)(f * f);
  for (int ch = lane; ch < C; ch += 64) {
    const int qq = ch >> 2, k = ch & 3;
    float a = 0.f;
    for (int sl = 0; sl < nsl; ++sl) a += park[wv][(sl * C4 + qq) * 4 + k];    // fixed order: deterministic
    st1(y + (int64_t)op * ycs + ch, a * inv);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int Hout,
                                                          int Wout, int f, int C, int xcs, int ycs) {
  extern __shared__ float red[];  // [C] | parked partials [4][256] | group sums (chan_block_reduce)
  const int C4 = C >> 2;
  const int NTH = (256 / C4) * C4, per = NTH / C4;
  const int tid = threadIdx.x;
  const int ox = blockIdx.x % Wout, oy = (blockIdx.x / Wout) % Hout, b = blockIdx.x / (Wout * Hout);
  f32x4 sv[1] = {f32x4{0, 0, 0, 0}};
  if (tid < NTH) {
    const int c = (tid % C4) * 4;
    f32x4 s = f32x4{0, 0, 0, 0};
    const int Hin = Hout * f, Win = Wout * f;
    for (int w = tid / C4; w < f * f; w += per) {
      const int wy = w / f, wx = w - wy * f;
      s += ld4(x + (((int64_t)b * Hin + oy * f + wy) * Win + ox * f + wx) * xcs + c);
    }
    sv[0] = s;
  }
  chan_block_reduce<1>(sv, red + C, red, C, C4, NTH, per, tid);   // (fixed order, no LDS atomics)
  const float inv = 1.0f / (float)(f * f);
  for (int i = tid; i < C; i += 256) st1(y + (((int64_t)b * Hout + oy) * Wout + ox) * ycs + i, red[i] * inv);
}

template <typename T>
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int B,
                                                          int Hout, int Wout, int f, int C4, int dycs, int dxcs,
                                                          int accumulate, uint32_t mC4, uint32_t mf, int prio) {
  lmn_setprio_level(prio);   // (lmn_set_priority_stream)
  // one block row (grid-strided) = one input row of one image: the vertical index arithmetic is block-uniform; columns by magic
  // multipliers (the flat-index form: five 64- / 32-bit divisions per element around one load and one store)
  const int Hin = Hout * f, Win = Wout * f;
  const float inv = 1.0f / (float)(f * f);
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= Win * C4) return;
  const int ix = (int)lmn_div_row((uint32_t)j, (uint32_t)C4, mC4), c = (j - ix * C4) * 4;
  const int oxc = (int)lmn_div_row((uint32_t)ix, (uint32_t)f, mf);
  for (int row = blockIdx.y; row < B * Hin; row += gridDim.y) {
    const int b = row / Hin, iy = row - b * Hin;
    f32x4 g = inv * ld4(dy + (((int64_t)b * Hout + iy / f) * Wout + oxc) * dycs + c);
    T* o = dx + ((int64_t)row * Win + ix) * dxcs + c;
    if (accumulate) g += ld4(o);
    st4(o, g);
  }
}

// ------------------------------------------------------------------------------------ layout + utilities
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ x, T* __restrict__ y, int B,
                                                           int C, int64_t HW, int ycs) {
  const int64_t total = (int64_t)B * HW;
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < total; p += (int64_t)gridDim.x * 256) {
    const int64_t b = p / HW, hw = p - b * HW;
    for (int c = 0; c < ycs; ++c) st1(y + p * ycs + c, c < C ? x[(b * C + c) * HW + hw] : 0.f);
  }
}
template <typename T>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const T* __restrict__ x, float* __restrict__ y, int B,
                                                           int C, int64_t HW, int xcs) {
  const int64_t total = (int64_t)B * HW;
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < total; p += (int64_t)gridDim.x * 256) {
    const int64_t b = p / HW, hw = p - b * HW;
    for (int c = 0; c < C; ++c) y[(b * C + c) * HW + hw] = ld1(x + p * xcs + c);
  }
}
__global__ __launch_bounds__(256) void fill_kernel(float* __restrict__ p, float v, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = v;
}
template <typename T>
__global__ __launch_bounds__(256) void add_kernel(const T* __restrict__ a, const T* __restrict__ b,
                                                  const T* __restrict__ c, const T* __restrict__ d,
                                                  T* __restrict__ y, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    f32x4 v = ld4(a + i * 4) + ld4(b + i * 4);
    if (c) v += ld4(c + i * 4);
    if (d) v += ld4(d + i * 4);
    st4(y + i * 4, v);
  }
}
// y[p][c] = a[c] * u[p][c] + b[c] * v[p][c] + k[c]  (coef = [3][C]: a, b, k) -- the BatchNorm backward of the expand conv on the
// z-path (dz = a * dh + b * z + c, lmn_reparam_fold), run beside the critical path for the weight gradient
template <typename T>
__global__ __launch_bounds__(256) void affine2_kernel(const T* __restrict__ u, const T* __restrict__ v, const float* __restrict__ coef,
                                                      T* __restrict__ y, int64_t rows, int C4, int rp_w) {
  const int64_t total = rows * C4;
  const int C = C4 * 4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)((rp_w ? i / rp_w : i) % C4) * 4;   // (row-planar: quad i is (image row, channel quad, x))
    const f32x4 a = ld4(coef + c), b = ld4(coef + C + c), k = ld4(coef + 2 * C + c);
    st4(y + i * 4, a * ld4(u + i * 4) + b * ld4(v + i * 4) + k);
  }
}
template <typename T>
__global__ __launch_bounds__(256) void copy_slice_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t rows,
                                                         int C4, int xcs, int ycs) {
  const int64_t total = rows * C4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / C4;
    const int c = (int)(i - r * C4) * 4;
    st4(y + r * ycs + c, ld4(x + r * xcs + c));
  }
}

inline int grid_for(int64_t work_items, int per_block = 256, int cap = 4096) {
  int64_t g = (work_items + per_block - 1) / per_block;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

inline int ln_group(int C4) {
  int g = 1;
  while (g < C4 && g < 64) g <<= 1;
  return g;
}

// AdamW over the flat parameter / gradient buffers (torch.optim.AdamW, amsgrad=False, maximize=False):
//   p *= 1 - lr*wd;  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                             float* __restrict__ v, int64_t n4, float lr, float b1, float b2, float eps, float wd,
                             float inv_bc1, float inv_sqrt_bc2) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    f32x4 pp = reinterpret_cast<f32x4*>(p)[i], mm = reinterpret_cast<f32x4*>(m)[i], vv = reinterpret_cast<f32x4*>(v)[i];
    const f32x4 gg = reinterpret_cast<const f32x4*>(g)[i];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float x = pp[k] * (1.f - lr * wd);
      mm[k] = b1 * mm[k] + (1.f - b1) * gg[k];
      vv[k] = b2 * vv[k] + (1.f - b2) * gg[k] * gg[k];
      const float denom = sqrtf(vv[k]) * inv_sqrt_bc2 + eps;
      pp[k] = x - (lr * inv_bc1) * (mm[k] / denom);
    }
    reinterpret_cast<f32x4*>(p)[i] = pp;
    reinterpret_cast<f32x4*>(m)[i] = mm;
    reinterpret_cast<f32x4*>(v)[i] = vv;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Segmentation loss around the path: CrossEntropyLoss(weight, label_smoothing) (train.py:157) +
// DiceLoss (utils/loss.py:170-206) on NCHW logits [B,C,HW], C in {2,3,4,8}.  Pass 1 reduces the 3 + 3C batch
// sums, a one-block kernel turns them into the scalar loss and the backward coefficients, pass 2 writes dlogits.
//   sums: [0] sum w_y   [1] sum w_y*(-log p_y)   [2] sum_i sum_c w_c*(-log p_c)
//         [3+c] sum p_c*t_c   [3+C+c] sum p_c^2   [3+2C+c] sum t_c


template <int C>
__device__ __forceinline__ void loss_softmax(const float* __restrict__ lg, int64_t hw, int64_t i, float (&p)[C], float& lse) {
  float z[C], mx = -3.0e38f;
#pragma unroll
  for (int c = 0; c < C; ++c) { z[c] = lg[c * hw + i]; mx = fmaxf(mx, z[c]); }
  float den = 0.f;
#pragma unroll
  for (int c = 0; c < C; ++c) { p[c] = __expf(z[c] - mx); den += p[c]; }
  const float r = 1.f / den;
  lse = mx + __logf(den);
#pragma unroll
  for (int c = 0; c < C; ++c) { p[c] *= r; z[c] = z[c]; }
}

template <int C>
__global__ __launch_bounds__(256) void segloss_sums_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target,
                                                           const float* __restrict__ wce, int B, int64_t hw,
                                                           float* __restrict__ sums, int det) {
  constexpr int NS = 3 + 3 * C;
  float acc[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) acc[k] = 0.f;
  float w[C];
#pragma unroll
  for (int c = 0; c < C; ++c) w[c] = wce[c];
  const int64_t total = (int64_t)B * hw;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = idx / hw, i = idx - b * hw;
    const float* lg = logits + b * C * hw;
    float p[C], lse;
    loss_softmax<C>(lg, hw, i, p, lse);
    const int y = (int)target[idx];
    float sm = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const float nlp = lse - lg[c * hw + i];  // -log p_c
      const float t = (c == y) ? 1.f : 0.f;
      sm += w[c] * nlp;
      acc[0] += t * w[c];
      acc[1] += t * w[c] * nlp;
      acc[3 + c] += p[c] * t;
      acc[3 + C + c] += p[c] * p[c];
      acc[3 + 2 * C + c] += t;
    }
    acc[2] += sm;
  }
  __shared__ float red[4][NS];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    float v = acc[k];
#pragma unroll
    for (int m = 1; m <= 32; m <<= 1) v += __shfl_xor(v, m, 64);
    if (lane == 0) red[wv][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < NS)   // (deterministic mode: sums addresses slot copies [blocks][NS])
    lmn_red_add(sums + (det ? (int64_t)blockIdx.x * NS : 0) + threadIdx.x, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x], det);
}

// coef: [0] (1-eps)/S_w  [1] (eps/C)/S_w  [2] sum_c w_c   [3+c] a_c  [3+C+c] b_c   with dL_dice/dp_c = a_c*t_c + b_c*p_c
__global__ void segloss_finish_kernel(const float* __restrict__ sums, const float* __restrict__ wce,
                                      const float* __restrict__ wdice, int C, float eps_ls, float smooth,
                                      float* __restrict__ loss, float* __restrict__ coef) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const float Sw = sums[0];
  float l = ((1.f - eps_ls) * sums[1] + (eps_ls / C) * sums[2]) / Sw;
  float wsum = 0.f;
  for (int c = 0; c < C; ++c) {
    const float I = sums[3 + c], Z = sums[3 + C + c], Y = sums[3 + 2 * C + c];
    const float num = 2.f * I + smooth, den = Z + Y + smooth;
    l += wdice[c] * (1.f - num / den) / C;
    coef[3 + c] = wdice[c] / C * (-2.f / den);
    coef[3 + C + c] = wdice[c] / C * (2.f * num / (den * den));
    wsum += wce[c];
  }
  coef[0] = (1.f - eps_ls) / Sw;
  coef[1] = (eps_ls / C) / Sw;
  coef[2] = wsum;
  loss[0] = l;
}

template <int C>
__global__ __launch_bounds__(256) void segloss_bwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target,
                                                          const float* __restrict__ wce, const float* __restrict__ coef,
                                                          const float* __restrict__ gscale, int B, int64_t hw,
                                                          float* __restrict__ dlogits) {
  float w[C], a[C], bq[C];
#pragma unroll
  for (int c = 0; c < C; ++c) { w[c] = wce[c]; a[c] = coef[3 + c]; bq[c] = coef[3 + C + c]; }
  const float k_nll = coef[0], k_sm = coef[1], wsum = coef[2];
  const float gs = gscale ? gscale[0] : 1.f;
  const int64_t total = (int64_t)B * hw;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = idx / hw, i = idx - b * hw;
    const float* lg = logits + b * C * hw;
    float p[C], lse;
    loss_softmax<C>(lg, hw, i, p, lse);
    const int y = (int)target[idx];
    float wy = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) wy = (c == y) ? w[c] : wy;
    float g[C], gp = 0.f;  // dice: dL/dp_c, then through the softmax Jacobian
#pragma unroll
    for (int c = 0; c < C; ++c) {
      g[c] = a[c] * ((c == y) ? 1.f : 0.f) + bq[c] * p[c];
      gp += g[c] * p[c];
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const float t = (c == y) ? 1.f : 0.f;
      const float dce = k_nll * wy * (p[c] - t) + k_sm * (p[c] * wsum - w[c]);
      dlogits[b * C * hw + c * hw + i] = gs * (dce + p[c] * (g[c] - gp));
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Input pipeline on the device: the validation transform of dataset/data_loading.py:203-206
// (A.Resize -> A.Normalize -> ToTensorV2) plus the two flips of the training transform (:213-214), for a batch of
// uint8 HWC images and grayscale masks already in HBM.
//   image: cv2.resize(INTER_LINEAR) on uint8 -- half-pixel centres, 11-bit fixed-point coefficients, result rounded
//          back to uint8 -- then (v - mean*255) * (1/(std*255)) with numpy's float32/float64 promotion;
//   mask:  cv2.threshold(127, 1, THRESH_BINARY) (data_loading.py:237) and INTER_NEAREST (floor(dst*scale));
//   flips: bit 0 of flips[b] = horizontal, bit 1 = vertical (applied after the resize, before Normalize).
struct PrepGeom {
  int B, Hs, Ws, H, W;
  double m255[3], inv[3];
};

__device__ __forceinline__ void prep_axis(int d, int n_dst, int n_src, bool clamp_frac, int& s0, int& s1, int& a0, int& a1) {
  const double scale = (double)n_src / (double)n_dst;
  float f = (float)((d + 0.5) * scale - 0.5);
  int s = (int)floorf(f);
  f -= (float)s;
  if (clamp_frac) {  // columns: cv2 zeroes the fraction at the borders; rows are clamped instead
    if (s < 0) { s = 0; f = 0.f; }
    if (s >= n_src - 1) { s = n_src - 1; f = 0.f; }
  }
  a0 = (int)rintf((1.f - f) * 2048.f);  // saturate_cast<short>: round half to even
  a1 = (int)rintf(f * 2048.f);
  s0 = min(max(s, 0), n_src - 1);
  s1 = min(max(s + 1, 0), n_src - 1);
}

__global__ __launch_bounds__(256) void preprocess_u8_kernel(const uint8_t* __restrict__ img, const uint8_t* __restrict__ mask,
                                                            const uint8_t* __restrict__ flips, float* __restrict__ out,
                                                            int64_t* __restrict__ labels, const PrepGeom g) {
  const int64_t n = (int64_t)g.B * g.H * g.W;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int x = (int)(i % g.W), y = (int)((i / g.W) % g.H), b = (int)(i / ((int64_t)g.W * g.H));
    const int fl = flips ? flips[b] : 0;
    const int rx = (fl & 1) ? g.W - 1 - x : x, ry = (fl & 2) ? g.H - 1 - y : y;  // pixel of the resized image
    if (img) {
      int x0, x1, ax0, ax1, y0, y1, ay0, ay1;
      prep_axis(rx, g.W, g.Ws, true, x0, x1, ax0, ax1);
      prep_axis(ry, g.H, g.Hs, false, y0, y1, ay0, ay1);
      const uint8_t* r0 = img + ((int64_t)b * g.Hs + y0) * g.Ws * 3;
      const uint8_t* r1 = img + ((int64_t)b * g.Hs + y1) * g.Ws * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int h0 = r0[x0 * 3 + c] * ax0 + r0[x1 * 3 + c] * ax1;  // horizontal pass, scale 2^11
        const int h1 = r1[x0 * 3 + c] * ax0 + r1[x1 * 3 + c] * ax1;
        int v = (((ay0 * (h0 >> 4)) >> 16) + ((ay1 * (h1 >> 4)) >> 16) + 2) >> 2;  // cv2 VResizeLinear<uchar>
        v = min(max(v, 0), 255);
        const float t = (float)((double)v - g.m255[c]);     // numpy: float32 array -= float64 mean
        out[(((int64_t)b * 3 + c) * g.H + y) * g.W + x] = (float)((double)t * g.inv[c]);
      }
    }
    if (mask) {
      const int sx = min((int)floor(rx * ((double)g.Ws / g.W)), g.Ws - 1);
      const int sy = min((int)floor(ry * ((double)g.Hs / g.H)), g.Hs - 1);
      labels[i] = mask[((int64_t)b * g.Hs + sy) * g.Ws + sx] > 127 ? 1 : 0;
    }
  }
}


// Confusion matrix of argmax(logits) against the labels: counts[t*C + p] += 1 (float counts are
// exact up to 2^24 per launch per cell; the host accumulates in int64/double).
template <int C>
__global__ __launch_bounds__(256) void confusion_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target,
                                                        int B, int64_t hw, float* __restrict__ counts) {
  __shared__ float sc[C * C];
  for (int i = threadIdx.x; i < C * C; i += 256) sc[i] = 0.f;
  __syncthreads();
  int cnt[C * C];
#pragma unroll
  for (int k = 0; k < C * C; ++k) cnt[k] = 0;
  const int64_t total = (int64_t)B * hw;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = idx / hw, i = idx - b * hw;
    const float* lg = logits + b * C * hw;
    int best = 0;
    float bv = lg[i];
#pragma unroll
    for (int c = 1; c < C; ++c) {
      const float v = lg[c * hw + i];
      if (v > bv) { bv = v; best = c; }  // first maximum wins, as torch.argmax
    }
    const int y = (int)target[idx];
#pragma unroll
    for (int k = 0; k < C * C; ++k) cnt[k] += (k == y * C + best) ? 1 : 0;
  }
#pragma unroll
  for (int k = 0; k < C * C; ++k) {
    int v = cnt[k];
#pragma unroll
    for (int m = 1; m <= 32; m <<= 1) v += __shfl_xor(v, m, 64);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(&sc[k], (float)v);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < C * C; i += 256)
    if (sc[i] != 0.f) atomicAdd(counts + i, sc[i]);
}

__global__ void copy2d_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t rows, int cols, int xs, int ys) {
  const int64_t n = rows * cols;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cols;
    const int c = (int)(i - r * cols);
    y[r * ys + c] = x[r * xs + c];
  }
}

}  // namespace

// dynamic LDS of chan_kernel: red[2C] | parked partials [8][256] | group sums (<= max(256, 2C))
static size_t chan_shmem(int C) { return (size_t)(2 * C + 8 * 256 + (2 * C > 256 ? 2 * C : 256)) * sizeof(float); }

extern "C" {

int lmn_copy2d(const float* x, float* y, int64_t rows, int cols, int x_stride, int y_stride, lmn_stream_t stream) {
  LMN_REC(lmn_copy2d(x, y, rows, cols, x_stride, y_stride, stream));
  LMN_REQUIRE(x && y && rows > 0 && cols > 0 && x_stride >= cols && y_stride >= cols, "copy2d: bad argument");
  LMN_LAUNCH(copy2d_kernel, dim3(grid_for(rows * cols)), dim3(256), 0, (hipStream_t)stream, x, y, rows, cols, x_stride, y_stride);
  return lmn_launch_status("copy2d");
}

int lmn_ln_fwd(const void* x, const float* gamma, const float* beta, void* y, int64_t rows, int C, int act_dtype,
               lmn_stream_t stream) {
  LMN_REC(lmn_ln_fwd(x, gamma, beta, y, rows, C, act_dtype, stream));
  LMN_REQUIRE_DT(act_dtype, "ln_fwd");
  LMN_REQUIRE(x && gamma && beta && y && rows > 0, "ln_fwd: bad argument");
  LMN_REQUIRE(C % 4 == 0 && C >= 4 && C <= 512, "ln_fwd: C=%d (need multiple of 4, <= 512)", C);
  const int G = ln_group(C / 4);
  const int grid = grid_for(rows, 256 / G, 2048);
  hipStream_t st = (hipStream_t)stream;
#define LN_CASE(g) case g: LMN_LAUNCH((ln_fwd_kernel<g, T>), dim3(grid), dim3(256), 0, st, (const T*)x, gamma, beta, (T*)y, rows, C); break;
  LMN_ACT_DISPATCH(act_dtype, switch (G) { LN_CASE(1) LN_CASE(2) LN_CASE(4) LN_CASE(8) LN_CASE(16) LN_CASE(32) LN_CASE(64) });
#undef LN_CASE
  return lmn_launch_status("ln_fwd");
}

int lmn_ln_bwd(const void* x, const float* gamma, const void* dy, const void* dres, void* dx, float* dgamma,
               float* dbeta, int64_t rows, int C, int act_dtype, lmn_stream_t stream) {
  LMN_REC(lmn_ln_bwd(x, gamma, dy, dres, dx, dgamma, dbeta, rows, C, act_dtype, stream));
  LMN_REQUIRE_DT(act_dtype, "ln_bwd");
  LMN_REQUIRE(x && gamma && dy && dx && dgamma && dbeta && rows > 0, "ln_bwd: bad argument");
  LMN_REQUIRE(C % 4 == 0 && C >= 4 && C <= 512, "ln_bwd: C=%d", C);
  const int G = ln_group(C / 4);
  // every block ends with one global atomic per channel: ~40 ns per block on the same address, so the grid is kept to
  // what the tensor needs for bandwidth (level 0: 143 MB -> 1024 blocks; the coarser maps: 512)
  const int grid = grid_for(rows, 256 / G, C <= 12 ? 1024 : 512);
  hipStream_t st = (hipStream_t)stream;
  const size_t sh = 8 * C * sizeof(float);
  float *dgs = dgamma, *dbs = dbeta;
  if (g_lmn_det) {   // per-block sums into slot copies of dgamma / dbeta, folded in fixed order below
    lmn_det_begin(st);
    dgs = lmn_det_slots(st, (size_t)2 * grid * C);
    LMN_REQUIRE(dgs, "ln_bwd: deterministic mode: no scratch");
    dbs = dgs + (size_t)grid * C;
  }
#define LN_CASE(g) case g: LMN_LAUNCH((ln_bwd_kernel<g, T>), dim3(grid), dim3(256), sh, st, (const T*)x, gamma, (const T*)dy, (const T*)dres, (T*)dx, dgs, dbs, rows, C, g_lmn_det); break;
  LMN_ACT_DISPATCH(act_dtype, switch (G) { LN_CASE(1) LN_CASE(2) LN_CASE(4) LN_CASE(8) LN_CASE(16) LN_CASE(32) LN_CASE(64) });
#undef LN_CASE
  if (g_lmn_det) { lmn_det_sum(st, dgs, grid, C, dgamma); lmn_det_sum(st, dbs, grid, C, dbeta); }
  return lmn_launch_status("ln_bwd");
}

int lmn_bnact_fwd(const void* z, const float* a, const float* b, void* y, int64_t rows, int C, int act, int act_dtype,
                  lmn_stream_t stream) {
  LMN_REC(lmn_bnact_fwd(z, a, b, y, rows, C, act, act_dtype, stream));
  LMN_REQUIRE_DT(act_dtype, "bnact_fwd");
  LMN_REQUIRE(z && a && b && y && rows > 0 && C > 0 && C % 4 == 0, "bnact_fwd: bad argument");
  const int64_t n4 = rows * (C / 4);
  LMN_ACT_DISPATCH(act_dtype, LMN_LAUNCH((bnact_fwd_kernel<T>), dim3(grid_for(n4)), dim3(256), 0, (hipStream_t)stream, (const T*)z, a, b, (T*)y, n4, C / 4, act));
  return lmn_launch_status("bnact_fwd");
}

int lmn_bnact_fwd_fin(const void* z, const lmn_bn_fin_t* fin, void* y, int64_t rows, int C, int act, int act_dtype, lmn_stream_t stream) {
  LMN_REQUIRE(fin && fin->mode == LMN_FIN_BN, "bnact_fwd_fin: fin.mode must be LMN_FIN_BN");
  const lmn_bn_fin_t F = *fin;
  if (g_lmn_rec) lmn_rec_push([=]() -> int { return lmn_bnact_fwd_fin(z, &F, y, rows, C, act, act_dtype, stream); }, "lmn_bnact_fwd_fin(");
  LMN_REQUIRE_DT(act_dtype, "bnact_fwd_fin");
  LMN_REQUIRE(z && y && rows > 0 && C > 0 && C % 4 == 0 && C <= 4096, "bnact_fwd_fin: bad argument");
  LMN_REQUIRE(F.sums && F.gamma && F.beta && F.nrep >= 1 && F.count > 0.f, "bnact_fwd_fin: fin needs sums / gamma / beta / nrep / count");
  LMN_REQUIRE(!F.about || !F.rmean || F.about != F.rmean, "bnact_fwd_fin: fin.about must not alias fin.rmean (pass the statistics pass's snapshot)");
  const int64_t n4 = rows * (C / 4);
  LMN_ACT_DISPATCH(act_dtype, LMN_LAUNCH((bnact_fwd_fin_kernel<T>), dim3(grid_for(n4)), dim3(256), (size_t)2 * C * sizeof(float), (hipStream_t)stream, (const T*)z, F, (T*)y, n4, C / 4, act));
  return lmn_launch_status("bnact_fwd_fin");
}

int lmn_bnact_bwd_stats(const void* z, const void* dy, const float* mean, const float* rstd, const float* gamma,
                        const float* beta, float* stats, int64_t rows, int C, int act, int act_dtype, lmn_stream_t stream) {
  LMN_REC(lmn_bnact_bwd_stats(z, dy, mean, rstd, gamma, beta, stats, rows, C, act, act_dtype, stream));
  LMN_REQUIRE_DT(act_dtype, "bnact_bwd_stats");
  LMN_REQUIRE(z && dy && mean && rstd && gamma && beta && stats && rows > 0, "bnact_bwd_stats: bad argument");
  LMN_REQUIRE(C % 4 == 0 && C >= 4 && C <= 1024, "bnact_bwd_stats: C=%d", C);
  const int rpb = 256 / (C / 4);
  const int grid = grid_for(rows, rpb * 8, 1024);
  hipStream_t st = (hipStream_t)stream;
  float* sd = stats;
  if (g_lmn_det) {
    lmn_det_begin(st);
    sd = lmn_det_slots(st, (size_t)grid * 2 * C);
    LMN_REQUIRE(sd, "bnact_bwd_stats: deterministic mode: no scratch");
  }
  LMN_ACT_DISPATCH(act_dtype, LMN_LAUNCH((chan_kernel<0, T>), dim3(grid), dim3(256), chan_shmem(C),
                     st, (const T*)z, (const T*)dy, mean, rstd, gamma, beta, nullptr, nullptr, nullptr, (void*)sd, rows, C, C, act, g_lmn_det, lmn_bn_fin_t{}));
  if (g_lmn_det) lmn_det_sum(st, sd, grid, (int64_t)2 * C, stats);
  return lmn_launch_status("bnact_bwd_stats");
}

int lmn_bnact_bwd(const void* z, const void* dy, const float* mean, const float* rstd, const float* gamma,
                  const float* beta, const float* c1, const float* c2, const float* c3, void* dz, int64_t rows, int C,
                  int act, int act_dtype, lmn_stream_t stream) {
  LMN_REC(lmn_bnact_bwd(z, dy, mean, rstd, gamma, beta, c1, c2, c3, dz, rows, C, act, act_dtype, stream));
  LMN_REQUIRE_DT(act_dtype, "bnact_bwd");
  LMN_REQUIRE(z && dy && mean && rstd && gamma && beta && c1 && c2 && c3 && dz && rows > 0, "bnact_bwd: bad argument");
  LMN_REQUIRE(C % 4 == 0 && C >= 4 && C <= 1024, "bnact_bwd: C=%d", C);
  const int rpb = 256 / (C / 4);
  LMN_ACT_DISPATCH(act_dtype, LMN_LAUNCH((chan_kernel<1, T>), dim3(grid_for(rows, rpb * 4, 4096)), dim3(256), chan_shmem(C),
                     (hipStream_t)stream, (const T*)z, (const T*)dy, mean, rstd, gamma, beta, c1, c2, c3, dz, rows, C, C, act, 0, lmn_bn_fin_t{}));
  return lmn_launch_status("bnact_bwd");
}

int lmn_bnact_bwd_fin(const void* z, const void* dy, const float* mean, const float* rstd, const float* gamma,
                      const float* beta, const lmn_bn_fin_t* fin, void* dz, int64_t rows, int C, int act, int act_dtype,
                      lmn_stream_t stream) {
  LMN_REQUIRE(fin && fin->mode == LMN_FIN_BN_BWD, "bnact_bwd_fin: fin.mode must be LMN_FIN_BN_BWD");
  const lmn_bn_fin_t F = *fin;
  if (g_lmn_rec) lmn_rec_push([=]() -> int { return lmn_bnact_bwd_fin(z, dy, mean, rstd, gamma, beta, &F, dz, rows, C, act, act_dtype, stream); }, "lmn_bnact_bwd_fin(");
  LMN_REQUIRE_DT(act_dtype, "bnact_bwd_fin");
  LMN_REQUIRE(z && dy && mean && rstd && gamma && beta && dz && rows > 0 && F.sums && F.Ain && F.nrep >= 1 && F.count > 0.f, "bnact_bwd_fin: bad argument");
  LMN_REQUIRE(C % 4 == 0 && C >= 4 && C <= 1024, "bnact_bwd_fin: C=%d", C);
  const int rpb = 256 / (C / 4);
  LMN_ACT_DISPATCH(act_dtype, LMN_LAUNCH((chan_kernel<1, T>), dim3(grid_for(rows, rpb * 4, 4096)), dim3(256), chan_shmem(C),
                     (hipStream_t)stream, (const T*)z, (const T*)dy, mean, rstd, gamma, beta, nullptr, nullptr, nullptr, dz, rows, C, C, act, 0, F));
  return lmn_launch_status("bnact_bwd_fin");
}

int lmn_colsum(const void* x, float* out, int64_t rows, int C, int cstride, int act_dtype, lmn_stream_t stream) {
  LMN_REC(lmn_colsum(x, out, rows, C, cstride, act_dtype, stream));
  LMN_REQUIRE_DT(act_dtype, "colsum");
  LMN_REQUIRE(x && out && rows > 0 && C % 4 == 0 && C >= 4 && C <= 1024 && cstride >= C && cstride % 4 == 0, "colsum: bad argument");
  const int rpb = 256 / (C / 4);
  const int grid = grid_for(rows, rpb * 8, 1024);
  hipStream_t st = (hipStream_t)stream;
  float* sd = out;
  if (g_lmn_det) {
    lmn_det_begin(st);
    sd = lmn_det_slots(st, (size_t)grid * C);
    LMN_REQUIRE(sd, "colsum: deterministic mode: no scratch");
  }
  LMN_ACT_DISPATCH(act_dtype, LMN_LAUNCH((chan_kernel<2, T>), dim3(grid), dim3(256), chan_shmem(C),
                     st, (const T*)x, (const T*)nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, (void*)sd,
                     rows, C, cstride, 0, g_lmn_det, lmn_bn_fin_t{}));
  if (g_lmn_det) lmn_det_sum(st, sd, grid, C, out);
  return lmn_launch_status("colsum");
}

int lmn_bn_finalize(const float* sums, int nrep, float count, const float* gamma, const float* beta, float eps,
                    float momentum, float* mean, float* rstd, float* A, float* shift, float* running_mean,
                    float* running_var, const float* about, int C, lmn_stream_t stream) {
  LMN_REC(lmn_bn_finalize(sums, nrep, count, gamma, beta, eps, momentum, mean, rstd, A, shift, running_mean, running_var, about, C, stream));
  LMN_REQUIRE(sums && nrep >= 1 && gamma && beta && C > 0 && count > 0.f, "bn_finalize: bad argument");
  LMN_LAUNCH(bn_finalize_kernel, dim3(lmn_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, sums, nrep, count, gamma,
                     beta, eps, momentum, mean, rstd, A, shift, running_mean, running_var, about, C);
  return lmn_launch_status("bn_finalize");
}

int lmn_bn_fold(const float* running_mean, const float* running_var, const float* gamma, const float* beta, float eps,
                float* mean, float* rstd, float* A, float* shift, int C, lmn_stream_t stream) {
  LMN_REC(lmn_bn_fold(running_mean, running_var, gamma, beta, eps, mean, rstd, A, shift, C, stream));
  LMN_REQUIRE(running_mean && running_var && gamma && beta && C > 0, "bn_fold: bad argument");
  LMN_LAUNCH(bn_fold_kernel, dim3(lmn_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, running_mean,
                     running_var, gamma, beta, eps, mean, rstd, A, shift, C);
  return lmn_launch_status("bn_fold");
}

int lmn_bn_bwd_coef(const float* bstats, int nrep, float count, int batch_stats, const float* A, float* dgamma,
                    float* dbeta, float* c1, float* c2, float* c3, int C, lmn_stream_t stream) {
  LMN_REC(lmn_bn_bwd_coef(bstats, nrep, count, batch_stats, A, dgamma, dbeta, c1, c2, c3, C, stream));
  LMN_REQUIRE(bstats && nrep >= 1 && A && c1 && c2 && c3 && C > 0 && count > 0.f, "bn_bwd_coef: bad argument");
  LMN_LAUNCH(bn_bwd_coef_kernel, dim3(lmn_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, bstats, nrep, count,
                     batch_stats, A, dgamma, dbeta, c1, c2, c3, C);
  return lmn_launch_status("bn_bwd_coef");
}

int lmn_se_fwd(const float* gsum, float inv_hw, const float* w1, const float* b1, const float* w2, const float* b2,
               float* s, float* hidden, int B, int E, int R, lmn_stream_t stream) {
  LMN_REC(lmn_se_fwd(gsum, inv_hw, w1, b1, w2, b2, s, hidden, B, E, R, stream));
  LMN_REQUIRE(gsum && w1 && b1 && w2 && b2 && s && hidden && B > 0 && E > 0 && R > 0, "se_fwd: bad argument");
  LMN_REQUIRE((E + R) * sizeof(float) <= 60000, "se_fwd: E=%d too large", E);
  LMN_LAUNCH(se_fwd_kernel, dim3(B), dim3(256), (E + R) * sizeof(float), (hipStream_t)stream, gsum, inv_hw, w1,
                     b1, w2, b2, s, hidden, E, R);
  return lmn_launch_status("se_fwd");
}

int lmn_se_bwd(const float* ds, const float* gsum, float inv_hw, const float* w1, const float* b1, const float* w2,
               const float* b2, const float* hidden, float* dm, float* dw1, float* db1, float* dw2, float* db2, int B,
               int E, int R, lmn_stream_t stream) {
  LMN_REC(lmn_se_bwd(ds, gsum, inv_hw, w1, b1, w2, b2, hidden, dm, dw1, db1, dw2, db2, B, E, R, stream));
  LMN_REQUIRE(!g_lmn_det, "se_bwd: the one-launch form sums with float atomics; deterministic mode takes lmn_se_bwd_dm + lmn_se_bwd_params");
  (void)b1;
  LMN_REQUIRE(ds && gsum && w1 && w2 && b2 && hidden && dm && dw1 && db1 && dw2 && db2 && B > 0 && E > 0 && R > 0, "se_bwd: bad argument");
  LMN_REQUIRE((2 * E + 2 * R) * sizeof(float) <= 60000, "se_bwd: E=%d too large", E);
  LMN_LAUNCH(se_bwd_kernel, dim3(B), dim3(256), (2 * E + 2 * R) * sizeof(float), (hipStream_t)stream, ds, gsum,
                     inv_hw, w1, w2, b2, hidden, dm, dw1, db1, dw2, db2, E, R);
  return lmn_launch_status("se_bwd");
}

int lmn_se_bwd_dm(const float* ds, const float* s, float inv_hw, const float* w1, const float* w2, const float* hidden,
                  float* dm, float* dvec, int B, int E, int R, lmn_stream_t stream) {
  LMN_REC(lmn_se_bwd_dm(ds, s, inv_hw, w1, w2, hidden, dm, dvec, B, E, R, stream));
  LMN_REQUIRE(ds && s && w1 && w2 && hidden && dm && dvec && B > 0 && E > 0 && R > 0, "se_bwd_dm: bad argument");
  LMN_REQUIRE((E + R + 256) * sizeof(float) <= 60000, "se_bwd_dm: E=%d too large", E);
  LMN_LAUNCH(se_bwd_dm_kernel, dim3(B), dim3(256), (E + R + 256) * sizeof(float), (hipStream_t)stream, ds, s, inv_hw, w1, w2,
             hidden, dm, dvec, E, R);
  return lmn_launch_status("se_bwd_dm");
}

int lmn_se_bwd_params(const float* dvec, const float* gsum, float inv_hw, const float* hidden, float* dw1, float* db1,
                      float* dw2, float* db2, int B, int E, int R, lmn_stream_t stream) {
  LMN_REC(lmn_se_bwd_params(dvec, gsum, inv_hw, hidden, dw1, db1, dw2, db2, B, E, R, stream));
  LMN_REQUIRE(dvec && gsum && hidden && dw1 && db1 && dw2 && db2 && B > 0 && E > 0 && R > 0, "se_bwd_params: bad argument");
  const int64_t total = 2 * (int64_t)E * R + E + R;
  LMN_LAUNCH(se_bwd_params_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dvec, gsum, inv_hw,
             hidden, dw1, db1, dw2, db2, B, E, R);
  return lmn_launch_status("se_bwd_params");
}

int lmn_up2_fwd(const void* x, void* y, int B, int Hin, int Win, int C, int x_cstride, int y_cstride, int act_dtype,
                lmn_stream_t stream) {
  LMN_REC(lmn_up2_fwd(x, y, B, Hin, Win, C, x_cstride, y_cstride, act_dtype, stream));
  LMN_REQUIRE_DT(act_dtype, "up2_fwd");
  LMN_REQUIRE(x && y && B > 0 && Hin > 0 && Win > 0 && C > 0 && C % 4 == 0 && x_cstride >= C && y_cstride >= C && x_cstride % 4 == 0 && y_cstride % 4 == 0, "up2_fwd: bad argument");
  LMN_REQUIRE((int64_t)4 * Win * Hin * x_cstride < (1LL << 31) && (int64_t)B * 2 * Hin < (1LL << 31), "up2_fwd: image too large for 32-bit row offsets");
  const dim3 grid((unsigned)lmn_cdiv((int64_t)2 * Win * (C / 4), 256), (unsigned)((int64_t)B * 2 * Hin < 65535 ? B * 2 * Hin : 65535));
  LMN_ACT_DISPATCH(act_dtype, LMN_LAUNCH((up2_fwd_kernel<T>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)x, (T*)y, B, Hin, Win, C / 4,
                     x_cstride, y_cstride, lmn_div_magic(C / 4)));
  return lmn_launch_status("up2_fwd");
}

int lmn_up2_bwd(const void* dy, void* dx, int B, int Hin, int Win, int C, int dy_cstride, int dx_cstride, int act_dtype,
                lmn_stream_t stream) {
  LMN_REC(lmn_up2_bwd(dy, dx, B, Hin, Win, C, dy_cstride, dx_cstride, act_dtype, stream));
  LMN_REQUIRE_DT(act_dtype, "up2_bwd");
  LMN_REQUIRE(dy && dx && B > 0 && Hin > 0 && Win > 0 && C > 0 && C % 4 == 0 && dy_cstride >= C && dx_cstride >= C && dy_cstride % 4 == 0 && dx_cstride % 4 == 0, "up2_bwd: bad argument");
  LMN_REQUIRE((int64_t)2 * Win * dy_cstride < (1LL << 31) && (int64_t)B * Hin < (1LL << 31), "up2_bwd: image too large for 32-bit row offsets");
  const dim3 grid((unsigned)lmn_cdiv((int64_t)Win * (C / 4), 256), (unsigned)((int64_t)B * Hin < 65535 ? B * Hin : 65535));
  LMN_ACT_DISPATCH(act_dtype, LMN_LAUNCH((up2_bwd_kernel<T>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)dy, (T*)dx, B, Hin, Win, C / 4,
                     dy_cstride, dx_cstride, lmn_div_magic(C / 4), lmn_prio_level((hipStream_t)stream)));
  return lmn_launch_status("up2_bwd");
}

int lmn_avgpool_fwd(const void* x, void* y, int B, int Hout, int Wout, int f, int C, int x_cstride, int y_cstride,
                    int act_dtype, lmn_stream_t stream) {
  LMN_REC(lmn_avgpool_fwd(x, y, B, Hout, Wout, f, C, x_cstride, y_cstride, act_dtype, stream));
  LMN_REQUIRE_DT(act_dtype, "avgpool_fwd");
  LMN_REQUIRE(x && y && B > 0 && Hout > 0 && Wout > 0 && f >= 1 && C % 4 == 0 && C >= 4 && C <= 1024 && x_cstride >= C && y_cstride >= C && x_cstride % 4 == 0, "avgpool_fwd: bad argument");
  if (C <= 256 && (int64_t)B * Hout * Wout < (1LL << 30)) {   // wave per output pixel
    const int total = B * Hout * Wout;
    LMN_ACT_DISPATCH(act_dtype, LMN_LAUNCH((avgpool_fwd_wave_kernel<T>), dim3(lmn_cdiv(total, 4)), dim3(256), 0, (hipStream_t)stream, (const T*)x, (T*)y,
                       Hout, Wout, f, C, x_cstride, y_cstride, total, lmn_div_magic(Wout), lmn_div_magic(Hout * Wout), lmn_div_magic(C / 4), lmn_prio_level((hipStream_t)stream)));
    return lmn_launch_status("avgpool_fwd");
  }
  LMN_ACT_DISPATCH(act_dtype, LMN_LAUNCH((avgpool_fwd_kernel<T>), dim3(B * Hout * Wout), dim3(256), (size_t)(C + 4 * 256 + (C > 256 ? C : 256)) * sizeof(float), (hipStream_t)stream, (const T*)x, (T*)y,
                     Hout, Wout, f, C, x_cstride, y_cstride));
  return lmn_launch_status("avgpool_fwd");
}

int lmn_avgpool_bwd(const void* dy, void* dx, int B, int Hout, int Wout, int f, int C, int dy_cstride, int dx_cstride,
                    int accumulate, int act_dtype, lmn_stream_t stream) {
  LMN_REC(lmn_avgpool_bwd(dy, dx, B, Hout, Wout, f, C, dy_cstride, dx_cstride, accumulate, act_dtype, stream));
  LMN_REQUIRE_DT(act_dtype, "avgpool_bwd");
  LMN_REQUIRE(dy && dx && B > 0 && Hout > 0 && Wout > 0 && f >= 1 && C % 4 == 0 && C >= 4 && dy_cstride >= C && dx_cstride >= C && dy_cstride % 4 == 0 && dx_cstride % 4 == 0, "avgpool_bwd: bad argument");
  LMN_REQUIRE((int64_t)B * Hout * f < (1LL << 31) && (int64_t)Wout * f * (C / 4) < (1LL << 31), "avgpool_bwd: image too large for 32-bit row / column indices");
  const dim3 grid((unsigned)lmn_cdiv((int64_t)Wout * f * (C / 4), 256), (unsigned)((int64_t)B * Hout * f < 65535 ? B * Hout * f : 65535));
  LMN_ACT_DISPATCH(act_dtype, LMN_LAUNCH((avgpool_bwd_kernel<T>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)dy, (T*)dx, B, Hout, Wout,
                     f, C / 4, dy_cstride, dx_cstride, accumulate, lmn_div_magic(C / 4), lmn_div_magic(f), lmn_prio_level((hipStream_t)stream)));
  return lmn_launch_status("avgpool_bwd");
}

int lmn_nchw_to_nhwc(const float* x, void* y, int B, int C, int H, int W, int y_cstride, int act_dtype, lmn_stream_t stream) {
  LMN_REC(lmn_nchw_to_nhwc(x, y, B, C, H, W, y_cstride, act_dtype, stream));
  LMN_REQUIRE_DT(act_dtype, "nchw_to_nhwc");
  LMN_REQUIRE(x && y && B > 0 && C > 0 && H > 0 && W > 0 && y_cstride >= C, "nchw_to_nhwc: bad argument");
  LMN_ACT_DISPATCH(act_dtype, LMN_LAUNCH((nchw_to_nhwc_kernel<T>), dim3(grid_for((int64_t)B * H * W)), dim3(256), 0, (hipStream_t)stream, x, (T*)y, B,
                     C, (int64_t)H * W, y_cstride));
  return lmn_launch_status("nchw_to_nhwc");
}

int lmn_nhwc_to_nchw(const void* x, float* y, int B, int C, int H, int W, int x_cstride, int act_dtype, lmn_stream_t stream) {
  LMN_REC(lmn_nhwc_to_nchw(x, y, B, C, H, W, x_cstride, act_dtype, stream));
  LMN_REQUIRE_DT(act_dtype, "nhwc_to_nchw");
  LMN_REQUIRE(x && y && B > 0 && C > 0 && H > 0 && W > 0 && x_cstride >= C, "nhwc_to_nchw: bad argument");
  LMN_ACT_DISPATCH(act_dtype, LMN_LAUNCH((nhwc_to_nchw_kernel<T>), dim3(grid_for((int64_t)B * H * W)), dim3(256), 0, (hipStream_t)stream, (const T*)x, y, B,
                     C, (int64_t)H * W, x_cstride));
  return lmn_launch_status("nhwc_to_nchw");
}

int lmn_segloss_fwd(const float* logits, const int64_t* target, const float* w_ce, const float* w_dice, int B, int C,
                    int64_t HW, float label_smoothing, float smooth, float* sums, float* coef, float* loss,
                    lmn_stream_t stream) {
  LMN_REQUIRE(logits && target && w_ce && w_dice && sums && coef && loss, "segloss_fwd: null pointer");
  LMN_REQUIRE(B > 0 && HW > 0 && (C == 2 || C == 3 || C == 4 || C == 8), "segloss_fwd: C=%d not in {2,3,4,8}", C);
  hipStream_t st = (hipStream_t)stream;
  const int grid = grid_for((int64_t)B * HW) > 1024 ? 1024 : grid_for((int64_t)B * HW);
  LMN_LAUNCH(fill_kernel, dim3(1), dim3(64), 0, st, sums, 0.f, (int64_t)(3 + 3 * C));
  float* sd = sums;
  if (g_lmn_det) {
    lmn_det_begin(st);
    sd = lmn_det_slots(st, (size_t)grid * (3 + 3 * C));
    LMN_REQUIRE(sd, "segloss_fwd: deterministic mode: no scratch");
  }
  switch (C) {
    case 2: LMN_LAUNCH((segloss_sums_kernel<2>), dim3(grid), dim3(256), 0, st, logits, target, w_ce, B, HW, sd, g_lmn_det); break;
    case 3: LMN_LAUNCH((segloss_sums_kernel<3>), dim3(grid), dim3(256), 0, st, logits, target, w_ce, B, HW, sd, g_lmn_det); break;
    case 4: LMN_LAUNCH((segloss_sums_kernel<4>), dim3(grid), dim3(256), 0, st, logits, target, w_ce, B, HW, sd, g_lmn_det); break;
    default: LMN_LAUNCH((segloss_sums_kernel<8>), dim3(grid), dim3(256), 0, st, logits, target, w_ce, B, HW, sd, g_lmn_det); break;
  }
  if (g_lmn_det) lmn_det_sum(st, sd, grid, 3 + 3 * C, sums);
  LMN_LAUNCH(segloss_finish_kernel, dim3(1), dim3(64), 0, st, sums, w_ce, w_dice, C, label_smoothing, smooth, loss, coef);
  return lmn_launch_status("segloss_fwd");
}

int lmn_segloss_bwd(const float* logits, const int64_t* target, const float* w_ce, const float* coef, const float* gscale,
                    int B, int C, int64_t HW, float* dlogits, lmn_stream_t stream) {
  LMN_REQUIRE(logits && target && w_ce && coef && dlogits, "segloss_bwd: null pointer");
  LMN_REQUIRE(B > 0 && HW > 0 && (C == 2 || C == 3 || C == 4 || C == 8), "segloss_bwd: C=%d not in {2,3,4,8}", C);
  hipStream_t st = (hipStream_t)stream;
  const int grid = grid_for((int64_t)B * HW);
  switch (C) {
    case 2: LMN_LAUNCH((segloss_bwd_kernel<2>), dim3(grid), dim3(256), 0, st, logits, target, w_ce, coef, gscale, B, HW, dlogits); break;
    case 3: LMN_LAUNCH((segloss_bwd_kernel<3>), dim3(grid), dim3(256), 0, st, logits, target, w_ce, coef, gscale, B, HW, dlogits); break;
    case 4: LMN_LAUNCH((segloss_bwd_kernel<4>), dim3(grid), dim3(256), 0, st, logits, target, w_ce, coef, gscale, B, HW, dlogits); break;
    default: LMN_LAUNCH((segloss_bwd_kernel<8>), dim3(grid), dim3(256), 0, st, logits, target, w_ce, coef, gscale, B, HW, dlogits); break;
  }
  return lmn_launch_status("segloss_bwd");
}

int lmn_confusion(const float* logits, const int64_t* target, int B, int C, int64_t HW, float* counts, lmn_stream_t stream) {
  LMN_REQUIRE(logits && target && counts, "confusion: null pointer");
  LMN_REQUIRE(B > 0 && HW > 0 && (C == 2 || C == 3 || C == 4), "confusion: C=%d not in {2,3,4}", C);
  LMN_REQUIRE((int64_t)B * HW < (1LL << 24) * 64, "confusion: more than 2^30 pixels per call");
  hipStream_t st = (hipStream_t)stream;
  const int grid = grid_for((int64_t)B * HW) > 512 ? 512 : grid_for((int64_t)B * HW);
  switch (C) {
    case 2: LMN_LAUNCH((confusion_kernel<2>), dim3(grid), dim3(256), 0, st, logits, target, B, HW, counts); break;
    case 3: LMN_LAUNCH((confusion_kernel<3>), dim3(grid), dim3(256), 0, st, logits, target, B, HW, counts); break;
    default: LMN_LAUNCH((confusion_kernel<4>), dim3(grid), dim3(256), 0, st, logits, target, B, HW, counts); break;
  }
  return lmn_launch_status("confusion");
}

int lmn_preprocess_u8(const uint8_t* images, const uint8_t* masks, const uint8_t* flips, int B, int Hs, int Ws, int H,
                      int W, const double* mean, const double* std, float* out, int64_t* labels, lmn_stream_t stream) {
  LMN_REQUIRE((images && out && mean && std) || (masks && labels), "preprocess_u8: nothing to do");
  LMN_REQUIRE(!images || (out && mean && std), "preprocess_u8: images need out, mean and std");
  LMN_REQUIRE(!masks || labels, "preprocess_u8: masks need labels");
  LMN_REQUIRE(B > 0 && Hs > 0 && Ws > 0 && H > 0 && W > 0, "preprocess_u8: empty tensor");
  LMN_REQUIRE(Hs < 32768 && Ws < 32768 && H < 32768 && W < 32768, "preprocess_u8: side above 32767");
  PrepGeom g{B, Hs, Ws, H, W, {0, 0, 0}, {1, 1, 1}};
  if (images) {
    for (int c = 0; c < 3; ++c) {
      LMN_REQUIRE(std[c] > 0.0, "preprocess_u8: std[%d] must be positive", c);
      g.m255[c] = mean[c] * 255.0;          // albumentations: mean * max_pixel_value, reciprocal(std * max_pixel_value)
      g.inv[c] = 1.0 / (std[c] * 255.0);
    }
  }
  const int64_t n = (int64_t)B * H * W;
  LMN_LAUNCH(preprocess_u8_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, images ? images : nullptr,
                     masks ? masks : nullptr, flips, images ? out : nullptr, masks ? labels : nullptr, g);
  return lmn_launch_status("preprocess_u8");
}

int lmn_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                   float eps, float weight_decay, float bias_corr1, float bias_corr2, lmn_stream_t stream) {
  LMN_REQUIRE(p && g && m && v && n > 0 && n % 4 == 0, "adamw_step: bad argument (n must be a multiple of 4)");
  LMN_REQUIRE(bias_corr1 > 0.f && bias_corr2 > 0.f, "adamw_step: bias corrections must be positive");
  LMN_LAUNCH(adamw_kernel, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n / 4, lr, beta1,
                     beta2, eps, weight_decay, 1.f / bias_corr1, 1.f / sqrtf(bias_corr2));
  return lmn_launch_status("adamw_step");
}

int lmn_fill(float* p, float v, int64_t n, lmn_stream_t stream) {
  LMN_REC(lmn_fill(p, v, n, stream));
  LMN_REQUIRE(p && n >= 0, "fill: bad argument");
  if (n == 0) return 0;
  LMN_LAUNCH(fill_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, v, n);
  return lmn_launch_status("fill");
}

int lmn_add(const void* a, const void* b, const void* c, const void* d, void* y, int64_t n, int act_dtype, lmn_stream_t stream) {
  LMN_REC(lmn_add(a, b, c, d, y, n, act_dtype, stream));
  LMN_REQUIRE_DT(act_dtype, "add");
  LMN_REQUIRE(a && b && y && n > 0 && n % 4 == 0, "add: bad argument (n must be a multiple of 4)");
  LMN_ACT_DISPATCH(act_dtype, LMN_LAUNCH((add_kernel<T>), dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, (const T*)a, (const T*)b, (const T*)c, (const T*)d, (T*)y, n / 4));
  return lmn_launch_status("add");
}

int lmn_affine2(const void* u, const void* v, const float* coef, void* y, int64_t rows, int C, int rp_w, int act_dtype, lmn_stream_t stream) {
  LMN_REC(lmn_affine2(u, v, coef, y, rows, C, rp_w, act_dtype, stream));
  LMN_REQUIRE_DT(act_dtype, "affine2");
  LMN_REQUIRE(u && v && coef && y && rows > 0 && C > 0 && C % 4 == 0, "affine2: bad argument (C must be a multiple of 4)");
  if (g_lmn_prof_on) lmn_prof_cost(4.0 * rows * C, (act_dtype == LMN_BF16 ? 2.0 : 4.0) * 3 * rows * C);
  LMN_ACT_DISPATCH(act_dtype, LMN_LAUNCH((affine2_kernel<T>), dim3(grid_for(rows * (C / 4))), dim3(256), 0, (hipStream_t)stream, (const T*)u, (const T*)v, coef,
                     (T*)y, rows, C / 4, rp_w));
  return lmn_launch_status("affine2");
}

int lmn_copy_slice(const void* x, void* y, int64_t rows, int C, int x_cstride, int y_cstride, int act_dtype, lmn_stream_t stream) {
  LMN_REC(lmn_copy_slice(x, y, rows, C, x_cstride, y_cstride, act_dtype, stream));
  LMN_REQUIRE_DT(act_dtype, "copy_slice");
  LMN_REQUIRE(x && y && rows > 0 && C > 0 && C % 4 == 0 && x_cstride >= C && y_cstride >= C && x_cstride % 4 == 0 && y_cstride % 4 == 0, "copy_slice: bad argument");
  LMN_ACT_DISPATCH(act_dtype, LMN_LAUNCH((copy_slice_kernel<T>), dim3(grid_for(rows * (C / 4))), dim3(256), 0, (hipStream_t)stream, (const T*)x, (T*)y, rows,
                     C / 4, x_cstride, y_cstride));
  return lmn_launch_status("copy_slice");
}

}  // extern "C"
