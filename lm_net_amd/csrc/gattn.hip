// Dense multi-head self-attention of the GFT bottleneck (core/modules.py:267-279): N = (H/16)*(W/16)
// tokens (484 at 352x352, 1024 at 512x512), 12 heads, head_dim = 31.  < 2 % of the step's FLOPs, so a
// compact flash-style VALU kernel: four lanes per query row (each owns 8 of the 32 padded dims: q, the output
// slice and the running max/sum in registers), keys/values streamed through LDS in 64-key chunks.  The backward
// is the standard two-sweep form (query-owned dq; key-owned dk, dv) -- no atomics, deterministic.
#include "common.h"

namespace {

constexpr int GA_D = 32;    // padded head_dim
constexpr int GA_KC = 64;   // keys (or queries) per LDS chunk
constexpr int GA_P = 4;     // lanes per row: each owns GA_D / GA_P = 8 consecutive dims
constexpr int GA_DP = GA_D / GA_P;
constexpr int GA_ROWS = 256 / GA_P;  // rows (queries or keys) per block

typedef float f32x4g __attribute__((ext_vector_type(4)));

struct GaGeom {
  int B, N, heads, hd, C;
  float scale;
  int prio;   // wave priority of the launch stream (lmn_set_priority_stream)
};

// copy rows [r0, r0+GA_KC) of one (b, head, which) slab into LDS as [GA_KC][GA_D] (zero padded)
template <typename TA>
__device__ __forceinline__ void stage_rows(float* lds, const TA* base, int64_t row_stride, int r0, int N, int hd) {
  for (int i = threadIdx.x; i < GA_KC * GA_D; i += blockDim.x) {
    const int r = i / GA_D, d = i - r * GA_D;
    float v = 0.f;
    if (r0 + r < N && d < hd) v = ld1(base + (int64_t)(r0 + r) * row_stride + d);
    lds[i] = v;
  }
}

// sum over the 4 lanes of a row (lanes 4k..4k+3): two DPP quad permutes, no LDS
__device__ __forceinline__ float quad_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));  // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));  // quad_perm [2,3,0,1]
  return v;
}

// this lane's 8-dim slice of LDS row r
__device__ __forceinline__ void row_slice(const float* lds, int r, int part, float (&x)[GA_DP]) {
  const f32x4g a = *reinterpret_cast<const f32x4g*>(lds + r * GA_D + part * GA_DP);
  const f32x4g c = *reinterpret_cast<const f32x4g*>(lds + r * GA_D + part * GA_DP + 4);
  x[0] = a[0]; x[1] = a[1]; x[2] = a[2]; x[3] = a[3];
  x[4] = c[0]; x[5] = c[1]; x[6] = c[2]; x[7] = c[3];
}

// The row (query or key) a thread works on is shared by 4 lanes that split head_dim: 4x the waves of the
// one-thread-per-row form (484 tokens x 12 heads x 8 images is only 726 waves of rows -- less than one per SIMD),
// a quarter of the registers, and the dot products close with two DPP adds.
template <typename TA>
__global__ __launch_bounds__(256) void gattn_fwd_kernel(const TA* __restrict__ qkv, TA* __restrict__ out,
                                                        float* __restrict__ lse, const GaGeom g) {
  __shared__ __attribute__((aligned(16))) float Ks[GA_KC * GA_D], Vs[GA_KC * GA_D];
  const int h = blockIdx.y, b = blockIdx.z;
  const int part = threadIdx.x & (GA_P - 1);
  const int i = blockIdx.x * GA_ROWS + (threadIdx.x >> 2);
  const bool ok = i < g.N;
  const int64_t rs = 3 * g.C;
  const TA* qb = qkv + (int64_t)b * g.N * rs + h * g.hd;
  float q[GA_DP], o[GA_DP];
#pragma unroll
  for (int d = 0; d < GA_DP; ++d) {
    const int dd = part * GA_DP + d;
    q[d] = (ok && dd < g.hd) ? ld1(qb + (int64_t)i * rs + dd) * g.scale : 0.f;
    o[d] = 0.f;
  }
  float m = -3.0e38f, l = 0.f;
  for (int k0 = 0; k0 < g.N; k0 += GA_KC) {
    __syncthreads();
    stage_rows(Ks, qb + g.C, rs, k0, g.N, g.hd);
    stage_rows(Vs, qb + 2 * g.C, rs, k0, g.N, g.hd);
    __syncthreads();
    const int kn = g.N - k0 < GA_KC ? g.N - k0 : GA_KC;
    for (int j = 0; j < kn; ++j) {
      float kk[GA_DP], vv[GA_DP];
      row_slice(Ks, j, part, kk);
      row_slice(Vs, j, part, vv);
      float s = 0.f;
#pragma unroll
      for (int d = 0; d < GA_DP; ++d) s += q[d] * kk[d];
      s = quad_sum(s);
      const float mn = fmaxf(m, s);
      const float corr = __expf(m - mn), pj = __expf(s - mn);
      l = l * corr + pj;
#pragma unroll
      for (int d = 0; d < GA_DP; ++d) o[d] = o[d] * corr + pj * vv[d];
      m = mn;
    }
  }
  if (ok) {
    const float inv = 1.0f / l;
    TA* ob = out + ((int64_t)b * g.N + i) * g.C + h * g.hd;
#pragma unroll
    for (int d = 0; d < GA_DP; ++d)
      if (part * GA_DP + d < g.hd) st1(ob + part * GA_DP + d, o[d] * inv);
    if (part == 0) lse[((int64_t)b * g.heads + h) * g.N + i] = m + __logf(l);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// MFMA forms (v_mfma_f32_16x16x4_f32).  A wave owns 16 query rows (a block 64), keys / values stream through LDS in chunks
// of 64.  Per 16-key tile: S = Q K^T (8 MFMAs over the 32 padded dims; Q fragments stay in registers), the online-softmax
// update in the MFMA result layout (lane (g, n) holds rows 4g..4g+3 of column n: row maxima / sums are DPP reductions over the
// 16 lanes of a row group), P goes through a wave-private LDS tile into A-operand order, O += P V (2 x 4 MFMAs).
// 16 MFMAs + ~80 VALU per tile instead of 16 x (16 FMAs + 2 exp + 2 DPP) per key in the VALU form: 158 -> ~25 us at 484
// tokens x 12 heads x 8 images.
constexpr int GM_LD = 36;  // LDS row stride of the K / V chunks (floats): lanes (g, n) of a B-operand read hit 4n + g (mod 32)

__device__ __forceinline__ float row16_max(float v) {  // over the 16 lanes of a DPP row
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xF, 0xF, false)));  // row_ror:8
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xF, 0xF, false)));  // row_ror:4
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122, 0xF, 0xF, false)));  // row_ror:2
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xF, 0xF, false)));  // row_ror:1
  return v;
}
__device__ __forceinline__ float row16_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xF, 0xF, false));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xF, 0xF, false));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122, 0xF, 0xF, false));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xF, 0xF, false));
  return v;
}
// rows [r0, r0+64) of one (b, head, which) slab -> LDS [64][GM_LD] (zero past N / past head_dim).  256 threads x 8 elements, in
// two halves: stage_issue requests a chunk into registers (all eight loads in flight), stage_commit stores it -- the kernels
// request chunk c+1 before they compute on chunk c, so a chunk's round trip runs under the MFMAs of the previous one
template <typename TA>
__device__ __forceinline__ void stage_issue(float (&v)[8], const TA* base, int64_t row_stride, int r0, int N, int hd) {
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int i = threadIdx.x + k * 256;
    const int r = i >> 5, d = i & 31;
    const bool in = r0 + r < N && d < hd;
    v[k] = ld1(base + (in ? (int64_t)(r0 + r) * row_stride + d : 0));
    if (!in) v[k] = 0.f;
  }
}
__device__ __forceinline__ void stage_commit(float* lds, const float (&v)[8]) {
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int i = threadIdx.x + k * 256;
    lds[(i >> 5) * GM_LD + (i & 31)] = v[k];
  }
}
template <typename TA>
__device__ __forceinline__ void stage_rows_ld(float* lds, const TA* base, int64_t row_stride, int r0, int N, int hd) {
  float v[8];
  stage_issue(v, base, row_stride, r0, N, hd);
  stage_commit(lds, v);
}

template <typename TA>
__global__ __launch_bounds__(256) void gattn_fwd_mfma_kernel(const TA* __restrict__ qkv, TA* __restrict__ out,
                                                             float* __restrict__ lse, const GaGeom g) {
  __shared__ __attribute__((aligned(16))) float Ks[GA_KC * GM_LD], Vs[GA_KC * GM_LD], Ps[4][16 * 17];
  lmn_setprio_level(g.prio);
  const int h = blockIdx.y, b = blockIdx.z;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int gq = lane >> 4, n = lane & 15;
  const int q0 = blockIdx.x * 64 + wv * 16;  // this wave's first query row
  const int64_t rs = 3 * g.C;
  const TA* qb = qkv + (int64_t)b * g.N * rs + h * g.hd;
  // A operand of S = Q K^T: lane (gq, n) holds Q[q0 + n][4j + gq], j = 0..7 (pre-scaled)
  float qf[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int d = 4 * j + gq;
    qf[j] = (q0 + n < g.N && d < g.hd) ? ld1(qb + (int64_t)(q0 + n) * rs + d) * g.scale : 0.f;
  }
  f32x4g o0 = f32x4g{0.f, 0.f, 0.f, 0.f}, o1 = o0;  // O[rows 4gq..4gq+3][dims n, 16 + n]
  float mrow[4], lrow[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) { mrow[r] = -3.0e38f; lrow[r] = 0.f; }
  float* ps = Ps[wv];
  float sk[8], sv8[8];
  stage_issue(sk, qb + g.C, rs, 0, g.N, g.hd);
  stage_issue(sv8, qb + 2 * g.C, rs, 0, g.N, g.hd);
  for (int k0 = 0; k0 < g.N; k0 += GA_KC) {
    __syncthreads();
    stage_commit(Ks, sk);
    stage_commit(Vs, sv8);
    __syncthreads();
    if (k0 + GA_KC < g.N) {  // the next chunk's round trip runs under this chunk's tiles
      stage_issue(sk, qb + g.C, rs, k0 + GA_KC, g.N, g.hd);
      stage_issue(sv8, qb + 2 * g.C, rs, k0 + GA_KC, g.N, g.hd);
    }
    const int ntile = (min(GA_KC, g.N - k0) + 15) >> 4;
    for (int t = 0; t < ntile; ++t) {
      // S tile: rows = this wave's queries, columns = keys k0 + 16t + n
      f32x4g sacc = f32x4g{0.f, 0.f, 0.f, 0.f};
      const float* kr = Ks + (t * 16 + n) * GM_LD + gq;
#pragma unroll
      for (int j = 0; j < 8; ++j) sacc = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[j], kr[4 * j], sacc, 0, 0, 0);
      const bool kin = k0 + t * 16 + n < g.N;
      float corr[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float sv = kin ? sacc[r] : -3.0e38f;
        const float mn = fmaxf(mrow[r], row16_max(sv));
        const float pj = kin ? __expf(sv - mn) : 0.f;
        corr[r] = __expf(mrow[r] - mn);
        lrow[r] = lrow[r] * corr[r] + row16_sum(pj);
        mrow[r] = mn;
        ps[(4 * gq + r) * 17 + n] = pj;   // P[row][key] for the A-operand read below (wave-private tile)
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) { o0[r] *= corr[r]; o1[r] *= corr[r]; }
      // O += P V: A = P[row n'][key 4j + gq'] (read back in operand order), B = V[key 4j + gq][dim n (+16)]
      const float* vr = Vs + (t * 16 + gq) * GM_LD + n;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float pa = ps[n * 17 + 4 * j + gq];
        o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(pa, vr[(4 * j) * GM_LD], o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(pa, vr[(4 * j) * GM_LD + 16], o1, 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = q0 + 4 * gq + r;
    if (i < g.N) {
      const float inv = 1.0f / lrow[r];
      TA* ob = out + ((int64_t)b * g.N + i) * g.C + h * g.hd;
      if (n < g.hd) st1(ob + n, o0[r] * inv);
      if (16 + n < g.hd) st1(ob + 16 + n, o1[r] * inv);
      if (n == 0) lse[((int64_t)b * g.heads + h) * g.N + i] = mrow[r] + __logf(lrow[r]);
    }
  }
}

// Backward, MFMA sweep A (query-owned, 16 query rows per wave): per 16-key tile S = Q K^T and dP = dO V^T (8 + 8 MFMAs, both in
// the result layout rows = queries, columns = keys), dS = exp(S - lse) * (dP - delta), then dQ += dS K (dS through the
// wave-private LDS tile into A-operand order; 2 x 4 MFMAs).  delta_i = dO_i . O_i is formed once per row and written for sweep B.
template <typename TA>
__global__ __launch_bounds__(256) void gattn_bwd_q_mfma_kernel(const TA* __restrict__ qkv, const TA* __restrict__ out,
                                                               const TA* __restrict__ dout, const float* __restrict__ lse,
                                                               TA* __restrict__ dqkv, float* __restrict__ delta,
                                                               const GaGeom g) {
  __shared__ __attribute__((aligned(16))) float Ks[GA_KC * GM_LD], Vs[GA_KC * GM_LD], Ps[4][16 * 17];
  lmn_setprio_level(g.prio);
  const int h = blockIdx.y, b = blockIdx.z;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int gq = lane >> 4, n = lane & 15;
  const int q0 = blockIdx.x * 64 + wv * 16;
  const int64_t rs = 3 * g.C;
  const TA* qb = qkv + (int64_t)b * g.N * rs + h * g.hd;
  const int64_t lb = ((int64_t)b * g.heads + h) * g.N;
  float qf[8], df[8];  // A operands: Q (pre-scaled) and dO of row q0 + n, dims 4j + gq
  float dl = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int d = 4 * j + gq;
    const bool ok = q0 + n < g.N && d < g.hd;
    const int64_t oi = ((int64_t)b * g.N + q0 + n) * g.C + h * g.hd + d;
    qf[j] = ok ? ld1(qb + (int64_t)(q0 + n) * rs + d) * g.scale : 0.f;
    df[j] = ok ? ld1(dout + oi) : 0.f;
    dl += df[j] * (ok ? ld1(out + oi) : 0.f);
  }
  dl += __shfl_xor(dl, 16, 64);   // sum over the four row groups: every lane with the same n holds delta of row q0 + n
  dl += __shfl_xor(dl, 32, 64);
  if (gq == 0 && q0 + n < g.N) delta[lb + q0 + n] = dl;
  float lr[4], dr[4];             // lse / delta of the rows this lane holds in the result layout (rows 4gq + r)
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    dr[r] = __shfl(dl, 4 * gq + r, 64);
    lr[r] = q0 + 4 * gq + r < g.N ? lse[lb + q0 + 4 * gq + r] : 0.f;
  }
  f32x4g a0 = f32x4g{0.f, 0.f, 0.f, 0.f}, a1 = a0;  // dQ[rows 4gq..4gq+3][dims n, 16 + n]
  float* ps = Ps[wv];
  float sk[8], sv8[8];
  stage_issue(sk, qb + g.C, rs, 0, g.N, g.hd);
  stage_issue(sv8, qb + 2 * g.C, rs, 0, g.N, g.hd);
  for (int k0 = 0; k0 < g.N; k0 += GA_KC) {
    __syncthreads();
    stage_commit(Ks, sk);
    stage_commit(Vs, sv8);
    __syncthreads();
    if (k0 + GA_KC < g.N) {  // the next chunk's round trip runs under this chunk's tiles
      stage_issue(sk, qb + g.C, rs, k0 + GA_KC, g.N, g.hd);
      stage_issue(sv8, qb + 2 * g.C, rs, k0 + GA_KC, g.N, g.hd);
    }
    const int ntile = (min(GA_KC, g.N - k0) + 15) >> 4;
    for (int t = 0; t < ntile; ++t) {
      f32x4g sacc = f32x4g{0.f, 0.f, 0.f, 0.f}, pacc = sacc;
      const float* kr = Ks + (t * 16 + n) * GM_LD + gq;
      const float* vr = Vs + (t * 16 + n) * GM_LD + gq;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        sacc = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[j], kr[4 * j], sacc, 0, 0, 0);
        pacc = __builtin_amdgcn_mfma_f32_16x16x4f32(df[j], vr[4 * j], pacc, 0, 0, 0);
      }
      const bool kin = k0 + t * 16 + n < g.N;
#pragma unroll
      for (int r = 0; r < 4; ++r) ps[(4 * gq + r) * 17 + n] = kin ? __expf(sacc[r] - lr[r]) * (pacc[r] - dr[r]) : 0.f;
      const float* kc = Ks + (t * 16 + gq) * GM_LD + n;   // B = K[key 4j + gq][dim n (+16)]
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float da = ps[n * 17 + 4 * j + gq];
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(da, kc[(4 * j) * GM_LD], a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(da, kc[(4 * j) * GM_LD + 16], a1, 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = q0 + 4 * gq + r;
    if (i < g.N) {
      TA* dqb = dqkv + ((int64_t)b * g.N + i) * rs + h * g.hd;
      if (n < g.hd) st1(dqb + n, a0[r] * g.scale);
      if (16 + n < g.hd) st1(dqb + 16 + n, a1[r] * g.scale);
    }
  }
}

// Backward, MFMA sweep B (key-owned, 16 key rows per wave; queries / dO / lse / delta stream through LDS): per 16-query tile
// S^T = K Q^T and dP^T = V dO^T (rows = keys, columns = queries: lse / delta are per COLUMN, i.e. lane-local), P^T and dS^T go
// through the wave-private tile into A-operand order, dV += P^T dO, dK += dS^T Q (scaled at the store).
template <typename TA>
__global__ __launch_bounds__(256) void gattn_bwd_kv_mfma_kernel(const TA* __restrict__ qkv, const TA* __restrict__ dout,
                                                                const float* __restrict__ lse, const float* __restrict__ delta,
                                                                TA* __restrict__ dqkv, const GaGeom g) {
  __shared__ __attribute__((aligned(16))) float Qs[GA_KC * GM_LD], Ds[GA_KC * GM_LD], Ps[4][2][16 * 17];
  __shared__ float Ls[GA_KC], Dl[GA_KC];
  lmn_setprio_level(g.prio);
  const int h = blockIdx.y, b = blockIdx.z;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int gq = lane >> 4, n = lane & 15;
  const int j0 = blockIdx.x * 64 + wv * 16;  // this wave's first key row
  const int64_t rs = 3 * g.C;
  const TA* qb = qkv + (int64_t)b * g.N * rs + h * g.hd;
  const int64_t lb = ((int64_t)b * g.heads + h) * g.N;
  float kf[8], vf[8];  // A operands: K and V of row j0 + n, dims 4j + gq
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int d = 4 * j + gq;
    const bool ok = j0 + n < g.N && d < g.hd;
    kf[j] = ok ? ld1(qb + (int64_t)(j0 + n) * rs + g.C + d) : 0.f;
    vf[j] = ok ? ld1(qb + (int64_t)(j0 + n) * rs + 2 * g.C + d) : 0.f;
  }
  f32x4g dk0 = f32x4g{0.f, 0.f, 0.f, 0.f}, dk1 = dk0, dv0 = dk0, dv1 = dk0;
  float* pp = Ps[wv][0];
  float* pd = Ps[wv][1];
  const TA* dob = dout + (int64_t)b * g.N * g.C + h * g.hd;
  float sq[8], sd[8], sl = 0.f, sdl = 0.f;
  auto issue_all = [&](int i0) __attribute__((always_inline)) {
    stage_issue(sq, qb, rs, i0, g.N, g.hd);     // (Q is scaled where it is used: S^T and dK below)
    stage_issue(sd, dob, g.C, i0, g.N, g.hd);
    if (threadIdx.x < GA_KC) {
      const bool in = i0 + (int)threadIdx.x < g.N;
      sl = in ? lse[lb + i0 + threadIdx.x] : 0.f;
      sdl = in ? delta[lb + i0 + threadIdx.x] : 0.f;
    }
  };
  issue_all(0);
  for (int i0 = 0; i0 < g.N; i0 += GA_KC) {
    __syncthreads();
    stage_commit(Qs, sq);
    stage_commit(Ds, sd);
    if (threadIdx.x < GA_KC) { Ls[threadIdx.x] = sl; Dl[threadIdx.x] = sdl; }
    __syncthreads();
    if (i0 + GA_KC < g.N) issue_all(i0 + GA_KC);
    const int ntile = (min(GA_KC, g.N - i0) + 15) >> 4;
    for (int t = 0; t < ntile; ++t) {
      f32x4g sacc = f32x4g{0.f, 0.f, 0.f, 0.f}, pacc = sacc;
      const float* qr = Qs + (t * 16 + n) * GM_LD + gq;   // B = Q^T / dO^T: [dim 4j + gq][query n]
      const float* dr = Ds + (t * 16 + n) * GM_LD + gq;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        sacc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[j], qr[4 * j], sacc, 0, 0, 0);
        pacc = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[j], dr[4 * j], pacc, 0, 0, 0);
      }
      const bool qin = i0 + t * 16 + n < g.N;
      const float L = Ls[t * 16 + n], D = Dl[t * 16 + n];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = qin ? __expf(sacc[r] * g.scale - L) : 0.f;
        pp[(4 * gq + r) * 17 + n] = p;
        pd[(4 * gq + r) * 17 + n] = p * (pacc[r] - D);
      }
      const float* dc = Ds + (t * 16 + gq) * GM_LD + n;   // B = dO[query 4j + gq][dim n (+16)]
      const float* qc = Qs + (t * 16 + gq) * GM_LD + n;   // B = Q [query 4j + gq][dim n (+16)]
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float pa = pp[n * 17 + 4 * j + gq], da = pd[n * 17 + 4 * j + gq];
        dv0 = __builtin_amdgcn_mfma_f32_16x16x4f32(pa, dc[(4 * j) * GM_LD], dv0, 0, 0, 0);
        dv1 = __builtin_amdgcn_mfma_f32_16x16x4f32(pa, dc[(4 * j) * GM_LD + 16], dv1, 0, 0, 0);
        dk0 = __builtin_amdgcn_mfma_f32_16x16x4f32(da, qc[(4 * j) * GM_LD], dk0, 0, 0, 0);
        dk1 = __builtin_amdgcn_mfma_f32_16x16x4f32(da, qc[(4 * j) * GM_LD + 16], dk1, 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int jrow = j0 + 4 * gq + r;
    if (jrow < g.N) {
      TA* db = dqkv + ((int64_t)b * g.N + jrow) * rs + h * g.hd;
      if (n < g.hd) { st1(db + g.C + n, dk0[r] * g.scale); st1(db + 2 * g.C + n, dv0[r]); }
      if (16 + n < g.hd) { st1(db + g.C + 16 + n, dk1[r] * g.scale); st1(db + 2 * g.C + 16 + n, dv1[r]); }
    }
  }
}

// sweep A: query-owned.  delta_i = do_i . o_i ;  dq_i = scale * sum_j p_ij (dp_ij - delta_i) k_j
template <typename TA>
__global__ __launch_bounds__(256) void gattn_bwd_q_kernel(const TA* __restrict__ qkv, const TA* __restrict__ out,
                                                          const TA* __restrict__ dout, const float* __restrict__ lse,
                                                          TA* __restrict__ dqkv, float* __restrict__ delta,
                                                          const GaGeom g) {
  __shared__ __attribute__((aligned(16))) float Ks[GA_KC * GA_D], Vs[GA_KC * GA_D];
  const int h = blockIdx.y, b = blockIdx.z;
  const int part = threadIdx.x & (GA_P - 1);
  const int i = blockIdx.x * GA_ROWS + (threadIdx.x >> 2);
  const bool ok = i < g.N;
  const int64_t rs = 3 * g.C;
  const TA* qb = qkv + (int64_t)b * g.N * rs + h * g.hd;
  float q[GA_DP], dO[GA_DP], dq[GA_DP];
  float dl = 0.f;
#pragma unroll
  for (int d = 0; d < GA_DP; ++d) {
    const int dd = part * GA_DP + d;
    const bool dk = ok && dd < g.hd;
    q[d] = dk ? ld1(qb + (int64_t)i * rs + dd) * g.scale : 0.f;
    dO[d] = dk ? ld1(dout + ((int64_t)b * g.N + i) * g.C + h * g.hd + dd) : 0.f;
    const float ov = dk ? ld1(out + ((int64_t)b * g.N + i) * g.C + h * g.hd + dd) : 0.f;
    dl += dO[d] * ov;
    dq[d] = 0.f;
  }
  dl = quad_sum(dl);
  const float L = ok ? lse[((int64_t)b * g.heads + h) * g.N + i] : 0.f;
  for (int k0 = 0; k0 < g.N; k0 += GA_KC) {
    __syncthreads();
    stage_rows(Ks, qb + g.C, rs, k0, g.N, g.hd);
    stage_rows(Vs, qb + 2 * g.C, rs, k0, g.N, g.hd);
    __syncthreads();
    const int kn = g.N - k0 < GA_KC ? g.N - k0 : GA_KC;
    for (int j = 0; j < kn; ++j) {
      float kk[GA_DP], vv[GA_DP];
      row_slice(Ks, j, part, kk);
      row_slice(Vs, j, part, vv);
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int d = 0; d < GA_DP; ++d) {
        s += q[d] * kk[d];
        dp += dO[d] * vv[d];
      }
      s = quad_sum(s);
      dp = quad_sum(dp);
      const float ds = __expf(s - L) * (dp - dl);
#pragma unroll
      for (int d = 0; d < GA_DP; ++d) dq[d] += ds * kk[d];
    }
  }
  if (ok) {
    TA* dqb = dqkv + ((int64_t)b * g.N + i) * rs + h * g.hd;
#pragma unroll
    for (int d = 0; d < GA_DP; ++d)
      if (part * GA_DP + d < g.hd) st1(dqb + part * GA_DP + d, dq[d] * g.scale);
    if (part == 0) delta[((int64_t)b * g.heads + h) * g.N + i] = dl;
  }
}

// sweep B: key-owned.  dv_j = sum_i p_ij do_i ;  dk_j = scale * sum_i p_ij (dp_ij - delta_i) q_i
template <typename TA>
__global__ __launch_bounds__(256) void gattn_bwd_kv_kernel(const TA* __restrict__ qkv, const TA* __restrict__ dout,
                                                           const float* __restrict__ lse, const float* __restrict__ delta,
                                                           TA* __restrict__ dqkv, const GaGeom g) {
  __shared__ __attribute__((aligned(16))) float Qs[GA_KC * GA_D], Ds[GA_KC * GA_D];
  __shared__ float Ls[GA_KC], Dl[GA_KC];
  const int h = blockIdx.y, b = blockIdx.z;
  const int part = threadIdx.x & (GA_P - 1);
  const int j = blockIdx.x * GA_ROWS + (threadIdx.x >> 2);
  const bool ok = j < g.N;
  const int64_t rs = 3 * g.C;
  const TA* qb = qkv + (int64_t)b * g.N * rs + h * g.hd;
  float k[GA_DP], v[GA_DP], dk[GA_DP], dv[GA_DP];
#pragma unroll
  for (int d = 0; d < GA_DP; ++d) {
    const int dd = part * GA_DP + d;
    const bool dok = ok && dd < g.hd;
    k[d] = dok ? ld1(qb + (int64_t)j * rs + g.C + dd) : 0.f;
    v[d] = dok ? ld1(qb + (int64_t)j * rs + 2 * g.C + dd) : 0.f;
    dk[d] = dv[d] = 0.f;
  }
  const TA* dob = dout + (int64_t)b * g.N * g.C + h * g.hd;
  for (int i0 = 0; i0 < g.N; i0 += GA_KC) {
    __syncthreads();
    stage_rows(Qs, qb, rs, i0, g.N, g.hd);
    stage_rows(Ds, dob, g.C, i0, g.N, g.hd);
    for (int t = threadIdx.x; t < GA_KC; t += 256) {
      const bool in = i0 + t < g.N;
      Ls[t] = in ? lse[((int64_t)b * g.heads + h) * g.N + i0 + t] : 0.f;
      Dl[t] = in ? delta[((int64_t)b * g.heads + h) * g.N + i0 + t] : 0.f;
    }
    __syncthreads();
    const int qn = g.N - i0 < GA_KC ? g.N - i0 : GA_KC;
    for (int i = 0; i < qn; ++i) {
      float qq[GA_DP], dd[GA_DP];
      row_slice(Qs, i, part, qq);
      row_slice(Ds, i, part, dd);
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int d = 0; d < GA_DP; ++d) {
        s += qq[d] * k[d];
        dp += dd[d] * v[d];
      }
      s = quad_sum(s);
      dp = quad_sum(dp);
      const float pij = __expf(s * g.scale - Ls[i]);
      const float ds = pij * (dp - Dl[i]);
#pragma unroll
      for (int d = 0; d < GA_DP; ++d) {
        dv[d] += pij * dd[d];
        dk[d] += ds * qq[d];
      }
    }
  }
  if (ok) {
    TA* db = dqkv + ((int64_t)b * g.N + j) * rs + h * g.hd;
#pragma unroll
    for (int d = 0; d < GA_DP; ++d)
      if (part * GA_DP + d < g.hd) {
        st1(db + g.C + part * GA_DP + d, dk[d] * g.scale);
        st1(db + 2 * g.C + part * GA_DP + d, dv[d]);
      }
  }
}

}  // namespace

extern "C" {

int lmn_gattn_fwd(const void* qkv, void* out, float* lse, int B, int N, int heads, int hd, float scale, int act_dtype,
                  lmn_stream_t stream) {
  LMN_REC(lmn_gattn_fwd(qkv, out, lse, B, N, heads, hd, scale, act_dtype, stream));
  LMN_REQUIRE_DT(act_dtype, "gattn_fwd");
  LMN_REQUIRE(qkv && out && lse && B > 0 && N > 0 && heads > 0, "gattn_fwd: bad argument");
  LMN_REQUIRE(hd >= 1 && hd <= GA_D, "gattn_fwd: head_dim %d > %d", hd, GA_D);
  GaGeom g{B, N, heads, hd, heads * hd, scale, lmn_prio_level((hipStream_t)stream)};
  static int mf = -1;
  if (mf < 0) { const char* e = getenv("LMN_GATTN_MFMA"); mf = e ? atoi(e) : 1; }
  if (mf) {  // MFMA form (LMN_GATTN_MFMA=0: the VALU form, A/B runs)
    LMN_ACT_DISPATCH(act_dtype, LMN_LAUNCH((gattn_fwd_mfma_kernel<T>), dim3(lmn_cdiv(N, 64), heads, B), dim3(256), 0, (hipStream_t)stream, (const T*)qkv, (T*)out, lse, g));
    return lmn_launch_status("gattn_fwd");
  }
  LMN_ACT_DISPATCH(act_dtype, LMN_LAUNCH((gattn_fwd_kernel<T>), dim3(lmn_cdiv(N, GA_ROWS), heads, B), dim3(256), 0, (hipStream_t)stream, (const T*)qkv, (T*)out, lse, g));
  return lmn_launch_status("gattn_fwd");
}

int lmn_gattn_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* delta,
                  int B, int N, int heads, int hd, float scale, int act_dtype, lmn_stream_t stream) {
  LMN_REC(lmn_gattn_bwd(qkv, out, dout, lse, dqkv, delta, B, N, heads, hd, scale, act_dtype, stream));
  LMN_REQUIRE_DT(act_dtype, "gattn_bwd");
  LMN_REQUIRE(qkv && out && dout && lse && dqkv && delta && B > 0 && N > 0 && heads > 0, "gattn_bwd: bad argument");
  LMN_REQUIRE(hd >= 1 && hd <= GA_D, "gattn_bwd: head_dim %d > %d", hd, GA_D);
  GaGeom g{B, N, heads, hd, heads * hd, scale, lmn_prio_level((hipStream_t)stream)};
  static int mf = -1;
  if (mf < 0) { const char* e = getenv("LMN_GATTN_MFMA"); mf = e ? atoi(e) : 1; }
  if (mf) {
    const dim3 mgrid(lmn_cdiv(N, 64), heads, B);
    LMN_ACT_DISPATCH(act_dtype, LMN_LAUNCH((gattn_bwd_q_mfma_kernel<T>), mgrid, dim3(256), 0, (hipStream_t)stream, (const T*)qkv, (const T*)out, (const T*)dout, lse, (T*)dqkv, delta, g);
                     LMN_LAUNCH((gattn_bwd_kv_mfma_kernel<T>), mgrid, dim3(256), 0, (hipStream_t)stream, (const T*)qkv, (const T*)dout, lse, delta, (T*)dqkv, g));
    return lmn_launch_status("gattn_bwd");
  }
  const dim3 grid(lmn_cdiv(N, GA_ROWS), heads, B);
  LMN_ACT_DISPATCH(act_dtype, LMN_LAUNCH((gattn_bwd_q_kernel<T>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)qkv, (const T*)out, (const T*)dout, lse, (T*)dqkv, delta, g);
                   LMN_LAUNCH((gattn_bwd_kv_kernel<T>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)qkv, (const T*)dout, lse, delta, (T*)dqkv, g));
  return lmn_launch_status("gattn_bwd");
}

}  // extern "C"
