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
  GaGeom g{B, N, heads, hd, heads * hd, scale};
  LMN_ACT_DISPATCH(act_dtype, LMN_LAUNCH((gattn_fwd_kernel<T>), dim3(lmn_cdiv(N, GA_ROWS), heads, B), dim3(256), 0, (hipStream_t)stream, (const T*)qkv, (T*)out, lse, g));
  return lmn_launch_status("gattn_fwd");
}

int lmn_gattn_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* delta,
                  int B, int N, int heads, int hd, float scale, int act_dtype, lmn_stream_t stream) {
  LMN_REC(lmn_gattn_bwd(qkv, out, dout, lse, dqkv, delta, B, N, heads, hd, scale, act_dtype, stream));
  LMN_REQUIRE_DT(act_dtype, "gattn_bwd");
  LMN_REQUIRE(qkv && out && dout && lse && dqkv && delta && B > 0 && N > 0 && heads > 0, "gattn_bwd: bad argument");
  LMN_REQUIRE(hd >= 1 && hd <= GA_D, "gattn_bwd: head_dim %d > %d", hd, GA_D);
  GaGeom g{B, N, heads, hd, heads * hd, scale};
  const dim3 grid(lmn_cdiv(N, GA_ROWS), heads, B);
  LMN_ACT_DISPATCH(act_dtype, LMN_LAUNCH((gattn_bwd_q_kernel<T>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)qkv, (const T*)out, (const T*)dout, lse, (T*)dqkv, delta, g);
                   LMN_LAUNCH((gattn_bwd_kv_kernel<T>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)qkv, (const T*)dout, lse, delta, (T*)dqkv, g));
  return lmn_launch_status("gattn_bwd");
}

}  // extern "C"
