// Dense multi-head self-attention of the GFT bottleneck (core/modules.py:267-279): N = (H/16)*(W/16)
// tokens (484 at 352x352, 1024 at 512x512), 12 heads, head_dim = 31.  < 2 % of the step's FLOPs, so a
// compact flash-style VALU kernel: one thread per query row (q, running max/sum and the output row in
// registers), keys/values streamed through LDS in 64-key chunks and read as broadcasts.  The backward
// is the standard two-sweep form (query-owned dq; key-owned dk, dv) -- no atomics, deterministic.
#include "common.h"

namespace {

constexpr int GA_D = 32;    // padded head_dim
constexpr int GA_KC = 64;   // keys (or queries) per LDS chunk

struct GaGeom {
  int B, N, heads, hd, C;
  float scale;
};

// copy rows [r0, r0+GA_KC) of one (b, head, which) slab into LDS as [GA_KC][GA_D] (zero padded)
__device__ __forceinline__ void stage_rows(float* lds, const float* base, int64_t row_stride, int r0, int N, int hd) {
  for (int i = threadIdx.x; i < GA_KC * GA_D; i += blockDim.x) {
    const int r = i / GA_D, d = i - r * GA_D;
    float v = 0.f;
    if (r0 + r < N && d < hd) v = base[(int64_t)(r0 + r) * row_stride + d];
    lds[i] = v;
  }
}

__global__ __launch_bounds__(256) void gattn_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                        float* __restrict__ lse, const GaGeom g) {
  __shared__ float Ks[GA_KC * GA_D], Vs[GA_KC * GA_D];
  const int h = blockIdx.y, b = blockIdx.z;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool ok = i < g.N;
  const int64_t rs = 3 * g.C;
  const float* qb = qkv + (int64_t)b * g.N * rs + h * g.hd;
  float q[GA_D], o[GA_D];
#pragma unroll
  for (int d = 0; d < GA_D; ++d) {
    q[d] = (ok && d < g.hd) ? qb[(int64_t)i * rs + d] * g.scale : 0.f;
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
      float s = 0.f;
#pragma unroll
      for (int d = 0; d < GA_D; ++d) s += q[d] * Ks[j * GA_D + d];
      const float mn = fmaxf(m, s);
      const float corr = __expf(m - mn), p = __expf(s - mn);
      l = l * corr + p;
#pragma unroll
      for (int d = 0; d < GA_D; ++d) o[d] = o[d] * corr + p * Vs[j * GA_D + d];
      m = mn;
    }
  }
  if (ok) {
    const float inv = 1.0f / l;
    float* ob = out + ((int64_t)b * g.N + i) * g.C + h * g.hd;
#pragma unroll
    for (int d = 0; d < GA_D; ++d)
      if (d < g.hd) ob[d] = o[d] * inv;
    lse[((int64_t)b * g.heads + h) * g.N + i] = m + __logf(l);
  }
}

// sweep A: query-owned.  delta_i = do_i . o_i ;  dq_i = scale * sum_j p_ij (dp_ij - delta_i) k_j
__global__ __launch_bounds__(256) void gattn_bwd_q_kernel(const float* __restrict__ qkv, const float* __restrict__ out,
                                                          const float* __restrict__ dout, const float* __restrict__ lse,
                                                          float* __restrict__ dqkv, float* __restrict__ delta,
                                                          const GaGeom g) {
  __shared__ float Ks[GA_KC * GA_D], Vs[GA_KC * GA_D];
  const int h = blockIdx.y, b = blockIdx.z;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool ok = i < g.N;
  const int64_t rs = 3 * g.C;
  const float* qb = qkv + (int64_t)b * g.N * rs + h * g.hd;
  float q[GA_D], dO[GA_D], dq[GA_D];
  float dl = 0.f;
#pragma unroll
  for (int d = 0; d < GA_D; ++d) {
    const bool dk = ok && d < g.hd;
    q[d] = dk ? qb[(int64_t)i * rs + d] * g.scale : 0.f;
    dO[d] = dk ? dout[((int64_t)b * g.N + i) * g.C + h * g.hd + d] : 0.f;
    const float ov = dk ? out[((int64_t)b * g.N + i) * g.C + h * g.hd + d] : 0.f;
    dl += dO[d] * ov;
    dq[d] = 0.f;
  }
  const float L = ok ? lse[((int64_t)b * g.heads + h) * g.N + i] : 0.f;
  for (int k0 = 0; k0 < g.N; k0 += GA_KC) {
    __syncthreads();
    stage_rows(Ks, qb + g.C, rs, k0, g.N, g.hd);
    stage_rows(Vs, qb + 2 * g.C, rs, k0, g.N, g.hd);
    __syncthreads();
    const int kn = g.N - k0 < GA_KC ? g.N - k0 : GA_KC;
    for (int j = 0; j < kn; ++j) {
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int d = 0; d < GA_D; ++d) {
        s += q[d] * Ks[j * GA_D + d];
        dp += dO[d] * Vs[j * GA_D + d];
      }
      const float ds = __expf(s - L) * (dp - dl);
#pragma unroll
      for (int d = 0; d < GA_D; ++d) dq[d] += ds * Ks[j * GA_D + d];
    }
  }
  if (ok) {
    float* dqb = dqkv + ((int64_t)b * g.N + i) * rs + h * g.hd;
#pragma unroll
    for (int d = 0; d < GA_D; ++d)
      if (d < g.hd) dqb[d] = dq[d] * g.scale;
    delta[((int64_t)b * g.heads + h) * g.N + i] = dl;
  }
}

// sweep B: key-owned.  dv_j = sum_i p_ij do_i ;  dk_j = scale * sum_i p_ij (dp_ij - delta_i) q_i
__global__ __launch_bounds__(256) void gattn_bwd_kv_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                           const float* __restrict__ lse, const float* __restrict__ delta,
                                                           float* __restrict__ dqkv, const GaGeom g) {
  __shared__ float Qs[GA_KC * GA_D], Ds[GA_KC * GA_D], Ls[GA_KC], Dl[GA_KC];
  const int h = blockIdx.y, b = blockIdx.z;
  const int j = blockIdx.x * 256 + threadIdx.x;
  const bool ok = j < g.N;
  const int64_t rs = 3 * g.C;
  const float* qb = qkv + (int64_t)b * g.N * rs + h * g.hd;
  float k[GA_D], v[GA_D], dk[GA_D], dv[GA_D];
#pragma unroll
  for (int d = 0; d < GA_D; ++d) {
    const bool dok = ok && d < g.hd;
    k[d] = dok ? qb[(int64_t)j * rs + g.C + d] : 0.f;
    v[d] = dok ? qb[(int64_t)j * rs + 2 * g.C + d] : 0.f;
    dk[d] = dv[d] = 0.f;
  }
  const float* dob = dout + (int64_t)b * g.N * g.C + h * g.hd;
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
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int d = 0; d < GA_D; ++d) {
        s += Qs[i * GA_D + d] * k[d];
        dp += Ds[i * GA_D + d] * v[d];
      }
      const float p = __expf(s * g.scale - Ls[i]);
      const float ds = p * (dp - Dl[i]);
#pragma unroll
      for (int d = 0; d < GA_D; ++d) {
        dv[d] += p * Ds[i * GA_D + d];
        dk[d] += ds * Qs[i * GA_D + d];
      }
    }
  }
  if (ok) {
    float* db = dqkv + ((int64_t)b * g.N + j) * rs + h * g.hd;
#pragma unroll
    for (int d = 0; d < GA_D; ++d)
      if (d < g.hd) {
        db[g.C + d] = dk[d] * g.scale;
        db[2 * g.C + d] = dv[d];
      }
  }
}

}  // namespace

extern "C" {

int lmn_gattn_fwd(const float* qkv, float* out, float* lse, int B, int N, int heads, int hd, float scale,
                  lmn_stream_t stream) {
  LMN_REQUIRE(qkv && out && lse && B > 0 && N > 0 && heads > 0, "gattn_fwd: bad argument");
  LMN_REQUIRE(hd >= 1 && hd <= GA_D, "gattn_fwd: head_dim %d > %d", hd, GA_D);
  GaGeom g{B, N, heads, hd, heads * hd, scale};
  hipLaunchKernelGGL(gattn_fwd_kernel, dim3(lmn_cdiv(N, 256), heads, B), dim3(256), 0, (hipStream_t)stream, qkv, out, lse, g);
  return lmn_launch_status("gattn_fwd");
}

int lmn_gattn_bwd(const float* qkv, const float* out, const float* dout, const float* lse, float* dqkv, float* delta,
                  int B, int N, int heads, int hd, float scale, lmn_stream_t stream) {
  LMN_REQUIRE(qkv && out && dout && lse && dqkv && delta && B > 0 && N > 0 && heads > 0, "gattn_bwd: bad argument");
  LMN_REQUIRE(hd >= 1 && hd <= GA_D, "gattn_bwd: head_dim %d > %d", hd, GA_D);
  GaGeom g{B, N, heads, hd, heads * hd, scale};
  const dim3 grid(lmn_cdiv(N, 256), heads, B);
  hipLaunchKernelGGL(gattn_bwd_q_kernel, grid, dim3(256), 0, (hipStream_t)stream, qkv, out, dout, lse, dqkv, delta, g);
  hipLaunchKernelGGL(gattn_bwd_kv_kernel, grid, dim3(256), 0, (hipStream_t)stream, qkv, dout, lse, delta, dqkv, g);
  return lmn_launch_status("gattn_bwd");
}

}  // extern "C"
