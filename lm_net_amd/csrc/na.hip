// Fused 2-D neighborhood attention core: qk + relative-position bias + softmax + av in
// one kernel, attention weights never leave registers (upstream natten materialises attn [B,h,H,W,9]
// and crosses HBM with it four times).  Semantics: oracle/natten_ref.py (clamped 3x3 window, rpb
// index = (neighbour - query) + 2).
//
// Layout-first design for head_dim in {1,2,4,8} and 12 heads: one thread owns a float4 of channels of
// one pixel (= 4, 2, 1 or 1/2 heads), lanes walk the channel axis first so a wave reads whole NHWC
// pixels (48..384 B) contiguously; the per-head dot product is a 1-, 2-, 4-wide horizontal add (plus
// one lane-pair shuffle for hd = 8).  The 9x re-use of k/v between neighbouring queries is served by
// L1/L2: HBM traffic stays at the algorithmic 4*C*4 B per pixel.
#include "common.h"

namespace {

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// per-component head sums of a float4 of products (replicated inside each head)
template <int HD>
__device__ __forceinline__ f32x4 head_sum(f32x4 p) {
  if (HD == 1) return p;
  if (HD == 2) {
    const float a = p[0] + p[1], b = p[2] + p[3];
    return f32x4{a, a, b, b};
  }
  float s = p[0] + p[1] + p[2] + p[3];
  if (HD == 8) s += __shfl_xor(s, 1, 64);
  if (HD > 8) {  // hd = 16, 32: more lane pairs
#pragma unroll
    for (int m = 1; m < HD / 4; m <<= 1) s += __shfl_xor(s, m, 64);
  }
  return f32x4{s, s, s, s};
}

struct NaGeom {
  int B, H, W, C, C4, heads;
  float scale;
};

// window start (clamped, never padded) -- oracle/natten_ref.py: window_start
__device__ __forceinline__ int wstart(int i, int L) {
  int s = i - 1;
  s = s < 0 ? 0 : s;
  return s > L - 3 ? L - 3 : s;
}

template <int HD>
__global__ __launch_bounds__(256) void na_fwd_kernel(const float* __restrict__ qkv, const float* __restrict__ rpb,
                                                     float* __restrict__ out, const NaGeom g) {
  const int64_t total = (int64_t)g.B * g.H * g.W * g.C4;
  const int64_t nit = (total + (int64_t)gridDim.x * 256 - 1) / ((int64_t)gridDim.x * 256);
  for (int64_t it = 0; it < nit; ++it) {
    int64_t idx = (it * gridDim.x + blockIdx.x) * 256 + threadIdx.x;
    const bool ok = idx < total;
    if (!ok) idx = total - 1;  // keep every lane in the shuffles
    const int c = (int)(idx % g.C4) * 4;
    const int64_t pix = idx / g.C4;
    const int x = (int)(pix % g.W);
    const int y = (int)((pix / g.W) % g.H);
    const int b = (int)(pix / ((int64_t)g.W * g.H));
    const int sy = wstart(y, g.H), sx = wstart(x, g.W);
    const float* base = qkv + (int64_t)b * g.H * g.W * 3 * g.C;
    const f32x4 q = ld4(base + ((int64_t)y * g.W + x) * 3 * g.C + c) * g.scale;
    int hidx[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) hidx[k] = ((c + k) / HD) * 25;
    f32x4 l[9];
    f32x4 mx = f32x4{-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
#pragma unroll
    for (int ki = 0; ki < 3; ++ki)
#pragma unroll
      for (int kj = 0; kj < 3; ++kj) {
        const int ny = sy + ki, nx = sx + kj;
        const f32x4 kk = ld4(base + ((int64_t)ny * g.W + nx) * 3 * g.C + g.C + c);
        f32x4 s = head_sum<HD>(q * kk);
        const int bo = (ny - y + 2) * 5 + (nx - x + 2);
#pragma unroll
        for (int k = 0; k < 4; ++k) s[k] += rpb[hidx[k] + bo];
        l[ki * 3 + kj] = s;
#pragma unroll
        for (int k = 0; k < 4; ++k) mx[k] = fmaxf(mx[k], s[k]);
      }
    f32x4 den = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int n = 0; n < 9; ++n) {
#pragma unroll
      for (int k = 0; k < 4; ++k) l[n][k] = __expf(l[n][k] - mx[k]);
      den += l[n];
    }
    f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ki = 0; ki < 3; ++ki)
#pragma unroll
      for (int kj = 0; kj < 3; ++kj) {
        const f32x4 vv = ld4(base + ((int64_t)(sy + ki) * g.W + sx + kj) * 3 * g.C + 2 * g.C + c);
        o += l[ki * 3 + kj] * vv;
      }
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = o[k] / den[k];
    if (ok) *reinterpret_cast<f32x4*>(out + pix * g.C + c) = o;
  }
}

// Backward.  One block = an 8x8 query tile x a channel chunk of one image.  The inverse neighbourhood
// (which queries read key j) is irregular under the clamped-window rule, so dK/dV are SCATTERED from the
// query side -- but into an LDS image of the tile + 1-pixel halo (ds_add_f32), and only that image is
// flushed to HBM with global atomics (non-zero entries only): ~(10x10)/(8x8) * 2C adds per pixel instead of 18C.  dQ is owned by
// its query (plain store), d rpb is reduced per block in LDS.
constexpr int NB_T = 8, NB_HALO = 2, NB_R = NB_T + 2 * NB_HALO;  // halo 2: a clamped window at the image edge reaches 2 rows back

template <int HD>
__global__ __launch_bounds__(256) void na_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ rpb,
                                                     const float* __restrict__ dout, float* __restrict__ dqkv,
                                                     float* __restrict__ drpb, const NaGeom g, int cch, int tiles_x,
                                                     int tiles_y) {
  extern __shared__ float smem[];  // acc[NB_R*NB_R][2][cch] | s_drpb[heads*25]
  float* acc = smem;
  float* s_drpb = smem + NB_R * NB_R * 2 * cch;
  const int nacc = NB_R * NB_R * 2 * cch;
  for (int i = threadIdx.x; i < nacc + g.heads * 25; i += 256) smem[i] = 0.f;
  __syncthreads();
  const int tile = blockIdx.x % (tiles_x * tiles_y), b = blockIdx.x / (tiles_x * tiles_y);
  const int ty0 = (tile / tiles_x) * NB_T, tx0 = (tile % tiles_x) * NB_T;
  const int ch0 = blockIdx.y * cch;
  const int cch4 = cch >> 2;
  const int64_t ib = (int64_t)b * g.H * g.W * 3 * g.C;
  const float* base = qkv + ib;
  float* dbase = dqkv + ib;
  const int items = NB_T * NB_T * cch4;
  const int nit = (items + 255) / 256;
  for (int it = 0; it < nit; ++it) {
    int idx = it * 256 + threadIdx.x;
    bool ok = idx < items;
    if (!ok) idx = items - 1;  // keep every lane in the shuffles
    const int c = ch0 + (idx % cch4) * 4;
    const int pl = idx / cch4;
    int y = ty0 + pl / NB_T, x = tx0 + pl % NB_T;
    if (y >= g.H || x >= g.W || c >= g.C) ok = false;
    y = y < g.H ? y : g.H - 1;
    x = x < g.W ? x : g.W - 1;
    const int cc = c < g.C ? c : g.C - 4;
    const int sy = wstart(y, g.H), sx = wstart(x, g.W);
    const int64_t pix = ((int64_t)b * g.H + y) * g.W + x;
    const f32x4 q = ld4(base + ((int64_t)y * g.W + x) * 3 * g.C + cc) * g.scale;
    const f32x4 dO = ld4(dout + pix * g.C + cc);
    int hidx[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) hidx[k] = ((cc + k) / HD) * 25;
    f32x4 p[9], dp[9];
    f32x4 mx = f32x4{-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
#pragma unroll
    for (int ki = 0; ki < 3; ++ki)
#pragma unroll
      for (int kj = 0; kj < 3; ++kj) {
        const int ny = sy + ki, nx = sx + kj;
        const int64_t po = ((int64_t)ny * g.W + nx) * 3 * g.C;
        const f32x4 kk = ld4(base + po + g.C + cc), vv = ld4(base + po + 2 * g.C + cc);
        f32x4 s = head_sum<HD>(q * kk);
        const int bo = (ny - y + 2) * 5 + (nx - x + 2);
#pragma unroll
        for (int k = 0; k < 4; ++k) s[k] += rpb[hidx[k] + bo];
        p[ki * 3 + kj] = s;
        dp[ki * 3 + kj] = head_sum<HD>(dO * vv);
#pragma unroll
        for (int k = 0; k < 4; ++k) mx[k] = fmaxf(mx[k], s[k]);
      }
    f32x4 den = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int n = 0; n < 9; ++n) {
#pragma unroll
      for (int k = 0; k < 4; ++k) p[n][k] = __expf(p[n][k] - mx[k]);
      den += p[n];
    }
    f32x4 dsum = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int n = 0; n < 9; ++n) {
#pragma unroll
      for (int k = 0; k < 4; ++k) p[n][k] = p[n][k] / den[k];
      dsum += p[n] * dp[n];
    }
    f32x4 dq = f32x4{0.f, 0.f, 0.f, 0.f};
    bool rep[4];  // the lane owning a head's FIRST channel reports that head's rpb gradient
#pragma unroll
    for (int k = 0; k < 4; ++k) rep[k] = ok && ((cc + k) % HD == 0);
#pragma unroll
    for (int ki = 0; ki < 3; ++ki)
#pragma unroll
      for (int kj = 0; kj < 3; ++kj) {
        const int n = ki * 3 + kj;
        const int ny = sy + ki, nx = sx + kj;
        const f32x4 ds = p[n] * (dp[n] - dsum);
        const f32x4 kk = ld4(base + ((int64_t)ny * g.W + nx) * 3 * g.C + g.C + cc);
        dq += ds * kk;
        if (ok) {
          const f32x4 dk = ds * q;  // q already carries the scale
          const f32x4 dv = p[n] * dO;
          float* a = acc + (((ny - ty0 + NB_HALO) * NB_R + (nx - tx0 + NB_HALO)) * 2) * cch + (cc - ch0);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            atomicAdd(a + k, dk[k]);
            atomicAdd(a + cch + k, dv[k]);
          }
          const int bo = (ny - y + 2) * 5 + (nx - x + 2);
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (rep[k]) atomicAdd(&s_drpb[hidx[k] + bo], ds[k]);
        }
      }
    if (ok) *reinterpret_cast<f32x4*>(dbase + ((int64_t)y * g.W + x) * 3 * g.C + cc) = dq * g.scale;
  }
  __syncthreads();
  // flush the tile+halo image of dK, dV
  for (int i = threadIdx.x; i < nacc; i += 256) {
    const int cl = i % cch, w = (i / cch) & 1, pp = i / (2 * cch);
    const int gy = ty0 - NB_HALO + pp / NB_R, gx = tx0 - NB_HALO + pp % NB_R;
    const float v = acc[i];
    if (gy >= 0 && gy < g.H && gx >= 0 && gx < g.W && ch0 + cl < g.C && v != 0.f)
      atomicAdd(dbase + ((int64_t)gy * g.W + gx) * 3 * g.C + (1 + w) * g.C + ch0 + cl, v);
  }
  for (int i = threadIdx.x; i < g.heads * 25; i += 256)
    if (s_drpb[i] != 0.f) atomicAdd(drpb + i, s_drpb[i]);
}

inline int na_grid(int64_t total) {
  int64_t gsz = (total + 255) / 256;
  if (gsz > 8192) gsz = 8192;
  return (int)(gsz < 1 ? 1 : gsz);
}

}  // namespace

extern "C" {

int lmn_na_fwd(const float* qkv, const float* rpb, float* out, int B, int H, int W, int heads, int hd, float scale,
               lmn_stream_t stream) {
  LMN_REQUIRE(qkv && rpb && out, "na_fwd: null pointer");
  LMN_REQUIRE(B > 0 && H >= 3 && W >= 3, "na_fwd: feature map %dx%d smaller than the 3x3 window", H, W);
  LMN_REQUIRE(hd == 1 || hd == 2 || hd == 4 || hd == 8 || hd == 16, "na_fwd: head_dim %d not in {1,2,4,8,16}", hd);
  LMN_REQUIRE((heads * hd) % 4 == 0, "na_fwd: C=%d must be a multiple of 4", heads * hd);
  NaGeom g{B, H, W, heads * hd, heads * hd / 4, heads, scale};
  LMN_REQUIRE(hd < 8 || (g.C4 % (hd / 4)) == 0, "na_fwd: geometry");
  const int grid = na_grid((int64_t)B * H * W * g.C4);
  hipStream_t st = (hipStream_t)stream;
  switch (hd) {
    case 1: hipLaunchKernelGGL((na_fwd_kernel<1>), dim3(grid), dim3(256), 0, st, qkv, rpb, out, g); break;
    case 2: hipLaunchKernelGGL((na_fwd_kernel<2>), dim3(grid), dim3(256), 0, st, qkv, rpb, out, g); break;
    case 4: hipLaunchKernelGGL((na_fwd_kernel<4>), dim3(grid), dim3(256), 0, st, qkv, rpb, out, g); break;
    case 8: hipLaunchKernelGGL((na_fwd_kernel<8>), dim3(grid), dim3(256), 0, st, qkv, rpb, out, g); break;
    default: hipLaunchKernelGGL((na_fwd_kernel<16>), dim3(grid), dim3(256), 0, st, qkv, rpb, out, g); break;
  }
  return lmn_launch_status("na_fwd");
}

int lmn_na_bwd(const float* qkv, const float* rpb, const float* dout, float* dqkv, float* drpb, int B, int H, int W,
               int heads, int hd, float scale, lmn_stream_t stream) {
  LMN_REQUIRE(qkv && rpb && dout && dqkv && drpb, "na_bwd: null pointer");
  LMN_REQUIRE(B > 0 && H >= 3 && W >= 3, "na_bwd: feature map %dx%d smaller than the 3x3 window", H, W);
  LMN_REQUIRE(hd == 1 || hd == 2 || hd == 4 || hd == 8 || hd == 16, "na_bwd: head_dim %d not in {1,2,4,8,16}", hd);
  LMN_REQUIRE((heads * hd) % 4 == 0 && heads * 25 * sizeof(float) <= 48000, "na_bwd: heads=%d hd=%d", heads, hd);
  NaGeom g{B, H, W, heads * hd, heads * hd / 4, heads, scale};
  // channel chunk: whole heads, a multiple of 8 channels (lane pairs for hd = 8), at most 24 (48 for hd = 16)
  int cch = g.C <= 24 ? g.C : (hd == 16 ? 48 : 24);
  while (g.C % cch) cch += (hd >= 4 ? hd : 4);
  LMN_REQUIRE(cch % 4 == 0 && (cch % hd == 0 || hd % cch == 0), "na_bwd: channel chunk %d for hd %d", cch, hd);
  const int tx = lmn_cdiv(W, NB_T), ty = lmn_cdiv(H, NB_T);
  const dim3 grid(B * tx * ty, g.C / cch);
  hipStream_t st = (hipStream_t)stream;
  const size_t sh = (size_t)(NB_R * NB_R * 2 * cch + heads * 25) * sizeof(float);
  switch (hd) {
    case 1: hipLaunchKernelGGL((na_bwd_kernel<1>), grid, dim3(256), sh, st, qkv, rpb, dout, dqkv, drpb, g, cch, tx, ty); break;
    case 2: hipLaunchKernelGGL((na_bwd_kernel<2>), grid, dim3(256), sh, st, qkv, rpb, dout, dqkv, drpb, g, cch, tx, ty); break;
    case 4: hipLaunchKernelGGL((na_bwd_kernel<4>), grid, dim3(256), sh, st, qkv, rpb, dout, dqkv, drpb, g, cch, tx, ty); break;
    case 8: hipLaunchKernelGGL((na_bwd_kernel<8>), grid, dim3(256), sh, st, qkv, rpb, dout, dqkv, drpb, g, cch, tx, ty); break;
    default: hipLaunchKernelGGL((na_bwd_kernel<16>), grid, dim3(256), sh, st, qkv, rpb, dout, dqkv, drpb, g, cch, tx, ty); break;
  }
  return lmn_launch_status("na_bwd");
}

}  // extern "C"
