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

// Bias entries of a channel quad's heads for window offset `bo` from the TRANSPOSED table tab[25][heads] (LDS): the heads of a quad are
// adjacent, so the quad reads ONE 16- / 8- / 4-byte word instead of four gathers at a 25-word head stride (hd = 1: 36 -> 9 LDS reads
// per query; all pixels of a wave share `bo` in the interior, so the reads are broadcasts).  heads * hd % 4 == 0 keeps them aligned.
template <int HD>
__device__ __forceinline__ f32x4 rpb4(const float* tab, int heads, int bo, int c) {
  const float* p = tab + bo * heads + c / HD;
  if constexpr (HD == 1) return *reinterpret_cast<const f32x4*>(p);
  else if constexpr (HD == 2) {
    const float2 a = *reinterpret_cast<const float2*>(p);
    return f32x4{a.x, a.x, a.y, a.y};
  } else {
    const float a = *p;
    return f32x4{a, a, a, a};
  }
}
// per-(pixel, head) scalars of a channel quad (lse / dsum): one vector store per quad at hd <= 2, the head's first lane otherwise
template <int HD>
__device__ __forceinline__ void stat_store(float* dst, int c, f32x4 v, bool ok) {
  if (!ok) return;
  if constexpr (HD == 1) *reinterpret_cast<f32x4*>(dst + c) = v;
  else if constexpr (HD == 2) *reinterpret_cast<float2*>(dst + c / 2) = float2{v[0], v[2]};
  else if (c % HD == 0) dst[c / HD] = v[0];
}

struct NaGeom {
  int B, H, W, C, C4, heads;
  float scale;
  uint32_t mW, mHW, mC4;   // floor(2^32 / d) for d = W, H*W, C4 (lmn_div_row): the index decodes below were 64-bit software divisions,
};                         // three to five per item -- several hundred instructions next to ~200 of arithmetic

// flat item index -> channel quad, image, row, column (all < 2^31: checked on the host)
struct NaPix { int c, b, y, x; };
__device__ __forceinline__ NaPix na_decode_pix(const NaGeom& g, uint32_t pix) {
  const uint32_t b = lmn_div_row(pix, (uint32_t)(g.H * g.W), g.mHW), r = pix - b * (uint32_t)(g.H * g.W);
  const uint32_t y = lmn_div_row(r, (uint32_t)g.W, g.mW);
  return NaPix{0, (int)b, (int)y, (int)(r - y * (uint32_t)g.W)};
}
__device__ __forceinline__ NaPix na_decode_item(const NaGeom& g, uint32_t idx) {
  const uint32_t pix = lmn_div_row(idx, (uint32_t)g.C4, g.mC4);
  NaPix p = na_decode_pix(g, pix);
  p.c = (int)(idx - pix * (uint32_t)g.C4) * 4;
  return p;
}

// window start (clamped, never padded) -- oracle/natten_ref.py: window_start
__device__ __forceinline__ int wstart(int i, int L) {
  int s = i - 1;
  s = s < 0 ? 0 : s;
  return s > L - 3 ? L - 3 : s;
}

// XCD-aware block order (guide T1, bijective): consecutive logical tiles share an XCD and therefore an L2.
__device__ __forceinline__ int xcd_swizzle(int bid, int nwg) {
  const int qd = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  return (xcd < r ? xcd * (qd + 1) : r * (qd + 1) + (xcd - r) * qd) + (bid >> 3);
}

// Forward.  One block = a T x T query tile x a channel chunk of one image.  The k and v rows the tile's clamped
// 3x3 windows can touch ((T+2)^2 pixels; the origin follows the clamping at the image border) are staged ONCE in
// LDS by coalesced float4 loads, so the 9x neighbour re-use never leaves the CU: measured HBM fetch of the
// direct-from-L1 form was 4.4x the algorithmic bytes (rocprof FETCH_SIZE), this form reads (T+2)^2/T^2.
template <int HD, typename TA>
__global__ __launch_bounds__(256) void na_fwd_kernel(const TA* __restrict__ qkv, const float* __restrict__ rpb,
                                                     TA* __restrict__ out, const NaGeom g, int T, int cch,
                                                     int tiles_x, int tiles, int chunks, uint32_t mC4c, uint32_t mR, uint32_t mT) {
  // (mC4c, mR, mT: floor(2^32 / d) for the per-item divisions by cch / 4, T + 2 and T -- lmn_div_row)
  extern __shared__ __attribute__((aligned(16))) float KV[];  // [(T+2)^2][2][cch]
  __shared__ __attribute__((aligned(16))) float s_rpb[16 * 25];  // the bias table, transposed to [25][heads] (rpb4)
  for (int i = threadIdx.x; i < g.heads * 25; i += 256) s_rpb[(i % 25) * g.heads + i / 25] = rpb[i];
  const int R = T + 2;
  const int lid = xcd_swizzle(blockIdx.x, gridDim.x);
  const int t = lid % tiles, cb = lid / tiles;
  const int ty0 = (t / tiles_x) * T, tx0 = (t % tiles_x) * T;
  const int ch0 = (cb % chunks) * cch, b = cb / chunks;
  const int rlo = max(0, min(ty0 - 1, g.H - 3)), clo = max(0, min(tx0 - 1, g.W - 3));
  const int cch4 = cch >> 2;
  const TA* base = qkv + (int64_t)b * g.H * g.W * 3 * g.C;
  for (int i = threadIdx.x; i < R * R * 2 * cch4; i += 256) {
    const int iw = (int)lmn_div_row((uint32_t)i, (uint32_t)cch4, mC4c);
    const int c4 = i - iw * cch4, w = iw & 1, pix = iw >> 1;
    const int pr = (int)lmn_div_row((uint32_t)pix, (uint32_t)R, mR);
    const int gy = rlo + pr, gx = clo + pix - pr * R;
    const bool in = gy < g.H && gx < g.W && ch0 + c4 * 4 < g.C;
    const int sy = in ? gy : 0, sx = in ? gx : 0, sc = in ? ch0 + c4 * 4 : 0;
    f32x4 v = ld4(base + ((int64_t)sy * g.W + sx) * 3 * g.C + (1 + w) * g.C + sc);
    if (!in) v = f32x4{0.f, 0.f, 0.f, 0.f};
    *reinterpret_cast<f32x4*>(&KV[(pix * 2 + w) * cch + c4 * 4]) = v;
  }
  __syncthreads();
  const int items = T * T * cch4;
  const int nit = (items + 255) / 256;
  for (int it = 0; it < nit; ++it) {
    int idx = it * 256 + threadIdx.x;
    bool ok = idx < items;
    if (!ok) idx = items - 1;  // keep every lane in the shuffles
    const int pl = (int)lmn_div_row((uint32_t)idx, (uint32_t)cch4, mC4c), cl = (idx - pl * cch4) * 4;
    const int plr = (int)lmn_div_row((uint32_t)pl, (uint32_t)T, mT);
    int y = ty0 + plr, x = tx0 + pl - plr * T;
    const int c = ch0 + cl;
    if (y >= g.H || x >= g.W || c >= g.C) ok = false;
    y = y < g.H ? y : g.H - 1;
    x = x < g.W ? x : g.W - 1;
    const int cc = c < g.C ? c : g.C - 4, ccl = cc - ch0;
    const int sy = wstart(y, g.H), sx = wstart(x, g.W);
    const f32x4 q = ld4(base + ((int64_t)y * g.W + x) * 3 * g.C + cc) * g.scale;
    const float* kv0 = KV + (((sy - rlo) * R + (sx - clo)) * 2) * cch + ccl;
    f32x4 l[9];
    f32x4 mx = f32x4{-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
#pragma unroll
    for (int ki = 0; ki < 3; ++ki)
#pragma unroll
      for (int kj = 0; kj < 3; ++kj) {
        const f32x4 kk = *reinterpret_cast<const f32x4*>(kv0 + ((ki * R + kj) * 2) * cch);
        f32x4 sc = head_sum<HD>(q * kk);
        const int bo = (sy + ki - y + 2) * 5 + (sx + kj - x + 2);
        sc += rpb4<HD>(s_rpb, g.heads, bo, cc);
        l[ki * 3 + kj] = sc;
#pragma unroll
        for (int k = 0; k < 4; ++k) mx[k] = fmaxf(mx[k], sc[k]);
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
      for (int kj = 0; kj < 3; ++kj) o += l[ki * 3 + kj] * *reinterpret_cast<const f32x4*>(kv0 + ((ki * R + kj) * 2 + 1) * cch);
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = o[k] * __builtin_amdgcn_rcpf(den[k]);  // v_rcp_f32: 1 ulp
    if (ok) st4(out + (((int64_t)b * g.H + y) * g.W + x) * g.C + cc, o);
  }
}

// Backward in two gather passes -- no atomics on dQ/dK/dV (LDS float atomics cost ~3 cycles per lane on gfx950 and
// global ones serialise on the 9x fan-in), deterministic:
//   pass 1 (query-owned): recompute softmax p and dp_n = dO.v_n; dsum = sum_n p_n dp_n; dQ = scale * sum_n ds_n k_n;
//           store per (pixel, head) the two scalars the key side needs: lse = max + log(sum exp) and dsum; d rpb is
//           reduced per block in LDS and flushed with one atomic per entry per block.
//   pass 2 (key-owned): key j gathers from every query i whose clamped window contains j (i within +-2 rows/cols):
//           p_ij = exp(q_i.k_j + rpb - lse_i), ds_ij = p_ij (dO_i.v_j - dsum_i); dK_j += scale ds_ij q_i; dV_j += p_ij dO_i.
// Workspace: 2 floats per (pixel, head).
template <int HD, typename TA>
__global__ __launch_bounds__(256) void na_bwd_q_kernel(const TA* __restrict__ qkv, const float* __restrict__ rpb,
                                                       const TA* __restrict__ dout, TA* __restrict__ dqkv,
                                                       float* __restrict__ drpb, float* __restrict__ stat,
                                                       const NaGeom g, int det) {
  extern __shared__ __attribute__((aligned(16))) float s_drpb[];  // [heads][25] | per-thread interior bins [256][36] | rpb copy | border bins per wave [4][heads][25]
  float* s_bins = s_drpb + g.heads * 25;
  float* s_rpb = s_bins + 256 * 36;  // [heads][25] copy of the bias table
  // border pixels add with LDS atomics into the table of THEIR wave (program order within a wave; the four tables and the interior
  // sums are added in a fixed order at the end: the block's result does not depend on how its waves interleave)
  float* s_wave = s_rpb + g.heads * 25 + (threadIdx.x >> 6) * g.heads * 25;
  for (int i = threadIdx.x; i < g.heads * 25; i += 256) { s_drpb[i] = 0.f; s_rpb[(i % 25) * g.heads + i / 25] = rpb[i]; }   // (s_rpb transposed: rpb4)
  for (int i = threadIdx.x; i < 4 * g.heads * 25; i += 256) s_rpb[g.heads * 25 + i] = 0.f;
  __syncthreads();
  // A thread keeps ONE channel quad for the whole kernel (quad = tid % C4, pixel slot = tid / C4), so the rpb
  // gradient of interior pixels -- whose 9 neighbours always hit the same 9 bins -- accumulates in 36 registers.
  // (Per-pixel LDS atomics on 9 shared bins serialise the whole block: 554 us at 352x352, 12 heads.)
  const int PB = 256 / g.C4;                       // pixels per block iteration
  const int c = (threadIdx.x % g.C4) * 4;
  const int slot = threadIdx.x / g.C4;
  const int npix = g.B * g.H * g.W;
  const int nit = (npix + (int)gridDim.x * PB - 1) / ((int)gridDim.x * PB);
  f32x4 bins[9];
#pragma unroll
  for (int n = 0; n < 9; ++n) bins[n] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < nit; ++it) {
    int pix = (it * (int)gridDim.x + (int)blockIdx.x) * PB + slot;
    const bool ok = slot < PB && pix < npix;
    if (!ok) pix = npix - 1;  // keep every lane in the shuffles
    const NaPix pd = na_decode_pix(g, (uint32_t)pix);
    const int x = pd.x, y = pd.y, b = pd.b;
    const int sy = wstart(y, g.H), sx = wstart(x, g.W);
    const int64_t ib = (int64_t)b * g.H * g.W * 3 * g.C;
    const TA* base = qkv + ib;
    const f32x4 q = ld4(base + ((int64_t)y * g.W + x) * 3 * g.C + c) * g.scale;
    const f32x4 dO = ld4(dout + pix * g.C + c);
    int hidx[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) hidx[k] = ((c + k) / HD) * 25;
    f32x4 p[9], dp[9];
    f32x4 mx = f32x4{-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
#pragma unroll
    for (int ki = 0; ki < 3; ++ki)
#pragma unroll
      for (int kj = 0; kj < 3; ++kj) {
        const int ny = sy + ki, nx = sx + kj;
        const int64_t po = ((int64_t)ny * g.W + nx) * 3 * g.C;
        const f32x4 kk = ld4(base + po + g.C + c), vv = ld4(base + po + 2 * g.C + c);
        f32x4 s = head_sum<HD>(q * kk);
        const int bo = (ny - y + 2) * 5 + (nx - x + 2);
        s += rpb4<HD>(s_rpb, g.heads, bo, c);
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
    f32x4 rden;
#pragma unroll
    for (int k = 0; k < 4; ++k) rden[k] = __builtin_amdgcn_rcpf(den[k]);  // v_rcp_f32 (1 ulp) instead of 36 IEEE divisions
#pragma unroll
    for (int n = 0; n < 9; ++n) {
#pragma unroll
      for (int k = 0; k < 4; ++k) p[n][k] = p[n][k] * rden[k];
      dsum += p[n] * dp[n];
    }
    f32x4 dq = f32x4{0.f, 0.f, 0.f, 0.f};
    bool rep[4];  // the lane owning a head's FIRST channel reports for that head
#pragma unroll
    for (int k = 0; k < 4; ++k) rep[k] = ok && ((c + k) % HD == 0);
    const bool inter = sy == y - 1 && sx == x - 1;  // unclamped window: neighbour n always lands in bin (ki+1, kj+1)
    f32x4 rm;
#pragma unroll
    for (int k = 0; k < 4; ++k) rm[k] = (rep[k] && inter) ? 1.f : 0.f;
#pragma unroll
    for (int ki = 0; ki < 3; ++ki)
#pragma unroll
      for (int kj = 0; kj < 3; ++kj) {
        const int n = ki * 3 + kj;
        const int ny = sy + ki, nx = sx + kj;
        const f32x4 ds = p[n] * (dp[n] - dsum);
        const f32x4 kk = ld4(base + ((int64_t)ny * g.W + nx) * 3 * g.C + g.C + c);
        dq += ds * kk;
        bins[n] += ds * rm;
        if (!inter) {  // border pixels (the clamped window shifts the bins): rare, through LDS atomics
          const int bo = (ny - y + 2) * 5 + (nx - x + 2);
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (rep[k]) atomicAdd(&s_wave[hidx[k] + bo], ds[k]);
        }
      }
    if (ok) st4(dqkv + ib + ((int64_t)y * g.W + x) * 3 * g.C + c, dq * g.scale);
    {
      f32x4 lse;
#pragma unroll
      for (int k = 0; k < 4; ++k) lse[k] = mx[k] + __logf(den[k]);
      stat_store<HD>(stat + pix * 2 * g.heads, c, lse, ok);
      stat_store<HD>(stat + pix * 2 * g.heads + g.heads, c, dsum, ok);
    }
  }
  // interior bins: park per thread, then one thread per (quad, neighbour, component) sums its PB pixel slots
#pragma unroll
  for (int n = 0; n < 9; ++n)
#pragma unroll
    for (int k = 0; k < 4; ++k) s_bins[threadIdx.x * 36 + n * 4 + k] = bins[n][k];
  __syncthreads();
  for (int o = threadIdx.x; o < g.C4 * 36; o += 256) {
    const int qd = o / 36, nk = o - qd * 36;
    const int n = nk >> 2, k = nk & 3;
    const int ch = qd * 4 + k;
    if (ch % HD != 0) continue;
    float v = 0.f;
    for (int sl = 0; sl < PB; ++sl) v += s_bins[(sl * g.C4 + qd) * 36 + nk];
    s_drpb[(ch / HD) * 25 + (n / 3 + 1) * 5 + (n % 3 + 1)] = v;   // (one writer per (head, bin))
  }
  __syncthreads();
  for (int i = threadIdx.x; i < g.heads * 25; i += 256) {
    const float* wt = s_rpb + g.heads * 25;
    const float v = (((s_drpb[i] + wt[i]) + wt[g.heads * 25 + i]) + wt[2 * g.heads * 25 + i]) + wt[3 * g.heads * 25 + i];
    if (v != 0.f) lmn_red_add(drpb + (det ? (int64_t)blockIdx.x * g.heads * 25 : 0) + i, v, det);   // (deterministic mode: slot copies [blocks][heads][25])
  }
}

template <int HD, typename TA>
__global__ __launch_bounds__(256) void na_bwd_q_tile_kernel(const TA* __restrict__ qkv, const float* __restrict__ rpb,
                                                            const TA* __restrict__ dout, TA* __restrict__ dqkv,
                                                            float* __restrict__ drpb, float* __restrict__ stat,
                                                            const NaGeom g, int TH, int TW, int tiles_x, int tiles_img,
                                                            int total_tiles, int det) {
  // Query pass for C <= 24 (levels 0-1) with the k/v window of a TH x TW query tile staged in LDS, as the forward
  // does: the direct form is bound by the L1 address path (PMC: TA busy 78 % of the kernel, 49 cache accesses per wave
  // load -- 21 pixels x 48 B at a 144 B stride).  Persistent blocks walk tiles; a thread keeps its channel quad, so
  // the interior rpb bins stay in registers across tiles and are flushed once (the parking area reuses the window).
  extern __shared__ __attribute__((aligned(16))) float s_drpb[];  // [heads][25] | rpb copy [heads][25] (padded to 16 B) | KV window / bins
  const int ntab = (g.heads * 25 + 3) & ~3;
  float* s_rpb = s_drpb + ntab;
  float* s_wt = s_rpb + ntab;                            // border bins per wave [4][ntab] (see na_bwd_q_kernel)
  float* s_wave = s_wt + (threadIdx.x >> 6) * ntab;
  float* KV = s_wt + 4 * ntab;   // [(TH+2)*(TW+2)][2][C]
  float* s_bins = KV;         // [256][36] after the last tile
  const int RW = TW + 2, RH = TH + 2;
  for (int i = threadIdx.x; i < g.heads * 25; i += 256) { s_drpb[i] = 0.f; s_rpb[(i % 25) * g.heads + i / 25] = rpb[i]; }   // (s_rpb transposed: rpb4)
  for (int i = threadIdx.x; i < 4 * ntab; i += 256) s_wt[i] = 0.f;
  __syncthreads();
  // A thread keeps ONE channel quad for the whole kernel (quad = tid % C4, pixel slot = tid / C4), so the rpb
  // gradient of interior pixels -- whose 9 neighbours always hit the same 9 bins -- accumulates in 36 registers.
  // (Per-pixel LDS atomics on 9 shared bins serialise the whole block: 554 us at 352x352, 12 heads.)
  const int PB = 256 / g.C4;                       // pixels per block iteration
  const int c = (threadIdx.x % g.C4) * 4;
  const int slot = threadIdx.x / g.C4;
  const int npx = TH * TW;
  f32x4 bins[9];
#pragma unroll
  for (int n = 0; n < 9; ++n) bins[n] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
   const int b = tile / tiles_img, tt = tile - b * tiles_img;
   const int ty0 = (tt / tiles_x) * TH, tx0 = (tt % tiles_x) * TW;
   const int rlo = max(0, min(ty0 - 1, g.H - 3)), clo = max(0, min(tx0 - 1, g.W - 3));
   const int64_t ib = (int64_t)b * g.H * g.W * 3 * g.C;
   const TA* base = qkv + ib;
   __syncthreads();  // the previous tile's window is consumed
   for (int i0 = 0; i0 < RH * RW * 2 * g.C4; i0 += 4 * 256) {   // rounds of 4 loads in flight per thread: the plain loop compiled to one exposed round trip per item (level 0: backward 330 -> 299 us; the forward loses more to the 16 extra VGPRs than it gains)
     f32x4 sv[4];
#pragma unroll
     for (int u = 0; u < 4; ++u) {
       const int i = i0 + u * 256 + threadIdx.x;
       const int c4 = i % g.C4, w = (i / g.C4) & 1, wp = i / (2 * g.C4);
       const int gy = rlo + wp / RW, gx = clo + wp % RW;
       const bool in = i < RH * RW * 2 * g.C4 && gy < g.H && gx < g.W;
       sv[u] = ld4(base + ((int64_t)(in ? gy : 0) * g.W + (in ? gx : 0)) * 3 * g.C + (1 + (in ? w : 0)) * g.C + (in ? c4 * 4 : 0));
       if (!in) sv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
     }
#pragma unroll
     for (int u = 0; u < 4; ++u) {
       const int i = i0 + u * 256 + threadIdx.x;
       if (i < RH * RW * 2 * g.C4) {
         const int c4 = i % g.C4, w = (i / g.C4) & 1, wp = i / (2 * g.C4);
         *reinterpret_cast<f32x4*>(&KV[(wp * 2 + w) * g.C + c4 * 4]) = sv[u];
       }
     }
   }
   __syncthreads();
   for (int p0 = 0; p0 < npx; p0 += PB) {
    const int pl = p0 + slot;
    int y = ty0 + pl / TW, x = tx0 + pl % TW;
    const bool ok = slot < PB && pl < npx && y < g.H && x < g.W;
    if (!ok) { y = ty0; x = tx0; }  // a safe pixel of this tile; nothing of it is stored or counted
    const int64_t pix = ((int64_t)b * g.H + y) * g.W + x;
    const int sy = wstart(y, g.H), sx = wstart(x, g.W);
    const float* kv0 = KV + (((sy - rlo) * RW + (sx - clo)) * 2) * g.C + c;
    const f32x4 q = ld4(base + ((int64_t)y * g.W + x) * 3 * g.C + c) * g.scale;
    const f32x4 dO = ld4(dout + pix * g.C + c);
    int hidx[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) hidx[k] = ((c + k) / HD) * 25;
    f32x4 p[9], dp[9];
    f32x4 mx = f32x4{-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
#pragma unroll
    for (int ki = 0; ki < 3; ++ki)
#pragma unroll
      for (int kj = 0; kj < 3; ++kj) {
        const int ny = sy + ki, nx = sx + kj;
        const float* kvn = kv0 + ((ki * RW + kj) * 2) * g.C;
        const f32x4 kk = *reinterpret_cast<const f32x4*>(kvn), vv = *reinterpret_cast<const f32x4*>(kvn + g.C);
        f32x4 s = head_sum<HD>(q * kk);
        const int bo = (ny - y + 2) * 5 + (nx - x + 2);
        s += rpb4<HD>(s_rpb, g.heads, bo, c);
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
    f32x4 rden;
#pragma unroll
    for (int k = 0; k < 4; ++k) rden[k] = __builtin_amdgcn_rcpf(den[k]);  // v_rcp_f32 (1 ulp) instead of 36 IEEE divisions
#pragma unroll
    for (int n = 0; n < 9; ++n) {
#pragma unroll
      for (int k = 0; k < 4; ++k) p[n][k] = p[n][k] * rden[k];
      dsum += p[n] * dp[n];
    }
    f32x4 dq = f32x4{0.f, 0.f, 0.f, 0.f};
    bool rep[4];  // the lane owning a head's FIRST channel reports for that head
#pragma unroll
    for (int k = 0; k < 4; ++k) rep[k] = ok && ((c + k) % HD == 0);
    const bool inter = sy == y - 1 && sx == x - 1;  // unclamped window: neighbour n always lands in bin (ki+1, kj+1)
    f32x4 rm;
#pragma unroll
    for (int k = 0; k < 4; ++k) rm[k] = (rep[k] && inter) ? 1.f : 0.f;
#pragma unroll
    for (int ki = 0; ki < 3; ++ki)
#pragma unroll
      for (int kj = 0; kj < 3; ++kj) {
        const int n = ki * 3 + kj;
        const int ny = sy + ki, nx = sx + kj;
        const f32x4 ds = p[n] * (dp[n] - dsum);
        const f32x4 kk = *reinterpret_cast<const f32x4*>(kv0 + ((ki * RW + kj) * 2) * g.C);
        dq += ds * kk;
        bins[n] += ds * rm;
        if (!inter) {  // border pixels (the clamped window shifts the bins): rare, through LDS atomics
          const int bo = (ny - y + 2) * 5 + (nx - x + 2);
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (rep[k]) atomicAdd(&s_wave[hidx[k] + bo], ds[k]);
        }
      }
    if (ok) st4(dqkv + ib + ((int64_t)y * g.W + x) * 3 * g.C + c, dq * g.scale);
    {
      f32x4 lse;
#pragma unroll
      for (int k = 0; k < 4; ++k) lse[k] = mx[k] + __logf(den[k]);
      stat_store<HD>(stat + pix * 2 * g.heads, c, lse, ok);
      stat_store<HD>(stat + pix * 2 * g.heads + g.heads, c, dsum, ok);
    }
   }
  }
  __syncthreads();  // every window read is done: the parking area reuses that memory
  // interior bins: park per thread, then one thread per (quad, neighbour, component) sums its PB pixel slots
#pragma unroll
  for (int n = 0; n < 9; ++n)
#pragma unroll
    for (int k = 0; k < 4; ++k) s_bins[threadIdx.x * 36 + n * 4 + k] = bins[n][k];
  __syncthreads();
  for (int o = threadIdx.x; o < g.C4 * 36; o += 256) {
    const int qd = o / 36, nk = o - qd * 36;
    const int n = nk >> 2, k = nk & 3;
    const int ch = qd * 4 + k;
    if (ch % HD != 0) continue;
    float v = 0.f;
    for (int sl = 0; sl < PB; ++sl) v += s_bins[(sl * g.C4 + qd) * 36 + nk];
    s_drpb[(ch / HD) * 25 + (n / 3 + 1) * 5 + (n % 3 + 1)] = v;   // (one writer per (head, bin))
  }
  __syncthreads();
  for (int i = threadIdx.x; i < g.heads * 25; i += 256) {
    const float v = (((s_drpb[i] + s_wt[i]) + s_wt[ntab + i]) + s_wt[2 * ntab + i]) + s_wt[3 * ntab + i];
    if (v != 0.f) lmn_red_add(drpb + (det ? (int64_t)blockIdx.x * g.heads * 25 : 0) + i, v, det);
  }
}

template <int HD, typename TA>
__global__ __launch_bounds__(256) void na_bwd_kv_kernel(const TA* __restrict__ qkv, const float* __restrict__ rpb,
                                                        const TA* __restrict__ dout, TA* __restrict__ dqkv,
                                                        const float* __restrict__ stat, const NaGeom g) {
  // (the bias table in LDS, transposed to [25][heads]: the quad's heads are one vector read (rpb4); in the [heads][25] order the LDS
  //  copy was slower than the L1-resident table -- a 25-word head stride = bank conflicts)
  __shared__ __attribute__((aligned(16))) float s_rpb[16 * 25];
  for (int i = threadIdx.x; i < g.heads * 25; i += 256) s_rpb[(i % 25) * g.heads + i / 25] = rpb[i];
  __syncthreads();
  const int total = g.B * g.H * g.W * g.C4;
  const int nit = (total + (int)gridDim.x * 256 - 1) / ((int)gridDim.x * 256);
  for (int it = 0; it < nit; ++it) {
    int idx = (it * (int)gridDim.x + (int)blockIdx.x) * 256 + (int)threadIdx.x;
    const bool ok = idx < total;
    if (!ok) idx = total - 1;
    const NaPix pd = na_decode_item(g, (uint32_t)idx);
    const int c = pd.c;
    const int jx = pd.x, jy = pd.y, b = pd.b;
    const int64_t ib = (int64_t)b * g.H * g.W * 3 * g.C;
    const TA* base = qkv + ib;
    const int64_t kpo = ((int64_t)jy * g.W + jx) * 3 * g.C;
    const f32x4 kj = ld4(base + kpo + g.C + c), vj = ld4(base + kpo + 2 * g.C + c);
    int hidx[4], hd_[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { hd_[k] = (c + k) / HD; hidx[k] = hd_[k] * 25; }
    f32x4 dk = f32x4{0.f, 0.f, 0.f, 0.f}, dv = dk;
    auto candidate = [&](int iy, int ix) {
      const int64_t ipix = ((int64_t)b * g.H + iy) * g.W + ix;
      const f32x4 qi = ld4(base + ((int64_t)iy * g.W + ix) * 3 * g.C + c) * g.scale;
      const f32x4 dOi = ld4(dout + ipix * g.C + c);
      f32x4 lse4, dsm4;
      const float* sp = stat + ipix * 2 * g.heads;
      if constexpr (HD == 1) {
        lse4 = ld4(sp + hd_[0]);
        dsm4 = ld4(sp + g.heads + hd_[0]);
      } else if constexpr (HD == 2) {
        const float2 a = *reinterpret_cast<const float2*>(sp + hd_[0]), bq = *reinterpret_cast<const float2*>(sp + g.heads + hd_[0]);
        lse4 = f32x4{a.x, a.x, a.y, a.y};
        dsm4 = f32x4{bq.x, bq.x, bq.y, bq.y};
      } else {
        const float a = sp[hd_[0]], bq = sp[g.heads + hd_[0]];
        lse4 = f32x4{a, a, a, a};
        dsm4 = f32x4{bq, bq, bq, bq};
      }
      const f32x4 s = head_sum<HD>(qi * kj);
      const f32x4 dp = head_sum<HD>(dOi * vj);
      const int bo = (jy - iy + 2) * 5 + (jx - ix + 2);
      const f32x4 rb = rpb4<HD>(s_rpb, g.heads, bo, c);
      f32x4 pij, ds;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        pij[k] = __expf(s[k] + rb[k] - lse4[k]);
        ds[k] = pij[k] * (dp[k] - dsm4[k]);
      }
      dk += ds * qi;   // qi carries the scale
      dv += pij * dOi;
    };
    // Interior keys (99 % of them): the queries that see key j are exactly its 3x3 neighbourhood, all with unclamped
    // windows -- a fixed, fully unrolled candidate set whose 36 loads can be issued together.  The generic loop with
    // its per-candidate window tests (16 of 25 candidates fail them) is kept for keys within 3 pixels of the border.
    // (The test is pixel-uniform, so the lane pairs of hd >= 8 stay converged in head_sum.)
    // (queries 2 pixels away see key j only through a CLAMPED window, i.e. when they sit on the border: j in [3, L-4])
    if (jy >= 3 && jy <= g.H - 4 && jx >= 3 && jx <= g.W - 4) {
#pragma unroll
      for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) candidate(jy + dy, jx + dx);
    } else {
      for (int iy = jy - 2; iy <= jy + 2; ++iy) {
        if (iy < 0 || iy >= g.H) continue;
        const int ki = jy - wstart(iy, g.H);
        if (ki < 0 || ki > 2) continue;
        for (int ix = jx - 2; ix <= jx + 2; ++ix) {
          if (ix < 0 || ix >= g.W) continue;
          const int kx = jx - wstart(ix, g.W);
          if (kx < 0 || kx > 2) continue;
          candidate(iy, ix);
        }
      }
    }
    if (ok) {
      st4(dqkv + ib + kpo + g.C + c, dk);
      st4(dqkv + ib + kpo + 2 * g.C + c, dv);
    }
  }
}

inline int na_grid(int64_t total) {
  int64_t gsz = (total + 255) / 256;
  if (gsz > 8192) gsz = 8192;
  return (int)(gsz < 1 ? 1 : gsz);
}

template <int HD, typename TA>
__global__ __launch_bounds__(256) void na_bwd_kv_tile_kernel(const TA* __restrict__ qkv, const float* __restrict__ rpb,
                                                             const TA* __restrict__ dout, TA* __restrict__ dqkv,
                                                             const float* __restrict__ stat, const NaGeom g, int T,
                                                             int tiles_x, int tiles_img) {
  // Key pass for C <= 24 with the strided operands of the 3x3 query neighbourhood -- q (48..96 B of a 144..288 B
  // pixel) and the per-head lse / dsum -- of a T x T key tile staged in LDS ([pixel][C + 2*heads], window T+2 squared);
  // dO is contiguous per pixel and stays a global load.  Keys within 3 pixels of the border (clamped windows: up to 25
  // candidate queries) keep the direct path.
  extern __shared__ __attribute__((aligned(16))) float QS[];
  __shared__ __attribute__((aligned(16))) float s_rpb[16 * 25];   // bias table, transposed to [25][heads] (rpb4)
  for (int i = threadIdx.x; i < g.heads * 25; i += 256) s_rpb[(i % 25) * g.heads + i / 25] = rpb[i];
  const int PS = g.C + 2 * g.heads, RW = T + 2;
  const int tile = blockIdx.x;
  const int b = tile / tiles_img, tt = tile - b * tiles_img;
  const int ty0 = (tt / tiles_x) * T, tx0 = (tt % tiles_x) * T;
  const int64_t ib = (int64_t)b * g.H * g.W * 3 * g.C;
  const TA* base = qkv + ib;
  {
    const int qf = g.C4, sf = g.heads / 2, per = qf + sf;  // float4 items per window pixel
    for (int i0 = 0; i0 < RW * RW * per; i0 += 4 * 256) {   // rounds of 4 loads in flight per thread: the plain loop compiled to one exposed round trip per item (level 0: backward 330 -> 299 us; the forward loses more to the 16 extra VGPRs than it gains)
      f32x4 sv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * 256 + threadIdx.x;
        const int f = i % per, wp = i / per;
        const int gy = ty0 - 1 + wp / RW, gx = tx0 - 1 + wp % RW;
        const bool in = i < RW * RW * per && gy >= 0 && gy < g.H && gx >= 0 && gx < g.W;
        const int64_t ipix = ((int64_t)b * g.H + (in ? gy : 0)) * g.W + (in ? gx : 0);
        sv[u] = f < qf ? ld4(base + ((int64_t)(in ? gy : 0) * g.W + (in ? gx : 0)) * 3 * g.C + f * 4)
                       : ld4(stat + ipix * 2 * g.heads + (f - qf) * 4);
        if (!in) sv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * 256 + threadIdx.x;
        if (i < RW * RW * per) *reinterpret_cast<f32x4*>(&QS[(i / per) * PS + (i % per) * 4]) = sv[u];
      }
    }
  }
  __syncthreads();
  const int items = T * T * g.C4;
  for (int it0 = 0; it0 < items; it0 += 256) {
    int idx = it0 + threadIdx.x;
    bool ok = idx < items;
    if (!ok) idx = 0;
    const int c = (idx % g.C4) * 4;
    const int pl = idx / g.C4;
    int jy = ty0 + pl / T, jx = tx0 + pl % T;
    if (jy >= g.H || jx >= g.W) { ok = false; jy = ty0; jx = tx0; }
    const int64_t kpo = ((int64_t)jy * g.W + jx) * 3 * g.C;
    const f32x4 kj = ld4(base + kpo + g.C + c), vj = ld4(base + kpo + 2 * g.C + c);
    int hidx[4], hd_[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { hd_[k] = (c + k) / HD; hidx[k] = hd_[k] * 25; }
    f32x4 dk = f32x4{0.f, 0.f, 0.f, 0.f}, dv = dk;
    auto candidate = [&](int iy, int ix, bool lds) {
      const int64_t ipix = ((int64_t)b * g.H + iy) * g.W + ix;
      const float* wq = QS + ((iy - ty0 + 1) * RW + (ix - tx0 + 1)) * PS;
      const f32x4 qi = (lds ? *reinterpret_cast<const f32x4*>(wq + c) : ld4(base + ((int64_t)iy * g.W + ix) * 3 * g.C + c)) * g.scale;
      const f32x4 dOi = ld4(dout + ipix * g.C + c);
      f32x4 lse4, dsm4;
      const float* sp = lds ? wq + g.C : stat + ipix * 2 * g.heads;
      if constexpr (HD == 1) {
        lse4 = ld4(sp + hd_[0]);
        dsm4 = ld4(sp + g.heads + hd_[0]);
      } else if constexpr (HD == 2) {
        const float2 a = *reinterpret_cast<const float2*>(sp + hd_[0]), bq = *reinterpret_cast<const float2*>(sp + g.heads + hd_[0]);
        lse4 = f32x4{a.x, a.x, a.y, a.y};
        dsm4 = f32x4{bq.x, bq.x, bq.y, bq.y};
      } else {
        const float a = sp[hd_[0]], bq = sp[g.heads + hd_[0]];
        lse4 = f32x4{a, a, a, a};
        dsm4 = f32x4{bq, bq, bq, bq};
      }
      const f32x4 s = head_sum<HD>(qi * kj);
      const f32x4 dp = head_sum<HD>(dOi * vj);
      const int bo = (jy - iy + 2) * 5 + (jx - ix + 2);
      const f32x4 rb = rpb4<HD>(s_rpb, g.heads, bo, c);
      f32x4 pij, ds;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        pij[k] = __expf(s[k] + rb[k] - lse4[k]);
        ds[k] = pij[k] * (dp[k] - dsm4[k]);
      }
      dk += ds * qi;   // qi carries the scale
      dv += pij * dOi;
    };
    // Interior keys (99 % of them): the queries that see key j are exactly its 3x3 neighbourhood, all with unclamped
    // windows -- a fixed, fully unrolled candidate set whose 36 loads can be issued together.  The generic loop with
    // its per-candidate window tests (16 of 25 candidates fail them) is kept for keys within 3 pixels of the border.
    // (The test is pixel-uniform, so the lane pairs of hd >= 8 stay converged in head_sum.)
    // (queries 2 pixels away see key j only through a CLAMPED window, i.e. when they sit on the border: j in [3, L-4])
    if (jy >= 3 && jy <= g.H - 4 && jx >= 3 && jx <= g.W - 4) {
#pragma unroll
      for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) candidate(jy + dy, jx + dx, true);
    } else {
      for (int iy = jy - 2; iy <= jy + 2; ++iy) {
        if (iy < 0 || iy >= g.H) continue;
        const int ki = jy - wstart(iy, g.H);
        if (ki < 0 || ki > 2) continue;
        for (int ix = jx - 2; ix <= jx + 2; ++ix) {
          if (ix < 0 || ix >= g.W) continue;
          const int kx = jx - wstart(ix, g.W);
          if (kx < 0 || kx > 2) continue;
          candidate(iy, ix, false);
        }
      }
    }
    if (ok) {
      st4(dqkv + ib + kpo + g.C + c, dk);
      st4(dqkv + ib + kpo + 2 * g.C + c, dv);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Backward in ONE pass for head_dim <= 2 (levels 0-1: C = 12 / 24).  At head_dim 1 the per-(pixel, head) scalars the two-pass form
// exchanges through HBM (lse, dsum) are as large as q or k themselves; here they live in LDS for one tile:
//   stage k, v of the tile +- 2 (the windows of the queries below)                                   -> LDS window W
//   phase A, queries of the tile +- 1 (every query that sees a key of the tile): softmax statistics   -> LDS table ST;
//            queries INSIDE the tile also: dq -> HBM, bias-table gradient (register bins as in na_bwd_q_tile_kernel)
//   restage q, dO of the tile +- 1 over W (second read of the same bytes: L2)
//   phase B, keys of the tile: dk, dv from the 3x3 (border: up to 5x5) candidate queries, everything from LDS -> HBM
// Queries 2 pixels from a key see it only through a window clamped at the image border (query row 0 sees key row 2, row H-1 sees
// H-3): the statistics region is the tile +- 1 clipped to the image, extended by one row / column where that reaches H-2 / W-2.
// A thread keeps one channel quad; blockDim = 85 pixel slots x C/4 quads (255 of 256 threads at C = 12, 510 of 512 at C = 24), and
// the 15 x 17 tile is 3 such rounds, its 17 x 19 statistics region 4.  HBM traffic: the algorithmic 7 C per pixel plus halo
// re-reads that neighbouring blocks find in L2; no lse / dsum workspace.
// q = (n * m) >> 16 with m = ceil(65536 / d): exact for n < 1024, d <= 64 (pixel indices inside a tile's windows); the generic
// 32-bit signed divisions the index arithmetic compiled to were a quarter of the kernel's instructions
__device__ __forceinline__ uint32_t na_magic16(uint32_t d) { return (65535u + d) / d; }
__device__ __forceinline__ int na_div16(int n, uint32_t m) { return (int)(((uint32_t)n * m) >> 16); }

#ifdef LMN_NA_DBG
__device__ unsigned g_na_dbg[8];   // debug build: {dO re-read differs, v window != memory, dp re-evaluation differs, dsum rebuild differs, k window != memory, -, -, items}
#endif
template <int HD, int C4, typename TA>
__global__ __launch_bounds__(512) void na_bwd_fused_kernel(const TA* __restrict__ qkv, const float* __restrict__ rpb,
                                                           const TA* __restrict__ dout, TA* __restrict__ dqkv,
                                                           float* __restrict__ drpb, const NaGeom g, int TH, int TW, int tiles_x,
                                                           int tiles_img, int total_tiles, int st_floats, int det) {
  extern __shared__ __attribute__((aligned(16))) float s_drpb[];  // [ntab] | bias table, transposed [ntab] | border bins per wave | ST | W
  const int NT = blockDim.x, nwave = NT >> 6;
  const int ntab = (g.heads * 25 + 3) & ~3;
  float* s_rpb = s_drpb + ntab;
  float* s_wt = s_rpb + ntab;                                  // [nwave][ntab]
  float* s_wave = s_wt + (threadIdx.x >> 6) * ntab;
  float* ST = s_wt + nwave * ntab;                             // [statistics region][2*heads]: lse | dsum  (st_floats: the largest region of this map)
  float* Wn = ST + st_floats;                                  // k, v window, then q, dO window, then the parked bins
  const int tid = threadIdx.x;
  // scores are kept in base 2 (q and the bias table carry log2(e)): softmax weights are v_exp_f32 of a difference, no multiply
  constexpr float LOG2E = 1.4426950408889634f;
  const float scale2 = g.scale * LOG2E;
  for (int i = tid; i < g.heads * 25; i += NT) { s_drpb[i] = 0.f; s_rpb[(i % 25) * g.heads + i / 25] = rpb[i] * LOG2E; }
  for (int i = tid; i < nwave * ntab; i += NT) s_wt[i] = 0.f;
  constexpr int C = 4 * C4;   // (= g.C: a template argument so that the channel-quad arithmetic is constant division)
  const int PB = NT / C4;
  const int c = (tid % C4) * 4;
  const int slot = tid / C4;
  const int SH2 = 2 * g.heads;
  f32x4 bins[9];
#pragma unroll
  for (int n = 0; n < 9; ++n) bins[n] = f32x4{0.f, 0.f, 0.f, 0.f};
  int hidx[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) hidx[k] = ((c + k) / HD) * 25;
  // (XCD-aware order: the blocks of one XCD walk CONSECUTIVE tiles, whose halos overlap, at the same time -- the re-reads hit that XCD's L2)
  for (int tile = xcd_swizzle(blockIdx.x, gridDim.x); tile < total_tiles; tile += gridDim.x) {
    const int b = tile / tiles_img, tt = tile - b * tiles_img;
    const int ty0 = (tt / tiles_x) * TH, tx0 = (tt % tiles_x) * TW;
    const int ty1 = min(ty0 + TH, g.H), tx1 = min(tx0 + TW, g.W);
    // statistics region (queries) and k / v region (their windows)
    int sy0 = max(ty0 - 1, 0), sy1 = min(ty1 + 1, g.H), sx0 = max(tx0 - 1, 0), sx1 = min(tx1 + 1, g.W);
    if (sy0 == 1) sy0 = 0;
    if (sx0 == 1) sx0 = 0;
    if (sy1 == g.H - 1) sy1 = g.H;
    if (sx1 == g.W - 1) sx1 = g.W;
    const int SHt = sy1 - sy0, SWd = sx1 - sx0, nsp = SHt * SWd;
    const int ky0 = wstart(sy0, g.H), kx0 = wstart(sx0, g.W);
    const int KH = wstart(sy1 - 1, g.H) + 3 - ky0, KW = wstart(sx1 - 1, g.W) + 3 - kx0;
    const uint32_t mKW = na_magic16((uint32_t)KW), mSW = na_magic16((uint32_t)SWd);
    const int64_t ib = (int64_t)b * g.H * g.W * 3 * C;
    const TA* base = qkv + ib;
    __syncthreads();  // the previous tile's windows are consumed (first tile: the tables are written)
    {  // ---- k, v of the K region -> Wn [(pixel)*2 + which][C]
      const int nit = KH * KW * 2 * C4;
      for (int i0 = 0; i0 < nit; i0 += 4 * NT) {
        f32x4 sv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = i0 + u * NT + tid;
          const bool in = i < nit;
          const int ii = in ? i : 0;
          const int c4 = ii % C4, w = (ii / C4) & 1, wp = ii / (2 * C4);
          const int wr = na_div16(wp, mKW);
          const int gy = ky0 + wr, gx = kx0 + wp - wr * KW;
          sv[u] = ld4(base + (gy * g.W + gx) * 3 * C + (1 + w) * C + c4 * 4);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = i0 + u * NT + tid;
          if (i < nit) *reinterpret_cast<f32x4*>(&Wn[i * 4]) = sv[u];      // [(pixel * 2 + which) * C + quad * 4] = 4 i
        }
      }
    }
    __syncthreads();
    // ---- phase A
    for (int p0 = 0; p0 < nsp; p0 += PB) {
      const int pl = p0 + slot;
      const bool ok = slot < PB && pl < nsp;
      const int pls = ok ? pl : 0;
      const int pr = na_div16(pls, mSW);
      const int y = sy0 + pr, x = sx0 + pls - pr * SWd;
      const bool own = ok && y >= ty0 && y < ty1 && x >= tx0 && x < tx1;   // a query of the tile: dq and the bias gradient are this block's
      const int64_t pix = ((int64_t)b * g.H + y) * g.W + x;
      const int sy = wstart(y, g.H), sx = wstart(x, g.W);
      const float* kv0 = Wn + (((sy - ky0) * KW + (sx - kx0)) * 2) * C + c;
      const f32x4 q = ld4(base + (y * g.W + x) * 3 * C + c) * scale2;
      const f32x4 dO = ld4(dout + pix * C + c);
      f32x4 p[9], dp[9];
      f32x4 mx = f32x4{-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
#pragma unroll
      for (int ki = 0; ki < 3; ++ki)
#pragma unroll
        for (int kj = 0; kj < 3; ++kj) {
          const float* kvn = kv0 + ((ki * KW + kj) * 2) * C;
          const f32x4 kk = *reinterpret_cast<const f32x4*>(kvn), vv = *reinterpret_cast<const f32x4*>(kvn + C);
          f32x4 sc = head_sum<HD>(q * kk);
          const int bo = (sy + ki - y + 2) * 5 + (sx + kj - x + 2);
          sc += rpb4<HD>(s_rpb, g.heads, bo, c);
          p[ki * 3 + kj] = sc;
          dp[ki * 3 + kj] = head_sum<HD>(dO * vv);
#pragma unroll
          for (int k = 0; k < 4; ++k) mx[k] = fmaxf(mx[k], sc[k]);
        }
      f32x4 den = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int n = 0; n < 9; ++n) {
#pragma unroll
        for (int k = 0; k < 4; ++k) p[n][k] = __builtin_amdgcn_exp2f(p[n][k] - mx[k]);
#ifdef LMN_NA_EXPNOP     // experiment build: the same for the v_exp_f32 results
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 7\n\ts_nop 7" : "+v"(p[n][0]), "+v"(p[n][1]), "+v"(p[n][2]), "+v"(p[n][3]));
        __builtin_amdgcn_sched_barrier(0);
#endif
        den += p[n];
      }
      f32x4 dsum = f32x4{0.f, 0.f, 0.f, 0.f}, rden, lse;   // lse in base 2 as well
#pragma unroll
      for (int k = 0; k < 4; ++k) { rden[k] = __builtin_amdgcn_rcpf(den[k]); lse[k] = mx[k] + __builtin_amdgcn_logf(den[k]); }
#ifdef LMN_NA_TRANSNOP   // experiment build (make x2na): the v_rcp_f32 / v_log_f32 results have 16 idle cycles to land before their first reader
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_nop 7\n\ts_nop 7" : "+v"(rden[0]), "+v"(rden[1]), "+v"(rden[2]), "+v"(rden[3]));
      __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
      for (int n = 0; n < 9; ++n) {
        p[n] = p[n] * rden;
        dsum += p[n] * dp[n];
      }
#ifdef LMN_NA_MEMBAR   // experiment build (make x2na): ONLY the compiler barrier of the debug block below -- does the changed schedule alone hide the failure?
      asm volatile("" ::: "memory");
#endif
#ifdef LMN_NA_WAIT0    // experiment build: every outstanding memory operation of the item has landed before the statistics leave
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
#ifdef LMN_NA_DBG
      {   // debug build (tools/gpu_x2_canary.py): read dO and the v window AGAIN and rebuild dsum -- which operand moved?
        asm volatile("" ::: "memory");
        const f32x4 dO2 = ld4(dout + pix * C + c);
        f32x4 dsum2 = f32x4{0.f, 0.f, 0.f, 0.f};
        unsigned bad_v = 0, bad_k = 0;
#pragma unroll
        for (int ki = 0; ki < 3; ++ki)
#pragma unroll
          for (int kj = 0; kj < 3; ++kj) {
            const float* kvn = kv0 + ((ki * KW + kj) * 2) * C;
            const f32x4 vv2 = *reinterpret_cast<const f32x4*>(kvn + C);
            const f32x4 dp2 = head_sum<HD>(dO * vv2);
            const int gy = sy + ki, gx = sx + kj;
            const f32x4 vg = ld4(base + (gy * g.W + gx) * 3 * C + 2 * C + c), kg = ld4(base + (gy * g.W + gx) * 3 * C + C + c);
            const f32x4 kk2 = *reinterpret_cast<const f32x4*>(kvn);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              bad_v += __float_as_uint(vv2[k]) != __float_as_uint(vg[k]);   // LDS window vs global memory
              bad_k += __float_as_uint(kk2[k]) != __float_as_uint(kg[k]);
              if (__float_as_uint(dp2[k]) != __float_as_uint(dp[ki * 3 + kj][k]) && ok) atomicAdd(&g_na_dbg[2], 1u);   // same inputs, second evaluation differs
            }
            dsum2 += p[ki * 3 + kj] * dp2;
          }
        if (ok) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            if (__float_as_uint(dO2[k]) != __float_as_uint(dO[k])) atomicAdd(&g_na_dbg[0], 1u);     // dO changed between two reads
            if (__float_as_uint(dsum2[k]) != __float_as_uint(dsum[k])) atomicAdd(&g_na_dbg[3], 1u); // rebuilt dsum differs
          }
          if (bad_v) atomicAdd(&g_na_dbg[1], bad_v);
          if (bad_k) atomicAdd(&g_na_dbg[4], bad_k);
          atomicAdd(&g_na_dbg[7], 1u);
        }
      }
#endif
      stat_store<HD>(ST + pls * SH2, c, lse, ok);
      stat_store<HD>(ST + pls * SH2 + g.heads, c, dsum, ok);
      if (own) {   // (block-divergent only at tile edges; head_sum has no shuffles at hd <= 2)
        f32x4 dq = f32x4{0.f, 0.f, 0.f, 0.f};
        const bool inter = sy == y - 1 && sx == x - 1;  // unclamped window: neighbour n always lands in bin (ki+1, kj+1)
        f32x4 rm;
#pragma unroll
        for (int k = 0; k < 4; ++k) rm[k] = (inter && (c + k) % HD == 0) ? 1.f : 0.f;
#pragma unroll
        for (int ki = 0; ki < 3; ++ki)
#pragma unroll
          for (int kj = 0; kj < 3; ++kj) {
            const int n = ki * 3 + kj;
            const f32x4 ds = p[n] * (dp[n] - dsum);
            dq += ds * *reinterpret_cast<const f32x4*>(kv0 + ((ki * KW + kj) * 2) * C);
            bins[n] += ds * rm;
            if (!inter) {  // border pixels (the clamped window shifts the bins): rare, through LDS atomics
              const int bo = (sy + ki - y + 2) * 5 + (sx + kj - x + 2);
#pragma unroll
              for (int k = 0; k < 4; ++k)
                if ((c + k) % HD == 0) atomicAdd(&s_wave[hidx[k] + bo], ds[k]);
            }
          }
        st4(dqkv + ib + (y * g.W + x) * 3 * C + c, dq * g.scale);
      }
    }
    __syncthreads();  // ST complete; the k / v window is consumed
    {  // ---- q, dO of the statistics region -> Wn [(pixel)*2 + which][C]
      const int nit = nsp * 2 * C4;
      const TA* dob = dout + (int64_t)b * g.H * g.W * C;
      for (int i0 = 0; i0 < nit; i0 += 4 * NT) {
        f32x4 sv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = i0 + u * NT + tid;
          const bool in = i < nit;
          const int ii = in ? i : 0;
          const int c4 = ii % C4, w = (ii / C4) & 1, wp = ii / (2 * C4);
          const int wr = na_div16(wp, mSW);
          const int gp = (sy0 + wr) * g.W + sx0 + wp - wr * SWd;
          sv[u] = w ? ld4(dob + gp * C + c4 * 4) : ld4(base + gp * 3 * C + c4 * 4) * scale2;   // (q staged with the score scale: once per element, not per candidate)
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = i0 + u * NT + tid;
          if (i < nit) *reinterpret_cast<f32x4*>(&Wn[i * 4]) = sv[u];
        }
      }
    }
    __syncthreads();
    // ---- phase B
    const int tw = tx1 - tx0, ntp = (ty1 - ty0) * tw;
    const uint32_t mtw = na_magic16((uint32_t)tw);
    for (int p0 = 0; p0 < ntp; p0 += PB) {
      const int pl = p0 + slot;
      const bool ok = slot < PB && pl < ntp;
      const int pls = ok ? pl : 0;
      const int jr = na_div16(pls, mtw);
      const int jy = ty0 + jr, jx = tx0 + pls - jr * tw;
      const int kpo = (jy * g.W + jx) * 3 * C;
      const f32x4 kj = ld4(base + kpo + C + c), vj = ld4(base + kpo + 2 * C + c);
      f32x4 dk = f32x4{0.f, 0.f, 0.f, 0.f}, dv = dk;
      auto candidate = [&](int iy, int ix) {
        const int sl = (iy - sy0) * SWd + (ix - sx0);
        const f32x4 qi = *reinterpret_cast<const f32x4*>(Wn + (sl * 2) * C + c);
        const f32x4 dOi = *reinterpret_cast<const f32x4*>(Wn + (sl * 2 + 1) * C + c);
        const float* sp = ST + sl * SH2;
        f32x4 lse4, dsm4;
        if constexpr (HD == 1) {
          lse4 = *reinterpret_cast<const f32x4*>(sp + c);
          dsm4 = *reinterpret_cast<const f32x4*>(sp + g.heads + c);
        } else {
          const float2 a = *reinterpret_cast<const float2*>(sp + c / 2), bq = *reinterpret_cast<const float2*>(sp + g.heads + c / 2);
          lse4 = f32x4{a.x, a.x, a.y, a.y};
          dsm4 = f32x4{bq.x, bq.x, bq.y, bq.y};
        }
        const f32x4 sc = head_sum<HD>(qi * kj);
        const f32x4 dpv = head_sum<HD>(dOi * vj);
        const int bo = (jy - iy + 2) * 5 + (jx - ix + 2);
        const f32x4 rb = rpb4<HD>(s_rpb, g.heads, bo, c);
        f32x4 pij, ds;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          pij[k] = __builtin_amdgcn_exp2f(sc[k] + rb[k] - lse4[k]);
          ds[k] = pij[k] * (dpv[k] - dsm4[k]);
        }
        dk += ds * qi;   // qi carries scale * log2(e): taken out once below
        dv += pij * dOi;
      };
      if (jy >= 3 && jy <= g.H - 4 && jx >= 3 && jx <= g.W - 4) {   // interior key: exactly its 3x3 neighbourhood, all unclamped
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
          for (int dx = -1; dx <= 1; ++dx) candidate(jy + dy, jx + dx);
      } else {
        for (int iy = jy - 2; iy <= jy + 2; ++iy) {
          if (iy < 0 || iy >= g.H) continue;
          const int ki = jy - wstart(iy, g.H);
          if (ki < 0 || ki > 2) continue;
          for (int ix = jx - 2; ix <= jx + 2; ++ix) {
            if (ix < 0 || ix >= g.W) continue;
            const int kx = jx - wstart(ix, g.W);
            if (kx < 0 || kx > 2) continue;
            candidate(iy, ix);
          }
        }
      }
      if (ok) {
        st4(dqkv + ib + kpo + C + c, dk * (1.0f / LOG2E));
        st4(dqkv + ib + kpo + 2 * C + c, dv);
      }
    }
  }
  __syncthreads();  // every window read is done: the parking area reuses that memory
  float* s_bins = Wn;   // [NT][36]
#pragma unroll
  for (int n = 0; n < 9; ++n)
#pragma unroll
    for (int k = 0; k < 4; ++k) s_bins[tid * 36 + n * 4 + k] = bins[n][k];
  __syncthreads();
  for (int o = tid; o < C4 * 36; o += NT) {
    const int qd = o / 36, nk = o - qd * 36;
    const int n = nk >> 2, k = nk & 3;
    const int ch = qd * 4 + k;
    if (ch % HD != 0) continue;
    float v = 0.f;
    for (int sl = 0; sl < PB; ++sl) v += s_bins[(sl * C4 + qd) * 36 + nk];
    s_drpb[(ch / HD) * 25 + (n / 3 + 1) * 5 + (n % 3 + 1)] = v;   // (one writer per (head, bin))
  }
  __syncthreads();
  for (int i = tid; i < g.heads * 25; i += NT) {
    float v = s_drpb[i];
    for (int w = 0; w < nwave; ++w) v += s_wt[w * ntab + i];
    if (v != 0.f) lmn_red_add(drpb + (det ? (int64_t)blockIdx.x * g.heads * 25 : 0) + i, v, det);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Window sizes other than 3 (natten's kernel_size 5, 7, ...: core/LM_Net.py:81-84 passes [3, 5], BASELINE configs[4] names a
// larger window): the same three passes in the direct form -- a thread owns a channel quad of one pixel, lanes walk the
// channel axis first, the K x K re-use of k / v is served by L1 / L2 -- with K a run-time argument (one instantiation per
// head_dim).  Scores are formed twice (max, then exp / sums) instead of being kept in K*K registers.  K = 3 keeps the
// LDS-tiled kernels above; tests/kernel_checks.py also runs these on K = 3 as a cross-check of the two forms.
__device__ __forceinline__ int wstart_k(int i, int L, int K) {
  int s = i - (K >> 1);
  s = s < 0 ? 0 : s;
  return s > L - K ? L - K : s;
}

template <int HD, typename TA>
__global__ __launch_bounds__(256) void na_fwd_gen_kernel(const TA* __restrict__ qkv, const float* __restrict__ rpb,
                                                         TA* __restrict__ out, const NaGeom g, int K) {
  const int RB = 2 * K - 1;
  const int total = g.B * g.H * g.W * g.C4;
  const int nit = (total + (int)gridDim.x * 256 - 1) / ((int)gridDim.x * 256);
  for (int it = 0; it < nit; ++it) {
    int idx = (it * (int)gridDim.x + (int)blockIdx.x) * 256 + (int)threadIdx.x;
    const bool ok = idx < total;
    if (!ok) idx = total - 1;   // keep every lane in the shuffles
    const NaPix pd = na_decode_item(g, (uint32_t)idx);
    const int c = pd.c;
    const int64_t pix = (int64_t)(pd.b * g.H + pd.y) * g.W + pd.x;
    const int x = pd.x, y = pd.y, b = pd.b;
    const TA* base = qkv + (int64_t)b * g.H * g.W * 3 * g.C;
    const int sy = wstart_k(y, g.H, K), sx = wstart_k(x, g.W, K);
    const f32x4 q = ld4(base + ((int64_t)y * g.W + x) * 3 * g.C + c) * g.scale;
    int hidx[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) hidx[k] = ((c + k) / HD) * RB * RB;
    auto score = [&](int ki, int kj) {
      const f32x4 kk = ld4(base + ((int64_t)(sy + ki) * g.W + sx + kj) * 3 * g.C + g.C + c);
      f32x4 sc = head_sum<HD>(q * kk);
      const int bo = (sy + ki - y + K - 1) * RB + (sx + kj - x + K - 1);
#pragma unroll
      for (int k = 0; k < 4; ++k) sc[k] += rpb[hidx[k] + bo];
      return sc;
    };
    f32x4 mx = f32x4{-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
    for (int ki = 0; ki < K; ++ki)
      for (int kj = 0; kj < K; ++kj) {
        const f32x4 sc = score(ki, kj);
#pragma unroll
        for (int k = 0; k < 4; ++k) mx[k] = fmaxf(mx[k], sc[k]);
      }
    f32x4 den = f32x4{0.f, 0.f, 0.f, 0.f}, o = den;
    for (int ki = 0; ki < K; ++ki)
      for (int kj = 0; kj < K; ++kj) {
        f32x4 e = score(ki, kj);
#pragma unroll
        for (int k = 0; k < 4; ++k) e[k] = __expf(e[k] - mx[k]);
        den += e;
        o += e * ld4(base + ((int64_t)(sy + ki) * g.W + sx + kj) * 3 * g.C + 2 * g.C + c);
      }
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = o[k] / den[k];
    if (ok) st4(out + pix * g.C + c, o);
  }
}

template <int HD, typename TA>
__global__ __launch_bounds__(256) void na_bwd_q_gen_kernel(const TA* __restrict__ qkv, const float* __restrict__ rpb,
                                                           const TA* __restrict__ dout, TA* __restrict__ dqkv,
                                                           float* __restrict__ drpb, float* __restrict__ stat,
                                                           const NaGeom g, int K, int det) {
  extern __shared__ float s_all[];  // bias-gradient bins per wave [4][heads][(2K-1)^2] (program order within a wave, fixed-order sum at the end)
  const int RB = 2 * K - 1, NB = g.heads * RB * RB;
  float* s_tab = s_all + (threadIdx.x >> 6) * NB;
  for (int i = threadIdx.x; i < 4 * NB; i += 256) s_all[i] = 0.f;
  __syncthreads();
  const int total = g.B * g.H * g.W * g.C4;
  const int nit = (total + (int)gridDim.x * 256 - 1) / ((int)gridDim.x * 256);
  for (int it = 0; it < nit; ++it) {
    int idx = (it * (int)gridDim.x + (int)blockIdx.x) * 256 + (int)threadIdx.x;
    const bool ok = idx < total;
    if (!ok) idx = total - 1;
    const NaPix pd = na_decode_item(g, (uint32_t)idx);
    const int c = pd.c;
    const int64_t pix = (int64_t)(pd.b * g.H + pd.y) * g.W + pd.x;
    const int x = pd.x, y = pd.y, b = pd.b;
    const int64_t ib = (int64_t)b * g.H * g.W * 3 * g.C;
    const TA* base = qkv + ib;
    const int sy = wstart_k(y, g.H, K), sx = wstart_k(x, g.W, K);
    const f32x4 q = ld4(base + ((int64_t)y * g.W + x) * 3 * g.C + c) * g.scale;
    const f32x4 dO = ld4(dout + pix * g.C + c);
    int hidx[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) hidx[k] = ((c + k) / HD) * RB * RB;
    auto score = [&](int ki, int kj, f32x4& kk) {
      kk = ld4(base + ((int64_t)(sy + ki) * g.W + sx + kj) * 3 * g.C + g.C + c);
      f32x4 sc = head_sum<HD>(q * kk);
      const int bo = (sy + ki - y + K - 1) * RB + (sx + kj - x + K - 1);
#pragma unroll
      for (int k = 0; k < 4; ++k) sc[k] += rpb[hidx[k] + bo];
      return sc;
    };
    f32x4 kk;
    f32x4 mx = f32x4{-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
    for (int ki = 0; ki < K; ++ki)
      for (int kj = 0; kj < K; ++kj) {
        const f32x4 sc = score(ki, kj, kk);
#pragma unroll
        for (int k = 0; k < 4; ++k) mx[k] = fmaxf(mx[k], sc[k]);
      }
    f32x4 den = f32x4{0.f, 0.f, 0.f, 0.f}, edp = den;
    for (int ki = 0; ki < K; ++ki)
      for (int kj = 0; kj < K; ++kj) {
        f32x4 e = score(ki, kj, kk);
#pragma unroll
        for (int k = 0; k < 4; ++k) e[k] = __expf(e[k] - mx[k]);
        den += e;
        edp += e * head_sum<HD>(dO * ld4(base + ((int64_t)(sy + ki) * g.W + sx + kj) * 3 * g.C + 2 * g.C + c));
      }
    f32x4 dsum, rden;
#pragma unroll
    for (int k = 0; k < 4; ++k) { rden[k] = 1.f / den[k]; dsum[k] = edp[k] * rden[k]; }
    bool rep[4];  // the lane owning a head's FIRST channel reports for that head
#pragma unroll
    for (int k = 0; k < 4; ++k) rep[k] = ok && ((c + k) % HD == 0);
    f32x4 dq = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int ki = 0; ki < K; ++ki)
      for (int kj = 0; kj < K; ++kj) {
        f32x4 e = score(ki, kj, kk);
        const f32x4 dp = head_sum<HD>(dO * ld4(base + ((int64_t)(sy + ki) * g.W + sx + kj) * 3 * g.C + 2 * g.C + c));
        f32x4 ds;
#pragma unroll
        for (int k = 0; k < 4; ++k) ds[k] = __expf(e[k] - mx[k]) * rden[k] * (dp[k] - dsum[k]);
        dq += ds * kk;
        const int bo = (sy + ki - y + K - 1) * RB + (sx + kj - x + K - 1);
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (rep[k]) atomicAdd(&s_tab[hidx[k] + bo], ds[k]);
      }
    if (ok) st4(dqkv + ib + ((int64_t)y * g.W + x) * 3 * g.C + c, dq * g.scale);
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (rep[k]) {
        const int h = (c + k) / HD;
        stat[pix * 2 * g.heads + h] = mx[k] + __logf(den[k]);
        stat[pix * 2 * g.heads + g.heads + h] = dsum[k];
      }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < NB; i += 256) {
    const float v = ((s_all[i] + s_all[NB + i]) + s_all[2 * NB + i]) + s_all[3 * NB + i];
    if (v != 0.f) lmn_red_add(drpb + (det ? (int64_t)blockIdx.x * NB : 0) + i, v, det);
  }
}

template <int HD, typename TA>
__global__ __launch_bounds__(256) void na_bwd_kv_gen_kernel(const TA* __restrict__ qkv, const float* __restrict__ rpb,
                                                            const TA* __restrict__ dout, TA* __restrict__ dqkv,
                                                            const float* __restrict__ stat, const NaGeom g, int K) {
  const int RB = 2 * K - 1;
  const int total = g.B * g.H * g.W * g.C4;
  const int nit = (total + (int)gridDim.x * 256 - 1) / ((int)gridDim.x * 256);
  for (int it = 0; it < nit; ++it) {
    int idx = (it * (int)gridDim.x + (int)blockIdx.x) * 256 + (int)threadIdx.x;
    const bool ok = idx < total;
    if (!ok) idx = total - 1;
    const NaPix pd = na_decode_item(g, (uint32_t)idx);
    const int c = pd.c;
    const int jx = pd.x, jy = pd.y, b = pd.b;
    const int64_t ib = (int64_t)b * g.H * g.W * 3 * g.C;
    const TA* base = qkv + ib;
    const int64_t kpo = ((int64_t)jy * g.W + jx) * 3 * g.C;
    const f32x4 kj = ld4(base + kpo + g.C + c), vj = ld4(base + kpo + 2 * g.C + c);
    int hd_[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) hd_[k] = (c + k) / HD;
    f32x4 dk = f32x4{0.f, 0.f, 0.f, 0.f}, dv = dk;
    // key j is seen by query i iff 0 <= j - wstart(i) < K on both axes (pixel-uniform tests: lane pairs stay converged)
    for (int iy = jy - K + 1; iy <= jy + K - 1; ++iy) {
      if (iy < 0 || iy >= g.H) continue;
      const int ki = jy - wstart_k(iy, g.H, K);
      if (ki < 0 || ki >= K) continue;
      for (int ix = jx - K + 1; ix <= jx + K - 1; ++ix) {
        if (ix < 0 || ix >= g.W) continue;
        const int kx = jx - wstart_k(ix, g.W, K);
        if (kx < 0 || kx >= K) continue;
        const int64_t ipix = ((int64_t)b * g.H + iy) * g.W + ix;
        const f32x4 qi = ld4(base + ((int64_t)iy * g.W + ix) * 3 * g.C + c) * g.scale;
        const f32x4 dOi = ld4(dout + ipix * g.C + c);
        const float* sp = stat + ipix * 2 * g.heads;
        const f32x4 s = head_sum<HD>(qi * kj), dp = head_sum<HD>(dOi * vj);
        const int bo = (jy - iy + K - 1) * RB + (jx - ix + K - 1);
        f32x4 pij, ds;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          pij[k] = __expf(s[k] + rpb[hd_[k] * RB * RB + bo] - sp[hd_[k]]);
          ds[k] = pij[k] * (dp[k] - sp[g.heads + hd_[k]]);
        }
        dk += ds * qi;   // qi carries the scale
        dv += pij * dOi;
      }
    }
    if (ok) {
      st4(dqkv + ib + kpo + g.C + c, dk);
      st4(dqkv + ib + kpo + 2 * g.C + c, dv);
    }
  }
}

}  // namespace

#ifdef LMN_NA_DBG
extern "C" int lmn_na_dbg(unsigned* out, int reset) {
  if (reset) { unsigned z[8] = {0, 0, 0, 0, 0, 0, 0, 0}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_na_dbg), z, sizeof(z)); }
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_na_dbg), sizeof(unsigned) * 8);
}
#endif
extern "C" {

int lmn_na_fwd(const void* qkv, const float* rpb, void* out, int B, int H, int W, int heads, int hd, int K, float scale,
               int act_dtype, lmn_stream_t stream) {
  LMN_REC(lmn_na_fwd(qkv, rpb, out, B, H, W, heads, hd, K, scale, act_dtype, stream));
  LMN_REQUIRE_DT(act_dtype, "na_fwd");
  LMN_REQUIRE(qkv && rpb && out, "na_fwd: null pointer");
  const bool gen = K < 0;   // -K: the direct (run-time K) form, also for K = 3 (tests)
  if (gen) K = -K;
  LMN_REQUIRE(K >= 3 && K <= 13 && (K & 1), "na_fwd: window %d (odd, 3..13)", K);
  LMN_REQUIRE(B > 0 && H >= K && W >= K, "na_fwd: feature map %dx%d smaller than the %dx%d window", H, W, K, K);
  LMN_REQUIRE(hd == 1 || hd == 2 || hd == 4 || hd == 8 || hd == 16, "na_fwd: head_dim %d not in {1,2,4,8,16}", hd);
  LMN_REQUIRE((heads * hd) % 4 == 0, "na_fwd: C=%d must be a multiple of 4", heads * hd);
  LMN_REQUIRE(heads <= 16, "na_fwd: %d heads (the LDS bias table holds 16)", heads);
  LMN_REQUIRE((int64_t)B * H * W * (heads * hd / 4) < (1LL << 31), "neighborhood attention: %d x %d x %d x %d channels exceeds the 32-bit item index", B, H, W, heads * hd);
  NaGeom g{B, H, W, heads * hd, heads * hd / 4, heads, scale, lmn_div_magic(W), lmn_div_magic(H * W), lmn_div_magic(heads * hd / 4)};
  if (gen || K != 3) {
    hipStream_t st = (hipStream_t)stream;
    const int grid = na_grid((int64_t)B * H * W * g.C4);
    if (g_lmn_prof_on) lmn_prof_cost(2.0 * 2 * K * K * (double)B * H * W * g.C, (act_dtype == LMN_BF16 ? 2.0 : 4.0) * 4 * (double)B * H * W * g.C);
#define LMN_NAG(HDV) LMN_LAUNCH((na_fwd_gen_kernel<HDV, T>), dim3(grid), dim3(256), 0, st, (const T*)qkv, rpb, (T*)out, g, K)
    LMN_ACT_DISPATCH(act_dtype, switch (hd) {
      case 1: LMN_NAG(1); break;
      case 2: LMN_NAG(2); break;
      case 4: LMN_NAG(4); break;
      case 8: LMN_NAG(8); break;
      default: LMN_NAG(16); break;
    });
#undef LMN_NAG
    return lmn_launch_status("na_fwd");
  }
  // channel chunk: whole heads and whole lane pairs; tile 16x16 while the k/v window fits 64 KB of LDS, else 8x8
  int cch = g.C <= 48 ? g.C : 48;
  while (g.C % cch || cch % (hd > 4 ? hd : 4)) cch -= 4;
  LMN_REQUIRE(cch >= 4 && cch % hd == 0, "na_fwd: channel chunk %d for hd %d", cch, hd);
  const int TS = (18 * 18 * 2 * cch * 4 <= 48 * 1024) ? 16 : 8;
  const int tx = lmn_cdiv(W, TS), ty = lmn_cdiv(H, TS), chunks = g.C / cch;
  const int grid = tx * ty * chunks * B;
  const size_t sh = (size_t)(TS + 2) * (TS + 2) * 2 * cch * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  if (g_lmn_prof_on) lmn_prof_cost(2.0 * 2 * 9 * (double)B * H * W * g.C, (act_dtype == LMN_BF16 ? 2.0 : 4.0) * 4 * (double)B * H * W * g.C);
#define LMN_NAF(HDV) LMN_LAUNCH((na_fwd_kernel<HDV, T>), dim3(grid), dim3(256), sh, st, (const T*)qkv, rpb, (T*)out, g, TS, cch, tx, tx * ty, chunks, \
                                 lmn_div_magic(cch / 4), lmn_div_magic(TS + 2), lmn_div_magic(TS))
  LMN_ACT_DISPATCH(act_dtype, switch (hd) {
    case 1: LMN_NAF(1); break;
    case 2: LMN_NAF(2); break;
    case 4: LMN_NAF(4); break;
    case 8: LMN_NAF(8); break;
    default: LMN_NAF(16); break;
  });
#undef LMN_NAF
  return lmn_launch_status("na_fwd");
}

int lmn_na_bwd(const void* qkv_, const float* rpb, const void* dout_, void* dqkv_, float* drpb, float* stat, int B,
               int H, int W, int heads, int hd, int K, float scale, int act_dtype, lmn_stream_t stream) {
  LMN_REC(lmn_na_bwd(qkv_, rpb, dout_, dqkv_, drpb, stat, B, H, W, heads, hd, K, scale, act_dtype, stream));
  LMN_REQUIRE_DT(act_dtype, "na_bwd");
  const void *qkv = qkv_, *dout = dout_;
  void* dqkv = dqkv_;
  LMN_REQUIRE(qkv && rpb && dout && dqkv && drpb && stat, "na_bwd: null pointer");
  const bool gen = K < 0;
  if (gen) K = -K;
  LMN_REQUIRE(K >= 3 && K <= 13 && (K & 1), "na_bwd: window %d (odd, 3..13)", K);
  LMN_REQUIRE(B > 0 && H >= K && W >= K, "na_bwd: feature map %dx%d smaller than the %dx%d window", H, W, K, K);
  LMN_REQUIRE(hd == 1 || hd == 2 || hd == 4 || hd == 8 || hd == 16, "na_bwd: head_dim %d not in {1,2,4,8,16}", hd);
  LMN_REQUIRE((heads * hd) % 4 == 0 && heads <= 16, "na_bwd: heads=%d hd=%d (C %% 4 == 0, at most 16 heads)", heads, hd);
  LMN_REQUIRE((int64_t)B * H * W * (heads * hd / 4) < (1LL << 31), "neighborhood attention: %d x %d x %d x %d channels exceeds the 32-bit item index", B, H, W, heads * hd);
  NaGeom g{B, H, W, heads * hd, heads * hd / 4, heads, scale, lmn_div_magic(W), lmn_div_magic(H * W), lmn_div_magic(heads * hd / 4)};
  const int grid = na_grid((int64_t)B * H * W * g.C4);
  // query pass: persistent blocks -- every block ends with one global atomic per (head, bias entry), and 1452 blocks x 300 entries
  // serialised on 300 addresses were most of the pass on the small maps (level 3: 72 -> 51 us for the backward pair at 512 blocks)
  static const int qcap = getenv("LMN_NA_QGRID") ? atoi(getenv("LMN_NA_QGRID")) : 512;
  const int gq = grid > qcap ? qcap : grid;
  hipStream_t st = (hipStream_t)stream;
  const size_t sh = (6 * heads * 25 + 256 * 36) * sizeof(float);   // bins | parked interior bins | bias table | border bins per wave [4]
  // deterministic mode: the bias-table gradient of every block into its own slot, folded in fixed order after the query pass
  const int ntabK = heads * (2 * K - 1) * (2 * K - 1);
  float* dslot = drpb;
  int dn = 0;
  auto det_prep = [&](int nblk) -> int {
    if (!g_lmn_det) return 0;
    lmn_det_begin(st);
    dslot = lmn_det_slots(st, (size_t)nblk * ntabK);
    LMN_REQUIRE(dslot, "na_bwd: deterministic mode: no scratch");
    dn = nblk;
    return 0;
  };
  if (gen || K != 3) {
    const size_t gsh = (size_t)4 * heads * (2 * K - 1) * (2 * K - 1) * sizeof(float);
    if (int rc = det_prep(gq)) return rc;
#define LMN_NAG(HDV)                                                                                                 \
  do {                                                                                                               \
    if (g_lmn_prof_on) lmn_prof_cost(2.0 * 3 * K * K * (double)B * H * W * g.C, (act_dtype == LMN_BF16 ? 2.0 : 4.0) * 5 * (double)B * H * W * g.C); \
    LMN_LAUNCH((na_bwd_q_gen_kernel<HDV, T>), dim3(gq), dim3(256), gsh, st, (const T*)qkv, rpb, (const T*)dout, (T*)dqkv, dslot, stat, g, K, g_lmn_det); \
    if (g_lmn_det) lmn_det_sum(st, dslot, dn, ntabK, drpb);                                                          \
    if (g_lmn_prof_on) lmn_prof_cost(2.0 * 3 * K * K * (double)B * H * W * g.C, (act_dtype == LMN_BF16 ? 2.0 : 4.0) * 2 * (double)B * H * W * g.C); \
    LMN_LAUNCH((na_bwd_kv_gen_kernel<HDV, T>), dim3(grid), dim3(256), 0, st, (const T*)qkv, rpb, (const T*)dout, (T*)dqkv, stat, g, K);          \
  } while (0)
    LMN_ACT_DISPATCH(act_dtype, switch (hd) {
      case 1: LMN_NAG(1); break;
      case 2: LMN_NAG(2); break;
      case 4: LMN_NAG(4); break;
      case 8: LMN_NAG(8); break;
      default: LMN_NAG(16); break;
    });
#undef LMN_NAG
    return lmn_launch_status("na_bwd");
  }
  // algorithmic cost (7*C per pixel for the pair): query pass reads q,k,v,dout and writes dq; key pass writes dk,dv
#define NA_COST_Q if (g_lmn_prof_on) lmn_prof_cost(2.0 * 3 * 9 * (double)B * H * W * g.C, (act_dtype == LMN_BF16 ? 2.0 : 4.0) * 5 * (double)B * H * W * g.C)
#define NA_COST_KV if (g_lmn_prof_on) lmn_prof_cost(2.0 * 3 * 9 * (double)B * H * W * g.C, (act_dtype == LMN_BF16 ? 2.0 : 4.0) * 2 * (double)B * H * W * g.C)
#define LMN_NA(HDV)                                                                                                  \
  do {                                                                                                               \
    if (int rc = det_prep(gq)) return rc;                                                                            \
    NA_COST_Q; LMN_LAUNCH((na_bwd_q_kernel<HDV, T>), dim3(gq), dim3(256), sh, st, (const T*)qkv, rpb, (const T*)dout, (T*)dqkv, dslot, stat, g, g_lmn_det); \
    if (g_lmn_det) lmn_det_sum(st, dslot, dn, ntabK, drpb);                                                          \
    NA_COST_KV; LMN_LAUNCH((na_bwd_kv_kernel<HDV, T>), dim3(grid), dim3(256), 0, st, (const T*)qkv, rpb, (const T*)dout, (T*)dqkv, stat, g);        \
  } while (0)
  // C <= 24 (levels 0-1): query pass with the k/v window in LDS; tile sizes are whole multiples of the 256/C4 pixels a
  // block handles per iteration (15x17 = 3 x 85, 12x14 = 4 x 42)
#define LMN_NAT(HDV)                                                                                                 \
  do {                                                                                                               \
    const int TH = g.C4 == 3 ? 15 : 12, TW = g.C4 == 3 ? 17 : 14;                                                    \
    const int tx = lmn_cdiv(W, TW), ty = lmn_cdiv(H, TH), total = tx * ty * B;                                       \
    const int ntab = (heads * 25 + 3) & ~3;                                                                          \
    const int win = (TH + 2) * (TW + 2) * 2 * g.C;                                                                   \
    const size_t tsh = (size_t)(6 * ntab + (win > 256 * 36 ? win : 256 * 36)) * sizeof(float);                       \
    const int gt = total > 1024 ? 1024 : total;                                                                      \
    if (int rc = det_prep(gt)) return rc;                                                                            \
    NA_COST_Q; LMN_LAUNCH((na_bwd_q_tile_kernel<HDV, T>), dim3(gt), dim3(256), tsh, st, (const T*)qkv, rpb, (const T*)dout, (T*)dqkv, dslot, stat, g, TH, TW, \
                       tx, tx * ty, total, g_lmn_det);                                                               \
    if (g_lmn_det) lmn_det_sum(st, dslot, dn, ntabK, drpb);                                                          \
    if (g.C4 == 3) {  /* key pass in LDS form only at C = 12 (measured: 208 -> 181 us; at C = 24 it is slower, 71 -> 97 us) */ \
      const int KT = 16;  /* key tile: window (KT+2)^2 x (C + 2*heads) floats of LDS */                              \
      const int kx = lmn_cdiv(W, KT), ky = lmn_cdiv(H, KT);                                                          \
      const size_t ksh = (size_t)(KT + 2) * (KT + 2) * (g.C + 2 * heads) * sizeof(float);                            \
      NA_COST_KV; LMN_LAUNCH((na_bwd_kv_tile_kernel<HDV, T>), dim3(kx * ky * B), dim3(256), ksh, st, (const T*)qkv, rpb, (const T*)dout, (T*)dqkv, stat, g, KT, \
                         kx, kx * ky);                                                                               \
    } else {                                                                                                         \
      NA_COST_KV; LMN_LAUNCH((na_bwd_kv_kernel<HDV, T>), dim3(grid), dim3(256), 0, st, (const T*)qkv, rpb, (const T*)dout, (T*)dqkv, stat, g);      \
    }                                                                                                                \
  } while (0)
  const bool tiled = (g.C4 == 3 || g.C4 == 6) && hd <= 2 && heads % 2 == 0 && H >= 16 && W >= 16;
  // ---- one-pass form (na_bwd_fused_kernel): 15 x 17 tiles = 3 rounds of 85 pixel slots, statistics region 17 x 19 (+1 where a tile
  // ends two short of the border), k / v window +2.  LMN_NA_FUSED=0: the two-pass tile kernels (A/B runs)
  static const int fused_env = getenv("LMN_NA_FUSED") ? atoi(getenv("LMN_NA_FUSED")) : 1;
  if (tiled && fused_env && heads == 12) {   // (the kernel's channel-quad count is a template argument: 12 heads x head_dim / 4)
    const int TH = 15, TW = 17, NT = g.C4 == 3 ? 256 : 512, nwave = NT / 64;
    const int tx = lmn_cdiv(W, TW), ty = lmn_cdiv(H, TH), total = tx * ty * B;
    const int ey = (H - 2 >= TH && (H - 2) % TH == 0) ? 1 : 0, ex = (W - 2 >= TW && (W - 2) % TW == 0) ? 1 : 0;
    const int SHm = (TH + 2 + ey < H ? TH + 2 + ey : H), SWm = (TW + 2 + ex < W ? TW + 2 + ex : W);
    const int KHm = (SHm + 2 < H ? SHm + 2 : H), KWm = (SWm + 2 < W ? SWm + 2 : W);
    const int ntab = (heads * 25 + 3) & ~3;
    const int st_floats = SHm * SWm * 2 * heads;
    int wfl = KHm * KWm * 2 * g.C;
    if (wfl < SHm * SWm * 2 * g.C) wfl = SHm * SWm * 2 * g.C;
    if (wfl < NT * 36) wfl = NT * 36;
    const size_t fsh = (size_t)(ntab * (2 + nwave) + st_floats + wfl) * sizeof(float);
    const int cap = g.C4 == 3 ? 512 : 256;   // resident blocks: two per CU at C = 12 (76 KB each), one of 512 threads at C = 24 (120 KB)
    const int gf = total > cap ? cap : total;
    if (fsh <= 160 * 1024) {
      if (int rc = det_prep(gf)) return rc;
      if (g_lmn_prof_on) lmn_prof_cost(2.0 * 6 * 9 * (double)B * H * W * g.C, (act_dtype == LMN_BF16 ? 2.0 : 4.0) * 7 * (double)B * H * W * g.C);
#define LMN_NAF(HDV)                                                                                                                \
  do {                                                                                                                              \
    if (fsh > 64 * 1024) (void)hipFuncSetAttribute((const void*)na_bwd_fused_kernel<HDV, 3 * HDV, T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fsh); \
    LMN_LAUNCH((na_bwd_fused_kernel<HDV, 3 * HDV, T>), dim3(gf), dim3(NT), fsh, st, (const T*)qkv, rpb, (const T*)dout, (T*)dqkv, dslot, g, TH, TW, tx, \
               tx * ty, total, st_floats, g_lmn_det);                                                                               \
  } while (0)
      LMN_ACT_DISPATCH(act_dtype, if (hd == 1) LMN_NAF(1); else LMN_NAF(2););
#undef LMN_NAF
      if (g_lmn_det) lmn_det_sum(st, dslot, dn, ntabK, drpb);
      return lmn_launch_status("na_bwd");
    }
  }
  LMN_ACT_DISPATCH(act_dtype, switch (hd) {
    case 1: if (tiled) LMN_NAT(1); else LMN_NA(1); break;
    case 2: if (tiled) LMN_NAT(2); else LMN_NA(2); break;
    case 4: LMN_NA(4); break;
    case 8: LMN_NA(8); break;
    default: LMN_NA(16); break;
  });
#undef LMN_NA
#undef LMN_NAT
  return lmn_launch_status("na_bwd");
}

}  // extern "C"
