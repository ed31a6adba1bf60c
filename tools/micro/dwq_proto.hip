// Prototype of the wave-private depthwise kernels on the Q64 layout (chunk-planar channel quads):
//   element (pixel gp, channel c) of an E-wide tensor lives at (gp >> 6) * 64 * E + (c >> 2) * 256 + (gp & 63) * 4 + (c & 3)
// so the 64 columns of a wave's channel pair are ONE 8-bytes-per-lane load (16-byte lane stride), no LDS transpose, no
// block barriers: a wave owns (image, row segment, 60-column strip, channel pair) and walks down the rows alone.
//   hipcc --offload-arch=gfx950 -O3 -o dwq_proto dwq_proto.hip ;  ./dwq_proto check ; ./dwq_proto bench
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __amdgpu_buffer_rsrc_t BufRsrc;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// RP4: element (row r = b*H + y, column x, channel c) at (r * (E/4) + (c >> 2)) * 4 * W + 4 * x + (c & 3)
static int g_W = 0;
static inline int64_t q64_elem(int64_t gp, int c, int E) { const int64_t r = gp / g_W, x = gp - r * g_W; return (r * (E / 4) + (c >> 2)) * 4 * g_W + 4 * x + (c & 3); }
static inline int64_t q64_size(int64_t npix, int E) { return npix * E; }

__device__ __forceinline__ BufRsrc make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ f32x2 bload2(BufRsrc r, unsigned off) { return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, 0)); }
__device__ __forceinline__ void bstore2(BufRsrc r, unsigned off, f32x2 v) { __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, (int)off, 0, 0); }

__device__ __forceinline__ float lmn_erf(float x) {
  const float t = fabsf(x);
  const float k = __builtin_amdgcn_rcpf(fmaf(0.3275911f, t, 1.0f));
  float p = fmaf(1.061405429f, k, -1.453152027f);
  p = fmaf(p, k, 1.421413741f);
  p = fmaf(p, k, -0.284496736f);
  p = fmaf(p, k, 0.254829592f);
  return copysignf(1.0f - p * k * __expf(-t * t), x);
}
__device__ __forceinline__ float lmn_gelu(float x) { return 0.5f * x * (1.0f + lmn_erf(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float lmn_dgelu(float x) {
  const float cdf = 0.5f * (1.0f + lmn_erf(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}
__device__ __forceinline__ float lmn_dhswish(float x) { return x < -3.0f ? 0.0f : (x <= 3.0f ? x * (1.0f / 3.0f) + 0.5f : 1.0f); }

__device__ __forceinline__ float dpp_wave_shr1(float a) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x138, 0xF, 0xF, true)); }
__device__ __forceinline__ float dpp_wave_shl1(float a) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x130, 0xF, 0xF, true)); }
__device__ __forceinline__ f32x2 lane_from_left(f32x2 v) { const float a0 = v.x, a1 = v.y; f32x2 r; r.x = dpp_wave_shr1(a0); r.y = dpp_wave_shr1(a1); return r; }
__device__ __forceinline__ f32x2 lane_from_right(f32x2 v) { const float a0 = v.x, a1 = v.y; f32x2 r; r.x = dpp_wave_shl1(a0); r.y = dpp_wave_shl1(a1); return r; }
// wave total of a pair: lands in lane 63
__device__ __forceinline__ f32x2 wave_total(f32x2 v) {
  float a = v.x, c = v.y;
#define DPP_ADD(CTRL)                                                                               \
  a += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), CTRL, 0xF, 0xF, true));     \
  c += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(c), CTRL, 0xF, 0xF, true));
  DPP_ADD(0x111) DPP_ADD(0x112) DPP_ADD(0x114) DPP_ADD(0x118) DPP_ADD(0x142) DPP_ADD(0x143)
#undef DPP_ADD
  return f32x2{a, c};
}

constexpr unsigned OOB = 0x80000000u;
constexpr unsigned NREC = 0x7FFFFFFFu;   // every in-tensor voffset is < one row; OOB lanes carry bit 31

// ---- value type V: f32x2 = a channel PAIR per wave (packed math, weights = SGPR pairs), float = ONE channel per wave
template <typename V> struct VT;
template <> struct VT<f32x2> { static constexpr int NC = 2; };
template <> struct VT<float> { static constexpr int NC = 1; };
__device__ __forceinline__ f32x2 vzero(f32x2) { return f32x2{0.f, 0.f}; }
__device__ __forceinline__ float vzero(float) { return 0.f; }
__device__ __forceinline__ f32x2 vload(const float* p, int c, bool ok, f32x2) { return ok ? f32x2{p[c], p[c + 1]} : f32x2{0.f, 0.f}; }
__device__ __forceinline__ float vload(const float* p, int c, bool ok, float) { return ok ? p[c] : 0.f; }
// weight tap t of channel(s) c from w[E][NT]
__device__ __forceinline__ f32x2 wload(const float* w, int c, int NT, int t, bool ok, f32x2) { return ok ? f32x2{w[c * NT + t], w[(c + 1) * NT + t]} : f32x2{0.f, 0.f}; }
__device__ __forceinline__ float wload(const float* w, int c, int NT, int t, bool ok, float) { return ok ? w[c * NT + t] : 0.f; }
__device__ __forceinline__ f32x2 med01(f32x2 t) { t.x = __builtin_amdgcn_fmed3f(t.x, 0.f, 1.f); t.y = __builtin_amdgcn_fmed3f(t.y, 0.f, 1.f); return t; }
__device__ __forceinline__ float med01(float t) { return __builtin_amdgcn_fmed3f(t, 0.f, 1.f); }
__device__ __forceinline__ f32x2 vgelu(f32x2 v) { return f32x2{lmn_gelu(v.x), lmn_gelu(v.y)}; }
__device__ __forceinline__ float vgelu(float v) { return lmn_gelu(v); }
__device__ __forceinline__ f32x2 vdgelu(f32x2 v) { return f32x2{lmn_dgelu(v.x), lmn_dgelu(v.y)}; }
__device__ __forceinline__ float vdgelu(float v) { return lmn_dgelu(v); }
__device__ __forceinline__ f32x2 vdhswish(f32x2 v) { return f32x2{lmn_dhswish(v.x), lmn_dhswish(v.y)}; }
__device__ __forceinline__ float vdhswish(float v) { return lmn_dhswish(v); }
__device__ __forceinline__ float lane_from_left(float v) { return dpp_wave_shr1(v); }
__device__ __forceinline__ float lane_from_right(float v) { return dpp_wave_shl1(v); }
__device__ __forceinline__ float wave_total(float a) {
#define DPP_ADD(CTRL) a += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), CTRL, 0xF, 0xF, true));
  DPP_ADD(0x111) DPP_ADD(0x112) DPP_ADD(0x114) DPP_ADD(0x118) DPP_ADD(0x142) DPP_ADD(0x143)
#undef DPP_ADD
  return a;
}
__device__ __forceinline__ void red_put(float* r, int base, f32x2 v) { r[base] = v.x; r[base + 1] = v.y; }
__device__ __forceinline__ void red_put(float* r, int base, float v) { r[base] = v; }
template <typename V> __device__ __forceinline__ V bloadv(BufRsrc r, unsigned voff, unsigned so);
template <> __device__ __forceinline__ f32x2 bloadv<f32x2>(BufRsrc r, unsigned voff, unsigned so) { return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)so, 0)); }
template <> __device__ __forceinline__ float bloadv<float>(BufRsrc r, unsigned voff, unsigned so) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)so, 0)); }
__device__ __forceinline__ void bstorev(BufRsrc r, unsigned voff, unsigned so, f32x2 v) { __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, (int)voff, (int)so, 0); }
__device__ __forceinline__ void bstorev(BufRsrc r, unsigned voff, unsigned so, float v) { __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)voff, (int)so, 0); }

// lanes of ONE wave exchange data through LDS: the compiler sees no per-thread alias between a lane's store and its reads of
// the neighbours' slots, so both directions need a (code-free) wave-level fence
#define WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

// x1 = Hardswish(a z + s)
template <typename V> __device__ __forceinline__ V pre2(V z, V a, V s) {
  const V x = z * a + s;
  const V t = med01(x * (1.f / 6.f) + 0.5f);
  return x * t;
}

// wave = (image, row segment, strip of QW output columns, channel pair or channel); block = 4 waves = 8 / 4 consecutive channels
struct Geo {
  int b, seg, strip, ch, ys, ye, xs;
  bool cok;
  unsigned rowb, qoff;   // bytes per image row of the tensor; byte offset of this wave's quad plane inside a row
  int r0;                // b * H
};
template <int QW, int NC>
__device__ __forceinline__ Geo decode(int E, int H, int W, int strips, int segs, int seg_rows, int chunks, int wv) {
  Geo g;
  int lid = blockIdx.x;
  const int chunk = lid % chunks; lid /= chunks;
  g.strip = lid % strips; lid /= strips;
  g.seg = lid % segs;
  g.b = lid / segs;
  g.ch = (chunk * 4 + wv) * NC;
  g.cok = g.ch < E;
  if (!g.cok) g.ch = 0;
  g.ys = g.seg * seg_rows;
  g.ye = min(g.ys + seg_rows, H);
  g.xs = g.strip * QW;
  g.rowb = (unsigned)(E * W) * 4u;
  g.qoff = (unsigned)((g.ch >> 2) * 4 * W) * 4u;
  g.r0 = g.b * H;
  return g;
}
__device__ __forceinline__ unsigned soff(const Geo& g, int iy, int H) {   // row clamped into the image (callers mask the value)
  const int y = min(max(iy, 0), H - 1);
  return (unsigned)(g.r0 + y) * g.rowb + g.qoff;
}

template <typename V> struct BranchW { V w5[25], w3[9], wv[3], wh[3]; };
template <typename V>
__device__ __forceinline__ void load_branch_w(BranchW<V>& bw, const float* w5, const float* w3, const float* wv, const float* wh, int ch, bool ok) {
#pragma unroll
  for (int t = 0; t < 25; ++t) bw.w5[t] = wload(w5, ch, 25, t, ok, V());
#pragma unroll
  for (int t = 0; t < 9; ++t) bw.w3[t] = wload(w3, ch, 9, t, ok, V());
#pragma unroll
  for (int t = 0; t < 3; ++t) { bw.wv[t] = wload(wv, ch, 3, t, ok, V()); bw.wh[t] = wload(wh, ch, 3, t, ok, V()); }
}
#define SB() __builtin_amdgcn_sched_barrier(0)

// The row exchange is software-pipelined: step j writes x1 row j+1 and reads its shifted copies (inn) while the FMAs of row j
// run on `in` (read during step j-1); two LDS rows alternate.

// ------------------------------------------------------------------------------------------------ K0: forward statistics
// stats[4][2][E] += (sum y_b, sum y_b^2) over the image; z row (ys-2+j) enters at step j, output row (ys+j-4) completes
template <typename V, int D, bool ATOM, bool OL, int WPS>
__global__ __launch_bounds__(256, WPS) void k_stats0(const float* __restrict__ z, const float* __restrict__ preA, const float* __restrict__ preS,
                                                const float* __restrict__ w5, const float* __restrict__ w3, const float* __restrict__ wvv,
                                                const float* __restrict__ whh, float* __restrict__ stats, int B, int H, int W, int E,
                                                int strips, int segs, int seg_rows, int chunks) {
  constexpr int NC = VT<V>::NC;
  __shared__ V XSa[4][2][68];
  __shared__ float red[8 * 8];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const Geo g = decode<60, NC>(E, H, W, strips, segs, seg_rows, chunks, wv);
  V* XS0 = XSa[wv][0];
  if (lane < 8) XS0[(lane >> 2) * 68 + ((lane & 3) < 2 ? (lane & 3) : 64 + (lane & 3))] = vzero(V());
  BranchW<V> bw;
  load_branch_w(bw, w5, w3, wvv, whh, g.ch, g.cok);
  if constexpr (NC == 2) {
#pragma unroll
    for (int k = 0; k < 9; ++k) asm volatile("" : "+v"(bw.w3[k]));
#pragma unroll
    for (int k = 0; k < 3; ++k) asm volatile("" : "+v"(bw.wv[k]), "+v"(bw.wh[k]));
  }
  const int cx = g.xs - 2 + lane;
  const bool col_in = cx >= 0 && cx < W && g.cok;
  const V pa = vload(preA, g.ch, g.cok, V()), ps = vload(preS, g.ch, g.cok, V());
  const float cm = col_in ? 1.f : 0.f;
  const bool ovalid = lane >= 2 && lane < 62 && cx < W && g.cok;
  const unsigned voff = col_in ? (unsigned)(cx * 4 + (g.ch & 3)) * 4u : OOB;
  const BufRsrc rz = make_rsrc(z, NREC);
  const int rows = g.ye - g.ys, nsteps = rows + 4;
  V pf[5];
#pragma unroll
  for (int d = 0; d < D; ++d) pf[d] = bloadv<V>(rz, voff, soff(g, g.ys - 2 + d, H));
  const V z2 = vzero(V());
  V a5[5], a3[5], av[5], ah[5], sum[8];
#pragma unroll
  for (int k = 0; k < 5; ++k) a5[k] = a3[k] = av[k] = ah[k] = z2;
#pragma unroll
  for (int k = 0; k < 8; ++k) sum[k] = z2;
  V in[5], inn[5];
  {  // row 0 (always a boundary or halo row of the segment: masked like a non-FAST step)
    const int iy = g.ys - 2;
    const float rm = (iy >= 0 && iy < H) ? cm : 0.f;
    const V x1 = pre2(pf[0], pa, ps) * rm;
    pf[D % 5] = bloadv<V>(rz, voff, soff(g, g.ys - 2 + D, H));
    V* XS = XS0;
    XS[lane + 2] = x1; WAVE_SYNC();
    in[0] = XS[lane]; in[1] = XS[lane + 1]; in[2] = x1; in[3] = XS[lane + 3]; in[4] = XS[lane + 4];
  }
  // step j: FMAs of row j on `in`; exchange of row j+1 -> inn
#define STEP(P, FAST)                                                                                              \
  {                                                                                                                \
    const int j = j0 + P;                                                                                          \
    V x1n = pre2(pf[(P + 1) % 5], pa, ps) * cm;                                                                    \
    if (!(FAST)) { const int iy = g.ys - 1 + j; const float rm = (iy >= 0 && iy < H) ? 1.f : 0.f; x1n *= rm; }     \
    pf[(P + 1 + D) % 5] = bloadv<V>(rz, voff, soff(g, g.ys - 1 + j + D, H));                                       \
    V* XS = XS0 + ((P + 1) & 1) * 68;                                                                              \
    WAVE_SYNC();                                                                                                   \
    XS[lane + 2] = x1n;                                                                                            \
    WAVE_SYNC();                                                                                                   \
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
    constexpr int DD = (P + 1) % 5;                                                                                \
    if ((FAST) || (j >= 4 && j < nsteps)) {                                                                        \
      const V y5 = a5[DD], y3 = a3[DD], yv = av[DD], yh = ah[DD];                                                  \
      sum[0] += y5; sum[1] += y3; sum[2] += yv; sum[3] += yh;                                                      \
      sum[4] += y5 * y5; sum[5] += y3 * y3; sum[6] += yv * yv; sum[7] += yh * yh;                                  \
    }                                                                                                              \
    _Pragma("unroll") for (int d = 0; d < 5; ++d) in[d] = inn[d];                                                  \
    SB();                                                                                                          \
  }
  int j0 = 0;
  if (OL) {
    for (; j0 < nsteps; j0 += 5) { STEP(0, false) STEP(1, false) STEP(2, false) STEP(3, false) STEP(4, false) }
  } else {
  const int jfe = min(nsteps, H - g.ys + 1) - 5;   // last batch start whose rows j+1 are all inside the image
  for (; j0 < nsteps && j0 < 5; j0 += 5) { STEP(0, false) STEP(1, false) STEP(2, false) STEP(3, false) STEP(4, false) }
  for (; j0 <= jfe; j0 += 5) { STEP(0, true) STEP(1, true) STEP(2, true) STEP(3, true) STEP(4, true) }
  for (; j0 < nsteps; j0 += 5) { STEP(0, false) STEP(1, false) STEP(2, false) STEP(3, false) STEP(4, false) }
  }
#undef STEP
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    V v = ovalid ? sum[k] : z2;
    v = wave_total(v);
    if (lane == 63) red_put(red, k * 4 * NC + wv * NC, v);
  }
  __syncthreads();
  if (ATOM && tid < 8 * 4 * NC) {
    const int k = tid / (4 * NC), c8 = tid - k * 4 * NC;
    const int row = (k & 3) * 2 + (k >> 2);
    const int e = (blockIdx.x % chunks) * 4 * NC + c8;
    if (e < E) atomicAdd(stats + row * E + e, red[tid]);
  }
}

// ------------------------------------------------------------------------------------------------ K1: forward
// pre = merged 5x5 (keff) + beff -> store; gsum[b][e] += sum GELU(pre)
template <typename V, int D, bool OL, int WPS>
__global__ __launch_bounds__(256, WPS) void k_fwd(const float* __restrict__ z, float* __restrict__ pre, float* __restrict__ gsum,
                                             const float* __restrict__ preA, const float* __restrict__ preS, const float* __restrict__ keff,
                                             const float* __restrict__ beff, int B, int H, int W, int E, int strips, int segs, int seg_rows,
                                             int chunks) {
  constexpr int NC = VT<V>::NC;
  __shared__ V XSa[4][2][68];
  __shared__ float red[8];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const Geo g = decode<60, NC>(E, H, W, strips, segs, seg_rows, chunks, wv);
  V* XS0 = XSa[wv][0];
  if (lane < 8) XS0[(lane >> 2) * 68 + ((lane & 3) < 2 ? (lane & 3) : 64 + (lane & 3))] = vzero(V());
  V w[25];
#pragma unroll
  for (int t = 0; t < 25; ++t) w[t] = wload(keff, g.ch, 25, t, g.cok, V());
  const V bias = vload(beff, g.ch, g.cok, V());
  const int cx = g.xs - 2 + lane;
  const bool col_in = cx >= 0 && cx < W && g.cok;
  const V pa = vload(preA, g.ch, g.cok, V()), ps = vload(preS, g.ch, g.cok, V());
  const float cm = col_in ? 1.f : 0.f;
  const bool ovalid = lane >= 2 && lane < 62 && cx < W && g.cok;
  const unsigned voff = col_in ? (unsigned)(cx * 4 + (g.ch & 3)) * 4u : OOB;
  const unsigned vst = ovalid ? voff : OOB;
  const BufRsrc rz = make_rsrc(z, NREC), ro = make_rsrc(pre, NREC);
  const int rows = g.ye - g.ys, nsteps = rows + 4;
  V pf[5];
#pragma unroll
  for (int d = 0; d < D; ++d) pf[d] = bloadv<V>(rz, voff, soff(g, g.ys - 2 + d, H));
  V acc[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) acc[k] = vzero(V());
  V gs = vzero(V());
  V in[5], inn[5];
  {
    const int iy = g.ys - 2;
    const float rm = (iy >= 0 && iy < H) ? cm : 0.f;
    const V x1 = pre2(pf[0], pa, ps) * rm;
    pf[D % 5] = bloadv<V>(rz, voff, soff(g, g.ys - 2 + D, H));
    V* XS = XS0;
    XS[lane + 2] = x1; WAVE_SYNC();
    in[0] = XS[lane]; in[1] = XS[lane + 1]; in[2] = x1; in[3] = XS[lane + 3]; in[4] = XS[lane + 4];
  }
#define STEP(P, FAST)                                                                                              \
  {                                                                                                                \
    const int j = j0 + P;                                                                                          \
    V x1n = pre2(pf[(P + 1) % 5], pa, ps) * cm;                                                                    \
    if (!(FAST)) { const int iy = g.ys - 1 + j; const float rm = (iy >= 0 && iy < H) ? 1.f : 0.f; x1n *= rm; }     \
    pf[(P + 1 + D) % 5] = bloadv<V>(rz, voff, soff(g, g.ys - 1 + j + D, H));                                       \
    V* XS = XS0 + ((P + 1) & 1) * 68;                                                                              \
    WAVE_SYNC();                                                                                                   \
    XS[lane + 2] = x1n;                                                                                            \
    WAVE_SYNC();                                                                                                   \
    inn[0] = XS[lane]; inn[1] = XS[lane + 1]; inn[2] = x1n; inn[3] = XS[lane + 3]; inn[4] = XS[lane + 4];          \
    _Pragma("unroll") for (int d = 0; d < 5; ++d) {                                                                \
      if (d == 0) acc[P] = w[0] * in[0];                                                                           \
      else acc[P] += w[d] * in[d];                                                                                 \
      _Pragma("unroll") for (int ky = 1; ky < 5; ++ky) acc[(P - ky + 5) % 5] += w[ky * 5 + d] * in[d];             \
    }                                                                                                              \
    constexpr int DD = (P + 1) % 5;                                                                                \
    if ((FAST) || (j >= 4 && j < nsteps)) {                                                                        \
      const V pv = acc[DD] + bias;                                                                                 \
      bstorev(ro, vst, soff(g, g.ys + j - 4, H), pv);                                                              \
      gs += vgelu(pv);                                                                                             \
    }                                                                                                              \
    _Pragma("unroll") for (int d = 0; d < 5; ++d) in[d] = inn[d];                                                  \
    SB();                                                                                                          \
  }
  int j0 = 0;
  if (OL) {
    for (; j0 < nsteps; j0 += 5) { STEP(0, false) STEP(1, false) STEP(2, false) STEP(3, false) STEP(4, false) }
  } else {
  const int jfe = min(nsteps, H - g.ys + 1) - 5;
  for (; j0 < nsteps && j0 < 5; j0 += 5) { STEP(0, false) STEP(1, false) STEP(2, false) STEP(3, false) STEP(4, false) }
  for (; j0 <= jfe; j0 += 5) { STEP(0, true) STEP(1, true) STEP(2, true) STEP(3, true) STEP(4, true) }
  for (; j0 < nsteps; j0 += 5) { STEP(0, false) STEP(1, false) STEP(2, false) STEP(3, false) STEP(4, false) }
  }
#undef STEP
  {
    V v = ovalid ? gs : vzero(V());
    v = wave_total(v);
    if (lane == 63) red_put(red, wv * NC, v);
  }
  __syncthreads();
  if (tid < 4 * NC) {
    const int e = (blockIdx.x % chunks) * 4 * NC + tid;
    if (e < E) atomicAdd(gsum + g.b * E + e, red[tid]);
  }
}

// ------------------------------------------------------------------------------------------------ K2: backward statistics
// dpre = (u * s + dm) * gelu'(pre) -> store;  bst[0][e] += sum dpre;  G[t] = sum_p dpre[p] x1[p + t] contracted with the four
// kernels at the end: bst[1 + b][e] += <w_b, G>
template <typename V, int D, bool OL, int WPS>
__global__ __launch_bounds__(256, WPS) void k_stats1(const float* __restrict__ z, const float* __restrict__ pre, const float* __restrict__ u,
                                                const float* __restrict__ sgate, const float* __restrict__ dm, float* __restrict__ dpre,
                                                const float* __restrict__ preA, const float* __restrict__ preS, const float* __restrict__ w5,
                                                const float* __restrict__ w3, const float* __restrict__ wvv, const float* __restrict__ whh,
                                                float* __restrict__ bst, int B, int H, int W, int E, int strips, int segs, int seg_rows,
                                                int chunks) {
  constexpr int NC = VT<V>::NC;
  __shared__ V XSa[4][2][68];
  __shared__ float red[5 * 8];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const Geo g = decode<60, NC>(E, H, W, strips, segs, seg_rows, chunks, wv);
  V* XS0 = XSa[wv][0];
  if (lane < 8) XS0[(lane >> 2) * 68 + ((lane & 3) < 2 ? (lane & 3) : 64 + (lane & 3))] = vzero(V());
  const int cx = g.xs - 2 + lane;
  const bool col_in = cx >= 0 && cx < W && g.cok;
  const V pa = vload(preA, g.ch, g.cok, V()), ps = vload(preS, g.ch, g.cok, V());
  const float cm = col_in ? 1.f : 0.f;
  const bool ovalid = lane >= 2 && lane < 62 && cx < W && g.cok;
  const float om = ovalid ? 1.f : 0.f;
  const V sv = vload(sgate, g.b * E + g.ch, g.cok, V()), dv = vload(dm, g.b * E + g.ch, g.cok, V());
  const unsigned voff = col_in ? (unsigned)(cx * 4 + (g.ch & 3)) * 4u : OOB;
  const unsigned vst = ovalid ? voff : OOB;
  const BufRsrc rz = make_rsrc(z, NREC), rp = make_rsrc(pre, NREC), ru = make_rsrc(u, NREC), ro = make_rsrc(dpre, NREC);
  const int rows = g.ye - g.ys, nsteps = rows + 4;
  V pf[5], pp[5], pu[5];
#pragma unroll
  for (int d = 0; d < D; ++d) {
    pf[d] = bloadv<V>(rz, voff, soff(g, g.ys - 2 + d, H));
    const unsigned so = soff(g, g.ys + d, H);
    pp[d] = bloadv<V>(rp, vst, so);
    pu[d] = bloadv<V>(ru, vst, so);
  }
  const V z2 = vzero(V());
  V G[25], hist[5], sum0 = z2;
#pragma unroll
  for (int k = 0; k < 25; ++k) G[k] = z2;
#pragma unroll
  for (int k = 0; k < 5; ++k) hist[k] = z2;
  V in[5], inn[5];
  {
    const int iy = g.ys - 2;
    const float rm = (iy >= 0 && iy < H) ? cm : 0.f;
    const V x1 = pre2(pf[0], pa, ps) * rm;
    pf[D % 5] = bloadv<V>(rz, voff, soff(g, g.ys - 2 + D, H));
    V* XS = XS0;
    XS[lane + 2] = x1; WAVE_SYNC();
    in[0] = XS[lane]; in[1] = XS[lane + 1]; in[2] = x1; in[3] = XS[lane + 3]; in[4] = XS[lane + 4];
  }
#define STEP(P, FAST)                                                                                              \
  {                                                                                                                \
    const int j = j0 + P;                                                                                          \
    V x1n = pre2(pf[(P + 1) % 5], pa, ps) * cm;                                                                    \
    const V pv = pp[P], uv = pu[P];                                                                                \
    if (!(FAST)) { const int iy = g.ys - 1 + j; const float rm = (iy >= 0 && iy < H) ? 1.f : 0.f; x1n *= rm; }     \
    pf[(P + 1 + D) % 5] = bloadv<V>(rz, voff, soff(g, g.ys - 1 + j + D, H));                                       \
    { const unsigned so = soff(g, g.ys + j + D, H); pp[(P + D) % 5] = bloadv<V>(rp, vst, so); pu[(P + D) % 5] = bloadv<V>(ru, vst, so); } \
    V* XS = XS0 + ((P + 1) & 1) * 68;                                                                              \
    WAVE_SYNC();                                                                                                   \
    XS[lane + 2] = x1n;                                                                                            \
    WAVE_SYNC();                                                                                                   \
    inn[0] = XS[lane]; inn[1] = XS[lane + 1]; inn[2] = x1n; inn[3] = XS[lane + 3]; inn[4] = XS[lane + 4];          \
    V d = (uv * sv + dv) * vdgelu(pv) * om;                                                                        \
    if (!(FAST)) d *= (j < rows) ? 1.f : 0.f;                                                                      \
    if ((FAST) || j < rows) bstorev(ro, vst, soff(g, g.ys + j, H), d);                                             \
    sum0 += d;                                                                                                     \
    hist[P] = d;                                                                                                   \
    _Pragma("unroll") for (int ky = 0; ky < 5; ++ky)                                                               \
      _Pragma("unroll") for (int kx = 0; kx < 5; ++kx) G[ky * 5 + kx] += hist[(P - ky + 5) % 5] * in[kx];          \
    _Pragma("unroll") for (int dd = 0; dd < 5; ++dd) in[dd] = inn[dd];                                             \
    SB();                                                                                                          \
  }
  int j0 = 0;
  if (OL) {
    for (; j0 < nsteps; j0 += 5) { STEP(0, false) STEP(1, false) STEP(2, false) STEP(3, false) STEP(4, false) }
  } else {
  const int jfb = g.ys >= 2 ? 0 : 5, jfe = min(rows, H - g.ys + 1) - 5;
  for (; j0 < nsteps && j0 < jfb; j0 += 5) { STEP(0, false) STEP(1, false) STEP(2, false) STEP(3, false) STEP(4, false) }
  for (; j0 <= jfe; j0 += 5) { STEP(0, true) STEP(1, true) STEP(2, true) STEP(3, true) STEP(4, true) }
  for (; j0 < nsteps; j0 += 5) { STEP(0, false) STEP(1, false) STEP(2, false) STEP(3, false) STEP(4, false) }
  }
#undef STEP
  V sum[5];
  sum[0] = sum0;
  sum[1] = sum[2] = sum[3] = sum[4] = z2;
  {
    BranchW<V> bw;
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
    V v = wave_total(sum[k]);   // (invalid lanes carry d = 0)
    if (lane == 63) red_put(red, k * 4 * NC + wv * NC, v);
  }
  __syncthreads();
  if (tid < 5 * 4 * NC) {
    const int k = tid / (4 * NC), c8 = tid - k * 4 * NC;
    const int e = (blockIdx.x % chunks) * 4 * NC + c8;
    if (e < E) atomicAdd(bst + k * E + e, red[tid]);
  }
}

// ------------------------------------------------------------------------------------------------ K3: backward
// f_b = cA_b dpre + cC_b y_b + cD_b (inside the image); dx1 = sum_b corr^T(f_b, w_b); dh = dx1 * Hardswish'(A z + shift) -> store;
// hst[2][E] += (sum dh, sum dh z); dW_b[t] += sum_p f_b[p] x1[p + t].
// lane l = column xs-2+l; the x1 row in LDS has 68 entries (columns xs-4 .. xs+63: lanes 0..3 load the four halo columns).
// step j: z row (ys-4+j) enters; f row (ys-6+j); dx row (ys-8+j) completes; dW products of x1 row (ys-8+j).
template <typename V> struct SwState {
  V a5[5], a3[5], av[5], ah[5];
  V h5[5], h3[5], hv[5], hh[5];
  V dxa[5];
  V g5[25], g3[9], gv[3], gh[3];
};
template <typename V> struct Coef { V a[4], c[4], d[4]; };

template <int P, int PART, bool FAST, bool CL, typename V>
__device__ __forceinline__ void sw_step(SwState<V>& S, const BranchW<V>& bw, const Coef<V>& CO, const float* CFS, float cm, const V* XSw, int lane,
                                        V x1, V dp, bool frow_in, bool own, bool dw_ok) {
  // ---- shifted copies of this step's x1 row, then the weight-gradient products that only need OLD f rows (x1 row j-4 against f
  //      rows j-3 .. j-6): they cover the latency of the reads
  V in[5];
  const V* xr = XSw + P * 68 + lane;
  in[0] = xr[0]; in[1] = xr[1]; in[2] = x1; in[3] = xr[3]; in[4] = xr[4];
  const V* x2 = XSw + ((P + 1) % 5) * 68 + lane;
  if (PART != 1 && (FAST || dw_ok)) {
    V i2[5];
#pragma unroll
    for (int d = 0; d < 5; ++d) i2[d] = x2[d];
#pragma unroll
    for (int ky = 1; ky < 5; ++ky)
#pragma unroll
      for (int kx = 0; kx < 5; ++kx) S.g5[ky * 5 + kx] += S.h5[(P + 8 - ky) % 5] * i2[kx];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) S.g3[ky * 3 + kx] += S.h3[(P + 7 - ky) % 5] * i2[1 + kx];
      S.gv[ky] += S.hv[(P + 7 - ky) % 5] * i2[2];
    }
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) S.gh[kx] += S.hh[(P + 1) % 5] * i2[1 + kx];
  }
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
  constexpr int Q = (P + 3) % 5;
  V f5, f3, fv, fh;
  if constexpr (CL) {
    const float* cfp = CFS;
    asm volatile("" : "+v"(cfp));
    const V* cf = reinterpret_cast<const V*>(cfp);
    f5 = (cf[1] * S.a5[Q] + (cf[0] * dp + cf[2])) * cm;
    f3 = (cf[4] * S.a3[Q] + (cf[3] * dp + cf[5])) * cm;
    fv = (cf[7] * S.av[Q] + (cf[6] * dp + cf[8])) * cm;
    fh = (cf[10] * S.ah[Q] + (cf[9] * dp + cf[11])) * cm;
  } else {   // coefficients in SGPRs
    f5 = (CO.c[0] * S.a5[Q] + (CO.a[0] * dp + CO.d[0])) * cm;
    f3 = (CO.c[1] * S.a3[Q] + (CO.a[1] * dp + CO.d[1])) * cm;
    fv = (CO.c[2] * S.av[Q] + (CO.a[2] * dp + CO.d[2])) * cm;
    fh = (CO.c[3] * S.ah[Q] + (CO.a[3] * dp + CO.d[3])) * cm;
  }
  if constexpr (FAST) {
    S.h5[Q] = f5; S.h3[Q] = f3; S.hv[Q] = fv; S.hh[Q] = fh;
  } else {
    const float mf = frow_in ? 1.f : 0.f, mo = own ? mf : 0.f;
    f5 *= mf; f3 *= mf; fv *= mf; fh *= mf;
    S.h5[Q] = f5 * mo; S.h3[Q] = f3 * mo; S.hv[Q] = fv * mo; S.hh[Q] = fh * mo;
  }
  if (PART != 1 && (FAST || dw_ok)) {   // f row j-2 (this step's) against x1 row j-4: kernel row 0 of the 5x5 gradient
#pragma unroll
    for (int kx = 0; kx < 5; ++kx) S.g5[kx] += S.h5[Q] * x2[kx];
  }
  if constexpr (PART != 2) {
    V sh[5];
    sh[2] = f5;
    sh[1] = lane_from_right(f5);
    sh[0] = lane_from_right(sh[1]);
    sh[3] = lane_from_left(f5);
    sh[4] = lane_from_left(sh[3]);
    V s3[3];
    s3[1] = f3;
    s3[0] = lane_from_right(f3);
    s3[2] = lane_from_left(f3);
    const V hr = lane_from_right(fh), hl = lane_from_left(fh);
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
}

template <typename V, int PART, int D, int D2, bool HALO, int WPS, bool ONELOOP>
__global__ __launch_bounds__(256, WPS) void k_bwd(const float* __restrict__ z, const float* __restrict__ dpre, float* __restrict__ dh,
                                                const float* __restrict__ preA, const float* __restrict__ preS, const float* __restrict__ w5,
                                                const float* __restrict__ w3, const float* __restrict__ wvv, const float* __restrict__ whh,
                                                const float* __restrict__ cA, const float* __restrict__ cC, const float* __restrict__ cD,
                                                float* __restrict__ dw5, float* __restrict__ dw3, float* __restrict__ dwv,
                                                float* __restrict__ dwh, float* __restrict__ hst, int B, int H, int W, int E, int strips,
                                                int segs, int seg_rows, int chunks) {
  constexpr int NC = VT<V>::NC;
  constexpr bool CL = NC == 2;   // channel pairs: the 24 coefficient floats in LDS (the scalar file is full of weights)
  constexpr int QW = HALO ? 60 : 56;
  __shared__ V XSa[4][5 * 68];
  __shared__ V ZSa[4][5 * 64];
  __shared__ __attribute__((aligned(16))) float coef_s[4 * 24];
  __shared__ float red[4 * 44 * 2];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const Geo g = decode<QW, NC>(E, H, W, strips, segs, seg_rows, chunks, wv);
  V* XS = XSa[wv];
  V* ZS = ZSa[wv];
  for (int i = lane; i < 5 * 68; i += 64) XS[i] = vzero(V());
  BranchW<V> bw;
  load_branch_w(bw, w5, w3, wvv, whh, g.ch, g.cok);
  Coef<V> CO;
#pragma unroll
  for (int k = 0; k < 4; ++k) { CO.a[k] = vload(cA, k * E + g.ch, g.cok, V()); CO.c[k] = vload(cC, k * E + g.ch, g.cok, V()); CO.d[k] = vload(cD, k * E + g.ch, g.cok, V()); }
  if (CL && lane < 24) {
    const int k = lane / 6, r = lane - k * 6, which = r >> 1, h = r & 1;
    const float* src = which == 0 ? cA : which == 1 ? cC : cD;
    coef_s[wv * 24 + lane] = g.cok ? src[k * E + g.ch + h] : 0.f;
  }
  const float* CFS = coef_s + wv * 24;
  // HALO: lane l = column xs-2+l, outputs on lanes 2..61, four extra x1 columns by lanes 0..3; else lane l = column xs-4+l,
  // outputs on lanes 4..59 (the x1 entries 0, 1, 66, 67 of the LDS row stay zero: they only feed y_b of lanes 0, 1, 62, 63)
  const int cx = g.xs - (HALO ? 2 : 4) + lane;
  const bool col_in = cx >= 0 && cx < W && g.cok;
  const V pa = vload(preA, g.ch, g.cok, V()), ps = vload(preS, g.ch, g.cok, V());
  const float cm = col_in ? 1.f : 0.f;
  const int hx = lane < 2 ? g.xs - 4 + lane : g.xs + 60 + lane;
  const bool hcol_in = HALO && lane < 4 && hx >= 0 && hx < W && g.cok;
  const float hm = hcol_in ? 1.f : 0.f;
  const int hidx = lane < 2 ? lane : 64 + lane;
  const bool own_col = HALO ? (lane >= 2 && lane < 62) : (lane >= 4 && lane < 60);
  const bool ovalid = own_col && cx < W && g.cok;
  const unsigned voff = col_in ? (unsigned)(cx * 4 + (g.ch & 3)) * 4u : OOB;
  const unsigned vhalo = hcol_in ? (unsigned)(hx * 4 + (g.ch & 3)) * 4u : OOB;
  const unsigned vst = ovalid ? voff : OOB;
  const BufRsrc rz = make_rsrc(z, NREC), rd = make_rsrc(dpre, NREC), ro = make_rsrc(dh, NREC);
  const int rows = g.ye - g.ys, nsteps = rows + 10, ndx = rows + 8;
  V pfz[5], pfh[5], pfd[5];
#pragma unroll
  for (int d = 0; d < D; ++d) {
    const unsigned so = soff(g, g.ys - 4 + d, H);
    pfz[d] = bloadv<V>(rz, voff, so);
    if (HALO) pfh[d] = bloadv<V>(rz, vhalo, so);
  }
#pragma unroll
  for (int d = 0; d < D2; ++d) pfd[d] = bloadv<V>(rd, voff, soff(g, g.ys - 6 + d, H));
  SwState<V> S;
  const V z2 = vzero(V());
#pragma unroll
  for (int k = 0; k < 5; ++k) S.a5[k] = S.a3[k] = S.av[k] = S.ah[k] = S.h5[k] = S.h3[k] = S.hv[k] = S.hh[k] = S.dxa[k] = z2;
#pragma unroll
  for (int k = 0; k < 25; ++k) S.g5[k] = z2;
#pragma unroll
  for (int k = 0; k < 9; ++k) S.g3[k] = z2;
#pragma unroll
  for (int k = 0; k < 3; ++k) S.gv[k] = S.gh[k] = z2;
  V hs0 = z2, hs1 = z2;
  __syncthreads();   // coef_s
#define STEP(P, FAST)                                                                                              \
  {                                                                                                                \
    const int j = j0 + P;                                                                                          \
    const V zv = pfz[P];                                                                                           \
    V x1 = pre2(zv, pa, ps) * cm;                                                                                  \
    V x1h = z2;                                                                                                    \
    if (HALO) x1h = pre2(pfh[P], pa, ps) * hm;                                                                     \
    const V dp = pfd[P];                                                                                           \
    if (!(FAST)) { const int iy = g.ys - 4 + j; const float rm = (iy >= 0 && iy < H) ? 1.f : 0.f; x1 *= rm; x1h *= rm; } \
    {                                                                                                              \
      const unsigned so = soff(g, g.ys - 4 + j + D, H);                                                            \
      pfz[(P + D) % 5] = bloadv<V>(rz, voff, so);                                                                  \
      if (HALO) pfh[(P + D) % 5] = bloadv<V>(rz, vhalo, so);                                                       \
      pfd[(P + D2) % 5] = bloadv<V>(rd, voff, soff(g, g.ys - 6 + j + D2, H));                                      \
    }                                                                                                              \
    WAVE_SYNC();                                                                                                   \
    XS[P * 68 + lane + 2] = x1;                                                                                    \
    if (HALO) { if (lane < 4) XS[P * 68 + hidx] = x1h; }                                                           \
    if (PART != 2) ZS[P * 64 + lane] = zv;                                                                         \
    WAVE_SYNC();                                                                                                   \
    const int fy = g.ys - 6 + j;                                                                                   \
    sw_step<P, PART, FAST, CL>(S, bw, CO, CFS, cm, XS, lane, x1, dp, j >= 4 && fy >= 0 && fy < H, fy >= g.ys && fy < g.ye, ONELOOP || j >= 4); \
    if (PART != 2 && (ONELOOP || (FAST) || (j >= 8 && j < ndx))) {                                                 \
      constexpr int DD = (P + 1) % 5;                                                                              \
      const V zr = ZS[DD * 64 + lane];                                                                             \
      const V hh = zr * pa + ps;                                                                                   \
      V dhv = S.dxa[DD] * vdhswish(hh);                                                                            \
      if (ONELOOP) { const bool ok = j >= 8 && j < ndx; dhv *= ok ? 1.f : 0.f; bstorev(ro, ok ? vst : OOB, soff(g, g.ys - 8 + j, H), dhv); } \
      else bstorev(ro, vst, soff(g, g.ys - 8 + j, H), dhv);                                                        \
      hs0 += dhv;                                                                                                  \
      hs1 += dhv * zr;                                                                                             \
    }                                                                                                              \
    SB();                                                                                                          \
  }
  int j0 = 0;
  if (ONELOOP) {
    for (; j0 < nsteps; j0 += 5) { STEP(0, false) STEP(1, false) STEP(2, false) STEP(3, false) STEP(4, false) }
  } else {
  // interior batch: every z row inside the image, every f row owned by the segment, every dx row stored
  const int jfe = min(rows + 6, H - g.ys + 4) - 5;
  for (; j0 < nsteps && j0 < 10; j0 += 5) { STEP(0, false) STEP(1, false) STEP(2, false) STEP(3, false) STEP(4, false) }
  for (; j0 <= jfe; j0 += 5) { STEP(0, true) STEP(1, true) STEP(2, true) STEP(3, true) STEP(4, true) }
  for (; j0 < nsteps; j0 += 5) { STEP(0, false) STEP(1, false) STEP(2, false) STEP(3, false) STEP(4, false) }
  }
#undef STEP
  // reductions: hstats (2) + 40 taps, per wave by DPP, then one atomic per (tap, channel)
  float* rw = red + wv * 44 * NC;
  auto put = [&](V v, int t, bool ok) {
    V m = ok ? v : z2;
    m = wave_total(m);
    if (lane == 63) red_put(rw, t * NC, m);
  };
  if (PART != 2) { put(hs0, 40, ovalid); put(hs1, 41, ovalid); }
  if (PART != 1) {
#pragma unroll
    for (int t = 0; t < 25; ++t) put(S.g5[t], t, own_col);
#pragma unroll
    for (int t = 0; t < 9; ++t) put(S.g3[t], 25 + t, own_col);
#pragma unroll
    for (int t = 0; t < 3; ++t) { put(S.gv[t], 34 + t, own_col); put(S.gh[t], 37 + t, own_col); }
  }
  __syncthreads();
  for (int i = tid; i < 4 * 44 * NC; i += 256) {
    const int w = i / (44 * NC), r = i - w * 44 * NC, t = r / NC, k = r - t * NC;
    const int e = (blockIdx.x % chunks) * 4 * NC + w * NC + k;
    if (e >= E) continue;
    const float v = red[i];
    if (t >= 40) { if (PART != 2 && t < 42) atomicAdd(hst + (t - 40) * E + e, v); }
    else if (PART == 1) continue;
    else if (t < 25) atomicAdd(dw5 + e * 25 + t, v);
    else if (t < 34) atomicAdd(dw3 + e * 9 + t - 25, v);
    else if (t < 37) atomicAdd(dwv + e * 3 + t - 34, v);
    else atomicAdd(dwh + e * 3 + t - 37, v);
  }
}


// ------------------------------------------------------------------------------------------------ K3': backward, packed over TWO ROW
// SEGMENTS of ONE channel: the halves of every f32x2 are (segment sa, segment sb = sa + segs/2) of the same strip and channel, so
// the 40 weights and 12 BatchNorm coefficients are plain scalars (52 SGPRs, broadcast into both halves of a packed FMA).
// block = 4 waves = the 4 channels of one quad (they share every cache line of the row-planar layout).
template <int P, int PART>
__device__ __forceinline__ void sw2_step(SwState<f32x2>& S, const BranchW<float>& bw, const Coef<float>& CO, float cm, const f32x2* XSw, int lane,
                                         f32x2 x1, f32x2 dp, f32x2 mf, f32x2 mo) {
  typedef f32x2 V;
  V in[5];
  const V* xr = XSw + P * 68 + lane;
  in[0] = xr[0]; in[1] = xr[1]; in[2] = x1; in[3] = xr[3]; in[4] = xr[4];
  const V* x2 = XSw + ((P + 1) % 5) * 68 + lane;
  if (PART != 1) {
    V i2[5];
#pragma unroll
    for (int d = 0; d < 5; ++d) i2[d] = x2[d];
#pragma unroll
    for (int ky = 1; ky < 5; ++ky)
#pragma unroll
      for (int kx = 0; kx < 5; ++kx) S.g5[ky * 5 + kx] += S.h5[(P + 8 - ky) % 5] * i2[kx];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) S.g3[ky * 3 + kx] += S.h3[(P + 7 - ky) % 5] * i2[1 + kx];
      S.gv[ky] += S.hv[(P + 7 - ky) % 5] * i2[2];
    }
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) S.gh[kx] += S.hh[(P + 1) % 5] * i2[1 + kx];
  }
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
  constexpr int Q = (P + 3) % 5;
  const V m = mf * cm;     // f_b = 0 in columns outside the image and in rows outside it (per half)
  V f5 = (CO.c[0] * S.a5[Q] + (CO.a[0] * dp + CO.d[0])) * m;
  V f3 = (CO.c[1] * S.a3[Q] + (CO.a[1] * dp + CO.d[1])) * m;
  V fv = (CO.c[2] * S.av[Q] + (CO.a[2] * dp + CO.d[2])) * m;
  V fh = (CO.c[3] * S.ah[Q] + (CO.a[3] * dp + CO.d[3])) * m;
  S.h5[Q] = f5 * mo; S.h3[Q] = f3 * mo; S.hv[Q] = fv * mo; S.hh[Q] = fh * mo;
  if (PART != 1) {
#pragma unroll
    for (int kx = 0; kx < 5; ++kx) S.g5[kx] += S.h5[Q] * x2[kx];
  }
  if constexpr (PART != 2) {
    V sh[5];
    sh[2] = f5;
    sh[1] = lane_from_right(f5);
    sh[0] = lane_from_right(sh[1]);
    sh[3] = lane_from_left(f5);
    sh[4] = lane_from_left(sh[3]);
    V s3[3];
    s3[1] = f3;
    s3[0] = lane_from_right(f3);
    s3[2] = lane_from_left(f3);
    const V hr = lane_from_right(fh), hl = lane_from_left(fh);
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
}

template <int PART, int D, int D2, bool HALO, int WPS>
__global__ __launch_bounds__(256, WPS) void k_bwd2(const float* __restrict__ z, const float* __restrict__ dpre, float* __restrict__ dh,
                                                 const float* __restrict__ preA, const float* __restrict__ preS, const float* __restrict__ w5,
                                                 const float* __restrict__ w3, const float* __restrict__ wvv, const float* __restrict__ whh,
                                                 const float* __restrict__ cA, const float* __restrict__ cC, const float* __restrict__ cD,
                                                 float* __restrict__ dw5, float* __restrict__ dw3, float* __restrict__ dwv,
                                                 float* __restrict__ dwh, float* __restrict__ hst, int B, int H, int W, int E, int strips,
                                                 int segs /* even */, int seg_rows, int chunks /* quads */) {
  typedef f32x2 V;
  constexpr int QW = HALO ? 60 : 56;
  __shared__ V XSa[4][5 * 68];
  __shared__ V ZSa[4][5 * 64];
  __shared__ float red[4 * 44];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  // decode: quad fastest, then strip, segment PAIR, image
  int lid = blockIdx.x;
  const int quad = lid % chunks; lid /= chunks;
  const int strip = lid % strips; lid /= strips;
  const int hs = segs >> 1;
  const int sa = lid % hs;
  const int b = lid / hs;
  int ch = quad * 4 + wv;
  const bool cok = ch < E;
  if (!cok) ch = 0;
  const int ysA = sa * seg_rows, yeA = min(ysA + seg_rows, H);
  const int ysB = min((sa + hs) * seg_rows, H), yeB = min(ysB + seg_rows, H);   // (may be empty: rowsB = 0)
  const int rowsA = yeA - ysA, rowsB = yeB - ysB;
  const int xs = strip * QW;
  const unsigned rowb = (unsigned)(E * W) * 4u, qoff = (unsigned)(quad * 4 * W) * 4u;
  const int r0 = b * H;
  auto so = [&](int iy) -> unsigned { const int y = min(max(iy, 0), H - 1); return (unsigned)(r0 + y) * rowb + qoff; };
  V* XS = XSa[wv];
  V* ZS = ZSa[wv];
  for (int i = lane; i < 5 * 68; i += 64) XS[i] = V{0.f, 0.f};
  BranchW<float> bw;
  load_branch_w(bw, w5, w3, wvv, whh, ch, cok);
  Coef<float> CO;
#pragma unroll
  for (int k = 0; k < 4; ++k) { CO.a[k] = cok ? cA[k * E + ch] : 0.f; CO.c[k] = cok ? cC[k * E + ch] : 0.f; CO.d[k] = cok ? cD[k * E + ch] : 0.f; }
  const float pa = cok ? preA[ch] : 0.f, ps = cok ? preS[ch] : 0.f;
  const int cx = xs - (HALO ? 2 : 4) + lane;
  const bool col_in = cx >= 0 && cx < W && cok;
  const float cm = col_in ? 1.f : 0.f;
  const int hx = lane < 2 ? xs - 4 + lane : xs + 60 + lane;
  const bool hcol_in = HALO && lane < 4 && hx >= 0 && hx < W && cok;
  const float hm = hcol_in ? 1.f : 0.f;
  const int hidx = lane < 2 ? lane : 64 + lane;
  const bool own_col = HALO ? (lane >= 2 && lane < 62) : (lane >= 4 && lane < 60);
  const bool ovalid = own_col && cx < W && cok;
  const unsigned voff = col_in ? (unsigned)(cx * 4 + (ch & 3)) * 4u : OOB;
  const unsigned vhalo = hcol_in ? (unsigned)(hx * 4 + (ch & 3)) * 4u : OOB;
  const unsigned vst = ovalid ? voff : OOB;
  const BufRsrc rz = make_rsrc(z, NREC), rd = make_rsrc(dpre, NREC), ro = make_rsrc(dh, NREC);
  const int nsteps = rowsA + 10;   // rowsA >= rowsB
  V pfz[5], pfh[5], pfd[5];
  auto ldz = [&](int j, V& zz, V& zh_) {
    const unsigned a = so(ysA - 4 + j), bb = so(ysB - 4 + j);
    zz = V{bloadv<float>(rz, voff, a), bloadv<float>(rz, voff, bb)};
    if (HALO) zh_ = V{bloadv<float>(rz, vhalo, a), bloadv<float>(rz, vhalo, bb)};
  };
  auto ldd = [&](int j, V& dd) { dd = V{bloadv<float>(rd, voff, so(ysA - 6 + j)), bloadv<float>(rd, voff, so(ysB - 6 + j))}; };
#pragma unroll
  for (int d = 0; d < D; ++d) ldz(d, pfz[d], pfh[d]);
#pragma unroll
  for (int d = 0; d < D2; ++d) ldd(d, pfd[d]);
  SwState<V> S;
  const V z2 = V{0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 5; ++k) S.a5[k] = S.a3[k] = S.av[k] = S.ah[k] = S.h5[k] = S.h3[k] = S.hv[k] = S.hh[k] = S.dxa[k] = z2;
#pragma unroll
  for (int k = 0; k < 25; ++k) S.g5[k] = z2;
#pragma unroll
  for (int k = 0; k < 9; ++k) S.g3[k] = z2;
#pragma unroll
  for (int k = 0; k < 3; ++k) S.gv[k] = S.gh[k] = z2;
  V hs0 = z2, hs1 = z2;
  auto msk = [&](bool a, bool bq) -> V { return V{a ? 1.f : 0.f, bq ? 1.f : 0.f}; };
#define STEP(P)                                                                                                    \
  {                                                                                                                \
    const int j = j0 + P;                                                                                          \
    const V zv = pfz[P];                                                                                           \
    const int iyA = ysA - 4 + j, iyB = ysB - 4 + j;                                                                \
    const V rm = msk(iyA >= 0 && iyA < H, iyB >= 0 && iyB < H && rowsB > 0);                                       \
    V x1 = pre2(zv, V{pa, pa}, V{ps, ps}) * (rm * cm);                                                             \
    V x1h = z2;                                                                                                    \
    if (HALO) x1h = pre2(pfh[P], V{pa, pa}, V{ps, ps}) * (rm * hm);                                                \
    const V dp = pfd[P];                                                                                           \
    ldz(j + D, pfz[(P + D) % 5], pfh[(P + D) % 5]);                                                                \
    ldd(j + D2, pfd[(P + D2) % 5]);                                                                                \
    WAVE_SYNC();                                                                                                   \
    XS[P * 68 + lane + 2] = x1;                                                                                    \
    if (HALO) { if (lane < 4) XS[P * 68 + hidx] = x1h; }                                                           \
    if (PART != 2) ZS[P * 64 + lane] = zv;                                                                         \
    WAVE_SYNC();                                                                                                   \
    const int fyA = ysA - 6 + j, fyB = ysB - 6 + j;                                                                \
    const V mf = msk(j >= 4 && fyA >= 0 && fyA < H, j >= 4 && fyB >= 0 && fyB < H && rowsB > 0);                   \
    const V mo = msk(fyA >= ysA && fyA < yeA, fyB >= ysB && fyB < yeB);                                            \
    sw2_step<P, PART>(S, bw, CO, cm, XS, lane, x1, dp, mf, mo);                                                    \
    if (PART != 2) {                                                                                               \
      constexpr int DD = (P + 1) % 5;                                                                              \
      const V zr = ZS[DD * 64 + lane];                                                                             \
      const V hh = zr * pa + ps;                                                                                   \
      const bool okA = j >= 8 && j < rowsA + 8, okB = j >= 8 && j < rowsB + 8;                                     \
      const V dhv = S.dxa[DD] * vdhswish(hh) * msk(okA, okB);                                                      \
      bstorev(ro, okA ? vst : OOB, so(ysA - 8 + j), dhv.x);                                                        \
      bstorev(ro, okB ? vst : OOB, so(ysB - 8 + j), dhv.y);                                                        \
      hs0 += dhv;                                                                                                  \
      hs1 += dhv * zr;                                                                                             \
    }                                                                                                              \
    SB();                                                                                                          \
  }
  for (int j0 = 0; j0 < nsteps; j0 += 5) { STEP(0) STEP(1) STEP(2) STEP(3) STEP(4) }
#undef STEP
  // reductions: both halves add up; per wave by DPP, then one atomic per (tap, channel)
  float* rw = red + wv * 44;
  auto put = [&](V v, int t, bool ok) {
    float m = ok ? v.x + v.y : 0.f;
    m = wave_total(m);
    if (lane == 63) rw[t] = m;
  };
  if (PART != 2) { put(hs0, 40, ovalid); put(hs1, 41, ovalid); }
  if (PART != 1) {
#pragma unroll
    for (int t = 0; t < 25; ++t) put(S.g5[t], t, own_col);
#pragma unroll
    for (int t = 0; t < 9; ++t) put(S.g3[t], 25 + t, own_col);
#pragma unroll
    for (int t = 0; t < 3; ++t) { put(S.gv[t], 34 + t, own_col); put(S.gh[t], 37 + t, own_col); }
  }
  __syncthreads();
  for (int i = tid; i < 4 * 44; i += 256) {
    const int w = i / 44, t = i - w * 44;
    const int e = quad * 4 + w;
    if (e >= E || t >= 42) continue;
    const float v = red[i];
    if (t >= 40) { if (PART != 2) atomicAdd(hst + (t - 40) * E + e, v); }
    else if (PART == 1) continue;
    else if (t < 25) atomicAdd(dw5 + e * 25 + t, v);
    else if (t < 34) atomicAdd(dw3 + e * 9 + t - 25, v);
    else if (t < 37) atomicAdd(dwv + e * 3 + t - 34, v);
    else atomicAdd(dwh + e * 3 + t - 37, v);
  }
}


// ------------------------------------------------------------------------------------------------ K3'': the two-segment backward with the
// packed FMAs written as asm: hipcc never folds a scalar splat into op_sel of an SGPR operand (it builds an {w, w} SGPR pair per
// weight: 80 + 24 SGPRs, spilled to VGPR lanes and read back with v_readlane before every use), so TWO weights share one SGPR pair
// and op_sel / op_sel_hi pick the half that is broadcast into both lanes of the packed operation.
// Packed results need one wait state before a dependent VALU read: consecutive statements never chain (different accumulators),
// the few that would are separated by s_nop 0.
__device__ __forceinline__ void pkfma(f32x2& acc, f32x2 x, f32x2 wp, int hi) {   // acc += x * wp[hi]  (hi folds after unrolling)
  if (hi) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(x), "s"(wp));
  else asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(x), "s"(wp));
}
__device__ __forceinline__ void pkmul(f32x2& acc, f32x2 x, f32x2 wp, int hi) {   // acc = x * wp[hi]
  if (hi) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(acc) : "v"(x), "s"(wp));
  else asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(acc) : "v"(x), "s"(wp));
}
__device__ __forceinline__ void pkfma_vv(f32x2& acc, f32x2 x, f32x2 y) { asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(y)); }
#define NOP0() asm volatile("s_nop 0")
struct W2 { f32x2 w5[13], w3[5], wv[2], wh[2]; };   // tap t of a kernel: pair t >> 1, half t & 1
#define W5(t) bw.w5[(t) >> 1], (t) & 1
#define W3(t) bw.w3[(t) >> 1], (t) & 1
#define WV(t) bw.wv[(t) >> 1], (t) & 1
#define WH(t) bw.wh[(t) >> 1], (t) & 1

template <int P, int PART>
__device__ __forceinline__ void sw3_step(SwState<f32x2>& S, const W2& bw, const f32x2 (&co)[6], float cm, const f32x2* XSw, int lane,
                                         f32x2 x1, f32x2 dp, f32x2 mf, f32x2 mo) {
  typedef f32x2 V;
  V in[5];
  const V* xr = XSw + P * 68 + lane;
  in[0] = xr[0]; in[1] = xr[1]; in[2] = x1; in[3] = xr[3]; in[4] = xr[4];
  const V* x2 = XSw + ((P + 1) % 5) * 68 + lane;
  V i2[5];
  if (PART != 1) {
#pragma unroll
    for (int d = 0; d < 5; ++d) i2[d] = x2[d];
    NOP0();
#pragma unroll
    for (int ky = 1; ky < 5; ++ky)
#pragma unroll
      for (int kx = 0; kx < 5; ++kx) pkfma_vv(S.g5[ky * 5 + kx], S.h5[(P + 8 - ky) % 5], i2[kx]);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) pkfma_vv(S.g3[ky * 3 + kx], S.h3[(P + 7 - ky) % 5], i2[1 + kx]);
      pkfma_vv(S.gv[ky], S.hv[(P + 7 - ky) % 5], i2[2]);
    }
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) pkfma_vv(S.gh[kx], S.hh[(P + 1) % 5], i2[1 + kx]);
  }
  NOP0();
#pragma unroll
  for (int d = 0; d < 5; ++d) {
    if (d == 0) pkmul(S.a5[(P + 2) % 5], in[0], W5(0));
    else pkfma(S.a5[(P + 2) % 5], in[d], W5(d));
#pragma unroll
    for (int ky = 1; ky < 5; ++ky) pkfma(S.a5[(P - ky + 7) % 5], in[d], W5(ky * 5 + d));
    if (d >= 1 && d <= 3) {
      if (d == 1) { pkmul(S.a3[(P + 1) % 5], in[1], W3(0)); pkmul(S.ah[P], in[1], WH(0)); }
      else { pkfma(S.a3[(P + 1) % 5], in[d], W3(d - 1)); pkfma(S.ah[P], in[d], WH(d - 1)); }
#pragma unroll
      for (int ky = 1; ky < 3; ++ky) pkfma(S.a3[(P - ky + 6) % 5], in[d], W3(ky * 3 + d - 1));
    }
    if (d == 2) {
      pkmul(S.av[(P + 1) % 5], in[2], WV(0));
#pragma unroll
      for (int ky = 1; ky < 3; ++ky) pkfma(S.av[(P - ky + 6) % 5], in[2], WV(ky));
    }
  }
  NOP0();
  constexpr int Q = (P + 3) % 5;
  const V m = mf * cm;     // f_b = 0 in columns outside the image and in rows outside it (per half)
  // co[k] = {cA_k, cC_k} for k < 4, co[4] = {cD_0, cD_1}, co[5] = {cD_2, cD_3}:  f = (cC a + cA dp + cD) m
  V f5, f3, fv, fh;
  pkmul(f5, dp, co[0], 0); pkmul(f3, dp, co[1], 0); pkmul(fv, dp, co[2], 0); pkmul(fh, dp, co[3], 0);
  pkfma(f5, S.a5[Q], co[0], 1); pkfma(f3, S.a3[Q], co[1], 1); pkfma(fv, S.av[Q], co[2], 1); pkfma(fh, S.ah[Q], co[3], 1);
  {
    V d0, d1, d2, d3;   // (m * cD_k)
    pkmul(d0, m, co[4], 0); pkmul(d1, m, co[4], 1); pkmul(d2, m, co[5], 0); pkmul(d3, m, co[5], 1);
    NOP0();
    f5 = f5 * m + d0; f3 = f3 * m + d1; fv = fv * m + d2; fh = fh * m + d3;
  }
  S.h5[Q] = f5 * mo; S.h3[Q] = f3 * mo; S.hv[Q] = fv * mo; S.hh[Q] = fh * mo;
  if constexpr (PART != 2) {
    V sh[5];
    sh[2] = f5;
    sh[1] = lane_from_right(f5);
    sh[0] = lane_from_right(sh[1]);
    sh[3] = lane_from_left(f5);
    sh[4] = lane_from_left(sh[3]);
    V s3[3];
    s3[1] = f3;
    s3[0] = lane_from_right(f3);
    s3[2] = lane_from_left(f3);
    const V hr = lane_from_right(fh), hl = lane_from_left(fh);
    NOP0();
#pragma unroll
    for (int kx = 0; kx < 5; ++kx) {
      if (kx == 0) pkmul(S.dxa[P], sh[0], W5(20));
      else pkfma(S.dxa[P], sh[kx], W5(20 + kx));
#pragma unroll
      for (int ky = 0; ky < 4; ++ky) pkfma(S.dxa[(P + ky + 1) % 5], sh[kx], W5(ky * 5 + kx));
    }
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) pkfma(S.dxa[(P + ky + 2) % 5], s3[kx], W3(ky * 3 + kx));
    }
    // 3x1 and 1x3: interleaved so that no statement reads the accumulator its predecessor wrote
    pkfma(S.dxa[(P + 3) % 5], fv, WV(1));
    pkfma(S.dxa[(P + 2) % 5], fv, WV(0));
    pkfma(S.dxa[Q], hr, WH(0));
    pkfma(S.dxa[(P + 4) % 5], fv, WV(2));
    pkfma(S.dxa[Q], fh, WH(1));
    NOP0();
    pkfma(S.dxa[Q], hl, WH(2));
    NOP0();
  }
  if (PART != 1) {   // f row j-2 (this step's) against x1 row j-4: kernel row 0 of the 5x5 gradient
#pragma unroll
    for (int kx = 0; kx < 5; ++kx) pkfma_vv(S.g5[kx], S.h5[Q], i2[kx]);
    NOP0();
  }
}

__device__ unsigned long long g_tim[4 * 8192];
__device__ unsigned long long g_tim2[2 * 8192];
template <int PART, int D, int D2, bool HALO, int WPS>
__global__ __launch_bounds__(256, WPS) void k_bwd3(const float* __restrict__ z, const float* __restrict__ dpre, float* __restrict__ dh,
                                                 const float* __restrict__ preA, const float* __restrict__ preS, const float* __restrict__ w5,
                                                 const float* __restrict__ w3, const float* __restrict__ wvv, const float* __restrict__ whh,
                                                 const float* __restrict__ cA, const float* __restrict__ cC, const float* __restrict__ cD,
                                                 float* __restrict__ dw5, float* __restrict__ dw3, float* __restrict__ dwv,
                                                 float* __restrict__ dwh, float* __restrict__ hst, int B, int H, int W, int E, int strips,
                                                 int segs /* even */, int seg_rows, int chunks /* quads */) {
  typedef f32x2 V;
  constexpr int QW = HALO ? 60 : 56;
  __shared__ V XSa[4][5 * 68];
  __shared__ V ZSa[4][5 * 64];
  __shared__ float red[4 * 44];
  const unsigned long long tr_in = __builtin_amdgcn_s_memrealtime();
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  int lid = blockIdx.x;
  const int quad = lid % chunks; lid /= chunks;
  const int strip = lid % strips; lid /= strips;
  const int hs = segs >> 1;
  const int sa = lid % hs;
  const int b = lid / hs;
  int ch = quad * 4 + wv;
  const bool cok = ch < E;
  if (!cok) ch = 0;
  const int ysA = sa * seg_rows, yeA = min(ysA + seg_rows, H);
  const int ysB = min((sa + hs) * seg_rows, H), yeB = min(ysB + seg_rows, H);   // (may be empty: rowsB = 0)
  const int rowsA = yeA - ysA, rowsB = yeB - ysB;
  const int xs = strip * QW;
  const unsigned rowb = (unsigned)(E * W) * 4u, qoff = (unsigned)(quad * 4 * W) * 4u;
  const int r0 = b * H;
  auto so = [&](int iy) -> unsigned { const int y = min(max(iy, 0), H - 1); return (unsigned)(r0 + y) * rowb + qoff; };
  V* XS = XSa[wv];
  V* ZS = ZSa[wv];
  for (int i = lane; i < 5 * 68; i += 64) XS[i] = V{0.f, 0.f};
  W2 bw;
  auto wl = [&](const float* w, int NT, int t) -> float { return (cok && t < NT) ? w[ch * NT + t] : 0.f; };
#pragma unroll
  for (int k = 0; k < 13; ++k) bw.w5[k] = V{wl(w5, 25, 2 * k), wl(w5, 25, 2 * k + 1)};
#pragma unroll
  for (int k = 0; k < 5; ++k) bw.w3[k] = V{wl(w3, 9, 2 * k), wl(w3, 9, 2 * k + 1)};
#pragma unroll
  for (int k = 0; k < 2; ++k) { bw.wv[k] = V{wl(wvv, 3, 2 * k), wl(wvv, 3, 2 * k + 1)}; bw.wh[k] = V{wl(whh, 3, 2 * k), wl(whh, 3, 2 * k + 1)}; }
  V co[6];
#pragma unroll
  for (int k = 0; k < 4; ++k) co[k] = V{cok ? cA[k * E + ch] : 0.f, cok ? cC[k * E + ch] : 0.f};
  co[4] = V{cok ? cD[ch] : 0.f, cok ? cD[E + ch] : 0.f};
  co[5] = V{cok ? cD[2 * E + ch] : 0.f, cok ? cD[3 * E + ch] : 0.f};
  const float pa = cok ? preA[ch] : 0.f, ps = cok ? preS[ch] : 0.f;
  const int cx = xs - (HALO ? 2 : 4) + lane;
  const bool col_in = cx >= 0 && cx < W && cok;
  const float cm = col_in ? 1.f : 0.f;
  const int hx = lane < 2 ? xs - 4 + lane : xs + 60 + lane;
  const bool hcol_in = HALO && lane < 4 && hx >= 0 && hx < W && cok;
  const float hm = hcol_in ? 1.f : 0.f;
  const int hidx = lane < 2 ? lane : 64 + lane;
  const bool own_col = HALO ? (lane >= 2 && lane < 62) : (lane >= 4 && lane < 60);
  const bool ovalid = own_col && cx < W && cok;
  const unsigned voff = col_in ? (unsigned)(cx * 4 + (ch & 3)) * 4u : OOB;
  const unsigned vhalo = hcol_in ? (unsigned)(hx * 4 + (ch & 3)) * 4u : OOB;
  const unsigned vst = ovalid ? voff : OOB;
  const BufRsrc rz = make_rsrc(z, NREC), rd = make_rsrc(dpre, NREC), ro = make_rsrc(dh, NREC);
  const int nsteps = rowsA + 10;   // rowsA >= rowsB
  V pfz[5], pfh[5], pfd[5];
  auto ldz = [&](int j, V& zz, V& zh_) {
    const unsigned a = so(ysA - 4 + j), bb = so(ysB - 4 + j);
    zz = V{bloadv<float>(rz, voff, a), bloadv<float>(rz, voff, bb)};
    if (HALO) zh_ = V{bloadv<float>(rz, vhalo, a), bloadv<float>(rz, vhalo, bb)};
  };
  auto ldd = [&](int j, V& dd) { dd = V{bloadv<float>(rd, voff, so(ysA - 6 + j)), bloadv<float>(rd, voff, so(ysB - 6 + j))}; };
#pragma unroll
  for (int d = 0; d < D; ++d) ldz(d, pfz[d], pfh[d]);
#pragma unroll
  for (int d = 0; d < D2; ++d) ldd(d, pfd[d]);
  SwState<V> S;
  const V z2 = V{0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 5; ++k) S.a5[k] = S.a3[k] = S.av[k] = S.ah[k] = S.h5[k] = S.h3[k] = S.hv[k] = S.hh[k] = S.dxa[k] = z2;
#pragma unroll
  for (int k = 0; k < 25; ++k) S.g5[k] = z2;
#pragma unroll
  for (int k = 0; k < 9; ++k) S.g3[k] = z2;
#pragma unroll
  for (int k = 0; k < 3; ++k) S.gv[k] = S.gh[k] = z2;
  V hs0 = z2, hs1 = z2;
  auto msk = [&](bool a, bool bq) -> V { return V{a ? 1.f : 0.f, bq ? 1.f : 0.f}; };
#define STEP(P)                                                                                                    \
  {                                                                                                                \
    const int j = j0 + P;                                                                                          \
    const V zv = pfz[P];                                                                                           \
    const int iyA = ysA - 4 + j, iyB = ysB - 4 + j;                                                                \
    const V rm = msk(iyA >= 0 && iyA < H, iyB >= 0 && iyB < H && rowsB > 0);                                       \
    V x1 = pre2(zv, V{pa, pa}, V{ps, ps}) * (rm * cm);                                                             \
    V x1h = z2;                                                                                                    \
    if (HALO) x1h = pre2(pfh[P], V{pa, pa}, V{ps, ps}) * (rm * hm);                                                \
    const V dp = pfd[P];                                                                                           \
    ldz(j + D, pfz[(P + D) % 5], pfh[(P + D) % 5]);                                                                \
    ldd(j + D2, pfd[(P + D2) % 5]);                                                                                \
    WAVE_SYNC();                                                                                                   \
    XS[P * 68 + lane + 2] = x1;                                                                                    \
    if (HALO) { if (lane < 4) XS[P * 68 + hidx] = x1h; }                                                           \
    if (PART != 2) ZS[P * 64 + lane] = zv;                                                                         \
    WAVE_SYNC();                                                                                                   \
    const int fyA = ysA - 6 + j, fyB = ysB - 6 + j;                                                                \
    const V mf = msk(j >= 4 && fyA >= 0 && fyA < H, j >= 4 && fyB >= 0 && fyB < H && rowsB > 0);                   \
    const V mo = msk(fyA >= ysA && fyA < yeA, fyB >= ysB && fyB < yeB);                                            \
    sw3_step<P, PART>(S, bw, co, cm, XS, lane, x1, dp, mf, mo);                                                    \
    if (PART != 2) {                                                                                               \
      constexpr int DD = (P + 1) % 5;                                                                              \
      const V zr = ZS[DD * 64 + lane];                                                                             \
      const V hh = zr * pa + ps;                                                                                   \
      const bool okA = j >= 8 && j < rowsA + 8, okB = j >= 8 && j < rowsB + 8;                                     \
      const V dhv = S.dxa[DD] * vdhswish(hh) * msk(okA, okB);                                                      \
      bstorev(ro, okA ? vst : OOB, so(ysA - 8 + j), dhv.x);                                                        \
      bstorev(ro, okB ? vst : OOB, so(ysB - 8 + j), dhv.y);                                                        \
      hs0 += dhv;                                                                                                  \
      hs1 += dhv * zr;                                                                                             \
    }                                                                                                              \
    SB();                                                                                                          \
  }
  const unsigned long long tm0 = __builtin_amdgcn_s_memtime(), tr0 = __builtin_amdgcn_s_memrealtime();
  for (int j0 = 0; j0 < nsteps; j0 += 5) { STEP(0) STEP(1) STEP(2) STEP(3) STEP(4) }
#undef STEP
  {
    const unsigned long long tm1 = __builtin_amdgcn_s_memtime(), tr1 = __builtin_amdgcn_s_memrealtime();
    const int wid = blockIdx.x * 4 + wv;
    if (lane == 0 && wid < 8192) { g_tim[wid * 4] = tm1 - tm0; g_tim[wid * 4 + 1] = tr1 - tr0; g_tim[wid * 4 + 2] = nsteps; g_tim[wid * 4 + 3] = tr_in; g_tim2[wid * 2] = tr0; g_tim2[wid * 2 + 1] = tr1; }
  }
  float* rw = red + wv * 44;
  auto put = [&](V v, int t, bool ok) {
    float m = ok ? v.x + v.y : 0.f;
    m = wave_total(m);
    if (lane == 63) rw[t] = m;
  };
  if (PART != 2) { put(hs0, 40, ovalid); put(hs1, 41, ovalid); }
  if (PART != 1) {
#pragma unroll
    for (int t = 0; t < 25; ++t) put(S.g5[t], t, own_col);
#pragma unroll
    for (int t = 0; t < 9; ++t) put(S.g3[t], 25 + t, own_col);
#pragma unroll
    for (int t = 0; t < 3; ++t) { put(S.gv[t], 34 + t, own_col); put(S.gh[t], 37 + t, own_col); }
  }
  __syncthreads();
  for (int i = tid; i < 4 * 44; i += 256) {
    const int w = i / 44, t = i - w * 44;
    const int e = quad * 4 + w;
    if (e >= E || t >= 42) continue;
    const float v = red[i];
    if (t >= 40) { if (PART != 2) atomicAdd(hst + (t - 40) * E + e, v); }
    else if (PART == 1) continue;
    else if (t < 25) atomicAdd(dw5 + e * 25 + t, v);
    else if (t < 34) atomicAdd(dw3 + e * 9 + t - 25, v);
    else if (t < 37) atomicAdd(dwv + e * 3 + t - 34, v);
    else atomicAdd(dwh + e * 3 + t - 37, v);
  }
}

// ================================================================================================ host side
static int pick_segs(int64_t waves_per_seg, int H, int halo, int slots, int* seg_rows) {
  int best = 1; double best_cost = -1;
  for (int sg = 1; sg <= H; ++sg) {
    const int rows = (H + sg - 1) / sg;
    if (sg > 1 && rows < 8) break;
    const int nseg = (H + rows - 1) / rows;
    if (nseg != sg) continue;
    const int steps = ((rows + halo + 4) / 5) * 5;
    const int64_t rounds = (waves_per_seg * nseg + slots - 1) / slots;
    const double cost = (double)rounds * steps;
    if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = sg; }
  }
  *seg_rows = (H + best - 1) / best;
  return (H + *seg_rows - 1) / *seg_rows;
}

struct Host {
  int B, H, W, E;
  std::vector<float> z, pre, u, dpre, dh, A, S, w5, w3, wv, wh, keff, beff, sg, dm, cA, cC, cD;
};
static float frand() { return (float)rand() / RAND_MAX * 2.f - 1.f; }

static double hsw(double x) { double t = x / 6 + 0.5; t = t < 0 ? 0 : (t > 1 ? 1 : t); return x * t; }
static double dhsw(double x) { return x < -3 ? 0 : (x <= 3 ? x / 3 + 0.5 : 1); }
static double gelu(double x) { return 0.5 * x * (1 + erf(x * 0.70710678118654752440)); }
static double dgelu(double x) { return 0.5 * (1 + erf(x * 0.70710678118654752440)) + x * 0.39894228040143267794 * exp(-0.5 * x * x); }

int main(int argc, char** argv) {
  const bool check = argc > 1 && !strcmp(argv[1], "check");
  struct Shape { int B, H, W, E; };
  std::vector<Shape> shapes;
  if (check) shapes = {{2, 23, 70, 12}, {1, 9, 130, 8}};
  else shapes = {{8, 352, 352, 24}, {8, 176, 176, 48}, {8, 88, 88, 96}, {8, 44, 44, 192}};
  for (const Shape& sh : shapes) {
    const int B = sh.B, H = sh.H, W = sh.W, E = sh.E;
    g_W = W;
    const int64_t npix = (int64_t)B * H * W, n = q64_size(npix, E);
    srand(1);
    std::vector<float> hz(n, 0.f), hpre(n, 0.f), hu(n, 0.f), hdpre(n, 0.f);
    std::vector<float> A(E), S(E), w5(E * 25), w3(E * 9), wv(E * 3), wh(E * 3), keff(E * 25), beff(E), sg(B * E), dm(B * E), cA(4 * E), cC(4 * E), cD(4 * E);
    for (int64_t p = 0; p < npix; ++p)
      for (int c = 0; c < E; ++c) {
        hz[q64_elem(p, c, E)] = frand() * 2.5f;
        hpre[q64_elem(p, c, E)] = frand() * 2.f;
        hu[q64_elem(p, c, E)] = frand();
        hdpre[q64_elem(p, c, E)] = frand();
      }
    for (int c = 0; c < E; ++c) { A[c] = 0.8f + 0.4f * frand(); S[c] = 0.5f * frand(); beff[c] = frand() * 0.1f; }
    for (auto& v : w5) v = frand() * 0.2f;
    for (auto& v : w3) v = frand() * 0.3f;
    for (auto& v : wv) v = frand() * 0.5f;
    for (auto& v : wh) v = frand() * 0.5f;
    for (auto& v : keff) v = frand() * 0.2f;
    for (auto& v : sg) v = 0.5f + 0.5f * frand();
    for (auto& v : dm) v = 0.01f * frand();
    for (auto& v : cA) v = 1.f + 0.3f * frand();
    for (auto& v : cC) v = 0.05f * frand();
    for (auto& v : cD) v = 0.02f * frand();
    auto dev = [&](const std::vector<float>& h) { float* d; CK(hipMalloc(&d, h.size() * 4)); CK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice)); return d; };
    const int NSET = check ? 1 : 3;
    std::vector<float*> dz(NSET), dpr(NSET), du(NSET), ddp(NSET), ddh(NSET), dpo(NSET);
    for (int s = 0; s < NSET; ++s) {
      dz[s] = dev(hz); dpr[s] = dev(hpre); du[s] = dev(hu); ddp[s] = dev(hdpre);
      CK(hipMalloc(&ddh[s], n * 4)); CK(hipMalloc(&dpo[s], n * 4));
      CK(hipMemset(ddh[s], 0, n * 4)); CK(hipMemset(dpo[s], 0, n * 4));
    }
    float *dA = dev(A), *dS = dev(S), *d5 = dev(w5), *d3 = dev(w3), *dv = dev(wv), *dhh = dev(wh), *dk = dev(keff), *db = dev(beff), *dsg = dev(sg),
          *ddm = dev(dm), *dcA = dev(cA), *dcC = dev(cC), *dcD = dev(cD);
    float *st0, *gsum, *bst, *g5, *g3, *gv, *gh, *hst;
    CK(hipMalloc(&st0, 8 * E * 4)); CK(hipMalloc(&gsum, B * E * 4)); CK(hipMalloc(&bst, 5 * E * 4));
    CK(hipMalloc(&g5, E * 25 * 4)); CK(hipMalloc(&g3, E * 9 * 4)); CK(hipMalloc(&gv, E * 3 * 4)); CK(hipMalloc(&gh, E * 3 * 4)); CK(hipMalloc(&hst, 2 * E * 4));
    auto zero_out = [&]() {
      CK(hipMemset(st0, 0, 8 * E * 4)); CK(hipMemset(gsum, 0, B * E * 4)); CK(hipMemset(bst, 0, 5 * E * 4));
      CK(hipMemset(g5, 0, E * 25 * 4)); CK(hipMemset(g3, 0, E * 9 * 4)); CK(hipMemset(gv, 0, E * 3 * 4)); CK(hipMemset(gh, 0, E * 3 * 4)); CK(hipMemset(hst, 0, 2 * E * 4));
    };
    zero_out();
    const int strips = (W + 59) / 60;
    const bool halo = (W + 59) / 60 < (W + 55) / 56;
    const int strips3 = halo ? strips : (W + 55) / 56;
    // variant table: {name, kernel id, single-channel?}
    auto geom = [&](bool single, bool bwd, int& chunks, int& segs, int& sr) {
      chunks = single ? (E + 3) / 4 : (E + 7) / 8;
      const int st = bwd ? strips3 : strips;
      segs = pick_segs((int64_t)B * st * chunks * 4, H, bwd ? 10 : 4, 256 * (bwd && !single ? 8 : 16), &sr);
      return B * st * chunks * segs;
    };
#define ARGS_B dz[s], ddp[s], ddh[s], dA, dS, d5, d3, dv, dhh, dcA, dcC, dcD, g5, g3, gv, gh, hst, B, H, W, E, strips3, segs, sr, chunks
#define BWD(V, PART, WPS) { if (halo) k_bwd<V, PART, 3, 2, true, (WPS == 4 && PART != 1 ? 3 : WPS), true><<<nb, 256>>>(ARGS_B); else k_bwd<V, PART, 3, 2, false, WPS, true><<<nb, 256>>>(ARGS_B); }
    auto run = [&](int which, int s) {
      int chunks, segs, sr, nb;
      switch (which) {
        case 0: nb = geom(false, false, chunks, segs, sr); k_stats0<f32x2, 5, true, false, 4><<<nb, 256>>>(dz[s], dA, dS, d5, d3, dv, dhh, st0, B, H, W, E, strips, segs, sr, chunks); break;
        case 1: nb = geom(false, false, chunks, segs, sr); k_fwd<f32x2, 5, false, 4><<<nb, 256>>>(dz[s], dpo[s], gsum, dA, dS, dk, db, B, H, W, E, strips, segs, sr, chunks); break;
        case 2: nb = geom(false, false, chunks, segs, sr); k_stats1<f32x2, 3, false, 3><<<nb, 256>>>(dz[s], dpr[s], du[s], dsg, ddm, dpo[s], dA, dS, d5, d3, dv, dhh, bst, B, H, W, E, strips, segs, sr, chunks); break;
        case 3: nb = geom(false, true, chunks, segs, sr); BWD(f32x2, 0, 2) break;
        case 4: nb = geom(false, true, chunks, segs, sr); BWD(f32x2, 1, 3) break;
        case 5: nb = geom(false, true, chunks, segs, sr); BWD(f32x2, 2, 2) break;
        case 6: nb = geom(true, true, chunks, segs, sr); BWD(float, 0, 4) break;
        case 7: nb = geom(true, true, chunks, segs, sr); BWD(float, 1, 4) break;
        case 8: nb = geom(true, true, chunks, segs, sr); BWD(float, 2, 4) break;
        case 9: nb = geom(true, false, chunks, segs, sr); k_stats0<float, 5, true, true, 4><<<nb, 256>>>(dz[s], dA, dS, d5, d3, dv, dhh, st0, B, H, W, E, strips, segs, sr, chunks); break;
        case 10: nb = geom(true, false, chunks, segs, sr); k_fwd<float, 5, true, 4><<<nb, 256>>>(dz[s], dpo[s], gsum, dA, dS, dk, db, B, H, W, E, strips, segs, sr, chunks); break;
        case 11: nb = geom(true, false, chunks, segs, sr); k_stats1<float, 3, true, 4><<<nb, 256>>>(dz[s], dpr[s], du[s], dsg, ddm, dpo[s], dA, dS, d5, d3, dv, dhh, bst, B, H, W, E, strips, segs, sr, chunks); break;
        case 12: nb = geom(false, false, chunks, segs, sr); k_stats0<f32x2, 5, false, false, 4><<<nb, 256>>>(dz[s], dA, dS, d5, d3, dv, dhh, st0, B, H, W, E, strips, segs, sr, chunks); break;
        case 14: nb = geom(false, false, chunks, segs, sr); k_fwd<f32x2, 5, true, 4><<<nb, 256>>>(dz[s], dpo[s], gsum, dA, dS, dk, db, B, H, W, E, strips, segs, sr, chunks); break;
        case 15: nb = geom(false, false, chunks, segs, sr); k_stats1<f32x2, 3, true, 3><<<nb, 256>>>(dz[s], dpr[s], du[s], dsg, ddm, dpo[s], dA, dS, d5, d3, dv, dhh, bst, B, H, W, E, strips, segs, sr, chunks); break;
        case 16: case 17: case 18: {
          chunks = (E + 3) / 4;
          int sg = pick_segs((int64_t)B * strips3 * chunks * 4 / 2, H, 10, 256 * 8, &sr);
          if (sg & 1) { ++sg; sr = (H + sg - 1) / sg; }
          nb = B * strips3 * chunks * (sg / 2);
#define ARGS_B2 dz[s], ddp[s], ddh[s], dA, dS, d5, d3, dv, dhh, dcA, dcC, dcD, g5, g3, gv, gh, hst, B, H, W, E, strips3, sg, sr, chunks
#define BWD2(PART, WPS) { if (halo) k_bwd2<PART, 3, 2, true, WPS><<<nb, 256>>>(ARGS_B2); else k_bwd2<PART, 3, 2, false, WPS><<<nb, 256>>>(ARGS_B2); }
          if (which == 16) BWD2(0, 2) else if (which == 17) BWD2(1, 3) else BWD2(2, 2)
        } break;
        case 19: case 20: case 21: {
          chunks = (E + 3) / 4;
          int sg = pick_segs((int64_t)B * strips3 * chunks * 4 / 2, H, 10, 256 * 8, &sr);
          if (sg & 1) { ++sg; sr = (H + sg - 1) / sg; }
          nb = B * strips3 * chunks * (sg / 2);
#define BWD3(PART, WPS) { if (halo) k_bwd3<PART, 3, 2, true, WPS><<<nb, 256>>>(ARGS_B2); else k_bwd3<PART, 3, 2, false, WPS><<<nb, 256>>>(ARGS_B2); }
          if (which == 19) BWD3(0, 2) else if (which == 20) BWD3(1, 4) else BWD3(2, 2)
        } break;
        case 13: nb = geom(false, false, chunks, segs, sr); k_stats0<f32x2, 3, true, true, 4><<<nb, 256>>>(dz[s], dA, dS, d5, d3, dv, dhh, st0, B, H, W, E, strips, segs, sr, chunks); break;
      }
    };
    { int c_, sg_, sr_; const int nb_ = geom(false, false, c_, sg_, sr_); int c2, sg2, sr2; const int nb2 = geom(true, true, c2, sg2, sr2);
      printf("shape B=%d %dx%d E=%d: strips %d (bwd %d, halo %d) | pair fwd-type: %d segs x %d rows, %d blocks | single bwd: %d segs x %d rows, %d blocks\n", B, H, W, E, strips, strips3, (int)halo, sg_, sr_, nb_, sg2, sr2, nb2); }
    if (check) {
      // ---- CPU reference in double
      auto at = [&](const std::vector<float>& t, int b, int y, int x, int c) -> double {
        if (y < 0 || y >= H || x < 0 || x >= W) return 0.0;
        return t[q64_elem(((int64_t)b * H + y) * W + x, c, E)];
      };
      std::vector<double> x1((size_t)npix * E), y4((size_t)npix * E * 4);
      auto X1 = [&](int b, int y, int x, int c) -> double { if (y < 0 || y >= H || x < 0 || x >= W) return 0.0; return x1[(((size_t)b * H + y) * W + x) * E + c]; };
      for (int b = 0; b < B; ++b) for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) for (int c = 0; c < E; ++c)
        x1[(((size_t)b * H + y) * W + x) * E + c] = hsw((double)A[c] * at(hz, b, y, x, c) + S[c]);
      std::vector<double> rst(8 * E, 0.0), rg(B * E, 0.0), rbst(5 * E, 0.0);
      std::vector<double> rpre((size_t)npix * E), rdpre((size_t)npix * E);
      for (int b = 0; b < B; ++b) for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) for (int c = 0; c < E; ++c) {
        double y5 = 0, y3 = 0, yv = 0, yh = 0, pm = 0;
        for (int ky = 0; ky < 5; ++ky) for (int kx = 0; kx < 5; ++kx) {
          const double v = X1(b, y + ky - 2, x + kx - 2, c);
          y5 += w5[c * 25 + ky * 5 + kx] * v;
          pm += keff[c * 25 + ky * 5 + kx] * v;
        }
        for (int ky = 0; ky < 3; ++ky) for (int kx = 0; kx < 3; ++kx) y3 += w3[c * 9 + ky * 3 + kx] * X1(b, y + ky - 1, x + kx - 1, c);
        for (int k = 0; k < 3; ++k) { yv += wv[c * 3 + k] * X1(b, y + k - 1, x, c); yh += wh[c * 3 + k] * X1(b, y, x + k - 1, c); }
        const size_t i = (((size_t)b * H + y) * W + x) * E + c;
        y4[i * 4] = y5; y4[i * 4 + 1] = y3; y4[i * 4 + 2] = yv; y4[i * 4 + 3] = yh;
        const double ys[4] = {y5, y3, yv, yh};
        for (int k = 0; k < 4; ++k) { rst[(k * 2) * E + c] += ys[k]; rst[(k * 2 + 1) * E + c] += ys[k] * ys[k]; }
        pm += beff[c];
        rpre[i] = pm;
        rg[b * E + c] += gelu(pm);
        const double d = ((double)at(hu, b, y, x, c) * sg[b * E + c] + dm[b * E + c]) * dgelu(at(hpre, b, y, x, c));
        rdpre[i] = d;
        rbst[c] += d;
        for (int k = 0; k < 4; ++k) rbst[(1 + k) * E + c] += d * ys[k];
      }
      // backward reference (uses the random dpre tensor hdpre)
      std::vector<double> f4((size_t)npix * E * 4);
      for (int b = 0; b < B; ++b) for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) for (int c = 0; c < E; ++c) {
        const size_t i = (((size_t)b * H + y) * W + x) * E + c;
        for (int k = 0; k < 4; ++k) f4[i * 4 + k] = (double)cA[k * E + c] * at(hdpre, b, y, x, c) + (double)cC[k * E + c] * y4[i * 4 + k] + cD[k * E + c];
      }
      auto F = [&](int b, int y, int x, int c, int k) -> double { if (y < 0 || y >= H || x < 0 || x >= W) return 0.0; return f4[((((size_t)b * H + y) * W + x) * E + c) * 4 + k]; };
      std::vector<double> rdh((size_t)npix * E), rhst(2 * E, 0.0), rg5(E * 25, 0.0), rg3(E * 9, 0.0), rgv(E * 3, 0.0), rgh(E * 3, 0.0);
      for (int b = 0; b < B; ++b) for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) for (int c = 0; c < E; ++c) {
        double dx = 0;
        for (int ky = 0; ky < 5; ++ky) for (int kx = 0; kx < 5; ++kx) dx += w5[c * 25 + ky * 5 + kx] * F(b, y - ky + 2, x - kx + 2, c, 0);
        for (int ky = 0; ky < 3; ++ky) for (int kx = 0; kx < 3; ++kx) dx += w3[c * 9 + ky * 3 + kx] * F(b, y - ky + 1, x - kx + 1, c, 1);
        for (int k = 0; k < 3; ++k) { dx += wv[c * 3 + k] * F(b, y - k + 1, x, c, 2); dx += wh[c * 3 + k] * F(b, y, x - k + 1, c, 3); }
        const double zz = at(hz, b, y, x, c);
        const double v = dx * dhsw((double)A[c] * zz + S[c]);
        rdh[(((size_t)b * H + y) * W + x) * E + c] = v;
        rhst[c] += v; rhst[E + c] += v * zz;
        for (int ky = 0; ky < 5; ++ky) for (int kx = 0; kx < 5; ++kx) rg5[c * 25 + ky * 5 + kx] += F(b, y, x, c, 0) * X1(b, y + ky - 2, x + kx - 2, c);
        for (int ky = 0; ky < 3; ++ky) for (int kx = 0; kx < 3; ++kx) rg3[c * 9 + ky * 3 + kx] += F(b, y, x, c, 1) * X1(b, y + ky - 1, x + kx - 1, c);
        for (int k = 0; k < 3; ++k) { rgv[c * 3 + k] += F(b, y, x, c, 2) * X1(b, y + k - 1, x, c); rgh[c * 3 + k] += F(b, y, x, c, 3) * X1(b, y, x + k - 1, c); }
      }
      auto cmp = [&](const char* name, const std::vector<float>& got, const std::vector<double>& ref) {
        double me = 0, mr = 0;
        for (size_t i = 0; i < ref.size(); ++i) { me = fmax(me, fabs(got[i] - ref[i])); mr = fmax(mr, fabs(ref[i])); }
        printf("  %-10s max err %.3e (max ref %.3e) %s\n", name, me, mr, me <= 2e-4 * (mr + 1e-3) + 1e-5 ? "ok" : "FAIL");
      };
      auto fetch = [&](float* d, size_t cnt) { std::vector<float> h(cnt); CK(hipMemcpy(h.data(), d, cnt * 4, hipMemcpyDeviceToHost)); return h; };
      auto fetch_t = [&](float* d) {  // Q64 tensor -> [pix][E]
        std::vector<float> q = fetch(d, n), o((size_t)npix * E);
        for (int64_t p = 0; p < npix; ++p) for (int c = 0; c < E; ++c) o[p * E + c] = q[q64_elem(p, c, E)];
        return o;
      };
      auto chk_fwd = [&](int k0, int k1, int k2, const char* tag) {
        zero_out();
        run(k0, 0); CK(hipDeviceSynchronize());
        printf(" [%s]\n", tag);
        cmp("stats0", fetch(st0, 8 * E), rst);
        run(k1, 0); CK(hipDeviceSynchronize());
        cmp("pre", fetch_t(dpo[0]), rpre); cmp("gsum", fetch(gsum, B * E), rg);
        run(k2, 0); CK(hipDeviceSynchronize());
        cmp("dpre", fetch_t(dpo[0]), rdpre); cmp("bst", fetch(bst, 5 * E), rbst);
      };
      auto chk_bwd = [&](int kf, int k1, int k2, const char* tag) {
        zero_out(); CK(hipMemset(ddh[0], 0, n * 4));
        printf(" [%s]\n", tag);
        run(kf, 0); CK(hipDeviceSynchronize());
        cmp("dh", fetch_t(ddh[0]), rdh); cmp("hst", fetch(hst, 2 * E), rhst);
        cmp("dw5", fetch(g5, E * 25), rg5); cmp("dw3", fetch(g3, E * 9), rg3); cmp("dwv", fetch(gv, E * 3), rgv); cmp("dwh", fetch(gh, E * 3), rgh);
        zero_out(); CK(hipMemset(ddh[0], 0, n * 4));
        run(k1, 0); run(k2, 0); CK(hipDeviceSynchronize());
        cmp("dh(p1)", fetch_t(ddh[0]), rdh); cmp("hst(p1)", fetch(hst, 2 * E), rhst);
        cmp("dw5(p2)", fetch(g5, E * 25), rg5); cmp("dw3(p2)", fetch(g3, E * 9), rg3); cmp("dwv(p2)", fetch(gv, E * 3), rgv); cmp("dwh(p2)", fetch(gh, E * 3), rgh);
      };
      chk_fwd(0, 1, 2, "pair");
      chk_fwd(9, 10, 11, "single");
      chk_bwd(3, 4, 5, "pair");
      chk_bwd(6, 7, 8, "single");
      chk_bwd(16, 17, 18, "2seg");
      chk_bwd(19, 20, 21, "2seg asm");
    } else {
      const char* names[22] = {"stats0 pair", "fwd pair", "stats1 pair", "bwd pair", "bwd pair dx", "bwd pair dW", "bwd single", "bwd single dx", "bwd single dW",
                               "stats0 single", "fwd single", "stats1 single", "stats0 pair noatom", "stats0 pair D3 OL", "fwd pair OL", "stats1 pair OL", "bwd 2seg", "bwd 2seg dx", "bwd 2seg dW", "bwd 2seg asm", "bwd 2seg asm dx", "bwd 2seg asm dW"};
      const double passes[22] = {1, 2, 4, 3, 3, 2, 3, 3, 2, 1, 2, 4, 1, 1, 2, 4, 3, 3, 2, 3, 3, 2};
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      for (int k = 0; k < 22; ++k) {
        for (int it = 0; it < 3; ++it) run(k, it % NSET);
        CK(hipDeviceSynchronize());
        const int iters = 12;
        CK(hipEventRecord(e0));
        for (int it = 0; it < iters; ++it) run(k, it % NSET);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / iters, gb = passes[k] * (double)npix * E * 4 / 1e9;
        printf("  %-20s %8.1f us   %6.2f TB/s (%.0f tensor passes)\n", names[k], us, gb / us * 1e6 / 1e3, passes[k]);
        if (k >= 19) {
          std::vector<unsigned long long> ht(4 * 8192);
          CK(hipMemcpyFromSymbol(ht.data(), HIP_SYMBOL(g_tim), ht.size() * 8));
          std::vector<unsigned long long> h2(2 * 8192);
          CK(hipMemcpyFromSymbol(h2.data(), HIP_SYMBOL(g_tim2), h2.size() * 8));
          double sc = 0, sr = 0, st = 0; int nw = 0; unsigned long long imin = ~0ull, imax = 0, l0max = 0, l1min = ~0ull, l1max = 0;
          for (int w = 0; w < 8192; ++w) if (ht[w * 4 + 2]) { sc += ht[w * 4]; sr += ht[w * 4 + 1]; st += ht[w * 4 + 2]; ++nw;
            imin = std::min(imin, ht[w * 4 + 3]); imax = std::max(imax, ht[w * 4 + 3]); l0max = std::max(l0max, h2[w * 2]); l1min = std::min(l1min, h2[w * 2 + 1]); l1max = std::max(l1max, h2[w * 2 + 1]); }
          printf("      waves %d: %.0f cycles/step, loop %.1f us/wave, clock %.2f GHz | wave entry spread %.1f us, last loop start +%.1f us, first/last loop end +%.1f / +%.1f us\n", nw, sc / st, sr / nw / 100.0, sc / sr * 0.1,
                 (imax - imin) / 100.0, (l0max - imin) / 100.0, (l1min - imin) / 100.0, (l1max - imin) / 100.0);
          std::vector<unsigned long long> zz(4 * 8192, 0); CK(hipMemcpyToSymbol(HIP_SYMBOL(g_tim), zz.data(), zz.size() * 8));
        }
      }
    }
    for (int s = 0; s < NSET; ++s) { hipFree(dz[s]); hipFree(dpr[s]); hipFree(du[s]); hipFree(ddp[s]); hipFree(ddh[s]); hipFree(dpo[s]); }
  }
  return 0;
}
