// Shared device/host helpers for liblmnet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/lmnet_hip.h"

#define LMN_WAVE 64

extern thread_local char g_lmn_err[256];

#define LMN_REQUIRE(cond, ...)                           \
  do {                                                   \
    if (!(cond)) {                                       \
      snprintf(g_lmn_err, sizeof(g_lmn_err), __VA_ARGS__); \
      return LMN_E_BADARG;                               \
    }                                                    \
  } while (0)

static inline int lmn_launch_status(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(g_lmn_err, sizeof(g_lmn_err), "%s: %s", what, hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

static inline int lmn_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ---------------------------------------------------------------- runtime hooks (runtime.hip)
// (1) in-library kernel timer (lmn_prof_begin / lmn_prof_end): HIP events around every kernel launch whose name matches
//     the active filter, recorded on the stream the kernel is launched on; bench.py reads the dominant kernel's average
//     duration from it inside the timed region.  Off: one predictable branch per launch.
// (2) plan recorder (lmn_plan_*): while a plan is recording, every C-ABI entry also appends a closure of itself
//     (arguments by value) to the plan; lmn_plan_run re-issues the recorded entries in order -- one FFI crossing per
//     pass instead of one per kernel.
extern int g_lmn_prof_on;
// (3) deterministic mode (lmn_set_deterministic): every cross-block float reduction writes per-block partials into SLOTS -- private
//     copies of its destination arrays in a per-stream scratch -- with plain stores instead of float atomics, and a fixed-order
//     sum kernel folds the slots into the real destination right after the producer (same stream, same entry).
extern int g_lmn_det;
// (4) lmn_set_priority_stream: the wave priority (0..3) of the kernels launched on `st` (3: the compute chain's stream)
int lmn_prio_level(hipStream_t st);
__device__ __forceinline__ void lmn_setprio_level(int lvl) {   // (s_setprio takes an immediate)
  switch (lvl) {
    case 1: __builtin_amdgcn_s_setprio(1); break;
    case 2: __builtin_amdgcn_s_setprio(2); break;
    case 3: __builtin_amdgcn_s_setprio(3); break;
    default: break;
  }
}
// a zeroed scratch region of `floats` floats on stream st (hipMemsetAsync; the scratch itself is hipMalloc'ed once per stream and
// grown on demand: the one place where the library owns device memory; a region stays valid when a later request of the same entry
// outgrows the block -- the old block is retired, not freed).  lmn_det_begin resets the stream's scratch (once per entry).
void lmn_det_begin(hipStream_t st);
float* lmn_det_slots(hipStream_t st, size_t floats);
// dst[i] += sum_{s < nslots} slots[s * size + i]  (s ascending in fixed groups: bit-reproducible)
void lmn_det_sum(hipStream_t st, const float* slots, int nslots, int64_t size, float* dst);
bool lmn_prof_start(const char* kernel, hipStream_t st);
void lmn_prof_stop(hipStream_t st);
void lmn_prof_cost(double flops, double bytes);  // algorithmic cost of the NEXT launch of this thread (consumed by it)
// The timer files a launch under the INSTANTIATED kernel name (`dw_fwd_kernel<float, true>`: template arguments resolved), the name
// rocprofv3 prints for the same launch minus `void ` and the parameter list -- so a line of profiles/*_kernel_stats.csv and a record
// of the timer are the same key.  lmn_kname demangles the type name of LmnKTag<&kernel> once per instantiation (runtime.hip).
#ifdef __cplusplus
#include <typeinfo>
template <auto K> struct LmnKTag {};
const char* lmn_kname(const char* tag_type_name);
#define LMN_KNAME(kern) lmn_kname(typeid(LmnKTag<kern>).name())
#endif
#define LMN_LAUNCH(kern, grid, block, shmem, stream, ...)                        \
  do {                                                                           \
    const bool _lmn_pf = g_lmn_prof_on && lmn_prof_start(LMN_KNAME(kern), (stream)); \
    hipLaunchKernelGGL(kern, grid, block, shmem, stream, __VA_ARGS__);           \
    if (_lmn_pf) lmn_prof_stop((stream));                                        \
  } while (0)

#ifdef __cplusplus
#include <functional>
extern thread_local void* g_lmn_rec;  // plan being recorded by this thread, or NULL
void lmn_rec_push(std::function<int()>&& f, const char* what);
// LMN_REC(call-expression using only by-value locals): record the entry, then fall through and execute it
#define LMN_REC(...)                                                    \
  do {                                                                  \
    if (g_lmn_rec) lmn_rec_push([=]() -> int { return __VA_ARGS__; }, #__VA_ARGS__);  \
  } while (0)
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Item i of the squeeze-excite parameter gradients (lmn_se_bwd_params; also run by extra blocks of lmn_reparam_wfin's launch):
// dw2[e][r] += sum_b dt[b][e] h[b][r] | dw1[r][e] += sum_b da[b][r] m[b][e] | db2[e], db1[r] += sum_b dvec[b][.]
__device__ __forceinline__ void lmn_se_bwd_params_item(int i, const float* __restrict__ dvec, const float* __restrict__ gsum, float inv_hw,
                                                       const float* __restrict__ hidden, float* dw1, float* db1, float* dw2, float* db2,
                                                       int B, int E, int R) {
  const int ER = E * R, S = E + R;
  if (i < ER) {
    const int e = i / R, r = i - e * R;
    float a = 0.f;
    for (int b = 0; b < B; ++b) a += dvec[(int64_t)b * S + e] * hidden[(int64_t)b * R + r];
    dw2[i] += a;
  } else if (i < 2 * ER) {
    const int k = i - ER, r = k / E, e = k - r * E;
    float a = 0.f;
    for (int b = 0; b < B; ++b) a += dvec[(int64_t)b * S + E + r] * (gsum[(int64_t)b * E + e] * inv_hw);
    dw1[k] += a;
  } else if (i < 2 * ER + S) {
    const int k = i - 2 * ER;
    float a = 0.f;
    for (int b = 0; b < B; ++b) a += dvec[(int64_t)b * S + k];
    if (k < E) db2[k] += a; else db1[k - E] += a;
  }
}
#ifdef __HIPCC__
// bilinear x2, align_corners=True: source coordinate of output index `dst` along an axis of `in` source samples, scale =
// (in - 1) / (out - 1) in fp32 -- ATen's upsample_bilinear2d arithmetic: i0 = (int)(scale * dst), second tap i0 + ip, weights l0 / l1.
// Used by lmn_up2_fwd / lmn_up2_bwd (rows.hip) and by the LMN_SRC_UP2 staging of the 3x3 convs, so that the fused and the
// materialised forms sample identical values.
__device__ __forceinline__ void lmn_up_coord(int dst, int in, float scale, int& i0, int& ip, float& l0, float& l1) {
  const float r = scale * (float)dst;
  i0 = (int)r;
  ip = (i0 < in - 1) ? 1 : 0;
  l1 = r - (float)i0;
  l0 = 1.f - l1;
}
// one block's contribution to a reduced value: float atomic (arrival order) or, in deterministic mode, a plain store into the
// block's slot (the pointer then addresses the slot copy of the destination)
__device__ __forceinline__ void lmn_red_add(float* p, float v, bool det) {
  if (det) *p = v;
  else atomicAdd(p, v);
}
#endif

// ---------------------------------------------------------------- activation storage: fp32 or bf16
// Every activation tensor of a call has ONE storage type (lmn act_dtype: LMN_F32 | LMN_BF16); kernels are templated on it
// and do all arithmetic in fp32: ld4 / st4 move 4 consecutive channels (16 B or 8 B per lane), conversion is one
// v_cvt_pk_bf16_f32 per pair on the way out (round to nearest even) and a shift on the way in.
typedef __bf16 lmn_bf16;
typedef __bf16 lmn_bf16x2 __attribute__((ext_vector_type(2)));
typedef float lmn_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t lmn_pk_bf16(float a, float b) {
  const lmn_f32x2 t = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(t, lmn_bf16x2));
}
__device__ __forceinline__ float lmn_bf16_lo(uint32_t u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float lmn_bf16_hi(uint32_t u) { return __builtin_bit_cast(float, u & 0xFFFF0000u); }
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 ld4(const lmn_bf16* p) {
  const uint2 r = *reinterpret_cast<const uint2*>(p);
  return f32x4{lmn_bf16_lo(r.x), lmn_bf16_hi(r.x), lmn_bf16_lo(r.y), lmn_bf16_hi(r.y)};
}
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ void st4(lmn_bf16* p, f32x4 v) {
  *reinterpret_cast<uint2*>(p) = uint2{lmn_pk_bf16(v[0], v[1]), lmn_pk_bf16(v[2], v[3])};
}
__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const lmn_bf16* p) { return lmn_bf16_lo((uint32_t)*reinterpret_cast<const uint16_t*>(p)); }
__device__ __forceinline__ void st1(float* p, float v) { *p = v; }
__device__ __forceinline__ void st1(lmn_bf16* p, float v) { *p = (lmn_bf16)v; }
// host side: run `...` once with T = the storage type named by dt
#define LMN_ACT_DISPATCH(dt, ...)                      \
  do {                                                 \
    if ((dt) == LMN_BF16) { typedef lmn_bf16 T; __VA_ARGS__; } \
    else { typedef float T; __VA_ARGS__; }             \
  } while (0)
#define LMN_REQUIRE_DT(dt, what) LMN_REQUIRE((dt) == LMN_F32 || (dt) == LMN_BF16, what ": act_dtype %d", (int)(dt))


// ---------------------------------------------------------------- RP4: row-planar layout of the E-wide ReparamConv tensors
// element (global pixel gp = (b*H + y)*W + x, channel c) of a row-planar tensor with C channels lives at
//   (gp / W) * (C * W) + (c >> 2) * 4 * W + 4 * (gp % W) + (c & 3)  =  gp * 4 + (c >> 2) * 4W + (c & 3) + (gp / W) * (C - 4) * W
// (one plane of W pixels x 4 channels per image row and channel quad: include/lmnet_hip.h), an NHWC one at gp * cstride + c.
// Both are ONE linear form, so the kernels carry no second code path and no layout struct per operand:
//   offset = gp * cs + (c >> 2) * qs + (c & 3) + row * rf      NHWC: cs = cstride, qs = 4, rf = 0;  RP4: cs = 4, qs = 4W, rf = (C - 4) W
// with row = gp / W needed only where rf != 0 (a wave-uniform test).
struct LmnLay { int32_t cs, qs, rf; };
static inline LmnLay lmn_lay_make(int rp_w, int C, int cstride) {
  LmnLay l;
  if (rp_w > 0) { l.cs = 4; l.qs = 4 * rp_w; l.rf = (C - 4) * rp_w; }
  else { l.cs = cstride; l.qs = 4; l.rf = 0; }
  return l;
}
static inline uint32_t lmn_div_magic(int w) { return w > 1 ? (uint32_t)(0x100000000ULL / (uint64_t)w) : (w == 1 ? 0xFFFFFFFFu : 0u); }
#ifdef __HIPCC__
// gp / w for any gp < 2^32: floor(2^32 / w) as the multiplier gives a quotient short by at most one
__device__ __forceinline__ uint32_t lmn_div_row(uint32_t gp, uint32_t w, uint32_t magic) {
  uint32_t row = __umulhi(gp, magic);
  if (gp - row * w >= w) ++row;
  return row;
}
#endif

// ---------------------------------------------------------------- activations
// erf: Abramowitz & Stegun 7.1.26 (|abs error| <= 1.5e-7 over the whole line), branch-free: the libm erff has two
// branches that a wave almost always takes both of (~45 instructions per element against ~13 here).  The
// resulting GELU differs from nn.GELU()'s exact erf form by < 2e-7 * |x| -- fp32 rounding noise of the network.
__device__ __forceinline__ float lmn_erf(float x) {
  const float t = fabsf(x);
  const float k = __builtin_amdgcn_rcpf(fmaf(0.3275911f, t, 1.0f));
  float p = fmaf(1.061405429f, k, -1.453152027f);
  p = fmaf(p, k, 1.421413741f);
  p = fmaf(p, k, -0.284496736f);
  p = fmaf(p, k, 0.254829592f);
  return copysignf(1.0f - p * k * __expf(-t * t), x);
}
__device__ __forceinline__ float lmn_gelu(float x) {  // nn.GELU() default (erf form)
  return 0.5f * x * (1.0f + lmn_erf(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float lmn_dgelu(float x) {
  const float cdf = 0.5f * (1.0f + lmn_erf(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}
__device__ __forceinline__ float lmn_hswish(float x) {  // x * relu6(x+3)/6
  return x * fminf(fmaxf(x + 3.0f, 0.0f), 6.0f) * (1.0f / 6.0f);
}
__device__ __forceinline__ float lmn_dhswish(float x) {  // ATen hardswish_backward
  return x < -3.0f ? 0.0f : (x <= 3.0f ? x * (1.0f / 3.0f) + 0.5f : 1.0f);
}
__device__ __forceinline__ float lmn_hsigmoid(float x) { return fminf(fmaxf(x + 3.0f, 0.0f), 6.0f) * (1.0f / 6.0f); }
__device__ __forceinline__ float lmn_dhsigmoid(float x) { return (x > -3.0f && x < 3.0f) ? (1.0f / 6.0f) : 0.0f; }

__device__ __forceinline__ float lmn_act(float x, int act) {
  return act == LMN_ACT_HSWISH ? lmn_hswish(x) : (act == LMN_ACT_GELU ? lmn_gelu(x) : x);
}
__device__ __forceinline__ float lmn_dact(float x, int act) {
  return act == LMN_ACT_HSWISH ? lmn_dhswish(x) : (act == LMN_ACT_GELU ? lmn_dgelu(x) : 1.0f);
}
// four-wide forms: ONE wave-uniform branch on the activation kind per vector (the scalar forms, called per element of an
// unrolled loop, compiled to a compare-and-branch ladder per element: 16 ladders per conv tile and wave)
typedef float lmn_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ lmn_f32x4 lmn_act4(lmn_f32x4 x, int act) {
  lmn_f32x4 y = x;
  if (act == LMN_ACT_HSWISH) {
#pragma unroll
    for (int r = 0; r < 4; ++r) y[r] = lmn_hswish(x[r]);
  } else if (act == LMN_ACT_GELU) {
#pragma unroll
    for (int r = 0; r < 4; ++r) y[r] = lmn_gelu(x[r]);
  }
  return y;
}
__device__ __forceinline__ lmn_f32x4 lmn_dact4(lmn_f32x4 x, int act) {
  lmn_f32x4 y = lmn_f32x4{1.f, 1.f, 1.f, 1.f};
  if (act == LMN_ACT_HSWISH) {
#pragma unroll
    for (int r = 0; r < 4; ++r) y[r] = lmn_dhswish(x[r]);
  } else if (act == LMN_ACT_GELU) {
#pragma unroll
    for (int r = 0; r < 4; ++r) y[r] = lmn_dgelu(x[r]);
  }
  return y;
}

// ---------------------------------------------------------------- counter-based dropout mask
// keep(seed, idx) is a pure function of the dropout stream id and the element's linear index in its
// tensor, so forward and backward regenerate the same mask without storing it.
__device__ __forceinline__ uint32_t lmn_hash32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7feb352dU;
  x ^= x >> 15;
  x *= 0x846ca68bU;
  x ^= x >> 16;
  return x;
}
// returns 0 or 1/(1-p)
__device__ __forceinline__ float lmn_drop_scale(uint32_t seed, uint32_t idx, float p, float inv_keep) {
  const uint32_t h = lmn_hash32(idx * 0x9E3779B9U + seed) ^ lmn_hash32(seed ^ 0x85ebca6bU);
  const float u = (float)(lmn_hash32(h) >> 8) * (1.0f / 16777216.0f);
  return u >= p ? inv_keep : 0.0f;
}

// ---------------------------------------------------------------- wave helpers (64 lanes)
__device__ __forceinline__ float lmn_wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

// ---------------------------------------------------------------- raw buffer access
// Raw buffer access (bounds-checked by the descriptor: out-of-range loads return 0, stores are dropped).
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t BufRsrc;
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ BufRsrc make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
// 4 consecutive channels of storage type TA (16 B of fp32 / 8 B of bf16) at a per-lane byte offset (range-checked)
template <typename TA> __device__ __forceinline__ f32x4 buf_load4(BufRsrc r, unsigned byte_off) {
  if constexpr (sizeof(TA) == 4) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0));
  } else {
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)byte_off, 0, 0);
    return f32x4{lmn_bf16_lo(v.x), lmn_bf16_hi(v.x), lmn_bf16_lo(v.y), lmn_bf16_hi(v.y)};
  }
}
template <typename TA> __device__ __forceinline__ void buf_store4(BufRsrc r, unsigned byte_off, f32x4 v) {
  if constexpr (sizeof(TA) == 4) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, (int)byte_off, 0, 0);
  } else {
    __builtin_amdgcn_raw_buffer_store_b64(u32x2{lmn_pk_bf16(v[0], v[1]), lmn_pk_bf16(v[2], v[3])}, r, (int)byte_off, 0, 0);
  }
}
