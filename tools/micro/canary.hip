// Canary kernel for the "v_mfma_f32_16x16x32_bf16 corrupts kernels running beside it" question.
// It contains NO neighborhood-attention code.  Each resident wave keeps known patterns in VGPRs and LDS and re-computes a fixed set of
// results with the units the attention backward uses -- packed / scalar FMA chains, v_exp_f32 / v_rcp_f32, DPP and ds_bpermute
// reductions, float4 global loads of a constant buffer -- and compares every result bit for bit with what the same wave computed in
// its first iteration.  Run beside the -DLMN_MFMA_X2 conv instances on a second stream (tools/gpu_x2_canary.py):
//   * a canary hit (any category) => a co-resident kernel can damage another wave's state / results: the instruction (or the power /
//     clock behaviour it causes) is at fault, whatever the attention kernels do;
//   * no hit in N runs while the attention backward still goes wrong beside the same conv => the defect is in na.hip.
// Round 5 result: categories 12, 17, 20, 21, 22 -- packed fp32 instructions with src1 = HIGH half for both lanes -- are hit beside
// v_mfma_f32_16x16x32_bf16 on the same compute unit and nothing else is; tests/test_na_stress_gpu.py runs this file beside the product convs.
// Build: hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o libcanary.so canary.hip     (extern "C" launch_canary, launch_canary_lds)
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

constexpr int NR = 48;       // pattern registers per thread
constexpr int NL = 16;       // LDS floats per thread

__device__ __forceinline__ float pat(unsigned tid, unsigned k, unsigned seed) {   // a finite float with a busy mantissa
  unsigned h = (tid * 2654435761u) ^ (k * 40503u + seed * 97u + 0x9E3779B9u);
  h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
  return __uint_as_float(0x3F000000u | (h & 0x007FFFFFu));   // [0.5, 1)
}

// categories: 0 VGPR patterns, 1 LDS patterns, 2 FMA chain, 3 exp / rcp (softmax-like sum p * dp), 4 DPP reduction, 5 ds_bpermute
// reduction, 6 global reload
__global__ __launch_bounds__(256) void canary_kernel(unsigned* __restrict__ err, const float* __restrict__ cbuf, int ncbuf, int iters,
                                                     unsigned seed) {
  __shared__ float lds[256 * NL];
  const unsigned tid = threadIdx.x, gid = blockIdx.x * 256 + tid;
  float r[NR];
#pragma unroll
  for (int k = 0; k < NR; ++k) r[k] = pat(gid, k, seed);
#pragma unroll
  for (int k = 0; k < NL; ++k) lds[tid * NL + k] = pat(gid, 100 + k, seed);
  __syncthreads();
  float ref_fma = 0.f, ref_sm = 0.f, ref_dpp = 0.f, ref_bp = 0.f, ref_gl = 0.f;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 ref_pk = {0.f, 0.f}, ref_sel[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
  constexpr int NF = 8;
  f2 ref_f[NF];
  unsigned bad[7] = {0, 0, 0, 0, 0, 0, 0}, bad_pk = 0, bad_sel[4] = {0, 0, 0, 0}, bad_f[NF];
#pragma unroll
  for (int c = 0; c < NF; ++c) { ref_f[c] = f2{0.f, 0.f}; bad_f[c] = 0; }
  for (int it = 0; it < iters; ++it) {
    // the compiler must treat the patterns as live VGPR values that it cannot re-materialise
#pragma unroll
    for (int k = 0; k < NR; ++k) asm volatile("" : "+v"(r[k]));
    // (2) FMA chain over the pattern registers
    float a = 0.25f;
#pragma unroll
    for (int k = 0; k < NR; ++k) a = __builtin_fmaf(a, 0.5f, r[k] * r[(k + 7) % NR]);
    // (2b) the same chain with PACKED fp32 FMAs / multiplies (v_pk_fma_f32, v_pk_mul_f32: what f32x4 arithmetic compiles to at -O3) -> err[11]
    f2 pa = {0.25f, 0.125f};
#pragma unroll
    for (int k = 0; k < NR / 2; ++k) {
      const f2 x = {r[2 * k], r[2 * k + 1]}, y = {r[(2 * k + 7) % NR], r[(2 * k + 8) % NR]};
      f2 m;
      asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(m) : "v"(x), "v"(y));
      asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(pa) : "v"(pa), "v"(f2{0.5f, 0.5f}), "v"(m));
    }
    // (2c) packed fp32 with OPERAND-HALF SELECTION, the forms the SLP vectoriser gives the attention backward at head_dim 2 (forensics of
    //      round 5: only the odd head of a channel quad goes wrong, and that head's sum p * dp is the chain
    //      `v_pk_fma_f32 d, a, b, d op_sel:[0,1,0] op_sel_hi:[0,1,1]`): err[12] that chain, err[13] v_pk_mov_b32 op_sel:[1,0] + v_pk_add_f32,
    //      err[14] v_pk_mul_f32 op_sel:[1,0] op_sel_hi:[0,1], err[15] v_pk_fma_f32 op_sel_hi:[0,0,1] (the EVEN head's form: control)
    f2 s0 = {0.25f, 0.125f}, s1 = s0, s2 = s0, s3 = s0;
#pragma unroll
    for (int k = 0; k < NR / 2; ++k) {
      const f2 x = {r[2 * k], r[2 * k + 1]}, y = {r[(2 * k + 7) % NR], r[(2 * k + 8) % NR]};
      asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[0,1,1]" : "+v"(s0) : "v"(x), "v"(y));
      f2 m;
      asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(m) : "v"(x), "v"(y));
      asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(s1) : "v"(m));
      s1 = s1 * 0.5f;
      asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(m) : "v"(x), "v"(y));
      s2 = s2 * 0.5f + m;
      asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,0,1]" : "+v"(s3) : "v"(x), "v"(y));
      s0 = s0 * 0.5f; s3 = s3 * 0.5f;
    }
    // (2d) which operand-half selections are affected?  err[16 + c], one chain per form (results compared with the first iteration):
    //   0 pk_fma op_sel:[1,0,0] op_sel_hi:[1,0,1]   src0: high half for BOTH lanes
    //   1 pk_fma op_sel:[0,1,0] op_sel_hi:[1,1,1]   src1: high half for both lanes, src0 default     (the failing form with src0 not broadcast)
    //   2 pk_fma op_sel:[0,0,1] op_sel_hi:[1,1,1]   src2: high half for both lanes
    //   3 pk_fma op_sel:[1,1,1] op_sel_hi:[0,0,0]   all three swapped
    //   4 pk_mul op_sel:[0,1]   op_sel_hi:[0,1]     src0 low for both, src1 high for both           (the failing selection, no addend)
    //   5 pk_add op_sel:[0,1]   op_sel_hi:[1,1]     src1 high for both
    //   6 pk_fma op_sel:[0,1,0] op_sel_hi:[0,1,1]   the failing form again, but on COPIES of the operands made by v_mov_b32 right before
    //   7 v_fma_f32 x2 (no packing)                 control
    f2 gm[NF];
#pragma unroll
    for (int c = 0; c < NF; ++c) gm[c] = f2{0.25f, 0.125f};
#pragma unroll
    for (int k = 0; k < NR / 2; ++k) {
      const f2 x = {r[2 * k], r[2 * k + 1]}, y = {r[(2 * k + 7) % NR], r[(2 * k + 8) % NR]};
      asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,1]" : "+v"(gm[0]) : "v"(x), "v"(y));
      asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(gm[1]) : "v"(x), "v"(y));
      asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,1] op_sel_hi:[1,1,1]" : "+v"(gm[2]) : "v"(x), "v"(y));
      asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,1] op_sel_hi:[0,0,0]" : "+v"(gm[3]) : "v"(x), "v"(y));
      f2 m;
      asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,1]" : "=v"(m) : "v"(x), "v"(y));
      gm[4] = gm[4] * 0.5f + m;
      asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(m) : "v"(x), "v"(y));
      gm[5] = gm[5] * 0.5f + m;
      f2 xc, yc;
      asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=&v"(xc.x), "=&v"(xc.y) : "v"(x.x), "v"(x.y));
      asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=&v"(yc.x), "=&v"(yc.y) : "v"(y.x), "v"(y.y));
      asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[0,1,1]" : "+v"(gm[6]) : "v"(xc), "v"(yc));
      gm[7].x = __builtin_fmaf(x.x, y.y, gm[7].x); gm[7].y = __builtin_fmaf(x.x, y.y, gm[7].y);
#pragma unroll
      for (int c = 0; c < NF; ++c) if (c != 4 && c != 5) gm[c] = gm[c] * 0.5f;
    }
    if (it == 0) {
#pragma unroll
      for (int c = 0; c < NF; ++c) ref_f[c] = gm[c];
    } else {
#pragma unroll
      for (int c = 0; c < NF; ++c) bad_f[c] += (__float_as_uint(gm[c].x) != __float_as_uint(ref_f[c].x)) | (__float_as_uint(gm[c].y) != __float_as_uint(ref_f[c].y));
    }
    // (3) softmax-like: nine scores -> max -> exp2 -> sum -> p; s = sum p * dp  (the quantity that went wrong in the attention backward)
    float sc[9], mx = -1e30f;
#pragma unroll
    for (int n = 0; n < 9; ++n) { sc[n] = r[n] * 4.f - r[n + 9] * 3.f; mx = fmaxf(mx, sc[n]); }
    float den = 0.f, pd = 0.f;
#pragma unroll
    for (int n = 0; n < 9; ++n) { const float e = __builtin_amdgcn_exp2f(sc[n] - mx); den += e; pd = __builtin_fmaf(e, r[18 + n] - 0.75f, pd); }
    const float sm = pd * __builtin_amdgcn_rcpf(den);
    // (4) DPP wave reduction, (5) ds_bpermute butterfly
    float d = r[30];
    d += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(d), 0x111, 0xF, 0xF, true));
    d += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(d), 0x112, 0xF, 0xF, true));
    d += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(d), 0x114, 0xF, 0xF, true));
    d += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(d), 0x118, 0xF, 0xF, true));
    float b = r[31];
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) b += __shfl_xor(b, m, 64);
    // (6) global reload of a constant buffer (L1 / L2 / TA path)
    float g = 0.f;
    {
      const float4* c4 = reinterpret_cast<const float4*>(cbuf);
      const int n4 = ncbuf >> 2;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float4 v = c4[(gid * 4 + k + (unsigned)it * 977u) % (unsigned)n4];
        g += (v.x + v.y) + (v.z + v.w) * (float)(((gid * 4 + k + (unsigned)it * 977u) % (unsigned)n4) & 7);
      }
    }
    // references from the first iteration; the reload address moves with `it`, so its check is against the buffer's closed form
    if (it == 0) { ref_sel[0] = s0; ref_sel[1] = s1; ref_sel[2] = s2; ref_sel[3] = s3; }
    else {
      const f2 sv[4] = {s0, s1, s2, s3};
#pragma unroll
      for (int c = 0; c < 4; ++c) bad_sel[c] += (__float_as_uint(sv[c].x) != __float_as_uint(ref_sel[c].x)) | (__float_as_uint(sv[c].y) != __float_as_uint(ref_sel[c].y));
    }
    if (it == 0) { ref_fma = a; ref_sm = sm; ref_dpp = d; ref_bp = b; ref_pk = pa; }
    else {
      bad_pk += (__float_as_uint(pa.x) != __float_as_uint(ref_pk.x)) | (__float_as_uint(pa.y) != __float_as_uint(ref_pk.y));
      bad[2] += __float_as_uint(a) != __float_as_uint(ref_fma);
      bad[3] += __float_as_uint(sm) != __float_as_uint(ref_sm);
      bad[4] += __float_as_uint(d) != __float_as_uint(ref_dpp);
      bad[5] += __float_as_uint(b) != __float_as_uint(ref_bp);
    }
    {   // cbuf[i] = (i & 1023) * 0.5 (exact in fp32): recompute the expected sum
      const int n4 = ncbuf >> 2;
      float e = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const unsigned j = (gid * 4 + k + (unsigned)it * 977u) % (unsigned)n4;
        const float v0 = (float)((4 * j) & 1023) * 0.5f, v1 = (float)((4 * j + 1) & 1023) * 0.5f, v2 = (float)((4 * j + 2) & 1023) * 0.5f,
                    v3 = (float)((4 * j + 3) & 1023) * 0.5f;
        e += (v0 + v1) + (v2 + v3) * (float)(j & 7);
      }
      bad[6] += __float_as_uint(g) != __float_as_uint(e);
      ref_gl = e;
    }
    // (0), (1) the patterns themselves
#pragma unroll
    for (int k = 0; k < NR; ++k) bad[0] += __float_as_uint(r[k]) != __float_as_uint(pat(gid, k, seed));
#pragma unroll
    for (int k = 0; k < NL; ++k) bad[1] += __float_as_uint(lds[tid * NL + k]) != __float_as_uint(pat(gid, 100 + k, seed));
  }
  (void)ref_gl;
#pragma unroll
  for (int c = 0; c < 7; ++c)
    if (bad[c]) atomicAdd(err + c, bad[c]);
  if (bad_pk) atomicAdd(err + 11, bad_pk);
#pragma unroll
  for (int c = 0; c < 4; ++c)
    if (bad_sel[c]) atomicAdd(err + 12 + c, bad_sel[c]);
#pragma unroll
  for (int c = 0; c < NF; ++c)
    if (bad_f[c]) atomicAdd(err + 16 + c, bad_f[c]);
  if (tid == 0 && blockIdx.x == 0) atomicAdd(err + 7, 1u);   // launches that ran to the end
}

// Second canary (round 5, after the CU-mask experiment: the failure needs the x2 convs on the SAME compute unit, and only the LDS-tiled
// attention kernels fail): the staging -> barrier -> window-read -> barrier cycle of those kernels with no attention arithmetic.
// 512-thread blocks, dynamic LDS of `nslots` float4: every iteration all threads write the tile (even slots: a closed-form pattern of
// (block, slot, iteration); odd slots: float4 loads of the constant buffer, as the k / v staging does), __syncthreads, every thread reads
// 18 slots other threads wrote (ds_read_b128, window-like offsets) and checks them against the closed forms, __syncthreads.
// err[8] pattern slots wrong, err[9] global-sourced slots wrong, err[10] launches.
__global__ __launch_bounds__(512) void canary_lds_kernel(unsigned* __restrict__ err, const float* __restrict__ cbuf, int ncbuf, int iters,
                                                         unsigned seed, int nslots) {
  extern __shared__ float4 sl[];
  const unsigned tid = threadIdx.x, NT = 512;
  const float4* c4 = reinterpret_cast<const float4*>(cbuf);
  const unsigned n4 = (unsigned)ncbuf >> 2;
  unsigned bad_p = 0, bad_g = 0;
  for (int it = 0; it < iters; ++it) {
    for (unsigned i0 = tid; i0 < (unsigned)nslots; i0 += 4 * NT) {   // four loads in flight per thread, then four LDS writes (the staging loop's shape)
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const unsigned i = i0 + u * NT;
        const unsigned is = i < (unsigned)nslots ? i : 0;
        if (is & 1) v[u] = c4[(blockIdx.x * 131u + is * 7u + (unsigned)it * 977u) % n4];
        else { const unsigned key = blockIdx.x * 4099u + is; v[u] = float4{pat(key, 4 * it, seed), pat(key, 4 * it + 1, seed), pat(key, 4 * it + 2, seed), pat(key, 4 * it + 3, seed)}; }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const unsigned i = i0 + u * NT;
        if (i < (unsigned)nslots) sl[i] = v[u];
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 18; ++k) {
      const unsigned i = (tid * 2u + (unsigned)(k >> 1) * 42u + (unsigned)(k & 1) + (unsigned)it * 6u) % (unsigned)nslots;
      const float4 v = sl[i];
      if (i & 1) {
        const unsigned j = (blockIdx.x * 131u + i * 7u + (unsigned)it * 977u) % n4;
        const float e0 = (float)((4 * j) & 1023) * 0.5f, e1 = (float)((4 * j + 1) & 1023) * 0.5f, e2 = (float)((4 * j + 2) & 1023) * 0.5f,
                    e3 = (float)((4 * j + 3) & 1023) * 0.5f;
        bad_g += (v.x != e0) | (v.y != e1) | (v.z != e2) | (v.w != e3);
      } else {
        const unsigned key = blockIdx.x * 4099u + i;
        bad_p += (__float_as_uint(v.x) != __float_as_uint(pat(key, 4 * it, seed))) | (__float_as_uint(v.y) != __float_as_uint(pat(key, 4 * it + 1, seed))) |
                 (__float_as_uint(v.z) != __float_as_uint(pat(key, 4 * it + 2, seed))) | (__float_as_uint(v.w) != __float_as_uint(pat(key, 4 * it + 3, seed)));
      }
    }
    __syncthreads();
  }
  if (bad_p) atomicAdd(err + 8, bad_p);
  if (bad_g) atomicAdd(err + 9, bad_g);
  if (tid == 0 && blockIdx.x == 0) atomicAdd(err + 10, 1u);
}

}  // namespace

extern "C" int launch_canary_lds(unsigned* err, const float* cbuf, int ncbuf, int blocks, int iters, unsigned seed, int nslots, void* stream) {
  (void)hipFuncSetAttribute((const void*)canary_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, nslots * 16);
  hipLaunchKernelGGL(canary_lds_kernel, dim3(blocks), dim3(512), (size_t)nslots * 16, (hipStream_t)stream, err, cbuf, ncbuf, iters, seed, nslots);
  return (int)hipGetLastError();
}

extern "C" int launch_canary(unsigned* err, const float* cbuf, int ncbuf, int blocks, int iters, unsigned seed, void* stream) {
  hipLaunchKernelGGL(canary_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, err, cbuf, ncbuf, iters, seed);
  return (int)hipGetLastError();
}
