// Canary kernel for the "v_mfma_f32_16x16x32_bf16 corrupts kernels running beside it" question.
// It contains NO neighborhood-attention code.  Each resident wave keeps known patterns in VGPRs and LDS and re-computes a fixed set of
// results with the units the attention backward uses -- packed / scalar FMA chains, v_exp_f32 / v_rcp_f32, DPP and ds_bpermute
// reductions, float4 global loads of a constant buffer -- and compares every result bit for bit with what the same wave computed in
// its first iteration.  Run beside the -DLMN_MFMA_X2 conv instances on a second stream (tools/gpu_x2_canary.py):
//   * a canary hit (any category) => a co-resident kernel can damage another wave's state / results: the instruction (or the power /
//     clock behaviour it causes) is at fault, whatever the attention kernels do;
//   * no hit in N runs while the attention backward still goes wrong beside the same conv => the defect is in na.hip.
// Build: hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o libcanary.so canary.hip     (extern "C" launch_canary)
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
  unsigned bad[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    // the compiler must treat the patterns as live VGPR values that it cannot re-materialise
#pragma unroll
    for (int k = 0; k < NR; ++k) asm volatile("" : "+v"(r[k]));
    // (2) FMA chain over the pattern registers
    float a = 0.25f;
#pragma unroll
    for (int k = 0; k < NR; ++k) a = __builtin_fmaf(a, 0.5f, r[k] * r[(k + 7) % NR]);
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
    if (it == 0) { ref_fma = a; ref_sm = sm; ref_dpp = d; ref_bp = b; }
    else {
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
  if (tid == 0 && blockIdx.x == 0) atomicAdd(err + 7, 1u);   // launches that ran to the end
}

}  // namespace

extern "C" int launch_canary(unsigned* err, const float* cbuf, int ncbuf, int blocks, int iters, unsigned seed, void* stream) {
  hipLaunchKernelGGL(canary_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, err, cbuf, ncbuf, iters, seed);
  return (int)hipGetLastError();
}
