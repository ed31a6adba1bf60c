// f32 VALU issue-rate probe for gfx950:  hipcc --offload-arch=gfx950 -O3 -o valu_peak valu_peak.hip
// Back-to-back independent FMAs with a wave-uniform (SGPR) multiplier, as the depthwise kernels issue them:
// v_fma_f32 (64 FMAs per instruction) against v_pk_fma_f32 (128), 1..4 waves per SIMD.  Prints FMAs per cycle and SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(256) void peak(float* out, const float* w, unsigned long long* clk, int iters) {
  constexpr int NA = 16;
  float a[NA]; f32x2 p[NA];
  for (int i = 0; i < NA; ++i) { a[i] = threadIdx.x * 1e-3f + i; p[i] = f32x2{a[i], a[i] + 1.f}; }
  const float x = 1.0f + threadIdx.x * 1e-4f;
  const f32x2 x2 = f32x2{x, x + 0.5f};
  float ws[8]; f32x2 wp[8];
  for (int k = 0; k < 8; ++k) { ws[k] = w[k]; wp[k] = f32x2{w[k], w[8 + k]}; }   // wave-uniform -> SGPRs
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        if constexpr (KIND == 0) a[i] = __builtin_fmaf(ws[k], x, a[i]);
        if constexpr (KIND == 1) p[i] = wp[k] * x2 + p[i];
        if constexpr (KIND == 2) { asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "s"(wp[k]), "v"(x2)); }
        if constexpr (KIND == 3) { asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(wp[k]), "v"(x2)); }
      }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < NA; ++i) s += a[i] + p[i].x + p[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <int KIND> void run(const char* name, double fma_per_inst, int wps) {
  const int blocks = 256 * wps, iters = 4000;
  float *out, *w; unsigned long long* clk;
  hipMalloc(&out, blocks * 256 * 4); hipMalloc(&clk, blocks * 8); hipMalloc(&w, 64);
  hipMemset(w, 0, 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  peak<KIND><<<blocks, 256>>>(out, w, clk, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  peak<KIND><<<blocks, 256>>>(out, w, clk, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks);
  hipMemcpy(h.data(), clk, blocks * 8, hipMemcpyDeviceToHost);
  double sc = 0;
  for (int i = 0; i < blocks; ++i) sc += h[i];
  const double ninst = (double)iters * 8 * 16;  // per wave
  printf("%-34s %d waves/SIMD: %6.1f TFLOP/s  cycles per instruction and SIMD = %5.2f  (FMAs per cycle and SIMD = %5.1f)\n", name, wps,
         blocks * 4.0 * ninst * fma_per_inst * 2 / (ms * 1e-3) / 1e12, sc / blocks / (ninst * wps), fma_per_inst * ninst * wps / (sc / blocks));
  hipFree(out); hipFree(clk); hipFree(w);
}

int main() {
  for (int w = 1; w <= 4; w *= 2) run<0>("v_fma_f32 (SGPR x VGPR)", 64.0, w);
  for (int w = 1; w <= 4; w *= 2) run<1>("f32x2 fma (compiler)", 128.0, w);
  for (int w = 1; w <= 4; w *= 2) run<2>("v_pk_fma_f32 (SGPR pair x VGPR)", 128.0, w);
  for (int w = 1; w <= 4; w *= 2) run<3>("v_pk_fma_f32 (VGPR x VGPR)", 128.0, w);
  return 0;
}
