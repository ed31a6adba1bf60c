// MFMA issue-rate / shader-clock probe for gfx950:  hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip
// Prints achieved TFLOP/s of back-to-back independent MFMAs (fp32 16x16x4, bf16 16x16x16, bf16 32x32x8) at 1..4 waves
// per SIMD, and the shader clock seen by s_memtime against the 100 MHz s_memrealtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ __launch_bounds__(256) void peak(float* out, unsigned long long* clk, int iters) {
  f32x4 acc[8];
  f32x16 big[2];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
  for (int i = 0; i < 2; ++i) for (int k = 0; k < 16; ++k) big[i][k] = 0.f;
  const float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  const s16x4 ah = {1, 2, 3, 4}, bh = {5, 6, 7, 8};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if constexpr (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
      if constexpr (KIND == 1) acc[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, bh, acc[i], 0, 0, 0);
      if constexpr (KIND == 2) big[i & 1] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(ah, bh, big[i & 1], 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
  s += big[0][0] + big[1][5];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int KIND> void run(const char* name, double flops_per_mfma, int wps) {
  const int blocks = 256 * wps, iters = 20000;
  float* out; unsigned long long* clk;
  hipMalloc(&out, blocks * 256 * 4); hipMalloc(&clk, blocks * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  peak<KIND><<<blocks, 256>>>(out, clk, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  peak<KIND><<<blocks, 256>>>(out, clk, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks * 2);
  hipMemcpy(h.data(), clk, blocks * 16, hipMemcpyDeviceToHost);
  double sc = 0, rc = 0;
  for (int i = 0; i < blocks; ++i) { sc += h[2 * i]; rc += h[2 * i + 1]; }
  const double nm = (double)blocks * 4 * iters * 8;
  printf("%-22s %d waves/SIMD: %7.1f TFLOP/s  %.2f ms  s_memtime/s_memrealtime = %.2f (x100 MHz = %.0f MHz)  cycles per MFMA per SIMD (s_memtime) = %.1f\n",
         name, wps, nm * flops_per_mfma / (ms * 1e-3) / 1e12, ms, sc / rc, sc / rc * 100, sc / blocks / ((double)iters * 8 * wps));
  hipFree(out); hipFree(clk);
}

int main() {
  for (int w = 1; w <= 4; w *= 2) run<0>("f32 16x16x4", 2048.0, w);
  for (int w = 1; w <= 4; w *= 2) run<1>("bf16 16x16x16", 8192.0, w);
  for (int w = 1; w <= 2; w *= 2) run<2>("bf16 32x32x8", 16384.0, w);
  return 0;
}
