// Which compute unit does a block run on?  out[block] = (XCC_ID << 16) | (SE_ID << 8) | CU_ID  (gfx950: HW_REG_XCC_ID = 20,
// HW_REG_HW_ID = 4 with CU_ID in bits 11:8, SH_ID bit 12, SE_ID bits 15:13).  tools/gpu_x2_cumask.py uses it to verify that two
// streams created with hipExtStreamCreateWithCUMask really run on disjoint / identical sets of compute units.
// Build: hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o libcuid.so cuid.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ __launch_bounds__(256) void cuid_kernel(uint32_t* __restrict__ out, int spin) {
  uint32_t hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  // keep the block resident for a while so that the grid spreads over every unit the stream may use
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)spin) {}
  if (threadIdx.x == 0) out[blockIdx.x] = ((xcc & 0xF) << 16) | (((hw >> 13) & 7) << 8) | ((hw >> 8) & 0xF) | (((hw >> 12) & 1) << 4);
}

extern "C" int launch_cuid(uint32_t* out, int nblocks, int spin, hipStream_t st) {
  hipLaunchKernelGGL(cuid_kernel, dim3(nblocks), dim3(256), 0, st, out, spin);
  return (int)hipGetLastError();
}
