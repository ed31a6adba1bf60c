// conv_tile_kernel instances for 1x1 convs / Linear layers (N-split form; NHWC and row-planar operands).
#include "conv_tile.h"

int lmn_launch_conv_tile_1x1(const ConvParams& T, dim3 grid, size_t shmem, hipStream_t st, int tnct, int pm, int ek, bool rp, bool ln) {
  if (ln) {   // LayerNorm on load (LMN_SRC_LN): plain-epilogue NHWC instances of their own
#define LMN_CL(NN, BFV) LMN_LAUNCH((conv_tile_kernel<1, NN, 0, false, BFV, false, true>), grid, dim3(256), shmem, st, T)
    switch ((tnct > 3 ? 3 : tnct) * 4 + pm) {
      case 4: LMN_CL(1, 0); break;
      case 5: LMN_CL(1, 1); break;
      case 6: LMN_CL(1, 2); break;
      case 8: LMN_CL(2, 0); break;
      case 9: LMN_CL(2, 1); break;
      case 10: LMN_CL(2, 2); break;
      case 13: LMN_CL(3, 1); break;
      case 14: LMN_CL(3, 2); break;
      default: LMN_CL(3, 0); break;
    }
#undef LMN_CL
    return 0;
  }
#define LMN_CT(NN, BFV)                                                                                  \
  do {                                                                                                   \
    switch (ek) {                                                                                        \
      case 0: LMN_LAUNCH((conv_tile_kernel<1, NN, 0, false, BFV>), grid, dim3(256), shmem, st, T); break; \
      case 2: LMN_LAUNCH((conv_tile_kernel<1, NN, 2, false, BFV>), grid, dim3(256), shmem, st, T); break; \
      case 3: LMN_LAUNCH((conv_tile_kernel<1, NN, 3, false, BFV>), grid, dim3(256), shmem, st, T); break; \
      case 4: LMN_LAUNCH((conv_tile_kernel<1, NN, 4, false, BFV>), grid, dim3(256), shmem, st, T); break; \
      case 5: LMN_LAUNCH((conv_tile_kernel<1, NN, 5, false, BFV>), grid, dim3(256), shmem, st, T); break; \
      LMN_CASE6(NN, BFV) \
      default: LMN_LAUNCH((conv_tile_kernel<1, NN, 1, false, BFV>), grid, dim3(256), shmem, st, T); break; \
    }                                                                                                    \
  } while (0)
  const int key = (tnct > 3 ? 3 : tnct) * 4 + pm;
  if (rp) {   // row-planar operands: their own instances (precision mode | 4); no LayerNorm-backward epilogue (NHWC calls only)
#define LMN_CASE6(NN, BFV)
    switch (key) {
      case 4: LMN_CT(1, 4); break;
      case 5: LMN_CT(1, 5); break;
      case 6: LMN_CT(1, 6); break;
      case 8: LMN_CT(2, 4); break;
      case 9: LMN_CT(2, 5); break;
      case 10: LMN_CT(2, 6); break;
      case 13: LMN_CT(3, 5); break;
      case 14: LMN_CT(3, 6); break;
      default: LMN_CT(3, 4); break;
    }
#undef LMN_CASE6
  } else {
#define LMN_CASE6(NN, BFV) case 6: LMN_LAUNCH((conv_tile_kernel<1, NN, 6, false, BFV>), grid, dim3(256), shmem, st, T); break;
    switch (key) {
      case 4: LMN_CT(1, 0); break;
      case 5: LMN_CT(1, 1); break;
      case 6: LMN_CT(1, 2); break;
      case 8: LMN_CT(2, 0); break;
      case 9: LMN_CT(2, 1); break;
      case 10: LMN_CT(2, 2); break;
      case 13: LMN_CT(3, 1); break;
      case 14: LMN_CT(3, 2); break;
      default: LMN_CT(3, 0); break;
    }
#undef LMN_CASE6
  }
#undef LMN_CT
  return 0;
}
