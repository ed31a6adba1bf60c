// conv_tile_kernel instances for 3x3 convs with FOUR pixel groups per wave (tiles of up to 256 pixels, NPG = 4): forward / stride-1 data
// gradient with one or two cout tiles per block.  A translation unit of its own so that it compiles beside conv_tile_3x3.hip.
#include "conv_tile.h"

int lmn_launch_conv_tile_3x3g(const ConvParams& T, dim3 grid, size_t shmem, hipStream_t st, int tnct, int pm, int ek, bool wlk) {
  const int ekk = ek > 2 ? 1 : ek;   // (the BatchNorm-backward / SE-gradient epilogue instances exist for 1x1 convs only)
#define LMN_CG(NN, BFV)                                                                                  \
  do {                                                                                                   \
    switch (ekk) {                                                                                       \
      case 0: if (wlk) LMN_LAUNCH((conv_tile_kernel<9, NN, 0, false, BFV, true, false, false, 4>), grid, dim3(256), shmem, st, T); \
              else LMN_LAUNCH((conv_tile_kernel<9, NN, 0, false, BFV, false, false, false, 4>), grid, dim3(256), shmem, st, T); break;   \
      case 2: if (wlk) LMN_LAUNCH((conv_tile_kernel<9, NN, 2, false, BFV, true, false, false, 4>), grid, dim3(256), shmem, st, T); \
              else LMN_LAUNCH((conv_tile_kernel<9, NN, 2, false, BFV, false, false, false, 4>), grid, dim3(256), shmem, st, T); break;   \
      default: if (wlk) LMN_LAUNCH((conv_tile_kernel<9, NN, 1, false, BFV, true, false, false, 4>), grid, dim3(256), shmem, st, T); \
               else LMN_LAUNCH((conv_tile_kernel<9, NN, 1, false, BFV, false, false, false, 4>), grid, dim3(256), shmem, st, T); break;  \
    }                                                                                                    \
  } while (0)
  switch ((tnct >= 2 ? 2 : 1) * 4 + pm) {
    case 4: LMN_CG(1, 0); break;
    case 5: LMN_CG(1, 1); break;
    case 6: LMN_CG(1, 2); break;
    case 9: LMN_CG(2, 1); break;
    case 10: LMN_CG(2, 2); break;
    default: LMN_CG(2, 0); break;
  }
#undef LMN_CG
  return 0;
}
