// conv_tile_kernel instances for 3x3 convs (N-split form): forward / stride-1 data gradient, with or without LDS-staged weights,
// and the parity-class data gradient of the stride-2 convs (S2T).
#include "conv_tile.h"

int lmn_launch_conv_tile_3x3(const ConvParams& T, dim3 grid, size_t shmem, hipStream_t st, int tnct, int pm, int ek, bool wlk, bool up) {
  if (up) {   // bilinear x2 on load (LMN_SRC_UP2): plain-epilogue instances of their own
#define LMN_CU(NN, BFV)                                                                                               \
  do {                                                                                                                \
    if (wlk) LMN_LAUNCH((conv_tile_kernel<9, NN, 0, false, BFV, true, false, true>), grid, dim3(256), shmem, st, T);   \
    else LMN_LAUNCH((conv_tile_kernel<9, NN, 0, false, BFV, false, false, true>), grid, dim3(256), shmem, st, T);      \
  } while (0)
    switch ((tnct > 3 ? 3 : tnct) * 4 + pm) {
      case 4: LMN_CU(1, 0); break;
      case 5: LMN_CU(1, 1); break;
      case 6: LMN_CU(1, 2); break;
      case 8: LMN_CU(2, 0); break;
      case 9: LMN_CU(2, 1); break;
      case 10: LMN_CU(2, 2); break;
      case 13: LMN_CU(3, 1); break;
      case 14: LMN_CU(3, 2); break;
      default: LMN_CU(3, 0); break;
    }
#undef LMN_CU
    return 0;
  }
  const int ekk = ek > 2 ? 1 : ek;   // (the BatchNorm-backward / SE-gradient epilogue instances exist for 1x1 convs only)
#define LMN_CT(NN, BFV)                                                                                  \
  do {                                                                                                   \
    switch (ekk) {                                                                                       \
      case 0: if (wlk) LMN_LAUNCH((conv_tile_kernel<9, NN, 0, false, BFV, true>), grid, dim3(256), shmem, st, T); \
              else LMN_LAUNCH((conv_tile_kernel<9, NN, 0, false, BFV>), grid, dim3(256), shmem, st, T); break;   \
      case 2: if (wlk) LMN_LAUNCH((conv_tile_kernel<9, NN, 2, false, BFV, true>), grid, dim3(256), shmem, st, T); \
              else LMN_LAUNCH((conv_tile_kernel<9, NN, 2, false, BFV>), grid, dim3(256), shmem, st, T); break;   \
      default: if (wlk) LMN_LAUNCH((conv_tile_kernel<9, NN, 1, false, BFV, true>), grid, dim3(256), shmem, st, T); \
               else LMN_LAUNCH((conv_tile_kernel<9, NN, 1, false, BFV>), grid, dim3(256), shmem, st, T); break;  \
    }                                                                                                    \
  } while (0)
  switch ((tnct > 3 ? 3 : tnct) * 4 + pm) {
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
#undef LMN_CT
  return 0;
}

int lmn_launch_conv_tile_s2t(const ConvParams& T, dim3 zgrid, size_t shmem, hipStream_t st, int tnct, int pm, int ek) {
#define LMN_CZ(NN, BFV)                                                                                       \
  do {                                                                                                        \
    if (ek == 0) LMN_LAUNCH((conv_tile_kernel<9, NN, 0, true, BFV>), zgrid, dim3(256), shmem, st, T); \
    else LMN_LAUNCH((conv_tile_kernel<9, NN, 1, true, BFV>), zgrid, dim3(256), shmem, st, T);         \
  } while (0)
#define LMN_CZB(NN) do { if (pm == 2) LMN_CZ(NN, 2); else if (pm == 1) LMN_CZ(NN, 1); else LMN_CZ(NN, 0); } while (0)
  switch (tnct) {
    case 1: LMN_CZB(1); break;
    case 2: LMN_CZB(2); break;
    default: LMN_CZB(3); break;
  }
#undef LMN_CZB
#undef LMN_CZ
  return 0;
}
