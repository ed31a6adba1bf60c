// conv_tileM_kernel instances (M-split form for wide layers: 1x1 incl. row-planar operands, 3x3).
#include "conv_tileM.h"

int lmn_launch_conv_tileM(const ConvParams& T, dim3 mgrid, size_t msh, hipStream_t st, int taps, int ncw, int pm, int ek, bool rp, bool ln, bool up) {
  if (up) {   // bilinear x2 on load (LMN_SRC_UP2): plain-epilogue 3x3 instances of their own
#define LMN_CU(NN) do { if (pm == 2) LMN_LAUNCH((conv_tileM_kernel<9, NN, 0, 2, false, true>), mgrid, dim3(256), msh, st, T); \
                        else if (pm == 1) LMN_LAUNCH((conv_tileM_kernel<9, NN, 0, 1, false, true>), mgrid, dim3(256), msh, st, T); \
                        else LMN_LAUNCH((conv_tileM_kernel<9, NN, 0, 0, false, true>), mgrid, dim3(256), msh, st, T); } while (0)
    if (ncw == 2) LMN_CU(2); else LMN_CU(1);
#undef LMN_CU
    return 0;
  }
  if (ln) {   // LayerNorm on load (LMN_SRC_LN): plain-epilogue NHWC 1x1 instances of their own
#define LMN_CL(NN) do { if (pm == 2) LMN_LAUNCH((conv_tileM_kernel<1, NN, 0, 2, true>), mgrid, dim3(256), msh, st, T); \
                        else if (pm == 1) LMN_LAUNCH((conv_tileM_kernel<1, NN, 0, 1, true>), mgrid, dim3(256), msh, st, T); \
                        else LMN_LAUNCH((conv_tileM_kernel<1, NN, 0, 0, true>), mgrid, dim3(256), msh, st, T); } while (0)
    if (ncw == 2) LMN_CL(2); else LMN_CL(1);
#undef LMN_CL
    return 0;
  }
// (3x3 calls on the small maps stage K chunks of up to 128 channels: more than the default 64 KB of dynamic LDS)
#define LMN_BIGLDS(kern) do { if (msh > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)msh); } while (0)
#define LMN_CM(TT, NN, BFV)                                                                              \
  do {                                                                                                   \
    switch ((TT) == 1 ? ek : (ek > 2 ? 1 : ek)) {                                                        \
      case 0: LMN_BIGLDS((conv_tileM_kernel<TT, NN, 0, BFV>)); LMN_LAUNCH((conv_tileM_kernel<TT, NN, 0, BFV>), mgrid, dim3(256), msh, st, T); break;   \
      case 2: LMN_BIGLDS((conv_tileM_kernel<TT, NN, 2, BFV>)); LMN_LAUNCH((conv_tileM_kernel<TT, NN, 2, BFV>), mgrid, dim3(256), msh, st, T); break;   \
      case 3: LMN_LAUNCH((conv_tileM_kernel<1, NN, 3, BFV>), mgrid, dim3(256), msh, st, T); break;    \
      case 4: LMN_LAUNCH((conv_tileM_kernel<1, NN, 4, BFV>), mgrid, dim3(256), msh, st, T); break;    \
      case 5: LMN_LAUNCH((conv_tileM_kernel<1, NN, 5, BFV>), mgrid, dim3(256), msh, st, T); break;    \
      default: LMN_BIGLDS((conv_tileM_kernel<TT, NN, 1, BFV>)); LMN_LAUNCH((conv_tileM_kernel<TT, NN, 1, BFV>), mgrid, dim3(256), msh, st, T); break;  \
    }                                                                                                    \
  } while (0)
#define LMN_CMB(TT, NN) do { if (pm == 2) LMN_CM(TT, NN, 2); else if (pm == 1) LMN_CM(TT, NN, 1); else LMN_CM(TT, NN, 0); } while (0)
#define LMN_CMR(NN) do { if (pm == 2) LMN_CM(1, NN, 6); else if (pm == 1) LMN_CM(1, NN, 5); else LMN_CM(1, NN, 4); } while (0)
  if (taps == 1 && rp) { if (ncw == 2) LMN_CMR(2); else LMN_CMR(1); }   // (row-planar operands: their own instances)
  else if (taps == 1) { if (ncw == 2) LMN_CMB(1, 2); else LMN_CMB(1, 1); }
  else { if (ncw == 2) LMN_CMB(9, 2); else LMN_CMB(9, 1); }
#undef LMN_CMR
#undef LMN_CMB
#undef LMN_CM
#undef LMN_BIGLDS
  return 0;
}
