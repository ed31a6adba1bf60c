// Shared declarations of the dense-convolution family (conv_fwd.hip, conv_tile_1x1.hip, conv_tile_3x3.hip, conv_tileM.hip,
// conv_wgrad.hip): kernel parameter blocks, bf16 operand helpers, the phase-clock macros of the debug build and the launchers
// the host entry (conv_fwd.hip) calls.  The family is split over several translation units so that they compile in parallel
// (one file took 3 min 45 s); `make timing` builds them as ONE unit (conv_unity.hip) so that the phase-clock arrays exist once.
#pragma once
#include "common.h"
#include <stdlib.h>

struct ConvParams {
  lmn_conv_args_t a;
  int nkb[3];      // K16 blocks per source
  int kb_off[3];   // first K16 block of each source
  int NKB;         // total K16 blocks
  int NCTT;        // total cout tiles
  int ncls;        // 1 or 4 parity classes
  int gpi;         // groups per image
  int ng_c[4];     // groups per class
  int Hc[4], Wc[4];
  int total_sets;
  float inv_keep_ep;      // 1/(1-p) of the epilogue dropout
  float inv_keep_src[3];
  // LDS-tiled kernel (conv_tile_kernel): output tile TH x TW (TP pixels, NG pixel groups), input window XH x XW,
  // LDS pixel stride CS floats, CKB K16-blocks per staged chunk
  int TH, TW, TP, NG, XH, XW, CS, CKB, tiles_x, tiles_y, total_tiles;
  uint32_t mTW, mXW;
  int strided;
  float* det_stats;       // deterministic mode: slot copies of the statistics destination (one slot per block / per wave), else NULL
  LmnLay lay_src[3], lay_out, lay_aux;   // address forms of the operands (common.h: NHWC or row-planar)
  int32_t rpw;                           // image width of the row-planar operands of the call (0: none)
  uint32_t rp_magic;                     // floor(2^32 / rpw)
  LmnLay lay_out2, lay_aux2;             // chained second conv (lmn_conv_chain_t): its out / aux address forms
  int32_t NCTT2;                         // ... and its cout tiles
  int32_t prio;                          // 4..6: launched on a stream of wave priority 3..1 (lmn_set_priority_stream): s_setprio for every wave;
                                         // LMN_CONV_PRIO (experiment): 1..3 = the co-resident waves of a SIMD get DISTINCT issue priorities;
                                         // >= 100: start stagger of (prio - 100) x 256 cycles per wave slot (lmn_wave_stagger)
};

// Distinct issue priorities for the waves that share a SIMD (one wave of each co-resident block): equal-priority waves share the matrix
// pipe round-robin, so co-resident blocks that entered their MFMA loops together also leave them together and then all stage at once
// (phase clocks, DESIGN 5h: the pipe idles 45 % of a block's life although it is saturated inside the loops).  With a strict order the
// highest-priority wave runs its loop at full rate and stages while the next one multiplies.
__device__ __forceinline__ void lmn_wave_prio(int mode) {
  uint32_t slot;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID, 0, 4)" : "=s"(slot));   // wave slot of this wave on its SIMD
  const uint32_t p = mode >= 4 ? (uint32_t)(7 - mode) : (mode == 2 ? 3u - (slot & 3u) : (mode == 3 ? (slot & 1u) * 3u : (slot & 3u)));   // 4..6: uniform 3..1 (lmn_set_priority_stream)
  switch (p) {
    case 0: __builtin_amdgcn_s_setprio(0); break;
    case 1: __builtin_amdgcn_s_setprio(1); break;
    case 2: __builtin_amdgcn_s_setprio(2); break;
    default: __builtin_amdgcn_s_setprio(3); break;
  }
}

// Start stagger: the persistent blocks of a launch are dispatched together, run the same program on tiles of the same size and share
// the matrix pipe fairly, so the co-resident blocks of a CU stay in lock-step for the whole kernel -- all stage, then all multiply
// (tile time = staging + N x loop) -- and nothing ever desynchronises them.  A one-off delay of slot x units x 256 cycles
// before the first tile spreads their phases; equal tile times then keep the spacing.
__device__ __forceinline__ void lmn_wave_stagger(int units) {
  uint32_t slot;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID, 0, 4)" : "=s"(slot));   // wave slot of this wave on its SIMD
  const int n = (int)(slot & 7u) * units;
  for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(4);   // 4 x 64 cycles
}

namespace {

// precision mode of a conv-family kernel instance: 0 = fp32 storage + fp32 MFMA, 1 = fp32 storage + bf16 MFMA operands,
// 2 = bf16 storage + bf16 MFMA operands (accumulators / epilogues / statistics are fp32 in every mode)
template <int PM> struct ActT { typedef float type; };
template <> struct ActT<2> { typedef lmn_bf16 type; };

// ---- bf16 operand forms (mixed-precision path: bf16 MFMA operands, fp32 accumulators / epilogues / statistics)
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_bf16(float a, float b) {  // one v_cvt_pk_bf16_f32 (round to nearest even, NaN kept)
  const f32x2_t t = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(t, bf16x2_t));
}
__device__ __forceinline__ uint2 pk4_bf16(f32x4 v) { return uint2{pk_bf16(v[0], v[1]), pk_bf16(v[2], v[3])}; }
// MFMA operand fragment of one K16 block: fp32 path = 4 floats (one per v_mfma_f32_16x16x4_f32, blocked channel order),
// bf16 path = 4 bf16 (channels 4q..4q+3 of the lane's row, ONE v_mfma_f32_16x16x16_bf16)
template <bool BF> struct Frag { typedef f32x4 type; };
template <> struct Frag<true> { typedef uint2 type; };
template <bool BF> __device__ __forceinline__ typename Frag<BF>::type ldfrag(const float* p) {
  if constexpr (BF) return *reinterpret_cast<const uint2*>(p);
  else return *reinterpret_cast<const f32x4*>(p);
}
__device__ __forceinline__ f32x4 mfma_bf16(uint2 a, uint2 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), c, 0, 0, 0);
}
// Two K16 blocks (or two 16-pixel K steps of a weight gradient) per call: TWO v_mfma_f32_16x16x16_bf16.
// gfx950 has a shape of its own for this, v_mfma_f32_16x16x32_bf16 (twice the K per instruction at the cycles of the 16x16x16 form;
// its k index of lane group q, element j is 8q + j, the fragments carry block 0: channels 4q..4q+3 | block 1: channels 16+4q..16+4q+3
// -- a different bijection k <-> channel, which is fine as long as A and B use the same one), and rounds 2-3 used it (+1.6 % at
// bf16 batch 64).  NOT USED ANY MORE (round 4): kernels issuing it corrupt the results of OTHER kernels running beside them on
// another stream.  Found with the deterministic-mode schedule check in bf16: the neighborhood-attention backward of level 1 came
// out with wrong sum_n p_n dp_n terms for ~0.4 % of the queries (errors of 0.2 on values of order 1: wrong dq / dk, correct dv)
// whenever a bf16 3x3 conv or weight gradient ran on a second stream; reproduced stand-alone (tools/gpu_na_stress2.py: 27 of 30
// runs wrong against a quiet re-run and against the fp32 oracle; the fp32 attention kernel beside a bf16 conv likewise; inputs,
// canaries around the buffers and the kernels' register allocations all intact); 0 of 80 with this function as two 16x16x16
// MFMAs, 0 with fp32 conv kernels, 0 with the bf16 instances that never reach the 32-wide form (one K16 block; the M-split and
// 1x1 instances were not seen to do it either).  -DLMN_MFMA_X2 restores the instruction for experiments.
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x4 mfma_bf16x2(uint2 a0, uint2 a1, uint2 b0, uint2 b1, f32x4 c) {
#ifdef LMN_MFMA_X2
  const uint4 a = uint4{a0.x, a0.y, a1.x, a1.y}, b = uint4{b0.x, b0.y, b1.x, b1.y};
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
#else
  return mfma_bf16(a1, b1, mfma_bf16(a0, b0, c));
#endif
}

#ifdef LMN_CT_TIMING
// phase clocks of conv_tile_kernel (debug builds): per block {barrier 1, staging, barrier 2, MFMA, epilogue, life, start, end}
__device__ unsigned long long g_ct_timing[4096 * 8];
#define LMN_TK(i) do { tk_b = __builtin_amdgcn_s_memtime(); tk_s[i] += tk_b - tk_a; tk_a = tk_b; } while (0)
#define LMN_TK_DRAIN() __builtin_amdgcn_s_waitcnt(0)
#else
#define LMN_TK(i) do { } while (0)
#define LMN_TK_DRAIN() do { } while (0)
#endif

}  // namespace

// launchers of the kernel instances (one translation unit per family).  tnct: cout tiles per block (1..3), pm: precision mode
// (0 fp32, 1 bf16 operands, 2 bf16 storage), ek: epilogue instance (see conv_tile_kernel), wlk: LDS-staged weights (3x3),
// rp: row-planar operands (1x1), ln: LayerNorm on load (LMN_SRC_LN; 1x1, ek 0), up: bilinear x2 on load (LMN_SRC_UP2; 3x3, ek 0), ncw: cout tiles per wave of the M-split kernel.
int lmn_launch_conv_tile_1x1(const ConvParams& T, dim3 grid, size_t shmem, hipStream_t st, int tnct, int pm, int ek, bool rp, bool ln);
int lmn_launch_conv_tile_3x3(const ConvParams& T, dim3 grid, size_t shmem, hipStream_t st, int tnct, int pm, int ek, bool wlk, bool up);
int lmn_launch_conv_tile_3x3g(const ConvParams& T, dim3 grid, size_t shmem, hipStream_t st, int tnct, int pm, int ek, bool wlk);   // NPG = 4 (tnct <= 2)
// LDS-DMA double-buffered 3x3 stride-1 kernel of the small-channel layers (conv_dma3.hip): one plain NHWC fp32 source of 12 / 24 channels,
// Cout <= 32, plain epilogue (bias, residual, SUM_SQ statistics); T.tiles_x / tiles_y / total_tiles describe 8 x 16-pixel tiles
size_t lmn_conv_dma3_lds(int C, int nct);
int lmn_launch_conv_dma3(const ConvParams& T, int blocks, hipStream_t st);
// conv_dmaM.hip: the M-split tile of the wide 3x3 stride-1 convs with window and weights by LDS-DMA
size_t lmn_conv_dmaM_lds(int XH, int XW);
bool lmn_conv_dmaM_fits(int XH, int XW);
int lmn_launch_conv_dmaM(const ConvParams& T, dim3 grid, hipStream_t st, int ek);
// LDS-DMA streaming kernel of ReparamConv's HBM-bound 1x1 convs at levels 0-1 (conv_dma1.hip): instance lookup (tile pixels, 0: none),
// LDS bytes, launch.  ks*: channel quads of the sources, aq: of the aux image, mode: 0 plain / 2 SUM_SQ / 5 SE_BWD, gs: GELU x gate on source 0
int lmn_conv_dma1_tp(int ks0, int ks1, int ks2, int aq, int nct, int mode, int gs, int nct2, int mode2);
size_t lmn_conv_dma1_lds(int ks0, int ks1, int ks2, int aq, int nct, int tp, int nct2);
int lmn_launch_conv_dma1(const ConvParams& T, int ks0, int ks1, int ks2, int aq, int nct, int mode, int gs, int nct2, int mode2, int blocks, hipStream_t st);
int lmn_launch_conv_tile_s2t(const ConvParams& T, dim3 grid, size_t shmem, hipStream_t st, int tnct, int pm, int ek);
int lmn_launch_conv_tileM(const ConvParams& T, dim3 grid, size_t shmem, hipStream_t st, int taps, int ncw, int pm, int ek, bool rp, bool ln, bool up);
