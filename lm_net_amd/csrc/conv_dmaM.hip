// conv_dmaM_kernel (round 6): the WIDE 3x3 stride-1 forward / data-gradient convs of the small feature maps (one plain NHWC fp32 source
// of 48 ... 372 channels, Cout > 80: the skip fusers' convs and the bottleneck / decoder convs at 88^2 / 44^2 / 22^2 -- SURVEY rows A4 /
// A9 / A10, core/modules.py:83-143, core/LM_Net.py:14-39) as an M-split tile (conv_tileM_kernel's) whose operands -- the window AND the
// weights -- arrive by LDS-DMA: the K loop of a tile runs over STAGES of one K16 block, and the window slice + the 9 x 4 weight
// fragments of stage s+1 (`buffer_load_dwordx4 ... lds` into the second LDS buffer) are in flight while the matrix cores work on stage s.
//
// Why: on these maps a block is alone (or one of two) on its CU.  conv_tileM_kernel fills its window with register-staged loads
// between two barriers per K chunk and fetches every weight fragment from L2 ONE step (28 MFMAs, ~0.45 us) ahead of its use: phase
// clocks put its MFMA duty at 54-60 % (372 -> 372 at 22^2: 146 us alone for 91 us of matrix-pipe time).  A first version of this kernel
// that moved only the window to LDS-DMA gained 4 % -- the weight loads are the stall (and, being vector-memory loads issued after the
// pieces, every wait on them also waited for the pieces in flight).  With the weights in the stage image the K loop issues no
// vector-memory instruction at all: a stage is 13 + 36 DMA pieces per block, one s_waitcnt + ONE barrier, 252 MFMAs per wave.
//
// Tile geometry, cout split (wave w owns cout tile ct0 + w, all pixel groups), packed weights, epilogue parameters, statistics tail:
// conv_tileM_kernel's with one cout tile per wave (the host computes one ConvParams for both).  Differences:
//   * LDS image of a stage: window slice [window pixel][5 chunks of 16 B] -- the 16 channels of the K16 block in NATURAL order + one
//     pad chunk (odd pixel stride: conflict-free ds_read_b128); pad chunks, channels past C and window pixels outside the image are
//     written as zeros by the DMA itself (out-of-range buffer offset), so the conv's zero padding costs no instruction -- then the
//     weights [tap][cout tile][fragment], each 1 KB fragment copied as it lies by ONE piece of the wave that owns the tile;
//   * K mapping: slice j of the 16x16x4 MFMA takes channel 16 kb + 4 q + j from lane group q (one b128 read = a lane's four slices);
//     the matching weight element sits at [(j * 16 + m) * 4 + q] of the packed fragment (conv_pack / conv_pack_t store channel
//     16 kb + 4 j' + q' at [(q' * 16 + m) * 4 + j']): four conflict-free ds_read_b32 per tap.
// Synchronisation per stage: `s_waitcnt vmcnt(0)` (the stage's pieces were issued a whole stage ago; the only younger operations are,
// after a tile's last stage, its stores) + ONE raw s_barrier.  ~100 KB of LDS: one block per CU, which is what these grids give anyway.
#include "conv_tile.h"

namespace {

constexpr unsigned DM_OOB = 0x80000000u;
typedef int dm_i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ dm_i32x4 dm_rsrc(const void* p, unsigned bytes) {
  const unsigned long long a = (unsigned long long)p;
  dm_i32x4 r;
  r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
  r.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
  r.z = __builtin_amdgcn_readfirstlane((int)bytes);
  r.w = 0x00020000;
  return r;
}
__device__ __forceinline__ void dm_piece(unsigned ldsaddr, unsigned voff, dm_i32x4 rsrc) {
  unsigned keep;   // (M0 is compiler-reserved and not preserved around a statement: saved and restored inside it)
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(ldsaddr), "v"(voff), "s"(rsrc) : "memory");
}

constexpr int DM_PS = 5;        // 16-byte chunks per window pixel and stage: 4 data (one K16 block) + 1 pad
constexpr int DM_NKMAX = 5;     // window pieces per thread and stage (4 waves x 5 x 64 chunks = 1280 chunks = 256 window pixels)
constexpr int DM_WFL = 9 * 4 * 256;   // floats of a stage's weights: [tap][cout tile of the block][fragment]

// EPI: 0 plain, 2 SUM_SQ statistics.  One cout tile per wave (block: 64 output channels).
template <int EPI>
__global__ __launch_bounds__(256, 1) void conv_dmaM_kernel(const ConvParams P) {
  constexpr int NCT = 4, NGM = 8;
  const lmn_conv_args_t& A = P.a;
  const lmn_src_t& S = A.src[0];
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int NCH = P.XH * P.XW * DM_PS;                 // chunks of one window slice
  const int BUFW = (NCH + 63) & ~63;                   // its buffer stride (whole pieces)
  const int BUF = BUFW * 4 + DM_WFL;                   // floats of one stage buffer: window slice, then weights
  float* const s_stats = smem + 2 * BUF;               // [2][NCT*16]
  float* const s_par = s_stats + 2 * NCT * 16;         // [9][NCT*16]
  if (P.prio >= 4) lmn_setprio_level(7 - P.prio);      // (uniform: lmn_set_priority_stream)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, n = lane & 15;
  const int ct0 = blockIdx.y * NCT;
  const int H = A.Hin, W = A.Win;                      // (stride 1: input and output maps have the same size)
  const unsigned xbytes = (unsigned)(((int64_t)A.B * H * W - 1) * S.cstride + S.C) * 4u;
  const dm_i32x4 rx = dm_rsrc(S.ptr, xbytes);
  const dm_i32x4 rw = dm_rsrc(A.wpack, (unsigned)((int64_t)9 * P.NKB * P.NCTT * 1024));
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem;

  // per-thread chunk descriptors of the window slice (tile-independent): window row << 16 | column << 8 | chunk; -1: pad chunk / past the image
  int drc[DM_NKMAX];
#pragma unroll
  for (int k = 0; k < DM_NKMAX; ++k) {
    const int i = (k * 4 + wv) * 64 + lane;
    const int wp = i / DM_PS, f = i - wp * DM_PS;
    const int r = (int)__umulhi((uint32_t)wp, P.mXW), c = wp - r * P.XW;
    drc[k] = (i < NCH && f < 4) ? (r << 16 | c << 8 | f) : -1;
  }
  int pixoff[DM_NKMAX];   // float offset of the chunk in the source at K16 block 0 (tile-dependent), -1: outside
  auto tile_addr = [&](int tile) __attribute__((always_inline)) {
    const int b = tile / (P.tiles_x * P.tiles_y), tt = tile - b * P.tiles_x * P.tiles_y;
    const int wy0 = (tt / P.tiles_x) * P.TH - 1, wx0 = (tt % P.tiles_x) * P.TW - 1;
#pragma unroll
    for (int k = 0; k < DM_NKMAX; ++k) {
      const int d = drc[k];
      const int iy = wy0 + (d >> 16), ix = wx0 + ((d >> 8) & 255);
      const bool ok = d >= 0 && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
      pixoff[k] = ok ? ((b * H + iy) * W + ix) * S.cstride + (d & 255) * 4 : -1;
    }
  };
  // this wave's weight pieces of a stage: the nine fragments (tap, kb, its own cout tile); a fragment is 64 lanes x 16 B = ONE piece,
  // copied as it lies (cout tiles past the end re-read the last real one)
  const unsigned wlane_b = (unsigned)(min(ct0 + wv, P.NCTT - 1) * 1024 + lane * 16);
  auto issue = [&](int kb, int buf) __attribute__((always_inline)) {
    const unsigned base = lds0 + (unsigned)buf * (unsigned)(BUF * 4);
    const int ch0 = kb * 16;
#pragma unroll
    for (int k = 0; k < DM_NKMAX; ++k) {
      if ((k * 4 + wv) * 64 >= NCH) continue;            // (wave-uniform: this wave's piece lies past the slice)
      const int po = pixoff[k];
      const bool ok = po >= 0 && ch0 + (drc[k] & 255) * 4 < S.C;
      dm_piece(__builtin_amdgcn_readfirstlane(base + (unsigned)(k * 4 + wv) * 1024u), ok ? (unsigned)(po + ch0) * 4u : DM_OOB, rx);
    }
    const unsigned wbase = base + (unsigned)(BUFW * 16) + (unsigned)wv * 1024u;
    const unsigned wk = (unsigned)((P.kb_off[0] + kb) * P.NCTT) * 1024u + wlane_b;
    const unsigned wtap = (unsigned)(P.NKB * P.NCTT) * 1024u;
#pragma unroll
    for (int t = 0; t < 9; ++t) dm_piece(__builtin_amdgcn_readfirstlane(wbase + (unsigned)t * 4096u), wk + (unsigned)t * wtap, rw);
  };

  const int nkb = P.nkb[0];                              // stages per tile: one K16 block each
  const int t0 = blockIdx.x, tstep = gridDim.x;
  if (t0 < P.total_tiles) { tile_addr(t0); issue(0, 0); }

  for (int i = tid; i < 2 * NCT * 16; i += 256) s_stats[i] = 0.f;
  conv_stage_params<NCT>(A, s_par, ct0, tid, blockIdx.x == 0);
  float st0[4], st1[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) st0[r] = st1[r] = 0.f;
  const int flip = A.transposed ? 2 : 0;                // tap t of the packed weights meets the window pixel at (ty, tx), flipped for the data gradient
  const int wrd = wv * 256 + n * 4 + q;                 // slice j of a fragment, lane (q, m = n): element j * 64 + m * 4 + q (header)
  __syncthreads();   // s_par / s_stats written

  int buf = 0;
  for (int tile = t0; tile < P.total_tiles; tile += tstep) {
    const int b = tile / (P.tiles_x * P.tiles_y), tt = tile - b * P.tiles_x * P.tiles_y;
    const int oy0 = (tt / P.tiles_x) * P.TH, ox0 = (tt % P.tiles_x) * P.TW;
    int pbase[NGM], opix[NGM];
    bool pvalid[NGM];
#pragma unroll
    for (int g = 0; g < NGM; ++g) {
      const int i = g * 16 + n;
      const bool in_t = i < P.TP;
      const int is = in_t ? i : 0;
      const int r = (int)__umulhi((uint32_t)is, P.mTW), c = is - r * P.TW;
      const int oy = oy0 + r, ox = ox0 + c;
      pvalid[g] = in_t && oy < A.Hout && ox < A.Wout;
      opix[g] = pvalid[g] ? (b * A.Hout + oy) * A.Wout + ox : 0;
      pbase[g] = ((r * P.XW + c) * DM_PS + q) * 4;
    }
    f32x4 acc[NGM];
    {
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(s_par + wv * 16 + q * 4);
#pragma unroll
      for (int g = 0; g < NGM; ++g) acc[g] = b4;
    }
    const bool has_next = tile + tstep < P.total_tiles;   // (block-uniform)

    for (int kb = 0; kb < nkb; ++kb) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this stage's pieces have landed (issued one stage ago)
      __builtin_amdgcn_s_barrier();                       // ... for every wave; every wave has left the stage that read the other buffer
      if (kb + 1 < nkb) issue(kb + 1, buf ^ 1);
      else if (has_next) { tile_addr(tile + tstep); issue(0, buf ^ 1); }
      const float* XS = smem + buf * BUF;
      const float* WS = XS + BUFW * 4 + wrd;
      f32x4 xa[4];
      float wcur[4];
      {
        const int toff0 = ((flip * P.XW) + flip) * (DM_PS * 4);
#pragma unroll
        for (int u = 0; u < 4; ++u) xa[u] = *reinterpret_cast<const f32x4*>(&XS[pbase[u] + toff0]);
#pragma unroll
        for (int j = 0; j < 4; ++j) wcur[j] = WS[j * 64];
      }
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int tapn = tap < 8 ? tap + 1 : tap;
        float wnext[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) wnext[j] = WS[tapn * 1024 + j * 64];
        const int ty = tap / 3, tx = tap - ty * 3;
        const int fy = flip ? 2 - ty : ty, fx = flip ? 2 - tx : tx;
        const int toff = (fy * P.XW + fx) * (DM_PS * 4);
        const int tyn = tapn / 3, txn = tapn - tyn * 3;
        const int fyn = flip ? 2 - tyn : tyn, fxn = flip ? 2 - txn : txn;
        const int toffn = (fyn * P.XW + fxn) * (DM_PS * 4);
        // two half-sets of four pixel groups in flight (conv_tileM_kernel's scheme): the second half of this tap is requested before
        // the MFMAs of the first, the first half of the next tap before the MFMAs of the second
#pragma unroll
        for (int h = 0; h < NGM; h += 4) {
          f32x4 xg[4];
          if (h == 0) {
#pragma unroll
            for (int u = 0; u < 4; ++u) { xg[u] = xa[u]; xa[u] = *reinterpret_cast<const f32x4*>(&XS[pbase[4 + u] + toff]); }
          } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) { xg[u] = xa[u]; xa[u] = *reinterpret_cast<const f32x4*>(&XS[pbase[u] + toffn]); }
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int g = h + u;
            if (g < P.NG) {   // (wave-uniform)
#pragma unroll
              for (int j = 0; j < 4; ++j) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(wcur[j], xg[u][j], acc[g], 0, 0, 0);
            }
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) wcur[j] = wnext[j];
      }
      buf ^= 1;
    }

    // ---- epilogue (lane holds channels co..co+3 of its pixel): conv_tileM_kernel's plain / SUM_SQ paths
    const int ctc = ct0 + wv;
    const int co = ctc * 16 + q * 4;
    const bool cok = ctc < P.NCTT && co < A.Cout;
    const int cos = cok ? co : 0;
    const f32x4 sh = *reinterpret_cast<const f32x4*>(s_par + 6 * NCT * 16 + wv * 16 + q * 4);   // SUM_SQ: sums about p4 (see conv_tileM_kernel)
#pragma unroll
    for (int g = 0; g < NGM; ++g) {
      const uint32_t opx = (uint32_t)opix[g];
      const bool live = pvalid[g] && cok && g < P.NG;
      f32x4 o = acc[g];
      if (EPI == 2 && live) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float d = o[r] - sh[r]; st0[r] += d; st1[r] += d * d; }
      }
      if (A.residual) o += ld4((const float*)A.residual + opx * A.res_cstride + cos);
      if (A.out && live) st4((float*)A.out + opx * A.out_cstride + cos, o);
    }
  }

  // ---- statistics: wave shuffle over the 16 pixels -> LDS -> one global atomic per channel per block (conv_tileM_kernel's tail)
  if (EPI == 2) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float a = st0[r], bb = st1[r];
#pragma unroll
      for (int m = 1; m <= 8; m <<= 1) {
        a += __shfl_xor(a, m, 64);
        bb += __shfl_xor(bb, m, 64);
      }
      if (n == 0) {  // each (tile, channel) is owned by exactly one wave: plain stores
        s_stats[wv * 16 + q * 4 + r] = a;
        s_stats[NCT * 16 + wv * 16 + q * 4 + r] = bb;
      }
    }
    __syncthreads();
    for (int i = tid; i < 2 * NCT * 16; i += 256) {
      const int which = i / (NCT * 16), cc = i - which * NCT * 16;
      const int co2 = ct0 * 16 + cc;
      if (co2 < A.Cout) {
        if (P.det_stats) P.det_stats[(int64_t)blockIdx.x * 2 * A.Cout + (int64_t)which * A.Cout + co2] = s_stats[i];
        else atomicAdd(A.stats + (A.stats_rep > 1 ? (int64_t)(blockIdx.x % A.stats_rep) * 2 * A.Cout : 0) + (int64_t)which * A.Cout + co2, s_stats[i]);
      }
    }
  }
}

}  // namespace

size_t lmn_conv_dmaM_lds(int XH, int XW) {
  const int nch = XH * XW * DM_PS, bufw = (nch + 63) & ~63;
  return ((size_t)2 * (bufw * 4 + DM_WFL) + (size_t)(2 + 9) * 4 * 16) * sizeof(float);
}
bool lmn_conv_dmaM_fits(int XH, int XW) { return XH * XW * DM_PS <= DM_NKMAX * 256 && XH < 256 && XW < 256; }

int lmn_launch_conv_dmaM(const ConvParams& T, dim3 grid, hipStream_t st, int ek) {
  const size_t sh = lmn_conv_dmaM_lds(T.XH, T.XW);
#define LMN_DM(EE)                                                                                                                        \
  do {                                                                                                                                    \
    if (sh > 64 * 1024) (void)hipFuncSetAttribute((const void*)conv_dmaM_kernel<EE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); \
    LMN_LAUNCH((conv_dmaM_kernel<EE>), grid, dim3(256), sh, st, T);                                                                       \
  } while (0)
  if (ek == 0) LMN_DM(0);
  else if (ek == 2) LMN_DM(2);
  else return -1;
#undef LMN_DM
  return 0;
}
