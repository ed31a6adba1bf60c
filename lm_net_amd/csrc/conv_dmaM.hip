// conv_dmaM_kernel (round 6): the WIDE 3x3 stride-1 forward / data-gradient convs of the small feature maps (one plain NHWC fp32 source
// of 48 ... 372 channels, Cout > 80: the skip fusers' convs and the bottleneck / decoder convs at 88^2 / 44^2 / 22^2 -- SURVEY rows A4 /
// A9 / A10, core/modules.py:83-143, core/LM_Net.py:14-39) as an M-split tile (conv_tileM_kernel's) whose operands -- the window AND the
// weights -- arrive by LDS-DMA (`buffer_load_dwordx4 ... lds`); the K loop issues no other vector-memory instruction.
//
// Why: on these maps a block is alone or one of two or three on its CU.  conv_tileM_kernel fills its window with register-staged loads
// between two barriers per K chunk and fetches every weight fragment from L2 ONE step (28 MFMAs, ~0.45 us) ahead of its use: phase
// clocks put its MFMA duty at 54-60 % (372 -> 372 at 22^2: 146 us alone for 91 us of matrix-pipe time).  Moving only the window to
// LDS-DMA gained 4 % -- the weight loads are the stall (and, being vector-memory loads issued after the pieces, every wait on them also
// waited for the pieces in flight: one in-order counter).
//
// Tile geometry, cout split (wave w owns cout tile ct0 + w, all pixel groups), packed weights, epilogue parameters, statistics tail:
// conv_tileM_kernel's with one cout tile per wave (the host computes one ConvParams for both).  The K loop runs over STAGES of one K16
// block x nine taps:
//   * window slice of a stage: [window pixel][5 chunks of 16 B] -- the 16 channels of the K16 block in NATURAL order + one pad chunk
//     (odd pixel stride: conflict-free ds_read_b128); pad chunks, channels past C and window pixels outside the image are written as
//     zeros by the DMA itself (out-of-range buffer offset), so the conv's zero padding costs no instruction.  Double-buffered for the
//     block: ONE barrier per stage;
//   * weights: a wave reads only the fragments of ITS cout tile and issues their pieces itself (a packed 1 KB fragment is exactly one
//     piece, copied as it lies), so they need no block barrier: they travel through a per-wave RING of DR_R 1 KB slots, DR_D fragments
//     ahead, ordered by counted `s_waitcnt vmcnt(N)` alone;
//   * K mapping: slice j of the 16x16x4 MFMA takes channel 16 kb + 4 q + j from lane group q (one b128 read = a lane's four slices);
//     the matching weight element sits at [(j * 16 + m) * 4 + q] of the packed fragment (conv_pack / conv_pack_t store channel
//     16 kb + 4 j' + q' at [(q' * 16 + m) * 4 + j']): four conflict-free ds_read_b32 per tap.
// ~41-57 KB of LDS, <= 168 VGPRs: two or three blocks per CU.  Measured alone on cold operands (MI355X, batch 8): 372 -> 372 at 22^2
// 147.7 -> 126-130 us (65 -> 75 TF/s), 96 -> 96 at 44^2 51.5 -> 46.5 us; a first form with the whole stage (window + 36 fragments)
// double-buffered for the block (100 KB, one block per CU) measured the same on the 22^2 layer and 58 us on the 44^2 one.
//
// Vector-memory operations of a wave, in issue order, per tap f of a tile (F = 9 nkb taps):
//   stage start (f % 9 == 0, not the last stage): NKW window pieces of stage + 1 (behind the barrier), then
//   every tap: the piece of fragment f + DR_D (while it exists).
// Fragment f + 1 is read from the ring at tap f (one tap ahead of its MFMAs): its piece was issued at tap f + 1 - DR_D, and younger than it
// are the pieces of the DR_D - 2 taps in between plus this stage's window burst when f % 9 <= DR_D - 2; in the last stage the pieces
// run out: min(DR_D - 2, 7 - t).  The window slice of stage s was issued at the first tap of stage s - 1, nine fragment pieces ago.  All
// counts are compile-time (the nine taps are unrolled, `last` selects the variant).  A tile starts drained (vmcnt(0) after its prologue;
// the previous tile's stores retire there too).  The pieces are inline asm (hipcc's wait-count bookkeeping does not see them: hidden
// operations can only make its own waits stricter, never too weak -- the counter retires in issue order).
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

#ifndef LMN_DR_BPC
#define LMN_DR_BPC 3
#endif
#ifndef LMN_DR_D
#define LMN_DR_D 3
#endif
constexpr int DR_D = LMN_DR_D;    // fragments in flight ahead of the tap that READS one (= DR_D - 1 ahead of the tap that multiplies it); <= 5
constexpr int DR_R = DR_D + 1;    // ring slots: the slot refilled at tap f held fragment f - 1, consumed a whole tap ago
static_assert(DR_D >= 2 && DR_D <= 5, "wait counts are written for two to five fragments in flight");

template <int N> __device__ __forceinline__ void dr_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int EPI, int NKW>
__global__ __launch_bounds__(256, LMN_DR_BPC) void conv_dmaM_kernel(const ConvParams P) {
  constexpr int NCT = 4, NGM = 8;
  constexpr int BUFW = NKW * 256;                      // chunks of one window buffer: NKW pieces per wave
  const lmn_conv_args_t& A = P.a;
  const lmn_src_t& S = A.src[0];
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int NCH = P.XH * P.XW * DM_PS;                 // chunks of one window slice (<= BUFW)
  float* const s_ring = smem + 2 * BUFW * 4;           // [wave][DR_R][256]
  float* const s_stats = s_ring + 4 * DR_R * 256;      // [2][NCT*16]
  float* const s_par = s_stats + 2 * NCT * 16;         // [9][NCT*16]
  if (P.prio >= 4) lmn_setprio_level(7 - P.prio);      // (uniform: lmn_set_priority_stream)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, n = lane & 15;
  const int ct0 = blockIdx.y * NCT;
  const int H = A.Hin, W = A.Win;
  const unsigned xbytes = (unsigned)(((int64_t)A.B * H * W - 1) * S.cstride + S.C) * 4u;
  const dm_i32x4 rx = dm_rsrc(S.ptr, xbytes);
  const dm_i32x4 rw = dm_rsrc(A.wpack, (unsigned)((int64_t)9 * P.NKB * P.NCTT * 1024));
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem;
  const unsigned ring0 = lds0 + (unsigned)(2 * BUFW * 16) + (unsigned)wv * (DR_R * 1024u);

  int drc[NKW];
#pragma unroll
  for (int k = 0; k < NKW; ++k) {
    const int i = (k * 4 + wv) * 64 + lane;
    const int wp = i / DM_PS, f = i - wp * DM_PS;
    const int r = (int)__umulhi((uint32_t)wp, P.mXW), c = wp - r * P.XW;
    drc[k] = (i < NCH && f < 4) ? (r << 16 | c << 8 | f) : -1;
  }
  int pixoff[NKW];
  auto tile_addr = [&](int tile) __attribute__((always_inline)) {
    const int b = tile / (P.tiles_x * P.tiles_y), tt = tile - b * P.tiles_x * P.tiles_y;
    const int wy0 = (tt / P.tiles_x) * P.TH - 1, wx0 = (tt % P.tiles_x) * P.TW - 1;
#pragma unroll
    for (int k = 0; k < NKW; ++k) {
      const int d = drc[k];
      const int iy = wy0 + (d >> 16), ix = wx0 + ((d >> 8) & 255);
      const bool ok = d >= 0 && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
      pixoff[k] = ok ? ((b * H + iy) * W + ix) * S.cstride + (d & 255) * 4 : -1;
    }
  };
  // EXACTLY NKW pieces per wave (the wait counts rely on it): a piece past the slice writes zeros into the buffer's tail
  auto issue_win = [&](int kb, int buf) __attribute__((always_inline)) {
    const unsigned base = lds0 + (unsigned)buf * (unsigned)(BUFW * 16);
    const int ch0 = kb * 16;
#pragma unroll
    for (int k = 0; k < NKW; ++k) {
      const int po = pixoff[k];
      const bool ok = po >= 0 && ch0 + (drc[k] & 255) * 4 < S.C;
      dm_piece(__builtin_amdgcn_readfirstlane(base + (unsigned)(k * 4 + wv) * 1024u), ok ? (unsigned)(po + ch0) * 4u : DM_OOB, rx);
    }
  };
  const unsigned wlane_b = (unsigned)(min(ct0 + wv, P.NCTT - 1) * 1024 + lane * 16);
  const unsigned wtap = (unsigned)(P.NKB * P.NCTT) * 1024u, wkb = (unsigned)P.NCTT * 1024u;
  const unsigned wbase0 = (unsigned)(P.kb_off[0] * P.NCTT) * 1024u + wlane_b;
  // ONE piece: fragment g = kb * 9 + tap of this wave's cout tile into ring slot g % DR_R
  auto issue_frag = [&](int g) __attribute__((always_inline)) {
    const int kb = g / 9, tap = g - kb * 9;
    dm_piece(__builtin_amdgcn_readfirstlane(ring0 + (unsigned)(g % DR_R) * 1024u), wbase0 + (unsigned)kb * wkb + (unsigned)tap * wtap, rw);
  };

  const int nkb = P.nkb[0], F = 9 * nkb;
  for (int i = tid; i < 2 * NCT * 16; i += 256) s_stats[i] = 0.f;
  conv_stage_params<NCT>(A, s_par, ct0, tid, blockIdx.x == 0);
  float st0[4], st1[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) st0[r] = st1[r] = 0.f;
  const int flip = A.transposed ? 2 : 0;
  const float* const WR = s_ring + wv * (DR_R * 256) + n * 4 + q;   // slice j of a fragment: + j * 64
  __syncthreads();

  for (int tile = blockIdx.x; tile < P.total_tiles; tile += gridDim.x) {
    const int b = tile / (P.tiles_x * P.tiles_y), tt = tile - b * P.tiles_x * P.tiles_y;
    const int oy0 = (tt / P.tiles_x) * P.TH, ox0 = (tt % P.tiles_x) * P.TW;
    // prologue: the first window slice and the first DR_D fragments, then a full drain (the previous tile's stores included); every
    // wave has left the previous tile's last stage before a buffer is refilled
    __builtin_amdgcn_s_barrier();
    tile_addr(tile);
    issue_win(0, 0);
#pragma unroll
    for (int g = 0; g < DR_D; ++g)
      if (g < F) issue_frag(g);
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
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    float wcur[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) wcur[j] = WR[j * 64];   // fragment 0 (slot 0)

    int buf = 0;
#pragma nounroll
    for (int kb = 0; kb < nkb; ++kb) {
      const bool last = kb + 1 == nkb;                  // (block-uniform)
      const int f0 = kb * 9;
      if (kb > 0) {
        dr_wait<9>();                                   // this stage's window slice: issued nine fragment pieces ago
        __builtin_amdgcn_s_barrier();                   // ... by every wave; every wave has left the stage that read the other buffer
      }
      if (!last) issue_win(kb + 1, buf ^ 1);
      const float* XS = smem + buf * (BUFW * 4);
      f32x4 xa[4];
      {
        const int toff0 = ((flip * P.XW) + flip) * (DM_PS * 4);
#pragma unroll
        for (int u = 0; u < 4; ++u) xa[u] = *reinterpret_cast<const f32x4*>(&XS[pbase[u] + toff0]);
      }
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int f = f0 + tap;
        // ---- fragment f + 1 from the ring (one tap ahead of its MFMAs), then the piece of fragment f + 1 + DR_D... see the header
        float wnext[4];
        if (!(last && tap == 8)) {
          constexpr int Y = DR_D - 2;                    // fragment pieces younger than the one waited for (while they exist)
          if (!last) { if (tap <= Y) dr_wait<Y + NKW>(); else dr_wait<Y>(); }
          else {
            const int y = 7 - tap < Y ? 7 - tap : Y;
            if (y >= 3) dr_wait<3>(); else if (y == 2) dr_wait<2>(); else if (y == 1) dr_wait<1>(); else dr_wait<0>();
          }
          const float* wr = WR + ((f + 1) % DR_R) * 256;
#pragma unroll
          for (int j = 0; j < 4; ++j) wnext[j] = wr[j * 64];
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) wnext[j] = 0.f;
        }
        if (f + DR_D < F) issue_frag(f + DR_D);
        const int tapn = tap < 8 ? tap + 1 : tap;
        const int ty = tap / 3, tx = tap - ty * 3;
        const int fy = flip ? 2 - ty : ty, fx = flip ? 2 - tx : tx;
        const int toff = (fy * P.XW + fx) * (DM_PS * 4);
        const int tyn = tapn / 3, txn = tapn - tyn * 3;
        const int fyn = flip ? 2 - tyn : tyn, fxn = flip ? 2 - txn : txn;
        const int toffn = (fyn * P.XW + fxn) * (DM_PS * 4);
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

    // ---- epilogue: conv_dmaM_kernel's
    const int ctc = ct0 + wv;
    const int co = ctc * 16 + q * 4;
    const bool cok = ctc < P.NCTT && co < A.Cout;
    const int cos = cok ? co : 0;
    const f32x4 sh = *reinterpret_cast<const f32x4*>(s_par + 6 * NCT * 16 + wv * 16 + q * 4);
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

  if (EPI == 2) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float a = st0[r], bb = st1[r];
#pragma unroll
      for (int m = 1; m <= 8; m <<= 1) {
        a += __shfl_xor(a, m, 64);
        bb += __shfl_xor(bb, m, 64);
      }
      if (n == 0) {
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

static int dmaM_nkw(int XH, int XW) { return (XH * XW * DM_PS + 255) / 256; }
bool lmn_conv_dmaM_fits(int XH, int XW) { const int k = dmaM_nkw(XH, XW); return k >= 1 && k <= 5 && XH < 256 && XW < 256; }
size_t lmn_conv_dmaM_lds(int XH, int XW) {
  int k = dmaM_nkw(XH, XW);
  k = k < 3 ? 3 : k;
  return ((size_t)2 * k * 256 * 4 + (size_t)4 * DR_R * 256 + (size_t)(2 + 9) * 4 * 16) * sizeof(float);
}
int lmn_launch_conv_dmaM(const ConvParams& T, dim3 grid, hipStream_t st, int ek) {
  int k = dmaM_nkw(T.XH, T.XW);
  k = k < 3 ? 3 : k;
  const size_t sh = lmn_conv_dmaM_lds(T.XH, T.XW);
#define LMN_DM(EE, KK)                                                                                                                    \
  do {                                                                                                                                    \
    if (sh > 64 * 1024) (void)hipFuncSetAttribute((const void*)conv_dmaM_kernel<EE, KK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); \
    LMN_LAUNCH((conv_dmaM_kernel<EE, KK>), grid, dim3(256), sh, st, T);                                                                   \
  } while (0)
#define LMN_DM2(KK) do { if (ek == 0) LMN_DM(0, KK); else LMN_DM(2, KK); } while (0)
  if (ek != 0 && ek != 2) return -1;
  if (k == 3) LMN_DM2(3); else if (k == 4) LMN_DM2(4); else if (k == 5) LMN_DM2(5); else return -1;
#undef LMN_DM2
#undef LMN_DM
  return 0;
}
