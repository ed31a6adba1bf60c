// conv_dma3_kernel (round 6): 3x3 stride-1 forward / data-gradient convs of the small-channel layers (one NHWC fp32 source of 12 or 24
// channels, Cout <= 32, plain epilogue: bias, residual, SUM_SQ statistics) with the input window of tile t+1 filled by LDS-DMA
// (`buffer_load_dwordx4 ... lds`) into a SECOND LDS buffer while the MFMA loop of tile t runs.  Rows A9 / A8 / A10 of SURVEY 8a: the
// `convl` / `convm` / `fuse_conv` of the skip fusers (core/modules.py:83-143), the `nat` patch embedding (core/modules.py:22-39) and the
// data gradients of those and of the decoder's up-convs (core/LM_Net.py:58-74) at levels 0-1.
//
// Why (tools/micro/conv3_dma.hip is the stand-alone prototype with its phase clocks): conv_tile_kernel stages a
// window with register loads + LDS commits between two barriers, so a block's tile is a serial chain (stage 6.9 K cycles -> MFMA loop ->
// epilogue, 31.6 K per tile at five co-resident blocks for 3.5 K of matrix-pipe time) and the co-resident blocks run it in lock-step.
// Here the staging costs the wave NO registers and no LDS-write instructions, the tile has ONE barrier, and the DMA pieces of the next
// tile are issued behind the MFMAs of the first taps of the current one (a piece costs ~55 cycles of the CU's load path: issued in a
// burst they stall the wave 1.1-2.7 K cycles per tile, interleaved they ride along).  Measured alone on cold operands (MI355X, batch 8):
// 24 -> 12 at 352^2 101.5 -> 76.4 us, 12 -> 12 55.2 -> 39.5 us, 24 -> 24 at 176^2 47.0 -> 37.9 us; 48-channel layers gain nothing (the
// old kernel's staging is already amortised over more cout tiles there) and keep conv_tile_kernel.
//
// LDS image of a window: [window pixel][PS chunks of 16 B] in NATURAL channel order, PS = chunks per pixel rounded up to an odd count
// (16 pixels 4*odd dwords apart fall on 16 distinct bank quads: conflict-free b64 / b128 operand reads).  An LDS-DMA writes
// M0 + lane * 16: the image is lane-linear per piece, every lane picks its SOURCE address -- lanes of a pad chunk and of window pixels
// outside the image pass an out-of-range offset, the buffer bounds check returns 0 and the DMA writes the zero padding itself.
// K mapping: slice s of the 16x16x4 MFMA takes channel q * KS + s from lane group q (KS = C / 4 slices), so a lane's operands of all
// slices are contiguous in the image; the block's weights are gathered once from the conv family's packed fragments (conv_pack /
// conv_pack_t: lane (q', m) element j = channel 16 kb + 4 j + q') into that order -- no second packing form on the host side.
// Synchronisation: per tile `s_waitcnt vmcnt(stores of the previous tile)` (everything older -- this tile's pieces -- has landed; never
// vmcnt(0) inside the loop) + ONE raw s_barrier (all waves' pieces are in, all waves have left the MFMA loop that read the other
// buffer).  The pieces are inline asm (hipcc's wait-count bookkeeping does not see them: hidden operations can only make its own waits
// stricter, never too weak -- the counter retires in issue order).
#include "conv_tile.h"

namespace {

constexpr unsigned DMA_OOB = 0x80000000u;

typedef int dma3_i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ dma3_i32x4 dma3_rsrc(const void* p, unsigned bytes) {   // the descriptor as four SGPR words for the asm statement
  const unsigned long long a = (unsigned long long)p;
  dma3_i32x4 r;
  r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
  r.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));   // stride 0
  r.z = __builtin_amdgcn_readfirstlane((int)bytes);
  r.w = 0x00020000;
  return r;
}
__device__ __forceinline__ void dma3_piece(unsigned ldsaddr, unsigned voff, dma3_i32x4 rsrc) {
  unsigned keep;   // (M0 is compiler-reserved and not preserved around a statement: saved and restored inside it)
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(ldsaddr), "v"(voff), "s"(rsrc) : "memory");
}

template <int KS> struct Dma3Ops { float v[KS]; };
template <int KS> __device__ __forceinline__ Dma3Ops<KS> dma3_ops(const float* p) {   // KS contiguous floats (8-byte aligned for even KS)
  Dma3Ops<KS> r;
  if constexpr (KS % 2 == 0) {
#pragma unroll
    for (int i = 0; i < KS / 2; ++i) { const f32x2_t t = *reinterpret_cast<const f32x2_t*>(p + 2 * i); r.v[2 * i] = t[0]; r.v[2 * i + 1] = t[1]; }
  } else {
#pragma unroll
    for (int i = 0; i < KS; ++i) r.v[i] = p[i];
  }
  return r;
}

// Tile: 8 x 16 output pixels, 4 waves, wave wv owns tile rows wv and wv + 4 (two pixel groups of 16).  KS = C / 4 (3 or 6).
template <int KS, int NCT, int BPC>
__global__ __launch_bounds__(256, BPC) void conv_dma3_kernel(const ConvParams P) {
  constexpr int TH = 8, TW = 16, XW = TW + 2, XH = TH + 2, NPG = 2;
  constexpr int CQ = KS, PS = (CQ & 1) ? CQ : CQ + 1;
  constexpr int NCH = XH * XW * PS;                 // 16-byte chunks of one window image
  constexpr int NK = (NCH + 255) / 256;             // LDS-DMA pieces per thread and tile
  constexpr int WFL = 9 * NCT * 64 * KS;            // floats of the block's weights
  const lmn_conv_args_t& A = P.a;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const s_w = smem + 2 * NCH * 4;            // [tap][ct][lane][KS]
  float* const s_stats = s_w + WFL;                 // [2][NCT*16] (conv_stage_params's layout: statistics slots, then 9 parameter vectors)
  float* const s_par = s_stats + 2 * NCT * 16;
  if (P.prio >= 4) lmn_setprio_level(7 - P.prio);   // (uniform: lmn_set_priority_stream)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, n = lane & 15;
  const lmn_src_t& S = A.src[0];
  const int H = A.Hin, W = A.Win;                   // (stride 1, pad 1: input and output maps have the same size)
  const unsigned xbytes = (unsigned)(((int64_t)A.B * H * W - 1) * S.cstride + S.C) * 4u;
  const unsigned obytes = A.out ? (unsigned)(((int64_t)A.B * H * W - 1) * A.out_cstride + A.Cout) * 4u : 0u;
  const dma3_i32x4 rx = dma3_rsrc(S.ptr, xbytes);
  const BufRsrc ro = make_rsrc(A.out, obytes);
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem;
  const bool st_on = A.stats_mode == LMN_STATS_SUM_SQ;

  // per-thread chunk descriptors (tile-independent): window row / column and chunk of the pixel; -1: pad chunk or past the image
  int drc[NK];
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    const int i = k * 256 + tid;
    const int wp = i / PS, f = i - wp * PS;
    const int r = wp / XW, c = wp - r * XW;
    drc[k] = (i < NCH && f < CQ) ? (r << 16 | c << 8 | f) : -1;
  }
  const int tiles_img = P.tiles_x * P.tiles_y;
  unsigned svoff[NK];   // source offsets of the next tile's pieces
  auto stage_addr = [&](int tile) __attribute__((always_inline)) {
    const int b = tile / tiles_img, tt = tile - b * tiles_img;
    const int ty = tt / P.tiles_x, tx = tt - ty * P.tiles_x;
    const int wy0 = ty * TH - 1, wx0 = tx * TW - 1;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const int d = drc[k];
      const int iy = wy0 + (d >> 16), ix = wx0 + ((d >> 8) & 255);
      const bool ok = d >= 0 && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
      svoff[k] = ok ? (unsigned)(((b * H + iy) * W + ix) * S.cstride + (d & 255) * 4) * 4u : DMA_OOB;
    }
  };
  auto stage_piece = [&](int k, int buf) __attribute__((always_inline)) {
    const unsigned base = lds0 + (unsigned)buf * (NCH * 16) + (unsigned)wv * 1024;
    if ((k + 1) * 256 <= NCH || k * 256 + tid < NCH) dma3_piece(base + (unsigned)k * 4096, svoff[k], rx);   // (last piece: lanes past the image stay out)
  };

  const int t0 = blockIdx.x, tstep = gridDim.x;
  if (t0 < P.total_tiles) {
    stage_addr(t0);
#pragma unroll
    for (int k = 0; k < NK; ++k) stage_piece(k, 0);
  }
  // the block's weights, gathered from the packed fragments [tap][kb][ct][lane (q', m)][j] (channel 16 kb + 4 j + q') into
  // [tap][ct][lane (q, m)][s] (channel q KS + s); cout tiles past the end re-read the last one (dropped at the store)
  for (int i = tid; i < WFL; i += 256) {
    const int s = i % KS, t = i / KS;
    const int l = t & 63, tc = t >> 6;
    const int ct = tc % NCT, tap = tc / NCT;
    const int ch = (l >> 4) * KS + s, m = l & 15;
    const int kb = ch >> 4, j = (ch & 15) >> 2, qq = ch & 3;
    const int ctg = min(ct, P.NCTT - 1);
    s_w[i] = A.wpack[(((int64_t)tap * P.NKB + kb) * P.NCTT + ctg) * 256 + (qq * 16 + m) * 4 + j];
  }
  for (int i = tid; i < 2 * NCT * 16; i += 256) s_stats[i] = 0.f;
  conv_stage_params<NCT>(A, s_par, 0, tid, blockIdx.x == 0);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  f32x4 bias4[NCT], shift4[NCT];
#pragma unroll
  for (int c = 0; c < NCT; ++c) {
    bias4[c] = *reinterpret_cast<const f32x4*>(s_par + c * 16 + q * 4);                      // slot 0 = bias + bias2
    shift4[c] = *reinterpret_cast<const f32x4*>(s_par + 6 * NCT * 16 + c * 16 + q * 4);     // slot 6 = p4 (statistics about it)
  }
  float st0[NCT][4], st1[NCT][4];
#pragma unroll
  for (int c = 0; c < NCT; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) st0[c][r] = st1[c][r] = 0.f;
  // tap t of the packed weights meets the window pixel at (ty, tx) -- flipped for the data gradient (conv_tile_kernel's convention)
  const int flip = A.transposed ? 2 : 0;

  int it = 0;
  for (int tile = t0; tile < P.total_tiles; tile += tstep, ++it) {
    const int cur = it & 1;
    if (it > 0) {
      // the NPG * NCT stores of the previous tile are this wave's youngest vector-memory operations: everything older -- the LDS-DMA
      // pieces of THIS tile, issued during the previous MFMA loop -- has landed once at most that many are outstanding
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPG * NCT) : "memory");
      __builtin_amdgcn_s_barrier();
    }
    const bool has_next = tile + tstep < P.total_tiles;   // (block-uniform)
    if (has_next) stage_addr(tile + tstep);
    const int b = tile / tiles_img, tt = tile - b * tiles_img;
    const int tyi = tt / P.tiles_x, txi = tt - tyi * P.tiles_x;
    const float* XS = smem + cur * (NCH * 4);
    f32x4 acc[NPG][NCT];
#pragma unroll
    for (int g = 0; g < NPG; ++g)
#pragma unroll
      for (int c = 0; c < NCT; ++c) acc[g][c] = bias4[c];
    const float* xb = XS + (wv * XW + n) * (PS * 4) + q * KS;     // group g adds 4 * XW pixels
    const float* wb = s_w + lane * KS;
    // operands of tap t+1 are requested BEFORE the MFMAs of tap t
    Dma3Ops<KS> w[2][NCT], x[2][NPG];
#pragma unroll
    for (int c = 0; c < NCT; ++c) w[0][c] = dma3_ops<KS>(wb + c * 64 * KS);
#pragma unroll
    for (int g = 0; g < NPG; ++g) x[0][g] = dma3_ops<KS>(xb + ((4 * g + flip) * XW + flip) * (PS * 4));
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int cb = tap & 1, nb = cb ^ 1;
      if (tap < 8) {
        const int ty = (tap + 1) / 3, tx = (tap + 1) - ty * 3;
        const int fy = flip ? 2 - ty : ty, fx = flip ? 2 - tx : tx;
#pragma unroll
        for (int c = 0; c < NCT; ++c) w[nb][c] = dma3_ops<KS>(wb + ((tap + 1) * NCT + c) * 64 * KS);
#pragma unroll
        for (int g = 0; g < NPG; ++g) x[nb][g] = dma3_ops<KS>(xb + ((4 * g + fy) * XW + fx) * (PS * 4));
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int c = 0; c < NCT; ++c)
#pragma unroll
          for (int g = 0; g < NPG; ++g) acc[g][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[cb][c].v[s], x[cb][g].v[s], acc[g][c], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      // the next tile's pieces ride behind the MFMAs of the first taps, into a load queue that is never backed up
#pragma unroll
      for (int k = 0; k < NK; ++k)
        if (k * 9 / NK == tap && has_next) stage_piece(k, cur ^ 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    // epilogue: lane holds channels c*16 + q*4 .. +3 of pixel (row wv + 4g, column n); unconditional stores (dead lanes out of range:
    // the count of stores per tile is what the wait above relies on)
#pragma unroll
    for (int g = 0; g < NPG; ++g) {
      const int oy = tyi * TH + wv + 4 * g, ox = txi * TW + n;
      const bool pok = oy < H && ox < W;
      const uint32_t opx = pok ? (uint32_t)((b * H + oy) * W + ox) : 0u;
#pragma unroll
      for (int c = 0; c < NCT; ++c) {
        const int co = c * 16 + q * 4;
        const bool live = pok && co < A.Cout;
        f32x4 o = acc[g][c];
        if (st_on && live) {
#pragma unroll
          for (int r = 0; r < 4; ++r) { const float d = o[r] - shift4[c][r]; st0[c][r] += d; st1[c][r] += d * d; }
        }
        if (A.residual) o += ld4((const float*)A.residual + opx * A.res_cstride + (live ? co : 0));
        const unsigned voff = (live && A.out) ? (opx * (uint32_t)A.out_cstride + (uint32_t)co) * 4u : DMA_OOB;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), ro, (int)voff, 0, 0);
      }
    }
  }

  // ---- statistics (conv_tile_kernel's tail): wave shuffle over the 16 pixels -> LDS, the four waves added in wave order -> one
  //      atomic per channel and block, or the block's slot in deterministic mode
  if (st_on) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    float* XS = smem;
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float a = st0[c][r], bb = st1[c][r];
#pragma unroll
        for (int m = 1; m <= 8; m <<= 1) {
          a += __shfl_xor(a, m, 64);
          bb += __shfl_xor(bb, m, 64);
        }
        if (n == 0) {
          XS[wv * 2 * NCT * 16 + c * 16 + q * 4 + r] = a;
          XS[wv * 2 * NCT * 16 + NCT * 16 + c * 16 + q * 4 + r] = bb;
        }
      }
    __syncthreads();
    const bool det = P.det_stats != nullptr;
    for (int i = tid; i < 2 * NCT * 16; i += 256) {
      const int which = i / (NCT * 16), co = i - which * NCT * 16;
      const float v = ((XS[i] + XS[2 * NCT * 16 + i]) + XS[4 * NCT * 16 + i]) + XS[6 * NCT * 16 + i];
      if (co < A.Cout) {
        if (det) P.det_stats[(int64_t)blockIdx.x * 2 * A.Cout + (int64_t)which * A.Cout + co] = v;
        else atomicAdd(A.stats + (A.stats_rep > 1 ? (int64_t)(blockIdx.x % A.stats_rep) * 2 * A.Cout : 0) + (int64_t)which * A.Cout + co, v);
      }
    }
  }
}

}  // namespace

// LDS bytes of an instance (window double buffer + weights + statistics / parameter vectors)
size_t lmn_conv_dma3_lds(int C, int nct) {
  const int ks = C / 4, ps = (ks & 1) ? ks : ks + 1;
  return (size_t)2 * 10 * 18 * ps * 16 + (size_t)9 * nct * 64 * ks * 4 + (size_t)(2 + 9) * nct * 16 * 4;
}

int lmn_launch_conv_dma3(const ConvParams& T, int blocks, hipStream_t st) {
  const int C = T.a.src[0].C, nct = T.NCTT;
  const size_t sh = lmn_conv_dma3_lds(C, nct);
#define LMN_D3(KSV, NN, BPCV)                                                                                                            \
  do {                                                                                                                                    \
    if (sh > 64 * 1024) (void)hipFuncSetAttribute((const void*)conv_dma3_kernel<KSV, NN, BPCV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); \
    LMN_LAUNCH((conv_dma3_kernel<KSV, NN, BPCV>), dim3(blocks), dim3(256), sh, st, T);                                                    \
  } while (0)
  if (C == 12 && nct == 1) LMN_D3(3, 1, 4);
  else if (C == 12 && nct == 2) LMN_D3(3, 2, 4);
  else if (C == 24 && nct == 1) LMN_D3(6, 1, 2);
  else if (C == 24 && nct == 2) LMN_D3(6, 2, 2);
  else return -1;
#undef LMN_D3
  return 0;
}
