// Weight gradients of the dense-convolution family (K = pixels): wgrad_lds_kernel (general 3x3 / small maps), wgrad3_kernel (3x3
// stride 1 on maps >= 32 wide), wgrad_1x1_kernel / wgrad_1x1w_kernel (1x1 / Linear), the K-split reductions, and their host entry
// (lmn_conv_wgrad).  See conv_fwd.hip for the family's design notes.
#include "conv_common.h"

namespace {

// ------------------------------------------------------------------------------------ weight gradient
struct WgradParams {
  LmnLay lay_src[3], lay_dy;   // address forms of the operands (common.h: NHWC or row-planar; 1x1 kernels only)
  int32_t rpw;                 // image width of the row-planar operands (0: none)
  uint32_t rp_magic;
  lmn_wgrad_args_t a;
  int ntile_src[3];  // 16-channel tiles per source
  int ntile_off[3];
  int cbase[3];
  int NNTT, NMTT;    // total cin tiles (sum over sources), total cout tiles
  int Cin;
  int nsets_n;       // number of cin tile sets
  int steps_per_img;
  int total_steps;
  float inv_keep_src[3];
  float inv_keep_dy;
  float* partial;    // two-stage reduction: [gridDim.y][gridDim.x][NT*256 + NMT*16] block partials, or NULL (atomics)
  // LDS-staged kernel: output-pixel tile TH x TW, its input window XH x XW, LDS pixel strides, tile counts
  int TH, TW, XH, XW, CSx, CSy, tiles_x, tiles_y, total_tiles;
  uint32_t mXW, mTW;  // magic multipliers: n / d == (n * m) >> 32 for n, d < 2^16
  int dbg;
};

// first element (tap 0) of dW[co][ch of source sidx]: the source's own gradient tensor if the caller gave one,
// else its column range of the concatenated dW
__device__ __forceinline__ float* wgrad_dst(const WgradParams& P, int co, int sidx, int ch, int taps) {
  float* own = P.a.dW_src[sidx];
  return own ? own + ((int64_t)co * P.a.src[sidx].C + ch) * taps
             : P.a.dW + ((int64_t)co * P.Cin + P.cbase[sidx] + ch) * taps;
}

// ------------------------------------------------------------------------------------ weight gradient, LDS-staged
// dW[co][ci][tap] = sum_p dy[p][co] * x[p*s + tap - pad][ci]   as an MFMA GEMM with K = pixels.
// A block walks a contiguous range of output-pixel tiles.  Per tile the input window (with halo) and the dy
// tile are staged ONCE into LDS by coalesced float4 loads (on-load transforms -- GELU, SE scale, dropout mask
// -- applied here, once per element); the four waves then split the tile's 4-pixel K steps and read both MFMA
// operands from LDS as conflict-free ds_read_b32 (lanes along channels; pixel stride = 16 mod 32 banks).
// One global load per element instead of one per (element, tap): the direct form was bound by the texture
// addresser (one 4-segment dword load per MFMA).  Accumulators stay in registers across all tiles of the block.
// BF: operands rounded to bf16 when staged ([tile][pixel][16 bf16] planes, pixel stride CS dwords), a K step is 16 pixels
// = ONE v_mfma_f32_16x16x16_bf16 per (tap, cout tile, cin tile): lane (q, n) gathers pixels 4q..4q+3 of channel n with
// four ds_read_u16 per operand (the same LDS instruction count per pixel as the fp32 form, an eighth of its MFMA time).
template <int TAPS, int NMT, int NNT, int PM = 0>
__global__ __launch_bounds__(256, 2) void wgrad_lds_kernel(const WgradParams P) {
  constexpr bool BF = PM >= 1;
  typedef typename ActT<PM>::type TA;
  const lmn_wgrad_args_t& A = P.a;
  const uint32_t soff = A.seed_ctr ? *A.seed_ctr : 0u;  // device-side dropout stream offset (graph replays: one bump per step)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // one 16-channel PLANE per cin / cout tile: [tile][pixel][16] -- a K step reads 4 pixels x 16 channels = 64
  // consecutive floats of one plane (conflict-free ds_read_b32 at stride 1), and no padding is needed
  const int XP = P.XH * P.XW;
  float* XS = smem;                                   // [NNT][XH*XW][16]
  float* YS = smem + NNT * XP * P.CSx;                // [NMT][TH*TW][16]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave index as an SGPR: branches on it stay scalar
  const int q = lane >> 4, n = lane & 15;
  const int mset = blockIdx.y / P.nsets_n, nset = blockIdx.y - mset * P.nsets_n;
  const int mt0 = mset * NMT, nt0 = nset * NNT;
  const int pad = A.ksize >> 1;
  const int NP = P.TH * P.TW;

  // per cin tile of this block: source and channel base (block-uniform, hoisted out of the tile loop)
  const TA* tptr[NNT];
  const float* tscale[NNT];
  int tC[NNT], tcs[NNT], tflags[NNT], tch0[NNT];
  uint32_t tseed[NNT];
  float tp_[NNT], tik[NNT];
#pragma unroll
  for (int t = 0; t < NNT; ++t) {
    const int nt = nt0 + t;
    int sidx = 0;
    while (sidx + 1 < A.nsrc && nt >= P.ntile_off[sidx + 1]) ++sidx;
    tptr[t] = (const TA*)A.src[sidx].ptr;
    tscale[t] = A.src[sidx].scale;
    tC[t] = nt < P.NNTT ? A.src[sidx].C : 0;
    tcs[t] = A.src[sidx].cstride;
    tflags[t] = A.src[sidx].flags;
    tseed[t] = A.src[sidx].drop_seed + soff;
    tp_[t] = A.src[sidx].drop_p;
    tik[t] = P.inv_keep_src[sidx];
    tch0[t] = (nt - P.ntile_off[sidx]) * 16;
  }

  f32x4 acc[TAPS][NMT][NNT];
  float bsum[NMT];  // bias gradient: running sum of this lane's dy values (pixel q of every K step, channel n)
#pragma unroll
  for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
    for (int m = 0; m < NMT; ++m)
#pragma unroll
      for (int t = 0; t < NNT; ++t) acc[tp][m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int m = 0; m < NMT; ++m) bsum[m] = 0.f;

  const int t_begin = (int)(((int64_t)blockIdx.x * P.total_tiles) / gridDim.x);
  const int t_end = (int)(((int64_t)(blockIdx.x + 1) * P.total_tiles) / gridDim.x);
  for (int tile = t_begin; tile < t_end; ++tile) {
    const int b = tile / (P.tiles_x * P.tiles_y), tt = tile - b * P.tiles_x * P.tiles_y;
    const int oy0 = (tt / P.tiles_x) * P.TH, ox0 = (tt % P.tiles_x) * P.TW;
    const int iy0 = oy0 * A.stride - pad, ix0 = ox0 * A.stride - pad;
    __syncthreads();  // previous tile's reads are done
    // ---- stage the input window and the dy tile: float4 items (pixel, quad).  ALL global loads of a round (up to
    //      UX x 256 window items and UY x 256 dy items -- normally the whole tile) are issued before the first one is
    //      consumed: one exposed memory latency per tile instead of one per 256 items (the K loop of a tile is
    //      shorter than two such latencies).
    constexpr int UX = (TAPS == 9 && NMT * NNT == 4) ? 3 : 4, UY = 2;
    const int NXI = P.XH * P.XW * 4, NYI = NP * 4;
    for (int rd = 0; rd * (UX * 256) < NXI || rd * (UY * 256) < NYI; ++rd) {
      f32x4 vx[UX][NNT], vy[UY][NMT];
      int gpx[UX], gpy[UY];  // global pixel index, -1 = outside the image
#pragma unroll
      for (int u = 0; u < UX; ++u) {
        const int i = rd * (UX * 256) + u * 256 + tid;
        const int j = i & 3, pix = i >> 2;
        bool inb;
        int gp;
        if constexpr (TAPS == 1) {  // 1x1: the image is one flat row (host), the window is the tile itself
          inb = i < NXI && ox0 + pix < A.Wout;
          gp = inb ? b * A.Wout + ox0 + pix : 0;
        } else {
          const int r = (int)__umulhi((uint32_t)pix, P.mXW), c = pix - r * P.XW;
          const int iy = iy0 + r, ix = ix0 + c;
          inb = i < NXI && (unsigned)iy < (unsigned)A.Hin && (unsigned)ix < (unsigned)A.Win;
          gp = inb ? (b * A.Hin + iy) * A.Win + ix : 0;
        }
        gpx[u] = inb ? gp : -1;
#pragma unroll
        for (int t = 0; t < NNT; ++t) {
          const int ch = tch0[t] + j * 4;
          vx[u][t] = ld4(tptr[t] + (uint32_t)(gp * tcs[t] + (ch < tC[t] ? ch : 0)));
        }
      }
#pragma unroll
      for (int u = 0; u < UY; ++u) {
        const int i = rd * (UY * 256) + u * 256 + tid;
        const int j = i & 3, pix = i >> 2;
        bool inb;
        int gp;
        if constexpr (TAPS == 1) {
          inb = i < NYI && ox0 + pix < A.Wout;
          gp = inb ? b * A.Wout + ox0 + pix : 0;
        } else {
          const int r = (int)__umulhi((uint32_t)pix, P.mTW), c = pix - r * P.TW;
          const int oy = oy0 + r, ox = ox0 + c;
          inb = i < NYI && oy < A.Hout && ox < A.Wout;
          gp = inb ? (b * A.Hout + oy) * A.Wout + ox : 0;
        }
        gpy[u] = inb ? gp : -1;
#pragma unroll
        for (int m = 0; m < NMT; ++m) {
          const int co = (mt0 + m) * 16 + j * 4;
          const bool cok = (mt0 + m) < P.NMTT && co < A.Cout;
          vy[u][m] = ld4((const TA*)A.dy + (uint32_t)(gp * A.dy_cstride + (cok ? co : 0)));
        }
      }
#pragma unroll
      for (int u = 0; u < UX; ++u) {
        const int i = rd * (UX * 256) + u * 256 + tid;
        if (i < NXI) {
          const int j = i & 3, pix = i >> 2;
          const bool inb = gpx[u] >= 0;
          const int gp = inb ? gpx[u] : 0;
#pragma unroll
          for (int t = 0; t < NNT; ++t) {
            const int ch = tch0[t] + j * 4;
            const bool ok = inb && ch < tC[t];
            const int chs = ch < tC[t] ? ch : 0;
            f32x4 w = vx[u][t];
            if (tflags[t] & LMN_SRC_GELU) {
#pragma unroll
              for (int k = 0; k < 4; ++k) w[k] = lmn_gelu(w[k]);
            }
            if (tflags[t] & LMN_SRC_DROP) {
#pragma unroll
              for (int k = 0; k < 4; ++k) w[k] *= lmn_drop_scale(tseed[t], (uint32_t)(gp * tC[t] + chs + k), tp_[t], tik[t]);
            }
            if (tscale[t]) w *= ld4(tscale[t] + (inb ? b : 0) * tC[t] + chs);
            if (!ok) w = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (BF) *reinterpret_cast<uint2*>(&XS[(t * XP + pix) * P.CSx + j * 2]) = pk4_bf16(w);
            else *reinterpret_cast<f32x4*>(&XS[(t * XP + pix) * P.CSx + j * 4]) = w;
          }
        }
      }
#pragma unroll
      for (int u = 0; u < UY; ++u) {
        const int i = rd * (UY * 256) + u * 256 + tid;
        if (i < NYI) {
          const int j = i & 3, pix = i >> 2;
          const bool inb = gpy[u] >= 0;
          const int gp = inb ? gpy[u] : 0;
#pragma unroll
          for (int m = 0; m < NMT; ++m) {
            const int co = (mt0 + m) * 16 + j * 4;
            const bool cok = (mt0 + m) < P.NMTT && co < A.Cout;
            const int cos = cok ? co : 0;
            f32x4 w = vy[u][m];
            if (A.dy_flags & LMN_SRC_DROP) {
#pragma unroll
              for (int k = 0; k < 4; ++k) w[k] *= lmn_drop_scale(A.dy_seed + soff, (uint32_t)(gp * A.Cout + cos + k), A.dy_p, P.inv_keep_dy);
            }
            if (!(inb && cok)) w = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (BF) *reinterpret_cast<uint2*>(&YS[(m * NP + pix) * P.CSy + j * 2]) = pk4_bf16(w);
            else *reinterpret_cast<f32x4*>(&YS[(m * NP + pix) * P.CSy + j * 4]) = w;
          }
        }
      }
    }
    __syncthreads();
    if constexpr (BF) {
      // ---- bf16: K steps of 16 consecutive tile pixels; lane (q, n) owns pixels 4q..4q+3 of the step, channel n
      const uint16_t* XH16 = reinterpret_cast<const uint16_t*>(XS);
      const uint16_t* YH16 = reinterpret_cast<const uint16_t*>(YS);
      for (int ks = wv; ks * 16 < NP; ks += 4) {
        int xb[4];
        bool pin[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int pix = ks * 16 + q * 4 + j;
          pin[j] = pix < NP;
          const int pixs = pin[j] ? pix : 0;
          if constexpr (TAPS == 1) {
            xb[j] = pixs * P.CSx * 2 + n;
          } else {
            const int pr = (int)__umulhi((uint32_t)pixs, P.mTW), pc = pixs - pr * P.TW;
            xb[j] = ((pr * A.stride) * P.XW + pc * A.stride) * P.CSx * 2 + n;
          }
        }
        uint2 av[NMT], bvv[TAPS][NNT];
#pragma unroll
        for (int m = 0; m < NMT; ++m) {
          uint32_t h[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int pixs = pin[j] ? ks * 16 + q * 4 + j : 0;
            h[j] = pin[j] ? (uint32_t)YH16[(m * NP + pixs) * P.CSy * 2 + n] : 0u;
            bsum[m] += __builtin_bit_cast(float, h[j] << 16);
          }
          av[m] = uint2{h[0] | (h[1] << 16), h[2] | (h[3] << 16)};
        }
#pragma unroll
        for (int tp = 0; tp < TAPS; ++tp) {
          const int ty = (TAPS == 9) ? tp / 3 : 0, tx = (TAPS == 9) ? tp % 3 : 0;
#pragma unroll
          for (int t = 0; t < NNT; ++t) {
            const int toff = (t * XP + ty * P.XW + tx) * P.CSx * 2;
            const uint32_t h0 = XH16[xb[0] + toff], h1 = XH16[xb[1] + toff], h2 = XH16[xb[2] + toff], h3 = XH16[xb[3] + toff];
            bvv[tp][t] = uint2{h0 | (h1 << 16), h2 | (h3 << 16)};
          }
        }
#pragma unroll
        for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
          for (int t = 0; t < NNT; ++t)
#pragma unroll
            for (int m = 0; m < NMT; ++m) acc[tp][m][t] = mfma_bf16(av[m], bvv[tp][t], acc[tp][m][t]);
      }
    } else
    // ---- MFMA over this wave's K steps (4 consecutive tile pixels each); all LDS reads of a step are issued
    //      before its MFMAs so their latency overlaps
    for (int ks = wv; ks * 4 < NP; ks += 4) {
      const int pix = ks * 4 + q;
      const bool pin = pix < NP;
      const int pixs = pin ? pix : 0;
      float av[NMT], bvv[TAPS][NNT];
#pragma unroll
      for (int m = 0; m < NMT; ++m) av[m] = YS[(m * NP + pixs) * P.CSy + n];
      int xb;
      if constexpr (TAPS == 1) {
        xb = pixs * P.CSx + n;
      } else {
        const int pr = (int)__umulhi((uint32_t)pixs, P.mTW), pc = pixs - pr * P.TW;
        xb = ((pr * A.stride) * P.XW + pc * A.stride) * P.CSx + n;
      }
#pragma unroll
      for (int tp = 0; tp < TAPS; ++tp) {
        const int ty = (TAPS == 9) ? tp / 3 : 0, tx = (TAPS == 9) ? tp % 3 : 0;
#pragma unroll
        for (int t = 0; t < NNT; ++t) bvv[tp][t] = XS[xb + (t * XP + ty * P.XW + tx) * P.CSx];
      }
      if (!pin) {
#pragma unroll
        for (int m = 0; m < NMT; ++m) av[m] = 0.f;
      }
#pragma unroll
      for (int m = 0; m < NMT; ++m) bsum[m] += av[m];
#pragma unroll
      for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
        for (int t = 0; t < NNT; ++t)
#pragma unroll
          for (int m = 0; m < NMT; ++m) acc[tp][m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], bvv[tp][t], acc[tp][m][t], 0, 0, 0);
    }
  }

  // ---- block-level reduction in LDS (4 waves -> 1): plain stores / read-add-stores in four wave rounds.
  //      (LDS float atomics cost ~3 cycles per LANE on gfx950 -- measured 180 us for this tail with ds_add_f32.)
  constexpr int NT = TAPS * NMT * NNT;
  __syncthreads();
  float* s_acc = smem;  // reuse the staging area (>= NT*256 + NMT*16 floats, checked on the host)
  for (int w = 0; w < 4; ++w) {
    if (wv == w) {
#pragma unroll
      for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
        for (int m = 0; m < NMT; ++m)
#pragma unroll
          for (int t = 0; t < NNT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float* d = &s_acc[(((tp * NMT + m) * NNT + t) * 4 + r) * 64 + lane];
              *d = (w == 0) ? acc[tp][m][t][r] : *d + acc[tp][m][t][r];
            }
#pragma unroll
      for (int m = 0; m < NMT; ++m) {  // bias: sum the 4 pixel lanes (q) of channel n
        float v = bsum[m];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (q == 0) {
          float* d = &s_acc[NT * 256 + m * 16 + n];
          *d = (w == 0) ? v : *d + v;
        }
      }
    }
    __syncthreads();
  }
  if (P.partial) {
    float* dst = P.partial + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * (NT * 256 + NMT * 16);
    for (int i = tid; i < NT * 256 + NMT * 16; i += 256) dst[i] = s_acc[i];
    return;
  }
  for (int i = tid; i < NT * 256; i += 256) {
    const int ln = i & 63, r = (i >> 6) & 3, tl = i >> 8;
    const int t = tl % NNT, m = (tl / NNT) % NMT, tp = tl / (NNT * NMT);
    const int qq = ln >> 4, nn = ln & 15;
    const int co = (mt0 + m) * 16 + qq * 4 + r;
    const int nt = nt0 + t;
    if ((mt0 + m) >= P.NMTT || co >= A.Cout || nt >= P.NNTT) continue;
    int sidx = 0;
    while (sidx + 1 < A.nsrc && nt >= P.ntile_off[sidx + 1]) ++sidx;
    const int ch = (nt - P.ntile_off[sidx]) * 16 + nn;
    if (ch >= A.src[sidx].C) continue;
    atomicAdd(wgrad_dst(P, co, sidx, ch, TAPS) + tp, s_acc[i]);
  }
  if (A.db && nset == 0) {
    for (int i = tid; i < NMT * 16; i += 256) {
      const int co = mt0 * 16 + i;
      if (co < A.Cout && (mt0 + i / 16) < P.NMTT) {
        atomicAdd(A.db + co, s_acc[NT * 256 + i]);
        if (A.db2) atomicAdd(A.db2 + co, s_acc[NT * 256 + i]);
      }
    }
  }
}

// ------------------------------------------------------------------------------------ weight gradient, 3x3 stride 1, maps >= 32 wide
// The LDS-staged form above specialised for the layers that carry most of the weight-gradient time (tile 32 pixels wide,
// window 34 wide, one-tile or 2 x 2-tile blocks).  Against the general kernel:
//  * 2 x 2-tile blocks give each wave ONE output tile (wave w: cout tile w / 2, cin tile w % 2) with all 9 taps over ALL K steps
//    of the tile, instead of all four tiles over every fourth step: 36 accumulator VGPRs instead of 144 (room for a third
//    resident block per CU and the prefetch registers below) and no cross-wave reduction at the end;
//  * the global loads of tile i+1 are issued BEFORE the K loop of tile i and committed to LDS after it, so a tile's memory
//    latency hides behind this block's own MFMAs, not only behind those of the CU's other blocks;
//  * index math is hoisted out of the per-tile / per-step paths (it ran at ~35 VALU per K step and ~25 per staged item, which
//    on 3-wave SIMDs competes with the MFMA issue slots -- measured: MFMA busy 47 % with the K phase at 70 % of a block's
//    life).  A K step lies in ONE tile row (32 | tile width): row / column are scalar shifts of the step index and the nine tap
//    reads are immediate offsets from one lane address.  Staged items are addressed by per-thread offsets computed once per
//    kernel plus a scalar tile base; raw buffer loads return 0 for the halo outside the image (offset forced out of range),
//    so the commit is a plain register -> LDS copy when the source has no on-load transform (bf16 storage: the loaded words
//    go to LDS untouched).
// LDS planes are padded to whole 64-pixel item rounds ([tile][UX * 64 px][CS]): every plane / round offset is an immediate.
template <int ESZ> struct RawOf { typedef f32x4 type; };
template <> struct RawOf<2> { typedef u32x2 type; };
__device__ __forceinline__ f32x4 raw_f32(f32x4 v) { return v; }
__device__ __forceinline__ f32x4 raw_f32(u32x2 v) { return f32x4{lmn_bf16_lo(v.x), lmn_bf16_hi(v.x), lmn_bf16_lo(v.y), lmn_bf16_hi(v.y)}; }
template <typename RawT> __device__ __forceinline__ RawT raw_load(BufRsrc r, uint32_t off) {
  if constexpr (sizeof(RawT) == 16) return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0));
  else return __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, 0);
}
#ifdef LMN_WG_TIMING
// phase clocks (debug builds only): per block {tile-top barrier, commit, second barrier + issue, rest, start, end}
__device__ unsigned long long g_wg_timing[4096 * 6];
#define LMN_TCLK() __builtin_amdgcn_s_memtime()
#endif

// UP (LMN_SRC_UP2, one source): the source tensor is the HALF-resolution map and the window is its bilinear x2 upsampling
// (align_corners=True, lmn_up_coord: the arithmetic of lmn_up2_fwd and of the forward conv's LMN_SRC_UP2 staging).  The prefetch holds the
// QUARTER-SIZE source window of the next tile (<= 5 x 19 / 7 x 19 pixels: 2 / 3 items per thread instead of 4 / 6); commit parks it in a
// small LDS plane, one more block barrier, and every thread interpolates its window items LDS -> LDS (four float4 reads per item).  `up`
// never exists in HBM: no lmn_up2_fwd launch in the backward, no second tensor of 4x the pixels.
template <int NMT, int NNT, int PM, bool UP = false>
__global__ __launch_bounds__(256, 3) void wgrad3_kernel(const WgradParams P) {
  static_assert(NMT * NNT == 4 || NMT * NNT == 1, "one-tile or 2 x 2-tile blocks");
  constexpr bool TS = NMT * NNT == 4;  // one output tile per wave
  constexpr bool BF = PM >= 1;
  typedef typename ActT<PM>::type TA;
  constexpr int ESZ = sizeof(TA);
  typedef typename RawOf<ESZ>::type RawT;
  constexpr int XWC = 34, CS = BF ? 12 : 16;  // window width; dwords per LDS pixel (bf16: 8 + 4 pad)
  constexpr int QW = BF ? 2 : 4;              // dwords of one staged (pixel, channel quad) item in LDS
  constexpr int UX = TS ? 4 : 6, UY = TS ? 2 : 4;  // staged items per thread: window, dy tile
  constexpr int XPA = UX * 64, NPA = UY * 64;      // padded plane sizes in pixels
  const lmn_wgrad_args_t& A = P.a;
  const uint32_t soff = A.seed_ctr ? *A.seed_ctr : 0u;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* XS = smem;                     // [NNT][XPA][CS]
  float* YS = smem + NNT * XPA * CS;    // [NMT][NPA][CS]
  constexpr int USX = TS ? 2 : 3;       // UP: source-window items per thread (19 columns x <= 5 / 7 rows x 4 quads)
  constexpr int SCW = 19, SPA = USX * 64;   // source window width; padded plane size in pixels
  float* S2 = YS + NMT * NPA * CS;      // UP: [NNT][SPA][16] fp32 source window
  (void)S2;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, n = lane & 15;
  const int m_w = TS ? wv / NNT : 0, t_w = TS ? wv % NNT : 0;
  const int mset = blockIdx.y / P.nsets_n, nset = blockIdx.y - mset * P.nsets_n;
  const int mt0 = mset * NMT, nt0 = nset * NNT;
  const int NP = P.TH * 32;
  const int j = tid & 3, pl = tid >> 2;  // this thread's channel quad and pixel-in-round of every staged item

  // per cin tile of this block: source, channel base, buffer descriptor
  const float* tscale[NNT];
  int tC[NNT], tcs[NNT], tflags[NNT], tch0[NNT];
  uint32_t tseed[NNT];
  float tp_[NNT], tik[NNT];
  BufRsrc rx[NNT];
  bool cx[NNT];
#pragma unroll
  for (int t = 0; t < NNT; ++t) {
    const int nt = nt0 + t;
    int sidx = 0;
    while (sidx + 1 < A.nsrc && nt >= P.ntile_off[sidx + 1]) ++sidx;
    tscale[t] = A.src[sidx].scale;
    tC[t] = nt < P.NNTT ? A.src[sidx].C : 0;
    tcs[t] = A.src[sidx].cstride;
    tflags[t] = A.src[sidx].flags;
    tseed[t] = A.src[sidx].drop_seed + soff;
    tp_[t] = A.src[sidx].drop_p;
    tik[t] = P.inv_keep_src[sidx];
    tch0[t] = (nt - P.ntile_off[sidx]) * 16;
    rx[t] = make_rsrc(A.src[sidx].ptr, (unsigned)((int64_t)A.B * (UP ? (A.Hin >> 1) * (A.Win >> 1) : A.Hin * A.Win) * tcs[t] * ESZ));
    cx[t] = tch0[t] + j * 4 < tC[t];
  }
  const BufRsrc ry = make_rsrc(A.dy, (unsigned)((int64_t)A.B * A.Hout * A.Wout * A.dy_cstride * ESZ));

  // tile-invariant item descriptors: window items (row, column) and byte offsets from the window's first pixel
  uint32_t xrc[UX], xo[UX][NNT];
#pragma unroll
  for (int u = 0; u < UX; ++u) {
    const int pix = u * 64 + pl, r = pix / XWC, c = pix - r * XWC;
    xrc[u] = r < P.XH ? (uint32_t)(r << 16 | c) : 0x7fff0000u;  // rows past the window never pass the bounds test
#pragma unroll
    for (int t = 0; t < NNT; ++t) xo[u][t] = (uint32_t)(((r * A.Win + c) * tcs[t] + tch0[t] + j * 4) * ESZ);
  }
  // dy items: pixel u * 64 + pl of the 32-wide tile = row 2u + (pl >> 5), column pl & 31
  const int yc = pl & 31, yr0 = pl >> 5;
  uint32_t yo0[NMT];
  bool cy[NMT];
#pragma unroll
  for (int m = 0; m < NMT; ++m) {
    const int co = (mt0 + m) * 16 + j * 4;
    cy[m] = (mt0 + m) < P.NMTT && co < A.Cout;
    yo0[m] = (uint32_t)(((yr0 * A.Wout + yc) * A.dy_cstride + (cy[m] ? co : 0)) * ESZ);
  }
  const uint32_t ystep = (uint32_t)(2 * A.Wout * A.dy_cstride * ESZ);  // two tile rows

  f32x4 acc[9];
  float bsum = 0.f;  // bias gradient: this lane's dy values (pixel q of every K step, channel n)
#pragma unroll
  for (int tp = 0; tp < 9; ++tp) acc[tp] = f32x4{0.f, 0.f, 0.f, 0.f};

  RawT vx[UP ? 1 : UX][NNT], vy[UY][NMT];
  RawT vs[UP ? USX : 1][NNT];           // UP: the source window of the next tile
  const int tpi = P.tiles_x * P.tiles_y;
  // UP: source geometry.  hs x ws = the source map; the window rows iy0 .. iy0 + XH - 1 / columns ix0 .. ix0 + 33 of the upsampled image
  // (clipped to it) read source rows ys0 .. and columns xs0 .. (first taps of the first clipped row / column)
  const int hs = A.Hin >> 1, ws = A.Win >> 1;
  const float up_sh = UP ? (float)(hs - 1) / (float)(A.Hin - 1) : 0.f, up_sw = UP ? (float)(ws - 1) / (float)(A.Win - 1) : 0.f;
  auto src_org = [&](int iy0, int ix0, int& ys0, int& xs0) __attribute__((always_inline)) {
    ys0 = (int)(up_sh * (float)(iy0 > 0 ? iy0 : 0));
    xs0 = (int)(up_sw * (float)(ix0 > 0 ? ix0 : 0));
  };
  auto issue = [&](int tile) __attribute__((always_inline)) {
    const int b = tile / tpi, tt = tile - b * tpi;
    const int ty_ = tt / P.tiles_x;
    const int oy0 = ty_ * P.TH, ox0 = (tt - ty_ * P.tiles_x) * 32;
    const int iy0 = oy0 - 1, ix0 = ox0 - 1;
    const int wpix = (b * A.Hin + iy0) * A.Win + ix0;  // window's first pixel (may lie before the tensor: wraps back below)
    if constexpr (UP) {
      int ys0, xs0;
      src_org(iy0, ix0, ys0, xs0);
#pragma unroll
      for (int u = 0; u < USX; ++u) {
        const int sp = u * 64 + pl, sr = sp / SCW, sc = sp - sr * SCW;
        const bool inb = ys0 + sr < hs && xs0 + sc < ws;
#pragma unroll
        for (int t = 0; t < NNT; ++t) {
          const uint32_t off = (uint32_t)((((b * hs + ys0 + sr) * ws + xs0 + sc) * tcs[t] + tch0[t] + j * 4) * ESZ);
          vs[u][t] = raw_load<RawT>(rx[t], (inb && cx[t]) ? off : 0xffffffffu);
        }
      }
    } else {
#pragma unroll
    for (int u = 0; u < UX; ++u) {
      const int r = (int)(xrc[u] >> 16), c = (int)(xrc[u] & 0xffffu);
      const bool inb = (unsigned)(iy0 + r) < (unsigned)A.Hin && (unsigned)(ix0 + c) < (unsigned)A.Win;
#pragma unroll
      for (int t = 0; t < NNT; ++t) {
        const uint32_t off = (uint32_t)(wpix * tcs[t] * ESZ) + xo[u][t];
        vx[u][t] = raw_load<RawT>(rx[t], (inb && cx[t]) ? off : 0xffffffffu);
      }
    }
    }   // (!UP)
    const uint32_t ybase = (uint32_t)(((b * A.Hout + oy0) * A.Wout + ox0) * A.dy_cstride * ESZ);
    const bool cok = ox0 + yc < A.Wout;
    const int rlim = (A.Hout - oy0 < P.TH ? A.Hout - oy0 : P.TH) - yr0;  // rows of this tile inside the image
#pragma unroll
    for (int u = 0; u < UY; ++u) {
      const bool inb = cok && 2 * u < rlim;
#pragma unroll
      for (int m = 0; m < NMT; ++m)
        vy[u][m] = raw_load<RawT>(ry, (inb && cy[m]) ? ybase + (uint32_t)u * ystep + yo0[m] : 0xffffffffu);
    }
  };
  float* xl = XS + pl * CS + j * QW;
  float* yl = YS + pl * CS + j * QW;
  auto commit = [&](int tile) __attribute__((always_inline)) {
    const int b = tile / tpi, tt = tile - b * tpi;
    const int ty_ = tt / P.tiles_x;
    const int oy0 = ty_ * P.TH, ox0 = (tt - ty_ * P.tiles_x) * 32;
    if constexpr (UP) {
      // the quarter-size source window -> S2 (fp32), barrier, then the window items by interpolation
#pragma unroll
      for (int t = 0; t < NNT; ++t)
#pragma unroll
        for (int u = 0; u < USX; ++u) *reinterpret_cast<f32x4*>(&S2[((t * SPA) + u * 64 + pl) * 16 + j * 4]) = raw_f32(vs[u][t]);
      __syncthreads();
      const int iy0 = oy0 - 1, ix0 = ox0 - 1;
      int ys0, xs0;
      src_org(iy0, ix0, ys0, xs0);
#pragma unroll
      for (int u = 0; u < UX; ++u) {
        const int r = (int)(xrc[u] >> 16), c = (int)(xrc[u] & 0xffffu);
        const int iy = iy0 + r, ix = ix0 + c;
        const bool inb = (unsigned)iy < (unsigned)A.Hin && (unsigned)ix < (unsigned)A.Win;
        int y0, yp, x0, xp;
        float ly0, ly1, lx0, lx1;
        lmn_up_coord(inb ? iy : 0, hs, up_sh, y0, yp, ly0, ly1);
        lmn_up_coord(inb ? ix : 0, ws, up_sw, x0, xp, lx0, lx1);
        const int l00 = inb ? (y0 - ys0) * SCW + (x0 - xs0) : 0;
#pragma unroll
        for (int t = 0; t < NNT; ++t) {
          const float* sp = &S2[(t * SPA + l00) * 16 + j * 4];
          const f32x4 v00 = *reinterpret_cast<const f32x4*>(sp), v01 = *reinterpret_cast<const f32x4*>(sp + xp * 16);
          const f32x4 v10 = *reinterpret_cast<const f32x4*>(sp + yp * SCW * 16), v11 = *reinterpret_cast<const f32x4*>(sp + (yp * SCW + xp) * 16);
          f32x4 w = ly0 * (lx0 * v00 + lx1 * v01) + ly1 * (lx0 * v10 + lx1 * v11);   // (the expression of up2_fwd_kernel)
          if (!inb) w = f32x4{0.f, 0.f, 0.f, 0.f};
          float* dst = xl + (t * XPA + u * 64) * CS;
          if constexpr (BF) *reinterpret_cast<uint2*>(dst) = pk4_bf16(w);
          else *reinterpret_cast<f32x4*>(dst) = w;
        }
      }
    } else {
#pragma unroll
    for (int t = 0; t < NNT; ++t) {
      const bool tf = tflags[t] != 0 || tscale[t] != nullptr;  // block-uniform
#pragma unroll
      for (int u = 0; u < UX; ++u) {
        float* dst = xl + (t * XPA + u * 64) * CS;
        if (!tf) {
          if constexpr (ESZ == 2) *reinterpret_cast<u32x2*>(dst) = vx[u][t];
          else if constexpr (BF) *reinterpret_cast<uint2*>(dst) = pk4_bf16(raw_f32(vx[u][t]));
          else *reinterpret_cast<f32x4*>(dst) = raw_f32(vx[u][t]);
          continue;
        }
        f32x4 w = raw_f32(vx[u][t]);  // lanes outside the image / past the channels hold 0 and stay 0 under every transform
        const int chs = cx[t] ? tch0[t] + j * 4 : 0;
        if (tflags[t] & LMN_SRC_GELU) {
#pragma unroll
          for (int k = 0; k < 4; ++k) w[k] = lmn_gelu(w[k]);
        }
        if (tflags[t] & LMN_SRC_DROP) {
          const int r = (int)(xrc[u] >> 16), c = (int)(xrc[u] & 0xffffu);
          const int gp = (b * A.Hin + oy0 - 1 + r) * A.Win + ox0 - 1 + c;
#pragma unroll
          for (int k = 0; k < 4; ++k) w[k] *= lmn_drop_scale(tseed[t], (uint32_t)(gp * tC[t] + chs + k), tp_[t], tik[t]);
        }
        if (tscale[t]) w *= ld4(tscale[t] + b * tC[t] + chs);
        if constexpr (BF) *reinterpret_cast<uint2*>(dst) = pk4_bf16(w);
        else *reinterpret_cast<f32x4*>(dst) = w;
      }
    }
    }   // (!UP)
#pragma unroll
    for (int m = 0; m < NMT; ++m) {
#pragma unroll
      for (int u = 0; u < UY; ++u) {
        float* dst = yl + (m * NPA + u * 64) * CS;
        if (!(A.dy_flags & LMN_SRC_DROP)) {
          if constexpr (ESZ == 2) *reinterpret_cast<u32x2*>(dst) = vy[u][m];
          else if constexpr (BF) *reinterpret_cast<uint2*>(dst) = pk4_bf16(raw_f32(vy[u][m]));
          else *reinterpret_cast<f32x4*>(dst) = raw_f32(vy[u][m]);
          continue;
        }
        f32x4 w = raw_f32(vy[u][m]);
        const int cos = cy[m] ? (mt0 + m) * 16 + j * 4 : 0;
        const int gp = (b * A.Hout + oy0 + 2 * u + yr0) * A.Wout + ox0 + yc;
#pragma unroll
        for (int k = 0; k < 4; ++k) w[k] *= lmn_drop_scale(A.dy_seed + soff, (uint32_t)(gp * A.Cout + cos + k), A.dy_p, P.inv_keep_dy);
        if constexpr (BF) *reinterpret_cast<uint2*>(dst) = pk4_bf16(w);
        else *reinterpret_cast<f32x4*>(dst) = w;
      }
    }
  };

  const int t_begin = (int)(((int64_t)blockIdx.x * P.total_tiles) / gridDim.x);
  const int t_end = (int)(((int64_t)(blockIdx.x + 1) * P.total_tiles) / gridDim.x);
#ifdef LMN_WG_TIMING
  unsigned long long tk0 = LMN_TCLK(), tk_s1 = 0, tk_s2 = 0, tk_s3 = 0, tk_a, tk_b;
#endif
  if (t_begin < t_end) issue(t_begin);
  const int K0 = TS ? 0 : wv, KSTEP = TS ? 1 : 4;
  for (int tile = t_begin; tile < t_end; ++tile) {
#ifdef LMN_WG_TIMING
    tk_a = LMN_TCLK();
#endif
    __syncthreads();  // previous tile's reads are done
#ifdef LMN_WG_TIMING
    tk_b = LMN_TCLK(); tk_s1 += tk_b - tk_a; tk_a = tk_b;
#endif
    commit(tile);
#ifdef LMN_WG_TIMING
    __builtin_amdgcn_s_waitcnt(0);
    tk_b = LMN_TCLK(); tk_s2 += tk_b - tk_a; tk_a = tk_b;
#endif
    __syncthreads();
    if (tile + 1 < t_end) issue(tile + 1);
#ifdef LMN_WG_TIMING
    tk_b = LMN_TCLK(); tk_s3 += tk_b - tk_a;
#endif
    if constexpr (BF) {
      // K steps of 16 consecutive tile pixels; lane (q, n) owns pixels 4q..4q+3 of the step, channel n
      const uint16_t* xw = reinterpret_cast<const uint16_t*>(XS) + (t_w * XPA + q * 4) * (CS * 2) + n;
      const uint16_t* yw = reinterpret_cast<const uint16_t*>(YS) + (m_w * NPA + q * 4) * (CS * 2) + n;
      auto kfrag = [&](int ks, uint2& av, uint2 (&bv)[9]) __attribute__((always_inline)) {   // operands of the 16-pixel K step ks
        const int pix0 = ks * 16, pr = pix0 >> 5, pc0 = pix0 & 31;
        const uint16_t* xp = xw + (pr * XWC + pc0) * (CS * 2);
        const uint16_t* yp = yw + pix0 * (CS * 2);
        uint32_t h[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          h[jj] = yp[jj * CS * 2];
          bsum += __builtin_bit_cast(float, h[jj] << 16);
        }
        av = uint2{h[0] | (h[1] << 16), h[2] | (h[3] << 16)};
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) {
          const uint16_t* xt = xp + ((tp / 3) * XWC + tp % 3) * (CS * 2);
          const uint32_t h0 = xt[0], h1 = xt[CS * 2], h2 = xt[2 * CS * 2], h3 = xt[3 * CS * 2];
          bv[tp] = uint2{h0 | (h1 << 16), h2 | (h3 << 16)};
        }
      };
      // two 16-pixel K steps per v_mfma_f32_16x16x32_bf16 (mfma_bf16x2: K = 32 pixels); an odd last step in the 16x16x16 form
      int ks = K0;
      for (; (ks + KSTEP) * 16 < NP; ks += 2 * KSTEP) {
        uint2 av0, av1, bv0[9], bv1[9];
        kfrag(ks, av0, bv0);
        kfrag(ks + KSTEP, av1, bv1);
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) acc[tp] = mfma_bf16x2(av0, av1, bv0[tp], bv1[tp], acc[tp]);
      }
      if (ks * 16 < NP) {
        uint2 av, bv[9];
        kfrag(ks, av, bv);
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) acc[tp] = mfma_bf16(av, bv[tp], acc[tp]);
      }
    } else {
      const float* xw = XS + (t_w * XPA + q) * CS + n;
      const float* yw = YS + (m_w * NPA + q) * CS + n;
      for (int ks = K0; ks * 4 < NP; ks += KSTEP) {
        const int pix0 = ks * 4, pr = pix0 >> 5, pc0 = pix0 & 31;
        const float* xp = xw + (pr * XWC + pc0) * CS;
        const float av = yw[pix0 * CS];
        float bv[9];
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) bv[tp] = xp[((tp / 3) * XWC + tp % 3) * CS];
        bsum += av;
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) acc[tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[tp], acc[tp], 0, 0, 0);
      }
    }
  }
#ifdef LMN_WG_TIMING
  {
    const unsigned long long tk_e = LMN_TCLK();
    const int bid = blockIdx.y * gridDim.x + blockIdx.x;
    if (tid == 0 && bid < 4096) {
      g_wg_timing[bid * 6 + 0] = tk_s1;
      g_wg_timing[bid * 6 + 1] = tk_s2;
      g_wg_timing[bid * 6 + 2] = tk_s3;
      g_wg_timing[bid * 6 + 3] = tk_e - tk0 - tk_s1 - tk_s2 - tk_s3;
      g_wg_timing[bid * 6 + 4] = tk0;
      g_wg_timing[bid * 6 + 5] = tk_e;
    }
  }
#endif

  constexpr int NT = 9 * NMT * NNT;
  float* pdst = P.partial ? P.partial + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * (NT * 256 + NMT * 16) : nullptr;
  if constexpr (TS) {
    // every wave holds the finished sums of ITS tile: straight to the block partial (the layout of the LDS-reduced form:
    // [tap][m][t][r][lane], then the bias sums) or to dW
    const int nt = nt0 + t_w;
    int sidx = 0;
    while (sidx + 1 < A.nsrc && nt >= P.ntile_off[sidx + 1]) ++sidx;
    const int ch = (nt - P.ntile_off[sidx]) * 16 + n;
#pragma unroll
    for (int tp = 0; tp < 9; ++tp)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (pdst) {
          pdst[(((tp * NMT + m_w) * NNT + t_w) * 4 + r) * 64 + lane] = acc[tp][r];
        } else {
          const int co = (mt0 + m_w) * 16 + q * 4 + r;
          if ((mt0 + m_w) < P.NMTT && co < A.Cout && nt < P.NNTT && ch < A.src[sidx].C)
            atomicAdd(wgrad_dst(P, co, sidx, ch, 9) + tp, acc[tp][r]);
        }
      }
    if (t_w == 0) {
      float v = bsum;
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (q == 0) {
        const int co = (mt0 + m_w) * 16 + n;
        if (pdst) pdst[NT * 256 + m_w * 16 + n] = v;
        else if (A.db && nset == 0 && co < A.Cout && (mt0 + m_w) < P.NMTT) {
          atomicAdd(A.db + co, v);
          if (A.db2) atomicAdd(A.db2 + co, v);
        }
      }
    }
    return;
  }
  // one-tile blocks: the four waves hold K-split partial sums of the same tile -- reduce in LDS in four wave rounds
  __syncthreads();
  float* s_acc = smem;
  for (int w = 0; w < 4; ++w) {
    if (wv == w) {
#pragma unroll
      for (int tp = 0; tp < 9; ++tp)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float* d = &s_acc[(tp * 4 + r) * 64 + lane];
          *d = (w == 0) ? acc[tp][r] : *d + acc[tp][r];
        }
      float v = bsum;
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (q == 0) {
        float* d = &s_acc[NT * 256 + n];
        *d = (w == 0) ? v : *d + v;
      }
    }
    __syncthreads();
  }
  if (pdst) {
    for (int i = tid; i < NT * 256 + 16; i += 256) pdst[i] = s_acc[i];
    return;
  }
  for (int i = tid; i < NT * 256; i += 256) {
    const int ln = i & 63, r = (i >> 6) & 3, tp = i >> 8;
    const int co = mt0 * 16 + (ln >> 4) * 4 + r;
    if (mt0 >= P.NMTT || co >= A.Cout || nt0 >= P.NNTT) continue;
    int sidx = 0;
    while (sidx + 1 < A.nsrc && nt0 >= P.ntile_off[sidx + 1]) ++sidx;
    const int ch = (nt0 - P.ntile_off[sidx]) * 16 + (ln & 15);
    if (ch >= A.src[sidx].C) continue;
    atomicAdd(wgrad_dst(P, co, sidx, ch, 9) + tp, s_acc[i]);
  }
  if (A.db && nset == 0 && tid < 16) {
    const int co = mt0 * 16 + tid;
    if (co < A.Cout && mt0 < P.NMTT) {
      atomicAdd(A.db + co, s_acc[NT * 256 + tid]);
      if (A.db2) atomicAdd(A.db2 + co, s_acc[NT * 256 + tid]);
    }
  }
}

// ------------------------------------------------------------------------------------ weight gradient, 1x1 direct
// 1x1 / stride 1: both operands of a K step (4 pixels x 16 channels) are 4 x 64 contiguous bytes in NHWC, so
// a lane's MFMA operand IS one dword of global memory -- no LDS staging, no barriers in the main loop.  Every wave
// owns a contiguous range of K steps and keeps U steps (U * (NMT + NNT) loads) in flight.  The pixel index is flat
// over the batch; the image index (needed only for the per-image SE scale) is tracked per batch of steps.
// BF: the U = 8 four-pixel steps of a batch form two 16-pixel K steps; a lane packs its four loaded values (pixels
// 4q..4q+3 of one K16 step, i.e. step 4h + q, pixel j) into one bf16 fragment: 2 instead of 8 MFMAs per tile and batch.
template <int NMT, int NNT, int PM = 0>
__global__ __launch_bounds__(256) void wgrad_1x1_kernel(const WgradParams P) {
  constexpr int PMB = PM & 3;
  constexpr bool RP = (PM & 4) != 0;   // some operand is row-planar (LmnLay): its own instantiation, as in conv_tile_kernel
  constexpr bool BF = PMB >= 1;
  typedef typename ActT<PMB>::type TA;
  constexpr int U = 8;
  const lmn_wgrad_args_t& A = P.a;
  const uint32_t soff = A.seed_ctr ? *A.seed_ctr : 0u;  // device-side dropout stream offset (graph replays: one bump per step)
  const int lane = threadIdx.x & 63;
  const int q = lane >> 4, n = lane & 15;
  const int wvb = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wave = blockIdx.x * 4 + wvb, nwaves = gridDim.x * 4;
  const int mset = blockIdx.y / P.nsets_n, nset = blockIdx.y - mset * P.nsets_n;
  const int mt0 = mset * NMT, nt0 = nset * NNT;
  const int HW = A.Hout * A.Wout;
  const int NPX = A.B * HW;

  const TA* sptr[NNT];
  const float* sscale[NNT];
  int sC[NNT], scs[NNT], sflags[NNT], sch[NNT];
  uint32_t scho[NNT], srf[NNT];   // channel part of the offset, row term (LmnLay)
  uint32_t sseed[NNT];
  float sp[NNT], sik[NNT];
  bool any_tf = (A.dy_flags & LMN_SRC_DROP) != 0;
#pragma unroll
  for (int t = 0; t < NNT; ++t) {
    const int nt = nt0 + t;
    int s = 0;
    while (s + 1 < A.nsrc && nt >= P.ntile_off[s + 1]) ++s;
    const int ch = (nt - P.ntile_off[s]) * 16 + n;
    sptr[t] = (const TA*)A.src[s].ptr;
    sscale[t] = A.src[s].scale;
    sC[t] = A.src[s].C;
    scs[t] = RP ? P.lay_src[s].cs : A.src[s].cstride;
    srf[t] = RP ? (uint32_t)P.lay_src[s].rf : 0u;
    sflags[t] = A.src[s].flags;
    sseed[t] = A.src[s].drop_seed + soff;
    sp[t] = A.src[s].drop_p;
    sik[t] = P.inv_keep_src[s];
    sch[t] = (nt < P.NNTT && ch < A.src[s].C) ? ch : -1;
    { const int chc = sch[t] >= 0 ? sch[t] : 0; scho[t] = (uint32_t)(chc >> 2) * (uint32_t)P.lay_src[s].qs + (uint32_t)(chc & 3); }
    any_tf = any_tf || sflags[t] != 0 || sscale[t] != nullptr;
  }
  int mco[NMT];  // this lane's cout per tile, or -1
#pragma unroll
  for (int m = 0; m < NMT; ++m) mco[m] = ((mt0 + m) < P.NMTT && (mt0 + m) * 16 + n < A.Cout) ? (mt0 + m) * 16 + n : -1;
  uint32_t mcho[NMT];
#pragma unroll
  for (int m = 0; m < NMT; ++m) { const int cc = mco[m] >= 0 ? mco[m] : 0; mcho[m] = (uint32_t)(cc >> 2) * (uint32_t)P.lay_dy.qs + (uint32_t)(cc & 3); }

  f32x4 acc[NMT][NNT];
  float bsum[NMT];
#pragma unroll
  for (int m = 0; m < NMT; ++m) {
    bsum[m] = 0.f;
#pragma unroll
    for (int t = 0; t < NNT; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  const int total_steps = (NPX + 3) >> 2;
  const int sb = (int)(((int64_t)wave * total_steps) / nwaves), se = (int)(((int64_t)(wave + 1) * total_steps) / nwaves);
  // per-(image, channel) source scale (the SE gate): a lane's channel is fixed and a batch spans at most two images, so the
  // two candidate values live in registers and are refetched only when the image changes (loading them per element put
  // a second, dependent memory round trip into every batch: 85 us against 29 us for the untransformed layer at level 0)
  float sc0[NNT], sc1[NNT];
  int sc_b = -1;
#pragma unroll
  for (int t = 0; t < NNT; ++t) sc0[t] = sc1[t] = 1.f;
  for (int step0 = sb; step0 < se; step0 += U) {
    float av[U][NMT], bv[U][NNT];
    // ---- all loads of the batch, straight-line from clamped addresses
#pragma unroll
    for (int u = 0; u < U; ++u) {
      // fp32: lane (q, n) holds pixel q of four-pixel step u; bf16: pixel (u & 3) of lane group q in 16-pixel step u >> 2
      const int px = BF ? step0 * 4 + (u >> 2) * 16 + q * 4 + (u & 3) : (step0 + u) * 4 + q;
      const bool ok = (BF ? px < se * 4 : step0 + u < se) && px < NPX;
      const int ps = ok ? px : 0;
      if constexpr (RP) {
        const uint32_t rowp = lmn_div_row((uint32_t)ps, (uint32_t)P.rpw, P.rp_magic);   // (row term of the row-planar operands)
#pragma unroll
        for (int m = 0; m < NMT; ++m) av[u][m] = ld1((const TA*)A.dy + ((uint32_t)(ps * P.lay_dy.cs) + mcho[m] + rowp * (uint32_t)P.lay_dy.rf));
#pragma unroll
        for (int t = 0; t < NNT; ++t) bv[u][t] = ld1(sptr[t] + ((uint32_t)(ps * scs[t]) + scho[t] + rowp * srf[t]));
      } else {
#pragma unroll
        for (int m = 0; m < NMT; ++m) av[u][m] = ld1((const TA*)A.dy + (uint32_t)(ps * A.dy_cstride + (mco[m] >= 0 ? mco[m] : 0)));
#pragma unroll
        for (int t = 0; t < NNT; ++t) bv[u][t] = ld1(sptr[t] + (uint32_t)(ps * scs[t] + (sch[t] >= 0 ? sch[t] : 0)));
      }
    }
    // ---- on-load transforms (wave-uniform flags) and masking
    int b0 = 0;
    if (any_tf) {
      b0 = (step0 * 4) / HW;  // image of the batch's first pixel (scalar); a batch spans <= 2 images (HW >= 4U)
      if (b0 != sc_b) {       // wave-uniform, a handful of times per wave
        sc_b = b0;
        const int b1 = b0 + 1 < A.B ? b0 + 1 : b0;
#pragma unroll
        for (int t = 0; t < NNT; ++t) {
          if (sscale[t]) {
            const int chs = sch[t] >= 0 ? sch[t] : 0;
            sc0[t] = sscale[t][b0 * sC[t] + chs];
            sc1[t] = sscale[t][b1 * sC[t] + chs];
          }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int px = BF ? step0 * 4 + (u >> 2) * 16 + q * 4 + (u & 3) : (step0 + u) * 4 + q;
      const bool ok = (BF ? px < se * 4 : step0 + u < se) && px < NPX;
      const int ps = ok ? px : 0;
#pragma unroll
      for (int m = 0; m < NMT; ++m) {
        float v = av[u][m];
        if (A.dy_flags & LMN_SRC_DROP) v *= lmn_drop_scale(A.dy_seed + soff, (uint32_t)(ps * A.Cout + (mco[m] >= 0 ? mco[m] : 0)), A.dy_p, P.inv_keep_dy);
        av[u][m] = (ok && mco[m] >= 0) ? v : 0.f;
      }
#pragma unroll
      for (int t = 0; t < NNT; ++t) {
        float v = bv[u][t];
        if (any_tf) {
          const int chs = sch[t] >= 0 ? sch[t] : 0;
          if (sflags[t] & LMN_SRC_GELU) v = lmn_gelu(v);
          if (sflags[t] & LMN_SRC_DROP) v *= lmn_drop_scale(sseed[t], (uint32_t)(ps * sC[t] + chs), sp[t], sik[t]);
          if (sscale[t]) v *= (ps >= (b0 + 1) * HW) ? sc1[t] : sc0[t];
        }
        bv[u][t] = (ok && sch[t] >= 0) ? v : 0.f;
      }
    }
    if constexpr (BF) {
#pragma unroll
      for (int h = 0; h < U / 4; ++h) {
        uint2 af[NMT], bf[NNT];
#pragma unroll
        for (int m = 0; m < NMT; ++m) {
          bsum[m] += (av[4 * h][m] + av[4 * h + 1][m]) + (av[4 * h + 2][m] + av[4 * h + 3][m]);
          af[m] = uint2{pk_bf16(av[4 * h][m], av[4 * h + 1][m]), pk_bf16(av[4 * h + 2][m], av[4 * h + 3][m])};
        }
#pragma unroll
        for (int t = 0; t < NNT; ++t) bf[t] = uint2{pk_bf16(bv[4 * h][t], bv[4 * h + 1][t]), pk_bf16(bv[4 * h + 2][t], bv[4 * h + 3][t])};
#pragma unroll
        for (int t = 0; t < NNT; ++t)
#pragma unroll
          for (int m = 0; m < NMT; ++m) acc[m][t] = mfma_bf16(af[m], bf[t], acc[m][t]);
      }
    } else {
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int m = 0; m < NMT; ++m) bsum[m] += av[u][m];
#pragma unroll
      for (int t = 0; t < NNT; ++t)
#pragma unroll
        for (int m = 0; m < NMT; ++m) acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][m], bv[u][t], acc[m][t], 0, 0, 0);
    }
    }
  }

  // ---- block-level reduction in LDS (4 waves -> 1), plain stores in four wave rounds; layout as wgrad_lds_kernel
  constexpr int NT = NMT * NNT;
  __shared__ float s_acc[NT * 256 + NMT * 16];
  for (int w = 0; w < 4; ++w) {
    if (wvb == w) {
#pragma unroll
      for (int m = 0; m < NMT; ++m)
#pragma unroll
        for (int t = 0; t < NNT; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float* d = &s_acc[((m * NNT + t) * 4 + r) * 64 + lane];
            *d = (w == 0) ? acc[m][t][r] : *d + acc[m][t][r];
          }
#pragma unroll
      for (int m = 0; m < NMT; ++m) {
        float v = bsum[m];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (q == 0) {
          float* d = &s_acc[NT * 256 + m * 16 + n];
          *d = (w == 0) ? v : *d + v;
        }
      }
    }
    __syncthreads();
  }
  const int tid = threadIdx.x;
  if (P.partial) {
    float* dst = P.partial + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * (NT * 256 + NMT * 16);
    for (int i = tid; i < NT * 256 + NMT * 16; i += 256) dst[i] = s_acc[i];
    return;
  }
  for (int i = tid; i < NT * 256; i += 256) {
    const int ln = i & 63, r = (i >> 6) & 3, tl = i >> 8;
    const int t = tl % NNT, m = tl / NNT;
    const int qq = ln >> 4, nn = ln & 15;
    const int co = (mt0 + m) * 16 + qq * 4 + r;
    const int nt = nt0 + t;
    if ((mt0 + m) >= P.NMTT || co >= A.Cout || nt >= P.NNTT) continue;
    int sidx = 0;
    while (sidx + 1 < A.nsrc && nt >= P.ntile_off[sidx + 1]) ++sidx;
    const int ch = (nt - P.ntile_off[sidx]) * 16 + nn;
    if (ch >= A.src[sidx].C) continue;
    atomicAdd(wgrad_dst(P, co, sidx, ch, 1), s_acc[i]);
  }
  if (A.db && nset == 0) {
    for (int i = tid; i < NMT * 16; i += 256) {
      const int co = mt0 * 16 + i;
      if (co < A.Cout && (mt0 + i / 16) < P.NMTT) {
        atomicAdd(A.db + co, s_acc[NT * 256 + i]);
        if (A.db2) atomicAdd(A.db2 + co, s_acc[NT * 256 + i]);
      }
    }
  }
}

// ------------------------------------------------------------------------------------ weight gradient, 1x1, wave-staged
// Same job and same block / K-split geometry as wgrad_1x1_kernel, different data path.  The direct form feeds every MFMA
// operand with ONE dword (or one bf16) per lane straight from global memory: 256 B (128 B) per wave instruction, and the
// kernel runs at the pace of its vector-memory instructions (1.1 TB/s over the step's 1x1 weight gradients).  Here a wave
// moves a CHUNK of 16 pixels x 16 channels per tile with ONE 16-byte (8-byte) load per lane -- 1 KiB per wave instruction,
// whole 64-byte runs per pixel -- applies the on-load transforms once per element, parks the chunk in a WAVE-PRIVATE LDS
// slice ([pixel][16 channels] fp32: conflict-free ds_write_b128 / ds_read_b32) and reads the MFMA operands back in the
// (pixel = k, channel = row / column) form.  No block barrier in the main loop; the loads of chunk c+1 are in flight while
// chunk c is multiplied.
template <int NMT, int NNT, int PM>
__global__ __launch_bounds__(256) void wgrad_1x1w_kernel(const WgradParams P) {
  constexpr int PMB = PM & 3;
  constexpr bool RP = (PM & 4) != 0;   // some operand is row-planar (LmnLay): its own instantiation, as in conv_tile_kernel
  constexpr bool BF = PMB >= 1;
  typedef typename ActT<PMB>::type TA;
  constexpr int NTT = NMT + NNT;              // staged tiles per chunk: dy tiles first, then source tiles
  const lmn_wgrad_args_t& A = P.a;
  const uint32_t soff = A.seed_ctr ? *A.seed_ctr : 0u;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63;
  const int q = lane >> 4, n = lane & 15;
  const int wvb = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* WS = smem + wvb * (NTT * 256);       // this wave's slice: NTT tiles x [16 px][16 ch]
  const int wave = blockIdx.x * 4 + wvb, nwaves = gridDim.x * 4;
  const int mset = blockIdx.y / P.nsets_n, nset = blockIdx.y - mset * P.nsets_n;
  const int mt0 = mset * NMT, nt0 = nset * NNT;
  const int HW = A.Hout * A.Wout;
  const int NPX = A.B * HW;
  // staging role of this lane: pixel lp of the chunk, channel quad lq of every tile
  const int lp = lane >> 2, lq = lane & 3;
  // LMN_SRC_LN (single source): the source is layer-normalised where a chunk is parked -- (mean, rstd) of a pixel from the table the
  // forward conv left (ln_stats, one 8-byte load per lane and chunk, in flight with the chunk), gamma / beta of the block's channel
  // tiles from LDS (behind the staging / reduction area: [NNT][gamma 16 | beta 16])
  constexpr int LNOFF = (4 * NTT * 256 > NMT * NNT * 256 + NMT * 16) ? 4 * NTT * 256 : NMT * NNT * 256 + NMT * 16;
  float* s_gb = smem + LNOFF;
  const bool ln = (A.src[0].flags & LMN_SRC_LN) != 0;   // block-uniform
  if (ln) {
    for (int i = threadIdx.x; i < NNT * 16; i += 256) {
      const int t = i >> 4, c = i & 15, ch = (nt0 + t) * 16 + c;
      const bool cok = (nt0 + t) < P.NNTT && ch < A.src[0].C;
      s_gb[t * 32 + c] = cok ? A.src[0].ln_gamma[ch] : 0.f;
      s_gb[t * 32 + 16 + c] = cok ? A.src[0].ln_beta[ch] : 0.f;
    }
    __syncthreads();
  }
  float2 lnmr = float2{0.f, 1.f};

  const TA* sptr[NNT];
  const float* sscale[NNT];
  int sC[NNT], scs[NNT], sflags[NNT], sch4[NNT];
  uint32_t scho[NNT], srf[NNT];   // channel part of the offset, row term (LmnLay)
  uint32_t sseed[NNT];
  float sp[NNT], sik[NNT];
#pragma unroll
  for (int t = 0; t < NNT; ++t) {
    const int nt = nt0 + t;
    int s = 0;
    while (s + 1 < A.nsrc && nt >= P.ntile_off[s + 1]) ++s;
    const int ch = (nt - P.ntile_off[s]) * 16 + lq * 4;
    sptr[t] = (const TA*)A.src[s].ptr;
    sscale[t] = A.src[s].scale;
    sC[t] = A.src[s].C;
    scs[t] = RP ? P.lay_src[s].cs : A.src[s].cstride;
    srf[t] = RP ? (uint32_t)P.lay_src[s].rf : 0u;
    sflags[t] = A.src[s].flags;
    sseed[t] = A.src[s].drop_seed + soff;
    sp[t] = A.src[s].drop_p;
    sik[t] = P.inv_keep_src[s];
    sch4[t] = (nt < P.NNTT && ch < A.src[s].C) ? ch : -1;   // first channel of this lane's quad, or -1
    scho[t] = (uint32_t)((sch4[t] >= 0 ? sch4[t] : 0) >> 2) * (uint32_t)P.lay_src[s].qs;
  }
  int mco4[NMT];
#pragma unroll
  for (int m = 0; m < NMT; ++m) mco4[m] = ((mt0 + m) < P.NMTT && (mt0 + m) * 16 + lq * 4 < A.Cout) ? (mt0 + m) * 16 + lq * 4 : -1;
  uint32_t mcho[NMT];
#pragma unroll
  for (int m = 0; m < NMT; ++m) mcho[m] = (uint32_t)((mco4[m] >= 0 ? mco4[m] : 0) >> 2) * (uint32_t)P.lay_dy.qs;

  f32x4 acc[NMT][NNT];
  float bsum[NMT];
#pragma unroll
  for (int m = 0; m < NMT; ++m) {
    bsum[m] = 0.f;
#pragma unroll
    for (int t = 0; t < NNT; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // this wave's pixels: whole chunks of 16
  const int total_chunks = (NPX + 15) >> 4;
  const int cb = (int)(((int64_t)wave * total_chunks) / nwaves), ce = (int)(((int64_t)(wave + 1) * total_chunks) / nwaves);
  f32x4 stg[NTT];
  auto chunk_load = [&](int c) {
    const int px = c * 16 + lp;
    const int ps = px < NPX ? px : 0;
    if (ln) lnmr = *reinterpret_cast<const float2*>(A.src[0].ln_stats + 2 * (int64_t)ps);
    if constexpr (RP) {
      const uint32_t rowp = lmn_div_row((uint32_t)ps, (uint32_t)P.rpw, P.rp_magic);   // (row term of the row-planar operands)
#pragma unroll
      for (int m = 0; m < NMT; ++m) stg[m] = ld4((const TA*)A.dy + ((uint32_t)(ps * P.lay_dy.cs) + mcho[m] + rowp * (uint32_t)P.lay_dy.rf));
#pragma unroll
      for (int t = 0; t < NNT; ++t) stg[NMT + t] = ld4(sptr[t] + ((uint32_t)(ps * scs[t]) + scho[t] + rowp * srf[t]));
    } else {
#pragma unroll
      for (int m = 0; m < NMT; ++m) stg[m] = ld4((const TA*)A.dy + (uint32_t)(ps * A.dy_cstride + (mco4[m] >= 0 ? mco4[m] : 0)));
#pragma unroll
      for (int t = 0; t < NNT; ++t) stg[NMT + t] = ld4(sptr[t] + (uint32_t)(ps * scs[t] + (sch4[t] >= 0 ? sch4[t] : 0)));
    }
  };
  auto chunk_put = [&](int c) {   // transforms, masking, LDS
    const int px = c * 16 + lp;
    const bool ok = px < NPX;
    const int ps = ok ? px : 0;
#pragma unroll
    for (int m = 0; m < NMT; ++m) {
      f32x4 v = stg[m];
      if (A.dy_flags & LMN_SRC_DROP) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] *= lmn_drop_scale(A.dy_seed + soff, (uint32_t)(ps * A.Cout + (mco4[m] >= 0 ? mco4[m] : 0) + k), A.dy_p, P.inv_keep_dy);
      }
      if (!(ok && mco4[m] >= 0)) v = f32x4{0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(&WS[m * 256 + lp * 16 + lq * 4]) = v;
    }
#pragma unroll
    for (int t = 0; t < NNT; ++t) {
      f32x4 v = stg[NMT + t];
      const int chs = sch4[t] >= 0 ? sch4[t] : 0;
      if (ln) v = (v - lnmr.x) * lnmr.y * (*reinterpret_cast<const f32x4*>(&s_gb[t * 32 + lq * 4])) + (*reinterpret_cast<const f32x4*>(&s_gb[t * 32 + 16 + lq * 4]));
      if (sflags[t] & LMN_SRC_GELU) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = lmn_gelu(v[k]);
      }
      if (sflags[t] & LMN_SRC_DROP) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] *= lmn_drop_scale(sseed[t], (uint32_t)(ps * sC[t] + chs + k), sp[t], sik[t]);
      }
      if (sscale[t]) v *= ld4(sscale[t] + (ps / HW) * sC[t] + chs);
      if (!(ok && sch4[t] >= 0)) v = f32x4{0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(&WS[(NMT + t) * 256 + lp * 16 + lq * 4]) = v;
    }
  };
  if (cb < ce) chunk_load(cb);
  for (int c = cb; c < ce; ++c) {
    chunk_put(c);                       // (LDS operations of one wave execute in order: the previous chunk's reads are done)
    if (c + 1 < ce) chunk_load(c + 1);  // in flight during the MFMAs below
    if constexpr (BF) {                 // one K16 step: lane (q, n) owns pixels 4q..4q+3, channel n
      uint2 af[NMT], bf_[NNT];
#pragma unroll
      for (int m = 0; m < NMT; ++m) {
        const float a0 = WS[m * 256 + (4 * q) * 16 + n], a1 = WS[m * 256 + (4 * q + 1) * 16 + n];
        const float a2 = WS[m * 256 + (4 * q + 2) * 16 + n], a3 = WS[m * 256 + (4 * q + 3) * 16 + n];
        bsum[m] += (a0 + a1) + (a2 + a3);
        af[m] = uint2{pk_bf16(a0, a1), pk_bf16(a2, a3)};
      }
#pragma unroll
      for (int t = 0; t < NNT; ++t) {
        const float* b = &WS[(NMT + t) * 256 + (4 * q) * 16 + n];
        bf_[t] = uint2{pk_bf16(b[0], b[16]), pk_bf16(b[32], b[48])};
      }
#pragma unroll
      for (int t = 0; t < NNT; ++t)
#pragma unroll
        for (int m = 0; m < NMT; ++m) acc[m][t] = mfma_bf16(af[m], bf_[t], acc[m][t]);
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {     // four K4 steps: lane (q, n) owns pixel 4k + q, channel n
        float av[NMT], bv[NNT];
#pragma unroll
        for (int m = 0; m < NMT; ++m) av[m] = WS[m * 256 + (4 * k + q) * 16 + n];
#pragma unroll
        for (int t = 0; t < NNT; ++t) bv[t] = WS[(NMT + t) * 256 + (4 * k + q) * 16 + n];
#pragma unroll
        for (int m = 0; m < NMT; ++m) bsum[m] += av[m];
#pragma unroll
        for (int t = 0; t < NNT; ++t)
#pragma unroll
          for (int m = 0; m < NMT; ++m) acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], bv[t], acc[m][t], 0, 0, 0);
      }
    }
  }

  // ---- block-level reduction in LDS (4 waves -> 1), plain stores in four wave rounds; layout as wgrad_lds_kernel
  constexpr int NT = NMT * NNT;
  __syncthreads();                      // every wave is done with its staging slice
  float* s_acc = smem;                  // reuse (host sizes the allocation to max(staging, NT*256 + NMT*16) floats)
  for (int w = 0; w < 4; ++w) {
    if (wvb == w) {
#pragma unroll
      for (int m = 0; m < NMT; ++m)
#pragma unroll
        for (int t = 0; t < NNT; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float* d = &s_acc[((m * NNT + t) * 4 + r) * 64 + lane];
            *d = (w == 0) ? acc[m][t][r] : *d + acc[m][t][r];
          }
#pragma unroll
      for (int m = 0; m < NMT; ++m) {
        float v = bsum[m];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (q == 0) {
          float* d = &s_acc[NT * 256 + m * 16 + n];
          *d = (w == 0) ? v : *d + v;
        }
      }
    }
    __syncthreads();
  }
  const int tid = threadIdx.x;
  if (P.partial) {
    float* dst = P.partial + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * (NT * 256 + NMT * 16);
    for (int i = tid; i < NT * 256 + NMT * 16; i += 256) dst[i] = s_acc[i];
    return;
  }
  for (int i = tid; i < NT * 256; i += 256) {
    const int ln = i & 63, r = (i >> 6) & 3, tl = i >> 8;
    const int t = tl % NNT, m = tl / NNT;
    const int qq = ln >> 4, nn = ln & 15;
    const int co = (mt0 + m) * 16 + qq * 4 + r;
    const int nt = nt0 + t;
    if ((mt0 + m) >= P.NMTT || co >= A.Cout || nt >= P.NNTT) continue;
    int sidx = 0;
    while (sidx + 1 < A.nsrc && nt >= P.ntile_off[sidx + 1]) ++sidx;
    const int ch = (nt - P.ntile_off[sidx]) * 16 + nn;
    if (ch >= A.src[sidx].C) continue;
    atomicAdd(wgrad_dst(P, co, sidx, ch, 1), s_acc[i]);
  }
  if (A.db && nset == 0) {
    for (int i = tid; i < NMT * 16; i += 256) {
      const int co = mt0 * 16 + i;
      if (co < A.Cout && (mt0 + i / 16) < P.NMTT) {
        atomicAdd(A.db + co, s_acc[NT * 256 + i]);
        if (A.db2) atomicAdd(A.db2 + co, s_acc[NT * 256 + i]);
      }
    }
  }
}

// second stage: dW[co][ci][tap] += sum over the K-split blocks of their partial tiles (fixed order => deterministic).
// 1024 threads = epq groups of 4 consecutive elements (one 16 B load each) x ksl K-slices, 4 loads in flight per thread: the
// 768 partials of a 3x3 layer are 28 MB, and with 4 B loads the 145 blocks of that launch had 2.4 MB in flight (1 TB/s).
template <int TAPS, int NMT, int NNT>
__global__ __launch_bounds__(1024) void wgrad_reduce_kernel(const WgradParams P, int nblk, int ksl) {
  // ksl = 1 .. 64, a power of two sized to nblk on the host (with 2-3 partials, many slices leave most waves idle)
  const lmn_wgrad_args_t& A = P.a;
  constexpr int NT = TAPS * NMT * NNT, PER = NT * 256 + NMT * 16;  // PER % 4 == 0
  __shared__ __attribute__((aligned(16))) float red[4096];
  const int mset = blockIdx.y / P.nsets_n, nset = blockIdx.y - mset * P.nsets_n;
  const int mt0 = mset * NMT, nt0 = nset * NNT;
  const float* src = P.partial + (int64_t)blockIdx.y * nblk * PER;
  const int epq = 1024 / ksl, epb = epq * 4;
  const int q = threadIdx.x & (epq - 1), ks = threadIdx.x / epq;
  const int i4 = (blockIdx.x * epq + q) * 4;
  f32x4 s4 = f32x4{0.f, 0.f, 0.f, 0.f};
  if (i4 < PER) {
    const float* sp = src + i4;
#pragma unroll 4
    for (int k = ks; k < nblk; k += ksl) s4 += *reinterpret_cast<const f32x4*>(sp + (int64_t)k * PER);
  }
  *reinterpret_cast<f32x4*>(&red[threadIdx.x * 4]) = s4;
  __syncthreads();
  for (int e = threadIdx.x; e < epb; e += 1024) {  // (epb > 1024 for ksl < 4)
  const int i = blockIdx.x * epb + e;
  if (i >= PER) return;
  float sum = 0.f;
  for (int k = 0; k < ksl; ++k) sum += red[k * epb + e];
  if (i < NT * 256) {
    const int ln = i & 63, r = (i >> 6) & 3, tile = i >> 8;
    const int t = tile % NNT, m = (tile / NNT) % NMT, tp = tile / (NNT * NMT);
    const int qq = ln >> 4, nn = ln & 15;
    const int co = (mt0 + m) * 16 + qq * 4 + r;
    const int nt = nt0 + t;
    if ((mt0 + m) >= P.NMTT || co >= A.Cout || nt >= P.NNTT) continue;
    int sidx = 0;
    while (sidx + 1 < A.nsrc && nt >= P.ntile_off[sidx + 1]) ++sidx;
    const int ch = (nt - P.ntile_off[sidx]) * 16 + nn;
    if (ch >= A.src[sidx].C) continue;
    wgrad_dst(P, co, sidx, ch, TAPS)[tp] += sum;
  } else if (A.db && nset == 0) {
    const int j = i - NT * 256, co = mt0 * 16 + j;
    if (co < A.Cout && (mt0 + j / 16) < P.NMTT) {
      A.db[co] += sum;
      if (A.db2) A.db2[co] += sum;
    }
  }
  }
}

// All deferred reductions of a pass (lmn_wgrad_args_t.defer_reduce) in ONE launch: block b finds its job by bisection over the
// jobs' first-block table, then sums one slice of one tile set exactly as wgrad_reduce_kernel does (same fixed order =>
// deterministic).  82 reduction launches of 8-11 us each per training step become one or a few.
__global__ __launch_bounds__(1024) void wgrad_reduce_batch_kernel(const lmn_reduce_job_t* __restrict__ jobs, int njobs) {
  __shared__ __attribute__((aligned(16))) float red[4096];
  int lo = 0, hi = njobs - 1;
  const int64_t b = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].first_block <= b) lo = mid; else hi = mid - 1;
  }
  const lmn_reduce_job_t& J = jobs[lo];
  const int rel = (int)(b - J.first_block);
  const int set = rel / J.blocks_per_set, bx = rel - set * J.blocks_per_set;
  const int NMT = J.NMT, NNT = J.NNT, TAPS = J.taps, PER = J.per, nblk = J.nblk, ksl = J.ksl;
  const int NT = TAPS * NMT * NNT;
  const int mset = set / J.nsets_n, nset = set - mset * J.nsets_n;
  const int mt0 = mset * NMT, nt0 = nset * NNT;
  const float* src = J.partial + (int64_t)set * nblk * PER;
  const int epq = 1024 / ksl, epb = epq * 4;
  const int q = threadIdx.x & (epq - 1), ks = threadIdx.x / epq;
  const int i4 = (bx * epq + q) * 4;
  f32x4 s4 = f32x4{0.f, 0.f, 0.f, 0.f};
  if (i4 < PER) {
    const float* sp = src + i4;
#pragma unroll 4
    for (int k = ks; k < nblk; k += ksl) s4 += *reinterpret_cast<const f32x4*>(sp + (int64_t)k * PER);
  }
  *reinterpret_cast<f32x4*>(&red[threadIdx.x * 4]) = s4;
  __syncthreads();
  for (int e = threadIdx.x; e < epb; e += 1024) {  // (epb > 1024 for ksl < 4)
    const int i = bx * epb + e;
    if (i >= PER) return;
    float sum = 0.f;
    for (int k = 0; k < ksl; ++k) sum += red[k * epb + e];
    if (i < NT * 256) {
      const int ln = i & 63, r = (i >> 6) & 3, tile = i >> 8;
      const int t = tile % NNT, m = (tile / NNT) % NMT, tp = tile / (NNT * NMT);
      const int qq = ln >> 4, nn = ln & 15;
      const int co = (mt0 + m) * 16 + qq * 4 + r;
      const int nt = nt0 + t;
      if ((mt0 + m) >= J.NMTT || co >= J.Cout || nt >= J.NNTT) continue;
      int sidx = 0;
      while (sidx + 1 < J.nsrc && nt >= J.ntile_off[sidx + 1]) ++sidx;
      const int ch = (nt - J.ntile_off[sidx]) * 16 + nn;
      if (ch >= J.srcC[sidx]) continue;
      float* own = J.dW_src[sidx];
      float* d = own ? own + ((int64_t)co * J.srcC[sidx] + ch) * TAPS : J.dW + ((int64_t)co * J.Cin + J.cbase[sidx] + ch) * TAPS;
      d[tp] += sum;
    } else if (J.db && nset == 0) {
      const int j = i - NT * 256, co = mt0 * 16 + j;
      if (co < J.Cout && (mt0 + j / 16) < J.NMTT) {
        J.db[co] += sum;
        if (J.db2) J.db2[co] += sum;
      }
    }
  }
}

}  // namespace

// (a function template of its own: only there is the `if constexpr` branch of the tile shapes wgrad3_kernel does not have discarded
// without being instantiated)
template <int M, int N, int PM>
static void wgrad3_launch(dim3 grid, size_t shmem, hipStream_t st, const WgradParams& P) {
  if constexpr (M * N == 4 || M * N == 1) {
    if (P.a.src[0].flags & LMN_SRC_UP2) LMN_LAUNCH((wgrad3_kernel<M, N, PM, true>), grid, dim3(256), shmem, st, P);
    else LMN_LAUNCH((wgrad3_kernel<M, N, PM>), grid, dim3(256), shmem, st, P);
  }
}

extern "C" {

int lmn_sizeof_wgrad_args(void) { return (int)sizeof(lmn_wgrad_args_t); }

// cout x cin 16-channel tiles per block.  The direct 1x1 kernel holds up to 4 accumulator tiles in any shape (a
// one-tile-wide side is read exactly once); the LDS-staged kernel is instantiated for 1 or 2 tiles per side.
static void wgrad_tile_shape(const lmn_wgrad_args_t& a, int nmtt, int nntt, int* NMT, int* NNT) {
  bool rp_any = a.dy_rp_w != 0;
  for (int s = 0; s < a.nsrc && s < 3; ++s) rp_any = rp_any || a.src[s].rp_w != 0 || (a.src[s].flags & LMN_SRC_LN) != 0;   // (LayerNorm on load: the wave-staged kernel only)
  const bool direct = a.ksize == 1 && a.stride == 1 && ((int64_t)a.Hout * a.Wout >= 32 || rp_any);   // (row-planar operands: direct kernels only)
  if (direct) {
    // 1x1: a block of (M x N) tiles reads its M dy tiles and N source tiles once per pixel; the grid's tile sets re-read
    // dy ceil(nntt/N) times and the sources ceil(nmtt/M) times: take the instantiated shape with the fewest bytes per
    // pixel (ties: fewer accumulator tiles)
    static const int shapes[][2] = {{1, 1}, {1, 2}, {1, 3}, {1, 4}, {2, 1}, {3, 1}, {4, 1}, {2, 2}, {2, 3}, {3, 2}, {2, 4}, {4, 2}};
    int cin = 0;
    for (int s = 0; s < a.nsrc && s < 3; ++s) cin += a.src[s].C;
    long best = -1;
    for (const auto& sh : shapes) {
      const int M = sh[0], N = sh[1];
      if (M > nmtt && M > 1) continue;
      if (N > nntt && N > 1) continue;
      const long cost = ((long)((nmtt + M - 1) / M) * cin + (long)((nntt + N - 1) / N) * a.Cout) * 64 + M * N;
      if (best < 0 || cost < best) { best = cost; *NMT = M; *NNT = N; }
    }
    return;
  }
  const bool small = nmtt == 1 || nntt == 1;
  *NMT = small ? 1 : 2; *NNT = small ? 1 : 2;  // LDS-staged kernel: (1,2)/(2,1) measured slower here
}

// 3x3 stride-1 layers on maps >= 32 wide run wgrad3_kernel: <= 154 VGPRs and <= 48 KB of LDS per block, so THREE blocks are
// resident per CU and the K-split aims at 768 blocks instead of 512.  LMN_WGRAD_V1=0 selects the
// general kernel (A/B runs), LMN_WGRAD_CAP overrides the block count.
static int wgrad_v1_env() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("LMN_WGRAD_V1"); v = e ? atoi(e) : 1; }
  return v;
}
static bool wgrad_v1(const lmn_wgrad_args_t& a) { return a.ksize == 3 && a.stride == 1 && wgrad_v1_env() != 0; }
static int wgrad_blocks_total(const lmn_wgrad_args_t& a) {
  static int c = -1;
  if (c < 0) { const char* e = getenv("LMN_WGRAD_CAP"); c = e ? atoi(e) : 0; }
  if (!wgrad_v1(a)) return 512;
  // 512 = two resident blocks per CU.  Alone the kernel is fastest at 768 (three per CU), but it runs on the weight-gradient stream
  // beside the backward chain: inside the step 512 blocks measured 14.59 ms against 14.68 (768), 14.63 (384), 14.65 (640), 14.71 (256)
  // -- fewer partials to write and re-read (28 -> 19 MB per 24-channel layer) and more CU time left to the chain (round 4, three
  // runs each on one box).  (bf16 storage would fit 4 per CU: 1024 blocks measured 3 % slower.)
  return c > 0 ? c : 512;
}

int64_t lmn_conv_wgrad_workspace(const lmn_wgrad_args_t* a) {
  if (!a) return 0;
  int nntt = 0;
  for (int s = 0; s < a->nsrc && s < 3; ++s) nntt += (a->src[s].C + 15) / 16;
  const int nmtt = (a->Cout + 15) / 16;
  int NMT, NNT;
  wgrad_tile_shape(*a, nmtt, nntt, &NMT, &NNT);
  const int gy = ((nmtt + NMT - 1) / NMT) * ((nntt + NNT - 1) / NNT);
  const int64_t per = (int64_t)a->ksize * a->ksize * NMT * NNT * 256 + NMT * 16;
  const int btot = wgrad_blocks_total(*a);
  const int64_t cap = btot / gy > 2 ? btot / gy : 2;  // upper bound of the K-split block count (see lmn_conv_wgrad)
  const int64_t need = gy * cap * per;
  return need <= (int64_t)(16 << 20) ? need : 0;  // at most 64 MB of partials; larger problems use atomics
}
#ifdef LMN_WG_TIMING
int lmn_wg_timing(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wg_timing), sizeof(unsigned long long) * n);
}
#endif

static int reduce_slices(int nblk) {  // k-slices of the reduction kernel: ~4 partials per slice, at most 64
  int k = 1;
  while (k < 64 && k * 4 < nblk) k <<= 1;
  return k;
}

// K-split blocks either add their LDS-reduced tile straight into dW with atomics or write it to the workspace for a
// second (reduction) launch.  Scattered float atomics retire at ~40 per nanosecond chip-wide (measured: 4.6 M of
// them made a 114 us tail on the 192->96 3x3 layer), the reduction launch costs ~6-10 us: atomics only for small totals.
static bool wgrad_two_stage(int64_t gy, int64_t blocks, int64_t per) {
  return blocks > 48 || gy * blocks * per > 256 * 1024;
}

// Geometry of one weight-gradient call: tile shape, K-split block count, reduction form.  ONE source of truth for the launch
// (lmn_conv_wgrad) and for the description of its deferred reduction (lmn_conv_wgrad_job).
struct WgGeom {
  int NMT, NNT, gy, taps, blocks;   // cout x cin tiles per block, tile sets (grid.y), K-split blocks (grid.x)
  int64_t per;                      // floats of one block partial
  bool direct, v1, wave_staged;
  size_t shmem;
  int pm;
};

static int wgrad_setup(const lmn_wgrad_args_t& A, WgradParams& P, WgGeom& G) {
  LMN_REQUIRE(A.ksize == 1 || A.ksize == 3, "conv_wgrad: ksize %d", A.ksize);
  LMN_REQUIRE(A.stride == 1 || A.stride == 2, "conv_wgrad: stride %d", A.stride);
  LMN_REQUIRE(A.nsrc >= 1 && A.nsrc <= 3, "conv_wgrad: nsrc %d", A.nsrc);
  LMN_REQUIRE(A.dy && A.Cout > 0 && A.dy_cstride >= A.Cout, "conv_wgrad: dy/Cout");
  for (int s = 0; s < A.nsrc && s < 3; ++s) LMN_REQUIRE(A.dW || A.dW_src[s], "conv_wgrad: no gradient tensor for source %d", s);
  LMN_REQUIRE(!A.db2 || A.db, "conv_wgrad: db2 without db");
  LMN_REQUIRE(A.B > 0 && A.Hout > 0 && A.Wout > 0, "conv_wgrad: empty tensor");
  LMN_REQUIRE(A.mma_dtype == LMN_F32 || A.mma_dtype == LMN_BF16, "conv_wgrad: mma_dtype %d", A.mma_dtype);
  LMN_REQUIRE(A.act_dtype == LMN_F32 || (A.act_dtype == LMN_BF16 && A.mma_dtype == LMN_BF16),
              "conv_wgrad: act_dtype %d with mma_dtype %d", A.act_dtype, A.mma_dtype);
  P.a = A;
  P.NNTT = 0;
  P.Cin = 0;
  for (int s = 0; s < 3; ++s) {
    P.ntile_src[s] = P.ntile_off[s] = P.cbase[s] = 0;
    P.inv_keep_src[s] = 1.f;
  }
  for (int s = 0; s < A.nsrc; ++s) {
    LMN_REQUIRE(A.src[s].ptr && A.src[s].C > 0 && A.src[s].cstride >= A.src[s].C, "conv_wgrad: source %d", s);
    P.ntile_src[s] = (A.src[s].C + 15) / 16;
    P.ntile_off[s] = P.NNTT;
    P.cbase[s] = P.Cin;
    P.NNTT += P.ntile_src[s];
    P.Cin += A.src[s].C;
    if (A.src[s].flags & LMN_SRC_DROP) P.inv_keep_src[s] = 1.f / (1.f - A.src[s].drop_p);
  }
  P.inv_keep_dy = (A.dy_flags & LMN_SRC_DROP) ? 1.f / (1.f - A.dy_p) : 1.f;
  {  // row-planar operands (RP4): whole tensors of a 1x1 stride-1 call, one image width for all of them
    int rw = 0;
    for (int s = 0; s < 3; ++s) {
      const int w = s < A.nsrc ? A.src[s].rp_w : 0;
      P.lay_src[s] = lmn_lay_make(w, s < A.nsrc ? A.src[s].C : 4, s < A.nsrc ? A.src[s].cstride : 4);
      if (w) {
        LMN_REQUIRE(w > 0 && A.src[s].cstride == A.src[s].C && (rw == 0 || rw == w), "conv_wgrad: row-planar source %d must be a whole tensor (cstride == C) of the call's width", s);
        rw = w;
      }
    }
    P.lay_dy = lmn_lay_make(A.dy_rp_w, A.Cout, A.dy_cstride);
    if (A.dy_rp_w) { LMN_REQUIRE(A.dy_rp_w > 0 && A.dy_cstride == A.Cout && (rw == 0 || rw == A.dy_rp_w), "conv_wgrad: row-planar dy must be a whole tensor of the call's width"); rw = A.dy_rp_w; }
    if (rw) LMN_REQUIRE(A.ksize == 1 && A.stride == 1 && rw == A.Wout && rw == A.Win, "conv_wgrad: row-planar operands belong to 1x1 stride-1 calls over images of their own width (rp_w %d, call %d)", rw, A.Wout);
    P.rpw = rw;
    P.rp_magic = lmn_div_magic(rw);
  }
  P.NMTT = (A.Cout + 15) / 16;
  P.steps_per_img = (A.Hout * A.Wout + 3) / 4;
  LMN_REQUIRE((int64_t)A.B * P.steps_per_img < (1LL << 31), "conv_wgrad: too many pixels");
  for (int s = 0; s < A.nsrc; ++s)
    LMN_REQUIRE((int64_t)A.B * A.Hin * A.Win * A.src[s].cstride < (1LL << 31), "conv_wgrad: source %d larger than 2^31 elements", s);
  LMN_REQUIRE((int64_t)A.B * A.Hout * A.Wout * A.dy_cstride < (1LL << 31), "conv_wgrad: dy larger than 2^31 elements");
  P.total_steps = A.B * P.steps_per_img;
  if (A.ksize == 1 && A.stride == 1) {  // 1x1: every image is one flat row of H*W pixels (all tiles full)
    P.a.Wout = A.Hout * A.Wout; P.a.Hout = 1;
    P.a.Win = A.Hin * A.Win; P.a.Hin = 1;
  }
  int NMT, NNT;  // cout x cin tiles per block
  wgrad_tile_shape(A, P.NMTT, P.NNTT, &NMT, &NNT);
  const int msets = (P.NMTT + NMT - 1) / NMT;
  P.nsets_n = (P.NNTT + NNT - 1) / NNT;
  const int gy = msets * P.nsets_n;
  const int taps = A.ksize * A.ksize;
  const int64_t per = (int64_t)taps * NMT * NNT * 256 + NMT * 16;
  // ---- tile geometry of the LDS-staged kernel
  const lmn_wgrad_args_t& Gm = P.a;  // (flattened) geometry
  // tile pixels: 256 for the 1x1 form, for one-tile blocks and on the small maps (fewer barriers and less halo per pixel:
  // -5..10 %), 128 where a 2x2-tile block would need 76 KB of LDS for it (+3 % there)
  const bool v1c = wgrad_v1(A) && Gm.Wout >= 32 && (NMT * NNT == 4 || NMT * NNT == 1);  // V1 candidate (item counts checked below)
  const int npmax = Gm.stride == 2 ? 64 : ((Gm.ksize == 1 || NMT * NNT == 1 || (Gm.Wout <= 32 && !v1c)) ? 256 : 128);
  P.TW = Gm.Wout < (Gm.ksize == 1 ? npmax : 32) ? Gm.Wout : (Gm.ksize == 1 ? npmax : 32);
  if (Gm.stride == 2 && P.TW > 16) P.TW = 16;
  P.TH = npmax / P.TW;
  if (P.TH > Gm.Hout) P.TH = Gm.Hout;
  if (P.TH < 1) P.TH = 1;
  P.XH = (P.TH - 1) * Gm.stride + Gm.ksize;
  P.XW = (P.TW - 1) * Gm.stride + Gm.ksize;
  // LDS planes [tile][pixel][CS]: the 4 pixels of a K step are CS*stride floats apart; 16 (stride 1) and 24
  // (stride 2: 48 = 16 mod 32 banks) keep the two 16-lane halves of a ds_read_b32 group on disjoint banks
  P.CSy = 16;
  P.CSx = A.stride == 1 ? 16 : 24;
  const bool bf = A.mma_dtype == LMN_BF16;
  const int pm = bf ? (A.act_dtype == LMN_BF16 ? 2 : 1) : 0;
  // bf16 planes: 16 bf16 = 8 dwords per pixel, +4: the lane groups q (pixels 4 apart) read disjoint bank ranges
  if (bf) P.CSy = P.CSx = 12;
  P.tiles_x = (Gm.Wout + P.TW - 1) / P.TW;
  P.tiles_y = (Gm.Hout + P.TH - 1) / P.TH;
  P.total_tiles = A.B * P.tiles_x * P.tiles_y;
  P.dbg = 0;
  LMN_REQUIRE(P.XH * P.XW < 65536 && P.TH * P.TW < 65536, "conv_wgrad: tile too large");
  P.mXW = (uint32_t)((1ull << 32) / (uint32_t)P.XW + 1);
  P.mTW = (uint32_t)((1ull << 32) / (uint32_t)P.TW + 1);
  int64_t lds_floats = (int64_t)NNT * P.XH * P.XW * P.CSx + (int64_t)NMT * P.TH * P.TW * P.CSy;
  if (lds_floats < per) lds_floats = per;
  LMN_REQUIRE(lds_floats * 4 <= 160 * 1024, "conv_wgrad: LDS tile too large (%lld B)", (long long)lds_floats * 4);
  // K-split: enough blocks to fill the chip (~1024 in total), each walking a contiguous range of tiles
  int64_t blocks64 = P.total_tiles;
  // wgrad3_kernel: the whole tile's items in the prefetch registers (6 + 4 per thread for one-tile, 4 + 2 for 2 x 2-tile
  // blocks), byte offsets in 32 bits (raw buffer descriptors)
  const int64_t esz = A.act_dtype == LMN_BF16 ? 2 : 4;
  bool v1 = v1c && P.TW == 32 && P.XW == 34 &&
            P.XH * P.XW * 4 <= (NMT * NNT == 1 ? 6 : 4) * 256 && P.TH * P.TW * 4 <= (NMT * NNT == 1 ? 4 : 2) * 256 &&
            (int64_t)A.B * A.Hout * A.Wout * A.dy_cstride * esz < 0xfffffff0LL;
  for (int s = 0; s < A.nsrc; ++s) v1 = v1 && (int64_t)A.B * A.Hin * A.Win * A.src[s].cstride * esz < 0xfffffff0LL;
  bool up2 = false;
  for (int s = 0; s < A.nsrc; ++s) up2 = up2 || (A.src[s].flags & LMN_SRC_UP2) != 0;
  if (v1) {  // padded planes: [tile][UX * 64 pixels][CS dwords]
    const int ux = NMT * NNT == 1 ? 6 : 4, uy = NMT * NNT == 1 ? 4 : 2, cs = bf ? 12 : 16;
    lds_floats = (int64_t)(NNT * ux + NMT * uy) * 64 * cs;
    if (up2) lds_floats += (int64_t)NNT * (NMT * NNT == 1 ? 3 : 2) * 64 * 16;   // the fp32 source window (LMN_SRC_UP2)
    if (NMT * NNT == 1 && lds_floats < per) lds_floats = per;
  }
  if (up2) {
    LMN_REQUIRE(A.nsrc == 1 && A.ksize == 3 && A.stride == 1 && A.Hin % 2 == 0 && A.Win % 2 == 0 && !(A.src[0].flags & ~LMN_SRC_UP2) && !A.src[0].scale,
                "conv_wgrad: LMN_SRC_UP2 belongs to single-source 3x3 stride-1 calls (Hin x Win = the upsampled size) without other source transforms");
    LMN_REQUIRE(v1 && (P.XH + 1) / 2 + 2 <= (NMT * NNT == 1 ? 3 : 2) * 64 / 19,
                "conv_wgrad: LMN_SRC_UP2 needs the wgrad3 form (upsampled map >= 32 wide; lmn_conv_wgrad_up2_ok says so beforehand)");
  }
  static int cap_other = -1;   // LMN_WGRAD_CAP_OTHER: K-split blocks of the other weight-gradient kernels (1x1, LDS-staged 3x3), A/B runs
  if (cap_other < 0) { const char* e = getenv("LMN_WGRAD_CAP_OTHER"); cap_other = e && atoi(e) > 0 ? atoi(e) : 512; }
  const int btot = v1 ? wgrad_blocks_total(A) : cap_other;
  const int64_t cap = btot / gy > 2 ? btot / gy : 2;
  if (blocks64 > cap) blocks64 = cap;
  if (blocks64 < 1) blocks64 = 1;
  bool rp_any = A.dy_rp_w != 0, ln_any = false;
  for (int s = 0; s < A.nsrc; ++s) { rp_any = rp_any || A.src[s].rp_w != 0; ln_any = ln_any || (A.src[s].flags & LMN_SRC_LN) != 0; }
  G.direct = A.ksize == 1 && A.stride == 1 && ((int64_t)Gm.Hout * Gm.Wout >= 32 || rp_any || ln_any);
  if (G.direct) {  // direct (no LDS) kernels: K steps split over ~1024*4/gy waves, at least 16 steps per wave
    const int64_t steps = ((int64_t)A.B * Gm.Hout * Gm.Wout + 3) / 4;
    blocks64 = (steps + 63) / 64;
    if (blocks64 > cap) blocks64 = cap;
    if (blocks64 < 1) blocks64 = 1;
  }
  // two-stage reduction when the caller's workspace holds every block partial; else LDS-reduced atomics with fewer blocks
  P.partial = nullptr;
  if (A.workspace && (wgrad_two_stage(gy, blocks64, per) || (g_lmn_det && blocks64 > 1)) && (int64_t)gy * blocks64 * per <= A.workspace_floats) {
    P.partial = A.workspace;
  } else if (g_lmn_det) {
    blocks64 = 1;         // deterministic mode without room for the partials: no K split (one block per tile set adds its sums alone)
  } else if (blocks64 > 512 / gy && 512 / gy >= 2) {
    blocks64 = 512 / gy;  // atomics: fewer blocks
  }
  G.NMT = NMT; G.NNT = NNT; G.gy = gy; G.taps = taps; G.per = per; G.blocks = (int)blocks64;
  G.v1 = v1; G.pm = pm; G.shmem = (size_t)lds_floats * 4;
  // 1x1 data path: wave-staged chunks (wgrad_1x1w_kernel) except for bf16-stored operands without on-load transforms in
  // one of the direct kernel's shapes, where one 2-byte load per lane and MFMA operand is faster (measured, level 0:
  // 28 vs 40 us; with transforms 70 vs 63 us; fp32 storage: wave-staged -19 % over the 13 probe layers)
  bool any_tf = (A.dy_flags & LMN_SRC_DROP) != 0;
  for (int s = 0; s < A.nsrc; ++s) any_tf = any_tf || A.src[s].flags != 0 || A.src[s].scale != nullptr;
  const bool old_shape = (NMT == 1 && NNT <= 4) || (NNT == 1 && NMT <= 4) || (NMT == 2 && NNT == 2);
  G.wave_staged = !(pm == 2 && !any_tf && old_shape) || (rp_any && (int64_t)Gm.Hout * Gm.Wout < 32);   // (tiny maps: the staged kernel divides per pixel)
  for (int s = 0; s < A.nsrc; ++s) {
    if (A.src[s].flags & LMN_SRC_LN) {   // LayerNorm on load: the wave-staged 1x1 kernel, statistics from the forward conv's table
      LMN_REQUIRE(A.nsrc == 1 && G.direct && G.wave_staged && !rp_any, "conv_wgrad: LMN_SRC_LN belongs to single-source NHWC 1x1 stride-1 calls");
      LMN_REQUIRE(A.src[s].ln_gamma && A.src[s].ln_beta && A.src[s].ln_stats, "conv_wgrad: LMN_SRC_LN needs ln_gamma / ln_beta and the ln_stats table of the forward conv");
      LMN_REQUIRE(!(A.src[s].flags & ~LMN_SRC_LN) && !A.src[s].scale, "conv_wgrad: LMN_SRC_LN does not combine with other source transforms");
    }
  }
  return 0;
}

int lmn_sizeof_reduce_job(void) { return (int)sizeof(lmn_reduce_job_t); }

// 1 when lmn_conv_wgrad accepts these arguments WITH LMN_SRC_UP2 on the source (the wgrad3 form samples the bilinear x2 where it
// stages its window), 0 when the caller has to materialise the upsampled tensor first.  Host arithmetic only: the geometry of the
// call without the flag (wgrad_setup: ONE predicate for the launch and for this query), then the two conditions of the fused form.
int lmn_conv_wgrad_up2_ok(const lmn_wgrad_args_t* args) {
  if (!args || args->nsrc != 1 || args->ksize != 3 || args->stride != 1 || (args->Hin & 1) || (args->Win & 1)) return 0;
  if ((args->src[0].flags & ~LMN_SRC_UP2) || args->src[0].scale) return 0;
  lmn_wgrad_args_t a = *args;
  a.src[0].flags = 0;
  // (the plain call describes the UPSAMPLED tensor: the 32-bit offset checks of wgrad_setup then bound the quarter-size source too)
  WgradParams P;
  WgGeom G;
  char keep[sizeof(g_lmn_err)];
  memcpy(keep, g_lmn_err, sizeof(keep));
  const int rc = wgrad_setup(a, P, G);
  memcpy(g_lmn_err, keep, sizeof(keep));   // a query leaves no error text behind
  if (rc != 0 || !G.v1) return 0;
  return (P.XH + 1) / 2 + 2 <= (G.NMT * G.NNT == 1 ? 3 : 2) * 64 / 19 ? 1 : 0;
}

// Description of the reduction a deferred weight-gradient call (args->defer_reduce) leaves behind: `out->nblk == 0` when the call
// reduces on its own (atomics: small totals, or no workspace).  Pure host arithmetic, nothing is launched.
int lmn_conv_wgrad_job(const lmn_wgrad_args_t* args, lmn_reduce_job_t* out) {
  LMN_REQUIRE(args && out, "conv_wgrad_job: null pointer");
  WgradParams P;
  WgGeom G;
  const int rc = wgrad_setup(*args, P, G);
  if (rc) return rc;
  memset(out, 0, sizeof(*out));
  if (!P.partial) return 0;
  const lmn_wgrad_args_t& A = *args;
  out->partial = P.partial;
  out->nblk = G.blocks; out->per = (int32_t)G.per; out->gy = G.gy; out->nsets_n = P.nsets_n;
  out->taps = G.taps; out->NMT = G.NMT; out->NNT = G.NNT; out->nsrc = A.nsrc;
  out->Cout = A.Cout; out->Cin = P.Cin; out->NMTT = P.NMTT; out->NNTT = P.NNTT;
  for (int s = 0; s < 3; ++s) {
    out->srcC[s] = s < A.nsrc ? A.src[s].C : 0;
    out->ntile_off[s] = P.ntile_off[s];
    out->cbase[s] = P.cbase[s];
    out->dW_src[s] = A.dW_src[s];
  }
  out->dW = A.dW; out->db = A.db; out->db2 = A.db2;
  out->ksl = reduce_slices(G.blocks);
  out->blocks_per_set = (int32_t)((G.per + 4096 / out->ksl - 1) / (4096 / out->ksl));
  return 0;
}
int lmn_wgrad_reduce_batch(const lmn_reduce_job_t* jobs_dev, int njobs, int64_t total_blocks, lmn_stream_t stream) {
  LMN_REC(lmn_wgrad_reduce_batch(jobs_dev, njobs, total_blocks, stream));
  LMN_REQUIRE(jobs_dev && njobs > 0 && total_blocks > 0 && total_blocks < (1LL << 31), "wgrad_reduce_batch: bad job table");
  LMN_LAUNCH(wgrad_reduce_batch_kernel, dim3((unsigned)total_blocks), dim3(1024), 0, (hipStream_t)stream, jobs_dev, njobs);
  return lmn_launch_status("wgrad_reduce_batch");
}

int lmn_conv_wgrad(const lmn_wgrad_args_t* args, lmn_stream_t stream) {
  if (args && g_lmn_rec) {
    const lmn_wgrad_args_t copy = *args;
    lmn_rec_push([copy, stream]() -> int { return lmn_conv_wgrad(&copy, stream); }, "lmn_conv_wgrad(");
  }
  LMN_REQUIRE(args, "conv_wgrad: null args");
  const lmn_wgrad_args_t& A = *args;
  WgradParams P;
  WgGeom G;
  {
    const int rc = wgrad_setup(A, P, G);
    if (rc) return rc;
  }
  if (g_lmn_prof_on) {  // algorithmic cost: MACs of the layer shape; every source and dy read once
    const double opix = (double)A.B * A.Hout * A.Wout, ipix = (double)A.B * A.Hin * A.Win;
    lmn_prof_cost(2.0 * opix * (double)P.Cin * A.Cout * A.ksize * A.ksize, (A.act_dtype == LMN_BF16 ? 2.0 : 4.0) * (ipix * P.Cin + opix * A.Cout));
  }
  const int NMT = G.NMT, NNT = G.NNT, gy = G.gy, pm = G.pm, blocks = G.blocks;
  const int64_t per = G.per;
  const bool v1 = G.v1;
  const bool reduce_now = P.partial && !A.defer_reduce;   // deferred: the caller batches the reductions (lmn_wgrad_reduce_batch)
  const size_t shmem = G.shmem;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(blocks, gy);
#define LMN_WG(T, M, N)                                                                                             \
  do {                                                                                                              \
    if (shmem > 64 * 1024) {                                                                                        \
      (void)hipFuncSetAttribute((const void*)wgrad_lds_kernel<T, M, N, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem); \
      (void)hipFuncSetAttribute((const void*)wgrad_lds_kernel<T, M, N, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem); \
      (void)hipFuncSetAttribute((const void*)wgrad_lds_kernel<T, M, N, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem); \
    }                                                                                                               \
    if constexpr (T == 9 && (M * N == 4 || M * N == 1)) {                                                           \
      if (v1) {                                                                                                     \
        if (pm == 2) wgrad3_launch<M, N, 2>(grid, shmem, st, P);                                                    \
        else if (pm == 1) wgrad3_launch<M, N, 1>(grid, shmem, st, P);                                               \
        else wgrad3_launch<M, N, 0>(grid, shmem, st, P);                                                            \
        goto wg_reduce_##T##M##N;                                                                                   \
      }                                                                                                             \
    }                                                                                                               \
    if (pm == 2) LMN_LAUNCH((wgrad_lds_kernel<T, M, N, 2>), grid, dim3(256), shmem, st, P);                 \
    else if (pm == 1) LMN_LAUNCH((wgrad_lds_kernel<T, M, N, 1>), grid, dim3(256), shmem, st, P);            \
    else LMN_LAUNCH((wgrad_lds_kernel<T, M, N, 0>), grid, dim3(256), shmem, st, P);                         \
  wg_reduce_##T##M##N:                                                                                              \
    if (reduce_now) {                                                                                               \
      const int ksl = reduce_slices(blocks);                                                                        \
      const int rb = (int)((per + 4096 / ksl - 1) / (4096 / ksl));                                                    \
      LMN_LAUNCH((wgrad_reduce_kernel<T, M, N>), dim3(rb, gy), dim3(1024), 0, st, P, blocks, ksl);          \
    }                                                                                                               \
  } while (0)
  if (G.direct) {
    const int64_t nb = blocks;
    const dim3 dgrid((unsigned)nb, gy);
    const bool wave_staged = G.wave_staged;
#define LMN_WDK(KERN, M, N, SH)                                                                                    \
  do {                                                                                                              \
    if (P.rpw) {   /* row-planar operands: their own instances (precision mode | 4) */                              \
      if (pm == 2) LMN_LAUNCH((KERN<M, N, 6>), dgrid, dim3(256), SH, st, P);                                        \
      else if (pm == 1) LMN_LAUNCH((KERN<M, N, 5>), dgrid, dim3(256), SH, st, P);                                   \
      else LMN_LAUNCH((KERN<M, N, 4>), dgrid, dim3(256), SH, st, P);                                                \
    } else if (pm == 2) LMN_LAUNCH((KERN<M, N, 2>), dgrid, dim3(256), SH, st, P);                                   \
    else if (pm == 1) LMN_LAUNCH((KERN<M, N, 1>), dgrid, dim3(256), SH, st, P);                                     \
    else LMN_LAUNCH((KERN<M, N, 0>), dgrid, dim3(256), SH, st, P);                                                  \
  } while (0)
#define LMN_WD(M, N)                                                                                               \
  do {                                                                                                              \
    if (wave_staged) {                                                                                              \
      const int64_t stf = 4 * (M + N) * 256, rdf = (int64_t)M * N * 256 + M * 16;                                   \
      const size_t wsh = (size_t)((stf > rdf ? stf : rdf) + 32 * N) * 4;   /* (+ the LayerNorm gamma / beta tiles) */ \
      LMN_WDK(wgrad_1x1w_kernel, M, N, wsh);                                                                        \
    } else LMN_WDK(wgrad_1x1_kernel, M, N, 0);                                                                      \
    if (reduce_now) {                                                                                               \
      const int ksl = reduce_slices((int)nb);                                                                       \
      const int rb = (int)((per + 4096 / ksl - 1) / (4096 / ksl));                                                    \
      LMN_LAUNCH((wgrad_reduce_kernel<1, M, N>), dim3(rb, gy), dim3(1024), 0, st, P, (int)nb, ksl);         \
    }                                                                                                               \
  } while (0)
#define LMN_WW(M, N)                                                                                               \
  do {                                                                                                              \
    const int64_t stf = 4 * (M + N) * 256, rdf = (int64_t)M * N * 256 + M * 16;                                     \
    const size_t wsh = (size_t)((stf > rdf ? stf : rdf) + 32 * N) * 4;                                              \
    LMN_WDK(wgrad_1x1w_kernel, M, N, wsh);                                                                          \
    if (reduce_now) {                                                                                               \
      const int ksl = reduce_slices((int)nb);                                                                       \
      const int rb = (int)((per + 4096 / ksl - 1) / (4096 / ksl));                                                    \
      LMN_LAUNCH((wgrad_reduce_kernel<1, M, N>), dim3(rb, gy), dim3(1024), 0, st, P, (int)nb, ksl);         \
    }                                                                                                               \
  } while (0)
    switch (NMT * 8 + NNT) {
      case 1 * 8 + 1: LMN_WD(1, 1); break;
      case 1 * 8 + 2: LMN_WD(1, 2); break;
      case 1 * 8 + 3: LMN_WD(1, 3); break;
      case 1 * 8 + 4: LMN_WD(1, 4); break;
      case 2 * 8 + 1: LMN_WD(2, 1); break;
      case 3 * 8 + 1: LMN_WD(3, 1); break;
      case 4 * 8 + 1: LMN_WD(4, 1); break;
      case 2 * 8 + 2: LMN_WD(2, 2); break;
      case 2 * 8 + 3: LMN_WW(2, 3); break;
      case 3 * 8 + 2: LMN_WW(3, 2); break;
      case 2 * 8 + 4: LMN_WW(2, 4); break;
      default: LMN_WW(4, 2); break;
    }
#undef LMN_WW
#undef LMN_WD
#undef LMN_WDK
    return lmn_launch_status("conv_wgrad(1x1)");
  }
#define LMN_WGS(T)                                                   \
  do {                                                               \
    if (NMT == 1 && NNT == 1) LMN_WG(T, 1, 1);                       \
    else if (NMT == 1) LMN_WG(T, 1, 2);                              \
    else if (NNT == 1) LMN_WG(T, 2, 1);                              \
    else LMN_WG(T, 2, 2);                                            \
  } while (0)
  if (A.ksize == 1) LMN_WGS(1); else LMN_WGS(9);
#undef LMN_WGS
#undef LMN_WG
  return lmn_launch_status("conv_wgrad");
}

}  // extern "C"
