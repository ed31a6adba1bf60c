// conv_dma1_kernel (round 6): the HBM-bound 1x1 convs of ReparamConv at levels 0-1 (SURVEY rows A1 / A3 and their data gradients;
// core/modules.py:537-539, 576-599) as an LDS-DMA streaming kernel: the operand tile of pixel tile t+1 -- every source, and the
// epilogue's `aux` tensor -- is filled by `buffer_load_dwordx4 ... lds` into a second LDS buffer while tile t is multiplied and stored.
//
//   F1  expand conv          z   = W_e x + b_e            x NHWC (4 / 12 / 24 ch) -> z ROW-PLANAR (24 / 48), SUM_SQ statistics
//   F2  pointwise + shortcut out = W_p (GELU(pre) s) + W_sc x + b   pre row-planar (on-load GELU x per-image gate) + x NHWC -> NHWC
//   B1  SE-gradient conv     u   = W_p^T dy               dy NHWC -> u row-planar, per-image sums of u GELU(pre) (aux = pre)
//   B2  folded data gradient dx  = [..] dh + [..] x + W_sc^T dy + k      dh row-planar + x + dy NHWC -> NHWC   (lmn_reparam_fold)
//
// Why: conv_tile_kernel stages a tile with register loads between two barriers; its five blocks per CU keep at most ~30 KB of loads in
// flight per CU and the level-0 layers stream 2.9-3.7 TB/s against 5.2-6.3 TB/s of a plain copy.  With the next tile's DMA in flight
// during the whole tile (no staging registers, no LDS commit, one barrier per tile) the stand-alone prototype (tools/micro/conv1_dma.hip)
// streams 4.7-5.2 TB/s on the same shapes (12 -> 24 at 352^2: 39.5 -> 30.7 us).
//
// LDS image of a tile: REGION-major -- [source 0: TP pixels x PS0 chunks][source 1][source 2][aux] -- every region a multiple of 64
// chunks (TP = 64 or 128 pixels), so each DMA piece (one wave-instruction, 64 x 16 B) lies in ONE region and takes that tensor's buffer
// descriptor; PS = chunks per pixel rounded up to odd (conflict-free operand reads), pad chunks are written by out-of-range lanes
// (zeros).  K mapping as conv_dma3.hip: slice s takes channel q * KS + s of its source from lane group q; the block's weights are
// gathered once from the packed fragments.  On-load transforms (F2: GELU x gate) are applied where a lane reads ITS operands from LDS
// (each element is read by exactly one lane, so once per element as before).
// Tiles never straddle images (the host requires H * W % TP == 0), blocks take CONTIGUOUS tile ranges (per-image state -- the gate
// vector, the SE sums -- changes a few times per block).
#include "conv_tile.h"

#ifdef D1_VAR
#define D1_VARV D1_VAR
#else
#define D1_VARV 0
#endif
namespace {

constexpr unsigned D1_OOB = 0x80000000u;
typedef int d1_i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ d1_i32x4 d1_rsrc(const void* p, unsigned bytes) {
  const unsigned long long a = (unsigned long long)p;
  d1_i32x4 r;
  r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
  r.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
  r.z = __builtin_amdgcn_readfirstlane((int)bytes);
  r.w = 0x00020000;
  return r;
}
__device__ __forceinline__ void d1_piece(unsigned ldsaddr, unsigned voff, d1_i32x4 rsrc) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(ldsaddr), "v"(voff), "s"(rsrc) : "memory");
}
template <int KS> struct D1Ops { float v[KS > 0 ? KS : 1]; };
template <int KS> __device__ __forceinline__ D1Ops<KS> d1_ops(const float* p) {   // KS contiguous floats at p (16- / 8- / 4-byte aligned for KS % 4 == 0 / even / odd)
  D1Ops<KS> r;
  if constexpr (KS % 4 == 0) {
#pragma unroll
    for (int i = 0; i < KS / 4; ++i) { const f32x4 t = *reinterpret_cast<const f32x4*>(p + 4 * i); r.v[4 * i] = t[0]; r.v[4 * i + 1] = t[1]; r.v[4 * i + 2] = t[2]; r.v[4 * i + 3] = t[3]; }
  } else if constexpr (KS % 2 == 0) {
#pragma unroll
    for (int i = 0; i < KS / 2; ++i) { const f32x2_t t = *reinterpret_cast<const f32x2_t*>(p + 2 * i); r.v[2 * i] = t[0]; r.v[2 * i + 1] = t[1]; }
  } else {
#pragma unroll
    for (int i = 0; i < KS; ++i) r.v[i] = p[i];
  }
  return r;
}
constexpr int d1_ps(int cq) { return cq == 0 ? 0 : ((cq & 1) ? cq : cq + 1); }

// KS0..2: channel quads of the sources (0: absent); AQ: channel quads of the aux image (SE_BWD: Cout / 4 of the stage that carries it, else 0);
// NCT: cout tiles; MODE: 0 plain, 2 SUM_SQ statistics, 5 SE_BWD; GS: source 0 carries GELU (+ per-image scale); TPG: pixel groups per wave
// (TP = 64 TPG).  NCT2 > 0: a CHAINED second 1x1 conv (lmn_conv_chain_t) with NCT2 cout tiles and epilogue MODE2 on the first conv's
// output tile (the first stage is then plain: MODE == 0, and AQ belongs to the second stage).  The accumulator fragment of the first
// MFMA chain -- lane (q, n): channels 16 c + 4 q + r of pixel n -- IS the B fragment of K slice (c, r) of the second one; its weights
// are gathered to match, nothing goes through LDS.
template <int KS0, int KS1, int KS2, int AQ, int NCT, int MODE, bool GS, int TPG, int BPC, int NCT2 = 0, int MODE2 = 0>
__global__ __launch_bounds__(256, BPC) void conv_dma1_kernel(const ConvParams P) {
  static_assert(NCT2 == 0 || MODE == 0, "a chained conv follows a plain first stage");
  constexpr int TP = 64 * TPG;
  constexpr int PS0 = d1_ps(KS0), PS1 = d1_ps(KS1), PS2 = d1_ps(KS2), PSA = d1_ps(AQ);
  constexpr int R0 = 0, R1 = R0 + TP * PS0, R2 = R1 + TP * PS1, RA = R2 + TP * PS2, NCH = RA + TP * PSA;   // region starts (chunks)
  static_assert(NCH % 64 == 0, "regions are whole DMA pieces");
  constexpr int KST = KS0 + KS1 + KS2;
  constexpr int WFL = NCT * 64 * KST;               // floats of the block's weights: [source][ct][lane][KS_s]
  constexpr int KS2ND = 4 * NCT;                    // K slices of the chained conv
  constexpr int WFL2 = NCT2 * 64 * KS2ND;           // its weights: [ct2][lane][slice (c, r)]
  constexpr int NS = TPG * (NCT + NCT2);            // stores per tile and lane
  constexpr int NCTS = NCT2 ? NCT2 : NCT;           // the stage that carries statistics: its cout tiles, its mode
  constexpr int MODES = NCT2 ? MODE2 : MODE;
  const lmn_conv_args_t& A = P.a;
  const lmn_conv_chain_t& CH = A.chain;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const s_w = smem + 2 * NCH * 4;
  float* const s_w2 = s_w + WFL;
  float* const s_red = s_w2 + WFL2;                 // [4 waves][NCTS*16]: block-level sums of the SE-gradient epilogue
  float* const s_par2 = s_red + 4 * NCTS * 16;      // chained conv: bias, shift [2][NCT2*16]
  float* const s_par = s_par2 + 2 * (NCT2 ? NCT2 : 1) * 16;   // 9 parameter vectors of the first conv (conv_stage_params)
  if (P.prio >= 4) lmn_setprio_level(7 - P.prio);   // (uniform: lmn_set_priority_stream)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, n = lane & 15;
  const int NPX = A.B * A.Win;                      // (1x1: the host flattened the image to one row of H*W pixels: A.Win = H*W)
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem;
  const int CoutS = NCT2 ? CH.Cout : A.Cout;        // channels / statistics destination / layouts of the statistics stage
  float* const statsS = NCT2 ? CH.stats : A.stats;
  const int srepS = NCT2 ? CH.stats_rep : A.stats_rep;
  const LmnLay layA = NCT2 ? P.lay_aux2 : P.lay_aux;
  // buffer descriptors of the regions' tensors (whole tensors or channel slices: the range covers every addressed element)
  auto span = [&](int C, const LmnLay& L) -> unsigned {   // bytes from the base to the end of the last addressed element
    const int64_t last = (int64_t)(NPX - 1) * L.cs + (int64_t)((C >> 2) - 1) * L.qs + 3 + (L.rf ? (int64_t)((NPX - 1) / P.rpw) * L.rf : 0);
    return (unsigned)((last + 1) * 4);
  };
  const d1_i32x4 rs0 = d1_rsrc(A.src[0].ptr, span(A.src[0].C, P.lay_src[0]));
  const d1_i32x4 rs1 = KS1 ? d1_rsrc(A.src[1].ptr, span(A.src[1].C, P.lay_src[1])) : rs0;
  const d1_i32x4 rs2 = KS2 ? d1_rsrc(A.src[2].ptr, span(A.src[2].C, P.lay_src[2])) : rs0;
  const d1_i32x4 rsa = AQ ? d1_rsrc(NCT2 ? CH.aux : A.aux, span(CoutS, layA)) : rs0;
  const BufRsrc ro = make_rsrc(A.out, A.out ? span(A.Cout, P.lay_out) : 0u);
  const BufRsrc ro2 = make_rsrc(NCT2 ? CH.out : nullptr, (NCT2 && CH.out) ? span(CH.Cout, P.lay_out2) : 0u);

  // per-thread piece descriptors (tile-independent), region by region (a region's tensor, layout and descriptor are compile-time
  // choices of its loop): pixel of the tile << 8 | chunk of the pixel, -1: pad chunk
  constexpr int NK0 = (TP * PS0 + 255) / 256, NK1 = (TP * PS1 + 255) / 256, NK2 = (TP * PS2 + 255) / 256, NKA = (TP * PSA + 255) / 256;
  int dpf[NK0 + NK1 + NK2 + NKA];
  auto mk = [&](int* d, const int nk, const int ps, const int cq) __attribute__((always_inline)) {
    for (int k = 0; k < nk; ++k) {
      const int j = k * 256 + tid;
      const int p = j / ps, f = j - p * ps;
      d[k] = (j < TP * ps && f < cq) ? (p << 8 | f) : -1;
    }
  };
  mk(dpf, NK0, PS0, KS0);
  if constexpr (KS1 > 0) mk(dpf + NK0, NK1, PS1, KS1);
  if constexpr (KS2 > 0) mk(dpf + NK0 + NK1, NK2, PS2, KS2);
  if constexpr (AQ > 0) mk(dpf + NK0 + NK1 + NK2, NKA, PSA, AQ);
  auto stage_region = [&](int tile, unsigned lbase, const int* d, const int nk, const int nchr, const LmnLay& L, const d1_i32x4& rr) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < nk; ++k) {
      if (k * 256 + wv * 64 >= nchr) continue;     // (wave-uniform: this wave's piece lies past the region)
      const int dd = d[k];
      const uint32_t gp = (uint32_t)(tile * TP + (dd >> 8));
      uint32_t off = gp * (uint32_t)L.cs + (uint32_t)(dd & 255) * (uint32_t)L.qs;
      if (L.rf) off += lmn_div_row(gp, (uint32_t)P.rpw, P.rp_magic) * (uint32_t)L.rf;   // (wave-uniform: row-planar tensor)
      d1_piece(__builtin_amdgcn_readfirstlane(lbase + (unsigned)k * 4096), dd >= 0 ? off * 4u : D1_OOB, rr);
    }
  };
  auto stage = [&](int tile, int buf) __attribute__((always_inline)) {
    const unsigned lb = lds0 + (unsigned)buf * (NCH * 16) + (unsigned)wv * 1024;
    stage_region(tile, lb + R0 * 16, dpf, NK0, TP * PS0, P.lay_src[0], rs0);
    if constexpr (KS1 > 0) stage_region(tile, lb + R1 * 16, dpf + NK0, NK1, TP * PS1, P.lay_src[1], rs1);
    if constexpr (KS2 > 0) stage_region(tile, lb + R2 * 16, dpf + NK0 + NK1, NK2, TP * PS2, P.lay_src[2], rs2);
    if constexpr (AQ > 0) stage_region(tile, lb + RA * 16, dpf + NK0 + NK1 + NK2, NKA, TP * PSA, layA, rsa);
  };

  // contiguous tile range of the block
  const int t_begin = (int)(((int64_t)blockIdx.x * P.total_tiles) / gridDim.x), t_end = (int)(((int64_t)(blockIdx.x + 1) * P.total_tiles) / gridDim.x);
  if (t_begin < t_end) stage(t_begin, 0);
  // weights: [source][ct][lane (q, m)][s] (channel q KS_s + s of the source) from the packed fragments [kb][ct][lane (q', m)][j]
  // (channel 16 kb + 4 j + q' of the source's K16 blocks); cout tiles past the end re-read the last one (dropped at the store)
  for (int i = tid; i < WFL; i += 256) {
    const int sidx = i >= NCT * 64 * (KS0 + KS1) ? 2 : (i >= NCT * 64 * KS0 ? 1 : 0);
    const int ks = sidx == 2 ? KS2 : (sidx == 1 ? KS1 : KS0);
    const int j0 = i - (sidx == 2 ? NCT * 64 * (KS0 + KS1) : (sidx == 1 ? NCT * 64 * KS0 : 0));
    const int s = j0 % ks, t = j0 / ks;
    const int l = t & 63, ct = t >> 6;
    const int ch = (l >> 4) * ks + s, m = l & 15;
    const int kb = ch >> 4, jj = (ch & 15) >> 2, qq = ch & 3;
    const int ctg = min(ct, P.NCTT - 1);
    s_w[i] = A.wpack[((int64_t)(P.kb_off[sidx] + kb) * P.NCTT + ctg) * 256 + (qq * 16 + m) * 4 + jj];
  }
  if constexpr (NCT2 > 0) {
    // chained conv: [ct2][lane (q, m)][slice (c, r)] (its input channel 16 c + 4 q + r: K16 block c, element j = q of lane (r, m))
    for (int i = tid; i < WFL2; i += 256) {
      const int sl = i % KS2ND, t = i / KS2ND;
      const int l = t & 63, ct2 = t >> 6;
      const int c = sl >> 2, r = sl & 3, qq = l >> 4, m = l & 15;
      const int ctg = min(ct2, P.NCTT2 - 1);
      s_w2[i] = CH.wpack[((int64_t)c * P.NCTT2 + ctg) * 256 + (r * 16 + m) * 4 + qq];
    }
    for (int i = tid; i < NCT2 * 16; i += 256) {
      const bool cok = i < CH.Cout;
      const float bi = (CH.bias && cok) ? CH.bias[i] : 0.f, sh = (CH.shift && cok) ? CH.shift[i] : 0.f;
      s_par2[i] = bi;
      s_par2[NCT2 * 16 + i] = sh;
      if (CH.stats_snap && blockIdx.x == 0 && cok && CH.stats) CH.stats[(int64_t)CH.stats_rep * 2 * CH.Cout + i] = sh;
    }
  }
  conv_stage_params<NCT>(A, s_par, 0, tid, blockIdx.x == 0);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  f32x4 bias4[NCT], biasS[NCTS], shiftS[NCTS];
#pragma unroll
  for (int c = 0; c < NCT; ++c) bias4[c] = *reinterpret_cast<const f32x4*>(s_par + c * 16 + q * 4);
#pragma unroll
  for (int c = 0; c < NCTS; ++c) {
    if constexpr (NCT2 > 0) {
      biasS[c] = *reinterpret_cast<const f32x4*>(s_par2 + c * 16 + q * 4);
      shiftS[c] = *reinterpret_cast<const f32x4*>(s_par2 + NCT2 * 16 + c * 16 + q * 4);
    } else {
      biasS[c] = bias4[c];
      shiftS[c] = *reinterpret_cast<const f32x4*>(s_par + 6 * NCT * 16 + c * 16 + q * 4);
    }
  }
  float st0[NCTS][4], st1[NCTS][4];
#pragma unroll
  for (int c = 0; c < NCTS; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) st0[c][r] = st1[c][r] = 0.f;
  const int tiles_img = A.Win / TP;                 // tiles per image (exact)
  int cur_b = -1;
  float gate[GS ? KS0 : 1];                         // per-image scale of this lane's source-0 channels (F2: the SE gate)
#pragma unroll
  for (int s = 0; s < (GS ? KS0 : 1); ++s) gate[s] = 1.f;
  const bool det = P.det_stats != nullptr;
  // SE_BWD: the finished image's sums.  16 pixel lanes -> wave sum -> the four waves through LDS -> ONE atomic instruction per block
  // and image (24 / 48 lanes).  Per-wave atomics (conv_tile_kernel's form) arrive from all 3,072 waves of the persistent grid at the
  // same moment here -- 384 adds per address, serialised at the memory side: 70 of the kernel's 125 us at level 0 (round 6 probe).
  // Called under a block-uniform condition (it holds two barriers).
  auto se_flush = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < NCTS; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float t = st0[c][r];
        t += __shfl_xor(t, 1, 64); t += __shfl_xor(t, 2, 64); t += __shfl_xor(t, 4, 64); t += __shfl_xor(t, 8, 64);
        if (n == 0) s_red[wv * NCTS * 16 + c * 16 + q * 4 + r] = t;
        st0[c][r] = 0.f;
      }
    __syncthreads();
    if (tid < NCTS * 16 && tid < CoutS) {
      const float v = ((s_red[tid] + s_red[NCTS * 16 + tid]) + s_red[2 * NCTS * 16 + tid]) + s_red[3 * NCTS * 16 + tid];
      lmn_red_add((det ? P.det_stats + (int64_t)(blockIdx.x * 4) * A.B * CoutS : statsS) + cur_b * CoutS + tid, v, det);
    }
    __syncthreads();
  };

  int it = 0;
  for (int tile = t_begin; tile < t_end; ++tile, ++it) {
    const int cur = it & 1;
    if (it > 0) {
      // the NS stores of the previous tile are this wave's youngest vector-memory operations (anything issued between the pieces and
      // them -- gate loads, SE atomics of an image change -- is younger than the pieces too): the pieces of THIS tile have landed
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NS) : "memory");
      __builtin_amdgcn_s_barrier();
    }
    if (tile + 1 < t_end) stage(tile + 1, cur ^ 1);
    const int b = tile / tiles_img;
    if (b != cur_b) {                               // (block-uniform) a new image
      if (MODES == 5 && cur_b >= 0) se_flush();
      cur_b = b;
      if constexpr (GS) {
        if (A.src[0].scale) {
#pragma unroll
          for (int s = 0; s < KS0; ++s) gate[s] = A.src[0].scale[b * A.src[0].C + q * KS0 + s];
        }
      }
    }
    const float* XS = smem + cur * (NCH * 4);
    f32x4 acc[TPG][NCT];
#pragma unroll
    for (int g = 0; g < TPG; ++g)
#pragma unroll
      for (int c = 0; c < NCT; ++c) acc[g][c] = bias4[c];
    // ---- source 0 (on-load transform), then sources 1 / 2: operands straight from the image, weights from LDS
    {
      D1Ops<KS0> w[NCT], x[TPG];
#pragma unroll
      for (int c = 0; c < NCT; ++c) w[c] = d1_ops<KS0>(s_w + (c * 64 + lane) * KS0);
#pragma unroll
      for (int g = 0; g < TPG; ++g) {
        x[g] = d1_ops<KS0>(XS + (R0 + ((wv + 4 * g) * 16 + n) * PS0) * 4 + q * KS0);
        if constexpr (GS) {
#pragma unroll
          for (int s = 0; s < KS0; ++s) x[g].v[s] = lmn_gelu(x[g].v[s]) * gate[s];
        }
      }
#pragma unroll
      for (int s = 0; s < KS0; ++s)
#pragma unroll
        for (int c = 0; c < NCT; ++c)
#pragma unroll
          for (int g = 0; g < TPG; ++g) acc[g][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[c].v[s], x[g].v[s], acc[g][c], 0, 0, 0);
    }
    if constexpr (KS1 > 0) {
      D1Ops<KS1> w[NCT], x[TPG];
#pragma unroll
      for (int c = 0; c < NCT; ++c) w[c] = d1_ops<KS1>(s_w + NCT * 64 * KS0 + (c * 64 + lane) * KS1);
#pragma unroll
      for (int g = 0; g < TPG; ++g) x[g] = d1_ops<KS1>(XS + (R1 + ((wv + 4 * g) * 16 + n) * PS1) * 4 + q * KS1);
#pragma unroll
      for (int s = 0; s < KS1; ++s)
#pragma unroll
        for (int c = 0; c < NCT; ++c)
#pragma unroll
          for (int g = 0; g < TPG; ++g) acc[g][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[c].v[s], x[g].v[s], acc[g][c], 0, 0, 0);
    }
    if constexpr (KS2 > 0) {
      D1Ops<KS2> w[NCT], x[TPG];
#pragma unroll
      for (int c = 0; c < NCT; ++c) w[c] = d1_ops<KS2>(s_w + NCT * 64 * (KS0 + KS1) + (c * 64 + lane) * KS2);
#pragma unroll
      for (int g = 0; g < TPG; ++g) x[g] = d1_ops<KS2>(XS + (R2 + ((wv + 4 * g) * 16 + n) * PS2) * 4 + q * KS2);
#pragma unroll
      for (int s = 0; s < KS2; ++s)
#pragma unroll
        for (int c = 0; c < NCT; ++c)
#pragma unroll
          for (int g = 0; g < TPG; ++g) acc[g][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[c].v[s], x[g].v[s], acc[g][c], 0, 0, 0);
    }
    // ---- chained conv: the output fragments are the B operands of its K slices
    f32x4 acc2[TPG][NCT2 ? NCT2 : 1];
    if constexpr (NCT2 > 0) {
      D1Ops<KS2ND> w2[NCT2];
#pragma unroll
      for (int c = 0; c < NCT2; ++c) w2[c] = d1_ops<KS2ND>(s_w2 + (c * 64 + lane) * KS2ND);
#pragma unroll
      for (int g = 0; g < TPG; ++g)
#pragma unroll
        for (int c = 0; c < NCT2; ++c) acc2[g][c] = biasS[c];
#pragma unroll
      for (int c1 = 0; c1 < NCT; ++c1)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int c = 0; c < NCT2; ++c)
#pragma unroll
            for (int g = 0; g < TPG; ++g) acc2[g][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(w2[c].v[c1 * 4 + r], acc[g][c1][r], acc2[g][c], 0, 0, 0);
    }
    // ---- epilogues: lane holds channels c*16 + q*4 .. +3 of pixel (group wv + 4g, lane n); unconditional stores (dead lanes out of range)
#pragma unroll
    for (int g = 0; g < TPG; ++g) {
      const int pl = (wv + 4 * g) * 16 + n;         // pixel of the tile
      const uint32_t gp = (uint32_t)(tile * TP + pl);
      const uint32_t rowp = (P.lay_out.rf || (NCT2 && P.lay_out2.rf)) ? lmn_div_row(gp, (uint32_t)P.rpw, P.rp_magic) : 0u;
      const uint32_t obase = gp * (uint32_t)P.lay_out.cs + rowp * (uint32_t)P.lay_out.rf;
      if constexpr (NCT2 > 0) {                     // first stage: plain store of its output
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
          const int co = c * 16 + q * 4;
          const unsigned voff = (co < A.Cout && A.out) ? (obase + (uint32_t)(co >> 2) * (uint32_t)P.lay_out.qs) * 4u : D1_OOB;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[g][c]), ro, (int)voff, 0, 0);
        }
      }
      const uint32_t obaseS = NCT2 ? gp * (uint32_t)P.lay_out2.cs + rowp * (uint32_t)P.lay_out2.rf : obase;
      const LmnLay& LO = NCT2 ? P.lay_out2 : P.lay_out;
      void* const outS = NCT2 ? CH.out : A.out;
#pragma unroll
      for (int c = 0; c < NCTS; ++c) {
        const int co = c * 16 + q * 4;
        const bool live = co < CoutS;
        f32x4 o;
        if constexpr (NCT2 > 0) o = acc2[g][c]; else o = acc[g][c];
        if (MODES == 2 && live) {
#pragma unroll
          for (int r = 0; r < 4; ++r) { const float d = o[r] - shiftS[c][r]; st0[c][r] += d; st1[c][r] += d * d; }
        }
        if constexpr (MODES == 5) {
          if (live) {
            const f32x4 ax = *reinterpret_cast<const f32x4*>(XS + (RA + pl * PSA + (co >> 2)) * 4);   // aux (pre) of this pixel and quad, from the image
#pragma unroll
            for (int r = 0; r < 4; ++r) st0[c][r] += o[r] * lmn_gelu(ax[r]);
          }
        }
        const unsigned voff = (live && outS) ? (obaseS + (uint32_t)(co >> 2) * (uint32_t)LO.qs) * 4u : D1_OOB;
        if constexpr (NCT2 > 0) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), ro2, (int)voff, 0, 0);
        else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), ro, (int)voff, 0, 0);
      }
    }
  }
  if (MODES == 5 && cur_b >= 0) se_flush();

  // ---- SUM_SQ statistics (conv_tile_kernel's tail)
  if constexpr (MODES == 2) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    float* XS = smem;
#pragma unroll
    for (int c = 0; c < NCTS; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float a = st0[c][r], bb = st1[c][r];
#pragma unroll
        for (int m = 1; m <= 8; m <<= 1) {
          a += __shfl_xor(a, m, 64);
          bb += __shfl_xor(bb, m, 64);
        }
        if (n == 0) {
          XS[wv * 2 * NCTS * 16 + c * 16 + q * 4 + r] = a;
          XS[wv * 2 * NCTS * 16 + NCTS * 16 + c * 16 + q * 4 + r] = bb;
        }
      }
    __syncthreads();
    for (int i = tid; i < 2 * NCTS * 16; i += 256) {
      const int which = i / (NCTS * 16), co = i - which * NCTS * 16;
      const float v = ((XS[i] + XS[2 * NCTS * 16 + i]) + XS[4 * NCTS * 16 + i]) + XS[6 * NCTS * 16 + i];
      if (co < CoutS) {
        if (det) P.det_stats[(int64_t)blockIdx.x * 2 * CoutS + (int64_t)which * CoutS + co] = v;
        else atomicAdd(statsS + (srepS > 1 ? (int64_t)(blockIdx.x % srepS) * 2 * CoutS : 0) + (int64_t)which * CoutS + co, v);
      }
    }
  }
}

}  // namespace

// LDS bytes of an instance: two images, the weights of both stages, SE reduction slots, parameter vectors
static size_t d1_lds(int ks0, int ks1, int ks2, int aq, int nct, int tp, int nct2) {
  const int nch = tp * (d1_ps(ks0) + d1_ps(ks1) + d1_ps(ks2) + d1_ps(aq));
  const int ncts = nct2 ? nct2 : nct;
  return (size_t)2 * nch * 16 + (size_t)nct * 64 * (ks0 + ks1 + ks2) * 4 + (size_t)nct2 * 64 * 4 * nct * 4 + (size_t)4 * ncts * 16 * 4 +
         (size_t)2 * (nct2 ? nct2 : 1) * 16 * 4 + (size_t)9 * nct * 16 * 4;
}

// Instance table: (quads of the three sources, aux quads, cout tiles, mode, GELU-scale source 0, chained cout tiles, chained mode) ->
// tile pixels; 0: no instance.  Rows 0-8: single convs (F1 stem / F1 / F2 / B1 / B2 at level 0, F1 / F2 / B1 / B2 at level 1); rows 9-11:
// chained pairs (F2 -> F1' at levels 0 and 1, B2 -> B1' at level 0).
#define D1_TABLE(X)                                                                                                   \
  X(1, 0, 0, 0, 2, 2, 0, 2, 4, 0, 0) X(3, 0, 0, 0, 2, 2, 0, 2, 4, 0, 0) X(6, 3, 0, 0, 1, 0, 1, 2, 3, 0, 0) X(3, 0, 0, 6, 2, 5, 0, 2, 3, 0, 0) \
  X(6, 3, 3, 0, 1, 0, 0, 2, 2, 0, 0) X(6, 0, 0, 0, 3, 2, 0, 2, 3, 0, 0) X(12, 6, 0, 0, 2, 0, 1, 1, 3, 0, 0) X(6, 0, 0, 12, 3, 5, 0, 1, 3, 0, 0) \
  X(12, 6, 6, 0, 2, 0, 0, 1, 2, 0, 0)                                                                                  \
  X(6, 3, 0, 0, 1, 0, 1, 2, 3, 2, 2) X(12, 6, 0, 0, 2, 0, 1, 1, 2, 3, 2) X(6, 3, 3, 6, 1, 0, 0, 1, 3, 2, 5)

int lmn_conv_dma1_tp(int ks0, int ks1, int ks2, int aq, int nct, int mode, int gs, int nct2, int mode2) {
  static int mask = -1;   // LMN_CONV_DMA1_MASK: bit i = instance i of the table (A/B runs)
  // (default: without row 8, the level-1 three-source data gradient -- 54 KB of LDS per block, slower inside the step: 13.95 vs 13.90 ms)
  if (mask < 0) { const char* e = getenv("LMN_CONV_DMA1_MASK"); mask = e ? (int)strtol(e, nullptr, 0) : 0xeff; }
  int idx = 0;
#define D1_ROW(a, b, c, d, e, f, g, tpg, bpc, e2, f2) { if (ks0 == a && ks1 == b && ks2 == c && aq == d && nct == e && mode == f && gs == g && nct2 == e2 && mode2 == f2) return ((mask >> idx) & 1) ? 64 * tpg : 0; ++idx; }
  D1_TABLE(D1_ROW)
#undef D1_ROW
  return 0;
}

int lmn_launch_conv_dma1(const ConvParams& T, int ks0, int ks1, int ks2, int aq, int nct, int mode, int gs, int nct2, int mode2, int blocks, hipStream_t st) {
#define D1_GO(a, b, c, d, e, f, g, tpg, bpc, e2, f2)                                                                                       \
  if (ks0 == a && ks1 == b && ks2 == c && aq == d && nct == e && mode == f && gs == g && nct2 == e2 && mode2 == f2) {                     \
    const size_t sh = d1_lds(a, b, c, d, e, 64 * tpg, e2);                                                                                 \
    if (sh > 64 * 1024) (void)hipFuncSetAttribute((const void*)conv_dma1_kernel<a, b, c, d, e, f, g != 0, tpg, bpc, e2, f2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); \
    LMN_LAUNCH((conv_dma1_kernel<a, b, c, d, e, f, g != 0, tpg, bpc, e2, f2>), dim3(blocks), dim3(256), sh, st, T);                       \
    return 0;                                                                                                                              \
  }
  D1_TABLE(D1_GO)
#undef D1_GO
  return -1;
}

size_t lmn_conv_dma1_lds(int ks0, int ks1, int ks2, int aq, int nct, int tp, int nct2) { return d1_lds(ks0, ks1, ks2, aq, nct, tp, nct2); }
