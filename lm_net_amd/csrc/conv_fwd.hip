// Dense convolution family on the CDNA4 matrix cores -- forward, data-gradient and weight-gradient.
//
// One kernel family serves every nn.Conv2d(k=1|3, stride 1|2) and nn.Linear on the LM-Net path
// (reference call sites listed in include/lmnet_hip.h).  Design (MI355X-first, not a translation):
//
//  * implicit GEMM on v_mfma_f32_16x16x4_f32 (exact fp32, k-ordered fma chain): M = output
//    channels, N = 16 output pixels ("pixel group"), K = (tap, input channel).  With M on the MFMA
//    row index each lane ends up holding 4 CONSECUTIVE output channels of ONE pixel, so the epilogue
//    (bias / BN affine / activation / dropout / residual) and the NHWC store are float4-wide.
//  * no LDS and no barriers: the B operand (activations) is loaded straight from HBM/L2 as float4
//    per lane -- lane (q = lane>>4, n = lane&15) loads channels 16t+4q..+3 of pixel n; element j of
//    that float4 feeds MFMA j of the K16 block, whose k index q therefore means channel 16t+4q+j.
//    The weights are pre-packed (lmn_conv_pack) in exactly that fragment order, so the A operand is
//    one coalesced 1 KiB float4 load per (tap, K16 block, cout tile), served by L1/L2.
//  * a wave owns NPG pixel groups x NCT cout tiles (accumulators in registers, 4 VGPR each) and a
//    contiguous range of group sets, so per-channel statistics (BatchNorm batch stats, SE gradient)
//    are reduced in registers first and hit global atomics once per block.
//  * pixel groups are flattened over one image (never straddle images: the SE scale and the
//    per-image statistics are wave-uniform).  For the data gradient of a stride-2 conv the groups
//    enumerate one parity class of the output at a time, so the valid taps are uniform per group.
//
// This file: the host entry of the forward / data-gradient convs (lmn_conv_fwd: validation, tiling, launch geometry), weight packing and
// the z-path fold kernels.  The kernel instances live in conv_tile_1x1.hip / conv_tile_3x3.hip / conv_tileM.hip (conv_tile.h,
// conv_tileM.h), the weight gradients in conv_wgrad.hip.
#include "conv_common.h"

namespace {

// ------------------------------------------------------------------------------------ weight packing
struct PackGeom {
  int taps, Cout, Cin, nsrc, cs[3], transposed, row_off, rows;
  int bf16;  // 1: fragment order of v_mfma_f32_16x16x16_bf16 (lane (q, n) holds channels 4q..4q+3), elements stored as bf16
};

// One element i of the packed stream [tap][K16 block][cout tile][64 lanes][4].
__device__ __forceinline__ float pack_element(const float* __restrict__ w, const PackGeom& g, int64_t i) {
  int nkb[3], kboff[3], cbase[3], NKB = 0, cb = 0;
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    nkb[s] = s < g.nsrc ? (g.cs[s] + 15) / 16 : 0;
    kboff[s] = NKB;
    cbase[s] = cb;
    NKB += nkb[s];
    cb += s < g.nsrc ? g.cs[s] : 0;
  }
  const int nrows = g.transposed ? g.rows : g.Cout;
  const int NCTT = (nrows + 15) / 16;
  const int j = (int)(i & 3);
  const int lane = (int)((i >> 2) & 63);
  uint32_t t = (uint32_t)(i >> 8);   // (a packed weight has < 2^31 elements: 32-bit divisions -- the 64-bit ones were most of this kernel's time)
  const int ct = (int)(t % (uint32_t)NCTT);
  t /= (uint32_t)NCTT;
  const int kb = (int)(t % (uint32_t)NKB);
  const int tap = (int)(t / (uint32_t)NKB);
  int s = 0;
  while (s + 1 < g.nsrc && kb >= kboff[s + 1]) ++s;
  // MFMA j of a K16 block covers channels 4j .. 4j+3 (k slot = lane >> 4): a source whose last block holds fewer than
  // 16 channels leaves whole MFMAs zero, which the kernels skip (12 channels: 3 of 4)
  const int kk = (kb - kboff[s]) * 16 + (g.bf16 ? (lane >> 4) * 4 + j : j * 4 + (lane >> 4));  // reduction index inside the source
  const int row = ct * 16 + (lane & 15);
  float v = 0.f;
  // (a source / row count may be the weight's padded to a multiple of 4 -- the RGB input travels as NHWC4, the 2-class
  //  head is computed on 4 rows: elements past the real weight are zeros)
  if (row < nrows && kk < g.cs[s]) {
    if (!g.transposed) {
      if (row < g.Cout && cbase[s] + kk < g.Cin) v = w[((int64_t)row * g.Cin + cbase[s] + kk) * g.taps + tap];
    } else {  // rows = forward input channels, reduction = forward output channels
      if (kk < g.Cout && g.row_off + row < g.Cin) v = w[((int64_t)kk * g.Cin + g.row_off + row) * g.taps + tap];
    }
  }
  return v;
}

__device__ __forceinline__ void pack_store(float* __restrict__ wp, int64_t i, float v, int bf16) {
  if (bf16) reinterpret_cast<__bf16*>(wp)[i] = (__bf16)v;
  else wp[i] = v;
}

__global__ void conv_pack_kernel(const float* __restrict__ w, float* __restrict__ wp, const PackGeom g, int64_t total) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
    pack_store(wp, i, pack_element(w, g, i), g.bf16);
}

// All weights of a pass in ONE launch: block b finds its job by bisection over the jobs' first-block table, then
// packs 1024 consecutive elements of it.  (192 separate pack launches cost ~1 ms per training step.)
__global__ __launch_bounds__(256) void conv_pack_batch_kernel(const lmn_pack_job_t* __restrict__ jobs, int njobs) {
  int lo = 0, hi = njobs - 1;
  const int64_t b = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].first_block <= b) lo = mid; else hi = mid - 1;
  }
  const lmn_pack_job_t J = jobs[lo];
  PackGeom g;
  g.taps = J.ksize * J.ksize; g.Cout = J.Cout; g.Cin = J.Cin; g.nsrc = J.nsrc;
  g.cs[0] = J.c[0]; g.cs[1] = J.c[1]; g.cs[2] = J.c[2];
  g.transposed = J.transposed; g.row_off = J.row_off; g.rows = J.rows;
  g.bf16 = J.dtype == LMN_BF16;
  const int64_t base = (b - J.first_block) * 1024;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t i = base + r * 256 + threadIdx.x;
    if (i < J.total) pack_store(J.wpack, i, pack_element(J.w, g, i), g.bf16);
  }
}

// ------------------------------------------------------------------------------------ ReparamConv backward, folded
// z-path (lmn_dw_pre_t): BatchNorm backward of the expand conv folded into WEIGHTS.  With dh = dL/d(BN output) and the sums
// S0 = sum dh, S1 = sum dh * z (lmn_dw_bwd_bn), the gradient w.r.t. the conv output is affine in two stored tensors,
//     dz = a * dh + b * z + c,   a = A,  b = -A * T * rstd / N,  c = -A * S0 / N - b * mean,  T = (S1 - mean * S0) * rstd
// (eval-mode BatchNorm: b = c = 0), and z = W_e x + bias_e, so the gradient w.r.t. the block input is ONE conv over three sources
//     dx = [W_e^T diag(a)] dh + [W_e^T diag(b) W_e] x + W_sc^T dy + W_e^T (b * bias_e + c)
// -- no statistics conv, no second conv that writes dz, no separate shortcut gradient on the critical path.  This kernel forms
// a / b / c (every block, in LDS; block 0 stores them for the weight-gradient side and adds the gamma / beta gradients) and
// writes the three operators straight into the packed fragment order of lmn_conv_fwd (sources dh [E], x [Cin], dy [Cout]).
struct FoldParams {
  const float* hstats; const float* mean; const float* rstd; const float* A;   // [2][E], [E] x 3
  const float* we; const float* be; const float* wsc;                          // [E][cinw], [E], [coutw][cinw]  (torch layouts, k = 1)
  float* wpack; float* kbias; float* coef; float* dgamma; float* dbeta;        // packed operators, [rows], [3][E], [E] +=, [E] +=
  float count; int batch_stats;
  int prio;                               // wave priority of the launch stream (lmn_set_priority_stream)
  int E, rows, cinw, cred, coutw, bf16;   // rows: channels of x / dx (a multiple of 4 >= cinw); cred: channels of dy (>= coutw)
};
__global__ __launch_bounds__(256) void reparam_fold_kernel(const FoldParams P) {
  // one block per packed fragment tile (K16 block kb, row tile ct): 64 lanes x 4 elements.  The Q tiles need E-long dot
  // products of two 16-column panels of W_e: both panels are staged in LDS with coalesced 64-byte rows (read straight from
  // global memory, the loop was a chain of 2 * E dependent L2 round trips per thread: 24 us for a 2 MFLOP problem).
  lmn_setprio_level(P.prio);
  extern __shared__ float sm[];   // a[E], b[E], c[E], panel R [E][16], panel K [E][16]
  float* sa = sm; float* sb = sm + P.E; float* sc = sm + 2 * P.E;
  float* pr = sm + 3 * P.E; float* pk = pr + P.E * 16;
  const int tid = threadIdx.x;
  for (int e = tid; e < P.E; e += 256) {
    const float S0 = P.hstats[e], S1 = P.hstats[P.E + e];
    const float mu = P.mean[e], rs = P.rstd[e], A = P.A[e];
    const float T = (S1 - mu * S0) * rs;   // sum dh * zhat
    const float b = P.batch_stats ? -A * T * rs / P.count : 0.f;
    const float c = P.batch_stats ? (-A * S0 / P.count - b * mu) : 0.f;
    sa[e] = A; sb[e] = b; sc[e] = c;
    if (blockIdx.x == 0) {
      P.coef[e] = A; P.coef[P.E + e] = b; P.coef[2 * P.E + e] = c;
      if (P.dgamma) P.dgamma[e] += T;
      if (P.dbeta) P.dbeta[e] += S0;
    }
  }
  const int nkb0 = (P.E + 15) / 16, nkb1 = (P.rows + 15) / 16, nkb2 = (P.cred + 15) / 16;
  const int NCTT = (P.rows + 15) / 16;
  const int ntiles = (nkb0 + nkb1 + nkb2) * NCTT;
  if ((int)blockIdx.x < ntiles) {
    const int ct = blockIdx.x % NCTT, kb = blockIdx.x / NCTT;
    const int s = kb < nkb0 ? 0 : (kb < nkb0 + nkb1 ? 1 : 2);
    const int kbl = kb - (s == 0 ? 0 : (s == 1 ? nkb0 : nkb0 + nkb1));
    if (s == 1) {   // panels W_e[:, ct*16 .. +16) and W_e[:, kbl*16 .. +16)
      for (int i = tid; i < P.E * 16; i += 256) {
        const int e = i >> 4, c = i & 15;
        const int r0 = ct * 16 + c, k0 = kbl * 16 + c;
        pr[i] = r0 < P.cinw ? P.we[(int64_t)e * P.cinw + r0] : 0.f;
        pk[i] = k0 < P.cinw ? P.we[(int64_t)e * P.cinw + k0] : 0.f;
      }
    }
    __syncthreads();
    const int lane = tid >> 2, j = tid & 3;
    const int kq = P.bf16 ? (lane >> 4) * 4 + j : j * 4 + (lane >> 4);   // reduction index inside the K16 block
    const int kk = kbl * 16 + kq, rl = lane & 15, row = ct * 16 + rl;
    float v = 0.f;
    if (row < P.cinw) {
      if (s == 0) {
        if (kk < P.E) v = P.we[(int64_t)kk * P.cinw + row] * sa[kk];
      } else if (s == 1) {
        if (kk < P.cinw) {
          float q = 0.f;
          for (int e = 0; e < P.E; ++e) q += pr[e * 16 + rl] * sb[e] * pk[e * 16 + kq];
          v = q;
        }
      } else {
        if (kk < P.coutw) v = P.wsc[(int64_t)kk * P.cinw + row];
      }
    }
    pack_store(P.wpack, (int64_t)blockIdx.x * 256 + tid, v, P.bf16);
  } else {   // the last block: kbias[row] = sum_e W_e[e][row] * (b[e] * bias_e[e] + c[e])  (16 lanes per row over e)
    __syncthreads();
    const int sub = tid & 15;
    for (int row = tid >> 4; row < P.rows; row += 16) {
      float k = 0.f;
      if (row < P.cinw)
        for (int e = sub; e < P.E; e += 16) k += P.we[(int64_t)e * P.cinw + row] * (sb[e] * P.be[e] + sc[e]);
      k += __shfl_xor(k, 1, 64); k += __shfl_xor(k, 2, 64); k += __shfl_xor(k, 4, 64); k += __shfl_xor(k, 8, 64);
      if (sub == 0) P.kbias[row] = k;
    }
  }
}

// Weight / bias gradient of the expand conv on the z-path without materialising dz = a dh + b z + c:
//   dW_e = sum_p dz x^T = diag(a) R + diag(b) (W_e M + b_e m^T) + c m^T,   R = sum dh x^T (the raw weight gradient),
//   M = sum x x^T, m = sum x (one weight-gradient launch over x alone, in the forward), since z = W_e x + b_e;
//   db_e = a S0 + b (W_e m + N b_e) + c N   (zero up to rounding under batch statistics, as in the reference's autograd).
__global__ __launch_bounds__(256) void reparam_wfin_kernel(const float* __restrict__ R, const float* __restrict__ M,
                                                           const float* __restrict__ m, const float* __restrict__ coef,
                                                           const float* __restrict__ hstats, const float* __restrict__ we,
                                                           const float* __restrict__ be, float count, int E, int rows, int cinw,
                                                           float* __restrict__ dW, float* __restrict__ db, const lmn_se_params_t se,
                                                           int nw) {
  if ((int)blockIdx.x >= nw) {   // (block-uniform) the squeeze-excite parameter gradients of the same block ride along
    lmn_se_bwd_params_item(((int)blockIdx.x - nw) * 256 + (int)threadIdx.x, se.dvec, se.gsum, se.inv_hw, se.hidden, se.dw1, se.db1, se.dw2,
                           se.db2, se.B, se.E, se.R);
    return;
  }
  // 16 lanes per output element split the cinw-long dot product (one thread per element was a chain of up to 192 dependent
  // L2 round trips: 24 us per launch, 16 launches per step)
  const int i = (blockIdx.x * 256 + threadIdx.x) >> 4, sub = threadIdx.x & 15;
  const bool live = i < E * (cinw + 1);
  const int e = live ? i / (cinw + 1) : 0, c = live ? i - e * (cinw + 1) : 0;
  float acc = 0.f;
  if (live) {
    if (c < cinw) {
      for (int j = sub; j < cinw; j += 16) acc += we[(int64_t)e * cinw + j] * M[(int64_t)j * rows + c];
    } else {
      for (int j = sub; j < cinw; j += 16) acc += we[(int64_t)e * cinw + j] * m[j];
    }
  }
  acc += __shfl_xor(acc, 1, 64); acc += __shfl_xor(acc, 2, 64); acc += __shfl_xor(acc, 4, 64); acc += __shfl_xor(acc, 8, 64);
  if (!live || sub != 0) return;
  const float a = coef[e], b = coef[E + e], k = coef[2 * E + e];
  if (c < cinw) {
    const float zx = be[e] * m[c] + acc;
    dW[(int64_t)e * cinw + c] += a * R[(int64_t)e * rows + c] + b * zx + k * m[c];
  } else if (db) {
    const float zs = count * be[e] + acc;
    db[e] += a * hstats[e] + b * zs + k * count;
  }
}

}  // namespace

thread_local char g_lmn_err[256] = {0};

extern "C" {

int lmn_abi_version(void) { return LMN_ABI_VERSION; }
int lmn_sizeof_conv_args(void) { return (int)sizeof(lmn_conv_args_t); }
int lmn_sizeof_src(void) { return (int)sizeof(lmn_src_t); }
int lmn_sizeof_pack_job(void) { return (int)sizeof(lmn_pack_job_t); }
const char* lmn_last_error(void) { return g_lmn_err; }
#ifdef LMN_CT_TIMING
int lmn_ct_timing(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ct_timing), sizeof(unsigned long long) * n);
}
#endif

int64_t lmn_conv_pack_size(int ksize, int Cout, int nsrc, const int32_t* c) {
  int64_t nkb = 0;
  for (int s = 0; s < nsrc; ++s) nkb += (c[s] + 15) / 16;
  return (int64_t)ksize * ksize * nkb * ((Cout + 15) / 16) * 256;
}

int lmn_conv_pack(const float* w, float* wpack, int ksize, int Cout, int Cin, int nsrc, const int32_t* c,
                  int transposed, int row_off, int rows, int dtype, lmn_stream_t stream) {
  LMN_REQUIRE(w && wpack && c, "conv_pack: null pointer");
  LMN_REQUIRE(dtype == LMN_F32 || dtype == LMN_BF16, "conv_pack: dtype %d", dtype);
  if (g_lmn_rec && nsrc >= 1 && nsrc <= 3) {
    int32_t cc[3] = {c[0], nsrc > 1 ? c[1] : 0, nsrc > 2 ? c[2] : 0};
    const int32_t c0 = cc[0], c1 = cc[1], c2 = cc[2];
    lmn_rec_push([=]() -> int { const int32_t ca[3] = {c0, c1, c2}; return lmn_conv_pack(w, wpack, ksize, Cout, Cin, nsrc, ca, transposed, row_off, rows, dtype, stream); }, "lmn_conv_pack(");
  }
  LMN_REQUIRE(ksize == 1 || ksize == 3, "conv_pack: ksize %d", ksize);
  LMN_REQUIRE(nsrc >= 1 && nsrc <= 3, "conv_pack: nsrc %d", nsrc);
  int csum = 0;
  for (int s = 0; s < nsrc; ++s) csum += c[s];
  if (!transposed) {
    LMN_REQUIRE(csum == Cin || (nsrc == 1 && csum > Cin && csum < Cin + 4),
                "conv_pack: sources sum to %d channels, weight has %d", csum, Cin);
  } else {
    LMN_REQUIRE(nsrc == 1 && c[0] >= Cout && c[0] < Cout + 4, "conv_pack(transposed): one source of Cout=%d channels expected", Cout);
    LMN_REQUIRE(row_off >= 0 && rows > 0 && row_off + rows < Cin + 4, "conv_pack(transposed): rows [%d,+%d) of %d", row_off, rows, Cin);
  }
  const int nrows = transposed ? rows : (Cout + 3) / 4 * 4;
  const int64_t total = lmn_conv_pack_size(ksize, nrows, nsrc, c);
  LMN_REQUIRE(total < (1LL << 31), "conv_pack: %lld packed elements (32-bit index arithmetic)", (long long)total);
  const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  PackGeom g;
  g.taps = ksize * ksize; g.Cout = Cout; g.Cin = Cin; g.nsrc = nsrc;
  g.cs[0] = c[0]; g.cs[1] = nsrc > 1 ? c[1] : 0; g.cs[2] = nsrc > 2 ? c[2] : 0;
  g.transposed = transposed; g.row_off = row_off; g.rows = rows;
  g.bf16 = dtype == LMN_BF16;
  LMN_LAUNCH(conv_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, wpack, g, total);
  return lmn_launch_status("conv_pack");
}

int lmn_conv_pack_batch(const lmn_pack_job_t* jobs_dev, int njobs, int64_t total_blocks, lmn_stream_t stream) {
  LMN_REC(lmn_conv_pack_batch(jobs_dev, njobs, total_blocks, stream));
  LMN_REQUIRE(jobs_dev && njobs > 0 && total_blocks > 0 && total_blocks < (1LL << 31), "conv_pack_batch: bad job table");
  LMN_LAUNCH(conv_pack_batch_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, jobs_dev, njobs);
  return lmn_launch_status("conv_pack_batch");
}

// LDS-DMA kernels (conv_dma3.hip, conv_dma1.hip): -1 = read LMN_CONV_DMA at the first call (default 3), bit 0 = the 3x3 kernel, bit 1 = the
// 1x1 kernel; calls with fewer tiles than the threshold keep conv_tile_kernel (the resident blocks want tiles to pipeline)
static int g_conv_dma_mode = -1;
static int g_conv_dma_min_tiles = 512;
int lmn_conv_dma_config(int mode, int min_tiles) {
  const int prev = g_conv_dma_mode;
  if (mode >= 0) g_conv_dma_mode = mode;
  if (min_tiles >= 0) g_conv_dma_min_tiles = min_tiles;
  return prev;
}

// Does the LDS-DMA streaming 1x1 kernel (conv_dma1.hip) take this call (UNFLATTENED arguments), and as which instance?
struct Dma1Sel { int ks[3]; int aq, nct, mode, gs, nct2, mode2, tp; };
static bool dma1_select(const lmn_conv_args_t& A, Dma1Sel& S) {
  if (g_conv_dma_mode < 0) { const char* e = getenv("LMN_CONV_DMA"); g_conv_dma_mode = e ? atoi(e) : 7; }
  if (!(g_conv_dma_mode & 2)) return false;
  if (A.ksize != 1 || A.stride != 1 || A.mma_dtype != LMN_F32 || A.act_dtype != LMN_F32 || A.drop_p != 0.f || A.fin.mode != LMN_FIN_NONE || A.residual) return false;
  if (A.nsrc < 1 || A.nsrc > 3 || A.Cout <= 0 || A.Cout % 4 || A.Cout > 48 || A.out_rp_w < 0) return false;
  if (A.Hin != A.Hout || A.Win != A.Wout) return false;
  S.ks[0] = S.ks[1] = S.ks[2] = 0;
  S.gs = 0;
  for (int s = 0; s < A.nsrc; ++s) {
    const lmn_src_t& X = A.src[s];
    if (!X.ptr || X.C <= 0 || X.C % 4 || X.C > 48) return false;
    if (s == 0 && X.flags == LMN_SRC_GELU) S.gs = 1;
    else if (X.flags != 0 || X.scale) return false;
    if (X.rp_w && X.rp_w != A.Win) return false;
    S.ks[s] = X.C / 4;
  }
  if ((A.out_rp_w && A.out_rp_w != A.Win) || (A.aux && A.aux_rp_w && A.aux_rp_w != A.Win)) return false;
  const bool sebwd = A.epilogue == LMN_EP_SE_BWD && A.stats_mode == LMN_STATS_EP && A.aux && A.stats;
  S.mode = sebwd ? 5 : ((A.epilogue == LMN_EP_LINEAR && A.stats_mode == LMN_STATS_SUM_SQ && A.stats) ? 2 : ((A.epilogue == LMN_EP_LINEAR && A.stats_mode == LMN_STATS_NONE) ? 0 : -1));
  if (S.mode < 0 || (!sebwd && A.aux)) return false;
  S.nct = (A.Cout + 15) / 16;
  S.aq = sebwd ? A.Cout / 4 : 0;
  S.nct2 = S.mode2 = 0;
  int64_t maxc = A.Cout > A.out_cstride ? A.Cout : A.out_cstride;
  if (sebwd && A.aux_cstride > maxc) maxc = A.aux_cstride;
  for (int s = 0; s < A.nsrc; ++s) maxc = A.src[s].cstride > maxc ? A.src[s].cstride : maxc;
  const lmn_conv_chain_t& C2 = A.chain;
  if (C2.wpack) {
    if (S.mode != 0 || !A.out || !C2.out || C2.Cout <= 0 || C2.Cout % 4 || C2.Cout > 48 || g_lmn_det) return false;
    if ((C2.out_rp_w && C2.out_rp_w != A.Win) || (C2.aux && C2.aux_rp_w && C2.aux_rp_w != A.Win)) return false;
    const bool se2 = C2.epilogue == LMN_EP_SE_BWD && C2.stats_mode == LMN_STATS_EP && C2.aux && C2.stats;
    const bool sq2 = C2.epilogue == LMN_EP_LINEAR && C2.stats_mode == LMN_STATS_SUM_SQ && C2.stats && !C2.aux && C2.stats_rep >= 1;
    if (!se2 && !sq2) return false;
    S.mode2 = se2 ? 5 : 2;
    S.nct2 = (C2.Cout + 15) / 16;
    S.aq = se2 ? C2.Cout / 4 : 0;
    if (C2.Cout > maxc) maxc = C2.Cout;
    if (C2.out_cstride > maxc) maxc = C2.out_cstride;
    if (se2 && C2.aux_cstride > maxc) maxc = C2.aux_cstride;
  }
  S.tp = lmn_conv_dma1_tp(S.ks[0], S.ks[1], S.ks[2], S.aq, S.nct, S.mode, S.gs, S.nct2, S.mode2);
  if (S.tp <= 0) return false;
  const int64_t hw = (int64_t)A.Hin * A.Win, npx = (int64_t)A.B * hw;
  return hw % S.tp == 0 && npx / S.tp >= g_conv_dma_min_tiles && npx * maxc * 4 < 0x7fffffffLL;
}

int lmn_conv_chain_ok(const lmn_conv_args_t* args) {
  if (!args || !args->chain.wpack) return 0;
  Dma1Sel S;
  return dma1_select(*args, S) ? 1 : 0;
}

int lmn_conv_fwd(const lmn_conv_args_t* args, lmn_stream_t stream) {
  LMN_REQUIRE(args, "conv_fwd: null args");
  if (g_lmn_rec) {
    const lmn_conv_args_t copy = *args;
    lmn_rec_push([copy, stream]() -> int { return lmn_conv_fwd(&copy, stream); }, "lmn_conv_fwd(");
  }
  const lmn_conv_args_t& A = *args;
  if (A.chain.wpack) {
    Dma1Sel Sc;
    LMN_REQUIRE(dma1_select(A, Sc), "conv_fwd: these arguments do not take a chained second conv (lmn_conv_chain_ok says so beforehand)");
  }
  LMN_REQUIRE(A.ksize == 1 || A.ksize == 3, "conv_fwd: ksize %d", A.ksize);
  LMN_REQUIRE(A.stride == 1 || A.stride == 2, "conv_fwd: stride %d", A.stride);
  LMN_REQUIRE(A.nsrc >= 1 && A.nsrc <= 3, "conv_fwd: nsrc %d", A.nsrc);
  LMN_REQUIRE(A.Cout > 0 && A.Cout % 4 == 0, "conv_fwd: Cout %d must be a positive multiple of 4", A.Cout);
  LMN_REQUIRE(A.mma_dtype == LMN_F32 || A.mma_dtype == LMN_BF16, "conv_fwd: mma_dtype %d", A.mma_dtype);
  LMN_REQUIRE(A.act_dtype == LMN_F32 || (A.act_dtype == LMN_BF16 && A.mma_dtype == LMN_BF16),
              "conv_fwd: act_dtype %d with mma_dtype %d (bf16 storage needs bf16 matrix-core operands)", A.act_dtype, A.mma_dtype);
  LMN_REQUIRE(A.B > 0 && A.Hout > 0 && A.Wout > 0 && A.Hin > 0 && A.Win > 0, "conv_fwd: empty tensor");
  LMN_REQUIRE(A.wpack, "conv_fwd: null packed weights");
  LMN_REQUIRE(A.out || A.stats, "conv_fwd: neither out nor stats requested");
  LMN_REQUIRE(!(A.transposed && A.nsrc != 1), "conv_fwd: transposed form takes one source");
  {
    static int trace = -1;   // LMN_CONV_TRACE=1: one line per call on stderr (which layer shapes reach which kernel form)
    if (trace < 0) { const char* e = getenv("LMN_CONV_TRACE"); trace = e ? atoi(e) : 0; }
    if (trace) {
      fprintf(stderr, "conv_fwd k%d s%d T%d B%d %dx%d->%dx%d nsrc%d C", A.ksize, A.stride, A.transposed, A.B, A.Hin, A.Win, A.Hout, A.Wout, A.nsrc);
      for (int s = 0; s < A.nsrc; ++s) fprintf(stderr, "%s%d/%d(f%d%s%s)", s ? "+" : "", A.src[s].C, A.src[s].cstride, A.src[s].flags, A.src[s].scale ? "s" : "", A.src[s].rp_w ? "r" : "");
      fprintf(stderr, " Cout%d/%d ep%d st%d act%d drop%g res%d aux%d fin%d mma%d adt%d orp%d bias%d%d\n", A.Cout, A.out_cstride, A.epilogue, A.stats_mode, A.act, (double)A.drop_p,
              A.residual != nullptr, A.aux != nullptr, A.fin.mode, A.mma_dtype, A.act_dtype, A.out_rp_w, A.bias != nullptr, A.bias2 != nullptr);
    }
  }
  ConvParams P;
  P.det_stats = nullptr;
  P.prio = lmn_prio_level((hipStream_t)stream);
  if (P.prio) P.prio = 7 - P.prio;   // 4..6: uniform raised issue priority 3..1 (lmn_set_priority_stream)
  P.a = A;
  P.NKB = 0;
  for (int s = 0; s < 3; ++s) {
    P.nkb[s] = P.kb_off[s] = 0;
    P.inv_keep_src[s] = 1.f;
  }
  for (int s = 0; s < A.nsrc; ++s) {
    LMN_REQUIRE(A.src[s].ptr && A.src[s].C > 0 && A.src[s].C % 4 == 0 && A.src[s].cstride >= A.src[s].C && A.src[s].cstride % 4 == 0,
                "conv_fwd: source %d: C=%d cstride=%d (need multiples of 4, cstride>=C)", s, A.src[s].C, A.src[s].cstride);
    P.nkb[s] = (A.src[s].C + 15) / 16;
    P.kb_off[s] = P.NKB;
    P.NKB += P.nkb[s];
    if (A.src[s].flags & LMN_SRC_DROP) {
      LMN_REQUIRE(A.src[s].drop_p >= 0.f && A.src[s].drop_p < 1.f, "conv_fwd: source dropout p");
      P.inv_keep_src[s] = 1.f / (1.f - A.src[s].drop_p);
    }
  }
  bool ln = false;   // LMN_SRC_LN: LayerNorm of the source where it is staged
  for (int s = 0; s < A.nsrc; ++s) ln = ln || (A.src[s].flags & LMN_SRC_LN) != 0;
  if (ln) {
    LMN_REQUIRE(A.nsrc == 1 && A.ksize == 1 && A.stride == 1 && !A.transposed && A.src[0].rp_w == 0 && !A.out_rp_w,
                "conv_fwd: LMN_SRC_LN belongs to single-source NHWC 1x1 stride-1 forward calls");
    LMN_REQUIRE(A.src[0].ln_gamma && A.src[0].ln_beta && A.src[0].ln_eps > 0.f, "conv_fwd: LMN_SRC_LN needs ln_gamma / ln_beta / ln_eps");
    LMN_REQUIRE(A.epilogue == LMN_EP_LINEAR && A.stats_mode == LMN_STATS_NONE && A.drop_p == 0.f && A.fin.mode == LMN_FIN_NONE,
                "conv_fwd: LMN_SRC_LN comes with the plain epilogue (bias / residual only)");
    LMN_REQUIRE(!(A.src[0].flags & ~LMN_SRC_LN) && !A.src[0].scale, "conv_fwd: LMN_SRC_LN does not combine with other source transforms");
  }
  bool up2 = false;  // LMN_SRC_UP2: the source is the half-resolution map, sampled through bilinear x2 where the window is staged
  for (int s = 0; s < A.nsrc; ++s) up2 = up2 || (A.src[s].flags & LMN_SRC_UP2) != 0;
  if (up2) {
    LMN_REQUIRE(A.nsrc == 1 && A.ksize == 3 && A.stride == 1 && !A.transposed && A.src[0].rp_w == 0 && !A.out_rp_w,
                "conv_fwd: LMN_SRC_UP2 belongs to single-source NHWC 3x3 stride-1 forward calls");
    LMN_REQUIRE(A.Hin % 2 == 0 && A.Win % 2 == 0 && A.Hin >= 4 && A.Win >= 4, "conv_fwd: LMN_SRC_UP2: Hin x Win (%d x %d) is the UPSAMPLED size, twice the source map", A.Hin, A.Win);
    LMN_REQUIRE(A.epilogue == LMN_EP_LINEAR && A.stats_mode == LMN_STATS_NONE && A.drop_p == 0.f && A.fin.mode == LMN_FIN_NONE,
                "conv_fwd: LMN_SRC_UP2 comes with the plain epilogue (bias / residual only)");
    LMN_REQUIRE(!(A.src[0].flags & ~LMN_SRC_UP2) && !A.src[0].scale, "conv_fwd: LMN_SRC_UP2 does not combine with other source transforms");
  }
  {  // row-planar operands (RP4): whole tensors of a 1x1 stride-1 call, one image width for all of them
    int rw = 0;
    for (int s = 0; s < 3; ++s) {
      const int w = s < A.nsrc ? A.src[s].rp_w : 0;
      P.lay_src[s] = lmn_lay_make(w, s < A.nsrc ? A.src[s].C : 4, s < A.nsrc ? A.src[s].cstride : 4);
      if (w) {
        LMN_REQUIRE(w > 0 && A.src[s].cstride == A.src[s].C && (rw == 0 || rw == w), "conv_fwd: row-planar source %d must be a whole tensor (cstride == C) of the call's width", s);
        rw = w;
      }
    }
    const int ow = A.out ? A.out_rp_w : 0, aw = A.aux ? A.aux_rp_w : 0;
    P.lay_out = lmn_lay_make(ow, A.Cout, A.out_cstride);
    P.lay_aux = lmn_lay_make(aw, A.Cout, A.aux_cstride);
    if (ow) { LMN_REQUIRE(ow > 0 && A.out_cstride == A.Cout && (rw == 0 || rw == ow), "conv_fwd: row-planar out must be a whole tensor of the call's width"); rw = ow; }
    if (aw) { LMN_REQUIRE(aw > 0 && A.aux_cstride == A.Cout && (rw == 0 || rw == aw), "conv_fwd: row-planar aux must be a whole tensor of the call's width"); rw = aw; }
    if (rw) {
      // (the plane stride of a row-planar tensor is its image width: a tensor marked with another width than the call's would be read
      //  with the wrong stride -- the width must BE the call's, not merely divide its pixel count)
      LMN_REQUIRE(A.ksize == 1 && A.stride == 1 && rw == A.Win && rw == A.Wout,
                  "conv_fwd: row-planar operands belong to 1x1 stride-1 calls over images of their own width (rp_w %d, call %d -> %d)", rw, A.Win, A.Wout);
    }
    P.rpw = rw;
    P.rp_magic = lmn_div_magic(rw);
  }
  {  // kernels index with 32-bit element offsets
    const int64_t lim = (1LL << 31) - 1;
    for (int s = 0; s < A.nsrc; ++s)
      LMN_REQUIRE((int64_t)A.B * A.Hin * A.Win * A.src[s].cstride <= lim, "conv_fwd: source %d larger than 2^31 elements", s);
    const int64_t ocs = A.out_cstride > A.aux_cstride ? A.out_cstride : A.aux_cstride;
    LMN_REQUIRE((int64_t)A.B * A.Hout * A.Wout * (ocs > A.res_cstride ? ocs : A.res_cstride) <= lim && (int64_t)A.B * A.Hout * A.Wout * A.Cout <= lim,
                "conv_fwd: output larger than 2^31 elements");
  }
  LMN_REQUIRE(A.drop_p >= 0.f && A.drop_p < 1.f, "conv_fwd: dropout p %f", A.drop_p);
  P.inv_keep_ep = 1.f / (1.f - A.drop_p);
  if (A.out) LMN_REQUIRE(A.out_cstride >= A.Cout && A.out_cstride % 4 == 0, "conv_fwd: out_cstride %d", A.out_cstride);
  if (A.residual) LMN_REQUIRE(A.res_cstride >= A.Cout && A.res_cstride % 4 == 0, "conv_fwd: res_cstride %d", A.res_cstride);
  if (A.aux) LMN_REQUIRE(A.aux_cstride >= A.Cout && A.aux_cstride % 4 == 0, "conv_fwd: aux_cstride %d", A.aux_cstride);
  switch (A.epilogue) {
    case LMN_EP_LINEAR: break;
    case LMN_EP_AFFINE_ACT: LMN_REQUIRE((A.p0 && A.p1) || A.fin.mode == LMN_FIN_BN, "conv_fwd: AFFINE_ACT needs p0,p1 (or fin)"); break;
    case LMN_EP_DGELU: LMN_REQUIRE(A.aux, "conv_fwd: DGELU needs aux"); break;
    case LMN_EP_BN_BWD1: LMN_REQUIRE(A.aux && A.p0 && A.p1 && A.p2 && A.p3 && A.stats, "conv_fwd: BN_BWD1 operands"); break;
    case LMN_EP_BN_BWD2:
      LMN_REQUIRE(A.aux && A.p0 && A.p1 && ((A.p2 && A.p3 && A.p4) || A.fin.mode == LMN_FIN_BN_BWD) && (!A.p5 == !A.p6), "conv_fwd: BN_BWD2 operands");
      break;
    case LMN_EP_SE_BWD: LMN_REQUIRE(A.aux && A.stats, "conv_fwd: SE_BWD operands"); break;
    case LMN_EP_LN_BWD:
      LMN_REQUIRE(A.aux && A.p0 && A.p6 && A.stats && A.out && A.ksize == 1 && A.Cout <= 48 && A.stats_mode == LMN_STATS_EP && A.out_rp_w == 0 && A.aux_rp_w == 0,
                  "conv_fwd: LN_BWD: 1x1 NHWC call with aux (LayerNorm input), p0 (gamma), p6 ((mean, rstd) table), stats [2][Cout], Cout <= 48 (got %d)", A.Cout);
      break;
    default: LMN_REQUIRE(false, "conv_fwd: epilogue %d", A.epilogue);
  }
  if (A.stats_mode == LMN_STATS_SUM_SQ) LMN_REQUIRE(A.stats, "conv_fwd: SUM_SQ needs stats");
  if (!A.transposed) {
    LMN_REQUIRE(A.Hout == (A.Hin + 2 * (A.ksize / 2) - A.ksize) / A.stride + 1 && A.Wout == (A.Win + 2 * (A.ksize / 2) - A.ksize) / A.stride + 1,
                "conv_fwd: out %dx%d inconsistent with in %dx%d k%d s%d", A.Hout, A.Wout, A.Hin, A.Win, A.ksize, A.stride);
  } else {
    LMN_REQUIRE(A.Hin == (A.Hout + 2 * (A.ksize / 2) - A.ksize) / A.stride + 1 && A.Win == (A.Wout + 2 * (A.ksize / 2) - A.ksize) / A.stride + 1,
                "conv_fwd(T): dy %dx%d inconsistent with dx %dx%d k%d s%d", A.Hin, A.Win, A.Hout, A.Wout, A.ksize, A.stride);
  }
  LMN_REQUIRE(A.fin.mode == LMN_FIN_NONE || A.fin.mode == LMN_FIN_BN || A.fin.mode == LMN_FIN_BN_BWD, "conv_fwd: fin.mode %d", A.fin.mode);
  if (A.fin.mode != LMN_FIN_NONE) {
    LMN_REQUIRE(A.fin.sums && A.fin.nrep >= 1 && A.fin.nrep <= 16 && A.fin.count > 0.f, "conv_fwd: fin needs sums, 1..16 slices, count > 0");
    if (A.fin.mode == LMN_FIN_BN)
      LMN_REQUIRE(A.epilogue == LMN_EP_AFFINE_ACT && A.fin.gamma && A.fin.beta && (!A.fin.about || A.fin.about != A.fin.rmean),
                  "conv_fwd: LMN_FIN_BN belongs to EP_AFFINE_ACT, needs gamma / beta, and `about` must not alias rmean");
    else
      LMN_REQUIRE(A.epilogue == LMN_EP_BN_BWD2 && A.fin.Ain, "conv_fwd: LMN_FIN_BN_BWD belongs to EP_BN_BWD2 and needs Ain");
  }
  LMN_REQUIRE(!A.stats_snap || (A.stats && A.stats_mode == LMN_STATS_SUM_SQ), "conv_fwd: stats_snap belongs to a SUM_SQ statistics pass");
  if (g_lmn_prof_on) {  // algorithmic cost of this launch (each HBM tensor once, MACs of the layer shape)
    int64_t cin = 0;
    for (int s = 0; s < A.nsrc; ++s) cin += A.src[s].C;
    const double opix = (double)A.B * A.Hout * A.Wout, ipix = (double)A.B * A.Hin * A.Win;
    const double macs = (A.transposed ? ipix : opix) * (double)cin * A.Cout * A.ksize * A.ksize;
    double by = ipix * cin + (A.out ? opix * A.Cout : 0.0);
    if (A.aux) by += opix * A.Cout;
    if (A.residual) by += opix * A.Cout;
    double macs2 = 0.0;
    if (A.chain.wpack) {   // chained second conv: its MACs and its output (+ aux); the first conv's output is written, never re-read
      macs2 = opix * (double)A.Cout * A.chain.Cout;
      by += opix * A.chain.Cout * (A.chain.aux ? 2.0 : 1.0);
    }
    lmn_prof_cost(2.0 * (macs + macs2), (A.act_dtype == LMN_BF16 ? 2.0 : 4.0) * by);
  }
  P.NCTT = (A.Cout + 15) / 16;
  P.ncls = (A.transposed && A.stride == 2) ? 4 : 1;
  P.gpi = 0;
  for (int c = 0; c < 4; ++c) {
    P.ng_c[c] = 0;
    P.Hc[c] = P.Wc[c] = 1;
  }
  for (int c = 0; c < P.ncls; ++c) {
    if (P.ncls == 4) {
      P.Hc[c] = (A.Hout - (c >> 1) + 1) >> 1;
      P.Wc[c] = (A.Wout - (c & 1) + 1) >> 1;
    } else {
      P.Hc[c] = A.Hout;
      P.Wc[c] = A.Wout;
    }
    if (P.Wc[c] < 1) P.Wc[c] = 1;  // empty class (Hout or Wout == 1): keep divisors sane
    const int64_t px = (int64_t)((P.ncls == 4) ? ((A.Hout - (c >> 1) + 1) >> 1) * (int64_t)((A.Wout - (c & 1) + 1) >> 1) : (int64_t)A.Hout * A.Wout);
    P.ng_c[c] = (int)((px + 15) / 16);
    P.gpi += P.ng_c[c];
  }
  const int64_t total_groups = (int64_t)A.B * P.gpi;
  LMN_REQUIRE(total_groups < (1LL << 30), "conv_fwd: too many pixel groups");
  int nct = P.NCTT >= 6 ? 6 : (P.NCTT == 5 ? 6 : P.NCTT);
  const int chunks = (P.NCTT + nct - 1) / nct;
  hipStream_t st = (hipStream_t)stream;
  // data gradient of a stride-2 3x3 conv: four parity classes in grid.z of the tile kernel (S2T)
  const bool s2t = A.transposed && A.stride == 2 && A.ksize == 3 && !(A.epilogue == LMN_EP_SE_BWD);
  if (!(A.transposed && A.stride == 2) || s2t) {
    // ---- LDS-tiled kernel.  1x1: the image is a flat row of H*W pixels.
    ConvParams T = P;
    lmn_conv_args_t& a = T.a;
    {
      static int prio_env = -1;   // LMN_CONV_PRIO: 0 off | 1, 2, 3 = distinct issue priorities per wave slot (lmn_wave_prio); 3x3 calls only unless +10
      if (prio_env < 0) { const char* e = getenv("LMN_CONV_PRIO"); prio_env = e ? atoi(e) : 0; }
      if (prio_env > 0) T.prio = (a.ksize == 3 || prio_env >= 10) ? prio_env % 10 : 0;
      static int stag3 = -1, stag1 = -1;   // LMN_CONV_STAGGER / LMN_CONV_STAGGER1: start stagger of the 3x3 / 1x1 tile kernels, units of 256 cycles per wave slot
      if (stag3 < 0) { const char* e = getenv("LMN_CONV_STAGGER"); stag3 = e ? atoi(e) : 0; }
      if (stag1 < 0) { const char* e = getenv("LMN_CONV_STAGGER1"); stag1 = e ? atoi(e) : 0; }
      const int stg = a.ksize == 3 ? stag3 : stag1;
      if (stg > 0) T.prio = 100 + stg;
    }
    if (a.ksize == 1) {
      a.Wout *= a.Hout; a.Win *= a.Hin; a.Hout = a.Hin = 1;
    }
    const int tpmax = (a.stride == 2 && !s2t) ? 64 : 128;
    const int gW = s2t ? (a.Wout + 1) / 2 : a.Wout, gH = s2t ? (a.Hout + 1) / 2 : a.Hout;  // tiled grid (S2T: class coordinates)
    int ncw = 0;  // > 0: M-split kernel with ncw cout tiles per wave
    int npg = 2;  // pixel groups per wave of the N-split kernel (4: 3x3 tiles of up to 256 pixels)
    // M-split from 6 cout tiles: 5 tiles split 2+1+1+1 over the four waves (24->72 3x3 at 176x176: 242 us, N-split 185 us)
    static int msplit_env = -1;   // LMN_MSPLIT_MIN: A/B runs
    if (msplit_env < 0) { const char* e = getenv("LMN_MSPLIT_MIN"); msplit_env = e ? atoi(e) : 6; }
    const int msplit_min = msplit_env;
    int tnct = nct, tchunks = chunks;  // N-split form: cout tiles per block (<= 3), cout chunks (grid.y)
    if (P.NCTT > 3) { tnct = 3; tchunks = (P.NCTT + 2) / 3; }
    if (a.ksize == 1) { T.TW = a.Wout < tpmax ? a.Wout : tpmax; T.TH = 1; }
    else {
      T.TW = gW <= 32 ? gW : 16;
      T.TH = tpmax / T.TW;
      if (T.TH > gH) T.TH = gH;
      if (T.TH < 1) T.TH = 1;
      static int nth_env = -1;   // LMN_CONV_TH: force the tile height of the N-split 3x3 kernel (A/B runs)
      if (nth_env < 0) { const char* e = getenv("LMN_CONV_TH"); nth_env = e ? atoi(e) : 0; }
      if (nth_env > 0 && nth_env < T.TH) T.TH = nth_env;
      // four pixel groups per wave (tiles of up to 256 pixels, conv_tile_kernel NPG = 4): stride-1 3x3 calls of the N-split form with at
      // most two cout tiles per block on maps that hold such tiles (LMN_CONV_NPG=2: off, A/B runs).  Measured: -5 ... -13 % on the layers
      // with 24+ channel sources alone, nothing inside the step (14.50 / 14.52 / 14.51 against 14.49 / 14.51 / 14.49 ms): not in the product
#ifdef LMN_CONV_NPG4   // experiment build (make npg4): the instances live in conv_tile_3x3g.hip, which the product library does not carry
      static int npg_env = -1;
      if (npg_env < 0) { const char* e = getenv("LMN_CONV_NPG"); npg_env = e ? atoi(e) : 4; }
      int maxkb0 = 0;
      for (int s_ = 0; s_ < a.nsrc; ++s_) maxkb0 = P.nkb[s_] > maxkb0 ? P.nkb[s_] : maxkb0;
      // (sources of <= 16 channels keep the 128-pixel tile: their window fits five blocks per CU WITH the weights in LDS, the larger one
      //  does not -- 12 -> 12 at 352^2: 54 -> 65 us; sources of 24+ channels: 24 -> 12 107 -> 100 us, 72 -> 24 125 -> 113, data gradients -5 ... -13 %)
      if (npg_env == 4 && !s2t && a.stride == 1 && !up2 && nth_env == 0 && P.NCTT < msplit_min && !ln && tnct <= 2 && gH >= 12 && gW >= 16 && maxkb0 >= 2) {
        npg = 4;
        T.TH = 256 / T.TW;
        if (T.TH > gH) T.TH = gH;
      }
#endif
    }
    // (LayerNorm on load: the N-split kernel reduces a pixel's statistics inside one staging round of <= 32 channels; wider sources
    //  take the M-split kernel, whose pre-pass handles any width)
    const bool ln_wide = ln && P.nkb[0] > 2;
    if ((P.NCTT >= msplit_min || ln_wide) && !s2t) {
      // Wide layer -> M-split kernel.  Wide layers sit on the small feature maps, where a 128-pixel tile times a
      // few cout chunks can leave most of the 256 CUs idle: pick (tile pixels, cout tiles per wave) by a cost model
      // -- rounds of 256 blocks x per-block MFMA work (+ staging, inflated by the halo for short 3x3 tiles).
      float best = 1e30f;
      int bTW = T.TW, bTH = T.TH;
      for (int cand = 0; cand < 16; ++cand) {
        int tw, th;
        if (a.ksize == 1) {
          static int tp_env = -1;   // LMN_CONVM_TP: force the 1x1 tile (pixels), A/B runs
          if (tp_env < 0) { const char* e = getenv("LMN_CONVM_TP"); tp_env = e ? atoi(e) : 0; }
          if (cand > 2) break;
          if (tp_env > 0 && (tpmax >> cand) != tp_env && tp_env <= a.Wout) continue;
          tw = tpmax >> cand; th = 1;
          if (tw > a.Wout) { if (cand) continue; tw = a.Wout; }
          if (cand && tw < 16) continue;
        } else {
          static int th_env = -1;   // LMN_CONVM_TH: force the 3x3 tile height (A/B runs)
          if (th_env < 0) { const char* e = getenv("LMN_CONVM_TH"); th_env = e ? atoi(e) : 0; }
          tw = T.TW; th = T.TH - cand;
          if (th < 1) break;
          if (th_env > 0 && th_env <= T.TH && th != th_env) continue;
        }
        const int ng = (tw * th + 15) / 16;
        const long tiles = (long)a.B * ((a.Wout + tw - 1) / tw) * ((a.Hout + th - 1) / th);
        for (int w = 1; w <= 2; ++w) {
          const long blk = tiles * ((P.NCTT + 4 * w - 1) / (4 * w));
          const float halo = a.ksize == 1 ? 1.f : (float)(th + 2) / th;
          const float cost = (float)((blk + 255) / 256) * (ng * (w + 0.3f * halo) + 1.f);
          if (cost < best * 0.999f) { best = cost; bTW = tw; bTH = th; ncw = w; }
        }
      }
      T.TW = bTW; T.TH = bTH;
      LMN_REQUIRE(ncw > 0, "conv_fwd: no tile candidate");
    }
    T.TP = T.TH * T.TW;
    T.NG = (T.TP + 15) / 16;
    LMN_REQUIRE(T.NG <= 4 * npg, "conv_fwd: tile of %d pixels", T.TP);
    const int st_in = a.transposed ? 1 : a.stride;
    T.XH = s2t ? T.TH + 1 : (T.TH - 1) * st_in + a.ksize;
    T.XW = s2t ? T.TW + 1 : (T.TW - 1) * st_in + a.ksize;
    int maxkb = 0;
    for (int s = 0; s < a.nsrc; ++s) maxkb = P.nkb[s] > maxkb ? P.nkb[s] : maxkb;
    T.CKB = maxkb < 2 ? maxkb : 2;
    if (ncw && a.ksize == 1) {  // M-split 1x1: K chunks of up to 8 K16 blocks while the window stays within ~36 KB of LDS
      static int ckb_cap = -1;
      if (ckb_cap < 0) { const char* e = getenv("LMN_CONVM_CKB"); ckb_cap = e ? atoi(e) : 8; }
      int c = maxkb < ckb_cap ? maxkb : ckb_cap;
      while (c > 2 && (size_t)T.TP * (c * 16 + 8) * sizeof(float) > 37 * 1024) --c;
      if (c > T.CKB) T.CKB = c;
    }
    if (ncw && a.ksize == 3 && !up2) {
      // M-split 3x3 on the small maps: a block is alone (or one of two) on its CU there, so nothing hides its staging round trips and
      // barriers -- one per 32-channel chunk, twelve per tile at 372 channels (phase: MFMA duty 54 % of a 149 us kernel whose 240 blocks
      // are one per CU).  K chunks of 4 / 8 K16 blocks while the grid leaves the LDS to at most two (one) blocks per CU.
      static int ckb3_env = -1;   // LMN_CONVM_CKB3: cap (2 = the 32-channel chunks of rounds 2-4, A/B runs)
      if (ckb3_env < 0) { const char* e = getenv("LMN_CONVM_CKB3"); ckb3_env = e ? atoi(e) : 8; }
      const long mblk = (long)a.B * ((gW + T.TW - 1) / T.TW) * ((gH + T.TH - 1) / T.TH) * ((P.NCTT + 4 * ncw - 1) / (4 * ncw));
      const size_t lds_cap = mblk <= 288 ? 144 * 1024 : (mblk <= 576 ? 72 * 1024 : 0);
      for (int c = 8; c > 2; c >>= 1) {
        if (c > ckb3_env || c > maxkb) continue;      // (a chunk size the layer's sources never fill)
        const bool bf0 = a.mma_dtype == LMN_BF16;
        const size_t need = ((size_t)T.XH * T.XW * (c * (bf0 ? 8 : 16) + (bf0 ? 4 : 8)) + (2 + 9) * 4 * ncw * 16) * sizeof(float);
        if (need <= lds_cap) { T.CKB = c; break; }
      }
    }
    // LDS pixel stride: conflict-free ds_read_b128 for 16 pixels st_in apart (brute-forced over the b128 lane groups):
    // +8 floats at unit stride, +4 at stride 2 (PMC: 0.5 conflict cycles per LDS cycle with +4 at unit stride)
    T.CS = T.CKB * 16 + (st_in == 1 ? 8 : 4);
    const bool bf = a.mma_dtype == LMN_BF16;
    const int pm = bf ? (a.act_dtype == LMN_BF16 ? 2 : 1) : 0;
    if (bf) T.CS = T.CKB * 8 + 4;  // dwords: 8 per K16 block of 4-bf16 fragments, +4: conflict-free ds_read_b64 (brute-forced)
    T.tiles_x = (gW + T.TW - 1) / T.TW;
    T.tiles_y = (gH + T.TH - 1) / T.TH;
    T.total_tiles = a.B * T.tiles_x * T.tiles_y;
    // strided tile assignment: -3 % over the conv family at batch 8 (contiguous ranges leave some blocks with twice the
    // tiles of others: 1936 tiles on 1280 blocks); SE_BWD keeps contiguous ranges (per-image sums live in registers)
    T.strided = a.epilogue == LMN_EP_SE_BWD ? 0 : 1;
    T.mTW = (uint32_t)((1ull << 32) / (uint32_t)T.TW + 1);
    T.mXW = (uint32_t)((1ull << 32) / (uint32_t)T.XW + 1);
    LMN_REQUIRE(T.XH * T.XW < 65536, "conv_fwd: window too large");
    size_t shmem = ((size_t)T.XH * T.XW * T.CS + (2 + 9) * tnct * 16) * sizeof(float);  // window, statistics, epilogue parameters
    // LDS-staged weights (3x3 tile kernel) while the block stays within LMN_CONV_WL_KB of LDS (default 32: 5 blocks per CU; level-0
    // 12 -> 12: 62 -> 52 us on cold operands; at 48 KB the 24-channel layers lose more to occupancy than they gain)
    static int wl_kb = -1;
    if (wl_kb < 0) { const char* e = getenv("LMN_CONV_WL_KB"); wl_kb = e ? atoi(e) : 32; }
    const size_t wl_bytes = (size_t)9 * T.CKB * tnct * (a.mma_dtype == LMN_BF16 ? 128 : 256) * sizeof(float);
    const bool wlk = a.ksize == 3 && !s2t && shmem + wl_bytes <= (size_t)wl_kb * 1024;
    if (wlk) shmem += wl_bytes;
    LMN_REQUIRE(ncw || shmem <= 64 * 1024, "conv_fwd: LDS window %zu B", shmem);
    int blocks = T.total_tiles;
    int maxb = 1280 / tchunks > 256 ? 1280 / tchunks : 256;  // ~5 resident blocks per CU: one round of persistent blocks
    if (npg == 4) {   // the larger window: as many blocks as the LDS of a CU holds (at most five)
      int per_cu = (int)((160 * 1024) / (shmem + 512));
      per_cu = per_cu > 5 ? 5 : (per_cu < 1 ? 1 : per_cu);
      maxb = 256 * per_cu / tchunks > 256 ? 256 * per_cu / tchunks : 256;
    }
    if (s2t) maxb = 320 / tchunks > 64 ? 320 / tchunks : 64;          // x 4 classes in grid.z
    {
      static int maxb_env = -1;   // LMN_CONV_MAXB: cap on the persistent blocks of a 3x3 call (occupancy experiments: 256 = one block per CU)
      if (maxb_env < 0) { const char* e = getenv("LMN_CONV_MAXB"); maxb_env = e ? atoi(e) : 0; }
      if (maxb_env > 0 && a.ksize == 3 && !s2t) maxb = maxb_env / tchunks > 1 ? maxb_env / tchunks : 1;
    }
    if (blocks > maxb) blocks = maxb;
    // epilogue instance (see the kernel): 0 plain, 2 LINEAR+SUM_SQ, 3 BN_BWD1, 4 BN_BWD2, 5 SE_BWD, 1 everything else
    int ek = 1;
    if (a.epilogue <= LMN_EP_AFFINE_ACT && a.stats_mode == LMN_STATS_NONE) ek = 0;
    else if (a.drop_p > 0.f) ek = 1;
    else if (a.epilogue == LMN_EP_LINEAR && a.stats_mode == LMN_STATS_SUM_SQ) ek = 2;
    else if (a.epilogue == LMN_EP_BN_BWD1 && a.stats_mode == LMN_STATS_EP) ek = 3;
    else if (a.epilogue == LMN_EP_BN_BWD2 && a.stats_mode == LMN_STATS_NONE) ek = 4;
    else if (a.epilogue == LMN_EP_SE_BWD && a.stats_mode == LMN_STATS_EP) ek = 5;
    else if (a.epilogue == LMN_EP_LN_BWD) ek = 6;
    // deterministic mode: the statistics go to one slot per block (SE_BWD: per wave) of the stream's scratch, summed in fixed order
    // by lmn_det_sum right after the launch into slice 0 of the caller's buffer
    int det_ns = 0;
    int64_t det_sz = 0;
    auto det_prep = [&](int nbx) -> int {
      if (!g_lmn_det || !a.stats || a.stats_mode == LMN_STATS_NONE) return 0;
      const bool sebwd = a.epilogue == LMN_EP_SE_BWD;
      det_ns = sebwd ? nbx * 4 : nbx;
      det_sz = sebwd ? (int64_t)a.B * a.Cout : 2 * (int64_t)a.Cout;
      lmn_det_begin(st);
      T.det_stats = lmn_det_slots(st, (size_t)det_ns * det_sz);
      LMN_REQUIRE(T.det_stats, "conv_fwd: deterministic mode: no scratch for %d x %lld statistics slots", det_ns, (long long)det_sz);
      return 0;
    };
    auto det_finish = [&]() { if (T.det_stats) lmn_det_sum(st, T.det_stats, det_ns, det_sz, a.stats); };
    if (ncw) {
      LMN_REQUIRE(a.epilogue != LMN_EP_LN_BWD, "conv_fwd: LN_BWD is an N-split epilogue (Cout <= 48)");
      const int mchunks = (P.NCTT + 4 * ncw - 1) / (4 * ncw);
      int mblocks = T.total_tiles;
      const int mmax = 2048 / mchunks > 256 ? 2048 / mchunks : 256;
      if (mblocks > mmax) mblocks = mmax;
      const dim3 mgrid(mblocks, mchunks);
      {
        // LDS-DMA form (conv_dmaM.hip) for the plain wide 3x3 stride-1 calls: one NHWC fp32 source without on-load transform,
        // bias / residual / SUM_SQ statistics only (LMN_CONV_DMA bit 2).  One cout tile per wave (64 output channels per block).
        if (g_conv_dma_mode < 0) { const char* e = getenv("LMN_CONV_DMA"); g_conv_dma_mode = e ? atoi(e) : 7; }
        const lmn_src_t& s0 = a.src[0];
        const bool dm_ok = (g_conv_dma_mode & 4) && a.ksize == 3 && a.stride == 1 && !up2 && !ln && pm == 0 && a.nsrc == 1 && s0.flags == 0 && !s0.scale && s0.rp_w == 0 &&
                           s0.C % 4 == 0 && s0.cstride % 4 == 0 && a.epilogue == LMN_EP_LINEAR && (ek == 0 || ek == 2) && a.drop_p == 0.f && !a.aux && a.fin.mode == LMN_FIN_NONE &&
                           !a.out_rp_w && a.Hin == a.Hout && a.Win == a.Wout && lmn_conv_dmaM_fits(T.XH, T.XW) &&
                           (int64_t)a.B * a.Hin * a.Win * s0.cstride * 4 < 0x7fffffffLL && (int64_t)9 * P.NKB * P.NCTT * 1024 < 0x7fffffffLL;
        if (dm_ok) {
          const int dchunks = (P.NCTT + 3) / 4;
          int dblocks = T.total_tiles;
          const int dmax = 2048 / dchunks > 256 ? 2048 / dchunks : 256;
          if (dblocks > dmax) dblocks = dmax;
          if (int rc = det_prep(dblocks)) return rc;
          LMN_REQUIRE(lmn_launch_conv_dmaM(T, dim3(dblocks, dchunks), st, ek) == 0, "conv_fwd: no LDS-DMA M-split instance for epilogue %d", ek);
          det_finish();
          return lmn_launch_status("conv_fwd(dmaM)");
        }
      }
      if (int rc = det_prep(mblocks)) return rc;
      const size_t msh = ((size_t)T.XH * T.XW * T.CS + (2 + 9) * 4 * ncw * 16 + (ln ? 2 * T.XH * T.XW : 0)) * sizeof(float);
      lmn_launch_conv_tileM(T, mgrid, msh, st, a.ksize == 1 ? 1 : 9, ncw, pm, ek, a.ksize == 1 && T.rpw != 0, ln, up2);
      det_finish();
      return lmn_launch_status("conv_fwd(tileM)");
    }
    if (s2t) {
      const dim3 zgrid(blocks, tchunks, 4);
      if (int rc = det_prep(blocks * 4)) return rc;
      lmn_launch_conv_tile_s2t(T, zgrid, shmem, st, tnct, pm, ek);
      det_finish();
      return lmn_launch_status("conv_fwd(tile, stride-2 data gradient)");
    }
    {
      // LDS-DMA double-buffered kernel (conv_dma3.hip) for the small-channel 3x3 stride-1 calls: one plain NHWC fp32 source of 12 / 24
      // channels, Cout <= 32, bias / residual / SUM_SQ statistics only, maps that give every CU tiles to pipeline (levels 0-1 at batch 8)
      if (g_conv_dma_mode < 0) { const char* e = getenv("LMN_CONV_DMA"); g_conv_dma_mode = e ? atoi(e) : 7; }   // LMN_CONV_DMA: bit 0 = 3x3 kernel, bit 1 = 1x1 kernel; 0 = off (conv_tile_kernel everywhere, A/B runs)
      const int dma_env = g_conv_dma_mode & 1;
      const lmn_src_t& s0 = a.src[0];
      const int64_t dtiles = (int64_t)a.B * ((a.Wout + 15) / 16) * ((a.Hout + 7) / 8);
      const bool dma_ok = dma_env != 0 && a.ksize == 3 && a.stride == 1 && !up2 && !ln && pm == 0 && a.nsrc == 1 && s0.flags == 0 && !s0.scale && s0.rp_w == 0 &&
                          (s0.C == 12 || s0.C == 24) && P.NCTT <= 2 && a.epilogue == LMN_EP_LINEAR && (a.stats_mode == LMN_STATS_NONE || a.stats_mode == LMN_STATS_SUM_SQ) &&
                          a.drop_p == 0.f && !a.aux && a.fin.mode == LMN_FIN_NONE && !a.out_rp_w && a.Hin == a.Hout && a.Win == a.Wout && dtiles >= g_conv_dma_min_tiles && dtiles < (1 << 30) &&
                          (int64_t)a.B * a.Hin * a.Win * s0.cstride * 4 < 0x7fffffffLL && (int64_t)a.B * a.Hout * a.Wout * a.out_cstride * 4 < 0x7fffffffLL &&
                          (!a.residual || (int64_t)a.B * a.Hout * a.Wout * a.res_cstride < 0x7fffffffLL);
      if (dma_ok) {
        ConvParams D = T;
        D.tiles_x = (a.Wout + 15) / 16; D.tiles_y = (a.Hout + 7) / 8; D.total_tiles = (int)dtiles;
        const size_t lds = lmn_conv_dma3_lds(s0.C, P.NCTT);
        int bpc = (int)((160 * 1024) / (lds + 256));
        bpc = bpc > 4 ? 4 : (bpc < 1 ? 1 : bpc);
        static int bpc3_env = -1;   // LMN_DMA3_BPC: cap on the resident blocks per CU of the 3x3 LDS-DMA kernel (A/B runs)
        if (bpc3_env < 0) { const char* e = getenv("LMN_DMA3_BPC"); bpc3_env = e ? atoi(e) : 0; }
        if (bpc3_env > 0 && bpc > bpc3_env) bpc = bpc3_env;
        int dblocks = 256 * bpc;
        if (dblocks > D.total_tiles) dblocks = D.total_tiles;
        if (int rc = det_prep(dblocks)) return rc;
        D.det_stats = T.det_stats;
        LMN_REQUIRE(lmn_launch_conv_dma3(D, dblocks, st) == 0, "conv_fwd: no LDS-DMA instance for C=%d, %d cout tiles", s0.C, P.NCTT);
        det_finish();
        return lmn_launch_status("conv_fwd(dma3)");
      }
    }
    if (a.ksize == 1) {
      // LDS-DMA streaming kernel (conv_dma1.hip) for the HBM-bound 1x1 convs of ReparamConv at levels 0-1 (expand conv, pointwise +
      // shortcut, SE-gradient conv, folded data gradient, and the chained pairs of lmn_conv_chain_t): dma1_select decides
      Dma1Sel S1;
      if (dma1_select(A, S1)) {
        ConvParams D = T;
        D.total_tiles = (int)((int64_t)A.B * A.Hin * A.Win / S1.tp);
        D.NCTT2 = S1.nct2;
        D.lay_out2 = lmn_lay_make(A.chain.out_rp_w, A.chain.Cout > 0 ? A.chain.Cout : 4, A.chain.out_cstride);
        D.lay_aux2 = lmn_lay_make(A.chain.aux_rp_w, A.chain.Cout > 0 ? A.chain.Cout : 4, A.chain.aux_cstride);
        if (S1.nct2) {
          LMN_REQUIRE((!A.chain.out_rp_w || A.chain.out_rp_w == A.Win) && (!A.chain.aux_rp_w || A.chain.aux_rp_w == A.Win), "conv_fwd: row-planar tensors of the chained conv belong to images of the call's width");
          if (A.chain.out_rp_w || A.chain.aux_rp_w) { D.rpw = A.Win; D.rp_magic = lmn_div_magic(A.Win); }
        }
        const size_t lds = lmn_conv_dma1_lds(S1.ks[0], S1.ks[1], S1.ks[2], S1.aq, S1.nct, S1.tp, S1.nct2);
        int bpc = (int)((160 * 1024) / (lds + 256));
        bpc = bpc > 4 ? 4 : (bpc < 1 ? 1 : bpc);
        static int bpc1_env = -1;   // LMN_DMA1_BPC: cap on the resident blocks per CU of the 1x1 LDS-DMA kernel (A/B runs)
        if (bpc1_env < 0) { const char* e = getenv("LMN_DMA1_BPC"); bpc1_env = e ? atoi(e) : 0; }
        if (bpc1_env > 0 && bpc > bpc1_env) bpc = bpc1_env;
        int dblocks = 256 * bpc;
        if (dblocks > D.total_tiles) dblocks = D.total_tiles;
        if (int rc = det_prep(dblocks)) return rc;
        D.det_stats = T.det_stats;
        LMN_REQUIRE(lmn_launch_conv_dma1(D, S1.ks[0], S1.ks[1], S1.ks[2], S1.aq, S1.nct, S1.mode, S1.gs, S1.nct2, S1.mode2, dblocks, st) == 0, "conv_fwd: LDS-DMA 1x1 instance table out of step");
        det_finish();
        return lmn_launch_status("conv_fwd(dma1)");
      }
    }
    const dim3 grid(blocks, tchunks);
    LMN_REQUIRE(a.epilogue != LMN_EP_LN_BWD || (tchunks == 1 && a.ksize == 1 && T.rpw == 0 && !ln), "conv_fwd: LN_BWD needs every cout in one block (N-split 1x1 NHWC call)");
    if (int rc = det_prep(blocks)) return rc;
    if (a.ksize == 1) lmn_launch_conv_tile_1x1(T, grid, shmem, st, tnct, pm, ek, T.rpw != 0, ln);
#ifdef LMN_CONV_NPG4
    else if (npg == 4) lmn_launch_conv_tile_3x3g(T, grid, shmem, st, tnct, pm, ek, wlk);
#endif
    else lmn_launch_conv_tile_3x3(T, grid, shmem, st, tnct, pm, ek, wlk, up2);
    det_finish();
    return lmn_launch_status("conv_fwd(tile)");
  }
  LMN_REQUIRE(false, "conv_fwd: the data gradient of a stride-2 conv is implemented for 3x3 kernels (got ksize %d, epilogue %d)", A.ksize, A.epilogue);
  return -1;
}

int lmn_reparam_fold(const float* hstats, const float* mean, const float* rstd, const float* A, float count, int batch_stats,
                     const float* w_expand, const float* b_expand, const float* w_shortcut, int E, int rows, int cin_w, int cred,
                     int cout_w, float* wpack, float* kbias, float* coef, float* dgamma, float* dbeta, int dtype,
                     lmn_stream_t stream) {
  LMN_REC(lmn_reparam_fold(hstats, mean, rstd, A, count, batch_stats, w_expand, b_expand, w_shortcut, E, rows, cin_w, cred, cout_w,
                           wpack, kbias, coef, dgamma, dbeta, dtype, stream));
  LMN_REQUIRE(hstats && mean && rstd && A && w_expand && b_expand && w_shortcut && wpack && kbias && coef && count > 0.f,
              "reparam_fold: null pointer");
  LMN_REQUIRE(E > 0 && rows > 0 && rows % 4 == 0 && cin_w > 0 && cin_w <= rows && cout_w > 0 && cout_w <= cred && cred % 4 == 0,
              "reparam_fold: E=%d rows=%d cin=%d cred=%d cout=%d", E, rows, cin_w, cred, cout_w);
  LMN_REQUIRE(dtype == LMN_F32 || dtype == LMN_BF16, "reparam_fold: dtype %d", dtype);
  FoldParams P;
  P.hstats = hstats; P.mean = mean; P.rstd = rstd; P.A = A; P.we = w_expand; P.be = b_expand; P.wsc = w_shortcut;
  P.wpack = wpack; P.kbias = kbias; P.coef = coef; P.dgamma = dgamma; P.dbeta = dbeta; P.count = count; P.batch_stats = batch_stats;
  P.E = E; P.rows = rows; P.cinw = cin_w; P.cred = cred; P.coutw = cout_w; P.bf16 = dtype == LMN_BF16;
  const int ntiles = ((E + 15) / 16 + (rows + 15) / 16 + (cred + 15) / 16) * ((rows + 15) / 16);
  LMN_REQUIRE((size_t)35 * E * sizeof(float) <= 64 * 1024, "reparam_fold: E = %d too wide for the block's scratch", E);
  P.prio = lmn_prio_level((hipStream_t)stream);
  LMN_LAUNCH(reparam_fold_kernel, dim3(ntiles + 1), dim3(256), (size_t)35 * E * sizeof(float), (hipStream_t)stream, P);
  return lmn_launch_status("reparam_fold");
}

int lmn_reparam_wfin(const float* R, const float* M, const float* m, const float* coef, const float* hstats, const float* w_expand,
                     const float* b_expand, float count, int E, int rows, int cin_w, float* dW, float* db, const lmn_se_params_t* se,
                     lmn_stream_t stream) {
  lmn_se_params_t sp;
  memset(&sp, 0, sizeof(sp));
  if (se) sp = *se;
  const bool has_se = se != nullptr;
  if (g_lmn_rec) lmn_rec_push([=]() -> int { return lmn_reparam_wfin(R, M, m, coef, hstats, w_expand, b_expand, count, E, rows, cin_w, dW, db, has_se ? &sp : nullptr, stream); }, "lmn_reparam_wfin(");
  LMN_REQUIRE(R && M && m && coef && hstats && w_expand && b_expand && dW && E > 0 && rows >= cin_w && cin_w > 0 && count > 0.f,
              "reparam_wfin: bad argument");
  int nse = 0;
  if (has_se) {
    LMN_REQUIRE(sp.dvec && sp.gsum && sp.hidden && sp.dw1 && sp.db1 && sp.dw2 && sp.db2 && sp.B > 0 && sp.E > 0 && sp.R > 0, "reparam_wfin: bad squeeze-excite argument");
    nse = (int)lmn_cdiv(2 * (int64_t)sp.E * sp.R + sp.E + sp.R, 256);
  }
  const int nw = (int)lmn_cdiv((int64_t)E * (cin_w + 1), 16);
  LMN_LAUNCH(reparam_wfin_kernel, dim3(nw + nse), dim3(256), 0, (hipStream_t)stream, R, M, m, coef,
             hstats, w_expand, b_expand, count, E, rows, cin_w, dW, db, sp, nw);
  return lmn_launch_status("reparam_wfin");
}

}  // extern "C"
