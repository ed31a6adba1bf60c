// Prototype (round 6): 3x3 stride-1 conv, NHWC fp32, on v_mfma_f32_16x16x4_f32 with the input window of tile
// t+1 filled by LDS-DMA (buffer_load_dwordx4 ... lds) into a SECOND LDS buffer while the MFMA loop of tile t runs -- zero staging
// VGPRs, one barrier per tile, counted s_waitcnt vmcnt(N) (never 0 in the loop), 2-3 blocks per CU.  Stand-alone: builds its own
// inputs, checks sampled outputs against a double-precision CPU restatement, times the launch on COLD operands (rotating sets).
//   hipcc --offload-arch=gfx950 -O3 -o conv3_dma conv3_dma.hip ;  ./conv3_dma [cin cout [H [B]]]
// LDS image of a window: [window pixel][PS chunks of 16 B], natural channel order, PS = chunks per pixel rounded up to an ODD count
// (16 pixels 4*odd dwords apart fall on 16 distinct bank quads: conflict-free b64 / b128 reads); the pad chunk is written by lanes
// whose source offset is out of range (buffer bounds check returns 0 -> the DMA writes zeros), and so is the zero padding of the conv.
// K mapping: slice s of the MFMA takes channel q*KS + s from lane group q (KS = CIN/4 slices), so a lane's operands of all slices
// are CONTIGUOUS in the natural-order image (no transposing commit); the weights are packed to match.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <math.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr unsigned OOB = 0x80000000u;

struct Params {
  unsigned long long* tk;
  const float* x; float* y; const float* wpack; const float* bias;
  int B, H, W, Cout;
  int tiles_x, tiles_y, total_tiles;
  unsigned xbytes, ybytes;
};

__device__ __forceinline__ i32x4 make_rsrc(const void* p, unsigned bytes) {
  const unsigned long long a = (unsigned long long)p;
  i32x4 r;
  r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
  r.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));   // stride 0
  r.z = __builtin_amdgcn_readfirstlane((int)bytes);
  r.w = 0x00020000;
  return r;
}

// one LDS-DMA piece: 64 lanes x 16 B from base + voff (out of range: zeros) to LDS[ldsaddr + lane * 16]
__device__ __forceinline__ void glds16(unsigned ldsaddr, unsigned voff, i32x4 rsrc) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(ldsaddr), "v"(voff), "s"(rsrc) : "memory");
}

template <int KS> struct OpRegs { float v[KS]; };
template <int KS> __device__ __forceinline__ OpRegs<KS> lds_ops(const float* p) {   // KS contiguous floats at p (alignment KS * 4 bytes mod 16)
  OpRegs<KS> r;
  if constexpr (KS % 4 == 0) {
#pragma unroll
    for (int i = 0; i < KS / 4; ++i) { const f32x4 t = *reinterpret_cast<const f32x4*>(p + 4 * i); r.v[4 * i] = t[0]; r.v[4 * i + 1] = t[1]; r.v[4 * i + 2] = t[2]; r.v[4 * i + 3] = t[3]; }
  } else if constexpr (KS % 2 == 0) {
#pragma unroll
    for (int i = 0; i < KS / 2; ++i) { const f32x2 t = *reinterpret_cast<const f32x2*>(p + 2 * i); r.v[2 * i] = t[0]; r.v[2 * i + 1] = t[1]; }
  } else {
#pragma unroll
    for (int i = 0; i < KS; ++i) r.v[i] = p[i];
  }
  return r;
}

// TH x 16 output pixels per tile, 4 waves, wave wv owns tile rows wv, wv + 4, ... (NPG = TH / 4 pixel groups of 16)
#ifndef PF
#define PF 1
#endif
#ifndef IL
#define IL 1
#endif
template <int CIN, int NCT, int TH, int BPC>
__global__ __launch_bounds__(256, BPC) void conv3_dma_kernel(const Params P) {
  constexpr int TW = 16, XW = TW + 2, XH = TH + 2, NPG = TH / 4;
  constexpr int CQ = CIN / 4, PS = (CQ & 1) ? CQ : CQ + 1, KS = CIN / 4;
  constexpr int NCH = XH * XW * PS;                 // 16-byte chunks of one window image
  constexpr int NK = (NCH + 255) / 256;             // LDS-DMA pieces per thread and tile
  constexpr int WFL = 9 * NCT * 64 * KS;            // floats of the packed weights
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const s_w = smem + 2 * NCH * 4;            // [tap][ct][lane][KS]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, n = lane & 15;
  const i32x4 rx = make_rsrc(P.x, P.xbytes), ry = make_rsrc(P.y, P.ybytes);
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem;   // LDS byte address of the window buffers

  // per-thread chunk descriptors (tile-independent): window row / column and byte offset of the chunk inside its pixel
  int drc[NK];   // r << 16 | c << 8 | f ; -1: pad chunk or past the image
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    const int i = k * 256 + tid;
    const int wp = i / PS, f = i - wp * PS;
    const int r = wp / XW, c = wp - r * XW;
    drc[k] = (i < NCH && f < CQ) ? (r << 16 | c << 8 | f) : -1;
  }
  unsigned svoff[NK];   // source offsets of the next tile's pieces
  auto stage_addr = [&](int tile) __attribute__((always_inline)) {
    const int b = tile / (P.tiles_x * P.tiles_y), tt = tile - b * P.tiles_x * P.tiles_y;
    const int ty = tt / P.tiles_x, tx = tt - ty * P.tiles_x;
    const int wy0 = ty * TH - 1, wx0 = tx * TW - 1;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const int d = drc[k];
      const int iy = wy0 + (d >> 16), ix = wx0 + ((d >> 8) & 255);
      const bool ok = d >= 0 && (unsigned)iy < (unsigned)P.H && (unsigned)ix < (unsigned)P.W;
      svoff[k] = ok ? (unsigned)(((b * P.H + iy) * P.W + ix) * CIN + (d & 255) * 4) * 4u : OOB;
    }
  };
  auto stage_piece = [&](int k, int buf) __attribute__((always_inline)) {
    const unsigned base = lds0 + (unsigned)buf * (NCH * 16) + (unsigned)wv * 1024;
    if ((k + 1) * 256 <= NCH || k * 256 + tid < NCH) glds16(base + (unsigned)k * 4096, svoff[k], rx);   // (last piece: lanes past the image stay out)
  };
  auto stage = [&](int tile, int buf) __attribute__((always_inline)) {
    stage_addr(tile);
#pragma unroll
    for (int k = 0; k < NK; ++k) stage_piece(k, buf);
  };

  const int t0 = blockIdx.x, tstep = gridDim.x;
  if (t0 < P.total_tiles) stage(t0, 0);
  for (int i = tid; i < WFL / 4; i += 256) *reinterpret_cast<f32x4*>(s_w + 4 * i) = *reinterpret_cast<const f32x4*>(P.wpack + 4 * i);
  f32x4 bias4[NCT];
#pragma unroll
  for (int c = 0; c < NCT; ++c) {
    const int co = c * 16 + q * 4;
    bias4[c] = co < P.Cout ? *reinterpret_cast<const f32x4*>(P.bias + co) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int c = 0; c < NCT; ++c) asm volatile("" : "+v"(bias4[c]));   // a use HERE: hipcc's own wait for the bias load must not land behind the next tile's DMA
  __builtin_amdgcn_s_barrier();

#ifdef TIMING
  unsigned long long tks[4] = {0, 0, 0, 0}, tk0 = __builtin_amdgcn_s_memtime(), tka, tkb;
#define TK(i) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tkb = __builtin_amdgcn_s_memtime(); tks[i] += tkb - tka; tka = tkb; } while (0)
#else
#define TK(i)
#endif
  int it = 0;
  for (int tile = t0; tile < P.total_tiles; tile += tstep, ++it) {
    const int cur = it & 1;
#ifdef TIMING
    tka = __builtin_amdgcn_s_memtime();
#endif
    if (it > 0) {
      // the NPG * NCT stores of the previous tile are the wave's youngest vector-memory operations; everything older -- the LDS-DMA
      // pieces of THIS tile, issued a whole MFMA loop ago -- must have landed.  Then the barrier: every wave's pieces are in, and
      // every wave has left the MFMA loop that read the other buffer.
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPG * NCT) : "memory");
      __builtin_amdgcn_s_barrier();
    }
    TK(0);
    const bool has_next = tile + tstep < P.total_tiles;   // (block-uniform)
    if (IL) { if (has_next) stage_addr(tile + tstep); }
    else if (has_next) stage(tile + tstep, cur ^ 1);
    TK(1);
    const int b = tile / (P.tiles_x * P.tiles_y), tt = tile - b * P.tiles_x * P.tiles_y;
    const int tyi = tt / P.tiles_x, txi = tt - tyi * P.tiles_x;
    const float* XS = smem + cur * (NCH * 4);
    f32x4 acc[NPG][NCT];
#pragma unroll
    for (int g = 0; g < NPG; ++g)
#pragma unroll
      for (int c = 0; c < NCT; ++c) acc[g][c] = bias4[c];
    const float* xb = XS + (wv * XW + n) * (PS * 4) + q * KS;     // group g adds 4 * XW pixels
    const float* wb = s_w + lane * KS;
    // operands of tap t+1 are requested BEFORE the MFMAs of tap t (sched_barrier: hipcc otherwise reads, waits and multiplies per tap,
    // an exposed LDS round trip nine times per tile at two waves per SIMD)
    OpRegs<KS> w[2][NCT], x[2][NPG];
#pragma unroll
    for (int c = 0; c < NCT; ++c) w[0][c] = lds_ops<KS>(wb + c * 64 * KS);
#pragma unroll
    for (int g = 0; g < NPG; ++g) x[0][g] = lds_ops<KS>(xb + (4 * g * XW) * (PS * 4));
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      constexpr int dummy = 0; (void)dummy;
      const int cb = tap & 1, nb = cb ^ 1;
      if (tap < 8) {
        const int ty = (tap + 1) / 3, tx = (tap + 1) - ty * 3;
#pragma unroll
        for (int c = 0; c < NCT; ++c) w[nb][c] = lds_ops<KS>(wb + ((tap + 1) * NCT + c) * 64 * KS);
#pragma unroll
        for (int g = 0; g < NPG; ++g) x[nb][g] = lds_ops<KS>(xb + ((4 * g + ty) * XW + tx) * (PS * 4));
      }
      if (PF) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int c = 0; c < NCT; ++c)
#pragma unroll
          for (int g = 0; g < NPG; ++g) acc[g][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[cb][c].v[s], x[cb][g].v[s], acc[g][c], 0, 0, 0);
      if (PF) __builtin_amdgcn_sched_barrier(0);
      if (IL) {   // the next tile's DMA pieces ride behind the MFMAs of the first taps: issued into a queue that is never backed up
#pragma unroll
        for (int k = 0; k < NK; ++k)
          if (k * 9 / NK == tap && has_next) stage_piece(k, cur ^ 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    TK(2);
    // epilogue: lane holds channels c*16 + q*4 .. +3 of pixel (row wv + 4g, column n) -- unconditional stores (dead lanes out of range)
#pragma unroll
    for (int g = 0; g < NPG; ++g) {
      const int oy = tyi * TH + wv + 4 * g, ox = txi * TW + n;
      const bool pok = oy < P.H && ox < P.W;
#pragma unroll
      for (int c = 0; c < NCT; ++c) {
        const int co = c * 16 + q * 4;
        const unsigned voff = (pok && co < P.Cout) ? (unsigned)(((b * P.H + oy) * P.W + ox) * P.Cout + co) * 4u : OOB;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, acc[g][c]), __builtin_amdgcn_make_buffer_rsrc((void*)P.y, 0, (int)P.ybytes, 0x00020000), (int)voff, 0, 0);
      }
    }
    (void)ry;
    TK(3);
  }
#ifdef TIMING
  if (lane == 0) {
    unsigned long long* o = P.tk + (size_t)(blockIdx.x * 4 + wv) * 8;
    o[0] = tks[0]; o[1] = tks[1]; o[2] = tks[2]; o[3] = tks[3]; o[4] = __builtin_amdgcn_s_memtime() - tk0; o[5] = it;
  }
#endif
}

// ---------------------------------------------------------------------------------------------------------------- host
static float frand(unsigned& s) { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 32768.f - 1.f; }

template <int CIN, int NCT, int TH, int BPC>
static int run(int Cout, int H, int W, int B, int nset, int iters) {
  constexpr int CQ = CIN / 4, PS = (CQ & 1) ? CQ : CQ + 1, KS = CIN / 4;
  constexpr int NCH = (TH + 2) * 18 * PS;
  const size_t xn = (size_t)B * H * W * CIN, yn = (size_t)B * H * W * Cout;
  std::vector<float> hx(xn), hw((size_t)Cout * CIN * 9), hb(Cout), hy(yn);
  unsigned s = 12345u;
  for (auto& v : hx) v = frand(s);
  for (auto& v : hw) v = frand(s) * 0.2f;
  for (auto& v : hb) v = frand(s);
  // packed weights [tap][ct][lane = (q, m)][s] = W[cout = ct*16 + m][cin = q*KS + s][tap]   (torch layout W[cout][cin][ky][kx])
  std::vector<float> hp((size_t)9 * NCT * 64 * KS, 0.f);
  for (int tap = 0; tap < 9; ++tap)
    for (int ct = 0; ct < NCT; ++ct)
      for (int l = 0; l < 64; ++l)
        for (int sl = 0; sl < KS; ++sl) {
          const int q = l >> 4, m = l & 15, co = ct * 16 + m, ci = q * KS + sl;
          hp[(((size_t)tap * NCT + ct) * 64 + l) * KS + sl] = co < Cout ? hw[((size_t)co * CIN + ci) * 9 + tap] : 0.f;
        }
  std::vector<float*> dx(nset), dy(nset);
  for (int i = 0; i < nset; ++i) {
    CK(hipMalloc(&dx[i], xn * 4)); CK(hipMalloc(&dy[i], yn * 4));
    CK(hipMemcpy(dx[i], hx.data(), xn * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dy[i], 0xFF, yn * 4));
  }
  float *dw, *db;
  CK(hipMalloc(&dw, hp.size() * 4)); CK(hipMalloc(&db, 64 * 4));
  CK(hipMemcpy(dw, hp.data(), hp.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemset(db, 0, 64 * 4));
  CK(hipMemcpy(db, hb.data(), Cout * 4, hipMemcpyHostToDevice));
  Params P;
  P.tk = nullptr;
  const int tk_cap = 4096;
  { unsigned long long* dtk0; CK(hipMalloc(&dtk0, (size_t)tk_cap * 4 * 8 * 8)); CK(hipMemset(dtk0, 0, (size_t)tk_cap * 4 * 8 * 8)); P.tk = dtk0; }
  P.wpack = dw; P.bias = db; P.B = B; P.H = H; P.W = W; P.Cout = Cout;
  P.tiles_x = (W + 15) / 16; P.tiles_y = (H + TH - 1) / TH; P.total_tiles = B * P.tiles_x * P.tiles_y;
  P.xbytes = (unsigned)(xn * 4); P.ybytes = (unsigned)(yn * 4);
  const size_t shmem = (size_t)2 * NCH * 16 + (size_t)9 * NCT * 64 * KS * 4;
  auto kern = conv3_dma_kernel<CIN, NCT, TH, BPC>;
  CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  int grid = 256 * BPC;
  if (const char* e = getenv("GRID")) grid = atoi(e);
  if (grid > P.total_tiles) grid = P.total_tiles;
  if (grid > tk_cap) grid = tk_cap;
  auto launch = [&](int i) { P.x = dx[i % nset]; P.y = dy[i % nset]; hipLaunchKernelGGL(kern, dim3(grid), dim3(256), shmem, 0, P); };
  launch(0);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(hy.data(), dy[0], yn * 4, hipMemcpyDeviceToHost));
  // check: every pixel of a few rows / columns incl. the borders and tile seams, plus random pixels
  double maxerr = 0.0, maxref = 0.0;
  long checked = 0;
  auto check_px = [&](int b, int oy, int ox) {
    for (int co = 0; co < Cout; ++co) {
      double a = hb[co];
      for (int ky = 0; ky < 3; ++ky)
        for (int kx = 0; kx < 3; ++kx) {
          const int iy = oy + ky - 1, ix = ox + kx - 1;
          if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
          const float* xp = &hx[(((size_t)b * H + iy) * W + ix) * CIN];
          for (int ci = 0; ci < CIN; ++ci) a += (double)xp[ci] * hw[((size_t)co * CIN + ci) * 9 + ky * 3 + kx];
        }
      const double g = hy[(((size_t)b * H + oy) * W + ox) * Cout + co];
      const double e = fabs(g - a);
      if (!(e <= maxerr)) maxerr = e;   // (NaN-proof)
      if (fabs(a) > maxref) maxref = fabs(a);
      ++checked;
    }
  };
  const int rows[] = {0, 1, TH - 1, TH, TH + 1, H / 2, H - TH - 1, H - 2, H - 1};
  for (int b = 0; b < B; b += (B > 1 ? B - 1 : 1))
    for (int r : rows)
      if (r >= 0 && r < H)
        for (int ox = 0; ox < W; ++ox) check_px(b, r, ox);
  const int cols[] = {0, 1, 15, 16, 17, W / 2, W - 17, W - 2, W - 1};
  for (int c : cols)
    if (c >= 0 && c < W)
      for (int oy = 0; oy < H; ++oy) check_px(B / 2, oy, c);
  for (int i = 0; i < 4000; ++i) { s = s * 1664525u + 1013904223u; const int b = (s >> 4) % B; s = s * 1664525u + 1013904223u; const int oy = (s >> 4) % H; s = s * 1664525u + 1013904223u; check_px(b, oy, (s >> 4) % W); }
  // every output written?
  long nanc = 0;
  for (size_t i = 0; i < yn; ++i) if (!(hy[i] == hy[i])) ++nanc;
  printf("check: %ld values, max abs err %.3e (max |ref| %.2f), unwritten outputs %ld  -> %s\n", checked, maxerr, maxref, nanc,
         (maxerr < 2e-4 * (maxref > 1 ? maxref : 1) && nanc == 0) ? "OK" : "FAIL");
#ifdef TIMING
  {
    unsigned long long* dtk = P.tk;
    for (int i = 0; i < 3; ++i) launch(i);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h((size_t)grid * 4 * 8);
    CK(hipMemcpy(h.data(), dtk, h.size() * 8, hipMemcpyDeviceToHost));
    double a[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < grid * 4; ++i) for (int k = 0; k < 6; ++k) a[k] += (double)h[(size_t)i * 8 + k];
    const double nt = a[5];
    printf("phase clocks per tile and wave (cycles): wait+barrier %.0f  stage %.0f  mfma %.0f  epilogue %.0f | kernel life per wave %.0f for %.1f tiles\n",
           a[0] / nt, a[1] / nt, a[2] / nt, a[3] / nt, a[4] / (grid * 4), nt / (grid * 4));
  }
#endif
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 5; ++i) launch(i);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) launch(i);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / iters, fl = 2.0 * B * H * W * Cout * CIN * 9, by = (double)(xn + yn) * 4;
  printf("conv3_dma %d->%d %dx%d B=%d TH=%d blocks/CU=%d grid=%d lds=%zu B: %.1f us  %.1f TF/s  %.0f GB/s (cold, %d sets)\n", CIN, Cout, H, W, B, TH, BPC, grid, shmem,
         us, fl / us * 1e-6, by / us * 1e-3, nset);
  for (int i = 0; i < nset; ++i) { CK(hipFree(dx[i])); CK(hipFree(dy[i])); }
  CK(hipFree(dw)); CK(hipFree(db));
  return 0;
}

int main(int argc, char** argv) {
  const int cin = argc > 1 ? atoi(argv[1]) : 24, cout = argc > 2 ? atoi(argv[2]) : 12;
  const int H = argc > 3 ? atoi(argv[3]) : 352, B = argc > 4 ? atoi(argv[4]) : 8;
  const int th = argc > 5 ? atoi(argv[5]) : 8;
  const size_t one = (size_t)B * H * H * (cin + cout) * 4;
  int nset = (int)(1.2e9 / one); nset = nset < 1 ? 1 : (nset > 6 ? 6 : nset);
  if (const char* e = getenv("NSET")) nset = atoi(e);
  const int iters = 30;
  const int nct = (cout + 15) / 16;
#define RUN(CI, NC, TH_, BPC_) if (cin == CI && nct == NC && th == TH_) return run<CI, NC, TH_, BPC_>(cout, H, H, B, nset, iters);
  RUN(24, 1, 8, 3) RUN(12, 1, 8, 4) RUN(24, 2, 8, 3) RUN(48, 2, 8, 2) RUN(48, 3, 8, 2)
  RUN(24, 1, 16, 2) RUN(12, 1, 16, 3) RUN(24, 2, 16, 2)
  printf("no instance for cin %d cout %d th %d\n", cin, cout, th);
  return 1;
}
