// Dense NHWC convolution as implicit GEMM on the gfx950 fp32 matrix cores
// (v_mfma_f32_32x32x2_f32: fp32 in, fp32 accumulate, bit-exact fmaf chain).
//
//   fwd   : Y[m=(n,oh,ow)][co]      = sum_k A[m][k=(kh,kw,ci)] * W[k][co]      (+bias)
//   dgrad : dX[m=(n,ih,iw)][ci]     = sum_k dY_gather[m][k=(kh,kw,co)] * W[(kh,kw)][ci][co]
//   wgrad : dW[m'=(kh,kw,ci)][co]   = sum_{p=(n,oh,ow)} X_gather[p][m'] * dY[p][co]   (split over p)
//
// One launch covers up to RN_MAX_SEG independent problems ("segments") -- the shared
// class/box heads applied to P3..P7 (reference retinanet.py:283-291) are ONE launch per layer,
// and for wgrad the pyramid levels are simply more reduction length for the shared kernel.
//
// Tiling: BMxBN block tile, BK=32, WMxWN waves (64 lanes) each owning TMxTN 32x32 MFMA tiles.
// Global -> register prefetch of tile t+1 overlaps the MFMAs of tile t (an f32 MFMA occupies
// its SIMD for 64 cycles, so one prefetch stage covers HBM/L2 latency); operands are staged in
// LDS, k-contiguous tiles padded to 36 floats so the ds_read_b128 fragments are conflict-free.
// Tile ids are remapped so neighbouring tiles (which share the activation rows / the weight
// panel) run on the same XCD and hit its L2.
#include <stdlib.h>

#include "rn_common.h"
#include "conv_tiles.h"

namespace {
using namespace rn_tiles;

struct SegDev {
  const float* a;     // fwd: x     dgrad: dy    wgrad: x
  const float* b;     // fwd: w     dgrad: w     wgrad: dy
  const float* bias;  // fwd only
  float* out;         // fwd: y     dgrad: dx    wgrad: unused
  int n, h, w, oh, ow, cout, pad_t, pad_l;
  int m;              // fwd: n*oh*ow   dgrad: n*h*w   wgrad: n*oh*ow (reduction length)
  int tiles_n;        // fwd/dgrad: tiles along N
  int start;          // fwd/dgrad: first tile id; wgrad: first split id
  int chunk;          // wgrad: reduction rows per split (multiple of BK)
  int x_ld, x_coff;   // pixel stride / first channel of the x (dx) tensor: a channel slice of a wider buffer
  // stride-2 dgrad, phase decomposition: this (pseudo-)segment covers the input pixels (2i+py, 2j+px) only, an
  // hc x wc grid per image; they see the taps kh = kh0, kh0+2, .. and kw = kw0, kw0+2, .. (nkh x nkw of them) --
  // the other 3/4 of the taps multiply structural zeros and are never touched.  par == 0: plain segment.
  int par, py, px, hc, wc, kh0, kw0, nkh, nkw;
};

template <int NS>
struct ConvArgsT {
  SegDev seg[NS];
  int nseg;
  int kh, kw, stride, cin;
  int groups, cin_g;  // grouped conv: cin_g = cin/groups input channels per group
  int tpg;            // N-tiles per group
  int ktotal;    // fwd: kh*kw*cin   dgrad: kh*kw*cout(seg)  (recomputed per seg)   wgrad: kh*kw*cin
  int tiles_mn;  // wgrad: output tiles per split
  int tiles_n;   // wgrad
  int cout;      // wgrad (all segments share it)
  float* slab;   // wgrad: [nsplit][ktotal][cout]
  // batched GEMM mode (Winograd: one 1x1 "conv" per transform point): nseg == 1, the grid is nbatch copies of
  // the segment's tiles, copy b uses a + b*bs_a, b + b*bs_b, out + b*bs_out (strides in floats)
  int nbatch, btiles;
  long bs_a, bs_b, bs_out;
  // split-K mode (fwd / dgrad of tiny grids, e.g. the 4x4 P7 map: 4 tiles x 72 K-tiles): the "batches" are K ranges of
  // `ksplit` K-tiles each (bs_a = bs_b = 0), written to slab rows bs_out apart and summed by reduce_rows
  int ksplit;
  // forward only: GroupNorm partial-sum rows of the output as a by-product (rn_conv2d_fwd_stats); rows == nullptr: off
  rn::StatDev st;
};
typedef ConvArgsT<RN_MAX_SEG> ConvArgs;
typedef ConvArgsT<4> ConvArgs4;  // compact copy (<= 4 segments) so that TWO argument blocks fit one 4 KB kernarg

template <typename A>
__device__ __forceinline__ int find_seg(const A& args, int id) {
  int s = 0;
  while (s + 1 < args.nseg && id >= args.seg[s + 1].start) ++s;
  return s;
}

// GroupNorm partial sums of the tile a block has just computed (rn::StatDev): per-channel (sum, sum of squares) over the
// tile's BM rows -> row `tile_m` of the rows tensor.  A tile never straddles two samples (the host checks that a sample's
// pixels are a multiple of BM), rows past the end do not exist (m % BM == 0), there is no bias.  `smem` = the dead operand tiles.
template <int BM, int BN, int WM, int WN>
__device__ __forceinline__ void conv_stats_epilogue(const f32x16 (&acc)[BM / WM / 32][BN / WN / 32], const rn::StatDev& st,
                                                    float* smem, int tile_m, int n0, int nmax, int cout, int wm, int wn, int lane) {
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  const int tid = threadIdx.x, l31 = lane & 31;
  float* red = smem;                                        // [WM][BN][2]
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int r = 0; r < 16; ++r) { const float v = acc[tm][tn][r]; s1 += v; s2 = fmaf(v, v, s2); }
    s1 += __shfl_xor(s1, 32, 64);                           // the other 16 rows of each 32-row slab
    s2 += __shfl_xor(s2, 32, 64);
    if (lane < 32) {
      const int col = wn * (BN / WN) + tn * 32 + l31;
      red[(wm * BN + col) * 2 + 0] = s1;
      red[(wm * BN + col) * 2 + 1] = s2;
    }
  }
  __syncthreads();
  if (tid < BN && n0 + tid < nmax) {
    float t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int w = 0; w < WM; ++w) { t1 += red[(w * BN + tid) * 2 + 0]; t2 += red[(w * BN + tid) * 2 + 1]; }
    st.rows[(size_t)tile_m * cout + n0 + tid] = make_float2(t1, t2);
  }
}

// =============================================================================================
// forward.  TAPU: cin % BK == 0, so every K-tile lies inside ONE filter tap and (kh, kw, ci0) are
// block-uniform scalars advanced incrementally -- no per-thread division in the K loop.
// =============================================================================================
template <int BM, int BN, int WM, int WN, int VEC, bool TAPU>
// waves per SIMD the 64 x 64 batched-product instantiations are compiled for (the register cap that goes with it: 72 / 80): one
// more resident block per CU than the default allocation gave -- 1 584 forward blocks then fit the chip in one round (1 792
// slots instead of 1 536).  Stand-alone time unchanged, +0.3 % in the step (interleaved A/B of two builds, RN_LIB_PATH).
#ifndef RN_OCC_FWD
#define RN_OCC_FWD 7
#endif
#ifndef RN_OCC_BWD
#define RN_OCC_BWD 6
#endif
__global__ __launch_bounds__(WM* WN * 64, (BM == 64 && BN == 64 && VEC == 4 && TAPU) ? RN_OCC_FWD : 1) void conv_fwd_kernel(const ConvArgs args) {
  constexpr int T = WM * WN * 64;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int KQ = BK / VEC, A_RPP = T / KQ, A_PASS = BM / A_RPP;
  constexpr int NQ = BN / VEC, B_RPP = T / NQ, B_PASS = BK / B_RPP;
  static_assert(A_PASS >= 1 && B_PASS >= 1 && BM % A_RPP == 0 && BK % B_RPP == 0, "tile/threads mismatch");
  typedef typename Vec<VEC>::type vec_t;
  __shared__ __attribute__((aligned(16))) float smem[BM * LDK + BK * BN];
  float* As = smem;
  float* Bs = smem + BM * LDK;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  int bid = rn::xcd_remap(blockIdx.x, gridDim.x);
  int batch = 0;
  if (args.nbatch > 1) { batch = bid / args.btiles; bid -= batch * args.btiles; }
  const int s = find_seg(args, bid);
  const SegDev& sg = args.seg[s];
  const int local = bid - sg.start;
  const int tile_n = local % sg.tiles_n, tile_m = local / sg.tiles_n;
  const int H = sg.h, W = sg.w, OW = sg.ow, OHW = sg.oh * sg.ow, M = sg.m, cout = sg.cout;
  const int ldx = sg.x_ld, kw = args.kw, stride = args.stride;
  // grouped conv (ResNeXt, resnet.py:53-59): group g owns cout/G output channels, fed by input channels
  // [g*cin_g, (g+1)*cin_g); kernel tensor is [kh,kw,cin_g,cout]; `tpg` N-tiles per group.
  const int G = args.groups, cin = args.cin_g;
  const int cout_g = cout / G;
  const int grp = tile_n / args.tpg, tn = tile_n - grp * args.tpg;
  const int m0 = tile_m * BM, n0 = grp * cout_g + tn * BN;
  const int nmax = (grp + 1) * cout_g;
  const int a_coff = sg.x_coff + grp * cin;
  const int ktotal = args.kh * args.kw * cin;
  const __amdgpu_buffer_rsrc_t xa = make_rsrc(sg.a + batch * args.bs_a, (unsigned)sg.n * H * W * ldx * 4u);
  const __amdgpu_buffer_rsrc_t wb = make_rsrc(sg.b + batch * args.bs_b, (unsigned)ktotal * cout * 4u);

  // per-thread im2col rows (fixed for the whole K loop): element offset of (n, ih0, iw0, 0)
  const int kq = tid % KQ;
  int ih0[A_PASS], iw0[A_PASS], rowoff[A_PASS];
#pragma unroll
  for (int i = 0; i < A_PASS; ++i) {
    const int m = m0 + tid / KQ + i * A_RPP;
    if (m < M) {
      const int n_ = m / OHW, rem = m - n_ * OHW;
      const int oh_ = rem / OW, ow_ = rem - oh_ * OW;
      ih0[i] = oh_ * stride - sg.pad_t;
      iw0[i] = ow_ * stride - sg.pad_l;
      rowoff[i] = ((n_ * H + ih0[i]) * W + iw0[i]) * ldx + a_coff;
    } else {
      ih0[i] = -0x40000000; iw0[i] = 0; rowoff[i] = 0;
    }
  }
  const int nq = tid % NQ;
  const int bcol = n0 + nq * VEC;
  // weight rows past ktotal fall outside the descriptor => zeros without a test
  const unsigned boff0 = bcol < nmax ? ((unsigned)(tid / NQ) * cout + bcol) * 4u : OOB;

  const int nk = (ktotal + BK - 1) / BK;
  int kt_begin = 0, kt_end = nk;
  if (args.ksplit) { kt_begin = batch * args.ksplit; kt_end = min(nk, kt_begin + args.ksplit); }
  int t_kh = 0, t_kw = 0, t_ci = 0;  // TAPU: block-uniform tap state of the tile being loaded
  if (TAPU && kt_begin) {
    const int tap0 = kt_begin * BK / cin;
    t_ci = kt_begin * BK - tap0 * cin; t_kh = tap0 / kw; t_kw = tap0 - t_kh * kw;
  }
  vec_t ra[A_PASS], rb[B_PASS];
  auto load_tiles = [&](int kt) {
    int khh, kww, tapoff;
    bool kok = true;
    if (TAPU) {
      khh = t_kh; kww = t_kw;
      tapoff = (khh * W + kww) * ldx + t_ci + kq * VEC;
      t_ci += BK;
      if (t_ci == cin) { t_ci = 0; if (++t_kw == kw) { t_kw = 0; ++t_kh; } }
    } else {
      const int k = kt * BK + kq * VEC;
      kok = k < ktotal;
      const int tap = k / cin, ci = k - tap * cin;
      khh = tap / kw; kww = tap - khh * kw;
      tapoff = (khh * W + kww) * ldx + ci;
    }
#pragma unroll
    for (int i = 0; i < A_PASS; ++i) {
      const int ih = ih0[i] + khh, iw = iw0[i] + kww;
      const bool ok = kok && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
      ra[i] = Vec<VEC>::load(xa, ok ? (unsigned)(rowoff[i] + tapoff) * 4u : OOB);
    }
    const unsigned bo = boff0 + (unsigned)kt * BK * cout * 4u;
#pragma unroll
    for (int j = 0; j < B_PASS; ++j) rb[j] = Vec<VEC>::load(wb, bo + (unsigned)j * B_RPP * cout * 4u);
  };
  auto store_tiles = [&]() {
#pragma unroll
    for (int i = 0; i < A_PASS; ++i)
      *reinterpret_cast<vec_t*>(&As[(tid / KQ + i * A_RPP) * LDK + kq * VEC]) = ra[i];
#pragma unroll
    for (int j = 0; j < B_PASS; ++j)
      *reinterpret_cast<vec_t*>(&Bs[(tid / NQ + j * B_RPP) * BN + nq * VEC]) = rb[j];
  };

  f32x16 acc[TM][TN];
  zero_acc<TM, TN>(acc);
  load_tiles(kt_begin);
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    store_tiles();
    __syncthreads();
    if (kt + 1 < kt_end) load_tiles(kt + 1);
    mma_ktile<BM, BN, WM, WN, false, false>(As, Bs, acc, wm, wn, lane);
    __syncthreads();
  }
  store_tile<BM, BN, WM, WN>(acc, sg.out + batch * args.bs_out, (args.ksplit && batch) ? nullptr : sg.bias, m0, n0, M, nmax, cout,
                             wm, wn, lane);
  if (args.st.rows) conv_stats_epilogue<BM, BN, WM, WN>(acc, args.st, smem, tile_m, n0, nmax, cout, wm, wn, lane);
}

// =============================================================================================
// dgrad: rows = input pixels, K = (kh,kw,co), N = ci.  W tile is read "NK" (ci rows, co contiguous).
// TAPU: cout % BK == 0 (tap uniform per K-tile).
// =============================================================================================
template <int BM, int BN>
constexpr int dgrad_lds_floats() { return BM * LDK + BN * LDK; }

// body of the data-gradient kernel: block `blk` of `nblk`, operand tiles in `smem` (dgrad_lds_floats floats)
template <int BM, int BN, int WM, int WN, int VEC, bool TAPU, typename A>
__device__ __forceinline__ void conv_dgrad_body(const A& args, float* smem, int blk, int nblk) {
  constexpr int T = WM * WN * 64;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int KQ = BK / VEC, RPP = T / KQ, A_PASS = BM / RPP, B_PASS = BN / RPP;
  static_assert(A_PASS >= 1 && B_PASS >= 1, "tile/threads mismatch");
  typedef typename Vec<VEC>::type vec_t;
  float* As = smem;
  float* Bs = smem + BM * LDK;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  // phase-decomposed launches keep the hardware's round-robin order: their pseudo-segments differ 4x in K (1 / 2 / 2 / 4
  // taps), and handing each XCD a contiguous tile range would give two XCDs all the 4-tap tiles (4/9 of the work)
  int bid = args.seg[0].par ? blk : rn::xcd_remap(blk, nblk);
  int batch = 0;
  if (args.nbatch > 1) { batch = bid / args.btiles; bid -= batch * args.btiles; }
  const int s = find_seg(args, bid);
  const SegDev& sg = args.seg[s];
  const int local = bid - sg.start;
  const int tile_n = local % sg.tiles_n, tile_m = local / sg.tiles_n;
  const int W = sg.w, OH = sg.oh, OW = sg.ow, HW = sg.h * sg.w, M = sg.m;
  const int kw = args.kw, stride = args.stride;
  const int par = sg.par, tstep = par ? 2 : 1, kh0 = par ? sg.kh0 : 0, kw0 = par ? sg.kw0 : 0;
  const int nkw_c = par ? sg.nkw : kw;                      // taps per kernel row seen by this segment
  // grouped: N-tile `tile_n` is group g; K runs over (tap, co within the group)
  const int G = args.groups, cin = args.cin_g, ldy = sg.cout, ldx = sg.x_ld;
  const int cout = ldy / G;                                  // output channels per group = K per tap
  const int grp = tile_n / args.tpg, tn = tile_n - grp * args.tpg;
  const int m0 = tile_m * BM, n0 = tn * BN;                  // n0: first ci (within the group) of the tile
  const int y_coff = grp * cout;                             // channel offset into dy and into w's cout axis
  const int x_coff = sg.x_coff + grp * cin;                  // channel offset into dx
  const int ktotal = (par ? sg.nkh * sg.nkw : args.kh * args.kw) * cout;
  const __amdgpu_buffer_rsrc_t dy = make_rsrc(sg.a + batch * args.bs_a, (unsigned)sg.n * OH * OW * ldy * 4u);
  const __amdgpu_buffer_rsrc_t wb = make_rsrc(sg.b + batch * args.bs_b, (unsigned)args.kh * args.kw * cin * ldy * 4u);

  const int kq = tid % KQ, r0 = tid / KQ;
  int ihp[A_PASS], iwp[A_PASS], nb[A_PASS];
#pragma unroll
  for (int i = 0; i < A_PASS; ++i) {
    const int m = m0 + r0 + i * RPP;
    if (m < M) {
      int n_, ih, iw;
      if (par) {
        const int hw_c = sg.hc * sg.wc;
        n_ = m / hw_c;
        const int rem = m - n_ * hw_c, i_ = rem / sg.wc;
        ih = 2 * i_ + sg.py; iw = 2 * (rem - i_ * sg.wc) + sg.px;
      } else {
        n_ = m / HW;
        const int rem = m - n_ * HW;
        ih = rem / W; iw = rem - ih * W;
      }
      ihp[i] = ih + sg.pad_t;
      iwp[i] = iw + sg.pad_l;
      nb[i] = n_ * OH;
    } else {
      ihp[i] = -0x40000000; iwp[i] = 0; nb[i] = 0;
    }
  }
  // weight row (ci) of this thread in each B pass; rows >= cin must not alias the next tap
  unsigned browoff[B_PASS];
#pragma unroll
  for (int j = 0; j < B_PASS; ++j) {
    const int ci = n0 + r0 + j * RPP;
    browoff[j] = ci < cin ? ((unsigned)ci * ldy + y_coff) * 4u : OOB;
  }

  const int nk = (ktotal + BK - 1) / BK;
  int kt_begin = 0, kt_end = nk;
  if (args.ksplit) { kt_begin = batch * args.ksplit; kt_end = min(nk, kt_begin + args.ksplit); }
  int t_kh = kh0, t_kw = kw0, t_co = 0;
  if (TAPU && kt_begin) {
    const int tap0 = kt_begin * BK / cout;
    t_co = kt_begin * BK - tap0 * cout; t_kh = kh0 + tstep * (tap0 / nkw_c); t_kw = kw0 + tstep * (tap0 % nkw_c);
  }
  vec_t ra[A_PASS], rb[B_PASS];
  auto load_tiles = [&](int kt) {
    int khh, kww, co, tap;
    bool kok = true;
    if (TAPU) {
      khh = t_kh; kww = t_kw; tap = khh * kw + kww;
      co = t_co + kq * VEC;
      t_co += BK;
      if (t_co == cout) { t_co = 0; t_kw += tstep; if (t_kw >= kw) { t_kw = kw0; t_kh += tstep; } }
    } else {
      const int k = kt * BK + kq * VEC;
      kok = k < ktotal;
      const int tc = k / cout;
      co = k - tc * cout;
      const int tr = tc / nkw_c;
      khh = kh0 + tstep * tr; kww = kw0 + tstep * (tc - tr * nkw_c);
      tap = khh * kw + kww;
    }
#pragma unroll
    for (int i = 0; i < A_PASS; ++i) {
      const int ohs = ihp[i] - khh, ows = iwp[i] - kww;
      int oh_, ow_;
      bool ok = kok && ohs >= 0 && ows >= 0;
      if (stride == 1) {
        oh_ = ohs; ow_ = ows;
      } else {
        oh_ = ohs / stride; ow_ = ows / stride;
        ok = ok && (oh_ * stride == ohs) && (ow_ * stride == ows);
      }
      ok = ok && oh_ < OH && ow_ < OW;
      ra[i] = Vec<VEC>::load(dy, ok ? (unsigned)(((nb[i] + oh_) * OW + ow_) * ldy + y_coff + co) * 4u : OOB);
    }
    const unsigned tapb = kok ? ((unsigned)tap * cin * ldy + co) * 4u : OOB;
#pragma unroll
    for (int j = 0; j < B_PASS; ++j) rb[j] = Vec<VEC>::load(wb, (tapb | browoff[j]) >= OOB ? OOB : tapb + browoff[j]);
  };
  auto store_tiles = [&]() {
#pragma unroll
    for (int i = 0; i < A_PASS; ++i) *reinterpret_cast<vec_t*>(&As[(r0 + i * RPP) * LDK + kq * VEC]) = ra[i];
#pragma unroll
    for (int j = 0; j < B_PASS; ++j) *reinterpret_cast<vec_t*>(&Bs[(r0 + j * RPP) * LDK + kq * VEC]) = rb[j];
  };

  f32x16 acc[TM][TN];
  zero_acc<TM, TN>(acc);
  if (kt_begin < kt_end) load_tiles(kt_begin);  // a phase can have no taps at all (1x1 / stride 2): it stores zeros
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    store_tiles();
    __syncthreads();
    if (kt + 1 < kt_end) load_tiles(kt + 1);
    mma_ktile<BM, BN, WM, WN, false, true>(As, Bs, acc, wm, wn, lane);
    __syncthreads();
  }
  if (!par) {
    store_tile<BM, BN, WM, WN>(acc, sg.out + batch * args.bs_out, nullptr, m0, x_coff + n0, M, x_coff + cin, ldx, wm, wn, lane);
    return;
  }
  // phase rows are not consecutive pixels: pixel index of each of the tile's rows via LDS (the operand tiles are dead)
  int* rowpix = reinterpret_cast<int*>(smem);
  for (int r = tid; r < BM; r += T) {
    const int m = m0 + r;
    int pix = -1;
    if (m < M) {
      const int hw_c = sg.hc * sg.wc, n_ = m / hw_c, rem = m - n_ * hw_c, i_ = rem / sg.wc;
      pix = (n_ * sg.h + 2 * i_ + sg.py) * W + 2 * (rem - i_ * sg.wc) + sg.px;
    }
    rowpix[r] = pix;
  }
  __syncthreads();
  {
    const int l31 = lane & 31, half = lane >> 5;
    float* out = sg.out;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(out, (unsigned)sg.n * sg.h * W * (unsigned)ldx * 4u);
#pragma unroll
    for (int tn2 = 0; tn2 < TN; ++tn2) {
      const int col = x_coff + n0 + wn * (BN / WN) + tn2 * 32 + l31;
      const bool cok = col < x_coff + cin;
#pragma unroll
      for (int tm2 = 0; tm2 < TM; ++tm2) {
        const int rbase = wm * (BM / WM) + tm2 * 32 + 4 * half;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int pix = rowpix[rbase + (r & 3) + 8 * (r >> 2)];
          const unsigned voff = (cok && pix >= 0) ? ((unsigned)pix * (unsigned)ldx + (unsigned)col) * 4u : OOB;
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[tm2][tn2][r]), rs, voff, 0, 0);
        }
      }
    }
  }
}

template <int BM, int BN, int WM, int WN, int VEC, bool TAPU>
__global__ __launch_bounds__(WM* WN * 64) void conv_dgrad_kernel(const ConvArgs args) {
  __shared__ __attribute__((aligned(16))) float smem[dgrad_lds_floats<BM, BN>()];
  conv_dgrad_body<BM, BN, WM, WN, VEC, TAPU>(args, smem, blockIdx.x, gridDim.x);
}

// =============================================================================================
// wgrad: rows m' = (kh,kw,ci), cols = co, reduction over output pixels p, split over blocks.
// A tile is "KM" (pixel rows, m' contiguous = ci contiguous in x), B tile is dY rows.
// The (n, oh, ow) decomposition of the block's pixel range is done ONCE into an LDS table
// (element offset of the window origin + packed ih0/iw0), so the K loop has no division.
// =============================================================================================
constexpr int WG_MAXPIX = 1024;  // pixels per split (plan_wgrad keeps chunks <= this)

template <int BM, int BN>
constexpr int wgrad_lds_floats() { return BK * BM + BK * BN + 2 * WG_MAXPIX; }  // operand tiles + the pixel table (int2)

// body of the weight-gradient kernel: block `blk` of `nblk`, LDS in `smem` (wgrad_lds_floats floats)
template <int BM, int BN, int WM, int WN, int VEC, typename A>
__device__ __forceinline__ void conv_wgrad_body(const A& args, float* smem, int blk, int nblk) {
  constexpr int T = WM * WN * 64;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int MQ = BM / VEC, A_RPP = T / MQ, A_PASS = BK / A_RPP;
  constexpr int NQ = BN / VEC, B_RPP = T / NQ, B_PASS = BK / B_RPP;
  static_assert(A_PASS >= 1 && B_PASS >= 1 && BK % A_RPP == 0 && BK % B_RPP == 0, "tile/threads mismatch");
  typedef typename Vec<VEC>::type vec_t;
  float* As = smem;
  float* Bs = smem + BK * BM;
  int2* pixtab = reinterpret_cast<int2*>(smem + BK * BM + BK * BN);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int bid = rn::xcd_remap(blk, nblk);
  int split = bid / args.tiles_mn;
  const int t = bid - split * args.tiles_mn;
  const int tile_n = t % args.tiles_n, tile_m = t / args.tiles_n;
  // batched mode (one segment): split id = batch * btiles + split within the batch; the slab is laid out
  // [split within batch][batch][ktotal][cout] so that ONE row reduction sums the splits of every batch
  int batch = 0, slab_row = split;
  if (args.nbatch > 1) {
    batch = split / args.btiles;
    split -= batch * args.btiles;
    slab_row = split * args.nbatch + batch;
  }
  const int s = find_seg(args, split);
  const SegDev& sg = args.seg[s];
  const int p0_all = (split - sg.start) * sg.chunk;
  const int p1_all = min(p0_all + sg.chunk, sg.m);
  int p0 = p0_all, p1 = min(p0_all + WG_MAXPIX, p1_all);  // current window of <= WG_MAXPIX pixels (the LDS table's size)
  const int H = sg.h, W = sg.w, OW = sg.ow, OHW = sg.oh * sg.ow;
  const int ldx = sg.x_ld, cout = args.cout, kw = args.kw, stride = args.stride;
  const int G = args.groups, cin = args.cin_g, cout_g = cout / G;
  const int grp = tile_n / args.tpg, tn = tile_n - grp * args.tpg;
  const int m0 = tile_m * BM, n0 = grp * cout_g + tn * BN;
  const int nmax = (grp + 1) * cout_g;
  const int x_coff = sg.x_coff + grp * cin;
  const int ktotal = args.ktotal;
  const __amdgpu_buffer_rsrc_t xa = make_rsrc(sg.a + batch * args.bs_a, (unsigned)sg.n * H * W * ldx * 4u);
  const __amdgpu_buffer_rsrc_t dy = make_rsrc(sg.b + batch * args.bs_b, (unsigned)sg.m * cout * 4u);

  auto fill_pixtab = [&]() {
    for (int i = tid; i < p1 - p0; i += T) {
      const int p = p0 + i;
      const int n_ = p / OHW, rem = p - n_ * OHW;
      const int oh_ = rem / OW, ow_ = rem - oh_ * OW;
      const int ih0 = oh_ * stride - sg.pad_t, iw0 = ow_ * stride - sg.pad_l;
      pixtab[i] = make_int2(((n_ * H + ih0) * W + iw0) * ldx + x_coff, (ih0 << 16) | (iw0 & 0xffff));
    }
    __syncthreads();
  };

  // this thread's m' (fixed): tap and channel
  const int mq = tid % MQ;
  const int mrow = m0 + mq * VEC;
  const bool mok = mrow < ktotal;
  const int tap = mrow / cin, ci = mrow - tap * cin;
  const int khh = tap / kw, kww = tap - khh * kw;
  const int tapoff = (khh * W + kww) * ldx + ci;
  const int nq = tid % NQ;
  const int bcol = n0 + nq * VEC;
  const unsigned boff0 = bcol < nmax ? (unsigned)bcol * 4u : OOB;

  vec_t ra[A_PASS], rb[B_PASS];
  auto load_tiles = [&](int kt) {
#pragma unroll
    for (int j = 0; j < A_PASS; ++j) {
      const int pl = kt * BK + tid / MQ + j * A_RPP;
      unsigned voff = OOB;
      if (mok && pl < p1 - p0) {
        const int2 e = pixtab[pl];
        const int ih = (e.y >> 16) + khh, iw = (int)(short)(e.y & 0xffff) + kww;
        if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) voff = (unsigned)(e.x + tapoff) * 4u;
      }
      ra[j] = Vec<VEC>::load(xa, voff);
    }
#pragma unroll
    for (int j = 0; j < B_PASS; ++j) {
      const int p = p0 + kt * BK + tid / NQ + j * B_RPP;
      rb[j] = Vec<VEC>::load(dy, p < p1 ? boff0 + (unsigned)p * cout * 4u : OOB);
    }
  };
  auto store_tiles = [&]() {
#pragma unroll
    for (int j = 0; j < A_PASS; ++j)
      *reinterpret_cast<vec_t*>(&As[(tid / MQ + j * A_RPP) * BM + mq * VEC]) = ra[j];
#pragma unroll
    for (int j = 0; j < B_PASS; ++j)
      *reinterpret_cast<vec_t*>(&Bs[(tid / NQ + j * B_RPP) * BN + nq * VEC]) = rb[j];
  };

  f32x16 acc[TM][TN];
  zero_acc<TM, TN>(acc);
  for (;;) {  // windows of the split's pixel range (one window unless the split is longer than the LDS table)
    fill_pixtab();
    const int nk = (p1 - p0 + BK - 1) / BK;
    if (nk > 0) load_tiles(0);
    for (int kt = 0; kt < nk; ++kt) {
      store_tiles();
      __syncthreads();
      if (kt + 1 < nk) load_tiles(kt + 1);
      mma_ktile<BM, BN, WM, WN, true, false>(As, Bs, acc, wm, wn, lane);
      __syncthreads();
    }
    if (p1 >= p1_all) break;
    p0 = p1;
    p1 = min(p0 + WG_MAXPIX, p1_all);
  }
  float* out = args.slab + (size_t)slab_row * ktotal * cout;
  store_tile<BM, BN, WM, WN>(acc, out, nullptr, m0, n0, ktotal, nmax, cout, wm, wn, lane);
}

template <int BM, int BN, int WM, int WN, int VEC>
__global__ __launch_bounds__(WM* WN * 64) void conv_wgrad_kernel(const ConvArgs args) {
  __shared__ __attribute__((aligned(16))) float smem[wgrad_lds_floats<BM, BN>()];
  conv_wgrad_body<BM, BN, WM, WN, VEC>(args, smem, blockIdx.x, gridDim.x);
}

// ---------------------------------------------------------------------------------------------
// The stem: 3 x 3 / stride 2, 3 -> 32 channels, no bias (mobilenet_v2.py:112-114).  27 multiply-adds per output: nothing for
// the matrix cores (the scalar-gather implicit GEMM spent 24 us on 23 MB) -- a direct kernel.  Block = 256 consecutive output
// pixels of one sample (8 runs of 32 along a row), thread = (pixel of the run, channel quad): the run's three input rows
// (65 pixels x 3 channels each) are staged in LDS (coalesced loads, the next run's in flight under this run's arithmetic),
// weights in LDS, 128-byte rows out.  The GroupNorm statistics of the 256 pixels leave as one row (the conv kernels' layout),
// so the GroupNorm behind it is one apply kernel.
// ---------------------------------------------------------------------------------------------
constexpr int STEM_PPB = 256, STEM_RUN = 32, STEM_PATCH = 3 * (2 * STEM_RUN + 1) * 3;   // 585 floats
struct StemArgs { const float* x; const float* w; float* y; float2* rows; int n, h, wd, oh, ow, pad_t, pad_l; };
__global__ __launch_bounds__(256) void stem_conv_fwd_kernel(const StemArgs a) {
  __shared__ float patch[2][STEM_PATCH + 3];
  __shared__ float red[4][64];
  const int tid = threadIdx.x, q = tid & 7, px = tid >> 3;
  const int ohw = a.oh * a.ow;
  const int blocks_per_sample = ohw / STEM_PPB;
  const int sample = blockIdx.x / blocks_per_sample, p_base = (blockIdx.x - sample * blocks_per_sample) * STEM_PPB;
  const float* __restrict__ xs = a.x + (size_t)sample * a.h * a.wd * 3;
  float4 wq[27];                                     // this thread's quad of every tap's weights (registers: LDS reads would bound the kernel)
#pragma unroll
  for (int t = 0; t < 27; ++t) wq[t] = *reinterpret_cast<const float4*>(a.w + t * 32 + q * 4);
  constexpr int NL = (STEM_PATCH + 255) / 256;       // 3 loads per thread and run
  float pre[NL];
  auto load_run = [&](int run) {
    const int p0 = p_base + run * STEM_RUN;
    const int oy = p0 / a.ow, ox0 = p0 - oy * a.ow;
    const int iy0 = oy * 2 - a.pad_t, ix0 = ox0 * 2 - a.pad_l;
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      const int idx = tid + k * 256;
      const int r = idx / 195, off = idx - r * 195;            // row of the patch, float inside it (65 pixels x 3)
      const int iy = iy0 + r, ix = ix0 + off / 3;
      const bool ok = idx < STEM_PATCH && (unsigned)iy < (unsigned)a.h && (unsigned)ix < (unsigned)a.wd;
      // (clamped address x 0/1 mask: a load under a condition is compiled to branch + load + wait, one round trip per element)
      const int iyc = min(max(iy, 0), a.h - 1), ixc = min(max(ix, 0), a.wd - 1);
      pre[k] = xs[((size_t)iyc * a.wd + ixc) * 3 + (off - (off / 3) * 3)] * (ok ? 1.f : 0.f);
    }
  };
  auto store_run = [&](int buf) {
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      const int idx = tid + k * 256;
      if (idx < STEM_PATCH) patch[buf][idx] = pre[k];
    }
  };
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  load_run(0);
  store_run(0);
  __syncthreads();
  constexpr int NRUN = STEM_PPB / STEM_RUN;
  for (int run = 0; run < NRUN; ++run) {
    if (run + 1 < NRUN) load_run(run + 1);
    const float* __restrict__ pt = patch[run & 1];
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int ci = 0; ci < 3; ++ci) {
          const float xv = pt[kh * 195 + (2 * px + kw) * 3 + ci];
          const float4 w4 = wq[(kh * 3 + kw) * 3 + ci];
          acc[0] = fmaf(xv, w4.x, acc[0]); acc[1] = fmaf(xv, w4.y, acc[1]);
          acc[2] = fmaf(xv, w4.z, acc[2]); acc[3] = fmaf(xv, w4.w, acc[3]);
        }
    const size_t o = ((size_t)sample * ohw + p_base + run * STEM_RUN + px) * 32 + q * 4;
    *reinterpret_cast<float4*>(a.y + o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
#pragma unroll
    for (int j = 0; j < 4; ++j) { s1[j] += acc[j]; s2[j] = fmaf(acc[j], acc[j], s2[j]); }
    if (run + 1 < NRUN) store_run((run + 1) & 1);
    __syncthreads();
  }
  if (!a.rows) return;
  // per-channel sums over the block's pixels: the 8 pixel lanes of a wave by shuffles (lane = q + 8 px), the 4 waves through LDS
#pragma unroll
  for (int j = 0; j < 4; ++j) {
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) { s1[j] += __shfl_xor(s1[j], o, 64); s2[j] += __shfl_xor(s2[j], o, 64); }
  }
  const int lane = tid & 63, wave = tid >> 6;
  if (lane < 8) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { red[wave][(lane * 4 + j) * 2] = s1[j]; red[wave][(lane * 4 + j) * 2 + 1] = s2[j]; }
  }
  __syncthreads();
  if (tid < 32) {
    float t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) { t1 += red[w][tid * 2]; t2 += red[w][tid * 2 + 1]; }
    a.rows[(size_t)blockIdx.x * 32 + tid] = make_float2(t1, t2);
  }
}

// Both gradients of one convolution in ONE launch: blocks [0, dblocks) run the data-gradient tiles, the rest the
// weight-gradient splits (independent work on the same dy).  The small backbone convs are launch-latency-bound, and two
// half-empty grids fill the chip better together.  Two compact argument blocks (<= 4 segments each) fit the kernarg.
template <int DBM, int DBN, int DWM, int DWN, bool DTAPU, int WBM, int WBN, int WWM, int WWN>
__global__ __launch_bounds__(256, (DTAPU && WBM == 64 && WBN == 64) ? RN_OCC_BWD : 1) void conv_bwd_kernel(const ConvArgs4 d, const ConvArgs4 w, int dblocks) {
  constexpr int LDSF = dgrad_lds_floats<DBM, DBN>() > wgrad_lds_floats<WBM, WBN>() ? dgrad_lds_floats<DBM, DBN>()
                                                                                   : wgrad_lds_floats<WBM, WBN>();
  static_assert(DWM * DWN == 4 && WWM * WWN == 4, "both halves use 256-thread blocks");
  __shared__ __attribute__((aligned(16))) float smem[LDSF];
  if ((int)blockIdx.x < dblocks) conv_dgrad_body<DBM, DBN, DWM, DWN, 4, DTAPU>(d, smem, blockIdx.x, dblocks);
  else conv_wgrad_body<WBM, WBN, WWM, WWN, 4>(w, smem, (int)blockIdx.x - dblocks, (int)gridDim.x - dblocks);
}

// out[i] = (accumulate ? out[i] : 0) + sum_r in[r][i], fixed order => bitwise reproducible.
// Block = (256 / RL) float4 columns x RL row lanes: lane l sums rows l, l+RL, ... (four loads in flight: the loop is
// latency-bound), the lanes are combined in lane order through LDS.  RL = 16 (coalesced 256-B row segments) for short
// reductions, RL = 64 for tall ones (>= REDUCE_TALL rows: the depthwise weight-gradient partials have up to 4096), where
// the length of a lane's dependent chain matters more than the segment width.
constexpr int REDUCE_TALL = 512;
// ... and RL = 4 for the shortest (< REDUCE_SHORT rows: the 2 - 10 split-K slabs of a weight gradient -- most of a step's ~130
// reductions): 1 KB row segments, a quarter of the blocks (the batched launch was bound by the block dispatch rate: 28 721
// blocks of 1 - 4 KB of work each took 45 us)
constexpr int REDUCE_SHORT = 32;
__host__ __device__ inline int64_t reduce_cols_per_block(int nrows) { return nrows >= REDUCE_TALL ? 16 : (nrows < REDUCE_SHORT ? 256 : 64); }

template <int RL>
__device__ __forceinline__ void reduce_rows_lanes(const float* __restrict__ in, float* __restrict__ out, int64_t count, int nrows,
                                                  int accumulate, int block, float4* sh) {
  constexpr int CL = 256 / RL;
  const int cl = threadIdx.x % CL, rl = threadIdx.x / CL;
  const int64_t i = ((int64_t)block * CL + cl) * 4;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < count) {
    const bool full = (i + 4 <= count) && ((count & 3) == 0);
    int r = rl;
    if (full) {
      float4 v1 = v, v2 = v, v3 = v;
      for (; r + 3 * RL < nrows; r += 4 * RL) {
        const float* p = in + (size_t)r * count + i;
        const float4 t0 = *reinterpret_cast<const float4*>(p);
        const float4 t1 = *reinterpret_cast<const float4*>(p + (size_t)RL * count);
        const float4 t2 = *reinterpret_cast<const float4*>(p + (size_t)(2 * RL) * count);
        const float4 t3 = *reinterpret_cast<const float4*>(p + (size_t)(3 * RL) * count);
        v.x += t0.x; v.y += t0.y; v.z += t0.z; v.w += t0.w;
        v1.x += t1.x; v1.y += t1.y; v1.z += t1.z; v1.w += t1.w;
        v2.x += t2.x; v2.y += t2.y; v2.z += t2.z; v2.w += t2.w;
        v3.x += t3.x; v3.y += t3.y; v3.z += t3.z; v3.w += t3.w;
      }
      v.x = (v.x + v1.x) + (v2.x + v3.x); v.y = (v.y + v1.y) + (v2.y + v3.y);
      v.z = (v.z + v1.z) + (v2.z + v3.z); v.w = (v.w + v1.w) + (v2.w + v3.w);
    }
    for (; r < nrows; r += RL) {
      const float* p = in + (size_t)r * count + i;
      if (full) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
      } else {
        v.x += p[0];
        if (i + 1 < count) v.y += p[1];
        if (i + 2 < count) v.z += p[2];
        if (i + 3 < count) v.w += p[3];
      }
    }
  }
  sh[rl * CL + cl] = v;
  __syncthreads();
  if (RL > 16) {  // 64 lanes -> 16 (lanes 4q..4q+3 in order), then the common tail
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    if (rl < 16) {
#pragma unroll
      for (int k = 0; k < RL / 16; ++k) {
        const float4 u = sh[(rl * (RL / 16) + k) * CL + cl];
        t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
      }
    }
    __syncthreads();
    if (rl < 16) sh[rl * CL + cl] = t;
    __syncthreads();
  }
  if (rl == 0 && i < count) {
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    if (accumulate) {
      t.x = out[i];
      if (i + 1 < count) t.y = out[i + 1];
      if (i + 2 < count) t.z = out[i + 2];
      if (i + 3 < count) t.w = out[i + 3];
    }
#pragma unroll
    for (int r = 0; r < (RL < 16 ? RL : 16); ++r) { const float4 u = sh[r * CL + cl]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
    out[i] = t.x;
    if (i + 1 < count) out[i + 1] = t.y;
    if (i + 2 < count) out[i + 2] = t.z;
    if (i + 3 < count) out[i + 3] = t.w;
  }
}

__device__ __forceinline__ void reduce_rows_block(const float* __restrict__ in, float* __restrict__ out, int64_t count, int nrows,
                                                  int accumulate, int block) {
  __shared__ float4 sh[256];
  if (nrows >= REDUCE_TALL) reduce_rows_lanes<64>(in, out, count, nrows, accumulate, block, sh);  // block-uniform
  else if (nrows < REDUCE_SHORT) reduce_rows_lanes<4>(in, out, count, nrows, accumulate, block, sh);
  else reduce_rows_lanes<16>(in, out, count, nrows, accumulate, block, sh);
}

__global__ __launch_bounds__(256) void reduce_rows_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                          int64_t count, int nrows, int accumulate) {
  reduce_rows_block(in, out, count, nrows, accumulate, blockIdx.x);
}

// Deferred mode (a `defer` list handed to the entry points): every row reduction recorded during a backward pass -- the split-K slabs of
// ~70 weight gradients, the GroupNorm parameter-gradient rows -- runs as ONE launch instead of one
// launch-latency-bound kernel each.
constexpr int RN_MAX_REDUCE = 140;  // 24-byte descriptors + block starts: the whole step's ~135 reductions in one 4 KB kernarg
struct ReduceDesc { const float* in; float* out; int count; unsigned short nrows, accumulate; };
struct ReduceManyArgs {
  ReduceDesc d[RN_MAX_REDUCE];
  int block_start[RN_MAX_REDUCE + 1];
  int n;
};
static_assert(sizeof(ReduceDesc) == 24 && sizeof(ReduceManyArgs) <= 4096, "descriptors must fit the kernel-argument segment");
__global__ __launch_bounds__(256) void reduce_rows_many_kernel(const ReduceManyArgs a) {
  int lo = 0, hi = a.n - 1;  // last descriptor whose first block is <= blockIdx.x
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (a.block_start[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const ReduceDesc& d = a.d[lo];
  reduce_rows_block(d.in, d.out, d.count, d.nrows, d.accumulate, (int)blockIdx.x - a.block_start[lo]);
}

// ---------------------------------------------------------------------------------------------
// host side: tile-shape choice and launch
// ---------------------------------------------------------------------------------------------
struct TileCfg {
  int bm, bn;
  double penalty;
};
const TileCfg kCfgs[] = {{128, 128, 1.00}, {128, 64, 1.06}, {64, 64, 1.12}, {128, 32, 1.20}};
constexpr int kNumCfg = 4;

// Pick the tile shape from a measured latency model (tools/conv_occupancy.py on MI355X): with b blocks of
// a shape resident per CU, one K-iteration of every resident block takes about t0 + t1*b nanoseconds
//   128x128: 700 + 1780 b     128x64: 550 + 1000 b     64x64: 430 + 515 b     128x32: 430 + 560 b
// (big tiles win once they fill the chip, small tiles win when there are only a few: a 2-block grid of
// 128x128 tiles is 2.6x slower per iteration than 8 blocks of 64x64).  All segments share K.
const double kT0[kNumCfg] = {700, 550, 430, 430}, kT1[kNumCfg] = {1780, 1000, 515, 560};
template <typename F>
int choose_cfg(F dims, int nseg, int nbatch = 1) {
  if (const char* force = getenv("RN_CONV_CFG")) {  // tuning aid: force a tile shape (0..3)
    const int c = atoi(force);
    if (c >= 0 && c < kNumCfg) return c;
  }
  int best = 0;
  double best_cost = 1e300;
  for (int c = 0; c < kNumCfg; ++c) {
    long tiles = 0;
    for (int s = 0; s < nseg; ++s) {
      long m, n;
      dims(s, &m, &n);
      tiles += ((m + kCfgs[c].bm - 1) / kCfgs[c].bm) * ((n + kCfgs[c].bn - 1) / kCfgs[c].bn);
    }
    tiles *= nbatch;
    const double b = (double)((tiles + 255) / 256);
    const double cost = kT0[c] + kT1[c] * b;
    if (cost < best_cost) { best_cost = cost; best = c; }
  }
  return best;
}

// grouped conv with narrow groups: one N-tile per group, as narrow as possible
int cfg_for_group_width(int width) { return width <= 32 ? 3 : (width <= 64 ? 2 : 0); }
inline void set_x_view(SegDev& d, const rn_conv_seg& s, int cin) {
  d.x_ld = s.x_ld > 0 ? s.x_ld : cin;
  d.x_coff = s.x_ld > 0 ? s.x_coff : 0;
}
inline int ngroups(const rn_conv_geom* g) { return g->groups > 1 ? g->groups : 1; }

int validate_geom(const rn_conv_seg* segs, int nseg, const rn_conv_geom* g) {
  RN_CHECK_ARG(segs && g, "conv: null argument");
  RN_CHECK_ARG(nseg >= 1 && nseg <= RN_MAX_SEG, "conv: nseg %d outside [1,%d]", nseg, RN_MAX_SEG);
  RN_CHECK_ARG(g->kh >= 1 && g->kw >= 1 && g->stride >= 1 && g->cin >= 1, "conv: bad geometry");
  RN_CHECK_ARG(g->groups >= 0 && (g->groups <= 1 || g->cin % g->groups == 0), "conv: cin %d not divisible by groups %d",
               g->cin, g->groups);
  if (g->groups > 1) {
    for (int s = 0; s < nseg; ++s) {
      RN_CHECK_ARG(segs[s].cout % g->groups == 0, "conv: cout %d not divisible by groups %d", segs[s].cout, g->groups);
    }
  }
  for (int s = 0; s < nseg; ++s)
    RN_CHECK_ARG(segs[s].x_ld == 0 || (segs[s].x_ld >= segs[s].x_coff + g->cin && segs[s].x_coff >= 0 &&
                                       segs[s].x_ld % 4 == 0 && segs[s].x_coff % 4 == 0),
                 "conv: bad x_ld/x_coff in segment %d", s);
  for (int s = 0; s < nseg; ++s) {
    RN_CHECK_ARG(segs[s].n >= 1 && segs[s].h >= 1 && segs[s].w >= 1 && segs[s].cout >= 1, "conv: bad segment %d", s);
    // the kernels address every tensor with 32-bit byte offsets below 2 GiB
    const double px = (double)segs[s].n * segs[s].h * segs[s].w * 4.0;
    int oh_, ow_, pt_, pl_;
    rn::same_pad(segs[s].h, g->kh, g->stride, &oh_, &pt_);
    rn::same_pad(segs[s].w, g->kw, g->stride, &ow_, &pl_);
    const double opx = (double)segs[s].n * oh_ * ow_ * 4.0;
    RN_UNSUPPORTED(px * g->cin >= 2147483648.0 || opx * segs[s].cout >= 2147483648.0 ||
                   (double)g->kh * g->kw * g->cin * segs[s].cout * 4.0 >= 2147483648.0,
                   "conv: a tensor of segment %d is >= 2 GiB", s);
    RN_UNSUPPORTED(segs[s].h >= 32768 || segs[s].w >= 32768, "conv: spatial size of segment %d too large", s);
  }
  return RN_OK;
}

}  // namespace

namespace {
// the list the current entry point was handed (rn::DeferScope): valid only for the duration of that call, per thread --
// nothing about a deferral outlives the call that records it
thread_local rn_reduce_list* tl_defer = nullptr;
}  // namespace

rn::DeferScope::DeferScope(rn_reduce_list* list) : prev_(tl_defer) { tl_defer = list; }
rn::DeferScope::~DeferScope() { tl_defer = (rn_reduce_list*)prev_; }

bool rn::reduce_deferred(hipStream_t) { return tl_defer != nullptr; }

int rn::launch_reduce_rows(const float* in, float* out, int64_t count, int nrows, int accumulate, hipStream_t st) {
  if (tl_defer) {
    rn_reduce_list* l = tl_defer;
    RN_CHECK_ARG(l->desc && l->capacity > 0 && l->count >= 0, "deferred reduction: bad list");
    RN_UNSUPPORTED(count >= ((int64_t)1 << 31) || nrows >= 65536, "deferred reduction: %lld x %d too large for a descriptor",
                   (long long)count, nrows);
    if (l->count >= l->capacity) {
      rn::set_error("deferred reduction: list full (%d entries)", l->capacity);
      return RN_EWORKSPACE;
    }
    l->desc[l->count++] = rn_reduce_desc{in, out, (int32_t)count, (uint16_t)nrows, (uint16_t)(accumulate ? 1 : 0)};
    return RN_OK;
  }
  hipLaunchKernelGGL(reduce_rows_kernel, dim3((unsigned)rn::ceil_div64(count, reduce_cols_per_block(nrows))), dim3(256), 0, st, in, out, count, nrows,
                     accumulate);
  RN_LAUNCH_CHECK();
  return RN_OK;
}

extern "C" int rn_flush_reductions(rn_reduce_list* list, rn_stream_t stream) {
  RN_CHECK_ARG(list && (list->count == 0 || list->desc) && list->count >= 0 && list->count <= list->capacity, "flush_reductions: bad list");
  hipStream_t st = (hipStream_t)stream;
  for (int first = 0; first < list->count; first += RN_MAX_REDUCE) {
    ReduceManyArgs a = {};
    a.n = list->count - first < RN_MAX_REDUCE ? list->count - first : RN_MAX_REDUCE;
    int blocks = 0;
    for (int i = 0; i < a.n; ++i) {
      const rn_reduce_desc& d = list->desc[first + i];
      a.d[i] = ReduceDesc{d.in, d.out, d.count, d.nrows, d.accumulate};
      a.block_start[i] = blocks;
      blocks += (int)rn::ceil_div64(a.d[i].count, reduce_cols_per_block(a.d[i].nrows));
    }
    a.block_start[a.n] = blocks;
    static const bool dump = getenv("RN_REDUCE_DBG") != nullptr;      // measurement aid: what a flush is made of
    if (dump)
      for (int i = 0; i < a.n; ++i)
        fprintf(stderr, "reduce[%d] cols %lld rows %d MB %.2f blocks %d\n", first + i, (long long)a.d[i].count, a.d[i].nrows,
                (double)a.d[i].count * a.d[i].nrows * 4e-6, a.block_start[i + 1 < a.n ? i + 1 : a.n] - a.block_start[i]);
    if (blocks > 0) hipLaunchKernelGGL(reduce_rows_many_kernel, dim3(blocks), dim3(256), 0, st, a);
  }
  RN_LAUNCH_CHECK();
  list->count = 0;
  return RN_OK;
}

extern "C" int rn_reduce_rows(const float* in, float* out, int64_t count, int nrows, int accumulate, rn_stream_t stream, rn_reduce_list* defer) {
  rn::DeferScope defer_scope_(defer);
  RN_CHECK_ARG(in && out && count >= 1 && nrows >= 1, "reduce_rows: bad argument");
  return rn::launch_reduce_rows(in, out, count, nrows, accumulate, (hipStream_t)stream);
}

extern "C" void rn_same_pad(int n, int k, int s, int* out, int* pad_before) { rn::same_pad(n, k, s, out, pad_before); }

namespace {
struct Batch { int n; long bs_a, bs_b, bs_out; };
// what a dgrad / wgrad call WOULD launch (filled instead of launching when requested): lets rn_conv2d_bwd put both
// gradients of a convolution into one launch
struct Planned {
  ConvArgs a;
  int cfg, blocks;
  bool vec, tapu, simple;  // simple: one plain kernel (no split-K, no phase decomposition, no batch)
  int nsplit; bool direct; int64_t count;  // wgrad: the row reduction that follows
};
// split-K scratch: ws == nullptr -> never split; need_out != nullptr -> dry run, only report the bytes split-K wants
struct Scratch { void* ws; size_t bytes; size_t* need_out; };
// rows request of rn_conv2d_fwd_stats: `rows` the caller's buffer, or (dry run) layout_out = the layout + bytes_out (0: cannot)
struct StatReq { const rn_gn_rows* rows; int groups; rn_gn_rows* layout_out; size_t* bytes_out; };
int conv_fwd_impl(const rn_conv_seg* segs, int nseg, const rn_conv_geom* g, const Batch& bt, rn_stream_t stream,
                  const Scratch& sc = Scratch{nullptr, 0, nullptr}, const StatReq& sr = StatReq{nullptr, 0, nullptr, nullptr});
int conv_dgrad_impl(const rn_conv_seg* segs, int nseg, const rn_conv_geom* g, const Batch& bt, rn_stream_t stream,
                    const Scratch& sc = Scratch{nullptr, 0, nullptr}, Planned* plan = nullptr);

// Tiny grids with a long reduction (the stride-2 convs that make P6 / P7, the 4x4 and 8x8 maps: 4..24 tiles x 72+
// K-tiles) are latency-bound on a handful of CUs: split K over ~384/tiles blocks per tile and sum the partial
// outputs with the fixed-order row reduction.
void plan_splitk(int tiles, int nk, int* nsplit, int* ksplit) {
  *nsplit = 1; *ksplit = 0;
  if (tiles >= 96 || nk < 16 || getenv("RN_NO_SPLITK")) return;
  int ns = (384 + tiles - 1) / tiles;
  if (ns > nk / 4) ns = nk / 4;
  if (ns < 2) return;
  const int ks = (nk + ns - 1) / ns;
  *ksplit = ks;
  *nsplit = (nk + ks - 1) / ks;
}
}  // namespace

extern "C" size_t rn_conv2d_fwd_workspace(const rn_conv_seg* segs, int nseg, const rn_conv_geom* g) {
  size_t need = 0;
  if (conv_fwd_impl(segs, nseg, g, Batch{1, 0, 0, 0}, nullptr, Scratch{nullptr, 0, &need})) return 0;
  return need;
}
extern "C" size_t rn_conv2d_dgrad_workspace(const rn_conv_seg* segs, int nseg, const rn_conv_geom* g) {
  size_t need = 0;
  if (conv_dgrad_impl(segs, nseg, g, Batch{1, 0, 0, 0}, nullptr, Scratch{nullptr, 0, &need})) return 0;
  return need;
}
extern "C" int rn_conv2d_fwd(const rn_conv_seg* segs, int nseg, const rn_conv_geom* g, void* workspace, size_t workspace_bytes,
                             rn_stream_t stream) {
  return conv_fwd_impl(segs, nseg, g, Batch{1, 0, 0, 0}, stream, Scratch{workspace, workspace_bytes, nullptr});
}
extern "C" size_t rn_conv2d_stats_rows(const rn_conv_seg* segs, int nseg, const rn_conv_geom* g, size_t workspace_bytes, int groups,
                                       rn_gn_rows* layout) {
  size_t bytes = 0;
  // (a non-null scratch pointer of the caller's size, never dereferenced in a dry run: the split-K decision depends on it)
  if (conv_fwd_impl(segs, nseg, g, Batch{1, 0, 0, 0}, nullptr, Scratch{workspace_bytes ? (void*)segs : nullptr, workspace_bytes, nullptr},
                    StatReq{nullptr, groups, layout, &bytes}))
    return 0;
  return bytes;
}
extern "C" int rn_conv2d_fwd_stats(const rn_conv_seg* segs, int nseg, const rn_conv_geom* g, void* workspace, size_t workspace_bytes,
                                   const rn_gn_rows* rows, rn_stream_t stream) {
  RN_CHECK_ARG(rows, "conv fwd stats: null rows");
  return conv_fwd_impl(segs, nseg, g, Batch{1, 0, 0, 0}, stream, Scratch{workspace, workspace_bytes, nullptr}, StatReq{rows, 0, nullptr, nullptr});
}
// conv -> Dropout (-> statistics of the dropped output) in one launch: dense 1x1 / stride-1 convs of one tensor on the split-bf16
// kernels (product mode 1).  See include/rn_hip.h.
namespace {
int fused_dropout_rows(const rn_conv_seg* segs, int nseg, const rn_conv_geom* g, int* ohw_out, long* m_out, int* x_ld, int* x_coff);
}
extern "C" size_t rn_conv2d_dropout_rows(const rn_conv_seg* segs, int nseg, const rn_conv_geom* g, int groups, rn_gn_rows* layout, int* fused_ok) {
  if (fused_ok) *fused_ok = 0;
  int ohw, ld, coff; long m;
  const int xrows = fused_dropout_rows(segs, nseg, g, &ohw, &m, &ld, &coff);
  if (!xrows) return 0;
  if (fused_ok) *fused_ok = 1;
  const int cout = segs[0].cout;
  const bool ok = ohw % xrows == 0 && groups >= 1 && cout % groups == 0 && (double)ohw * (cout / groups) < 16777216.0 &&
                  rn_group_norm_rows_ok(cout, groups, ohw / xrows, 0);
  if (!ok) return 0;
  if (layout) { layout->rows_per_sample = ohw / xrows; layout->per_group = 0; layout->groups = groups; }
  return (size_t)(m / xrows) * cout * 8;
}
extern "C" int rn_conv2d_fwd_dropout(const rn_conv_seg* segs, int nseg, const rn_conv_geom* g, float rate, uint64_t seed, const uint64_t* seed_dev,
                                     const rn_gn_rows* rows, rn_stream_t stream) {
  RN_CHECK_ARG(rate >= 0.f && rate < 1.f, "conv + dropout: rate %g outside [0, 1)", (double)rate);
  int ohw, ld, coff; long m;
  const int xrows = fused_dropout_rows(segs, nseg, g, &ohw, &m, &ld, &coff);
  RN_UNSUPPORTED(!xrows, "conv + dropout: not a dense 1x1 / stride-1 conv the split-bf16 kernels take (rn_conv2d_dropout_rows: fused_ok == 0)");
  RN_CHECK_ARG(segs[0].x && segs[0].wgt && segs[0].y, "conv + dropout: null pointer in segment 0");
  float2* r = nullptr;
  if (rows) {
    rn_gn_rows want = {};
    RN_UNSUPPORTED(!rn_conv2d_dropout_rows(segs, nseg, g, rows->groups, &want, nullptr) || want.rows_per_sample != rows->rows_per_sample ||
                       rows->per_group != 0, "conv + dropout: this shape / layout cannot produce GroupNorm rows (rn_conv2d_dropout_rows)");
    RN_CHECK_ARG(rows->rows, "conv + dropout: null rows");
    r = (float2*)rows->rows;
  }
  return rn::launch_conv1x1_fwd_x3(segs[0].x + coff, ld, segs[0].wgt, segs[0].y, (int)m, g->cin, segs[0].cout, r, (hipStream_t)stream, rate, seed, seed_dev);
}

extern "C" int rn_conv2d_dgrad(const rn_conv_seg* segs, int nseg, const rn_conv_geom* g, void* workspace, size_t workspace_bytes,
                               rn_stream_t stream) {
  return conv_dgrad_impl(segs, nseg, g, Batch{1, 0, 0, 0}, stream, Scratch{workspace, workspace_bytes, nullptr});
}

// C_b [M x N] = A_b [M x K] * B_b   for b = 0..nbatch-1 in one launch (Winograd's per-point products).
// b_nk == 0: B_b is [K x N] (forward kernel);  b_nk != 0: B_b is [N x K] (the data-gradient kernel's layout).
int rn::launch_batched_gemm(const float* A, const float* B, float* C, int M, int K, int N, int nbatch, int b_nk,
                            hipStream_t st) {
  if (rn::product_mode() == 1 && rn::gemm_x3_ok(M, K, N)) return rn::launch_batched_gemm_x3(A, B, C, M, K, N, nbatch, b_nk, st);
  rn_conv_seg sg = {};
  sg.n = 1; sg.h = 1; sg.w = M; sg.wgt = B;
  rn_conv_geom g1 = {1, 1, 1, b_nk ? N : K, 1};
  const Batch bt = {nbatch, (long)M * K, (long)K * N, (long)M * N};
  if (b_nk) {
    sg.dy = A; sg.dx = C; sg.cout = K;
    return conv_dgrad_impl(&sg, 1, &g1, bt, (rn_stream_t)st);
  }
  sg.x = A; sg.y = C; sg.cout = N;
  return conv_fwd_impl(&sg, 1, &g1, bt, (rn_stream_t)st);
}

namespace {
// Dense 1x1 / stride-1 convolutions of ONE tensor are plain products x [M x Cin] W [Cin x Cout]: in product mode 1 they run on
// the split-bf16 kernels of gemm_x3.hip (round 6) where the grid is large enough to need no split-K (the tiny maps keep the
// fp32 split-K path).  Returns the m-tile rows (64 / 128) or 0.
int x3_conv1x1(const rn_conv_seg* segs, int nseg, const rn_conv_geom* g, const Batch& bt, long m, int x_ld, bool with_bias) {
  if (nseg != 1 || bt.n != 1 || ngroups(g) != 1 || g->kh != 1 || g->kw != 1 || g->stride != 1 || with_bias) return 0;
  if (g->cin % 4 || segs[0].cout % 4 || x_ld % 4 || (segs[0].x_ld > 0 && segs[0].x_coff % 4)) return 0;     // 16-byte operand pieces and stores
  const int rows = rn::conv1x1_x3_tile(m, g->cin, segs[0].cout, x_ld);
  if (!rows) return 0;
  const long tiles = rn::ceil_div64(m, rows) * rn::ceil_div(segs[0].cout, rows);
  return tiles >= 96 ? rows : 0;
}

// m-tile rows of the fused conv + dropout launch (0: the shape is not taken)
int fused_dropout_rows(const rn_conv_seg* segs, int nseg, const rn_conv_geom* g, int* ohw_out, long* m_out, int* x_ld, int* x_coff) {
  if (validate_geom(segs, nseg, g)) return 0;
  SegDev v = {};
  set_x_view(v, segs[0], g->cin);
  const long m = (long)segs[0].n * segs[0].h * segs[0].w;
  *ohw_out = segs[0].h * segs[0].w; *m_out = m; *x_ld = v.x_ld; *x_coff = v.x_coff;
  return x3_conv1x1(segs, nseg, g, Batch{1, 0, 0, 0}, m, v.x_ld, segs[0].bias != nullptr);
}

// Dense k x k convs that are not plain products (3x3 / stride 2 ...) through an explicit patch matrix on the split-bf16 kernels
// (im2col.hip); returns the patch matrix's bytes (0: not taken).  Dense x / dx only, one tensor, no bias.
size_t x3_im2col(const rn_conv_seg* segs, int nseg, const rn_conv_geom* g, const Batch& bt, bool with_bias) {
  if (nseg != 1 || bt.n != 1 || ngroups(g) != 1 || with_bias) return 0;
  if (segs[0].x_ld > 0 && (segs[0].x_ld != g->cin || segs[0].x_coff != 0)) return 0;
  return rn::im2col_x3_bytes(segs[0].n, segs[0].h, segs[0].w, g->cin, segs[0].cout, g->kh, g->kw, g->stride);
}
// Grouped 3x3 / stride-1 convs with 4 .. 32 channels per group (the ResNeXt bottlenecks' conv 2): direct kernels of grouped_conv.hip
bool gconv_direct(const rn_conv_seg* segs, int nseg, const rn_conv_geom* g, const Batch& bt, bool with_bias) {
  if (nseg != 1 || bt.n != 1 || ngroups(g) < 2 || g->kh != 3 || g->kw != 3 || g->stride != 1 || with_bias) return false;
  if (segs[0].cout != g->cin || (segs[0].x_ld > 0 && (segs[0].x_ld != g->cin || segs[0].x_coff != 0))) return false;
  return rn::gconv3x3_ok(segs[0].n, segs[0].h, segs[0].w, g->cin, ngroups(g));
}

int conv_fwd_impl(const rn_conv_seg* segs, int nseg, const rn_conv_geom* g, const Batch& bt, rn_stream_t stream,
                  const Scratch& sc, const StatReq& sr) {
  if (int e = validate_geom(segs, nseg, g)) return e;
  if (!sr.rows && !sr.bytes_out && gconv_direct(segs, nseg, g, bt, segs[0].bias != nullptr)) {
    if (sc.need_out) { *sc.need_out = 0; return RN_OK; }
    RN_CHECK_ARG(segs[0].x && segs[0].wgt && segs[0].y, "conv fwd: null pointer in segment 0");
    return rn::launch_gconv3x3(segs[0].x, segs[0].wgt, segs[0].y, segs[0].n, segs[0].h, segs[0].w, g->cin, ngroups(g), 0, (hipStream_t)stream);
  }
  RN_CHECK_ARG(bt.n == 1 || nseg == 1, "conv: batched mode takes one segment");
  ConvArgs a = {};
  a.nbatch = bt.n; a.bs_a = bt.bs_a; a.bs_b = bt.bs_b; a.bs_out = bt.bs_out;
  const int G = ngroups(g);
  a.nseg = nseg; a.kh = g->kh; a.kw = g->kw; a.stride = g->stride; a.cin = g->cin;
  a.groups = G; a.cin_g = g->cin / G;
  bool vec = (a.cin_g % 4 == 0);
  for (int s = 0; s < nseg; ++s) {
    RN_CHECK_ARG(sc.need_out || (segs[s].x && segs[s].wgt && segs[s].y), "conv fwd: null pointer in segment %d", s);
    SegDev& d = a.seg[s];
    d.a = segs[s].x; d.b = segs[s].wgt; d.bias = segs[s].bias; d.out = segs[s].y;
    d.n = segs[s].n; d.h = segs[s].h; d.w = segs[s].w; d.cout = segs[s].cout;
    rn::same_pad(d.h, g->kh, g->stride, &d.oh, &d.pad_t);
    rn::same_pad(d.w, g->kw, g->stride, &d.ow, &d.pad_l);
    d.m = d.n * d.oh * d.ow;
    set_x_view(d, segs[s], g->cin);
    vec = vec && ((d.cout / G) % 4 == 0);
  }
  {
    // the stem (see stem_conv_fwd_kernel): its own direct kernel, rows of 256 pixels
    static const bool stem_on = !(getenv("RN_STEM_DIRECT") && atoi(getenv("RN_STEM_DIRECT")) == 0);
    const SegDev& d = a.seg[0];
    if (stem_on && nseg == 1 && bt.n == 1 && G == 1 && g->kh == 3 && g->kw == 3 && g->cin == 3 && g->stride == 2 && d.cout == 32 &&
        !d.bias && d.x_ld == 3 && d.x_coff == 0 && d.ow % STEM_RUN == 0 && (d.oh * d.ow) % STEM_PPB == 0 && (long)d.m * 32 < (1l << 31)) {
      const int ohw = d.oh * d.ow;
      if (sc.need_out) { *sc.need_out = 0; return RN_OK; }
      float2* rows = nullptr;
      if (sr.rows || sr.bytes_out) {
        const int groups = sr.rows ? sr.rows->groups : sr.groups;
        const bool ok = groups >= 1 && 32 % groups == 0 && rn_group_norm_rows_ok(32, groups, ohw / STEM_PPB, 0);
        if (sr.bytes_out) {
          *sr.bytes_out = ok ? (size_t)(d.m / STEM_PPB) * 32 * 8 : 0;
          if (ok && sr.layout_out) { sr.layout_out->rows_per_sample = ohw / STEM_PPB; sr.layout_out->per_group = 0; sr.layout_out->groups = groups; }
          return RN_OK;
        }
        RN_UNSUPPORTED(!ok || sr.rows->rows_per_sample != ohw / STEM_PPB || sr.rows->per_group != 0,
                       "conv fwd stats: this shape / layout cannot produce GroupNorm rows (rn_conv2d_stats_rows)");
        RN_CHECK_ARG(sr.rows->rows, "conv fwd stats: null rows");
        rows = (float2*)sr.rows->rows;
      }
      StemArgs st_a = {d.a, d.b, d.out, rows, d.n, d.h, d.w, d.oh, d.ow, d.pad_t, d.pad_l};
      hipLaunchKernelGGL(stem_conv_fwd_kernel, dim3((unsigned)(d.m / STEM_PPB)), dim3(256), 0, (hipStream_t)stream, st_a);
      RN_LAUNCH_CHECK();
      return RN_OK;
    }
  }
  // dense convs of one tensor as split-bf16 products (product mode 1): 1x1 / stride 1 directly on x, any other kernel through the patch
  // matrix written into the caller's scratch (rn_conv2d_fwd_workspace asks for it; a caller that brings none keeps the fp32 kernels)
  int xrows = x3_conv1x1(segs, nseg, g, bt, a.seg[0].m, a.seg[0].x_ld, a.seg[0].bias != nullptr);
  size_t colb = 0;
  int n_piece = 0;                            // the patch-matrix path runs the batch in equal pieces of this many samples (<= 1 GiB each)
  if (!xrows && nseg == 1 && bt.n == 1 && G == 1 && !a.seg[0].bias && a.seg[0].x_ld == g->cin && a.seg[0].x_coff == 0 &&
      (colb = rn::im2col_x3_fwd_pieces(a.seg[0].n, a.seg[0].h, a.seg[0].w, g->cin, a.seg[0].cout, g->kh, g->kw, g->stride, &n_piece)) != 0) {
    if (sc.need_out || (sc.ws && sc.bytes >= colb))
      xrows = rn::conv1x1_x3_tile((long)n_piece * a.seg[0].oh * a.seg[0].ow, g->kh * g->kw * g->cin, a.seg[0].cout, g->kh * g->kw * g->cin);
    else colb = 0;
  }
  if (xrows) {
    const SegDev& d = a.seg[0];
    const int ohw = d.oh * d.ow;
    bool take = true;
    float2* rows = nullptr;
    if (sr.rows || sr.bytes_out) {
      const int groups = sr.rows ? sr.rows->groups : sr.groups;
      const bool ok = ohw % xrows == 0 && groups >= 1 && d.cout % groups == 0 && (double)ohw * (d.cout / groups) < 16777216.0 &&
                      rn_group_norm_rows_ok(d.cout, groups, ohw / xrows, 0);
      if (!ok) {
        take = false;                       // (the fp32 kernels' tiles may still fit the map: fall through)
      } else if (sr.bytes_out) {
        *sr.bytes_out = (size_t)(d.m / xrows) * d.cout * 8;
        if (sr.layout_out) { sr.layout_out->rows_per_sample = ohw / xrows; sr.layout_out->per_group = 0; sr.layout_out->groups = groups; }
        return RN_OK;
      } else {
        RN_UNSUPPORTED(sr.rows->rows_per_sample != ohw / xrows || sr.rows->per_group != 0,
                       "conv fwd stats: this shape / layout cannot produce GroupNorm rows (rn_conv2d_stats_rows)");
        RN_CHECK_ARG(sr.rows->rows, "conv fwd stats: null rows");
        rows = (float2*)sr.rows->rows;
      }
    }
    if (take) {
      if (sc.need_out) { *sc.need_out = colb; return RN_OK; }
      if (!colb) return rn::launch_conv1x1_fwd_x3(d.a + d.x_coff, d.x_ld, d.b, d.out, d.m, g->cin, d.cout, rows, (hipStream_t)stream);
      const int K = g->kh * g->kw * g->cin;                    // W in HWIO is the [K][cout] matrix
      const int mp = n_piece * ohw;                            // rows of a piece (a whole number of m-tiles when rows are asked for)
      for (int s0 = 0; s0 < d.n; s0 += n_piece) {
        if (int e = rn::launch_im2col(d.a + (size_t)s0 * d.h * d.w * g->cin, (float*)sc.ws, n_piece, d.h, d.w, g->cin, g->kh, g->kw, g->stride,
                                      (hipStream_t)stream))
          return e;
        if (int e = rn::launch_conv1x1_fwd_x3((const float*)sc.ws, K, d.b, d.out + (size_t)s0 * ohw * d.cout, mp, K, d.cout,
                                              rows ? rows + (size_t)(s0 * (ohw / xrows)) * d.cout : nullptr, (hipStream_t)stream))
          return e;
      }
      return RN_OK;
    }
  }
  const int c = (G > 1 && a.seg[0].cout / G <= 64)
                    ? cfg_for_group_width(a.seg[0].cout / G)
                    : choose_cfg([&](int s, long* m, long* n) { *m = a.seg[s].m * G; *n = a.seg[s].cout / G; }, nseg, bt.n);
  a.tpg = G > 1 ? rn::ceil_div(a.seg[0].cout / G, kCfgs[c].bn) : (1 << 20);  // dense: every N-tile is "group 0"
  int tiles = 0;
  for (int s = 0; s < nseg; ++s) {
    SegDev& d = a.seg[s];
    RN_UNSUPPORTED(G > 1 && d.cout != a.seg[0].cout, "conv: grouped segments must share cout");
    d.tiles_n = G > 1 ? G * a.tpg : rn::ceil_div(d.cout, kCfgs[c].bn);
    d.start = tiles;
    tiles += rn::ceil_div(d.m, kCfgs[c].bm) * d.tiles_n;
  }
  a.btiles = tiles;
  tiles *= bt.n;
  hipStream_t st = (hipStream_t)stream;
  const bool tapu = vec && (a.cin_g % BK == 0);
  // split-K (single dense segment only)
  int nsplit = 1, ksplit = 0;
  float* const y_final = a.seg[0].out;
  const int64_t out_elems = (int64_t)a.seg[0].m * a.seg[0].cout;
  if (bt.n == 1 && nseg == 1 && G == 1) plan_splitk(tiles, rn::ceil_div(g->kh * g->kw * g->cin, BK), &nsplit, &ksplit);
  if (sc.need_out) {
    *sc.need_out = nsplit > 1 ? (size_t)nsplit * out_elems * sizeof(float) : 0;
    return RN_OK;
  }
  if (nsplit > 1 && sc.ws && sc.bytes >= (size_t)nsplit * out_elems * sizeof(float)) {
    a.ksplit = ksplit; a.nbatch = nsplit; a.bs_a = 0; a.bs_b = 0; a.bs_out = out_elems;
    a.seg[0].out = (float*)sc.ws;
    tiles *= nsplit;
  } else {
    nsplit = 1;
  }
  if (sr.rows || sr.bytes_out) {
    // rows ride on the plain single-segment kernel only, and the tile rows of a sample must be whole m-tiles
    const SegDev& d = a.seg[0];
    const int ohw = d.oh * d.ow, bm = kCfgs[c].bm;
    const int groups = sr.rows ? sr.rows->groups : sr.groups;
    const bool ok = nseg == 1 && bt.n == 1 && G == 1 && nsplit == 1 && !d.bias && ohw % bm == 0 && groups >= 1 && d.cout % groups == 0 &&
                    (double)ohw * (d.cout / groups) < 16777216.0 &&
                    rn_group_norm_rows_ok(d.cout, groups, ohw / bm, 0);
    if (sr.bytes_out) {
      *sr.bytes_out = ok ? (size_t)(d.m / bm) * d.cout * 8 : 0;
      if (ok && sr.layout_out) { sr.layout_out->rows_per_sample = ohw / bm; sr.layout_out->per_group = 0; sr.layout_out->groups = groups; }
      return RN_OK;
    }
    RN_UNSUPPORTED(!ok || sr.rows->rows_per_sample != ohw / bm || sr.rows->per_group != 0,
                   "conv fwd stats: this shape / layout cannot produce GroupNorm rows (rn_conv2d_stats_rows)");
    RN_CHECK_ARG(sr.rows->rows, "conv fwd stats: null rows");
    a.st.rows = (float2*)sr.rows->rows; a.st.groups = groups; a.st.cpg = d.cout / groups;
  }
  static const int fpad_env = getenv("RN_PROD_LDS_PAD_FWD") ? atoi(getenv("RN_PROD_LDS_PAD_FWD")) : 0;
  static const int fpad_only_m = getenv("RN_PROD_LDS_PAD_FWD_ONLY_M") ? atoi(getenv("RN_PROD_LDS_PAD_FWD_ONLY_M")) : 0;   // (probe: only products of this many rows)
  const int fpad = (bt.n > 1 && (!fpad_only_m || a.seg[0].m == fpad_only_m)) ? fpad_env : 0;      // (batched = the Winograd products; see RN_PROD_LDS_PAD_BWD)
#define RN_FWD(BM_, BN_, WM_, WN_)                                                                   \
  do {                                                                                               \
    if (tapu) hipLaunchKernelGGL((conv_fwd_kernel<BM_, BN_, WM_, WN_, 4, true>), dim3(tiles), dim3(WM_* WN_ * 64), fpad, st, a); \
    else if (vec) hipLaunchKernelGGL((conv_fwd_kernel<BM_, BN_, WM_, WN_, 4, false>), dim3(tiles), dim3(WM_* WN_ * 64), 0, st, a); \
    else hipLaunchKernelGGL((conv_fwd_kernel<BM_, BN_, WM_, WN_, 1, false>), dim3(tiles), dim3(WM_* WN_ * 64), 0, st, a);     \
  } while (0)
  switch (c) {
    case 0: RN_FWD(128, 128, 2, 2); break;
    case 1: RN_FWD(128, 64, 2, 2); break;
    case 2: RN_FWD(64, 64, 2, 2); break;
    default: RN_FWD(128, 32, 4, 1); break;
  }
#undef RN_FWD
  RN_LAUNCH_CHECK();
  if (nsplit > 1) {
    hipLaunchKernelGGL(reduce_rows_kernel, dim3((unsigned)rn::ceil_div64(out_elems, reduce_cols_per_block(nsplit))), dim3(256), 0, st, (const float*)sc.ws,
                       y_final, out_elems, nsplit, 0);
    RN_LAUNCH_CHECK();
  }
  return RN_OK;
}

int conv_dgrad_impl(const rn_conv_seg* segs, int nseg, const rn_conv_geom* g, const Batch& bt, rn_stream_t stream,
                    const Scratch& sc, Planned* plan) {
  if (int e = validate_geom(segs, nseg, g)) return e;
  if (!plan) {
    if (const size_t colb = x3_im2col(segs, nseg, g, bt, false)) {
      if (sc.need_out) { *sc.need_out = colb; return RN_OK; }
      if (sc.ws && sc.bytes >= colb) {
        RN_CHECK_ARG(segs[0].dy && segs[0].wgt && segs[0].dx, "conv dgrad: null pointer in segment 0");
        int oh, ow, pt, pl;
        rn::same_pad(segs[0].h, g->kh, g->stride, &oh, &pt);
        rn::same_pad(segs[0].w, g->kw, g->stride, &ow, &pl);
        const int M = segs[0].n * oh * ow, K = g->kh * g->kw * g->cin;
        // dcol [M][K] = dy [M][cout] W^T  (W [K][cout]: the product's [N = K][k = cout] operand), then the gather
        if (int e = rn::launch_conv1x1_dgrad_x3(segs[0].dy, segs[0].wgt, (float*)sc.ws, K, M, K, segs[0].cout, (hipStream_t)stream)) return e;
        return rn::launch_col2im((const float*)sc.ws, segs[0].dx, segs[0].n, segs[0].h, segs[0].w, g->cin, g->kh, g->kw, g->stride, (hipStream_t)stream);
      }
    }
  }
  if (!plan && gconv_direct(segs, nseg, g, bt, false)) {       // the same direct kernel on dy, the kernel rotated / transposed per group
    if (sc.need_out) { *sc.need_out = 0; return RN_OK; }
    RN_CHECK_ARG(segs[0].dy && segs[0].wgt && segs[0].dx, "conv dgrad: null pointer in segment 0");
    return rn::launch_gconv3x3(segs[0].dy, segs[0].wgt, segs[0].dx, segs[0].n, segs[0].h, segs[0].w, g->cin, ngroups(g), 1, (hipStream_t)stream);
  }
  RN_CHECK_ARG(bt.n == 1 || nseg == 1, "conv: batched mode takes one segment");
  ConvArgs a = {};
  a.nbatch = bt.n; a.bs_a = bt.bs_a; a.bs_b = bt.bs_b; a.bs_out = bt.bs_out;
  const int G = ngroups(g);
  a.nseg = nseg; a.kh = g->kh; a.kw = g->kw; a.stride = g->stride; a.cin = g->cin;
  a.groups = G; a.cin_g = g->cin / G;
  bool vec = true;  // float4 gathers run along cout
  for (int s = 0; s < nseg; ++s) {
    RN_CHECK_ARG(sc.need_out || (segs[s].dy && segs[s].wgt && segs[s].dx), "conv dgrad: null pointer in segment %d", s);
    vec = vec && ((segs[s].cout / G) % 4 == 0);
    SegDev& d = a.seg[s];
    d.a = segs[s].dy; d.b = segs[s].wgt; d.bias = nullptr; d.out = segs[s].dx;
    d.n = segs[s].n; d.h = segs[s].h; d.w = segs[s].w; d.cout = segs[s].cout;
    rn::same_pad(d.h, g->kh, g->stride, &d.oh, &d.pad_t);
    rn::same_pad(d.w, g->kw, g->stride, &d.ow, &d.pad_l);
    d.m = d.n * d.h * d.w;
    set_x_view(d, segs[s], g->cin);
  }
  if (!plan && x3_conv1x1(segs, nseg, g, bt, a.seg[0].m, a.seg[0].x_ld, false)) {      // (a `plan` request = the merged kernel: see rn_conv2d_bwd)
    if (sc.need_out) { *sc.need_out = 0; return RN_OK; }
    const SegDev& d = a.seg[0];
    return rn::launch_conv1x1_dgrad_x3(d.a, d.b, d.out + d.x_coff, d.x_ld, d.m, g->cin, d.cout, (hipStream_t)stream);
  }
  // stride 2: an input pixel only sees the taps of its own row / column parity -- the plain implicit GEMM multiplies
  // 3/4 structural zeros (ResNeXt's three 3x3/2 identity convs alone: 3 x 142 of 189 GFLOP per cfg-3 step).  Phase
  // decomposition: every segment becomes up to four pseudo-segments, one per (row, column) parity, each with its own
  // tap subset.  Large maps only (small ones are latency-bound and take the split-K path instead).
  long total_rows = 0;
  for (int s = 0; s < nseg; ++s) total_rows += a.seg[s].m;
  if (g->stride == 2 && bt.n == 1 && nseg * 4 <= RN_MAX_SEG && total_rows >= 4096 && !getenv("RN_NO_PHASE_DGRAD")) {
    SegDev src[RN_MAX_SEG];
    for (int s = 0; s < nseg; ++s) src[s] = a.seg[s];
    int k = 0;
    for (int s = 0; s < nseg; ++s) {
      for (int py = 0; py < 2; ++py) {
        for (int px = 0; px < 2; ++px) {
          SegDev d = src[s];
          d.par = 1; d.py = py; d.px = px;
          d.hc = (d.h - py + 1) / 2; d.wc = (d.w - px + 1) / 2;
          if (d.hc <= 0 || d.wc <= 0) continue;
          d.kh0 = (py + d.pad_t) & 1; d.kw0 = (px + d.pad_l) & 1;
          d.nkh = d.kh0 < g->kh ? (g->kh - d.kh0 + 1) / 2 : 0;
          d.nkw = d.kw0 < g->kw ? (g->kw - d.kw0 + 1) / 2 : 0;
          if (d.nkh == 0 || d.nkw == 0) { d.nkh = 0; d.nkw = 1; }  // no taps: the phase is all zeros
          d.m = d.n * d.hc * d.wc;
          a.seg[k++] = d;
        }
      }
    }
    nseg = k;
    a.nseg = k;
  }
  const int cin_g = a.cin_g;
  const int c = (G > 1 && cin_g <= 64) ? cfg_for_group_width(cin_g)
                                        : choose_cfg([&](int s, long* m, long* n) { *m = a.seg[s].m * G; *n = cin_g; }, nseg, bt.n);
  a.tpg = rn::ceil_div(cin_g, kCfgs[c].bn);
  int tiles = 0;
  for (int s = 0; s < nseg; ++s) {
    SegDev& d = a.seg[s];
    d.tiles_n = G * a.tpg;
    d.start = tiles;
    tiles += rn::ceil_div(d.m, kCfgs[c].bm) * d.tiles_n;
  }
  a.btiles = tiles;
  tiles *= bt.n;
  hipStream_t st = (hipStream_t)stream;
  bool tapu = vec;
  for (int s = 0; s < nseg; ++s) tapu = tapu && ((a.seg[s].cout / G) % BK == 0);
  // split-K (single segment writing a dense dx only)
  int nsplit = 1, ksplit = 0;
  float* const dx_final = a.seg[0].out;
  const int64_t out_elems = (int64_t)a.seg[0].m * g->cin;
  if (bt.n == 1 && nseg == 1 && !a.seg[0].par && G == 1 && a.seg[0].x_ld == g->cin && a.seg[0].x_coff == 0)
    plan_splitk(tiles, rn::ceil_div(g->kh * g->kw * segs[0].cout, BK), &nsplit, &ksplit);
  if (sc.need_out) {
    *sc.need_out = nsplit > 1 ? (size_t)nsplit * out_elems * sizeof(float) : 0;
    return RN_OK;
  }
  if (nsplit > 1 && sc.ws && sc.bytes >= (size_t)nsplit * out_elems * sizeof(float)) {
    a.ksplit = ksplit; a.nbatch = nsplit; a.bs_a = 0; a.bs_b = 0; a.bs_out = out_elems;
    a.seg[0].out = (float*)sc.ws;
    tiles *= nsplit;
  } else {
    nsplit = 1;
  }
  if (plan) {
    plan->a = a; plan->cfg = c; plan->blocks = tiles; plan->vec = vec; plan->tapu = tapu;
    plan->simple = nsplit == 1 && !a.seg[0].par && bt.n == 1;
    return RN_OK;
  }
#define RN_DG(BM_, BN_, WM_, WN_)                                                                    \
  do {                                                                                               \
    if (tapu) hipLaunchKernelGGL((conv_dgrad_kernel<BM_, BN_, WM_, WN_, 4, true>), dim3(tiles), dim3(WM_* WN_ * 64), 0, st, a); \
    else if (vec) hipLaunchKernelGGL((conv_dgrad_kernel<BM_, BN_, WM_, WN_, 4, false>), dim3(tiles), dim3(WM_* WN_ * 64), 0, st, a); \
    else hipLaunchKernelGGL((conv_dgrad_kernel<BM_, BN_, WM_, WN_, 1, false>), dim3(tiles), dim3(WM_* WN_ * 64), 0, st, a);     \
  } while (0)
  switch (c) {
    case 0: RN_DG(128, 128, 2, 2); break;
    case 1: RN_DG(128, 64, 2, 2); break;
    case 2: RN_DG(64, 64, 2, 2); break;
    default: RN_DG(128, 32, 4, 1); break;
  }
#undef RN_DG
  RN_LAUNCH_CHECK();
  if (nsplit > 1) {
    hipLaunchKernelGGL(reduce_rows_kernel, dim3((unsigned)rn::ceil_div64(out_elems, reduce_cols_per_block(nsplit))), dim3(256), 0, st, (const float*)sc.ws,
                       dx_final, out_elems, nsplit, 0);
    RN_LAUNCH_CHECK();
  }
  return RN_OK;
}
}  // namespace

namespace {
struct WgradPlan {
  int cfg, tiles_m, tiles_n, tpg, nsplit, ktotal, cout;
  int chunk[RN_MAX_SEG], start[RN_MAX_SEG], pixels[RN_MAX_SEG], oh[RN_MAX_SEG], ow[RN_MAX_SEG], pt[RN_MAX_SEG],
      pl[RN_MAX_SEG];
};

int plan_wgrad(const rn_conv_seg* segs, int nseg, const rn_conv_geom* g, WgradPlan* p, int nbatch = 1) {
  const int G = ngroups(g);
  p->ktotal = g->kh * g->kw * (g->cin / G);
  p->cout = segs[0].cout;
  long total_pixels = 0;
  for (int s = 0; s < nseg; ++s) {
    RN_CHECK_ARG(segs[s].cout == p->cout, "conv wgrad: segments must share cout (one kernel tensor)");
    rn::same_pad(segs[s].h, g->kh, g->stride, &p->oh[s], &p->pt[s]);
    rn::same_pad(segs[s].w, g->kw, g->stride, &p->ow[s], &p->pl[s]);
    p->pixels[s] = segs[s].n * p->oh[s] * p->ow[s];
    total_pixels += p->pixels[s];
  }
  const long kt = p->ktotal, co = p->cout;
  p->cfg = choose_cfg([&](int, long* m, long* n) { *m = kt; *n = co; }, 1);
  // wgrad output tiles are few: the fill comes from splitting the reduction, so prefer the
  // shape with the least padding only.
  {
    double best = 1e300;
    for (int c = 0; c < kNumCfg; ++c) {
      double w = (double)rn::ceil_div((int)kt, kCfgs[c].bm) * kCfgs[c].bm * rn::ceil_div((int)co / G, kCfgs[c].bn) *
                 kCfgs[c].bn * kCfgs[c].penalty;
      if (w < best) { best = w; p->cfg = c; }
    }
    if (nbatch > 1) p->cfg = 2;  // batched (Winograd) products: many small outputs, 64x64 measured best
    if (const char* force = getenv("RN_WGRAD_CFG")) {  // tuning aid
      const int c = atoi(force);
      if (c >= 0 && c < kNumCfg) p->cfg = c;
    }
    if (G > 1 && p->cout / G <= 64) p->cfg = cfg_for_group_width(p->cout / G);
  }
  p->tiles_m = rn::ceil_div(p->ktotal, kCfgs[p->cfg].bm);
  p->tpg = rn::ceil_div(p->cout / G, kCfgs[p->cfg].bn);
  p->tiles_n = G * p->tpg;
  const int tiles_mn = p->tiles_m * p->tiles_n;
  // aim for ~768 blocks (3 per CU): enough to fill the chip, few enough that the slab traffic
  // (nsplit x |dW| written + read) stays small; each split reduces >= 64 pixels
  long want_splits = (768 + (long)tiles_mn * nbatch - 1) / ((long)tiles_mn * nbatch);
  if (const char* force = getenv("RN_WGRAD_SPLITS")) {  // tuning aid
    if (atoi(force) > 0) want_splits = atoi(force);
  }
  long chunk = (total_pixels + want_splits - 1) / want_splits;
  chunk = (chunk + BK - 1) / BK * BK;
  if (chunk < 2 * BK) chunk = 2 * BK;
  // (a split longer than WG_MAXPIX pixels walks its range in windows: big-kernel layers -- many output tiles already
  // fill the chip -- then need no split at all, and no slab traffic)
  int nsplit = 0;
  for (int s = 0; s < nseg; ++s) {
    p->chunk[s] = (int)chunk;
    p->start[s] = nsplit;
    nsplit += rn::ceil_div(p->pixels[s], (int)chunk);
  }
  p->nsplit = nsplit;
  return RN_OK;
}
}  // namespace

extern "C" size_t rn_conv2d_wgrad_workspace(const rn_conv_seg* segs, int nseg, const rn_conv_geom* g) {
  if (validate_geom(segs, nseg, g)) return 0;
  WgradPlan p;
  if (plan_wgrad(segs, nseg, g, &p)) return 0;
  size_t need = (size_t)p.nsplit * p.ktotal * p.cout * sizeof(float);
  if (nseg == 1 && g->kh == 1 && g->kw == 1 && g->stride == 1 && ngroups(g) == 1 && g->cin % 4 == 0 && segs[0].cout % 4 == 0) {
    // (either product mode may run: the split-bf16 kernels have their own slab count)
    const size_t x3 = rn::conv1x1_wgrad_workspace_x3(segs[0].n * segs[0].h * segs[0].w, g->cin, segs[0].cout);
    if (x3 > need) need = x3;
  }
  if (gconv_direct(segs, nseg, g, Batch{1, 0, 0, 0}, false)) {
    const size_t gc = rn::gconv3x3_wgrad_workspace(segs[0].n, segs[0].h, segs[0].w, g->cin, ngroups(g));
    if (gc > need) need = gc;
  }
  if (const size_t colb = x3_im2col(segs, nseg, g, Batch{1, 0, 0, 0}, false)) {      // patch matrix + the split product's slabs
    const size_t ic = colb + rn::conv1x1_wgrad_workspace_x3((int)(colb / 4 / ((size_t)g->kh * g->kw * g->cin)), g->kh * g->kw * g->cin, segs[0].cout);
    if (ic > need) need = ic;
  }
  return need;
}

namespace {
int conv_wgrad_impl(const rn_conv_seg* segs, int nseg, const rn_conv_geom* g, float* dw, int accumulate, void* workspace,
                    size_t workspace_bytes, const Batch& bt, rn_stream_t stream, int* nsplit_out = nullptr,
                    Planned* plan = nullptr);
}
extern "C" int rn_conv2d_wgrad(const rn_conv_seg* segs, int nseg, const rn_conv_geom* g, float* dw, int accumulate,
                               void* workspace, size_t workspace_bytes, rn_stream_t stream, rn_reduce_list* defer) {
  rn::DeferScope defer_scope_(defer);
  return conv_wgrad_impl(segs, nseg, g, dw, accumulate, workspace, workspace_bytes, Batch{1, 0, 0, 0}, stream);
}

// C_b [K x N] = A_b^T B_b for A_b [M x K], B_b [M x N], b = 0..nbatch-1 (Winograd weight gradient: the sum over
// tiles of V_xi^T dM_xi).  The reduction over M is split across blocks; the partial products go through
// `workspace` (rn::batched_gemm_tn_workspace bytes) and one fixed-order row reduction.
size_t rn::batched_gemm_tn_workspace(int M, int K, int N, int nbatch) {
  rn_conv_seg sg = {};
  sg.n = 1; sg.h = 1; sg.w = M; sg.cout = N;
  rn_conv_geom g1 = {1, 1, 1, K, 1};
  WgradPlan p;
  if (plan_wgrad(&sg, 1, &g1, &p, nbatch)) return 0;
  const size_t f32 = (size_t)p.nsplit * nbatch * K * N * sizeof(float);
  const size_t x3 = rn::gemm_x3_ok(M, K, N) ? rn::batched_gemm_tn_workspace_x3(M, K, N, nbatch) : 0;    // (either mode may run)
  return f32 > x3 ? f32 : x3;
}
int rn::launch_batched_gemm_tn(const float* A, const float* B, float* C, int M, int K, int N, int nbatch, void* workspace,
                               size_t workspace_bytes, hipStream_t st, int* nsplit_out, int background) {
  if (nsplit_out && !C && rn::product_mode() == 1 && rn::gemm_x3_ok(M, K, N))
    return rn::launch_batched_gemm_tn_x3(A, B, M, K, N, nbatch, workspace, workspace_bytes, st, nsplit_out, background);
  rn_conv_seg sg = {};
  sg.n = 1; sg.h = 1; sg.w = M; sg.cout = N; sg.x = A; sg.dy = B;
  rn_conv_geom g1 = {1, 1, 1, K, 1};
  return conv_wgrad_impl(&sg, 1, &g1, C, 0, workspace, workspace_bytes, Batch{nbatch, (long)M * K, (long)M * N, 0},
                         (rn_stream_t)st, nsplit_out);
}

namespace {
int conv_wgrad_impl(const rn_conv_seg* segs, int nseg, const rn_conv_geom* g, float* dw, int accumulate, void* workspace,
                    size_t workspace_bytes, const Batch& bt, rn_stream_t stream, int* nsplit_out, Planned* plan) {
  if (int e = validate_geom(segs, nseg, g)) return e;
  RN_CHECK_ARG((dw || nsplit_out) && workspace, "conv wgrad: null dw/workspace");
  RN_CHECK_ARG(bt.n == 1 || nseg == 1, "conv: batched mode takes one segment");
  if (!plan && !nsplit_out) {
    if (const size_t colb = x3_im2col(segs, nseg, g, bt, false)) {
      const int K = g->kh * g->kw * g->cin, M = (int)(colb / 4 / (size_t)K);
      const size_t slabs = rn::conv1x1_wgrad_workspace_x3(M, K, segs[0].cout);
      if (workspace_bytes >= colb + slabs) {
        RN_CHECK_ARG(segs[0].x && segs[0].dy && dw, "conv wgrad: null pointer in segment 0");
        float* col = (float*)workspace;
        if (int e = rn::launch_im2col(segs[0].x, col, segs[0].n, segs[0].h, segs[0].w, g->cin, g->kh, g->kw, g->stride, (hipStream_t)stream)) return e;
        int ns = 0;
        void* sl = (char*)workspace + colb;
        if (int e = rn::launch_conv1x1_wgrad_x3(col, K, segs[0].dy, M, K, segs[0].cout, sl, workspace_bytes - colb, (hipStream_t)stream, &ns)) return e;
        return rn::launch_reduce_rows((const float*)sl, dw, (int64_t)K * segs[0].cout, ns, accumulate, (hipStream_t)stream);
      }
    }
  }
  if (!plan && !nsplit_out && gconv_direct(segs, nseg, g, bt, false)) {
    RN_CHECK_ARG(segs[0].x && segs[0].dy && dw, "conv wgrad: null pointer in segment 0");
    return rn::launch_gconv3x3_wgrad(segs[0].x, segs[0].dy, dw, accumulate, segs[0].n, segs[0].h, segs[0].w, g->cin, ngroups(g), workspace,
                                     workspace_bytes, (hipStream_t)stream);
  }
  if (!plan && !nsplit_out && nseg == 1) {
    SegDev v = {};
    set_x_view(v, segs[0], g->cin);
    const long m = (long)segs[0].n * segs[0].h * segs[0].w;
    if (x3_conv1x1(segs, nseg, g, bt, m, v.x_ld, false)) {
      RN_CHECK_ARG(segs[0].x && segs[0].dy, "conv wgrad: null pointer in segment 0");
      int ns = 0;
      if (int e = rn::launch_conv1x1_wgrad_x3(segs[0].x + v.x_coff, v.x_ld, segs[0].dy, (int)m, g->cin, segs[0].cout, workspace, workspace_bytes,
                                              (hipStream_t)stream, &ns))
        return e;
      return rn::launch_reduce_rows((const float*)workspace, dw, (int64_t)g->cin * segs[0].cout, ns, accumulate, (hipStream_t)stream);
    }
  }
  WgradPlan p;
  if (int e = plan_wgrad(segs, nseg, g, &p, bt.n)) return e;
  const size_t need = (size_t)p.nsplit * bt.n * p.ktotal * p.cout * sizeof(float);
  if (workspace_bytes < need) {
    rn::set_error("conv wgrad: workspace %zu < %zu bytes", workspace_bytes, need);
    return RN_EWORKSPACE;
  }
  ConvArgs a = {};
  a.nseg = nseg; a.kh = g->kh; a.kw = g->kw; a.stride = g->stride; a.cin = g->cin;
  a.groups = ngroups(g); a.cin_g = g->cin / a.groups; a.tpg = p.tpg;
  a.ktotal = p.ktotal; a.cout = p.cout; a.tiles_n = p.tiles_n; a.tiles_mn = p.tiles_m * p.tiles_n;
  a.slab = (float*)workspace;
  const bool direct = p.nsplit == 1 && bt.n == 1 && !accumulate && !nsplit_out;  // one split: straight into dw
  if (direct) a.slab = dw;
  a.nbatch = bt.n; a.btiles = p.nsplit; a.bs_a = bt.bs_a; a.bs_b = bt.bs_b;
  for (int s = 0; s < nseg; ++s) {
    RN_CHECK_ARG(segs[s].x && segs[s].dy, "conv wgrad: null pointer in segment %d", s);
    SegDev& d = a.seg[s];
    d.a = segs[s].x; d.b = segs[s].dy; d.n = segs[s].n; d.h = segs[s].h; d.w = segs[s].w; d.cout = p.cout;
    d.oh = p.oh[s]; d.ow = p.ow[s]; d.pad_t = p.pt[s]; d.pad_l = p.pl[s];
    d.m = p.pixels[s]; d.start = p.start[s]; d.chunk = p.chunk[s];
    set_x_view(d, segs[s], g->cin);
  }
  const bool vec = (a.cin_g % 4 == 0) && ((p.cout / a.groups) % 4 == 0);
  const int blocks = p.nsplit * bt.n * a.tiles_mn;
  hipStream_t st = (hipStream_t)stream;
  if (plan) {
    plan->a = a; plan->cfg = p.cfg; plan->blocks = blocks; plan->vec = vec; plan->tapu = false; plan->simple = bt.n == 1;
    plan->nsplit = p.nsplit; plan->direct = direct; plan->count = (int64_t)p.ktotal * p.cout;
    return RN_OK;
  }
#define RN_WG(BM_, BN_, WM_, WN_)                                                                    \
  do {                                                                                               \
    if (vec) hipLaunchKernelGGL((conv_wgrad_kernel<BM_, BN_, WM_, WN_, 4>), dim3(blocks), dim3(WM_* WN_ * 64), 0, st, a); \
    else hipLaunchKernelGGL((conv_wgrad_kernel<BM_, BN_, WM_, WN_, 1>), dim3(blocks), dim3(WM_* WN_ * 64), 0, st, a);     \
  } while (0)
  switch (p.cfg) {
    case 0: RN_WG(128, 128, 2, 2); break;
    case 1: RN_WG(128, 64, 2, 2); break;
    case 2: RN_WG(64, 64, 2, 2); break;
    default: RN_WG(128, 32, 4, 1); break;
  }
#undef RN_WG
  RN_LAUNCH_CHECK();
  if (nsplit_out) {  // the caller sums the [nsplit][batch][ktotal][cout] partial products itself
    *nsplit_out = p.nsplit;
    return RN_OK;
  }
  if (direct) return RN_OK;
  return rn::launch_reduce_rows((const float*)workspace, dw, (int64_t)bt.n * p.ktotal * p.cout, p.nsplit, accumulate, st);
}
}  // namespace

namespace {
ConvArgs4 compact(const ConvArgs& a) {
  ConvArgs4 c = {};
  for (int s = 0; s < a.nseg && s < 4; ++s) c.seg[s] = a.seg[s];
  c.nseg = a.nseg; c.kh = a.kh; c.kw = a.kw; c.stride = a.stride; c.cin = a.cin; c.groups = a.groups; c.cin_g = a.cin_g;
  c.tpg = a.tpg; c.ktotal = a.ktotal; c.tiles_mn = a.tiles_mn; c.tiles_n = a.tiles_n; c.cout = a.cout; c.slab = a.slab;
  c.nbatch = a.nbatch; c.btiles = a.btiles; c.bs_a = a.bs_a; c.bs_b = a.bs_b; c.bs_out = a.bs_out; c.ksplit = a.ksplit;
  return c;
}
}  // namespace

// dx (segments' dx) and dw of one convolution.  Small problems (<= 4 segments, the 64x64 dgrad tile, no split-K / phase
// decomposition) run as ONE launch of conv_bwd_kernel; everything else as rn_conv2d_dgrad followed by rn_conv2d_wgrad.
// workspace: rn_conv2d_wgrad_workspace bytes.
extern "C" int rn_conv2d_bwd(const rn_conv_seg* segs, int nseg, const rn_conv_geom* g, float* dw, void* workspace,
                             size_t workspace_bytes, rn_stream_t stream, rn_reduce_list* defer) {
  rn::DeferScope defer_scope_(defer);
  static const bool enabled = getenv("RN_NO_MERGED_BWD") == nullptr;
  Planned pd, pw;
  bool merge = enabled && nseg <= 4;
  if (merge && nseg == 1 && validate_geom(segs, nseg, g) == 0) {      // dense 1x1 in product mode 1: two split-bf16 launches instead
    SegDev v = {};
    set_x_view(v, segs[0], g->cin);
    if (x3_conv1x1(segs, nseg, g, Batch{1, 0, 0, 0}, (long)segs[0].n * segs[0].h * segs[0].w, v.x_ld, false)) merge = false;
    if (gconv_direct(segs, nseg, g, Batch{1, 0, 0, 0}, false)) merge = false;          // ... grouped 3x3: the direct kernels
    if (x3_im2col(segs, nseg, g, Batch{1, 0, 0, 0}, false)) merge = false;             // ... dense k x k / stride 2: patch matrix + split products
  }
  if (merge) {
    if (int e = conv_dgrad_impl(segs, nseg, g, Batch{1, 0, 0, 0}, stream, Scratch{nullptr, 0, nullptr}, &pd)) return e;
    if (int e = conv_wgrad_impl(segs, nseg, g, dw, 0, workspace, workspace_bytes, Batch{1, 0, 0, 0}, stream, nullptr, &pw)) return e;
    merge = pd.simple && pd.vec && pw.vec && pd.cfg == 2 && pd.a.nseg <= 4 && pw.a.nseg <= 4;
  }
  if (!merge) {
    // (the patch-matrix path of the data gradient needs the scratch -- the weight gradient's is at least as large; every other
    // shape keeps the scratch-free data gradient it always had)
    const bool ic = nseg == 1 && validate_geom(segs, nseg, g) == 0 && x3_im2col(segs, nseg, g, Batch{1, 0, 0, 0}, false) != 0;
    if (int e = conv_dgrad_impl(segs, nseg, g, Batch{1, 0, 0, 0}, stream, ic ? Scratch{workspace, workspace_bytes, nullptr} : Scratch{nullptr, 0, nullptr}))
      return e;
    return conv_wgrad_impl(segs, nseg, g, dw, 0, workspace, workspace_bytes, Batch{1, 0, 0, 0}, stream);
  }
  const ConvArgs4 d4 = compact(pd.a), w4 = compact(pw.a);
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)(pd.blocks + pw.blocks));
#define RN_BWD(WBM_, WBN_, WWM_, WWN_)                                                                                          \
  do {                                                                                                                          \
    if (pd.tapu) hipLaunchKernelGGL((conv_bwd_kernel<64, 64, 2, 2, true, WBM_, WBN_, WWM_, WWN_>), grid, dim3(256), 0, st, d4, w4, pd.blocks); \
    else hipLaunchKernelGGL((conv_bwd_kernel<64, 64, 2, 2, false, WBM_, WBN_, WWM_, WWN_>), grid, dim3(256), 0, st, d4, w4, pd.blocks);        \
  } while (0)
  switch (pw.cfg) {
    case 0: RN_BWD(128, 128, 2, 2); break;
    case 1: RN_BWD(128, 64, 2, 2); break;
    case 2: RN_BWD(64, 64, 2, 2); break;
    default: RN_BWD(128, 32, 4, 1); break;
  }
#undef RN_BWD
  RN_LAUNCH_CHECK();
  if (pw.direct) return RN_OK;
  return rn::launch_reduce_rows((const float*)workspace, dw, pw.count, pw.nsplit, 0, st);
}

// The two batched products of a Winograd backward pass in ONE launch (same trick as rn_conv2d_bwd):
//   data gradient    Cd_b [M x Nd] = Ad_b [M x Kd] * Bd_b^T          (Bd_b stored [Nd x Kd])
//   weight gradient  partial sums of Aw_b^T [Kw x M] * Bw_b [M x Nw] (split over M; *nsplit_out slabs
//                    [nsplit][nbatch][Kw][Nw] are left in `workspace` for the caller's back-transform to add up)
int rn::launch_winograd_bwd_products(const float* Ad, const float* Bd, float* Cd, int M, int Kd, int Nd, const float* Aw,
                                     const float* Bw, int Kw, int Nw, int nbatch, void* workspace, size_t workspace_bytes,
                                     hipStream_t st, int* nsplit_out) {
  if (rn::product_mode() == 1 && rn::gemm_x3_ok(M, Kd, Nd) && rn::gemm_x3_ok(M, Kw, Nw)) {
    if (int e = rn::launch_batched_gemm_x3(Ad, Bd, Cd, M, Kd, Nd, nbatch, 1, st)) return e;
    return rn::launch_batched_gemm_tn_x3(Aw, Bw, M, Kw, Nw, nbatch, workspace, workspace_bytes, st, nsplit_out);
  }
  rn_conv_seg sd = {}, sw = {};
  sd.n = 1; sd.h = 1; sd.w = M; sd.wgt = Bd; sd.dy = Ad; sd.dx = Cd; sd.cout = Kd;
  rn_conv_geom gd = {1, 1, 1, Nd, 1};
  const Batch bd = {nbatch, (long)M * Kd, (long)Kd * Nd, (long)M * Nd};
  sw.n = 1; sw.h = 1; sw.w = M; sw.cout = Nw; sw.x = Aw; sw.dy = Bw;
  rn_conv_geom gw = {1, 1, 1, Kw, 1};
  const Batch bw = {nbatch, (long)M * Kw, (long)M * Nw, 0};
  Planned pd, pw;
  static const bool enabled = getenv("RN_NO_MERGED_BWD") == nullptr;
  bool merge = enabled;
  if (merge) {
    if (int e = conv_dgrad_impl(&sd, 1, &gd, bd, (rn_stream_t)st, Scratch{nullptr, 0, nullptr}, &pd)) return e;
    if (int e = conv_wgrad_impl(&sw, 1, &gw, nullptr, 0, workspace, workspace_bytes, bw, (rn_stream_t)st, nsplit_out, &pw)) return e;
    merge = pd.vec && pw.vec && pd.cfg == 2 && !pd.a.seg[0].par && pd.a.ksplit == 0;
  }
  if (!merge) {
    if (int e = conv_dgrad_impl(&sd, 1, &gd, bd, (rn_stream_t)st)) return e;
    return conv_wgrad_impl(&sw, 1, &gw, nullptr, 0, workspace, workspace_bytes, bw, (rn_stream_t)st, nsplit_out);
  }
  *nsplit_out = pw.nsplit;
  const ConvArgs4 d4 = compact(pd.a), w4 = compact(pw.a);
  // RN_DEBUG_SKIP_WGRAD_BLOCKS=1 (TIMING ONLY, wrong weight gradients): the weight-gradient blocks are not launched -- what the
  // heads' backward pass would take with its weight gradients computed elsewhere
  static const bool skip_w = getenv("RN_DEBUG_SKIP_WGRAD_BLOCKS") && atoi(getenv("RN_DEBUG_SKIP_WGRAD_BLOCKS")) == 1;
  const dim3 grid((unsigned)(pd.blocks + (skip_w ? 0 : pw.blocks)));
  // tuning aid: unused dynamic LDS caps the products' blocks per CU, leaving registers / wave slots for the OTHER subnet's
  // transform kernel that runs beside them (RN_PROD_LDS_PAD_BWD bytes)
  static const int pad = getenv("RN_PROD_LDS_PAD_BWD") ? atoi(getenv("RN_PROD_LDS_PAD_BWD")) : 0;
#define RN_BWD(WBM_, WBN_, WWM_, WWN_)                                                                                          \
  do {                                                                                                                          \
    if (pd.tapu) hipLaunchKernelGGL((conv_bwd_kernel<64, 64, 2, 2, true, WBM_, WBN_, WWM_, WWN_>), grid, dim3(256), pad, st, d4, w4, pd.blocks); \
    else hipLaunchKernelGGL((conv_bwd_kernel<64, 64, 2, 2, false, WBM_, WBN_, WWM_, WWN_>), grid, dim3(256), pad, st, d4, w4, pd.blocks);        \
  } while (0)
  switch (pw.cfg) {
    case 0: RN_BWD(128, 128, 2, 2); break;
    case 1: RN_BWD(128, 64, 2, 2); break;
    case 2: RN_BWD(64, 64, 2, 2); break;
    default: RN_BWD(128, 32, 4, 1); break;
  }
#undef RN_BWD
  RN_LAUNCH_CHECK();
  return RN_OK;
}

// ---------------------------------------------------------------------------------------------
// bias gradient: column sums of dy over every pixel of every segment (two fixed-order stages)
// ---------------------------------------------------------------------------------------------
namespace {
constexpr int BG_BLOCKS = 256;  // row groups: ~43 rows each for the 10912-row head gradient, and a 256-row reduce after
struct BiasArgs {
  const float* dy[RN_MAX_SEG];
  int rows[RN_MAX_SEG];
  int nseg, cout;
  float* partial;  // [BG_BLOCKS][cout]
};
__global__ __launch_bounds__(256) void bias_partial_kernel(const BiasArgs a) {
  // grid (row groups, 256-column groups): thread owns one column, rows strided by the row groups, four loads in flight
  const int c = blockIdx.y * 256 + threadIdx.x;
  if (c >= a.cout) return;
  const int g = gridDim.x;
  float acc = 0.f;
  for (int s = 0; s < a.nseg; ++s) {
    const float* p = a.dy[s] + c;
    const int rows = a.rows[s];
    int r = blockIdx.x;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (; r + 3 * g < rows; r += 4 * g) {
      a0 += p[(size_t)r * a.cout];
      a1 += p[(size_t)(r + g) * a.cout];
      a2 += p[(size_t)(r + 2 * g) * a.cout];
      a3 += p[(size_t)(r + 3 * g) * a.cout];
    }
    for (; r < rows; r += g) a0 += p[(size_t)r * a.cout];
    acc += (a0 + a1) + (a2 + a3);
  }
  a.partial[(size_t)blockIdx.x * a.cout + c] = acc;
}
}  // namespace

extern "C" size_t rn_conv2d_bias_grad_workspace(int cout) { return (size_t)BG_BLOCKS * (size_t)(cout > 0 ? cout : 0) * sizeof(float); }

extern "C" int rn_conv2d_bias_grad(const rn_conv_seg* segs, int nseg, const rn_conv_geom* g, float* dbias, void* workspace,
                                   size_t workspace_bytes, rn_stream_t stream, rn_reduce_list* defer) {
  rn::DeferScope defer_scope_(defer);
  if (int e = validate_geom(segs, nseg, g)) return e;
  RN_CHECK_ARG(dbias && workspace, "bias grad: null pointer");
  BiasArgs a = {};
  a.nseg = nseg; a.cout = segs[0].cout; a.partial = (float*)workspace;
  for (int s = 0; s < nseg; ++s) {
    RN_CHECK_ARG(segs[s].dy && segs[s].cout == a.cout, "bias grad: bad segment %d", s);
    int oh, ow, pt, pl;
    rn::same_pad(segs[s].h, g->kh, g->stride, &oh, &pt);
    rn::same_pad(segs[s].w, g->kw, g->stride, &ow, &pl);
    a.dy[s] = segs[s].dy; a.rows[s] = segs[s].n * oh * ow;
  }
  if (workspace_bytes < rn_conv2d_bias_grad_workspace(a.cout)) { rn::set_error("bias grad: workspace too small"); return RN_EWORKSPACE; }
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(bias_partial_kernel, dim3(BG_BLOCKS, (unsigned)rn::ceil_div(a.cout, 256)), dim3(256), 0, st, a);
  RN_LAUNCH_CHECK();
  return rn::launch_reduce_rows((const float*)workspace, dbias, a.cout, BG_BLOCKS, 0, st);
}
