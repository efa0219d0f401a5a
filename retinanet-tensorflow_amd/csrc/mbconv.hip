// MobileNetV2 bottleneck chain (mobilenet_v2.py:41-94, GroupNorm variant normalization.py:20-35) with every GroupNorm
// applied by its CONSUMER (rn_hip.h, "MobileNetV2 bottleneck chain"):
//
//   forward   y1 = A W1            A  = x, or drop(GN3'(y3')) + x' of the previous bottleneck, normalised WHILE LOADING the
//                                  A tile (and written out once: the bottleneck's materialised input / a pyramid tap)
//             y2 = dw3x3(a1)       a1 = drop(act(GN1(y1))) formed once per input-patch element on its way into LDS
//             y3 = a2 W3           a2 = drop(act(GN2(y2))) normalised while loading the A tile
//   every kernel emits the per-group (sum, sum sq) rows of its raw output from its epilogue; the consumer merges the rows
//   of its sample (fixed order, fp64) before its first operand store.
//   backward  the data-gradient epilogues form g = d act'(z) mask of the GroupNorm block they enter, with its rows
//             (sum gamma g, sum gamma g xhat); the next kernel computes dy = rstd (gamma g - c1 - xhat c2) while loading.
//
// Pointwise convs run on the fp32 matrix cores (v_mfma_f32_32x32x2_f32, conv_tiles.h); the depthwise kernels stage their
// input patch through LDS (one transform per element instead of one per tap).  HBM-bound layers: what this file removes is
// the write + re-read of every normalised tensor and 7 of 13 kernel boundaries per bottleneck.
#include "mb_common.h"

namespace {

// =====================================================================================================================
// pointwise forward:  y[M, N] = A[M, K] W[K, N],  A plain or normalised while loading
// =====================================================================================================================

// KS > 1: the K-tiles are dealt round-robin to KS groups of TG = WM WN 64 threads (few output tiles, long K: the small
// maps' linear convs), each with its own operand tiles; the groups' accumulators are summed at the end.  The smallest maps
// take 32 x 32 tiles with one wave per group: more blocks -- the operand transform (ELU, dropout mask: VALU work) and the
// fp32 MFMAs of a 64 x 64 x 960 tile would keep ONE compute unit busy for 13 us while 230 others idle.
// NST register stages: the operand tiles of NST K-steps are in flight at once (a dependent round trip to memory costs
// ~3 us here -- L2 is per XCD, data written by the previous kernel sits in another one -- an MFMA step 0.4 us).
template <int BM, int BN, int WM, int WN, bool NORM, int ACT, int KS, int NST>
__global__ __launch_bounds__(WM* WN * 64 * KS) void mb_pw_fwd_kernel(const PwFwdArgs a) {
  constexpr int TG = WM * WN * 64;
  constexpr bool RES = NORM && ACT != RN_ACT_ELU && ACT != RN_ACT_RELU6 && ACT != RN_ACT_RELU;   // a residual comes with a linear block only
  static_assert(TG * KS >= T || !NORM, "the prologue needs 256 threads");
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int KQ = BK / 4, A_RPP = TG / KQ, A_PASS = BM / A_RPP;
  constexpr int NQ = BN / 4, B_RPP = TG / NQ, B_PASS = BK / B_RPP;
  static_assert(A_PASS >= 1 && B_PASS >= 1 && BM % A_RPP == 0 && BK % B_RPP == 0, "tile/threads mismatch");
  constexpr int OPF = BM * LDK + BK * BN;
  static_assert(KS * OPF * 4 >= (T + GMAX) * 16 && OPF >= (WM + 1) * BN * 2, "operand tiles double as scratch");
  static_assert(KS == 1 || KS * OPF >= (KS - 1) * TM * TN * 16 * TG, "operand tiles double as the split-K exchange");
  __shared__ __attribute__((aligned(16))) float smem[KS * OPF];
  __shared__ __attribute__((aligned(16))) float tab[NORM ? 2 * KMAX : 4];
  __shared__ float gstat[NORM ? GMAX : 1][2];
  const int tid = threadIdx.x, grp = tid / TG, lt = tid % TG, lane = lt & 63, wave = lt >> 6;
  float* As = smem + grp * OPF;
  float* Bs = As + BM * LDK;
  const int wm = wave / WN, wn = wave % WN;
  const int bid = rn::xcd_remap(blockIdx.x, gridDim.x);
  const int tile_n = bid % a.tiles_n, tile_m = bid / a.tiles_n;
  const int K = a.cin, N = a.cout, M = a.n * a.hw;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int sample = m0 / a.hw;                       // a tile never straddles samples (hw % BM == 0, host-checked)
  const float* asrc = NORM ? a.in.y : a.x;
  const __amdgpu_buffer_rsrc_t xa = make_rsrc(asrc, (unsigned)M * K * 4u);
  const __amdgpu_buffer_rsrc_t xr = make_rsrc(a.res ? a.res : asrc, (unsigned)M * K * 4u);
  const __amdgpu_buffer_rsrc_t wb = make_rsrc(a.w, (unsigned)K * N * 4u);
  const bool has_res = RES && a.res != nullptr;

  const int kq = lt % KQ, arow = lt / KQ;
  const int nq = lt % NQ;
  const int bcol = n0 + nq * 4;
  const unsigned boff0 = bcol < N ? ((unsigned)(lt / NQ) * N + bcol) * 4u : OOB;
  const int nk = (K + BK - 1) / BK, nit = (nk + KS - 1) / KS;
  float4 ra[NST][A_PASS], rr[RES ? NST : 1][A_PASS], rb[NST][B_PASS];
  auto load_tiles = [&](int it, const int st) {
    const int kt = it * KS + grp;
    const int k = kt * BK + kq * 4;
    const bool kok = k < K;
#pragma unroll
    for (int i = 0; i < A_PASS; ++i) {
      const unsigned off = kok ? ((unsigned)(m0 + arow + i * A_RPP) * K + k) * 4u : OOB;
      ra[st][i] = Vec<4>::load(xa, off);
      if (RES) rr[st][i] = Vec<4>::load(xr, has_res ? off : OOB);
    }
    const unsigned bo = (kt < nk && boff0 != OOB) ? boff0 + (unsigned)kt * BK * N * 4u : OOB;
#pragma unroll
    for (int j = 0; j < B_PASS; ++j) rb[st][j] = Vec<4>::load(wb, bo == OOB ? OOB : bo + (unsigned)j * B_RPP * N * 4u);
  };
  ChanPre<NORM ? KMAX / T : 1> pre;
  GroupPre gpre = {0.f, 1.f};
  if (NORM) {
    prefetch_chan(a.in.gamma, a.in.beta, 0, K, tid, pre);
    gpre = prefetch_groups(a.in, sample, 0, a.in.groups, tid);
  }
#pragma unroll
  for (int st = 0; st < NST; ++st)
    if (st < nit) load_tiles(st, st);
  uint64_t seed = 0;
  bool drop = false;
  if (NORM) {
    // the sample's statistics while the first tiles are in flight; the first block of a sample publishes them
    group_stats(a.in, sample, a.hw, 0, a.in.groups, tile_n == 0 && m0 == sample * a.hw, smem, gstat, gpre, tid);
    scale_shift_table(a.in, 0, K, 0, gstat, tab, tab + KMAX, pre, tid);
    drop = a.in.drop_rate > 0.f;
    seed = a.in.seed + (a.in.seed_dev ? *a.in.seed_dev : 0ull);
  }
  if (a.dbg == 1) return;
  const bool write_mat = NORM && a.mat != nullptr && tile_n == 0;
  auto store_tiles = [&](int it, const int st) {
    const int k = (it * KS + grp) * BK + kq * 4;
#pragma unroll
    for (int i = 0; i < A_PASS; ++i) {
      float4 v = ra[st][i];
      if (NORM) {
        if (k < K) {
          const int m = m0 + arow + i * A_RPP;
          const float4 sc = *reinterpret_cast<const float4*>(&tab[k]);
          const float4 sh = *reinterpret_cast<const float4*>(&tab[KMAX + k]);
          v = norm_act_drop<ACT>(v, sc, sh, a.in.act, drop, a.in.drop_rate, a.in.keep_scale, seed, (uint64_t)m * K + k);
          if (RES) { v.x += rr[st][i].x; v.y += rr[st][i].y; v.z += rr[st][i].z; v.w += rr[st][i].w; }
          if (write_mat) *reinterpret_cast<float4*>(a.mat + (size_t)m * K + k) = v;
        } else {
          v = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
      *reinterpret_cast<float4*>(&As[(arow + i * A_RPP) * LDK + kq * 4]) = v;
    }
#pragma unroll
    for (int j = 0; j < B_PASS; ++j) *reinterpret_cast<float4*>(&Bs[(lt / NQ + j * B_RPP) * BN + nq * 4]) = rb[st][j];
  };

  f32x16 acc[TM][TN];
  zero_acc<TM, TN>(acc);
  for (int it0 = 0; it0 < nit; it0 += NST) {
#pragma unroll
    for (int st = 0; st < NST; ++st) {
      const int it = it0 + st;
      if (it < nit) {                       // (block-uniform)
        store_tiles(it, st);
        __syncthreads();
        if (it + NST < nit) load_tiles(it + NST, st);
        if (it * KS + grp < nk) mma_ktile<BM, BN, WM, WN, false, false>(As, Bs, acc, wm, wn, lane);
        __syncthreads();
      }
    }
  }
  if (a.dbg == 2) { if (acc[0][0][0] == 123.456f) a.y[0] = 0.f; return; }
  sum_groups<KS, TM * TN, TG>(&acc[0][0], smem, grp, lt);
  if (grp == 0) store_tile<BM, BN, WM, WN>(acc, a.y, nullptr, m0, n0, M, N, N, wm, wn, lane);
  if (a.dbg == 3) return;
  if (a.ost.rows) {
    float2* row = a.ost.rows + ((size_t)sample * a.ost.R + (m0 - sample * a.hw) / BM) * a.ost.W;
    float s1[TN], s2[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      s1[tn] = 0.f; s2[tn] = 0.f;
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float v = acc[tm][tn][r]; s1[tn] += v; s2[tn] = fmaf(v, v, s2[tn]); }
    }
    reduce_group_rows<BM, BN, WM, WN>(s1, s2, smem, row, n0, N, a.ocpg, tile_n, tid, nullptr, nullptr, nullptr);
  }
}

// =====================================================================================================================
// depthwise 3x3 forward on the normalised input: block = (sample, TH x TW output tile, channel slab of whole groups)
// =====================================================================================================================

template <int ACT, bool PF>
__global__ __launch_bounds__(T) void mb_dw_fwd_kernel(const DwFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float dsm[];   // patch [ph * pw][sw]; before / after: merge and reduction scratch
  __shared__ __attribute__((aligned(16))) float tab[2 * 128];    // scale | shift of the slab's channels
  __shared__ float gstat[GMAX][2];
  const int tid = threadIdx.x;
  const int bid = rn::xcd_remap(blockIdx.x, gridDim.x);
  const int slab = bid % a.nslab;
  const int tt = bid / a.nslab;
  const int ntile = a.tiles_h * a.tiles_w;
  // a block walks `tpb` consecutive tiles of one sample: the prologue (rows merge, tables, weights) and the final reduction
  // are paid once, the raw patch of tile t + 1 is in flight while tile t's stencil runs
  const int blk = tt % a.nblk, sample = tt / a.nblk;
  const int tile_lo = blk * a.tpb, tile_hi = min(tile_lo + a.tpb, ntile);
  const int C = a.c, SW = a.sw, SQ = SW >> 2, c0 = slab * SW;
  const int g0 = c0 / a.in.cpg, ng = SW / a.in.cpg;
  // everything from memory first: the slab's gamma / beta, the first raw patch, the stencil weights
  ChanPre<1> pre;
  prefetch_chan(a.in.gamma, a.in.beta, c0, SW, tid, pre);
  const GroupPre gpre = prefetch_groups(a.in, sample, g0, ng, tid);
  const int total = a.ph * a.pw * SQ;
  // (raw buffer loads: a slot that is not part of the patch, or lies outside the image, costs no memory access)
  const __amdgpu_buffer_rsrc_t xs = make_rsrc(a.in.y + (size_t)sample * a.h * a.wd * C + c0, (unsigned)(a.h * a.wd * C - c0) * 4u);
  float4 pv[NP];
  int pk[NP];
#pragma unroll
  for (int j = 0; j < NP; ++j) pk[j] = patch_pack(tid + j * T, total, SQ, a.pw);
  auto load_patch = [&](int tile) {
    const int ih0 = (tile / a.tiles_w) * a.th * a.stride - a.pad_t, iw0 = (tile % a.tiles_w) * a.tw * a.stride - a.pad_l;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const PatchElem e = patch_at(pk[j], ih0, iw0, a.h, a.wd);
      pv[j] = Vec<4>::load(xs, (e.live && e.inside) ? ((unsigned)e.pix * C + e.q * 4) * 4u : OOB);
    }
  };
  load_patch(tile_lo);
  const int lanes = T / SQ, q4 = tid % SQ, pl = tid / SQ;
  float4 wv[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) wv[t] = *reinterpret_cast<const float4*>(a.w + (size_t)t * C + c0 + min(q4, SQ - 1) * 4);
  group_stats(a.in, sample, a.h * a.wd, g0, ng, blk == 0, dsm, gstat, gpre, tid);
  scale_shift_table(a.in, c0, SW, g0, gstat, tab, tab + 128, pre, tid);
  if (a.dbg == 1) { if (pv[0].x == 123.456f && wv[0].x == 1.f) a.y[0] = 0.f; return; }
  const bool drop = a.in.drop_rate > 0.f;
  const uint64_t seed = a.in.seed + (a.in.seed_dev ? *a.in.seed_dev : 0ull);
  const uint64_t samp_off = (uint64_t)sample * a.h * a.wd * C;
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  float* __restrict__ ys = a.y + (size_t)sample * a.oh * a.ow * C + c0 + q4 * 4;
  auto process_tile = [&](int tile) {
    const int oh0 = (tile / a.tiles_w) * a.th, ow0 = (tile % a.tiles_w) * a.tw;
    const int ih0 = oh0 * a.stride - a.pad_t, iw0 = ow0 * a.stride - a.pad_l;
    // patch -> LDS, normalised once per element; zero outside the image (SAME padding pads the ACTIVATED tensor)
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const PatchElem e = patch_at(pk[j], ih0, iw0, a.h, a.wd);
      if (e.live) {
        float4 o = norm_act_drop<ACT>(pv[j], *reinterpret_cast<const float4*>(&tab[e.q * 4]), *reinterpret_cast<const float4*>(&tab[128 + e.q * 4]),
                                      a.in.act, drop, a.in.drop_rate, a.in.keep_scale, seed, samp_off + (uint64_t)e.pix * C + c0 + e.q * 4);
        if (!e.inside) o = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(&dsm[(size_t)(tid + j * T) * 4]) = o;     // slot = pp * SQ + q: [pp][sw] rows
      }
    }
    __syncthreads();
    if (PF && tile + 1 < tile_hi) load_patch(tile + 1);
    // stencil from LDS: thread = (channel quad, pixel lane)
    if (pl < lanes) {
      for (int p = pl; p < a.th * a.tw; p += lanes) {
        const int oy = p / a.tw, ox = p - oy * a.tw;
        const int oh_ = oh0 + oy, ow_ = ow0 + ox;
        if (oh_ < a.oh && ow_ < a.ow) {
          float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
          for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
              const float4 xv = *reinterpret_cast<const float4*>(&dsm[((size_t)(oy * a.stride + kh) * a.pw + ox * a.stride + kw) * SW + q4 * 4]);
              const float4 w4 = wv[kh * 3 + kw];
              acc.x = fmaf(xv.x, w4.x, acc.x); acc.y = fmaf(xv.y, w4.y, acc.y);
              acc.z = fmaf(xv.z, w4.z, acc.z); acc.w = fmaf(xv.w, w4.w, acc.w);
            }
          *reinterpret_cast<float4*>(ys + (size_t)(oh_ * a.ow + ow_) * C) = acc;
          s1[0] += acc.x; s1[1] += acc.y; s1[2] += acc.z; s1[3] += acc.w;
          s2[0] = fmaf(acc.x, acc.x, s2[0]); s2[1] = fmaf(acc.y, acc.y, s2[1]);
          s2[2] = fmaf(acc.z, acc.z, s2[2]); s2[3] = fmaf(acc.w, acc.w, s2[3]);
        }
      }
    }
    __syncthreads();                               // the patch is dead: the next tile's, or the reduction scratch
  };
  if (PF) {
#pragma nounroll
    for (int tile = tile_lo; tile < tile_hi; ++tile) process_tile(tile);
  } else {                                         // (nothing carried around the loop)
    process_tile(tile_lo);
#pragma nounroll
    for (int tile = tile_lo + 1; tile < tile_hi; ++tile) {
      load_patch(tile);
      process_tile(tile);
    }
  }
  if (!a.ost.rows || a.dbg == 3) return;
  float (*red)[8] = reinterpret_cast<float (*)[8]>(dsm);
  float (*chan)[2] = reinterpret_cast<float (*)[2]>(dsm + T * 8);
#pragma unroll
  for (int j = 0; j < 4; ++j) { red[tid][j] = s1[j]; red[tid][4 + j] = s2[j]; }
  __syncthreads();
  for (int e = tid; e < SQ * 8; e += T) {          // pixel lanes in order
    const int qd = e >> 3, comp = e & 7;
    float t = 0.f;
    for (int l = 0; l < lanes; ++l) t += red[l * SQ + qd][comp];
    chan[qd * 4 + (comp & 3)][comp >> 2] = t;
  }
  __syncthreads();
  const int ong = SW / a.ocpg, og0 = c0 / a.ocpg;
  if (tid < ong) {                                 // channels of a group in order
    float t1 = 0.f, t2 = 0.f;
    for (int j = 0; j < a.ocpg; ++j) { t1 += chan[tid * a.ocpg + j][0]; t2 += chan[tid * a.ocpg + j][1]; }
    a.ost.rows[((size_t)sample * a.ost.R + blk) * a.ost.W + og0 + tid] = make_float2(t1, t2);
  }
}

// =====================================================================================================================
// out = drop(act(GN(y))) [+ residual]: the end of a chain.  grid (pixel chunks, samples); thread = (channel quad, pixel lane)
// =====================================================================================================================
struct ApplyArgs { NormDev in; const float* res; float* out; int n, hw, ppb; };

__global__ __launch_bounds__(T) void mb_apply_kernel(const ApplyArgs a) {
  __shared__ __attribute__((aligned(16))) double scratch[(T + GMAX) * 2];
  __shared__ __attribute__((aligned(16))) float tab[2 * KMAX];
  __shared__ float gstat[GMAX][2];
  const int tid = threadIdx.x, sample = blockIdx.y, C = a.in.c, CQ = C >> 2;
  ChanPre<KMAX / T> pre;
  prefetch_chan(a.in.gamma, a.in.beta, 0, C, tid, pre);
  const GroupPre gpre = prefetch_groups(a.in, sample, 0, a.in.groups, tid);
  group_stats(a.in, sample, a.hw, 0, a.in.groups, blockIdx.x == 0, scratch, gstat, gpre, tid);
  scale_shift_table(a.in, 0, C, 0, gstat, tab, tab + KMAX, pre, tid);
  const bool drop = a.in.drop_rate > 0.f;
  const uint64_t seed = a.in.seed + (a.in.seed_dev ? *a.in.seed_dev : 0ull);
  const int p_begin = blockIdx.x * a.ppb, p_end = min(p_begin + a.ppb, a.hw);
  const size_t base = (size_t)sample * a.hw * C;
  for (int i = tid; i < (p_end - p_begin) * CQ; i += T) {
    const int p = p_begin + i / CQ, q = i % CQ;
    const size_t e = base + (size_t)p * C + q * 4;
    const float4 v = *reinterpret_cast<const float4*>(a.in.y + e);
    float4 o = norm_act_drop<-1>(v, *reinterpret_cast<const float4*>(&tab[q * 4]), *reinterpret_cast<const float4*>(&tab[KMAX + q * 4]),
                                 a.in.act, drop, a.in.drop_rate, a.in.keep_scale, seed, (uint64_t)e);
    if (a.res) {
      const float4 r = *reinterpret_cast<const float4*>(a.res + e);
      o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
    }
    *reinterpret_cast<float4*>(a.out + e) = o;
  }
}

// =====================================================================================================================
// Row compaction: the 256 x 256 maps produce more rows per sample (one per tile) than a consumer block wants to merge;
// F consecutive rows are summed into one (fp64, fixed order).  grid (out rows, samples), thread = (entry, row lane).
// =====================================================================================================================
struct CompactArgs { const float2* in; float2* out; int R, W, F, Rout; };
__global__ __launch_bounds__(T) void mb_compact_rows_kernel(const CompactArgs a) {
  __shared__ double part[T][2];
  const int tid = threadIdx.x, ro = blockIdx.x, sample = blockIdx.y;
  const int RL = T / a.W, e = tid % a.W, rl = tid / a.W;
  const int r_begin = ro * a.F, r_end = min(r_begin + a.F, a.R);
  double S = 0.0, Q = 0.0;
  if (rl < RL) {
    const float2* __restrict__ base = a.in + (size_t)sample * a.R * a.W + e;
    for (int r0 = r_begin + rl; r0 < r_end; r0 += 8 * RL) {
      float2 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = base[(size_t)min(r0 + j * RL, r_end - 1) * a.W];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (r0 + j * RL < r_end) { S += (double)v[j].x; Q += (double)v[j].y; }
    }
  }
  part[tid][0] = S; part[tid][1] = Q;
  __syncthreads();
  if (tid < a.W) {
    S = 0.0; Q = 0.0;
    for (int l = 0; l < RL; ++l) { S += part[l * a.W + tid][0]; Q += part[l * a.W + tid][1]; }
    a.out[((size_t)sample * a.Rout + ro) * a.W + tid] = make_float2((float)S, (float)Q);
  }
}

// =====================================================================================================================
// backward
// =====================================================================================================================
struct DyDev {            // the gradient of a raw conv output y as a kernel loads it (rn_mb_dy)
  const float* dy;        // plain, or nullptr:
  const float* g; int g_plain; NormDev nd; RowsDev grows;   // dy = rstd (gamma g' - c1 - xhat c2), g' = g (* dropout mask when g_plain)
};
struct GoutDev {          // rn_mb_gout
  float* out; const float* add1; const float* add2;
  int has_norm; NormDev nd; int store_plain; RowsDev grows; float* planes; long plane_stride;   // planes: [2][plane rows][c]
};

// per-channel coefficients of dy = P g' + Q + R y for the channels [c0, c0 + nc) of `sample`: the GroupNorm's statistics
// (read back: `gp`) and the merged rows c1 = sum(gamma g) / m, c2 = sum(gamma g xhat) / m.  tab: P | Q | R, `stride` apart.
template <int NJ>
__device__ __forceinline__ void dy_table(const DyDev& d, int sample, int hw, int c0, int nc, void* scratch, float (*gstat)[2], float (*gc)[2],
                                         float* tab, int stride, const GroupPre& gp, const ChanPre<NJ>& p, int tid) {
  const int g0 = c0 / d.nd.cpg, ng = (c0 + nc - 1) / d.nd.cpg - g0 + 1;
  double (*part)[2] = reinterpret_cast<double (*)[2]>(scratch);
  double (*tot)[2] = part + T;
  merge_rows(d.grows, sample, d.nd.cpg, d.nd.c, g0, ng, part, tot, tid);
  if (tid < ng) {
    const double m = (double)hw * (double)d.nd.cpg;
    gc[tid][0] = (float)(tot[tid][0] / m); gc[tid][1] = (float)(tot[tid][1] / m);
    gstat[tid][0] = gp.mean; gstat[tid][1] = gp.rstd;
  }
  __syncthreads();
  if (tid < T) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int i = tid + j * T;
      if (i < nc) {
        const int g = (c0 + i) / d.nd.cpg - g0;
        const float mean = gstat[g][0], rstd = gstat[g][1], c1 = gc[g][0], c2 = gc[g][1];
        tab[i] = rstd * p.g[j];
        tab[stride + i] = rstd * (rstd * c2 * mean - c1);
        tab[2 * stride + i] = -rstd * rstd * c2;
      }
    }
  }
  __syncthreads();
}

__device__ __forceinline__ float4 dy_of(float4 g, float4 y, const float* tab, int stride, int i, bool mask, float rate, float keep, uint64_t seed,
                                        uint64_t eidx) {
  const float4 P = *reinterpret_cast<const float4*>(&tab[i]);
  const float4 Q = *reinterpret_cast<const float4*>(&tab[stride + i]);
  const float4 R = *reinterpret_cast<const float4*>(&tab[2 * stride + i]);
  if (mask) {
    float m[4];
    keep4(seed, eidx, rate, keep, m);
    g.x *= m[0]; g.y *= m[1]; g.z *= m[2]; g.w *= m[3];
  }
  return make_float4(fmaf(P.x, g.x, fmaf(R.x, y.x, Q.x)), fmaf(P.y, g.y, fmaf(R.y, y.y, Q.y)), fmaf(P.z, g.z, fmaf(R.z, y.z, Q.z)),
                     fmaf(P.w, g.w, fmaf(R.w, y.w, Q.w)));
}

struct PwBwdArgs {
  int dbg;
  const float* x; NormDev in; int has_in;     // A operand of the weight gradient: x, or the block of `in`
  DyDev dy; const float* w; GoutDev go;
  int n, hw, cin, cout;
  int d_tiles_n, dblocks;                     // data gradient: N-tiles over cin; blocks
  int w_tiles_m, w_tiles_n, chunk, sps;       // weight gradient: tiles over (cin, cout), pixels per split, splits per sample
  float* slab;                                // [n * sps][cin][cout]
  int wfirst;                                 // the weight-gradient blocks take the LOWEST block indices (dispatched first)
};
// both halves use PB x PB tiles: 64 (2 x 2 waves per group) or, on the smallest maps, 32 (one wave per group; more blocks)
constexpr int pw_lds(int pb) { return 2 * pb * LDK; }   // operand floats per group (dgrad: two k-contiguous tiles; wgrad needs 2 * BK * pb, less)

// data gradient  d[M, cin] = dy[M, cout] W^T  and what rn_mb_gout asks for
template <int ACT_OUT, int PB, int KS, int NST>
__device__ __forceinline__ void mb_pw_dgrad_body(const PwBwdArgs& a, float* smem_all, float* tabD, float (*gstat)[2], float (*gc)[2], int blk) {
  constexpr int BM = PB, BN = PB, WM = PB / 32, WN = PB / 32, TM = 1, TN = 1, TG = WM * WN * 64;
  constexpr int KQ = BK / 4, RPP = TG / KQ, A_PASS = BM / RPP, B_PASS = BN / RPP;
  const int tid = threadIdx.x, grp = tid / TG, lt = tid % TG, lane = lt & 63, wave = lt >> 6;
  float* As = smem_all + grp * pw_lds(PB);
  float* Bs = As + BM * LDK;
  const int wm = wave / WN, wn = wave % WN;
  const int bid = rn::xcd_remap(blk, a.dblocks);
  const int tile_n = bid % a.d_tiles_n, tile_m = bid / a.d_tiles_n;
  const int KD = a.cout, ND = a.cin, M = a.n * a.hw;      // reduction over cout, output columns = cin
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int sample = m0 / a.hw;
  const bool plain = a.dy.dy != nullptr;
  const __amdgpu_buffer_rsrc_t ga = make_rsrc(plain ? a.dy.dy : a.dy.g, (unsigned)M * KD * 4u);
  const __amdgpu_buffer_rsrc_t ya = make_rsrc(plain ? a.dy.dy : a.dy.nd.y, (unsigned)M * KD * 4u);
  const __amdgpu_buffer_rsrc_t wb = make_rsrc(a.w, (unsigned)ND * KD * 4u);
  const int kq = lt % KQ, r0 = lt / KQ;
  unsigned browoff[B_PASS];
#pragma unroll
  for (int j = 0; j < B_PASS; ++j) {
    const int ci = n0 + r0 + j * RPP;
    browoff[j] = ci < ND ? (unsigned)ci * KD * 4u : OOB;
  }
  const int nk = (KD + BK - 1) / BK, nit = (nk + KS - 1) / KS;
  float4 ra[NST][A_PASS], ry[NST][A_PASS], rb[NST][B_PASS];
  auto load_tiles = [&](int it, const int st) {
    const int k = (it * KS + grp) * BK + kq * 4;
    const bool kok = k < KD;
#pragma unroll
    for (int i = 0; i < A_PASS; ++i) {
      const unsigned off = kok ? ((unsigned)(m0 + r0 + i * RPP) * KD + k) * 4u : OOB;
      ra[st][i] = Vec<4>::load(ga, off);
      ry[st][i] = Vec<4>::load(ya, plain ? OOB : off);
    }
#pragma unroll
    for (int j = 0; j < B_PASS; ++j) rb[st][j] = Vec<4>::load(wb, (kok && browoff[j] != OOB) ? browoff[j] + (unsigned)k * 4u : OOB);
  };
  // everything from memory first: the dy table's gamma and statistics, the epilogue's per-column constants, the first tiles
  ChanPre<KMAX / T> pre;
  GroupPre gpre = {0.f, 1.f};
  if (!plain) {
    prefetch_chan<KMAX / T>(a.dy.nd.gamma, nullptr, 0, KD, tid, pre);
    gpre = prefetch_groups(a.dy.nd, sample, 0, a.dy.nd.groups, tid);
  }
  const GoutDev& go = a.go;
  const int l31 = lane & 31, half = lane >> 5;
  const int col = n0 + wn * 32 + l31;
  const bool cok = col < ND;
  float o_mean = 0.f, o_rstd = 1.f, o_gam = 1.f, o_bet = 0.f;
  if (go.has_norm && grp == 0) {
    const int cc = cok ? col : 0, og = cc / go.nd.cpg;
    o_mean = go.nd.mean[sample * go.nd.groups + og]; o_rstd = go.nd.rstd[sample * go.nd.groups + og];
    o_gam = go.nd.gamma[cc]; o_bet = go.nd.beta[cc];
  }
#pragma unroll
  for (int st = 0; st < NST; ++st)
    if (st < nit) load_tiles(st, st);
  bool mask = false;
  uint64_t seed = 0;
  if (!plain) {
    dy_table(a.dy, sample, a.hw, 0, KD, smem_all, gstat, gc, tabD, KMAX, gpre, pre, tid);
    mask = a.dy.g_plain && a.dy.nd.drop_rate > 0.f;
    seed = a.dy.nd.seed + (a.dy.nd.seed_dev ? *a.dy.nd.seed_dev : 0ull);
  }
  if (a.dbg == 1) return;
  auto store_tiles = [&](int it, const int st) {
    const int k = (it * KS + grp) * BK + kq * 4;
#pragma unroll
    for (int i = 0; i < A_PASS; ++i) {
      float4 v = ra[st][i];
      if (!plain) {
        if (k < KD) v = dy_of(v, ry[st][i], tabD, KMAX, k, mask, a.dy.nd.drop_rate, a.dy.nd.keep_scale, seed,
                              (uint64_t)(m0 + r0 + i * RPP) * KD + k);
        else v = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      *reinterpret_cast<float4*>(&As[(r0 + i * RPP) * LDK + kq * 4]) = v;
    }
#pragma unroll
    for (int j = 0; j < B_PASS; ++j) *reinterpret_cast<float4*>(&Bs[(r0 + j * RPP) * LDK + kq * 4]) = rb[st][j];
  };
  f32x16 acc[TM][TN];
  zero_acc<TM, TN>(acc);
  for (int it0 = 0; it0 < nit; it0 += NST) {
#pragma unroll
    for (int st = 0; st < NST; ++st) {
      const int it = it0 + st;
      if (it < nit) {
        store_tiles(it, st);
        __syncthreads();
        if (it + NST < nit) load_tiles(it + NST, st);
        if (it * KS + grp < nk) mma_ktile<BM, BN, WM, WN, false, true>(As, Bs, acc, wm, wn, lane);
        __syncthreads();
      }
    }
  }
  if (a.dbg == 2) { if (acc[0][0][0] == 123.456f) a.go.out[0] = 0.f; return; }
  sum_groups<KS, 1, TG>(&acc[0][0], smem_all, grp, lt);
  // ---- epilogue (group 0): d (+ addends) -> out; with a GroupNorm block behind the conv's input: g, its rows and planes
  const int rbase = m0 + wm * 32 + 4 * half;
  const __amdgpu_buffer_rsrc_t ro = make_rsrc(go.out, (unsigned)M * ND * 4u);
  const bool on = grp == 0;
  float d[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) d[r] = acc[0][0][r];
  float t1[16], t2[16], yv[16];
  {
    const __amdgpu_buffer_rsrc_t r1 = make_rsrc(go.add1 ? go.add1 : go.out, (unsigned)M * ND * 4u);
    const __amdgpu_buffer_rsrc_t r2 = make_rsrc(go.add2 ? go.add2 : go.out, (unsigned)M * ND * 4u);
    const __amdgpu_buffer_rsrc_t ry2 = make_rsrc(go.has_norm ? go.nd.y : go.out, (unsigned)M * ND * 4u);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const unsigned off = (cok && on) ? ((unsigned)(rbase + (r & 3) + 8 * (r >> 2)) * ND + col) * 4u : OOB;
      t1[r] = Vec<1>::load(r1, go.add1 ? off : OOB);
      t2[r] = Vec<1>::load(r2, go.add2 ? off : OOB);
      yv[r] = Vec<1>::load(ry2, go.has_norm ? off : OOB);
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) d[r] += t1[r] + t2[r];
  if (!go.has_norm) {
    if (on) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(d[r]), ro, cok ? ((unsigned)(rbase + (r & 3) + 8 * (r >> 2)) * ND + col) * 4u : OOB, 0, 0);
    }
    return;
  }
  const NormDev& nd = go.nd;
  const bool drop = nd.drop_rate > 0.f;
  const uint64_t oseed = nd.seed + (nd.seed_dev ? *nd.seed_dev : 0ull);
  float s1[1] = {0.f}, s2[1] = {0.f};
  if (on) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = rbase + (r & 3) + 8 * (r >> 2);
      const float xh = (yv[r] - o_mean) * o_rstd;
      const float z = fmaf(xh, o_gam, o_bet);
      float g = d[r] * actgrad_of<ACT_OUT>(z, nd.act);
      if (drop) g *= keep1(oseed, (uint64_t)row * ND + col, nd.drop_rate, nd.keep_scale);
      if (!cok) g = 0.f;
      s1[0] += g; s2[0] = fmaf(g, xh, s2[0]);
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(go.store_plain ? d[r] : g), ro, cok ? ((unsigned)row * ND + col) * 4u : OOB, 0, 0);
    }
  }
  const int prow = sample * go.grows.R + (m0 - sample * a.hw) / BM;
  reduce_group_rows<BM, BN, WM, WN>(s1, s2, smem_all, go.grows.rows + (size_t)prow * go.grows.W, n0, ND, nd.cpg, tile_n, tid, nd.gamma,
                                    go.planes + (size_t)prow * ND, go.planes + go.plane_stride + (size_t)prow * ND);
}

// weight gradient  dW[cin, cout] = A^T dy over one split's pixels (inside one sample) -> slab[split]
template <int ACT_IN, int PB, int KS, int NST>
__device__ __forceinline__ void mb_pw_wgrad_body(const PwBwdArgs& a, float* smem_all, float* tabA, float* tabD, float (*gstat)[2], float (*gc)[2],
                                                 int blk, int nblk) {
  constexpr int BM = PB, BN = PB, WM = PB / 32, WN = PB / 32, TM = 1, TN = 1, TG = WM * WN * 64;
  constexpr int MQ = BM / 4, A_RPP = TG / MQ, A_PASS = BK / A_RPP;
  constexpr int NQ = BN / 4, B_RPP = TG / NQ, B_PASS = BK / B_RPP;
  const int tid = threadIdx.x, grp = tid / TG, lt = tid % TG, lane = lt & 63, wave = lt >> 6;
  float* As = smem_all + grp * pw_lds(PB);   // [BK][BM]  (pixel rows, cin contiguous)
  float* Bs = As + BK * BM;              // [BK][BN]
  const int wm = wave / WN, wn = wave % WN;
  const int bid = rn::xcd_remap(blk, nblk);
  const int tiles_mn = a.w_tiles_m * a.w_tiles_n;
  const int split = bid / tiles_mn, t = bid - split * tiles_mn;
  const int tile_n = t % a.w_tiles_n, tile_m = t / a.w_tiles_n;
  const int sample = split / a.sps;
  const int p0 = sample * a.hw + (split - sample * a.sps) * a.chunk;   // chunk divides hw (host)
  const int KI = a.cin, NO = a.cout, M = a.n * a.hw;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const bool plain = a.dy.dy != nullptr;
  const float* asrc = a.has_in ? a.in.y : a.x;
  const __amdgpu_buffer_rsrc_t xa = make_rsrc(asrc, (unsigned)M * KI * 4u);
  const __amdgpu_buffer_rsrc_t ga = make_rsrc(plain ? a.dy.dy : a.dy.g, (unsigned)M * NO * 4u);
  const __amdgpu_buffer_rsrc_t ya = make_rsrc(plain ? a.dy.dy : a.dy.nd.y, (unsigned)M * NO * 4u);
  const int mq = lt % MQ, acol = m0 + mq * 4;
  const bool aok = acol < KI;
  const int nq = lt % NQ, bcol = n0 + nq * 4;
  const bool bok = bcol < NO;
  const int nk = a.chunk / BK, nit = (nk + KS - 1) / KS;
  float4 ra[NST][A_PASS], rg[NST][B_PASS], ry[NST][B_PASS];
  auto load_tiles = [&](int it, const int st) {
    const int kt = it * KS + grp;
    const bool tok = kt < nk;
#pragma unroll
    for (int j = 0; j < A_PASS; ++j) {
      const int p = p0 + kt * BK + lt / MQ + j * A_RPP;
      ra[st][j] = Vec<4>::load(xa, (aok && tok) ? ((unsigned)p * KI + acol) * 4u : OOB);
    }
#pragma unroll
    for (int j = 0; j < B_PASS; ++j) {
      const int p = p0 + kt * BK + lt / NQ + j * B_RPP;
      const unsigned off = (bok && tok) ? ((unsigned)p * NO + bcol) * 4u : OOB;
      rg[st][j] = Vec<4>::load(ga, off);
      ry[st][j] = Vec<4>::load(ya, plain ? OOB : off);
    }
  };
  const int na = min(BM, KI - m0), nb = min(BN, NO - n0);
  ChanPre<1> pre_d, pre_a;
  GroupPre gp_d = {0.f, 1.f}, gp_a = {0.f, 1.f};
  int ga0 = 0, nga = 1;
  if (!plain) {
    prefetch_chan<1>(a.dy.nd.gamma, nullptr, n0, nb, tid, pre_d);
    const int g0 = n0 / a.dy.nd.cpg, ng = (n0 + nb - 1) / a.dy.nd.cpg - g0 + 1;
    gp_d = prefetch_groups(a.dy.nd, sample, g0, ng, tid);
  }
  if (a.has_in) {
    prefetch_chan<1>(a.in.gamma, a.in.beta, m0, na, tid, pre_a);
    ga0 = m0 / a.in.cpg; nga = (m0 + na - 1) / a.in.cpg - ga0 + 1;
    gp_a = prefetch_groups(a.in, sample, ga0, nga, tid);
  }
#pragma unroll
  for (int st = 0; st < NST; ++st)
    if (st < nit) load_tiles(st, st);
  bool mask = false, drop_in = false;
  uint64_t seed = 0, seed_in = 0;
  if (!plain) {
    dy_table(a.dy, sample, a.hw, n0, nb, smem_all, gstat, gc, tabD, BN, gp_d, pre_d, tid);
    mask = a.dy.g_plain && a.dy.nd.drop_rate > 0.f;
    seed = a.dy.nd.seed + (a.dy.nd.seed_dev ? *a.dy.nd.seed_dev : 0ull);
  }
  if (a.has_in) {
    group_stats(a.in, sample, a.hw, ga0, nga, false, smem_all, gstat, gp_a, tid);
    scale_shift_table(a.in, m0, na, ga0, gstat, tabA, tabA + BM, pre_a, tid);
    drop_in = a.in.drop_rate > 0.f;
    seed_in = a.in.seed + (a.in.seed_dev ? *a.in.seed_dev : 0ull);
  }
  if (a.dbg == 1) return;
  auto store_tiles = [&](int it, const int st) {
    const int kt = it * KS + grp;
#pragma unroll
    for (int j = 0; j < A_PASS; ++j) {
      float4 v = ra[st][j];
      if (a.has_in) {
        if (aok && kt < nk) {
          const int p = p0 + kt * BK + lt / MQ + j * A_RPP;
          v = norm_act_drop<ACT_IN>(v, *reinterpret_cast<const float4*>(&tabA[mq * 4]), *reinterpret_cast<const float4*>(&tabA[BM + mq * 4]),
                                    a.in.act, drop_in, a.in.drop_rate, a.in.keep_scale, seed_in, (uint64_t)p * KI + acol);
        } else {
          v = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
      *reinterpret_cast<float4*>(&As[(lt / MQ + j * A_RPP) * BM + mq * 4]) = v;
    }
#pragma unroll
    for (int j = 0; j < B_PASS; ++j) {
      float4 v = rg[st][j];
      if (!plain) {
        if (bok && kt < nk) {
          const int p = p0 + kt * BK + lt / NQ + j * B_RPP;
          v = dy_of(v, ry[st][j], tabD, BN, nq * 4, mask, a.dy.nd.drop_rate, a.dy.nd.keep_scale, seed, (uint64_t)p * NO + bcol);
        } else {
          v = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
      *reinterpret_cast<float4*>(&Bs[(lt / NQ + j * B_RPP) * BN + nq * 4]) = v;
    }
  };
  f32x16 acc[TM][TN];
  zero_acc<TM, TN>(acc);
  for (int it0 = 0; it0 < nit; it0 += NST) {
#pragma unroll
    for (int st = 0; st < NST; ++st) {
      const int it = it0 + st;
      if (it < nit) {
        store_tiles(it, st);
        __syncthreads();
        if (it + NST < nit) load_tiles(it + NST, st);
        if (it * KS + grp < nk) mma_ktile<BM, BN, WM, WN, true, false>(As, Bs, acc, wm, wn, lane);
        __syncthreads();
      }
    }
  }
  if (a.dbg == 2) { if (acc[0][0][0] == 123.456f) a.slab[0] = 0.f; return; }
  sum_groups<KS, 1, TG>(&acc[0][0], smem_all, grp, lt);
  if (grp == 0) store_tile<BM, BN, WM, WN>(acc, a.slab + (size_t)split * KI * NO, nullptr, m0, n0, KI, NO, NO, wm, wn, lane);
}

// ACT_IN: activation of the weight gradient's A block (-2: plain x); ACT_OUT: activation of the block the data gradient enters.
// The two halves have their own tile size and split-K group count (same number of threads): on the smallest maps the data
// gradient takes 32 x 32 tiles (more blocks: its reduction is long), the weight gradient keeps 64 x 64 (its output is large).
template <int ACT_IN, int ACT_OUT, int DPB, int DKS, int WPB, int WKS>
__global__ __launch_bounds__((DPB / 32) * (DPB / 32) * 64 * DKS) void mb_pw_bwd_kernel(const PwBwdArgs a) {
  constexpr int DTG = (DPB / 32) * (DPB / 32) * 64, WTG = (WPB / 32) * (WPB / 32) * 64;
  static_assert(DTG * DKS == WTG * WKS && DTG * DKS >= T, "both halves run in the same block size; the prologues need 256 threads");
  constexpr int LDSF = DKS * pw_lds(DPB) > WKS * pw_lds(WPB) ? DKS * pw_lds(DPB) : WKS * pw_lds(WPB);
  static_assert(LDSF * 4 >= (T + GMAX) * 16 && pw_lds(DPB) >= 3 * DPB * 2 && pw_lds(WPB) >= 2 * BK * WPB, "operand tiles double as scratch");
  static_assert((DKS == 1 || DKS * pw_lds(DPB) >= (DKS - 1) * 16 * DTG) && (WKS == 1 || WKS * pw_lds(WPB) >= (WKS - 1) * 16 * WTG),
                "operand tiles double as the split-K exchange");
  __shared__ __attribute__((aligned(16))) float smem[LDSF];
  __shared__ __attribute__((aligned(16))) float tabD[3 * KMAX];
  __shared__ __attribute__((aligned(16))) float tabA[2 * 64];
  __shared__ float gstat[GMAX][2];
  __shared__ float gc[GMAX][2];
  constexpr int NST = 1;      // operand tiles in flight per thread (measured: 4 stages cost occupancy on the big maps and buy nothing on the small)
  // big maps: a weight-gradient block walks 8 - 32 K-tiles of pixels, a data-gradient block 1 - 5 of channels -- with more
  // blocks than the chip holds at once the long ones go first (block indices are dispatched in order), or they form the tail
  const int nw = (int)gridDim.x - a.dblocks;
  const int b = (int)blockIdx.x;
  const bool is_w = a.wfirst ? b < nw : b >= a.dblocks;
  if (!is_w) mb_pw_dgrad_body<ACT_OUT, DPB, DKS, NST>(a, smem, tabD, gstat, gc, a.wfirst ? b - nw : b);
  else mb_pw_wgrad_body<(ACT_IN == -2 ? 0 : ACT_IN), WPB, WKS, NST>(a, smem, tabA, tabD, gstat, gc, a.wfirst ? b : b - a.dblocks, nw);
}

// ---------------------------------------------------------------------------------------------------------------------
// Pointwise backward of the LARGE maps (256^2 / 128^2: narrow convs, 16 .. 144 channels), both gradients from ONE pass over
// the pixels.  The tile kernels above read g and y twice (data-gradient blocks + weight-gradient blocks) and pay a prologue per
// 64-row tile; here a block owns a run of pixels of one sample and each of its four WAVES walks 16-row chunks on its own:
//   chunk rows are contiguous in memory -> every operand arrives as coalesced float4 rows (g, y -> dy formed on the way into
//   the wave's LDS tile; the conv's input likewise), the next chunk's loads are in flight while this one is multiplied;
//   d[16, cin]  = dy[16, cout] W^T              v_mfma_f32_16x16x4_f32, K = cout, W resident in LDS for the whole block
//   dW[cin,cout] += A^T[cin, 16] dy[16, cout]   same tiles, K = the chunk's 16 pixels, accumulators live across the chunks
//   epilogue as the tile kernel's (g = d act'(z) mask, per-channel sums), rows / planes ONCE per block, dW partial per block.
// Y_IS_A (the linear conv: the GroupNorm block the data gradient enters IS the conv's input block): the wave's input tile holds
// the RAW tensor -- the weight gradient's operand is normalised as it is read, the epilogue reads the raw value and leaves its
// result in place.  Otherwise (the expand conv) the tile holds the operand, and the narrow tensors of the epilogue (the output
// block's raw y, the addends) come as scalar loads in the accumulator layout, in flight with the chunk.
// No barrier inside the loop (a wave reads only the LDS tiles it wrote).  HBM-bound by construction: 2 cout + cin (.. 4 cin)
// floats read and cin written per pixel; the MFMA work is ~1/10 of that time.
typedef float f32x4 __attribute__((ext_vector_type(4)));
struct PwBigArgs {
  int dbg;                                    // tuning aid (RN_MB_DBG=pwB:<phase>): stop after a phase
  const float* x; NormDev in; int has_in;
  DyDev dy; const float* w; GoutDev go;
  int n, hw, cin, cout;
  int ppb, bps;                               // pixels per block (multiple of 16), blocks per sample
  float* slab;                                // [n * bps][cin][cout]
};
constexpr int pw_big_lds_floats(int nti, int nto) {      // W [CI][CO + 4] | per wave: Dt [16][CO + 4], At [16][CI + 4]
  return nti * 16 * (nto * 16 + 4) + 4 * 16 * ((nto * 16 + 4) + (nti * 16 + 4));
}

template <int NTI, int NTO, bool Y_IS_A>
__global__ __launch_bounds__(T, 2) void mb_pw_bwd_big_kernel(const PwBigArgs a) {
  constexpr int CI = NTI * 16, CO = NTO * 16, SA = CI + 4, SD = CO + 4;
  constexpr int NG = (CO * 4 + 63) / 64, NA = (CI * 4 + 63) / 64;     // float4 loads per lane and chunk: dy side, input side
  constexpr int NE = Y_IS_A ? 1 : 4 * NTI;                            // scalar epilogue loads per lane and tensor
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ __attribute__((aligned(16))) float tabD[3 * CO];
  __shared__ __attribute__((aligned(16))) float tabA[2 * CI];
  __shared__ __attribute__((aligned(16))) float tabO[4 * CI];         // mean | rstd | gamma | beta per output channel
  __shared__ float gstat[GMAX][2];
  __shared__ float gc[GMAX][2];
  __shared__ __attribute__((aligned(16))) double mscratch[(T + GMAX) * 2];   // the row merges' own scratch: W goes to its tile at once
  static_assert(pw_big_lds_floats(NTI, NTO) >= 2 * NTI * NTO * 256 && pw_big_lds_floats(NTI, NTO) >= 10 * CI, "the operand tiles double as the final exchange");
  constexpr bool PF = NTI * NTO < 18;            // the next chunk's rows in flight under the products (the widest tiles: registers)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int blk = blockIdx.x, sample = blk / a.bps, bis = blk - sample * a.bps;
  const int KI = a.cin, NO = a.cout, M = a.n * a.hw;
  const int p_begin = sample * a.hw + bis * a.ppb;
  const bool plain = a.dy.dy != nullptr;
  const GoutDev& go = a.go;
  float* Wl = lds;
  float* Dt = lds + CI * SD + wave * 16 * (SD + SA);
  float* At = Dt + 16 * SD;
  // ---- the chunk loads: float4 slot lane + 64 j of a chunk's 16 * C contiguous floats is (row, col) = divmod(4 (lane + 64 j), C);
  // slot 0's pair is kept, the others follow by adding divmod(256, C) (no per-slot registers, no division in the loop)
  const int g_r0 = (lane * 4) / NO, g_c0 = lane * 4 - g_r0 * NO, g_dr = 256 / NO, g_dc = 256 - g_dr * NO;
  const int a_r0 = (lane * 4) / KI, a_c0 = lane * 4 - a_r0 * KI, a_dr = 256 / KI, a_dc = 256 - a_dr * KI;
  auto g_slot = [&](int j, int& row, int& col) {      // (j is a compile-time constant at every call: the loop below folds)
    row = g_r0; col = g_c0;
    for (int t = 0; t < j; ++t) { row += g_dr; col += g_dc; if (col >= NO) { col -= NO; ++row; } }
    return (lane + 64 * j) * 4 < 16 * NO;
  };
  auto a_slot = [&](int j, int& row, int& col) {
    row = a_r0; col = a_c0;
    for (int t = 0; t < j; ++t) { row += a_dr; col += a_dc; if (col >= KI) { col -= KI; ++row; } }
    return (lane + 64 * j) * 4 < 16 * KI;
  };
  const float* gsrc = plain ? a.dy.dy : a.dy.g;
  const float* ysrc = plain ? a.dy.dy : a.dy.nd.y;
  const float* asrc = a.has_in ? a.in.y : a.x;
  const bool y_sep = !Y_IS_A && go.has_norm;
  const __amdgpu_buffer_rsrc_t rg = make_rsrc(gsrc, (unsigned)M * NO * 4u);
  const __amdgpu_buffer_rsrc_t ry = make_rsrc(ysrc, (unsigned)M * NO * 4u);
  const __amdgpu_buffer_rsrc_t ra = make_rsrc(asrc, (unsigned)M * KI * 4u);
  const __amdgpu_buffer_rsrc_t ryo = make_rsrc(y_sep ? go.nd.y : asrc, (unsigned)M * KI * 4u);
  const __amdgpu_buffer_rsrc_t r1 = make_rsrc(go.add1 ? go.add1 : asrc, (unsigned)M * KI * 4u);
  const __amdgpu_buffer_rsrc_t r2 = make_rsrc(go.add2 ? go.add2 : asrc, (unsigned)M * KI * 4u);
  const int l15 = lane & 15, lq = lane >> 4;
  float4 vg[NG], vy[NG], va[NA];
  float eyo[NE], e1[NE], e2[NE];                  // accumulator layout: [i * 4 + r] = (row 4 lq + r, col 16 i + l15)
#pragma unroll
  for (int t = 0; t < NE; ++t) { eyo[t] = 0.f; e1[t] = 0.f; e2[t] = 0.f; }
  auto load_chunk = [&](int p) {                  // p: first pixel of the chunk
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      const unsigned off = (lane + 64 * j) * 4 < 16 * NO ? ((unsigned)p * NO + (unsigned)(lane + 64 * j) * 4u) * 4u : OOB;
      vg[j] = Vec<4>::load(rg, off);
      vy[j] = Vec<4>::load(ry, plain ? OOB : off);
    }
#pragma unroll
    for (int j = 0; j < NA; ++j)
      va[j] = Vec<4>::load(ra, (lane + 64 * j) * 4 < 16 * KI ? ((unsigned)p * KI + (unsigned)(lane + 64 * j) * 4u) * 4u : OOB);
    if (!Y_IS_A) {
#pragma unroll
      for (int i = 0; i < NTI; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int col = i * 16 + l15;
          const unsigned off = col < KI ? ((unsigned)(p + 4 * lq + r) * KI + col) * 4u : OOB;
          eyo[(i * 4 + r) % NE] = Vec<1>::load(ryo, y_sep ? off : OOB);
          e1[(i * 4 + r) % NE] = Vec<1>::load(r1, go.add1 ? off : OOB);
          e2[(i * 4 + r) % NE] = Vec<1>::load(r2, go.add2 ? off : OOB);
        }
    }
  };
  const int nchunk = a.ppb >> 4;
  // ---- prologue (all 256 threads).  Everything from memory first: W (-> its LDS tile, zero padding included), the tables' inputs
  constexpr int WQ = (CI * (CO >> 2) + T - 1) / T;
  {
    float4 wv[WQ];
#pragma unroll
    for (int j = 0; j < WQ; ++j) {
      const int e = tid + j * T, ci = e / (CO >> 2), q = e - ci * (CO >> 2);
      const bool ok = ci < KI && q * 4 < NO;
      wv[j] = ok ? *reinterpret_cast<const float4*>(a.w + (size_t)ci * NO + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int j = 0; j < WQ; ++j) {
      const int e = tid + j * T, ci = e / (CO >> 2), q = e - ci * (CO >> 2);
      if (ci < CI) *reinterpret_cast<float4*>(&Wl[ci * SD + q * 4]) = wv[j];
    }
    // zero padding of this thread's share of the waves' tiles: dy columns [NO, CO), input columns [KI, CI)
    for (int e = tid; e < 4 * 16 * (CO - NO + CI - KI); e += T) {
      const int per = CO - NO + CI - KI, wv_ = e / (16 * per), r_ = (e / per) % 16, c_ = e % per;
      float* dt = lds + CI * SD + wv_ * 16 * (SD + SA);
      if (c_ < CO - NO) dt[r_ * SD + NO + c_] = 0.f;
      else dt[16 * SD + r_ * SA + KI + (c_ - (CO - NO))] = 0.f;
    }
  }
  ChanPre<1> pre_d, pre_a;
  GroupPre gp_d = {0.f, 1.f}, gp_a = {0.f, 1.f};
  if (!plain) {
    prefetch_chan<1>(a.dy.nd.gamma, nullptr, 0, NO, tid, pre_d);
    gp_d = prefetch_groups(a.dy.nd, sample, 0, a.dy.nd.groups, tid);
  }
  if (a.has_in) {
    prefetch_chan<1>(a.in.gamma, a.in.beta, 0, KI, tid, pre_a);
    gp_a = prefetch_groups(a.in, sample, 0, a.in.groups, tid);
  }
  if (go.has_norm) {
    for (int c = tid; c < CI; c += T) {
      const int cc = min(c, KI - 1), og = cc / go.nd.cpg;
      tabO[c] = go.nd.mean[sample * go.nd.groups + og];
      tabO[CI + c] = go.nd.rstd[sample * go.nd.groups + og];
      tabO[2 * CI + c] = go.nd.gamma[cc];
      tabO[3 * CI + c] = go.nd.beta[cc];
    }
  }
  bool mask = false, drop_in = false;
  uint64_t seed = 0, seed_in = 0;
  if (!plain) {
    dy_table(a.dy, sample, a.hw, 0, NO, mscratch, gstat, gc, tabD, CO, gp_d, pre_d, tid);
    mask = a.dy.g_plain && a.dy.nd.drop_rate > 0.f;
    seed = a.dy.nd.seed + (a.dy.nd.seed_dev ? *a.dy.nd.seed_dev : 0ull);
  }
  if (a.has_in) {
    group_stats(a.in, sample, a.hw, 0, a.in.groups, false, mscratch, gstat, gp_a, tid);
    scale_shift_table(a.in, 0, KI, 0, gstat, tabA, tabA + CI, pre_a, tid);
    drop_in = a.in.drop_rate > 0.f;
    seed_in = a.in.seed + (a.in.seed_dev ? *a.in.seed_dev : 0ull);
  }
  __syncthreads();
  f32x4 accw[NTI][NTO];
#pragma unroll
  for (int i = 0; i < NTI; ++i)
#pragma unroll
    for (int o = 0; o < NTO; ++o) accw[i][o] = f32x4{0.f, 0.f, 0.f, 0.f};
  float s1[NTI], s2[NTI];
#pragma unroll
  for (int i = 0; i < NTI; ++i) { s1[i] = 0.f; s2[i] = 0.f; }
  const NormDev& ond = go.nd;
  const bool odrop = go.has_norm && ond.drop_rate > 0.f;
  const uint64_t oseed = go.has_norm ? ond.seed + (ond.seed_dev ? *ond.seed_dev : 0ull) : 0ull;
  if (a.dbg == 1) return;
  if (PF && wave < nchunk) load_chunk(p_begin + wave * 16);
  for (int ch = wave; ch < nchunk; ch += 4) {
    const int p = p_begin + ch * 16;
    if (!PF) load_chunk(p);
    // ---- operands -> this wave's LDS tiles
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      int row, col;
      if (g_slot(j, row, col)) {
        float4 v = vg[j];
        if (!plain) v = dy_of(v, vy[j], tabD, CO, col, mask, a.dy.nd.drop_rate, a.dy.nd.keep_scale, seed, (uint64_t)(p + row) * NO + col);
        *reinterpret_cast<float4*>(&Dt[row * SD + col]) = v;
      }
    }
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      int row, col;
      if (a_slot(j, row, col)) {
        float4 v = va[j];
        if (!Y_IS_A && a.has_in)
          v = norm_act_drop<-1>(v, *reinterpret_cast<const float4*>(&tabA[col]), *reinterpret_cast<const float4*>(&tabA[CI + col]),
                                a.in.act, drop_in, a.in.drop_rate, a.in.keep_scale, seed_in, (uint64_t)(p + row) * KI + col);
        *reinterpret_cast<float4*>(&At[row * SA + col]) = v;
      }
    }
    float yo[NE], ad[NE];
#pragma unroll
    for (int t = 0; t < NE; ++t) { yo[t] = eyo[t]; ad[t] = e1[t] + e2[t]; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (PF && ch + 4 < nchunk) load_chunk(p + 64); // the next chunk's rows are in flight under the products
    if (a.dbg == 2) continue;
    // ---- d = dy W^T  (A: dy[row = l15][k], B: W[ci = l15][k], four k per step: k = 4 ks + lq)
    f32x4 accd[NTI];
#pragma unroll
    for (int i = 0; i < NTI; ++i) accd[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    // (k is summed in the order that lets a lane fetch FOUR of its k values with one b128 read: in step 4 s + t lane group lq
    // supplies k = 16 s + 4 lq + t, of dy and of W alike; the padding columns up to CO are zeros)
#pragma unroll 2
    for (int ks = 0; ks < NTO; ++ks) {
      const float4 a4 = *reinterpret_cast<const float4*>(&Dt[l15 * SD + 16 * ks + 4 * lq]);
      const float aa[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
      for (int i = 0; i < NTI; ++i) {
        const float4 b4 = *reinterpret_cast<const float4*>(&Wl[(i * 16 + l15) * SD + 16 * ks + 4 * lq]);
        const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
        for (int t = 0; t < 4; ++t) accd[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(aa[t], bb[t], accd[i], 0, 0, 0);
      }
    }
    // ---- dW += A^T dy over the chunk's 16 pixels.  k (the pixel) is summed in the order px = 4 lq + t: in step t lane group lq
    // supplies pixel 4 lq + t of A and of dy -- the (row, col) set a lane feeds is then exactly the set it OWNS in the
    // accumulator layout of d, so with Y_IS_A one pass over the element does both jobs: z, exp and the dropout mask once ->
    // the operand drop(act(z)) for the MFMA, g = d act'(z) mask for the epilogue, left in place of the raw value.
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int row = 4 * lq + t;
      float bv[NTO];
#pragma unroll
      for (int o = 0; o < NTO; ++o) bv[o] = Dt[row * SD + o * 16 + l15];
#pragma unroll
      for (int i = 0; i < NTI; ++i) {
        const int col = i * 16 + l15;
        const bool cok = col < KI;
        float av = At[row * SA + col];
        if (Y_IS_A) {
          const float raw = av;
          const float z = fmaf(raw, tabA[col], tabA[CI + col]);
          const float m = drop_in ? keep1(seed_in, (uint64_t)(p + row) * KI + col, a.in.drop_rate, a.in.keep_scale) : 1.f;
          float fa, fg;                              // act(z), act'(z)
          if (a.in.act == RN_ACT_ELU) { const float ez = __expf(fminf(z, 0.f)); fa = z > 0.f ? z : ez - 1.f; fg = z > 0.f ? 1.f : ez; }
          else { fa = act_of<-1>(z, a.in.act); fg = actgrad_of<-1>(z, a.in.act); }
          av = cok ? fa * m : 0.f;
          const float xh = (raw - tabO[col]) * tabO[CI + col];
          const float d = accd[i][t];
          float g = cok ? d * fg * m : 0.f;
          s1[i] += g; s2[i] = fmaf(g, xh, s2[i]);
          At[row * SA + col] = cok ? (go.store_plain ? d : g) : 0.f;
        }
#pragma unroll
        for (int o = 0; o < NTO; ++o) accw[i][o] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[o], accw[i][o], 0, 0, 0);
      }
    }
    // ---- epilogue in the accumulator layout: lane holds d[row = 4 lq + r][col = 16 i + l15]; result -> At -> coalesced rows
    if (!Y_IS_A) {
#pragma unroll
      for (int i = 0; i < NTI; ++i) {
        const int col = i * 16 + l15;
        const bool cok = col < KI;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 4 * lq + r;
          const float d = accd[i][r] + ad[(i * 4 + r) % NE];
          float outv = d;
          if (go.has_norm) {
            const float xh = (yo[(i * 4 + r) % NE] - tabO[col]) * tabO[CI + col];
            const float z = fmaf(xh, tabO[2 * CI + col], tabO[3 * CI + col]);
            float g = d * actgrad_of<-1>(z, ond.act);
            if (odrop) g *= keep1(oseed, (uint64_t)(p + row) * KI + col, ond.drop_rate, ond.keep_scale);
            if (!cok) g = 0.f;
            s1[i] += g; s2[i] = fmaf(g, xh, s2[i]);
            outv = go.store_plain ? d : g;
          }
          if (!cok) outv = 0.f;
          At[row * SA + col] = outv;                 // (this lane alone reads / writes the element in this phase)
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      int row, col;
      if (a_slot(j, row, col))
        *reinterpret_cast<float4*>(go.out + (size_t)p * KI + (size_t)(lane + 64 * j) * 4) = *reinterpret_cast<const float4*>(&At[row * SA + col]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  if (a.dbg == 3) { if (accw[0][0][0] == 123.456f) a.slab[0] = s1[0]; return; }
  // ---- block totals: the four waves' dW accumulators, (w0 + w2) + (w1 + w3), through LDS; then the rows / planes
  __syncthreads();
  float* ex = lds;                                // [2][tile][r][lane]
  auto put = [&](int slot) {
#pragma unroll
    for (int i = 0; i < NTI; ++i)
#pragma unroll
      for (int o = 0; o < NTO; ++o)
#pragma unroll
        for (int r = 0; r < 4; ++r) ex[((slot * NTI * NTO + i * NTO + o) * 4 + r) * 64 + lane] = accw[i][o][r];
  };
  auto add = [&](int slot) {
#pragma unroll
    for (int i = 0; i < NTI; ++i)
#pragma unroll
      for (int o = 0; o < NTO; ++o)
#pragma unroll
        for (int r = 0; r < 4; ++r) accw[i][o][r] += ex[((slot * NTI * NTO + i * NTO + o) * 4 + r) * 64 + lane];
  };
  if (wave >= 2) put(wave - 2);
  __syncthreads();
  if (wave < 2) add(wave);
  __syncthreads();
  if (wave < 2) put(wave);
  __syncthreads();
  {
    float* slab = a.slab + (size_t)blk * KI * NO;
    for (int e = tid; e < NTI * NTO * 256; e += T) {     // e = (tile, r, lane): (w0 + w2) + (w1 + w3)
      const int ln = e & 63, r = (e >> 6) & 3, tl = e >> 8;
      const int i = tl / NTO, o = tl - i * NTO;
      const int ci = i * 16 + 4 * (ln >> 4) + r, co = o * 16 + (ln & 15);
      if (ci < KI && co < NO) slab[(size_t)ci * NO + co] = ex[((0 * NTI * NTO + tl) * 4 + r) * 64 + ln] + ex[((1 * NTI * NTO + tl) * 4 + r) * 64 + ln];
    }
  }
  if (!go.has_norm) return;
  __syncthreads();
  float* red = lds;                               // [wave][2][CI]
#pragma unroll
  for (int i = 0; i < NTI; ++i) {
    float t1 = s1[i], t2 = s2[i];
    t1 += __shfl_xor(t1, 16, 64); t1 += __shfl_xor(t1, 32, 64);
    t2 += __shfl_xor(t2, 16, 64); t2 += __shfl_xor(t2, 32, 64);
    if (lane < 16) { red[(wave * 2 + 0) * CI + i * 16 + lane] = t1; red[(wave * 2 + 1) * CI + i * 16 + lane] = t2; }
  }
  __syncthreads();
  float* chan = lds + 8 * CI;                     // [CI][2]
  const int prow = sample * a.bps + bis;
  for (int c = tid; c < KI; c += T) {
    float t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) { t1 += red[(w * 2 + 0) * CI + c]; t2 += red[(w * 2 + 1) * CI + c]; }
    chan[c * 2 + 0] = t1; chan[c * 2 + 1] = t2;
    go.planes[(size_t)prow * KI + c] = t1;
    go.planes[go.plane_stride + (size_t)prow * KI + c] = t2;
  }
  __syncthreads();
  if (tid < ond.groups) {
    float t1 = 0.f, t2 = 0.f;
    for (int c = tid * ond.cpg; c < (tid + 1) * ond.cpg; ++c) {
      const float w = ond.gamma[c];
      t1 += w * chan[c * 2 + 0]; t2 += w * chan[c * 2 + 1];
    }
    go.grows.rows[(size_t)prow * go.grows.W + tid] = make_float2(t1, t2);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// depthwise 3x3 backward: block = (sample, TH x TW INPUT tile, channel slab).  LDS: the dy patch (dy = rstd (gamma g - c1
// - xhat c2) formed once per element) and the a1 patch (drop(act(GN(y1))), formed once per element).  Data gradient of the
// tile's pixels -> g1 = d act'(z1) mask with its rows / planes; weight-gradient partial sums over the outputs whose window
// origin lies in the tile (a disjoint cover of the outputs) -> partial[block row][9][c].
struct DwBwdArgs {
  int dbg;
  NormDev in; DyDev dy; const float* w; float* partial; GoutDev go;
  int n, h, wd, c, stride, oh, ow, pad_t, pad_l;
  int th, tw, tiles_h, tiles_w, sw, nslab;
  int oph, opw;      // dy patch (output pixels)
  int tpb, nblk;     // tiles per block, blocks per (sample, slab)
};

__device__ __forceinline__ int floor_div(int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); }

// S: the stride as a compile-time constant (1 | 2) -- the tap -> output index arithmetic of the data gradient divides by it
// nine times per pixel and thread; with a runtime divisor that arithmetic, not memory, bounded the 256^2 / 128^2 maps
// NPA / NPD: patch loads in flight per thread (input patch / dy patch), sized for the launch's tile and slab by the host (NP is the
// upper bound any plan may need): the registers they free let PF = the next tile's three raw patches be fetched while this tile's
// stencils run -- the large maps walk 4 - 8 tiles per block and ran their loop at 2 TB/s with every tile's loads exposed
template <int ACT, bool MULTI, int S, int NPA = NP, int NPD = NP, bool PF = false>
__global__ __launch_bounds__(T, 2) void mb_dw_bwd_kernel(const DwBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float dsm[];   // a1 patch [(th+2)(tw+2)][sw] | dy patch [oph*opw][sw]
  __shared__ __attribute__((aligned(16))) float tabA[4 * 128];   // scale | shift | mean | rstd of the slab's channels (GN1)
  __shared__ __attribute__((aligned(16))) float tabD[3 * 128];   // P | Q | R (GN2)
  __shared__ __attribute__((aligned(16))) float wtab[9 * 128];   // the slab's stencil weights [tap][sw] (registers are short here)
  __shared__ float gstat[GMAX][2];
  __shared__ float gc[GMAX][2];
  const int tid = threadIdx.x;
  const int bid = rn::xcd_remap(blockIdx.x, gridDim.x);
  const int slab = bid % a.nslab;
  const int tt = bid / a.nslab;
  const int ntile = a.tiles_h * a.tiles_w;
  // MULTI: a block walks `tpb` consecutive tiles of one sample: prologue and final reduction once (the weight-gradient sums
  // then live across the loop, so the stencil weights move from registers to LDS)
  const int blk = tt % a.nblk, sample = tt / a.nblk;
  const int tile_lo = MULTI ? blk * a.tpb : blk, tile_hi = MULTI ? min(tile_lo + a.tpb, ntile) : tile_lo + 1;
  const int C = a.c, SW = a.sw, SQ = SW >> 2, c0 = slab * SW;
  constexpr int s = S;
  const int tw_shift = 31 - __builtin_clz(a.tw);          // (the planner's tiles are powers of two)
  const int aph = a.th + 2, apw = a.tw + 2;
  float* a1p = dsm;
  float* dyp = dsm + (size_t)aph * apw * SW;
  const int g0 = c0 / a.in.cpg, ng = SW / a.in.cpg;
  // ---- everything from memory first: channel constants, statistics, both raw patches, the stencil weights
  ChanPre<1> pre_a, pre_d;
  prefetch_chan<1>(a.in.gamma, a.in.beta, c0, SW, tid, pre_a);
  prefetch_chan<1>(a.dy.nd.gamma, nullptr, c0, SW, tid, pre_d);
  const GroupPre gp_a = prefetch_groups(a.in, sample, g0, ng, tid);
  const GroupPre gp_d = prefetch_groups(a.dy.nd, sample, g0, ng, tid);
  const int atotal = aph * apw * SQ;
  const int dtotal = a.oph * a.opw * SQ;
  const __amdgpu_buffer_rsrc_t xs = make_rsrc(a.in.y + (size_t)sample * a.h * a.wd * C + c0, (unsigned)(a.h * a.wd * C - c0) * 4u);
  const __amdgpu_buffer_rsrc_t gs = make_rsrc(a.dy.g + (size_t)sample * a.oh * a.ow * C + c0, (unsigned)(a.oh * a.ow * C - c0) * 4u);
  const __amdgpu_buffer_rsrc_t ys = make_rsrc(a.dy.nd.y + (size_t)sample * a.oh * a.ow * C + c0, (unsigned)(a.oh * a.ow * C - c0) * 4u);
  float4 av[NPA], gv[NPD], yv2[NPD];
  int pka[NPA], pkd[NPD];
#pragma unroll
  for (int j = 0; j < NPA; ++j) pka[j] = patch_pack(tid + j * T, atotal, SQ, apw);
#pragma unroll
  for (int j = 0; j < NPD; ++j) pkd[j] = patch_pack(tid + j * T, dtotal, SQ, a.opw);
  auto load_patches = [&](int tile) {
    const int ih0 = (tile / a.tiles_w) * a.th, iw0 = (tile % a.tiles_w) * a.tw;
    const int ay0 = ih0 - a.pad_t, ax0 = iw0 - a.pad_l;
#pragma unroll
    for (int j = 0; j < NPA; ++j) {
      const PatchElem e = patch_at(pka[j], ay0, ax0, a.h, a.wd);
      av[j] = Vec<4>::load(xs, (e.live && e.inside) ? ((unsigned)e.pix * C + e.q * 4) * 4u : OOB);
    }
    // dy patch: output rows [oy0, +oph), cols [ox0, +opw): every output that touches the tile
    const int oy0 = floor_div(ih0 + a.pad_t - 2 + (s - 1), s), ox0 = floor_div(iw0 + a.pad_l - 2 + (s - 1), s);
#pragma unroll
    for (int j = 0; j < NPD; ++j) {
      const PatchElem e = patch_at(pkd[j], oy0, ox0, a.oh, a.ow);
      const unsigned off = (e.live && e.inside) ? ((unsigned)e.pix * C + e.q * 4) * 4u : OOB;
      gv[j] = Vec<4>::load(gs, off);
      yv2[j] = Vec<4>::load(ys, off);
    }
  };
  load_patches(tile_lo);
  const int lanes = T / SQ, q4 = tid % SQ, pl = tid / SQ;
  const bool active = pl < lanes;
  float4 wv[9];
  if (MULTI) {
    for (int e = tid; e < 9 * SQ; e += T)
      *reinterpret_cast<float4*>(&wtab[(e / SQ) * 128 + (e % SQ) * 4]) = *reinterpret_cast<const float4*>(a.w + (size_t)(e / SQ) * C + c0 + (e % SQ) * 4);
  } else {
#pragma unroll
    for (int t = 0; t < 9; ++t) wv[t] = *reinterpret_cast<const float4*>(a.w + (size_t)t * C + c0 + q4 * 4);
  }
  // ---- coefficient tables (the patches' LDS is the merge scratch until they are filled)
  dy_table(a.dy, sample, a.oh * a.ow, c0, SW, dsm, gstat, gc, tabD, 128, gp_d, pre_d, tid);
  group_stats(a.in, sample, a.h * a.wd, g0, ng, false, dsm, gstat, gp_a, tid);
  scale_shift_table(a.in, c0, SW, g0, gstat, tabA, tabA + 128, pre_a, tid);
  for (int i = tid; i < SW; i += T) {
    const int g = (c0 + i) / a.in.cpg - g0;
    tabA[256 + i] = gstat[g][0]; tabA[384 + i] = gstat[g][1];
  }
  __syncthreads();
  if (a.dbg == 1) { if (av[0].x == 123.456f && gv[0].x == 1.f && yv2[0].x == 1.f) a.go.out[0] = 0.f; return; }
  const bool drop = a.in.drop_rate > 0.f;
  const uint64_t seed = a.in.seed + (a.in.seed_dev ? *a.in.seed_dev : 0ull);
  const uint64_t samp_off = (uint64_t)sample * a.h * a.wd * C;
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  float4 wacc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) wacc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
  const float* __restrict__ y1 = a.in.y + (size_t)sample * a.h * a.wd * C + c0 + q4 * 4;
  float* __restrict__ go = a.go.out + (size_t)sample * a.h * a.wd * C + c0 + q4 * 4;
  auto process_tile = [&](int tile) {
    const int ih0 = (tile / a.tiles_w) * a.th, iw0 = (tile % a.tiles_w) * a.tw;
    const int ay0 = ih0 - a.pad_t, ax0 = iw0 - a.pad_l;
    const int oy0 = floor_div(ih0 + a.pad_t - 2 + (s - 1), s), ox0 = floor_div(iw0 + a.pad_l - 2 + (s - 1), s);
#pragma unroll
    for (int j = 0; j < NPA; ++j) {
      const PatchElem e = patch_at(pka[j], ay0, ax0, a.h, a.wd);
      if (e.live) {
        float4 o = norm_act_drop<ACT>(av[j], *reinterpret_cast<const float4*>(&tabA[e.q * 4]), *reinterpret_cast<const float4*>(&tabA[128 + e.q * 4]),
                                      a.in.act, drop, a.in.drop_rate, a.in.keep_scale, seed, samp_off + (uint64_t)e.pix * C + c0 + e.q * 4);
        if (!e.inside) o = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(&a1p[(size_t)(tid + j * T) * 4]) = o;
      }
    }
#pragma unroll
    for (int j = 0; j < NPD; ++j) {
      const PatchElem e = patch_at(pkd[j], oy0, ox0, a.oh, a.ow);
      if (e.live) {
        float4 o = dy_of(gv[j], yv2[j], tabD, 128, e.q * 4, false, 0.f, 1.f, 0ull, 0ull);
        if (!e.inside) o = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(&dyp[(size_t)(tid + j * T) * 4]) = o;
      }
    }
    __syncthreads();
    if (PF && tile + 1 < tile_hi) load_patches(tile + 1);     // (the raw patches of the next tile, under this tile's stencils)
    // ---- data gradient of the tile's pixels, then g1 and its sums
    if (active) {
      const float4 sc = *reinterpret_cast<const float4*>(&tabA[q4 * 4]), sh = *reinterpret_cast<const float4*>(&tabA[128 + q4 * 4]);
      const float4 mn = *reinterpret_cast<const float4*>(&tabA[256 + q4 * 4]), rs = *reinterpret_cast<const float4*>(&tabA[384 + q4 * 4]);
      for (int p = pl; p < a.th * a.tw; p += lanes) {
        const int ty = p >> tw_shift, tx = p - (ty << tw_shift);
        const int ih = ih0 + ty, iw = iw0 + tx;
        if (ih < a.h && iw < a.wd) {
          const float4 yv = *reinterpret_cast<const float4*>(y1 + (size_t)(ih * a.wd + iw) * C);
          float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
          for (int kh = 0; kh < 3; ++kh) {
            const int ohs = ih + a.pad_t - kh;
            const int oh_ = S == 1 ? ohs : (ohs >> 1);       // (ohs < 0 only with a 0 weight: see m below)
            const bool rok = ohs >= 0 && oh_ * s == ohs && oh_ < a.oh;
            const int py = min(max(oh_ - oy0, 0), a.oph - 1);
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
              const int ows = iw + a.pad_l - kw;
              const int ow_ = S == 1 ? ows : (ows >> 1);
              const float m = (rok && ows >= 0 && ow_ * s == ows && ow_ < a.ow) ? 1.f : 0.f;
              const int px = min(max(ow_ - ox0, 0), a.opw - 1);
              const float4 dv = *reinterpret_cast<const float4*>(&dyp[((size_t)py * a.opw + px) * SW + q4 * 4]);
              const float4 w4 = MULTI ? *reinterpret_cast<const float4*>(&wtab[(kh * 3 + kw) * 128 + q4 * 4]) : wv[kh * 3 + kw];
              acc.x = fmaf(dv.x * m, w4.x, acc.x); acc.y = fmaf(dv.y * m, w4.y, acc.y);
              acc.z = fmaf(dv.z * m, w4.z, acc.z); acc.w = fmaf(dv.w * m, w4.w, acc.w);
            }
          }
          const float yy[4] = {yv.x, yv.y, yv.z, yv.w}, dd[4] = {acc.x, acc.y, acc.z, acc.w};
          const float scv[4] = {sc.x, sc.y, sc.z, sc.w}, shv[4] = {sh.x, sh.y, sh.z, sh.w};
          const float mnv[4] = {mn.x, mn.y, mn.z, mn.w}, rsv[4] = {rs.x, rs.y, rs.z, rs.w};
          float g[4], mk[4] = {1.f, 1.f, 1.f, 1.f};
          if (drop) keep4(seed, samp_off + (uint64_t)(ih * a.wd + iw) * C + c0 + q4 * 4, a.in.drop_rate, a.in.keep_scale, mk);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float z = fmaf(yy[j], scv[j], shv[j]);
            const float xh = (yy[j] - mnv[j]) * rsv[j];
            const float t = dd[j] * actgrad_of<ACT>(z, a.in.act) * mk[j];
            g[j] = t;
            s1[j] += t; s2[j] = fmaf(t, xh, s2[j]);
          }
          *reinterpret_cast<float4*>(go + (size_t)(ih * a.wd + iw) * C) = make_float4(g[0], g[1], g[2], g[3]);
        }
      }
    }
    // ---- weight-gradient partial sums over the owned outputs (window origin inside the tile)
    if (active) {
      const int noy = (a.th + s - 1) / s, nox = (a.tw + s - 1) / s;      // tiles are aligned to the stride (host)
      const int oyb = ih0 / s, oxb = iw0 / s;
      for (int p = pl; p < noy * nox; p += lanes) {
        const int ty = p / nox, tx = p - ty * nox;
        const int oh_ = oyb + ty, ow_ = oxb + tx;
        if (oh_ < a.oh && ow_ < a.ow) {
          const float4 dv = *reinterpret_cast<const float4*>(&dyp[((size_t)(oh_ - oy0) * a.opw + (ow_ - ox0)) * SW + q4 * 4]);
#pragma unroll
          for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
              // input pixel (oh s - pad_t + kh, ..) in a1-patch coordinates (origin ih0 - pad_t)
              const float4 xv = *reinterpret_cast<const float4*>(&a1p[((size_t)(ty * s + kh) * apw + tx * s + kw) * SW + q4 * 4]);
              float4& t = wacc[kh * 3 + kw];
              t.x = fmaf(xv.x, dv.x, t.x); t.y = fmaf(xv.y, dv.y, t.y); t.z = fmaf(xv.z, dv.z, t.z); t.w = fmaf(xv.w, dv.w, t.w);
            }
        }
      }
    }
    __syncthreads();                               // the patches are dead: the next tile's, or the reduction scratch
  };
  // (no loads carried around the loop: ~100 loop-carried registers are copied at the back edge, which doubled them)
  process_tile(tile_lo);
  if (MULTI) {
#pragma nounroll
    for (int tile = tile_lo + 1; tile < tile_hi; ++tile) {
      if (!PF) load_patches(tile);
      process_tile(tile);
    }
  }
  if (a.dbg == 4) { if (wacc[0].x == 123.456f && s1[0] == 1.f) a.go.out[0] = 0.f; return; }
  // one exchange for all nine taps and both statistics: red[thread][36 + 8], then (tap, channel) outputs over the pixel lanes
  constexpr int RW = 44;
  float* red = dsm;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    red[tid * RW + t * 4 + 0] = wacc[t].x; red[tid * RW + t * 4 + 1] = wacc[t].y;
    red[tid * RW + t * 4 + 2] = wacc[t].z; red[tid * RW + t * 4 + 3] = wacc[t].w;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) { red[tid * RW + 36 + j] = s1[j]; red[tid * RW + 40 + j] = s2[j]; }
  __syncthreads();
  float* chan = dsm + T * RW;                       // [sw][2]
  const int brow = sample * a.nblk + blk;
  for (int e = tid; e < 11 * SW; e += T) {          // 9 taps + 2 statistics, channel fastest; pixel lanes in order
    const int k = e / SW, ch = e - k * SW;
    const int qd = ch >> 2, comp = ch & 3;
    float t = 0.f;
    for (int l = 0; l < lanes; ++l) t += red[(l * SQ + qd) * RW + k * 4 + comp];
    if (k < 9) {
      a.partial[((size_t)brow * 9 + k) * C + c0 + ch] = t;
    } else {
      chan[ch * 2 + (k - 9)] = t;
      a.go.planes[(size_t)(k - 9) * a.go.plane_stride + (size_t)brow * C + c0 + ch] = t;
    }
  }
  __syncthreads();
  const int cpg = a.in.cpg;
  if (tid < ng) {
    float t1 = 0.f, t2 = 0.f;
    for (int j = 0; j < cpg; ++j) {
      const float w = a.in.gamma[c0 + tid * cpg + j];
      t1 += w * chan[(tid * cpg + j) * 2 + 0]; t2 += w * chan[(tid * cpg + j) * 2 + 1];
    }
    a.go.grows.rows[(size_t)brow * a.go.grows.W + g0 + tid] = make_float2(t1, t2);
  }
}

}  // namespace

extern "C" size_t rn_mb_pointwise_rows(int n, int hw, int cin, int cout, int groups, rn_mb_rows* layout) {
  if (n < 1 || hw < 1 || cin < 4 || cout < 4 || groups < 1 || cout % groups) return 0;
  const PwCfg c = pw_cfg(n, hw, cin, cout);
  if (hw % c.bm) return 0;
  const int R = hw / c.bm, W = groups + rn::ceil_div(cout, c.bn);
  if (cout / groups > c.bn) return 0;      // (R > rn_mb_rows_max(): rn_mb_compact_rows before a consumer reads them)
  if (layout) { layout->rows_per_sample = R; layout->width = W; layout->bn = c.bn; }
  return (size_t)n * R * W * 8;
}

extern "C" int rn_mb_pointwise_fwd(const float* x, const rn_mb_norm* in, const float* residual, float* materialise, const float* w, float* y,
                                   int n, int hw, int cin, int cout, const rn_mb_rows* stat_out, int stat_groups, rn_stream_t stream) {
  RN_CHECK_ARG((x != nullptr) != (in != nullptr), "mb pointwise fwd: exactly one of x / in");
  RN_CHECK_ARG(w && y && n >= 1 && hw >= 1, "mb pointwise fwd: bad argument");
  RN_UNSUPPORTED(cin % 4 || cout % 4 || cin > KMAX || cout > KMAX, "mb pointwise fwd: cin=%d cout=%d (multiples of 4, <= %d)", cin, cout, KMAX);
  RN_UNSUPPORTED((double)n * hw * cin >= 536870912.0 || (double)n * hw * cout >= 536870912.0, "mb pointwise fwd: tensor >= 2 GiB");
  RN_CHECK_ARG(in || (!residual && !materialise), "mb pointwise fwd: residual / materialise come with `in`");
  PwFwdArgs a = {};
  a.dbg = dbg_word("pwf");
  a.x = x; a.res = residual; a.mat = materialise; a.w = w; a.y = y;
  a.n = n; a.hw = hw; a.cin = cin; a.cout = cout;
  if (in) {
    if (int e = fill_norm(in, &a.in, n, true, "mb pointwise fwd")) return e;
    RN_CHECK_ARG(in->c == cin, "mb pointwise fwd: in->c %d != cin %d", in->c, cin);
  }
  const PwCfg c = pw_cfg(n, hw, cin, cout);
  RN_UNSUPPORTED(hw % c.bm, "mb pointwise fwd: %d pixels per sample, tile height %d", hw, c.bm);
  a.tiles_n = rn::ceil_div(cout, c.bn);
  if (stat_out) {
    RN_CHECK_ARG(stat_out->rows && stat_groups >= 1 && cout % stat_groups == 0, "mb pointwise fwd: bad stat_out");
    rn_mb_rows want = {};
    RN_UNSUPPORTED(!rn_mb_pointwise_rows(n, hw, cin, cout, stat_groups, &want), "mb pointwise fwd: this shape cannot emit rows");
    RN_CHECK_ARG(want.rows_per_sample == stat_out->rows_per_sample && want.width == stat_out->width && want.bn == stat_out->bn,
                 "mb pointwise fwd: stat_out layout differs from rn_mb_pointwise_rows");
    a.ost.rows = (float2*)stat_out->rows; a.ost.R = want.rows_per_sample; a.ost.W = want.width; a.ost.bn = want.bn;
    a.ocpg = cout / stat_groups;
  }
  const long blocks = (long)n * hw / c.bm * a.tiles_n;
  const dim3 grid((unsigned)blocks);
  hipStream_t st = (hipStream_t)stream;
#define RN_PW(BM_, BN_, WM_, WN_, KS_, NST_)                                                                              \
  do {                                                                                                                \
    constexpr int TB = WM_ * WN_ * 64 * KS_;                                                                          \
    if (!in) hipLaunchKernelGGL((mb_pw_fwd_kernel<BM_, BN_, WM_, WN_, false, 0, KS_, NST_>), grid, dim3(TB), 0, st, a);      \
    else if (in->act == RN_ACT_NONE) hipLaunchKernelGGL((mb_pw_fwd_kernel<BM_, BN_, WM_, WN_, true, RN_ACT_NONE, KS_, NST_>), grid, dim3(TB), 0, st, a); \
    else if (in->act == RN_ACT_ELU) hipLaunchKernelGGL((mb_pw_fwd_kernel<BM_, BN_, WM_, WN_, true, RN_ACT_ELU, KS_, NST_>), grid, dim3(TB), 0, st, a);   \
    else hipLaunchKernelGGL((mb_pw_fwd_kernel<BM_, BN_, WM_, WN_, true, -1, KS_, NST_>), grid, dim3(TB), 0, st, a);          \
  } while (0)
  switch (c.id) {
    // (register stages: measured, more than one tile in flight per thread buys nothing here and costs occupancy on the big maps)
    case 1: RN_PW(128, 32, 4, 1, 1, 1); break;
    case 2: RN_PW(128, 64, 2, 2, 1, 1); break;
    case 3:
      if (c.ks == 8) RN_PW(32, 32, 1, 1, 8, 2);
      else RN_PW(32, 32, 1, 1, 4, 2);
      break;
    default:
      if (c.ks == 4) RN_PW(64, 64, 2, 2, 4, 1);
      else RN_PW(64, 64, 2, 2, 1, 1);
      break;
  }
#undef RN_PW
  RN_LAUNCH_CHECK();
  return RN_OK;
}

extern "C" size_t rn_mb_depthwise_rows(int n, int h, int w, int c, int stride, int groups, rn_mb_rows* layout) {
  if (n < 1 || h < 1 || w < 1 || c < 4 || c % 4 || groups < 1 || c % groups || (stride != 1 && stride != 2)) return 0;
  int oh, ow, pt, pl;
  rn::same_pad(h, 3, stride, &oh, &pt);
  rn::same_pad(w, 3, stride, &ow, &pl);
  DwPlan p;
  if (!dw_plan(n, oh, ow, c, stride, c / groups, &p)) return 0;
  const int R = p.nblk;
  if (layout) { layout->rows_per_sample = R; layout->width = groups; layout->bn = c; }
  return (size_t)n * R * groups * 8;
}

extern "C" int rn_mb_depthwise_fwd(const rn_mb_norm* in, const float* w, float* y, int n, int h, int wd, int stride, const rn_mb_rows* stat_out,
                                   int stat_groups, rn_stream_t stream) {
  RN_CHECK_ARG(in && w && y && n >= 1 && h >= 1 && wd >= 1 && (stride == 1 || stride == 2), "mb depthwise fwd: bad argument");
  DwFwdArgs a = {};
  a.dbg = dbg_word("dwf");
  if (int e = fill_norm(in, &a.in, n, true, "mb depthwise fwd")) return e;
  const int c = in->c;
  RN_UNSUPPORTED((double)n * h * wd * c >= 536870912.0, "mb depthwise fwd: tensor >= 2 GiB");
  a.w = w; a.y = y; a.n = n; a.h = h; a.wd = wd; a.c = c; a.stride = stride;
  rn::same_pad(h, 3, stride, &a.oh, &a.pad_t);
  rn::same_pad(wd, 3, stride, &a.ow, &a.pad_l);
  DwPlan p;
  // the slab must hold whole groups of the input GroupNorm and of the one that follows y (same c: stat_groups == in->groups here)
  RN_UNSUPPORTED(stat_out && stat_groups != in->groups, "mb depthwise fwd: the GroupNorms around a depthwise conv share their grouping");
  RN_UNSUPPORTED(!dw_plan(n, a.oh, a.ow, c, stride, a.in.cpg, &p), "mb depthwise fwd: no channel slab for c=%d groups=%d", c, in->groups);
  a.th = p.th; a.tw = p.tw; a.tiles_h = p.tiles_h; a.tiles_w = p.tiles_w; a.sw = p.sw; a.nslab = p.nslab; a.ph = p.ph; a.pw = p.pw;
  a.tpb = p.tpb; a.nblk = p.nblk;
  if (stat_out) {
    rn_mb_rows want = {};
    RN_UNSUPPORTED(!rn_mb_depthwise_rows(n, h, wd, c, stride, stat_groups, &want), "mb depthwise fwd: this shape cannot emit rows");
    RN_CHECK_ARG(stat_out->rows && want.rows_per_sample == stat_out->rows_per_sample && want.width == stat_out->width && want.bn == stat_out->bn,
                 "mb depthwise fwd: stat_out layout differs from rn_mb_depthwise_rows");
    a.ost.rows = (float2*)stat_out->rows; a.ost.R = want.rows_per_sample; a.ost.W = want.width; a.ost.bn = want.bn;
    a.ocpg = c / stat_groups;
  }
  const size_t lds = dw_lds_bytes(p);
  RN_UNSUPPORTED(lds > 64 * 1024, "mb depthwise fwd: patch of %zu bytes", lds);
  const dim3 grid((unsigned)((long)n * p.nblk * p.nslab));
  hipStream_t st = (hipStream_t)stream;
  static const bool pf = !(getenv("RN_MB_DW_NO_PF") && atoi(getenv("RN_MB_DW_NO_PF")));
  if (in->act == RN_ACT_ELU && pf && p.tpb > 1) hipLaunchKernelGGL((mb_dw_fwd_kernel<RN_ACT_ELU, true>), grid, dim3(T), lds, st, a);
  else if (in->act == RN_ACT_ELU) hipLaunchKernelGGL((mb_dw_fwd_kernel<RN_ACT_ELU, false>), grid, dim3(T), lds, st, a);
  else if (in->act == RN_ACT_RELU6) hipLaunchKernelGGL((mb_dw_fwd_kernel<RN_ACT_RELU6, false>), grid, dim3(T), lds, st, a);
  else hipLaunchKernelGGL((mb_dw_fwd_kernel<-1, false>), grid, dim3(T), lds, st, a);
  RN_LAUNCH_CHECK();
  return RN_OK;
}

extern "C" int rn_mb_apply(const rn_mb_norm* in, const float* residual, float* out, int n, int hw, rn_stream_t stream) {
  RN_CHECK_ARG(in && out && n >= 1 && hw >= 1, "mb apply: bad argument");
  ApplyArgs a = {};
  if (int e = fill_norm(in, &a.in, n, false, "mb apply")) return e;
  RN_UNSUPPORTED((double)n * hw * in->c >= 536870912.0, "mb apply: tensor >= 2 GiB");
  a.res = residual; a.out = out; a.n = n; a.hw = hw;
  const long elems = (long)hw * (in->c / 4);
  int chunks = (int)((elems + 4 * T - 1) / (4 * T));   // ~4 quads per thread
  if (chunks > hw) chunks = hw;
  if (chunks < 1) chunks = 1;
  a.ppb = rn::ceil_div(hw, chunks);
  hipLaunchKernelGGL(mb_apply_kernel, dim3((unsigned)rn::ceil_div(hw, a.ppb), (unsigned)n), dim3(T), 0, (hipStream_t)stream, a);
  RN_LAUNCH_CHECK();
  return RN_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// backward entry points
// ---------------------------------------------------------------------------------------------------------------------
namespace {
int fill_dy(const rn_mb_dy* s, DyDev* d, int n, int channels, const char* what) {
  RN_CHECK_ARG(s, "%s: null rn_mb_dy", what);
  if (s->dy) { d->dy = s->dy; return RN_OK; }
  RN_CHECK_ARG(s->norm && s->g && s->grows.rows, "%s: rn_mb_dy needs dy, or norm + g + grows", what);
  if (int e = fill_norm(s->norm, &d->nd, n, false, what)) return e;
  RN_CHECK_ARG(s->norm->c == channels, "%s: rn_mb_dy.norm->c %d != %d", what, s->norm->c, channels);
  RN_CHECK_ARG(!s->g_plain || s->norm->act == RN_ACT_NONE, "%s: g_plain needs a block without activation", what);
  d->nd.st.rows = nullptr;                       // backward kernels read mean / rstd back
  d->g = s->g; d->g_plain = s->g_plain ? 1 : 0;
  d->grows.rows = (float2*)s->grows.rows; d->grows.R = s->grows.rows_per_sample; d->grows.W = s->grows.width; d->grows.bn = s->grows.bn;
  RN_CHECK_ARG(d->grows.R >= 1 && d->grows.W >= d->nd.groups && d->grows.bn >= 1, "%s: bad grows layout", what);
  RN_UNSUPPORTED(d->grows.R > RMAX || d->nd.cpg > d->grows.bn, "%s: %d gradient rows per sample (<= %d)", what, d->grows.R, RMAX);
  return RN_OK;
}
int fill_gout(const rn_mb_gout* s, GoutDev* d, int n, int channels, const rn_mb_rows& want, const char* what) {
  RN_CHECK_ARG(s && s->out, "%s: null rn_mb_gout", what);
  d->out = s->out; d->add1 = s->add1; d->add2 = s->add2; d->store_plain = s->store_plain ? 1 : 0;
  if (!s->norm) return RN_OK;
  if (int e = fill_norm(s->norm, &d->nd, n, false, what)) return e;
  RN_CHECK_ARG(s->norm->c == channels, "%s: rn_mb_gout.norm->c %d != %d", what, s->norm->c, channels);
  d->nd.st.rows = nullptr;
  d->has_norm = 1;
  RN_CHECK_ARG(s->grows.rows && s->planes, "%s: rn_mb_gout with a norm needs grows + planes", what);
  RN_CHECK_ARG(s->grows.rows_per_sample == want.rows_per_sample && s->grows.width == want.width && s->grows.bn == want.bn,
               "%s: gout->grows layout differs from the *_bwd_rows query", what);
  d->grows.rows = (float2*)s->grows.rows; d->grows.R = want.rows_per_sample; d->grows.W = want.width; d->grows.bn = want.bn;
  d->planes = s->planes; d->plane_stride = (long)n * want.rows_per_sample * channels;
  return RN_OK;
}
// tile sizes and intra-block split-K of the pointwise backward's two halves: few data-gradient tiles with a long reduction
// (the small maps) take 32 x 32 tiles and several groups of waves per block.  A function of the shape alone (the row
// layout follows the data gradient's tile).
struct PwBwdCfg { int dpb, dks, wpb, wks; };
PwBwdCfg pw_bwd_cfg(int n, int hw, int cin, int cout) {
  static const bool no_ks = getenv("RN_MB_NO_SPLITK") != nullptr;
  static const bool no_t32 = getenv("RN_MB_NO_TILE32") != nullptr;
  const long dblocks = (long)n * hw / 64 * rn::ceil_div(cin, 64);
  const int nkt = rn::ceil_div(cout, BK);
  if (!no_ks && dblocks <= 96 && nkt >= 4) {
    if (no_t32) return {64, 4, 64, 4};
    return nkt >= 16 ? PwBwdCfg{32, 8, 64, 2} : PwBwdCfg{32, 4, 64, 1};
  }
  return {64, 1, 64, 1};
}
// weight-gradient split of the pointwise backward: pixels per split (divides hw, multiple of BK) and splits per sample;
// a block (ks groups) reduces >= 128 ks pixels -- every block pays for its coefficient tables once
void pw_wgrad_plan(int n, int hw, int cin, int cout, int pb, int ks, int* chunk, int* sps) {
  const int tiles = rn::ceil_div(cin, pb) * rn::ceil_div(cout, pb);
  int want = rn::ceil_div(384, tiles * n);             // splits per sample for ~384 blocks
  static const int big_blocks = getenv("RN_MB_WGRAD_BIG_BLOCKS") ? atoi(getenv("RN_MB_WGRAD_BIG_BLOCKS")) : 384;
  if (hw >= 16384 && big_blocks > 0) want = rn::ceil_div(big_blocks, tiles * n);   // (the 128^2 / 256^2 maps: tuning aid)
  if (const char* f = getenv("RN_MB_WGRAD_SPS")) { if (atoi(f) > 0) want = atoi(f); }   // tuning aid
  const int unit = 128 * ks;
  const int units = hw % unit == 0 ? hw / unit : 1;    // (hw is a multiple of 64; odd multiples: one split per sample)
  int best = 1;
  for (int d = 1; d <= units && d <= want; ++d)
    if (units % d == 0) best = d;
  *sps = best; *chunk = hw / best;
}
// the one-pass kernel of the large maps (mb_pw_bwd_big_kernel): which (cin, cout) it is built for, pixels per block
struct PwBigCfg { int nti, nto, ppb, bps; };
bool pw_big_cfg(int n, int hw, int cin, int cout, PwBigCfg* c) {
  static const bool on = !(getenv("RN_MB_PW_BIG") && atoi(getenv("RN_MB_PW_BIG")) == 0);
  static const int min_hw = getenv("RN_MB_PW_BIG_MIN_HW") ? atoi(getenv("RN_MB_PW_BIG_MIN_HW")) : 16384;
  if (!on || hw < min_hw || hw % 64 || cin % 4 || cout % 4) return false;
  const int nti = rn::ceil_div(cin, 16), nto = rn::ceil_div(cout, 16);
  const bool built = (nti == 2 && nto == 2) || (nti == 1 && nto == 6) || (nti == 2 && nto == 9) || (nti == 2 && nto == 1) ||
                     (nti == 6 && nto == 2) || (nti == 9 && nto == 2);
  if (!built) return false;
  if (nti == 9 && hw < 65536 && !getenv("RN_MB_PW_BIG_MIN_HW")) return false;   // (144 -> 24 at 128^2: measured slower than the tile kernel: its 36 + 72 accumulator
                                                                                 // registers leave the fused operand / epilogue pass spilling)
  // ~512 blocks (two per CU), a block's pixels a multiple of 64 that divides the sample; at most rn_mb_rows_max() rows per sample
  static const int target = getenv("RN_MB_PW_BIG_BLOCKS") ? atoi(getenv("RN_MB_PW_BIG_BLOCKS")) : 512;
  int ppb = 64;
  while ((long)n * hw / ppb > target && hw % (ppb * 2) == 0) ppb *= 2;
  while (hw / ppb > RMAX && hw % (ppb * 2) == 0) ppb *= 2;
  if (hw / ppb > RMAX) return false;
  c->nti = nti; c->nto = nto; c->ppb = ppb; c->bps = hw / ppb;
  return true;
}
struct DwBwdPlan { int th, tw, tiles_h, tiles_w, sw, nslab, oph, opw, tpb, nblk; };
bool dw_bwd_plan(int n, int h, int w, int c, int stride, int cpg, DwBwdPlan* p) {
  p->sw = dw_slab(c, cpg);
  if (!p->sw) return false;
  p->nslab = c / p->sw;
  static const int th_big = getenv("RN_MB_DWB_TH") ? atoi(getenv("RN_MB_DWB_TH")) : 8;
  static const int tw_big = getenv("RN_MB_DWB_TW") ? atoi(getenv("RN_MB_DWB_TW")) : 8;
  const bool big = (long)h * w >= 16384;
  int th = big ? th_big : 8, tw = big ? tw_big : 8;
  auto blocks = [&]() { return (long)n * rn::ceil_div(h, th) * rn::ceil_div(w, tw) * p->nslab; };
  auto too_big = [&]() { return (long)(th + 2) * (tw + 2) * (p->sw / 4) > NP * T || (long)(th / stride + 2) * (tw / stride + 2) * (p->sw / 4) > NP * T; };
  while ((blocks() < 384 && th * tw > 16) || (too_big() && th * tw > 16)) {
    if (th >= tw) th /= 2; else tw /= 2;
  }
  p->th = th; p->tw = tw;                              // powers of two >= 2: aligned to stride 1 / 2
  p->tiles_h = rn::ceil_div(h, th); p->tiles_w = rn::ceil_div(w, tw);
  p->oph = th / stride + 2; p->opw = tw / stride + 2;
  if ((long)(th + 2) * (tw + 2) * (p->sw / 4) > NP * T || (long)p->oph * p->opw * (p->sw / 4) > NP * T) return false;
  p->tpb = dw_tiles_per_block(blocks(), p->tiles_h * p->tiles_w);
  p->nblk = rn::ceil_div(p->tiles_h * p->tiles_w, p->tpb);
  return true;
}
size_t dw_bwd_lds_bytes(const DwBwdPlan& p) {
  size_t patches = ((size_t)(p.th + 2) * (p.tw + 2) + (size_t)p.oph * p.opw) * p.sw * 4;
  size_t scratch = (size_t)(T + GMAX) * 16;
  size_t red = (size_t)(T * 44 + 2 * p.sw) * 4;
  size_t m = patches > scratch ? patches : scratch;
  return m > red ? m : red;
}
}  // namespace

extern "C" size_t rn_mb_pointwise_bwd_rows(int n, int hw, int cin, int cout, int groups, rn_mb_rows* layout) {
  if (n < 1 || hw < 1 || cin < 4 || cout < 4 || groups < 1 || cin % groups || hw % 64) return 0;
  PwBigCfg big;
  if (pw_big_cfg(n, hw, cin, cout, &big)) {      // one row per block, one N-tile that spans cin
    if (layout) { layout->rows_per_sample = big.bps; layout->width = groups + 1; layout->bn = big.nti * 16; }
    return (size_t)n * big.bps * (groups + 1) * 8;
  }
  const int pb = pw_bwd_cfg(n, hw, cin, cout).dpb;
  const int R = hw / pb, W = groups + rn::ceil_div(cin, pb);
  if (cin / groups > pb) return 0;
  if (layout) { layout->rows_per_sample = R; layout->width = W; layout->bn = pb; }
  return (size_t)n * R * W * 8;
}
extern "C" size_t rn_mb_pointwise_bwd_workspace(int n, int hw, int cin, int cout) {
  if (n < 1 || hw < 64 || hw % 64 || cin < 4 || cout < 4) return 0;
  PwBigCfg big;
  if (pw_big_cfg(n, hw, cin, cout, &big)) return (size_t)n * big.bps * cin * cout * sizeof(float);
  int chunk, sps;
  const PwBwdCfg c = pw_bwd_cfg(n, hw, cin, cout);
  pw_wgrad_plan(n, hw, cin, cout, c.wpb, c.wks, &chunk, &sps);
  return (size_t)n * sps * cin * cout * sizeof(float);
}

extern "C" int rn_mb_pointwise_bwd(const float* x, const rn_mb_norm* in, const rn_mb_dy* dy, const float* w, float* dw, const rn_mb_gout* gout,
                                   int n, int hw, int cin, int cout, void* workspace, size_t workspace_bytes, rn_stream_t stream,
                                   rn_reduce_list* defer) {
  rn::DeferScope defer_scope_(defer);
  RN_CHECK_ARG((x != nullptr) != (in != nullptr), "mb pointwise bwd: exactly one of x / in");
  RN_CHECK_ARG(w && dw && dy && gout && workspace && n >= 1 && hw >= 1, "mb pointwise bwd: bad argument");
  RN_UNSUPPORTED(cin % 4 || cout % 4 || cin > KMAX || cout > KMAX, "mb pointwise bwd: cin=%d cout=%d (multiples of 4, <= %d)", cin, cout, KMAX);
  RN_UNSUPPORTED(hw % 64, "mb pointwise bwd: %d pixels per sample, tile height 64", hw);
  RN_UNSUPPORTED((double)n * hw * cin >= 536870912.0 || (double)n * hw * cout >= 536870912.0, "mb pointwise bwd: tensor >= 2 GiB");
  PwBwdArgs a = {};
  a.dbg = dbg_word("pwb");
  a.x = x; a.w = w; a.n = n; a.hw = hw; a.cin = cin; a.cout = cout;
  if (in) {
    if (int e = fill_norm(in, &a.in, n, false, "mb pointwise bwd")) return e;
    RN_CHECK_ARG(in->c == cin, "mb pointwise bwd: in->c %d != cin %d", in->c, cin);
    a.in.st.rows = nullptr;
    a.has_in = 1;
  }
  if (int e = fill_dy(dy, &a.dy, n, cout, "mb pointwise bwd")) return e;
  rn_mb_rows want = {};
  if (gout->norm) RN_UNSUPPORTED(!rn_mb_pointwise_bwd_rows(n, hw, cin, cout, gout->norm->groups, &want), "mb pointwise bwd: this shape cannot emit gradient rows");
  if (int e = fill_gout(gout, &a.go, n, cin, want, "mb pointwise bwd")) return e;
  PwBigCfg big;
  if (pw_big_cfg(n, hw, cin, cout, &big)) {
    PwBigArgs b = {};
    b.dbg = dbg_word("pwB");
    b.x = a.x; b.in = a.in; b.has_in = a.has_in; b.dy = a.dy; b.w = w; b.go = a.go;
    b.n = n; b.hw = hw; b.cin = cin; b.cout = cout; b.ppb = big.ppb; b.bps = big.bps;
    const int nsplit = n * big.bps;
    const size_t need = (size_t)nsplit * cin * cout * sizeof(float);
    if (workspace_bytes < need) { rn::set_error("mb pointwise bwd: workspace %zu < %zu bytes", workspace_bytes, need); return RN_EWORKSPACE; }
    b.slab = (float*)workspace;
    // the linear conv: the block the data gradient enters is the conv's own input block (no addends there)
    const bool y_is_a = a.has_in && a.go.has_norm && a.go.nd.y == a.in.y && !a.go.add1 && !a.go.add2;
    const dim3 grid((unsigned)nsplit);
    hipStream_t st = (hipStream_t)stream;
#define RN_PWBIG(NTI_, NTO_)                                                                                                    \
  do {                                                                                                                          \
    constexpr size_t lds = (size_t)pw_big_lds_floats(NTI_, NTO_) * 4;                                                           \
    static const bool attr_ = (hipFuncSetAttribute((const void*)mb_pw_bwd_big_kernel<NTI_, NTO_, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess) & \
                              (hipFuncSetAttribute((const void*)mb_pw_bwd_big_kernel<NTI_, NTO_, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess); \
    RN_UNSUPPORTED(!attr_, "mb pointwise bwd: %zu bytes of LDS per block refused", lds);                                                                                                                \
    if (y_is_a) hipLaunchKernelGGL((mb_pw_bwd_big_kernel<NTI_, NTO_, true>), grid, dim3(T), lds, st, b);                        \
    else hipLaunchKernelGGL((mb_pw_bwd_big_kernel<NTI_, NTO_, false>), grid, dim3(T), lds, st, b);                              \
  } while (0)
    if (big.nti == 2 && big.nto == 2) RN_PWBIG(2, 2);
    else if (big.nti == 1 && big.nto == 6) RN_PWBIG(1, 6);
    else if (big.nti == 2 && big.nto == 9) RN_PWBIG(2, 9);
    else if (big.nti == 2 && big.nto == 1) RN_PWBIG(2, 1);
    else if (big.nti == 6 && big.nto == 2) RN_PWBIG(6, 2);
    else RN_PWBIG(9, 2);
#undef RN_PWBIG
    RN_LAUNCH_CHECK();
    return rn::launch_reduce_rows((const float*)workspace, dw, (int64_t)cin * cout, nsplit, 0, st);
  }
  const PwBwdCfg cfg = pw_bwd_cfg(n, hw, cin, cout);
  a.d_tiles_n = rn::ceil_div(cin, cfg.dpb);
  a.dblocks = n * hw / cfg.dpb * a.d_tiles_n;
  a.w_tiles_m = rn::ceil_div(cin, cfg.wpb); a.w_tiles_n = rn::ceil_div(cout, cfg.wpb);
  pw_wgrad_plan(n, hw, cin, cout, cfg.wpb, cfg.wks, &a.chunk, &a.sps);
  const int nsplit = n * a.sps;
  const size_t need = (size_t)nsplit * cin * cout * sizeof(float);
  if (workspace_bytes < need) { rn::set_error("mb pointwise bwd: workspace %zu < %zu bytes", workspace_bytes, need); return RN_EWORKSPACE; }
  a.slab = nsplit == 1 ? dw : (float*)workspace;
  static const int wfirst_env = getenv("RN_MB_WGRAD_FIRST") ? atoi(getenv("RN_MB_WGRAD_FIRST")) : 1;
  a.wfirst = (wfirst_env && a.dblocks > 1024) ? 1 : 0;      // (only where the grid does not fit the chip in one go)
  const dim3 grid((unsigned)(a.dblocks + nsplit * a.w_tiles_m * a.w_tiles_n));
  hipStream_t st = (hipStream_t)stream;
  const int act_in = in ? in->act : -2, act_out = gout->norm ? gout->norm->act : RN_ACT_NONE;
#define RN_PWB(DPB_, DKS_, WPB_, WKS_)                                                                                               \
  do {                                                                                                                               \
    constexpr int TB = (DPB_ / 32) * (DPB_ / 32) * 64 * DKS_;                                                                        \
    if (act_in == -2 && act_out == RN_ACT_NONE) hipLaunchKernelGGL((mb_pw_bwd_kernel<-2, RN_ACT_NONE, DPB_, DKS_, WPB_, WKS_>), grid, dim3(TB), 0, st, a); \
    else if (act_in == -2) hipLaunchKernelGGL((mb_pw_bwd_kernel<-2, -1, DPB_, DKS_, WPB_, WKS_>), grid, dim3(TB), 0, st, a);          \
    else if (act_in == RN_ACT_ELU && act_out == RN_ACT_ELU) hipLaunchKernelGGL((mb_pw_bwd_kernel<RN_ACT_ELU, RN_ACT_ELU, DPB_, DKS_, WPB_, WKS_>), grid, dim3(TB), 0, st, a); \
    else hipLaunchKernelGGL((mb_pw_bwd_kernel<-1, -1, DPB_, DKS_, WPB_, WKS_>), grid, dim3(TB), 0, st, a);                            \
  } while (0)
  if (cfg.dpb == 32 && cfg.dks == 8) RN_PWB(32, 8, 64, 2);
  else if (cfg.dpb == 32) RN_PWB(32, 4, 64, 1);
  else if (cfg.dks == 4) RN_PWB(64, 4, 64, 4);
  else RN_PWB(64, 1, 64, 1);
#undef RN_PWB
  RN_LAUNCH_CHECK();
  if (nsplit == 1) return RN_OK;
  return rn::launch_reduce_rows((const float*)workspace, dw, (int64_t)cin * cout, nsplit, 0, st);
}

extern "C" size_t rn_mb_depthwise_bwd_rows(int n, int h, int w, int c, int stride, int groups, rn_mb_rows* layout) {
  if (n < 1 || h < 1 || w < 1 || c < 4 || c % 4 || groups < 1 || c % groups || (stride != 1 && stride != 2)) return 0;
  DwBwdPlan p;
  if (!dw_bwd_plan(n, h, w, c, stride, c / groups, &p)) return 0;
  const int R = p.nblk;
  if (layout) { layout->rows_per_sample = R; layout->width = groups; layout->bn = c; }
  return (size_t)n * R * groups * 8;
}
extern "C" size_t rn_mb_depthwise_bwd_workspace(int n, int h, int w, int c, int stride) {
  if (n < 1 || h < 1 || w < 1 || c < 4 || c % 4 || (stride != 1 && stride != 2)) return 0;
  // weight-gradient partial sums, one row per (sample, input tile): the planner's tiles are never smaller than 4 x 4
  return (size_t)n * rn::ceil_div(h, 4) * rn::ceil_div(w, 4) * 9 * c * sizeof(float);
}

extern "C" int rn_mb_depthwise_bwd(const rn_mb_norm* in, const rn_mb_dy* dy, const float* w, float* dw, const rn_mb_gout* gout, int n, int h,
                                   int wd, int stride, void* workspace, size_t workspace_bytes, rn_stream_t stream, rn_reduce_list* defer) {
  rn::DeferScope defer_scope_(defer);
  RN_CHECK_ARG(in && dy && w && dw && gout && gout->norm && workspace && n >= 1 && h >= 1 && wd >= 1 && (stride == 1 || stride == 2),
               "mb depthwise bwd: bad argument");
  RN_CHECK_ARG(!dy->dy && !gout->store_plain && !gout->add1 && !gout->add2, "mb depthwise bwd: dy comes as (g, norm, grows); gout stores g");
  DwBwdArgs a = {};
  a.dbg = dbg_word("dwb");
  if (int e = fill_norm(in, &a.in, n, false, "mb depthwise bwd")) return e;
  a.in.st.rows = nullptr;
  const int c = in->c;
  RN_UNSUPPORTED((double)n * h * wd * c >= 536870912.0, "mb depthwise bwd: tensor >= 2 GiB");
  if (int e = fill_dy(dy, &a.dy, n, c, "mb depthwise bwd")) return e;
  RN_CHECK_ARG(!a.dy.g_plain && dy->norm->groups == in->groups, "mb depthwise bwd: the GroupNorms around a depthwise conv share their grouping");
  a.w = w; a.n = n; a.h = h; a.wd = wd; a.c = c; a.stride = stride;
  rn::same_pad(h, 3, stride, &a.oh, &a.pad_t);
  rn::same_pad(wd, 3, stride, &a.ow, &a.pad_l);
  DwBwdPlan p;
  RN_UNSUPPORTED(!dw_bwd_plan(n, h, wd, c, stride, a.in.cpg, &p), "mb depthwise bwd: no channel slab for c=%d groups=%d", c, in->groups);
  a.th = p.th; a.tw = p.tw; a.tiles_h = p.tiles_h; a.tiles_w = p.tiles_w; a.sw = p.sw; a.nslab = p.nslab; a.oph = p.oph; a.opw = p.opw;
  a.tpb = p.tpb; a.nblk = p.nblk;
  rn_mb_rows want = {};
  RN_UNSUPPORTED(!rn_mb_depthwise_bwd_rows(n, h, wd, c, stride, in->groups, &want), "mb depthwise bwd: this shape cannot emit gradient rows");
  if (int e = fill_gout(gout, &a.go, n, c, want, "mb depthwise bwd")) return e;
  RN_CHECK_ARG(gout->norm->y == in->y, "mb depthwise bwd: gout->norm is the block of `in`");
  const int nrows = n * p.nblk;
  const size_t need = (size_t)nrows * 9 * c * sizeof(float);
  if (workspace_bytes < need) { rn::set_error("mb depthwise bwd: workspace %zu < %zu bytes", workspace_bytes, need); return RN_EWORKSPACE; }
  a.partial = (float*)workspace;
  const size_t lds = dw_bwd_lds_bytes(p);
  RN_UNSUPPORTED(lds > 64 * 1024, "mb depthwise bwd: patches of %zu bytes", lds);
  const dim3 grid((unsigned)((long)nrows * p.nslab));
  hipStream_t st = (hipStream_t)stream;
  RN_CHECK_ARG((p.tw & (p.tw - 1)) == 0, "mb depthwise bwd: tile width %d is not a power of two", p.tw);
#define RN_DWB(ACT_, MULTI_)                                                                                           \
  do {                                                                                                                 \
    if (stride == 1) hipLaunchKernelGGL((mb_dw_bwd_kernel<ACT_, MULTI_, 1>), grid, dim3(T), lds, st, a);               \
    else hipLaunchKernelGGL((mb_dw_bwd_kernel<ACT_, MULTI_, 2>), grid, dim3(T), lds, st, a);                           \
  } while (0)
  // (the large maps' launches: loads sized for the plan, next tile prefetched; RN_MB_DWB_PF=0: the plain variant)
  static const bool pf_on = !(getenv("RN_MB_DWB_PF") && atoi(getenv("RN_MB_DWB_PF")) == 0);
  const int npa = rn::ceil_div((p.th + 2) * (p.tw + 2) * (p.sw / 4), T), npd = rn::ceil_div(p.oph * p.opw * (p.sw / 4), T);
  if (in->act == RN_ACT_ELU && p.tpb > 1 && pf_on && stride == 1 && npa <= 4 && npd <= 4)
    hipLaunchKernelGGL((mb_dw_bwd_kernel<RN_ACT_ELU, true, 1, 4, 4, true>), grid, dim3(T), lds, st, a);
  else if (in->act == RN_ACT_ELU && p.tpb > 1 && pf_on && stride == 1 && npa <= 5 && npd <= 5)
    hipLaunchKernelGGL((mb_dw_bwd_kernel<RN_ACT_ELU, true, 1, 5, 5, true>), grid, dim3(T), lds, st, a);
  else if (in->act == RN_ACT_ELU && p.tpb > 1 && pf_on && stride == 2 && npa <= 4 && npd <= 2)
    hipLaunchKernelGGL((mb_dw_bwd_kernel<RN_ACT_ELU, true, 2, 4, 2, true>), grid, dim3(T), lds, st, a);
  else if (in->act == RN_ACT_ELU && p.tpb > 1 && pf_on && stride == 2 && npa <= 5 && npd <= 2)
    hipLaunchKernelGGL((mb_dw_bwd_kernel<RN_ACT_ELU, true, 2, 5, 2, true>), grid, dim3(T), lds, st, a);
  else if (in->act == RN_ACT_ELU && p.tpb == 1) RN_DWB(RN_ACT_ELU, false);
  else if (in->act == RN_ACT_ELU) RN_DWB(RN_ACT_ELU, true);
  else if (in->act == RN_ACT_RELU6) RN_DWB(RN_ACT_RELU6, true);
  else RN_DWB(-1, true);
#undef RN_DWB
  RN_LAUNCH_CHECK();
  return rn::launch_reduce_rows((const float*)workspace, dw, (int64_t)9 * c, nrows, 0, st);
}

// ---------------------------------------------------------------------------------------------------------------------
// row compaction
// ---------------------------------------------------------------------------------------------------------------------
extern "C" int rn_mb_rows_max(void) { return RMAX; }

extern "C" size_t rn_mb_compact_rows_layout(int n, const rn_mb_rows* in, rn_mb_rows* out) {
  if (!in || !out || n < 1 || in->rows_per_sample < 1 || in->width < 1 || in->width > T) return 0;
  const int F = rn::ceil_div(in->rows_per_sample, 8);
  out->rows_per_sample = rn::ceil_div(in->rows_per_sample, F); out->width = in->width; out->bn = in->bn;
  return (size_t)n * out->rows_per_sample * out->width * 8;
}

extern "C" int rn_mb_compact_rows(const rn_mb_rows* in, const rn_mb_rows* out, int n, rn_stream_t stream) {
  RN_CHECK_ARG(in && out && in->rows && out->rows && n >= 1, "mb compact rows: bad argument");
  rn_mb_rows want = {};
  RN_CHECK_ARG(rn_mb_compact_rows_layout(n, in, &want) && want.rows_per_sample == out->rows_per_sample && want.width == out->width && want.bn == out->bn,
               "mb compact rows: `out` layout differs from rn_mb_compact_rows_layout");
  CompactArgs a = {(const float2*)in->rows, (float2*)out->rows, in->rows_per_sample, in->width, rn::ceil_div(in->rows_per_sample, 8), want.rows_per_sample};
  hipLaunchKernelGGL(mb_compact_rows_kernel, dim3((unsigned)a.Rout, (unsigned)n), dim3(T), 0, (hipStream_t)stream, a);
  RN_LAUNCH_CHECK();
  return RN_OK;
}
