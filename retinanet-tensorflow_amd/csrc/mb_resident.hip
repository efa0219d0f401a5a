// XCD-resident small-map section of the MobileNetV2 bottleneck chain (mobilenet_v2.py:41-94,120-223; GroupNorm variant
// normalization.py:20-35): the kernels of mbconv.hip for the maps of <= 32 x 32 pixels -- about 33 dependent launches of
// 9 - 20 us each that move 1 - 3 us of data -- as PHASES of one launch.
//
// Why it can be one launch: GroupNorm statistics never cross samples, and a sample's working set on these maps (<= 3 MB) fits
// the 4 MB L2 of ONE XCD.  The grid is 8 x B blocks; the blocks with equal blockIdx.x % 8 are dispatched to one XCD
// (round-robin placement, verified at run time from HW_REG_XCC_ID: a mismatch sets the error word) and form the CLUSTER of
// sample blockIdx.x % 8 (+ 8, + 16, ... for batches beyond 8).  A phase is what one launch-ordered kernel was: its
// virtual blocks are dealt to the cluster's blocks (rank, rank + B, ...), then the cluster meets at a barrier:
//   s_waitcnt vmcnt(0) (every thread: its stores have reached the XCD's L2) -> one atomic add on the cluster's counter ->
//   relaxed sc1 polling with s_sleep -> the next phase reads what the phase wrote from the SAME L2 with plain loads.
// No buffer_wbl2, no cross-XCD fence, no chip-wide barrier: measured 0.8 - 1.4 us per barrier for 32 - 64 blocks
// (tools/micro/xcd_cluster.hip) against 4 - 9 us chip-wide, and against ~2 us of launch boundary + the cold first round
// trips of a new kernel (tools/micro/boundary.hip).
// Requirements on the caller (rn_mb_resident_fwd checks what it can):
//   * every buffer a phase writes is written by that phase ONLY and read by LATER phases only (then no L1 can hold a stale
//     line: the vector L1 is invalidated at launch and write-through afterwards);
//   * `sync` (rn_mb_resident_sync_bytes(), zero-initialised, private to the stream) is written by these launches only.
// Co-residency: B <= 32 CUs x blocks per CU of this kernel (queried); a block that never arrives (the chip is shared with a
// kernel that never ends) ends the wait after `spin` polls with the error word set instead of hanging.
// Results: the same arithmetic per element as the launch-ordered kernels (same tile code, same fixed-order row merges), a
// different TILING on some layers (rows are summed in a different grouping: differences of one fp32 rounding in the
// statistics).  Bit-reproducible from run to run: the virtual block -> block assignment is static.
#include "mb_common.h"

namespace {

constexpr int RES_MAXP = 14;        // phases per launch: the kernarg segment is limited to 4 KB
constexpr int RES_BMAX = 96;        // blocks per cluster at most (barrier cost grows with the arrivals: 0.8 / 1.4 / 2.6 us at 32 / 64 / 128)
constexpr int PW64_FLOATS = 64 * LDK + BK * 64;             // 64 x 64 tile, one group
constexpr int PW32_FLOATS = 4 * (32 * LDK + BK * 32);       // 32 x 32 tile, four split-K groups of one wave

enum { RES_PW64 = 0, RES_PW32 = 1, RES_DW = 2 };

struct ResPhase {
  int kind, nvb;                    // virtual blocks per sample
  const void* warm; unsigned warm_bytes, pad_;      // this phase's weights: touched during the PREVIOUS phase (L2 warm-up)
  union U { PwFwdArgs pw; DwFwdArgs dw; } u;
};
struct ResArgs {
  unsigned* sync;                   // [8][64] words: per cluster {barrier counter, exit ticket, xcc mask}; word 512: error word
  int n, nphase, B, spin;
  int stamp0, pad_;                 // >= 0 (RN_MB_RES_STAMPS=1, measurements): rank 0 of cluster 0 writes three clock stamps per phase
                                    // (start, work done, barrier passed) to the 64-bit words sync[640 / 2 + 3 (stamp0 + p) ...]
  ResPhase ph[RES_MAXP];
};
static_assert(sizeof(ResArgs) <= 4096, "kernarg segment");

__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15u; }   // hwreg(HW_REG_XCC_ID, 0, 4)

// All blocks of the cluster have finished the phase and their stores are visible in the XCD's L2.  false: timed out.
__device__ __forceinline__ bool cluster_barrier(unsigned* ctr, unsigned target, int spin, unsigned* err, int* s_flag, int tid) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int tries = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (++tries > spin) {
        *s_flag = 1;
        __hip_atomic_fetch_or(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
  }
  __syncthreads();
  return *s_flag == 0;
}

// One 4-byte load per 128-byte line of [p, p + bytes), dealt over the threads of the blocks that call: brings the NEXT phase's
// weights into this XCD's L2.  Called by the blocks that have no tile in the current phase (a wave's loads return in order:
// a working block would wait for these HBM round trips at its next operand wait; an idle one only arrives a little later at
// the barrier it would have waited at anyway).  The XOR of the words is returned so that the loads are real.
__device__ __forceinline__ unsigned warm_l2(const void* p, unsigned bytes, int idle_rank, int nidle, int tid) {
  const unsigned* c = reinterpret_cast<const unsigned*>(p);
  unsigned acc = 0;
  for (unsigned off = ((unsigned)idle_rank * T + tid) * 32u; off < bytes / 4u; off += (unsigned)nidle * T * 32u) acc ^= c[off];
  return acc;
}

// ---------------------------------------------------------------------------------------------------------------------
// pointwise phase: the tile code of mb_pw_fwd_kernel<BM, BN, WM, WN, NORM = true, ACT, KS, NST>; the sample's statistics
// and the per-channel tables are built ONCE per block and phase (not per tile), then the block walks its virtual blocks.
// ---------------------------------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, int ACT, int KS, int NST>
__device__ __forceinline__ void res_pw_phase(const PwFwdArgs& a, const int sample, const int rank, const int B, const int nvb, float* smem,
                                             float* tab, float (*gstat)[2], unsigned long long* stamps) {
  constexpr int TG = WM * WN * 64;
  constexpr bool RES = ACT != RN_ACT_ELU && ACT != RN_ACT_RELU6 && ACT != RN_ACT_RELU;   // a residual comes with a linear block only
  static_assert(TG * KS == T, "256-thread blocks");
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int KQ = BK / 4, A_RPP = TG / KQ, A_PASS = BM / A_RPP;
  constexpr int NQ = BN / 4, B_RPP = TG / NQ, B_PASS = BK / B_RPP;
  static_assert(A_PASS >= 1 && B_PASS >= 1 && BM % A_RPP == 0 && BK % B_RPP == 0, "tile/threads mismatch");
  constexpr int OPF = BM * LDK + BK * BN;
  static_assert(KS * OPF * 4 >= (T + GMAX) * 16 && OPF >= (WM + 1) * BN * 2, "operand tiles double as scratch");
  static_assert(KS == 1 || KS * OPF >= (KS - 1) * TM * TN * 16 * TG, "operand tiles double as the split-K exchange");
  if (rank >= nvb) return;                            // (block-uniform: this block has no tile in this phase)
  const int tid = threadIdx.x, grp = tid / TG, lt = tid % TG, lane = lt & 63, wave = lt >> 6;
  float* As = smem + grp * OPF;
  float* Bs = As + BM * LDK;
  const int wm = wave / WN, wn = wave % WN;
  const int K = a.cin, N = a.cout, M = a.n * a.hw;
  const int tiles_m = a.hw / BM;                      // per sample
  const __amdgpu_buffer_rsrc_t xa = make_rsrc(a.in.y, (unsigned)M * K * 4u);
  const __amdgpu_buffer_rsrc_t xr = make_rsrc(a.res ? a.res : a.in.y, (unsigned)M * K * 4u);
  const __amdgpu_buffer_rsrc_t wb = make_rsrc(a.w, (unsigned)K * N * 4u);
  const bool has_res = RES && a.res != nullptr;
  const int kq = lt % KQ, arow = lt / KQ;
  const int nq = lt % NQ;
  const int nk = (K + BK - 1) / BK, nit = (nk + KS - 1) / KS;
  int vb = rank;
  int tile_n = vb % a.tiles_n, m0 = (sample * tiles_m + vb / a.tiles_n) * BM, n0 = tile_n * BN;
  unsigned boff0 = (n0 + nq * 4) < N ? ((unsigned)(lt / NQ) * N + n0 + nq * 4) * 4u : OOB;
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
  ChanPre<KMAX / T> pre;
  prefetch_chan(a.in.gamma, a.in.beta, 0, K, tid, pre);
  const GroupPre gpre = {0.f, 1.f};
#pragma unroll
  for (int st = 0; st < NST; ++st)
    if (st < nit) load_tiles(st, st);
  if (stamps) stamps[3] = wall_clock64();
  // the sample's statistics while the first tiles are in flight; the cluster's first block publishes them (the backward pass reads them)
  group_stats(a.in, sample, a.hw, 0, a.in.groups, rank == 0, smem, gstat, gpre, tid);
  if (stamps) stamps[4] = wall_clock64();
  scale_shift_table(a.in, 0, K, 0, gstat, tab, tab + KMAX, pre, tid);
  if (stamps) stamps[5] = wall_clock64();
  int nstamp = 6;
  const bool drop = a.in.drop_rate > 0.f;
  const uint64_t seed = a.in.seed + (a.in.seed_dev ? *a.in.seed_dev : 0ull);
  for (;;) {
    const bool write_mat = a.mat != nullptr && tile_n == 0;
    auto store_tiles = [&](int it, const int st) {
      const int k = (it * KS + grp) * BK + kq * 4;
#pragma unroll
      for (int i = 0; i < A_PASS; ++i) {
        float4 v = ra[st][i];
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
    if (stamps && nstamp < 15) stamps[nstamp++] = wall_clock64();      // K loop done
    // the next virtual block's first operand tiles go out before this one's epilogue
    const int m0_cur = m0, n0_cur = n0, tile_n_cur = tile_n;
    vb += B;
    const bool more = vb < nvb;
    if (more) {
      tile_n = vb % a.tiles_n; m0 = (sample * tiles_m + vb / a.tiles_n) * BM; n0 = tile_n * BN;
      boff0 = (n0 + nq * 4) < N ? ((unsigned)(lt / NQ) * N + n0 + nq * 4) * 4u : OOB;
#pragma unroll
      for (int st = 0; st < NST; ++st)
        if (st < nit) load_tiles(st, st);
    }
    sum_groups<KS, TM * TN, TG>(&acc[0][0], smem, grp, lt);
    if (grp == 0) store_tile<BM, BN, WM, WN>(acc, a.y, nullptr, m0_cur, n0_cur, M, N, N, wm, wn, lane);
    if (a.ost.rows) {
      float2* row = a.ost.rows + ((size_t)sample * a.ost.R + (m0_cur - sample * a.hw) / BM) * a.ost.W;
      float s1[TN], s2[TN];
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        s1[tn] = 0.f; s2[tn] = 0.f;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int r = 0; r < 16; ++r) { const float v = acc[tm][tn][r]; s1[tn] += v; s2[tn] = fmaf(v, v, s2[tn]); }
      }
      reduce_group_rows<BM, BN, WM, WN>(s1, s2, smem, row, n0_cur, N, a.ocpg, tile_n_cur, tid, nullptr, nullptr, nullptr);
    }
    if (stamps && nstamp < 15) stamps[nstamp++] = wall_clock64();      // epilogue done
    if (!more) break;
    __syncthreads();                          // the epilogue's scratch is the next tile's operand space
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// depthwise phase: the code of mb_dw_fwd_kernel<ACT, PF = true>; a block owns (channel slab, run of tiles) of its sample
// ---------------------------------------------------------------------------------------------------------------------
template <int ACT>
__device__ __forceinline__ void res_dw_phase(const DwFwdArgs& a, const int sample, const int rank, const int nvb, float* dsm, float* tab,
                                             float (*gstat)[2], unsigned long long* stamps) {
  if (rank >= nvb) return;
  const int tid = threadIdx.x;
  const int slab = rank % a.nslab, blk = rank / a.nslab;
  const int ntile = a.tiles_h * a.tiles_w;
  const int tile_lo = blk * a.tpb, tile_hi = min(tile_lo + a.tpb, ntile);
  const int C = a.c, SW = a.sw, SQ = SW >> 2, c0 = slab * SW;
  const int g0 = c0 / a.in.cpg, ng = SW / a.in.cpg;
  ChanPre<1> pre;
  prefetch_chan(a.in.gamma, a.in.beta, c0, SW, tid, pre);
  const GroupPre gpre = {0.f, 1.f};
  const int total = a.ph * a.pw * SQ;
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
  if (stamps) stamps[3] = wall_clock64();
  group_stats(a.in, sample, a.h * a.wd, g0, ng, blk == 0, dsm, gstat, gpre, tid);
  if (stamps) stamps[4] = wall_clock64();
  scale_shift_table(a.in, c0, SW, g0, gstat, tab, tab + 128, pre, tid);
  if (stamps) stamps[5] = wall_clock64();
  int nstamp = 6;
  const bool drop = a.in.drop_rate > 0.f;
  const uint64_t seed = a.in.seed + (a.in.seed_dev ? *a.in.seed_dev : 0ull);
  const uint64_t samp_off = (uint64_t)sample * a.h * a.wd * C;
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  float* __restrict__ ys = a.y + (size_t)sample * a.oh * a.ow * C + c0 + q4 * 4;
#pragma nounroll
  for (int tile = tile_lo; tile < tile_hi; ++tile) {
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
    if (tile + 1 < tile_hi) load_patch(tile + 1);
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
    if (stamps && nstamp < 15) stamps[nstamp++] = wall_clock64();
  }
  if (!a.ost.rows) return;
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

// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(T, 2) void mb_resident_fwd_kernel(const ResArgs a) {
  extern __shared__ __attribute__((aligned(16))) float dsm[];   // operand tiles / input patch / scratch: the largest phase
  __shared__ __attribute__((aligned(16))) float tab[2 * KMAX];
  __shared__ float gstat[GMAX][2];
  __shared__ int s_flag;
  const int tid = threadIdx.x, cluster = blockIdx.x & 7, rank = blockIdx.x >> 3, B = a.B;
  if (cluster >= a.n) return;
  unsigned* ctr = a.sync + cluster * 64;
  unsigned* err = a.sync + 512;
  if (tid == 0) {
    s_flag = 0;
    __hip_atomic_fetch_or(ctr + 2, 1u << xcc_id(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  unsigned bars = 0, warm_acc = 0;
  for (int sample = cluster; sample < a.n; sample += 8) {
    for (int p = 0; p < a.nphase; ++p) {
      const ResPhase& ph = a.ph[p];
      const bool stamp = a.stamp0 >= 0 && blockIdx.x == 0 && tid == 0;
      unsigned long long* stamps = stamp ? reinterpret_cast<unsigned long long*>(a.sync + 1024) + 16 * (a.stamp0 + p) : nullptr;
      if (stamp) stamps[0] = wall_clock64();
      if (rank >= ph.nvb && p + 1 < a.nphase && a.ph[p + 1].warm)
        warm_acc ^= warm_l2(a.ph[p + 1].warm, a.ph[p + 1].warm_bytes, rank - ph.nvb, B - ph.nvb, tid);
      switch (ph.kind) {
        case RES_PW64:
          if (ph.u.pw.in.act == RN_ACT_ELU) res_pw_phase<64, 64, 2, 2, RN_ACT_ELU, 1, 1>(ph.u.pw, sample, rank, B, ph.nvb, dsm, tab, gstat, stamps);
          else res_pw_phase<64, 64, 2, 2, RN_ACT_NONE, 1, 1>(ph.u.pw, sample, rank, B, ph.nvb, dsm, tab, gstat, stamps);
          break;
        case RES_PW32:
          if (ph.u.pw.in.act == RN_ACT_ELU) res_pw_phase<32, 32, 1, 1, RN_ACT_ELU, 4, 2>(ph.u.pw, sample, rank, B, ph.nvb, dsm, tab, gstat, stamps);
          else res_pw_phase<32, 32, 1, 1, RN_ACT_NONE, 4, 2>(ph.u.pw, sample, rank, B, ph.nvb, dsm, tab, gstat, stamps);
          break;
        default:
          res_dw_phase<RN_ACT_ELU>(ph.u.dw, sample, rank, ph.nvb, dsm, tab, gstat, stamps);
          break;
      }
      if (stamp) stamps[1] = wall_clock64();
      const bool last = p + 1 == a.nphase;
      if (!last) {                            // (nothing of this launch reads the last phase's output; the next sample's first phase reads none of ours)
        ++bars;
        if (!cluster_barrier(ctr, bars * (unsigned)B, a.spin, err, &s_flag, tid)) return;
        if (stamp) stamps[2] = wall_clock64();
      } else {
        if (stamp) stamps[2] = wall_clock64();
        __syncthreads();                      // this block's LDS before the next sample
      }
    }
  }
  if (warm_acc == 0x9E3779B9u && tid == 77) a.sync[600] = warm_acc;      // (keeps the warm-up loads alive; never true in practice, harmless if it is)
  // the last block of the cluster to leave clears the cluster's words for the next launch and checks the placement
  if (tid == 0) {
    const unsigned old = __hip_atomic_fetch_add(ctr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == (unsigned)B - 1u) {
      const unsigned mask = __hip_atomic_load(ctr + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (mask & (mask - 1u)) __hip_atomic_fetch_or(err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // blocks of one cluster on two XCDs
      __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(ctr + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(ctr + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
int res_blocks() {                  // blocks per cluster (RN_MB_RES_B: tuning aid)
  static const int b = [] {
    const char* e = getenv("RN_MB_RES_B");
    int v = e ? atoi(e) : 64;
    if (v < 8) v = 8;
    if (v > RES_BMAX) v = RES_BMAX;
    return v;
  }();
  return b;
}

struct ResPw { int kind, bm, bn, tiles_n, nvb; };
bool res_pw_plan(int hw, int cin, int cout, ResPw* p) {
  const int B = res_blocks();
  if (hw % 32) return false;
  const long vb64 = hw % 64 == 0 ? (long)(hw / 64) * rn::ceil_div(cout, 64) : 0;
  const int nkt = rn::ceil_div(cin, BK);
  if (vb64 == 0 || (vb64 < B * 3 / 4 && nkt >= 4)) { p->kind = RES_PW32; p->bm = 32; p->bn = 32; }
  else { p->kind = RES_PW64; p->bm = 64; p->bn = 64; }
  p->tiles_n = rn::ceil_div(cout, p->bn);
  p->nvb = hw / p->bm * p->tiles_n;
  return true;
}
// depthwise: 8 x 8 output tiles (4 x 8 behind a stride-2 window: the patch stays within a thread's NP loads), a block owns a
// slab and a run of consecutive tiles; slabs x runs <= B virtual blocks per sample
bool res_dw_plan(int oh, int ow, int c, int stride, int cpg, DwPlan* p) {
  const int B = res_blocks();
  p->sw = dw_slab(c, cpg);
  if (!p->sw) return false;
  p->nslab = c / p->sw;
  if (p->nslab > B) return false;
  int th = stride == 1 ? 8 : 4, tw = 8;
  if (th > oh) th = oh;
  if (tw > ow) tw = ow;
  auto patch = [&]() { return (long)((th - 1) * stride + 3) * ((tw - 1) * stride + 3) * (p->sw / 4); };
  while (patch() > NP * T && th * tw > 1) {
    if (th >= tw) th = (th + 1) / 2; else tw = (tw + 1) / 2;
  }
  if (patch() > NP * T) return false;
  p->th = th; p->tw = tw;
  p->tiles_h = rn::ceil_div(oh, th); p->tiles_w = rn::ceil_div(ow, tw);
  p->ph = (th - 1) * stride + 3; p->pw = (tw - 1) * stride + 3;
  const int ntile = p->tiles_h * p->tiles_w;
  int nblk = B / p->nslab;
  if (nblk > ntile) nblk = ntile;
  p->tpb = rn::ceil_div(ntile, nblk);
  p->nblk = rn::ceil_div(ntile, p->tpb);
  return true;
}

}  // namespace

extern "C" size_t rn_mb_resident_sync_bytes(void) { return 16384; }

extern "C" size_t rn_mb_resident_rows(int kind, int n, int h, int wd, int cin, int cout, int stride, int groups, rn_mb_rows* layout) {
  if (n < 1 || h < 1 || wd < 1 || cin < 4 || cout < 4 || groups < 1 || cout % groups || groups > GMAX) return 0;
  if (kind == RN_MB_PHASE_POINTWISE) {
    ResPw p;
    if (cin % 4 || cout % 4 || cin > KMAX || cout > KMAX || !res_pw_plan(h * wd, cin, cout, &p)) return 0;
    if (cout / groups > p.bn) return 0;
    const int R = h * wd / p.bm, W = groups + p.tiles_n;
    if (R > RMAX) return 0;
    if (layout) { layout->rows_per_sample = R; layout->width = W; layout->bn = p.bn; }
    return (size_t)n * R * W * 8;
  }
  if (kind == RN_MB_PHASE_DEPTHWISE) {
    if (cin != cout || cin % 4 || (stride != 1 && stride != 2)) return 0;
    int oh, ow, pt, pl;
    rn::same_pad(h, 3, stride, &oh, &pt);
    rn::same_pad(wd, 3, stride, &ow, &pl);
    DwPlan p;
    if (!res_dw_plan(oh, ow, cin, stride, cin / groups, &p)) return 0;
    if (dw_lds_bytes(p) > 60 * 1024) return 0;
    if (layout) { layout->rows_per_sample = p.nblk; layout->width = groups; layout->bn = cin; }
    return (size_t)n * p.nblk * groups * 8;
  }
  return 0;
}

extern "C" int rn_mb_resident_fwd(const rn_mb_phase* phases, int nphase, int n, void* sync, rn_stream_t stream) {
  RN_CHECK_ARG(phases && nphase >= 1 && n >= 1 && sync, "mb resident fwd: bad argument");
  const int B = res_blocks();
  static const int spin = getenv("RN_MB_RES_SPIN") ? atoi(getenv("RN_MB_RES_SPIN")) : (1 << 21);
  static int occ = -1;
  hipStream_t st = (hipStream_t)stream;
  // every phase is planned (and refused) before the first launch
  ResPhase* plan = (ResPhase*)calloc((size_t)nphase, sizeof(ResPhase));
  RN_CHECK_ARG(plan, "mb resident fwd: out of host memory");
  size_t lds = 0;
  int rc = RN_OK;
  for (int i = 0; i < nphase && rc == RN_OK; ++i) {
    const rn_mb_phase& s = phases[i];
    ResPhase& d = plan[i];
    const char* what = "mb resident fwd";
    if (!s.in || !s.w || !s.y || s.h < 1 || s.wd < 1) { rn::set_error("%s: phase %d: null pointer / empty map", what, i); rc = RN_EINVAL; break; }
    if (s.in->act != RN_ACT_ELU && s.in->act != RN_ACT_NONE) { rn::set_error("%s: phase %d: activation %d (ELU / none)", what, i, s.in->act); rc = RN_EUNSUPPORTED; break; }
    if ((double)n * s.h * s.wd * (s.cin > s.cout ? s.cin : s.cout) >= 536870912.0) { rn::set_error("%s: phase %d: tensor >= 2 GiB", what, i); rc = RN_EUNSUPPORTED; break; }
    rn_mb_rows want = {};
    if (s.stat_out.rows && !rn_mb_resident_rows(s.kind, n, s.h, s.wd, s.cin, s.cout, s.stride, s.stat_groups, &want)) {
      rn::set_error("%s: phase %d cannot emit rows (see rn_mb_resident_rows)", what, i); rc = RN_EUNSUPPORTED; break;
    }
    if (s.stat_out.rows && (want.rows_per_sample != s.stat_out.rows_per_sample || want.width != s.stat_out.width || want.bn != s.stat_out.bn)) {
      rn::set_error("%s: phase %d: stat_out layout differs from rn_mb_resident_rows", what, i); rc = RN_EINVAL; break;
    }
    if (s.kind == RN_MB_PHASE_POINTWISE) {
      PwFwdArgs& a = d.u.pw;
      ResPw p;
      if (s.cin % 4 || s.cout % 4 || s.cin > KMAX || s.cout > KMAX || !res_pw_plan(s.h * s.wd, s.cin, s.cout, &p)) {
        rn::set_error("%s: phase %d: pointwise %d -> %d on %d pixels is not supported", what, i, s.cin, s.cout, s.h * s.wd); rc = RN_EUNSUPPORTED; break;
      }
      if ((rc = fill_norm(s.in, &a.in, n, true, what)) != RN_OK) break;
      if (s.in->c != s.cin) { rn::set_error("%s: phase %d: in->c %d != cin %d", what, i, s.in->c, s.cin); rc = RN_EINVAL; break; }
      if (s.in->act != RN_ACT_NONE && (s.residual || s.materialise)) { rn::set_error("%s: phase %d: residual / materialise come with a linear block", what, i); rc = RN_EINVAL; break; }
      a.res = s.residual; a.mat = s.materialise; a.w = s.w; a.y = s.y;
      a.n = n; a.hw = s.h * s.wd; a.cin = s.cin; a.cout = s.cout; a.tiles_n = p.tiles_n;
      if (s.stat_out.rows) {
        a.ost.rows = (float2*)s.stat_out.rows; a.ost.R = want.rows_per_sample; a.ost.W = want.width; a.ost.bn = want.bn;
        a.ocpg = s.cout / s.stat_groups;
      }
      d.kind = p.kind; d.nvb = p.nvb;
      d.warm = s.w; d.warm_bytes = (unsigned)s.cin * s.cout * 4u;
      const size_t need = (size_t)(p.kind == RES_PW64 ? PW64_FLOATS : PW32_FLOATS) * 4;
      if (need > lds) lds = need;
    } else if (s.kind == RN_MB_PHASE_DEPTHWISE) {
      DwFwdArgs& a = d.u.dw;
      if (s.cin != s.cout || (s.stride != 1 && s.stride != 2)) { rn::set_error("%s: phase %d: depthwise needs cin == cout, stride 1 / 2", what, i); rc = RN_EINVAL; break; }
      if (s.in->act != RN_ACT_ELU) { rn::set_error("%s: phase %d: the depthwise phase is built for ELU", what, i); rc = RN_EUNSUPPORTED; break; }
      if ((rc = fill_norm(s.in, &a.in, n, true, what)) != RN_OK) break;
      if (s.in->c != s.cin) { rn::set_error("%s: phase %d: in->c %d != channels %d", what, i, s.in->c, s.cin); rc = RN_EINVAL; break; }
      if (s.stat_out.rows && s.stat_groups != s.in->groups) { rn::set_error("%s: phase %d: the GroupNorms around a depthwise conv share their grouping", what, i); rc = RN_EUNSUPPORTED; break; }
      a.w = s.w; a.y = s.y; a.n = n; a.h = s.h; a.wd = s.wd; a.c = s.cin; a.stride = s.stride;
      rn::same_pad(s.h, 3, s.stride, &a.oh, &a.pad_t);
      rn::same_pad(s.wd, 3, s.stride, &a.ow, &a.pad_l);
      DwPlan p;
      if (!res_dw_plan(a.oh, a.ow, s.cin, s.stride, a.in.cpg, &p) || dw_lds_bytes(p) > 60 * 1024) {
        rn::set_error("%s: phase %d: no depthwise plan for c=%d", what, i, s.cin); rc = RN_EUNSUPPORTED; break;
      }
      a.th = p.th; a.tw = p.tw; a.tiles_h = p.tiles_h; a.tiles_w = p.tiles_w; a.sw = p.sw; a.nslab = p.nslab; a.ph = p.ph; a.pw = p.pw;
      a.tpb = p.tpb; a.nblk = p.nblk;
      if (s.stat_out.rows) {
        a.ost.rows = (float2*)s.stat_out.rows; a.ost.R = want.rows_per_sample; a.ost.W = want.width; a.ost.bn = want.bn;
        a.ocpg = s.cin / s.stat_groups;
      }
      d.kind = RES_DW; d.nvb = p.nslab * p.nblk;
      d.warm = s.w; d.warm_bytes = 9u * (unsigned)s.cin * 4u;
      const size_t need = dw_lds_bytes(p);
      if (need > lds) lds = need;
    } else {
      rn::set_error("%s: phase %d: kind %d", what, i, s.kind); rc = RN_EINVAL;
    }
  }
  if (rc != RN_OK) { free(plan); return rc; }
  // co-residency: the cluster's B blocks must fit the 32 CUs of an XCD at this kernel's footprint
  if (occ < 0) {
    int o = 0;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(mb_resident_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, mb_resident_fwd_kernel, T, 60 * 1024) != hipSuccess) {
      (void)hipGetLastError();
      o = 0;
    }
    occ = o;
  }
  if (occ < 1 || B > 32 * (occ < 2 ? occ : 2)) {
    free(plan);
    rn::set_error("mb resident fwd: %d blocks per cluster do not fit an XCD (%d blocks per CU)", B, occ);
    return RN_EUNSUPPORTED;
  }
  for (int p0 = 0; p0 < nphase; p0 += RES_MAXP) {
    ResArgs a = {};
    a.sync = (unsigned*)sync; a.n = n; a.B = B; a.spin = spin;
    static const bool stamps = getenv("RN_MB_RES_STAMPS") && atoi(getenv("RN_MB_RES_STAMPS"));
    a.stamp0 = stamps ? p0 : -1;
    a.nphase = nphase - p0 < RES_MAXP ? nphase - p0 : RES_MAXP;
    memcpy(a.ph, plan + p0, (size_t)a.nphase * sizeof(ResPhase));
    hipLaunchKernelGGL(mb_resident_fwd_kernel, dim3(8 * B), dim3(T), lds, st, a);
  }
  free(plan);
  RN_LAUNCH_CHECK();
  return RN_OK;
}
