// Shared device / host helpers of the MobileNetV2 bottleneck-chain kernels (mbconv.hip: the launch-ordered kernels;
// mb_resident.hip: the XCD-resident small-map section): statistic rows, per-channel tables, the counter-based dropout mask,
// argument structs of the forward kernels, tile / slab planning.  Not part of the public ABI.  Everything lives in an
// anonymous namespace: each translation unit gets its own copy, the kernels inline what they use.
#pragma once
#include <stdlib.h>
#include <string.h>

#include "conv_tiles.h"
#include "rn_common.h"

namespace {
using namespace rn_tiles;

constexpr int T = 256;
constexpr int KMAX = 1024;     // widest channel count (tables of per-channel coefficients live in LDS)
constexpr int GMAX = 32;       // GroupNorm groups (normalization.py:24: min(32, c))
constexpr int RMAX = 256;      // rows of a sample a consumer block is willing to merge

struct RowsDev { float2* rows; int R, W, bn; };
struct NormDev {
  const float* y; RowsDev st; float* mean; float* rstd; const float* gamma; const float* beta;
  int c, groups, cpg, act;
  float eps, drop_rate, keep_scale;
  uint64_t seed; const uint64_t* seed_dev;
};

// ---------------------------------------------------------------------------------------------------------------------
// Prologue helpers.  A block may have more than T threads (intra-block split-K: KS groups of T); the first T ("leaders")
// do the work, every thread of the block calls (the helpers contain block-wide barriers).  Everything a prologue needs
// from global memory that does not depend on another load -- gamma / beta of its channels, mean / rstd when they are read
// back -- is fetched into registers FIRST (prefetch_*), together with the kernel's first operand loads: the merge of the
// rows, the tables and the first tile then cost one round trip to memory, not one each.
template <int NJ>
struct ChanPre { float g[NJ], b[NJ]; };             // gamma / beta of the channels c0 + tid + j T
template <int NJ>
__device__ __forceinline__ void prefetch_chan(const float* gamma, const float* beta, int c0, int nc, int tid, ChanPre<NJ>& p) {
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int i = min(tid % T + j * T, nc - 1);
    p.g[j] = (tid < T) ? gamma[c0 + i] : 0.f;
    p.b[j] = (tid < T && beta) ? beta[c0 + i] : 0.f;
  }
}
struct GroupPre { float mean, rstd; };             // read-back statistics of group g0 + tid (tid < ng)
__device__ __forceinline__ GroupPre prefetch_groups(const NormDev& nd, int sample, int g0, int ng, int tid) {
  GroupPre p = {0.f, 1.f};
  if (nd.st.rows == nullptr && tid < T) {
    const int g = g0 + min(tid, ng - 1);
    p.mean = nd.mean[sample * nd.groups + g];
    p.rstd = nd.rstd[sample * nd.groups + g];
  }
  return p;
}

// Totals (a, b) of the groups [g0, g0 + ng) over the rows of `sample` -> tot[ng][2] (fp64, fixed order); ng <= T.  A group's
// entries of a row: position g + t for the producer N-tiles t that cut it (<= 2, the host checks cpg <= bn): both are loaded
// unconditionally (the second weighted 0 when there is none), eight rows in flight per lane.
__device__ __forceinline__ void merge_rows(const RowsDev& st, int sample, int cpg, int C, int g0, int ng, double (*part)[2],
                                           double (*tot)[2], int tid) {
  const int RL = T / ng, gl = tid % ng, rl = tid / ng;
  double S = 0.0, Q = 0.0;
  if (tid < T && rl < RL) {
    const int g = g0 + gl;
    const int t0 = (g * cpg) / st.bn, t1 = (min((g + 1) * cpg, C) - 1) / st.bn;
    const double w1 = t1 > t0 ? 1.0 : 0.0;
    const float2* __restrict__ base = st.rows + (size_t)sample * st.R * st.W + g;
    for (int r0 = rl; r0 < st.R; r0 += 8 * RL) {
      float2 a[8], b[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const size_t rr = (size_t)min(r0 + j * RL, st.R - 1) * st.W;
        a[j] = base[rr + t0];
        b[j] = base[rr + t1];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (r0 + j * RL < st.R) {
          S += (double)a[j].x + w1 * (double)b[j].x;
          Q += (double)a[j].y + w1 * (double)b[j].y;
        }
    }
  }
  if (tid < T) { part[tid][0] = S; part[tid][1] = Q; }
  __syncthreads();
  if (tid < ng) {
    S = 0.0; Q = 0.0;
    for (int l = 0; l < RL; ++l) { S += part[l * ng + tid][0]; Q += part[l * ng + tid][1]; }
    tot[tid][0] = S; tot[tid][1] = Q;
  }
  __syncthreads();
}

// (mean, rstd) of the groups [g0, g0 + ng) of `sample` -> gstat[ng][2]: merged from the rows (and written to nd.mean /
// nd.rstd when `publish`), or the prefetched read-back values (nd.st.rows == nullptr: the backward kernels).
// `scratch` >= (T + GMAX) * 16 bytes.
__device__ __forceinline__ void group_stats(const NormDev& nd, int sample, int hw, int g0, int ng, bool publish, void* scratch,
                                            float (*gstat)[2], const GroupPre& pre, int tid) {
  if (nd.st.rows) {
    double (*part)[2] = reinterpret_cast<double (*)[2]>(scratch);
    double (*tot)[2] = part + T;
    merge_rows(nd.st, sample, nd.cpg, nd.c, g0, ng, part, tot, tid);
    if (tid < ng) {
      const double m = (double)hw * (double)nd.cpg;
      const double mean = tot[tid][0] / m;
      double var = tot[tid][1] / m - mean * mean;
      if (var < 0.0) var = 0.0;
      const float rstd = (float)(1.0 / sqrt(var + (double)nd.eps));
      gstat[tid][0] = (float)mean; gstat[tid][1] = rstd;
      if (publish) { nd.mean[sample * nd.groups + g0 + tid] = (float)mean; nd.rstd[sample * nd.groups + g0 + tid] = rstd; }
    }
  } else if (tid < ng) {
    gstat[tid][0] = pre.mean; gstat[tid][1] = pre.rstd;
  }
  __syncthreads();
}

// per-channel z = x * sc + sh  (sc = rstd gamma, sh = beta - mean sc) of the channels [c0, c0 + nc) -> sc[0..nc) | sh[0..nc)
template <int NJ>
__device__ __forceinline__ void scale_shift_table(const NormDev& nd, int c0, int nc, int g0, const float (*gstat)[2], float* sc, float* sh,
                                                  const ChanPre<NJ>& p, int tid) {
  if (tid < T) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int i = tid + j * T;
      if (i < nc) {
        const int g = (c0 + i) / nd.cpg - g0;
        const float s = gstat[g][1] * p.g[j];
        sc[i] = s;
        sh[i] = p.b[j] - gstat[g][0] * s;
      }
    }
  }
  __syncthreads();
}

template <int ACT>
__device__ __forceinline__ float act_of(float z, int act_rt) { return ACT >= 0 ? rn::act_fwd(z, ACT) : rn::act_fwd(z, act_rt); }
template <int ACT>
__device__ __forceinline__ float actgrad_of(float z, int act_rt) { return ACT >= 0 ? rn::act_grad(z, ACT) : rn::act_grad(z, act_rt); }

// rn::uniform01(seed, idx .. idx + 3) >= rate for four consecutive element indices, the same bits as the per-element
// function: every tensor here has < 2^32 elements (host-checked: < 2 GiB), so the index's high word contributes only the
// seed's high word; the first multiply is shared (idx + j) * C = idx * C + j * C.  (The two avalanche multiplies per element
// stay: v_mul_lo_u32 is quarter rate, this is what an element's mask costs.)
__device__ __forceinline__ void keep4(uint64_t seed, uint64_t eidx, float rate, float keep, float (&m)[4]) {
  const uint32_t s_lo = (uint32_t)seed, s_hi = (uint32_t)(seed >> 32);
  uint32_t h0 = (uint32_t)eidx * 0x9E3779B1u + s_lo;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    uint32_t h = h0 ^ s_hi;
    h0 += 0x9E3779B1u;
    h ^= h >> 16; h *= 0x85EBCA6Bu;
    h ^= h >> 13; h *= 0xC2B2AE35u;
    h ^= h >> 16;
    m[j] = ((float)(h >> 8) * (1.0f / 16777216.0f) >= rate) ? keep : 0.f;
  }
}
__device__ __forceinline__ float keep1(uint64_t seed, uint64_t eidx, float rate, float keep) {
  uint32_t h = ((uint32_t)eidx * 0x9E3779B1u + (uint32_t)seed) ^ (uint32_t)(seed >> 32);
  h ^= h >> 16; h *= 0x85EBCA6Bu;
  h ^= h >> 13; h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return ((float)(h >> 8) * (1.0f / 16777216.0f) >= rate) ? keep : 0.f;
}

// drop(act(z)) of 4 consecutive channels of one pixel; `eidx` = element index of the first one in the GroupNorm's tensor
template <int ACT>
__device__ __forceinline__ float4 norm_act_drop(float4 v, float4 sc, float4 sh, int act_rt, bool drop, float rate, float keep,
                                                uint64_t seed, uint64_t eidx) {
  float o[4] = {fmaf(v.x, sc.x, sh.x), fmaf(v.y, sc.y, sh.y), fmaf(v.z, sc.z, sh.z), fmaf(v.w, sc.w, sh.w)};
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = act_of<ACT>(o[j], act_rt);
  if (drop) {
    float m[4];
    keep4(seed, eidx, rate, keep, m);
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] *= m[j];
  }
  return make_float4(o[0], o[1], o[2], o[3]);
}

// ---------------------------------------------------------------------------------------------------------------------
// Per-group rows from per-lane column sums of an accumulator tile.  A lane holds 16 rows x TN columns of each 32-row slab:
// the caller has summed its (v1, v2) over those rows into s1[tn], s2[tn]; here: one cross-half shuffle, the WM waves
// through LDS, then the channels of every group the N-tile touches, in channel order -> row[g + tile_n].  Optionally the
// per-channel sums go to plane1 / plane2 (the parameter-gradient planes) and the group sums are weighted by wgt[c]
// (gamma).  `smem`: dead operand tiles, >= (WM + 1) * BN * 2 floats.  Every thread of the block calls; `tid` < T works.
template <int BM, int BN, int WM, int WN>
__device__ __forceinline__ void reduce_group_rows(const float (&s1)[BN / WN / 32], const float (&s2)[BN / WN / 32], float* smem, float2* row,
                                                  int n0, int C, int cpg, int tile_n, int tid, const float* wgt, float* plane1, float* plane2) {
  constexpr int TN = BN / WN / 32;
  const bool on = tid < WM * WN * 64;                 // group 0
  const int lane = tid & 63, wave = (tid >> 6) % (WM * WN), wm = wave / WN, wn = wave % WN, l31 = lane & 31;
  float* red = smem;                 // [WM][BN][2]
  float* chan = smem + WM * BN * 2;  // [BN][2]
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    float a = s1[tn], b = s2[tn];
    a += __shfl_xor(a, 32, 64);
    b += __shfl_xor(b, 32, 64);
    if (on && lane < 32) {
      const int col = wn * (BN / WN) + tn * 32 + l31;
      red[(wm * BN + col) * 2 + 0] = a;
      red[(wm * BN + col) * 2 + 1] = b;
    }
  }
  __syncthreads();
  if (tid < BN) {
    float t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int w = 0; w < WM; ++w) { t1 += red[(w * BN + tid) * 2 + 0]; t2 += red[(w * BN + tid) * 2 + 1]; }
    chan[tid * 2 + 0] = t1; chan[tid * 2 + 1] = t2;
    if (plane1 && n0 + tid < C) { plane1[n0 + tid] = t1; plane2[n0 + tid] = t2; }
  }
  __syncthreads();
  const int cend = min(n0 + BN, C);
  const int g_lo = n0 / cpg, g_hi = (cend - 1) / cpg;
  if (tid <= g_hi - g_lo) {
    const int g = g_lo + tid;
    const int c_lo = max(g * cpg, n0), c_hi = min((g + 1) * cpg, cend);
    float t1 = 0.f, t2 = 0.f;
    for (int c = c_lo; c < c_hi; ++c) {
      const float w = wgt ? wgt[c] : 1.f;
      t1 += w * chan[(c - n0) * 2 + 0]; t2 += w * chan[(c - n0) * 2 + 1];
    }
    row[g + tile_n] = make_float2(t1, t2);
  }
}

// Intra-block split-K: the accumulators of the groups 1 .. KS-1 are added to group 0's, in group order, through LDS
// (`red` >= (KS - 1) * NACC * T floats; the operand tiles are dead).  Afterwards group 0 holds the tile.
template <int KS, int NACC, int TG = T>
__device__ __forceinline__ void sum_groups(f32x16* acc, float* red, int grp, int lt) {
  if (KS == 1) return;
  if (grp > 0) {
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) red[((size_t)((grp - 1) * NACC + i) * 16 + r) * TG + lt] = acc[i][r];
  }
  __syncthreads();
  if (grp == 0) {
    for (int g = 1; g < KS; ++g)
#pragma unroll
      for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] += red[((size_t)((g - 1) * NACC + i) * 16 + r) * TG + lt];
  }
  __syncthreads();
}

// ---- argument structs of the forward kernels
struct PwFwdArgs {
  int dbg;   // tuning aid (RN_MB_DBG): stop after a phase
  const float* x; const float* res; float* mat; const float* w; float* y;
  NormDev in;
  int n, hw, cin, cout, tiles_n;
  RowsDev ost; int ocpg;
};

struct DwFwdArgs {
  int dbg;
  NormDev in; const float* w; float* y;
  int n, h, wd, c, stride, oh, ow, pad_t, pad_l;
  int th, tw, tiles_h, tiles_w, sw, nslab, ph, pw;
  int tpb, nblk;     // tiles per block, blocks per (sample, slab)
  RowsDev ost; int ocpg;
};
constexpr int NP = 8;            // patch loads in flight per thread (the host keeps a patch at <= NP * T float4)

// one patch element of thread `tid`: its LDS slot idx = tid + j T = pp * SQ + q, pixel, validity
struct PatchElem { int q, pix; bool inside, live; };
__device__ __forceinline__ PatchElem patch_elem(int idx, int total, int SQ, int pw, int y0, int x0, int h, int w) {
  PatchElem e;
  e.live = idx < total;
  const int i = min(idx, total - 1);
  const int pp = i / SQ;
  e.q = i - pp * SQ;
  const int py = pp / pw, px = pp - py * pw;
  const int yy = y0 + py, xx = x0 + px;
  e.inside = (unsigned)yy < (unsigned)h && (unsigned)xx < (unsigned)w;
  e.pix = min(max(yy, 0), h - 1) * w + min(max(xx, 0), w - 1);
  return e;
}
// the tile-independent half of patch_elem, packed into one register per slot (the tile loops keep NP of them alive instead of
// letting the compiler hoist three or four values per slot): q | py << 8 | px << 16 | live << 24
__device__ __forceinline__ int patch_pack(int idx, int total, int SQ, int pw) {
  const int i = min(idx, total - 1);
  const int pp = i / SQ, q = i - pp * SQ;
  const int py = pp / pw, px = pp - py * pw;
  return q | (py << 8) | (px << 16) | (idx < total ? (1 << 24) : 0);
}
__device__ __forceinline__ PatchElem patch_at(int pk, int y0, int x0, int h, int w) {
  PatchElem e;
  e.live = (pk >> 24) != 0;
  e.q = pk & 255;
  const int yy = y0 + ((pk >> 8) & 255), xx = x0 + ((pk >> 16) & 255);
  e.inside = (unsigned)yy < (unsigned)h && (unsigned)xx < (unsigned)w;
  e.pix = min(max(yy, 0), h - 1) * w + min(max(xx, 0), w - 1);
  return e;
}

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
int dbg_word(const char* kernel) {      // RN_MB_DBG="<kernel>:<phase>" (pwf, dwf, pwb, dwb): stop that kernel after a phase (timing aid; results invalid)
  const char* e = getenv("RN_MB_DBG");
  if (!e || strncmp(e, kernel, 3) != 0 || e[3] != ':') return 0;
  return atoi(e + 4);
}
int fill_norm(const rn_mb_norm* s, NormDev* d, int n, bool need_rows, const char* what) {
  RN_CHECK_ARG(s && s->y && s->mean && s->rstd && s->gamma && s->beta, "%s: null pointer in rn_mb_norm", what);
  RN_CHECK_ARG(s->c >= 4 && s->groups >= 1 && s->c % s->groups == 0, "%s: c=%d groups=%d", what, s->c, s->groups);
  RN_UNSUPPORTED(s->c % 4 != 0 || s->c > KMAX || s->groups > GMAX, "%s: c=%d (multiple of 4, <= %d), groups=%d (<= %d)", what, s->c, KMAX,
                 s->groups, GMAX);
  RN_CHECK_ARG(s->drop_rate >= 0.f && s->drop_rate < 1.f, "%s: drop_rate %f", what, s->drop_rate);
  d->y = s->y; d->mean = s->mean; d->rstd = s->rstd; d->gamma = s->gamma; d->beta = s->beta;
  d->c = s->c; d->groups = s->groups; d->cpg = s->c / s->groups; d->act = s->act;
  d->eps = s->eps; d->drop_rate = s->drop_rate; d->keep_scale = s->drop_rate > 0.f ? 1.f / (1.f - s->drop_rate) : 1.f;
  d->seed = s->drop_seed; d->seed_dev = s->drop_seed_dev;
  d->st.rows = (float2*)s->stat.rows; d->st.R = s->stat.rows_per_sample; d->st.W = s->stat.width; d->st.bn = s->stat.bn;
  if (need_rows) RN_CHECK_ARG(s->stat.rows, "%s: the forward pass needs the statistic rows", what);
  if (s->stat.rows) {
    RN_CHECK_ARG(d->st.R >= 1 && d->st.W >= d->groups && d->st.bn >= 1, "%s: bad row layout", what);
    RN_UNSUPPORTED(d->st.R > RMAX || d->cpg > d->st.bn, "%s: %d rows per sample (<= %d) / groups wider than an N-tile", what, d->st.R, RMAX);
  }
  (void)n;
  return RN_OK;
}

// tile shape of the pointwise forward: 0 = 64x64, 1 = 128x32, 2 = 128x64, 3 = 32x32 (one wave per split-K group); ks = groups
// of waves that share a block's K-tiles.  A function of the shape alone: the row layout (rn_mb_pointwise_rows) follows it.
struct PwCfg { int id, bm, bn, ks; };
PwCfg pw_cfg(int n, int hw, int kdim, int ncols) {
  static const bool no_ks = getenv("RN_MB_NO_SPLITK") != nullptr;
  static const bool no_t32 = getenv("RN_MB_NO_TILE32") != nullptr;
  if (const char* f = getenv("RN_MB_PW_CFG")) {  // tuning aid
    const int c = atoi(f);
    if (c == 1 && hw % 128 == 0) return {1, 128, 32, 1};
    if (c == 2 && hw % 128 == 0) return {2, 128, 64, 1};
    if (c == 0) return {0, 64, 64, 1};
  }
  const long m = (long)n * hw;
  if (hw % 128 == 0 && ncols <= 32 && m / 128 >= 128) return {1, 128, 32, 1};
  if (hw % 128 == 0 && hw > 4096) return {2, 128, 64, 1};       // large maps: fewer rows for the consumers to merge
  // few output tiles with a long K (the small maps): groups of waves share the K-tiles, 32 x 32 tiles give more blocks
  const long blocks64 = m / 64 * rn::ceil_div(ncols, 64);
  const int nkt = rn::ceil_div(kdim, BK);
  if (blocks64 <= 96 && nkt >= 4 && !no_ks) {
    if (!no_t32) return {3, 32, 32, nkt >= 16 ? 8 : 4};
    return {0, 64, 64, 4};
  }
  return {0, 64, 64, 1};
}

// channel slab of the depthwise kernels: whole groups (of both GroupNorms around it: same channel count, same rule) and
// whole float4 quads, the narrowest one of >= 32 channels that tiles c (or all of c), at most 128
int dw_slab(int c, int cpg) {
  int unit = cpg;
  while (unit % 4) unit += cpg;
  for (int sw = unit; sw <= c; sw += unit)
    if (c % sw == 0 && sw >= 32) return sw <= 128 ? sw : 0;
  return c <= 128 ? c : 0;
}

struct DwPlan { int th, tw, tiles_h, tiles_w, sw, nslab, ph, pw, tpb, nblk; };
// tiles per block of the depthwise kernels: about two blocks per CU (RN_MB_DW_TPB overrides: measurements)
int dw_tiles_per_block(long blocks, int ntile) {
  static const int forced = getenv("RN_MB_DW_TPB") ? atoi(getenv("RN_MB_DW_TPB")) : 0;
  static const int target = getenv("RN_MB_DW_BLOCKS") ? atoi(getenv("RN_MB_DW_BLOCKS")) : 512;    // blocks per launch (tuning aid)
  int tpb = forced > 0 ? forced : (int)(blocks / (target > 0 ? target : 512));
  if (tpb < 1) tpb = 1;
  if (tpb > ntile) tpb = ntile;
  return tpb;
}
bool dw_plan(int n, int oh, int ow, int c, int stride, int cpg, DwPlan* p) {
  p->sw = dw_slab(c, cpg);
  if (!p->sw) return false;
  p->nslab = c / p->sw;
  // (tuning aids: the tile the planner starts from on the large maps -- a taller tile halves the halo and the round trips per pixel)
  static const int th_big = getenv("RN_MB_DWF_TH") ? atoi(getenv("RN_MB_DWF_TH")) : 8;
  static const int tw_big = getenv("RN_MB_DWF_TW") ? atoi(getenv("RN_MB_DWF_TW")) : 8;
  const bool big = (long)oh * ow >= 16384;
  int th = stride == 1 ? (big ? th_big : 8) : (big ? th_big / 2 : 4), tw = big ? tw_big : 8;
  if (th > oh) th = oh;
  if (tw > ow) tw = ow;
  auto blocks = [&]() { return (long)n * rn::ceil_div(oh, th) * rn::ceil_div(ow, tw) * p->nslab; };
  auto patch = [&]() { return (long)((th - 1) * stride + 3) * ((tw - 1) * stride + 3) * (p->sw / 4); };
  while ((blocks() < 384 && th * tw > 16) || (patch() > NP * T && th * tw > 1)) {   // (a thread holds <= NP patch loads)
    if (th >= tw) th = (th + 1) / 2; else tw = (tw + 1) / 2;
  }
  if (patch() > NP * T) return false;
  p->th = th; p->tw = tw;
  p->tiles_h = rn::ceil_div(oh, th); p->tiles_w = rn::ceil_div(ow, tw);
  p->ph = (th - 1) * stride + 3; p->pw = (tw - 1) * stride + 3;
  p->tpb = dw_tiles_per_block(blocks(), p->tiles_h * p->tiles_w);
  p->nblk = rn::ceil_div(p->tiles_h * p->tiles_w, p->tpb);
  return true;
}
size_t dw_lds_bytes(const DwPlan& p) {
  size_t patch = (size_t)p.ph * p.pw * p.sw * 4;
  size_t scratch = (size_t)(T + GMAX) * 16;
  size_t red = (size_t)(T * 8 + 2 * p.sw) * 4;
  size_t m = patch > scratch ? patch : scratch;
  return m > red ? m : red;
}

}  // namespace
