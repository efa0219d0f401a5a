// The middle of a MobileNetV2 inverted-residual bottleneck (mobilenet_v2.py:56-80) as ONE kernel per direction:
//
//     a = drop2(act(GN2( depthwise3x3( drop1(act(GN1(x))) ) )))
//
// instead of GroupNorm kernel + depthwise kernel + GroupNorm kernel (three launch-latency-bound launches each way, a chain of
// ~25 us for ~5 us of work at 1/16 resolution).  A depthwise conv keeps channels apart and both GroupNorms use the same
// grouping, so ONE block can own one (sample, group) slice end to end: it holds the normalised, activated input slice in
// LDS (all pixels of the group's C/groups channels: 24..98 KB), runs the 3x3 stencil out of LDS, and gets BOTH sets of
// statistics exactly (two-pass mean / variance over values it already holds) without any exchange between blocks.  The
// raw depthwise output and the activated input are never written: the backward kernel recomputes them from x.
// Backward, same ownership: g2 = dy*mask2*act'(z2) -> GroupNorm-2 gradient -> depthwise data + weight gradient ->
// g1 -> GroupNorm-1 gradient, with per-sample rows for the parameter gradients (summed by rn_reduce_rows).
// Dropout masks are the same counter-based hash of (seed, element index) as the stand-alone GroupNorm kernels.
#include "rn_common.h"

namespace {
constexpr int T = 1024;
constexpr int U = 8;   // global loads a thread keeps in flight in the slice loops
constexpr size_t LDS_LIMIT = 156 * 1024;   // dynamic LDS of the CU's 160 KB (the rest: ~3 KB of static reduction scratch)

struct Args {
  const float* x; const float* dy;
  const float* gamma1; const float* beta1; const float* wgt; const float* gamma2; const float* beta2;
  float* y; float* stats;        // stats [4][n][groups]: mean1, rstd1, mean2, rstd2
  float* dx; float* rows;        // backward: dx, rows = [dg1: n x C][db1: n x C][dg2: n x C][db2: n x C][dw: n x 9C]
  int n, h, w, c, stride, groups, cpg, act, oh, ow, pad_t, pad_l;
  float eps, drop_rate;
  uint64_t seed1, seed2;
  const uint64_t* seed_dev;
};

// block-wide sum of NV values per thread; result broadcast to every thread (fixed order: waves in order)
template <int NV>
__device__ __forceinline__ void block_sum(float (&v)[NV], float* sh /* [NV][T/64] */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = rn::wave_sum(v[i]);
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) sh[i * (T / 64) + wave] = v[i];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    float t = 0.f;
#pragma unroll
    for (int wv = 0; wv < T / 64; ++wv) t += sh[i * (T / 64) + wv];
    v[i] = t;
  }
}

// per-channel sum of one per-thread value over the block's pixel lanes -> out[ch]  (thread t: channel t % cpg); fixed order
__device__ __forceinline__ void channel_sum(float v, int cpg, int plc, bool active, float* red /* [T] */, float* red2 /* [8*64] */,
                                            float* out /* [64] */) {
  const int tid = threadIdx.x;
  __syncthreads();
  red[tid] = active ? v : 0.f;
  __syncthreads();
  if (tid < cpg * 8) {
    const int ch = tid % cpg, j = tid / cpg;
    float acc = 0.f;
    for (int pl = j; pl < plc; pl += 8) acc += red[pl * cpg + ch];
    red2[j * 64 + ch] = acc;
  }
  __syncthreads();
  if (tid < cpg) {
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) t += red2[j * 64 + tid];
    out[tid] = t;
  }
  __syncthreads();
}

// floor(i / d) for 0 <= i < 2^20 and 1 <= d <= 2^12 without an integer division (a runtime-divisor division is ~40
// instructions, and these loops did five per element): (i + 0.5) * (1 / d) is off by < 1e-6 * i / d, far inside the
// 0.5 / d margin to the next integer
__device__ __forceinline__ int fast_div(int i, float inv_d) { return (int)(((float)i + 0.5f) * inv_d); }

__device__ __forceinline__ float drop_apply(float v, float rate, float keep_scale, uint64_t seed, uint64_t idx) {
  return (rn::uniform01(seed, idx) >= rate) ? v * keep_scale : 0.f;
}

// the 3x3 stencil at output pixel (oy, ox) for channel lane ch, out of the LDS slice a1[p * cpg + ch]; same fmaf order as
// dw_fwd_kernel (kh, kw ascending; clamped address times a 0/1 mask)
__device__ __forceinline__ float stencil(const Args& a, const float* a1, const float (&wt)[9], int oy, int ox, int ch) {
  float acc = 0.f;
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    const int ih = oy * a.stride - a.pad_t + kh;
    const int ihc = min(max(ih, 0), a.h - 1);
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int iw = ox * a.stride - a.pad_l + kw;
      const int iwc = min(max(iw, 0), a.w - 1);
      const float m = ((unsigned)ih < (unsigned)a.h && (unsigned)iw < (unsigned)a.w) ? 1.f : 0.f;
      acc = fmaf(a1[(ihc * a.w + iwc) * a.cpg + ch] * m, wt[kh * 3 + kw], acc);
    }
  }
  return acc;
}

// x slice -> LDS as a1 = drop1(act(GN1(x))); returns mean1 / rstd1 (computed when stats_known == false).
// Thread t owns channel lane t % cpg and the pixels t / cpg + k * (T / cpg): no index arithmetic per element.
template <int ACT>
__device__ __forceinline__ void load_a1(const Args& a, float* a1, int n_, int g, bool stats_known, float* mean_io, float* rstd_io,
                                        float* sh) {
  const int tid = threadIdx.x, cpg = a.cpg, hw = a.h * a.w, C = a.c, cnt = hw * cpg;
  const int plc = T / cpg, ch = tid % cpg, pl = tid / cpg, cg = g * cpg + ch;
  const int p0 = pl < plc ? pl : hw;           // idle lanes (T is not a multiple of cpg) skip every loop
  const float* xg = a.x + (size_t)n_ * hw * C + cg;
  float mean = *mean_io, rstd = *rstd_io;
  if (!stats_known) {
    float s[1] = {0.f};
    for (int p = p0; p < hw; p += U * plc) {   // U loads in flight per thread (a one-load-per-iteration loop waits for each)
      float v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = xg[(size_t)min(p + u * plc, hw - 1) * C];
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (p + u * plc < hw) { a1[(p + u * plc) * cpg + ch] = v[u]; s[0] += v[u]; }
    }
    block_sum<1>(s, sh);
    mean = s[0] / (float)cnt;
    float q[1] = {0.f};
    for (int p = p0; p < hw; p += plc) { const float d = a1[p * cpg + ch] - mean; q[0] += d * d; }
    block_sum<1>(q, sh);
    rstd = 1.f / sqrtf(q[0] / (float)cnt + a.eps);
    *mean_io = mean; *rstd_io = rstd;
  } else {
    for (int p = p0; p < hw; p += U * plc) {
      float v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = xg[(size_t)min(p + u * plc, hw - 1) * C];
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (p + u * plc < hw) a1[(p + u * plc) * cpg + ch] = v[u];
    }
  }
  const bool drop = a.drop_rate > 0.f;
  const float keep_scale = drop ? 1.f / (1.f - a.drop_rate) : 1.f;
  const uint64_t seed = a.seed1 + (a.seed_dev ? *a.seed_dev : 0ull);
  const uint64_t samp = (uint64_t)n_ * (uint64_t)hw * (uint64_t)C + (uint64_t)cg;
  const float sc = rstd * a.gamma1[cg], shf = a.beta1[cg] - mean * sc;
  for (int p = p0; p < hw; p += plc) {          // own elements only: no barrier needed before this loop
    float v = rn::act_fwd(a1[p * cpg + ch] * sc + shf, ACT);
    if (drop) v = drop_apply(v, a.drop_rate, keep_scale, seed, samp + (uint64_t)p * C);
    a1[p * cpg + ch] = v;
  }
  __syncthreads();
}

template <int ACT, int R>
__global__ __launch_bounds__(T) void dwgn_fwd_kernel(const Args a) {
  extern __shared__ float smem[];
  __shared__ float sh[2 * (T / 64)];
  float* a1 = smem;                       // [hw][cpg]
  const int tid = threadIdx.x, cpg = a.cpg, C = a.c;
  const int n_ = blockIdx.x / a.groups, g = blockIdx.x - n_ * a.groups;
  const int ohw = a.oh * a.ow, NG = a.n * a.groups;
  float mean1 = 0.f, rstd1 = 1.f;
  load_a1<ACT>(a, a1, n_, g, false, &mean1, &rstd1, sh);
  if (tid == 0) { a.stats[0 * NG + blockIdx.x] = mean1; a.stats[1 * NG + blockIdx.x] = rstd1; }
  // thread t: channel lane t % cpg, output pixels t / cpg + k * plc
  const int plc = T / cpg, ch = tid % cpg, pl = tid / cpg;
  const bool active = pl < plc;
  float wt[9];
#pragma unroll
  for (int t9 = 0; t9 < 9; ++t9) wt[t9] = a.wgt[(size_t)t9 * C + g * cpg + ch];
  const float inv_ow = 1.f / (float)a.ow;
  float y2[R];
  float s[1] = {0.f};
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const int op = pl + k * plc;
    const bool ok = active && op < ohw;
    const int opc = min(op, ohw - 1), oy = fast_div(opc, inv_ow), ox = opc - oy * a.ow;
    const float v = stencil(a, a1, wt, oy, ox, ch);
    y2[k] = ok ? v : 0.f;
    s[0] += y2[k];
  }
  block_sum<1>(s, sh);
  const float cnt2 = (float)ohw * (float)cpg;
  const float mean2 = s[0] / cnt2;
  float q[1] = {0.f};
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const int op = pl + k * plc;
    const float d = y2[k] - mean2;
    q[0] += (active && op < ohw) ? d * d : 0.f;
  }
  block_sum<1>(q, sh);
  const float rstd2 = 1.f / sqrtf(q[0] / cnt2 + a.eps);
  if (tid == 0) { a.stats[2 * NG + blockIdx.x] = mean2; a.stats[3 * NG + blockIdx.x] = rstd2; }
  if (!active) return;
  const int cg = g * cpg + ch;
  const float sc = rstd2 * a.gamma2[cg], shf = a.beta2[cg] - mean2 * sc;
  const bool drop = a.drop_rate > 0.f;
  const float keep_scale = drop ? 1.f / (1.f - a.drop_rate) : 1.f;
  const uint64_t seed = a.seed2 + (a.seed_dev ? *a.seed_dev : 0ull);
  const uint64_t samp = (uint64_t)n_ * (uint64_t)ohw * (uint64_t)C;
  float* yo = a.y + (size_t)n_ * ohw * C;
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const int op = pl + k * plc;
    if (op < ohw) {
      float v = rn::act_fwd(y2[k] * sc + shf, ACT);
      if (drop) v = drop_apply(v, a.drop_rate, keep_scale, seed, samp + (uint64_t)op * C + cg);
      yo[(size_t)op * C + cg] = v;
    }
  }
}

template <int ACT>
__global__ __launch_bounds__(T) void dwgn_bwd_kernel(const Args a) {
  extern __shared__ float smem[];
  __shared__ float sh[2 * (T / 64)];
  __shared__ float red2[8 * 64], csum[64], csum2[64], coef[2];
  const int tid = threadIdx.x, cpg = a.cpg, C = a.c;
  const int n_ = blockIdx.x / a.groups, g = blockIdx.x - n_ * a.groups;
  const int hw = a.h * a.w, ohw = a.oh * a.ow, NG = a.n * a.groups;
  float* a1 = smem;                        // [hw][cpg]: a1, later g1
  float* d2 = smem + (size_t)hw * cpg;     // [ohw][cpg]: g2, then dy2
  float* red = d2 + (size_t)ohw * cpg;     // [T]
  float mean1 = a.stats[0 * NG + blockIdx.x], rstd1 = a.stats[1 * NG + blockIdx.x];
  const float mean2 = a.stats[2 * NG + blockIdx.x], rstd2 = a.stats[3 * NG + blockIdx.x];
  load_a1<ACT>(a, a1, n_, g, true, &mean1, &rstd1, sh);
  const int plc = T / cpg, ch = tid % cpg, pl = tid / cpg, cg = g * cpg + ch;
  const bool active = pl < plc;
  const float inv_ow = 1.f / (float)a.ow, inv_w = 1.f / (float)a.w;
  const int sshift = a.stride - 1;             // stride 1 or 2
  const bool drop = a.drop_rate > 0.f;
  const float keep_scale = drop ? 1.f / (1.f - a.drop_rate) : 1.f;
  const uint64_t dev_seed = a.seed_dev ? *a.seed_dev : 0ull;
  float wt[9];
#pragma unroll
  for (int t9 = 0; t9 < 9; ++t9) wt[t9] = a.wgt[(size_t)t9 * C + cg];
  const float gam2 = a.gamma2[cg], bet2 = a.beta2[cg], gam1 = a.gamma1[cg], bet1 = a.beta1[cg];
  const size_t nC = (size_t)a.n * C;
  // ---- GroupNorm 2 backward, pass a: g2 = dy * mask2 * act'(z2) -> d2; per-channel sums (g2, g2 * xhat2)
  {
    const uint64_t seed2 = a.seed2 + dev_seed, samp2 = (uint64_t)n_ * (uint64_t)ohw * (uint64_t)C;
    const float* dyg = a.dy + (size_t)n_ * ohw * C;
    float s[2] = {0.f, 0.f};
    if (active)
      for (int op0 = pl; op0 < ohw; op0 += U * plc) {
        float dyv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) dyv[u] = dyg[(size_t)min(op0 + u * plc, ohw - 1) * C + cg];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int op = op0 + u * plc;
          if (op < ohw) {
            const int oy = fast_div(op, inv_ow), ox = op - oy * a.ow;
            const float xh = (stencil(a, a1, wt, oy, ox, ch) - mean2) * rstd2;
            float gg = dyv[u];
            if (drop) gg = drop_apply(gg, a.drop_rate, keep_scale, seed2, samp2 + (uint64_t)op * C + cg);
            gg *= rn::act_grad(xh * gam2 + bet2, ACT);
            d2[op * cpg + ch] = gg;
            s[0] += gg; s[1] += gg * xh;
          }
        }
      }
    channel_sum(s[0], cpg, plc, active, red, red2, csum);
    channel_sum(s[1], cpg, plc, active, red, red2, csum2);
    if (tid < cpg) {
      a.rows[3 * nC + (size_t)n_ * C + g * cpg + tid] = csum[tid];         // dbeta2 row of this sample
      a.rows[2 * nC + (size_t)n_ * C + g * cpg + tid] = csum2[tid];        // dgamma2
    }
    if (tid == 0) {
      float t1 = 0.f, t2 = 0.f;
      for (int j = 0; j < cpg; ++j) { const float gm = a.gamma2[g * cpg + j]; t1 += gm * csum[j]; t2 += gm * csum2[j]; }
      const float inv_m = 1.f / ((float)ohw * (float)cpg);
      coef[0] = t1 * inv_m; coef[1] = t2 * inv_m;
    }
    __syncthreads();
    // pass b: dy2 = rstd2 (gamma2 g2 - c1 - xhat2 c2) -> d2; depthwise weight gradient: sum over pixels a1(tap) * dy2
    const float c1 = coef[0], c2 = coef[1];
    float wacc[9];
#pragma unroll
    for (int t9 = 0; t9 < 9; ++t9) wacc[t9] = 0.f;
    if (active)
      for (int op = pl; op < ohw; op += plc) {
        const int oy = fast_div(op, inv_ow), ox = op - oy * a.ow;
        const float xh = (stencil(a, a1, wt, oy, ox, ch) - mean2) * rstd2;
        const float dv = rstd2 * (gam2 * d2[op * cpg + ch] - c1 - xh * c2);
        d2[op * cpg + ch] = dv;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          const int ih = oy * a.stride - a.pad_t + kh;
          const int ihc = min(max(ih, 0), a.h - 1);
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const int iw = ox * a.stride - a.pad_l + kw;
            const int iwc = min(max(iw, 0), a.w - 1);
            const float m = ((unsigned)ih < (unsigned)a.h && (unsigned)iw < (unsigned)a.w) ? 1.f : 0.f;
            wacc[kh * 3 + kw] = fmaf(a1[(ihc * a.w + iwc) * cpg + ch] * m, dv, wacc[kh * 3 + kw]);
          }
        }
      }
    // the nine taps' per-channel sums, one at a time through the same scratch
#pragma unroll
    for (int t9 = 0; t9 < 9; ++t9) {
      channel_sum(wacc[t9], cpg, plc, active, red, red2, csum);
      if (tid < cpg) a.rows[4 * nC + ((size_t)n_ * 9 + t9) * C + g * cpg + tid] = csum[tid];
    }
  }
  __syncthreads();   // d2 (dy2) complete; a1 no longer needed: it is overwritten with g1 below
  // ---- depthwise data gradient + GroupNorm 1 backward, pass a: g1 -> a1 buffer, per-channel sums
  const float* xg = a.x + (size_t)n_ * hw * C;
  {
    const uint64_t seed1 = a.seed1 + dev_seed, samp1 = (uint64_t)n_ * (uint64_t)hw * (uint64_t)C;
    float s[2] = {0.f, 0.f};
    if (active)
      for (int ip0 = pl; ip0 < hw; ip0 += U * plc) {
       float xv[U];
#pragma unroll
       for (int u = 0; u < U; ++u) xv[u] = xg[(size_t)min(ip0 + u * plc, hw - 1) * C + cg];
#pragma unroll
       for (int u = 0; u < U; ++u) {
        const int ip = ip0 + u * plc;
        if (ip >= hw) break;
        const int ih = fast_div(ip, inv_w), iw = ip - ih * a.w;
        float da = 0.f;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {     // same visiting order as dw_dgrad_body's generic 3x3 branch
          const int ohs = ih + a.pad_t - kh;
          const int oh_ = ohs >> sshift;       // arithmetic shift: negative ohs stays negative and fails `ohs >= 0`
          const bool rok = ohs >= 0 && (oh_ << sshift) == ohs && oh_ < a.oh;
          const int ohc = min(max(oh_, 0), a.oh - 1);
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const int ows = iw + a.pad_l - kw;
            const int ow_ = ows >> sshift;
            const float m = (rok && ows >= 0 && (ow_ << sshift) == ows && ow_ < a.ow) ? 1.f : 0.f;
            const int owc = min(max(ow_, 0), a.ow - 1);
            da = fmaf(d2[(ohc * a.ow + owc) * cpg + ch] * m, wt[kh * 3 + kw], da);
          }
        }
        const float xh = (xv[u] - mean1) * rstd1;
        float gg = da;
        if (drop) gg = drop_apply(gg, a.drop_rate, keep_scale, seed1, samp1 + (uint64_t)ip * C + cg);
        gg *= rn::act_grad(xh * gam1 + bet1, ACT);
        a1[ip * cpg + ch] = gg;               // only this thread reads / writes (ip, ch) from here on
        s[0] += gg; s[1] += gg * xh;
       }
      }
    channel_sum(s[0], cpg, plc, active, red, red2, csum);
    channel_sum(s[1], cpg, plc, active, red, red2, csum2);
    if (tid < cpg) {
      a.rows[1 * nC + (size_t)n_ * C + g * cpg + tid] = csum[tid];         // dbeta1
      a.rows[0 * nC + (size_t)n_ * C + g * cpg + tid] = csum2[tid];        // dgamma1
    }
    if (tid == 0) {
      float t1 = 0.f, t2 = 0.f;
      for (int j = 0; j < cpg; ++j) { const float gm = a.gamma1[g * cpg + j]; t1 += gm * csum[j]; t2 += gm * csum2[j]; }
      const float inv_m = 1.f / ((float)hw * (float)cpg);
      coef[0] = t1 * inv_m; coef[1] = t2 * inv_m;
    }
    __syncthreads();
    const float c1 = coef[0], c2 = coef[1];
    float* dxg = a.dx + (size_t)n_ * hw * C;
    if (active)
      for (int ip0 = pl; ip0 < hw; ip0 += U * plc) {
        float xv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) xv[u] = xg[(size_t)min(ip0 + u * plc, hw - 1) * C + cg];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int ip = ip0 + u * plc;
          if (ip < hw) dxg[(size_t)ip * C + cg] = rstd1 * (gam1 * a1[ip * cpg + ch] - c1 - (xv[u] - mean1) * rstd1 * c2);
        }
      }
  }
}

int fill(const rn_dwgn_params* p, Args* a) {
  RN_CHECK_ARG(p, "dwgn: null params");
  RN_CHECK_ARG(p->n >= 1 && p->h >= 1 && p->w >= 1 && p->c >= 1 && (p->stride == 1 || p->stride == 2) && p->groups >= 1 &&
                   p->c % p->groups == 0, "dwgn: bad shape");
  RN_CHECK_ARG(p->drop_rate >= 0.f && p->drop_rate < 1.f, "dwgn: drop_rate %f", p->drop_rate);
  a->n = p->n; a->h = p->h; a->w = p->w; a->c = p->c; a->stride = p->stride; a->groups = p->groups; a->cpg = p->c / p->groups;
  a->act = p->act; a->eps = p->eps; a->drop_rate = p->drop_rate; a->seed1 = p->drop_seed1; a->seed2 = p->drop_seed2;
  a->seed_dev = p->drop_seed_dev;
  rn::same_pad(p->h, 3, p->stride, &a->oh, &a->pad_t);
  rn::same_pad(p->w, 3, p->stride, &a->ow, &a->pad_l);
  return RN_OK;
}
size_t fwd_lds(const Args& a) { return (size_t)a.h * a.w * a.cpg * 4; }
size_t bwd_lds(const Args& a) { return ((size_t)a.h * a.w + (size_t)a.oh * a.ow) * a.cpg * 4 + (size_t)T * 4; }
int fwd_depth(const Args& a) { return rn::ceil_div(a.oh * a.ow, T / a.cpg); }
bool shape_ok(const Args& a, bool backward) {
  if (a.cpg > 64 || a.cpg < 1 || a.act == RN_ACT_SIGMOID) return false;
  if ((double)a.n * a.h * a.w * a.c * 4.0 >= 2147483648.0) return false;
  if (backward) return bwd_lds(a) <= LDS_LIMIT;
  return fwd_lds(a) <= LDS_LIMIT && fwd_depth(a) <= 32;
}
}  // namespace

extern "C" int rn_dwgn_supported(const rn_dwgn_params* p, int backward) {
  Args a = {};
  if (fill(p, &a)) return 0;
  return shape_ok(a, backward != 0) ? 1 : 0;
}

extern "C" int rn_dwgn_fwd(const float* x, const float* gamma1, const float* beta1, const float* wgt, const float* gamma2,
                           const float* beta2, float* y, float* stats, const rn_dwgn_params* p, rn_stream_t stream) {
  Args a = {};
  if (int e = fill(p, &a)) return e;
  RN_CHECK_ARG(x && gamma1 && beta1 && wgt && gamma2 && beta2 && y && stats, "dwgn fwd: null pointer");
  RN_UNSUPPORTED(!shape_ok(a, false), "dwgn fwd: a (sample, group) slice of %d x %d x %d does not fit (rn_dwgn_supported)", a.h, a.w, a.cpg);
  a.x = x; a.gamma1 = gamma1; a.beta1 = beta1; a.wgt = wgt; a.gamma2 = gamma2; a.beta2 = beta2; a.y = y; a.stats = stats;
  const size_t lds = fwd_lds(a);
  const int depth = fwd_depth(a);
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)(a.n * a.groups));
#define RN_DWGN_FWD2(ACT_, R_)                                                                                              \
  do {                                                                                                                      \
    static const hipError_t attr_ = hipFuncSetAttribute((const void*)dwgn_fwd_kernel<ACT_, R_>,                              \
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_LIMIT);        \
    (void)attr_;                                                                                                            \
    hipLaunchKernelGGL((dwgn_fwd_kernel<ACT_, R_>), grid, dim3(T), lds, st, a);                                             \
  } while (0)
#define RN_DWGN_FWD(ACT_)                              \
  do {                                                 \
    if (depth <= 4) RN_DWGN_FWD2(ACT_, 4);             \
    else if (depth <= 8) RN_DWGN_FWD2(ACT_, 8);        \
    else if (depth <= 16) RN_DWGN_FWD2(ACT_, 16);      \
    else RN_DWGN_FWD2(ACT_, 32);                       \
  } while (0)
  switch (a.act) {
    case RN_ACT_RELU: RN_DWGN_FWD(RN_ACT_RELU); break;
    case RN_ACT_ELU: RN_DWGN_FWD(RN_ACT_ELU); break;
    case RN_ACT_RELU6: RN_DWGN_FWD(RN_ACT_RELU6); break;
    default: RN_DWGN_FWD(RN_ACT_NONE); break;
  }
#undef RN_DWGN_FWD
#undef RN_DWGN_FWD2
  RN_LAUNCH_CHECK();
  return RN_OK;
}

extern "C" int rn_dwgn_bwd(const float* x, const float* dy, const float* gamma1, const float* beta1, const float* wgt,
                           const float* gamma2, const float* beta2, const float* stats, float* dx, float* rows,
                           const rn_dwgn_params* p, rn_stream_t stream) {
  Args a = {};
  if (int e = fill(p, &a)) return e;
  RN_CHECK_ARG(x && dy && gamma1 && beta1 && wgt && gamma2 && beta2 && stats && dx && rows, "dwgn bwd: null pointer");
  RN_UNSUPPORTED(!shape_ok(a, true), "dwgn bwd: the slices of %d x %d x %d do not fit (rn_dwgn_supported)", a.h, a.w, a.cpg);
  a.x = x; a.dy = dy; a.gamma1 = gamma1; a.beta1 = beta1; a.wgt = wgt; a.gamma2 = gamma2; a.beta2 = beta2;
  a.stats = const_cast<float*>(stats); a.dx = dx; a.rows = rows;
  const size_t lds = bwd_lds(a);
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)(a.n * a.groups));
#define RN_DWGN_BWD(ACT_)                                                                                                \
  do {                                                                                                                   \
    static const hipError_t attr_ = hipFuncSetAttribute((const void*)dwgn_bwd_kernel<ACT_>,                              \
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_LIMIT);    \
    (void)attr_;                                                                                                         \
    hipLaunchKernelGGL((dwgn_bwd_kernel<ACT_>), grid, dim3(T), lds, st, a);                                              \
  } while (0)
  switch (a.act) {
    case RN_ACT_RELU: RN_DWGN_BWD(RN_ACT_RELU); break;
    case RN_ACT_ELU: RN_DWGN_BWD(RN_ACT_ELU); break;
    case RN_ACT_RELU6: RN_DWGN_BWD(RN_ACT_RELU6); break;
    default: RN_DWGN_BWD(RN_ACT_NONE); break;
  }
#undef RN_DWGN_BWD
  RN_LAUNCH_CHECK();
  return RN_OK;
}
