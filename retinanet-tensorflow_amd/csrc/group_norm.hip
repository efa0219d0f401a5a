// GroupNorm fused with the activation, dropout and residual add that follow it in every
// reference block (normalization.py:20-35; Sequential([conv, Normalization(), activation,
// Dropout]) e.g. mobilenet_v2.py:56-80; residual mobilenet_v2.py:91-92).
//
// HBM-bound.  Forward = 1 read for the statistics + 1 read / 1 write for the apply pass;
// backward = 2 reads of (x, dy) + 1 write.  Statistics are reduced in a fixed order
// (per-thread fp32 over a short run -> LDS tree -> per-chunk partial -> fp64 finalize), so the
// result is bitwise reproducible and safe against E[x^2]-E[x]^2 cancellation.
// Several tensors that share gamma/beta (one head layer over P3..P7) are one launch.
#include "rn_common.h"

namespace {

constexpr int T = 256;

struct GnSeg {
  const float* x; float* y; const float* res; const float* dy; float* dx; float* dres; float* mean; float* rstd;
  int n, hw;
  int sample_start;  // first global sample id
  int chunk_start;   // first partial row
  int chunks;        // chunks per sample
  int ppc;           // pixels per chunk
  int x_ld;          // floats between consecutive pixels of x (c when dense): x may be the first c channels of a wider buffer
  int dx_ld;         // the same for dx; dx_acc != 0: dx += (several layers of a DenseNet block feed one gradient buffer)
  int dx_acc;
};

struct GnArgs {
  GnSeg seg[RN_MAX_SEG];
  int nseg, c, groups, cpg, act;
  int in_half, out_half;  // forward only: fp16 storage of x / of y and residual (inference path)
  int act_after_res;  // 1: y = act(GN(x) + residual) (ResNeXt, resnet.py:99-101) instead of act(GN(x)) + residual
  float eps, drop_rate;
  uint64_t seed;
  const uint64_t* seed_dev;  // optional device counter added to seed (graph replays get fresh masks)
  const float* gamma; const float* beta;
  float* partial;  // [2][total_chunks][c]: plane 0 = sum x (bwd: sum g), plane 1 = sum x^2 (bwd: sum g*xhat)
  float* coef;     // bwd: [samples][groups][2]
  float* dgamma; float* dbeta;
  int total_chunks, total_samples;
  unsigned long long* xchg;  // grid-resident path: [total_chunks][groups][2] tagged per-block group sums (inside the sync region)
  unsigned* sync;            // grid-resident path: leave counter, call counter, error word (caller-owned, zero-initialised)
  int coop_ppc, coop_r;  // grid-resident path: pixels per block, float4 values per thread
  int strided;     // some segment reads a channel prefix of a wider buffer / accumulates dx: three-kernel path only
  int slice_wc;    // slice-resident path: channels per block (a whole number of groups), 0 = not used
  float* pgrad;    // slice-resident bwd: [2][total_samples][c] per-sample sum g (plane 0) and sum g*xhat (plane 1)
  // rows path: [chunk][per_group ? groups : c] pairs (the producer's, rn_gn_params.stat_rows, or rows_out below); slab width
  const float2* rows; int rows_per_group, rows_sw;   // (rows of a sample: seg.chunks, first row: seg.chunk_start)
  float2* rows_out;                                  // gn_rows_partial_kernel: [total_chunks][groups]
  int nt_loads;    // fp16 apply: x and the residual are read for the last time here -- non-temporal loads (large inference batches)
  // fp16 apply: the residual is itself a RAW conv output whose (activation-free) GroupNorm is applied here (ResNeXt's projection branch)
  const float* res_mean; const float* res_rstd; const float* res_gamma; const float* res_beta; int res_groups;
};

__device__ __forceinline__ int seg_of_sample(const GnArgs& a, int q) {
  int s = 0;
  while (s + 1 < a.nseg && q >= a.seg[s + 1].sample_start) ++s;
  return s;
}
__device__ __forceinline__ int seg_of_chunk(const GnArgs& a, int ch) {
  int s = 0;
  while (s + 1 < a.nseg && ch >= a.seg[s + 1].chunk_start) ++s;
  return s;
}

// Per-channel partial sums of (v1, v2) over a chunk of pixels.  BWD=false: (x, x^2);
// BWD=true: (dya, dya*xhat) with dya = dy * dropmask * act'(z).
template <bool BWD>
__global__ __launch_bounds__(T) void gn_partial_kernel(const GnArgs a) {
  __shared__ float red[T][8];
  __shared__ float tab[2048][4];  // bwd: mean, rstd, gamma, beta per channel
  const int tid = threadIdx.x;
  const int ch = blockIdx.x;
  const int s = seg_of_chunk(a, ch);
  const GnSeg& sg = a.seg[s];
  const int local = ch - sg.chunk_start;
  const int nl = local / sg.chunks, ck = local - nl * sg.chunks;
  const int C = a.c, CQ = C >> 2;
  const int QP = CQ < T ? CQ : T;      // channel quads handled per pass (C > 1024 needs several passes)
  const int lanes = T / QP;
  const int p_begin = ck * sg.ppc, p_end = min(p_begin + sg.ppc, sg.hw);
  const size_t base = (size_t)nl * sg.hw * C;
  const size_t xbase = (size_t)nl * sg.hw * sg.x_ld;
  const float* __restrict__ x = sg.x + xbase;
  const float* __restrict__ dy = BWD ? sg.dy + base : nullptr;
  const float* __restrict__ rs = (BWD && a.act_after_res && sg.res) ? sg.res + base : nullptr;

  if (BWD) {
    for (int c = tid; c < C; c += T) {
      const int g = c / a.cpg;
      tab[c][0] = sg.mean[nl * a.groups + g];
      tab[c][1] = sg.rstd[nl * a.groups + g];
      tab[c][2] = a.gamma[c];
      tab[c][3] = a.beta[c];
    }
    __syncthreads();
  }
  const bool drop = a.drop_rate > 0.f;
  const float keep_scale = drop ? 1.f / (1.f - a.drop_rate) : 1.f;
  const uint64_t samp_off = (uint64_t)(sg.sample_start + nl) * (uint64_t)sg.hw * (uint64_t)C;
  const uint64_t seed = a.seed + (a.seed_dev ? *a.seed_dev : 0ull);

  for (int qbase = 0; qbase < CQ; qbase += QP) {
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    const int q4 = qbase + tid % QP, pl = tid / QP;
    if (pl < lanes && q4 < CQ) {
#pragma unroll 4
      for (int p = p_begin + pl; p < p_end; p += lanes) {
        const size_t off = (size_t)p * C + q4 * 4, xoff = (size_t)p * sg.x_ld + q4 * 4;
        const float4 xv = BWD ? *reinterpret_cast<const float4*>(x + xoff) : rn::ld4(sg.x, xbase + xoff, a.in_half);
        const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
        if (!BWD) {
#pragma unroll
          for (int j = 0; j < 4; ++j) { s1[j] += xs[j]; s2[j] += xs[j] * xs[j]; }
        } else {
          const float4 dv = *reinterpret_cast<const float4*>(dy + off);
          const float ds[4] = {dv.x, dv.y, dv.z, dv.w};
          float rr[4] = {0.f, 0.f, 0.f, 0.f};
          if (rs) {
            const float4 rv = *reinterpret_cast<const float4*>(rs + off);
            rr[0] = rv.x; rr[1] = rv.y; rr[2] = rv.z; rr[3] = rv.w;
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int c = q4 * 4 + j;
            const float xh = (xs[j] - tab[c][0]) * tab[c][1];
            const float z = xh * tab[c][2] + tab[c][3] + rr[j];
            float g = ds[j];
            if (drop) g = (rn::uniform01(seed, samp_off + off + j) >= a.drop_rate) ? g * keep_scale : 0.f;
            g *= rn::act_grad(z, a.act);
            s1[j] += g;
            s2[j] += g * xh;
          }
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) { red[tid][j] = s1[j]; red[tid][4 + j] = s2[j]; }
    __syncthreads();
    // combine the pixel lanes: thread (q, comp) sums component comp of channel quad q in lane order
    const int nq = min(QP, CQ - qbase);
    for (int e = tid; e < nq * 8; e += T) {
      const int q = e >> 3, comp = e & 7;
      float t = 0.f;
      for (int l = 0; l < lanes; ++l) t += red[l * QP + q][comp];
      a.partial[(size_t)(comp >> 2) * a.total_chunks * C + (size_t)ch * C + (qbase + q) * 4 + (comp & 3)] = t;
    }
  }
}

// The forward statistics pass over fp16 storage (the inference path: the head towers' maps, 179 MB per GroupNorm at cfg 5): 16 bytes
// (8 channels) per lane and load, four loads in flight -- the generic kernel above walks 8-byte quads through a converting load and
// streams at 1.9 TB/s there.  Same chunks, same output layout; channel octets x pixel lanes, the lanes summed in order.
typedef _Float16 part_half8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(T) void gn_partial_f16x8_kernel(const GnArgs a) {
  __shared__ float red[T][16];
  const int tid = threadIdx.x;
  const int ch = blockIdx.x;
  const int s = seg_of_chunk(a, ch);
  const GnSeg& sg = a.seg[s];
  const int local = ch - sg.chunk_start;
  const int nl = local / sg.chunks, ck = local - nl * sg.chunks;
  const int C = a.c, C8 = C >> 3;                  // (the host checks: C8 divides T)
  const int lanes = T / C8;
  const int p_begin = ck * sg.ppc, p_end = min(p_begin + sg.ppc, sg.hw);
  const _Float16* __restrict__ x = reinterpret_cast<const _Float16*>(sg.x) + (size_t)nl * sg.hw * sg.x_ld;
  const int q8 = tid % C8, pl = tid / C8;
  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s1[j] = s2[j] = 0.f;
  for (int p0 = p_begin + pl; p0 < p_end; p0 += 4 * lanes) {
    part_half8 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const part_half8*>(x + (size_t)min(p0 + u * lanes, p_end - 1) * sg.x_ld + q8 * 8);
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (p0 + u * lanes < p_end) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float f = (float)v[u][j]; s1[j] += f; s2[j] = fmaf(f, f, s2[j]); }
      }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) { red[tid][j] = s1[j]; red[tid][8 + j] = s2[j]; }
  __syncthreads();
  for (int e = tid; e < C8 * 16; e += T) {
    const int q = e >> 4, comp = e & 15;
    float t = 0.f;
    for (int l = 0; l < lanes; ++l) t += red[l * C8 + q][comp];
    a.partial[(size_t)(comp >> 3) * a.total_chunks * C + (size_t)ch * C + q * 8 + (comp & 7)] = t;
  }
}

// the forward finalize by 64-channel slabs (see gn_finalize_cols_kernel below), any number of segments: block = (sample, slab)
__global__ __launch_bounds__(T) void gn_finalize_cols_segs_kernel(const GnArgs a) {
  __shared__ double red[4][64][2];
  __shared__ double chan[64][2];
  const int tid = threadIdx.x, cl = tid & 63, rl = tid >> 6;
  const int slabs = a.c >> 6;
  const int q = blockIdx.x / slabs, slab = blockIdx.x - q * slabs;
  const int s = seg_of_sample(a, q);
  const GnSeg& sg = a.seg[s];
  const int nl = q - sg.sample_start, rows = sg.chunks;
  const float* __restrict__ p1 = a.partial + (size_t)(sg.chunk_start + nl * rows) * a.c + slab * 64 + cl;
  const float* __restrict__ p2 = p1 + (size_t)a.total_chunks * a.c;
  double s1 = 0.0, s2 = 0.0;
  if (rows == 0) {       // rn_group_norm_fwd_f16_tiles: a segment nobody wrote rows for (a small map): the sums straight from the fp16 tensor
    const _Float16* __restrict__ x = reinterpret_cast<const _Float16*>(sg.x) + (size_t)nl * sg.hw * sg.x_ld + slab * 64 + cl;
    for (int p = rl; p < sg.hw; p += 4) { const float f = (float)x[(size_t)p * sg.x_ld]; s1 += (double)f; s2 += (double)f * (double)f; }
  }
  for (int r0 = rl; r0 < rows; r0 += 32) {
    float u[8], v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const size_t rr = (size_t)min(r0 + 4 * j, rows - 1) * a.c;
      u[j] = p1[rr]; v[j] = p2[rr];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (r0 + 4 * j < rows) { s1 += (double)u[j]; s2 += (double)v[j]; }
  }
  red[rl][cl][0] = s1; red[rl][cl][1] = s2;
  __syncthreads();
  if (tid < 64) {
    chan[tid][0] = ((red[0][tid][0] + red[1][tid][0]) + red[2][tid][0]) + red[3][tid][0];
    chan[tid][1] = ((red[0][tid][1] + red[1][tid][1]) + red[2][tid][1]) + red[3][tid][1];
  }
  __syncthreads();
  const int gps = 64 / a.cpg;
  if (tid < gps) {
    double v1 = 0.0, v2 = 0.0;
    for (int j = 0; j < a.cpg; ++j) { v1 += chan[tid * a.cpg + j][0]; v2 += chan[tid * a.cpg + j][1]; }
    const double m = (double)sg.hw * (double)a.cpg;
    const double mean = v1 / m;
    double var = v2 / m - mean * mean;
    if (var < 0.0) var = 0.0;
    const int g = slab * gps + tid;
    sg.mean[nl * a.groups + g] = (float)mean;
    sg.rstd[nl * a.groups + g] = (float)(1.0 / sqrt(var + (double)a.eps));
  }
}

// Finalize: fp64 reduction of the chunk partials, one 256-thread block per (sample, group):
//   BWD=false -> mean / rstd;  BWD=true -> coef[g] = (sum_c gamma_c*S1_c, sum_c gamma_c*S2_c)/m.
// In the backward launch the blocks after samples*groups compute the parameter gradients
// (dgamma_c = sum over all samples/chunks of S2_c, dbeta_c = sum of S1_c): 16 channels x 16 row
// lanes per block, coalesced float2 rows, fixed order => bitwise reproducible.
__device__ __forceinline__ void block_sum2(double& v1, double& v2, double (*sh)[T / 64]) {
  v1 = rn::wave_sum_d(v1);
  v2 = rn::wave_sum_d(v2);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { sh[0][wave] = v1; sh[1][wave] = v2; }
  __syncthreads();
  v1 = 0.0; v2 = 0.0;
#pragma unroll
  for (int w = 0; w < T / 64; ++w) { v1 += sh[0][w]; v2 += sh[1][w]; }
}

template <bool BWD>
__global__ __launch_bounds__(T) void gn_finalize_kernel(const GnArgs a) {
  __shared__ double sh[2][T / 64];
  __shared__ double pg[16][16][2];
  const int tid = threadIdx.x;
  const int ngroup_blocks = a.total_samples * a.groups;
  if ((int)blockIdx.x < ngroup_blocks) {
    const int q = blockIdx.x / a.groups, g = blockIdx.x - q * a.groups;
    const int s = seg_of_sample(a, q);
    const GnSeg& sg = a.seg[s];
    const int nl = q - sg.sample_start;
    const int items = sg.chunks * a.cpg;
    const float* base = a.partial + (size_t)(sg.chunk_start + nl * sg.chunks) * a.c;
    const size_t plane = (size_t)a.total_chunks * a.c;
    double v1 = 0.0, v2 = 0.0;
    for (int i = tid; i < items; i += T) {
      const int ck = i / a.cpg, c = g * a.cpg + (i - ck * a.cpg);
      const float p1 = base[(size_t)ck * a.c + c], p2 = base[plane + (size_t)ck * a.c + c];
      const double w = BWD ? (double)a.gamma[c] : 1.0;
      v1 += w * (double)p1;
      v2 += w * (double)p2;
    }
    block_sum2(v1, v2, sh);
    if (tid == 0) {
      const double m = (double)sg.hw * (double)a.cpg;
      if (!BWD) {
        const double mean = v1 / m;
        double var = v2 / m - mean * mean;
        if (var < 0.0) var = 0.0;
        sg.mean[nl * a.groups + g] = (float)mean;
        sg.rstd[nl * a.groups + g] = (float)(1.0 / sqrt(var + (double)a.eps));
      } else {
        a.coef[((size_t)q * a.groups + g) * 2 + 0] = (float)(v1 / m);
        a.coef[((size_t)q * a.groups + g) * 2 + 1] = (float)(v2 / m);
      }
    }
  } else if (BWD) {
    const int cl = tid & 15, rl = tid >> 4;
    const int c = ((int)blockIdx.x - ngroup_blocks) * 16 + cl;
    double b = 0.0, g = 0.0;
    if (c < a.c) {
      const size_t plane = (size_t)a.total_chunks * a.c;
      for (int r = rl; r < a.total_chunks; r += 16) {
        b += (double)a.partial[(size_t)r * a.c + c];
        g += (double)a.partial[plane + (size_t)r * a.c + c];
      }
    }
    pg[rl][cl][0] = b; pg[rl][cl][1] = g;
    __syncthreads();
    if (rl == 0 && c < a.c) {
      double sb = 0.0, sgm = 0.0;
#pragma unroll
      for (int r = 0; r < 16; ++r) { sb += pg[r][cl][0]; sgm += pg[r][cl][1]; }
      a.dbeta[c] = (float)sb;
      a.dgamma[c] = (float)sgm;
    }
  }
}

// grid (blocks_x, samples).  FWD: y = drop(act((x-mean)*rstd*gamma+beta)) + res.
// BWD: dx = rstd*(gamma*dya - c1 - xhat*c2).
template <bool BWD>
__global__ __launch_bounds__(T) void gn_apply_kernel(const GnArgs a) {
  __shared__ float tab[2048][6];  // mean, rstd, gamma, beta, c1, c2 per channel
  const int q = blockIdx.y, tid = threadIdx.x;
  const int s = seg_of_sample(a, q);
  const GnSeg& sg = a.seg[s];
  const int nl = q - sg.sample_start;
  const int C = a.c, CQ = C >> 2;
  if ((int64_t)blockIdx.x * T >= (int64_t)sg.hw * CQ) return;  // the grid is sized for the largest segment
  for (int c = tid; c < C; c += T) {
    const int g = c / a.cpg;
    const float mean = sg.mean[nl * a.groups + g], rstd = sg.rstd[nl * a.groups + g];
    tab[c][0] = mean;
    tab[c][1] = rstd;
    tab[c][2] = a.gamma[c];
    tab[c][3] = a.beta[c];
    if (BWD) {
      tab[c][4] = a.coef[((size_t)q * a.groups + g) * 2 + 0];
      tab[c][5] = a.coef[((size_t)q * a.groups + g) * 2 + 1];
    }
  }
  __syncthreads();
  const size_t base = (size_t)nl * sg.hw * C;
  const size_t xbase = (size_t)nl * sg.hw * sg.x_ld;
  const float* __restrict__ x = sg.x + xbase;
  const bool drop = a.drop_rate > 0.f;
  const float keep_scale = drop ? 1.f / (1.f - a.drop_rate) : 1.f;
  const uint64_t samp_off = (uint64_t)q * (uint64_t)sg.hw * (uint64_t)C;
  const uint64_t seed = a.seed + (a.seed_dev ? *a.seed_dev : 0ull);
  const int64_t total = (int64_t)sg.hw * CQ;
  const bool strided = sg.x_ld != C || (BWD && sg.dx_ld != C);
#pragma unroll 2
  for (int64_t i = (int64_t)blockIdx.x * T + tid; i < total; i += (int64_t)gridDim.x * T) {
    const int q4 = (int)(i % CQ);
    const size_t off = (size_t)i * 4;
    const int64_t pix = strided ? i / CQ : 0;
    const size_t xoff = strided ? (size_t)pix * sg.x_ld + q4 * 4 : off;
    const float4 xv = BWD ? *reinterpret_cast<const float4*>(x + xoff) : rn::ld4(sg.x, xbase + xoff, a.in_half);
    const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
    float o[4];
    if (!BWD) {
      float r[4] = {0.f, 0.f, 0.f, 0.f};
      if (sg.res) {
        const float4 rv = rn::ld4(sg.res, base + off, a.out_half);
        r[0] = rv.x; r[1] = rv.y; r[2] = rv.z; r[3] = rv.w;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = q4 * 4 + j;
        const float z = (xs[j] - tab[c][0]) * tab[c][1] * tab[c][2] + tab[c][3];
        float v = rn::act_fwd(a.act_after_res ? z + r[j] : z, a.act);
        if (drop) v = (rn::uniform01(seed, samp_off + off + j) >= a.drop_rate) ? v * keep_scale : 0.f;
        o[j] = a.act_after_res ? v : v + r[j];
      }
      rn::st4(sg.y, base + off, a.out_half, make_float4(o[0], o[1], o[2], o[3]));
    } else {
      const float4 dv = *reinterpret_cast<const float4*>(sg.dy + base + off);
      const float ds[4] = {dv.x, dv.y, dv.z, dv.w};
      float rr[4] = {0.f, 0.f, 0.f, 0.f}, gr[4];
      const bool aar = a.act_after_res && sg.res;
      if (aar) {
        const float4 rv = *reinterpret_cast<const float4*>(sg.res + base + off);
        rr[0] = rv.x; rr[1] = rv.y; rr[2] = rv.z; rr[3] = rv.w;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = q4 * 4 + j;
        const float xh = (xs[j] - tab[c][0]) * tab[c][1];
        const float z = xh * tab[c][2] + tab[c][3] + rr[j];
        float g = ds[j];
        if (drop) g = (rn::uniform01(seed, samp_off + off + j) >= a.drop_rate) ? g * keep_scale : 0.f;
        g *= rn::act_grad(z, a.act);
        gr[j] = g;
        o[j] = tab[c][1] * (tab[c][2] * g - tab[c][4] - xh * tab[c][5]);
      }
      float* dxp = sg.dx + (strided ? (size_t)nl * sg.hw * sg.dx_ld + (size_t)pix * sg.dx_ld + q4 * 4 : base + off);
      if (sg.dx_acc) {
        const float4 old = *reinterpret_cast<const float4*>(dxp);
        o[0] += old.x; o[1] += old.y; o[2] += old.z; o[3] += old.w;
      }
      *reinterpret_cast<float4*>(dxp) = make_float4(o[0], o[1], o[2], o[3]);
      if (aar && sg.dres) *reinterpret_cast<float4*>(sg.dres + base + off) = make_float4(gr[0], gr[1], gr[2], gr[3]);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Rows path: GroupNorm as "partial-sum rows, then merge + apply", with no exchange between running blocks.
//   rows     [chunk][width] pairs (sum, sum of squares) -- or, backward, (sum gamma g, sum gamma g xhat) -- per block of the
//            PRODUCER: the conv / depthwise forward that computed x (rn_gn_params.stat_rows: the statistics pass over x
//            disappears), or gn_rows_partial_kernel below (one read of x, or of dy and x);
//   merge    every block of the apply kernel owns a channel slab (a whole number of groups and of float4 quads, ~32-64
//            channels) of a pixel chunk; it first issues the loads of its activations, then adds up its slab's share of the
//            sample's rows -- row lanes in fp64, fixed order -- while those loads are in flight, and applies.
// fp32 tensors; x / dx may be channel prefixes of a wider buffer (concat-free DenseNet blocks).  The path for whatever
// neither a slice-resident block nor the grid-resident kernel takes, and the configuration without any spinning kernel.
constexpr int AT = 256, RT = 512, ROWS_MAX_SLAB = 128, ROWS_MAX_ENTRIES = 8192;

// the slab's per-group totals (S, Q) of sample-local index nl -> gsum[group of slab][2] (fp64)
__device__ __forceinline__ void rows_merge(const GnArgs& a, const GnSeg& sg, int nl, int c0, double (*part)[2], double (*gsum)[2]) {
  const int tid = threadIdx.x, C = a.c, SW = a.rows_sw, cpg = a.cpg;
  const int per_group = a.rows_per_group;
  const int W = per_group ? a.groups : C;             // entries per row
  const int gw = per_group ? SW / cpg : SW;           // this slab's entries of a row
  const int e0 = per_group ? c0 / cpg : c0;
  const int RL = AT / gw;                             // row lanes
  const int el = tid % gw, rl = tid / gw;
  const float2* __restrict__ rp = a.rows + (size_t)(sg.chunk_start + nl * sg.chunks) * W + e0 + el;
  double S = 0.0, Q = 0.0;
  if (rl < RL) {
    for (int r0 = rl; r0 < sg.chunks; r0 += 8 * RL) {
      float2 w[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) w[j] = rp[(size_t)min(r0 + j * RL, sg.chunks - 1) * W];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (r0 + j * RL < sg.chunks) { S += (double)w[j].x; Q += (double)w[j].y; }
    }
  }
  part[tid][0] = S; part[tid][1] = Q;
  __syncthreads();
  const int ngs = SW / cpg, epg = per_group ? 1 : cpg;  // groups of the slab, row entries per group
  if (tid < ngs) {
    S = 0.0; Q = 0.0;
    for (int j = 0; j < epg; ++j)
      for (int l = 0; l < RL; ++l) { S += part[l * gw + tid * epg + j][0]; Q += part[l * gw + tid * epg + j][1]; }
    gsum[tid][0] = S; gsum[tid][1] = Q;
  }
  __syncthreads();
}

// grid (pixel chunks of the largest segment, channel slabs, samples)
template <int ACT, int R>
__global__ __launch_bounds__(AT) void gn_apply_rows_kernel(const GnArgs a) {
  __shared__ double part[AT][2];
  __shared__ double gsum[ROWS_MAX_SLAB][2];
  __shared__ float gstat[ROWS_MAX_SLAB][2];
  const int q = blockIdx.z, slab = blockIdx.y, tid = threadIdx.x;
  const GnSeg& sg = a.seg[seg_of_sample(a, q)];
  const int nl = q - sg.sample_start;
  const int C = a.c, SW = a.rows_sw, SQ = SW >> 2, lanes = AT / SQ;
  if ((int)blockIdx.x * lanes * R >= sg.hw) return;     // (whole block: the grid is sized for the largest segment)
  const int q4 = tid % SQ, pl = tid / SQ, c0 = slab * SW;
  const int p0 = (int)blockIdx.x * lanes * R + pl;
  const size_t base = (size_t)nl * sg.hw * C + c0 + q4 * 4;
  const float* __restrict__ x = sg.x + (size_t)nl * sg.hw * sg.x_ld + c0 + q4 * 4;   // (x may be a channel prefix of a wider buffer)
  const float* __restrict__ res = sg.res ? sg.res + base : nullptr;
  float4 v[R], rv[R];
#pragma unroll
  for (int k = 0; k < R; ++k) v[k] = *reinterpret_cast<const float4*>(x + (size_t)min(p0 + k * lanes, sg.hw - 1) * sg.x_ld);
  if (res) {
#pragma unroll
    for (int k = 0; k < R; ++k) rv[k] = *reinterpret_cast<const float4*>(res + (size_t)min(p0 + k * lanes, sg.hw - 1) * C);
  } else {
#pragma unroll
    for (int k = 0; k < R; ++k) rv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float gam[4], bet[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { gam[j] = a.gamma[c0 + q4 * 4 + j]; bet[j] = a.beta[c0 + q4 * 4 + j]; }
  const uint64_t seed = a.seed + (a.seed_dev ? *a.seed_dev : 0ull);
  rows_merge(a, sg, nl, c0, part, gsum);
  if (tid < SW / a.cpg) {
    const double m = (double)sg.hw * (double)a.cpg;
    const double mean = gsum[tid][0] / m;
    double var = gsum[tid][1] / m - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)a.eps));
    gstat[tid][0] = (float)mean; gstat[tid][1] = rstd;
    if (blockIdx.x == 0) { sg.mean[nl * a.groups + c0 / a.cpg + tid] = (float)mean; sg.rstd[nl * a.groups + c0 / a.cpg + tid] = rstd; }
  }
  __syncthreads();
  if (pl >= lanes || p0 >= sg.hw) return;
  float sc[4], sh[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int g = (q4 * 4 + j) / a.cpg;
    sc[j] = gstat[g][1] * gam[j];
    sh[j] = bet[j] - gstat[g][0] * sc[j];
  }
  const bool drop = a.drop_rate > 0.f;
  const float keep_scale = drop ? 1.f / (1.f - a.drop_rate) : 1.f;
  const uint64_t samp_off = (uint64_t)q * (uint64_t)sg.hw * (uint64_t)C;
  const bool aar = a.act_after_res != 0;
  float* __restrict__ y = sg.y + base;
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const int p = p0 + k * lanes;
    if (p < sg.hw) {
      const float xs[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
      const float r[4] = {rv[k].x, rv[k].y, rv[k].z, rv[k].w};
      float o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float z = xs[j] * sc[j] + sh[j];
        float t = rn::act_fwd(aar ? z + r[j] : z, ACT);
        if (drop) t = (rn::uniform01(seed, samp_off + (uint64_t)p * C + c0 + q4 * 4 + j) >= a.drop_rate) ? t * keep_scale : 0.f;
        o[j] = aar ? t : t + r[j];
      }
      *reinterpret_cast<float4*>(y + (size_t)p * C) = make_float4(o[0], o[1], o[2], o[3]);
    }
  }
}

// backward twin: dx = rstd (gamma g - c1 - xhat c2), g = dy * dropmask * act'(z) recomputed from dy and x; c1 / c2 are the
// merged rows (sum gamma g, sum gamma g xhat) / m of the group
template <int ACT, int R>
__global__ __launch_bounds__(AT) void gn_bwd_apply_rows_kernel(const GnArgs a) {
  __shared__ double part[AT][2];
  __shared__ double gsum[ROWS_MAX_SLAB][2];
  __shared__ float gstat[ROWS_MAX_SLAB][2];
  const int q = blockIdx.z, slab = blockIdx.y, tid = threadIdx.x;
  const GnSeg& sg = a.seg[seg_of_sample(a, q)];
  const int nl = q - sg.sample_start;
  const int C = a.c, SW = a.rows_sw, SQ = SW >> 2, lanes = AT / SQ;
  if ((int)blockIdx.x * lanes * R >= sg.hw) return;
  const int q4 = tid % SQ, pl = tid / SQ, c0 = slab * SW;
  const int p0 = (int)blockIdx.x * lanes * R + pl;
  const size_t base = (size_t)nl * sg.hw * C + c0 + q4 * 4;
  const float* __restrict__ x = sg.x + (size_t)nl * sg.hw * sg.x_ld + c0 + q4 * 4;
  const float* __restrict__ dy = sg.dy + base;
  const bool aar = a.act_after_res && sg.res;
  const float* __restrict__ res = aar ? sg.res + base : nullptr;
  float4 xv[R], dv[R], rv[R];
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const size_t pc = (size_t)min(p0 + k * lanes, sg.hw - 1);
    xv[k] = *reinterpret_cast<const float4*>(x + pc * sg.x_ld);
    dv[k] = *reinterpret_cast<const float4*>(dy + pc * C);
  }
  if (res) {
#pragma unroll
    for (int k = 0; k < R; ++k) rv[k] = *reinterpret_cast<const float4*>(res + (size_t)min(p0 + k * lanes, sg.hw - 1) * C);
  } else {
#pragma unroll
    for (int k = 0; k < R; ++k) rv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float mean[4], rstd[4], gam[4], bet[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = c0 + q4 * 4 + j, g = c / a.cpg;
    mean[j] = sg.mean[nl * a.groups + g]; rstd[j] = sg.rstd[nl * a.groups + g];
    gam[j] = a.gamma[c]; bet[j] = a.beta[c];
  }
  const uint64_t seed = a.seed + (a.seed_dev ? *a.seed_dev : 0ull);
  rows_merge(a, sg, nl, c0, part, gsum);
  if (tid < SW / a.cpg) {
    const double m = (double)sg.hw * (double)a.cpg;
    gstat[tid][0] = (float)(gsum[tid][0] / m); gstat[tid][1] = (float)(gsum[tid][1] / m);
  }
  __syncthreads();
  if (pl >= lanes || p0 >= sg.hw) return;
  float c1[4], c2[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int g = (q4 * 4 + j) / a.cpg;
    c1[j] = gstat[g][0]; c2[j] = gstat[g][1];
  }
  const bool drop = a.drop_rate > 0.f;
  const float keep_scale = drop ? 1.f / (1.f - a.drop_rate) : 1.f;
  const uint64_t samp_off = (uint64_t)q * (uint64_t)sg.hw * (uint64_t)C;
  float* __restrict__ dx = sg.dx + (size_t)nl * sg.hw * sg.dx_ld + c0 + q4 * 4;   // (dx may be a channel prefix too, and accumulate)
  float* __restrict__ dres = (aar && sg.dres) ? sg.dres + base : nullptr;
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const int p = p0 + k * lanes;
    if (p < sg.hw) {
      const float xs[4] = {xv[k].x, xv[k].y, xv[k].z, xv[k].w};
      const float ds[4] = {dv[k].x, dv[k].y, dv[k].z, dv[k].w};
      const float rr[4] = {rv[k].x, rv[k].y, rv[k].z, rv[k].w};
      float o[4], g[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float h = (xs[j] - mean[j]) * rstd[j];
        const float z = h * gam[j] + bet[j] + rr[j];
        float t = ds[j];
        if (drop) t = (rn::uniform01(seed, samp_off + (uint64_t)p * C + c0 + q4 * 4 + j) >= a.drop_rate) ? t * keep_scale : 0.f;
        g[j] = t * rn::act_grad(z, ACT);
        o[j] = rstd[j] * (gam[j] * g[j] - c1[j] - h * c2[j]);
      }
      float* dxp = dx + (size_t)p * sg.dx_ld;
      if (sg.dx_acc) {
        const float4 old = *reinterpret_cast<const float4*>(dxp);
        o[0] += old.x; o[1] += old.y; o[2] += old.z; o[3] += old.w;
      }
      *reinterpret_cast<float4*>(dxp) = make_float4(o[0], o[1], o[2], o[3]);
      if (dres) *reinterpret_cast<float4*>(dres + (size_t)p * C) = make_float4(g[0], g[1], g[2], g[3]);
    }
  }
}

// The rows of a tensor that did not come with any: one block per chunk (the three-kernel path's chunking), all channels,
// thread = (channel quad, pixel lane).  FWD: per-group (sum x, sum x^2).  BWD: per-group (sum gamma g, sum gamma g xhat),
// and the per-channel planes (sum g, sum g xhat) of every chunk for the parameter gradients (a.partial).
template <bool BWD, int ACT>
__global__ __launch_bounds__(RT) void gn_rows_partial_kernel(const GnArgs a) {
  __shared__ float red[RT][8];
  __shared__ float chan[2048][2];
  const int tid = threadIdx.x, C = a.c, CQ = C >> 2, lanes = RT / CQ;
  const int ch = blockIdx.x;
  const GnSeg& sg = a.seg[seg_of_chunk(a, ch)];
  const int local = ch - sg.chunk_start;
  const int nl = local / sg.chunks, ck = local - nl * sg.chunks;
  const int q = sg.sample_start + nl;
  const int p_begin = ck * sg.ppc, p_end = min(p_begin + sg.ppc, sg.hw);
  const size_t base = (size_t)nl * sg.hw * C;
  const int q4 = tid % CQ, pl = tid / CQ;
  const float* __restrict__ x = sg.x + (size_t)nl * sg.hw * sg.x_ld + q4 * 4;
  const float* __restrict__ dy = BWD ? sg.dy + base + q4 * 4 : nullptr;
  const float* __restrict__ res = (BWD && a.act_after_res && sg.res) ? sg.res + base + q4 * 4 : nullptr;
  float mean[4] = {0.f, 0.f, 0.f, 0.f}, rstd[4] = {1.f, 1.f, 1.f, 1.f}, gam[4], bet[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = q4 * 4 + j;
    gam[j] = a.gamma[c]; bet[j] = a.beta[c];
    if (BWD) { mean[j] = sg.mean[nl * a.groups + c / a.cpg]; rstd[j] = sg.rstd[nl * a.groups + c / a.cpg]; }
  }
  const bool drop = BWD && a.drop_rate > 0.f;
  const float keep_scale = drop ? 1.f / (1.f - a.drop_rate) : 1.f;
  const uint64_t samp_off = (uint64_t)q * (uint64_t)sg.hw * (uint64_t)C;
  const uint64_t seed = a.seed + (a.seed_dev ? *a.seed_dev : 0ull);
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  if (pl < lanes) {
    for (int pb = p_begin + pl; pb < p_end; pb += 4 * lanes) {   // four pixels of loads in flight per thread
      float4 xv[4], dv[4], rv[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const size_t off = (size_t)min(pb + k * lanes, sg.hw - 1) * C;
        xv[k] = *reinterpret_cast<const float4*>(x + (size_t)min(pb + k * lanes, sg.hw - 1) * sg.x_ld);
        if (BWD) dv[k] = *reinterpret_cast<const float4*>(dy + off);
        rv[k] = (BWD && res) ? *reinterpret_cast<const float4*>(res + off) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int p = pb + k * lanes;
        if (p < p_end) {
          const float xs[4] = {xv[k].x, xv[k].y, xv[k].z, xv[k].w};
          if (!BWD) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { s1[j] += xs[j]; s2[j] += xs[j] * xs[j]; }
          } else {
            const float ds[4] = {dv[k].x, dv[k].y, dv[k].z, dv[k].w};
            const float rr[4] = {rv[k].x, rv[k].y, rv[k].z, rv[k].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float h = (xs[j] - mean[j]) * rstd[j];
              const float z = h * gam[j] + bet[j] + rr[j];
              float t = ds[j];
              if (drop) t = (rn::uniform01(seed, samp_off + (uint64_t)p * C + q4 * 4 + j) >= a.drop_rate) ? t * keep_scale : 0.f;
              const float g = t * rn::act_grad(z, ACT);
              s1[j] += g;
              s2[j] += g * h;
            }
          }
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) { red[tid][j] = s1[j]; red[tid][4 + j] = s2[j]; }
  __syncthreads();
  for (int e = tid; e < CQ * 8; e += RT) {      // pixel lanes in order
    const int qd = e >> 3, comp = e & 7;
    float t = 0.f;
    for (int l = 0; l < lanes; ++l) t += red[l * CQ + qd][comp];
    const int c = qd * 4 + (comp & 3);
    chan[c][comp >> 2] = t;
    if (BWD) a.partial[(size_t)(comp >> 2) * a.total_chunks * C + (size_t)ch * C + c] = t;
  }
  __syncthreads();
  for (int g = tid; g < a.groups; g += RT) {    // channels of a group in order
    float t1 = 0.f, t2 = 0.f;
    for (int j = 0; j < a.cpg; ++j) {
      const int c = g * a.cpg + j;
      const float w = BWD ? a.gamma[c] : 1.f;
      t1 += w * chan[c][0]; t2 += w * chan[c][1];
    }
    a.rows_out[(size_t)ch * a.groups + g] = make_float2(t1, t2);
  }
}

// fp16-storage inference apply: 8 channels (16 B) per thread, no dropout.  y = act(z) + r  or  act(z + r).
typedef _Float16 gn_half8 __attribute__((ext_vector_type(8)));
template <int ACT, bool RES_NORM = false>
__global__ __launch_bounds__(T) void gn_apply_f16x8_kernel(const GnArgs a) {
  __shared__ float tab[2048][2];  // scale = rstd*gamma, shift = beta - mean*rstd*gamma  (z = x*scale + shift)
  __shared__ float tabr[RES_NORM ? 2048 : 1][2];   // the same for the residual's own GroupNorm
  const int q = blockIdx.y, tid = threadIdx.x;
  const int s = seg_of_sample(a, q);
  const GnSeg& sg = a.seg[s];
  const int nl = q - sg.sample_start;
  const int C = a.c, C8 = C >> 3;
  if ((int64_t)blockIdx.x * T >= (int64_t)sg.hw * C8) return;  // the grid is sized for the largest segment
  for (int c = tid; c < C; c += T) {
    const int g = c / a.cpg;
    const float mean = sg.mean[nl * a.groups + g], rstd = sg.rstd[nl * a.groups + g];
    const float sc = rstd * a.gamma[c];
    tab[c][0] = sc;
    tab[c][1] = a.beta[c] - mean * sc;
    if (RES_NORM) {
      const int gr = c / (C / a.res_groups);
      const float scr = a.res_rstd[nl * a.res_groups + gr] * a.res_gamma[c];
      tabr[c][0] = scr;
      tabr[c][1] = a.res_beta[c] - a.res_mean[nl * a.res_groups + gr] * scr;
    }
  }
  __syncthreads();
  const size_t base = (size_t)nl * sg.hw * C;
  const _Float16* __restrict__ x = reinterpret_cast<const _Float16*>(sg.x) + base;
  const _Float16* __restrict__ r = sg.res ? reinterpret_cast<const _Float16*>(sg.res) + base : nullptr;
  _Float16* __restrict__ y = reinterpret_cast<_Float16*>(sg.y) + base;
  const int64_t total = (int64_t)sg.hw * C8;
  const bool aar = a.act_after_res != 0;
#pragma unroll 2
  for (int64_t i = (int64_t)blockIdx.x * T + tid; i < total; i += (int64_t)gridDim.x * T) {
    const int c0 = (int)(i % C8) * 8;
    const gn_half8* xp = reinterpret_cast<const gn_half8*>(x + i * 8);
    const gn_half8 xv = a.nt_loads ? __builtin_nontemporal_load(xp) : *xp;
    gn_half8 rv;
    if (r) {
      const gn_half8* rp = reinterpret_cast<const gn_half8*>(r + i * 8);
      rv = a.nt_loads ? __builtin_nontemporal_load(rp) : *rp;
    }
    gn_half8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float2 t = *reinterpret_cast<const float2*>(&tab[c0 + j][0]);
      const float z = (float)xv[j] * t.x + t.y;
      float rr = r ? (float)rv[j] : 0.f;
      if (RES_NORM) { const float2 tr = *reinterpret_cast<const float2*>(&tabr[c0 + j][0]); rr = rr * tr.x + tr.y; }
      const float v = aar ? rn::act_fwd(z + rr, ACT) : rn::act_fwd(z, ACT) + rr;
      o[j] = (_Float16)v;
    }
    *reinterpret_cast<gn_half8*>(y + i * 8) = o;
  }
}

// ---------------------------------------------------------------------------------------------
// Slice-resident path: when one (sample, group-block) slice fits the registers of a 512-thread block
// (hw * channels <= 16 K values: the GroupNorms at 1/16 resolution and below, about half of the calls) the whole GroupNorm is ONE kernel -- x is read once, the statistics are an exact two-pass
// mean / variance over registers, and y is written once -- instead of partial + finalize + apply (three
// launch-latency-bound kernels and a second read of x).  Backward likewise: x and dy read once, dx written
// once, per-sample (sum g, sum g*xhat) rows for the parameter gradients summed by a tiny second kernel.
// A block owns `wc` consecutive channels (gb whole groups, so that narrow groups still give >= 32-byte runs);
// thread t owns channel t % wc and pixels t / wc + k * ppt, k < R.  All reductions run in a fixed order.
// ---------------------------------------------------------------------------------------------
constexpr int ST = 512;  // 8 waves: 256 VGPRs per lane, so 32 values (+ 32 residuals / gradients) stay in registers
constexpr int SLICE_MAX_R = 32, SLICE_MAX_WC = 64, SLICE_J = ST / SLICE_MAX_WC;

// per-channel sums of NV values over the pixel lanes of the block -> out[cw * NV + i]
template <int NV>
__device__ __forceinline__ void slice_channel_sums(const float (&val)[NV], int tid, int wc, int ppt, bool active, float* red,
                                                   float* red2, float* out) {
#pragma unroll
  for (int i = 0; i < NV; ++i) red[tid * NV + i] = active ? val[i] : 0.f;
  __syncthreads();
  if (tid < wc * SLICE_J) {
    const int cw = tid % wc, j = tid / wc;
    float acc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] = 0.f;
    for (int pl = j; pl < ppt; pl += SLICE_J)
#pragma unroll
      for (int i = 0; i < NV; ++i) acc[i] += red[(pl * wc + cw) * NV + i];
#pragma unroll
    for (int i = 0; i < NV; ++i) red2[(j * SLICE_MAX_WC + cw) * NV + i] = acc[i];
  }
  __syncthreads();
  if (tid < wc) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      float t = 0.f;
#pragma unroll
      for (int j = 0; j < SLICE_J; ++j) t += red2[(j * SLICE_MAX_WC + tid) * NV + i];
      out[tid * NV + i] = t;
    }
  }
  __syncthreads();
}

template <int R, int ACT>
__global__ __launch_bounds__(ST) void gn_slice_fwd_kernel(const GnArgs a) {
  __shared__ float red[ST], red2[SLICE_J * SLICE_MAX_WC], csum[SLICE_MAX_WC], gstat[SLICE_MAX_WC];
  const int tid = threadIdx.x, wc = a.slice_wc, cpg = a.cpg, C = a.c;
  const int gb = wc / cpg, gblocks = a.groups / gb;
  const int q = blockIdx.x / gblocks, gbk = blockIdx.x - q * gblocks;
  const GnSeg& sg = a.seg[seg_of_sample(a, q)];
  const int nl = q - sg.sample_start, hw = sg.hw;
  const int ppt = ST / wc, cw = tid % wc, pl = tid / wc;
  const bool active = pl < ppt;
  const int c = gbk * wc + cw;
  const size_t base = (size_t)nl * hw * C;
  const float* __restrict__ x = sg.x + base;  // block-uniform base + 32-bit per-lane offsets (p * C + c)
  float v[R];
  float s1 = 0.f;
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const int p = pl + k * ppt;
    // branch-free: load from a clamped pixel, then zero the lanes past the end (a conditional load compiles to
    // a branch + wait per pixel and serialises the R round trips)
    const float t = x[(unsigned)(min(p, hw - 1) * C + c)];
    v[k] = t * ((active && p < hw) ? 1.f : 0.f);  // a multiply, not a select: the compiler sinks a selected load
    s1 += v[k];                                    // back under a branch

  }
  const float inv_m = 1.f / ((float)hw * (float)cpg);
  {
    const float val[1] = {s1};
    slice_channel_sums<1>(val, tid, wc, ppt, active, red, red2, csum);
  }
  if (tid < gb) {
    float t = 0.f;
    for (int j = 0; j < cpg; ++j) t += csum[tid * cpg + j];
    gstat[tid] = t * inv_m;
  }
  __syncthreads();
  const float mean = gstat[cw / cpg];
  float s2 = 0.f;
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const int p = pl + k * ppt;
    const float d = v[k] - mean;
    s2 += (active && p < hw) ? d * d : 0.f;
  }
  __syncthreads();  // gstat (means) read by everyone before it is overwritten below
  {
    const float val[1] = {s2};
    slice_channel_sums<1>(val, tid, wc, ppt, active, red, red2, csum);
  }
  if (tid < gb) {
    float t = 0.f;
    for (int j = 0; j < cpg; ++j) t += csum[tid * cpg + j];
    const float rstd = 1.f / sqrtf(t * inv_m + a.eps);
    const int g = gbk * gb + tid;
    sg.mean[nl * a.groups + g] = gstat[tid];
    sg.rstd[nl * a.groups + g] = rstd;
    gstat[tid] = rstd;
  }
  __syncthreads();
  if (!active) return;
  const float rstd = gstat[cw / cpg];
  const float sc = rstd * a.gamma[c], sh = a.beta[c] - mean * sc;
  const bool drop = a.drop_rate > 0.f;
  const float keep_scale = drop ? 1.f / (1.f - a.drop_rate) : 1.f;
  const uint64_t samp_off = (uint64_t)q * (uint64_t)hw * (uint64_t)C;
  const uint64_t seed = a.seed + (a.seed_dev ? *a.seed_dev : 0ull);
  const float* __restrict__ res = sg.res ? sg.res + base : nullptr;
  float* __restrict__ y = sg.y + base;
  float rr[R];
#pragma unroll
  for (int k = 0; k < R; ++k) rr[k] = 0.f;
  if (res) {  // all residual loads in flight together
#pragma unroll
    for (int k = 0; k < R; ++k) rr[k] = res[(unsigned)(min(pl + k * ppt, hw - 1) * C + c)];
  }
  const bool aar = a.act_after_res != 0;
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const int p = pl + k * ppt;
    const unsigned off = (unsigned)(p * C + c);
    const float z = v[k] * sc + sh;
    const float r = rr[k];
    float o = rn::act_fwd(aar ? z + r : z, ACT);
    if (drop) o = (rn::uniform01(seed, samp_off + off) >= a.drop_rate) ? o * keep_scale : 0.f;
    if (p < hw) y[off] = aar ? o : o + r;
  }
}

template <int R, int ACT>
__global__ __launch_bounds__(ST) void gn_slice_bwd_kernel(const GnArgs a) {
  __shared__ float red[ST * 2], red2[SLICE_J * SLICE_MAX_WC * 2], csum[SLICE_MAX_WC * 2], gcoef[SLICE_MAX_WC][2];
  const int tid = threadIdx.x, wc = a.slice_wc, cpg = a.cpg, C = a.c;
  const int gb = wc / cpg, gblocks = a.groups / gb;
  const int q = blockIdx.x / gblocks, gbk = blockIdx.x - q * gblocks;
  const GnSeg& sg = a.seg[seg_of_sample(a, q)];
  const int nl = q - sg.sample_start, hw = sg.hw;
  const int ppt = ST / wc, cw = tid % wc, pl = tid / wc;
  const bool active = pl < ppt;
  const int c = gbk * wc + cw;
  const int g = c / cpg;
  const size_t base = (size_t)nl * hw * C;
  const float mean = sg.mean[nl * a.groups + g], rstd = sg.rstd[nl * a.groups + g];
  const float gam = a.gamma[c], bet = a.beta[c];
  const bool aar = a.act_after_res && sg.res;
  const bool drop = a.drop_rate > 0.f;
  const float keep_scale = drop ? 1.f / (1.f - a.drop_rate) : 1.f;
  const uint64_t samp_off = (uint64_t)q * (uint64_t)hw * (uint64_t)C;
  const uint64_t seed = a.seed + (a.seed_dev ? *a.seed_dev : 0ull);
  const float* __restrict__ x = sg.x + base;  // block-uniform bases + 32-bit per-lane offsets (p * C + c)
  const float* __restrict__ dy = sg.dy + base;
  const float* __restrict__ res = aar ? sg.res + base : nullptr;
  // x / dy are loaded once and overwritten in place by xhat / g = dy * dropmask * act'(z)
  constexpr int RR = R;
  float xh[RR], gr[RR];
  auto grad_at = [&](float xv, float dyv, unsigned off, float& h) {
    h = (xv - mean) * rstd;
    float z = h * gam + bet;
    if (res) z += res[off];
    float gg = dyv;
    if (drop) gg = (rn::uniform01(seed, samp_off + off) >= a.drop_rate) ? gg * keep_scale : 0.f;
    return gg * rn::act_grad(z, ACT);
  };
  float sb = 0.f, sgm = 0.f;
  {
#pragma unroll
    for (int k = 0; k < RR; ++k) {  // clamped pixel, masked below: unconditional loads, all in flight together
      const unsigned off = (unsigned)(min(pl + k * ppt, hw - 1) * C + c);
      xh[k] = x[off];
      gr[k] = dy[off];
    }
#pragma unroll
    for (int k = 0; k < RR; ++k) {
      const int p = pl + k * ppt;
      const unsigned off = (unsigned)(min(p, hw - 1) * C + c);
      float h;
      float gg = grad_at(xh[k], gr[k], off, h);
      gg *= (active && p < hw) ? 1.f : 0.f;  // multiply, not select (see the forward kernel)
      xh[k] = h;
      gr[k] = gg;
      sb += gg; sgm += gg * h;
    }
  }
  {
    const float val[2] = {sb, sgm};
    slice_channel_sums<2>(val, tid, wc, ppt, active, red, red2, csum);
  }
  if (tid < wc) {  // this sample's contribution to dbeta / dgamma of channel gbk*wc + tid: two [samples][C] planes
    float* pg = a.pgrad + (size_t)q * C + gbk * wc + tid;
    pg[0] = csum[tid * 2];
    pg[(size_t)a.total_samples * C] = csum[tid * 2 + 1];
  }
  if (tid < gb) {
    float t1 = 0.f, t2 = 0.f;
    for (int j = 0; j < cpg; ++j) {
      const float gm = a.gamma[gbk * wc + tid * cpg + j];
      t1 += gm * csum[(tid * cpg + j) * 2];
      t2 += gm * csum[(tid * cpg + j) * 2 + 1];
    }
    const float inv_m = 1.f / ((float)hw * (float)cpg);
    gcoef[tid][0] = t1 * inv_m;
    gcoef[tid][1] = t2 * inv_m;
  }
  __syncthreads();
  if (!active) return;
  const float c1 = gcoef[cw / cpg][0], c2 = gcoef[cw / cpg][1];
  float* __restrict__ dx = sg.dx + base;
  float* __restrict__ dres = (aar && sg.dres) ? sg.dres + base : nullptr;
  {
#pragma unroll
    for (int k = 0; k < RR; ++k) {
      const int p = pl + k * ppt;
      if (p < hw) {
        const unsigned off = (unsigned)(p * C + c);
        dx[off] = rstd * (gam * gr[k] - c1 - xh[k] * c2);
        if (dres) dres[off] = gr[k];
      }
    }
  }
}

// dbeta_c / dgamma_c = sum over all samples (fixed order) of the per-sample rows
__global__ __launch_bounds__(256) void gn_param_grad_kernel(const GnArgs a) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= a.c) return;
  const float* pb = a.pgrad + c;
  const float* pgm = a.pgrad + (size_t)a.total_samples * a.c + c;
  float b = 0.f, g = 0.f;
  int q = 0;
  for (; q + 4 <= a.total_samples; q += 4) {  // eight loads in flight, added in sample order
    float vb[4], vg[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { vb[j] = pb[(size_t)(q + j) * a.c]; vg[j] = pgm[(size_t)(q + j) * a.c]; }
#pragma unroll
    for (int j = 0; j < 4; ++j) { b += vb[j]; g += vg[j]; }
  }
  for (; q < a.total_samples; ++q) { b += pb[(size_t)q * a.c]; g += pgm[(size_t)q * a.c]; }
  a.dbeta[c] = b;
  a.dgamma[c] = g;
}

// ---------------------------------------------------------------------------------------------
// Grid-resident path: maps too large for one block per slice but small enough for the registers of <= 128 co-resident
// blocks (<= 4 M elements: the head towers over P3..P7, the backbone at 1/8 resolution) are ONE kernel as well: every
// block loads its pixel range (all channels, full 16-byte coalesced rows) into registers, publishes its per-group
// partial sums, collects the sums of the other blocks of its sample, derives the statistics and applies them to the
// registers -- x (and dy) are read once, y (dx) written once, instead of partial + finalize + apply with a second
// read.  Co-residency: <= 128 blocks of 512 threads, one per CU; two such kernels (the two head streams) still fit the
// 256 CUs together, nothing else in the step spins.  The wait is bounded: if a sample's blocks are not all resident
// within ~10^6 polls it gives up (error word set, results of that call undefined) instead of hanging.
// ---------------------------------------------------------------------------------------------
constexpr int CT = 512, COOP_MAX_R = 16, COOP_MAX_BLOCKS = 128, COOP_MAX_C = 1024;

// The blocks of a sample exchange their per-group sums WITHOUT a barrier and without a cache write-back / invalidate:
// every sum travels as one 8-byte word {value, tag} written with a device-scope (write-through, sc1) store into a
// region only these kernels ever write (rn_gn_params.sync + 256 bytes); readers poll the words they need with
// device-scope loads until the tag is the current call's.  The tag is a device-side call counter (word 1) + 1: it is read
// at kernel start, and bumped by the last block to leave the kernel (word 0 counts the leavers and is reset by that
// block), so a replayed hipGraph and eager calls share it without host involvement.  Stale words always carry an
// older tag.  Critical path: store -> L2 -> load, instead of store-ack + arrive-atomic + poll + leave-atomic + load.
// The poll is bounded (~10^6 tries): if a sample's blocks are not all resident it gives up and sets the error word
// (word 2) instead of hanging.
constexpr size_t COOP_XCHG_OFFSET = 256;  // bytes from rn_gn_params.sync to the exchange rows
constexpr size_t COOP_XCHG_BYTES = 1 << 20;  // 256 blocks x 256 groups x 2 sums x 8 bytes
__device__ __forceinline__ unsigned coop_tag(const unsigned* sync) {
  return __hip_atomic_load(&sync[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
}
__device__ __forceinline__ void coop_leave(unsigned* sync, unsigned nblocks) {  // one thread per block, after its polls
  const unsigned d = __hip_atomic_fetch_add(&sync[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (d == nblocks - 1) {  // every block has read the tag and finished polling: next call gets a new one
    __hip_atomic_store(&sync[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(&sync[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// per-channel (sum v1, sum v2) over the block's pixels -> chan[c][0..1] in LDS, optionally the two partial planes (row
// `chunk`; the backward's dbeta / dgamma rows), and the per-group sums (weighted by gamma when WEIGHTED) -> tagged exchange words
template <bool WEIGHTED, bool PLANES>
__device__ __forceinline__ void coop_partials(const GnArgs& a, float (*red)[8], float (*chan)[2], const float (&s1)[4],
                                              const float (&s2)[4], int chunk, int QP, int lanes, int CQ, unsigned tag) {
  const int tid = threadIdx.x, C = a.c;
#pragma unroll
  for (int j = 0; j < 4; ++j) { red[tid][j] = s1[j]; red[tid][4 + j] = s2[j]; }
  __syncthreads();
  for (int e = tid; e < CQ * 8; e += CT) {
    const int q = e >> 3, comp = e & 7;
    float t = 0.f;
    for (int l = 0; l < lanes; ++l) t += red[l * QP + q][comp];
    const int c = q * 4 + (comp & 3);
    chan[c][comp >> 2] = t;
    if (PLANES) a.partial[(size_t)(comp >> 2) * a.total_chunks * C + (size_t)chunk * C + c] = t;
  }
  __syncthreads();
  for (int e = tid; e < a.groups * 2; e += CT) {
    const int g = e >> 1, comp = e & 1;
    float t = 0.f;
    for (int j = 0; j < a.cpg; ++j) {
      const int c = g * a.cpg + j;
      t += WEIGHTED ? a.gamma[c] * chan[c][comp] : chan[c][comp];
    }
    const unsigned long long word = ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(t);
    __hip_atomic_store(&a.xchg[((size_t)chunk * a.groups + g) * 2 + comp], word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// per-group totals of this block's sample over its `rows` chunks (fp64, fixed order) -> out[g][0..1]; polls the other
// blocks' tagged words (four in flight per thread).  Returns false if a word never arrived.
__device__ __forceinline__ bool coop_group_totals(const GnArgs& a, const unsigned long long* gp, int rows, unsigned tag,
                                                  double (*lane_sum)[2], double (*out)[2]) {
  __shared__ int fail_sh;
  const int tid = threadIdx.x, G2 = a.groups * 2;
  const int RL = CT / G2 >= 1 ? CT / G2 : 1;  // row lanes per (group, component)
  if (tid == 0) fail_sh = 0;
  if (G2 >= CT) __syncthreads();  // (the other branch has barriers of its own before fail_sh is written again)
  bool fail = false;
  for (int e0 = 0; e0 < G2; e0 += CT) {      // G2 > CT: several passes with one row lane
    const int e = e0 + (tid % (G2 < CT ? G2 : CT)), rl = tid / (G2 < CT ? G2 : CT);
    double t = 0.0;
    if (e < G2 && rl < RL) {
      for (int r0 = rl; r0 < rows; r0 += 4 * RL) {
        unsigned long long w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int r = min(r0 + j * RL, rows - 1);
          w[j] = __hip_atomic_load(gp + (size_t)r * G2 + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int r = r0 + j * RL;
          if (r < rows) {
            unsigned spins = 0;
            while ((unsigned)(w[j] >> 32) != tag) {
              if (++spins > (1u << 20)) { fail = true; break; }  // that block is not resident: give up instead of hanging
              __builtin_amdgcn_s_sleep(1);
              w[j] = __hip_atomic_load(gp + (size_t)r * G2 + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            t += (double)__uint_as_float((unsigned)w[j]);
          }
        }
      }
    }
    if (G2 < CT) {
      __syncthreads();
      lane_sum[tid][0] = t;
      __syncthreads();
      if (tid < G2) {
        double u = 0.0;
        for (int l = 0; l < RL; ++l) u += lane_sum[l * G2 + tid][0];
        out[tid >> 1][tid & 1] = u;
      }
    } else if (e < G2) {
      out[e >> 1][e & 1] = t;
    }
  }
  if (fail) fail_sh = 1;
  __syncthreads();
  const bool ok = fail_sh == 0;
  if (tid == 0) {
    if (!ok) atomicExch(&a.sync[2], 1u);
    coop_leave(a.sync, gridDim.x);
  }
  return ok;
}

template <int ACT, int COOP_R>
__global__ __launch_bounds__(CT) void gn_coop_fwd_kernel(const GnArgs a) {
  __shared__ float red[CT][8];
  __shared__ float chan[COOP_MAX_C][2];
  __shared__ double lane_sum[CT][2], gtot[COOP_MAX_C][2];
  __shared__ float gstat[COOP_MAX_C][2];
  const int tid = threadIdx.x, C = a.c, CQ = C >> 2, QP = CQ, lanes = CT / QP;
  const int ch = blockIdx.x;
  const GnSeg& sg = a.seg[seg_of_chunk(a, ch)];
  const int local = ch - sg.chunk_start;
  const int nl = local / sg.chunks, ck = local - nl * sg.chunks;
  const int q = sg.sample_start + nl;
  const int p_begin = ck * sg.ppc, p_end = min(p_begin + sg.ppc, sg.hw);
  const size_t base = (size_t)nl * sg.hw * C;
  const float* __restrict__ x = sg.x + base;
  const int q4 = tid % QP, pl = tid / QP;
  const bool active = pl < lanes;
  const unsigned tag = coop_tag(a.sync);  // in flight together with the data loads, like the parameters below
  float gam[4], bet[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { gam[j] = a.gamma[min(q4 * 4 + j, C - 1)]; bet[j] = a.beta[min(q4 * 4 + j, C - 1)]; }
  const uint64_t seed = a.seed + (a.seed_dev ? *a.seed_dev : 0ull);
  float4 v[COOP_R];
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < COOP_R; ++k) {
    const int p = p_begin + pl + k * lanes;
    const float m = (active && p < p_end) ? 1.f : 0.f;
    const float4 t = *reinterpret_cast<const float4*>(x + (size_t)min(p, sg.hw - 1) * C + q4 * 4);  // clamped, masked
    v[k] = make_float4(t.x * m, t.y * m, t.z * m, t.w * m);
    s1[0] += v[k].x; s1[1] += v[k].y; s1[2] += v[k].z; s1[3] += v[k].w;
    s2[0] += v[k].x * v[k].x; s2[1] += v[k].y * v[k].y; s2[2] += v[k].z * v[k].z; s2[3] += v[k].w * v[k].w;
  }
  if (!active) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
  }
  // the residual rows are fetched now (clamped addresses, all in flight) and wait in registers for the statistics
  const float* __restrict__ res = sg.res ? sg.res + base : nullptr;
  float4 rres[COOP_R];
  if (res) {
#pragma unroll
    for (int k = 0; k < COOP_R; ++k)
      rres[k] = *reinterpret_cast<const float4*>(res + (size_t)min(p_begin + pl + k * lanes, sg.hw - 1) * C + q4 * 4);
  } else {
#pragma unroll
    for (int k = 0; k < COOP_R; ++k) rres[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  coop_partials<false, false>(a, red, chan, s1, s2, ch, QP, lanes, CQ, tag);
  // statistics of this block's sample from the group rows of all its chunks (fp64, fixed order)
  const bool ok = coop_group_totals(a, a.xchg + (size_t)(sg.chunk_start + nl * sg.chunks) * a.groups * 2, sg.chunks, tag, lane_sum, gtot);
  for (int g = tid; g < a.groups; g += CT) {
    const double m = (double)sg.hw * (double)a.cpg;
    const double mean = gtot[g][0] / m;
    double var = gtot[g][1] / m - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)a.eps));
    gstat[g][0] = (float)mean; gstat[g][1] = rstd;
    if (ck == 0) { sg.mean[nl * a.groups + g] = (float)mean; sg.rstd[nl * a.groups + g] = rstd; }
  }
  __syncthreads();
  if (!active || !ok) return;
  float sc[4], sh[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = q4 * 4 + j, g = c / a.cpg;
    sc[j] = gstat[g][1] * gam[j];
    sh[j] = bet[j] - gstat[g][0] * sc[j];
  }
  const bool drop = a.drop_rate > 0.f;
  const float keep_scale = drop ? 1.f / (1.f - a.drop_rate) : 1.f;
  const uint64_t samp_off = (uint64_t)q * (uint64_t)sg.hw * (uint64_t)C;
  float* __restrict__ y = sg.y + base;
  const bool aar = a.act_after_res != 0;
#pragma unroll
  for (int k = 0; k < COOP_R; ++k) {
    const int p = p_begin + pl + k * lanes;
    if (p < p_end) {
      const size_t off = (size_t)p * C + q4 * 4;
      const float4 rv = rres[k];
      const float xs[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
      const float r[4] = {rv.x, rv.y, rv.z, rv.w};
      float o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float z = xs[j] * sc[j] + sh[j];
        float t = rn::act_fwd(aar ? z + r[j] : z, ACT);
        if (drop) t = (rn::uniform01(seed, samp_off + off + j) >= a.drop_rate) ? t * keep_scale : 0.f;
        o[j] = aar ? t : t + r[j];
      }
      *reinterpret_cast<float4*>(y + off) = make_float4(o[0], o[1], o[2], o[3]);
    }
  }
}

template <int ACT, int COOP_R>
__global__ __launch_bounds__(CT) void gn_coop_bwd_kernel(const GnArgs a) {
  __shared__ float red[CT][8];
  __shared__ float chan[COOP_MAX_C][2];
  __shared__ double lane_sum[CT][2], gtot[COOP_MAX_C][2];
  __shared__ float gcoef[COOP_MAX_C][2];
  const int tid = threadIdx.x, C = a.c, CQ = C >> 2, QP = CQ, lanes = CT / QP;
  const int ch = blockIdx.x;
  const GnSeg& sg = a.seg[seg_of_chunk(a, ch)];
  const int local = ch - sg.chunk_start;
  const int nl = local / sg.chunks, ck = local - nl * sg.chunks;
  const int q = sg.sample_start + nl;
  const int p_begin = ck * sg.ppc, p_end = min(p_begin + sg.ppc, sg.hw);
  const size_t base = (size_t)nl * sg.hw * C;
  const float* __restrict__ x = sg.x + base;
  const float* __restrict__ dy = sg.dy + base;
  const bool aar = a.act_after_res && sg.res;
  const float* __restrict__ res = aar ? sg.res + base : nullptr;
  const int q4 = tid % QP, pl = tid / QP;
  const bool active = pl < lanes;
  const unsigned tag = coop_tag(a.sync);
  float mean[4], rstd[4], gam[4], bet[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = min(q4 * 4 + j, C - 1), g = c / a.cpg;
    mean[j] = sg.mean[nl * a.groups + g]; rstd[j] = sg.rstd[nl * a.groups + g];
    gam[j] = a.gamma[c]; bet[j] = a.beta[c];
  }
  const bool drop = a.drop_rate > 0.f;
  const float keep_scale = drop ? 1.f / (1.f - a.drop_rate) : 1.f;
  const uint64_t samp_off = (uint64_t)q * (uint64_t)sg.hw * (uint64_t)C;
  const uint64_t seed = a.seed + (a.seed_dev ? *a.seed_dev : 0ull);
  float4 xh[COOP_R], gr[COOP_R];  // loaded as x / dy, overwritten by xhat / g = dy * dropmask * act'(z)
#pragma unroll
  for (int k = 0; k < COOP_R; ++k) {
    const size_t off = (size_t)min(p_begin + pl + k * lanes, sg.hw - 1) * C + q4 * 4;
    xh[k] = *reinterpret_cast<const float4*>(x + off);
    gr[k] = *reinterpret_cast<const float4*>(dy + off);
  }
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < COOP_R; ++k) {
    const int p = p_begin + pl + k * lanes;
    const float m = (active && p < p_end) ? 1.f : 0.f;
    const size_t off = (size_t)min(p, sg.hw - 1) * C + q4 * 4;
    float4 rv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (res) rv = *reinterpret_cast<const float4*>(res + off);
    const float xs[4] = {xh[k].x, xh[k].y, xh[k].z, xh[k].w};
    const float ds[4] = {gr[k].x, gr[k].y, gr[k].z, gr[k].w};
    const float rr[4] = {rv.x, rv.y, rv.z, rv.w};
    float h[4], g[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      h[j] = (xs[j] - mean[j]) * rstd[j];
      const float z = h[j] * gam[j] + bet[j] + rr[j];
      float t = ds[j];
      if (drop) t = (rn::uniform01(seed, samp_off + off + j) >= a.drop_rate) ? t * keep_scale : 0.f;
      g[j] = t * rn::act_grad(z, ACT) * m;
      s1[j] += g[j];
      s2[j] += g[j] * h[j];
    }
    xh[k] = make_float4(h[0], h[1], h[2], h[3]);
    gr[k] = make_float4(g[0], g[1], g[2], g[3]);
  }
  coop_partials<true, true>(a, red, chan, s1, s2, ch, QP, lanes, CQ, tag);
  const bool ok = coop_group_totals(a, a.xchg + (size_t)(sg.chunk_start + nl * sg.chunks) * a.groups * 2, sg.chunks, tag, lane_sum, gtot);
  for (int g = tid; g < a.groups; g += CT) {
    const double m = (double)sg.hw * (double)a.cpg;
    gcoef[g][0] = (float)(gtot[g][0] / m); gcoef[g][1] = (float)(gtot[g][1] / m);
  }
  __syncthreads();
  if (!active || !ok) return;
  float c1[4], c2[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int g = (q4 * 4 + j) / a.cpg;
    c1[j] = gcoef[g][0]; c2[j] = gcoef[g][1];
  }
  float* __restrict__ dx = sg.dx + base;
  float* __restrict__ dres = (aar && sg.dres) ? sg.dres + base : nullptr;
#pragma unroll
  for (int k = 0; k < COOP_R; ++k) {
    const int p = p_begin + pl + k * lanes;
    if (p < p_end) {
      const size_t off = (size_t)p * C + q4 * 4;
      const float hs[4] = {xh[k].x, xh[k].y, xh[k].z, xh[k].w};
      const float gs[4] = {gr[k].x, gr[k].y, gr[k].z, gr[k].w};
      float o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = rstd[j] * (gam[j] * gs[j] - c1[j] - hs[j] * c2[j]);
      *reinterpret_cast<float4*>(dx + off) = make_float4(o[0], o[1], o[2], o[3]);
      if (dres) *reinterpret_cast<float4*>(dres + off) = gr[k];
    }
  }
}

// chunking of the grid-resident path; false when the call does not qualify
bool plan_coop(GnArgs* a) {
  if (!a->sync || a->strided || getenv("RN_GN_NO_COOP")) return false;
  if (a->in_half || a->out_half || a->act == RN_ACT_SIGMOID) return false;
  const int CQ = a->c / 4;
  if (a->c > COOP_MAX_C || a->groups > COOP_MAX_C || CQ > CT) return false;
  const int lanes = CT / CQ;
  // smallest per-thread depth (most blocks, most parallelism) that keeps the grid within COOP_MAX_BLOCKS
  int r = 0, ppc = 0, chunks = 0;
  for (int cand = 4; cand <= COOP_MAX_R; cand *= 2) {
    ppc = lanes * cand;
    chunks = 0;
    for (int s = 0; s < a->nseg; ++s) chunks += a->seg[s].n * rn::ceil_div(a->seg[s].hw, ppc);
    // depth 4 needs ~80 VGPRs: three blocks fit a CU, so two concurrent kernels of 256 blocks are still co-resident;
    // deeper variants (up to 245 VGPRs, one block per CU) stay within 128 blocks each
    if (chunks <= (cand == 4 ? 2 * COOP_MAX_BLOCKS : COOP_MAX_BLOCKS)) { r = cand; break; }
  }
  if (!r) return false;
  if ((size_t)chunks * a->groups * 2 * sizeof(unsigned long long) > COOP_XCHG_BYTES) return false;  // rows must fit the sync region
  a->coop_r = r;
  chunks = 0;
  for (int s = 0; s < a->nseg; ++s) {
    GnSeg& d = a->seg[s];
    d.ppc = ppc;
    d.chunks = rn::ceil_div(d.hw, ppc);
    d.chunk_start = chunks;
    chunks += d.n * d.chunks;
  }
  a->total_chunks = chunks;
  a->coop_ppc = ppc;
  return true;
}

template <bool BWD>
void launch_coop(const GnArgs& a, hipStream_t st) {
#define RN_GN_COOP2(ACT_, R_)                                                                                   \
  do {                                                                                                          \
    if (BWD) hipLaunchKernelGGL((gn_coop_bwd_kernel<ACT_, R_>), dim3(a.total_chunks), dim3(CT), 0, st, a);      \
    else hipLaunchKernelGGL((gn_coop_fwd_kernel<ACT_, R_>), dim3(a.total_chunks), dim3(CT), 0, st, a);          \
  } while (0)
#define RN_GN_COOP(ACT_)                                                                                       \
  do {                                                                                                          \
    if (a.coop_r == 4) RN_GN_COOP2(ACT_, 4);                                                                    \
    else if (a.coop_r == 8) RN_GN_COOP2(ACT_, 8);                                                               \
    else RN_GN_COOP2(ACT_, 16);                                                                                 \
  } while (0)
  switch (a.act) {
    case RN_ACT_RELU: RN_GN_COOP(RN_ACT_RELU); break;
    case RN_ACT_ELU: RN_GN_COOP(RN_ACT_ELU); break;
    case RN_ACT_RELU6: RN_GN_COOP(RN_ACT_RELU6); break;
    default: RN_GN_COOP(RN_ACT_NONE); break;
  }
#undef RN_GN_COOP2
#undef RN_GN_COOP
}

// choose the block width of the slice-resident path; returns R (pixels per thread) or 0 when a slice does not fit
int plan_slices(GnArgs* a) {
  a->slice_wc = 0;
  if (a->strided || getenv("RN_GN_NO_SLICE")) return 0;  // (tuning aid: force the three-kernel path)
  if (a->in_half || a->out_half || a->cpg > SLICE_MAX_WC || a->act == RN_ACT_SIGMOID) return 0;
  int max_hw = 0;
  for (int s = 0; s < a->nseg; ++s) max_hw = max_hw > a->seg[s].hw ? max_hw : a->seg[s].hw;
  // gb whole groups per block: the widest run of channels (up to a 128-byte line) that still fits the registers --
  // a block that uses only part of every line it touches multiplies its L2 traffic
  int min_wc = 8;
  if (const char* force = getenv("RN_GN_SLICE_WC")) min_wc = atoi(force);  // tuning aid
  int best = 0;
  for (int gb = 1; gb <= a->groups; ++gb) {
    if (a->groups % gb) continue;
    const int wc = gb * a->cpg;
    if (wc > SLICE_MAX_WC) break;
    if ((long)max_hw > (long)SLICE_MAX_R * (ST / wc)) break;
    best = wc;
    if (wc >= min_wc) break;
  }
  if (!best) return 0;
  a->slice_wc = best;
  const int need = rn::ceil_div(max_hw, ST / best);
  int r = 1;
  while (r < need) r *= 2;
  return r;
}

// The slice kernel reads runs of slice_wc channels: fine for small maps and for runs of >= 64 bytes; larger maps with
// narrow groups (ResNeXt's per-channel GroupNorm, MobileNet's 1..3-channel groups) go to the grid-resident kernel,
// which reads whole rows.
bool slice_preferred(const GnArgs& a) {
  long elems = 0;
  for (int s = 0; s < a.nseg; ++s) elems += (long)a.seg[s].n * a.seg[s].hw * a.c;
  static const long big = getenv("RN_GN_SLICE_BIG") ? atol(getenv("RN_GN_SLICE_BIG")) : 262144;
  return a.slice_wc >= 16 || (a.slice_wc >= 8 && elems < big);
}

// After the grid-resident path has declined (map too large): a slice kernel that reads runs narrower than 32 bytes
// (ResNeXt's per-channel GroupNorm: 4 bytes of every 128-byte line) only pays while the whole tensor is small and
// launch latency is what counts; past that the three full-row passes are 2x faster (measured at 2 x 100 x 100 x 256).
bool narrow_slice_ok(const GnArgs& a) {
  if (a.slice_wc >= 8) return true;
  long elems = 0;
  for (int s = 0; s < a.nseg; ++s) elems += (long)a.seg[s].n * a.seg[s].hw * a.c;
  return elems * (8 / a.slice_wc) <= (1L << 21);
}

template <bool BWD, int ACT>
void launch_slices_act(const GnArgs& a, int r, hipStream_t st) {
  const unsigned blocks = (unsigned)(a.total_samples * (a.groups / (a.slice_wc / a.cpg)));
#define RN_GN_SLICE(R_)                                                                                  \
  do {                                                                                                   \
    if (BWD) hipLaunchKernelGGL((gn_slice_bwd_kernel<R_, ACT>), dim3(blocks), dim3(ST), 0, st, a);       \
    else hipLaunchKernelGGL((gn_slice_fwd_kernel<R_, ACT>), dim3(blocks), dim3(ST), 0, st, a);           \
  } while (0)
  switch (r) {
    case 1: case 2: RN_GN_SLICE(2); break;
    case 4: RN_GN_SLICE(4); break;
    case 8: RN_GN_SLICE(8); break;
    case 16: RN_GN_SLICE(16); break;
    default: RN_GN_SLICE(32); break;
  }
#undef RN_GN_SLICE
}
template <bool BWD>
void launch_slices(const GnArgs& a, int r, hipStream_t st) {
  switch (a.act) {  // the activation is a compile-time constant of the slice kernels (no per-element switch)
    case RN_ACT_RELU: launch_slices_act<BWD, RN_ACT_RELU>(a, r, st); break;
    case RN_ACT_ELU: launch_slices_act<BWD, RN_ACT_ELU>(a, r, st); break;
    case RN_ACT_RELU6: launch_slices_act<BWD, RN_ACT_RELU6>(a, r, st); break;
    default: launch_slices_act<BWD, RN_ACT_NONE>(a, r, st); break;
  }
}

int build_args(const rn_gn_seg* segs, int nseg, const rn_gn_params* p, GnArgs* a, bool bwd) {
  RN_CHECK_ARG(segs && p, "group_norm: null argument");
  RN_CHECK_ARG(nseg >= 1 && nseg <= RN_MAX_SEG, "group_norm: nseg %d outside [1,%d]", nseg, RN_MAX_SEG);
  RN_CHECK_ARG(p->c >= 4 && p->groups >= 1 && p->c % p->groups == 0, "group_norm: c=%d groups=%d", p->c, p->groups);
  RN_UNSUPPORTED(p->c % 4 != 0 || p->c > 2048, "group_norm: c=%d must be a multiple of 4 and <= 2048", p->c);
  RN_CHECK_ARG(p->drop_rate >= 0.f && p->drop_rate < 1.f, "group_norm: drop_rate %f", p->drop_rate);
  a->nseg = nseg; a->c = p->c; a->groups = p->groups; a->cpg = p->c / p->groups; a->act = p->act;
  a->act_after_res = p->act_after_residual ? 1 : 0;
  a->in_half = (!bwd && p->in_f16) ? 1 : 0;
  a->out_half = (!bwd && p->out_f16) ? 1 : 0;
  RN_UNSUPPORTED(bwd && (p->in_f16 || p->out_f16), "group_norm bwd: fp16 storage is forward-only");
  a->eps = p->eps; a->drop_rate = p->drop_rate; a->seed = p->drop_seed; a->seed_dev = p->drop_seed_dev;
  a->sync = (unsigned*)p->sync;
  int samples = 0, chunks = 0;
  for (int s = 0; s < nseg; ++s) {
    RN_CHECK_ARG(segs[s].x && segs[s].mean && segs[s].rstd && segs[s].n >= 1 && segs[s].hw >= 1,
                 "group_norm: bad segment %d", s);
    if (bwd) RN_CHECK_ARG(segs[s].dy && segs[s].dx, "group_norm bwd: null dy/dx in segment %d", s);
    else RN_CHECK_ARG(segs[s].y, "group_norm fwd: null y in segment %d", s);
    GnSeg& d = a->seg[s];
    d.x = segs[s].x; d.y = segs[s].y; d.res = segs[s].residual; d.dy = segs[s].dy; d.dx = segs[s].dx;
    d.dres = segs[s].dresidual;
    d.mean = segs[s].mean; d.rstd = segs[s].rstd; d.n = segs[s].n; d.hw = segs[s].hw;
    d.x_ld = segs[s].x_ld > 0 ? segs[s].x_ld : p->c;
    d.dx_ld = segs[s].dx_ld > 0 ? segs[s].dx_ld : p->c;
    d.dx_acc = segs[s].dx_accumulate ? 1 : 0;
    RN_CHECK_ARG(d.x_ld >= p->c && d.x_ld % 4 == 0 && d.dx_ld >= p->c && d.dx_ld % 4 == 0, "group_norm: bad x_ld / dx_ld in segment %d", s);
    RN_UNSUPPORTED((d.x_ld != p->c || d.dx_ld != p->c || d.dx_acc) && (p->in_f16 || p->out_f16), "group_norm: strided views are fp32 only");
    if (d.x_ld != p->c || d.dx_ld != p->c || d.dx_acc) a->strided = 1;
    d.sample_start = samples; d.chunk_start = chunks;
    long elems = (long)d.hw * p->c;
    const long per = elems >= (4L << 20) ? 16384 : 4096;  // small tensors are latency-bound: more, shorter blocks
    int ck = (int)((elems + per - 1) / per);
    if (ck > 256) ck = 256;
    if (ck > d.hw) ck = d.hw;
    if (ck < 1) ck = 1;
    d.ppc = rn::ceil_div(d.hw, ck);
    d.chunks = rn::ceil_div(d.hw, d.ppc);
    samples += d.n;
    chunks += d.n * d.chunks;
  }
  a->total_samples = samples; a->total_chunks = chunks;
  return RN_OK;
}

size_t ws_bytes(const GnArgs& a) {
  const size_t rows = (size_t)a.total_samples > (size_t)2 * COOP_MAX_BLOCKS ? a.total_samples : 2 * COOP_MAX_BLOCKS;
  return rn::align_up((size_t)a.total_chunks * a.c * 2 * sizeof(float), 256) +
         rn::align_up(rows * a.groups * 2 * sizeof(float), 256) +  // bwd coefficients / grid-resident group rows
         rn::align_up((size_t)a.total_chunks * a.groups * 2 * sizeof(float), 256);  // rows path: per-chunk group rows
}
// the rows path's group rows: the third area of the workspace
float2* rows_area(const GnArgs& a, void* workspace) {
  const size_t rows = (size_t)a.total_samples > (size_t)2 * COOP_MAX_BLOCKS ? a.total_samples : 2 * COOP_MAX_BLOCKS;
  return (float2*)((char*)workspace + rn::align_up((size_t)a.total_chunks * a.c * 2 * sizeof(float), 256) +
                   rn::align_up(rows * a.groups * 2 * sizeof(float), 256));
}

// blocks per sample of the apply pass (grid.y = samples).  Every block first builds the per-channel scale / shift table
// (a ~1.5 us dependent chain), so blocks get `per` vectors per thread: 4 when the launch is small (latency-bound: more
// blocks), 16 once there are already >= 8 blocks per CU without it (bandwidth-bound: amortise the table).
unsigned apply_blocks(const GnArgs& a, int vec = 4) {
  long mx = 0, all = 0;
  for (int s = 0; s < a.nseg; ++s) {
    const long v = (long)a.seg[s].hw * (a.c / vec);
    mx = mx > v ? mx : v;
    all += v * a.seg[s].n;
  }
  const long per = all >= 16L * 2048 * T ? 16 : 4;
  long b = (mx + T * per - 1) / (T * per);
  if (b < 1) b = 1;
  if (b > 4096) b = 4096;
  return (unsigned)b;
}

}  // namespace

extern "C" size_t rn_group_norm_sync_bytes(void) {
  return COOP_XCHG_OFFSET + COOP_XCHG_BYTES;
}

extern "C" size_t rn_group_norm_workspace(const rn_gn_seg* segs, int nseg, const rn_gn_params* p) {
  GnArgs a = {};
  // workspace query tolerates null data pointers
  if (!segs || !p || nseg < 1 || nseg > RN_MAX_SEG || p->c < 4 || p->groups < 1) return 0;
  int samples = 0, chunks = 0;
  for (int s = 0; s < nseg; ++s) {
    long elems = (long)segs[s].hw * p->c;
    const long per = elems >= (4L << 20) ? 16384 : 4096;
    int ck = (int)((elems + per - 1) / per);
    if (ck > 256) ck = 256;
    if (ck > segs[s].hw) ck = segs[s].hw;
    if (ck < 1) ck = 1;
    int ppc = rn::ceil_div(segs[s].hw, ck);
    samples += segs[s].n;
    chunks += segs[s].n * rn::ceil_div(segs[s].hw, ppc);
  }
  a.total_chunks = chunks; a.total_samples = samples; a.c = p->c; a.groups = p->groups;
  return ws_bytes(a);
}

namespace {
// channel slab of gn_apply_rows_kernel: a whole number of groups and of float4 quads that tiles c, the narrowest one of at
// least 32 channels (128-byte runs per pixel), or all of c
int rows_slab(int c, int cpg) {
  int unit = cpg;
  while (unit % 4) unit += cpg;   // lcm(cpg, 4)
  int best = c;
  for (int sw = unit; sw <= c; sw += unit)
    if (c % sw == 0 && sw >= 32) { best = sw; break; }
  return best;
}
}  // namespace

extern "C" int rn_group_norm_rows_ok(int c, int groups, int rows_per_sample, int per_group) {
  if (c < 4 || c % 4 || groups < 1 || c % groups || rows_per_sample < 1) return 0;
  const int cpg = c / groups, sw = rows_slab(c, cpg);
  if (sw > ROWS_MAX_SLAB || sw / 4 > AT) return 0;
  const long entries = (long)rows_per_sample * (per_group ? sw / cpg : sw);   // a block's share of the sample's rows
  return (entries <= ROWS_MAX_ENTRIES && (per_group ? sw / cpg : sw) <= AT) ? 1 : 0;
}

namespace {
// the two-kernel rows path (gn_rows_partial_kernel + an apply-rows kernel) for this call: dense fp32, every segment's
// chunk rows mergeable
bool rows_path_ok(const GnArgs& a) {
  if (getenv("RN_GN_NO_ROWS")) return false;   // (tuning aid)
  if (a.in_half || a.out_half || a.act == RN_ACT_SIGMOID || a.c / 4 > RT) return false;   // (channel-prefix views are fine)
  for (int s = 0; s < a.nseg; ++s)
    if (!rn_group_norm_rows_ok(a.c, a.groups, a.seg[s].chunks, 1)) return false;
  return true;
}
template <bool BWD>
void launch_apply_rows(const GnArgs& a, hipStream_t st) {
  int max_hw = 0, max_chunks = 0;
  for (int s = 0; s < a.nseg; ++s) {
    max_hw = max_hw > a.seg[s].hw ? max_hw : a.seg[s].hw;
    max_chunks = max_chunks > a.seg[s].chunks ? max_chunks : a.seg[s].chunks;
  }
  const int lanes = AT / (a.rows_sw / 4);
  // every pixel chunk of a slab merges the same rows again (L2 hits): few chunks where that share is large
  const long entries = (long)max_chunks * (a.rows_per_group ? a.rows_sw / a.cpg : a.rows_sw);
  const int chunk_cap = entries <= 2048 ? 4096 : 48;
  int R = 4;
  while (R < (BWD ? 8 : 16) && rn::ceil_div(max_hw, lanes * R) > chunk_cap) R *= 2;   // (backward holds x, dy and the residual)
  const dim3 grid((unsigned)rn::ceil_div(max_hw, lanes * R), (unsigned)(a.c / a.rows_sw), (unsigned)a.total_samples);
#define RN_GN_ROWS2(ACT_, R_)                                                                                \
  do {                                                                                                       \
    if (BWD) hipLaunchKernelGGL((gn_bwd_apply_rows_kernel<ACT_, R_>), grid, dim3(AT), 0, st, a);             \
    else hipLaunchKernelGGL((gn_apply_rows_kernel<ACT_, R_>), grid, dim3(AT), 0, st, a);                     \
  } while (0)
#define RN_GN_ROWS(ACT_)                                                       \
  do {                                                                         \
    if (R == 4) RN_GN_ROWS2(ACT_, 4);                                          \
    else if (R == 8 || BWD) RN_GN_ROWS2(ACT_, 8);                              \
    else hipLaunchKernelGGL((gn_apply_rows_kernel<ACT_, 16>), grid, dim3(AT), 0, st, a); \
  } while (0)
  switch (a.act) {
    case RN_ACT_RELU: RN_GN_ROWS(RN_ACT_RELU); break;
    case RN_ACT_ELU: RN_GN_ROWS(RN_ACT_ELU); break;
    case RN_ACT_RELU6: RN_GN_ROWS(RN_ACT_RELU6); break;
    default: RN_GN_ROWS(RN_ACT_NONE); break;
  }
#undef RN_GN_ROWS
#undef RN_GN_ROWS2
}
template <bool BWD>
void launch_rows_partial(const GnArgs& a, hipStream_t st) {
  switch (a.act) {
    case RN_ACT_RELU: hipLaunchKernelGGL((gn_rows_partial_kernel<BWD, RN_ACT_RELU>), dim3(a.total_chunks), dim3(RT), 0, st, a); break;
    case RN_ACT_ELU: hipLaunchKernelGGL((gn_rows_partial_kernel<BWD, RN_ACT_ELU>), dim3(a.total_chunks), dim3(RT), 0, st, a); break;
    case RN_ACT_RELU6: hipLaunchKernelGGL((gn_rows_partial_kernel<BWD, RN_ACT_RELU6>), dim3(a.total_chunks), dim3(RT), 0, st, a); break;
    default: hipLaunchKernelGGL((gn_rows_partial_kernel<BWD, RN_ACT_NONE>), dim3(a.total_chunks), dim3(RT), 0, st, a); break;
  }
}
// Where both qualify, the grid-resident kernel goes first: one launch instead of two, measured 2 % faster on the headline
// step (373 vs 366 images/s).  RN_GN_ROWS_FIRST=1 (or no rn_gn_params.sync region) puts the rows path in front: then no
// kernel of a step ever waits for another block.
bool rows_first() {
  static const bool v = getenv("RN_GN_ROWS_FIRST") && atoi(getenv("RN_GN_ROWS_FIRST")) != 0;
  return v;
}
}  // namespace

extern "C" int rn_group_norm_fwd(const rn_gn_seg* segs, int nseg, const rn_gn_params* p, const float* gamma,
                                 const float* beta, void* workspace, size_t workspace_bytes, rn_stream_t stream) {
  GnArgs a = {};
  if (int e = build_args(segs, nseg, p, &a, false)) return e;
  RN_CHECK_ARG(gamma && beta && workspace, "group_norm fwd: null gamma/beta/workspace");
  if (workspace_bytes < ws_bytes(a)) {
    rn::set_error("group_norm fwd: workspace %zu < %zu", workspace_bytes, ws_bytes(a));
    return RN_EWORKSPACE;
  }
  a.gamma = gamma; a.beta = beta; a.partial = (float*)workspace;
  hipStream_t st = (hipStream_t)stream;
  if (p->stat_rows && p->stat_rows->rows && a.nseg == 1 && !a.in_half && !a.out_half && !a.strided && a.act != RN_ACT_SIGMOID &&
      p->stat_rows->groups == a.groups && rn_group_norm_rows_ok(a.c, a.groups, p->stat_rows->rows_per_sample, p->stat_rows->per_group)) {
    // x came with partial-sum rows from its producer: merge them and apply in one pass
    a.rows = (const float2*)p->stat_rows->rows; a.rows_per_group = p->stat_rows->per_group;
    a.seg[0].chunks = p->stat_rows->rows_per_sample; a.seg[0].chunk_start = 0;
    a.rows_sw = rows_slab(a.c, a.cpg);
    launch_apply_rows<false>(a, st);
    RN_LAUNCH_CHECK();
    return RN_OK;
  }
  // slice-resident kernel when its blocks read runs of >= 32 bytes; else the grid-resident kernel (full rows); else
  // the narrow slice kernel; else three kernels
  const int r = plan_slices(&a);
  if (r && slice_preferred(a)) {
    launch_slices<false>(a, r, st);
    RN_LAUNCH_CHECK();
    return RN_OK;
  }
  const bool rows_ok = rows_path_ok(a);
  auto rows_path = [&]() {   // statistics rows per chunk, then merge + apply: two kernels, nothing waits on another block
    a.rows_out = rows_area(a, workspace); a.rows = a.rows_out; a.rows_per_group = 1; a.rows_sw = rows_slab(a.c, a.cpg);
    launch_rows_partial<false>(a, st);
    launch_apply_rows<false>(a, st);
  };
  if (rows_ok && rows_first()) {
    rows_path();
    RN_LAUNCH_CHECK();
    return RN_OK;
  }
  {
    GnArgs c = a;  // the chunking differs from the three-kernel path's; the workspace (chunks <= 128 rows) always fits
    if (plan_coop(&c) && ws_bytes(c) <= workspace_bytes) {
      c.xchg = (unsigned long long*)((char*)c.sync + COOP_XCHG_OFFSET);
      launch_coop<false>(c, st);
      RN_LAUNCH_CHECK();
      return RN_OK;
    }
  }
  if (r && narrow_slice_ok(a)) {
    launch_slices<false>(a, r, st);
    RN_LAUNCH_CHECK();
    return RN_OK;
  }
  if (rows_ok) {
    rows_path();
    RN_LAUNCH_CHECK();
    return RN_OK;
  }
  {
    static const bool f16_stats = !(getenv("RN_GN_F16_STATS") && atoi(getenv("RN_GN_F16_STATS")) == 0);     // (0: the generic kernels; A/B)
    const int c8 = a.c >> 3;
    if (f16_stats && a.in_half && !a.strided && a.c % 8 == 0 && c8 <= T && T % c8 == 0)
      hipLaunchKernelGGL(gn_partial_f16x8_kernel, dim3(a.total_chunks), dim3(T), 0, st, a);
    else
      hipLaunchKernelGGL(gn_partial_kernel<false>, dim3(a.total_chunks), dim3(T), 0, st, a);
    if (f16_stats && a.c % 64 == 0 && a.cpg <= 64 && 64 % a.cpg == 0)
      hipLaunchKernelGGL(gn_finalize_cols_segs_kernel, dim3(a.total_samples * (a.c / 64)), dim3(T), 0, st, a);
    else
      hipLaunchKernelGGL(gn_finalize_kernel<false>, dim3(a.total_samples * a.groups), dim3(T), 0, st, a);
  }
  if (a.in_half && a.out_half && a.c % 8 == 0 && a.drop_rate == 0.f)
    switch (a.act) {  // activation as a template parameter: no per-element switch in a bandwidth-bound pass
      case RN_ACT_RELU: hipLaunchKernelGGL(gn_apply_f16x8_kernel<RN_ACT_RELU>, dim3(apply_blocks(a, 8), a.total_samples), dim3(T), 0, st, a); break;
      case RN_ACT_ELU: hipLaunchKernelGGL(gn_apply_f16x8_kernel<RN_ACT_ELU>, dim3(apply_blocks(a, 8), a.total_samples), dim3(T), 0, st, a); break;
      case RN_ACT_RELU6: hipLaunchKernelGGL(gn_apply_f16x8_kernel<RN_ACT_RELU6>, dim3(apply_blocks(a, 8), a.total_samples), dim3(T), 0, st, a); break;
      case RN_ACT_SIGMOID: hipLaunchKernelGGL(gn_apply_f16x8_kernel<RN_ACT_SIGMOID>, dim3(apply_blocks(a, 8), a.total_samples), dim3(T), 0, st, a); break;
      default: hipLaunchKernelGGL(gn_apply_f16x8_kernel<RN_ACT_NONE>, dim3(apply_blocks(a, 8), a.total_samples), dim3(T), 0, st, a); break;
    }
  else
    hipLaunchKernelGGL(gn_apply_kernel<false>, dim3(apply_blocks(a), a.total_samples), dim3(T), 0, st, a);
  RN_LAUNCH_CHECK();
  return RN_OK;
}

// fp16 inference, several segments: GroupNorm + activation of conv outputs whose statistics came out of the conv's epilogue
// (rn_conv2d_fwd_f16_fold with seg_chunk_start: per (m-tile, channel) sums in `partial` [2][total rows][c], segment s at the rows
// behind those of segments 0 .. s-1, n_s * tiles_per_sample[s] of them) -- the statistics pass over the tensors is not run; a
// segment with tiles_per_sample[s] == 0 (a small map whose conv tiles straddle samples) is summed from its tensor by the finalise
// blocks themselves.  Two launches: finalise, apply.
extern "C" int rn_group_norm_fwd_f16_tiles(const rn_gn_seg* segs, int nseg, const rn_gn_params* p, const float* gamma, const float* beta,
                                           const float* partial, const int32_t* tiles_per_sample, rn_stream_t stream) {
  GnArgs a = {};
  if (int e = build_args(segs, nseg, p, &a, false)) return e;
  RN_CHECK_ARG(gamma && beta && partial && tiles_per_sample, "group_norm f16 tiles: null argument");
  RN_UNSUPPORTED(!(a.in_half && a.out_half) || a.strided || a.c % 64 != 0 || a.cpg > 64 || 64 % a.cpg != 0 || a.drop_rate != 0.f,
                 "group_norm f16 tiles: fp16 in and out, dense, c %% 64 == 0, whole groups per 64 channels, no dropout");
  int chunks = 0;
  for (int s = 0; s < nseg; ++s) {
    GnSeg& d = a.seg[s];
    const int t = tiles_per_sample[s];
    RN_CHECK_ARG(t >= 0 && (t == 0 || d.hw % t == 0), "group_norm f16 tiles: segment %d: %d tiles for %d pixels", s, t, d.hw);
    RN_UNSUPPORTED(t == 0 && d.hw > 4096, "group_norm f16 tiles: segment %d has no rows and %d pixels per sample", s, d.hw);
    d.chunks = t; d.ppc = t ? d.hw / t : d.hw; d.chunk_start = chunks;
    chunks += d.n * t;
  }
  a.total_chunks = chunks;
  a.gamma = gamma; a.beta = beta; a.partial = const_cast<float*>(partial);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(gn_finalize_cols_segs_kernel, dim3(a.total_samples * (a.c / 64)), dim3(T), 0, st, a);
  switch (a.act) {
    case RN_ACT_RELU: hipLaunchKernelGGL(gn_apply_f16x8_kernel<RN_ACT_RELU>, dim3(apply_blocks(a, 8), a.total_samples), dim3(T), 0, st, a); break;
    case RN_ACT_ELU: hipLaunchKernelGGL(gn_apply_f16x8_kernel<RN_ACT_ELU>, dim3(apply_blocks(a, 8), a.total_samples), dim3(T), 0, st, a); break;
    case RN_ACT_RELU6: hipLaunchKernelGGL(gn_apply_f16x8_kernel<RN_ACT_RELU6>, dim3(apply_blocks(a, 8), a.total_samples), dim3(T), 0, st, a); break;
    case RN_ACT_SIGMOID: hipLaunchKernelGGL(gn_apply_f16x8_kernel<RN_ACT_SIGMOID>, dim3(apply_blocks(a, 8), a.total_samples), dim3(T), 0, st, a); break;
    default: hipLaunchKernelGGL(gn_apply_f16x8_kernel<RN_ACT_NONE>, dim3(apply_blocks(a, 8), a.total_samples), dim3(T), 0, st, a); break;
  }
  RN_LAUNCH_CHECK();
  return RN_OK;
}

extern "C" int rn_group_norm_bwd(const rn_gn_seg* segs, int nseg, const rn_gn_params* p, const float* gamma,
                                 const float* beta, float* dgamma, float* dbeta, void* workspace,
                                 size_t workspace_bytes, rn_stream_t stream, rn_reduce_list* defer) {
  rn::DeferScope defer_scope_(defer);
  GnArgs a = {};
  if (int e = build_args(segs, nseg, p, &a, true)) return e;
  RN_CHECK_ARG(gamma && beta && dgamma && dbeta && workspace, "group_norm bwd: null argument");
  if (workspace_bytes < ws_bytes(a)) {
    rn::set_error("group_norm bwd: workspace %zu < %zu", workspace_bytes, ws_bytes(a));
    return RN_EWORKSPACE;
  }
  a.gamma = gamma; a.beta = beta; a.dgamma = dgamma; a.dbeta = dbeta;
  a.partial = (float*)workspace;
  a.coef = (float*)((char*)workspace + rn::align_up((size_t)a.total_chunks * a.c * 2 * sizeof(float), 256));
  hipStream_t st = (hipStream_t)stream;
  int r = plan_slices(&a);
  const bool rows_ok = rows_path_ok(a);
  auto rows_path = [&]() -> int {   // (sum g, sum g xhat) rows per chunk, then merge + apply; parameter gradients = column sums
    a.rows_out = rows_area(a, workspace); a.rows = a.rows_out; a.rows_per_group = 1; a.rows_sw = rows_slab(a.c, a.cpg);
    launch_rows_partial<true>(a, st);
    launch_apply_rows<true>(a, st);
    RN_LAUNCH_CHECK();
    if (int e = rn::launch_reduce_rows(a.partial, dbeta, a.c, a.total_chunks, 0, st)) return e;
    return rn::launch_reduce_rows(a.partial + (size_t)a.total_chunks * a.c, dgamma, a.c, a.total_chunks, 0, st);
  };
  if (rows_ok && rows_first() && !(r && slice_preferred(a))) return rows_path();
  bool coop_first = r && !slice_preferred(a);
  if (coop_first) {
    GnArgs c = a;
    coop_first = plan_coop(&c) && ws_bytes(c) <= workspace_bytes;
    if (!coop_first && !narrow_slice_ok(a)) r = 0;  // neither: three kernels
  }
  if (r && !coop_first) {
    a.pgrad = (float*)workspace;  // [2][total_samples][c] <= the chunk-partial area (chunks >= samples)
    launch_slices<true>(a, r, st);
    if (rn::reduce_deferred(st)) {  // the two row sums join the step's single deferred reduction launch
      RN_LAUNCH_CHECK();
      if (int e = rn::launch_reduce_rows(a.pgrad, dbeta, a.c, a.total_samples, 0, st)) return e;
      return rn::launch_reduce_rows(a.pgrad + (size_t)a.total_samples * a.c, dgamma, a.c, a.total_samples, 0, st);
    }
    hipLaunchKernelGGL(gn_param_grad_kernel, dim3(rn::ceil_div(a.c, 256)), dim3(256), 0, st, a);
    RN_LAUNCH_CHECK();
    return RN_OK;
  }
  {
    GnArgs c = a;
    if (plan_coop(&c) && ws_bytes(c) <= workspace_bytes) {
      c.xchg = (unsigned long long*)((char*)c.sync + COOP_XCHG_OFFSET);
      launch_coop<true>(c, st);
      RN_LAUNCH_CHECK();
      // dbeta / dgamma = column sums of the two partial planes (one deferred launch per step, or two small ones now)
      if (int e = rn::launch_reduce_rows(c.partial, dbeta, c.c, c.total_chunks, 0, st)) return e;
      return rn::launch_reduce_rows(c.partial + (size_t)c.total_chunks * c.c, dgamma, c.c, c.total_chunks, 0, st);
    }
  }
  if (rows_ok) return rows_path();
  hipLaunchKernelGGL(gn_partial_kernel<true>, dim3(a.total_chunks), dim3(T), 0, st, a);
  // dbeta / dgamma = the column sums of the two partial planes: blocks appended to the finalize launch, or -- while
  // the step's reductions are deferred -- two rows of the single batched reduction
  const bool deferring = rn::reduce_deferred(st);
  hipLaunchKernelGGL(gn_finalize_kernel<true>, dim3(a.total_samples * a.groups + (deferring ? 0 : rn::ceil_div(a.c, 16))), dim3(T), 0,
                     st, a);
  hipLaunchKernelGGL(gn_apply_kernel<true>, dim3(apply_blocks(a), a.total_samples), dim3(T), 0, st, a);
  RN_LAUNCH_CHECK();
  if (deferring) {
    if (int e = rn::launch_reduce_rows(a.partial, dbeta, a.c, a.total_chunks, 0, st)) return e;
    return rn::launch_reduce_rows(a.partial + (size_t)a.total_chunks * a.c, dgamma, a.c, a.total_chunks, 0, st);
  }
  return RN_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// The fp16 inference path with the statistics taken from the producing conv (rn_conv2d_fwd_f16_fold): the conv's epilogue
// writes per (m-tile, channel) sums in the layout of gn_partial_kernel, so the same fp64 finalize and the same fp16 apply
// kernel run behind it -- without the statistics pass, and without the apply pass where the consumer is another folded conv.
// ---------------------------------------------------------------------------------------------------------------------
namespace {
// fp16 apply passes: tensors that cannot stay in the caches until their next use anyway (>= 32 MB: the cfg-5 batch) are read
// with non-temporal loads (RN_F16_NT=0: never; =1: always)
int f16_nt_loads(double bytes) {
  static const int mode = getenv("RN_F16_NT") ? atoi(getenv("RN_F16_NT")) : -1;
  return mode >= 0 ? (mode != 0) : (bytes >= 33554432.0);
}
// rn_group_norm_finalize for rows whose channels split into 64-channel slabs of whole groups: block = (sample, slab), thread =
// (channel of the slab, one of four row lanes) -- every load instruction of a wave reads 256 contiguous bytes of a row (the
// per-group kernel above reads one float per lane from `cpg`-wide pieces of `rows` different rows: 12.7 us per launch at cfg 5,
// 68 launches per forward pass).  fp64, fixed order: rows of a lane in order, lanes in order, channels of a group in order.
struct FinColsArgs { const float* partial; float* mean; float* rstd; int rows, c, groups, cpg, hw; size_t plane; float eps; };
__global__ __launch_bounds__(T) void gn_finalize_cols_kernel(const FinColsArgs a) {
  __shared__ double red[4][64][2];
  __shared__ double chan[64][2];
  const int tid = threadIdx.x, cl = tid & 63, rl = tid >> 6;
  const int slabs = a.c >> 6;
  const int sample = blockIdx.x / slabs, slab = blockIdx.x - sample * slabs;
  const float* __restrict__ p1 = a.partial + (size_t)sample * a.rows * a.c + slab * 64 + cl;
  const float* __restrict__ p2 = p1 + a.plane;
  double s1 = 0.0, s2 = 0.0;
  for (int r0 = rl; r0 < a.rows; r0 += 32) {
    float u[8], v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const size_t rr = (size_t)min(r0 + 4 * j, a.rows - 1) * a.c;
      u[j] = p1[rr]; v[j] = p2[rr];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (r0 + 4 * j < a.rows) { s1 += (double)u[j]; s2 += (double)v[j]; }
  }
  red[rl][cl][0] = s1; red[rl][cl][1] = s2;
  __syncthreads();
  if (tid < 64) {
    chan[tid][0] = ((red[0][tid][0] + red[1][tid][0]) + red[2][tid][0]) + red[3][tid][0];
    chan[tid][1] = ((red[0][tid][1] + red[1][tid][1]) + red[2][tid][1]) + red[3][tid][1];
  }
  __syncthreads();
  const int gps = 64 / a.cpg;            // groups per slab
  if (tid < gps) {
    double v1 = 0.0, v2 = 0.0;
    for (int j = 0; j < a.cpg; ++j) { v1 += chan[tid * a.cpg + j][0]; v2 += chan[tid * a.cpg + j][1]; }
    const double m = (double)a.hw * (double)a.cpg;
    const double mean = v1 / m;
    double var = v2 / m - mean * mean;
    if (var < 0.0) var = 0.0;
    const int g = slab * gps + tid;
    a.mean[sample * a.groups + g] = (float)mean;
    a.rstd[sample * a.groups + g] = (float)(1.0 / sqrt(var + (double)a.eps));
  }
}
}  // namespace

extern "C" int rn_group_norm_finalize(const float* partial, int n, int rows_per_sample, int hw, int c, int groups, float eps, float* mean,
                                      float* rstd, rn_stream_t stream) {
  RN_CHECK_ARG(partial && mean && rstd && n >= 1 && rows_per_sample >= 1 && hw >= 1 && c >= 1 && groups >= 1 && c % groups == 0,
               "group_norm finalize: bad argument");
  const int cpg = c / groups;
  static const bool no_cols = getenv("RN_GN_FINALIZE_COLS") && atoi(getenv("RN_GN_FINALIZE_COLS")) == 0;    // (A/B measurements)
  if (!no_cols && c % 64 == 0 && cpg <= 64 && 64 % cpg == 0) {
    FinColsArgs f = {partial, mean, rstd, rows_per_sample, c, groups, cpg, hw, (size_t)n * rows_per_sample * c, eps};
    hipLaunchKernelGGL(gn_finalize_cols_kernel, dim3(n * (c / 64)), dim3(T), 0, (hipStream_t)stream, f);
    RN_LAUNCH_CHECK();
    return RN_OK;
  }
  GnArgs a = {};
  a.nseg = 1; a.c = c; a.groups = groups; a.cpg = c / groups; a.eps = eps;
  a.partial = const_cast<float*>(partial);
  GnSeg& d = a.seg[0];
  d.mean = mean; d.rstd = rstd; d.n = n; d.hw = hw; d.sample_start = 0; d.chunk_start = 0; d.chunks = rows_per_sample;
  a.total_samples = n; a.total_chunks = n * rows_per_sample;
  hipLaunchKernelGGL(gn_finalize_kernel<false>, dim3(n * groups), dim3(T), 0, (hipStream_t)stream, a);
  RN_LAUNCH_CHECK();
  return RN_OK;
}

namespace {
int apply_f16_impl(const void* x, const void* residual, void* y, int n, int hw, int c, int groups, const float* mean, const float* rstd,
                   const float* gamma, const float* beta, int act, int act_after_residual, const rn_gn_residual_norm* rn_, rn_stream_t stream);
}
extern "C" int rn_group_norm_apply_f16(const void* x, const void* residual, void* y, int n, int hw, int c, int groups, const float* mean,
                                       const float* rstd, const float* gamma, const float* beta, int act, int act_after_residual,
                                       rn_stream_t stream) {
  return apply_f16_impl(x, residual, y, n, hw, c, groups, mean, rstd, gamma, beta, act, act_after_residual, nullptr, stream);
}
extern "C" int rn_group_norm_apply_res_f16(const void* x, const void* residual, const rn_gn_residual_norm* residual_norm, void* y, int n, int hw,
                                           int c, int groups, const float* mean, const float* rstd, const float* gamma, const float* beta, int act,
                                           int act_after_residual, rn_stream_t stream) {
  RN_CHECK_ARG(residual && residual_norm && residual_norm->mean && residual_norm->rstd && residual_norm->gamma && residual_norm->beta &&
               residual_norm->groups >= 1 && c % residual_norm->groups == 0, "group_norm apply res f16: bad residual GroupNorm");
  return apply_f16_impl(x, residual, y, n, hw, c, groups, mean, rstd, gamma, beta, act, act_after_residual, residual_norm, stream);
}
namespace {
int apply_f16_impl(const void* x, const void* residual, void* y, int n, int hw, int c, int groups, const float* mean, const float* rstd,
                   const float* gamma, const float* beta, int act, int act_after_residual, const rn_gn_residual_norm* rn_, rn_stream_t stream) {
  RN_CHECK_ARG(x && y && mean && rstd && gamma && beta && n >= 1 && hw >= 1 && groups >= 1 && c % groups == 0, "group_norm apply f16: bad argument");
  RN_UNSUPPORTED(c % 8 != 0 || c > 2048, "group_norm apply f16: c=%d must be a multiple of 8 and <= 2048", c);
  GnArgs a = {};
  a.nseg = 1; a.c = c; a.groups = groups; a.cpg = c / groups; a.act = act; a.act_after_res = act_after_residual ? 1 : 0;
  a.in_half = a.out_half = 1;
  a.nt_loads = f16_nt_loads((double)n * hw * c * 2.0);
  a.gamma = gamma; a.beta = beta;
  GnSeg& d = a.seg[0];
  d.x = (const float*)x; d.y = (float*)y; d.res = (const float*)residual;
  d.mean = const_cast<float*>(mean); d.rstd = const_cast<float*>(rstd); d.n = n; d.hw = hw; d.sample_start = 0;
  a.total_samples = n;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(apply_blocks(a, 8), n);
  if (rn_) {
    a.res_mean = rn_->mean; a.res_rstd = rn_->rstd; a.res_gamma = rn_->gamma; a.res_beta = rn_->beta; a.res_groups = rn_->groups;
    switch (act) {
      case RN_ACT_RELU: hipLaunchKernelGGL((gn_apply_f16x8_kernel<RN_ACT_RELU, true>), grid, dim3(T), 0, st, a); break;
      case RN_ACT_ELU: hipLaunchKernelGGL((gn_apply_f16x8_kernel<RN_ACT_ELU, true>), grid, dim3(T), 0, st, a); break;
      case RN_ACT_NONE: hipLaunchKernelGGL((gn_apply_f16x8_kernel<RN_ACT_NONE, true>), grid, dim3(T), 0, st, a); break;
      default: RN_UNSUPPORTED(true, "group_norm apply res f16: activation %d (relu, elu, none)", act);
    }
    RN_LAUNCH_CHECK();
    return RN_OK;
  }
  switch (act) {
    case RN_ACT_RELU: hipLaunchKernelGGL(gn_apply_f16x8_kernel<RN_ACT_RELU>, grid, dim3(T), 0, st, a); break;
    case RN_ACT_ELU: hipLaunchKernelGGL(gn_apply_f16x8_kernel<RN_ACT_ELU>, grid, dim3(T), 0, st, a); break;
    case RN_ACT_RELU6: hipLaunchKernelGGL(gn_apply_f16x8_kernel<RN_ACT_RELU6>, grid, dim3(T), 0, st, a); break;
    case RN_ACT_SIGMOID: hipLaunchKernelGGL(gn_apply_f16x8_kernel<RN_ACT_SIGMOID>, grid, dim3(T), 0, st, a); break;
    default: hipLaunchKernelGGL(gn_apply_f16x8_kernel<RN_ACT_NONE>, grid, dim3(T), 0, st, a); break;
  }
  RN_LAUNCH_CHECK();
  return RN_OK;
}
}  // namespace
