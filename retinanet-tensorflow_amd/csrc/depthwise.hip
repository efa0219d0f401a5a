// Depthwise kxk convolution, NHWC fp32, TF SAME padding (mobilenet_v2.py:35-36,
// tf.nn.depthwise_conv2d with channel multiplier 1; kernel stored [kh,kw,C]).
// HBM-bound (9 MAC per loaded element): one thread owns 4 channels of one pixel, float4
// loads along C are fully coalesced, the k*k window re-reads hit L1/L2.
#include "rn_common.h"

namespace {

constexpr int T = 256;

struct DwArgs {
  const float* x; const float* w; const float* dy; float* out;
  int n, h, wd, c, k, stride, oh, ow, pad_t, pad_l;
  int chunks, ppc;  // wgrad
  float* partial;
};

template <bool K3>
__global__ __launch_bounds__(T) void dw_fwd_kernel(const DwArgs a) {
  // 32-bit index arithmetic: a tensor has < 2^31 bytes (fill), and a 64-bit division by a runtime divisor is ~150
  // instructions -- three of them per element were several times the nine taps' work
  const unsigned CQ = (unsigned)a.c >> 2;
  const unsigned total = (unsigned)a.n * a.oh * a.ow * CQ;
  // consecutive blocks (neighbouring pixels: shared taps) on ONE XCD's L2 -- dealt round-robin over the eight XCDs, every
  // XCD fetched the three input rows of each output row itself (measured: 3x the algorithmic fetch traffic)
  const unsigned bid = (unsigned)rn::xcd_remap(blockIdx.x, gridDim.x);
  for (unsigned i = bid * T + threadIdx.x; i < total; i += gridDim.x * T) {
    const int q4 = (int)(i % CQ);
    unsigned p = i / CQ;
    const int ow_ = (int)(p % (unsigned)a.ow); p /= (unsigned)a.ow;
    const int oh_ = (int)(p % (unsigned)a.oh);
    const int n_ = (int)(p / (unsigned)a.oh);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (K3) {
      // branch-free 3x3: every tap is loaded from a clamped address and multiplied by a 0/1 mask -- loads under
      // `if (in bounds)` compile to a branch + wait per tap, i.e. nine serialised round trips
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int ih = oh_ * a.stride - a.pad_t + kh;
        const int ihc = min(max(ih, 0), a.h - 1);
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int iw = ow_ * a.stride - a.pad_l + kw;
          const int iwc = min(max(iw, 0), a.wd - 1);
          const float m = ((unsigned)ih < (unsigned)a.h && (unsigned)iw < (unsigned)a.wd) ? 1.f : 0.f;
          const float4 xv = *reinterpret_cast<const float4*>(a.x + ((size_t)(n_ * a.h + ihc) * a.wd + iwc) * a.c + q4 * 4);
          const float4 wv = *reinterpret_cast<const float4*>(a.w + (size_t)(kh * 3 + kw) * a.c + q4 * 4);
          acc.x = fmaf(xv.x * m, wv.x, acc.x); acc.y = fmaf(xv.y * m, wv.y, acc.y);
          acc.z = fmaf(xv.z * m, wv.z, acc.z); acc.w = fmaf(xv.w * m, wv.w, acc.w);
        }
      }
    } else {
      for (int kh = 0; kh < a.k; ++kh) {
        const int ih = oh_ * a.stride - a.pad_t + kh;
        if ((unsigned)ih >= (unsigned)a.h) continue;
        for (int kw = 0; kw < a.k; ++kw) {
          const int iw = ow_ * a.stride - a.pad_l + kw;
          if ((unsigned)iw >= (unsigned)a.wd) continue;
          const float4 xv = *reinterpret_cast<const float4*>(a.x + ((size_t)(n_ * a.h + ih) * a.wd + iw) * a.c + q4 * 4);
          const float4 wv = *reinterpret_cast<const float4*>(a.w + (size_t)(kh * a.k + kw) * a.c + q4 * 4);
          acc.x = fmaf(xv.x, wv.x, acc.x); acc.y = fmaf(xv.y, wv.y, acc.y);
          acc.z = fmaf(xv.z, wv.z, acc.z); acc.w = fmaf(xv.w, wv.w, acc.w);
        }
      }
    }
    *reinterpret_cast<float4*>(a.out + (size_t)i * 4) = acc;
  }
}

// dx[n,ih,iw,c] = sum_{kh,kw} dy[n,oh,ow,c]*w[kh,kw,c] with oh*s + kh - pad_t == ih
template <bool K3>
__device__ __forceinline__ void dw_dgrad_body(const DwArgs& a, int bid, int nblk) {
  const unsigned CQ = (unsigned)a.c >> 2;           // (32-bit index arithmetic: see dw_fwd_kernel)
  const unsigned total = (unsigned)a.n * a.h * a.wd * CQ;
  for (unsigned i = (unsigned)rn::xcd_remap(bid, nblk) * T + threadIdx.x; i < total; i += (unsigned)nblk * T) {   // (XCD locality: see dw_fwd_kernel)
    const int q4 = (int)(i % CQ);
    unsigned p = i / CQ;
    const int iw = (int)(p % (unsigned)a.wd); p /= (unsigned)a.wd;
    const int ih = (int)(p % (unsigned)a.h);
    const int n_ = (int)(p / (unsigned)a.h);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (K3 && a.stride == 2) {
      // stride 2: only the taps whose parity matches the input pixel's reach an output (<= 2 x 2 of the 3 x 3): visit
      // just those, in the same ascending order as the full loop (same bits), branch-free
      const int kh0 = (ih + a.pad_t) & 1, kw0 = (iw + a.pad_l) & 1;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int kh = kh0 + 2 * j;
        const int ohs = ih + a.pad_t - kh;  // even
        const int oh_ = ohs >> 1;
        const bool rok = kh < 3 && ohs >= 0 && oh_ < a.oh;
        const int ohc = min(max(oh_, 0), a.oh - 1), khc = min(kh, 2);
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2) {
          const int kw = kw0 + 2 * i2;
          const int ows = iw + a.pad_l - kw;
          const int ow_ = ows >> 1;
          const float m = (rok && kw < 3 && ows >= 0 && ow_ < a.ow) ? 1.f : 0.f;
          const int owc = min(max(ow_, 0), a.ow - 1), kwc = min(kw, 2);
          const float4 dv = *reinterpret_cast<const float4*>(a.dy + ((size_t)(n_ * a.oh + ohc) * a.ow + owc) * a.c + q4 * 4);
          const float4 wv = *reinterpret_cast<const float4*>(a.w + (size_t)(khc * 3 + kwc) * a.c + q4 * 4);
          acc.x = fmaf(dv.x * m, wv.x, acc.x); acc.y = fmaf(dv.y * m, wv.y, acc.y);
          acc.z = fmaf(dv.z * m, wv.z, acc.z); acc.w = fmaf(dv.w * m, wv.w, acc.w);
        }
      }
    } else if (K3) {  // branch-free (see dw_fwd_kernel); stride 1 (or any other stride)
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int ohs = ih + a.pad_t - kh;
        const int oh_ = ohs / a.stride;
        const bool rok = ohs >= 0 && oh_ * a.stride == ohs && oh_ < a.oh;
        const int ohc = min(max(oh_, 0), a.oh - 1);
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int ows = iw + a.pad_l - kw;
          const int ow_ = ows / a.stride;
          const float m = (rok && ows >= 0 && ow_ * a.stride == ows && ow_ < a.ow) ? 1.f : 0.f;
          const int owc = min(max(ow_, 0), a.ow - 1);
          const float4 dv = *reinterpret_cast<const float4*>(a.dy + ((size_t)(n_ * a.oh + ohc) * a.ow + owc) * a.c + q4 * 4);
          const float4 wv = *reinterpret_cast<const float4*>(a.w + (size_t)(kh * 3 + kw) * a.c + q4 * 4);
          acc.x = fmaf(dv.x * m, wv.x, acc.x); acc.y = fmaf(dv.y * m, wv.y, acc.y);
          acc.z = fmaf(dv.z * m, wv.z, acc.z); acc.w = fmaf(dv.w * m, wv.w, acc.w);
        }
      }
    } else {
      for (int kh = 0; kh < a.k; ++kh) {
        const int ohs = ih + a.pad_t - kh;
        if (ohs < 0 || ohs % a.stride) continue;
        const int oh_ = ohs / a.stride;
        if (oh_ >= a.oh) continue;
        for (int kw = 0; kw < a.k; ++kw) {
          const int ows = iw + a.pad_l - kw;
          if (ows < 0 || ows % a.stride) continue;
          const int ow_ = ows / a.stride;
          if (ow_ >= a.ow) continue;
          const float4 dv = *reinterpret_cast<const float4*>(a.dy + ((size_t)(n_ * a.oh + oh_) * a.ow + ow_) * a.c + q4 * 4);
          const float4 wv = *reinterpret_cast<const float4*>(a.w + (size_t)(kh * a.k + kw) * a.c + q4 * 4);
          acc.x = fmaf(dv.x, wv.x, acc.x); acc.y = fmaf(dv.y, wv.y, acc.y);
          acc.z = fmaf(dv.z, wv.z, acc.z); acc.w = fmaf(dv.w, wv.w, acc.w);
        }
      }
    }
    *reinterpret_cast<float4*>(a.out + (size_t)i * 4) = acc;
  }
}

template <bool K3>
__global__ __launch_bounds__(T) void dw_dgrad_kernel(const DwArgs a) {
  dw_dgrad_body<K3>(a, blockIdx.x, gridDim.x);
}

// partial[chunk][tap][c] = sum over the chunk's output pixels of x(tap)*dy ; k == 3 only.
__device__ __forceinline__ void dw_wgrad_partial_body(const DwArgs& a, int chunk) {
  __shared__ float red[T][4];
  const int tid = threadIdx.x;
  const int CQ = a.c >> 2, lanes = T / CQ;
  const int q4 = tid % CQ, pl = tid / CQ;
  const unsigned npix = (unsigned)a.n * a.oh * a.ow;
  const unsigned p_begin = (unsigned)chunk * a.ppc;
  const unsigned p_end = p_begin + a.ppc < npix ? p_begin + a.ppc : npix;
  float4 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (pl < lanes) {
    for (unsigned p = p_begin + pl; p < p_end; p += lanes) {
      const int ow_ = (int)(p % (unsigned)a.ow);
      const unsigned r = p / (unsigned)a.ow;
      const int oh_ = (int)(r % (unsigned)a.oh);
      const int n_ = (int)(r / (unsigned)a.oh);
      const float4 dv = *reinterpret_cast<const float4*>(a.dy + (size_t)p * a.c + q4 * 4);
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int ih = oh_ * a.stride - a.pad_t + kh;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int iw = ow_ * a.stride - a.pad_l + kw;
          const float m = ((unsigned)ih < (unsigned)a.h && (unsigned)iw < (unsigned)a.wd) ? 1.f : 0.f;  // branch-free
          const int ihc = min(max(ih, 0), a.h - 1), iwc = min(max(iw, 0), a.wd - 1);
          const float4 xv = *reinterpret_cast<const float4*>(a.x + ((size_t)(n_ * a.h + ihc) * a.wd + iwc) * a.c + q4 * 4);
          float4& t = acc[kh * 3 + kw];
          t.x = fmaf(xv.x * m, dv.x, t.x); t.y = fmaf(xv.y * m, dv.y, t.y);
          t.z = fmaf(xv.z * m, dv.z, t.z); t.w = fmaf(xv.w * m, dv.w, t.w);
        }
      }
    }
  }
  for (int t = 0; t < 9; ++t) {
    __syncthreads();
    red[tid][0] = acc[t].x; red[tid][1] = acc[t].y; red[tid][2] = acc[t].z; red[tid][3] = acc[t].w;
    __syncthreads();
    if (tid < CQ) {
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
      for (int l = 0; l < lanes; ++l) {
        s0 += red[l * CQ + tid][0]; s1 += red[l * CQ + tid][1];
        s2 += red[l * CQ + tid][2]; s3 += red[l * CQ + tid][3];
      }
      *reinterpret_cast<float4*>(a.partial + ((size_t)chunk * 9 + t) * a.c + tid * 4) = make_float4(s0, s1, s2, s3);
    }
  }
}

__global__ __launch_bounds__(T) void dw_wgrad_partial_kernel(const DwArgs a) { dw_wgrad_partial_body(a, rn::xcd_remap(blockIdx.x, gridDim.x)); }

// both gradients of a 3x3 depthwise conv in ONE launch: blocks [0, dgrad_blocks) compute dx, the rest the weight-gradient
// partials (independent work on the same dy; two launch-latency-bound kernels otherwise)
struct DwBwdArgs { DwArgs d, w; int dgrad_blocks; };
__global__ __launch_bounds__(T) void dw_bwd_kernel(const DwBwdArgs b) {
  if ((int)blockIdx.x < b.dgrad_blocks) dw_dgrad_body<true>(b.d, blockIdx.x, b.dgrad_blocks);
  else dw_wgrad_partial_body(b.w, rn::xcd_remap((int)blockIdx.x - b.dgrad_blocks, (int)gridDim.x - b.dgrad_blocks));
}

// ---------------------------------------------------------------------------------------------------------------------
// 3x3 forward that also emits the GroupNorm partial sums of its output (rn_depthwise_fwd_stats, rn::StatDev): a block owns
// a run of `ppc` output pixels of ONE sample and all channels -- thread = (channel quad, pixel lane), R pixels each, the
// nine weights in registers -- so its per-channel sums fold into one row of per-group (sum, sum of squares) inside the
// block; the GroupNorm that follows merges the sample's rows.  Same fmaf order as dw_fwd_kernel: the same output bits.
constexpr int ST = 512;
struct DwStatArgs { DwArgs a; rn::StatDev st; int chunks_ps, ppc; };

template <int R>
__global__ __launch_bounds__(ST) void dw_fwd_stats_kernel(const DwStatArgs b) {
  __shared__ float red[ST][8];
  __shared__ float chan[1024][2];
  const DwArgs& a = b.a;
  const rn::StatDev& st = b.st;
  const int tid = threadIdx.x, C = a.c, CQ = C >> 2, lanes = ST / CQ;
  const int q4 = tid % CQ, pl = tid / CQ;
  const bool active = pl < lanes;
  const int chunk = rn::xcd_remap(blockIdx.x, gridDim.x);      // neighbouring pixel chunks share taps: one XCD's L2
  const int sample = chunk / b.chunks_ps, ck = chunk - sample * b.chunks_ps;
  const int ohw = a.oh * a.ow;
  const int p_begin = ck * b.ppc, p_end = min(p_begin + b.ppc, ohw);
  float4 wv[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) wv[t] = *reinterpret_cast<const float4*>(a.w + (size_t)t * C + q4 * 4);
  const float inv_ow = 1.f / (float)a.ow;
  const float* __restrict__ xs = a.x + (size_t)sample * a.h * a.wd * C + q4 * 4;
  float* __restrict__ ys = a.out + (size_t)sample * ohw * C + q4 * 4;
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const int p = p_begin + pl + k * lanes;
    const bool ok = active && p < p_end;
    const int pc = min(p, ohw - 1);
    const int oh_ = (int)(((float)pc + 0.5f) * inv_ow);   // floor(pc / ow): exact for pc < 2^20, ow <= 2^12 (host-checked)
    const int ow_ = pc - oh_ * a.ow;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int ih = oh_ * a.stride - a.pad_t + kh;
      const int ihc = min(max(ih, 0), a.h - 1);
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int iw = ow_ * a.stride - a.pad_l + kw;
        const int iwc = min(max(iw, 0), a.wd - 1);
        const float m = ((unsigned)ih < (unsigned)a.h && (unsigned)iw < (unsigned)a.wd) ? 1.f : 0.f;
        const float4 xv = *reinterpret_cast<const float4*>(xs + ((size_t)ihc * a.wd + iwc) * C);
        const float4 w4 = wv[kh * 3 + kw];
        acc.x = fmaf(xv.x * m, w4.x, acc.x); acc.y = fmaf(xv.y * m, w4.y, acc.y);
        acc.z = fmaf(xv.z * m, w4.z, acc.z); acc.w = fmaf(xv.w * m, w4.w, acc.w);
      }
    }
    if (ok) {
      *reinterpret_cast<float4*>(ys + (size_t)p * C) = acc;
      s1[0] += acc.x; s1[1] += acc.y; s1[2] += acc.z; s1[3] += acc.w;
      s2[0] = fmaf(acc.x, acc.x, s2[0]); s2[1] = fmaf(acc.y, acc.y, s2[1]);
      s2[2] = fmaf(acc.z, acc.z, s2[2]); s2[3] = fmaf(acc.w, acc.w, s2[3]);
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) { red[tid][j] = s1[j]; red[tid][4 + j] = s2[j]; }
  __syncthreads();
  for (int e = tid; e < CQ * 8; e += ST) {      // pixel lanes in order
    const int q = e >> 3, comp = e & 7;
    float t = 0.f;
    for (int l = 0; l < lanes; ++l) t += red[l * CQ + q][comp];
    chan[q * 4 + (comp & 3)][comp >> 2] = t;
  }
  __syncthreads();
  for (int g = tid; g < st.groups; g += ST) {   // channels of a group in order
    float t1 = 0.f, t2 = 0.f;
    for (int j = 0; j < st.cpg; ++j) { t1 += chan[g * st.cpg + j][0]; t2 += chan[g * st.cpg + j][1]; }
    st.rows[(size_t)chunk * st.groups + g] = make_float2(t1, t2);
  }
}

int fill(DwArgs* a, int n, int h, int w, int c, int k, int stride) {
  RN_CHECK_ARG(n >= 1 && h >= 1 && w >= 1 && c >= 1 && k >= 1 && stride >= 1, "depthwise: bad shape");
  RN_UNSUPPORTED(c % 4 != 0 || c > 1024, "depthwise: c=%d must be a multiple of 4 and <= 1024", c);
  RN_UNSUPPORTED((double)n * h * w * c >= 536870912.0, "depthwise: tensors of 2 GiB and more are not supported");   // 32-bit indices
  a->n = n; a->h = h; a->wd = w; a->c = c; a->k = k; a->stride = stride;
  rn::same_pad(h, k, stride, &a->oh, &a->pad_t);
  rn::same_pad(w, k, stride, &a->ow, &a->pad_l);
  return RN_OK;
}

unsigned grid_for(int64_t total) {
  int64_t b = (total + T - 1) / T;
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  return (unsigned)b;
}

void wgrad_plan(const DwArgs& a, int* chunks, int* ppc) {
  const int64_t npix = (int64_t)a.n * a.oh * a.ow;
  int64_t ck = (npix * a.c + 2047) / 2048;   // short blocks: the tap loop is latency-bound
  if (ck > 4096) ck = 4096;
  if (ck < 1) ck = 1;
  *ppc = (int)((npix + ck - 1) / ck);
  *chunks = (int)((npix + *ppc - 1) / *ppc);
}

}  // namespace

extern "C" int rn_depthwise_fwd(const float* x, const float* wgt, float* y, int n, int h, int w, int c, int k,
                                int stride, rn_stream_t stream) {
  DwArgs a = {};
  if (int e = fill(&a, n, h, w, c, k, stride)) return e;
  RN_CHECK_ARG(x && wgt && y, "depthwise fwd: null pointer");
  a.x = x; a.w = wgt; a.out = y;
  if (k == 3) hipLaunchKernelGGL(dw_fwd_kernel<true>, dim3(grid_for((int64_t)n * a.oh * a.ow * (c / 4))), dim3(T), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(dw_fwd_kernel<false>, dim3(grid_for((int64_t)n * a.oh * a.ow * (c / 4))), dim3(T), 0, (hipStream_t)stream, a);
  RN_LAUNCH_CHECK();
  return RN_OK;
}

namespace {
// plan of the statistics kernel: pixels per thread R (4, or 8 / 16 where that keeps a sample at <= 1024 rows)
bool stats_plan(const DwArgs& a, int groups, int* R, int* ppc, int* chunks_ps) {
  if (a.k != 3 || groups < 1 || a.c % groups) return false;
  const long ohw = (long)a.oh * a.ow;
  if (ohw >= (1 << 20) || a.ow > 4096 || (double)ohw * (a.c / groups) >= 16777216.0) return false;
  const int lanes = ST / (a.c / 4);
  int r = 4;
  while (r < 16 && rn::ceil_div((int)ohw, lanes * r) > 1024) r *= 2;  // (per-group rows: a thousand are still cheap to merge; short blocks: a thread's pixels are a serial chain)
  *R = r; *ppc = lanes * r; *chunks_ps = rn::ceil_div((int)ohw, lanes * r);
  return rn_group_norm_rows_ok(a.c, groups, *chunks_ps, 1) != 0;
}
}  // namespace

extern "C" size_t rn_depthwise_stats_rows(int n, int h, int w, int c, int k, int stride, int groups, rn_gn_rows* layout) {
  DwArgs a = {};
  if (n < 1 || h < 1 || w < 1 || c < 4 || c % 4 || c > 1024 || k < 1 || stride < 1) return 0;
  if (fill(&a, n, h, w, c, k, stride)) return 0;
  int R, ppc, cps;
  if (!stats_plan(a, groups, &R, &ppc, &cps)) return 0;
  if (layout) { layout->rows_per_sample = cps; layout->per_group = 1; layout->groups = groups; }
  return (size_t)n * cps * groups * 8;
}

extern "C" int rn_depthwise_fwd_stats(const float* x, const float* wgt, float* y, int n, int h, int w, int c, int k, int stride,
                                      const rn_gn_rows* rows, rn_stream_t stream) {
  DwStatArgs b = {};
  if (int e = fill(&b.a, n, h, w, c, k, stride)) return e;
  RN_CHECK_ARG(x && wgt && y && rows && rows->rows, "depthwise fwd stats: null pointer");
  int R;
  RN_UNSUPPORTED(!stats_plan(b.a, rows->groups, &R, &b.ppc, &b.chunks_ps) || rows->rows_per_sample != b.chunks_ps || rows->per_group != 1,
                 "depthwise fwd stats: unsupported shape / layout (rn_depthwise_stats_rows)");
  b.a.x = x; b.a.w = wgt; b.a.out = y;
  b.st.rows = (float2*)rows->rows; b.st.groups = rows->groups; b.st.cpg = c / rows->groups;
  const dim3 grid((unsigned)(n * b.chunks_ps));
  if (R == 4) hipLaunchKernelGGL(dw_fwd_stats_kernel<4>, grid, dim3(ST), 0, (hipStream_t)stream, b);
  else if (R == 8) hipLaunchKernelGGL(dw_fwd_stats_kernel<8>, grid, dim3(ST), 0, (hipStream_t)stream, b);
  else hipLaunchKernelGGL(dw_fwd_stats_kernel<16>, grid, dim3(ST), 0, (hipStream_t)stream, b);
  RN_LAUNCH_CHECK();
  return RN_OK;
}

extern "C" int rn_depthwise_dgrad(const float* dy, const float* wgt, float* dx, int n, int h, int w, int c, int k,
                                  int stride, rn_stream_t stream) {
  DwArgs a = {};
  if (int e = fill(&a, n, h, w, c, k, stride)) return e;
  RN_CHECK_ARG(dy && wgt && dx, "depthwise dgrad: null pointer");
  a.dy = dy; a.w = wgt; a.out = dx;
  if (k == 3) hipLaunchKernelGGL(dw_dgrad_kernel<true>, dim3(grid_for((int64_t)n * h * w * (c / 4))), dim3(T), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(dw_dgrad_kernel<false>, dim3(grid_for((int64_t)n * h * w * (c / 4))), dim3(T), 0, (hipStream_t)stream, a);
  RN_LAUNCH_CHECK();
  return RN_OK;
}

extern "C" size_t rn_depthwise_wgrad_workspace(int n, int h, int w, int c, int k, int stride) {
  DwArgs a = {};
  if (fill(&a, n, h, w, c, k, stride)) return 0;
  int chunks, ppc;
  wgrad_plan(a, &chunks, &ppc);
  return (size_t)chunks * k * k * c * sizeof(float);
}

extern "C" int rn_depthwise_wgrad(const float* x, const float* dy, float* dw, int n, int h, int w, int c, int k,
                                  int stride, void* workspace, size_t workspace_bytes, rn_stream_t stream, rn_reduce_list* defer) {
  rn::DeferScope defer_scope_(defer);
  DwArgs a = {};
  if (int e = fill(&a, n, h, w, c, k, stride)) return e;
  RN_CHECK_ARG(x && dy && dw && workspace, "depthwise wgrad: null pointer");
  RN_UNSUPPORTED(k != 3, "depthwise wgrad: only 3x3 kernels (got %d)", k);
  wgrad_plan(a, &a.chunks, &a.ppc);
  const size_t need = (size_t)a.chunks * 9 * c * sizeof(float);
  if (workspace_bytes < need) {
    rn::set_error("depthwise wgrad: workspace %zu < %zu", workspace_bytes, need);
    return RN_EWORKSPACE;
  }
  a.x = x; a.dy = dy; a.partial = (float*)workspace;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(dw_wgrad_partial_kernel, dim3(a.chunks), dim3(T), 0, st, a);
  RN_LAUNCH_CHECK();
  return rn::launch_reduce_rows((const float*)workspace, dw, 9 * c, a.chunks, 0, st);
}

// dx and dw of a 3x3 depthwise conv from one launch (+ the fixed-order row reduction of the dw partials, which joins
// the step's deferred reduction when that is active).  Workspace: rn_depthwise_wgrad_workspace bytes.
extern "C" int rn_depthwise_bwd(const float* x, const float* dy, const float* wgt, float* dx, float* dw, int n, int h, int w, int c,
                                int k, int stride, void* workspace, size_t workspace_bytes, rn_stream_t stream, rn_reduce_list* defer) {
  rn::DeferScope defer_scope_(defer);
  DwBwdArgs b = {};
  if (int e = fill(&b.d, n, h, w, c, k, stride)) return e;
  RN_CHECK_ARG(x && dy && wgt && dx && dw && workspace, "depthwise bwd: null pointer");
  RN_UNSUPPORTED(k != 3, "depthwise bwd: only 3x3 kernels (got %d)", k);
  b.w = b.d;
  wgrad_plan(b.w, &b.w.chunks, &b.w.ppc);
  const size_t need = (size_t)b.w.chunks * 9 * c * sizeof(float);
  if (workspace_bytes < need) {
    rn::set_error("depthwise bwd: workspace %zu < %zu", workspace_bytes, need);
    return RN_EWORKSPACE;
  }
  b.d.dy = dy; b.d.w = wgt; b.d.out = dx;
  b.w.x = x; b.w.dy = dy; b.w.partial = (float*)workspace;
  b.dgrad_blocks = (int)grid_for((int64_t)n * h * w * (c / 4));
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(dw_bwd_kernel, dim3(b.dgrad_blocks + b.w.chunks), dim3(T), 0, st, b);
  RN_LAUNCH_CHECK();
  return rn::launch_reduce_rows((const float*)workspace, dw, 9 * c, b.w.chunks, 0, st);
}
