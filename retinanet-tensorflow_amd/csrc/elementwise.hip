// Small HBM-bound elementwise ops of the FPN: stand-alone activation (retinanet.py:180-181)
// and lateral + nearest-neighbour upsample of the coarser level (retinanet.py:153-157,
// ResizeNearestNeighbor align_corners=True) with its gradient.
#include "rn_common.h"

namespace {
constexpr int T = 256;

__global__ void act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t count, int act) {
  for (int64_t i = (int64_t)blockIdx.x * T + threadIdx.x; i < count; i += (int64_t)gridDim.x * T)
    y[i] = rn::act_fwd(x[i], act);
}
__global__ void act_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx,
                               int64_t count, int act) {
  for (int64_t i = (int64_t)blockIdx.x * T + threadIdx.x; i < count; i += (int64_t)gridDim.x * T)
    dx[i] = dy[i] * rn::act_grad(x[i], act);
}

__global__ void act_fwd_f16_kernel(const void* __restrict__ x, void* __restrict__ y, int64_t quads, int act) {
  for (int64_t i = (int64_t)blockIdx.x * T + threadIdx.x; i < quads; i += (int64_t)gridDim.x * T) {
    float4 v = rn::ld4(x, (size_t)i * 4, 1);
    v.x = rn::act_fwd(v.x, act); v.y = rn::act_fwd(v.y, act); v.z = rn::act_fwd(v.z, act); v.w = rn::act_fwd(v.w, act);
    rn::st4(y, (size_t)i * 4, 1, v);
  }
}

// [TF-sem] src = min(roundf(dst * (in-1)/(out-1)), in-1); scale evaluated in float32
__device__ __forceinline__ int nn_src(int dst, int in, int out) {
  const float scale = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
  const int s = (int)roundf((float)dst * scale);
  return s < in - 1 ? s : in - 1;
}

__global__ void upsample_add_kernel(const void* __restrict__ lat, const void* __restrict__ top, void* __restrict__ y,
                                    int n, int h, int w, int th, int tw, int c, int is_half) {
  const int CQ = c >> 2;
  const int64_t total = (int64_t)n * h * w * CQ;
  for (int64_t i = (int64_t)blockIdx.x * T + threadIdx.x; i < total; i += (int64_t)gridDim.x * T) {
    const int q4 = (int)(i % CQ);
    int64_t p = i / CQ;
    const int x_ = (int)(p % w); p /= w;
    const int y_ = (int)(p % h);
    const int n_ = (int)(p / h);
    const int sy = nn_src(y_, th, h), sx = nn_src(x_, tw, w);
    const float4 a = rn::ld4(lat, (size_t)i * 4, is_half);
    const float4 b = rn::ld4(top, ((size_t)(n_ * th + sy) * tw + sx) * c + q4 * 4, is_half);
    rn::st4(y, (size_t)i * 4, is_half, make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w));
  }
}

// dtop[n,sy,sx,c] = sum of dy over the fine cells whose source is (sy,sx); deterministic gather.
__global__ void upsample_add_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dtop, int n, int h, int w,
                                        int th, int tw, int c) {
  const int CQ = c >> 2;
  const int64_t total = (int64_t)n * th * tw * CQ;
  for (int64_t i = (int64_t)blockIdx.x * T + threadIdx.x; i < total; i += (int64_t)gridDim.x * T) {
    const int q4 = (int)(i % CQ);
    int64_t p = i / CQ;
    const int sx = (int)(p % tw); p /= tw;
    const int sy = (int)(p % th);
    const int n_ = (int)(p / th);
    // candidate fine rows/cols: the inverse image of a monotone map is a contiguous range
    int y_lo = (int)((int64_t)sy * (h - 1) / (th > 1 ? th - 1 : 1)) - (h / th + 2);
    int x_lo = (int)((int64_t)sx * (w - 1) / (tw > 1 ? tw - 1 : 1)) - (w / tw + 2);
    if (y_lo < 0) y_lo = 0;
    if (x_lo < 0) x_lo = 0;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int y_ = y_lo; y_ < h; ++y_) {
      const int my = nn_src(y_, th, h);
      if (my > sy) break;
      if (my < sy) continue;
      for (int x_ = x_lo; x_ < w; ++x_) {
        const int mx = nn_src(x_, tw, w);
        if (mx > sx) break;
        if (mx < sx) continue;
        const float4 v = *reinterpret_cast<const float4*>(dy + ((size_t)(n_ * h + y_) * w + x_) * c + q4 * 4);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
    }
    *reinterpret_cast<float4*>(dtop + (size_t)i * 4) = acc;
  }
}

// ---- stand-alone dropout (DenseNet places tf.layers.Dropout after a conv, densenet.py:44,67,77,143)
__global__ void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t count, float rate, uint64_t seed,
                               const uint64_t* __restrict__ seed_dev) {
  const uint64_t sd = seed + (seed_dev ? *seed_dev : 0ull);
  const float ks = 1.f / (1.f - rate);
  for (int64_t i = (int64_t)blockIdx.x * T + threadIdx.x; i < count; i += (int64_t)gridDim.x * T)
    y[i] = rn::uniform01(sd, (uint64_t)i) >= rate ? x[i] * ks : 0.f;
}

// ---- 3x3/2 (k x k / s) max pool, TF SAME: padded cells never win (resnet.py:200, densenet.py:180)
struct PoolArgs { const float* x; const float* dy; float* out; int n, h, w, c, k, s, oh, ow, pt, pl; int is_half; uint8_t* arg; };

__global__ void maxpool_fwd_kernel(const PoolArgs a) {
  const int CQ = a.c >> 2;
  const int64_t total = (int64_t)a.n * a.oh * a.ow * CQ;
  for (int64_t i = (int64_t)blockIdx.x * T + threadIdx.x; i < total; i += (int64_t)gridDim.x * T) {
    const int q4 = (int)(i % CQ);
    int64_t p = i / CQ;
    const int ow_ = (int)(p % a.ow); p /= a.ow;
    const int oh_ = (int)(p % a.oh);
    const int n_ = (int)(p / a.oh);
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    uchar4 am = make_uchar4(0, 0, 0, 0);  // tap index (kh * k + kw) of the FIRST maximum, for the backward pass
    for (int kh = 0; kh < a.k; ++kh) {
      const int ih = oh_ * a.s - a.pt + kh;
      if ((unsigned)ih >= (unsigned)a.h) continue;
      for (int kw = 0; kw < a.k; ++kw) {
        const int iw = ow_ * a.s - a.pl + kw;
        if ((unsigned)iw >= (unsigned)a.w) continue;
        const float4 v = rn::ld4(a.x, ((size_t)(n_ * a.h + ih) * a.w + iw) * a.c + q4 * 4, a.is_half);
        const unsigned char t = (unsigned char)(kh * a.k + kw);
        if (v.x > m.x) { m.x = v.x; am.x = t; }
        if (v.y > m.y) { m.y = v.y; am.y = t; }
        if (v.z > m.z) { m.z = v.z; am.z = t; }
        if (v.w > m.w) { m.w = v.w; am.w = t; }
      }
    }
    rn::st4(a.out, (size_t)i * 4, a.is_half, m);
    if (a.arg) *reinterpret_cast<uchar4*>(a.arg + (size_t)i * 4) = am;
  }
}

// fp16 inference: max pool of act(GN(y)) straight from the RAW conv output y (ResNeXt's stem: conv -> GroupNorm -> ReLU -> 3x3/2
// max pool at 512^2 x 64 x 16 images) -- the normalised 537 MB tensor is neither written nor read back.  Thread = (output pixel,
// 8 channels); (scale, shift) of the sample's channels in LDS; fp32 arithmetic, one rounding to fp16 at the end: rounding is
// monotone, so the result equals the max pool of the fp16-rounded normalised tensor bit for bit.
struct PoolGnArgs {
  const _Float16* x; _Float16* out; const float* mean; const float* rstd; const float* gamma; const float* beta;
  int n, h, w, c, k, s, oh, ow, pt, pl, groups, act;
};
typedef _Float16 pool_half8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(T) void maxpool_gn_f16_kernel(const PoolGnArgs a) {
  __shared__ float2 tab[2048];
  const int smp = blockIdx.y, tid = threadIdx.x;
  const int cpg = a.c / a.groups;
  for (int c = tid; c < a.c; c += T) {
    const int g = c / cpg;
    const float sc = a.rstd[smp * a.groups + g] * a.gamma[c];
    tab[c] = make_float2(sc, a.beta[c] - a.mean[smp * a.groups + g] * sc);
  }
  __syncthreads();
  const int C8 = a.c >> 3;
  const int64_t total = (int64_t)a.oh * a.ow * C8;
  const _Float16* __restrict__ x = a.x + (size_t)smp * a.h * a.w * a.c;
  _Float16* __restrict__ out = a.out + (size_t)smp * a.oh * a.ow * a.c;
  for (int64_t i = (int64_t)blockIdx.x * T + tid; i < total; i += (int64_t)gridDim.x * T) {
    const int q = (int)(i % C8);
    const int p = (int)(i / C8);
    const int ow_ = p % a.ow, oh_ = p / a.ow;
    float sc[8], sh[8], m[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float2 t = tab[q * 8 + j]; sc[j] = t.x; sh[j] = t.y; m[j] = -INFINITY; }
    for (int kh = 0; kh < a.k; ++kh) {
      const int ih = oh_ * a.s - a.pt + kh;
      if ((unsigned)ih >= (unsigned)a.h) continue;
      for (int kw = 0; kw < a.k; ++kw) {
        const int iw = ow_ * a.s - a.pl + kw;
        if ((unsigned)iw >= (unsigned)a.w) continue;           // (padded cells never win)
        const pool_half8 v = *reinterpret_cast<const pool_half8*>(x + ((size_t)ih * a.w + iw) * a.c + q * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) m[j] = fmaxf(m[j], rn::act_fwd(fmaf((float)v[j], sc[j], sh[j]), a.act));
      }
    }
    pool_half8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (_Float16)m[j];
    *reinterpret_cast<pool_half8*>(out + (size_t)i * 8) = o;
  }
}

// gradient goes to the FIRST maximum of each window (row-major window order); gather form: every
// input cell sums dy of the windows in which it is that first maximum => deterministic, no atomics
__global__ void maxpool_bwd_kernel(const PoolArgs a) {
  const int64_t total = (int64_t)a.n * a.h * a.w * a.c;
  for (int64_t i = (int64_t)blockIdx.x * T + threadIdx.x; i < total; i += (int64_t)gridDim.x * T) {
    const int c = (int)(i % a.c);
    int64_t p = i / a.c;
    const int iw = (int)(p % a.w); p /= a.w;
    const int ih = (int)(p % a.h);
    const int n_ = (int)(p / a.h);
    const float xv = a.x[i];
    float g = 0.f;
    for (int kh = 0; kh < a.k; ++kh) {
      const int ohs = ih + a.pt - kh;
      if (ohs < 0 || ohs % a.s) continue;
      const int oh_ = ohs / a.s;
      if (oh_ >= a.oh) continue;
      for (int kw = 0; kw < a.k; ++kw) {
        const int ows = iw + a.pl - kw;
        if (ows < 0 || ows % a.s) continue;
        const int ow_ = ows / a.s;
        if (ow_ >= a.ow) continue;
        // is (ih, iw) the first max of window (oh_, ow_)?
        bool first = true;
        for (int a2 = 0; a2 < a.k && first; ++a2) {
          const int yy = oh_ * a.s - a.pt + a2;
          if ((unsigned)yy >= (unsigned)a.h) continue;
          for (int b2 = 0; b2 < a.k; ++b2) {
            const int xx = ow_ * a.s - a.pl + b2;
            if ((unsigned)xx >= (unsigned)a.w) continue;
            const float v = a.x[((size_t)(n_ * a.h + yy) * a.w + xx) * a.c + c];
            const bool before = (a2 < kh) || (a2 == kh && b2 < kw);
            if (v > xv || (before && v == xv)) { first = false; break; }
          }
        }
        if (first) g += a.dy[((size_t)(n_ * a.oh + oh_) * a.ow + ow_) * a.c + c];
      }
    }
    a.out[i] = g;
  }
}

// ---- 2x2/2 (k x k / s) average pool, TF SAME: divide by the number of VALID cells (densenet.py:144)
__global__ void avgpool_fwd_kernel(const PoolArgs a) {
  const int CQ = a.c >> 2;
  const int64_t total = (int64_t)a.n * a.oh * a.ow * CQ;
  for (int64_t i = (int64_t)blockIdx.x * T + threadIdx.x; i < total; i += (int64_t)gridDim.x * T) {
    const int q4 = (int)(i % CQ);
    int64_t p = i / CQ;
    const int ow_ = (int)(p % a.ow); p /= a.ow;
    const int oh_ = (int)(p % a.oh);
    const int n_ = (int)(p / a.oh);
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
    int cnt = 0;
    for (int kh = 0; kh < a.k; ++kh) {
      const int ih = oh_ * a.s - a.pt + kh;
      if ((unsigned)ih >= (unsigned)a.h) continue;
      for (int kw = 0; kw < a.k; ++kw) {
        const int iw = ow_ * a.s - a.pl + kw;
        if ((unsigned)iw >= (unsigned)a.w) continue;
        const float4 v = *reinterpret_cast<const float4*>(a.x + ((size_t)(n_ * a.h + ih) * a.w + iw) * a.c + q4 * 4);
        sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
        ++cnt;
      }
    }
    const float inv = 1.f / (float)cnt;
    *reinterpret_cast<float4*>(a.out + (size_t)i * 4) = make_float4(sum.x * inv, sum.y * inv, sum.z * inv, sum.w * inv);
  }
}

__global__ void avgpool_bwd_kernel(const PoolArgs a) {
  const int CQ = a.c >> 2;
  const int64_t total = (int64_t)a.n * a.h * a.w * CQ;
  for (int64_t i = (int64_t)blockIdx.x * T + threadIdx.x; i < total; i += (int64_t)gridDim.x * T) {
    const int q4 = (int)(i % CQ);
    int64_t p = i / CQ;
    const int iw = (int)(p % a.w); p /= a.w;
    const int ih = (int)(p % a.h);
    const int n_ = (int)(p / a.h);
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int kh = 0; kh < a.k; ++kh) {
      const int ohs = ih + a.pt - kh;
      if (ohs < 0 || ohs % a.s) continue;
      const int oh_ = ohs / a.s;
      if (oh_ >= a.oh) continue;
      for (int kw = 0; kw < a.k; ++kw) {
        const int ows = iw + a.pl - kw;
        if (ows < 0 || ows % a.s) continue;
        const int ow_ = ows / a.s;
        if (ow_ >= a.ow) continue;
        // valid cells of that window
        const int y0 = oh_ * a.s - a.pt, x0 = ow_ * a.s - a.pl;
        const int vy = min(y0 + a.k, a.h) - max(y0, 0), vx = min(x0 + a.k, a.w) - max(x0, 0);
        const float inv = 1.f / (float)(vy * vx);
        const float4 d = *reinterpret_cast<const float4*>(a.dy + ((size_t)(n_ * a.oh + oh_) * a.ow + ow_) * a.c + q4 * 4);
        g.x += d.x * inv; g.y += d.y * inv; g.z += d.z * inv; g.w += d.w * inv;
      }
    }
    *reinterpret_cast<float4*>(a.out + (size_t)i * 4) = g;
  }
}

int fill_pool(PoolArgs* a, int n, int h, int w, int c, int k, int s) {
  RN_CHECK_ARG(n >= 1 && h >= 1 && w >= 1 && c >= 1 && k >= 1 && s >= 1, "pool: bad shape");
  RN_UNSUPPORTED(c % 4 != 0, "pool: c=%d not a multiple of 4", c);
  a->n = n; a->h = h; a->w = w; a->c = c; a->k = k; a->s = s;
  rn::same_pad(h, k, s, &a->oh, &a->pt);
  rn::same_pad(w, k, s, &a->ow, &a->pl);
  return RN_OK;
}

unsigned grid_for(int64_t total) {
  int64_t b = (total + T - 1) / T;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (unsigned)b;
}
}  // namespace

extern "C" int rn_act_fwd(const float* x, float* y, int64_t count, int act, rn_stream_t stream) {
  RN_CHECK_ARG(x && y && count >= 0, "act fwd: bad argument");
  if (count == 0) return RN_OK;
  hipLaunchKernelGGL(act_fwd_kernel, dim3(grid_for(count)), dim3(T), 0, (hipStream_t)stream, x, y, count, act);
  RN_LAUNCH_CHECK();
  return RN_OK;
}

extern "C" int rn_act_bwd(const float* x, const float* dy, float* dx, int64_t count, int act, rn_stream_t stream) {
  RN_CHECK_ARG(x && dy && dx && count >= 0, "act bwd: bad argument");
  if (count == 0) return RN_OK;
  hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for(count)), dim3(T), 0, (hipStream_t)stream, x, dy, dx, count, act);
  RN_LAUNCH_CHECK();
  return RN_OK;
}

extern "C" int rn_upsample_add_fwd(const float* lateral, const float* top, float* y, int n, int h, int w, int th, int tw,
                                   int c, rn_stream_t stream) {
  RN_CHECK_ARG(lateral && top && y && n >= 1 && h >= 1 && w >= 1 && th >= 1 && tw >= 1, "upsample_add: bad argument");
  RN_UNSUPPORTED(c % 4 != 0, "upsample_add: c=%d not a multiple of 4", c);
  hipLaunchKernelGGL(upsample_add_kernel, dim3(grid_for((int64_t)n * h * w * (c / 4))), dim3(T), 0, (hipStream_t)stream,
                     (const void*)lateral, (const void*)top, (void*)y, n, h, w, th, tw, c, 0);
  RN_LAUNCH_CHECK();
  return RN_OK;
}

extern "C" int rn_upsample_add_fwd_f16(const void* lateral, const void* top, void* y, int n, int h, int w, int th, int tw,
                                       int c, rn_stream_t stream) {
  RN_CHECK_ARG(lateral && top && y && n >= 1 && h >= 1 && w >= 1 && th >= 1 && tw >= 1, "upsample_add f16: bad argument");
  RN_UNSUPPORTED(c % 4 != 0, "upsample_add f16: c=%d not a multiple of 4", c);
  hipLaunchKernelGGL(upsample_add_kernel, dim3(grid_for((int64_t)n * h * w * (c / 4))), dim3(T), 0, (hipStream_t)stream,
                     lateral, top, y, n, h, w, th, tw, c, 1);
  RN_LAUNCH_CHECK();
  return RN_OK;
}

extern "C" int rn_upsample_add_bwd_top(const float* dy, float* dtop, int n, int h, int w, int th, int tw, int c,
                                       rn_stream_t stream) {
  RN_CHECK_ARG(dy && dtop && n >= 1 && h >= 1 && w >= 1 && th >= 1 && tw >= 1, "upsample_add bwd: bad argument");
  RN_UNSUPPORTED(c % 4 != 0, "upsample_add bwd: c=%d not a multiple of 4", c);
  hipLaunchKernelGGL(upsample_add_bwd_kernel, dim3(grid_for((int64_t)n * th * tw * (c / 4))), dim3(T), 0,
                     (hipStream_t)stream, dy, dtop, n, h, w, th, tw, c);
  RN_LAUNCH_CHECK();
  return RN_OK;
}

extern "C" int rn_dropout(const float* x, float* y, int64_t count, float rate, uint64_t seed, const uint64_t* seed_dev,
                          rn_stream_t stream) {
  RN_CHECK_ARG(x && y && count >= 0 && rate >= 0.f && rate < 1.f, "dropout: bad argument");
  if (count == 0) return RN_OK;
  hipLaunchKernelGGL(dropout_kernel, dim3(grid_for(count)), dim3(T), 0, (hipStream_t)stream, x, y, count, rate, seed, seed_dev);
  RN_LAUNCH_CHECK();
  return RN_OK;
}

// backward from the forward pass's arg-max taps: each input cell (4 channels per thread) adds dy of the <= (k/s)^2
// windows whose first maximum it is -- gather form, deterministic, 5 bytes read per window instead of re-scanning it
__global__ void maxpool_bwd_arg_kernel(const PoolArgs a) {
  const int CQ = a.c >> 2;
  const int64_t total = (int64_t)a.n * a.h * a.w * CQ;
  for (int64_t i = (int64_t)blockIdx.x * T + threadIdx.x; i < total; i += (int64_t)gridDim.x * T) {
    const int q4 = (int)(i % CQ);
    int64_t p = i / CQ;
    const int iw = (int)(p % a.w); p /= a.w;
    const int ih = (int)(p % a.h);
    const int n_ = (int)(p / a.h);
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int kh = 0; kh < a.k; ++kh) {
      const int ohs = ih + a.pt - kh;
      if (ohs < 0 || ohs % a.s) continue;
      const int oh_ = ohs / a.s;
      if (oh_ >= a.oh) continue;
      for (int kw = 0; kw < a.k; ++kw) {
        const int ows = iw + a.pl - kw;
        if (ows < 0 || ows % a.s) continue;
        const int ow_ = ows / a.s;
        if (ow_ >= a.ow) continue;
        const size_t o = ((size_t)(n_ * a.oh + oh_) * a.ow + ow_) * a.c + q4 * 4;
        const uchar4 am = *reinterpret_cast<const uchar4*>(a.arg + o);
        const float4 d = *reinterpret_cast<const float4*>(a.dy + o);
        const unsigned char t = (unsigned char)(kh * a.k + kw);
        if (am.x == t) g.x += d.x;
        if (am.y == t) g.y += d.y;
        if (am.z == t) g.z += d.z;
        if (am.w == t) g.w += d.w;
      }
    }
    *reinterpret_cast<float4*>(a.out + (size_t)i * 4) = g;
  }
}

extern "C" int rn_maxpool_fwd(const float* x, float* y, uint8_t* argmax, int n, int h, int w, int c, int k, int stride,
                              rn_stream_t stream) {
  PoolArgs a = {};
  if (int e = fill_pool(&a, n, h, w, c, k, stride)) return e;
  RN_CHECK_ARG(x && y, "maxpool fwd: null pointer");
  RN_UNSUPPORTED(argmax && k * k > 255, "maxpool fwd: window %dx%d too large for byte tap indices", k, k);
  a.x = x; a.out = y; a.arg = argmax;
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid_for((int64_t)n * a.oh * a.ow * (c / 4))), dim3(T), 0, (hipStream_t)stream, a);
  RN_LAUNCH_CHECK();
  return RN_OK;
}

extern "C" int rn_maxpool_fwd_f16(const void* x, void* y, int n, int h, int w, int c, int k, int stride, rn_stream_t stream) {
  PoolArgs a = {};
  if (int e = fill_pool(&a, n, h, w, c, k, stride)) return e;
  RN_CHECK_ARG(x && y, "maxpool fwd f16: null pointer");
  a.x = (const float*)x; a.out = (float*)y; a.is_half = 1;
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid_for((int64_t)n * a.oh * a.ow * (c / 4))), dim3(T), 0, (hipStream_t)stream, a);
  RN_LAUNCH_CHECK();
  return RN_OK;
}

extern "C" int rn_maxpool_gn_fwd_f16(const void* x, void* y, int n, int h, int w, int c, int k, int stride, const float* mean,
                                    const float* rstd, const float* gamma, const float* beta, int groups, int act, rn_stream_t stream) {
  PoolArgs pa = {};
  if (int e = fill_pool(&pa, n, h, w, c, k, stride)) return e;
  RN_CHECK_ARG(x && y && mean && rstd && gamma && beta && groups >= 1 && c % groups == 0, "maxpool gn fwd f16: bad argument");
  RN_UNSUPPORTED(c % 8 != 0 || c > 2048, "maxpool gn fwd f16: c=%d must be a multiple of 8 and <= 2048", c);
  PoolGnArgs a = {};
  a.x = (const _Float16*)x; a.out = (_Float16*)y; a.mean = mean; a.rstd = rstd; a.gamma = gamma; a.beta = beta;
  a.n = n; a.h = h; a.w = w; a.c = c; a.k = k; a.s = stride; a.oh = pa.oh; a.ow = pa.ow; a.pt = pa.pt; a.pl = pa.pl;
  a.groups = groups; a.act = act;
  const int64_t per = (int64_t)a.oh * a.ow * (c / 8);
  int64_t bx = (per + T - 1) / T;
  if (bx > 4096) bx = 4096;
  hipLaunchKernelGGL(maxpool_gn_f16_kernel, dim3((unsigned)bx, (unsigned)n), dim3(T), 0, (hipStream_t)stream, a);
  RN_LAUNCH_CHECK();
  return RN_OK;
}

extern "C" int rn_maxpool_bwd(const float* x, const float* dy, float* dx, int n, int h, int w, int c, int k, int stride,
                              rn_stream_t stream) {
  PoolArgs a = {};
  if (int e = fill_pool(&a, n, h, w, c, k, stride)) return e;
  RN_CHECK_ARG(x && dy && dx, "maxpool bwd: null pointer");
  a.x = x; a.dy = dy; a.out = dx;
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for((int64_t)n * h * w * c)), dim3(T), 0, (hipStream_t)stream, a);
  RN_LAUNCH_CHECK();
  return RN_OK;
}

extern "C" int rn_maxpool_bwd_arg(const uint8_t* argmax, const float* dy, float* dx, int n, int h, int w, int c, int k, int stride,
                                  rn_stream_t stream) {
  PoolArgs a = {};
  if (int e = fill_pool(&a, n, h, w, c, k, stride)) return e;
  RN_CHECK_ARG(argmax && dy && dx, "maxpool bwd: null pointer");
  a.arg = const_cast<uint8_t*>(argmax); a.dy = dy; a.out = dx;
  hipLaunchKernelGGL(maxpool_bwd_arg_kernel, dim3(grid_for((int64_t)n * h * w * (c / 4))), dim3(T), 0, (hipStream_t)stream, a);
  RN_LAUNCH_CHECK();
  return RN_OK;
}

extern "C" int rn_avgpool_fwd(const float* x, float* y, int n, int h, int w, int c, int k, int stride, rn_stream_t stream) {
  PoolArgs a = {};
  if (int e = fill_pool(&a, n, h, w, c, k, stride)) return e;
  RN_CHECK_ARG(x && y, "avgpool fwd: null pointer");
  a.x = x; a.out = y;
  hipLaunchKernelGGL(avgpool_fwd_kernel, dim3(grid_for((int64_t)n * a.oh * a.ow * (c / 4))), dim3(T), 0, (hipStream_t)stream, a);
  RN_LAUNCH_CHECK();
  return RN_OK;
}

extern "C" int rn_avgpool_bwd(const float* dy, float* dx, int n, int h, int w, int c, int k, int stride, rn_stream_t stream) {
  PoolArgs a = {};
  if (int e = fill_pool(&a, n, h, w, c, k, stride)) return e;
  RN_CHECK_ARG(dy && dx, "avgpool bwd: null pointer");
  a.dy = dy; a.out = dx;
  hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(grid_for((int64_t)n * h * w * (c / 4))), dim3(T), 0, (hipStream_t)stream, a);
  RN_LAUNCH_CHECK();
  return RN_OK;
}

extern "C" int rn_act_fwd_f16(const void* x, void* y, int64_t count, int act, rn_stream_t stream) {
  RN_CHECK_ARG(x && y && count >= 0 && count % 4 == 0, "act fwd f16: bad argument (count must be a multiple of 4)");
  if (count == 0) return RN_OK;
  hipLaunchKernelGGL(act_fwd_f16_kernel, dim3(grid_for(count / 4)), dim3(T), 0, (hipStream_t)stream, x, y, count / 4, act);
  RN_LAUNCH_CHECK();
  return RN_OK;
}

// ---- horizontal flip of an [outer, W, inner] tensor (augmentation.flip, augmentation.py:5-22): reverse W;
// for regression maps additionally negate component `neg_idx` of every group of `neg_mod` values (the x shift)
namespace {
template <typename TT>
__global__ void flip_w_kernel(const TT* __restrict__ x, TT* __restrict__ y, int64_t outer, int w, int64_t inner, int neg_mod,
                              int neg_idx) {
  const int64_t total = outer * w * inner;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t in_ = i % inner;
    const int64_t r = i / inner;
    const int j = (int)(r % w);
    const int64_t o = r / w;
    TT v = x[(o * w + (w - 1 - j)) * inner + in_];
    if (neg_mod > 0 && (in_ % neg_mod) == neg_idx) v = (TT)(-(float)v);
    y[i] = v;
  }
}
}  // namespace

extern "C" int rn_flip_width(const void* x, void* y, int64_t outer, int w, int64_t inner, int elem_bytes, int neg_mod,
                             int neg_idx, rn_stream_t stream) {
  RN_CHECK_ARG(x && y && outer >= 0 && w >= 1 && inner >= 1 && (elem_bytes == 1 || elem_bytes == 4), "flip_width: bad argument");
  RN_CHECK_ARG(elem_bytes == 4 || neg_mod == 0, "flip_width: negation needs fp32 data");
  const int64_t total = outer * w * inner;
  if (total == 0) return RN_OK;
  int64_t b = (total + 255) / 256;
  if (b > 4096) b = 4096;
  if (elem_bytes == 4)
    hipLaunchKernelGGL(flip_w_kernel<float>, dim3((unsigned)b), dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)y,
                       outer, w, inner, neg_mod, neg_idx);
  else
    hipLaunchKernelGGL(flip_w_kernel<uint8_t>, dim3((unsigned)b), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)x,
                       (uint8_t*)y, outer, w, inner, 0, 0);
  RN_LAUNCH_CHECK();
  return RN_OK;
}

// Stand-in for a ring all-reduce sharing the GPU with the backward pass (measurement aid, rn_hip.h): `blocks` workgroups of
// 256 threads stream `bytes` from src to dst, paced by the constant 100 MHz clock so that the whole copy takes ~target_us.
namespace {
__global__ __launch_bounds__(256) void standin_kernel(const float4* __restrict__ src, float4* __restrict__ dst, int64_t n16,
                                                      int64_t ticks_total) {
  constexpr int64_t CHUNK = 4096;                       // float4 per block and round: 64 KB
  const int64_t per = (n16 + gridDim.x - 1) / gridDim.x;
  const int64_t lo = per * blockIdx.x, hi = min(lo + per, n16);
  const int64_t rounds = (per + CHUNK - 1) / CHUNK;
  const uint64_t t0 = wall_clock64();
  for (int64_t r = 0; r < rounds; ++r) {
    const int64_t b = lo + r * CHUNK, e = min(b + CHUNK, hi);
    for (int64_t i = b + threadIdx.x; i < e; i += 256) dst[i] = src[i];
    // pace: round r may end no earlier than its share of the target duration
    const uint64_t due = t0 + (uint64_t)(ticks_total * (r + 1) / rounds);
    while (wall_clock64() < due) __builtin_amdgcn_s_sleep(32);
  }
}
}  // namespace

extern "C" int rn_debug_collective_standin(const void* src, void* dst, int64_t bytes, int blocks, float target_us, rn_stream_t stream) {
  RN_CHECK_ARG(src && dst && bytes >= 16 && bytes % 16 == 0 && blocks >= 1 && blocks <= 1024 && target_us >= 0.f && target_us < 1e6f,
               "collective_standin: bad argument");
  hipLaunchKernelGGL(standin_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const float4*)src, (float4*)dst,
                     bytes / 16, (int64_t)(target_us * 100.0f));
  RN_LAUNCH_CHECK();
  return RN_OK;
}

// out_s = a_s + b_s for up to RN_MAX_SEG tensors in one launch: the sum of the two gradients of a tensor that two
// branches consume (a bottleneck's input: expand conv + residual; a pyramid level: class + box subnet).
namespace {
struct AddArgs { rn_add_seg seg[RN_MAX_SEG]; int64_t start[RN_MAX_SEG + 1]; int nseg; };
__global__ __launch_bounds__(256) void add_segs_kernel(const AddArgs a) {
  const int64_t total = a.start[a.nseg];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int s = 0;
    while (s + 1 < a.nseg && i >= a.start[s + 1]) ++s;
    const int64_t q = i - a.start[s];            // quad index inside the segment
    const rn_add_seg& g = a.seg[s];
    if (q * 4 + 4 <= g.count) {
      const float4 x = *reinterpret_cast<const float4*>(g.a + q * 4), y = *reinterpret_cast<const float4*>(g.b + q * 4);
      *reinterpret_cast<float4*>(g.out + q * 4) = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
    } else {
      for (int64_t j = q * 4; j < g.count; ++j) g.out[j] = g.a[j] + g.b[j];
    }
  }
}
}  // namespace

extern "C" int rn_add_segs(const rn_add_seg* segs, int nseg, rn_stream_t stream) {
  RN_CHECK_ARG(segs && nseg >= 1 && nseg <= RN_MAX_SEG, "add_segs: nseg outside [1,%d]", RN_MAX_SEG);
  AddArgs a = {};
  a.nseg = nseg;
  int64_t quads = 0;
  for (int s = 0; s < nseg; ++s) {
    RN_CHECK_ARG(segs[s].a && segs[s].b && segs[s].out && segs[s].count >= 1, "add_segs: bad segment %d", s);
    RN_CHECK_ARG((((uintptr_t)segs[s].a | (uintptr_t)segs[s].b | (uintptr_t)segs[s].out) & 15) == 0, "add_segs: segment %d not 16-byte aligned", s);
    a.seg[s] = segs[s];
    a.start[s] = quads;
    quads += (segs[s].count + 3) / 4;
  }
  a.start[nseg] = quads;
  int64_t blocks = (quads + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(add_segs_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  RN_LAUNCH_CHECK();
  return RN_OK;
}

// Dropout (rate 0: a plain copy) between channel slices of wider buffers: the growth layers of a concat-free DenseNet
// block write their k channels straight into the block's buffer, and read their gradient slice out of it.
namespace {
__global__ __launch_bounds__(256) void dropout_strided_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t pixels, int c,
                                                              int x_ld, int x_coff, int y_ld, int y_coff, float rate, uint64_t seed,
                                                              const uint64_t* __restrict__ seed_dev) {
  const uint64_t sd = seed + (seed_dev ? *seed_dev : 0ull);
  const float ks = rate > 0.f ? 1.f / (1.f - rate) : 1.f;
  const int cq = c >> 2;
  const int64_t total = pixels * cq;
  const bool small = total < (1ll << 31);   // a 64-bit division by a runtime divisor is ~150 instructions per element
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t p = small ? (int64_t)((unsigned)i / (unsigned)cq) : i / cq;
    const int q = (int)(i - p * cq);
    float4 v = *reinterpret_cast<const float4*>(x + p * x_ld + x_coff + q * 4);
    if (rate > 0.f) {
      const uint64_t e = (uint64_t)p * (uint64_t)c + (uint64_t)q * 4;   // element index in the dense [pixels, c] tensor
      v.x = rn::uniform01(sd, e) >= rate ? v.x * ks : 0.f;
      v.y = rn::uniform01(sd, e + 1) >= rate ? v.y * ks : 0.f;
      v.z = rn::uniform01(sd, e + 2) >= rate ? v.z * ks : 0.f;
      v.w = rn::uniform01(sd, e + 3) >= rate ? v.w * ks : 0.f;
    }
    *reinterpret_cast<float4*>(y + p * y_ld + y_coff + q * 4) = v;
  }
}
}  // namespace

extern "C" int rn_dropout_strided(const float* x, float* y, int64_t pixels, int c, int x_ld, int x_coff, int y_ld, int y_coff,
                                  float rate, uint64_t seed, const uint64_t* seed_dev, rn_stream_t stream) {
  RN_CHECK_ARG(x && y && pixels >= 0 && c >= 4 && rate >= 0.f && rate < 1.f, "dropout_strided: bad argument");
  RN_CHECK_ARG(c % 4 == 0 && x_ld % 4 == 0 && y_ld % 4 == 0 && x_coff % 4 == 0 && y_coff % 4 == 0 && x_coff >= 0 && y_coff >= 0 &&
                   x_ld >= x_coff + c && y_ld >= y_coff + c, "dropout_strided: slices must be 16-byte aligned and inside their rows");
  if (pixels == 0) return RN_OK;
  hipLaunchKernelGGL(dropout_strided_kernel, dim3(grid_for(pixels * (c / 4))), dim3(256), 0, (hipStream_t)stream, x, y, pixels, c, x_ld,
                     x_coff, y_ld, y_coff, rate, seed, seed_dev);
  RN_LAUNCH_CHECK();
  return RN_OK;
}
