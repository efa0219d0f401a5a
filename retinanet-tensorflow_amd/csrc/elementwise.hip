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

// [TF-sem] src = min(roundf(dst * (in-1)/(out-1)), in-1); scale evaluated in float32
__device__ __forceinline__ int nn_src(int dst, int in, int out) {
  const float scale = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
  const int s = (int)roundf((float)dst * scale);
  return s < in - 1 ? s : in - 1;
}

__global__ void upsample_add_kernel(const float* __restrict__ lat, const float* __restrict__ top, float* __restrict__ y,
                                    int n, int h, int w, int th, int tw, int c) {
  const int CQ = c >> 2;
  const int64_t total = (int64_t)n * h * w * CQ;
  for (int64_t i = (int64_t)blockIdx.x * T + threadIdx.x; i < total; i += (int64_t)gridDim.x * T) {
    const int q4 = (int)(i % CQ);
    int64_t p = i / CQ;
    const int x_ = (int)(p % w); p /= w;
    const int y_ = (int)(p % h);
    const int n_ = (int)(p / h);
    const int sy = nn_src(y_, th, h), sx = nn_src(x_, tw, w);
    const float4 a = *reinterpret_cast<const float4*>(lat + (size_t)i * 4);
    const float4 b = *reinterpret_cast<const float4*>(top + ((size_t)(n_ * th + sy) * tw + sx) * c + q4 * 4);
    *reinterpret_cast<float4*>(y + (size_t)i * 4) = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
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
                     lateral, top, y, n, h, w, th, tw, c);
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
