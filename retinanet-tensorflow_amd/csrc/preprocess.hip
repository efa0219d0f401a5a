// Input pipeline on the device: convert_image_dtype (uint8 -> [0,1] float), bilinear resize with
// align_corners=True (dataset.py:145-151 rescale_image -> tf.image.resize_images, the ResizeBilinear kernel with
// half_pixel_centers = false) and the mean / std normalisation of train.py:48-49, fused in one pass.
// Compiled with -ffp-contract=off: every operation rounds separately, in the order of the TF kernel
//   in = dst * scale;  lo = floor(in);  hi = min(ceil(in), size-1);  lerp = in - lo
//   top = tl + (tr - tl) * xl;  bot = bl + (br - bl) * xl;  out = top + (bot - top) * yl
// HBM-bound: 4 gathered reads (cached) + 1 write per output element.
#include "rn_common.h"

namespace {
struct ResizeArgs {
  const void* x; float* y;
  int n, h, w, c, oh, ow, in_u8, normalize;
  float hs, ws;  // (in-1)/(out-1) with align_corners, 0 when out == 1
  float mean[8], stdv[8];
};

__device__ __forceinline__ float fetch(const ResizeArgs& a, size_t idx) {
  if (a.in_u8) return (float)reinterpret_cast<const uint8_t*>(a.x)[idx] * (1.0f / 255.0f);  // convert_image_dtype
  return reinterpret_cast<const float*>(a.x)[idx];
}

__global__ __launch_bounds__(256) void resize_bilinear_kernel(const ResizeArgs a) {
  const int64_t total = (int64_t)a.n * a.oh * a.ow * a.c;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int64_t q = i;
    const int ch = (int)(q % a.c); q /= a.c;
    const int ox = (int)(q % a.ow); q /= a.ow;
    const int oy = (int)(q % a.oh);
    const int n_ = (int)(q / a.oh);
    const float iny = (float)oy * a.hs, inx = (float)ox * a.ws;
    const float fy = floorf(iny), fx = floorf(inx);
    const int y0 = max((int)fy, 0), x0 = max((int)fx, 0);
    const int y1 = min((int)ceilf(iny), a.h - 1), x1 = min((int)ceilf(inx), a.w - 1);
    const float yl = iny - fy, xl = inx - fx;
    const size_t base = (size_t)n_ * a.h * a.w;
    const float tl = fetch(a, ((base + (size_t)y0 * a.w + x0) * a.c) + ch), tr = fetch(a, ((base + (size_t)y0 * a.w + x1) * a.c) + ch);
    const float bl = fetch(a, ((base + (size_t)y1 * a.w + x0) * a.c) + ch), br = fetch(a, ((base + (size_t)y1 * a.w + x1) * a.c) + ch);
    const float top = tl + (tr - tl) * xl;
    const float bot = bl + (br - bl) * xl;
    float v = top + (bot - top) * yl;
    if (a.normalize) v = (v - a.mean[ch]) / a.stdv[ch];
    a.y[i] = v;
  }
}
}  // namespace

extern "C" int rn_resize_bilinear_normalize(const void* x, int in_u8, float* y, int n, int h, int w, int c, int oh, int ow,
                                            const float* mean, const float* stdv, rn_stream_t stream) {
  RN_CHECK_ARG(x && y && n >= 1 && h >= 1 && w >= 1 && c >= 1 && oh >= 1 && ow >= 1, "resize: bad argument");
  RN_UNSUPPORTED(c > 8, "resize: c = %d (at most 8 channels)", c);
  RN_CHECK_ARG((mean == nullptr) == (stdv == nullptr), "resize: mean and std go together");
  ResizeArgs a = {};
  a.x = x; a.y = y; a.n = n; a.h = h; a.w = w; a.c = c; a.oh = oh; a.ow = ow; a.in_u8 = in_u8 ? 1 : 0;
  a.hs = oh > 1 ? (float)(h - 1) / (float)(oh - 1) : 0.f;
  a.ws = ow > 1 ? (float)(w - 1) / (float)(ow - 1) : 0.f;
  a.normalize = mean ? 1 : 0;
  for (int i = 0; i < c && mean; ++i) { a.mean[i] = mean[i]; a.stdv[i] = stdv[i]; }
  const int64_t total = (int64_t)n * oh * ow * c;
  int64_t b = (total + 255) / 256;
  if (b > 16384) b = 16384;
  hipLaunchKernelGGL(resize_bilinear_kernel, dim3((unsigned)b), dim3(256), 0, (hipStream_t)stream, a);
  RN_LAUNCH_CHECK();
  return RN_OK;
}
