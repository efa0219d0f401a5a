// Dense k x k convolutions that are NOT Winograd layers (3 x 3 / stride 2: the ResNeXt "down" identity convs, resnet.py:60-69, 141-150)
// on the split-bf16 product kernels of gemm_x3.hip, through an explicit patch matrix (round 6).
//
// At the cfg-3 shape these three convs are 47 GFLOP each per pass and ran at the fp32 matrix-core roofline (393 - 580 us per launch,
// 98 - 120 TFLOP/s): 4.3 ms of a 19.5 ms step.  The split-bf16 kernel holds ~175 TFLOP/s-equivalent on products with a long K, but it
// multiplies plain matrices; the patch matrix col [M = n oh ow][K = kh kw cin] is 46 - 184 MB here -- written and read once at
// ~5 TB/s that is 20 - 75 us, a fraction of what the product saves:
//   forward   col = im2col(x);  y = col W                       (W in HWIO IS the [K][cout] matrix)
//   dgrad     dcol = dy W^T;    dx = col2im(dcol)               (a gather: every input pixel sums the <= kh kw entries that read it)
//   wgrad     col = im2col(x);  dW = col^T dy                   (split over the pixels, fixed-order row reduction)
// A wave copies the cin channels of one (output pixel, tap) as 16-byte pieces: 256-byte .. 4 KB contiguous runs on both sides.
#include "rn_common.h"

namespace {
struct ColArgs {
  const float* src; float* dst;
  int n, h, wd, cin, oh, ow, kh, kw, stride, pad_t, pad_l;
};

// col[m][tap][ci] = x[n, oy s - pad_t + kh, ox s - pad_l + kw, ci] (0 outside); one wave per (m, tap), lanes over 4-channel pieces
__global__ __launch_bounds__(256) void im2col_kernel(const ColArgs a) {
  const int lane = threadIdx.x & 63;
  const int ntap = a.kh * a.kw, q4 = a.cin >> 2;
  const long nwork = (long)a.n * a.oh * a.ow * ntap;
  for (long wk = (long)blockIdx.x * 4 + (threadIdx.x >> 6); wk < nwork; wk += (long)gridDim.x * 4) {
    const int tap = (int)(wk % ntap);
    const long m = wk / ntap;
    const int ox = (int)(m % a.ow);
    const long t = m / a.ow;
    const int oy = (int)(t % a.oh), s = (int)(t / a.oh);
    const int kh = tap / a.kw, kw = tap - kh * a.kw;
    const int iy = oy * a.stride - a.pad_t + kh, ix = ox * a.stride - a.pad_l + kw;
    const bool ok = (unsigned)iy < (unsigned)a.h && (unsigned)ix < (unsigned)a.wd;
    const float4* src = reinterpret_cast<const float4*>(a.src + ((size_t)(s * a.h + (ok ? iy : 0)) * a.wd + (ok ? ix : 0)) * a.cin);
    float4* dst = reinterpret_cast<float4*>(a.dst + ((size_t)m * ntap + tap) * a.cin);
    for (int q = lane; q < q4; q += 64) dst[q] = ok ? src[q] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

// dx[n, iy, ix, ci] = sum over the taps (kh, kw) whose window origin is an output pixel: dcol[m(n, oy, ox)][tap][ci], taps in
// ascending order (a fixed order: reproducible); one wave per input pixel
__global__ __launch_bounds__(256) void col2im_kernel(const ColArgs a) {
  const int lane = threadIdx.x & 63;
  const int ntap = a.kh * a.kw, q4 = a.cin >> 2;
  const long npix = (long)a.n * a.h * a.wd;
  for (long p = (long)blockIdx.x * 4 + (threadIdx.x >> 6); p < npix; p += (long)gridDim.x * 4) {
    const int ix = (int)(p % a.wd);
    const long t = p / a.wd;
    const int iy = (int)(t % a.h), s = (int)(t / a.h);
    for (int q = lane; q < q4; q += 64) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int kh = 0; kh < a.kh; ++kh) {
        const int oys = iy + a.pad_t - kh;
        if (oys < 0 || oys % a.stride) continue;
        const int oy = oys / a.stride;
        if (oy >= a.oh) continue;
        for (int kw = 0; kw < a.kw; ++kw) {
          const int oxs = ix + a.pad_l - kw;
          if (oxs < 0 || oxs % a.stride) continue;
          const int ox = oxs / a.stride;
          if (ox >= a.ow) continue;
          const size_t m = ((size_t)s * a.oh + oy) * a.ow + ox;
          const float4 v = reinterpret_cast<const float4*>(a.src + (m * ntap + kh * a.kw + kw) * a.cin)[q];
          acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
      }
      reinterpret_cast<float4*>(a.dst + (size_t)p * a.cin)[q] = acc;
    }
  }
}

const bool g_on = !(getenv("RN_X3_IM2COL") && atoi(getenv("RN_X3_IM2COL")) == 0);
}  // namespace

namespace rn {
size_t im2col_x3_bytes(int n, int h, int wd, int cin, int cout, int kh, int kw, int stride);
// patch-matrix bytes of the conv, or 0 when the path is not taken (product mode 0, RN_X3_IM2COL=0, channels not multiples of 4, 1 x 1 /
// stride-1 kernels -- those are plain products already --, too few tiles, or > 2 GiB)
// forward only: the batch in equal pieces of `*n_piece` samples whose patch matrix stays under 1 GiB (large inference batches: the first
// "down" conv of a 1024^2 batch of 16 would need 2.4 GB at once) -- bytes of ONE piece, or 0 when the path is not taken
size_t im2col_x3_fwd_pieces(int n, int h, int wd, int cin, int cout, int kh, int kw, int stride, int* n_piece) {
  for (int np = n; np >= 1; --np) {
    if (n % np) continue;
    int oh, ow, pt, pl;
    rn::same_pad(h, kh, stride, &oh, &pt);
    rn::same_pad(wd, kw, stride, &ow, &pl);
    static const double cap = (getenv("RN_X3_IM2COL_PIECE_MB") ? atof(getenv("RN_X3_IM2COL_PIECE_MB")) : 1024.0) * 1048576.0;   // (tests: a small cap)
    if ((double)np * oh * ow * kh * kw * cin * 4.0 > cap && np > 1) continue;
    *n_piece = np;
    return im2col_x3_bytes(np, h, wd, cin, cout, kh, kw, stride);
  }
  return 0;
}

size_t im2col_x3_bytes(int n, int h, int wd, int cin, int cout, int kh, int kw, int stride) {
  if (!g_on || rn::product_mode() != 1 || cin % 4 || cout % 4 || (kh == 1 && kw == 1 && stride == 1)) return 0;
  int oh, ow, pt, pl;
  rn::same_pad(h, kh, stride, &oh, &pt);
  rn::same_pad(wd, kw, stride, &ow, &pl);
  const long M = (long)n * oh * ow, K = (long)kh * kw * cin;
  if (M < 1 || M > 0x7fffffffL || K > 0x7fffffffL || (double)M * K * 4.0 >= 2147483648.0) return 0;
  if (!rn::conv1x1_x3_tile(M, (int)K, cout, (int)K)) return 0;
  if (rn::ceil_div64(M, 128) * rn::ceil_div(cout, 128) < 96) return 0;      // small grids keep the fp32 split-K path
  // profitable only where the product dominates the patch matrix's traffic: 2 M K N FLOP gain ~4e-15 s each over the fp32 kernels,
  // the matrix costs ~2 x 4 M K bytes at 5 TB/s -- break-even at N ~ 200 output channels; taken from 512 (the ResNeXt "down" convs:
  // 512 / 1 024 / 2 048), NOT for e.g. DenseNet's 3x3 convs into 32 channels
  if (cout < 512 || cin < 128) return 0;
  return (size_t)M * K * sizeof(float);
}

int launch_im2col(const float* x, float* col, int n, int h, int wd, int cin, int kh, int kw, int stride, hipStream_t st) {
  ColArgs a = {x, col, n, h, wd, cin, 0, 0, kh, kw, stride, 0, 0};
  rn::same_pad(h, kh, stride, &a.oh, &a.pad_t);
  rn::same_pad(wd, kw, stride, &a.ow, &a.pad_l);
  const long nwork = (long)n * a.oh * a.ow * kh * kw;
  const long blocks = rn::ceil_div64(nwork, 4);
  hipLaunchKernelGGL(im2col_kernel, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, st, a);
  RN_LAUNCH_CHECK();
  return RN_OK;
}

int launch_col2im(const float* dcol, float* dx, int n, int h, int wd, int cin, int kh, int kw, int stride, hipStream_t st) {
  ColArgs a = {dcol, dx, n, h, wd, cin, 0, 0, kh, kw, stride, 0, 0};
  rn::same_pad(h, kh, stride, &a.oh, &a.pad_t);
  rn::same_pad(wd, kw, stride, &a.ow, &a.pad_l);
  const long blocks = rn::ceil_div64((long)n * h * wd, 4);
  hipLaunchKernelGGL(col2im_kernel, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, st, a);
  RN_LAUNCH_CHECK();
  return RN_OK;
}
}  // namespace rn
