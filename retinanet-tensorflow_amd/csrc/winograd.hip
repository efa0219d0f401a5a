// Winograd F(m x m, 3x3), m = 2 or 4, for the 3x3 / stride-1 / SAME convolutions (the RetinaNet head towers
// and FPN merge convs: retinanet.py:39-46,87-94,138-145 -- 88 % of the network's multiply-adds at 512x512).
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A       per (m+2)x(m+2) input patch d -> m x m outputs
//
// 2.25x (m = 2) or 4x (m = 4) fewer multiply-adds than the direct form; the products still run as exact fp32
// MFMA arithmetic.  Three stages, pyramid levels (segments) concatenated along the tile axis T, P = (m+2)^2:
//   1. wino_input_kernel   : x [n,h,w,C]  -> V [P][T][C]        (B^T d B, zero padding at the borders)
//   2. P GEMMs M_xi = V_xi [T x C] * U_xi [C x Cout] as ONE batched launch of the implicit-GEMM kernels of
//      conv_gemm.hip (P x T/128 x Cout/128 tiles: fills the chip even for small pyramids)
//   3. wino_output_kernel  : M [P][T][Cout] -> y [n,h,w,Cout]   (A^T m A, + bias)
// U = G g G^T is produced per call by wino_weight_kernel (weights change every training step).
// The same three stages compute the data gradient, dX = conv(dY, rot180(W)^T): U from the rotated kernel,
// read by the GEMM in the data-gradient kernel's [N][K] layout (no transpose pass).  V and M (P/m^2 = 4x or
// 2.25x the activation size each) live in the caller's workspace and stay in the 256 MB Infinity Cache
// between the stages.  Transform matrices: Lavin & Gray 2016, points {0, +-1, (+-2,) inf}.
#include "rn_common.h"

namespace {
constexpr int NT = 256;

struct WSeg { const float* x; float* y; int n, h, w, th, tw, tile_start; };
struct WArgs {
  WSeg seg[RN_MAX_SEG];
  int nseg, c, total_tiles;
  float* buf;         // V (input transform) or M (output transform): [P][total_tiles][c]
  const float* bias;  // output transform only
};

__device__ __forceinline__ int seg_of_tile(const WArgs& a, int t) {
  int s = 0;
  while (s + 1 < a.nseg && t >= a.seg[s + 1].tile_start) ++s;
  return s;
}

// W channels per thread (clang vector types: elementwise + - and scalar * vector).  W = 4 (16-byte accesses) when
// there is enough work to fill the chip with it, else 2 or 1 -- small pyramids are latency-, not bandwidth-bound.
template <int W>
struct VecW { typedef float type __attribute__((ext_vector_type(W))); };
template <>
struct VecW<1> { typedef float type; };

template <int M>
struct Wino;

template <>
struct Wino<2> {
  // B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
  template <typename V>
  static __device__ __forceinline__ void bt(const V (&d)[4], V (&t)[4]) {
    t[0] = d[0] - d[2]; t[1] = d[1] + d[2]; t[2] = d[2] - d[1]; t[3] = d[1] - d[3];
  }
  // A^T = [1 1 1 0; 0 1 -1 -1]
  template <typename V>
  static __device__ __forceinline__ void at(const V (&m)[4], V (&o)[2]) {
    o[0] = (m[0] + m[1]) + m[2]; o[1] = (m[1] - m[2]) - m[3];
  }
  // G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]
  static __device__ __forceinline__ void g(const float (&w)[3], float (&u)[4]) {
    u[0] = w[0]; u[1] = 0.5f * (w[0] + w[1] + w[2]); u[2] = 0.5f * (w[0] - w[1] + w[2]); u[3] = w[2];
  }
  // A (4x2): the gradient of the output transform, dM = A dY A^T
  template <typename V>
  static __device__ __forceinline__ void a(const V (&y)[2], V (&m)[4]) {
    m[0] = y[0]; m[1] = y[0] + y[1]; m[2] = y[0] - y[1]; m[3] = (V)(0.f) - y[1];
  }
  // G^T (3x4): the gradient of the kernel transform, dg = G^T dU G
  static __device__ __forceinline__ void gt(const float (&u)[4], float (&w)[3]) {
    const float s = 0.5f * (u[1] + u[2]);
    w[0] = u[0] + s; w[1] = 0.5f * (u[1] - u[2]); w[2] = s + u[3];
  }
};

template <>
struct Wino<4> {
  // B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
  template <typename V>
  static __device__ __forceinline__ void bt(const V (&d)[6], V (&t)[6]) {
    const V a = d[4] - 4.f * d[2], b = d[3] - 4.f * d[1];
    const V c = d[4] - d[2], e = 2.f * (d[3] - d[1]);
    t[0] = (4.f * d[0] - 5.f * d[2]) + d[4];
    t[1] = a + b; t[2] = a - b; t[3] = c + e; t[4] = c - e;
    t[5] = (4.f * d[1] - 5.f * d[3]) + d[5];
  }
  // A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
  template <typename V>
  static __device__ __forceinline__ void at(const V (&m)[6], V (&o)[4]) {
    const V s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
    o[0] = (m[0] + s12) + s34;
    o[1] = d12 + 2.f * d34;
    o[2] = s12 + 4.f * s34;
    o[3] = (d12 + 8.f * d34) + m[5];
  }
  // G = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]
  static __device__ __forceinline__ void g(const float (&w)[3], float (&u)[6]) {
    const float s = w[0] + w[2];
    u[0] = 0.25f * w[0];
    u[1] = (-1.f / 6.f) * (s + w[1]);
    u[2] = (-1.f / 6.f) * (s - w[1]);
    const float q = (1.f / 24.f) * w[0] + (1.f / 6.f) * w[2];
    u[3] = q + (1.f / 12.f) * w[1];
    u[4] = q - (1.f / 12.f) * w[1];
    u[5] = w[2];
  }
  // A (6x4) = [1 0 0 0; 1 1 1 1; 1 -1 1 -1; 1 2 4 8; 1 -2 4 -8; 0 0 0 1]
  template <typename V>
  static __device__ __forceinline__ void a(const V (&y)[4], V (&m)[6]) {
    const V e = y[0] + y[2], o = y[1] + y[3];
    const V e4 = y[0] + 4.f * y[2], o4 = 2.f * y[1] + 8.f * y[3];
    m[0] = y[0]; m[1] = e + o; m[2] = e - o; m[3] = e4 + o4; m[4] = e4 - o4; m[5] = y[3];
  }
  // G^T (3x6)
  static __device__ __forceinline__ void gt(const float (&u)[6], float (&w)[3]) {
    const float s12 = u[1] + u[2], s34 = u[3] + u[4];
    w[0] = 0.25f * u[0] - (1.f / 6.f) * s12 + (1.f / 24.f) * s34;
    w[1] = (1.f / 6.f) * (u[2] - u[1]) + (1.f / 12.f) * (u[3] - u[4]);
    w[2] = (1.f / 6.f) * (s34 - s12) + u[5];
  }
};

// one thread: one tile x W channels
template <int M, int W>
__device__ __forceinline__ void wino_input_body(const WArgs& a, int blk, int nblk) {
  typedef typename VecW<W>::type VT;
  constexpr int P = M + 2;
  const int CQ = a.c / W;
  const int64_t total = (int64_t)a.total_tiles * CQ;
  const size_t plane = (size_t)a.total_tiles * a.c;
  for (int64_t i = (int64_t)blk * NT + threadIdx.x; i < total; i += (int64_t)nblk * NT) {
    const int q4 = (int)(i % CQ);
    const int t = (int)(i / CQ);
    const WSeg& sg = a.seg[seg_of_tile(a, t)];
    int lt = t - sg.tile_start;
    const int tx = lt % sg.tw; lt /= sg.tw;
    const int ty = lt % sg.th;
    const int n_ = lt / sg.th;
    const int y0 = M * ty - 1, x0 = M * tx - 1;
    VT d[P][P];  // d[c][r]: column-major so that the first pass transforms contiguous arrays
#pragma unroll
    for (int r = 0; r < P; ++r) {
      const int yy = y0 + r;
#pragma unroll
      for (int c = 0; c < P; ++c) {
        const int xx = x0 + c;
        // branch-free: always load from a clamped (valid) address, then zero the padding taps -- conditional loads
        // compile to one branch + wait per tap and serialise the 36 round trips
        const bool ok = (unsigned)yy < (unsigned)sg.h && (unsigned)xx < (unsigned)sg.w;
        const int yc = min(max(yy, 0), sg.h - 1), xc = min(max(xx, 0), sg.w - 1);
        const VT v = *reinterpret_cast<const VT*>(sg.x + ((size_t)(n_ * sg.h + yc) * sg.w + xc) * a.c + q4 * W);
        d[c][r] = ok ? v : (VT)(0.f);
      }
    }
    VT tm[P][P];  // tm[r][c] = (B^T d)[r][c]
#pragma unroll
    for (int c = 0; c < P; ++c) {
      VT col[P];
      Wino<M>::bt(d[c], col);
#pragma unroll
      for (int r = 0; r < P; ++r) tm[r][c] = col[r];
    }
    float* out = a.buf + (size_t)t * a.c + q4 * W;
#pragma unroll
    for (int r = 0; r < P; ++r) {
      VT row[P];
      Wino<M>::bt(tm[r], row);  // (B^T d) B
#pragma unroll
      for (int c = 0; c < P; ++c) *reinterpret_cast<VT*>(out + (size_t)(r * P + c) * plane) = row[c];
    }
  }
}

template <int M, int W>
__device__ __forceinline__ void wino_output_body(const WArgs& a, int blk, int nblk) {
  typedef typename VecW<W>::type VT;
  constexpr int P = M + 2;
  const int CQ = a.c / W;
  const int64_t total = (int64_t)a.total_tiles * CQ;
  const size_t plane = (size_t)a.total_tiles * a.c;
  for (int64_t i = (int64_t)blk * NT + threadIdx.x; i < total; i += (int64_t)nblk * NT) {
    const int q4 = (int)(i % CQ);
    const int t = (int)(i / CQ);
    const WSeg& sg = a.seg[seg_of_tile(a, t)];
    int lt = t - sg.tile_start;
    const int tx = lt % sg.tw; lt /= sg.tw;
    const int ty = lt % sg.th;
    const int n_ = lt / sg.th;
    const float* in = a.buf + (size_t)t * a.c + q4 * W;
    VT rr[M][P];  // rr[i][c] = (A^T m)[i][c]
#pragma unroll
    for (int c = 0; c < P; ++c) {
      VT col[P], o[M];
#pragma unroll
      for (int r = 0; r < P; ++r) col[r] = *reinterpret_cast<const VT*>(in + (size_t)(r * P + c) * plane);
      Wino<M>::at(col, o);
#pragma unroll
      for (int i2 = 0; i2 < M; ++i2) rr[i2][c] = o[i2];
    }
    VT b = (VT)(0.f);
    if (a.bias) b = *reinterpret_cast<const VT*>(a.bias + q4 * W);
    const int y0 = M * ty, x0 = M * tx;
#pragma unroll
    for (int i2 = 0; i2 < M; ++i2) {
      VT o[M];
      Wino<M>::at(rr[i2], o);
      if (y0 + i2 < sg.h) {
        float* yb = sg.y + ((size_t)(n_ * sg.h + y0 + i2) * sg.w + x0) * a.c + q4 * W;
#pragma unroll
        for (int j = 0; j < M; ++j)
          if (x0 + j < sg.w) *reinterpret_cast<VT*>(yb + (size_t)j * a.c) = o[j] + b;
      }
    }
  }
}

// weight gradient, stage 2: dM [P][T][C] = A dY A^T per m x m output-gradient tile (zeros outside the map)
template <int M, int W>
__device__ __forceinline__ void wino_dy_body(const WArgs& a, int blk, int nblk) {
  typedef typename VecW<W>::type VT;
  constexpr int P = M + 2;
  const int CQ = a.c / W;
  const int64_t total = (int64_t)a.total_tiles * CQ;
  const size_t plane = (size_t)a.total_tiles * a.c;
  for (int64_t i = (int64_t)blk * NT + threadIdx.x; i < total; i += (int64_t)nblk * NT) {
    const int q4 = (int)(i % CQ);
    const int t = (int)(i / CQ);
    const WSeg& sg = a.seg[seg_of_tile(a, t)];
    int lt = t - sg.tile_start;
    const int tx = lt % sg.tw; lt /= sg.tw;
    const int ty = lt % sg.th;
    const int n_ = lt / sg.th;
    const int y0 = M * ty, x0 = M * tx;
    VT tm[P][M];  // tm[xi][c] = (A dY)[xi][c]
#pragma unroll
    for (int c = 0; c < M; ++c) {
      VT col[M], o[P];
#pragma unroll
      for (int r = 0; r < M; ++r) {
        const bool ok = y0 + r < sg.h && x0 + c < sg.w;
        const int yc = min(y0 + r, sg.h - 1), xc = min(x0 + c, sg.w - 1);
        const VT v = *reinterpret_cast<const VT*>(sg.x + ((size_t)(n_ * sg.h + yc) * sg.w + xc) * a.c + q4 * W);
        col[r] = ok ? v : (VT)(0.f);
      }
      Wino<M>::a(col, o);
#pragma unroll
      for (int r = 0; r < P; ++r) tm[r][c] = o[r];
    }
    float* out = a.buf + (size_t)t * a.c + q4 * W;
#pragma unroll
    for (int r = 0; r < P; ++r) {
      VT row[P];
      Wino<M>::a(tm[r], row);
#pragma unroll
      for (int c = 0; c < P; ++c) *reinterpret_cast<VT*>(out + (size_t)(r * P + c) * plane) = row[c];
    }
  }
}

template <int M, int W>
__global__ __launch_bounds__(NT) void wino_input_kernel(const WArgs a) { wino_input_body<M, W>(a, blockIdx.x, gridDim.x); }
template <int M, int W>
__global__ __launch_bounds__(NT) void wino_output_kernel(const WArgs a) { wino_output_body<M, W>(a, blockIdx.x, gridDim.x); }
template <int M, int W>
__global__ __launch_bounds__(NT) void wino_dy_kernel(const WArgs a) { wino_dy_body<M, W>(a, blockIdx.x, gridDim.x); }

// weight gradient, stage 4: dw[3][3][k][n] (+)= G^T dU G for dU [P][k][n]
// du holds `nsplit` partial sums of dU (the split reduction of the batched GEMM), added here in split order.
template <int M>
__device__ __forceinline__ void wino_dw_body(const float* __restrict__ du, float* __restrict__ dw, int64_t kn, int nsplit,
                                             int accumulate, int blk) {
  constexpr int P = M + 2;
  const int64_t i = (int64_t)blk * NT + threadIdx.x;
  if (i >= kn) return;
  const size_t split_stride = (size_t)P * P * kn;
  float t[P][3];  // t[b][.] = (G^T dU)[., b]
#pragma unroll
  for (int b = 0; b < P; ++b) {
    float col[P];
#pragma unroll
    for (int a = 0; a < P; ++a) col[a] = du[(size_t)(a * P + b) * kn + i];
    for (int sp = 1; sp < nsplit; ++sp)
#pragma unroll
      for (int a = 0; a < P; ++a) col[a] += du[sp * split_stride + (size_t)(a * P + b) * kn + i];
    Wino<M>::gt(col, t[b]);
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float row[P], o[3];
#pragma unroll
    for (int b = 0; b < P; ++b) row[b] = t[b][a];
    Wino<M>::gt(row, o);
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      float* q = dw + (size_t)(a * 3 + b) * kn + i;
      *q = accumulate ? *q + o[b] : o[b];
    }
  }
}

// U[xi][k][n] = (G g G^T)[xi] for g = w[:, :, k, n] (rot == 0) or its 180-degree rotation (rot != 0, the
// data-gradient kernel); [k][n] = [cin][cout] either way.
// u2 != nullptr: also the transform of the rotated kernel (what the data gradient will need) from the same 9 loads.
template <int M>
__device__ __forceinline__ void wino_weight_body(const float* __restrict__ w, float* __restrict__ u, float* __restrict__ u2,
                                                 int64_t kn, int rot, int blk) {
  constexpr int P = M + 2;
  const int64_t i = (int64_t)blk * NT + threadIdx.x;
  if (i >= kn) return;
  float g9[3][3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) g9[a][b] = w[(size_t)(a * 3 + b) * kn + i];
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const bool r = pass ? true : (rot != 0);
    float* out = pass ? u2 : u;
    if (pass && !u2) break;
    float t[3][P];  // t[b][a] = (G g)[a][b]
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      float col[3];
#pragma unroll
      for (int a = 0; a < 3; ++a) col[a] = r ? g9[2 - a][2 - b] : g9[a][b];
      Wino<M>::g(col, t[b]);
    }
#pragma unroll
    for (int a = 0; a < P; ++a) {
      const float row[3] = {t[0][a], t[1][a], t[2][a]};
      float o[P];
      Wino<M>::g(row, o);
#pragma unroll
      for (int b = 0; b < P; ++b) out[(size_t)(a * P + b) * kn + i] = o[b];
    }
  }
}

template <int M>
__global__ __launch_bounds__(NT) void wino_dw_kernel(const float* __restrict__ du, float* __restrict__ dw, int64_t kn, int nsplit,
                                                    int accumulate) {
  wino_dw_body<M>(du, dw, kn, nsplit, accumulate, blockIdx.x);
}
template <int M>
__global__ __launch_bounds__(NT) void wino_weight_kernel(const float* __restrict__ w, float* __restrict__ u, float* __restrict__ u2,
                                                        int64_t kn, int rot) {
  wino_weight_body<M>(w, u, u2, kn, rot, blockIdx.x);
}

// Independent stages of one layer share a launch (blocks of two kinds) -- every launch saved is ~5 us of a step that
// is a chain of ~500 short kernels:
//   forward  : input transform of x            | kernel transform (+ the rotated one kept for backward)
//   backward : B^T dy B (for the data gradient) | A dy A^T (for the weight gradient)      -- both read dy
//   backward : output transform -> dx           | G^T dU G -> dw
struct WeightArgs { const float* w; float* u; float* u2; int64_t kn; int rot; };
struct DwArgs2 { const float* du; float* dw; int64_t kn; int nsplit, accumulate; };

template <int M, int W>
__global__ __launch_bounds__(NT) void wino_fwd_pre_kernel(const WArgs in, const WeightArgs wa, int nb_in) {
  if ((int)blockIdx.x < nb_in) wino_input_body<M, W>(in, blockIdx.x, nb_in);
  else wino_weight_body<M>(wa.w, wa.u, wa.u2, wa.kn, wa.rot, (int)blockIdx.x - nb_in);
}
template <int M, int W>
__global__ __launch_bounds__(NT) void wino_bwd_pre_kernel(const WArgs dgrad_in, const WArgs wgrad_dy, int nb_first) {
  if ((int)blockIdx.x < nb_first) wino_input_body<M, W>(dgrad_in, blockIdx.x, nb_first);
  else wino_dy_body<M, W>(wgrad_dy, (int)blockIdx.x - nb_first, (int)gridDim.x - nb_first);
}
template <int M, int W>
__global__ __launch_bounds__(NT) void wino_bwd_post_kernel(const WArgs out, const DwArgs2 d, int nb_out) {
  if ((int)blockIdx.x < nb_out) wino_output_body<M, W>(out, blockIdx.x, nb_out);
  else wino_dw_body<M>(d.du, d.dw, d.kn, d.nsplit, d.accumulate, (int)blockIdx.x - nb_out);
}

int fill(const rn_conv_seg* segs, int nseg, int cin, int cout, int m, bool dgrad, WArgs* in, WArgs* out,
         bool input_only = false) {
  RN_CHECK_ARG(segs && nseg >= 1 && nseg <= RN_MAX_SEG, "winograd: bad segments");
  RN_CHECK_ARG(m == 2 || m == 4, "winograd: tile %d (2 or 4)", m);
  RN_UNSUPPORTED(cin % 4 != 0 || cout % 4 != 0, "winograd: cin %d / cout %d must be multiples of 4", cin, cout);
  int64_t tiles = 0;
  for (int s = 0; s < nseg; ++s) {
    RN_CHECK_ARG(segs[s].n >= 1 && segs[s].h >= 1 && segs[s].w >= 1, "winograd: bad segment %d", s);
    WSeg& a = in->seg[s];
    WSeg& b = out->seg[s];
    a.n = b.n = segs[s].n; a.h = b.h = segs[s].h; a.w = b.w = segs[s].w;
    a.th = b.th = (segs[s].h + m - 1) / m; a.tw = b.tw = (segs[s].w + m - 1) / m;
    a.tile_start = b.tile_start = (int)tiles;
    a.x = dgrad ? segs[s].dy : segs[s].x;
    b.y = dgrad ? segs[s].dx : segs[s].y;
    RN_CHECK_ARG(a.x && (b.y || input_only), "winograd: null tensor in segment %d", s);
    RN_UNSUPPORTED((double)a.n * a.h * a.w * (cin > cout ? cin : cout) * 4.0 >= 2147483648.0, "winograd: segment %d >= 2 GiB", s);
    tiles += (int64_t)a.n * a.th * a.tw;
  }
  const int cmax = cin > cout ? cin : cout;
  RN_UNSUPPORTED((double)tiles * cmax * 4.0 >= 2147483648.0, "winograd: %lld tiles x %d channels: a transform plane is >= 2 GiB",
                 (long long)tiles, cmax);
  in->nseg = out->nseg = nseg;
  in->total_tiles = out->total_tiles = (int)tiles;
  in->c = dgrad ? cout : cin;    // channels of the transformed input
  out->c = dgrad ? cin : cout;   // channels of the result
  return RN_OK;
}

unsigned grid_for(int64_t total) {
  int64_t b = (total + NT - 1) / NT;
  if (b > 32768) b = 32768;
  if (b < 1) b = 1;
  return (unsigned)b;
}
// widest access that still leaves ~4 waves per SIMD of work on 256 CUs
int width_for(int64_t tiles, int c) {
  if (const char* force = getenv("RN_WINO_W")) {  // tuning aid
    const int w = atoi(force);
    if (w == 1 || w == 2 || w == 4) return w;
  }
  const int64_t want = 256 * 1024;
  if (tiles * (c / 4) >= want) return 4;
  return tiles * (c / 2) >= want ? 2 : 1;
}
#define RN_WINO_LAUNCH(KERNEL, WIDTH, TOTAL_C, ARGS)                                                                \
  do {                                                                                                             \
    if (WIDTH == 4) hipLaunchKernelGGL((KERNEL<M, 4>), dim3(grid_for((TOTAL_C) / 4)), dim3(NT), 0, st, ARGS);      \
    else if (WIDTH == 2) hipLaunchKernelGGL((KERNEL<M, 2>), dim3(grid_for((TOTAL_C) / 2)), dim3(NT), 0, st, ARGS); \
    else hipLaunchKernelGGL((KERNEL<M, 1>), dim3(grid_for(TOTAL_C)), dim3(NT), 0, st, ARGS);                       \
  } while (0)

size_t tiles_of(const rn_conv_seg* segs, int nseg, int m) {
  size_t tiles = 0;
  for (int s = 0; s < nseg; ++s) tiles += (size_t)segs[s].n * ((segs[s].h + m - 1) / m) * ((segs[s].w + m - 1) / m);
  return tiles;
}

template <int M>
int run(const WArgs& ia_, const WArgs& oa_, int cin, int cout, const float* w, const float* bias, bool dgrad, float* U, float* V,
        float* Mb, float* v_buf, float* urot_buf, hipStream_t st) {
  constexpr int P2 = (M + 2) * (M + 2);
  WArgs ia = ia_, oa = oa_;
  const int T_ = ia.total_tiles, kc = ia.c, nc = oa.c;
  const int64_t kn = (int64_t)cin * cout;
  if (!dgrad && v_buf) V = v_buf;  // keep the transformed input for the weight gradient
  ia.buf = V;
  const int wi = width_for(T_, kc), wo = width_for(T_, nc);
  if (dgrad && urot_buf) {
    U = urot_buf;  // transformed (rotated) kernel kept by the forward call
    RN_WINO_LAUNCH(wino_input_kernel, wi, (int64_t)T_ * kc, ia);
  } else {  // kernel transform and input transform: independent, one launch
    const WeightArgs wa = {w, U, dgrad ? nullptr : urot_buf, kn, dgrad ? 1 : 0};
    const int nb_in = (int)grid_for((int64_t)T_ * kc / wi), nb_w = (int)rn::ceil_div64(kn, NT);
    if (wi == 4) hipLaunchKernelGGL((wino_fwd_pre_kernel<M, 4>), dim3(nb_in + nb_w), dim3(NT), 0, st, ia, wa, nb_in);
    else if (wi == 2) hipLaunchKernelGGL((wino_fwd_pre_kernel<M, 2>), dim3(nb_in + nb_w), dim3(NT), 0, st, ia, wa, nb_in);
    else hipLaunchKernelGGL((wino_fwd_pre_kernel<M, 1>), dim3(nb_in + nb_w), dim3(NT), 0, st, ia, wa, nb_in);
  }
  RN_LAUNCH_CHECK();
  // forward: M_xi [T x cout] = V_xi [T x cin] * U_xi [cin x cout];  dgrad: [T x cin] = V_xi [T x cout] * U_xi^T
  if (int e = rn::launch_batched_gemm(V, U, Mb, T_, kc, nc, P2, dgrad ? 1 : 0, st)) return e;
  oa.buf = Mb;
  oa.bias = dgrad ? nullptr : bias;
  RN_WINO_LAUNCH(wino_output_kernel, wo, (int64_t)T_ * nc, oa);
  RN_LAUNCH_CHECK();
  return RN_OK;
}
template <int M>
int run_wgrad(const WArgs& xa_, const WArgs& ya_, int cin, int cout, float* dw, int accumulate, float* dU, float* V, float* dM,
              void* gemm_ws, size_t gemm_ws_bytes, const float* v_buf, hipStream_t st) {
  constexpr int P2 = (M + 2) * (M + 2);
  WArgs xa = xa_, ya = ya_;
  const int T_ = xa.total_tiles;
  xa.buf = V;
  ya.buf = dM;
  const int wi = width_for(T_, cin), wo = width_for(T_, cout);
  if (v_buf) V = const_cast<float*>(v_buf);  // the forward call's transformed input
  else RN_WINO_LAUNCH(wino_input_kernel, wi, (int64_t)T_ * cin, xa);
  RN_WINO_LAUNCH(wino_dy_kernel, wo, (int64_t)T_ * cout, ya);
  RN_LAUNCH_CHECK();
  // dU_xi [cin x cout] = V_xi^T [cin x T] * dM_xi [T x cout]
  int nsplit = 1;
  if (int e = rn::launch_batched_gemm_tn(V, dM, nullptr, T_, cin, cout, P2, gemm_ws, gemm_ws_bytes, st, &nsplit)) return e;
  const int64_t kn = (int64_t)cin * cout;
  hipLaunchKernelGGL(wino_dw_kernel<M>, dim3((unsigned)rn::ceil_div64(kn, 256)), dim3(256), 0, st, (const float*)gemm_ws, dw, kn,
                     nsplit, accumulate);
  RN_LAUNCH_CHECK();
  return RN_OK;
}
}  // namespace

// bytes: dU (P*cin*cout) + V (P*T*cin) + dM (P*T*cout) + the split-reduction slab of the batched GEMM
extern "C" size_t rn_conv3x3_winograd_wgrad_workspace(const rn_conv_seg* segs, int nseg, int cin, int cout, int tile) {
  if (!segs || nseg < 1 || nseg > RN_MAX_SEG || (tile != 2 && tile != 4)) return 0;
  const size_t tiles = tiles_of(segs, nseg, tile), p2 = (size_t)(tile + 2) * (tile + 2);
  return rn::align_up(p2 * cin * cout * 4, 256) + rn::align_up(p2 * tiles * cin * 4, 256) + rn::align_up(p2 * tiles * cout * 4, 256) +
         rn::batched_gemm_tn_workspace((int)tiles, cin, cout, (int)p2);
}

// dw[3,3,cin,cout] (+)= sum over segments of the weight gradient of y = conv3x3_same(x, w), from the segments'
// x and dy:  dU_xi = sum_tiles (B^T d B)_xi^T (A dY A^T)_xi,  dw = G^T dU G.
extern "C" int rn_conv3x3_winograd_wgrad(const rn_conv_seg* segs, int nseg, int cin, int cout, float* dw, int accumulate, int tile,
                                         void* workspace, size_t workspace_bytes, const float* v_buf, rn_stream_t stream) {
  WArgs xa = {}, ya = {}, unused_in = {}, unused_out = {};
  // x through the input transform (channels cin); dy through the A-transform (channels cout): reuse fill() twice
  if (int e = fill(segs, nseg, cin, cout, tile, false, &xa, &unused_out, true)) return e;
  if (int e = fill(segs, nseg, cin, cout, tile, true, &ya, &unused_in, true)) return e;
  RN_CHECK_ARG(dw && workspace, "winograd wgrad: null dw / workspace");
  const size_t need = rn_conv3x3_winograd_wgrad_workspace(segs, nseg, cin, cout, tile);
  if (workspace_bytes < need) {
    rn::set_error("winograd wgrad: workspace %zu < %zu bytes", workspace_bytes, need);
    return RN_EWORKSPACE;
  }
  const size_t p2 = (size_t)(tile + 2) * (tile + 2);
  const size_t tiles = xa.total_tiles;
  char* base = (char*)workspace;
  float* dU = (float*)base;                 base += rn::align_up(p2 * cin * cout * 4, 256);
  float* V = (float*)base;                  base += rn::align_up(p2 * tiles * cin * 4, 256);
  float* dM = (float*)base;                 base += rn::align_up(p2 * tiles * cout * 4, 256);
  const size_t gemm_ws = workspace_bytes - (size_t)(base - (char*)workspace);
  hipStream_t st = (hipStream_t)stream;
  return tile == 2 ? run_wgrad<2>(xa, ya, cin, cout, dw, accumulate, dU, V, dM, base, gemm_ws, v_buf, st)
                   : run_wgrad<4>(xa, ya, cin, cout, dw, accumulate, dU, V, dM, base, gemm_ws, v_buf, st);
}

// bytes: U (P*cin*cout) + V (P*T*cin) + M (P*T*cout)
extern "C" size_t rn_conv3x3_winograd_workspace(const rn_conv_seg* segs, int nseg, int cin, int cout, int tile) {
  if (!segs || nseg < 1 || nseg > RN_MAX_SEG || (tile != 2 && tile != 4)) return 0;
  const size_t tiles = tiles_of(segs, nseg, tile), p2 = (size_t)(tile + 2) * (tile + 2);
  return rn::align_up(p2 * cin * cout * 4, 256) + rn::align_up(p2 * tiles * cin * 4, 256) + rn::align_up(p2 * tiles * cout * 4, 256);
}

// dgrad == 0: y = conv3x3_same(x, w) + bias.   dgrad != 0: dx = conv3x3_same(dy, rot180(w)^T)  (w is always
// the forward kernel [3,3,cin,cout]).
extern "C" int rn_conv3x3_winograd(const rn_conv_seg* segs, int nseg, int cin, int cout, const float* w, const float* bias,
                                   int dgrad, int tile, void* workspace, size_t workspace_bytes, float* v_buf, float* urot_buf,
                                   rn_stream_t stream) {
  WArgs ia = {}, oa = {};
  if (int e = fill(segs, nseg, cin, cout, tile, dgrad != 0, &ia, &oa)) return e;
  RN_CHECK_ARG(w && workspace, "winograd: null weights / workspace");
  const size_t need = rn_conv3x3_winograd_workspace(segs, nseg, cin, cout, tile);
  if (workspace_bytes < need) {
    rn::set_error("winograd: workspace %zu < %zu bytes", workspace_bytes, need);
    return RN_EWORKSPACE;
  }
  const size_t p2 = (size_t)(tile + 2) * (tile + 2);
  float* U = (float*)workspace;
  float* V = (float*)((char*)workspace + rn::align_up(p2 * cin * cout * 4, 256));
  float* Mb = (float*)((char*)V + rn::align_up(p2 * ia.total_tiles * ia.c * 4, 256));
  hipStream_t st = (hipStream_t)stream;
  return tile == 2 ? run<2>(ia, oa, cin, cout, w, bias, dgrad != 0, U, V, Mb, v_buf, urot_buf, st)
                   : run<4>(ia, oa, cin, cout, w, bias, dgrad != 0, U, V, Mb, v_buf, urot_buf, st);
}

extern "C" int rn_gemm_batched(const float* A, const float* B, float* C, int M, int K, int N, int nbatch, int b_nk,
                               rn_stream_t stream) {
  RN_CHECK_ARG(A && B && C && M >= 1 && K >= 1 && N >= 1 && nbatch >= 1, "gemm_batched: bad argument");
  RN_UNSUPPORTED(K % 4 != 0 || N % 4 != 0, "gemm_batched: K %d / N %d must be multiples of 4", K, N);
  return rn::launch_batched_gemm(A, B, C, M, K, N, nbatch, b_nk, (hipStream_t)stream);
}

// bytes of the two optional buffers a forward call can fill for its backward pass: the transformed input V
// ([P][tiles][cin], reused by the weight gradient) and the transformed rotated kernel ([P][cin][cout], reused by the
// data gradient)
extern "C" int rn_conv3x3_winograd_keep_bytes(const rn_conv_seg* segs, int nseg, int cin, int cout, int tile, size_t* v_bytes,
                                              size_t* urot_bytes) {
  RN_CHECK_ARG(segs && nseg >= 1 && nseg <= RN_MAX_SEG && (tile == 2 || tile == 4) && v_bytes && urot_bytes, "winograd: bad argument");
  const size_t p2 = (size_t)(tile + 2) * (tile + 2);
  *v_bytes = p2 * tiles_of(segs, nseg, tile) * cin * 4;
  *urot_bytes = p2 * cin * cout * 4;
  return RN_OK;
}

// ---- the whole backward pass of a Winograd layer in three launches (+ the rare two that rebuild what forward did
// not keep): [B^T dy B | A dy A^T]  ->  [data-gradient products | weight-gradient partial products]  ->
// [output transform -> dx | G^T dU G -> dw].  Segments: x, dy, dx.  v_buf / urot_buf as for rn_conv3x3_winograd.
namespace {
struct BwdLayout { size_t vdy, mdx, dm, v, urot, slab, total; };
BwdLayout bwd_layout(const rn_conv_seg* segs, int nseg, int cin, int cout, int tile, bool need_v, bool need_urot) {
  const size_t tiles = tiles_of(segs, nseg, tile), p2 = (size_t)(tile + 2) * (tile + 2);
  BwdLayout L;
  size_t off = 0;
  L.vdy = off; off += rn::align_up(p2 * tiles * cout * 4, 256);
  L.mdx = off; off += rn::align_up(p2 * tiles * cin * 4, 256);
  L.dm = off; off += rn::align_up(p2 * tiles * cout * 4, 256);
  L.v = off; off += need_v ? rn::align_up(p2 * tiles * cin * 4, 256) : 0;
  L.urot = off; off += need_urot ? rn::align_up(p2 * cin * cout * 4, 256) : 0;
  L.slab = off; off += rn::batched_gemm_tn_workspace((int)tiles, cin, cout, (int)p2);
  L.total = off;
  return L;
}

template <int M>
int run_bwd(const rn_conv_seg* segs, int nseg, int cin, int cout, const float* w, float* dw, int accumulate, char* ws,
            const BwdLayout& L, size_t ws_bytes, const float* v_buf, const float* urot_buf, hipStream_t st) {
  constexpr int P2 = (M + 2) * (M + 2);
  WArgs ia = {}, oa = {}, ya = {}, xa = {}, unused = {};
  if (int e = fill(segs, nseg, cin, cout, M, true, &ia, &oa)) return e;            // dy -> (products) -> dx
  if (int e = fill(segs, nseg, cin, cout, M, true, &ya, &unused, true)) return e;  // dy, for the weight gradient
  const int T_ = ia.total_tiles;
  const int64_t kn = (int64_t)cin * cout;
  float* Vdy = (float*)(ws + L.vdy);
  float* Mdx = (float*)(ws + L.mdx);
  float* dM = (float*)(ws + L.dm);
  const float* V = v_buf;
  const float* Urot = urot_buf;
  if (!Urot) {
    float* u = (float*)(ws + L.urot);
    hipLaunchKernelGGL(wino_weight_kernel<M>, dim3((unsigned)rn::ceil_div64(kn, NT)), dim3(NT), 0, st, w, u, (float*)nullptr, kn, 1);
    Urot = u;
  }
  if (!V) {
    if (int e = fill(segs, nseg, cin, cout, M, false, &xa, &unused, true)) return e;
    xa.buf = (float*)(ws + L.v);
    const int wx = width_for(T_, cin);
    RN_WINO_LAUNCH(wino_input_kernel, wx, (int64_t)T_ * cin, xa);
    V = xa.buf;
  }
  ia.buf = Vdy;
  ya.buf = dM;
  const int wy = width_for(T_, cout), wo = width_for(T_, cin);
  {
    const int nb = (int)grid_for((int64_t)T_ * cout / wy);
    if (wy == 4) hipLaunchKernelGGL((wino_bwd_pre_kernel<M, 4>), dim3(2 * nb), dim3(NT), 0, st, ia, ya, nb);
    else if (wy == 2) hipLaunchKernelGGL((wino_bwd_pre_kernel<M, 2>), dim3(2 * nb), dim3(NT), 0, st, ia, ya, nb);
    else hipLaunchKernelGGL((wino_bwd_pre_kernel<M, 1>), dim3(2 * nb), dim3(NT), 0, st, ia, ya, nb);
  }
  RN_LAUNCH_CHECK();
  int nsplit = 1;
  if (int e = rn::launch_winograd_bwd_products(Vdy, Urot, Mdx, T_, cout, cin, V, dM, cin, cout, P2, ws + L.slab, ws_bytes - L.slab, st,
                                               &nsplit))
    return e;
  oa.buf = Mdx;
  oa.bias = nullptr;
  {
    const DwArgs2 d = {(const float*)(ws + L.slab), dw, kn, nsplit, accumulate};
    const int nb_out = (int)grid_for((int64_t)T_ * cin / wo), nb_dw = (int)rn::ceil_div64(kn, NT);
    if (wo == 4) hipLaunchKernelGGL((wino_bwd_post_kernel<M, 4>), dim3(nb_out + nb_dw), dim3(NT), 0, st, oa, d, nb_out);
    else if (wo == 2) hipLaunchKernelGGL((wino_bwd_post_kernel<M, 2>), dim3(nb_out + nb_dw), dim3(NT), 0, st, oa, d, nb_out);
    else hipLaunchKernelGGL((wino_bwd_post_kernel<M, 1>), dim3(nb_out + nb_dw), dim3(NT), 0, st, oa, d, nb_out);
  }
  RN_LAUNCH_CHECK();
  return RN_OK;
}
}  // namespace

extern "C" size_t rn_conv3x3_winograd_bwd_workspace(const rn_conv_seg* segs, int nseg, int cin, int cout, int tile, int have_v,
                                                    int have_urot) {
  if (!segs || nseg < 1 || nseg > RN_MAX_SEG || (tile != 2 && tile != 4)) return 0;
  return bwd_layout(segs, nseg, cin, cout, tile, !have_v, !have_urot).total;
}

extern "C" int rn_conv3x3_winograd_bwd(const rn_conv_seg* segs, int nseg, int cin, int cout, const float* w, float* dw, int accumulate,
                                       int tile, void* workspace, size_t workspace_bytes, const float* v_buf, const float* urot_buf,
                                       rn_stream_t stream) {
  RN_CHECK_ARG(segs && nseg >= 1 && nseg <= RN_MAX_SEG && (tile == 2 || tile == 4), "winograd bwd: bad segments / tile");
  RN_CHECK_ARG(w && dw && workspace, "winograd bwd: null pointer");
  for (int s = 0; s < nseg; ++s) RN_CHECK_ARG(segs[s].x && segs[s].dy && segs[s].dx, "winograd bwd: null tensor in segment %d", s);
  const BwdLayout L = bwd_layout(segs, nseg, cin, cout, tile, v_buf == nullptr, urot_buf == nullptr);
  if (workspace_bytes < L.total) {
    rn::set_error("winograd bwd: workspace %zu < %zu bytes", workspace_bytes, L.total);
    return RN_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  return tile == 2 ? run_bwd<2>(segs, nseg, cin, cout, w, dw, accumulate, (char*)workspace, L, workspace_bytes, v_buf, urot_buf, st)
                   : run_bwd<4>(segs, nseg, cin, cout, w, dw, accumulate, (char*)workspace, L, workspace_bytes, v_buf, urot_buf, st);
}
