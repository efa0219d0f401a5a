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
  // (contraction off: every rounding of the kernel transform is pinned, so that U has the same bits whichever kernel computes it --
  // wino_weight_body and wino_weight_frag_body; left to itself the compiler fuses these products into adds of the NEXT call, and
  // differently in the two: 30 % of the elements of some points came out one ulp apart, tools/dbg_u.py)
  static __device__ __forceinline__ void g(const float (&w)[3], float (&u)[4]) {
#pragma clang fp contract(off)
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
#pragma clang fp contract(off)
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
    if (!out) continue;
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

// The same transform written as the forward product kernel's pre-split B operand (gemm_x3_bfrag.hip FragB: three bf16 planes in
// MFMA-fragment order, [n block][16-k step][plane][lane][8 bf16]; B(k = ci, n = co)) instead of fp32 [xi][cin][cout].
// One thread per dword = two consecutive k of one column n, for all (M+2)^2 points and the three planes; a wave covers 16 columns x
// 4 k-pairs, so each of its stores fills sixteen whole 16-byte lane slots (256 contiguous bytes).  Threads of padding columns
// (n >= cout, inside the last 32-block) write zeros.  The arithmetic is Wino<M>::g twice, as in wino_weight_body: the same bits
// (g compiles without contraction).
__device__ __forceinline__ void split3w(float x, unsigned& h1, unsigned& h2, unsigned& h3) {
  h1 = __float_as_uint(x) & 0xffff0000u;
  const float r1 = x - __uint_as_float(h1);
  h2 = __float_as_uint(r1) & 0xffff0000u;
  h3 = __float_as_uint(r1 - __uint_as_float(h2));
}
template <int M>
__device__ __forceinline__ void wino_weight_frag_body(const float* __restrict__ w, unsigned* __restrict__ out, int cin, int cout, int blk) {
  constexpr int P = M + 2;
  const int nb16 = ((cout + 31) / 32) * 2, ks16 = cin / 16;
  const int64_t i = (int64_t)blk * NT + threadIdx.x;
  if (i >= (int64_t)(cin / 2) * nb16 * 16) return;
  const int nl = (int)(i & 15), q = (int)((i >> 4) & 3);
  const int64_t rest = i >> 6;
  const int n = (int)(rest % nb16) * 16 + nl, k = (int)(rest / nb16) * 8 + 2 * q;
  const int64_t kn = (int64_t)cin * cout;
  float g9[2][3][3];
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) g9[e][a][b] = n < cout ? w[(size_t)(a * 3 + b) * kn + (int64_t)(k + e) * cout + n] : 0.f;      // w[tap][ci][co]
  float t[2][3][P];
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const float col[3] = {g9[e][0][b], g9[e][1][b], g9[e][2][b]};
      Wino<M>::g(col, t[e][b]);
    }
  // dword (n, k pair) of batch xi, plane p: ((((n / 32) ks16 + k / 16) 3 + p) 64 + (n % 32) + 32 ((k / 8) % 2)) 4 + (k % 8) / 2
  const size_t bstride = (size_t)(nb16 / 2) * ks16 * 768;      // dwords per batch
  unsigned* o = out + ((size_t)(n >> 5) * ks16 + (k >> 4)) * 768 + ((n & 31) + 32 * ((k >> 3) & 1)) * 4 + ((k & 7) >> 1);
#pragma unroll
  for (int a = 0; a < P; ++a) {
    float oe[2][P];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const float row[3] = {t[e][0][a], t[e][1][a], t[e][2][a]};
      Wino<M>::g(row, oe[e]);
    }
#pragma unroll
    for (int b = 0; b < P; ++b) {
      unsigned h0[3], h1[3];
      split3w(oe[0][b], h0[0], h0[1], h0[2]);
      split3w(oe[1][b], h1[0], h1[1], h1[2]);
      unsigned* ob = o + (size_t)(a * P + b) * bstride;
#pragma unroll
      for (int p = 0; p < 3; ++p) ob[p * 256] = __builtin_amdgcn_perm(h1[p], h0[p], 0x07060302u);      // (the lower k in the low half)
    }
  }
}
// blocks of a kernel-transform launch: [fp32 outputs (u and / or u2) | U as a fragment image], each part optional
struct WeightArgs {
  const float* w; float* u; float* u2; int64_t kn; int rot;
  unsigned* uf; int cin, cout;      // fragment-ordered U (null: not wanted)
};
__host__ __device__ inline int weight_blocks_f32(const WeightArgs& wa) { return (wa.u || wa.u2) ? (int)((wa.kn + NT - 1) / NT) : 0; }
__host__ __device__ inline int weight_blocks_frag(const WeightArgs& wa) {
  return wa.uf ? (int)(((int64_t)(wa.cin / 2) * ((wa.cout + 31) / 32) * 32 + NT - 1) / NT) : 0;
}
__host__ __device__ inline int weight_blocks(const WeightArgs& wa) { return weight_blocks_f32(wa) + weight_blocks_frag(wa); }
template <int M>
__device__ __forceinline__ void wino_weight_any_body(const WeightArgs& wa, int blk) {
  const int n0 = weight_blocks_f32(wa);
  if (blk < n0) wino_weight_body<M>(wa.w, wa.u, wa.u2, wa.kn, wa.rot, blk);
  else wino_weight_frag_body<M>(wa.w, wa.uf, wa.cin, wa.cout, blk - n0);
}

template <int M>
__global__ __launch_bounds__(NT) void wino_weight_any_kernel(const WeightArgs wa) {
  wino_weight_any_body<M>(wa, blockIdx.x);
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

// bytes of the forward pass's transformed kernel U (p2 points) in the workspace: fp32 [p2][cin][cout], or the forward product
// kernel's pre-split fragment image (6 bytes per element, the columns padded to 32: gemm_x3_bfrag.hip FragB) -- whichever the run-time
// switches pick, the region holds either
size_t u_bytes(size_t p2, int cin, int cout) {
  size_t b = (size_t)cin * cout * 4;
  if (cin % 16 == 0 && rn::x3_bfrag_bytes(cin, cout) > b) b = rn::x3_bfrag_bytes(cin, cout);
  return p2 * b;
}

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
  return rn::align_up(u_bytes(p2, cin, cout), 256) + rn::align_up(p2 * tiles * cin * 4, 256) + rn::align_up(p2 * tiles * cout * 4, 256);
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
  float* V = (float*)((char*)workspace + rn::align_up(u_bytes(p2, cin, cout), 256));
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

// The merged backward products of a Winograd layer, exposed for measurement like rn_gemm_batched: ONE launch whose blocks
// are of two kinds -- dgrad products Cd_b [M x Nd] = Ad_b [M x Kd] * Bd_b^T (Bd_b is [Nd x Kd]) and the split partial
// products of the weight gradient dU_b = Aw_b^T [Kw x M] * Bw_b [M x Nw] (left in `workspace`, *nsplit partial slabs).
extern "C" size_t rn_winograd_bwd_products_workspace(int M, int Kw, int Nw, int nbatch) {
  return rn::batched_gemm_tn_workspace(M, Kw, Nw, nbatch);
}
extern "C" int rn_winograd_bwd_products(const float* Ad, const float* Bd, float* Cd, int M, int Kd, int Nd, const float* Aw,
                                        const float* Bw, int Kw, int Nw, int nbatch, void* workspace, size_t workspace_bytes,
                                        int* nsplit, rn_stream_t stream) {
  RN_CHECK_ARG(Ad && Bd && Cd && Aw && Bw && workspace && nsplit && M >= 1 && nbatch >= 1, "winograd_bwd_products: bad argument");
  RN_UNSUPPORTED(Kd % 4 || Nd % 4 || Kw % 4 || Nw % 4, "winograd_bwd_products: K / N must be multiples of 4");
  return rn::launch_winograd_bwd_products(Ad, Bd, Cd, M, Kd, Nd, Aw, Bw, Kw, Nw, nbatch, workspace, workspace_bytes, (hipStream_t)stream,
                                          nsplit);
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

// =============================================================================================================
// GroupNorm folded into a chain of Winograd layers (the class / box towers: 4 x [conv3x3 256->256, GroupNorm,
// activation] + an output conv, retinanet.py:37-71).  Between two such convs the normalised, activated tensor is never
// written and never read:
//   forward   output transform of layer i : writes the RAW conv output y_i and, per chunk of 16 tiles, the per-group
//                                           partial statistics (count, mean, M2)                ("stat rows")
//             input transform of layer i+1: finalises mean / rstd of its sample's groups from <= a few rows (fp64,
//                                           fixed order, a few hundred bytes per block) and loads act(GN(y_i)) on the fly
//   backward  output transform of layer i+1's data gradient: from dA (in registers) and y_i forms
//                                           g = dA * act'(z), writes g, and per chunk the sums (sum g, sum g xhat)
//                                           per channel (-> dbeta / dgamma) and gamma-weighted per group
//             dy transforms of layer i    : load dy = rstd (gamma g - c1 - xhat c2) on the fly from g and y_i
// so a tower layer is three launches forward and three backward with NO GroupNorm kernel, no exchange between
// blocks and no spinning: every reduction is a fixed-order sum over rows written by an earlier launch.
// Organisation: a block (256 threads) = one chunk (16 consecutive tiles of ONE sample) x 16 channel lanes of W channels
// each (whole groups): W = 4 when the launch fills the chip with that, else 2 or 1 -- the pyramids of a 512^2 batch
// are latency-, not bandwidth-bound, and narrow lanes put 4x the blocks in flight.  Needs C % 64 == 0 and
// 64 % (C / groups) == 0 on the folded side.
// =============================================================================================================
namespace {
constexpr int CK_TILES = 16, CK_LANES = 16, CK_T = CK_TILES * CK_LANES;  // a block: 16 tiles x 16 channel lanes of W channels
constexpr int CK_MAXCH = 64;  // channels per block with W = 4
constexpr int ROW_F = 4;  // floats per (chunk, group) of a statistics row: count, mean, M2, unused

struct CSeg { const float* x; const float* aux; float* y; int n, h, w, th, tw, tile_start, chunk_start, cps; };
struct CArgs {
  CSeg seg[RN_MAX_SEG];
  int nseg, c, total_tiles, total_chunks;
  float* buf;          // V or M planes [P][total_tiles][c]
  const float* bias;
};
// GroupNorm parameters of the folded side
struct Fold {
  const float* rows;        // [chunks][groups][4] (count, mean, M2, -) of the normalised tensor
  const float* grows;       // backward, output side: [chunks][groups][2] (sum gamma g, sum gamma g xhat)
  const float* gamma; const float* beta;
  float* out_rows;          // forward, output side: rows to write
  float* g_rows_group;      // backward, input side: rows to write [chunks][groups][2]
  float* g_rows_chan;       // backward, input side: [2][chunks][c]
  int groups, cpg, act;
  float eps;
};

__device__ __forceinline__ int cseg_of_chunk(const CArgs& a, int ch) {
  int s = 0;
  while (s + 1 < a.nseg && ch >= a.seg[s + 1].chunk_start) ++s;
  return s;
}

struct BlockPos { int s, sample, k, tl, q, lt, t_global, c0, cl; bool tile_ok; };
template <int W>
__device__ __forceinline__ BlockPos block_pos(const CArgs& a, int chunk, int slab) {
  constexpr int LANES = CK_LANES;
  BlockPos p;
  p.s = cseg_of_chunk(a, chunk);
  const CSeg& sg = a.seg[p.s];
  const int local = chunk - sg.chunk_start;
  p.sample = local / sg.cps;
  p.k = local - p.sample * sg.cps;
  p.tl = threadIdx.x / LANES;
  p.q = threadIdx.x % LANES;
  p.lt = p.k * CK_TILES + p.tl;
  const int tps = sg.th * sg.tw;
  p.tile_ok = p.lt < tps;
  p.t_global = sg.tile_start + p.sample * tps + min(p.lt, tps - 1);
  p.cl = p.q * W;
  p.c0 = slab * (CK_LANES * W) + p.cl;
  return p;
}

template <int W>
__device__ __forceinline__ typename VecW<W>::type ldv(const float* p) { return *reinterpret_cast<const typename VecW<W>::type*>(p); }
template <int W>
__device__ __forceinline__ typename VecW<W>::type lanes(const float (*st)[2], int cl, int cpg, int comp) {
  if constexpr (W == 1) {
    return st[cl / cpg][comp];
  } else {
    typename VecW<W>::type v;
#pragma unroll
    for (int i = 0; i < W; ++i) v[i] = st[(cl + i) / cpg][comp];
    return v;
  }
}
template <int W, bool GRAD, int ACT>
__device__ __forceinline__ typename VecW<W>::type map_act(typename VecW<W>::type z) {
  if constexpr (W == 1) {
    return GRAD ? rn::act_grad(z, ACT) : rn::act_fwd(z, ACT);
  } else {
    typename VecW<W>::type o;
#pragma unroll
    for (int i = 0; i < W; ++i) o[i] = GRAD ? rn::act_grad(z[i], ACT) : rn::act_fwd(z[i], ACT);
    return o;
  }
}
// the activation is a runtime argument of the call but a compile-time constant of the per-element code: ONE uniform
// switch around the whole loop (a switch per element puts a branch between the loads of consecutive taps, and the
// compiler then waits for each load before issuing the next: 36 serialised round trips)
template <int A>
struct ActTag { static constexpr int value = A; };
template <typename F>
__device__ __forceinline__ void dispatch_act(int act, F fn) {
  switch (act) {
    case RN_ACT_RELU: fn(ActTag<RN_ACT_RELU>{}); break;
    case RN_ACT_ELU: fn(ActTag<RN_ACT_ELU>{}); break;
    case RN_ACT_RELU6: fn(ActTag<RN_ACT_RELU6>{}); break;
    default: fn(ActTag<RN_ACT_NONE>{}); break;
  }
}
template <int W>
__device__ __forceinline__ float lane_of(typename VecW<W>::type v, int i) {
  if constexpr (W == 1) return v; else return v[i];
}

// mean / rstd of the block's groups -> st[g_local][0..1]: the stat rows hold (count, mean, M2 = sum (y - mean)^2) per
// (chunk, group); they are combined in fp64 in chunk order, mean first, then M2 about it -- no E[y^2] - E[y]^2
// cancellation, however few values a group has (a 1x1 pyramid level has C / groups of them).  The rows of the sample
// are fetched by all threads at once into LDS (one round trip instead of one per row), then summed in order.
constexpr int STAGE_ROWS = 1024;  // (chunk, group) rows staged per block: 16 KB
__device__ __forceinline__ void block_stats(const Fold& f, const CSeg& sg, int sample, int slab, int ch, float (*st)[2]) {
  __shared__ float4 stage[STAGE_ROWS];
  const int ng = ch / f.cpg, g0 = slab * ng, total = ng * sg.cps, tid = threadIdx.x;
  const float* p = f.rows + ((size_t)(sg.chunk_start + sample * sg.cps) * f.groups + g0) * ROW_F;
  const bool staged = total <= STAGE_ROWS;
  if (staged) {
    for (int i = tid; i < total; i += blockDim.x) {
      const int k = i / ng, gl = i - k * ng;
      stage[i] = *reinterpret_cast<const float4*>(p + ((size_t)k * f.groups + gl) * ROW_F);
    }
    __syncthreads();
  }
  if (tid < ng) {
    auto row = [&](int k) { return staged ? stage[k * ng + tid] : *reinterpret_cast<const float4*>(p + ((size_t)k * f.groups + tid) * ROW_F); };
    double n = 0.0, sm = 0.0;
    for (int k = 0; k < sg.cps; ++k) {
      const float4 r = row(k);
      n += (double)r.x;
      sm += (double)r.x * (double)r.y;
    }
    const double mean = n > 0.0 ? sm / n : 0.0;
    double m2 = 0.0;
    for (int k = 0; k < sg.cps; ++k) {
      const float4 r = row(k);
      const double d = (double)r.y - mean;
      m2 += (double)r.z + (double)r.x * d * d;
    }
    const double var = n > 0.0 ? m2 / n : 0.0;
    st[tid][0] = (float)mean;
    st[tid][1] = (float)(1.0 / sqrt(var + (double)f.eps));
  }
  __syncthreads();
}
// (sum gamma g, sum gamma g xhat) / m of the block's groups -> co[g_local][0..1]
__device__ __forceinline__ void block_coefs(const Fold& f, const CSeg& sg, int sample, int slab, int ch, float (*co)[2]) {
  __shared__ float2 stage2[STAGE_ROWS];
  const int ng = ch / f.cpg, g0 = slab * ng, total = ng * sg.cps, tid = threadIdx.x;
  const float* p = f.grows + ((size_t)(sg.chunk_start + sample * sg.cps) * f.groups + g0) * 2;
  const bool staged = total <= STAGE_ROWS;
  if (staged) {
    for (int i = tid; i < total; i += blockDim.x) {
      const int k = i / ng, gl = i - k * ng;
      stage2[i] = *reinterpret_cast<const float2*>(p + ((size_t)k * f.groups + gl) * 2);
    }
    __syncthreads();
  }
  if (tid < ng) {
    double a0 = 0.0, a1 = 0.0;
    for (int k = 0; k < sg.cps; ++k) {
      const float2 r = staged ? stage2[k * ng + tid] : *reinterpret_cast<const float2*>(p + ((size_t)k * f.groups + tid) * 2);
      a0 += (double)r.x; a1 += (double)r.y;
    }
    const double m = (double)sg.h * (double)sg.w * (double)f.cpg;
    co[tid][0] = (float)(a0 / m);
    co[tid][1] = (float)(a1 / m);
  }
  __syncthreads();
}

// ---- tap loaders: fetch() issues the loads of one tap (element offset `off`, this thread's W channels), fin<ACT>() turns
// them into the value the transform sees.  The bodies fetch every tap first and finish them in a second loop, so all
// the loads of a thread are in flight together.
template <int W>
struct LoadPlain {
  typedef typename VecW<W>::type VT;
  static constexpr int NLOAD = 1;
  const float* x;
  __device__ __forceinline__ void init(const Fold&, const CSeg& sg, const BlockPos&, int) { x = sg.x; }
  __device__ __forceinline__ int act_id() const { return RN_ACT_NONE; }
  __device__ __forceinline__ void fetch(size_t off, VT& a, VT&) const { a = ldv<W>(x + off); }
  template <int ACT>
  __device__ __forceinline__ VT fin(VT a, VT) const { return a; }
};
template <int W>
struct LoadGnAct {   // act(GN(x)): x * sc + sh per channel
  typedef typename VecW<W>::type VT;
  static constexpr int NLOAD = 1;
  const float* x; VT sc, sh; int act;
  __device__ __forceinline__ void init(const Fold& f, const CSeg& sg, const BlockPos& p, int slab) {
    __shared__ float st[CK_MAXCH][2];
    block_stats(f, sg, p.sample, slab, CK_LANES * W, st);
    const VT mean = lanes<W>(st, p.cl, f.cpg, 0), rstd = lanes<W>(st, p.cl, f.cpg, 1);
    x = sg.x; act = f.act;
    sc = rstd * ldv<W>(f.gamma + p.c0);
    sh = ldv<W>(f.beta + p.c0) - mean * sc;
  }
  __device__ __forceinline__ int act_id() const { return act; }
  __device__ __forceinline__ void fetch(size_t off, VT& a, VT&) const { a = ldv<W>(x + off); }
  template <int ACT>
  __device__ __forceinline__ VT fin(VT a, VT) const { return map_act<W, false, ACT>(a * sc + sh); }
};
template <int W>
struct LoadGnBwd {   // dy = rstd (gamma g - c1 - xhat c2), xhat = (y - mean) rstd; g in `x`, y in `aux`
  typedef typename VecW<W>::type VT;
  static constexpr int NLOAD = 2;
  const float* x; const float* aux; VT mean, rstd, gam, c1, c2;
  __device__ __forceinline__ void init(const Fold& f, const CSeg& sg, const BlockPos& p, int slab) {
    __shared__ float st[CK_MAXCH][2], co[CK_MAXCH][2];
    block_stats(f, sg, p.sample, slab, CK_LANES * W, st);
    block_coefs(f, sg, p.sample, slab, CK_LANES * W, co);
    x = sg.x; aux = sg.aux;
    mean = lanes<W>(st, p.cl, f.cpg, 0); rstd = lanes<W>(st, p.cl, f.cpg, 1);
    c1 = lanes<W>(co, p.cl, f.cpg, 0); c2 = lanes<W>(co, p.cl, f.cpg, 1);
    gam = ldv<W>(f.gamma + p.c0);
  }
  __device__ __forceinline__ int act_id() const { return RN_ACT_NONE; }
  __device__ __forceinline__ void fetch(size_t off, VT& g, VT& y) const { g = ldv<W>(x + off); y = ldv<W>(aux + off); }
  template <int ACT>
  __device__ __forceinline__ VT fin(VT g, VT y) const { return rstd * (gam * g - c1 - (y - mean) * rstd * c2); }
};

// ---- input transform of a chunk: V[xi][tile][c] = (B^T d B)[xi], d = the (M+2)^2 patch read through loader L
template <int M, int W, template <int> class LT>
__device__ __forceinline__ void chunk_input_body(const CArgs& a, const Fold& f, int chunk, int slab) {
  typedef typename VecW<W>::type VT;
  constexpr int P = M + 2;
  const BlockPos p = block_pos<W>(a, chunk, slab);
  const CSeg& sg = a.seg[p.s];
  LT<W> ld;
  ld.init(f, sg, p, slab);
  if (!p.tile_ok) return;
  const int ty = p.lt / sg.tw, tx = p.lt - ty * sg.tw;
  const int y0 = M * ty - 1, x0 = M * tx - 1;
  const size_t plane = (size_t)a.total_tiles * a.c;
  VT d[P][P], e[LT<W>::NLOAD == 2 ? P : 1][LT<W>::NLOAD == 2 ? P : 1];
#pragma unroll
  for (int r = 0; r < P; ++r) {
    const int yc = min(max(y0 + r, 0), sg.h - 1);
#pragma unroll
    for (int c = 0; c < P; ++c) {   // clamped addresses: every load is unconditional
      const int xc = min(max(x0 + c, 0), sg.w - 1);
      ld.fetch(((size_t)(p.sample * sg.h + yc) * sg.w + xc) * a.c + p.c0, d[c][r], e[LT<W>::NLOAD == 2 ? c : 0][LT<W>::NLOAD == 2 ? r : 0]);
    }
  }
  dispatch_act(ld.act_id(), [&](auto tag) {
#pragma unroll
    for (int r = 0; r < P; ++r)
#pragma unroll
      for (int c = 0; c < P; ++c) {  // 0/1 mask: the padding is zero AFTER the activation
        const bool ok = (unsigned)(y0 + r) < (unsigned)sg.h && (unsigned)(x0 + c) < (unsigned)sg.w;
        d[c][r] = ld.template fin<decltype(tag)::value>(d[c][r], e[LT<W>::NLOAD == 2 ? c : 0][LT<W>::NLOAD == 2 ? r : 0]) * (ok ? 1.f : 0.f);
      }
  });
  VT tm[P][P];
#pragma unroll
  for (int c = 0; c < P; ++c) {
    VT col[P];
    Wino<M>::bt(d[c], col);
#pragma unroll
    for (int r = 0; r < P; ++r) tm[r][c] = col[r];
  }
  float* out = a.buf + (size_t)p.t_global * a.c + p.c0;
#pragma unroll
  for (int r = 0; r < P; ++r) {
    VT row[P];
    Wino<M>::bt(tm[r], row);
#pragma unroll
    for (int c = 0; c < P; ++c) *reinterpret_cast<VT*>(out + (size_t)(r * P + c) * plane) = row[c];
  }
}

// ---- A dY A^T of a chunk (weight gradient), dY read through loader L
template <int M, int W, template <int> class LT>
__device__ __forceinline__ void chunk_dy_body(const CArgs& a, const Fold& f, int chunk, int slab) {
  typedef typename VecW<W>::type VT;
  constexpr int P = M + 2;
  const BlockPos p = block_pos<W>(a, chunk, slab);
  const CSeg& sg = a.seg[p.s];
  LT<W> ld;
  ld.init(f, sg, p, slab);
  if (!p.tile_ok) return;
  const int ty = p.lt / sg.tw, tx = p.lt - ty * sg.tw;
  const int y0 = M * ty, x0 = M * tx;
  const size_t plane = (size_t)a.total_tiles * a.c;
  VT d[M][M], e[LT<W>::NLOAD == 2 ? M : 1][LT<W>::NLOAD == 2 ? M : 1];
#pragma unroll
  for (int c = 0; c < M; ++c)
#pragma unroll
    for (int r = 0; r < M; ++r) {
      const int yc = min(y0 + r, sg.h - 1), xc = min(x0 + c, sg.w - 1);
      ld.fetch(((size_t)(p.sample * sg.h + yc) * sg.w + xc) * a.c + p.c0, d[c][r], e[LT<W>::NLOAD == 2 ? c : 0][LT<W>::NLOAD == 2 ? r : 0]);
    }
  dispatch_act(ld.act_id(), [&](auto tag) {
#pragma unroll
    for (int c = 0; c < M; ++c)
#pragma unroll
      for (int r = 0; r < M; ++r) {
        const bool ok = y0 + r < sg.h && x0 + c < sg.w;
        d[c][r] = ld.template fin<decltype(tag)::value>(d[c][r], e[LT<W>::NLOAD == 2 ? c : 0][LT<W>::NLOAD == 2 ? r : 0]) * (ok ? 1.f : 0.f);
      }
  });
  VT tm[P][M];
#pragma unroll
  for (int c = 0; c < M; ++c) {
    VT col[M], o[P];
#pragma unroll
    for (int r = 0; r < M; ++r) col[r] = d[c][r];
    Wino<M>::a(col, o);
#pragma unroll
    for (int r = 0; r < P; ++r) tm[r][c] = o[r];
  }
  float* out = a.buf + (size_t)p.t_global * a.c + p.c0;
#pragma unroll
  for (int r = 0; r < P; ++r) {
    VT row[P];
    Wino<M>::a(tm[r], row);
#pragma unroll
    for (int c = 0; c < P; ++c) *reinterpret_cast<VT*>(out + (size_t)(r * P + c) * plane) = row[c];
  }
}

// ---- output transform of a chunk.  MODE 0: y = A^T m A (+ bias), optional stat rows of y.
//      MODE 1 (data gradient into a folded GroupNorm): o = dA; g = o * act'(z) from the raw tensor in seg.aux; writes g
//      and the rows (sum g, sum g xhat) per channel and gamma-weighted per group.
template <int M, int W, int MODE>
__device__ __forceinline__ void chunk_output_body(const CArgs& a, const Fold& f, int chunk, int slab) {
  typedef typename VecW<W>::type VT;
  constexpr int P = M + 2, LANES = CK_LANES, T = CK_T, CH = CK_LANES * W;
  __shared__ float red[T][2 * W];
  __shared__ float chan[CH][3];
  __shared__ float st[CH][2];
  __shared__ float cnt[CK_TILES];
  const BlockPos p = block_pos<W>(a, chunk, slab);
  const CSeg& sg = a.seg[p.s];
  const int tid = threadIdx.x;
  VT gam = (VT)(0.f), bet = (VT)(0.f), mean = (VT)(0.f), rstd = (VT)(0.f);
  if (MODE == 1) {
    block_stats(f, sg, p.sample, slab, CH, st);
    mean = lanes<W>(st, p.cl, f.cpg, 0); rstd = lanes<W>(st, p.cl, f.cpg, 1);
    gam = ldv<W>(f.gamma + p.c0); bet = ldv<W>(f.beta + p.c0);
  }
  VT s1 = (VT)(0.f), s2 = (VT)(0.f);
  float nvalid = 0.f;
  const bool want_rows = MODE == 1 || f.out_rows != nullptr;
  if (p.tile_ok) {
    const int ty = p.lt / sg.tw, tx = p.lt - ty * sg.tw;
    const size_t plane = (size_t)a.total_tiles * a.c;
    const float* in = a.buf + (size_t)p.t_global * a.c + p.c0;
    VT rr[M][P];
#pragma unroll
    for (int c = 0; c < P; ++c) {
      VT col[P], o[M];
#pragma unroll
      for (int r = 0; r < P; ++r) col[r] = ldv<W>(in + (size_t)(r * P + c) * plane);
      Wino<M>::at(col, o);
#pragma unroll
      for (int i2 = 0; i2 < M; ++i2) rr[i2][c] = o[i2];
    }
    VT b = (VT)(0.f);
    if (a.bias) b = ldv<W>(a.bias + p.c0);
    const int y0 = M * ty, x0 = M * tx;
    VT vals[M][M], xh[MODE == 1 ? M : 1][MODE == 1 ? M : 1];
    if (MODE == 1) {   // the raw tensor at the tile's pixels (clamped addresses: always valid), all loads in flight
#pragma unroll
      for (int i2 = 0; i2 < M; ++i2)
#pragma unroll
        for (int j = 0; j < M; ++j)
          xh[MODE == 1 ? i2 : 0][MODE == 1 ? j : 0] =
              ldv<W>(sg.aux + ((size_t)(p.sample * sg.h + min(y0 + i2, sg.h - 1)) * sg.w + min(x0 + j, sg.w - 1)) * a.c + p.c0);
    }
#pragma unroll
    for (int i2 = 0; i2 < M; ++i2) {
      VT o[M];
      Wino<M>::at(rr[i2], o);
#pragma unroll
      for (int j = 0; j < M; ++j) vals[i2][j] = o[j] + b;
    }
    if (MODE == 1) {
      dispatch_act(f.act, [&](auto tag) {
#pragma unroll
        for (int i2 = 0; i2 < M; ++i2)
#pragma unroll
          for (int j = 0; j < M; ++j) {
            const VT h = (xh[MODE == 1 ? i2 : 0][MODE == 1 ? j : 0] - mean) * rstd;
            xh[MODE == 1 ? i2 : 0][MODE == 1 ? j : 0] = h;
            vals[i2][j] = vals[i2][j] * map_act<W, true, decltype(tag)::value>(h * gam + bet);
          }
      });
    }
#pragma unroll
    for (int i2 = 0; i2 < M; ++i2)
#pragma unroll
      for (int j = 0; j < M; ++j) {
        const bool ok = y0 + i2 < sg.h && x0 + j < sg.w;
        const size_t off = ((size_t)(p.sample * sg.h + min(y0 + i2, sg.h - 1)) * sg.w + min(x0 + j, sg.w - 1)) * a.c + p.c0;
        const VT v = vals[i2][j];
        if (ok) {
          s1 += v;
          if (MODE == 1) s2 += v * xh[MODE == 1 ? i2 : 0][MODE == 1 ? j : 0];
          else nvalid += 1.f;
          *reinterpret_cast<VT*>(sg.y + off) = v;
        }
      }
    if (MODE == 0 && want_rows) {   // second pass over the registers: M2 about this thread's own per-channel mean
      const VT mu = s1 * (1.f / fmaxf(nvalid, 1.f));
      s1 = mu;
#pragma unroll
      for (int i2 = 0; i2 < M; ++i2)
#pragma unroll
        for (int j = 0; j < M; ++j) {
          const VT d = vals[i2][j] - mu;
          if (y0 + i2 < sg.h && x0 + j < sg.w) s2 += d * d;
        }
    }
  }
  if (!want_rows) return;
#pragma unroll
  for (int i = 0; i < W; ++i) { red[tid][i] = lane_of<W>(s1, i); red[tid][W + i] = lane_of<W>(s2, i); }
  if (p.q == 0) cnt[p.tl] = nvalid;
  __syncthreads();
  const int ng = CH / f.cpg;
  if (MODE == 1) {
    if (tid < CH * 2) {   // per-channel sums over the chunk's 16 tiles, tile order
      const int cl = tid >> 1, comp = tid & 1;
      float t = 0.f;
#pragma unroll
      for (int tl = 0; tl < CK_TILES; ++tl) t += red[tl * LANES + cl / W][comp * W + cl % W];
      chan[cl][comp] = t;
      f.g_rows_chan[((size_t)comp * a.total_chunks + chunk) * a.c + slab * CH + cl] = t;
    }
    __syncthreads();
    if (tid < ng * 2) {
      const int gl = tid >> 1, comp = tid & 1;
      float t = 0.f;
      for (int j = 0; j < f.cpg; ++j) {
        const int cl = gl * f.cpg + j;
        t += f.gamma[slab * CH + cl] * chan[cl][comp];
      }
      f.g_rows_group[((size_t)chunk * f.groups + slab * ng + gl) * 2 + comp] = t;
    }
    return;
  }
  // MODE 0: (count, mean, M2) per channel over the chunk's tiles (mean first, then M2 about it), then per group
  if (tid < CH) {
    const int cl = tid;
    float n = 0.f, sm = 0.f;
#pragma unroll
    for (int tl = 0; tl < CK_TILES; ++tl) { n += cnt[tl]; sm += cnt[tl] * red[tl * LANES + cl / W][cl % W]; }
    const float mu = n > 0.f ? sm / n : 0.f;
    float m2 = 0.f;
#pragma unroll
    for (int tl = 0; tl < CK_TILES; ++tl) {
      const float d = red[tl * LANES + cl / W][cl % W] - mu;
      m2 += red[tl * LANES + cl / W][W + cl % W] + cnt[tl] * d * d;
    }
    chan[cl][0] = n; chan[cl][1] = mu; chan[cl][2] = m2;
  }
  __syncthreads();
  if (tid < ng) {
    float n = 0.f, sm = 0.f;
    for (int j = 0; j < f.cpg; ++j) { n += chan[tid * f.cpg + j][0]; sm += chan[tid * f.cpg + j][0] * chan[tid * f.cpg + j][1]; }
    const float mu = n > 0.f ? sm / n : 0.f;
    float m2 = 0.f;
    for (int j = 0; j < f.cpg; ++j) {
      const float d = chan[tid * f.cpg + j][1] - mu;
      m2 += chan[tid * f.cpg + j][2] + chan[tid * f.cpg + j][0] * d * d;
    }
    *reinterpret_cast<float4*>(f.out_rows + ((size_t)chunk * f.groups + slab * ng + tid) * ROW_F) = make_float4(n, mu, m2, 0.f);
  }
}

// launches: blocks [0, nb_chunk) run a chunk body (chunk = b / slabs, slab = b % slabs) with all T threads; the second
// kind of block (kernel transform, G^T dU G) uses the first NT threads of as many blocks as it needs
template <int W>
constexpr int chunk_threads() { return CK_T; }
// Occupancy target of the backward transforms (waves per SIMD).  A head-tower layer is 736 / 624 blocks of 4 waves: at the
// 172 / 182 registers the compiler picks on its own, two blocks fit a CU and the launch needs two rounds of 512 (the second
// one 44 % / 22 % full); at <= 168 registers three fit and every block is resident at once.
template <int W>
constexpr int chunk_waves() { return W <= 2 ? 3 : 1; }

template <int M, int W, template <int> class LT>
__global__ __launch_bounds__(chunk_threads<W>()) void wino_gn_fwd_pre_kernel(const CArgs in, const Fold f, const WeightArgs wa,
                                                                               int nb_chunk, int slabs) {
  // (the kernel-transform blocks FIRST: they are the longer ones -- 100+ stores per thread -- and would otherwise be the launch's tail)
  const int nb_w = (int)gridDim.x - nb_chunk, b = (int)blockIdx.x - nb_w;
  if (b >= 0) chunk_input_body<M, W, LT>(in, f, b / slabs, b % slabs);
  else wino_weight_any_body<M>(wa, (int)blockIdx.x);
}
template <int M, int W, template <int> class LT>
__global__ __launch_bounds__(chunk_threads<W>()) void wino_gn_input_kernel(const CArgs in, const Fold f, int slabs) {
  chunk_input_body<M, W, LT>(in, f, blockIdx.x / slabs, blockIdx.x % slabs);
}
template <int M, int W>
__global__ __launch_bounds__(chunk_threads<W>()) void wino_gn_output_kernel(const CArgs out, const Fold f, int slabs) {
  chunk_output_body<M, W, 0>(out, f, blockIdx.x / slabs, blockIdx.x % slabs);
}
template <int M, int W, template <int> class LT>
__global__ __launch_bounds__(chunk_threads<W>(), chunk_waves<W>()) void wino_gn_bwd_pre_kernel(const CArgs dgrad_in, const CArgs wgrad_dy, const Fold f,
                                                                               int nb_first, int slabs) {
  const int b = blockIdx.x;
  if (b < nb_first) chunk_input_body<M, W, LT>(dgrad_in, f, b / slabs, b % slabs);
  else chunk_dy_body<M, W, LT>(wgrad_dy, f, (b - nb_first) / slabs, (b - nb_first) % slabs);
}
template <int M, int W, int MODE>
__global__ __launch_bounds__(chunk_threads<W>(), chunk_waves<W>()) void wino_gn_bwd_post_kernel(const CArgs out, const Fold f, const DwArgs2 d, int nb_out,
                                                                                int slabs) {
  const int b = blockIdx.x;
  if (b < nb_out) chunk_output_body<M, W, MODE>(out, f, b / slabs, b % slabs);
  else wino_dw_body<M>(d.du, d.dw, d.kn, d.nsplit, d.accumulate, b - nb_out);
}

int cfill(const rn_conv_seg* segs, int nseg, int c, int m, CArgs* a, const float* const* xs, const float* const* auxs, float* const* ys) {
  int64_t tiles = 0, chunks = 0;
  for (int s = 0; s < nseg; ++s) {
    CSeg& d = a->seg[s];
    d.n = segs[s].n; d.h = segs[s].h; d.w = segs[s].w;
    d.th = (d.h + m - 1) / m; d.tw = (d.w + m - 1) / m;
    d.tile_start = (int)tiles; d.chunk_start = (int)chunks;
    d.cps = (d.th * d.tw + CK_TILES - 1) / CK_TILES;
    d.x = xs ? xs[s] : nullptr; d.aux = auxs ? auxs[s] : nullptr; d.y = ys ? ys[s] : nullptr;
    tiles += (int64_t)d.n * d.th * d.tw;
    chunks += (int64_t)d.n * d.cps;
  }
  a->nseg = nseg; a->c = c; a->total_tiles = (int)tiles; a->total_chunks = (int)chunks;
  return RN_OK;
}

// a block covers 16 W channels, W in {1, 2, 4}: they must be whole groups on a folded side
bool width_ok(int c, int cpg, int w) { return c % (CK_LANES * w) == 0 && (cpg == 0 || (CK_LANES * w) % cpg == 0); }
bool fold_ok(int c, int groups) { return groups >= 1 && c % groups == 0 && width_ok(c, c / groups, 4); }
// channels per thread of the chunked kernels: 4 once that still fills the chip (>= 4 waves per SIMD on 256 CUs), else 2
// -- the pyramids of a 512^2 batch are latency-, not bandwidth-bound, and narrower lanes put more blocks in flight
// (measured on one head subnet, forward + backward: W = 1 / 2 / 4 -> 1215 / 1144 / 1176 us; layer by layer with
// stand-alone GroupNorm kernels: 1185 us)
int chunk_width(int64_t tiles, int c, int cpg) {
  if (const char* force = getenv("RN_WINO_GN_W")) {  // tuning aid
    const int w = atoi(force);
    if ((w == 1 || w == 2 || w == 4) && width_ok(c, cpg, w)) return w;
  }
  if (tiles * (c / 4) >= 256 * 1024 || !width_ok(c, cpg, 2)) return 4;
  return 2;
}
}  // namespace

extern "C" size_t rn_wino_gn_rows(const rn_conv_seg* segs, int nseg, int tile) {
  if (!segs || nseg < 1 || nseg > RN_MAX_SEG || (tile != 2 && tile != 4)) return 0;
  size_t chunks = 0;
  for (int s = 0; s < nseg; ++s) {
    const size_t tps = (size_t)((segs[s].h + tile - 1) / tile) * ((segs[s].w + tile - 1) / tile);
    chunks += (size_t)segs[s].n * ((tps + CK_TILES - 1) / CK_TILES);
  }
  return chunks;
}

namespace {
#define RN_WGN(W_, CALL)                          \
  do {                                            \
    if (W_ == 4) { constexpr int W = 4; CALL; }   \
    else if (W_ == 2) { constexpr int W = 2; CALL; } \
    else { constexpr int W = 1; CALL; }           \
  } while (0)

template <int M>
int run_gn_fwd(const rn_conv_seg* segs, int nseg, int cin, int cout, const float* w, const float* bias, const rn_wino_gn* gn, float* U,
               float* V, float* Mb, float* v_buf, float* urot_buf, hipStream_t st) {
  constexpr int P2 = (M + 2) * (M + 2);
  const float* xs[RN_MAX_SEG]; float* ys[RN_MAX_SEG];
  for (int s = 0; s < nseg; ++s) { xs[s] = segs[s].x; ys[s] = segs[s].y; }
  CArgs ia = {}, oa = {};
  cfill(segs, nseg, cin, M, &ia, xs, nullptr, nullptr);
  cfill(segs, nseg, cout, M, &oa, nullptr, nullptr, ys);
  if (v_buf) V = v_buf;
  ia.buf = V;
  const int64_t kn = (int64_t)cin * cout;
  // the kernel transform writes U as the forward product kernel's pre-split fragment-ordered operand where that kernel takes one
  // (rn::x3_bfrag_ok: product mode 1, cin % 16 == 0): the same workspace region, 6 instead of 4 bytes per element.  Urot (kept for the
  // backward pass) stays fp32: the data-gradient product reads its kernel operand k-contiguous, whole 16-byte fragments from LDS, and
  // gains nothing from the image (measured: 25.9 vs 25.8 us; the forward product 25.7 vs 27.8 us)
  const bool u_frag = rn::x3_bfrag_ok(ia.total_tiles, cin, cout);
  WeightArgs wa = {w, u_frag ? nullptr : U, urot_buf, kn, 0, u_frag ? (unsigned*)U : nullptr, cin, cout};
  // the caller transformed the kernel ahead of the layer (rn_conv3x3_winograd_gn_weights: once per step, off the layers' critical
  // path): no kernel-transform blocks in this launch
  if (gn->u_ready) {
    RN_UNSUPPORTED(u_frag != rn::x3_bfrag_format(cin, cout), "winograd gn: the prepared kernel transform has another format than this launch needs");
    U = (float*)gn->u_ready;
    wa.u = nullptr; wa.uf = nullptr;
  }
  if (gn->urot_ready) wa.u2 = nullptr;
  Fold fi = {};
  const bool fold_in = gn->in_rows != nullptr;
  if (fold_in) {
    fi.rows = gn->in_rows; fi.gamma = gn->in_gamma; fi.beta = gn->in_beta; fi.groups = gn->in_groups; fi.cpg = cin / gn->in_groups;
    fi.act = gn->in_act; fi.eps = gn->in_eps;
  }
  const int wi = chunk_width(ia.total_tiles, cin, fold_in ? fi.cpg : 0);
  const int slabs_in = cin / (CK_LANES * wi), nb_chunk = ia.total_chunks * slabs_in, nb_w = weight_blocks(wa);
  if (fold_in) RN_WGN(wi, hipLaunchKernelGGL((wino_gn_fwd_pre_kernel<M, W, LoadGnAct>), dim3(nb_chunk + nb_w), dim3(chunk_threads<W>()), 0, st, ia, fi, wa, nb_chunk, slabs_in));
  else RN_WGN(wi, hipLaunchKernelGGL((wino_gn_fwd_pre_kernel<M, W, LoadPlain>), dim3(nb_chunk + nb_w), dim3(chunk_threads<W>()), 0, st, ia, fi, wa, nb_chunk, slabs_in));
  RN_LAUNCH_CHECK();
  if (u_frag) {
    if (int e = rn::launch_batched_gemm_x3_bfrag(V, U, Mb, ia.total_tiles, cin, cout, P2, 1, st)) return e;
  } else if (int e = rn::launch_batched_gemm(V, U, Mb, ia.total_tiles, cin, cout, P2, 0, st)) {
    return e;
  }
  if (!gn->out_rows) {  // no statistics wanted: the grid-stride output transform (any cout % 4 == 0, e.g. the 720 class maps)
    WArgs wia = {}, woa = {};
    if (int e = fill(segs, nseg, cin, cout, M, false, &wia, &woa)) return e;
    woa.buf = Mb; woa.bias = bias;
    const int wd = width_for(woa.total_tiles, cout);
    RN_WINO_LAUNCH(wino_output_kernel, wd, (int64_t)woa.total_tiles * cout, woa);
    RN_LAUNCH_CHECK();
    return RN_OK;
  }
  oa.buf = Mb; oa.bias = bias;
  Fold fo = {};
  fo.out_rows = gn->out_rows; fo.groups = gn->out_groups; fo.cpg = cout / gn->out_groups;
  const int wo = chunk_width(oa.total_tiles, cout, fo.cpg);
  const int slabs_out = cout / (CK_LANES * wo);
  RN_WGN(wo, hipLaunchKernelGGL((wino_gn_output_kernel<M, W>), dim3(oa.total_chunks * slabs_out), dim3(chunk_threads<W>()), 0, st, oa, fo, slabs_out));
  RN_LAUNCH_CHECK();
  return RN_OK;
}

template <int M>
int run_gn_bwd(const rn_conv_seg* segs, int nseg, int cin, int cout, const float* w, float* dw, int accumulate, const rn_wino_gn_bwd* gn,
               char* ws, const BwdLayout& L, size_t ws_bytes, const float* v_buf, const float* urot_buf, hipStream_t st) {
  constexpr int P2 = (M + 2) * (M + 2);
  const float* dys[RN_MAX_SEG]; const float* youts[RN_MAX_SEG]; const float* xins[RN_MAX_SEG]; float* dxs[RN_MAX_SEG];
  for (int s = 0; s < nseg; ++s) { dys[s] = segs[s].dy; youts[s] = segs[s].y; xins[s] = segs[s].x; dxs[s] = segs[s].dx; }
  const bool fold_out = gn->out_rows != nullptr, fold_in = gn->in_rows != nullptr;
  CArgs ia = {}, ya = {}, oa = {};
  cfill(segs, nseg, cout, M, &ia, dys, fold_out ? youts : nullptr, nullptr);   // dy (or g + raw y) -> B^T . B
  ya = ia;                                                                      // the same tensor  -> A . A^T
  cfill(segs, nseg, cin, M, &oa, nullptr, fold_in ? xins : nullptr, dxs);      // products -> dx (or g of the input's GroupNorm)
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
  Fold fi = {};
  if (fold_in) {
    fi.rows = gn->in_rows; fi.gamma = gn->in_gamma; fi.beta = gn->in_beta; fi.groups = gn->in_groups; fi.cpg = cin / gn->in_groups;
    fi.act = gn->in_act; fi.eps = gn->in_eps; fi.g_rows_group = gn->in_g_rows_group; fi.g_rows_chan = gn->in_g_rows_chan;
  }
  const int wi = chunk_width(T_, cin, fold_in ? fi.cpg : 0);
  if (!V) {  // the forward pass did not keep its transformed input: rebuild it (through the folded GroupNorm if there is one)
    CArgs xa = {};
    cfill(segs, nseg, cin, M, &xa, xins, nullptr, nullptr);
    xa.buf = (float*)(ws + L.v);
    const int slabs = cin / (CK_LANES * wi);
    if (fold_in) RN_WGN(wi, hipLaunchKernelGGL((wino_gn_input_kernel<M, W, LoadGnAct>), dim3(xa.total_chunks * slabs), dim3(chunk_threads<W>()), 0, st, xa, fi, slabs));
    else RN_WGN(wi, hipLaunchKernelGGL((wino_gn_input_kernel<M, W, LoadPlain>), dim3(xa.total_chunks * slabs), dim3(chunk_threads<W>()), 0, st, xa, fi, slabs));
    V = xa.buf;
  }
  ia.buf = Vdy;
  ya.buf = dM;
  if (fold_out) {
    Fold fo = {};
    fo.rows = gn->out_rows; fo.grows = gn->out_g_rows_group; fo.gamma = gn->out_gamma; fo.groups = gn->out_groups;
    fo.cpg = cout / gn->out_groups; fo.eps = gn->out_eps;
    const int wy = chunk_width(T_, cout, fo.cpg);
    const int slabs = cout / (CK_LANES * wy), nb = ia.total_chunks * slabs;
    RN_WGN(wy, hipLaunchKernelGGL((wino_gn_bwd_pre_kernel<M, W, LoadGnBwd>), dim3(2 * nb), dim3(chunk_threads<W>()), 0, st, ia, ya, fo, nb, slabs));
  } else {  // plain dy (any cout % 4 == 0): the grid-stride transforms
    WArgs wia = {}, woa = {}, wya = {}, unused = {};
    if (int e = fill(segs, nseg, cin, cout, M, true, &wia, &woa, true)) return e;
    if (int e = fill(segs, nseg, cin, cout, M, true, &wya, &unused, true)) return e;
    wia.buf = Vdy; wya.buf = dM;
    const int wy = width_for(T_, cout);
    const int nb = (int)grid_for((int64_t)T_ * cout / wy);
    if (wy == 4) hipLaunchKernelGGL((wino_bwd_pre_kernel<M, 4>), dim3(2 * nb), dim3(NT), 0, st, wia, wya, nb);
    else if (wy == 2) hipLaunchKernelGGL((wino_bwd_pre_kernel<M, 2>), dim3(2 * nb), dim3(NT), 0, st, wia, wya, nb);
    else hipLaunchKernelGGL((wino_bwd_pre_kernel<M, 1>), dim3(2 * nb), dim3(NT), 0, st, wia, wya, nb);
  }
  RN_LAUNCH_CHECK();
  int nsplit = 1;
  const bool defer = gn->defer_wgrad != 0;      // (the weight-gradient half runs later from this workspace: run_gn_bwd_wgrad)
  if (defer) {
    if (int e = rn::launch_batched_gemm(Vdy, Urot, Mdx, T_, cout, cin, P2, 1, st)) return e;
  } else if (int e = rn::launch_winograd_bwd_products(Vdy, Urot, Mdx, T_, cout, cin, V, dM, cin, cout, P2, ws + L.slab, ws_bytes - L.slab, st,
                                                      &nsplit)) {
    return e;
  }
  oa.buf = Mdx; oa.bias = nullptr;
  {
    const DwArgs2 d = {(const float*)(ws + L.slab), dw, kn, nsplit, accumulate};
    const int slabs = cin / (CK_LANES * wi), nb_out = oa.total_chunks * slabs, nb_dw = defer ? 0 : (int)rn::ceil_div64(kn, NT);
    if (!fold_in) { fi.groups = 1; fi.cpg = CK_LANES * wi; }
    if (fold_in) RN_WGN(wi, hipLaunchKernelGGL((wino_gn_bwd_post_kernel<M, W, 1>), dim3(nb_out + nb_dw), dim3(chunk_threads<W>()), 0, st, oa, fi, d, nb_out, slabs));
    else RN_WGN(wi, hipLaunchKernelGGL((wino_gn_bwd_post_kernel<M, W, 0>), dim3(nb_out + nb_dw), dim3(chunk_threads<W>()), 0, st, oa, fi, d, nb_out, slabs));
  }
  RN_LAUNCH_CHECK();
  return RN_OK;
}
#undef RN_WGN

// the deferred half of run_gn_bwd: dU = V^T dM over the tiles (partial slabs), dw (+)= G^T dU G
template <int M>
int run_gn_bwd_wgrad(const rn_conv_seg* segs, int nseg, int cin, int cout, float* dw, int accumulate, char* ws, const BwdLayout& L,
                     size_t ws_bytes, const float* v_buf, hipStream_t st) {
  constexpr int P2 = (M + 2) * (M + 2);
  const int T_ = (int)tiles_of(segs, nseg, M);
  const int64_t kn = (int64_t)cin * cout;
  const float* V = v_buf ? v_buf : (const float*)(ws + L.v);
  const float* dM = (const float*)(ws + L.dm);
  int nsplit = 1;
  if (int e = rn::launch_batched_gemm_tn(V, dM, nullptr, T_, cin, cout, P2, ws + L.slab, ws_bytes - L.slab, st, &nsplit, 1)) return e;
  hipLaunchKernelGGL(wino_dw_kernel<M>, dim3((unsigned)rn::ceil_div64(kn, NT)), dim3(NT), 0, st, (const float*)(ws + L.slab), dw, kn, nsplit,
                     accumulate);
  RN_LAUNCH_CHECK();
  return RN_OK;
}

int check_fold(const rn_conv_seg* segs, int nseg, int cin, int cout, int tile) {
  RN_CHECK_ARG(segs && nseg >= 1 && nseg <= RN_MAX_SEG && (tile == 2 || tile == 4), "winograd gn: bad segments / tile");
  RN_UNSUPPORTED(cin % CK_MAXCH != 0 || cout % 4 != 0, "winograd gn: cin %d must be a multiple of %d, cout %d of 4", cin, CK_MAXCH, cout);
  for (int s = 0; s < nseg; ++s) {
    RN_CHECK_ARG(segs[s].n >= 1 && segs[s].h >= 1 && segs[s].w >= 1, "winograd gn: bad segment %d", s);
    RN_UNSUPPORTED((double)segs[s].n * segs[s].h * segs[s].w * (cin > cout ? cin : cout) * 4.0 >= 2147483648.0,
                   "winograd gn: segment %d >= 2 GiB", s);
  }
  return RN_OK;
}
}  // namespace

// The kernel transforms of a folded layer ahead of the layer: U in the format its forward product reads (fp32 [P][cin][cout], or the
// pre-split fragment image when rn_x3_bfrag_ok) -> u_out (rn_conv3x3_winograd_gn_u_bytes), the rotated kernel's transform (fp32, for
// the data gradient) -> urot_out (rn_conv3x3_winograd_keep_bytes' urot_bytes; NULL: not wanted).  Pass them to
// rn_conv3x3_winograd_gn as gn->u_ready / urot_buf + gn->urot_ready.
extern "C" size_t rn_conv3x3_winograd_gn_u_bytes(int cin, int cout, int tile) {
  if (cin < 1 || cout < 1 || (tile != 2 && tile != 4)) return 0;
  return u_bytes((size_t)(tile + 2) * (tile + 2), cin, cout);
}
extern "C" int rn_conv3x3_winograd_gn_weights(const float* w, int cin, int cout, int tile, void* u_out, size_t u_out_bytes, float* urot_out,
                                              rn_stream_t stream) {
  RN_CHECK_ARG(w && u_out && (tile == 2 || tile == 4), "winograd gn weights: bad argument");
  RN_UNSUPPORTED(cin % CK_MAXCH != 0 || cout % 4 != 0, "winograd gn weights: cin %d must be a multiple of %d, cout %d of 4", cin, CK_MAXCH, cout);
  const size_t need = rn_conv3x3_winograd_gn_u_bytes(cin, cout, tile);
  if (u_out_bytes < need) {
    rn::set_error("winograd gn weights: u_out %zu < %zu bytes", u_out_bytes, need);
    return RN_EWORKSPACE;
  }
  const bool u_frag = rn::x3_bfrag_format(cin, cout);
  const WeightArgs wa = {w, u_frag ? nullptr : (float*)u_out, urot_out, (int64_t)cin * cout, 0, u_frag ? (unsigned*)u_out : nullptr, cin, cout};
  hipStream_t st = (hipStream_t)stream;
  if (tile == 2) hipLaunchKernelGGL(wino_weight_any_kernel<2>, dim3((unsigned)weight_blocks(wa)), dim3(NT), 0, st, wa);
  else hipLaunchKernelGGL(wino_weight_any_kernel<4>, dim3((unsigned)weight_blocks(wa)), dim3(NT), 0, st, wa);
  RN_LAUNCH_CHECK();
  return RN_OK;
}

extern "C" int rn_conv3x3_winograd_gn_bwd_wgrad(const rn_conv_seg* segs, int nseg, int cin, int cout, float* dw, int accumulate, int tile,
                                                void* workspace, size_t workspace_bytes, const float* v_buf, int urot_was_given,
                                                rn_stream_t stream) {
  if (int e = check_fold(segs, nseg, cin, cout, tile)) return e;
  RN_CHECK_ARG(dw && workspace, "winograd gn bwd wgrad: null pointer");
  const BwdLayout L = bwd_layout(segs, nseg, cin, cout, tile, v_buf == nullptr, urot_was_given == 0);
  if (workspace_bytes < L.total) {
    rn::set_error("winograd gn bwd wgrad: workspace %zu < %zu bytes", workspace_bytes, L.total);
    return RN_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  return tile == 2 ? run_gn_bwd_wgrad<2>(segs, nseg, cin, cout, dw, accumulate, (char*)workspace, L, workspace_bytes, v_buf, st)
                   : run_gn_bwd_wgrad<4>(segs, nseg, cin, cout, dw, accumulate, (char*)workspace, L, workspace_bytes, v_buf, st);
}

extern "C" int rn_conv3x3_winograd_gn(const rn_conv_seg* segs, int nseg, int cin, int cout, const float* w, const float* bias, int tile,
                                      const rn_wino_gn* gn, void* workspace, size_t workspace_bytes, float* v_buf, float* urot_buf,
                                      rn_stream_t stream) {
  if (int e = check_fold(segs, nseg, cin, cout, tile)) return e;
  RN_CHECK_ARG(w && workspace && gn, "winograd gn: null pointer");
  for (int s = 0; s < nseg; ++s) RN_CHECK_ARG(segs[s].x && segs[s].y, "winograd gn: null tensor in segment %d", s);
  if (gn->in_rows) {
    RN_CHECK_ARG(gn->in_gamma && gn->in_beta, "winograd gn: null gamma / beta");
    RN_UNSUPPORTED(!fold_ok(cin, gn->in_groups), "winograd gn: %d channels in %d groups cannot be folded", cin, gn->in_groups);
  }
  if (gn->out_rows) RN_UNSUPPORTED(!fold_ok(cout, gn->out_groups), "winograd gn: %d channels in %d groups cannot be folded", cout, gn->out_groups);
  const size_t need = rn_conv3x3_winograd_workspace(segs, nseg, cin, cout, tile);
  if (workspace_bytes < need) {
    rn::set_error("winograd gn: workspace %zu < %zu bytes", workspace_bytes, need);
    return RN_EWORKSPACE;
  }
  const size_t p2 = (size_t)(tile + 2) * (tile + 2), tiles = tiles_of(segs, nseg, tile);
  RN_UNSUPPORTED((double)tiles * (cin > cout ? cin : cout) * 4.0 >= 2147483648.0, "winograd gn: a transform plane is >= 2 GiB");
  float* U = (float*)workspace;
  float* V = (float*)((char*)workspace + rn::align_up(u_bytes(p2, cin, cout), 256));
  float* Mb = (float*)((char*)V + rn::align_up(p2 * tiles * cin * 4, 256));
  hipStream_t st = (hipStream_t)stream;
  return tile == 2 ? run_gn_fwd<2>(segs, nseg, cin, cout, w, bias, gn, U, V, Mb, v_buf, urot_buf, st)
                   : run_gn_fwd<4>(segs, nseg, cin, cout, w, bias, gn, U, V, Mb, v_buf, urot_buf, st);
}

extern "C" int rn_conv3x3_winograd_gn_bwd(const rn_conv_seg* segs, int nseg, int cin, int cout, const float* w, float* dw, int accumulate,
                                          int tile, const rn_wino_gn_bwd* gn, void* workspace, size_t workspace_bytes, const float* v_buf,
                                          const float* urot_buf, rn_stream_t stream) {
  if (int e = check_fold(segs, nseg, cin, cout, tile)) return e;
  RN_CHECK_ARG(w && dw && workspace && gn, "winograd gn bwd: null pointer");
  for (int s = 0; s < nseg; ++s) RN_CHECK_ARG(segs[s].x && segs[s].dy && segs[s].dx, "winograd gn bwd: null tensor in segment %d", s);
  if (gn->in_rows) {
    RN_CHECK_ARG(gn->in_gamma && gn->in_beta && gn->in_g_rows_group && gn->in_g_rows_chan, "winograd gn bwd: null input-side pointer");
    RN_UNSUPPORTED(!fold_ok(cin, gn->in_groups), "winograd gn bwd: %d channels in %d groups cannot be folded", cin, gn->in_groups);
  }
  if (gn->out_rows) {
    RN_CHECK_ARG(gn->out_g_rows_group && gn->out_gamma, "winograd gn bwd: null output-side pointer");
    for (int s = 0; s < nseg; ++s) RN_CHECK_ARG(segs[s].y, "winograd gn bwd: the folded output side needs the raw conv output y (segment %d)", s);
    RN_UNSUPPORTED(!fold_ok(cout, gn->out_groups), "winograd gn bwd: %d channels in %d groups cannot be folded", cout, gn->out_groups);
  }
  const BwdLayout L = bwd_layout(segs, nseg, cin, cout, tile, v_buf == nullptr, urot_buf == nullptr);
  if (workspace_bytes < L.total) {
    rn::set_error("winograd gn bwd: workspace %zu < %zu bytes", workspace_bytes, L.total);
    return RN_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  return tile == 2 ? run_gn_bwd<2>(segs, nseg, cin, cout, w, dw, accumulate, gn, (char*)workspace, L, workspace_bytes, v_buf, urot_buf, st)
                   : run_gn_bwd<4>(segs, nseg, cin, cout, w, dw, accumulate, gn, (char*)workspace, L, workspace_bytes, v_buf, urot_buf, st);
}
