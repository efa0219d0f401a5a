// Grouped 3x3 / stride-1 convolutions with few channels per group: the 32 parallel 3x3 convs of a ResNeXt bottleneck
// (resnet.py:36-49: `tf.split(input, 32, -1)`, one Conv2D per piece, `tf.concat`), 4 / 8 / 16 / 32 channels per group, fp32,
// forward + data gradient + weight gradient (round 6).
//
// As an implicit GEMM (conv_gemm.hip) each group is its own product with N = channels per group: at 4 channels a 32-column
// matrix-core tile is 7/8 padding and every block gathers 16-byte pieces of its group's input -- measured at the cfg-3 shape
// (800^2, batch 2): forward 105 us, data gradient 135 us, weight gradient 141 us per conv on average, for 82 MB of tensors
// (16 us at 5 TB/s) and 0.74 GFLOP.  The arithmetic is small enough for the vector unit, so these kernels are direct
// convolutions shaped for the memory system instead:
//   block = (sample, 8 x 16 tile of output pixels, slab of 32 channels = 32 / CG whole groups)
//   LDS   = the 10 x 18 x 32 input patch, read ONCE from HBM as whole 128-byte pixel rows (pixel stride 40 floats: the b128 reads
//           of a 16-lane group fall on distinct banks) + the slab's kernel [tap][ci][32 co]
//   thread = 4 consecutive pixels x 4 consecutive output channels (16 accumulators): per (kernel row, 4 input channels) it reads
//           6 input float4 + 12 kernel float4 from LDS for 192 fused multiply-adds
// The data gradient is the same kernel on dy with the kernel rotated and transposed inside each group while it is copied to LDS.
// The weight gradient: a block walks `tpb` tiles of one sample and slab, keeps x and dy patches in LDS, a thread owns (tap, 4 input
// channels, 4 output channels) = 16 accumulators over a share of the pixels; partial sums [row][9][CG][C] leave through the
// fixed-order row reduction (rn::launch_reduce_rows) like every other weight gradient of the library.
#include "rn_common.h"

namespace {
constexpr int GT = 256;                  // threads per block
constexpr int TH = 8, TW = 16;           // output tile
constexpr int PH = TH + 2, PW = TW + 2;  // input patch
constexpr int SLAB = 32;                 // channels per block
constexpr int PS = 40;                   // floats per patch pixel in LDS (32 + 8 pad)

struct GcArgs {
  const float* x; const float* w; float* y;
  int n, h, wd, c;                       // NHWC, c = cin = cout
  int tiles_h, tiles_w, nslab;
  int transpose;                         // data gradient: kernel rotated by 180 degrees, ci <-> co inside each group
};

// the slab's kernel -> LDS [tap][ci][32 co]; global layout HWIO [3][3][CG][C]
template <int CG>
__device__ __forceinline__ void load_kernel(float* wl, const float* __restrict__ w, int c, int c0, int transpose, int tid) {
  if (!transpose) {
    for (int e = tid; e < 9 * CG * (SLAB / 4); e += GT) {
      const int row = e / (SLAB / 4), q = e - row * (SLAB / 4);           // row = tap * CG + ci
      *reinterpret_cast<float4*>(&wl[row * SLAB + q * 4]) = *reinterpret_cast<const float4*>(w + (size_t)row * c + c0 + q * 4);
    }
  } else {
    // w'[tap][a][g CG + b] = w[8 - tap][b][g CG + a]   (a: dy channel inside the group, b: dx channel inside the group)
    for (int e = tid; e < 9 * CG * SLAB; e += GT) {
      const int col = e % SLAB, row = e / SLAB;
      const int tap = row / CG, a = row - tap * CG;
      const int g = col / CG, b = col - g * CG;
      wl[row * SLAB + col] = w[(size_t)((8 - tap) * CG + b) * c + c0 + g * CG + a];
    }
  }
}

// rows [y0, y0 + PH) x cols [x0, x0 + PW) of sample `s`, channels [c0, c0 + 32) -> LDS [pixel][PS], zeros outside the map
__device__ __forceinline__ void load_patch(float* pl, const float* __restrict__ x, int s, int h, int wd, int c, int c0, int y0, int x0, int tid) {
  for (int e = tid; e < PH * PW * (SLAB / 4); e += GT) {
    const int pix = e / (SLAB / 4), q = e - pix * (SLAB / 4);
    const int r = pix / PW, cc = pix - r * PW;
    const int iy = y0 + r, ix = x0 + cc;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((unsigned)iy < (unsigned)h && (unsigned)ix < (unsigned)wd)
      v = *reinterpret_cast<const float4*>(x + ((size_t)(s * h + iy) * wd + ix) * c + c0 + q * 4);
    *reinterpret_cast<float4*>(&pl[pix * PS + q * 4]) = v;
  }
}

template <int CG>
__global__ __launch_bounds__(GT, 2) void gconv3x3_kernel(const GcArgs a) {
  extern __shared__ __attribute__((aligned(16))) float gsm[];
  float* pl = gsm;                              // [PH * PW][PS]
  float* wl = gsm + PH * PW * PS;               // [9 * CG][32]
  const int tid = threadIdx.x;
  const int bid = rn::xcd_remap(blockIdx.x, gridDim.x);
  const int slab = bid % a.nslab;
  int t = bid / a.nslab;
  const int tx = t % a.tiles_w; t /= a.tiles_w;
  const int ty = t % a.tiles_h;
  const int s = t / a.tiles_h;
  const int c0 = slab * SLAB, oy0 = ty * TH, ox0 = tx * TW;
  load_patch(pl, a.x, s, a.h, a.wd, a.c, c0, oy0 - 1, ox0 - 1, tid);
  load_kernel<CG>(wl, a.w, a.c, c0, a.transpose, tid);
  __syncthreads();
  const int q = tid & 7, pq = tid >> 3;         // output-channel quad of the slab; pixel quad of the tile
  const int py = pq >> 2, px4 = (pq & 3) * 4;
  const int gbase = (q * 4) / CG * CG;          // first input channel (inside the slab) of this quad's group
  float4 acc[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) acc[p] = make_float4(0.f, 0.f, 0.f, 0.f);
  // (neither loop unrolled: a step is 18 b128 LDS reads + 192 FMAs on 72 + 16 registers; unrolled, the compiler hoists every
  // read of the block in front of the arithmetic and spills -- 250 registers at CG = 4, 6.8 KB of scratch at CG = 32)
#pragma unroll 1
  for (int kh = 0; kh < 3; ++kh) {
    const float* prow = pl + ((py + kh) * PW + px4) * PS + gbase;
#pragma unroll 1
    for (int c4 = 0; c4 < CG / 4; ++c4) {
      float4 in[6];
#pragma unroll
      for (int cc = 0; cc < 6; ++cc) in[cc] = *reinterpret_cast<const float4*>(prow + cc * PS + c4 * 4);
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const float* wrow = wl + ((kh * 3 + kw) * CG + c4 * 4) * SLAB + q * 4;
        const float4 w0 = *reinterpret_cast<const float4*>(wrow), w1 = *reinterpret_cast<const float4*>(wrow + SLAB);
        const float4 w2 = *reinterpret_cast<const float4*>(wrow + 2 * SLAB), w3 = *reinterpret_cast<const float4*>(wrow + 3 * SLAB);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const float4 v = in[p + kw];
          acc[p].x = fmaf(v.x, w0.x, acc[p].x); acc[p].y = fmaf(v.x, w0.y, acc[p].y); acc[p].z = fmaf(v.x, w0.z, acc[p].z); acc[p].w = fmaf(v.x, w0.w, acc[p].w);
          acc[p].x = fmaf(v.y, w1.x, acc[p].x); acc[p].y = fmaf(v.y, w1.y, acc[p].y); acc[p].z = fmaf(v.y, w1.z, acc[p].z); acc[p].w = fmaf(v.y, w1.w, acc[p].w);
          acc[p].x = fmaf(v.z, w2.x, acc[p].x); acc[p].y = fmaf(v.z, w2.y, acc[p].y); acc[p].z = fmaf(v.z, w2.z, acc[p].z); acc[p].w = fmaf(v.z, w2.w, acc[p].w);
          acc[p].x = fmaf(v.w, w3.x, acc[p].x); acc[p].y = fmaf(v.w, w3.y, acc[p].y); acc[p].z = fmaf(v.w, w3.z, acc[p].z); acc[p].w = fmaf(v.w, w3.w, acc[p].w);
        }
      }
    }
  }
  const int oy = oy0 + py;
  if (oy < a.h) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int ox = ox0 + px4 + p;
      if (ox < a.wd) *reinterpret_cast<float4*>(a.y + ((size_t)(s * a.h + oy) * a.wd + ox) * a.c + c0 + q * 4) = acc[p];
    }
  }
}

// ---- weight gradient
struct GwArgs {
  const float* x; const float* dy; float* partial;    // partial [n * nblk][9][CG][C]
  int n, h, wd, c;
  int tiles_h, tiles_w, nslab, tpb, nblk;              // tiles per block, blocks per (sample, slab)
};

template <int CG>
__global__ __launch_bounds__(GT, 2) void gconv3x3_wgrad_kernel(const GwArgs a) {
  // work items: (tap, input-channel quad of the group, output-channel quad of the slab) = 9 * (CG / 4) * 8; with fewer items than
  // threads (CG = 4: 72, CG = 8: 144) the tile's pixel rows are shared out among PP copies of the items
  constexpr int ITEMS = 9 * (CG / 4) * 8;
  constexpr int PP = ITEMS >= GT ? 1 : GT / ITEMS;     // 3, 1, 1, 1
  constexpr int UNITS = ITEMS * PP;
  constexpr int UPT = (UNITS + GT - 1) / GT;           // units per thread: 1, 1, 2, 3
  extern __shared__ __attribute__((aligned(16))) float gsm[];
  float* xl = gsm;                                     // x patch [PH * PW][PS]
  float* dl = gsm + PH * PW * PS;                      // dy tile [TH * TW][PS]
  const int tid = threadIdx.x;
  const int bid = rn::xcd_remap(blockIdx.x, gridDim.x);
  const int slab = bid % a.nslab;
  const int rowid = bid / a.nslab;                     // sample * nblk + blk
  const int blk = rowid % a.nblk, s = rowid / a.nblk;
  const int c0 = slab * SLAB;
  const int ntile = a.tiles_h * a.tiles_w;
  const int t_lo = blk * a.tpb, t_hi = min(t_lo + a.tpb, ntile);
  float4 acc[UPT][4];
#pragma unroll
  for (int u = 0; u < UPT; ++u)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[u][j] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int tile = t_lo; tile < t_hi; ++tile) {
    const int ty = tile / a.tiles_w, tx = tile - ty * a.tiles_w;
    const int oy0 = ty * TH, ox0 = tx * TW;
    __syncthreads();                                   // the previous tile's patches are dead
    load_patch(xl, a.x, s, a.h, a.wd, a.c, c0, oy0 - 1, ox0 - 1, tid);
    for (int e = tid; e < TH * TW * (SLAB / 4); e += GT) {
      const int pix = e / (SLAB / 4), q = e - pix * (SLAB / 4);
      const int r = pix / TW, cc = pix - r * TW;
      const int oy = oy0 + r, ox = ox0 + cc;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (oy < a.h && ox < a.wd) v = *reinterpret_cast<const float4*>(a.dy + ((size_t)(s * a.h + oy) * a.wd + ox) * a.c + c0 + q * 4);
      *reinterpret_cast<float4*>(&dl[pix * PS + q * 4]) = v;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < UPT; ++u) {
      const int unit = tid + u * GT;
      if (unit < UNITS) {
        const int part = unit / ITEMS, item = unit - part * ITEMS;
        const int q = item & 7, r = item >> 3;         // r = tap * (CG / 4) + ci quad
        const int tap = r / (CG / 4), c4 = r - tap * (CG / 4);
        const int kh = tap / 3, kw = tap - kh * 3;
        const int gbase = (q * 4) / CG * CG;
        // this unit's share of the tile's rows: part, part + PP, ...
        for (int py = part; py < TH; py += PP) {
          const float* xr = xl + ((py + kh) * PW + kw) * PS + gbase + c4 * 4;
          const float* dr = dl + (py * TW) * PS + q * 4;
#pragma unroll 4
          for (int px = 0; px < TW; ++px) {
            const float4 xv = *reinterpret_cast<const float4*>(xr + px * PS);
            const float4 dv = *reinterpret_cast<const float4*>(dr + px * PS);
            acc[u][0].x = fmaf(xv.x, dv.x, acc[u][0].x); acc[u][0].y = fmaf(xv.x, dv.y, acc[u][0].y); acc[u][0].z = fmaf(xv.x, dv.z, acc[u][0].z); acc[u][0].w = fmaf(xv.x, dv.w, acc[u][0].w);
            acc[u][1].x = fmaf(xv.y, dv.x, acc[u][1].x); acc[u][1].y = fmaf(xv.y, dv.y, acc[u][1].y); acc[u][1].z = fmaf(xv.y, dv.z, acc[u][1].z); acc[u][1].w = fmaf(xv.y, dv.w, acc[u][1].w);
            acc[u][2].x = fmaf(xv.z, dv.x, acc[u][2].x); acc[u][2].y = fmaf(xv.z, dv.y, acc[u][2].y); acc[u][2].z = fmaf(xv.z, dv.z, acc[u][2].z); acc[u][2].w = fmaf(xv.z, dv.w, acc[u][2].w);
            acc[u][3].x = fmaf(xv.w, dv.x, acc[u][3].x); acc[u][3].y = fmaf(xv.w, dv.y, acc[u][3].y); acc[u][3].z = fmaf(xv.w, dv.z, acc[u][3].z); acc[u][3].w = fmaf(xv.w, dv.w, acc[u][3].w);
          }
        }
      }
    }
  }
  // the PP copies of an item are summed through LDS in a fixed order (part 0 + part 1 + part 2), then one row of the partial tensor
  __syncthreads();
  float* red = gsm;                                    // [PP][ITEMS][16]
  if (PP > 1) {
#pragma unroll
    for (int u = 0; u < UPT; ++u) {
      const int unit = tid + u * GT;
      if (unit < UNITS) {
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<float4*>(&red[(size_t)unit * 16 + j * 4]) = acc[u][j];
      }
    }
    __syncthreads();
  }
  float* out = a.partial + (size_t)rowid * 9 * CG * a.c;
#pragma unroll
  for (int u = 0; u < UPT; ++u) {
    const int unit = tid + u * GT;
    if (unit < ITEMS) {                                // (part 0 of every item writes)
      const int q = unit & 7, r = unit >> 3;
      const int tap = r / (CG / 4), c4 = r - tap * (CG / 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float4 v = acc[u][j];
        if (PP > 1) {
          v = *reinterpret_cast<const float4*>(&red[(size_t)unit * 16 + j * 4]);
#pragma unroll
          for (int p = 1; p < PP; ++p) {
            const float4 o = *reinterpret_cast<const float4*>(&red[(size_t)(p * ITEMS + unit) * 16 + j * 4]);
            v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
          }
        }
        *reinterpret_cast<float4*>(out + (size_t)(tap * CG + c4 * 4 + j) * a.c + c0 + q * 4) = v;
      }
    }
  }
}

constexpr size_t fwd_lds(int cg) { return (size_t)(PH * PW * PS + 9 * cg * SLAB) * 4; }
constexpr size_t wgrad_lds() { return (size_t)(PH * PW * PS + TH * TW * PS) * 4; }     // (>= 3 * 72 * 16 floats of the final exchange)

const bool g_on = !(getenv("RN_GCONV_DIRECT") && atoi(getenv("RN_GCONV_DIRECT")) == 0);

void wgrad_plan(int n, int h, int wd, int* tiles_h, int* tiles_w, int* tpb, int* nblk) {
  *tiles_h = rn::ceil_div(h, TH); *tiles_w = rn::ceil_div(wd, TW);
  const int ntile = *tiles_h * *tiles_w;
  int per = 1;
  while ((long)n * rn::ceil_div(ntile, per) > 128) ++per;       // <= 128 partial rows: the row reduction reads rows x |dW|
  *tpb = per; *nblk = rn::ceil_div(ntile, per);
}
}  // namespace

namespace rn {
// channels per group for which the direct kernels are built (cin_g == cout_g), maps large enough to fill the chip
bool gconv3x3_ok(int n, int h, int wd, int c, int groups) {
  if (!g_on || groups < 2 || c % groups || c % SLAB) return false;
  const int cg = c / groups;
  if (cg != 4 && cg != 8 && cg != 16 && cg != 32) return false;
  if ((double)n * h * wd * c >= 536870912.0) return false;
  return (long)n * rn::ceil_div(h, TH) * rn::ceil_div(wd, TW) * (c / SLAB) >= 256;
}

int launch_gconv3x3(const float* x, const float* w, float* y, int n, int h, int wd, int c, int groups, int transpose, hipStream_t st) {
  GcArgs a = {x, w, y, n, h, wd, c, rn::ceil_div(h, TH), rn::ceil_div(wd, TW), c / SLAB, transpose};
  const dim3 grid((unsigned)((long)n * a.tiles_h * a.tiles_w * a.nslab));
  const int cg = c / groups;
#define RN_GC(CG_)                                                                                                           \
  do {                                                                                                                       \
    static const bool attr_ = hipFuncSetAttribute((const void*)gconv3x3_kernel<CG_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fwd_lds(CG_)) == hipSuccess; \
    RN_UNSUPPORTED(!attr_, "grouped conv: %zu bytes of LDS per block refused", fwd_lds(CG_));                                \
    hipLaunchKernelGGL((gconv3x3_kernel<CG_>), grid, dim3(GT), fwd_lds(CG_), st, a);                                         \
  } while (0)
  if (cg == 4) RN_GC(4);
  else if (cg == 8) RN_GC(8);
  else if (cg == 16) RN_GC(16);
  else RN_GC(32);
#undef RN_GC
  RN_LAUNCH_CHECK();
  return RN_OK;
}

size_t gconv3x3_wgrad_workspace(int n, int h, int wd, int c, int groups) {
  int th, tw, tpb, nblk;
  wgrad_plan(n, h, wd, &th, &tw, &tpb, &nblk);
  return (size_t)n * nblk * 9 * (c / groups) * c * sizeof(float);
}

// dw [3][3][CG][C] (accumulate != 0: added to) from x and dy [n][h][wd][c]
int launch_gconv3x3_wgrad(const float* x, const float* dy, float* dw, int accumulate, int n, int h, int wd, int c, int groups, void* workspace,
                          size_t workspace_bytes, hipStream_t st) {
  GwArgs a = {};
  a.x = x; a.dy = dy; a.partial = (float*)workspace; a.n = n; a.h = h; a.wd = wd; a.c = c; a.nslab = c / SLAB;
  wgrad_plan(n, h, wd, &a.tiles_h, &a.tiles_w, &a.tpb, &a.nblk);
  const int cg = c / groups;
  const size_t need = (size_t)n * a.nblk * 9 * cg * c * sizeof(float);
  if (workspace_bytes < need) {
    rn::set_error("grouped conv wgrad: workspace %zu < %zu bytes", workspace_bytes, need);
    return RN_EWORKSPACE;
  }
  const dim3 grid((unsigned)((long)n * a.nblk * a.nslab));
#define RN_GW(CG_)                                                                                                           \
  do {                                                                                                                       \
    static const bool attr_ = hipFuncSetAttribute((const void*)gconv3x3_wgrad_kernel<CG_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wgrad_lds()) == hipSuccess; \
    RN_UNSUPPORTED(!attr_, "grouped conv wgrad: %zu bytes of LDS per block refused", wgrad_lds());                           \
    hipLaunchKernelGGL((gconv3x3_wgrad_kernel<CG_>), grid, dim3(GT), wgrad_lds(), st, a);                                    \
  } while (0)
  if (cg == 4) RN_GW(4);
  else if (cg == 8) RN_GW(8);
  else if (cg == 16) RN_GW(16);
  else RN_GW(32);
#undef RN_GW
  RN_LAUNCH_CHECK();
  return rn::launch_reduce_rows((const float*)workspace, dw, (int64_t)9 * cg * c, n * a.nblk, accumulate, st);
}
}  // namespace rn
