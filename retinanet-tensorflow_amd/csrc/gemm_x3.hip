// Batched fp32 GEMMs on the bf16 matrix cores with fp32-grade accuracy: the products of the Winograd convolutions (head
// towers, FPN merges, class / box output convs: 88 % of the network's multiply-adds, retinanet.py:37-62,85-106,118-221).
//
// gfx950 has no reduced-precision fast path for fp32 inputs (no xf32), and its exact fp32 MFMA (v_mfma_f32_32x32x2_f32) runs at
// 1/16 of the bf16 rate: the fp32 product kernels are MFMA-bound at 0.5 - 0.6 of that peak.  Here every fp32
// operand element x is split EXACTLY into three bf16 values by truncation,
//     x = x1 + x2 + x3,   x1 = top 8 significant bits of x, x2 = top 8 bits of x - x1, x3 = x - x1 - x2 (<= 8 bits: exact),
// on its way from global memory into LDS, and a product sum_k a_k b_k is evaluated as SIX bf16 MFMA products with fp32
// accumulation (v_mfma_f32_32x32x16_bf16: bf16 x bf16 is exact in fp32):
//     a1 b1 + a1 b2 + a2 b1 + a2 b2 + a1 b3 + a3 b1          (dropped: a2 b3 + a3 b2 + a3 b3 <= 2^-23 |a b|)
// i.e. every elementary product carries a relative error of at most ~2^-23 -- the size of ONE fp32 rounding, which the fp32 MFMA's
// fmaf chain commits per term as well.  6 x 32 cycles per 32x32x16 step against 8 x 64 for the fp32 instruction: 2.7 x less
// matrix-core time; the kernels become bound by their operand traffic and the split's VALU work instead.
// Storage stays fp32 everywhere (inputs, outputs, accumulators): this is a different EVALUATION of the same fp32 product, and it
// is tested against an fp64 reference to the same bar as the fp32 kernels (tests/test_gpu_x3.py) besides every parity test.
//
//   C[r1][r2] = sum_{k in [k0, k1)} Op1(r1, k) * Op2(r2, k)        one launch for `nbatch` matrices x `nsplit` ranges of k
// Each operand is either k-contiguous in memory (KC: element (r, k) at base[r * ld + k]) or k-strided (KS: base[k * ld + r]):
//   forward products       M_xi = V_xi U_xi         Op1 = V [T x Cin] KC,  Op2 = U [Cin x Cout]     KS
//   data-gradient products dV_xi = dM_xi Urot_xi^T  Op1 = dM          KC,  Op2 = Urot [Cin x Cout..] KC ([N][K] layout)
//   weight-gradient        dU_xi = V_xi^T dM_xi     Op1 = V  KS (k = tile index), Op2 = dM KS; k split over blocks, partial slabs
// Tile 64 x 64 x 32, 4 waves (32 x 32 each), LDS [row][k] bf16 per plane with 80-byte rows (conflict-free b128 fragment reads);
// the KS loader reads 8 consecutive k of one row as 8 coalesced dword loads and writes them as ONE b128 per plane -- the
// transpose happens in registers.
#include "conv_tiles.h"
#include "rn_common.h"

namespace {
using namespace rn_tiles;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2v __attribute__((ext_vector_type(2)));

constexpr int XM = 64, XN = 64, XK = 32, XT = 256;
constexpr int LDR = 40;               // halfs per LDS row: 32 k + 8 pad = 80 bytes
constexpr int PLANE = 64 * LDR;       // halfs per (operand, plane)

struct X3Op { const float* p; long bstride; int ld, rows; };
struct X3Args {
  X3Op a, b;
  float* c; long c_bstride, c_sstride; int ldc;
  int K, chunk, nsplit, nbatch, tiles_m, tiles_n;
};

// x = h1 + h2 + h3 exactly, each with <= 8 significant bits (fp32 bit patterns whose low 16 bits are zero)
__device__ __forceinline__ void split3(float x, unsigned& h1, unsigned& h2, unsigned& h3) {
  h1 = __float_as_uint(x) & 0xffff0000u;
  const float r1 = x - __uint_as_float(h1);
  h2 = __float_as_uint(r1) & 0xffff0000u;
  h3 = __float_as_uint(r1 - __uint_as_float(h2));
}
// two bf16 (the high halves of lo and hi) in one dword, lo in the low half (the lower k)
__device__ __forceinline__ unsigned pack_hi(unsigned lo, unsigned hi) { return __builtin_amdgcn_perm(hi, lo, 0x07060302u); }

// One operand tile [64 rows][32 k]: 8 fp32 values per thread.
//   KC: two float4 (4 consecutive k of rows t/8 and t/8 + 32);  KS: 8 dwords (8 consecutive k of row t % 64)
template <bool KS>
struct TileLoad {
  float v[8];
  __device__ __forceinline__ void load(const __amdgpu_buffer_rsrc_t rs, const X3Op& op, int row0, int k0, int k1, int t) {
    if (KS) {
      const int row = row0 + (t & 63), k = k0 + (t >> 6) * 8;
      const bool rok = row < op.rows;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        v[j] = Vec<1>::load(rs, (rok && k + j < k1) ? ((unsigned)(k + j) * (unsigned)op.ld + (unsigned)row) * 4u : OOB);
    } else {
      const int k = k0 + (t & 7) * 4;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = row0 + (t >> 3) + 32 * i;
        const float4 q = Vec<4>::load(rs, (row < op.rows && k < k1) ? ((unsigned)row * (unsigned)op.ld + (unsigned)k) * 4u : OOB);
        v[4 * i] = q.x; v[4 * i + 1] = q.y; v[4 * i + 2] = q.z; v[4 * i + 3] = q.w;
      }
    }
  }
  // the three planes of this thread's values -> LDS (`tile`: the operand's plane 0; planes PLANE halfs apart)
  __device__ __forceinline__ void store(unsigned short* tile, int t) const {
    unsigned h[3][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) split3(v[j], h[0][j], h[1][j], h[2][j]);
    if (KS) {
      unsigned short* dst = tile + (t & 63) * LDR + (t >> 6) * 8;
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        u32x4 w;
        w.x = pack_hi(h[p][0], h[p][1]); w.y = pack_hi(h[p][2], h[p][3]);
        w.z = pack_hi(h[p][4], h[p][5]); w.w = pack_hi(h[p][6], h[p][7]);
        *reinterpret_cast<u32x4*>(dst + p * PLANE) = w;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        unsigned short* dst = tile + ((t >> 3) + 32 * i) * LDR + (t & 7) * 4;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          u32x2v w;
          w.x = pack_hi(h[p][4 * i], h[p][4 * i + 1]); w.y = pack_hi(h[p][4 * i + 2], h[p][4 * i + 3]);
          *reinterpret_cast<u32x2v*>(dst + p * PLANE) = w;
        }
      }
    }
  }
};

template <bool A_KS, bool B_KS>
__global__ __launch_bounds__(XT) void gemm_x3_kernel(const X3Args a) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[6 * PLANE];
  unsigned short* At = lds;
  unsigned short* Bt = lds + 3 * PLANE;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;
  const int bid = rn::xcd_remap(blockIdx.x, gridDim.x);
  const int tile_n = bid % a.tiles_n;
  const int tile_m = (bid / a.tiles_n) % a.tiles_m;
  const int rest = bid / (a.tiles_n * a.tiles_m);
  const int batch = rest % a.nbatch, split = rest / a.nbatch;
  const int m0 = tile_m * XM, n0 = tile_n * XN;
  const int kbeg = split * a.chunk, kend = min(a.K, kbeg + a.chunk);
  const float* pa = a.a.p + (size_t)batch * a.a.bstride;
  const float* pb = a.b.p + (size_t)batch * a.b.bstride;
  // (the descriptor covers the whole batch matrix: rows x ld for KC, K x ld for KS)
  const __amdgpu_buffer_rsrc_t ra = make_rsrc(pa, (unsigned)(A_KS ? a.K : a.a.rows) * (unsigned)a.a.ld * 4u);
  const __amdgpu_buffer_rsrc_t rb = make_rsrc(pb, (unsigned)(B_KS ? a.K : a.b.rows) * (unsigned)a.b.ld * 4u);
  TileLoad<A_KS> la;
  TileLoad<B_KS> lb;
  f32x16 acc[1][1];
  zero_acc<1, 1>(acc);
  const int nk = (kend - kbeg + XK - 1) / XK;
  if (nk > 0) {
    la.load(ra, a.a, m0, kbeg, kend, t);
    lb.load(rb, a.b, n0, kbeg, kend, t);
  }
  const unsigned short* afr = At + (wm * 32 + r) * LDR + h * 8;
  const unsigned short* bfr = Bt + (wn * 32 + r) * LDR + h * 8;
  for (int it = 0; it < nk; ++it) {
    la.store(At, t);
    lb.store(Bt, t);
    __syncthreads();
    if (it + 1 < nk) {
      la.load(ra, a.a, m0, kbeg + (it + 1) * XK, kend, t);
      lb.load(rb, a.b, n0, kbeg + (it + 1) * XK, kend, t);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 fa[3], fb[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        fa[p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(afr + p * PLANE + s * 16));
        fb[p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(bfr + p * PLANE + s * 16));
      }
      // (the small terms first)
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[2], fb[0], acc[0][0], 0, 0, 0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[2], acc[0][0], 0, 0, 0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fb[1], acc[0][0], 0, 0, 0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fb[0], acc[0][0], 0, 0, 0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[1], acc[0][0], 0, 0, 0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[0], acc[0][0], 0, 0, 0);
    }
    __syncthreads();
  }
  float* pc = a.c + (size_t)split * a.c_sstride + (size_t)batch * a.c_bstride;
  store_tile<XM, XN, 2, 2>(acc, pc, nullptr, m0, n0, a.a.rows, a.b.rows, a.ldc, wm, wn, lane);
}

int g_mode = -1;
int mode() {
  if (g_mode < 0) {
    const char* e = getenv("RN_PROD_X3");
    g_mode = e ? (atoi(e) != 0) : 0;
  }
  return g_mode;
}

bool fits(long elems) { return elems > 0 && (double)elems * 4.0 < 2147483648.0; }

int launch(const X3Args& a, bool a_ks, bool b_ks, hipStream_t st) {
  const long blocks = (long)a.nsplit * a.nbatch * a.tiles_m * a.tiles_n;
  RN_UNSUPPORTED(blocks <= 0 || blocks > 0x7fffffffL, "gemm x3: %ld blocks", blocks);
  const dim3 grid((unsigned)blocks);
  if (a_ks && b_ks) hipLaunchKernelGGL((gemm_x3_kernel<true, true>), grid, dim3(XT), 0, st, a);
  else if (!a_ks && b_ks) hipLaunchKernelGGL((gemm_x3_kernel<false, true>), grid, dim3(XT), 0, st, a);
  else if (!a_ks && !b_ks) hipLaunchKernelGGL((gemm_x3_kernel<false, false>), grid, dim3(XT), 0, st, a);
  else hipLaunchKernelGGL((gemm_x3_kernel<true, false>), grid, dim3(XT), 0, st, a);
  RN_LAUNCH_CHECK();
  return RN_OK;
}

// k-ranges of the weight-gradient product: enough blocks to fill the chip, chunks of whole K-tiles
void tn_split(int M, int K, int N, int nbatch, int* nsplit, int* chunk) {
  const long tiles = (long)nbatch * rn::ceil_div(K, XM) * rn::ceil_div(N, XN);
  int ns = (int)rn::ceil_div64(1024, tiles > 0 ? tiles : 1);
  const int kt = rn::ceil_div(M, XK);
  if (ns > kt) ns = kt;
  if (ns > 16) ns = 16;
  if (ns < 1) ns = 1;
  const int ck = rn::ceil_div(kt, ns) * XK;
  *chunk = ck;
  *nsplit = rn::ceil_div(M, ck);
}
}  // namespace

namespace rn {
// 0: the exact fp32 MFMA kernels of conv_gemm.hip; 1: the split-bf16 kernels of this file (where the shape allows)
int product_mode() { return mode(); }
void set_product_mode(int m) { g_mode = m ? 1 : 0; }

bool gemm_x3_ok(int M, int K, int N) {
  return M >= 1 && K >= 4 && N >= 4 && K % 4 == 0 && N % 4 == 0 && fits((long)M * K) && fits((long)K * N) && fits((long)M * N);
}

int launch_batched_gemm_x3(const float* A, const float* B, float* C, int M, int K, int N, int nbatch, int b_nk, hipStream_t st) {
  X3Args a = {};
  a.a = {A, (long)M * K, K, M};
  a.b = b_nk ? X3Op{B, (long)K * N, K, N} : X3Op{B, (long)K * N, N, N};
  a.c = C; a.c_bstride = (long)M * N; a.c_sstride = 0; a.ldc = N;
  a.K = K; a.chunk = rn::ceil_div(K, XK) * XK; a.nsplit = 1; a.nbatch = nbatch;
  a.tiles_m = rn::ceil_div(M, XM); a.tiles_n = rn::ceil_div(N, XN);
  return launch(a, false, b_nk == 0, st);
}

size_t batched_gemm_tn_workspace_x3(int M, int K, int N, int nbatch) {
  int ns, ck;
  tn_split(M, K, N, nbatch, &ns, &ck);
  return (size_t)ns * nbatch * K * N * sizeof(float);
}

// slabs [nsplit][nbatch][K][N] of A_b^T B_b (A_b [M x K], B_b [M x N]) in `workspace`
int launch_batched_gemm_tn_x3(const float* A, const float* B, int M, int K, int N, int nbatch, void* workspace, size_t workspace_bytes,
                              hipStream_t st, int* nsplit_out) {
  int ns, ck;
  tn_split(M, K, N, nbatch, &ns, &ck);
  const size_t need = (size_t)ns * nbatch * K * N * sizeof(float);
  if (workspace_bytes < need) {
    rn::set_error("gemm x3 tn: workspace %zu < %zu bytes", workspace_bytes, need);
    return RN_EWORKSPACE;
  }
  X3Args a = {};
  a.a = {A, (long)M * K, K, K};          // rows = the K output rows, ld = K, contraction index = the M rows of A
  a.b = {B, (long)M * N, N, N};
  a.c = (float*)workspace; a.c_bstride = (long)K * N; a.c_sstride = (long)nbatch * K * N; a.ldc = N;
  a.K = M; a.chunk = ck; a.nsplit = ns; a.nbatch = nbatch;
  a.tiles_m = rn::ceil_div(K, XM); a.tiles_n = rn::ceil_div(N, XN);
  *nsplit_out = ns;
  return launch(a, true, true, st);
}
}  // namespace rn

extern "C" int rn_set_product_mode(int mode_) {
  rn::set_product_mode(mode_);
  return RN_OK;
}
extern "C" int rn_get_product_mode(void) { return rn::product_mode(); }
