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
// Tile 128 x 128 x 32 (64 x 64 x 32 for small launches), 4 waves, LDS [row][k] bf16 per plane with 80-byte rows (conflict-free b128
// fragment reads);
// a KS tile is kept [k pair][row] in LDS (see TileLoad): float4 loads along the rows, one b128 store per plane, fragments as
// four b32 reads -- the transpose costs nothing.
#include "conv_tiles.h"
#include "rn_common.h"

namespace {
using namespace rn_tiles;
#include "x3_tiles.h"

// Block tile (64 WT) x (64 WT) x 32, 4 waves as 2 x 2, each wave WT x WT MFMA tiles of 32 x 32.  WT = 2 (128 x 128) halves
// the LDS traffic and the split's VALU work per matrix-core instruction (the 64 x 64 tile is bound by their SUM: 28 us for the
// head-tower product against 8 us of matrix-core time); WT = 1 is for launches too small to fill the chip with 128 x 128 tiles.
// NST register stages: the operand tiles of NST K-steps are in flight at once.
// DBG: the leave-one-out timing aid (RN_X3_DBG, wrong results) as its own instantiation -- in the production kernel (DBG = false)
// the K-step is ONE basic block: run-time tests between its phases would keep the scheduler from moving the next step's fragment
// reads and the global loads in between the matrix-core instructions.
template <bool A_KS, bool B_KS, int WT, int NST, bool DBG = false>
__global__ __launch_bounds__(XT, WT == 2 ? 2 : 4) void gemm_x3_kernel(const X3Args a) {
  const int dbg = DBG ? a.dbg : 0;
  constexpr int ROWS = 64 * WT;
  typedef TileGeom<ROWS> G;
  __shared__ __attribute__((aligned(16))) unsigned short lds[6 * G::PLANE];
  unsigned short* At = lds;
  unsigned short* Bt = lds + 3 * G::PLANE;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;
  const int bid = rn::xcd_remap(blockIdx.x, gridDim.x);
  const int tile_n = bid % a.tiles_n;
  const int tile_m = (bid / a.tiles_n) % a.tiles_m;
  const int rest = bid / (a.tiles_n * a.tiles_m);
  const int batch = rest % a.nbatch, split = rest / a.nbatch;
  const int m0 = tile_m * ROWS, n0 = tile_n * ROWS;
  const int kbeg = split * a.chunk, kend = min(a.K, kbeg + a.chunk);
  const float* pa = a.a.p + (size_t)batch * a.a.bstride;
  const float* pb = a.b.p + (size_t)batch * a.b.bstride;
  // (the descriptor covers the whole batch matrix: rows x ld for KC, K x ld for KS)
  const __amdgpu_buffer_rsrc_t ra = make_rsrc(pa, (unsigned)(A_KS ? a.K : a.a.rows) * (unsigned)a.a.ld * 4u);
  const __amdgpu_buffer_rsrc_t rb = make_rsrc(pb, (unsigned)(B_KS ? a.K : a.b.rows) * (unsigned)a.b.ld * 4u);
  TileLoad<A_KS, ROWS> la[NST];
  TileLoad<B_KS, ROWS> lb[NST];
  f32x16 acc[WT][WT];
  zero_acc<WT, WT>(acc);
  const int nk = (kend - kbeg + XK - 1) / XK;
#pragma unroll
  for (int st = 0; st < NST; ++st)
    if (st < nk) {
      la[st].load(ra, a.a, m0, kbeg + st * XK, kend, t);
      lb[st].load(rb, a.b, n0, kbeg + st * XK, kend, t);
    }
  for (int it0 = 0; it0 < nk; it0 += NST) {
#pragma unroll
    for (int st = 0; st < NST; ++st) {
      const int it = it0 + st;
      if (it < nk) {                         // (block-uniform)
        if (!(dbg & 2)) {
          la[st].store(At, t);
          lb[st].store(Bt, t);
        }
        __syncthreads();
        if (it + NST < nk && !(dbg & 4)) {
          la[st].load(ra, a.a, m0, kbeg + (it + NST) * XK, kend, t);
          lb[st].load(rb, a.b, n0, kbeg + (it + NST) * XK, kend, t);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          bf16x8 fa[WT][3], fb[WT][3];
          if (!(dbg & 8)) {
#pragma unroll
            for (int i = 0; i < WT; ++i)
#pragma unroll
              for (int p = 0; p < 3; ++p) {
                fa[i][p] = fragment<A_KS, ROWS>(At, p, (wm * WT + i) * 32 + r, h, s);
                fb[i][p] = fragment<B_KS, ROWS>(Bt, p, (wn * WT + i) * 32 + r, h, s);
              }
          } else {
#pragma unroll
            for (int i = 0; i < WT; ++i)
#pragma unroll
              for (int p = 0; p < 3; ++p) { fa[i][p] = __builtin_bit_cast(bf16x8, u32x4{(unsigned)t, 1u, 2u, 3u}); fb[i][p] = fa[i][p]; }
          }
          if (!(dbg & 1))
#pragma unroll
          for (int i = 0; i < WT; ++i)
#pragma unroll
            for (int j = 0; j < WT; ++j) {   // (the small terms first)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][2], fb[j][0], acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][2], acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][1], acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][0], acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][1], acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][0], acc[i][j], 0, 0, 0);
            }
        }
        __syncthreads();
      }
    }
  }
  float* pc = a.c + (size_t)split * a.c_sstride + (size_t)batch * a.c_bstride;
  if (a.drop_rate > 0.f) {                                   // (block-uniform; only the 1x1-conv entry points set it)
    const uint64_t sd = a.drop_seed + (a.drop_seed_dev ? *a.drop_seed_dev : 0ull);
#pragma unroll
    for (int i = 0; i < WT; ++i)
#pragma unroll
      for (int j = 0; j < WT; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int row = m0 + (wm * WT + i) * 32 + (q & 3) + 8 * (q >> 2) + 4 * h, col = n0 + (wn * WT + j) * 32 + r;
          acc[i][j][q] = rn::uniform01(sd, (uint64_t)row * (uint64_t)a.ldc + (uint64_t)col) >= a.drop_rate ? acc[i][j][q] * a.drop_keep : 0.f;
        }
  }
  if (WT == 2 && !(dbg & 16)) {
    // 16-byte stores through LDS: the accumulator layout (lane = column, registers = rows) would go out as 64 dword stores per
    // lane in 128-byte pieces; staged per wave as [32 rows][64 columns] it leaves as 8 x 16-byte stores per lane and 32-row half,
    // every instruction writing four whole 256-byte row segments.  (The operand tiles are dead: the K loop ended with a barrier.)
    float* stage = reinterpret_cast<float*>(lds) + wave * (32 * 68);           // 68-float rows: 16-byte aligned, rows 4 banks apart
    const __amdgpu_buffer_rsrc_t rc = make_rsrc(pc, (unsigned)a.a.rows * (unsigned)a.ldc * 4u);
    const int col0 = n0 + wn * 64;
#pragma unroll
    for (int i = 0; i < WT; ++i) {
#pragma unroll
      for (int j = 0; j < WT; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) stage[((q & 3) + 8 * (q >> 2) + 4 * h) * 68 + j * 32 + r] = acc[i][j][q];
      // (one wave writes and reads its own region: no block barrier, the wave's LDS operations complete in order)
      const int row_base = m0 + (wm * WT + i) * 32;
#pragma unroll
      for (int v = 0; v < 8; ++v) {
        const int rr = v * 4 + (lane >> 4), cc = (lane & 15) * 4;
        const float4 o = *reinterpret_cast<const float4*>(&stage[rr * 68 + cc]);
        const int row = row_base + rr, col = col0 + cc;
        const unsigned voff = (row < a.a.rows && col < a.b.rows) ? ((unsigned)row * (unsigned)a.ldc + (unsigned)col) * 4u : OOB;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rc, voff, 0, 0);
      }
    }
  } else {
    store_tile<ROWS, ROWS, 2, 2>(acc, pc, nullptr, m0, n0, a.a.rows, a.b.rows, a.ldc, wm, wn, lane);
  }
  if (a.stat) {      // (block-uniform) GroupNorm partial sums of the tile: rows past the end were multiplied as zeros
    __syncthreads();                                         // the staging area / operand tiles are dead
    float* red = reinterpret_cast<float*>(lds);              // [2 wm][ROWS columns][2]
#pragma unroll
    for (int j = 0; j < WT; ++j) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < WT; ++i)
#pragma unroll
        for (int q = 0; q < 16; ++q) { const float v = acc[i][j][q]; s1 += v; s2 = fmaf(v, v, s2); }
      s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 32, 64);
      if (lane < 32) {
        const int col = (wn * WT + j) * 32 + r;
        red[(wm * ROWS + col) * 2 + 0] = s1;
        red[(wm * ROWS + col) * 2 + 1] = s2;
      }
    }
    __syncthreads();
    if (t < ROWS && n0 + t < a.b.rows)
      a.stat[(size_t)tile_m * a.ldc + n0 + t] = make_float2(red[t * 2] + red[(ROWS + t) * 2], red[t * 2 + 1] + red[(ROWS + t) * 2 + 1]);
  }
}

// Measured and not kept (round 5): the same tiles as a "ping-pong" block of 512 threads -- two groups of 4 waves, each with its
// own 128 x 128 tile and LDS image, held half a K-step apart by the block's barriers so that every SIMD always has one wave on the
// matrix cores and one on the vector unit -- 27.1 us against 27.6 us for the head-tower product: the pieces of this kernel add up
// (leave-one-out, RN_X3_DBG: 10.6 us launch + epilogue, + 8.3 matrix cores, + 4.5 split and LDS stores, + 3.1 fragment reads,
// + 1.4 global loads) whatever the interleaving, as they would under a power-limited clock: what shortens it is less WORK
// (operands split once by their producers instead of once per consuming block), not a different schedule.

// Measured and not kept (round 6): the K loop software-pipelined inside every wave -- the split of tile it + 1 (registers -> packed
// registers) issued in the matrix-core phase of tile it, in the shadow of its MFMAs (the compiler does interleave them: 5 - 8 VALU
// per MFMA), only the 24 LDS stores of the packed values left between the barriers, no branch in the step: 28.0 us against 26.8 us
// for the head-tower product, 54.5 against 52.1 for the backward pair, the step unchanged (499.9 vs 499.9 images/s).  What DID
// help (round 6): the leave-one-out switches (RN_X3_DBG) are a template parameter now -- as run-time tests they split the K-step
// into basic blocks the scheduler could not move fragment reads and global loads across.
int g_mode = -1;
int mode() {
  if (g_mode < 0) {
    const char* e = getenv("RN_PROD_X3");
    g_mode = e ? (atoi(e) != 0) : 1;
  }
  return g_mode;
}

bool fits(long elems) { return elems > 0 && (double)elems * 4.0 < 2147483648.0; }

// tiles of 128 x 128 where they still give every CU work (>= 1.25 blocks per CU), 64 x 64 otherwise
int tile_rows(long rows_a, long rows_b, long batches) {
  static const int forced = getenv("RN_X3_TILE") ? atoi(getenv("RN_X3_TILE")) : 0;      // tuning aid: 64 / 128
  if (forced == 64 || forced == 128) return forced;
  const long big = batches * rn::ceil_div64(rows_a, 128) * rn::ceil_div64(rows_b, 128);
  return big >= 320 ? 128 : 64;
}

int launch(X3Args a, bool a_ks, bool b_ks, int rows, hipStream_t st, unsigned lds_pad = 0) {
  static const int dbg = getenv("RN_X3_DBG") ? atoi(getenv("RN_X3_DBG")) : 0;
  a.dbg = dbg;
  a.tiles_m = rn::ceil_div(a.a.rows, rows); a.tiles_n = rn::ceil_div(a.b.rows, rows);
  const long blocks = (long)a.nsplit * a.nbatch * a.tiles_m * a.tiles_n;
  RN_UNSUPPORTED(blocks <= 0 || blocks > 0x7fffffffL, "gemm x3: %ld blocks", blocks);
  const dim3 grid((unsigned)blocks);
  static const int nst_env = getenv("RN_X3_NST") ? atoi(getenv("RN_X3_NST")) : 0;      // register stages (tuning aid)
#define RN_X3K(WT_, NST_)                                                                                                \
  do {                                                                                                               \
    if (a_ks && b_ks) hipLaunchKernelGGL((gemm_x3_kernel<true, true, WT_, NST_>), grid, dim3(XT), lds_pad, st, a);        \
    else if (!a_ks && b_ks) hipLaunchKernelGGL((gemm_x3_kernel<false, true, WT_, NST_>), grid, dim3(XT), 0, st, a);       \
    else if (!a_ks && !b_ks) hipLaunchKernelGGL((gemm_x3_kernel<false, false, WT_, NST_>), grid, dim3(XT), 0, st, a);     \
    else hipLaunchKernelGGL((gemm_x3_kernel<true, false, WT_, NST_>), grid, dim3(XT), 0, st, a);                          \
  } while (0)
  if (rows == 128 && dbg) {     // (the timing aid measures the head-tower configuration)
    if (a_ks && b_ks) hipLaunchKernelGGL((gemm_x3_kernel<true, true, 2, 2, true>), grid, dim3(XT), lds_pad, st, a);
    else if (!a_ks && b_ks) hipLaunchKernelGGL((gemm_x3_kernel<false, true, 2, 2, true>), grid, dim3(XT), 0, st, a);
    else if (!a_ks && !b_ks) hipLaunchKernelGGL((gemm_x3_kernel<false, false, 2, 2, true>), grid, dim3(XT), 0, st, a);
    else hipLaunchKernelGGL((gemm_x3_kernel<true, false, 2, 2, true>), grid, dim3(XT), 0, st, a);
  } else if (rows == 128) {
    if (nst_env == 1) RN_X3K(2, 1);
    else RN_X3K(2, 2);
  } else {
    if (nst_env == 1) RN_X3K(1, 1);
    else if (nst_env == 2) RN_X3K(1, 2);
    else RN_X3K(1, 3);
  }
#undef RN_X3K
  RN_LAUNCH_CHECK();
  return RN_OK;
}

// k-ranges of the weight-gradient product: enough blocks to fill the chip, chunks of whole K-tiles
void tn_split(int M, int K, int N, int nbatch, int* nsplit, int* chunk, int* rows) {
  // (the tile follows the UNSPLIT problem: with 128 x 128 tiles a head-tower layer is 144 tiles x 3 ranges of the 682 tiles)
  const int tr = tile_rows(K, N, (long)nbatch * 3);
  *rows = tr;
  const long tiles = (long)nbatch * rn::ceil_div(K, tr) * rn::ceil_div(N, tr);
  int ns = (int)rn::ceil_div64(tr == 128 ? 400 : 1024, tiles > 0 ? tiles : 1);
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

// (K and N multiples of 4: float4 loads along k of the KC operands and along the rows of the KS operands)
bool gemm_x3_ok(int M, int K, int N) {
  return M >= 1 && K >= 4 && N >= 4 && K % 4 == 0 && N % 4 == 0 && fits((long)M * K) && fits((long)K * N) && fits((long)M * N);
}

int launch_batched_gemm_x3(const float* A, const float* B, float* C, int M, int K, int N, int nbatch, int b_nk, hipStream_t st) {
  X3Args a = {};
  a.a = {A, (long)M * K, K, M};
  a.b = b_nk ? X3Op{B, (long)K * N, K, N} : X3Op{B, (long)K * N, N, N};
  a.c = C; a.c_bstride = (long)M * N; a.c_sstride = 0; a.ldc = N;
  a.K = K; a.chunk = rn::ceil_div(K, XK) * XK; a.nsplit = 1; a.nbatch = nbatch;
  return launch(a, false, b_nk == 0, tile_rows(M, N, nbatch), st);
}

int x3_tile_rows(long rows_a, long rows_b, long batches) { return tile_rows(rows_a, rows_b, batches); }
bool x3_fits(long elems) { return fits(elems); }

// ---- dense 1x1 / stride-1 convolutions as plain products (round 6: the ResNeXt / DenseNet bottleneck 1x1 convs and the FPN laterals,
// resnet.py:38-49,67-69, densenet.py:61-66,137-142, retinanet.py:127-133,195-201): y [M x Cout] = x [M x Cin] W [Cin x Cout], x a
// channel slice (pixel stride x_ld) of a possibly wider buffer.  conv_gemm.hip routes them here in product mode 1.
static const bool conv1x1_on = !(getenv("RN_X3_CONV1X1") && atoi(getenv("RN_X3_CONV1X1")) == 0);
int conv1x1_x3_tile(long M, int Cin, int Cout, int x_ld) {
  if (mode() != 1 || !conv1x1_on || M < 1 || M > 0x7fffffffL || !gemm_x3_ok((int)M, Cin, Cout) || !fits(M * (long)x_ld)) return 0;
  return tile_rows(M, Cout, 1);
}
int launch_conv1x1_fwd_x3(const float* x, int x_ld, const float* w, float* y, int M, int Cin, int Cout, float2* stat_rows, hipStream_t st,
                          float drop_rate, uint64_t drop_seed, const uint64_t* drop_seed_dev) {
  X3Args a = {};
  if (drop_rate > 0.f) { a.drop_rate = drop_rate; a.drop_keep = 1.f / (1.f - drop_rate); a.drop_seed = drop_seed; a.drop_seed_dev = drop_seed_dev; }
  a.a = {x, 0, x_ld, M};
  a.b = {w, 0, Cout, Cout};                 // W [Cin][Cout]: k-strided
  a.c = y; a.c_bstride = 0; a.c_sstride = 0; a.ldc = Cout;
  a.K = Cin; a.chunk = rn::ceil_div(Cin, XK) * XK; a.nsplit = 1; a.nbatch = 1;
  a.stat = stat_rows;
  return launch(a, false, true, tile_rows(M, Cout, 1), st);
}
// dx [M x Cin] (pixel stride dx_ld) = dy [M x Cout] W^T
int launch_conv1x1_dgrad_x3(const float* dy, const float* w, float* dx, int dx_ld, int M, int Cin, int Cout, hipStream_t st) {
  X3Args a = {};
  a.a = {dy, 0, Cout, M};
  a.b = {w, 0, Cout, Cin};                  // Op2(ci, co) = W[ci * Cout + co]: k-contiguous
  a.c = dx; a.c_bstride = 0; a.c_sstride = 0; a.ldc = dx_ld;
  a.K = Cout; a.chunk = rn::ceil_div(Cout, XK) * XK; a.nsplit = 1; a.nbatch = 1;
  return launch(a, false, false, tile_rows(M, Cin, 1), st);
}
size_t conv1x1_wgrad_workspace_x3(int M, int Cin, int Cout) { return batched_gemm_tn_workspace_x3(M, Cin, Cout, 1); }
// slabs [nsplit][Cin][Cout] of x^T dy in `workspace` (the caller sums them: rn::launch_reduce_rows)
int launch_conv1x1_wgrad_x3(const float* x, int x_ld, const float* dy, int M, int Cin, int Cout, void* workspace, size_t workspace_bytes,
                            hipStream_t st, int* nsplit_out) {
  int ns, ck, tr;
  tn_split(M, Cin, Cout, 1, &ns, &ck, &tr);
  const size_t need = (size_t)ns * Cin * Cout * sizeof(float);
  if (workspace_bytes < need) {
    rn::set_error("conv 1x1 wgrad x3: workspace %zu < %zu bytes", workspace_bytes, need);
    return RN_EWORKSPACE;
  }
  X3Args a = {};
  a.a = {x, 0, x_ld, Cin};                  // rows = the Cin output rows, contraction index = the pixel
  a.b = {dy, 0, Cout, Cout};
  a.c = (float*)workspace; a.c_bstride = 0; a.c_sstride = (long)Cin * Cout; a.ldc = Cout;
  a.K = M; a.chunk = ck; a.nsplit = ns; a.nbatch = 1;
  *nsplit_out = ns;
  return launch(a, true, true, tr, st);
}

size_t batched_gemm_tn_workspace_x3(int M, int K, int N, int nbatch) {
  int ns, ck, tr;
  tn_split(M, K, N, nbatch, &ns, &ck, &tr);
  return (size_t)ns * nbatch * K * N * sizeof(float);
}

// slabs [nsplit][nbatch][K][N] of A_b^T B_b (A_b [M x K], B_b [M x N]) in `workspace`
// background != 0 (and RN_X3_BG_PAD > 0; measured, no gain: 482 vs 483 images/s, so off): the launch runs BESIDE latency-bound
// kernels of another stream (the deferred weight gradients of
// the head towers beside the backbone's backward pass): unused dynamic LDS keeps it at one block per CU, so that half of every
// SIMD's registers stay free for the other stream's blocks (at two blocks per CU -- 2 x 220 of 512 registers -- they would wait
// for a 25 us block to retire before every launch)
int launch_batched_gemm_tn_x3(const float* A, const float* B, int M, int K, int N, int nbatch, void* workspace, size_t workspace_bytes,
                              hipStream_t st, int* nsplit_out, int background) {
  int ns, ck, tr;
  tn_split(M, K, N, nbatch, &ns, &ck, &tr);
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
  *nsplit_out = ns;
  static const int bg_pad = getenv("RN_X3_BG_PAD") ? atoi(getenv("RN_X3_BG_PAD")) : 0;
  unsigned pad = 0;
  if (background && tr == 128 && bg_pad > 0) {
    static const bool ok =
        hipFuncSetAttribute((const void*)gemm_x3_kernel<true, true, 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, bg_pad) == hipSuccess &&
        hipFuncSetAttribute((const void*)gemm_x3_kernel<true, true, 2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, bg_pad) == hipSuccess;
    if (ok) pad = (unsigned)bg_pad;
    else (void)hipGetLastError();
  }
  return launch(a, true, true, tr, st, pad);
}
}  // namespace rn

extern "C" int rn_set_product_mode(int mode_) {
  rn::set_product_mode(mode_);
  return RN_OK;
}
extern "C" int rn_get_product_mode(void) { return rn::product_mode(); }

