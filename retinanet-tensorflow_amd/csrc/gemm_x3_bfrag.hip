// The forward products of the GroupNorm-folded Winograd tower layers (retinanet.py:37-62,85-106) with their kernel operand U pre-split
// by its producer: see FragB.  A translation unit of its own (x3_tiles.h says why); the host entry points of this path live here too.
#include <stdlib.h>

#include "conv_tiles.h"
#include "rn_common.h"

namespace {
using namespace rn_tiles;
#include "x3_tiles.h"

// gemm_x3_bfrag_kernel: Op2 comes pre-split and in MFMA-fragment order (see `FragB` below) -- `p` per batch `bstride` bytes apart,
// `nblk` 32-row blocks x `ks16` 16-k steps; X3Args::b.rows (the column bound of the epilogue) as ever, b.p / b.ld unused.
// (A second kernel argument, not more fields of X3Args: with the larger struct the compiler laid the k-strided tile loads of
// gemm_x3_kernel out as branches instead of selects -- the weight-gradient kernel <true, true, 1, 3> ran 47 % longer.)
struct X3Frag { const void* p; long bstride; int nblk, ks16; };

// Op2 pre-split by its PRODUCER and laid out in MFMA-fragment order (the Winograd kernel transform writes U / Urot this way once per
// layer and step, winograd.hip wino_weight_frag_body; rn_x3_pack_bfrag for stand-alone products): per batch matrix
//     [n block nb = n / 32][k step kk = k / 16][plane p < 3][lane l < 64][8 bf16]       (6 bytes per element, K % 16 == 0, n padded to 32)
// with lane l holding column n = 32 nb + (l & 31) at k = 16 kk + 8 (l >> 5) .. + 7 -- what lane l feeds v_mfma_f32_32x32x16_bf16 as
// its B operand is ONE 16-byte global load (1 KB contiguous per wave and fragment; L2-resident: every m-tile of a batch re-reads
// the same 48 KB per column block) straight into the registers the instruction reads.  Against the LDS route that removes, per
// K-step of a 128 x 128 tile and thread: the split of 16 elements (~90 VALU operations), 6 ds_write_b128 and 48 ds_read_b32 (the
// k-strided operand's fragments are dword gathers), and half of the block's LDS.  The fragments of K-step t + 1 are requested as
// the matrix-core instructions of K-step t release their registers.
struct FragB {
  __amdgpu_buffer_rsrc_t rs;
  unsigned base[2];       // byte offset of (column block of this wave's tile j, k step 0, plane 0, this lane); OOB past the last block
  int ks16;
  template <int WT>
  __device__ __forceinline__ void init(const X3Frag& a, int batch, int n0, int wn, int lane) {
    rs = make_rsrc(static_cast<const char*>(a.p) + (size_t)batch * a.bstride, (unsigned)a.nblk * (unsigned)a.ks16 * 3072u);
    ks16 = a.ks16;
#pragma unroll
    for (int j = 0; j < WT; ++j) {
      const int nb = (n0 >> 5) + wn * WT + j;
      base[j] = nb < a.nblk ? ((unsigned)nb * (unsigned)a.ks16 * 192u + (unsigned)lane) * 16u : OOB;
    }
  }
  // the three planes of (tile column block j, 16-k step kk); steps past K: zeros (out of the descriptor's range)
  __device__ __forceinline__ void load(int j, int kk, bf16x8 (&f)[3]) const {
    const unsigned vo = (kk < ks16 && base[j] != OOB) ? base[j] + (unsigned)kk * 3072u : OOB;
#pragma unroll
    for (int p = 0; p < 3; ++p) f[p] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, p * 1024, 0));
  }
};

// The forward products of the GroupNorm-folded tower layers with their kernel operand as a fragment image (FragB): the K loop of
// gemm_x3_kernel<false, *, WT, NST> with Op2 out of LDS, as a kernel of its own -- as one more template parameter of gemm_x3_kernel
// the compiler laid the OTHER instantiations out differently (the weight-gradient kernel <true, true, 1, 3> of the dense 1 x 1 convs
// ran 47 % longer, cfg 3 / cfg 4 lost 6 %: profiles/r06_ab_runs.txt (14)).  FWD_NAME only names the instantiation (forward / data
// gradient in a profile).  Op1 k-contiguous, one k range, no statistics / dropout epilogue.
template <bool FWD_NAME, int WT, int NST, bool DBG = false>
__global__ __launch_bounds__(XT, WT == 2 ? 2 : 4) void gemm_x3_bfrag_kernel(const X3Args a, const X3Frag bf) {
  const int dbg = DBG ? a.dbg : 0;
  constexpr int ROWS = 64 * WT;
  typedef TileGeom<ROWS> G;
  // (the A planes only -- or the epilogue's staging area, 4 waves x 32 rows x 68 floats, where that is larger)
  constexpr int LDS_HALFS = (WT == 2 && 3 * G::PLANE < 4 * 32 * 68 * 2) ? 4 * 32 * 68 * 2 : 3 * G::PLANE;
  __shared__ __attribute__((aligned(16))) unsigned short lds[LDS_HALFS];
  unsigned short* At = lds;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;
  const int bid = rn::xcd_remap(blockIdx.x, gridDim.x);
  const int tile_n = bid % a.tiles_n;
  const int tile_m = (bid / a.tiles_n) % a.tiles_m;
  const int batch = bid / (a.tiles_n * a.tiles_m);
  const int m0 = tile_m * ROWS, n0 = tile_n * ROWS;
  const int kend = a.K;
  const float* pa = a.a.p + (size_t)batch * a.a.bstride;
  const __amdgpu_buffer_rsrc_t ra = make_rsrc(pa, (unsigned)a.a.rows * (unsigned)a.a.ld * 4u);
  TileLoad<false, ROWS> la[NST];
  FragB fragb;
  bf16x8 fbq[2][WT][3];                       // the B fragments of the K-step at hand (16-k halves s = 0, 1), refilled in place
  f32x16 acc[WT][WT];
  zero_acc<WT, WT>(acc);
  const int nk = (kend + XK - 1) / XK;
  fragb.template init<WT>(bf, batch, n0, wn, lane);
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int j = 0; j < WT; ++j) fragb.load(j, s, fbq[s][j]);
#pragma unroll
  for (int st = 0; st < NST; ++st)
    if (st < nk) la[st].load(ra, a.a, m0, st * XK, kend, t);
  for (int it0 = 0; it0 < nk; it0 += NST) {
#pragma unroll
    for (int st = 0; st < NST; ++st) {
      const int it = it0 + st;
      if (it < nk) {                         // (block-uniform)
        if (!(dbg & 2)) la[st].store(At, t);
        __syncthreads();
        if (it + NST < nk && !(dbg & 4)) la[st].load(ra, a.a, m0, (it + NST) * XK, kend, t);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          bf16x8 fa[WT][3];
          if (!(dbg & 8)) {
#pragma unroll
            for (int i = 0; i < WT; ++i)
#pragma unroll
              for (int p = 0; p < 3; ++p) fa[i][p] = fragment<false, ROWS>(At, p, (wm * WT + i) * 32 + r, h, s);
          } else {
#pragma unroll
            for (int i = 0; i < WT; ++i)
#pragma unroll
              for (int p = 0; p < 3; ++p) fa[i][p] = __builtin_bit_cast(bf16x8, u32x4{(unsigned)t, 1u, 2u, 3u});
          }
          if (!(dbg & 1))
#pragma unroll
          for (int i = 0; i < WT; ++i)
#pragma unroll
            for (int j = 0; j < WT; ++j) {   // (the small terms first: the order of gemm_x3_kernel -- bit-identical results)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][2], fbq[s][j][0], acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fbq[s][j][2], acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fbq[s][j][1], acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fbq[s][j][0], acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fbq[s][j][1], acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fbq[s][j][0], acc[i][j], 0, 0, 0);
            }
          // the same half of the NEXT K-step into the registers just read (past the last step: out of range, zeros, no traffic)
          if (!(dbg & 4)) {
#pragma unroll
            for (int j = 0; j < WT; ++j) fragb.load(j, it + 1 < nk ? 2 * (it + 1) + s : 0x7fffffff, fbq[s][j]);
          }
        }
        __syncthreads();
      }
    }
  }
  float* pc = a.c + (size_t)batch * a.c_bstride;
  if (WT == 2 && !(dbg & 16)) {              // 16-byte stores through LDS (see gemm_x3_kernel)
    float* stage = reinterpret_cast<float*>(lds) + wave * (32 * 68);
    const __amdgpu_buffer_rsrc_t rc = make_rsrc(pc, (unsigned)a.a.rows * (unsigned)a.ldc * 4u);
    const int col0 = n0 + wn * 64;
#pragma unroll
    for (int i = 0; i < WT; ++i) {
#pragma unroll
      for (int j = 0; j < WT; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) stage[((q & 3) + 8 * (q >> 2) + 4 * h) * 68 + j * 32 + r] = acc[i][j][q];
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
}

// fp32 B_b ([K][N], or [N][K] when b_nk) -> its fragment image (FragB): stand-alone products (tests, bench.py, tools); inside the
// network the Winograd kernel transform writes the image itself.  One thread per dword (two consecutive k of one column).
__global__ __launch_bounds__(256) void pack_bfrag_kernel(const float* __restrict__ B, unsigned* __restrict__ out, int K, int N, int b_nk,
                                                         int ks16, long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int j = (int)(i & 3), lane = (int)((i >> 2) & 63);
  const long blk = i >> 8;
  const int kk = (int)(blk % ks16), nb = (int)(blk / ks16);
  const int n = nb * 32 + (lane & 31), k = kk * 16 + (lane >> 5) * 8 + 2 * j;
  const float* Bb = B + (size_t)blockIdx.y * K * N;
  float v0 = 0.f, v1 = 0.f;
  if (n < N) {
    v0 = b_nk ? Bb[(size_t)n * K + k] : Bb[(size_t)k * N + n];
    v1 = b_nk ? Bb[(size_t)n * K + k + 1] : Bb[(size_t)(k + 1) * N + n];
  }
  unsigned h0[3], h1[3];
  split3(v0, h0[0], h0[1], h0[2]);
  split3(v1, h1[0], h1[1], h1[2]);
  unsigned* o = out + (size_t)blockIdx.y * (size_t)(total / 256) * 768 + (size_t)blk * 768 + lane * 4 + j;
#pragma unroll
  for (int p = 0; p < 3; ++p) o[p * 256] = pack_hi(h0[p], h1[p]);
}


int launch_bfrag(X3Args a, const X3Frag& bf, bool fwd_name, int rows, hipStream_t st) {
  static const int dbg = getenv("RN_X3_DBG") ? atoi(getenv("RN_X3_DBG")) : 0;
  a.dbg = dbg;
  a.tiles_m = rn::ceil_div(a.a.rows, rows); a.tiles_n = rn::ceil_div(a.b.rows, rows);
  const long blocks = (long)a.nbatch * a.tiles_m * a.tiles_n;
  RN_UNSUPPORTED(blocks <= 0 || blocks > 0x7fffffffL, "gemm x3 bfrag: %ld blocks", blocks);
  const dim3 grid((unsigned)blocks);
  if (rows == 128 && dbg) {     // (the leave-one-out timing aid measures the head-tower configuration)
    if (fwd_name) hipLaunchKernelGGL((gemm_x3_bfrag_kernel<true, 2, 2, true>), grid, dim3(XT), 0, st, a, bf);
    else hipLaunchKernelGGL((gemm_x3_bfrag_kernel<false, 2, 2, true>), grid, dim3(XT), 0, st, a, bf);
  } else if (rows == 128) {
    if (fwd_name) hipLaunchKernelGGL((gemm_x3_bfrag_kernel<true, 2, 2>), grid, dim3(XT), 0, st, a, bf);
    else hipLaunchKernelGGL((gemm_x3_bfrag_kernel<false, 2, 2>), grid, dim3(XT), 0, st, a, bf);
  } else {
    if (fwd_name) hipLaunchKernelGGL((gemm_x3_bfrag_kernel<true, 1, 3>), grid, dim3(XT), 0, st, a, bf);
    else hipLaunchKernelGGL((gemm_x3_bfrag_kernel<false, 1, 3>), grid, dim3(XT), 0, st, a, bf);
  }
  RN_LAUNCH_CHECK();
  return RN_OK;
}
}  // namespace

namespace rn {
// ---- Op2 pre-split in fragment order (FragB above) ----
static int g_bfrag = -1;
int bfrag_on() {
  if (g_bfrag < 0) g_bfrag = !(getenv("RN_X3_BFRAG") && atoi(getenv("RN_X3_BFRAG")) == 0);
  return g_bfrag;
}
void set_bfrag(int on) { g_bfrag = on ? 1 : 0; }
// (k in whole 16-steps; the columns are padded to whole 32-blocks inside the image)
// (the M-independent half: which FORMAT a [K x N] kernel operand has under the current switches)
bool x3_bfrag_format(int K, int N) { return product_mode() == 1 && bfrag_on() && K >= 16 && K % 16 == 0 && N >= 4 && N % 4 == 0 && x3_fits((long)(x3_bfrag_bytes(K, N) / 4)); }
bool x3_bfrag_ok(int M, int K, int N) { return x3_bfrag_format(K, N) && gemm_x3_ok(M, K, N); }
size_t x3_bfrag_bytes(int K, int N) { return (size_t)rn::ceil_div(N, 32) * (size_t)(K / 16) * 3072u; }
// C_b [M x N] = A_b [M x K] * B_b, B_b given as its fragment image (`fwd_name`: which of the two identical instantiations runs --
// the profiler then tells forward products from data-gradient products by name)
int launch_batched_gemm_x3_bfrag(const float* A, const void* Bfrag, float* C, int M, int K, int N, int nbatch, int fwd_name, hipStream_t st) {
  X3Args a = {};
  a.a = {A, (long)M * K, K, M};
  a.b = {nullptr, 0, N, N};
  const X3Frag bf = {Bfrag, (long)x3_bfrag_bytes(K, N), rn::ceil_div(N, 32), K / 16};
  a.c = C; a.c_bstride = (long)M * N; a.c_sstride = 0; a.ldc = N;
  a.K = K; a.chunk = rn::ceil_div(K, XK) * XK; a.nsplit = 1; a.nbatch = nbatch;
  return launch_bfrag(a, bf, fwd_name != 0, x3_tile_rows(M, N, nbatch), st);
}

int launch_pack_bfrag(const float* B, void* out, int K, int N, int nbatch, int b_nk, hipStream_t st) {
  const int ks16 = K / 16;
  const long total = (long)rn::ceil_div(N, 32) * ks16 * 256;
  hipLaunchKernelGGL(pack_bfrag_kernel, dim3((unsigned)rn::ceil_div64(total, 256), (unsigned)nbatch), dim3(256), 0, st, B, (unsigned*)out, K, N, b_nk,
                     ks16, total);
  RN_LAUNCH_CHECK();
  return RN_OK;
}

}  // namespace rn

extern "C" int rn_set_x3_bfrag(int on) {
  rn::set_bfrag(on);
  return RN_OK;
}
extern "C" int rn_get_x3_bfrag(void) { return rn::bfrag_on(); }
extern "C" int rn_x3_bfrag_ok(int M, int K, int N) { return rn::x3_bfrag_ok(M, K, N) ? 1 : 0; }
extern "C" size_t rn_x3_bfrag_bytes(int K, int N, int nbatch) { return (K > 0 && N > 0 && nbatch > 0 && K % 16 == 0) ? rn::x3_bfrag_bytes(K, N) * (size_t)nbatch : 0; }
extern "C" int rn_x3_pack_bfrag(const float* B, void* out, int K, int N, int nbatch, int b_nk, rn_stream_t stream) {
  RN_CHECK_ARG(B && out && K >= 16 && K % 16 == 0 && N >= 1 && nbatch >= 1 && nbatch <= 65535, "x3 pack: bad argument");
  return rn::launch_pack_bfrag(B, out, K, N, nbatch, b_nk, (hipStream_t)stream);
}
extern "C" int rn_gemm_batched_bfrag(const float* A, const void* Bfrag, float* C, int M, int K, int N, int nbatch, int fwd_name, rn_stream_t stream) {
  RN_CHECK_ARG(A && Bfrag && C && nbatch >= 1, "gemm bfrag: bad argument");
  RN_UNSUPPORTED(!rn::x3_bfrag_ok(M, K, N), "gemm bfrag: [%d x %d] x [%d x %d] cannot take a fragment-ordered operand (mode / RN_X3_BFRAG / K %% 16)", M, K, K, N);
  return rn::launch_batched_gemm_x3_bfrag(A, Bfrag, C, M, K, N, nbatch, fwd_name, (hipStream_t)stream);
}
