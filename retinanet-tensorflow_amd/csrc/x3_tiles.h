// Shared device code of the split-bf16 product kernels (gemm_x3.hip, gemm_x3_bfrag.hip): the kernel arguments, the exact three-way
// split, the operand tiles in LDS and their fragments.  Included INSIDE each file's anonymous namespace (after conv_tiles.h and
// `using namespace rn_tiles`): every translation unit gets its own copies, nothing here has external linkage.
// (Two translation units, not one: with gemm_x3_bfrag_kernel instantiated next to them the compiler laid gemm_x3_kernel's k-strided
// tile loads out as branches instead of selects -- the weight-gradient kernel <true, true, 1, 3> of the dense 1 x 1 convs ran 47 %
// longer, cfg 3 / cfg 4 lost 6 % -- profiles/r06_ab_runs.txt (14).)
#pragma once


typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2v __attribute__((ext_vector_type(2)));

constexpr int XK = 32, XT = 256;
constexpr int LDR = 40;               // halfs per LDS row of a KC tile: 32 k + 8 pad = 80 bytes

struct X3Op { const float* p; long bstride; int ld, rows; };
struct X3Args {
  X3Op a, b;
  float* c; long c_bstride, c_sstride; int ldc;
  int K, chunk, nsplit, nbatch, tiles_m, tiles_n;
  float2* stat;       // dense 1x1 convs (rn::launch_conv1x1_fwd_x3): per (m-tile, column) sums (sum y, sum y^2) over the tile's rows -> stat[tile_m * ldc + col]
                      // (the GroupNorm statistic rows of conv_gemm.hip's conv_stats_epilogue, same layout); nullptr: off
  // dense 1x1 convs followed by a Dropout (DenseNet's composite function, densenet.py:61-67): the mask of rn_dropout -- keep element i of the
  // output tensor iff uniform01(seed + *seed_dev, i) >= rate, scaled by 1 / (1 - rate) -- applied to the accumulators before they are stored
  // and summed: the conv's output never exists un-dropped, `stat` holds the sums of the DROPPED tensor.  rate == 0: off
  float drop_rate, drop_keep; uint64_t drop_seed; const uint64_t* drop_seed_dev;
  int dbg;      // RN_X3_DBG (timing aid, wrong results): bit 0 no MFMAs, bit 1 no split / LDS stores, bit 2 no global loads after the first, bit 3 no fragment reads, bit 4 the dword epilogue
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

// One operand tile [ROWS rows][32 k] (ROWS = 64 or 128): ROWS / 32 float4 loads per thread either way.
//   KC (k contiguous in memory): 4 consecutive k of rows t/8 + 32 i  -> LDS [row][k] bf16, 80-byte rows: a lane's MFMA fragment
//      (8 consecutive k of one row) is ONE ds_read_b128
//   KS (rows contiguous in memory): rows 4 (t%16) + 64 i2 .. +3 at k = 2 (t/16) and 2 (t/16) + 1 -> LDS [k pair][row] dwords (a
//      dword = the bf16 pair (k, k + 1) of one row; rows of ROWS + 4 dwords): the thread's four rows of a plane are ONE
//      ds_write_b128, a fragment is four ds_read_b32 a k-pair apart (lanes = consecutive rows: conflict-free) -- the transpose
//      costs nothing
template <int ROWS>
struct TileGeom {
  static constexpr int PLANE = ROWS * LDR;        // halfs per plane (the KS image, 16 x (ROWS + 4) dwords, is smaller)
  static constexpr int KS_LD = ROWS + 4;          // dwords per k-pair row of a KS tile
  static constexpr int NQ = ROWS / 32;            // float4 per thread
};
template <bool KS, int ROWS>
struct TileLoad {
  typedef TileGeom<ROWS> G;
  float4 q[G::NQ];
  __device__ __forceinline__ void load(const __amdgpu_buffer_rsrc_t rs, const X3Op& op, int row0, int k0, int k1, int t) {
    if (KS) {
      const int k = k0 + (t >> 4) * 2;
#pragma unroll
      for (int i2 = 0; i2 < G::NQ / 2; ++i2) {
        const int row = row0 + (t & 15) * 4 + 64 * i2;
        const bool rok = row < op.rows;               // (rows % 4 == 0: a quad is inside or outside)
#pragma unroll
        for (int i = 0; i < 2; ++i)
          q[2 * i2 + i] = Vec<4>::load(rs, (rok && k + i < k1) ? ((unsigned)(k + i) * (unsigned)op.ld + (unsigned)row) * 4u : OOB);
      }
    } else {
      const int k = k0 + (t & 7) * 4;
#pragma unroll
      for (int i = 0; i < G::NQ; ++i) {
        const int row = row0 + (t >> 3) + 32 * i;
        q[i] = Vec<4>::load(rs, (row < op.rows && k < k1) ? ((unsigned)row * (unsigned)op.ld + (unsigned)k) * 4u : OOB);
      }
    }
  }
  // the three planes of this thread's values -> LDS (`tile`: the operand's plane 0; planes PLANE halfs apart)
  __device__ __forceinline__ void store(unsigned short* tile, int t) const {
    if (KS) {       // q[2 i2] = rows r..r+3 at k, q[2 i2 + 1] = the same rows at k + 1
#pragma unroll
      for (int i2 = 0; i2 < G::NQ / 2; ++i2) {
        const float v[8] = {q[2 * i2].x, q[2 * i2].y, q[2 * i2].z, q[2 * i2].w, q[2 * i2 + 1].x, q[2 * i2 + 1].y, q[2 * i2 + 1].z, q[2 * i2 + 1].w};
        unsigned h[3][8];
#pragma unroll
        for (int j = 0; j < 8; ++j) split3(v[j], h[0][j], h[1][j], h[2][j]);
        unsigned* dst = reinterpret_cast<unsigned*>(tile) + (t >> 4) * G::KS_LD + (t & 15) * 4 + 64 * i2;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          u32x4 w;
          w.x = pack_hi(h[p][0], h[p][4]); w.y = pack_hi(h[p][1], h[p][5]);
          w.z = pack_hi(h[p][2], h[p][6]); w.w = pack_hi(h[p][3], h[p][7]);
          *reinterpret_cast<u32x4*>(dst + p * (G::PLANE / 2)) = w;
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < G::NQ; ++i) {
        unsigned h[3][4];
        split3(q[i].x, h[0][0], h[1][0], h[2][0]); split3(q[i].y, h[0][1], h[1][1], h[2][1]);
        split3(q[i].z, h[0][2], h[1][2], h[2][2]); split3(q[i].w, h[0][3], h[1][3], h[2][3]);
        unsigned short* dst = tile + ((t >> 3) + 32 * i) * LDR + (t & 7) * 4;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          u32x2v w;
          w.x = pack_hi(h[p][0], h[p][1]); w.y = pack_hi(h[p][2], h[p][3]);
          *reinterpret_cast<u32x2v*>(dst + p * G::PLANE) = w;
        }
      }
    }
  }
};
// a lane's fragment of k-step s (k = 16 s + 8 h .. + 7) of row `row` of the tile's plane p
template <bool KS, int ROWS>
__device__ __forceinline__ bf16x8 fragment(const unsigned short* tile, int p, int row, int h, int s) {
  typedef TileGeom<ROWS> G;
  if (KS) {
    const unsigned* src = reinterpret_cast<const unsigned*>(tile) + p * (G::PLANE / 2) + (8 * s + 4 * h) * G::KS_LD + row;
    u32x4 w;
    w.x = src[0]; w.y = src[G::KS_LD]; w.z = src[2 * G::KS_LD]; w.w = src[3 * G::KS_LD];
    return __builtin_bit_cast(bf16x8, w);
  }
  return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(tile + p * G::PLANE + row * LDR + s * 16 + h * 8));
}

