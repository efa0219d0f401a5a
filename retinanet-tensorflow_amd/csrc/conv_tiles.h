// Tile-level building blocks of the fp32-MFMA implicit-GEMM kernels (conv_gemm.hip, mbconv.hip): raw buffer loads with
// hardware range checks, the K-tile MFMA loop over LDS-staged operands, the accumulator store.  Not part of the public ABI.
#pragma once
#include "rn_common.h"

namespace rn_tiles {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;   // K-tile
constexpr int LDK = 36;  // row stride (floats) of k-contiguous LDS tiles: 144 B, 16-B aligned,
                         // rows r and r+1 are 4 banks apart => b128 fragment reads conflict-free

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0x80000000u;  // voffset beyond any tensor (< 2 GiB each): buffer loads return 0,
                                       // buffer stores are dropped -- no branches around memory ops

// Raw buffer access: 32-bit byte offsets + hardware range check against the tensor size.
template <int VEC>
struct Vec;
template <>
struct Vec<4> {
  typedef float4 type;
  static __device__ __forceinline__ float4 load(__amdgpu_buffer_rsrc_t rs, unsigned voff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
  }
};
template <>
struct Vec<1> {
  typedef float type;
  static __device__ __forceinline__ float load(__amdgpu_buffer_rsrc_t rs, unsigned voff) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voff, 0, 0));
  }
};
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

// ---------------------------------------------------------------------------------------------
// MFMA over one staged K-tile.  A fragment: lane l supplies A[row l&31][k-slot l>>5]; we let
// lane-half h own the 4 consecutive k = 8*kg + 4*h .. +3 of each 8-wide k-group, so an
// MK-layout fragment is ONE ds_read_b128.  B uses the same k per half, so the sum is complete.
// ---------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, bool A_KM, bool B_NK>
__device__ __forceinline__ void mma_ktile(const float* __restrict__ As, const float* __restrict__ Bs,
                                          f32x16 (&acc)[BM / WM / 32][BN / WN / 32], int wm, int wn, int lane) {
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  const int l31 = lane & 31, half = lane >> 5;
#pragma unroll
  for (int kg = 0; kg < BK / 8; ++kg) {
    const int k0 = kg * 8 + half * 4;
    float a[TM][4], b[TN][4];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const int row = wm * (BM / WM) + tm * 32 + l31;
      if (A_KM) {
#pragma unroll
        for (int j = 0; j < 4; ++j) a[tm][j] = As[(k0 + j) * BM + row];
      } else {
        const float4 v = *reinterpret_cast<const float4*>(&As[row * LDK + k0]);
        a[tm][0] = v.x; a[tm][1] = v.y; a[tm][2] = v.z; a[tm][3] = v.w;
      }
    }
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int col = wn * (BN / WN) + tn * 32 + l31;
      if (B_NK) {
        const float4 v = *reinterpret_cast<const float4*>(&Bs[col * LDK + k0]);
        b[tn][0] = v.x; b[tn][1] = v.y; b[tn][2] = v.z; b[tn][3] = v.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) b[tn][j] = Bs[(k0 + j) * BN + col];
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][j], b[tn][j], acc[tm][tn], 0, 0, 0);
  }
}

// C/D map of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
// Stores go through a buffer descriptor sized to exactly `mmax` rows: rows past the end are
// dropped by the range check, only the column needs a test => 16 back-to-back stores per tile.
template <int BM, int BN, int WM, int WN>
__device__ __forceinline__ void store_tile(const f32x16 (&acc)[BM / WM / 32][BN / WN / 32], float* __restrict__ out,
                                           const float* __restrict__ bias, int m0, int n0, int mmax, int nmax,
                                           int ldc, int wm, int wn, int lane) {
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  const int l31 = lane & 31, half = lane >> 5;
  const __amdgpu_buffer_rsrc_t rs = make_rsrc(out, (unsigned)mmax * (unsigned)ldc * 4u);
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int col = n0 + wn * (BN / WN) + tn * 32 + l31;
    const bool cok = col < nmax;
    const float bv = (bias != nullptr && cok) ? bias[col] : 0.f;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const int rbase = m0 + wm * (BM / WM) + tm * 32 + 4 * half;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rbase + (r & 3) + 8 * (r >> 2);
        const unsigned voff = cok ? ((unsigned)row * (unsigned)ldc + (unsigned)col) * 4u : OOB;
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[tm][tn][r] + bv), rs, voff, 0, 0);
      }
    }
  }
}

template <int TM, int TN>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[TM][TN]) {
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
}

}  // namespace rn_tiles
