// fp16 inference convolution: NHWC half activations, implicit GEMM on the f16 matrix cores
// (v_mfma_f32_32x32x16_f16: fp16 in, fp32 accumulate), forward only.
// BASELINE configs[4]: "Inference-only ResNeXt-50-FPN 1024x1024 bs=16, fp16".
//
//   Y[m=(n,oh,ow)][co] = sum_k A[m][k=(kh,kw,ci)] * Wt[co][k]   (+bias), fp32 accumulate
//
// Weights are packed ONCE (rn_pack_weights_f16) from the fp32 HWIO kernel [kh,kw,cin_g,cout] to
// Wt[cout][K] half (k contiguous), so both operands are k-contiguous in LDS and every MFMA fragment
// (8 consecutive k of one row) is one ds_read_b128.  K-tile = 64 halfs, rows padded to 72 halfs
// (144 B, the same conflict-free stride as the fp32 kernels).  Everything else follows conv_gemm.hip:
// multi-segment launches, buffer loads with hardware range checks, block-uniform filter-tap state,
// groups (ResNeXt), channel-slice input views, XCD-aware tile order.
#include <stdlib.h>

#include "rn_common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int BK = 64;   // halfs per K-tile
constexpr int LDH = 72;  // LDS row stride in halfs (144 B)
constexpr unsigned OOB = 0x80000000u;

struct SegH {
  const _Float16* x; const _Float16* wt; const float* bias; void* y;
  const _Float16* wf;   // the kernel once more in MFMA-fragment order (frag_offset_halfs behind wt), nullptr when K % 16 != 0
  int n, h, w, oh, ow, cout, pad_t, pad_l, m, tiles_n, start, x_ld, x_coff;
  // FOUT kernels: this segment's (m-tile, channel) statistic rows -- row `tile_m` of [2][prows][cout] (the second half prows * cout
  // floats behind the first); nullptr: the segment emits none (its tiles straddle samples: the caller takes them from the tensor)
  float* partial; int prows;
};
// rn_f16_fold on the device: the GroupNorm in FRONT of the conv applied to the A operand between the global load and the LDS
// store (x holds the raw output of the previous conv), and / or the statistics of the GroupNorm BEHIND the conv from the
// epilogue ([2][prows][cout] per (m-tile, channel) sums of y and y^2 of the values as stored, gn_partial_kernel's layout)
struct FoldH {
  const float* in_mean; const float* in_rstd; const float* in_gamma; const float* in_beta;
  int in_groups, in_cpg, in_act, in_c;
  float* partial; int prows;
};
struct ArgsH {
  SegH seg[RN_MAX_SEG];
  int nseg, kh, kw, stride, cin, groups, cin_g, tpg, out_f32;
  int a_nt;      // the activation operand is read exactly once by the whole launch (1 x 1 conv, one N-tile): non-temporal loads
  FoldH fold;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

template <int VEC>
struct VecH;
template <>
struct VecH<8> {
  typedef half8 type;
  static __device__ __forceinline__ half8 load(__amdgpu_buffer_rsrc_t rs, unsigned voff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, 0);
    return __builtin_bit_cast(half8, v);
  }
  static __device__ __forceinline__ half8 load_nt(__amdgpu_buffer_rsrc_t rs, unsigned voff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, 2);      // aux bit 1: nt
    return __builtin_bit_cast(half8, v);
  }
};
template <>
struct VecH<4> {
  typedef half4 type;
  static __device__ __forceinline__ half4 load(__amdgpu_buffer_rsrc_t rs, unsigned voff) {
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, 0, 0);
    return __builtin_bit_cast(half4, v);
  }
  static __device__ __forceinline__ half4 load_nt(__amdgpu_buffer_rsrc_t rs, unsigned voff) {
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, 0, 2);
    return __builtin_bit_cast(half4, v);
  }
};

// (fp16) fma(fp32(x), sc, sh) of the two halfs of `p`, each with its own fp32 (sc, sh): v_fma_mixlo_f16 / v_fma_mixhi_f16 read the
// fp16 operand from the chosen half, compute in fp32 and write the chosen half of the destination (the compiler selects
// v_fma_mix_f32 + a separate convert for the same expression)
__device__ __forceinline__ uint32_t fma_mix_pair(uint32_t p, float sc0, float sh0, float sc1, float sh1) {
  uint32_t d;
  asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=&v"(d) : "v"(p), "v"(sc0), "v"(sh0));
  asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(d) : "v"(p), "v"(sc1), "v"(sh1));
  return d;
}

__device__ __forceinline__ int find_seg(const ArgsH& a, int id) {
  int s = 0;
  while (s + 1 < a.nseg && id >= a.seg[s + 1].start) ++s;
  return s;
}

// FOLD bit 0: GroupNorm + activation of the input applied on load; bit 1: statistics of the output from the epilogue; bit 2
// (with bit 0): the input's activation is ReLU, known at compile time.
// Both need a tile's rows inside one sample (oh * ow a multiple of BM: host-checked) and a single segment.
// PIPE (the 256 x 256 tile: ONE block of 8 waves per CU, so nothing else covers a block's own stalls): the operand tiles are
// double-buffered in (dynamic) LDS -- tile t+1 goes from the staging registers into the other buffer between the MFMAs of tile
// t and the loads of tile t+2 are issued behind it: one barrier per K-tile instead of two, and no wave waits on LDS stores or
// on the first fragment reads with the matrix pipe idle.
template <int BM, int BN, int WM, int WN, int VEC, bool TAPU, int FOLD, int PIPE = 0, int DBG = 0>
__global__ __launch_bounds__(WM* WN * 64) void conv_f16_kernel(const ArgsH args) {
  constexpr int T = WM * WN * 64;
  constexpr bool FIN = (FOLD & 1) != 0, FOUT = (FOLD & 2) != 0, FIN_RELU = (FOLD & 4) != 0;
  // (scale, shift) of every input channel of the tile's sample; the pipelined variant keeps the table in the half of its dynamic
  // LDS that the weight tiles do not use (PIPE == 2 reads its weights from the fragment-ordered copy) -- 160 KB are all there is
  __shared__ float2 ntab_st[(FIN && !PIPE) ? 2048 : 1];
  __shared__ float sred[FOUT ? WM * BN * 2 : 1];
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int KQ = BK / VEC, RPP = T / KQ, A_PASS = BM / RPP, B_PASS = BN / RPP;
  static_assert(A_PASS >= 1 && B_PASS >= 1 && BM % RPP == 0 && BN % RPP == 0, "tile/threads mismatch");
  typedef typename VecH<VEC>::type vec_t;
  constexpr int BUF = (BM + BN) * LDH;
  extern __shared__ __attribute__((aligned(16))) _Float16 dyn_smem[];
  __shared__ __attribute__((aligned(16))) _Float16 st_smem[PIPE ? 8 : BUF];
  _Float16* smem = PIPE ? dyn_smem : st_smem;
  float2* ntab = (FIN && PIPE) ? reinterpret_cast<float2*>(dyn_smem + 2 * BM * LDH) : ntab_st;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int bid = rn::xcd_remap(blockIdx.x, gridDim.x);
  const int s = find_seg(args, bid);
  const SegH& sg = args.seg[s];
  const int local = bid - sg.start;
  const int tile_n = local % sg.tiles_n, tile_m = local / sg.tiles_n;
  const int H = sg.h, W = sg.w, OW = sg.ow, OHW = sg.oh * sg.ow, M = sg.m, cout = sg.cout;
  const int ldx = sg.x_ld, kw = args.kw, stride = args.stride;
  const int G = args.groups, cin = args.cin_g, cout_g = cout / G;
  const int grp = tile_n / args.tpg, tn_ = tile_n - grp * args.tpg;
  const int m0 = tile_m * BM, n0 = grp * cout_g + tn_ * BN;
  const int nmax = (grp + 1) * cout_g;
  const int a_coff = sg.x_coff + grp * cin;
  const int ktotal = args.kh * args.kw * cin;
  const __amdgpu_buffer_rsrc_t xa = make_rsrc(sg.x, (unsigned)sg.n * H * W * ldx * 2u);
  const __amdgpu_buffer_rsrc_t wb = make_rsrc(sg.wt, (unsigned)cout * ktotal * 2u);

  if (FIN) {
    const FoldH& f = args.fold;
    const int smp = m0 / OHW;
    for (int c = tid; c < f.in_c; c += T) {
      const int g_ = c / f.in_cpg;
      const float mean = f.in_mean[smp * f.in_groups + g_], rstd = f.in_rstd[smp * f.in_groups + g_];
      const float sc = rstd * f.in_gamma[c];
      ntab[c] = make_float2(sc, f.in_beta[c] - mean * sc);
    }
    __syncthreads();
  }
  const int kq = tid % KQ, r0 = tid / KQ;
  int ih0[A_PASS], iw0[A_PASS], rowoff[A_PASS];
#pragma unroll
  for (int i = 0; i < A_PASS; ++i) {
    const int m = m0 + r0 + i * RPP;
    if (m < M) {
      const int n_ = m / OHW, rem = m - n_ * OHW;
      const int oh_ = rem / OW, ow_ = rem - oh_ * OW;
      ih0[i] = oh_ * stride - sg.pad_t;
      iw0[i] = ow_ * stride - sg.pad_l;
      rowoff[i] = ((n_ * H + ih0[i]) * W + iw0[i]) * ldx + a_coff;
    } else {
      ih0[i] = -0x40000000; iw0[i] = 0; rowoff[i] = 0;
    }
  }
  unsigned browoff[B_PASS];
#pragma unroll
  for (int j = 0; j < B_PASS; ++j) {
    const int n = n0 + r0 + j * RPP;
    browoff[j] = n < nmax ? (unsigned)n * ktotal * 2u : OOB;
  }

  int t_kh = 0, t_kw = 0, t_ci = 0;
  vec_t ra[A_PASS], rb[B_PASS];
  unsigned okbits = 0;      // FIN: which of the A vectors in flight lie inside the image
  int ld_c = 0;             // FIN: input channel (of the whole tensor) of this thread's A vectors in flight
  auto load_tiles = [&](int kt) {
    int khh, kww, tapoff;
    const int k = kt * BK + kq * VEC;
    const bool kok = k < ktotal;
    if (TAPU) {
      khh = t_kh; kww = t_kw;
      tapoff = (khh * W + kww) * ldx + t_ci + kq * VEC;
      ld_c = grp * cin + t_ci + kq * VEC;
      t_ci += BK;
      if (t_ci == cin) { t_ci = 0; if (++t_kw == kw) { t_kw = 0; ++t_kh; } }
    } else {
      const int tap = k / cin, ci = k - tap * cin;
      khh = tap / kw; kww = tap - khh * kw;
      tapoff = (khh * W + kww) * ldx + ci;
      ld_c = grp * cin + ci;
    }
    okbits = 0;
    unsigned aoff[A_PASS];
#pragma unroll
    for (int i = 0; i < A_PASS; ++i) {
      const int ih = ih0[i] + khh, iw = iw0[i] + kww;
      const bool ok = kok && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
      aoff[i] = ok ? (unsigned)(rowoff[i] + tapoff) * 2u : OOB;
      okbits |= ok ? (1u << i) : 0u;
    }
    if (args.a_nt) {
#pragma unroll
      for (int i = 0; i < A_PASS; ++i) ra[i] = VecH<VEC>::load_nt(xa, aoff[i]);
    } else {
#pragma unroll
      for (int i = 0; i < A_PASS; ++i) ra[i] = VecH<VEC>::load(xa, aoff[i]);
    }
#pragma unroll
    for (int j = 0; j < B_PASS; ++j)
      rb[j] = VecH<VEC>::load(wb, (kok && browoff[j] != OOB) ? browoff[j] + (unsigned)k * 2u : OOB);
  };
  auto store_tiles = [&](int buf) {
    _Float16* As = smem + buf * BUF;
    _Float16* Bs = As + BM * LDH;
    if (FIN) {            // act(GN(x)) of the vectors in flight; zero where the tap lies in the padding (SAME pads the ACTIVATED tensor)
      float2 t[VEC];
#pragma unroll
      for (int j = 0; j < VEC; ++j) t[j] = ntab[min(ld_c + j, 2047)];
#pragma unroll
      for (int i = 0; i < A_PASS; ++i) {
        vec_t v = ra[i];
        const bool ok = (okbits >> i) & 1u;
        if constexpr (FIN_RELU) {
          // ReLU commutes with the rounding to fp16 (monotone, 0 exact): fp16 in -> fp32 fma -> fp16 out is ONE mixed-precision
          // instruction per element (v_fma_mixlo / mixhi_f16), the ReLU one packed max per pair -- 1.5 operations per element
          // where convert, fma, max, convert are 4; that pass is as long as the tile's MFMAs otherwise
          static_assert(VEC % 2 == 0, "pairs of halfs");
          typedef uint32_t pairs_t __attribute__((ext_vector_type(VEC / 2)));
          pairs_t pv = __builtin_bit_cast(pairs_t, v);
#pragma unroll
          for (int j = 0; j < VEC / 2; ++j) pv[j] = fma_mix_pair(pv[j], t[2 * j].x, t[2 * j].y, t[2 * j + 1].x, t[2 * j + 1].y);
          const vec_t zero = {};
          v = ok ? __builtin_elementwise_max(__builtin_bit_cast(vec_t, pv), zero) : zero;
        } else {
#pragma unroll
          for (int j = 0; j < VEC; ++j) {
            const float z = rn::act_fwd(fmaf((float)v[j], t[j].x, t[j].y), args.fold.in_act);
            v[j] = ok ? (_Float16)z : (_Float16)0.f;
          }
        }
        ra[i] = v;
      }
    }
#pragma unroll
    for (int i = 0; i < A_PASS; ++i) *reinterpret_cast<vec_t*>(&As[(r0 + i * RPP) * LDH + kq * VEC]) = ra[i];
#pragma unroll
    for (int j = 0; j < B_PASS; ++j) *reinterpret_cast<vec_t*>(&Bs[(r0 + j * RPP) * LDH + kq * VEC]) = rb[j];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int l31 = lane & 31, half = lane >> 5;
  const int nk = (ktotal + BK - 1) / BK;
  auto mma_step = [&](const _Float16* As, const _Float16* Bs, int ks) {
    half8 a[TM], b[TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
      a[tm] = *reinterpret_cast<const half8*>(&As[(wm * (BM / WM) + tm * 32 + l31) * LDH + ks * 16 + half * 8]);
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
      b[tn] = *reinterpret_cast<const half8*>(&Bs[(wn * (BN / WN) + tn * 32 + l31) * LDH + ks * 16 + half * 8]);
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
  };
  if constexpr (PIPE == 1) {
    // ONE block of 8 waves per CU (128 accumulator registers per lane): nothing but the block's own instruction order hides its
    // memory operations, and a wave issues in order.  Here every K-step issues ONE memory operation behind each MFMA: the
    // fragment reads of the next K-step (two register slots), the LDS stores of tile t+1 (other buffer) and, right behind each
    // store, the global load of the same vector for tile t+2 (a full tile ahead of its store).  One barrier per K-tile; the
    // first fragments of the next tile are read behind it, under the last MFMAs of this one.  The order is pinned with
    // scheduling barriers (the compiler sinks reads to their uses and hoists stores otherwise); the loop body is branch-free:
    // loads past the last tile go out of range (zeros, no traffic), their stores land in the buffer nobody reads any more.
    // Measured on the cfg-5 head conv by leaving one kind of operation out (RN_F16_DBG_BUILD): no operand traffic at all
    // 1.6 - 1.7 PFLOP/s (the clock the chip holds under this load; bare 32x32x16 loops on random data: 1.66), no fragment reads
    // 1.33, no global loads 1.17, no LDS stores 1.56 -- the eight `ds_write_b128` of a K-tile are what is left to remove
    // (13 cycles of VGPR -> LDS transfer each, MI355X_MICROARCH.md LDS table; they cost the same 25 - 35 % in every order
    // tried).  Also built and measured, not kept: LDS-DMA (`buffer_load_dwordx4 ... lds` into XOR-swizzled unpadded rows, no
    // staging registers, no store instructions; correct, lanes out of range deposit zeros) -- the same time as this kernel,
    // the DMAs' issue cost replaces the stores'; and the two waves of a SIMD in opposite phases (memory phase / 16 MFMAs back
    // to back, four barriers per K-tile, wave row 1 one barrier behind): 10 % slower.
    static_assert(!FIN && TAPU && VEC == 8 && TM == 4 && TN == 2 && A_PASS == 4 && B_PASS == 4 && BK == 64, "PIPE: the 256 x 256 / 8-wave shape");
#define SB() __builtin_amdgcn_sched_barrier(0)
    half8 fa[2][TM], fb[2][TN];
    const int arow = (wm * (BM / WM) + l31) * LDH + half * 8, brow = BM * LDH + (wn * (BN / WN) + l31) * LDH + half * 8;
    auto rdA = [&](const _Float16* buf, int ks, int slot, int tm) {
      fa[slot][tm] = *reinterpret_cast<const half8*>(&buf[arow + tm * 32 * LDH + ks * 16]);
    };
    auto rdB = [&](const _Float16* buf, int ks, int slot, int tn) {
      fb[slot][tn] = *reinterpret_cast<const half8*>(&buf[brow + tn * 32 * LDH + ks * 16]);
    };
    auto mf = [&](int slot, int tm, int tn) {
      acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[slot][tm], fb[slot][tn], acc[tm][tn], 0, 0, 0);
    };
    // tile being loaded: block-uniform tap state -> (khh, kww, element offset of the tap, byte offset into a weight row)
    int l_kh = 0, l_kw = 0, l_tap = 0;
    unsigned l_kb = 0;
    bool l_in = true;
    auto prep = [&](int kt) {
      l_kh = t_kh; l_kw = t_kw;
      l_tap = (t_kh * W + t_kw) * ldx + t_ci + kq * VEC;
      l_kb = (unsigned)(kt * BK + kq * VEC) * 2u;
      l_in = kt < nk;
      t_ci += BK;
      if (t_ci == cin) { t_ci = 0; if (++t_kw == kw) { t_kw = 0; ++t_kh; } }
    };
    auto ldA = [&](int i) {
      const int ok = (int)((unsigned)(ih0[i] + l_kh) < (unsigned)H) & (int)((unsigned)(iw0[i] + l_kw) < (unsigned)W) & (int)l_in;
      ra[i] = VecH<VEC>::load(xa, ok ? (unsigned)(rowoff[i] + l_tap) * 2u : OOB);
    };
    auto ldB = [&](int j) {
      const int ok = (int)(browoff[j] != OOB) & (int)l_in;
      rb[j] = VecH<VEC>::load(wb, ok ? browoff[j] + l_kb : OOB);
    };
    const int st_off = r0 * LDH + kq * VEC;
    auto stA = [&](_Float16* buf, int i) { *reinterpret_cast<vec_t*>(&buf[st_off + i * RPP * LDH]) = ra[i]; };
    auto stB = [&](_Float16* buf, int j) { *reinterpret_cast<vec_t*>(&buf[BM * LDH + st_off + j * RPP * LDH]) = rb[j]; };

    prep(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) { ldA(i); ldB(i); }
#pragma unroll
    for (int i = 0; i < 4; ++i) { stA(smem, i); stB(smem, i); }
    prep(1);
#pragma unroll
    for (int i = 0; i < 4; ++i) { ldA(i); ldB(i); }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 4; ++t) rdA(smem, 0, 0, t);
    rdB(smem, 0, 0, 0); rdB(smem, 0, 0, 1);
    for (int kt = 0; kt < nk; ++kt) {
      const _Float16* Ac = smem + (kt & 1) * BUF;
      _Float16* An = smem + ((kt + 1) & 1) * BUF;
      prep(kt + 2);
      SB();
      // K-step 0 (slot 0) | fragments of K-step 1 -> slot 1 | vectors A0 A1 A2: tile t+1 -> LDS, tile t+2 <- memory
      mf(0, 0, 0); SB(); if constexpr (!(DBG & 4)) rdA(Ac, 1, 1, 0); if constexpr (!(DBG & 2)) stA(An, 0); SB();
      mf(0, 0, 1); SB(); if constexpr (!(DBG & 4)) rdB(Ac, 1, 1, 0); if constexpr (!(DBG & 1)) ldA(0); SB();
      mf(0, 1, 0); SB(); if constexpr (!(DBG & 4)) rdB(Ac, 1, 1, 1); if constexpr (!(DBG & 2)) stA(An, 1); SB();
      mf(0, 1, 1); SB(); if constexpr (!(DBG & 4)) rdA(Ac, 1, 1, 1); if constexpr (!(DBG & 1)) ldA(1); SB();
      mf(0, 2, 0); SB(); if constexpr (!(DBG & 4)) rdA(Ac, 1, 1, 2); if constexpr (!(DBG & 2)) stA(An, 2); SB();
      mf(0, 2, 1); SB(); if constexpr (!(DBG & 4)) rdA(Ac, 1, 1, 3); if constexpr (!(DBG & 1)) ldA(2); SB();
      mf(0, 3, 0); SB();
      mf(0, 3, 1); SB();
      // K-step 1 (slot 1) | K-step 2 -> slot 0 | A3 B0 B1
      mf(1, 0, 0); SB(); if constexpr (!(DBG & 4)) rdA(Ac, 2, 0, 0); if constexpr (!(DBG & 2)) stA(An, 3); SB();
      mf(1, 0, 1); SB(); if constexpr (!(DBG & 4)) rdB(Ac, 2, 0, 0); if constexpr (!(DBG & 1)) ldA(3); SB();
      mf(1, 1, 0); SB(); if constexpr (!(DBG & 4)) rdB(Ac, 2, 0, 1); if constexpr (!(DBG & 2)) stB(An, 0); SB();
      mf(1, 1, 1); SB(); if constexpr (!(DBG & 4)) rdA(Ac, 2, 0, 1); if constexpr (!(DBG & 1)) ldB(0); SB();
      mf(1, 2, 0); SB(); if constexpr (!(DBG & 4)) rdA(Ac, 2, 0, 2); if constexpr (!(DBG & 2)) stB(An, 1); SB();
      mf(1, 2, 1); SB(); if constexpr (!(DBG & 4)) rdA(Ac, 2, 0, 3); if constexpr (!(DBG & 1)) ldB(1); SB();
      mf(1, 3, 0); SB();
      mf(1, 3, 1); SB();
      // K-step 2 (slot 0) | K-step 3 -> slot 1 | B2 B3
      mf(0, 0, 0); SB(); if constexpr (!(DBG & 4)) rdA(Ac, 3, 1, 0); if constexpr (!(DBG & 2)) stB(An, 2); SB();
      mf(0, 0, 1); SB(); if constexpr (!(DBG & 4)) rdB(Ac, 3, 1, 0); if constexpr (!(DBG & 1)) ldB(2); SB();
      mf(0, 1, 0); SB(); if constexpr (!(DBG & 4)) rdB(Ac, 3, 1, 1); if constexpr (!(DBG & 2)) stB(An, 3); SB();
      mf(0, 1, 1); SB(); if constexpr (!(DBG & 4)) rdA(Ac, 3, 1, 1); if constexpr (!(DBG & 1)) ldB(3); SB();
      mf(0, 2, 0); SB(); if constexpr (!(DBG & 4)) rdA(Ac, 3, 1, 2); SB();
      mf(0, 2, 1); SB(); if constexpr (!(DBG & 4)) rdA(Ac, 3, 1, 3); SB();
      mf(0, 3, 0); SB();
      mf(0, 3, 1); SB();
      // K-step 3 (slot 1) | barrier: this buffer's fragments are all in registers, the other buffer is complete | K-step 0 of
      // the next tile -> slot 0
      mf(1, 0, 0); SB();
      mf(1, 0, 1); SB();
      if constexpr (!(DBG & 8)) __syncthreads(); SB();
      mf(1, 1, 0); SB(); if constexpr (!(DBG & 4)) rdA(An, 0, 0, 0); SB();
      mf(1, 1, 1); SB(); if constexpr (!(DBG & 4)) rdB(An, 0, 0, 0); SB();
      mf(1, 2, 0); SB(); if constexpr (!(DBG & 4)) rdB(An, 0, 0, 1); SB();
      mf(1, 2, 1); SB(); if constexpr (!(DBG & 4)) rdA(An, 0, 0, 1); SB();
      mf(1, 3, 0); SB(); if constexpr (!(DBG & 4)) rdA(An, 0, 0, 2); SB();
      mf(1, 3, 1); SB(); if constexpr (!(DBG & 4)) rdA(An, 0, 0, 3); SB();
    }
#undef SB
  } else if constexpr (PIPE == 2) {
    // The pipelined loop above with the WEIGHT operand kept out of LDS: rn_pack_weights_f16 leaves a second copy of the kernel
    // in MFMA-fragment order -- Wf[cout / 32][K / 16][lane][8 halfs]: what lane l of a wave feeds `v_mfma_f32_32x32x16_f16`
    // for 32 output channels x 16 k is 16 contiguous bytes, a wave's fragment 1 KB contiguous -- so a B fragment is ONE global
    // load straight into the registers the MFMA reads (L2-resident data; the two wave rows of a tile fetch the same lines).
    // Per K-tile and wave that removes 4 of the 8 `ds_write_b128` (what the leave-one-out builds name as the loop's cost), 8 of
    // the 24 fragment reads and the B tiles' LDS; the 8 fragments of tile t+1 are requested as the MFMAs of tile t release
    // their registers, a full tile ahead of their use.
    static_assert((!FIN || FIN_RELU) && TAPU && VEC == 8 && TM == 4 && TN == 2 && A_PASS == 4 && BK == 64, "PIPE: the 256 x 256 / 8-wave shape");
#define SB() __builtin_amdgcn_sched_barrier(0)
    half8 fa[2][TM], fbg[4][TN];
    const int arow = (wm * (BM / WM) + l31) * LDH + half * 8;
    auto rdA = [&](const _Float16* buf, int ks, int slot, int tm) {
      fa[slot][tm] = *reinterpret_cast<const half8*>(&buf[arow + tm * 32 * LDH + ks * 16]);
    };
    auto mf = [&](int slot, int ks, int tm, int tn) {
      acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[slot][tm], fbg[ks][tn], acc[tm][tn], 0, 0, 0);
    };
    int l_kh = 0, l_kw = 0, l_tap = 0;
    bool l_in = true;
    auto prep = [&](int kt) {
      l_kh = t_kh; l_kw = t_kw;
      l_tap = (t_kh * W + t_kw) * ldx + t_ci + kq * VEC;
      l_in = kt < nk;
      t_ci += BK;
      if (t_ci == cin) { t_ci = 0; if (++t_kw == kw) { t_kw = 0; ++t_kh; } }
    };
    auto ldA = [&](int i) {
      const int ok = (int)((unsigned)(ih0[i] + l_kh) < (unsigned)H) & (int)((unsigned)(iw0[i] + l_kw) < (unsigned)W) & (int)l_in;
      ra[i] = VecH<VEC>::load(xa, ok ? (unsigned)(rowoff[i] + l_tap) * 2u : OOB);
    };
    const int st_off = r0 * LDH + kq * VEC;
    // FIN (1 x 1 convs only: the host checks; a tile is then the channels [64 kt, 64 kt + 64) of 256 pixels, no padding taps):
    // ReLU(GN(x)) of a vector between its staging registers and LDS -- the table entries of the tile on its way to LDS (two float4
    // pairs per lane, read once per K-tile), one mixed-precision fma per element, one packed max per pair
    float4 tq[FIN ? 4 : 1];
    auto ldT = [&](int kt) {
      if constexpr (FIN) {
        const float4* tp = reinterpret_cast<const float4*>(ntab + min(kt * BK + kq * VEC, 2040));
#pragma unroll
        for (int j = 0; j < 4; ++j) tq[j] = tp[j];
      }
    };
    auto stA = [&](_Float16* buf, int i) {
      if constexpr (FIN) {
        typedef uint32_t pairs_t __attribute__((ext_vector_type(4)));
        pairs_t pv = __builtin_bit_cast(pairs_t, ra[i]);
#pragma unroll
        for (int j = 0; j < 4; ++j) pv[j] = fma_mix_pair(pv[j], tq[j].x, tq[j].y, tq[j].z, tq[j].w);
        const vec_t zero = {};
        ra[i] = __builtin_elementwise_max(__builtin_bit_cast(vec_t, pv), zero);
      }
      *reinterpret_cast<vec_t*>(&buf[st_off + i * RPP * LDH]) = ra[i];
    };
    // fragment (32-channel block nt, 16-k step kidx) starts at ((nt * K/16 + kidx) * 64 lanes) * 16 bytes; channel blocks past
    // the (padded) kernel are out of the descriptor's range: zeros
    const int ks16 = ktotal >> 4;
    const int nblk = (cout + 31) >> 5;
    const __amdgpu_buffer_rsrc_t wfr = make_rsrc(sg.wf, (unsigned)nblk * 32u * (unsigned)ktotal * 2u);
    unsigned bbase[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int nt = (n0 >> 5) + wn * (BN / WN / 32) + tn;
      bbase[tn] = nt < nblk ? ((unsigned)nt * (unsigned)ks16 * 64u + (unsigned)lane) * 16u : OOB;
    }
    auto ldBf = [&](int kt, int ks, int tn) {   // fragments of tile kt (past the last tile: not requested)
      const unsigned vo = kt < nk ? bbase[tn] : OOB;
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wfr, vo, (kt * 4 + ks) * 1024, 0);
      fbg[ks][tn] = __builtin_bit_cast(half8, v);
    };

    prep(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) ldA(i);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { ldBf(0, ks, 0); ldBf(0, ks, 1); }
    ldT(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) stA(smem, i);
    prep(1);
#pragma unroll
    for (int i = 0; i < 4; ++i) ldA(i);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 4; ++t) rdA(smem, 0, 0, t);
    constexpr int ABUF = BM * LDH;                  // halfs per A buffer
    for (int kt = 0; kt < nk; ++kt) {
      const _Float16* Ac = smem + (kt & 1) * ABUF;
      _Float16* An = smem + ((kt + 1) & 1) * ABUF;
      prep(kt + 2);
      ldT(kt + 1);
      SB();
      // K-step 0 (slot 0) | A fragments of K-step 1 -> slot 1 | A0 A1: tile t+1 -> LDS, tile t+2 <- memory | B fragments (t+1, 0)
      mf(0, 0, 0, 0); SB(); rdA(Ac, 1, 1, 0); stA(An, 0); SB();
      mf(0, 0, 0, 1); SB(); rdA(Ac, 1, 1, 1); ldA(0); SB();
      mf(0, 0, 1, 0); SB(); rdA(Ac, 1, 1, 2); stA(An, 1); SB();
      mf(0, 0, 1, 1); SB(); rdA(Ac, 1, 1, 3); ldA(1); SB();
      mf(0, 0, 2, 0); SB();
      mf(0, 0, 2, 1); SB();
      mf(0, 0, 3, 0); SB(); ldBf(kt + 1, 0, 0); SB();
      mf(0, 0, 3, 1); SB(); ldBf(kt + 1, 0, 1); SB();
      // K-step 1 (slot 1) | K-step 2 -> slot 0 | A2 A3 | B fragments (t+1, 1)
      mf(1, 1, 0, 0); SB(); rdA(Ac, 2, 0, 0); stA(An, 2); SB();
      mf(1, 1, 0, 1); SB(); rdA(Ac, 2, 0, 1); ldA(2); SB();
      mf(1, 1, 1, 0); SB(); rdA(Ac, 2, 0, 2); stA(An, 3); SB();
      mf(1, 1, 1, 1); SB(); rdA(Ac, 2, 0, 3); ldA(3); SB();
      mf(1, 1, 2, 0); SB();
      mf(1, 1, 2, 1); SB();
      mf(1, 1, 3, 0); SB(); ldBf(kt + 1, 1, 0); SB();
      mf(1, 1, 3, 1); SB(); ldBf(kt + 1, 1, 1); SB();
      // K-step 2 (slot 0) | K-step 3 -> slot 1 | B fragments (t+1, 2)
      mf(0, 2, 0, 0); SB(); rdA(Ac, 3, 1, 0); SB();
      mf(0, 2, 0, 1); SB(); rdA(Ac, 3, 1, 1); SB();
      mf(0, 2, 1, 0); SB(); rdA(Ac, 3, 1, 2); SB();
      mf(0, 2, 1, 1); SB(); rdA(Ac, 3, 1, 3); SB();
      mf(0, 2, 2, 0); SB();
      mf(0, 2, 2, 1); SB();
      mf(0, 2, 3, 0); SB(); ldBf(kt + 1, 2, 0); SB();
      mf(0, 2, 3, 1); SB(); ldBf(kt + 1, 2, 1); SB();
      // K-step 3 (slot 1) | barrier: this buffer's fragments are all in registers, the other buffer is complete | K-step 0 of
      // the next tile -> slot 0 | B fragments (t+1, 3)
      mf(1, 3, 0, 0); SB();
      mf(1, 3, 0, 1); SB();
      __syncthreads(); SB();
      mf(1, 3, 1, 0); SB(); rdA(An, 0, 0, 0); SB();
      mf(1, 3, 1, 1); SB(); rdA(An, 0, 0, 1); SB();
      mf(1, 3, 2, 0); SB(); rdA(An, 0, 0, 2); SB();
      mf(1, 3, 2, 1); SB(); rdA(An, 0, 0, 3); SB();
      mf(1, 3, 3, 0); SB(); ldBf(kt + 1, 3, 0); SB();
      mf(1, 3, 3, 1); SB(); ldBf(kt + 1, 3, 1); SB();
    }
#undef SB
  } else {
    load_tiles(0);
    for (int kt = 0; kt < nk; ++kt) {
      store_tiles(0);
      __syncthreads();
      if (kt + 1 < nk) load_tiles(kt + 1);
#pragma unroll
      for (int ks = 0; ks < BK / 16; ++ks) mma_step(smem, smem + BM * LDH, ks);
      __syncthreads();
    }
  }

  // epilogue: C/D map col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  if (!args.out_f32 && (cout_g & 7) == 0) {
    // fp16 output: stage the tile through LDS (the operand tiles are dead) and write it as whole 16-byte row pieces --
    // 8 global stores per thread instead of 64 two-byte ones, every row of the tile a contiguous 2*BN-byte run.
    constexpr int LDS_ROW = BN + 8;  // halfs; +8 keeps the two half-waves (rows r, r+4) on different banks
    // tiles taller than the operand LDS can stage go in WM passes of BM / WM rows (the rows of one wave row)
    constexpr int CAP = (PIPE ? 2 : 1) * BUF;   // (the host passes 2 * BUF halfs of dynamic LDS to both pipelined variants)   // halfs of LDS the staged tile may use
    constexpr int PASSES = (BM * LDS_ROW <= CAP) ? 1 : WM;
    constexpr int ROWS = BM / PASSES;
    static_assert(ROWS * LDS_ROW <= CAP, "staging tile must fit the operand tiles' LDS");
    _Float16* Cs = smem;
    constexpr int VPR = BN / 8;  // 16-byte vectors per tile row
    const __amdgpu_buffer_rsrc_t ys = make_rsrc(sg.y, (unsigned)M * (unsigned)cout * 2u);
    float st1[TN], st2[TN];    // FOUT: this lane's column sums of y and y^2 (of the values as stored)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) st1[tn] = st2[tn] = 0.f;
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      if (PASSES == 1 || wm == ps) {
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
          const int lc = wn * (BN / WN) + tn * 32 + l31;
          const int col = n0 + lc;
          const float bv = (sg.bias != nullptr && col < nmax) ? sg.bias[col] : 0.f;
#pragma unroll
          for (int tm = 0; tm < TM; ++tm) {
            const int rb = (PASSES == 1 ? wm * (BM / WM) : 0) + tm * 32 + 4 * half;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const _Float16 hv = (_Float16)(acc[tm][tn][r] + bv);
              Cs[(rb + (r & 3) + 8 * (r >> 2)) * LDS_ROW + lc] = hv;
              if (FOUT) { const float fv = (float)hv; st1[tn] += fv; st2[tn] = fmaf(fv, fv, st2[tn]); }
            }
          }
        }
      }
      __syncthreads();
#pragma unroll
      for (int v = tid; v < ROWS * VPR; v += T) {
        const int row = v / VPR, cv = v - row * VPR;
        const int col = n0 + cv * 8;
        const u32x4 d = *reinterpret_cast<const u32x4*>(&Cs[row * LDS_ROW + cv * 8]);
        const bool ok = col < nmax;  // cout_g % 8 == 0: a vector never straddles a group
        __builtin_amdgcn_raw_buffer_store_b128(d, ys, ok ? ((unsigned)(m0 + ps * ROWS + row) * (unsigned)cout + (unsigned)col) * 2u : OOB,
                                               0, 0);
      }
      if (ps + 1 < PASSES) __syncthreads();
    }
    if (FOUT && sg.partial != nullptr) {   // the two lane halves, then the WM wave rows -> one (sum, sum of squares) per column of the tile
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        st1[tn] += __shfl_xor(st1[tn], 32, 64);
        st2[tn] += __shfl_xor(st2[tn], 32, 64);
        if (lane < 32) {
          const int lc = wn * (BN / WN) + tn * 32 + l31;
          sred[(wm * BN + lc) * 2 + 0] = st1[tn];
          sred[(wm * BN + lc) * 2 + 1] = st2[tn];
        }
      }
      __syncthreads();
      if (tid < BN && n0 + tid < nmax) {
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int w_ = 0; w_ < WM; ++w_) { t1 += sred[(w_ * BN + tid) * 2 + 0]; t2 += sred[(w_ * BN + tid) * 2 + 1]; }
        float* p1 = sg.partial + (size_t)tile_m * cout + n0 + tid;
        p1[0] = t1;
        p1[(size_t)sg.prows * cout] = t2;
      }
    }
    return;
  }
  const unsigned esz = args.out_f32 ? 4u : 2u;
  const __amdgpu_buffer_rsrc_t ys = make_rsrc(sg.y, (unsigned)M * (unsigned)cout * esz);
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int col = n0 + wn * (BN / WN) + tn * 32 + l31;
    const bool cok = col < nmax;
    const float bv = (sg.bias != nullptr && cok) ? sg.bias[col] : 0.f;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const int rbase = m0 + wm * (BM / WM) + tm * 32 + 4 * half;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rbase + (r & 3) + 8 * (r >> 2);
        const unsigned idx = (unsigned)row * (unsigned)cout + (unsigned)col;
        const float v = acc[tm][tn][r] + bv;
        if (args.out_f32) {
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ys, cok ? idx * 4u : OOB, 0, 0);
        } else {
          const _Float16 hv = (_Float16)v;
          __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, hv), ys, cok ? idx * 2u : OOB, 0, 0);
        }
      }
    }
  }
}

// Wt[co][k] (half) <- w[k][co] (fp32 HWIO flattened: k = (kh,kw,ci_g)); tiny, runs once per model
__global__ void pack_weights_kernel(const float* __restrict__ w, _Float16* __restrict__ wt, int ktotal, int cout) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)ktotal * cout) return;
  const int co = (int)(i / ktotal), k = (int)(i - (int64_t)co * ktotal);
  wt[i] = (_Float16)w[(size_t)k * cout + co];
}

// Wf[cout/32 (padded)][K/16][lane][8] <- w: lane l of a fragment holds channel (l & 31) of the block and k = 16 step + 8 (l >> 5) .. + 7
__global__ void pack_weights_frag_kernel(const float* __restrict__ w, _Float16* __restrict__ wf, int ktotal, int cout, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int e = (int)(i & 7), l = (int)((i >> 3) & 63);
  const int64_t f = i >> 9;                      // fragment index = nt * (K/16) + kstep
  const int ks16 = ktotal >> 4;
  const int nt = (int)(f / ks16), kstep = (int)(f - (int64_t)nt * ks16);
  const int n = nt * 32 + (l & 31), k = kstep * 16 + (l >> 5) * 8 + e;
  wf[i] = n < cout ? (_Float16)w[(size_t)k * cout + n] : (_Float16)0.f;
}

// =====================================================================================================================
// Grouped 3 x 3 / stride 1 conv with as many input as output channels per group (ResNeXt's conv 2: 32 groups of 4 / 8 / 16 / 32):
// the implicit-GEMM kernel above runs one group per N-tile -- a [128 x 9 cin_g] x [9 cin_g x cout_g] product whose operand
// gather reads 8 - 64 bytes per (pixel, tap) and whose 32-wide N-tile holds 4 - 32 useful columns: 2.0 - 2.5 x the time its
// bytes take.  Here a block owns a 16 x 16 tile of output pixels of one sample and one SUPER-GROUP of 32 consecutive channels
// (= 8 / 4 / 2 / 1 groups): the 18 x 18 x 32 input patch goes to LDS ONCE (64 contiguous bytes per pixel; 1.27 x the tile, not
// 9 x), the super-group's kernel is block-diagonal over its groups (zeros elsewhere: the matrix cores are not the bound) and
// sits in registers in MFMA-fragment order (18 fragments per lane from rn_pack_weights_f16's third copy), and the nine taps are
// nine shifted fragment reads of the same patch: 36 MFMAs and 36 `ds_read_b128` per wave, no barrier inside.
//   wave w: output rows [4 w, 4 w + 4) of the tile = two 32-pixel M-tiles (2 rows x 16 columns each)
//   patch pixel stride 80 bytes: the 16 pixels of a row a half-wave reads land on 16 different 16-byte bank groups
// The epilogue is the one of the kernel above (tile staged through LDS, 16-byte stores, statistics of the values as stored).
// =====================================================================================================================
struct G3Args {
  const _Float16* x; const _Float16* ws; _Float16* y; float* partial;
  int n, h, w, c, tiles_w, tiles_per_sample, nsg, prows;
  // FIN: x is the RAW output of the conv in front; act(GN(x)) is formed between the patch's staging registers and LDS -- once per
  // patch element (1.27 x the tile; the implicit GEMM would do it once per tap) -- and the GroupNorm's apply pass disappears
  const float* in_mean; const float* in_rstd; const float* in_gamma; const float* in_beta;
  int in_groups, in_cpg, in_act;
};
constexpr int G3_PST = 40;      // pixel stride of the LDS patch and of the staged output tile (halfs): 80 bytes

// S = 1: 16 x 16 output pixels per block, wave w the rows [4 w, 4 w + 4) = two 32-pixel M-tiles; 18 x 18 input patch.
// S = 2 (the first bottleneck of a stage; even maps: SAME pads bottom / right only): 8 x 16 output pixels, wave w the rows
// [2 w, 2 w + 2) = one M-tile; 17 x 33 patch with the even and the odd columns of a row in two runs, so that the 16 pixels a
// half-wave reads for one tap (input columns 2 x + kw) are neighbours in LDS as they are for S = 1.
template <int S>
struct G3Geom {
  static constexpr int TH = S == 1 ? 16 : 8, TW = 16;
  static constexpr int PH = (TH - 1) * S + 3, PW = (TW - 1) * S + 3;          // 18 x 18 | 17 x 33
  static constexpr int ROW = S == 1 ? PW : 34;                                 // LDS entries per patch row
  static constexpr int NT = S == 1 ? 2 : 1;                                    // M-tiles per wave
  static constexpr int NCH = PH * PW * 4, PER = (NCH + 255) / 256;             // 16-byte chunks of the patch, per thread
  static __device__ __forceinline__ int lds_pix(int py, int px) { return S == 1 ? py * ROW + px : py * ROW + (px & 1) * 17 + (px >> 1); }
};

template <bool FOUT, int FIN, int S>       // FIN: 0 none, 1 any activation, 2 ReLU (one mixed-precision fma per element + a packed max per pair)
__global__ __launch_bounds__(256) void conv3x3_sg32_f16_kernel(const G3Args a) {
  typedef G3Geom<S> G;
  __shared__ __attribute__((aligned(16))) _Float16 patch[G::PH * G::ROW * G3_PST];     // 25.9 | 46.2 KB; later: the staged output tile
  __shared__ float sred[FOUT ? 4 * 32 * 2 : 1];
  __shared__ __attribute__((aligned(16))) float2 ntab[FIN ? 32 : 1];                  // (scale, shift) of the super-group's channels
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bid = rn::xcd_remap(blockIdx.x, gridDim.x);
  const int sg = bid % a.nsg;                        // super-groups of one tile are neighbours: they share the pixels' cache lines
  const int t = bid / a.nsg;
  const int tile = t % a.tiles_per_sample, smp = t / a.tiles_per_sample;
  const int ty0 = (tile / a.tiles_w) * G::TH, tx0 = (tile % a.tiles_w) * G::TW;      // output pixels
  const int H = a.h, W = a.w, C = a.c, OH = H / S, OW = W / S;
  const int pad = S == 1 ? 1 : 0;                    // SAME, top / left (even maps)
  const __amdgpu_buffer_rsrc_t xs = make_rsrc(a.x + (size_t)smp * H * W * C, (unsigned)H * W * C * 2u);
  // ---- everything from memory first: the patch (6 | 9 chunks of 16 bytes per thread), the 18 weight fragments
  half8 pv[G::PER];
  unsigned okbits = 0;
#pragma unroll
  for (int j = 0; j < G::PER; ++j) {
    const int e = tid + j * 256;
    const int pix = min(e, G::NCH - 1) >> 2, q = e & 3;
    const int py = pix / G::PW, px = pix - py * G::PW;
    const int gy = ty0 * S - pad + py, gx = tx0 * S - pad + px;
    const bool ok = e < G::NCH && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
    pv[j] = VecH<8>::load(xs, ok ? (unsigned)((gy * W + gx) * C + sg * 32 + q * 8) * 2u : OOB);     // (SAME padding: zeros)
    okbits |= ok ? (1u << j) : 0u;
  }
  if (FIN && tid < 32) {
    const int c = sg * 32 + tid, g_ = c / a.in_cpg;
    const float sc = a.in_rstd[smp * a.in_groups + g_] * a.in_gamma[c];
    ntab[tid] = make_float2(sc, a.in_beta[c] - a.in_mean[smp * a.in_groups + g_] * sc);
  }
  half8 bw[9][2];
  const half8* wsp = reinterpret_cast<const half8*>(a.ws) + (size_t)sg * 18 * 64 + lane;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) bw[tap][ks] = wsp[(tap * 2 + ks) * 64];
  if (FIN) {
    __syncthreads();
    // a thread's chunks all sit at the same channel offset (256 % 4 == 0): its eight (scale, shift) pairs once.  Padding taps stay
    // zero: SAME pads the ACTIVATED tensor
    const float4* tp = reinterpret_cast<const float4*>(ntab + (tid & 3) * 8);
    const float4 t0 = tp[0], t1 = tp[1], t2 = tp[2], t3 = tp[3];
    const half8 zero = {};
#pragma unroll
    for (int j = 0; j < G::PER; ++j) {
      const bool ok = (okbits >> j) & 1u;
      if (FIN == 2) {
        typedef uint32_t pairs_t __attribute__((ext_vector_type(4)));
        pairs_t p4 = __builtin_bit_cast(pairs_t, pv[j]);
        p4[0] = fma_mix_pair(p4[0], t0.x, t0.y, t0.z, t0.w); p4[1] = fma_mix_pair(p4[1], t1.x, t1.y, t1.z, t1.w);
        p4[2] = fma_mix_pair(p4[2], t2.x, t2.y, t2.z, t2.w); p4[3] = fma_mix_pair(p4[3], t3.x, t3.y, t3.z, t3.w);
        pv[j] = ok ? __builtin_elementwise_max(__builtin_bit_cast(half8, p4), zero) : zero;
      } else {
        const float scv[8] = {t0.x, t0.z, t1.x, t1.z, t2.x, t2.z, t3.x, t3.z}, shv[8] = {t0.y, t0.w, t1.y, t1.w, t2.y, t2.w, t3.y, t3.w};
        half8 v = pv[j];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = ok ? (_Float16)rn::act_fwd(fmaf((float)v[i], scv[i], shv[i]), a.in_act) : (_Float16)0.f;
        pv[j] = v;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < G::PER; ++j) {
    const int e = tid + j * 256;
    if (e < G::NCH) {
      const int pix = e >> 2;
      const int py = pix / G::PW, px = pix - py * G::PW;
      *reinterpret_cast<half8*>(&patch[G::lds_pix(py, px) * G3_PST + (e & 3) * 8]) = pv[j];
    }
  }
  __syncthreads();
  f32x16 acc[G::NT];
#pragma unroll
  for (int i = 0; i < G::NT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const int l31 = lane & 31, khalf = lane >> 5;
  const int ry = l31 >> 4, rx = l31 & 15;
  const int oy = (G::TH / 4) * wave + ry;             // first M-tile's output row of this lane (the second one: + 2)
#pragma unroll
  for (int kh = 0; kh < 3; ++kh)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int i = 0; i < G::NT; ++i) {
          const int lp = G::lds_pix((oy + 2 * i) * S + kh, rx * S + kw);
          const half8 av = *reinterpret_cast<const half8*>(&patch[lp * G3_PST + ks * 16 + khalf * 8]);
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bw[kh * 3 + kw][ks], acc[i], 0, 0, 0);
        }
      }
  __syncthreads();                                 // the patch is dead: the staged output tile takes its place
  // C/D map: col (channel) = lane & 31, pixel of the M-tile = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
  constexpr int WPX = 32 * G::NT;                  // output pixels per wave
  _Float16* cs = patch + wave * (WPX * G3_PST);    // [pixels of the wave][32 channels], pixel stride 80 bytes
  float st1 = 0.f, st2 = 0.f;
#pragma unroll
  for (int i = 0; i < G::NT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const _Float16 hv = (_Float16)acc[i][r];
      cs[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf) * G3_PST + l31] = hv;
      if (FOUT) { const float fv = (float)hv; st1 += fv; st2 = fmaf(fv, fv, st2); }
    }
  __builtin_amdgcn_s_waitcnt(0xc07f);              // lgkmcnt(0): the wave's own stores have landed (a wave reads only its own region)
  __builtin_amdgcn_wave_barrier();
  const __amdgpu_buffer_rsrc_t ys = make_rsrc(a.y + (size_t)smp * OH * OW * C, (unsigned)OH * OW * C * 2u);
#pragma unroll
  for (int j = 0; j < WPX / 16; ++j) {
    const int e = lane + j * 64;                   // chunk of the wave's pixels x 4 x 16 bytes
    const int p = e >> 2, q = e & 3;
    const int gy = ty0 + (G::TH / 4) * wave + (p >> 4), gx = tx0 + (p & 15);
    const u32x4 d = *reinterpret_cast<const u32x4*>(&cs[p * G3_PST + q * 8]);
    __builtin_amdgcn_raw_buffer_store_b128(d, ys, (unsigned)((gy * OW + gx) * C + sg * 32 + q * 8) * 2u, 0, 0);
  }
  if (FOUT) {
    st1 += __shfl_xor(st1, 32, 64);
    st2 += __shfl_xor(st2, 32, 64);
    if (lane < 32) { sred[(wave * 32 + lane) * 2 + 0] = st1; sred[(wave * 32 + lane) * 2 + 1] = st2; }
    __syncthreads();
    if (tid < 32) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int w_ = 0; w_ < 4; ++w_) { t1 += sred[(w_ * 32 + tid) * 2 + 0]; t2 += sred[(w_ * 32 + tid) * 2 + 1]; }
      float* p1 = a.partial + (size_t)(smp * a.tiles_per_sample + tile) * C + sg * 32 + tid;
      p1[0] = t1;
      p1[(size_t)a.prows * C] = t2;
    }
  }
}

// Ws[cout / 32][9 taps][2 k-steps][lane][8]: the block-diagonal 32 x 32 kernel of a super-group in MFMA-fragment order -- lane l
// holds output channel (l & 31) and input channels 16 step + 8 (l >> 5) .. + 7 of the super-group; zero where the two channels
// belong to different groups (groups of cin_g input AND cin_g output channels)
__global__ void pack_weights_sg_kernel(const float* __restrict__ w, _Float16* __restrict__ ws, int cin_g, int cout, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int e = (int)(i & 7), l = (int)((i >> 3) & 63);
  const int64_t f = i >> 9;
  const int ks = (int)(f & 1), tap = (int)((f >> 1) % 9), sg = (int)(f / 18);
  const int k = ks * 16 + (l >> 5) * 8 + e, n = l & 31;
  const int ci = sg * 32 + k, co = sg * 32 + n;
  const bool same = ci / cin_g == co / cin_g;
  ws[i] = same ? (_Float16)w[(size_t)(tap * cin_g + ci % cin_g) * cout + co] : (_Float16)0.f;
}

__global__ void cast_f32_to_f16_kernel(const float* __restrict__ x, _Float16* __restrict__ y, int64_t count) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = (_Float16)x[i];
}

// image [pixels][3] fp32 -> [pixels][4] fp16 (4th channel 0): lets the 7x7/2 stem use 8-byte fp16 gathers
__global__ void pad_cast_rgb_kernel(const float* __restrict__ x, _Float16* __restrict__ y, int64_t pixels) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < pixels; i += (int64_t)gridDim.x * blockDim.x) {
    half4 v;
    v.x = (_Float16)x[i * 3]; v.y = (_Float16)x[i * 3 + 1]; v.z = (_Float16)x[i * 3 + 2]; v.w = (_Float16)0.f;
    *reinterpret_cast<half4*>(y + i * 4) = v;
  }
}

struct TileCfg { int bm, bn; };
// 4, 5: large tiles for compute-heavy launches (the 3x3 head convs of a 1024^2 batch of 16): per MFMA fewer LDS
// fragment reads (wave tile 128x64: 6 reads per 8 MFMAs instead of 4 per 4)
const TileCfg kCfgs[6] = {{128, 128}, {128, 64}, {64, 64}, {128, 32}, {256, 128}, {256, 256}};
const double kT0[4] = {700, 550, 430, 430}, kT1[4] = {1780, 1000, 515, 560};

}  // namespace

namespace {
// the fragment-ordered copy starts this many halfs behind Wt (256-byte aligned); 0: there is none (K % 16 != 0)
inline int64_t frag_offset_halfs(int64_t ktotal, int64_t cout) { return (ktotal & 15) ? 0 : (int64_t)(rn::align_up((size_t)(ktotal * cout * 2), 256) / 2); }
inline int64_t frag_halfs(int64_t ktotal, int64_t cout) { return (ktotal & 15) ? 0 : ((cout + 31) / 32) * 32 * ktotal; }
// the super-group copy of a 3 x 3 kernel with 4 / 8 / 16 / 32 input channels per group (pack_weights_sg_kernel; read as a grouped
// kernel with as many output channels per group) starts 256-byte aligned behind whatever precedes it; 0: there is none
inline bool sg_shape(int kh, int kw, int cin_g, int cout) {
  return kh == 3 && kw == 3 && (cin_g == 4 || cin_g == 8 || cin_g == 16 || cin_g == 32) && cout % 32 == 0;
}
inline int64_t sg_offset_halfs(int kh, int kw, int cin_g, int cout) {
  if (!sg_shape(kh, kw, cin_g, cout)) return 0;
  const int64_t k = (int64_t)kh * kw * cin_g, off = frag_offset_halfs(k, cout);
  const int64_t end = off ? off + frag_halfs(k, cout) : k * cout;
  return (int64_t)(rn::align_up((size_t)(end * 2), 256) / 2);
}
inline int64_t sg_halfs(int cout) { return (int64_t)(cout / 32) * 18 * 64 * 8; }
}  // namespace

extern "C" size_t rn_pack_weights_f16_bytes(int kh, int kw, int cin_g, int cout) {
  if (kh < 1 || kw < 1 || cin_g < 1 || cout < 1) return 0;
  const int64_t k = (int64_t)kh * kw * cin_g;
  const int64_t off = frag_offset_halfs(k, cout);
  if (const int64_t so = sg_offset_halfs(kh, kw, cin_g, cout)) return (size_t)(so + sg_halfs(cout)) * 2;
  return (size_t)(off ? (off + frag_halfs(k, cout)) * 2 : k * cout * 2);
}

extern "C" int rn_pack_weights_f16(const float* w, void* wt, int kh, int kw, int cin_g, int cout, rn_stream_t stream) {
  RN_CHECK_ARG(w && wt && kh >= 1 && kw >= 1 && cin_g >= 1 && cout >= 1, "pack_weights_f16: bad argument");
  const int64_t total = (int64_t)kh * kw * cin_g * cout;
  hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)rn::ceil_div64(total, 256)), dim3(256), 0, (hipStream_t)stream, w,
                     (_Float16*)wt, kh * kw * cin_g, cout);
  RN_LAUNCH_CHECK();
  const int64_t k = (int64_t)kh * kw * cin_g, off = frag_offset_halfs(k, cout);
  if (off) {
    const int64_t ftotal = frag_halfs(k, cout);
    hipLaunchKernelGGL(pack_weights_frag_kernel, dim3((unsigned)rn::ceil_div64(ftotal, 256)), dim3(256), 0, (hipStream_t)stream, w,
                       (_Float16*)wt + off, (int)k, cout, ftotal);
    RN_LAUNCH_CHECK();
  }
  if (const int64_t so = sg_offset_halfs(kh, kw, cin_g, cout)) {
    const int64_t stotal = sg_halfs(cout);
    hipLaunchKernelGGL(pack_weights_sg_kernel, dim3((unsigned)rn::ceil_div64(stotal, 256)), dim3(256), 0, (hipStream_t)stream, w,
                       (_Float16*)wt + so, cin_g, cout, stotal);
    RN_LAUNCH_CHECK();
  }
  return RN_OK;
}

extern "C" int rn_cast_f32_to_f16(const float* x, void* y, int64_t count, rn_stream_t stream) {
  RN_CHECK_ARG(x && y && count >= 0, "cast: bad argument");
  if (count == 0) return RN_OK;
  int64_t b = rn::ceil_div64(count, 256);
  if (b > 4096) b = 4096;
  hipLaunchKernelGGL(cast_f32_to_f16_kernel, dim3((unsigned)b), dim3(256), 0, (hipStream_t)stream, x, (_Float16*)y, count);
  RN_LAUNCH_CHECK();
  return RN_OK;
}

extern "C" int rn_pad_cast_rgb_f16(const float* x, void* y, int64_t pixels, rn_stream_t stream) {
  RN_CHECK_ARG(x && y && pixels >= 0, "pad_cast_rgb: bad argument");
  if (pixels == 0) return RN_OK;
  int64_t b = rn::ceil_div64(pixels, 256);
  if (b > 8192) b = 8192;
  hipLaunchKernelGGL(pad_cast_rgb_kernel, dim3((unsigned)b), dim3(256), 0, (hipStream_t)stream, x, (_Float16*)y, pixels);
  RN_LAUNCH_CHECK();
  return RN_OK;
}

namespace {
// fold == nullptr: the plain convolution.  rows_out != nullptr: dry run -- *rows_out = m-tile rows per sample the statistics
// would take (0: this shape cannot fold), nothing is launched.
int conv_f16_impl(const rn_conv_seg* segs, int nseg, const rn_conv_geom* g, int out_f32, const rn_f16_fold* fold, int* rows_out,
                  rn_stream_t stream, int* tiles_out = nullptr) {
  RN_CHECK_ARG(segs && g && nseg >= 1 && nseg <= RN_MAX_SEG, "conv f16: bad segments");
  RN_CHECK_ARG(g->kh >= 1 && g->kw >= 1 && g->stride >= 1 && g->cin >= 1, "conv f16: bad geometry");
  const int G = g->groups > 1 ? g->groups : 1;
  RN_CHECK_ARG(g->cin % G == 0, "conv f16: cin %d not divisible by groups %d", g->cin, G);
  ArgsH a = {};
  a.nseg = nseg; a.kh = g->kh; a.kw = g->kw; a.stride = g->stride; a.cin = g->cin;
  a.groups = G; a.cin_g = g->cin / G; a.out_f32 = out_f32 ? 1 : 0;
  RN_UNSUPPORTED(a.cin_g % 4 != 0, "conv f16: input channels per group (%d) must be a multiple of 4", a.cin_g);
  for (int s = 0; s < nseg; ++s) {
    RN_CHECK_ARG(segs[s].x && segs[s].wgt && segs[s].y && segs[s].n >= 1 && segs[s].h >= 1 && segs[s].w >= 1 &&
                 segs[s].cout >= 1 && segs[s].cout % G == 0, "conv f16: bad segment %d", s);
    RN_UNSUPPORTED(G > 1 && segs[s].cout != segs[0].cout, "conv f16: grouped segments must share cout");
    SegH& d = a.seg[s];
    d.x = (const _Float16*)segs[s].x; d.wt = (const _Float16*)segs[s].wgt; d.bias = segs[s].bias; d.y = segs[s].y;
    {
      // the fragment-ordered copy is used only when the caller's buffer size proves it exists (a Wt-only buffer of an older
      // packing must not be read past its end)
      const int64_t kk = (int64_t)g->kh * g->kw * (g->cin / G), off = frag_offset_halfs(kk, segs[s].cout);
      const bool there = off && segs[s].wgt_bytes >= (int64_t)rn_pack_weights_f16_bytes(g->kh, g->kw, g->cin / G, segs[s].cout);
      d.wf = there ? d.wt + off : nullptr;
    }
    d.n = segs[s].n; d.h = segs[s].h; d.w = segs[s].w; d.cout = segs[s].cout;
    rn::same_pad(d.h, g->kh, g->stride, &d.oh, &d.pad_t);
    rn::same_pad(d.w, g->kw, g->stride, &d.ow, &d.pad_l);
    d.m = d.n * d.oh * d.ow;
    d.x_ld = segs[s].x_ld > 0 ? segs[s].x_ld : g->cin;
    d.x_coff = segs[s].x_ld > 0 ? segs[s].x_coff : 0;
    const double in_b = (double)d.n * d.h * d.w * d.x_ld * 2.0, out_b = (double)d.m * d.cout * 4.0;
    RN_UNSUPPORTED(in_b >= 2147483648.0 || out_b >= 2147483648.0 || d.h >= 32768 || d.w >= 32768,
                   "conv f16: a tensor of segment %d is >= 2 GiB", s);
  }
  const int cout_g = a.seg[0].cout / G;
  {
    // ResNeXt's conv 2 on 16 x 16 pixel tiles of 32-channel super-groups (conv3x3_sg32_f16_kernel); RN_F16_SG=0: the implicit GEMM
    static const bool sg_on = !(getenv("RN_F16_SG") && atoi(getenv("RN_F16_SG")) == 0);
    const SegH& d = a.seg[0];
    const int64_t so = sg_offset_halfs(g->kh, g->kw, a.cin_g, d.cout);
    const int S_ = g->stride;
    const bool shape_ok = sg_on && nseg == 1 && G > 1 && (S_ == 1 || S_ == 2) && so != 0 && cout_g == a.cin_g && d.x_ld == g->cin && d.x_coff == 0 &&
                          d.cout == g->cin && d.h % (16 * S_) == 0 && d.w % (16 * S_) == 0 && !out_f32 && d.bias == nullptr &&
                          segs[0].wgt_bytes >= (int64_t)rn_pack_weights_f16_bytes(g->kh, g->kw, a.cin_g, d.cout);
    if (shape_ok) {
      const int th = S_ == 1 ? 16 : 8;
      const int rows = (d.h / S_ / th) * (d.w / S_ / 16);
      if (rows_out) { *rows_out = rows; return RN_OK; }
      G3Args ga = {};
      ga.x = d.x; ga.ws = d.wt + so; ga.y = (_Float16*)d.y; ga.partial = fold ? fold->partial : nullptr;
      ga.n = d.n; ga.h = d.h; ga.w = d.w; ga.c = d.cout; ga.tiles_w = d.w / S_ / 16; ga.tiles_per_sample = rows; ga.nsg = d.cout / 32;
      ga.prows = d.n * rows;
      int fin = 0;
      if (fold && fold->in_mean) {
        RN_CHECK_ARG(fold->in_rstd && fold->in_gamma && fold->in_beta && fold->in_groups >= 1 && g->cin % fold->in_groups == 0 && fold->partial,
                     "conv f16 fold: incomplete input GroupNorm");
        ga.in_mean = fold->in_mean; ga.in_rstd = fold->in_rstd; ga.in_gamma = fold->in_gamma; ga.in_beta = fold->in_beta;
        ga.in_groups = fold->in_groups; ga.in_cpg = g->cin / fold->in_groups; ga.in_act = fold->in_act;
        fin = fold->in_act == RN_ACT_RELU ? 2 : 1;
      }
      const unsigned blocks = (unsigned)((long)d.n * rows * ga.nsg);
      hipStream_t st_ = (hipStream_t)stream;
#define RN_G3(S__)                                                                                                          \
      do {                                                                                                                      \
        if (fin == 2) hipLaunchKernelGGL((conv3x3_sg32_f16_kernel<true, 2, S__>), dim3(blocks), dim3(256), 0, st_, ga);           \
        else if (fin == 1) hipLaunchKernelGGL((conv3x3_sg32_f16_kernel<true, 1, S__>), dim3(blocks), dim3(256), 0, st_, ga);      \
        else if (ga.partial) hipLaunchKernelGGL((conv3x3_sg32_f16_kernel<true, 0, S__>), dim3(blocks), dim3(256), 0, st_, ga);    \
        else hipLaunchKernelGGL((conv3x3_sg32_f16_kernel<false, 0, S__>), dim3(blocks), dim3(256), 0, st_, ga);                   \
      } while (0)
      if (S_ == 1) RN_G3(1); else RN_G3(2);
#undef RN_G3
      RN_LAUNCH_CHECK();
      return RN_OK;
    }
  }
  int c;
  if (const char* force = getenv("RN_CONV_CFG")) {
    c = atoi(force) % 6;
  } else if (G > 1 && cout_g <= 64) {
    c = cout_g <= 32 ? 3 : 2;
  } else {
    double best = 1e300;
    c = 0;
    for (int k = 0; k < 4; ++k) {
      long tiles = 0;
      for (int s = 0; s < nseg; ++s)
        tiles += (long)rn::ceil_div(a.seg[s].m, kCfgs[k].bm) * G * rn::ceil_div(a.seg[s].cout / G, kCfgs[k].bn);
      const double cost = kT0[k] + kT1[k] * (double)((tiles + 255) / 256);
      if (cost < best) { best = cost; c = k; }
    }
    // compute-heavy launches that still fill the chip with 256x256 tiles (8 waves, 128x64 per wave): measured
    // 861-904 vs 760 TFLOP/s on the 3x3 head convs of a 1024^2 batch of 16, 619 vs 544 on a 1x1 512->256
    long big = 0;
    for (int s = 0; s < nseg; ++s) big += (long)rn::ceil_div(a.seg[s].m, 256) * rn::ceil_div(a.seg[s].cout, 256);
    // (1 x 1 convs whose output channels fill whole 256-wide tiles take it from K = 128: cfg 5's conv 3 / projection convs
    // 237 vs 290 us at 128 -> 256, 155 vs 196 us at 256 -> 512; a 128-wide output would waste half of every tile: 262 vs 189 us)
    static const long min_k_env = getenv("RN_F16_PIPE_MIN_K") ? atol(getenv("RN_F16_PIPE_MIN_K")) : 0;     // (tuning aid)
    const bool whole_n = g->kh == 1 && g->kw == 1 && nseg == 1 && a.seg[0].cout % 256 == 0;
    const long min_k = min_k_env > 0 ? min_k_env : (whole_n ? 128 : 512);
    if (G == 1 && (long)g->kh * g->kw * g->cin >= min_k && big >= 512 && a.cin_g % 8 == 0) c = 5;
  }
  a.tpg = G > 1 ? rn::ceil_div(cout_g, kCfgs[c].bn) : (1 << 20);
  int tiles = 0;
  for (int s = 0; s < nseg; ++s) {
    SegH& d = a.seg[s];
    d.tiles_n = G > 1 ? G * a.tpg : rn::ceil_div(d.cout, kCfgs[c].bn);
    d.start = tiles;
    tiles += rn::ceil_div(d.m, kCfgs[c].bm) * d.tiles_n;
  }
  const bool vec8 = a.cin_g % 8 == 0;
  const bool tapu = vec8 && a.cin_g % BK == 0;
  {
    // (RN_F16_NT: 0 never, 1 whenever the operand is read once; default: only where it cannot stay cached anyway, >= 32 MB)
    static const int nt_mode = getenv("RN_F16_NT") ? atoi(getenv("RN_F16_NT")) : -1;
    const SegH& d0 = a.seg[0];
    const bool once = nseg == 1 && G == 1 && g->kh == 1 && g->kw == 1 && g->stride == 1 && d0.tiles_n == 1;
    const double bytes = (double)d0.n * d0.h * d0.w * d0.x_ld * 2.0;
    a.a_nt = once && (nt_mode >= 0 ? nt_mode != 0 : bytes >= 33554432.0);
  }
  int fbits = 0;
  if (tiles_out || (fold && fold->seg_chunk_start)) {
    // Several segments (the head towers' pyramid levels in one launch), output statistics only: every segment whose m-tiles lie
    // inside one sample writes its rows into ONE array [2][total_chunks][cout] at its own first row (seg_chunk_start[s]; its
    // rows: sample-major, tiles_out[s] per sample); a segment whose tiles straddle samples (P7: 64 pixels per sample) gets 0.
    RN_UNSUPPORTED(out_f32 || (cout_g & 7) != 0 || G != 1, "conv f16 stats: fp16 output, cout %% 8 == 0, no groups");
    RN_UNSUPPORTED(fold && fold->in_mean, "conv f16 stats: several segments fold the output side only");
    for (int s = 0; s < nseg; ++s) {
      SegH& d = a.seg[s];
      const int ohw = d.oh * d.ow, t = ohw % kCfgs[c].bm == 0 ? ohw / kCfgs[c].bm : 0;
      if (tiles_out) { tiles_out[s] = t; continue; }
      RN_CHECK_ARG(d.bias == nullptr && fold->partial && fold->total_chunks >= 1, "conv f16 stats: bias-free conv, a partial array");
      d.partial = t ? fold->partial + (size_t)fold->seg_chunk_start[s] * d.cout : nullptr;
      d.prows = fold->total_chunks;
    }
    if (tiles_out) return RN_OK;
    fbits = 2;
  } else if (fold || rows_out) {
    // a tile's rows must lie inside one sample, the fp16 staged epilogue must be the one that runs, one segment
    const SegH& d = a.seg[0];
    const int ohw = d.oh * d.ow;
    // (the output statistics do not care how the operands are gathered: the RGB stem's 4-channel vectors qualify; the input-side
    // fold transforms whole 8-channel vectors)
    const bool ok = nseg == 1 && !out_f32 && (cout_g & 7) == 0 && ohw % kCfgs[c].bm == 0;
    if (rows_out) { *rows_out = ok ? ohw / kCfgs[c].bm : 0; return RN_OK; }
    RN_UNSUPPORTED(!ok, "conv f16 fold: this shape cannot fold its GroupNorms (see rn_conv2d_f16_fold_rows)");
    RN_UNSUPPORTED(fold->in_mean && !vec8, "conv f16 fold: a GroupNorm on the operand load needs input channels in multiples of 8");
    if (fold->in_mean) {
      RN_CHECK_ARG(fold->in_rstd && fold->in_gamma && fold->in_beta && fold->in_groups >= 1 && g->cin % fold->in_groups == 0,
                   "conv f16 fold: incomplete input GroupNorm");
      RN_UNSUPPORTED(g->cin > 2048 || d.x_ld != g->cin, "conv f16 fold: input GroupNorm over %d channels (<= 2048, dense)", g->cin);
      a.fold.in_mean = fold->in_mean; a.fold.in_rstd = fold->in_rstd; a.fold.in_gamma = fold->in_gamma; a.fold.in_beta = fold->in_beta;
      a.fold.in_groups = fold->in_groups; a.fold.in_cpg = g->cin / fold->in_groups; a.fold.in_act = fold->in_act; a.fold.in_c = g->cin;
      fbits |= 1;
    }
    RN_UNSUPPORTED(!fold->partial, "conv f16 fold: the input-side fold is built together with the output statistics only (pass `partial`)");
    if (fold->partial) {
      RN_CHECK_ARG(d.bias == nullptr, "conv f16 fold: the statistics are those of a bias-free conv (the GroupNorm behind it absorbs a bias)");
      a.fold.partial = fold->partial; a.fold.prows = d.n * (ohw / kCfgs[c].bm);
      a.seg[0].partial = fold->partial; a.seg[0].prows = a.fold.prows;
      fbits |= 2;
    }
  }
  hipStream_t st = (hipStream_t)stream;
#define RN_F16K(BM_, BN_, WM_, WN_, F_)                                                                             \
  do {                                                                                                              \
    if (tapu) hipLaunchKernelGGL((conv_f16_kernel<BM_, BN_, WM_, WN_, 8, true, F_>), dim3(tiles), dim3(WM_* WN_ * 64), 0, st, a); \
    else hipLaunchKernelGGL((conv_f16_kernel<BM_, BN_, WM_, WN_, 8, false, F_>), dim3(tiles), dim3(WM_* WN_ * 64), 0, st, a); \
  } while (0)
#define RN_F16(BM_, BN_, WM_, WN_)                                                                                  \
  do {                                                                                                              \
    if (fbits == 2 && vec8) RN_F16K(BM_, BN_, WM_, WN_, 2);                                                    \
    else if (fbits == 3 && a.fold.in_act == RN_ACT_RELU) RN_F16K(BM_, BN_, WM_, WN_, 7);                            \
    else if (fbits == 3) RN_F16K(BM_, BN_, WM_, WN_, 3);                                                            \
    else if (vec8) RN_F16K(BM_, BN_, WM_, WN_, 0);                                                                  \
    else if (fbits == 2) hipLaunchKernelGGL((conv_f16_kernel<BM_, BN_, WM_, WN_, 4, false, 2>), dim3(tiles), dim3(WM_* WN_ * 64), 0, st, a); \
    else hipLaunchKernelGGL((conv_f16_kernel<BM_, BN_, WM_, WN_, 4, false, 0>), dim3(tiles), dim3(WM_* WN_ * 64), 0, st, a); \
  } while (0)
  // the 256 x 256 tile, dense taps, no input-side fold: the software-pipelined variants (147 KB of dynamic LDS); RN_F16_PIPE: 0 the
  // plain loop, 1 both operands through LDS, 2 (default) the weight fragments straight from the fragment-ordered copy
  static const int pipe_mode = getenv("RN_F16_PIPE") ? atoi(getenv("RN_F16_PIPE")) : 2;
  // (an input-side fold rides along where it is a ReLU GroupNorm in front of a 1 x 1 conv: ResNeXt's conv 3)
  const bool fin_pipe = fbits == 3 && pipe_mode == 2 && g->kh == 1 && g->kw == 1 && g->stride == 1 && a.fold.in_act == RN_ACT_RELU &&
                        nseg == 1 && a.seg[0].wf != nullptr;
  if (c == 5 && tapu && pipe_mode && G == 1 && (fbits == 0 || fbits == 2 || fin_pipe)) {
    constexpr size_t lds = 2 * (256 + 256) * LDH * sizeof(_Float16);
    static const bool attr_ = (hipFuncSetAttribute((const void*)conv_f16_kernel<256, 256, 2, 4, 8, true, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess) &&
                              (hipFuncSetAttribute((const void*)conv_f16_kernel<256, 256, 2, 4, 8, true, 2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess) &&
                              (hipFuncSetAttribute((const void*)conv_f16_kernel<256, 256, 2, 4, 8, true, 0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess) &&
                              (hipFuncSetAttribute((const void*)conv_f16_kernel<256, 256, 2, 4, 8, true, 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess) &&
                              (hipFuncSetAttribute((const void*)conv_f16_kernel<256, 256, 2, 4, 8, true, 7, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess);
    if (attr_) {           // (refused: the plain loop below computes the same bits from 36 KB of static LDS)
      bool frag = pipe_mode == 2;
      for (int s = 0; s < nseg; ++s) frag = frag && a.seg[s].wf != nullptr;
      if (fin_pipe) {
        hipLaunchKernelGGL((conv_f16_kernel<256, 256, 2, 4, 8, true, 7, 2>), dim3(tiles), dim3(512), lds, st, a);
      } else if (frag) {
        if (fbits == 2) hipLaunchKernelGGL((conv_f16_kernel<256, 256, 2, 4, 8, true, 2, 2>), dim3(tiles), dim3(512), lds, st, a);
        else hipLaunchKernelGGL((conv_f16_kernel<256, 256, 2, 4, 8, true, 0, 2>), dim3(tiles), dim3(512), lds, st, a);
      } else {
        if (fbits == 2) hipLaunchKernelGGL((conv_f16_kernel<256, 256, 2, 4, 8, true, 2, 1>), dim3(tiles), dim3(512), lds, st, a);
        else hipLaunchKernelGGL((conv_f16_kernel<256, 256, 2, 4, 8, true, 0, 1>), dim3(tiles), dim3(512), lds, st, a);
      }
      RN_LAUNCH_CHECK();
      return RN_OK;
    }
  }
  switch (c) {
    case 0: RN_F16(128, 128, 2, 2); break;
    case 1: RN_F16(128, 64, 2, 2); break;
    case 2: RN_F16(64, 64, 2, 2); break;
    case 4: RN_F16(256, 128, 2, 2); break;
    case 5: RN_F16(256, 256, 2, 4); break;
    default: RN_F16(128, 32, 4, 1); break;
  }
#undef RN_F16
#undef RN_F16K
  RN_LAUNCH_CHECK();
  return RN_OK;
}
}  // namespace

extern "C" int rn_conv2d_fwd_f16(const rn_conv_seg* segs, int nseg, const rn_conv_geom* g, int out_f32, rn_stream_t stream) {
  return conv_f16_impl(segs, nseg, g, out_f32, nullptr, nullptr, stream);
}

extern "C" int rn_conv2d_f16_fold_rows(const rn_conv_seg* segs, int nseg, const rn_conv_geom* g) {
  int rows = 0;
  if (conv_f16_impl(segs, nseg, g, 0, nullptr, &rows, nullptr) != RN_OK) return 0;
  return rows;
}

extern "C" int rn_conv2d_f16_stats_tiles(const rn_conv_seg* segs, int nseg, const rn_conv_geom* g, int32_t* tiles_per_sample) {
  RN_CHECK_ARG(tiles_per_sample, "conv f16 stats tiles: null argument");
  return conv_f16_impl(segs, nseg, g, 0, nullptr, nullptr, nullptr, tiles_per_sample);
}

extern "C" int rn_conv2d_fwd_f16_fold(const rn_conv_seg* segs, int nseg, const rn_conv_geom* g, const rn_f16_fold* fold, rn_stream_t stream) {
  RN_CHECK_ARG(fold && (fold->in_mean || fold->partial), "conv f16 fold: nothing to fold");
  return conv_f16_impl(segs, nseg, g, 0, fold, nullptr, stream);
}
