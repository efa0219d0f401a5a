// Anchor decode, candidate extraction and batched class-wise greedy NMS.
// (utils.regression_postprocess utils.py:108-117; boxes_decode :183-195; merge_boxes_decoded
//  :223-227; nms_classwise / nms :198-220 = tf.image.non_max_suppression(max 1000, IoU>0.5).)
// Compiled with -ffp-contract=off: the IoU / threshold arithmetic is float32 op-for-op the
// oracle's, so the kept indices are bit-identical (north_star: "bit-exact ... NMS indices").
//
// Pipeline for a whole batch, no host round trip:
//   1. scan   : one wave per anchor row: max / first-arg-max over the C class probabilities
//               (coalesced row reads, HBM-bound: this is where the bytes are), per-wave counts
//   2. offsets: exclusive scan of the per-wave counts (order-preserving compaction)
//   3. emit   : candidates (box, score, class, image, anchor) in anchor order, and a count per (image, class) segment
//   4. sort   : hand-written, in two steps: (a) the candidates are dropped into their segment's slot range (exclusive scan
//               of the segment counts; the slot inside the range comes from an atomic counter -- any order), as 64-bit
//               keys ~score_bits << 32 | candidate position; (b) one block per segment sorts its keys with a bitonic
//               network in LDS (all-ascending formulation, so ragged lengths need no padding values).  The key orders by
//               score descending, then by lower anchor position: the result does not depend on the order of step (a)
//               => segments (image, class), score descending, ties by lower anchor index
//   5. nms    : one wave per (image, class) segment, exact greedy semantics: 64 candidates at
//               a time are tested against the kept list (LDS) in parallel, then resolved in
//               order with wave ballots; stops at max_per_class
//   6. gather : kept boxes of all segments, (image, class)-major = the reference's output order
#include <cstring>
#include <string.h>

#include "rn_common.h"

namespace {

constexpr int T = 256;
constexpr int MAXL = 8;

struct DetLevel {
  const void* prob; const float* boxes; int64_t rows; int64_t row_off; int wave_off;
  const void* reg; const float* anch; int h, w, A;  // boxes == nullptr: decode candidates from the raw regression
  int half_prob, half_reg, logit;
};
struct DetArgs {
  DetLevel lv[MAXL];
  int nlv, n, C, max_keep;
  float score_thr, iou_thr;
  int64_t cap;            // candidate capacity
  int64_t rows_per_image; // sum over levels
  int waves_per_image;
  // workspace
  float* row_score; int32_t* row_class; int32_t* wave_count; int32_t* wave_off; unsigned long long* wave_mask;
  uint64_t* keys; uint32_t* vals_out; int32_t* seg_count; int32_t* seg_fill;
  float* cand_box; float* cand_score; int32_t* cand_class; int32_t* cand_image; int64_t* cand_anchor;
  int32_t* seg_start; int32_t* seg_keep; int32_t* seg_off; int32_t* keep_idx;
  // outputs
  float* out_boxes; float* out_scores; int32_t* out_class; int32_t* out_image; int64_t* out_anchor; int64_t* counts;
};

__device__ __forceinline__ void locate_wave(const DetArgs& a, int64_t wid, int* img, int* lvl, int64_t* row0) {
  *img = (int)(wid / a.waves_per_image);
  const int w = (int)(wid - (int64_t)(*img) * a.waves_per_image);
  int l = 0;
  while (l + 1 < a.nlv && w >= a.lv[l + 1].wave_off) ++l;
  *lvl = l;
  *row0 = (int64_t)(w - a.lv[l].wave_off) * 64;
}

__device__ __forceinline__ void locate_wave32(const DetArgs& a, uint32_t wid, int* img, int* lvl, int64_t* row0) {
  const uint32_t wpi = (uint32_t)a.waves_per_image;
  const uint32_t im = wid / wpi;
  const int w = (int)(wid - im * wpi);
  *img = (int)im;
  int l = 0;
  while (l + 1 < a.nlv && w >= a.lv[l + 1].wave_off) ++l;
  *lvl = l;
  *row0 = (int64_t)(w - a.lv[l].wave_off) * 64;
}

// ---- 1. scan: lane i of a wave ends up holding (max prob, arg-max) of row row0+i.
// Four lanes share a row (float4 loads, 64-B contiguous per row and instruction), 16 rows per
// pass, 2 shuffle steps to combine; first index wins ties (tf.argmax).  C % 4 == 0 fast path.
// per-row (max, lowest arg-max) of the wave's (up to) 64 rows of C = 4 * C4 probabilities: 4 lanes per row, 16 rows per
// pass, Q float4 chunks per lane; lane l ends up with row l's result
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ float to_prob(float z, bool logit) { return logit ? 1.f / (1.f + __expf(-z)) : z; }

// V = elements per 16-byte load: 4 (fp32) or 8 (fp16)
template <bool HALF>
struct Chunk;
template <>
struct Chunk<false> {
  static constexpr int V = 4;
  float e[4];
  __device__ __forceinline__ void load(const void* row, int ci) {
    const float4 t = reinterpret_cast<const float4*>(row)[ci];
    e[0] = t.x; e[1] = t.y; e[2] = t.z; e[3] = t.w;
  }
  __device__ __forceinline__ float get(int i) const { return e[i]; }
};
template <>
struct Chunk<true> {   // stays packed in registers (4 VGPRs per chunk, not 8): the scan's occupancy is set by its registers
  static constexpr int V = 8;
  half8 t;
  __device__ __forceinline__ void load(const void* row, int ci) { t = reinterpret_cast<const half8*>(row)[ci]; }
  __device__ __forceinline__ float get(int i) const { return (float)t[i]; }
};

// LOGIT: the rows hold logits and the result must be that of "sigmoid every element, then max / first arg-max of the
// probabilities".  Evaluating 80 exponentials per anchor made the scan ALU-bound (184 us for 528 MB); instead:
//   1. max logit m of the row and its first index (compares only);
//   2. p = sigmoid(m); an element can tie with or (through the last-ulp wobble of the hardware exponential) exceed p only
//      if its logit lies within margin(m) = 5e-7 (1 + e^m) + 1e-6 |m| below m -- that is >= 8 ulp(p) / sigmoid'(m) -- or
//      both are in the saturated range (> 16: p == 1.0f); only those elements (normally none) get their own sigmoid and
//      the (greater probability, lower index) rule.
// Bit-identical to the element-wise form for every input the margin covers, two exponentials per row instead of C.
template <int Q, bool HALF, bool LOGIT>
__device__ __forceinline__ void scan_rows_unrolled(const void* __restrict__ base, int C, int nrow, int lane, float* out_s, int* out_c) {
  constexpr int V = Chunk<HALF>::V, ESZ = HALF ? 2 : 4;
  const int sub = lane & 3, rsel = lane >> 2, CV = C / V;
  Chunk<HALF> v[4][Q];
  int cc[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) cc[q] = min(sub + 4 * q, CV - 1);
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int r = min(pass * 16 + rsel, nrow - 1);
    const void* row = reinterpret_cast<const char*>(base) + (size_t)r * C * ESZ;
#pragma unroll
    for (int q = 0; q < Q; ++q) v[pass][q].load(row, cc[q]);
  }
  float my_s = 0.f; int my_c = 0;
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    float best = -1e30f; int bi = 0x7fffffff;
#pragma unroll
    for (int q = 0; q < Q; ++q) {  // ascending chunk index per lane: strict > keeps the lowest index on ties
      const int c = cc[q] * V;
#pragma unroll
      for (int i = 0; i < V; ++i) {
        const float pv = v[pass][q].get(i);
        if (pv > best) { best = pv; bi = c + i; }
      }
    }
#pragma unroll
    for (int o = 1; o <= 2; o <<= 1) {
      const float ob = __shfl_xor(best, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (LOGIT) {   // best / bi: the row's max LOGIT and its first index, the same in the row's 4 lanes
      const float m = best;
      float pb = to_prob(m, true);
      int pi = bi;
      float zlo = m - (5e-7f * (1.f + __expf(m)) + 1e-6f * fabsf(m));
      zlo = fminf(zlo, 16.f);
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        const int c = cc[q] * V;
#pragma unroll
        for (int i = 0; i < V; ++i) {
          const float z = v[pass][q].get(i);
          if (z >= zlo && c + i != bi) {            // rare: a possible tie / inversion in probability space
            const float pz = to_prob(z, true);
            if (pz > pb || (pz == pb && c + i < pi)) { pb = pz; pi = c + i; }
          }
        }
      }
#pragma unroll
      for (int o = 1; o <= 2; o <<= 1) {
        const float ob = __shfl_xor(pb, o, 64);
        const int oi = __shfl_xor(pi, o, 64);
        if (ob > pb || (ob == pb && oi < pi)) { pb = ob; pi = oi; }
      }
      best = pb; bi = pi;
    }
    const float sv = __shfl(best, (lane & 15) * 4, 64);
    const int sc = __shfl(bi, (lane & 15) * 4, 64);
    if ((lane >> 4) == pass) { my_s = sv; my_c = sc; }
  }
  *out_s = my_s; *out_c = my_c;
}

template <bool HALF, bool LOGIT>
__device__ __forceinline__ void scan_wave(const DetArgs& a, const DetLevel& lv, int img, int64_t row0, int nrow, int lane, float* out_s,
                                          int* out_c) {
  constexpr int V = Chunk<HALF>::V, ESZ = HALF ? 2 : 4;
  const void* base = reinterpret_cast<const char*>(lv.prob) + ((size_t)img * lv.rows + row0) * a.C * ESZ;
  float my_s = 0.f; int my_c = 0;
  const int q_per_lane = ((a.C / V) + 3) >> 2;  // 16-byte chunks per lane when 4 lanes share a row
  if (a.C % V == 0 && q_per_lane <= 8) {
    // every load of the wave's 64 rows is issued before the first compare (clamped addresses: a duplicated chunk or row
    // cannot change a max / lowest-index arg-max), instead of one dependent load per loop iteration
    switch (q_per_lane) {
      case 1: scan_rows_unrolled<1, HALF, LOGIT>(base, a.C, nrow, lane, &my_s, &my_c); break;
      case 2: scan_rows_unrolled<2, HALF, LOGIT>(base, a.C, nrow, lane, &my_s, &my_c); break;
      case 3: scan_rows_unrolled<3, HALF, LOGIT>(base, a.C, nrow, lane, &my_s, &my_c); break;
      case 4: scan_rows_unrolled<4, HALF, LOGIT>(base, a.C, nrow, lane, &my_s, &my_c); break;
      case 5: scan_rows_unrolled<5, HALF, LOGIT>(base, a.C, nrow, lane, &my_s, &my_c); break;
      case 6: scan_rows_unrolled<6, HALF, LOGIT>(base, a.C, nrow, lane, &my_s, &my_c); break;
      case 7: scan_rows_unrolled<7, HALF, LOGIT>(base, a.C, nrow, lane, &my_s, &my_c); break;
      default: scan_rows_unrolled<8, HALF, LOGIT>(base, a.C, nrow, lane, &my_s, &my_c); break;
    }
  } else {   // any C: one row at a time, lanes stride over the classes
    for (int i = 0; i < nrow; ++i) {
      float best = -1e30f; int bi = 0x7fffffff;
      for (int c = lane; c < a.C; c += 64) {
        const size_t e = (size_t)i * a.C + c;
        const float raw = HALF ? (float)reinterpret_cast<const _Float16*>(base)[e] : reinterpret_cast<const float*>(base)[e];
        const float pv = to_prob(raw, LOGIT);
        if (pv > best) { best = pv; bi = c; }   // ascending c per lane: keeps the lowest index on ties
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
      }
      if (lane == i) { my_s = best; my_c = bi; }
    }
  }
  *out_s = my_s; *out_c = my_c;
}

// ---- 1'. the same scan with the wave's rows staged through LDS: the wave's (up to) 64 rows are ONE contiguous range of
// memory, fetched with fully coalesced 16-byte lane loads (lane l takes chunks l, l + 64, ...: 1 KB per instruction), parked
// in LDS with a row stride of an odd number of 16-byte units (conflict-free ds_read_b128 when every lane then walks its own
// row), and lane l reduces row l sequentially -- no shuffles, ascending class order, so strict > keeps the first maximum.
// The four-lanes-per-row variant above touches ~20 cache lines per load instruction (16 rows x 64 bytes) and ran at
// 2.8 TB/s whatever the element size; this one is bound by HBM.  Needs row bytes % 16 == 0 and 64 rows <= SCAN_LDS_WAVE.
constexpr int SCAN_LDS_WAVE = 24 * 1024;   // most LDS bytes a wave may take (4 waves per block)
template <bool HALF, bool LOGIT>
__device__ __forceinline__ void scan_wave_lds(const DetArgs& a, const DetLevel& lv, int img, int64_t row0, int nrow, int lane, char* lds,
                                              float* out_s, int* out_c) {
  constexpr int V = Chunk<HALF>::V, ESZ = HALF ? 2 : 4;
  const int CV = a.C / V;                              // 16-byte chunks per row
  const int RS = (CV | 1) * 16;                        // row stride in LDS: an odd number of chunks
  const char* base = reinterpret_cast<const char*>(lv.prob) + ((size_t)img * lv.rows + row0) * a.C * ESZ;
  const int total = nrow * CV;
  const float inv_cv = 1.f / (float)CV;
  for (int g0 = 0; g0 < total; g0 += 64 * 8) {         // 8 loads in flight per lane
    uint4 t[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = reinterpret_cast<const uint4*>(base)[min(g0 + j * 64 + lane, total - 1)];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int g = g0 + j * 64 + lane;
      if (g < total) {
        const int r = (int)(((float)g + 0.5f) * inv_cv), ch = g - r * CV;     // exact for these sizes (see dw_gn.hip fast_div)
        *reinterpret_cast<uint4*>(lds + r * RS + ch * 16) = t[j];
      }
    }
  }
  // (one wave owns this LDS region: no block barrier needed, only the wave's own LDS writes must have landed)
  __builtin_amdgcn_s_waitcnt(0xc07f);                  // lgkmcnt(0)
  __builtin_amdgcn_wave_barrier();
  const char* row = lds + min(lane, nrow - 1) * RS;
  float best = -1e30f; int bi = 0;
  for (int c4 = 0; c4 < CV; ++c4) {
    Chunk<HALF> ck;
    ck.load(row, c4);
#pragma unroll
    for (int i = 0; i < V; ++i)
      if (ck.get(i) > best) { best = ck.get(i); bi = c4 * V + i; }
  }
  if (LOGIT) {   // see scan_rows_unrolled: sigmoid of the max logit, exact handling of possible ties in probability space
    const float m = best;
    float pb = to_prob(m, true);
    int pi = bi;
    const float zlo = fminf(m - (5e-7f * (1.f + __expf(m)) + 1e-6f * fabsf(m)), 16.f);
    bool any = false;
    for (int c4 = 0; c4 < CV; ++c4) {
      Chunk<HALF> ck;
      ck.load(row, c4);
#pragma unroll
      for (int i = 0; i < V; ++i) any = any || (ck.get(i) >= zlo && c4 * V + i != bi);
    }
    if (any) {   // rare
      for (int c4 = 0; c4 < CV; ++c4) {
        Chunk<HALF> ck;
        ck.load(row, c4);
#pragma unroll
        for (int i = 0; i < V; ++i) {
          const float z = ck.get(i);
          if (z >= zlo && c4 * V + i != bi) {
            const float pz = to_prob(z, true);
            if (pz > pb || (pz == pb && c4 * V + i < pi)) { pb = pz; pi = c4 * V + i; }
          }
        }
      }
    }
    best = pb; bi = pi;
  }
  *out_s = best; *out_c = bi;
}

__global__ __launch_bounds__(T) void det_scan_lds_kernel(const DetArgs a, int lds_per_wave) {
  extern __shared__ char scan_lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t wid = ((int64_t)blockIdx.x * T + threadIdx.x) >> 6;
  const int64_t nw = (int64_t)a.n * a.waves_per_image;
  if (wid >= nw) return;
  int img, l; int64_t row0;
  locate_wave(a, wid, &img, &l, &row0);
  const DetLevel& lv = a.lv[l];
  const int nrow = (int)((lv.rows - row0) < 64 ? (lv.rows - row0) : 64);
  char* lds = scan_lds + (size_t)wave * lds_per_wave;
  float my_s = 0.f; int my_c = 0;
  if (lv.half_prob) {
    if (lv.logit) scan_wave_lds<true, true>(a, lv, img, row0, nrow, lane, lds, &my_s, &my_c);
    else scan_wave_lds<true, false>(a, lv, img, row0, nrow, lane, lds, &my_s, &my_c);
  } else {
    if (lv.logit) scan_wave_lds<false, true>(a, lv, img, row0, nrow, lane, lds, &my_s, &my_c);
    else scan_wave_lds<false, false>(a, lv, img, row0, nrow, lane, lds, &my_s, &my_c);
  }
  // only the candidates' (score, class) are kept (~1 % of the rows: sparse stores instead of 8 bytes for every row), with the
  // wave's candidate mask: the emit pass reads 12 bytes per wave and returns at once where the mask is empty
  const bool flag = lane < nrow && my_s > a.score_thr;
  const int64_t g = (int64_t)img * a.rows_per_image + lv.row_off + row0 + lane;
  if (flag) { a.row_score[g] = my_s; a.row_class[g] = my_c; }
  const unsigned long long m = __ballot(flag);
  if (lane == 0) { a.wave_count[wid] = __popcll(m); a.wave_mask[wid] = m; }
}

// the unrolled scan for ONE (storage type, logit mode, chunks per lane): registers sized for this case only.  (The
// catch-all kernel below contains every Q up to 8 and is allocated 250 VGPRs -- two waves per SIMD, which made the scan
// latency-bound at 2.8 TB/s whatever the element size.)
template <bool HALF, bool LOGIT, int Q>
__global__ __launch_bounds__(T) void det_scan_q_kernel(const DetArgs a) {
  constexpr int ESZ = HALF ? 2 : 4;
  const int lane = threadIdx.x & 63;
  const int64_t wid = ((int64_t)blockIdx.x * T + threadIdx.x) >> 6;
  const int64_t nw = (int64_t)a.n * a.waves_per_image;
  if (wid >= nw) return;
  int img, l; int64_t row0;
  locate_wave(a, wid, &img, &l, &row0);
  const DetLevel& lv = a.lv[l];
  const int nrow = (int)((lv.rows - row0) < 64 ? (lv.rows - row0) : 64);
  const void* base = reinterpret_cast<const char*>(lv.prob) + ((size_t)img * lv.rows + row0) * a.C * ESZ;
  float my_s = 0.f; int my_c = 0;
  scan_rows_unrolled<Q, HALF, LOGIT>(base, a.C, nrow, lane, &my_s, &my_c);
  // only the candidates' (score, class) are kept (~1 % of the rows: sparse stores instead of 8 bytes for every row), with the
  // wave's candidate mask: the emit pass reads 12 bytes per wave and returns at once where the mask is empty
  const bool flag = lane < nrow && my_s > a.score_thr;
  const int64_t g = (int64_t)img * a.rows_per_image + lv.row_off + row0 + lane;
  if (flag) { a.row_score[g] = my_s; a.row_class[g] = my_c; }
  const unsigned long long m = __ballot(flag);
  if (lane == 0) { a.wave_count[wid] = __popcll(m); a.wave_mask[wid] = m; }
}

// fp16 maps with C = 8 CVT classes (CVT even): the wave's 64 rows are ONE contiguous range of memory, read as whole 1 KB
// instructions (lane l takes chunks l, l + 64, ...: 128-byte requests only -- the four-lanes-per-row kernels above issue
// 64-byte pieces of 16 rows per instruction and stream at 4.4 TB/s, this pattern at 5.4), every load of the wave in flight at
// once.  The chunks are then transposed through the wave's own LDS region, 32 rows at a time (5.6 KB per wave: 28 waves per CU;
// row stride an odd number of 16-byte units: conflict-free both ways): lanes l and l + 32 hold the two halves of row l's
// classes, reduce them in ascending class order (strict >: the first maximum wins) and exchange once; the exact logit ->
// probability rule of scan_rows_reduce follows from the same registers.  No block barrier: a wave touches only its own region.
template <int CVT, bool LOGIT, bool NT_LOADS = true>
__global__ __launch_bounds__(T) void det_scan_t_kernel(const DetArgs a) {
  static_assert(CVT % 2 == 0, "the two lanes of a row take CVT / 2 chunks each");
  extern __shared__ __attribute__((aligned(16))) char scan_t_lds[];
  constexpr int RS = (CVT | 1) * 16;                   // LDS row stride
  constexpr int HC = CVT / 2;                          // chunks per lane and half
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t wid = ((int64_t)blockIdx.x * T + threadIdx.x) >> 6;
  const int64_t nw = (int64_t)a.n * a.waves_per_image;
  if (wid >= nw) return;
  int img, l; int64_t row0;
  locate_wave(a, wid, &img, &l, &row0);
  const DetLevel& lv = a.lv[l];
  const int nrow = (int)((lv.rows - row0) < 64 ? (lv.rows - row0) : 64);
  const char* base = reinterpret_cast<const char*>(lv.prob) + ((size_t)img * lv.rows + row0) * (CVT * 16);
  char* lds = scan_t_lds + wave * (32 * RS);
  const int total = nrow * CVT;
  half8 v[CVT];
#pragma unroll
  for (int j = 0; j < CVT; ++j) {
    const half8* src = reinterpret_cast<const half8*>(base) + min(lane + 64 * j, total - 1);
    // the 528 MB of a cfg-5 batch are read once: non-temporal loads keep them from displacing each other (and the candidate
    // lists) in L2 / the memory-side cache: 5.2 -> 5.8 TB/s (0.174 -> 0.160 ms per batch; RN_SCAN_NT=0: plain loads)
    v[j] = NT_LOADS ? __builtin_nontemporal_load(src) : *src;
  }
  const int part = lane >> 5;                          // 0: classes [0, 4 CVT), 1: the rest
  const char* rd = lds + (lane & 31) * RS + part * HC * 16;
  float my_s = 0.f; int my_c = 0;
#pragma unroll
  for (int h = 0; h < 2; ++h) {                        // rows [32 h, 32 h + 32) = chunks [32 CVT h, + 32 CVT) = loads HC h .. HC h + HC - 1
#pragma unroll
    for (int j = 0; j < HC; ++j) {
      const int c = lane + 64 * j;
      const int r = c / CVT, ps = c - r * CVT;
      *reinterpret_cast<half8*>(lds + r * RS + ps * 16) = v[HC * h + j];
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);                // lgkmcnt(0): the wave's own stores have landed
    __builtin_amdgcn_wave_barrier();
    half8 u[HC];
#pragma unroll
    for (int j = 0; j < HC; ++j) u[j] = *reinterpret_cast<const half8*>(rd + j * 16);
    __builtin_amdgcn_s_waitcnt(0xc07f);                // (the next half overwrites the region)
    __builtin_amdgcn_wave_barrier();
    float best = -1e30f; int bi = 0x7fffffff;
#pragma unroll
    for (int j = 0; j < HC; ++j)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float pv = (float)u[j][i];
        if (pv > best) { best = pv; bi = (part * HC + j) * 8 + i; }
      }
    {
      const float ob = __shfl_xor(best, 32, 64);
      const int oi = __shfl_xor(bi, 32, 64);
      if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (LOGIT) {   // see scan_rows_reduce: sigmoid of the max logit, exact handling of possible ties in probability space
      const float m = best;
      float pb = to_prob(m, true);
      int pi = bi;
      float zlo = m - (5e-7f * (1.f + __expf(m)) + 1e-6f * fabsf(m));
      zlo = fminf(zlo, 16.f);
#pragma unroll
      for (int j = 0; j < HC; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float z = (float)u[j][i];
          const int ci = (part * HC + j) * 8 + i;
          if (z >= zlo && ci != bi) {                  // rare: a possible tie / inversion in probability space
            const float pz = to_prob(z, true);
            if (pz > pb || (pz == pb && ci < pi)) { pb = pz; pi = ci; }
          }
        }
      const float ob = __shfl_xor(pb, 32, 64);
      const int oi = __shfl_xor(pi, 32, 64);
      if (ob > pb || (ob == pb && oi < pi)) { pb = ob; pi = oi; }
      best = pb; bi = pi;
    }
    if (part == h) { my_s = best; my_c = bi; }         // lanes l and l + 32 both hold row 32 h + (l & 31): lane 32 h + (l & 31) keeps it
  }
  const bool flag = lane < nrow && my_s > a.score_thr;
  const int64_t g = (int64_t)img * a.rows_per_image + lv.row_off + row0 + lane;
  if (flag) { a.row_score[g] = my_s; a.row_class[g] = my_c; }
  const unsigned long long m = __ballot(flag);
  if (lane == 0) { a.wave_count[wid] = __popcll(m); a.wave_mask[wid] = m; }
}

__global__ __launch_bounds__(T) void det_scan_kernel(const DetArgs a) {
  const int lane = threadIdx.x & 63;
  const int64_t wid = ((int64_t)blockIdx.x * T + threadIdx.x) >> 6;
  const int64_t nw = (int64_t)a.n * a.waves_per_image;
  if (wid >= nw) return;
  int img, l; int64_t row0;
  locate_wave(a, wid, &img, &l, &row0);
  const DetLevel& lv = a.lv[l];
  const int nrow = (int)((lv.rows - row0) < 64 ? (lv.rows - row0) : 64);
  float my_s = 0.f; int my_c = 0;
  if (lv.half_prob) {
    if (lv.logit) scan_wave<true, true>(a, lv, img, row0, nrow, lane, &my_s, &my_c);
    else scan_wave<true, false>(a, lv, img, row0, nrow, lane, &my_s, &my_c);
  } else {
    if (lv.logit) scan_wave<false, true>(a, lv, img, row0, nrow, lane, &my_s, &my_c);
    else scan_wave<false, false>(a, lv, img, row0, nrow, lane, &my_s, &my_c);
  }
  // only the candidates' (score, class) are kept (~1 % of the rows: sparse stores instead of 8 bytes for every row), with the
  // wave's candidate mask: the emit pass reads 12 bytes per wave and returns at once where the mask is empty
  const bool flag = lane < nrow && my_s > a.score_thr;
  const int64_t g = (int64_t)img * a.rows_per_image + lv.row_off + row0 + lane;
  if (flag) { a.row_score[g] = my_s; a.row_class[g] = my_c; }
  const unsigned long long m = __ballot(flag);
  if (lane == 0) { a.wave_count[wid] = __popcll(m); a.wave_mask[wid] = m; }
}

// exclusive scan of one int per thread across a 1024-thread block (wave shuffles + one LDS hop); returns the
// exclusive prefix of `v`, *total = block sum
__device__ __forceinline__ int block_exclusive_scan_1024(int v, int* total) {
  __shared__ int wsum[16];
  __shared__ int wtot;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  if (wave == 0) {
    int w = lane < 16 ? wsum[lane] : 0;
    int winc = w;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
      const int t = __shfl_up(winc, o, 64);
      if (lane >= o) winc += t;
    }
    if (lane < 16) wsum[lane] = winc - w;  // exclusive prefix of the wave totals
    if (lane == 15) wtot = winc;
  }
  __syncthreads();
  const int ex = wsum[wave] + inc - v;
  *total = wtot;
  __syncthreads();
  return ex;
}

// ---- 2. exclusive scan of the wave counts (a few 10^4 entries) in ONE launch of independent blocks: block b scans its 1024
// entries and adds the sum of every entry in front of its chunk, which it forms itself (b x 1024 coalesced 4-byte loads
// from L2, eight in flight per thread: ~50 k loads for the last block of the 1024^2 x 16 shape, 2-3 us) -- no second
// launch for the chunk totals, no block waits for another.  Block 0 also clears what the next stages count into.
// (A single block walking a contiguous run per thread took 77 us at that shape -- 64 uncoalesced lines per load
// instruction through one CU's address path; the two-launch chunk scan this replaces: 4.8 + 10.8 us.)
__global__ __launch_bounds__(1024) void det_offsets_kernel(const DetArgs a) {
  __shared__ int part[16];
  const int64_t nw = (int64_t)a.n * a.waves_per_image;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (blockIdx.x == 0) {
    for (int k = tid; k < a.n * a.C; k += 1024) { a.seg_count[k] = 0; a.seg_fill[k] = 0; }   // emit counts, scatter fills
    for (int k = tid; k < 1 + a.n; k += 1024) a.counts[1 + k] = 0;
  }
  const int64_t begin = (int64_t)blockIdx.x * 1024;
  int before = 0;
  for (int64_t k0 = tid; k0 < begin; k0 += 8 * 1024) {
    int v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = a.wave_count[min(k0 + u * 1024, begin - 1)];
#pragma unroll
    for (int u = 0; u < 8; ++u) before += (k0 + u * 1024 < begin) ? v[u] : 0;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o, 64);
  if (lane == 0) part[wave] = before;
  __syncthreads();
  int base = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) base += part[w];
  const int64_t i = begin + tid;
  const int v = i < nw ? a.wave_count[i] : 0;
  int total;
  const int ex = block_exclusive_scan_1024(v, &total);
  if (i < nw) a.wave_off[i] = base + ex;
  if (blockIdx.x == gridDim.x - 1 && tid == 0) a.counts[0] = (int64_t)base + total;
}

// ---- anchor decode (utils.py:108-117)
__device__ __forceinline__ float cell_center(int i, int size) {
  const float cell = (float)(1.0 / (double)size);
  const float start = cell / 2.0f;
  if (size == 1) return start;
  const float stop = 1.0f - start;
  const float step = (stop - start) / (float)(size - 1);
  const float t = step * (float)i;
  return start + t;
}

// one anchor's box from its raw regression (utils.regression_postprocess, utils.py:100-117): the ONLY copy of this
// arithmetic -- the full-map kernel and the candidate-only path of det_emit_kernel both call it (same bits)
__device__ __forceinline__ float4 decode_one(const float4 r, float ah, float aw, int y_, int x_, int h, int w) {
  const float sy = r.x * ah, sx = r.y * aw;
  const float bh = expf(r.z) * ah, bw = expf(r.w) * aw;
  const float cy = sy + cell_center(y_, h), cx = sx + cell_center(x_, w);
  const float hh = bh / 2.0f, hw = bw / 2.0f;
  return make_float4(cy - hh, cx - hw, cy + hh, cx + hw);
}

constexpr int EMIT_MAXSEG = 8192;
// ---- 3. emit candidates in anchor order; pad the key array with sentinels
__global__ __launch_bounds__(T) void det_emit_kernel(const DetArgs a) {
  // a few thousand blocks walk the waves' masks (most are empty: 12 bytes read, nothing to do) instead of one block per four
  // waves -- 12 276 blocks of almost no work each took 37 us at the 1024^2 x 16 shape, bound by the block dispatch rate.
  // A block's waves are a CONTIGUOUS run (its four waves interleaved inside it): with many candidates the per-segment counts
  // are then formed in LDS and reach seg_count as one atomic per (block, segment it met) -- a run lies in one or two images --
  // instead of one per candidate (3.14 M atomics on n * C counters: 0.47 ms on the all-candidates input).
  extern __shared__ int hist[];                    // n * C counters in the many-candidates mode, nothing otherwise (the host sizes it)
  const int lane = threadIdx.x & 63;
  const int64_t nw = (int64_t)a.n * a.waves_per_image;
  const int nseg = a.n * a.C;
  // which of the two: decided from the caller's capacity (a kernel argument: no load in front of the walk) -- a capacity of an
  // eighth of the rows or more announces the many-candidates case; the ~1 % case keeps the strided walk and one atomic per candidate
  const bool use_hist = nseg <= EMIT_MAXSEG && a.cap * 8 >= (int64_t)a.n * a.rows_per_image;
  int64_t w_begin, w_end, stride;
  if (use_hist) {
    for (int k = threadIdx.x; k < nseg; k += T) hist[k] = 0;
    __syncthreads();
    const int64_t per_blk = (nw + gridDim.x - 1) / gridDim.x;
    w_begin = (int64_t)blockIdx.x * per_blk + (threadIdx.x >> 6);
    w_end = (int64_t)blockIdx.x * per_blk + per_blk < nw ? (int64_t)blockIdx.x * per_blk + per_blk : nw;
    stride = T >> 6;
  } else {
    w_begin = ((int64_t)blockIdx.x * T + threadIdx.x) >> 6; w_end = nw; stride = ((int64_t)gridDim.x * T) >> 6;
  }
  // the masks and offsets of the next EMIT_AHEAD waves this wave will visit are fetched together (a lane each), so the
  // chain per visited wave is one round trip to memory (score / class / raw box), not two
  constexpr int EMIT_AHEAD = 8;
  for (int64_t wid0 = w_begin; wid0 < w_end; wid0 += stride * EMIT_AHEAD) {
    const int64_t wl = wid0 + (int64_t)min(lane, EMIT_AHEAD - 1) * stride;
    const unsigned long long mine = (lane < EMIT_AHEAD && wl < w_end) ? a.wave_mask[wl] : 0ull;
    const int offmine = (lane < EMIT_AHEAD && wl < w_end) ? a.wave_off[wl] : 0;
#pragma unroll
    for (int u = 0; u < EMIT_AHEAD; ++u) {
    const int64_t wid = wid0 + (int64_t)u * stride;
    const unsigned long long m = __shfl(mine, u, 64);    // (wave-uniform)
    const int woff = __shfl(offmine, u, 64);
    if (m == 0ull || !((m >> lane) & 1ull)) continue;
    int img, l; int64_t row0;
    locate_wave32(a, (uint32_t)wid, &img, &l, &row0);     // (32-bit divisions: a 64-bit one is ~100 instructions)
    const DetLevel& lv = a.lv[l];
    const int64_t in_img = lv.row_off + row0 + lane;
    const int64_t g = (int64_t)img * a.rows_per_image + in_img;
    const float s = a.row_score[g];
    const int c = a.row_class[g];
    const int64_t pos = (int64_t)woff + __popcll(m & ((1ull << lane) - 1ull));
    if (pos >= a.cap) continue;  // overflow is reported through counts[0] > capacity
    float4 b;
    if (lv.boxes) {
      b = *reinterpret_cast<const float4*>(lv.boxes + ((size_t)img * lv.rows + row0 + lane) * 4);
    } else {  // decode this candidate only
      uint32_t q = (uint32_t)(row0 + lane);             // (rows per level < 2^31: host-checked)
      const int an = (int)(q % (uint32_t)lv.A); q /= (uint32_t)lv.A;
      const int x_ = (int)(q % (uint32_t)lv.w);
      const int y_ = (int)(q / (uint32_t)lv.w);
      const size_t ri = ((size_t)img * lv.rows + row0 + lane) * 4;
      float4 r;
      if (lv.half_reg) {
        const rn::rn_half4 h = *reinterpret_cast<const rn::rn_half4*>(reinterpret_cast<const _Float16*>(lv.reg) + ri);
        r = make_float4((float)h.x, (float)h.y, (float)h.z, (float)h.w);
      } else {
        r = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(lv.reg) + ri);
      }
      b = decode_one(r, lv.anch[an * 2], lv.anch[an * 2 + 1], y_, x_, lv.h, lv.w);
    }
    *reinterpret_cast<float4*>(a.cand_box + pos * 4) = b;
    a.cand_score[pos] = s; a.cand_class[pos] = c; a.cand_image[pos] = img; a.cand_anchor[pos] = in_img;
    if (use_hist) atomicAdd(&hist[img * a.C + c], 1);
    else atomicAdd(&a.seg_count[img * a.C + c], 1);   // integer count: the same whatever the order of arrival
    }
  }
  if (use_hist) {
    __syncthreads();
    for (int k = threadIdx.x; k < nseg; k += T) {
      const int c = hist[k];
      if (c) atomicAdd(&a.seg_count[k], c);
    }
  }
}

// order-preserving map of ANY float onto unsigned integers (scores handed to rn_nms_classwise may be <= 0)
__device__ __forceinline__ uint32_t float_order(float s) {
  const uint32_t sb = __float_as_uint(s);
  return (sb & 0x80000000u) ? ~sb : (sb | 0x80000000u);
}

// ---- 4a + 4b. every candidate takes a slot of its segment's range (the order inside the range is irrelevant: 4c sorts it).
// Every block forms the exclusive scan of the per-segment counts itself (nseg = n * C ints through LDS: 1280 at the
// 1024^2 x 16 shape) instead of waiting for a one-block scan launch; block 0 publishes seg_start[0..nseg] for the stages
// behind it.  seg_fill was cleared by det_offsets_kernel / det_zero_seg_kernel.
constexpr int SCAT_T = 256, SCAT_MAXSEG = 8192;
__global__ __launch_bounds__(SCAT_T) void det_seg_scatter_kernel(const DetArgs a) {
  __shared__ int start[SCAT_MAXSEG + 1];
  __shared__ int tsum[SCAT_T];
  const int nseg = a.n * a.C, tid = threadIdx.x;
  const int per = (nseg + SCAT_T - 1) / SCAT_T;
  const int b = tid * per, e = min(b + per, nseg);
  int local = 0;
  for (int k = b; k < e; ++k) { const int c = a.seg_count[k]; start[k] = c; local += c; }
  tsum[tid] = local;
  __syncthreads();
  if (tid < 64) {                                  // exclusive scan of the 256 thread totals by one wave (4 per lane)
    int v[4], run = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) { v[u] = tsum[tid * 4 + u]; run += v[u]; }
    int inc = run;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(inc, o, 64);
      if (tid >= o) inc += t;
    }
    int ex = inc - run;
#pragma unroll
    for (int u = 0; u < 4; ++u) { tsum[tid * 4 + u] = ex; ex += v[u]; }
    if (tid == 63) start[nseg] = inc;
  }
  __syncthreads();
  int run = tsum[tid];
  for (int k = b; k < e; ++k) { const int c = start[k]; start[k] = run; run += c; }
  __syncthreads();
  if (blockIdx.x == 0)
    for (int k = tid; k <= nseg; k += SCAT_T) a.seg_start[k] = start[k];
  const int64_t ncand = a.counts[0] < a.cap ? a.counts[0] : a.cap;
  const int64_t per_blk = (ncand + gridDim.x - 1) / gridDim.x;
  if (per_blk < 4 * SCAT_T) {                      // few candidates (a trained detector's ~1 %): one global atomic each is cheapest
    for (int64_t i = (int64_t)blockIdx.x * SCAT_T + tid; i < ncand; i += (int64_t)gridDim.x * SCAT_T) {
      const int seg = a.cand_image[i] * a.C + a.cand_class[i];
      const int slot = start[seg] + atomicAdd(&a.seg_fill[seg], 1);
      a.keys[slot] = ((uint64_t)(0xFFFFFFFFu - float_order(a.cand_score[i])) << 32) | (uint64_t)(uint32_t)i;
    }
    return;
  }
  // many candidates (every anchor a candidate: 3.14 M atomics on n * C counters took 1.0 ms): a block takes a CONTIGUOUS run
  // of candidates (one image's: <= C segments), counts them per segment in LDS, reserves each segment's share with ONE
  // global atomic, then hands out the slots from LDS.  (The order inside a segment is irrelevant: the sort follows.)
  __shared__ int lcnt[SCAT_MAXSEG];
  const int64_t c0 = (int64_t)blockIdx.x * per_blk, c1 = c0 + per_blk < ncand ? c0 + per_blk : ncand;
  for (int k = tid; k < nseg; k += SCAT_T) lcnt[k] = 0;
  __syncthreads();                                 // (also: block 0 has published seg_start before `start` is reused below)
  for (int64_t i = c0 + tid; i < c1; i += SCAT_T) atomicAdd(&lcnt[a.cand_image[i] * a.C + a.cand_class[i]], 1);
  __syncthreads();
  for (int k = tid; k < nseg; k += SCAT_T) {
    const int c = lcnt[k];
    if (c) { start[k] += atomicAdd(&a.seg_fill[k], c); lcnt[k] = 0; }
  }
  __syncthreads();
  for (int64_t i = c0 + tid; i < c1; i += SCAT_T) {
    const int seg = a.cand_image[i] * a.C + a.cand_class[i];
    const int slot = start[seg] + atomicAdd(&lcnt[seg], 1);
    a.keys[slot] = ((uint64_t)(0xFFFFFFFFu - float_order(a.cand_score[i])) << 32) | (uint64_t)(uint32_t)i;
  }
}
// (more than SCAT_MAXSEG segments: the scan as its own launch, the scatter reading it from global memory)
__global__ __launch_bounds__(1024) void det_seg_scan_kernel(const DetArgs a) {
  const int nseg = a.n * a.C;
  const int per = (nseg + 1023) / 1024;
  const int b = threadIdx.x * per, e = min(b + per, nseg);
  int local = 0;
  for (int k = b; k < e; ++k) local += a.seg_count[k];
  int total;
  int run = block_exclusive_scan_1024(local, &total);
  for (int k = b; k < e; ++k) {
    a.seg_start[k] = run;
    run += a.seg_count[k];
  }
  if (threadIdx.x == 0) a.seg_start[nseg] = total;
}
__global__ void det_seg_scatter_global_kernel(const DetArgs a) {
  const int64_t ncand = a.counts[0] < a.cap ? a.counts[0] : a.cap;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ncand; i += (int64_t)gridDim.x * blockDim.x) {
    const int seg = a.cand_image[i] * a.C + a.cand_class[i];
    const int slot = a.seg_start[seg] + atomicAdd(&a.seg_fill[seg], 1);
    a.keys[slot] = ((uint64_t)(0xFFFFFFFFu - float_order(a.cand_score[i])) << 32) | (uint64_t)(uint32_t)i;
  }
}

// ---- 4c. one block per segment: ascending bitonic sort of its 64-bit keys = score descending, then lower candidate
// position (= lower anchor index) first.  All-ascending formulation (each merge starts with the "flip" step i <-> i^(k-1),
// then half-cleaners i <-> i^j): elements only ever move towards the end when they are larger, so positions >= m behave as
// +infinity without being stored -- ragged lengths need no padding.  Up to SORT_LDS keys are sorted in LDS, longer
// segments in place in global memory (same network, block-wide barriers; a single-class NMS over a whole image).
constexpr int SORT_T = 256, SORT_LDS = 8192;
template <typename P>
__device__ __forceinline__ void bitonic_ascending(P keys, int m) {
  int p2 = 1;
  while (p2 < m) p2 <<= 1;
  for (int k = 2; k <= p2; k <<= 1) {
    for (int i = threadIdx.x; i < p2; i += SORT_T) {
      const int l = i ^ (k - 1);
      if (l > i && l < m) {
        const uint64_t x = keys[i], y = keys[l];
        if (x > y) { keys[i] = y; keys[l] = x; }
      }
    }
    __syncthreads();
    for (int j = k >> 2; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < p2; i += SORT_T) {
        const int l = i ^ j;
        if (l > i && l < m) {
          const uint64_t x = keys[i], y = keys[l];
          if (x > y) { keys[i] = y; keys[l] = x; }
        }
      }
      __syncthreads();
    }
  }
}
__global__ __launch_bounds__(SORT_T) void det_seg_sort_kernel(const DetArgs a) {
  __shared__ uint64_t sk[SORT_LDS];
  const int k = blockIdx.x;
  const int s0 = a.seg_start[k], m = a.seg_start[k + 1] - s0;
  if (m <= 0) return;
  uint64_t* g = a.keys + s0;
  if (m <= SORT_LDS) {
    for (int i = threadIdx.x; i < m; i += SORT_T) sk[i] = g[i];
    __syncthreads();
    bitonic_ascending(sk, m);
    for (int i = threadIdx.x; i < m; i += SORT_T) a.vals_out[s0 + i] = (uint32_t)sk[i];
  } else {
    __syncthreads();
    bitonic_ascending(g, m);   // same block wrote and reads these addresses: __syncthreads orders them
    for (int i = threadIdx.x; i < m; i += SORT_T) a.vals_out[s0 + i] = (uint32_t)g[i];
  }
}

// [TF-sem] IoU of TF's NonMaxSuppression kernel (corner normalisation, area<=0 -> 0, clamped extents)
struct NBox { float ymin, xmin, ymax, xmax, area; };
__device__ __forceinline__ NBox norm_box(const float4 b) {
  NBox r;
  r.ymin = fminf(b.x, b.z); r.ymax = fmaxf(b.x, b.z);
  r.xmin = fminf(b.y, b.w); r.xmax = fmaxf(b.y, b.w);
  r.area = (r.ymax - r.ymin) * (r.xmax - r.xmin);
  return r;
}
__device__ __forceinline__ bool suppresses(const NBox& i, const NBox& j, float thr) {
  if (i.area <= 0.f || j.area <= 0.f) return false;
  const float iy = fmaxf(fminf(i.ymax, j.ymax) - fmaxf(i.ymin, j.ymin), 0.f);
  const float ix = fmaxf(fminf(i.xmax, j.xmax) - fmaxf(i.xmin, j.xmin), 0.f);
  const float inter = iy * ix;
  const float v = inter / ((i.area + j.area) - inter);
  return v > thr;
}

// The same decision without the division wherever it is safe: v = inter / u > thr  <=>  inter > thr u up to rounding; outside a
// relative band of 2^-20 around equality the rounded quotient cannot land on the other side of thr (each of the roundings
// involved is <= 2^-23 relative), inside it the exact expression decides.  Bit-identical to `suppresses`.
__device__ __forceinline__ bool suppresses_fast(const NBox& i, const NBox& j, float thr) {
  const float iy = fmaxf(fminf(i.ymax, j.ymax) - fmaxf(i.ymin, j.ymin), 0.f);
  const float ix = fmaxf(fminf(i.xmax, j.xmax) - fmaxf(i.xmin, j.xmin), 0.f);
  const float inter = iy * ix;
  const float u = (i.area + j.area) - inter;
  const float t = thr * u;
  const bool valid = i.area > 0.f && j.area > 0.f;
  bool r = inter > t;
  if (fabsf(inter - t) <= 9.5367431640625e-07f * fabsf(t)) r = (inter / u) > thr;      // (rare: wave-divergent exact path)
  return valid && r;
}

// ---- 5. one block of four waves per (image, class) segment.  Greedy NMS is sequential in the candidates, but the test of a batch
// of 64 candidates against everything kept so far is not: the four waves each take a quarter of the kept boxes (interleaved in
// groups of four), their survivor masks are ANDed through LDS, and wave 0 resolves the batch in order.  The predicate is the same
// whatever the order it is evaluated in: index-for-index the results of the one-wave kernel (stress input: 1.24 -> see DESIGN).
constexpr int KEEP_LDS = 1024;
constexpr int NMS_WAVES = 4;
__global__ __launch_bounds__(64 * NMS_WAVES) void det_nms_kernel(const DetArgs a) {
  __shared__ NBox kept[KEEP_LDS];
  __shared__ unsigned long long masks[NMS_WAVES];
  __shared__ int s_nk;
  const int k = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int s0 = a.seg_start[k], s1 = a.seg_start[k + 1];
  int nk = 0;
  int* keep_out = a.keep_idx + (size_t)k * a.max_keep;
  for (int base = s0; base < s1 && nk < a.max_keep; base += 64) {
    const int idx = base + lane;
    const bool have = idx < s1;
    uint32_t cand = 0;
    NBox me = {0.f, 0.f, 0.f, 0.f, 0.f};
    if (have) {
      cand = a.vals_out[idx];
      me = norm_box(*reinterpret_cast<const float4*>(a.cand_box + (size_t)cand * 4));
    }
    bool alive = have;
    // against this wave's share of everything kept so far: four boxes per step, all loads of a step independent
    for (int j = 4 * wave; j < nk; j += 4 * NMS_WAVES) {
      const NBox k0 = kept[j], k1 = kept[min(j + 1, nk - 1)], k2 = kept[min(j + 2, nk - 1)], k3 = kept[min(j + 3, nk - 1)];
      const bool s0_ = suppresses_fast(k0, me, a.iou_thr), s1_ = suppresses_fast(k1, me, a.iou_thr);   // (a repeated last box changes nothing)
      const bool s2_ = suppresses_fast(k2, me, a.iou_thr), s3_ = suppresses_fast(k3, me, a.iou_thr);
      alive = alive && !(s0_ || s1_ || s2_ || s3_);
      if (__ballot(alive) == 0ull) break;
    }
    const unsigned long long mine = __ballot(alive);
    if (lane == 0) masks[wave] = mine;
    __syncthreads();
    if (wave == 0) {
      // resolve the 64 candidates in order
      unsigned long long live = masks[0] & masks[1] & masks[2] & masks[3];
      alive = ((live >> lane) & 1ull) != 0ull;
      while (live != 0ull && nk < a.max_keep) {
        const int i = __ffsll((long long)live) - 1;
        NBox bi;
        bi.ymin = __shfl(me.ymin, i, 64); bi.xmin = __shfl(me.xmin, i, 64);
        bi.ymax = __shfl(me.ymax, i, 64); bi.xmax = __shfl(me.xmax, i, 64); bi.area = __shfl(me.area, i, 64);
        if (lane == i) { kept[nk] = me; keep_out[nk] = (int)cand; alive = false; }
        if (alive && lane > i && suppresses_fast(bi, me, a.iou_thr)) alive = false;
        ++nk;
        live = __ballot(alive);
      }
      if (lane == 0) s_nk = nk;
    }
    __syncthreads();                               // kept[] and the count of this batch are visible to every wave's next checks
    nk = s_nk;
  }
  if (threadIdx.x == 0) a.seg_keep[k] = nk;
}

// ---- 6. gather the survivors.  A segment's output offset = the kept counts of the segments in front of it, summed by the
// block itself (n * C ints: 20 loads per lane at the 1024^2 x 16 shape) -- no one-block offsets launch in between; the
// first segment of every image also writes the image's total, block 0 the grand total.
__global__ __launch_bounds__(64) void det_gather_kernel(const DetArgs a) {
  const int k = blockIdx.x, lane = threadIdx.x;
  const int nseg = a.n * a.C;
  int before = 0;
  for (int j0 = lane; j0 < k; j0 += 64 * 8) {
    int v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = a.seg_keep[min(j0 + u * 64, nseg - 1)];
#pragma unroll
    for (int u = 0; u < 8; ++u) before += (j0 + u * 64 < k) ? v[u] : 0;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o, 64);
  if (k % a.C == 0) {                              // image total (and, for the first image's block, the grand total)
    const int img = k / a.C;
    int t = 0;
    for (int j = lane; j < a.C; j += 64) t += a.seg_keep[k + j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
    if (lane == 0) a.counts[2 + img] = t;
    if (k == 0) {
      int all = 0;
      for (int j = lane; j < nseg; j += 64) all += a.seg_keep[j];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) all += __shfl_xor(all, o, 64);
      if (lane == 0) a.counts[1] = all;
    }
  }
  const int nk = a.seg_keep[k], off = before;
  const int* keep = a.keep_idx + (size_t)k * a.max_keep;
  for (int i = lane; i < nk; i += 64) {
    const int c = keep[i];
    const int64_t o = (int64_t)off + i;
    *reinterpret_cast<float4*>(a.out_boxes + o * 4) = *reinterpret_cast<const float4*>(a.cand_box + (size_t)c * 4);
    a.out_scores[o] = a.cand_score[c]; a.out_class[o] = a.cand_class[c]; a.out_image[o] = a.cand_image[c];
    a.out_anchor[o] = a.cand_anchor[c];
  }
}

__global__ void decode_kernel(const float* __restrict__ reg, const float* __restrict__ anchors, float* __restrict__ out,
                              int n, int h, int w, int A) {
  const int64_t total = (int64_t)n * h * w * A;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t q = i;
    const int an = (int)(q % A); q /= A;
    const int x_ = (int)(q % w); q /= w;
    const int y_ = (int)(q % h);
    const float4 r = *reinterpret_cast<const float4*>(reg + i * 4);
    *reinterpret_cast<float4*>(out + i * 4) = decode_one(r, anchors[an * 2], anchors[an * 2 + 1], y_, x_, h, w);
  }
}

struct WsLayout { size_t off[20]; size_t total; };

int plan(const rn_det_level* levels, int nlevels, const rn_det_params* p, DetArgs* a, WsLayout* L) {
  RN_CHECK_ARG(levels && p && nlevels >= 1 && nlevels <= MAXL, "detect: bad levels");
  RN_CHECK_ARG(p->n >= 1 && p->num_classes >= 1 && p->max_per_class >= 1 && p->max_candidates >= 1, "detect: bad params");
  RN_UNSUPPORTED(p->max_per_class > KEEP_LDS, "detect: max_per_class %d > %d", p->max_per_class, KEEP_LDS);
  RN_UNSUPPORTED((int64_t)p->n * p->num_classes >= (1ll << 31) || p->max_candidates >= (1ll << 31), "detect: too large");
  a->nlv = nlevels; a->n = p->n; a->C = p->num_classes; a->max_keep = p->max_per_class;
  a->score_thr = p->score_threshold; a->iou_thr = p->iou_threshold; a->cap = p->max_candidates;
  int64_t rows = 0; int waves = 0;
  for (int l = 0; l < nlevels; ++l) {
    RN_CHECK_ARG(levels[l].rows_per_image >= 1, "detect: empty level %d", l);
    a->lv[l].prob = levels[l].prob; a->lv[l].boxes = levels[l].boxes; a->lv[l].rows = levels[l].rows_per_image;
    a->lv[l].reg = levels[l].regression; a->lv[l].anch = levels[l].anchor_sizes;
    a->lv[l].h = levels[l].grid_h; a->lv[l].w = levels[l].grid_w; a->lv[l].A = levels[l].num_anchors;
    a->lv[l].half_prob = levels[l].prob_f16 ? 1 : 0; a->lv[l].half_reg = levels[l].regression_f16 ? 1 : 0;
    a->lv[l].logit = levels[l].prob_is_logit ? 1 : 0;
    if (!levels[l].boxes && levels[l].regression)
      RN_CHECK_ARG(levels[l].anchor_sizes && levels[l].grid_h >= 1 && levels[l].grid_w >= 1 && levels[l].num_anchors >= 1 &&
                       (int64_t)levels[l].grid_h * levels[l].grid_w * levels[l].num_anchors == levels[l].rows_per_image,
                   "detect: level %d: regression grid %d x %d x %d does not match %lld rows", l, levels[l].grid_h,
                   levels[l].grid_w, levels[l].num_anchors, (long long)levels[l].rows_per_image);
    a->lv[l].row_off = rows; a->lv[l].wave_off = waves;
    rows += levels[l].rows_per_image;
    waves += (int)((levels[l].rows_per_image + 63) / 64);
  }
  a->rows_per_image = rows; a->waves_per_image = waves;
  const int64_t nrows = rows * p->n, nw = (int64_t)waves * p->n, cap = p->max_candidates;
  RN_UNSUPPORTED(rows >= (1ll << 31) || nw >= (1ll << 31), "detect: %lld rows per image / %lld waves: the emit pass indexes them in 32 bits", (long long)rows, (long long)nw);
  const int64_t nseg = (int64_t)p->n * p->num_classes;
  const size_t sizes[18] = {
      (size_t)nrows * 4, (size_t)nrows * 4, (size_t)nw * 4, (size_t)nw * 4,            // row_score,row_class,wave_count,wave_off
      (size_t)cap * 8, (size_t)cap * 4, (size_t)nseg * 4, (size_t)nseg * 4,            // keys, vals_out, seg_count, seg_fill
      (size_t)cap * 16, (size_t)cap * 4, (size_t)cap * 4, (size_t)cap * 4, (size_t)cap * 8,  // cand box,score,class,image,anchor
      (size_t)(nseg + 1) * 4, (size_t)nseg * 4, (size_t)nseg * 4, (size_t)nseg * p->max_per_class * 4,   // seg_*, keep_idx
      (size_t)nw * 8};                                                                                    // wave_mask
  size_t o = 0;
  for (int i = 0; i < 18; ++i) { L->off[i] = o; o += rn::align_up(sizes[i], 256); }
  L->total = o;
  return RN_OK;
}

void bind(DetArgs* a, const WsLayout& L, void* ws) {
  char* b = (char*)ws;
  a->row_score = (float*)(b + L.off[0]); a->row_class = (int32_t*)(b + L.off[1]);
  a->wave_count = (int32_t*)(b + L.off[2]); a->wave_off = (int32_t*)(b + L.off[3]);
  a->keys = (uint64_t*)(b + L.off[4]); a->vals_out = (uint32_t*)(b + L.off[5]);
  a->seg_count = (int32_t*)(b + L.off[6]); a->seg_fill = (int32_t*)(b + L.off[7]);
  a->cand_box = (float*)(b + L.off[8]); a->cand_score = (float*)(b + L.off[9]);
  a->cand_class = (int32_t*)(b + L.off[10]); a->cand_image = (int32_t*)(b + L.off[11]);
  a->cand_anchor = (int64_t*)(b + L.off[12]);
  a->seg_start = (int32_t*)(b + L.off[13]); a->seg_keep = (int32_t*)(b + L.off[14]);
  a->seg_off = (int32_t*)(b + L.off[15]); a->keep_idx = (int32_t*)(b + L.off[16]);
  a->wave_mask = (unsigned long long*)(b + L.off[17]);
}
}  // namespace

extern "C" int rn_decode_boxes(const float* reg, const float* anchor_sizes, float* boxes, int n, int h, int w,
                               int num_anchors, rn_stream_t stream) {
  RN_CHECK_ARG(reg && anchor_sizes && boxes && n >= 1 && h >= 1 && w >= 1 && num_anchors >= 1, "decode: bad argument");
  const int64_t total = (int64_t)n * h * w * num_anchors;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(decode_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, reg, anchor_sizes, boxes, n,
                     h, w, num_anchors);
  RN_LAUNCH_CHECK();
  return RN_OK;
}

extern "C" size_t rn_detect_workspace(const rn_det_level* levels, int nlevels, const rn_det_params* p) {
  DetArgs a = {};
  WsLayout L;
  if (plan(levels, nlevels, p, &a, &L)) return 0;
  return L.total;
}

namespace {
// copy the (un-sorted) candidates to the outputs in anchor order (boxes_decode semantics)
__global__ void det_copy_candidates_kernel(const DetArgs a) {
  const int64_t ncand = a.counts[0] < a.cap ? a.counts[0] : a.cap;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ncand; i += (int64_t)gridDim.x * blockDim.x) {
    *reinterpret_cast<float4*>(a.out_boxes + i * 4) = *reinterpret_cast<const float4*>(a.cand_box + i * 4);
    a.out_scores[i] = a.cand_score[i]; a.out_class[i] = a.cand_class[i]; a.out_image[i] = a.cand_image[i];
    a.out_anchor[i] = a.cand_anchor[i];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) a.counts[1] = ncand;
}

// candidates handed in as arrays (rn_nms_classwise): count them per segment
__global__ void det_zero_seg_kernel(const DetArgs a) {
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < a.n * a.C; k += gridDim.x * blockDim.x) { a.seg_count[k] = 0; a.seg_fill[k] = 0; }
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < 1 + a.n; k += gridDim.x * blockDim.x) a.counts[1 + k] = 0;
}
__global__ void det_count_arrays_kernel(const DetArgs a, const int64_t* count_dev) {
  const int64_t ncand = count_dev[0] < a.cap ? count_dev[0] : a.cap;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ncand; i += (int64_t)gridDim.x * blockDim.x) {
    atomicAdd(&a.seg_count[a.cand_image[i] * a.C + a.cand_class[i]], 1);
    a.cand_anchor[i] = i;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) a.counts[0] = count_dev[0];
}

int sort_and_suppress(DetArgs& a, const WsLayout& L, void* workspace, hipStream_t st) {
  const int nseg = a.n * a.C;
  if (nseg <= SCAT_MAXSEG) {
    hipLaunchKernelGGL(det_seg_scatter_kernel, dim3(256), dim3(SCAT_T), 0, st, a);
  } else {
    hipLaunchKernelGGL(det_seg_scan_kernel, dim3(1), dim3(1024), 0, st, a);
    hipLaunchKernelGGL(det_seg_scatter_global_kernel, dim3(256), dim3(256), 0, st, a);
  }
  hipLaunchKernelGGL(det_seg_sort_kernel, dim3(nseg), dim3(SORT_T), 0, st, a);
  hipLaunchKernelGGL(det_nms_kernel, dim3(nseg), dim3(64 * NMS_WAVES), 0, st, a);
  hipLaunchKernelGGL(det_gather_kernel, dim3(nseg), dim3(64), 0, st, a);
  RN_LAUNCH_CHECK();
  return RN_OK;
}

int run_detect(const rn_det_level* levels, int nlevels, const rn_det_params* p, float* out_boxes, float* out_scores,
               int32_t* out_class, int32_t* out_image, int64_t* out_anchor, int64_t* counts, void* workspace,
               size_t workspace_bytes, rn_stream_t stream, bool decode_only) {
  DetArgs a = {};
  WsLayout L;
  if (int e = plan(levels, nlevels, p, &a, &L)) return e;
  RN_CHECK_ARG(out_boxes && out_scores && out_class && out_image && out_anchor && counts && workspace, "detect: null pointer");
  for (int l = 0; l < nlevels; ++l)
    RN_CHECK_ARG(levels[l].prob && (levels[l].boxes || levels[l].regression), "detect: null level pointer %d", l);
  if (workspace_bytes < L.total) {
    rn::set_error("detect: workspace %zu < %zu", workspace_bytes, L.total);
    return RN_EWORKSPACE;
  }
  bind(&a, L, workspace);
  a.out_boxes = out_boxes; a.out_scores = out_scores; a.out_class = out_class; a.out_image = out_image;
  a.out_anchor = out_anchor; a.counts = counts;
  hipStream_t st = (hipStream_t)stream;
  const int64_t nw = (int64_t)a.n * a.waves_per_image;
  const unsigned wblocks = (unsigned)((nw * 64 + T - 1) / T);
  {
    // LDS-staged scan when every level's rows are whole 16-byte chunks and 64 of them (padded) fit a wave's LDS share
    bool lds_ok = getenv("RN_SCAN_NO_LDS") == nullptr;
    for (int l = 0; l < a.nlv; ++l) {
      const int esz = a.lv[l].half_prob ? 2 : 4, v = 16 / esz;
      lds_ok = lds_ok && a.C % v == 0 && 64 * (((a.C / v) | 1) * 16) <= SCAN_LDS_WAVE;
    }
    // all levels in one storage type / logit mode with whole 16-byte chunks: the kernel specialised for that case
    bool uniform = true;
    for (int l = 1; l < a.nlv; ++l) uniform = uniform && a.lv[l].half_prob == a.lv[0].half_prob && a.lv[l].logit == a.lv[0].logit;
    const int v0 = a.lv[0].half_prob ? 8 : 4;
    const int qpl = a.C % v0 == 0 ? (a.C / v0 + 3) / 4 : 0;
    lds_ok = lds_ok && getenv("RN_SCAN_LDS") != nullptr;   // measured slower than the specialised kernels: opt-in only
    // fp16 maps of 80 classes (cfg 5): the transposing scan; RN_SCAN_T=0: the four-lanes-per-row kernels
    static const bool scan_t = !(getenv("RN_SCAN_T") && atoi(getenv("RN_SCAN_T")) == 0);
    if (scan_t && uniform && a.lv[0].half_prob && a.C == 80 && !lds_ok) {
      constexpr size_t tl = (size_t)(T / 64) * 32 * ((10 | 1) * 16);
      static const bool nt = !(getenv("RN_SCAN_NT") && atoi(getenv("RN_SCAN_NT")) == 0);   // (0: A/B measurements)
      if (!nt) {
        if (a.lv[0].logit) hipLaunchKernelGGL((det_scan_t_kernel<10, true, false>), dim3(wblocks), dim3(T), tl, st, a);
        else hipLaunchKernelGGL((det_scan_t_kernel<10, false, false>), dim3(wblocks), dim3(T), tl, st, a);
      } else if (a.lv[0].logit) hipLaunchKernelGGL((det_scan_t_kernel<10, true>), dim3(wblocks), dim3(T), tl, st, a);
      else hipLaunchKernelGGL((det_scan_t_kernel<10, false>), dim3(wblocks), dim3(T), tl, st, a);
    } else if (uniform && qpl >= 1 && qpl <= 8 && !lds_ok) {
#define RN_SCAN_Q(H_, L_)                                                                                         \
      switch (qpl) {                                                                                              \
        case 1: hipLaunchKernelGGL((det_scan_q_kernel<H_, L_, 1>), dim3(wblocks), dim3(T), 0, st, a); break;      \
        case 2: hipLaunchKernelGGL((det_scan_q_kernel<H_, L_, 2>), dim3(wblocks), dim3(T), 0, st, a); break;      \
        case 3: hipLaunchKernelGGL((det_scan_q_kernel<H_, L_, 3>), dim3(wblocks), dim3(T), 0, st, a); break;      \
        case 4: hipLaunchKernelGGL((det_scan_q_kernel<H_, L_, 4>), dim3(wblocks), dim3(T), 0, st, a); break;      \
        case 5: hipLaunchKernelGGL((det_scan_q_kernel<H_, L_, 5>), dim3(wblocks), dim3(T), 0, st, a); break;      \
        case 6: hipLaunchKernelGGL((det_scan_q_kernel<H_, L_, 6>), dim3(wblocks), dim3(T), 0, st, a); break;      \
        case 7: hipLaunchKernelGGL((det_scan_q_kernel<H_, L_, 7>), dim3(wblocks), dim3(T), 0, st, a); break;      \
        default: hipLaunchKernelGGL((det_scan_q_kernel<H_, L_, 8>), dim3(wblocks), dim3(T), 0, st, a); break;     \
      }
      if (a.lv[0].half_prob) { if (a.lv[0].logit) { RN_SCAN_Q(true, true) } else { RN_SCAN_Q(true, false) } }
      else { if (a.lv[0].logit) { RN_SCAN_Q(false, true) } else { RN_SCAN_Q(false, false) } }
#undef RN_SCAN_Q
    } else if (lds_ok) {
      static const hipError_t attr_ = hipFuncSetAttribute((const void*)det_scan_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                          (T / 64) * SCAN_LDS_WAVE);
      (void)attr_;
      size_t need = 0;
      for (int l = 0; l < a.nlv; ++l) {
        const int esz = a.lv[l].half_prob ? 2 : 4, v = 16 / esz;
        const size_t w = (size_t)64 * (((a.C / v) | 1) * 16);
        need = need > w ? need : w;
      }
      hipLaunchKernelGGL(det_scan_lds_kernel, dim3(wblocks), dim3(T), (size_t)(T / 64) * need, st, a, (int)need);
    } else {
      hipLaunchKernelGGL(det_scan_kernel, dim3(wblocks), dim3(T), 0, st, a);
    }
  }
  hipLaunchKernelGGL(det_offsets_kernel, dim3((unsigned)((nw + 1023) / 1024)), dim3(1024), 0, st, a);
  {
    const int nseg_ = a.n * a.C;
    const bool hist = nseg_ <= EMIT_MAXSEG && a.cap * 8 >= (int64_t)a.n * a.rows_per_image;     // (the kernel's own rule)
    hipLaunchKernelGGL(det_emit_kernel, dim3(wblocks < 2048u ? wblocks : 2048u), dim3(T), hist ? (size_t)nseg_ * sizeof(int) : 0, st, a);
  }
  if (decode_only) {
    hipLaunchKernelGGL(det_copy_candidates_kernel, dim3(256), dim3(256), 0, st, a);
    RN_LAUNCH_CHECK();
    return RN_OK;
  }
  return sort_and_suppress(a, L, workspace, st);
}
}  // namespace

extern "C" int rn_detect(const rn_det_level* levels, int nlevels, const rn_det_params* p, float* out_boxes,
                         float* out_scores, int32_t* out_class, int32_t* out_image, int64_t* out_anchor, int64_t* counts,
                         void* workspace, size_t workspace_bytes, rn_stream_t stream) {
  return run_detect(levels, nlevels, p, out_boxes, out_scores, out_class, out_image, out_anchor, counts, workspace,
                    workspace_bytes, stream, false);
}

extern "C" int rn_boxes_decode(const rn_det_level* levels, int nlevels, const rn_det_params* p, float* out_boxes,
                               float* out_scores, int32_t* out_class, int32_t* out_image, int64_t* out_anchor,
                               int64_t* counts, void* workspace, size_t workspace_bytes, rn_stream_t stream) {
  return run_detect(levels, nlevels, p, out_boxes, out_scores, out_class, out_image, out_anchor, counts, workspace,
                    workspace_bytes, stream, true);
}

extern "C" size_t rn_nms_classwise_workspace(const rn_det_params* p) {
  if (!p) return 0;
  rn_det_level lv = {nullptr, nullptr, 64};
  DetArgs a = {};
  WsLayout L;
  if (plan(&lv, 1, p, &a, &L)) return 0;
  return L.total;
}

extern "C" int rn_nms_classwise(const float* boxes, const float* scores, const int32_t* class_ids, const int32_t* image_ids,
                                const int64_t* count_dev, const rn_det_params* p, float* out_boxes, float* out_scores,
                                int32_t* out_class, int32_t* out_image, int64_t* out_index, int64_t* counts, void* workspace,
                                size_t workspace_bytes, rn_stream_t stream) {
  RN_CHECK_ARG(boxes && scores && class_ids && image_ids && count_dev && p, "nms_classwise: null input");
  rn_det_level lv = {nullptr, nullptr, 64};
  DetArgs a = {};
  WsLayout L;
  if (int e = plan(&lv, 1, p, &a, &L)) return e;
  RN_CHECK_ARG(out_boxes && out_scores && out_class && out_image && out_index && counts && workspace, "nms_classwise: null output");
  if (workspace_bytes < L.total) {
    rn::set_error("nms_classwise: workspace %zu < %zu", workspace_bytes, L.total);
    return RN_EWORKSPACE;
  }
  bind(&a, L, workspace);
  // candidates are the caller's arrays (read-only use)
  a.cand_box = const_cast<float*>(boxes); a.cand_score = const_cast<float*>(scores);
  a.cand_class = const_cast<int32_t*>(class_ids); a.cand_image = const_cast<int32_t*>(image_ids);
  a.out_boxes = out_boxes; a.out_scores = out_scores; a.out_class = out_class; a.out_image = out_image;
  a.out_anchor = out_index; a.counts = counts;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(det_zero_seg_kernel, dim3(16), dim3(256), 0, st, a);
  hipLaunchKernelGGL(det_count_arrays_kernel, dim3(256), dim3(256), 0, st, a, count_dev);
  RN_LAUNCH_CHECK();
  return sort_and_suppress(a, L, workspace, st);
}
