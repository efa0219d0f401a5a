// Fused detection loss over all anchors of all pyramid levels (losses.py:155-175 behind
// utils.process_labels_and_logits, utils.py:240-284).
//
// The reference compacts the trainable rows with boolean_mask and concatenates P3..P7; the
// same numbers fall out of treating the mask as a 0/1 row weight, so nothing is gathered:
//   M      = #trainable rows, fg = trainable row whose max label > 0.5 (utils.py:171-179)
//   bce_dice: mean_{M,C} BCE + mean_C (1 - 2 I_c / (L_c + P_c))          (losses.py:124-139)
//   focal   : sum focal / max(#fg, 1)                                     (losses.py:6-15,119-122)
//   regr    : sum_{fg,4} huber_1(label - pred) / (4 #fg), 0 if no fg      (losses.py:144-152)
// One wave owns one row at a time (lanes stride over the classes, coalesced 4-B loads of a
// contiguous row), so the per-class dice sums stay in registers.  Two passes: reduce -> stats,
// then an elementwise gradient pass that reads the stats.  Fixed-order reduction (per-wave
// registers -> LDS -> per-block partial -> fp64 finalize): bitwise reproducible.
#include "rn_common.h"

namespace {

constexpr int T = 256;
constexpr int WAVES = T / 64;
constexpr int MAXJ = 4;          // classes per lane => C <= 256
constexpr int NSCAL = 5;         // M, fg, bce, focal, huber
constexpr int BLOCKS = 2048;      // 8 waves per SIMD: the per-row load -> math chain is latency-bound

struct LossSeg {
  const float* zl; const float* ll; const float* rp; const float* rl; const uint8_t* tm;
  float* dz; float* dr;
  int64_t rows; int64_t row_start;
};
struct LossArgs {
  LossSeg seg[RN_MAX_SEG];
  int nseg, C, mode;
  int64_t total_rows;
  float* partial;  // [BLOCKS][NSCAL + 3*C]
  float* stats;
  float* cls_out; float* reg_out;  // optional separate scalars (forward)
  const float* g_cls; const float* g_reg;
};

__device__ __forceinline__ int seg_of_row(const LossArgs& a, int64_t r) {
  int s = 0;
  while (s + 1 < a.nseg && r >= a.seg[s + 1].row_start) ++s;
  return s;
}
__device__ __forceinline__ float sigmoidf(float z) { return 1.f / (1.f + expf(-z)); }
// d huber_1(l - p) / dp = clip(p - l, -1, 1); written with compares so that a NaN target stays NaN (fminf / fmaxf drop it)
__device__ __forceinline__ float clip1(float e) { return e < -1.f ? -1.f : (e > 1.f ? 1.f : e); }
// Masking rule of BOTH kernel families, forward and backward = the reference's: rows outside the trainable mask are REMOVED
// (boolean_mask, utils.py:270-278: a select -- whatever they hold never reaches a sum); the foreground weight of the Huber term
// is a MULTIPLY (tf.losses.huber_loss(weights=...), losses.py:146-152: a NaN target on a trainable background row gives
// NaN * 0 = NaN in the loss and in the gradient, as in the reference graph).
__device__ __forceinline__ float huber1(float e) {
  const float a = fabsf(e), q = fminf(a, 1.f);
  return 0.5f * q * q + (a - q);
}

__global__ __launch_bounds__(T) void loss_reduce_kernel(const LossArgs a) {
  __shared__ float red[WAVES][NSCAL + 3 * 64 * MAXJ];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int C = a.C;
  float I[MAXJ], L[MAXJ], P[MAXJ];
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) I[j] = L[j] = P[j] = 0.f;
  float s_m = 0.f, s_fg = 0.f, s_bce = 0.f, s_focal = 0.f, s_hub = 0.f;

  const int64_t wave_id = (int64_t)blockIdx.x * WAVES + wave, nwaves = (int64_t)gridDim.x * WAVES;
  for (int64_t r = wave_id; r < a.total_rows; r += nwaves) {
    const int s = seg_of_row(a, r);
    const LossSeg& sg = a.seg[s];
    const int64_t lr = r - sg.row_start;
    if (!sg.tm[lr]) continue;  // wave-uniform
    const float* __restrict__ zrow = sg.zl + lr * C;
    const float* __restrict__ lrow = sg.ll + lr * C;
    float lmax = -1e30f;
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
      const int c = lane + 64 * j;
      if (c < C) {
        const float z = zrow[c], l = lrow[c];
        const float p = sigmoidf(z);
        lmax = fmaxf(lmax, l);
        I[j] += l * p; L[j] += l; P[j] += p;
        if (a.mode == RN_LOSS_BCE_DICE) {
          s_bce += fmaxf(z, 0.f) - z * l + log1pf(expf(-fabsf(z)));
        } else {
          const bool pos = (l == 1.f);
          const float pt = pos ? p : 1.f - p;
          const float al = pos ? 0.25f : 0.75f;
          const float om = 1.f - pt;
          s_focal += -al * om * om * logf(pt + 1e-7f);
        }
      }
    }
    lmax = rn::wave_max(lmax);
    const bool fg = lmax > 0.5f;
    if (lane == 0) { s_m += 1.f; s_fg += fg ? 1.f : 0.f; }
    if (lane < 4) s_hub = fmaf(fg ? 1.f : 0.f, huber1(sg.rl[lr * 4 + lane] - sg.rp[lr * 4 + lane]), s_hub);
  }
  // wave-level sums of the scalars
  s_m = rn::wave_sum(s_m); s_fg = rn::wave_sum(s_fg); s_bce = rn::wave_sum(s_bce);
  s_focal = rn::wave_sum(s_focal); s_hub = rn::wave_sum(s_hub);
  if (lane == 0) { red[wave][0] = s_m; red[wave][1] = s_fg; red[wave][2] = s_bce; red[wave][3] = s_focal; red[wave][4] = s_hub; }
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) {
    const int c = lane + 64 * j;
    red[wave][NSCAL + 3 * c + 0] = I[j];
    red[wave][NSCAL + 3 * c + 1] = L[j];
    red[wave][NSCAL + 3 * c + 2] = P[j];
  }
  __syncthreads();
  const int n = NSCAL + 3 * C;
  for (int i = tid; i < n; i += T) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) v += red[w][i];
    a.partial[(size_t)blockIdx.x * n + i] = v;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// C % 4 == 0, C <= 128 (the COCO / VOC class counts): FOUR lanes per row, 16 rows per wave and pass.  Lane (row, j) owns the
// class quads j, j + 4, ..: NK 16-byte loads of the logits and NK of the labels, all issued before the first is used, two
// row groups per pass (4 NK loads in flight per lane); the row's four lanes also are the four box coordinates.  The
// wave-per-row kernels above read 4 bytes per lane with one row in flight per wave: 44 us for 63 MB at the headline shape.
// Same sums (other order), same `partial` layout.
constexpr int RPW = 16;          // rows per wave and pass
template <int NK, int MODE>
__global__ __launch_bounds__(T) void loss_reduce4_kernel(const LossArgs a) {
  __shared__ float red[WAVES][NSCAL + 3 * 16 * NK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int C = a.C, CQ = C >> 2, j = lane & 3, r16 = lane >> 2;
  float I[NK][4], L[NK][4], P[NK][4];
#pragma unroll
  for (int k = 0; k < NK; ++k)
#pragma unroll
    for (int e = 0; e < 4; ++e) I[k][e] = L[k][e] = P[k][e] = 0.f;
  float s_m = 0.f, s_fg = 0.f, s_cls = 0.f, s_hub = 0.f;
  const int64_t ngroups = (a.total_rows + RPW - 1) / RPW;
  const int64_t wave_id = (int64_t)blockIdx.x * WAVES + wave, nwaves = (int64_t)gridDim.x * WAVES;
  for (int64_t g0 = wave_id * 2; g0 < ngroups; g0 += nwaves * 2) {
    float4 z[2][NK], l[2][NK];
    float rl[2], rp[2], tmv[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int64_t r = min((g0 + u) * RPW + r16, a.total_rows - 1);
      const bool live = (g0 + u) * RPW + r16 < a.total_rows;
      const LossSeg& sg = a.seg[seg_of_row(a, r)];
      const int64_t lr = r - sg.row_start;
      tmv[u] = (live && sg.tm[lr]) ? 1.f : 0.f;
      rl[u] = sg.rl[lr * 4 + j]; rp[u] = sg.rp[lr * 4 + j];
#pragma unroll
      for (int k = 0; k < NK; ++k) {
        const int q = min(j + 4 * k, CQ - 1);
        z[u][k] = *reinterpret_cast<const float4*>(sg.zl + lr * C + q * 4);
        l[u][k] = *reinterpret_cast<const float4*>(sg.ll + lr * C + q * 4);
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      float lmax = -1e30f, cls = 0.f;
#pragma unroll
      for (int k = 0; k < NK; ++k) {
        const float w = (j + 4 * k < CQ) ? tmv[u] : 0.f;        // 0: the quad does not exist, or the row is not trainable
        const float zz[4] = {z[u][k].x, z[u][k].y, z[u][k].z, z[u][k].w}, ll[4] = {l[u][k].x, l[u][k].y, l[u][k].z, l[u][k].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float zv = zz[e], lv = ll[e];
          const float p = sigmoidf(zv);
          if (j + 4 * k < CQ) lmax = fmaxf(lmax, lv);
          if (MODE == RN_LOSS_BCE_DICE) {
            I[k][e] = fmaf(w, lv * p, I[k][e]); L[k][e] = fmaf(w, lv, L[k][e]); P[k][e] = fmaf(w, p, P[k][e]);
            cls = fmaf(w, fmaxf(zv, 0.f) - zv * lv + log1pf(expf(-fabsf(zv))), cls);
          } else {
            const bool pos = (lv == 1.f);
            const float pt = pos ? p : 1.f - p;
            const float al = pos ? 0.25f : 0.75f;
            const float om = 1.f - pt;
            cls = fmaf(w, -al * om * om * logf(pt + 1e-7f), cls);
          }
        }
      }
      lmax = fmaxf(lmax, __shfl_xor(lmax, 1, 64));
      lmax = fmaxf(lmax, __shfl_xor(lmax, 2, 64));
      const float fgw = (lmax > 0.5f) ? 1.f : 0.f;
      const float fg = fgw * tmv[u];
      s_cls += cls;
      if (j == 0) { s_m += tmv[u]; s_fg += fg; }
      const float hub = fgw * huber1(rl[u] - rp[u]);             // multiply: the foreground weight
      s_hub += (tmv[u] != 0.f) ? hub : 0.f;                      // select: the trainable mask
    }
  }
  s_m = rn::wave_sum(s_m); s_fg = rn::wave_sum(s_fg); s_cls = rn::wave_sum(s_cls); s_hub = rn::wave_sum(s_hub);
  if (lane == 0) {
    red[wave][0] = s_m; red[wave][1] = s_fg;
    red[wave][2] = MODE == RN_LOSS_BCE_DICE ? s_cls : 0.f; red[wave][3] = MODE == RN_LOSS_BCE_DICE ? 0.f : s_cls;
    red[wave][4] = s_hub;
  }
  // per-class sums (the dice term's; the focal mode neither needs nor forms them: zeros): over the 16 row lanes that share j
#pragma unroll
  for (int k = 0; k < NK; ++k)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float vi = I[k][e], vl = L[k][e], vp = P[k][e];
      if (MODE == RN_LOSS_BCE_DICE) {
#pragma unroll
        for (int m = 4; m < 64; m <<= 1) { vi += __shfl_xor(vi, m, 64); vl += __shfl_xor(vl, m, 64); vp += __shfl_xor(vp, m, 64); }
      }
      const int c = 4 * (j + 4 * k) + e;
      if (lane < 4 && j + 4 * k < CQ) { red[wave][NSCAL + 3 * c + 0] = vi; red[wave][NSCAL + 3 * c + 1] = vl; red[wave][NSCAL + 3 * c + 2] = vp; }
    }
  __syncthreads();
  const int n = NSCAL + 3 * C;
  for (int i = tid; i < n; i += T) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) v += red[w][i];
    a.partial[(size_t)blockIdx.x * n + i] = v;
  }
}

template <int NK, int MODE>
__global__ __launch_bounds__(T) void loss_grad4_kernel(const LossArgs a) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int C = a.C, CQ = C >> 2, j = lane & 3, r16 = lane >> 2;
  const float M = a.stats[2], nfg = a.stats[3];
  const float gc = a.g_cls[0], gr = a.g_reg[0];
  const float k_bce = gc / (M * (float)C);
  const float k_dice = gc * 2.f / (float)C;
  const float k_focal = gc / fmaxf(nfg, 1.f);
  const float k_reg = nfg > 0.f ? gr / (4.f * nfg) : 0.f;
  float Ic[NK][4], Uc[NK][4];
  if (MODE == RN_LOSS_BCE_DICE) {
#pragma unroll
    for (int k = 0; k < NK; ++k)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = min(4 * (j + 4 * k) + e, C - 1);
        Ic[k][e] = a.stats[RN_LOSS_STATS_HEADER + 3 * c];
        Uc[k][e] = a.stats[RN_LOSS_STATS_HEADER + 3 * c + 1] + a.stats[RN_LOSS_STATS_HEADER + 3 * c + 2];
      }
  }
  const int64_t ngroups = (a.total_rows + RPW - 1) / RPW;
  const int64_t wave_id = (int64_t)blockIdx.x * WAVES + wave, nwaves = (int64_t)gridDim.x * WAVES;
  for (int64_t g0 = wave_id * 2; g0 < ngroups; g0 += nwaves * 2) {
    float4 z[2][NK], l[2][NK];
    float rl[2], rp[2], tmv[2];
    float* dzp[2]; float* drp[2]; bool live[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int64_t r = min((g0 + u) * RPW + r16, a.total_rows - 1);
      live[u] = (g0 + u) * RPW + r16 < a.total_rows;
      const LossSeg& sg = a.seg[seg_of_row(a, r)];
      const int64_t lr = r - sg.row_start;
      tmv[u] = (live[u] && sg.tm[lr]) ? 1.f : 0.f;
      rl[u] = sg.rl[lr * 4 + j]; rp[u] = sg.rp[lr * 4 + j];
      dzp[u] = sg.dz + lr * C; drp[u] = sg.dr + lr * 4 + j;
#pragma unroll
      for (int k = 0; k < NK; ++k) {
        const int q = min(j + 4 * k, CQ - 1);
        z[u][k] = *reinterpret_cast<const float4*>(sg.zl + lr * C + q * 4);
        l[u][k] = *reinterpret_cast<const float4*>(sg.ll + lr * C + q * 4);
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      float lmax = -1e30f;
#pragma unroll
      for (int k = 0; k < NK; ++k) {
        const float zz[4] = {z[u][k].x, z[u][k].y, z[u][k].z, z[u][k].w}, ll[4] = {l[u][k].x, l[u][k].y, l[u][k].z, l[u][k].w};
        float g[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float zv = zz[e], lv = ll[e];
          const float p = sigmoidf(zv);
          if (j + 4 * k < CQ) lmax = fmaxf(lmax, lv);
          if (MODE == RN_LOSS_BCE_DICE) {
            const float U = Uc[k][e];
            g[e] = k_bce * (p - lv) - k_dice * p * (1.f - p) * (lv * U - Ic[k][e]) / (U * U);
          } else {
            const bool pos = (lv == 1.f);
            const float pt = pos ? p : 1.f - p;
            const float al = pos ? 0.25f : 0.75f;
            const float om = 1.f - pt;
            const float dfdpt = -al * (-2.f * om * logf(pt + 1e-7f) + om * om / (pt + 1e-7f));
            g[e] = k_focal * dfdpt * (pos ? 1.f : -1.f) * p * (1.f - p);
          }
          g[e] = tmv[u] != 0.f ? g[e] : 0.f;
        }
        if (live[u] && j + 4 * k < CQ) *reinterpret_cast<float4*>(dzp[u] + (j + 4 * k) * 4) = make_float4(g[0], g[1], g[2], g[3]);
      }
      lmax = fmaxf(lmax, __shfl_xor(lmax, 1, 64));
      lmax = fmaxf(lmax, __shfl_xor(lmax, 2, 64));
      if (live[u]) {
        float g = 0.f;
        if (tmv[u] != 0.f) g = ((lmax > 0.5f) ? 1.f : 0.f) * (k_reg * clip1(rp[u] - rl[u]));
        *drp[u] = g;
      }
    }
  }
}

// stage 1: sums[i] = sum over blocks of partial[b][i]; 16 values x 16 row lanes per block, fixed order
__global__ __launch_bounds__(256) void loss_sum_kernel(const LossArgs a, int nblocks, double* __restrict__ sums) {
  __shared__ double sh[16][16];
  const int n = NSCAL + 3 * a.C;
  const int il = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int i = blockIdx.x * 16 + il;
  double v = 0.0;
  if (i < n) {
    int b = rl;
    for (; b + 16 * 7 < nblocks; b += 16 * 8) {  // eight loads in flight, added in row order
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = a.partial[(size_t)(b + 16 * u) * n + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) v += (double)t[u];
    }
    for (; b < nblocks; b += 16) v += (double)a.partial[(size_t)b * n + i];
  }
  sh[rl][il] = v;
  __syncthreads();
  if (rl == 0 && i < n) {
    double t = 0.0;
#pragma unroll
    for (int r = 0; r < 16; ++r) t += sh[r][il];
    sums[i] = t;
  }
}

// stage 2: loss values and the per-class statistics the gradient pass needs
__global__ void loss_finalize_kernel(const LossArgs a, const double* __restrict__ sh) {
  __shared__ double dice_part[256];
  for (int i = threadIdx.x; i < 3 * a.C; i += blockDim.x) a.stats[RN_LOSS_STATS_HEADER + i] = (float)sh[NSCAL + i];
  double d = 0.0;
  for (int c = threadIdx.x; c < a.C; c += blockDim.x) {
    const double I = sh[NSCAL + 3 * c], U = sh[NSCAL + 3 * c + 1] + sh[NSCAL + 3 * c + 2];
    d += 1.0 - 2.0 * I / U;
  }
  dice_part[threadIdx.x] = d;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double M = sh[0], fg = sh[1];
    double cls;
    if (a.mode == RN_LOSS_BCE_DICE) {
      double dice = 0.0;
      for (int t = 0; t < (int)blockDim.x; ++t) dice += dice_part[t];
      cls = sh[2] / (M * a.C) + dice / a.C;
    } else {
      cls = sh[3] / (fg > 1.0 ? fg : 1.0);
    }
    const double reg = fg > 0.0 ? sh[4] / (4.0 * fg) : 0.0;
    a.stats[0] = (float)cls; a.stats[1] = (float)reg; a.stats[2] = (float)M; a.stats[3] = (float)fg;
    if (a.cls_out) *a.cls_out = (float)cls;
    if (a.reg_out) *a.reg_out = (float)reg;
    a.stats[4] = (float)sh[2]; a.stats[5] = (float)sh[3]; a.stats[6] = (float)sh[4]; a.stats[7] = 0.f;
  }
}

__global__ __launch_bounds__(T) void loss_grad_kernel(const LossArgs a) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int C = a.C;
  const float M = a.stats[2], nfg = a.stats[3];
  const float gc = a.g_cls[0], gr = a.g_reg[0];
  const float k_bce = gc / (M * (float)C);
  const float k_dice = gc * 2.f / (float)C;
  const float k_focal = gc / fmaxf(nfg, 1.f);
  const float k_reg = nfg > 0.f ? gr / (4.f * nfg) : 0.f;
  float Ic[MAXJ], Uc[MAXJ];
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) {
    const int c = lane + 64 * j;
    Ic[j] = c < C ? a.stats[RN_LOSS_STATS_HEADER + 3 * c] : 0.f;
    Uc[j] = c < C ? a.stats[RN_LOSS_STATS_HEADER + 3 * c + 1] + a.stats[RN_LOSS_STATS_HEADER + 3 * c + 2] : 1.f;
  }
  const int64_t wave_id = (int64_t)blockIdx.x * WAVES + wave, nwaves = (int64_t)gridDim.x * WAVES;
  for (int64_t r = wave_id; r < a.total_rows; r += nwaves) {
    const int s = seg_of_row(a, r);
    const LossSeg& sg = a.seg[s];
    const int64_t lr = r - sg.row_start;
    float* __restrict__ dzrow = sg.dz + lr * C;
    if (!sg.tm[lr]) {
#pragma unroll
      for (int j = 0; j < MAXJ; ++j) { const int c = lane + 64 * j; if (c < C) dzrow[c] = 0.f; }
      if (lane < 4) sg.dr[lr * 4 + lane] = 0.f;
      continue;
    }
    const float* __restrict__ zrow = sg.zl + lr * C;
    const float* __restrict__ lrow = sg.ll + lr * C;
    float lmax = -1e30f;
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
      const int c = lane + 64 * j;
      if (c < C) {
        const float z = zrow[c], l = lrow[c];
        const float p = sigmoidf(z);
        lmax = fmaxf(lmax, l);
        float g;
        if (a.mode == RN_LOSS_BCE_DICE) {
          // d/dz [bce/(M C)] = (p - l)/(M C);  d/dz [-(2/C) I/U] = -(2/C) p(1-p) (l U - I)/U^2
          const float U = Uc[j];
          g = k_bce * (p - l) - k_dice * p * (1.f - p) * (l * U - Ic[j]) / (U * U);
        } else {
          // f = -al (1-pt)^2 log(pt+eps); dpt/dz = +-p(1-p)
          const bool pos = (l == 1.f);
          const float pt = pos ? p : 1.f - p;
          const float al = pos ? 0.25f : 0.75f;
          const float om = 1.f - pt;
          const float dfdpt = -al * (-2.f * om * logf(pt + 1e-7f) + om * om / (pt + 1e-7f));
          g = k_focal * dfdpt * (pos ? 1.f : -1.f) * p * (1.f - p);
        }
        dzrow[c] = g;
      }
    }
    lmax = rn::wave_max(lmax);
    if (lane < 4) {
      const float e = sg.rp[lr * 4 + lane] - sg.rl[lr * 4 + lane];    // d huber(l - p)/dp = clip(p - l)
      sg.dr[lr * 4 + lane] = ((lmax > 0.5f) ? 1.f : 0.f) * (k_reg * clip1(e));
    }
  }
}

int build(const rn_loss_seg* segs, int nseg, int C, int mode, LossArgs* a, bool bwd) {
  RN_CHECK_ARG(segs && nseg >= 1 && nseg <= RN_MAX_SEG, "loss: bad segments");
  RN_UNSUPPORTED(C < 1 || C > 64 * MAXJ, "loss: num_classes %d outside [1,%d]", C, 64 * MAXJ);
  RN_CHECK_ARG(mode == RN_LOSS_BCE_DICE || mode == RN_LOSS_FOCAL, "loss: bad mode %d", mode);
  a->nseg = nseg; a->C = C; a->mode = mode;
  int64_t rows = 0;
  for (int s = 0; s < nseg; ++s) {
    RN_CHECK_ARG(segs[s].cls_logit && segs[s].cls_label && segs[s].reg_pred && segs[s].reg_label && segs[s].trainable &&
                 segs[s].rows >= 0, "loss: null pointer in segment %d", s);
    if (bwd) RN_CHECK_ARG(segs[s].d_cls_logit && segs[s].d_reg_pred, "loss bwd: null gradient buffer in segment %d", s);
    LossSeg& d = a->seg[s];
    d.zl = segs[s].cls_logit; d.ll = segs[s].cls_label; d.rp = segs[s].reg_pred; d.rl = segs[s].reg_label;
    d.tm = segs[s].trainable; d.dz = segs[s].d_cls_logit; d.dr = segs[s].d_reg_pred;
    d.rows = segs[s].rows; d.row_start = rows;
    rows += segs[s].rows;
  }
  a->total_rows = rows;
  return RN_OK;
}

// the four-lanes-per-row kernels: C a multiple of 4, at most 128 (RN_LOSS_WAVE_PER_ROW=1: the wave-per-row kernels, measurements)
bool use4(int C) {
  static const bool off = getenv("RN_LOSS_WAVE_PER_ROW") && atoi(getenv("RN_LOSS_WAVE_PER_ROW"));
  return !off && C % 4 == 0 && C >= 4 && C <= 128;
}
int nblocks4(int64_t rows) {      // one pass of two row groups per wave
  int64_t b = (rows + RPW * 2 * WAVES - 1) / (RPW * 2 * WAVES);
  if (b > BLOCKS) b = BLOCKS;
  if (b < 1) b = 1;
  return (int)b;
}
template <bool REDUCE, int NK>
void launch4_nk(const LossArgs& a, int mode, hipStream_t st) {
  const dim3 grid(nblocks4(a.total_rows)), block(T);
  if (REDUCE) {
    if (mode == RN_LOSS_BCE_DICE) hipLaunchKernelGGL((loss_reduce4_kernel<NK, RN_LOSS_BCE_DICE>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((loss_reduce4_kernel<NK, RN_LOSS_FOCAL>), grid, block, 0, st, a);
  } else {
    if (mode == RN_LOSS_BCE_DICE) hipLaunchKernelGGL((loss_grad4_kernel<NK, RN_LOSS_BCE_DICE>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((loss_grad4_kernel<NK, RN_LOSS_FOCAL>), grid, block, 0, st, a);
  }
}
template <bool REDUCE>
bool launch4(const LossArgs& a, int C, int mode, hipStream_t st) {
  if (!use4(C)) return false;
  switch ((C / 4 + 3) / 4) {
    case 1: launch4_nk<REDUCE, 1>(a, mode, st); break;
    case 2: launch4_nk<REDUCE, 2>(a, mode, st); break;
    case 3: launch4_nk<REDUCE, 3>(a, mode, st); break;
    case 4: launch4_nk<REDUCE, 4>(a, mode, st); break;
    case 5: launch4_nk<REDUCE, 5>(a, mode, st); break;
    case 6: launch4_nk<REDUCE, 6>(a, mode, st); break;
    case 7: launch4_nk<REDUCE, 7>(a, mode, st); break;
    default: launch4_nk<REDUCE, 8>(a, mode, st); break;
  }
  return true;
}

int nblocks_for(int64_t rows) {
  int64_t b = (rows + WAVES * 4 - 1) / (WAVES * 4);
  if (b > BLOCKS) b = BLOCKS;
  if (b < 1) b = 1;
  return (int)b;
}
}  // namespace

extern "C" size_t rn_loss_workspace(const rn_loss_seg*, int, int num_classes) {
  return rn::align_up((size_t)BLOCKS * (NSCAL + 3 * (size_t)num_classes) * sizeof(float), 256) +
         (NSCAL + 3 * (size_t)num_classes) * sizeof(double);
}

extern "C" int rn_loss_fwd(const rn_loss_seg* segs, int nseg, int num_classes, int mode, float* stats, float* class_loss_out,
                           float* regr_loss_out, void* workspace, size_t workspace_bytes, rn_stream_t stream) {
  LossArgs a = {};
  if (int e = build(segs, nseg, num_classes, mode, &a, false)) return e;
  RN_CHECK_ARG(stats && workspace, "loss fwd: null stats/workspace");
  if (workspace_bytes < rn_loss_workspace(segs, nseg, num_classes)) {
    rn::set_error("loss fwd: workspace too small");
    return RN_EWORKSPACE;
  }
  a.partial = (float*)workspace; a.stats = stats; a.cls_out = class_loss_out; a.reg_out = regr_loss_out;
  hipStream_t st = (hipStream_t)stream;
  const int nb = use4(num_classes) ? nblocks4(a.total_rows) : nblocks_for(a.total_rows);
  if (!launch4<true>(a, num_classes, mode, st)) hipLaunchKernelGGL(loss_reduce_kernel, dim3(nb), dim3(T), 0, st, a);
  double* sums = (double*)((char*)workspace + rn::align_up((size_t)BLOCKS * (NSCAL + 3 * (size_t)num_classes) * sizeof(float), 256));
  hipLaunchKernelGGL(loss_sum_kernel, dim3(rn::ceil_div(NSCAL + 3 * num_classes, 16)), dim3(256), 0, st, a, nb, sums);
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, st, a, (const double*)sums);
  RN_LAUNCH_CHECK();
  return RN_OK;
}

extern "C" int rn_loss_bwd(const rn_loss_seg* segs, int nseg, int num_classes, int mode, const float* stats,
                           const float* g_cls, const float* g_reg, rn_stream_t stream) {
  LossArgs a = {};
  if (int e = build(segs, nseg, num_classes, mode, &a, true)) return e;
  RN_CHECK_ARG(stats && g_cls && g_reg, "loss bwd: null stats/upstream gradient");
  a.stats = const_cast<float*>(stats); a.g_cls = g_cls; a.g_reg = g_reg;
  if (!launch4<false>(a, num_classes, mode, (hipStream_t)stream))
    hipLaunchKernelGGL(loss_grad_kernel, dim3(nblocks_for(a.total_rows)), dim3(T), 0, (hipStream_t)stream, a);
  RN_LAUNCH_CHECK();
  return RN_OK;
}
