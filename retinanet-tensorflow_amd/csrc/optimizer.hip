// Optimizer apply on one flat fp32 parameter arena (train.py:111-134,221):
//   g' = grad*grad_scale + wd*w            (L2 regulariser gradient, scale per parameter)
//   g' *= clip/max(||g'||, clip)           (tf.clip_by_global_norm, optional)
//   Momentum(0.9) | RMSProp(0.9, 0.9, 1e-10) | Adam(0.9, 0.999, 1e-8)   [TF-sem]
// HBM-bound: one pass, float4 accesses; 16-20 B/element.
#include "rn_common.h"

namespace {
constexpr int T = 256;
constexpr int NB = 1024;  // reduction blocks

__global__ __launch_bounds__(T) void norm_reg_kernel(const float* __restrict__ w, const float* __restrict__ g,
                                                     const float* __restrict__ wd, int64_t count, float gs,
                                                     double* __restrict__ partial) {
  __shared__ double red[2][T / 64];
  double n2 = 0.0, rg = 0.0;
  const int64_t nquad = count / 4;  // count is a multiple of RN_OPT_BLOCK
  for (int64_t i = (int64_t)blockIdx.x * T + threadIdx.x; i < nquad; i += (int64_t)gridDim.x * T) {
    const float d = wd[(i * 4) / RN_OPT_BLOCK];
    const float4 wv = *reinterpret_cast<const float4*>(w + i * 4);
    const float4 gv = *reinterpret_cast<const float4*>(g + i * 4);
    const float t0 = gv.x * gs + d * wv.x, t1 = gv.y * gs + d * wv.y, t2 = gv.z * gs + d * wv.z, t3 = gv.w * gs + d * wv.w;
    n2 += (double)(t0 * t0 + t1 * t1 + t2 * t2 + t3 * t3);
    rg += (double)(0.5f * d * (wv.x * wv.x + wv.y * wv.y + wv.z * wv.z + wv.w * wv.w));
  }
  n2 = rn::wave_sum_d(n2); rg = rn::wave_sum_d(rg);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[0][wave] = n2; red[1][wave] = rg; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0.0, b = 0.0;
    for (int i = 0; i < T / 64; ++i) { a += red[0][i]; b += red[1][i]; }
    partial[blockIdx.x * 2] = a; partial[blockIdx.x * 2 + 1] = b;
  }
}

__global__ void norm_reg_finalize_kernel(const double* __restrict__ partial, int nb, float* __restrict__ out2) {
  // one wave; lane-strided fixed-order sums
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < nb; i += 64) { a += partial[2 * i]; b += partial[2 * i + 1]; }
  a = rn::wave_sum_d(a); b = rn::wave_sum_d(b);
  if (threadIdx.x == 0) { out2[0] = (float)a; out2[1] = (float)b; }
}

// NORM: the pass that applies the update also forms this launch's share of (sum g'^2, L2 regulariser value) -- both at the
// weights BEFORE the update, as rn_grad_norm_l2reg does -- into partial[block] (no clipping then: the norm is not an input)
template <int KIND, bool NORM>
__global__ __launch_bounds__(T) void opt_step_kernel(float* __restrict__ w, const float* __restrict__ g,
                                                     float* __restrict__ s1, float* __restrict__ s2,
                                                     const float* __restrict__ wd, int64_t count, float lr, float gs,
                                                     float clip, const float* __restrict__ norm_sq, unsigned long long* advance, unsigned long long advance_by,
                                                     double* __restrict__ partial) {
  __shared__ double red[2][T / 64];
  double n2 = 0.0, rg = 0.0;
  if (advance && blockIdx.x == 0 && threadIdx.x == 0) *advance += advance_by;  // the step counter the dropout masks hash (fresh masks next step)
  float cs = 1.f;
  if (clip > 0.f) {
    const float gn = sqrtf(norm_sq[0]);
    cs = clip / fmaxf(gn, clip);
  }
  const int64_t nquad = count / 4;
  for (int64_t i = (int64_t)blockIdx.x * T + threadIdx.x; i < nquad; i += (int64_t)gridDim.x * T) {
    const float d = wd[(i * 4) / RN_OPT_BLOCK];
    float4 wv = *reinterpret_cast<float4*>(w + i * 4);
    const float4 gv = *reinterpret_cast<const float4*>(g + i * 4);
    float4 av = *reinterpret_cast<float4*>(s1 + i * 4);
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (KIND != RN_OPT_MOMENTUM) bv = *reinterpret_cast<float4*>(s2 + i * 4);
    float* wp = &wv.x; const float* gp = &gv.x; float* ap = &av.x; float* bp = &bv.x;
    if (NORM) {
      const float t0 = gv.x * gs + d * wv.x, t1 = gv.y * gs + d * wv.y, t2 = gv.z * gs + d * wv.z, t3 = gv.w * gs + d * wv.w;
      n2 += (double)(t0 * t0 + t1 * t1 + t2 * t2 + t3 * t3);
      rg += (double)(0.5f * d * (wv.x * wv.x + wv.y * wv.y + wv.z * wv.z + wv.w * wv.w));
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float gg = (gp[j] * gs + d * wp[j]) * cs;
      if (KIND == RN_OPT_MOMENTUM) {
        ap[j] = 0.9f * ap[j] + gg;
        wp[j] -= lr * ap[j];
      } else if (KIND == RN_OPT_RMSPROP) {
        ap[j] = 0.9f * ap[j] + 0.1f * gg * gg;
        bp[j] = 0.9f * bp[j] + lr * gg / sqrtf(ap[j] + 1e-10f);
        wp[j] -= bp[j];
      } else {
        ap[j] = 0.9f * ap[j] + 0.1f * gg;
        bp[j] = 0.999f * bp[j] + 0.001f * gg * gg;
        wp[j] -= lr * ap[j] / (sqrtf(bp[j]) + 1e-8f);  // lr already carries the bias correction
      }
    }
    *reinterpret_cast<float4*>(w + i * 4) = wv;
    *reinterpret_cast<float4*>(s1 + i * 4) = av;
    if (KIND != RN_OPT_MOMENTUM) *reinterpret_cast<float4*>(s2 + i * 4) = bv;
  }
  if (NORM) {
    n2 = rn::wave_sum_d(n2); rg = rn::wave_sum_d(rg);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[0][wave] = n2; red[1][wave] = rg; }
    __syncthreads();
    if (threadIdx.x == 0) {
      double a = 0.0, b = 0.0;
      for (int i = 0; i < T / 64; ++i) { a += red[0][i]; b += red[1][i]; }
      partial[blockIdx.x * 2] = a; partial[blockIdx.x * 2 + 1] = b;
    }
  }
}

unsigned grid_for(int64_t nquad) {
  int64_t b = (nquad + T - 1) / T;
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (unsigned)b;
}
}  // namespace

extern "C" size_t rn_optimizer_workspace(int64_t) { return (size_t)NB * 2 * sizeof(double); }

extern "C" int rn_grad_norm_l2reg(const float* w, const float* grad, const float* wd_per_block, int64_t count,
                                  float grad_scale, float* out2, void* workspace, size_t workspace_bytes,
                                  rn_stream_t stream) {
  RN_CHECK_ARG(w && grad && wd_per_block && out2 && workspace, "grad_norm: null pointer");
  RN_CHECK_ARG(count > 0 && count % RN_OPT_BLOCK == 0, "grad_norm: count %lld not a multiple of %d", (long long)count,
               RN_OPT_BLOCK);
  if (workspace_bytes < rn_optimizer_workspace(count)) { rn::set_error("grad_norm: workspace too small"); return RN_EWORKSPACE; }
  unsigned nb = grid_for(count / 4);
  if (nb > NB) nb = NB;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(norm_reg_kernel, dim3(nb), dim3(T), 0, st, w, grad, wd_per_block, count, grad_scale, (double*)workspace);
  hipLaunchKernelGGL(norm_reg_finalize_kernel, dim3(1), dim3(64), 0, st, (const double*)workspace, (int)nb, out2);
  RN_LAUNCH_CHECK();
  return RN_OK;
}

namespace {
int launch_opt(int kind, float* w, const float* grad, float* state1, float* state2, const float* wd_per_block, int64_t count, float lr,
               float grad_scale, float clip_norm, const float* norm_sq, int64_t step, uint64_t* advance_counter, uint64_t advance_by,
               double* partial, hipStream_t st) {
  RN_CHECK_ARG(w && grad && state1 && wd_per_block, "optimizer: null pointer");
  RN_CHECK_ARG(count > 0 && count % RN_OPT_BLOCK == 0, "optimizer: count %lld not a multiple of %d", (long long)count,
               RN_OPT_BLOCK);
  RN_CHECK_ARG(clip_norm <= 0.f || norm_sq, "optimizer: clipping needs norm_sq");
  RN_CHECK_ARG(kind == RN_OPT_MOMENTUM || state2, "optimizer: state2 required for rmsprop/adam");
  const unsigned nb = grid_for(count / 4);
  float lr_eff = lr;
  if (kind == RN_OPT_ADAM) {
    RN_CHECK_ARG(step >= 1, "optimizer: adam step must be >= 1");
    lr_eff = (float)((double)lr * sqrt(1.0 - pow(0.999, (double)step)) / (1.0 - pow(0.9, (double)step)));
  } else if (kind != RN_OPT_MOMENTUM && kind != RN_OPT_RMSPROP) {
    rn::set_error("optimizer: unknown kind %d", kind);
    return RN_EINVAL;
  }
#define RN_OPT_LAUNCH(KIND_)                                                                                                  \
  do {                                                                                                                        \
    if (partial) hipLaunchKernelGGL((opt_step_kernel<KIND_, true>), dim3(nb), dim3(T), 0, st, w, grad, state1, state2, wd_per_block, count, lr_eff, \
                                    grad_scale, clip_norm, norm_sq, (unsigned long long*)advance_counter, (unsigned long long)advance_by, partial); \
    else hipLaunchKernelGGL((opt_step_kernel<KIND_, false>), dim3(nb), dim3(T), 0, st, w, grad, state1, state2, wd_per_block, count, lr_eff, \
                            grad_scale, clip_norm, norm_sq, (unsigned long long*)advance_counter, (unsigned long long)advance_by, partial); \
  } while (0)
  if (kind == RN_OPT_MOMENTUM) RN_OPT_LAUNCH(RN_OPT_MOMENTUM);
  else if (kind == RN_OPT_RMSPROP) RN_OPT_LAUNCH(RN_OPT_RMSPROP);
  else RN_OPT_LAUNCH(RN_OPT_ADAM);
#undef RN_OPT_LAUNCH
  RN_LAUNCH_CHECK();
  return RN_OK;
}
}  // namespace

extern "C" int rn_optimizer_step(int kind, float* w, const float* grad, float* state1, float* state2,
                                 const float* wd_per_block, int64_t count, float lr, float grad_scale, float clip_norm,
                                 const float* norm_sq, int64_t step, uint64_t* advance_counter, uint64_t advance_by, rn_stream_t stream) {
  return launch_opt(kind, w, grad, state1, state2, wd_per_block, count, lr, grad_scale, clip_norm, norm_sq, step, advance_counter, advance_by,
                    nullptr, (hipStream_t)stream);
}

extern "C" int64_t rn_optimizer_norm_pairs(int64_t count) { return count > 0 ? (int64_t)grid_for(count / 4) : 0; }

extern "C" int rn_optimizer_step_norm(int kind, float* w, const float* grad, float* state1, float* state2, const float* wd_per_block,
                                      int64_t count, float lr, float grad_scale, int64_t step, uint64_t* advance_counter,
                                      uint64_t advance_by, double* partial, rn_stream_t stream) {
  RN_CHECK_ARG(partial, "optimizer step + norm: null partial buffer");
  return launch_opt(kind, w, grad, state1, state2, wd_per_block, count, lr, grad_scale, 0.f, nullptr, step, advance_counter, advance_by,
                    partial, (hipStream_t)stream);
}

extern "C" int rn_norm_reg_finalize(const double* partial, int64_t npairs, float* out2, rn_stream_t stream) {
  RN_CHECK_ARG(partial && out2 && npairs >= 1 && npairs < (1 << 30), "norm_reg_finalize: bad argument");
  hipLaunchKernelGGL(norm_reg_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, partial, (int)npairs, out2);
  RN_LAUNCH_CHECK();
  return RN_OK;
}

namespace {
__global__ void counter_add_kernel(unsigned long long* c, unsigned long long inc) { *c += inc; }
}  // namespace

extern "C" int rn_counter_add(uint64_t* counter, uint64_t inc, rn_stream_t stream) {
  RN_CHECK_ARG(counter, "counter_add: null pointer");
  hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned long long*)counter, (unsigned long long)inc);
  RN_LAUNCH_CHECK();
  return RN_OK;
}

namespace {
__global__ __launch_bounds__(T) void zero_kernel(float* __restrict__ p, int64_t count) {
  const int64_t nquad = count / 4;
  for (int64_t i = (int64_t)blockIdx.x * T + threadIdx.x; i < nquad; i += (int64_t)gridDim.x * T)
    *reinterpret_cast<float4*>(p + i * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
  if (blockIdx.x == 0 && threadIdx.x < (int)(count - nquad * 4)) p[nquad * 4 + threadIdx.x] = 0.f;
}
}  // namespace

extern "C" int rn_zero(float* p, int64_t count, rn_stream_t stream) {
  RN_CHECK_ARG(p && count >= 0 && ((uintptr_t)p & 15) == 0, "zero: null / unaligned pointer");
  if (count == 0) return RN_OK;
  hipLaunchKernelGGL(zero_kernel, dim3(grid_for(count / 4 + 1)), dim3(T), 0, (hipStream_t)stream, p, count);
  RN_LAUNCH_CHECK();
  return RN_OK;
}
