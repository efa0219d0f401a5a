// Shared helpers for the gfx950 kernels of librn_hip.so (not part of the public ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "rn_hip.h"

namespace rn {

void set_error(const char* fmt, ...);

#define RN_CHECK_ARG(cond, ...)            \
  do {                                     \
    if (!(cond)) {                         \
      rn::set_error(__VA_ARGS__);          \
      return RN_EINVAL;                    \
    }                                      \
  } while (0)

#define RN_UNSUPPORTED(cond, ...)          \
  do {                                     \
    if (cond) {                            \
      rn::set_error(__VA_ARGS__);          \
      return RN_EUNSUPPORTED;              \
    }                                      \
  } while (0)

#define RN_LAUNCH_CHECK()                                                     \
  do {                                                                        \
    hipError_t e__ = hipGetLastError();                                       \
    if (e__ != hipSuccess) {                                                  \
      rn::set_error("%s:%d HIP launch error: %s", __FILE__, __LINE__,         \
                    hipGetErrorString(e__));                                  \
      return RN_EHIP;                                                         \
    }                                                                         \
  } while (0)

// C_b [M x N] = A_b [M x K] * B_b (b_nk == 0: B_b is [K x N]; else [N x K]) for nbatch contiguous matrices, one
// launch of the implicit-GEMM conv kernels; defined in conv_gemm.hip.
int launch_batched_gemm(const float* A, const float* B, float* C, int M, int K, int N, int nbatch, int b_nk,
                        hipStream_t st);
// C_b [K x N] = A_b^T B_b for A_b [M x K], B_b [M x N]; split reduction through `workspace`, fixed order.
size_t batched_gemm_tn_workspace(int M, int K, int N, int nbatch);
// nsplit_out != nullptr: no final reduction; the partial products stay in workspace as [nsplit][nbatch][K][N].
int launch_batched_gemm_tn(const float* A, const float* B, float* C, int M, int K, int N, int nbatch, void* workspace,
                           size_t workspace_bytes, hipStream_t st, int* nsplit_out = nullptr, int background = 0);
// data-gradient product + weight-gradient partial products of a Winograd backward pass in one launch
int launch_winograd_bwd_products(const float* Ad, const float* Bd, float* Cd, int M, int Kd, int Nd, const float* Aw,
                                 const float* Bw, int Kw, int Nw, int nbatch, void* workspace, size_t workspace_bytes,
                                 hipStream_t st, int* nsplit_out);
// The same three products on the bf16 matrix cores from exact three-way splits of the fp32 operands (gemm_x3.hip); used by
// the functions above when product_mode() == 1 and gemm_x3_ok().
int product_mode();
void set_product_mode(int m);
bool gemm_x3_ok(int M, int K, int N);
int launch_batched_gemm_x3(const float* A, const float* B, float* C, int M, int K, int N, int nbatch, int b_nk, hipStream_t st);
size_t batched_gemm_tn_workspace_x3(int M, int K, int N, int nbatch);
// Op2 pre-split into three bf16 planes in MFMA-fragment order by its producer (gemm_x3_bfrag.hip FragB; RN_X3_BFRAG=0 / rn_set_x3_bfrag(0): off):
// the Winograd kernel transform writes U / Urot that way, the product kernel loads its B fragments straight from global memory
bool x3_bfrag_ok(int M, int K, int N);
bool x3_bfrag_format(int K, int N);             // the M-independent half of x3_bfrag_ok
int x3_tile_rows(long rows_a, long rows_b, long batches);      // (gemm_x3.hip's tile choice / size limit, for gemm_x3_bfrag.hip)
bool x3_fits(long elems);
size_t x3_bfrag_bytes(int K, int N);            // per batch matrix
int bfrag_on();
void set_bfrag(int on);
int launch_batched_gemm_x3_bfrag(const float* A, const void* Bfrag, float* C, int M, int K, int N, int nbatch, int fwd_name, hipStream_t st);
int launch_pack_bfrag(const float* B, void* out, int K, int N, int nbatch, int b_nk, hipStream_t st);
// dense 1x1 / stride-1 convolutions on the same kernels (0: not taken -- product mode 0, RN_X3_CONV1X1=0, or the shape; else the m-tile rows)
int conv1x1_x3_tile(long M, int Cin, int Cout, int x_ld);
int launch_conv1x1_fwd_x3(const float* x, int x_ld, const float* w, float* y, int M, int Cin, int Cout, float2* stat_rows, hipStream_t st,
                          float drop_rate = 0.f, uint64_t drop_seed = 0, const uint64_t* drop_seed_dev = nullptr);
int launch_conv1x1_dgrad_x3(const float* dy, const float* w, float* dx, int dx_ld, int M, int Cin, int Cout, hipStream_t st);
size_t conv1x1_wgrad_workspace_x3(int M, int Cin, int Cout);
int launch_conv1x1_wgrad_x3(const float* x, int x_ld, const float* dy, int M, int Cin, int Cout, void* workspace, size_t workspace_bytes,
                            hipStream_t st, int* nsplit_out);
// dense k x k convs (not 1 x 1 / stride 1) as split-bf16 products of an explicit patch matrix (im2col.hip); bytes == 0: not taken
size_t im2col_x3_bytes(int n, int h, int wd, int cin, int cout, int kh, int kw, int stride);
size_t im2col_x3_fwd_pieces(int n, int h, int wd, int cin, int cout, int kh, int kw, int stride, int* n_piece);
int launch_im2col(const float* x, float* col, int n, int h, int wd, int cin, int kh, int kw, int stride, hipStream_t st);
int launch_col2im(const float* dcol, float* dx, int n, int h, int wd, int cin, int kh, int kw, int stride, hipStream_t st);
// grouped 3x3 / stride-1 convs with 4 / 8 / 16 / 32 channels per group as direct convolutions (grouped_conv.hip)
bool gconv3x3_ok(int n, int h, int wd, int c, int groups);
int launch_gconv3x3(const float* x, const float* w, float* y, int n, int h, int wd, int c, int groups, int transpose, hipStream_t st);
size_t gconv3x3_wgrad_workspace(int n, int h, int wd, int c, int groups);
int launch_gconv3x3_wgrad(const float* x, const float* dy, float* dw, int accumulate, int n, int h, int wd, int c, int groups, void* workspace,
                          size_t workspace_bytes, hipStream_t st);
int launch_batched_gemm_tn_x3(const float* A, const float* B, int M, int K, int N, int nbatch, void* workspace, size_t workspace_bytes,
                              hipStream_t st, int* nsplit_out, int background = 0);
// Entry points that end with a row reduction open one of these with their `defer` argument: while it is alive (this
// call, this thread) launch_reduce_rows records into the caller's list instead of launching.
struct DeferScope {
  explicit DeferScope(rn_reduce_list* list);
  ~DeferScope();
  void* prev_;
};
// true while the current call records its row reductions for rn_flush_reductions instead of launching them
bool reduce_deferred(hipStream_t st);
// out[i] = (accumulate ? out[i] : 0) + sum_r in[r][i]  (r = 0..nrows-1, fixed order => reproducible).
// 16 float4 columns x 16 row lanes per block; defined in conv_gemm.hip.
int launch_reduce_rows(const float* in, float* out, int64_t count, int nrows, int accumulate, hipStream_t st);

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// TF SAME padding
static inline void same_pad(int n, int k, int s, int* out, int* before) {
  int o = (n + s - 1) / s;
  int total = (o - 1) * s + k - n;
  if (total < 0) total = 0;
  *out = o;
  *before = total / 2;
}

// Blocks are dealt round-robin over the 8 XCDs (each with a private L2).  Remap the linear
// block id so that consecutive logical tiles (which share operand panels) land on ONE XCD.
// Bijective for any grid size (cdna_hip_programming.md T1).
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

// ELU / sigmoid use the hardware exponential (v_exp_f32, ~2 ulp): exp(z) - 1 is within 6e-8 ABSOLUTE of expm1(z) --
// far inside the 1e-4 parity bar -- and about 10x cheaper than the library expm1f, which made the GroupNorm apply
// passes ALU-bound (ELU follows every head / FPN GroupNorm).
__device__ __forceinline__ float act_fwd(float z, int act) {
  switch (act) {
    case RN_ACT_RELU: return z > 0.f ? z : 0.f;
    case RN_ACT_ELU: return z > 0.f ? z : __expf(z) - 1.f;
    case RN_ACT_RELU6: return fminf(fmaxf(z, 0.f), 6.f);
    case RN_ACT_SIGMOID: return 1.f / (1.f + __expf(-z));
    default: return z;
  }
}
// derivative w.r.t. the pre-activation z
__device__ __forceinline__ float act_grad(float z, int act) {
  switch (act) {
    case RN_ACT_RELU: return z > 0.f ? 1.f : 0.f;
    case RN_ACT_ELU: return z > 0.f ? 1.f : __expf(z);
    case RN_ACT_RELU6: return (z > 0.f && z < 6.f) ? 1.f : 0.f;
    case RN_ACT_SIGMOID: { const float p = 1.f / (1.f + __expf(-z)); return p * (1.f - p); }
    default: return 1.f;
  }
}

// counter-based uniform in [0,1): 32-bit murmur3-style avalanche of (seed, 64-bit element index)
__device__ __forceinline__ float uniform01(uint64_t seed, uint64_t idx) {
  uint32_t h = (uint32_t)idx * 0x9E3779B1u + (uint32_t)seed;
  h ^= (uint32_t)(idx >> 32) * 0x85EBCA77u + (uint32_t)(seed >> 32);
  h ^= h >> 16; h *= 0x85EBCA6Bu;
  h ^= h >> 13; h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return (float)(h >> 8) * (1.0f / 16777216.0f);
}

// 4 consecutive elements at element index i of an fp32 or fp16 tensor (fp16 storage is the inference path)
typedef _Float16 rn_half4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4(const void* p, size_t i, int is_half) {
  if (is_half) {
    const rn_half4 h = *reinterpret_cast<const rn_half4*>(reinterpret_cast<const _Float16*>(p) + i);
    return make_float4((float)h.x, (float)h.y, (float)h.z, (float)h.w);
  }
  return *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p) + i);
}
__device__ __forceinline__ void st4(void* p, size_t i, int is_half, float4 v) {
  if (is_half) {
    rn_half4 h;
    h.x = (_Float16)v.x; h.y = (_Float16)v.y; h.z = (_Float16)v.z; h.w = (_Float16)v.w;
    *reinterpret_cast<rn_half4*>(reinterpret_cast<_Float16*>(p) + i) = h;
  } else {
    *reinterpret_cast<float4*>(reinterpret_cast<float*>(p) + i) = v;
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---------------------------------------------------------------------------------------------------------------------
// GroupNorm partial sums as a by-product of the PRODUCING kernel (rn_gn_rows, rn_hip.h): every block of a conv / depthwise
// forward stores one row of (sum, sum of squares) pairs for the outputs it has just computed; the GroupNorm kernel that
// follows merges the rows of its sample (fixed order, fp64).  Plain stores, read by the next kernel on the stream.
struct StatDev {
  float2* rows;     // conv: [m-tile][channel]; depthwise: [pixel chunk][group]
  int groups, cpg;  // depthwise: the block folds its channels into groups
};

}  // namespace rn
