// Bare MFMA loops on random operands (tuning aid, GPU box): does the clock the chip holds under load depend on the MFMA shape?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_shape tools/micro/mfma_shape.hip && /tmp/mfma_shape
// One wave per SIMD x WPS, operands in registers (re-randomised from memory once), 8 independent accumulator groups per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

template <int SHAPE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ in, float* __restrict__ out, int iters) {
  const int tid = threadIdx.x + blockIdx.x * blockDim.x;
  float a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = in[(tid * 16 + i) & 0xfffff]; b[i] = in[(tid * 16 + 8 + i) & 0xfffff]; }
  float sum = 0.f;
  if (SHAPE == 0) {  // v_mfma_f32_32x32x2_f32: 8 accumulators of 16
    f32x16 acc[8];
    for (int j = 0; j < 8; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[(j * 3 + 1) & 7], acc[j], 0, 0, 0);
    for (int j = 0; j < 8; ++j) for (int r = 0; r < 16; ++r) sum += acc[j][r];
  } else if (SHAPE == 1) {  // v_mfma_f32_16x16x4_f32: 32 accumulators of 4 (same registers, same flops per byte of accumulator)
    f32x4 acc[32];
    for (int j = 0; j < 32; ++j) for (int r = 0; r < 4; ++r) acc[j][r] = 0.f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int j = 0; j < 32; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j & 7], b[(j * 3 + 1) & 7], acc[j], 0, 0, 0);
    for (int j = 0; j < 32; ++j) for (int r = 0; r < 4; ++r) sum += acc[j][r];
  } else if (SHAPE == 2) {  // v_mfma_f32_32x32x16_f16
    half8 ha[4], hb[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) { ha[i][e] = (_Float16)(a[(i + e) & 7]); hb[i][e] = (_Float16)(b[(i * 3 + e) & 7]); }
    f32x16 acc[8];
    for (int j = 0; j < 8; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha[j & 3], hb[(j * 3 + 1) & 3], acc[j], 0, 0, 0);
    for (int j = 0; j < 8; ++j) for (int r = 0; r < 16; ++r) sum += acc[j][r];
  } else {  // v_mfma_f32_16x16x32_f16
    half8 ha[4], hb[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) { ha[i][e] = (_Float16)(a[(i + e) & 7]); hb[i][e] = (_Float16)(b[(i * 3 + e) & 7]); }
    f32x4 acc[32];
    for (int j = 0; j < 32; ++j) for (int r = 0; r < 4; ++r) acc[j][r] = 0.f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int j = 0; j < 32; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha[j & 3], hb[(j * 3 + 1) & 3], acc[j], 0, 0, 0);
    for (int j = 0; j < 32; ++j) for (int r = 0; r < 4; ++r) sum += acc[j][r];
  }
  out[tid] = sum;
}

template <int SHAPE>
void run(const char* name, double flop_per_iter_per_wave, const float* in, float* out, int wps) {
  const int blocks = 256 * wps, iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k<SHAPE>, dim3(blocks), dim3(256), 0, 0, in, out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(k<SHAPE>, dim3(blocks), dim3(256), 0, 0, in, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = flop_per_iter_per_wave * iters * 4.0 * blocks * 5;
  printf("%-28s waves/SIMD %d: %8.2f ms  %8.1f TFLOP/s\n", name, wps, ms, flops / ms / 1e9);
}

int main(int argc, char** argv) {
  const bool zeros = argc > 1 && atoi(argv[1]) == 0;
  std::vector<float> h(1 << 20);
  srand(1);
  for (auto& v : h) v = zeros ? 0.f : (float)rand() / RAND_MAX * 2.f - 1.f;
  float *in, *out;
  hipMalloc(&in, h.size() * 4); hipMalloc(&out, 256 * 8 * 256 * 4);
  hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  printf("operands: %s\n", zeros ? "zeros" : "uniform random [-1, 1)");
  for (int wps = 1; wps <= 2; ++wps) {
    run<0>("f32 32x32x2  (8 acc x 16)", 8.0 * 32 * 32 * 2 * 2, in, out, wps);
    run<1>("f32 16x16x4  (32 acc x 4)", 32.0 * 16 * 16 * 4 * 2, in, out, wps);
    run<2>("f16 32x32x16 (8 acc x 16)", 8.0 * 32 * 32 * 16 * 2, in, out, wps);
    run<3>("f16 16x16x32 (32 acc x 4)", 32.0 * 16 * 16 * 32 * 2, in, out, wps);
  }
  return 0;
}
