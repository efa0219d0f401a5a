// What does `buffer_load_dwordx4 ... lds` write for lanes whose offset is out of the descriptor's range?  (GPU box)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/glds_oob tools/micro/glds_oob.hip && /tmp/glds_oob
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float* x, float* y, unsigned bytes) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  for (int i = threadIdx.x; i < 256 * 4; i += 256) smem[i] = -7.f;   // sentinel
  __syncthreads();
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, bytes, 0x00020000);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned voff = threadIdx.x * 16u;
  if (lane & 1) voff = 0x80000000u;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + wave * 256), 16, voff, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (int i = 0; i < 4; ++i) y[threadIdx.x * 4 + i] = smem[threadIdx.x * 4 + i];
}
int main() {
  std::vector<float> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = (float)(i + 1);
  float *x, *y;
  (void)hipMalloc(&x, 4096); (void)hipMalloc(&y, 4096);
  (void)hipMemcpy(x, h.data(), 4096, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 4096, 0, x, y, 4096u);
  (void)hipMemcpy(h.data(), y, 4096, hipMemcpyDeviceToHost);
  int ok_in = 0, zero_oob = 0, sentinel_oob = 0, other = 0;
  for (int t = 0; t < 256; ++t)
    for (int i = 0; i < 4; ++i) {
      const float v = h[t * 4 + i];
      if (t & 1) { if (v == 0.f) ++zero_oob; else if (v == -7.f) ++sentinel_oob; else ++other; }
      else { if (v == (float)(t * 4 + i + 1)) ++ok_in; else ++other; }
    }
  printf("in-range lanes correct: %d / 512; out-of-range lanes: %d zeros, %d untouched (sentinel), other %d\n", ok_in, zero_oob, sentinel_oob, other);
  return 0;
}
