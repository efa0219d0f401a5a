// Go / no-go for an XCD-resident section of the MobileNetV2 chain: what does a PHASE BARRIER among the
// blocks of ONE XCD cost, and is data handed over through that XCD's L2 with plain stores visible to the other blocks' loads?
//
//   hipcc --offload-arch=gfx950 -O3 tools/micro/xcd_cluster.hip -o /tmp/xcd_cluster && /tmp/xcd_cluster
//
// Grid = 8 x B blocks of 256 threads; cluster = blockIdx.x % 8 (round-robin placement: one XCD), rank = blockIdx.x / 8.
// Every block reads HW_REG_XCC_ID and reports whether its cluster really is one XCD.  Clusters >= `nclusters` exit at once.
// P phases: every block writes `bytes` of its own slice (plain 16-byte stores), s_waitcnt vmcnt(0), arrives on the cluster's
// counter (relaxed device-scope atomic: executed in the XCD's L2), polls it with sc1 loads, then reads the slice its
// neighbour (rank + 1) wrote in this phase and checks every value (the buffers are re-used every second phase, so a stale
// L1 line would show as a mismatch) -- with plain loads and with sc1 (L1-bypassing) loads.
// Prints us per phase (device clock), placement, mismatches.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

struct Args {
  unsigned* counters;      // [8][64] words: one counter per cluster (own 256-byte line)
  float* buf;              // [2][8][B][bytes / 4]
  unsigned long long* out; // per block: start, end, xcc, mismatches
  int B, P, bytes, nclusters, sc1_loads, chipwide, local_atomics;
};

__device__ __forceinline__ unsigned xcc_id() {
  return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15u;   // hwreg(HW_REG_XCC_ID, 0, 4)
}

__device__ __forceinline__ float4 load_sc1(const float* p) {
  float4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

__global__ void __launch_bounds__(256) cluster_kernel(const Args a) {
  const int cluster = blockIdx.x & 7, rank = blockIdx.x >> 3, tid = threadIdx.x;
  if (cluster >= a.nclusters) return;
  unsigned* ctr = a.counters + (a.chipwide ? 0 : cluster * 64);
  const unsigned members = a.chipwide ? a.B * a.nclusters : a.B;
  const int words = a.bytes / 4;
  unsigned long long mism = 0;
  __shared__ int timed_out;
  if (tid == 0) timed_out = 0;
  __syncthreads();
  const unsigned long long t0 = wall_clock64();
  for (int p = 0; p < a.P; ++p) {
    float* mine = a.buf + ((size_t)((p & 1) * 8 + cluster) * a.B + rank) * words;
    const float tag = (float)(p * 1024 + rank);
    for (int i = tid * 4; i < words; i += 1024) *reinterpret_cast<float4*>(mine + i) = make_float4(tag, tag + 0.25f, tag + 0.5f, (float)i);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the stores have reached L2
    __syncthreads();
    if (tid == 0) {
      const unsigned target = members * (unsigned)(p + 1);
      int tries = 0;
      if (a.local_atomics) {      // no scope bits on the atomic (performed in THIS XCD's L2), sc0 on the poll (L1 bypass only)
        const unsigned one = 1u;
        asm volatile("global_atomic_add %0, %1, off" ::"v"(ctr), "v"(one) : "memory");
        unsigned seen = 0;
        do {
          asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(seen) : "v"(ctr) : "memory");
          if (seen >= target) break;
          __builtin_amdgcn_s_sleep(1);
        } while (++tries < (1 << 16));
      } else {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++tries < (1 << 16)) __builtin_amdgcn_s_sleep(1);
      }
      if (tries >= (1 << 16)) timed_out = 1;
    }
    __syncthreads();
    if (timed_out) { mism |= 1ull << 40; break; }
    if (words) {
      const int nb = (rank + 1) % a.B;
      const float* theirs = a.buf + ((size_t)((p & 1) * 8 + cluster) * a.B + nb) * words;
      const float want = (float)(p * 1024 + nb);
      for (int i = tid * 4; i < words; i += 1024) {
        const float4 v = a.sc1_loads ? load_sc1(theirs + i) : *reinterpret_cast<const float4*>(theirs + i);
        if (v.x != want || v.y != want + 0.25f || v.z != want + 0.5f || v.w != (float)i) ++mism;
      }
    }
  }
  const unsigned long long t1 = wall_clock64();
  // block-wide mismatch count
  __shared__ unsigned long long red[256];
  red[tid] = mism;
  __syncthreads();
  if (tid == 0) {
    unsigned long long s = 0;
    for (int i = 0; i < 256; ++i) s += red[i];
    unsigned long long* o = a.out + (size_t)blockIdx.x * 4;
    o[0] = t0; o[1] = t1; o[2] = xcc_id(); o[3] = s;
  }
}

int main() {
  Args a{};
  const int BMAX = 160, P = 64;
  CK(hipMalloc(&a.counters, 8 * 64 * 4));
  CK(hipMalloc(&a.buf, (size_t)2 * 8 * BMAX * 65536));
  CK(hipMalloc(&a.out, (size_t)8 * BMAX * 32));
  std::vector<unsigned long long> h(8 * BMAX * 4);
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  int occ = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, cluster_kernel, 256, 0));
  printf("occupancy query: %d blocks of 256 threads per CU\n", occ);
  printf("%-10s %-4s %-9s %-8s %-6s | us/phase (max over blocks) | one XCD per cluster | mismatches\n", "scope", "B", "clusters", "bytes", "loads");
  for (int chipwide = 0; chipwide <= 1; ++chipwide)
   for (int local = 0; local <= 1 - chipwide; ++local)
    for (int ncl : {1, 2, 8})
      for (int B : {32, 64, 128})
        for (int bytes : {0, 4096, 65536})
          for (int sc1 = 0; sc1 <= 1; ++sc1) {
            if (bytes == 0 && sc1) continue;
            if (chipwide && (ncl != 8 || bytes == 65536)) continue;
            a.B = B; a.P = P; a.bytes = bytes; a.nclusters = ncl; a.sc1_loads = sc1; a.chipwide = chipwide; a.local_atomics = local;
            double best = 1e30;
            unsigned long long mism = 0;
            bool one_xcd = true;
            for (int rep = 0; rep < 3; ++rep) {
              CK(hipMemsetAsync(a.counters, 0, 8 * 64 * 4, st));
              CK(hipMemsetAsync(a.out, 0, (size_t)8 * BMAX * 32, st));
              hipLaunchKernelGGL(cluster_kernel, dim3(8 * B), dim3(256), 0, st, a);
              CK(hipStreamSynchronize(st));
              CK(hipMemcpy(h.data(), a.out, (size_t)8 * B * 32, hipMemcpyDeviceToHost));
              double worst = 0;
              for (int c = 0; c < ncl; ++c) {
                const unsigned long long x0 = h[(size_t)c * 4 + 2];
                for (int r = 0; r < B; ++r) {
                  const unsigned long long* o = &h[(size_t)(r * 8 + c) * 4];
                  worst = std::max(worst, (double)(o[1] - o[0]) * 0.01 / P);
                  mism += o[3];
                  if (o[2] != x0) one_xcd = false;
                }
              }
              best = std::min(best, worst);
            }
            printf("%-10s %-4d %-9d %-8d %-6s | %8.2f | %s | %llu\n", chipwide ? "chip-wide" : (local ? "XCD-L2" : "XCD-agent"), B, ncl, bytes, sc1 ? "sc1" : "plain", best,
                   one_xcd ? "yes" : "NO", mism);
          }
  // placement map of the last launch: which XCC did cluster c's blocks report
  printf("xcc of cluster 0..7 (rank 0): ");
  for (int c = 0; c < 8; ++c) printf("%llu ", h[(size_t)c * 4 + 2]);
  printf("\n");
  return 0;
}
