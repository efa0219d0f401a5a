// What does one dependent kernel boundary cost on this box, and what makes it 4.5 - 5 us inside the captured training step
//  when the platform guide measures 1.45 - 1.9 us (MI355X_MICROARCH.md, row "boundary")?
//
//   hipcc --offload-arch=gfx950 -O3 tools/micro/boundary.hip -o /tmp/boundary && /tmp/boundary
//
// Every kernel stamps the 100 MHz wall clock (s_memrealtime) when its first wave starts and when its last instruction
// runs; the GAP between kernel i's end stamp and kernel i + 1's start stamp is the boundary itself (end-of-kernel cache
// write-back, the dependency resolution of the queue / graph, dispatch, first wave launch), free of host effects.
// Variants: eager one stream | captured one-stream graph | graph with a parallel branch on a second / third stream |
// cross-stream edges every 8 kernels; kernarg 16 B | 512 B; grid 1 x 64 | 256 x 256 | streaming 4 MB through HBM;
// body ~0 | ~5 us.  Prints median / p10 / p90 of the gaps per variant and the whole-chain time per kernel.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

struct Small { uint64_t* stamps; float* buf; int slot; int spin; };
struct Big { uint64_t* stamps; float* buf; int slot; int spin; char pad[488]; };   // 512 bytes like the rn_* segment structs

template <class A>
__global__ void __launch_bounds__(256) node_kernel(const A a, int stream_bytes) {
  uint64_t t0 = wall_clock64();
  // streaming body: read what the predecessor wrote, write for the successor (float4 per thread, grid-stride)
  if (stream_bytes) {
    float4* p = reinterpret_cast<float4*>(a.buf);
    const int n = stream_bytes / 16;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
      float4 v = p[i];
      v.x += 1.f;
      p[i] = v;
    }
  }
  if (a.spin) {                                       // a body of ~spin x 10 ns
    while (wall_clock64() - t0 < (uint64_t)a.spin) __builtin_amdgcn_s_sleep(1);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) a.stamps[2 * a.slot] = t0;
  __syncthreads();
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) a.stamps[2 * a.slot + 1] = wall_clock64();
}

struct Cfg { const char* name; int grid, block, stream_bytes, spin, big; };

template <class A>
static void launch(const Cfg& c, uint64_t* stamps, float* buf, int slot, hipStream_t st) {
  A a{};
  a.stamps = stamps; a.buf = buf; a.slot = slot; a.spin = c.spin;
  hipLaunchKernelGGL(node_kernel<A>, dim3(c.grid), dim3(c.block), 0, st, a, c.stream_bytes);
}
static void launch_any(const Cfg& c, uint64_t* stamps, float* buf, int slot, hipStream_t st) {
  if (c.big) launch<Big>(c, stamps, buf, slot, st); else launch<Small>(c, stamps, buf, slot, st);
}

static void report(const char* what, const Cfg& c, const std::vector<uint64_t>& h, int first, int n, double wall_us) {
  std::vector<double> gaps, bodies;
  for (int i = first; i + 1 < first + n; ++i) gaps.push_back((double)((int64_t)h[2 * (i + 1)] - (int64_t)h[2 * i + 1]) * 0.01);
  for (int i = first; i < first + n; ++i) bodies.push_back((double)((int64_t)h[2 * i + 1] - (int64_t)h[2 * i]) * 0.01);
  std::sort(gaps.begin(), gaps.end());
  std::sort(bodies.begin(), bodies.end());
  const double span = (double)((int64_t)h[2 * (first + n - 1) + 1] - (int64_t)h[2 * first]) * 0.01;
  printf("%-34s %-22s gap med %6.2f  p10 %6.2f  p90 %6.2f us | body med %6.2f | chain %7.2f us = %5.2f / kernel | host %8.1f us\n", what, c.name,
         gaps[gaps.size() / 2], gaps[gaps.size() / 10], gaps[gaps.size() * 9 / 10], bodies[bodies.size() / 2], span, span / n, wall_us);
}

int main() {
  const int N = 256;
  uint64_t* stamps;
  float *buf, *buf2, *buf3;
  CK(hipMalloc(&stamps, 16 * 4 * N));
  CK(hipMalloc(&buf, 64 << 20));
  CK(hipMalloc(&buf2, 64 << 20));
  CK(hipMalloc(&buf3, 64 << 20));
  CK(hipMemset(buf, 0, 64 << 20));
  CK(hipMemset(buf2, 0, 64 << 20));
  CK(hipMemset(buf3, 0, 64 << 20));
  hipStream_t s0, s1, s2;
  CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t e0, e1, ef, ej1, ej2;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ej1, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ej2, hipEventDisableTiming));
  std::vector<uint64_t> h(2 * 4 * N);

  const Cfg cfgs[] = {
      {"1x64 empty, 16 B arg", 1, 64, 0, 0, 0},
      {"1x64 empty, 512 B arg", 1, 64, 0, 0, 1},
      {"256x256 empty, 16 B", 256, 256, 0, 0, 0},
      {"256x256 empty, 512 B", 256, 256, 0, 0, 1},
      {"32x256 5us body, 512 B", 32, 256, 0, 500, 1},
      {"256x256 5us body, 512 B", 256, 256, 0, 500, 1},
      {"1024x256 stream 4 MB", 1024, 256, 4 << 20, 0, 1},
      {"1024x256 stream 16 MB", 1024, 256, 16 << 20, 0, 1},
  };
  for (const Cfg& c : cfgs) {
    float ms;
    // ---- eager, one stream
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0, s0));
      for (int i = 0; i < N; ++i) launch_any(c, stamps, buf, i, s0);
      CK(hipEventRecord(e1, s0));
      CK(hipStreamSynchronize(s0));
    }
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(h.data(), stamps, 16 * N, hipMemcpyDeviceToHost));
    report("eager, one stream", c, h, 0, N, ms * 1e3);

    // ---- one-stream graph
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < N; ++i) launch_any(c, stamps, buf, i, s0);
    CK(hipStreamEndCapture(s0, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0, s0));
      CK(hipGraphLaunch(ge, s0));
      CK(hipEventRecord(e1, s0));
      CK(hipStreamSynchronize(s0));
    }
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(h.data(), stamps, 16 * N, hipMemcpyDeviceToHost));
    report("graph, one stream", c, h, 0, N, ms * 1e3);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));

    // ---- graph: the chain on s0 + an INDEPENDENT chain of the same kernels on s1 (fork at the start, join at the end)
    for (int branches = 2; branches <= 3; ++branches) {
      CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
      CK(hipEventRecord(ef, s0));
      CK(hipStreamWaitEvent(s1, ef, 0));
      if (branches == 3) CK(hipStreamWaitEvent(s2, ef, 0));
      for (int i = 0; i < N; ++i) {
        launch_any(c, stamps, buf, i, s0);
        launch_any(c, stamps, buf2, N + i, s1);
        if (branches == 3) launch_any(c, stamps, buf3, 2 * N + i, s2);
      }
      CK(hipEventRecord(ej1, s1)); CK(hipStreamWaitEvent(s0, ej1, 0));
      if (branches == 3) { CK(hipEventRecord(ej2, s2)); CK(hipStreamWaitEvent(s0, ej2, 0)); }
      CK(hipStreamEndCapture(s0, &g));
      CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, s0));
        CK(hipGraphLaunch(ge, s0));
        CK(hipEventRecord(e1, s0));
        CK(hipStreamSynchronize(s0));
      }
      CK(hipEventElapsedTime(&ms, e0, e1));
      CK(hipMemcpy(h.data(), stamps, 16 * 3 * N, hipMemcpyDeviceToHost));
      report(branches == 2 ? "graph, 2 parallel chains: chain 0" : "graph, 3 parallel chains: chain 0", c, h, 0, N, ms * 1e3);
      report(branches == 2 ? "graph, 2 parallel chains: chain 1" : "graph, 3 parallel chains: chain 1", c, h, N, N, ms * 1e3);
      CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }

    // ---- graph: ONE logical chain that hops between two streams every 8 kernels (a cross-stream edge per hop)
    CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
    {
      hipStream_t cur = s0, other = s1;
      std::vector<hipEvent_t> evs;
      for (int i = 0; i < N; ++i) {
        launch_any(c, stamps, buf, i, cur);
        if (i % 8 == 7 && i + 1 < N) {
          hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming)); evs.push_back(ev);
          CK(hipEventRecord(ev, cur)); CK(hipStreamWaitEvent(other, ev, 0));
          std::swap(cur, other);
        }
      }
      if (cur != s0) { CK(hipEventRecord(ej1, cur)); CK(hipStreamWaitEvent(s0, ej1, 0)); }
      CK(hipStreamEndCapture(s0, &g));
      for (hipEvent_t ev : evs) CK(hipEventDestroy(ev));
    }
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0, s0));
      CK(hipGraphLaunch(ge, s0));
      CK(hipEventRecord(e1, s0));
      CK(hipStreamSynchronize(s0));
    }
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(h.data(), stamps, 16 * N, hipMemcpyDeviceToHost));
    report("graph, chain hopping 2 streams / 8", c, h, 0, N, ms * 1e3);
    {   // the gaps AT the hops only
      std::vector<double> hop, stay;
      for (int i = 0; i + 1 < N; ++i) {
        const double gp = (double)((int64_t)h[2 * (i + 1)] - (int64_t)h[2 * i + 1]) * 0.01;
        (i % 8 == 7 ? hop : stay).push_back(gp);
      }
      std::sort(hop.begin(), hop.end()); std::sort(stay.begin(), stay.end());
      printf("%-34s %-22s gap at a hop med %6.2f us, inside a run med %6.2f us\n", "", c.name, hop[hop.size() / 2], stay[stay.size() / 2]);
    }
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    printf("\n");
  }
  return 0;
}
