// Dispatch-rate microbenchmark: N host threads, one stream each, every thread launches M dependent (same-stream)
// tiny kernels. Reports kernels/s summed over the streams -- the ceiling for pipelines made of many small kernels.
//   hipcc -O2 --offload-arch=gfx950 -o tools/launch_rate tools/launch_rate.hip -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

__global__ void tiny(unsigned* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1u; }
__global__ void small_grid(unsigned* p, unsigned n) { unsigned i = blockIdx.x * 256u + threadIdx.x; if (i < n) p[i] += 1u; }

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 4000;
  hipSetDevice(0);
  for (int use_graph = 0; use_graph < 2; ++use_graph)
    for (int N : {1, 2, 4, 6, 8}) {
      std::vector<hipStream_t> st(N);
      std::vector<unsigned*> buf(N);
      std::vector<hipGraphExec_t> ge(N);
      for (int i = 0; i < N; ++i) {
        hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking);
        hipMalloc(&buf[i], 1 << 20);
        hipMemset(buf[i], 0, 1 << 20);
        if (use_graph) {
          hipGraph_t g;
          hipStreamBeginCapture(st[i], hipStreamCaptureModeThreadLocal);
          for (int k = 0; k < 40; ++k) hipLaunchKernelGGL(small_grid, dim3(256), dim3(256), 0, st[i], buf[i], 65536u);
          hipStreamEndCapture(st[i], &g);
          hipGraphInstantiate(&ge[i], g, nullptr, nullptr, 0);
        }
      }
      hipDeviceSynchronize();
      auto t0 = std::chrono::steady_clock::now();
      std::vector<std::thread> th;
      for (int i = 0; i < N; ++i)
        th.emplace_back([&, i] {
          if (use_graph) { for (int k = 0; k < M / 40; ++k) hipGraphLaunch(ge[i], st[i]); }
          else { for (int k = 0; k < M; ++k) hipLaunchKernelGGL(small_grid, dim3(256), dim3(256), 0, st[i], buf[i], 65536u); }
          hipStreamSynchronize(st[i]);
        });
      for (auto& t : th) t.join();
      const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      const long total = (long)N * (use_graph ? (M / 40) * 40 : M);
      printf("%s streams=%d: %.0f kernels/s total, %.2f us per kernel per stream\n", use_graph ? "graph(40 nodes)" : "plain launches",
             N, total / dt, dt / (total / N) * 1e6);
      for (int i = 0; i < N; ++i) { hipStreamDestroy(st[i]); hipFree(buf[i]); }
    }
  return 0;
}
