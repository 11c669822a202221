// How do N independent chains of short kernels behave on N streams (fork from / join into a main stream with events, then a small D2H
// read-back + stream synchronise, as the WSOS cone's sections do)?  Direct launches against one captured hipGraph of the same section.
//   hipcc --offload-arch=gfx950 -O2 tools/probe_lanes.hip -o tools/_bin/probe_lanes
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
// ~spin cycles of dependent FMAs per thread: a kernel of `wgs` workgroups that lasts about spin * 32 cycles
__global__ void chain_kernel(double* d, int spin) {
  double x = d[blockIdx.x * 256 + threadIdx.x];
  for (int i = 0; i < spin; ++i) x = x * 1.0000001 + 1e-9;
  d[blockIdx.x * 256 + threadIdx.x] = x;
}
int main(int argc, char** argv) {
  const int M = 16;          // kernels per chain
  const int wgs = 300;       // workgroups per kernel
  const int spin = argc > 1 ? atoi(argv[1]) : 3000;   // 3000 x 32 cycles = 40 us
  const int NMAX = 6;
  std::vector<double*> bufs(NMAX);
  for (auto& b : bufs) { CK(hipMalloc(&b, wgs * 256 * 8)); CK(hipMemset(b, 0, wgs * 256 * 8)); }
  int plo, phi; CK(hipDeviceGetStreamPriorityRange(&plo, &phi));
  std::vector<hipStream_t> st(NMAX);
  for (int i = 0; i < NMAX; ++i) CK(hipStreamCreateWithPriority(&st[i], hipStreamNonBlocking, i == 1 ? plo : phi));
  std::vector<hipEvent_t> done(NMAX);
  for (auto& ev : done) CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  hipEvent_t fork; CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
  double* hp; CK(hipHostMalloc(&hp, 64, hipHostMallocDefault));
  auto section = [&](int nl, int chains) -> int {
    CK(hipEventRecord(fork, st[0]));
    for (int i = 1; i < nl; ++i) CK(hipStreamWaitEvent(st[i], fork, 0));
    for (int k = 0; k < chains; ++k) {
      hipStream_t s = st[k % nl];
      for (int j = 0; j < M; ++j) hipLaunchKernelGGL(chain_kernel, dim3(wgs), dim3(256), 0, s, bufs[k], spin);
    }
    for (int i = 1; i < nl; ++i) { CK(hipEventRecord(done[i], st[i])); CK(hipStreamWaitEvent(st[0], done[i], 0)); }
    return 0;
  };
  const int chains = 5, R = 50;
  for (int nl = 1; nl <= NMAX; ++nl) {
    for (int w = 0; w < 3; ++w) { if (section(nl, chains)) return 1; CK(hipStreamSynchronize(st[0])); }
    auto t0 = std::chrono::steady_clock::now();
    double enq = 0;
    for (int r = 0; r < R; ++r) {
      auto a = std::chrono::steady_clock::now();
      if (section(nl, chains)) return 1;
      auto b = std::chrono::steady_clock::now();
      enq += std::chrono::duration<double, std::micro>(b - a).count();
      CK(hipMemcpyAsync(hp, bufs[0], 8, hipMemcpyDeviceToHost, st[0]));
      CK(hipStreamSynchronize(st[0]));
    }
    auto t1 = std::chrono::steady_clock::now();
    printf("direct   %d streams, %d chains x %d kernels: %.1f us per section (host enqueue %.1f us)\n", nl, chains, M,
           std::chrono::duration<double, std::micro>(t1 - t0).count() / R, enq / R);
  }
  // the same section captured once into a graph and replayed
  for (int nl = 1; nl <= NMAX; ++nl) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st[0], hipStreamCaptureModeGlobal));
    if (section(nl, chains)) return 1;
    CK(hipStreamEndCapture(st[0], &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int w = 0; w < 3; ++w) { CK(hipGraphLaunch(ge, st[0])); CK(hipStreamSynchronize(st[0])); }
    auto t0 = std::chrono::steady_clock::now();
    double enq = 0;
    for (int r = 0; r < R; ++r) {
      auto a = std::chrono::steady_clock::now();
      CK(hipGraphLaunch(ge, st[0]));
      auto b = std::chrono::steady_clock::now();
      enq += std::chrono::duration<double, std::micro>(b - a).count();
      CK(hipMemcpyAsync(hp, bufs[0], 8, hipMemcpyDeviceToHost, st[0]));
      CK(hipStreamSynchronize(st[0]));
    }
    auto t1 = std::chrono::steady_clock::now();
    printf("graph    %d branches, %d chains x %d kernels: %.1f us per section (host launch %.1f us)\n", nl, chains, M,
           std::chrono::duration<double, std::micro>(t1 - t0).count() / R, enq / R);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  return 0;
}
