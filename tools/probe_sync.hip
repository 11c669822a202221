// How long does the host wait for a small result?  (a) hipMemcpyAsync D2H into pinned memory + hipStreamSynchronize,
// (b) a kernel that stores the result and a sequence number into host-mapped coherent memory, host spins on the number.
// Both after a short dependent kernel, 2000 repetitions.   hipcc --offload-arch=gfx950 -O2 tools/probe_sync.hip -o tools/_bin/probe_sync
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void work(double* d, int it) { d[threadIdx.x] = d[threadIdx.x] * 1.0000001 + it; }
__global__ void publish(const double* d, volatile double* h, volatile int* seq, int s) {
  h[threadIdx.x] = d[threadIdx.x];
  __threadfence_system();
  if (threadIdx.x == 0) *seq = s;
}
int main() {
  double* d; CK(hipMalloc(&d, 64 * 8)); CK(hipMemset(d, 0, 64 * 8));
  double* hp; CK(hipHostMalloc(&hp, 64 * 8, hipHostMallocDefault));
  double* hm; int* hs;
  CK(hipHostMalloc(&hm, 64 * 8, hipHostMallocMapped | hipHostMallocCoherent));
  CK(hipHostMalloc(&hs, 64, hipHostMallocMapped | hipHostMallocCoherent));
  *hs = 0;
  double* dm; int* ds; CK(hipHostGetDevicePointer((void**)&dm, hm, 0)); CK(hipHostGetDevicePointer((void**)&ds, hs, 0));
  hipStream_t st; CK(hipStreamCreate(&st));
  const int R = 2000;
  for (int mode = 0; mode < 3; ++mode) {
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 1; i <= R; ++i) {
      hipLaunchKernelGGL(work, dim3(1), dim3(64), 0, st, d, i);
      if (mode == 0) {
        CK(hipMemcpyAsync(hp, d, 8, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
      } else if (mode == 1) {
        hipLaunchKernelGGL(publish, dim3(1), dim3(64), 0, st, d, dm, ds, i);
        while (*(volatile int*)hs != i) { }
      } else {
        CK(hipStreamSynchronize(st));
      }
    }
    auto t1 = std::chrono::steady_clock::now();
    const double us = std::chrono::duration<double, std::micro>(t1 - t0).count() / R;
    printf("%s: %.1f us per round trip (launch + result on the host)\n",
           mode == 0 ? "memcpyAsync D2H + streamSynchronize" : mode == 1 ? "publish kernel to mapped host memory + spin" : "streamSynchronize only", us);
  }
  return 0;
}
