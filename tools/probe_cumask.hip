// Which compute units does a CU-masked stream use on MI355X (8 XCDs x 32 CUs)?  hipExtStreamCreateWithCUMask takes a bit array;
// this probe clears a run of bits and reports, per XCD, which (SE, CU) slots workgroups of a large grid landed on -- so that the
// blocked Cholesky's helper queue can leave CUs free in EVERY XCD for the latency-bound kernels of the main queue (dense.hip).
// Build: hipcc --offload-arch=gfx950 -O3 tools/probe_cumask.hip -o tools/_bin/probe_cumask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1;} } while (0)

__global__ void where(unsigned* out, int spin) {
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  double x = threadIdx.x;
  for (int i = 0; i < spin; ++i) x = fma(x, 1.0000001, 1e-9);   // long enough that the grid spreads over every allowed CU
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc + (x == 123.0 ? 1 : 0); }
}

static int report(const char* what, hipStream_t st, unsigned* d, int grid) {
  hipLaunchKernelGGL(where, dim3(grid), dim3(64), 0, st, d, 20000);
  CK(hipStreamSynchronize(st));
  std::vector<unsigned> h(2 * grid);
  CK(hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost));
  std::set<unsigned> per[16];
  for (int b = 0; b < grid; ++b) {
    const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
    const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
    per[xcc].insert((se << 8) | (sh << 4) | cu);
  }
  int total = 0;
  printf("%s:", what);
  for (int x = 0; x < 8; ++x) { printf(" xcd%d=%zu", x, per[x].size()); total += (int)per[x].size(); }
  printf("  total %d CUs used\n", total);
  return 0;
}

int main() {
  unsigned* d;
  const int grid = 8192;
  CK(hipMalloc(&d, 2 * grid * 4));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("multiProcessorCount %d\n", prop.multiProcessorCount);
  hipStream_t s0; CK(hipStreamCreate(&s0));
  report("no mask", s0, d, grid);
  for (int clear : {8, 16, 32}) {
    std::vector<uint32_t> m(8, 0xffffffffu);
    for (int b = 0; b < clear; ++b) m[b / 32] &= ~(1u << (b % 32));
    hipStream_t s; CK(hipExtStreamCreateWithCUMask(&s, 8, m.data()));
    char buf[64]; snprintf(buf, 64, "bits 0..%d cleared", clear - 1);
    report(buf, s, d, grid);
    CK(hipStreamDestroy(s));
  }
  {   // one bit per 32: the XCC-major reading
    std::vector<uint32_t> m(8, 0xfffffffeu);
    hipStream_t s; CK(hipExtStreamCreateWithCUMask(&s, 8, m.data()));
    report("bit 0 of every word cleared", s, d, grid);
    CK(hipStreamDestroy(s));
  }
  return 0;
}
