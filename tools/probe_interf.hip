// What slows the Cholesky's critical-path kernels down 2 - 2.5x while a trailing GEMM is resident (profiles/r04_cholesky_experiments.txt B)?
// A ONE-workgroup latency chain with the three ingredients of potrf_tiles4_kernel -- dependent v_fma_f64, LDS round trips with a
// barrier, dependent loads that miss L1 -- times each section in SHADER cycles (clock64) and in WALL ticks (wall_clock64, 100 MHz),
// alone and beside an FP64 MFMA load on another stream that occupies (a) every CU, (b) every CU but the chain's own XCD's, (c) half
// of the SIMDs of every CU.  cycles unchanged + wall longer = the clock; cycles longer = contention inside the CU / memory system.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_interf.hip -o /tmp/probe_interf && /tmp/probe_interf
#include <hip/hip_runtime.h>
#include <unistd.h>
#include <cstdio>
#include <vector>
typedef double d4_t __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1;} } while (0)

// HBM streaming load: every workgroup sums its slice of a 2 GB buffer, `passes` times
typedef double dv2_t __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k_stream(const dv2_t* buf, long n2, int passes, double* out) {
  double s = 0;
  for (int p = 0; p < passes; ++p)
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n2; i += (long)gridDim.x * 256) {
      const dv2_t v = __builtin_nontemporal_load(buf + i);
      s += v.x + v.y;
    }
  out[(blockIdx.x * blockDim.x + threadIdx.x) % 4096] = s;
}

__global__ __launch_bounds__(256) void k_load(double* out, int iters, volatile int* stop, int nacc_dummy) {
  d4_t acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = (d4_t){0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = threadIdx.x * 2e-3;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 64; ++r)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    (void)stop;   // (a first form polled a pinned host word here: 2.6e5 threads reading over PCIe stalled every other load of the chip)
  }
  double s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[(blockIdx.x * blockDim.x + threadIdx.x) % 4096] = s;
}

// sections: [0] 4096 dependent v_fma_f64; [1] 512 x (LDS write, barrier, LDS read of a neighbour's word); [2] 512 dependent 8-byte loads
// (pointer chase through a 64 MB ring: L2 / MALL / HBM latency); [3] 256 x 16 x 16 x 4 FP64 MFMAs in a dependent chain
__global__ __launch_bounds__(256) void k_chain(const long* ring, long* res, double* sink, int prio) {
  __shared__ double lds[256];
  const int t = threadIdx.x;
  if (prio) __builtin_amdgcn_s_setprio(3);   // (wavefront priority 3: the instruction arbiter prefers this wavefront to priority-0 ones)
  long long c[5], w[5];
  double x = 1.0 + t * 1e-9;
  c[0] = clock64(); w[0] = wall_clock64();
  for (int i = 0; i < 4096; ++i) x = fma(x, 1.0000001, 1e-12);
  c[1] = clock64(); w[1] = wall_clock64();
  for (int i = 0; i < 512; ++i) {
    lds[t] = x;
    __syncthreads();
    x += lds[(t + 17) & 255] * 1e-20;
    __syncthreads();
  }
  c[2] = clock64(); w[2] = wall_clock64();
  long p = t * 8;
  for (int i = 0; i < 512; ++i) p = ring[p];
  c[3] = clock64(); w[3] = wall_clock64();
  d4_t acc = (d4_t){0, 0, 0, 0};
  for (int i = 0; i < 256; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, 1e-30, acc, 0, 0, 0);
  c[4] = clock64(); w[4] = wall_clock64();
  sink[t] = x + (double)p + acc[0];
  if (t == 0) {
    for (int i = 0; i < 4; ++i) { res[2 * i] = c[i + 1] - c[i]; res[2 * i + 1] = w[i + 1] - w[i]; }
    unsigned xcc = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    res[8] = xcc & 0xf;
  }
}

int main() {
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  const long NR = 8L << 20;   // 64 MB ring
  std::vector<long> h(NR);
  { // a permutation with a long cycle: stride walk by a large odd step
    const long step = 5242881;   // odd, ~ NR * 0.625
    for (long i = 0; i < NR; ++i) h[i] = (i + step) % NR;
  }
  long* ring; long* res; double* sink; double* out; int* stop;
  CK(hipMalloc(&ring, NR * 8)); CK(hipMalloc(&res, 16 * 8)); CK(hipMalloc(&sink, 256 * 8)); CK(hipMalloc(&out, 4096 * 8));
  CK(hipHostMalloc(&stop, 4, hipHostMallocDefault));
  CK(hipMemcpy(ring, h.data(), NR * 8, hipMemcpyHostToDevice));
  long hr[16];
  const char* names[4] = {"4096 dependent v_fma_f64", "512 x (LDS write, barrier, read)", "512 dependent loads (64 MB ring)", "256 dependent FP64 MFMAs"};
  auto run_chain = [&](const char* tag) -> int {
    for (int prio = 0; prio < 2; ++prio) {
      for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k_chain, dim3(1), dim3(256), 0, s1, ring, res, sink, prio);
        CK(hipStreamSynchronize(s1));
      }
      CK(hipMemcpy(hr, res, 16 * 8, hipMemcpyDeviceToHost));
      printf("%-58s (chain on XCC %ld)%s\n", tag, hr[8], prio ? "  WITH s_setprio 3" : "");
      for (int i = 0; i < 4; ++i)
        printf("    %-36s %9ld shader cycles  %8.2f us wall  -> %7.1f MHz\n", names[i], hr[2 * i], hr[2 * i + 1] * 0.01, hr[2 * i] / (hr[2 * i + 1] * 0.01));
    }
    return 0;
  };
  *stop = 0;
  if (run_chain("alone (idle chip)")) return 1;
  // warm: the chip's power state after sustained load
  hipLaunchKernelGGL(k_load, dim3(1024), dim3(256), 0, s2, out, 2000, stop, 0);
  CK(hipStreamSynchronize(s2));
  if (run_chain("alone, right after 2000 x 256 MFMAs per wavefront on every CU")) return 1;
  // loads of a fixed length (~150 ms: they end by themselves), on a second stream; a third stream for the two loads together
  hipStream_t s3;
  CK(hipStreamCreateWithFlags(&s3, hipStreamNonBlocking));
  const long NB2 = 1L << 27;   // 2 GB of double2
  dv2_t* big;
  CK(hipMalloc(&big, NB2 * 16));
  CK(hipMemset(big, 0, NB2 * 16));
  struct Cfg { const char* tag; int grid; int threads; int stream_grid; } cfgs[] = {
    {"beside FP64 MFMA on every SIMD (1024 x 256 threads)", 1024, 256, 0},
    {"beside FP64 MFMA, one wavefront per SIMD on every CU (256 x 256)", 256, 256, 0},
    {"beside an HBM stream (1024 x 256 threads reading 2 GB over and over)", 0, 0, 1024},
    {"beside an HBM stream from 256 workgroups", 0, 0, 256},
    {"beside both: MFMA one wavefront per SIMD + HBM stream from 512 workgroups", 256, 256, 512},
  };
  for (auto& cf : cfgs) {
    if (cf.grid) hipLaunchKernelGGL(k_load, dim3(cf.grid), dim3(cf.threads), 0, s2, out, 30000, stop, 0);
    if (cf.stream_grid) hipLaunchKernelGGL(k_stream, dim3(cf.stream_grid), dim3(256), 0, s3, big, NB2, 400, out);
    usleep(30000);   // (load resident, power state settled)
    int rc = run_chain(cf.tag);
    CK(hipDeviceSynchronize());
    if (rc) return 1;
  }
  return 0;
}
