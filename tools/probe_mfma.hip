// FP64 MFMA issue-rate probe: cycles per v_mfma_f64_16x16x4_f64 per SIMD as a function of waves/SIMD,
// effective clock (s_memtime vs wall_clock64), chip TFLOP/s.  Roofline denominator for DESIGN.md.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4_t __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1;} } while (0)

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma(double* out, long long* cyc, int iters) {
  d4_t acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (d4_t){0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = threadIdx.x * 2e-3;
  long long t0 = clock64();
  long long w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  long long t1 = clock64();
  long long w1 = wall_clock64();
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { cyc[2 * blockIdx.x] = t1 - t0; cyc[2 * blockIdx.x + 1] = w1 - w0; }
}

__global__ __launch_bounds__(256) void k_fma(double* out, int iters) {
  double x[16];
  for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 1e-3 + i;
  double a = 1.0000001, b = 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = fma(x[i], a, b);
  }
  double s = 0; for (int i = 0; i < 16; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(1024) void k_mfma1024(double* out, long long* cyc, int iters) {
  d4_t acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = (d4_t){0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = threadIdx.x * 2e-3;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[(blockIdx.x * blockDim.x + threadIdx.x) % 4096] = s;
}

__global__ __launch_bounds__(1024) void k_mfma1024_rand(double* out, int iters) {
  d4_t acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = (d4_t){0, 0, 0, 0};
  unsigned long long s = 88172645463325252ULL + threadIdx.x * 7919ULL + blockIdx.x * 104729ULL;
  double a[4], b[4];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {   // cheap xorshift -> doubles in [1, 2)
      s ^= s << 13; s ^= s >> 7; s ^= s << 17;
      a[i] = __longlong_as_double((long long)((s >> 12) | 0x3FF0000000000000ULL)) - 1.5;
      b[i] = __longlong_as_double((long long)(((s * 2654435761ULL) >> 12) | 0x3FF0000000000000ULL)) - 1.5;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[i], acc[i], 0, 0, 0);
  }
  double t = 0;
  for (int i = 0; i < 4; ++i) t += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[(blockIdx.x * blockDim.x + threadIdx.x) % 4096] = t;
}

__global__ __launch_bounds__(512) void k_mixed(double* out, int iters) {
  int wave = threadIdx.x >> 6;
  double s = 0;
  if (wave & 1) {   // waves alternate SIMDs 0->2->1->3; odd/even split puts both kinds on every SIMD pair
    double x[16];
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 1e-3 + i;
    double a = 1.0000001, b = 1e-9;
    for (int it = 0; it < iters * 2; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) x[i] = fma(x[i], a, b);
    }
    for (int i = 0; i < 16; ++i) s += x[i];
  } else {
    d4_t acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (d4_t){0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = threadIdx.x * 2e-3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  }
  out[(blockIdx.x * blockDim.x + threadIdx.x) % 4096] = s;
}

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  double* out; CK(hipMalloc(&out, (size_t)cus * 8 * 256 * 8));
  long long* cyc; CK(hipMalloc(&cyc, (size_t)cus * 8 * 2 * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int iters = 40000;
  for (int bpc : {1, 2, 3, 4, 6, 8}) {
    int blocks = cus * bpc;
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_mfma<4>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep == 1) {
        std::vector<long long> h(2 * blocks); CK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
        double c = 0, w = 0; for (int i = 0; i < blocks; ++i) { c += h[2 * i]; w += h[2 * i + 1]; }
        c /= blocks; w /= blocks;
        double flops = (double)blocks * 4 * iters * 4 * 2048.0;
        printf("mfma_f64 waves/SIMD=%d: %.3f ms  %.2f TFLOP/s | clock64/MFMA(per wave)=%.1f  clk64=%.0f wall100MHz=%.0f -> clk64 rate %.1f MHz\n", bpc, ms,
               flops / ms * 1e-9, c / (iters * 4.0), c, w, c / (w / 100.0));
      }
    }
  }
  // few-CU run (1024-thread blocks = 4 waves/SIMD): is the MFMA rate a power/clock limit or architectural?
  for (int blocks : {8, 32, 128, 256, 512}) {
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_mfma1024, dim3(blocks), dim3(1024), 0, 0, out, cyc, iters);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep == 1) printf("mfma_f64 1024-thread blocks=%d: %.3f ms  ns per MFMA per SIMD = %.2f  (%.2f TFLOP/s)\n", blocks, ms,
                           ms * 1e6 / (iters * 4.0 * 4.0 * ((blocks + cus - 1) / cus)), (double)blocks * 16 * iters * 4 * 2048.0 / ms * 1e-9);
    }
  }
  // random-ish operand data that changes every iteration (DVFS: data toggling lowers the sustained clock)
  for (int blocks : {256, 512}) {
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_mfma1024_rand, dim3(blocks), dim3(1024), 0, 0, out, iters);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep == 1) printf("mfma_f64 1024-thread RANDOM operands blocks=%d: %.3f ms (%.2f TFLOP/s)\n", blocks, ms, (double)blocks * 16 * iters * 4 * 2048.0 / ms * 1e-9);
    }
  }
  // concurrent MFMA + VALU FMA (mixed kernel: even waves MFMA, odd waves VALU)
  for (int bpc : {2, 4}) {
    int blocks = cus * bpc;
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_mixed, dim3(blocks), dim3(512), 0, 0, out, iters);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      double fm = (double)blocks * 4 * iters * 4 * 2048.0, fv = (double)blocks * 256 * (iters * 2) * 16 * 2.0;
      if (rep == 1) printf("mixed (4 MFMA waves + 4 VALU waves per block) blocks/CU=%d: %.3f ms  mfma %.2f + valu %.2f = %.2f TFLOP/s\n", bpc, ms, fm / ms * 1e-9, fv / ms * 1e-9, (fm + fv) / ms * 1e-9);
    }
  }
  for (int bpc : {1, 2, 4, 8}) {
    int blocks = cus * bpc;
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(256), 0, 0, out, iters);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep == 1) printf("v_fma_f64 waves/SIMD=%d: %.3f ms %.2f TFLOP/s\n", bpc, ms, (double)blocks * 256 * iters * 16 * 2.0 / ms * 1e-9);
    }
  }
  return 0;
}
