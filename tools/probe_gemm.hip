// GPU probe: (1) v_mfma_f64_16x16x4_f64 fragment layout check, (2) FP64 MFMA peak microbenchmark
// (the roofline denominator for the Schur assembly / factorization kernels), (3) gemm_f64 correctness
// against a host triple loop, (4) syrk-shaped timing at the config-2 size (n = 5000, q = 20100).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe_gemm.hip -o tools/probe_gemm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "../hypatia.jl_amd/csrc/gemm_f64_kernel.hpp"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

using hyp::d4_t;

__global__ void mfma_layout_kernel(const double* a, const double* b, double* d) {
  // a: 16x4 row-major [i][k]; b: 4x16 row-major [k][j]; d: 16x16 row-major [i][j]
  int l = threadIdx.x;
  d4_t acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[(l & 15) * 4 + (l >> 4)], b[(l >> 4) * 16 + (l & 15)], acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) d[((l >> 4) + 4 * r) * 16 + (l & 15)] = acc[r];
}

template <int NACC>
__global__ __launch_bounds__(256) void mfma_peak_kernel(double* out, int iters) {
  d4_t acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (d4_t){0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = threadIdx.x * 2e-3;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static double frand() { return (double)rand() / RAND_MAX * 2.0 - 1.0; }

static int check_gemm(bool transa, int M, int N, int K, int tri, int krange, double alpha, double beta) {
  long lda = transa ? K + 3 : M + 1, ldb = K + 2, ldc = M + 5;
  std::vector<double> A((size_t)lda * (transa ? M : K)), B((size_t)ldb * N), C((size_t)ldc * N), C0;
  for (auto& v : A) v = frand();
  for (auto& v : B) v = frand();
  for (auto& v : C) v = frand();
  // enforce structural zeros for triangular ranges
  auto opA = [&](int m, int k) -> double& { return transa ? A[(size_t)m * lda + k] : A[(size_t)k * lda + m]; };
  for (int m = 0; m < M; ++m) for (int k = 0; k < K; ++k) {
    if (krange == hyp::KR_LE_M && k > m) opA(m, k) = 0;
    if (krange == hyp::KR_GE_M && k < m) opA(m, k) = 0;
  }
  for (int n = 0; n < N; ++n) for (int k = 0; k < K; ++k) {
    if (krange == hyp::KR_LE_N && k > n) B[(size_t)n * ldb + k] = 0;
    if (krange == hyp::KR_GE_N && k < n) B[(size_t)n * ldb + k] = 0;
  }
  C0 = C;
  double *dA, *dB, *dC;
  CK(hipMalloc(&dA, A.size() * 8)); CK(hipMalloc(&dB, B.size() * 8)); CK(hipMalloc(&dC, C.size() * 8));
  CK(hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dC, C.data(), C.size() * 8, hipMemcpyHostToDevice));
  hyp::GemmArgs g{};
  g.M = M; g.N = N; g.K = K; g.A = dA; g.lda = lda; g.B = dB; g.ldb = ldb; g.C = dC; g.ldc = ldc;
  g.alpha = alpha; g.beta = beta; g.tri = tri; g.krange = krange; g.batch = 1;
  CK(hyp::gemm_f64_launch(0, transa, g));
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(C.data(), dC, C.size() * 8, hipMemcpyDeviceToHost));
  double maxerr = 0; long untouched_bad = 0;
  for (int n = 0; n < N; ++n) for (int m = 0; m < M; ++m) {
    size_t ci = (size_t)n * ldc + m;
    if (tri == hyp::GEMM_UPPER && m > n) { if (C[ci] != C0[ci]) ++untouched_bad; continue; }
    double s = 0;
    for (int k = 0; k < K; ++k) s += opA(m, k) * B[(size_t)n * ldb + k];
    double ref = alpha * s + beta * C0[ci];
    maxerr = fmax(maxerr, fabs(ref - C[ci]));
  }
  // padding rows untouched?
  for (int n = 0; n < N; ++n) for (long m = M; m < ldc; ++m) if (C[(size_t)n * ldc + m] != C0[(size_t)n * ldc + m]) ++untouched_bad;
  printf("gemm %s M=%d N=%d K=%d tri=%d kr=%d alpha=%g beta=%g : maxerr=%.3e untouched_bad=%ld %s\n", transa ? "TN" : "NN", M, N, K, tri, krange,
         alpha, beta, maxerr, untouched_bad, (maxerr < 1e-11 * K && untouched_bad == 0) ? "OK" : "FAIL");
  hipFree(dA); hipFree(dB); hipFree(dC);
  return (maxerr < 1e-11 * K && untouched_bad == 0) ? 0 : 1;
}

int main(int argc, char** argv) {
  int fails = 0;
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("device %s CUs=%d clock=%d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
  // 1. layout
  {
    std::vector<double> a(64), b(64), d(256), ref(256, 0.0);
    for (auto& v : a) v = frand();
    for (auto& v : b) v = frand();
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 4; ++k) ref[i * 16 + j] += a[i * 4 + k] * b[k * 16 + j];
    double *da, *db, *dd; CK(hipMalloc(&da, 512)); CK(hipMalloc(&db, 512)); CK(hipMalloc(&dd, 2048));
    CK(hipMemcpy(da, a.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), 512, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(mfma_layout_kernel, dim3(1), dim3(64), 0, 0, da, db, dd);
    CK(hipMemcpy(d.data(), dd, 2048, hipMemcpyDeviceToHost));
    double e = 0; for (int i = 0; i < 256; ++i) e = fmax(e, fabs(d[i] - ref[i]));
    printf("mfma f64 16x16x4 layout check: maxerr=%.3e %s\n", e, e < 1e-14 ? "OK" : "FAIL");
    if (!(e < 1e-14)) ++fails;
  }
  // 2. peak
  {
    double* out; CK(hipMalloc(&out, (size_t)4096 * 256 * 8));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int nacc : {4, 8}) {
      for (int wpb : {4, 8}) {   // waves per block -> waves per SIMD with 1 block/CU... use 256 CU * k blocks
        int iters = 20000; int blocks = prop.multiProcessorCount * (wpb / 4);
        for (int rep = 0; rep < 2; ++rep) {
          hipEventRecord(e0);
          if (nacc == 4) hipLaunchKernelGGL(mfma_peak_kernel<4>, dim3(blocks), dim3(256), 0, 0, out, iters);
          else hipLaunchKernelGGL(mfma_peak_kernel<8>, dim3(blocks), dim3(256), 0, 0, out, iters);
          hipEventRecord(e1); hipEventSynchronize(e1);
          float ms; hipEventElapsedTime(&ms, e0, e1);
          double flops = (double)blocks * 4 * iters * nacc * 2.0 * 16 * 16 * 4;
          if (rep == 1) printf("mfma f64 peak: nacc=%d blocks=%d : %.3f ms, %.2f TFLOP/s\n", nacc, blocks, ms, flops / ms * 1e-9);
        }
      }
    }
    hipFree(out);
  }
  // 3. correctness
  fails += check_gemm(true, 300, 200, 100, hyp::GEMM_FULL, hyp::KR_ALL, 1.0, 0.0);
  fails += check_gemm(true, 129, 257, 33, hyp::GEMM_FULL, hyp::KR_ALL, -0.5, 1.0);
  fails += check_gemm(false, 300, 200, 100, hyp::GEMM_FULL, hyp::KR_ALL, 1.0, 0.0);
  fails += check_gemm(false, 131, 77, 45, hyp::GEMM_FULL, hyp::KR_ALL, 2.0, -1.0);
  fails += check_gemm(true, 300, 300, 77, hyp::GEMM_UPPER, hyp::KR_ALL, 1.0, 0.0);
  fails += check_gemm(true, 300, 300, 300, hyp::GEMM_FULL, hyp::KR_LE_M, 1.0, 0.0);
  fails += check_gemm(true, 300, 300, 300, hyp::GEMM_FULL, hyp::KR_GE_M, 1.0, 0.0);
  fails += check_gemm(false, 300, 300, 300, hyp::GEMM_FULL, hyp::KR_LE_N, 1.0, 0.0);
  fails += check_gemm(false, 300, 300, 300, hyp::GEMM_FULL, hyp::KR_GE_N, 1.0, 1.0);
  // 4. syrk timing at config-2 shape
  {
    int n = 5000, q = 20100;
    if (argc > 2) { n = atoi(argv[1]); q = atoi(argv[2]); }
    double *dA, *dC; CK(hipMalloc(&dA, (size_t)q * n * 8)); CK(hipMalloc(&dC, (size_t)n * n * 8));
    std::vector<double> h((size_t)q * n);
    for (auto& v : h) v = frand();
    CK(hipMemcpy(dA, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    hyp::GemmArgs g{};
    g.M = n; g.N = n; g.K = q; g.A = dA; g.lda = q; g.B = dA; g.ldb = q; g.C = dC; g.ldc = n; g.alpha = 1; g.beta = 0;
    g.tri = hyp::GEMM_UPPER; g.krange = hyp::KR_ALL; g.batch = 1;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 4; ++rep) {
      hipEventRecord(e0);
      CK(hyp::gemm_f64_launch(0, true, g));
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("syrk upper n=%d q=%d: %.3f ms, %.2f TFLOP/s algorithmic (n^2 q)\n", n, q, ms, (double)n * n * q / ms * 1e-9);
    }
    // spot check a few entries
    std::vector<double> c((size_t)n * n);
    CK(hipMemcpy(c.data(), dC, c.size() * 8, hipMemcpyDeviceToHost));
    double me = 0;
    for (int t = 0; t < 20; ++t) {
      int i = rand() % n, j = rand() % n; if (i > j) std::swap(i, j);
      double s = 0; for (int k = 0; k < q; ++k) s += h[(size_t)i * q + k] * h[(size_t)j * q + k];
      me = fmax(me, fabs(s - c[(size_t)j * n + i]) / (fabs(s) + 1));
    }
    printf("syrk spot check rel err %.3e %s\n", me, me < 1e-11 ? "OK" : "FAIL");
    if (!(me < 1e-11)) ++fails;
    // DVFS check: the same syrk on zero-filled operands (no data toggling) and with the split-K path
    {
      std::vector<double> z((size_t)q * n, 0.0);
      double* dZ; CK(hipMalloc(&dZ, (size_t)q * n * 8));
      CK(hipMemcpy(dZ, z.data(), z.size() * 8, hipMemcpyHostToDevice));
      hyp::GemmArgs gz = g; gz.A = dZ; gz.B = dZ;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        CK(hyp::gemm_f64_launch(0, true, gz));
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("syrk upper ZERO operands: %.3f ms, %.2f TFLOP/s\n", ms, (double)n * n * q / ms * 1e-9);
      }
      hyp::GemmArgs gs = g; gs.tag = 1;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        CK(hyp::gemm_f64_launch(0, true, gs));
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("syrk upper split-K (tag 1) random: %.3f ms, %.2f TFLOP/s\n", ms, (double)n * n * q / ms * 1e-9);
      }
      gs.A = dZ; gs.B = dZ;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        CK(hyp::gemm_f64_launch(0, true, gs));
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("syrk upper split-K ZERO operands: %.3f ms, %.2f TFLOP/s\n", ms, (double)n * n * q / ms * 1e-9);
      }
      hipFree(dZ);
    }
    g.tri = hyp::GEMM_FULL;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      CK(hyp::gemm_f64_launch(0, true, g));
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("gemm full n=%d q=%d: %.3f ms, %.2f TFLOP/s (2 n^2 q)\n", n, q, ms, 2.0 * n * n * q / ms * 1e-9);
    }
  }
  printf("PROBE %s (%d failures)\n", fails ? "FAIL" : "PASS", fails);
  return fails ? 1 : 0;
}
