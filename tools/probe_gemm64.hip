// Where do the wavefronts of the 64 x 64-tile GEMM spend their K steps?  The WSOS cone's products of config 5 (U = 4845, L = 495 / 330):
// (a) Gram product Lambda = (diag(v) P)' P, upper, split-K;  (b) LU = Lambda LFLP (L x L times L x U);  and for comparison the 128-tile
// Hessian-type product.  Timed back to back with HIP events, then once more with the phase stamps of -DHYP_GEMM_PROBE summed over all
// wavefronts (shader cycles: s_memtime runs at 100 MHz on gfx950 -> the stamps are in 10 ns ticks; ratios are what matters).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DHYP_GEMM_PROBE tools/probe_gemm64.hip -o tools/_bin/probe_gemm64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../hypatia.jl_amd/csrc/gemm_f64_kernel.hpp"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

static void report(const char* what, hyp::GemmArgs g, bool transa, hyp::GemmScratch* gs, double flop) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) CK(hyp::gemm_f64_launch(0, transa, g, gs));
  const int reps = 20;
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) CK(hyp::gemm_f64_launch(0, transa, g, gs));
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0}, acc[8];
  CK(hipMemcpyToSymbol(HIP_SYMBOL(hyp::gemm_probe_acc), z, sizeof(z)));
  CK(hyp::gemm_f64_launch(0, transa, g, gs));
  CK(hipDeviceSynchronize());
  CK(hipMemcpyFromSymbol(acc, HIP_SYMBOL(hyp::gemm_probe_acc), sizeof(acc)));
  const double w = (double)acc[5], steps = (double)acc[4];
  const double tot = (double)(acc[0] + acc[1] + acc[2] + acc[3]);
  printf("%-44s %7.1f us per launch (incl. reduce), %5.1f TFLOP/s executed-as-written | wavefronts %6.0f, K steps per wavefront %5.1f | per K step (ticks): "
         "request %.1f, reads+MFMA issue %.1f, wait+LDS stores %.1f, barrier %.1f | share: %.2f %.2f %.2f %.2f | life %.0f ticks, in K loop %.2f\n",
         what, ms * 1e3 / reps, flop / (ms * 1e-3 / reps) * 1e-12, w, steps / w, acc[0] / steps, acc[1] / steps, acc[2] / steps, acc[3] / steps, acc[0] / tot, acc[1] / tot,
         acc[2] / tot, acc[3] / tot, acc[6] / w, tot / (double)acc[6]);
}

int main() {
  const int U = 4845;
  hyp::GemmScratch gs;
  for (int L : {495, 330}) {
    double *P, *SP, *LL, *LF, *LU;
    CK(hipMalloc(&P, (size_t)U * L * 8)); CK(hipMalloc(&SP, (size_t)U * L * 8)); CK(hipMalloc(&LL, (size_t)L * L * 8));
    CK(hipMalloc(&LF, (size_t)U * L * 8)); CK(hipMalloc(&LU, (size_t)U * L * 8));
    std::vector<double> h((size_t)U * L);
    for (auto& v : h) v = (double)rand() / RAND_MAX - 0.5;
    CK(hipMemcpy(P, h.data(), h.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(SP, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(LF, h.data(), h.size() * 8, hipMemcpyHostToDevice)); CK(hipMemset(LL, 0, (size_t)L * L * 8));
    char name[96];
    {   // (a) WsosCone::update_feas / lambda_of
      hyp::GemmArgs a{};
      a.M = L; a.N = L; a.K = U; a.A = SP; a.lda = U; a.B = P; a.ldb = U; a.C = LL; a.ldc = L; a.alpha = 1; a.beta = 0; a.tri = hyp::GEMM_UPPER; a.batch = 1;
      snprintf(name, sizeof name, "Gram %d x %d x %d upper (split-K)", L, L, U);
      report(name, a, true, &gs, (double)L * L * U);
    }
    {   // (b) WsosCone::partial_lambda
      hyp::GemmArgs b{};
      b.M = L; b.N = U; b.K = L; b.A = LL; b.lda = L; b.B = LF; b.ldb = L; b.C = LU; b.ldc = L; b.alpha = 1; b.beta = 0; b.batch = 1;
      snprintf(name, sizeof name, "LU = LL LFLP: %d x %d x %d", L, U, L);
      report(name, b, true, &gs, 2.0 * L * L * U);
    }
    hipFree(P); hipFree(SP); hipFree(LL); hipFree(LF); hipFree(LU);
  }
  {   // 128-tile comparison: the Hessian's U x U x L product (upper), as WsosCone::update_hess launches it (epilogue squares)
    const int L = 495;
    double *LF, *H;
    CK(hipMalloc(&LF, (size_t)U * L * 8)); CK(hipMalloc(&H, (size_t)U * U * 8));
    CK(hipMemset(LF, 0, (size_t)U * L * 8));
    hyp::GemmArgs g{};
    g.M = U; g.N = U; g.K = L; g.A = LF; g.lda = L; g.B = LF; g.ldb = L; g.C = H; g.ldc = U; g.alpha = 1; g.beta = 0; g.tri = hyp::GEMM_UPPER; g.epi = 1; g.batch = 1;
    report("Hessian part U x U x 495 upper (128 tiles)", g, true, &gs, (double)U * U * L);
    hipFree(LF); hipFree(H);
  }
  return 0;
}
