// Cycle counts (s_memtime) of the building blocks of csrc/potrf_mfma.hip inside one wavefront, and HIP-event times of the
// diagonal-block / panel kernels alone on an idle chip.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I hypatia.jl_amd/csrc
// tools/probe_potrf.hip -o tools/_bin/probe_potrf
#define HYP_PROBE 1
#include "../hypatia.jl_amd/csrc/potrf_mfma.hip"
#include <cstdio>
#include <vector>
#include <cmath>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1;} } while (0)

namespace hyp {
__global__ __launch_bounds__(64) void k_blocks(const double* in, double* out, long long* cyc, int reps) {
  __shared__ double T[TL + 16];
  const int lane = threadIdx.x, q = lane >> 4, nn = lane & 15;
  for (int e = lane; e < TL + 16; e += 64) T[e] = in[e];
  __syncthreads();
  d4_t c;
  for (int r = 0; r < 4; ++r) c[r] = in[512 + lane * 4 + r];
  double x[16], ri[16];
  long long t[9];
  t[0] = clock64();
  for (int it = 0; it < reps; ++it) { tile_gather(c, nn, x); c[0] += x[3] * 1e-30; c[1] += x[6] * 1e-30; c[2] += x[9] * 1e-30; c[3] += x[12] * 1e-30; }
  t[1] = clock64();
  for (int it = 0; it < reps; ++it) { tile_subst(x, T, T + TL); }
  t[2] = clock64();
  for (int it = 0; it < reps; ++it) { tile_scatter(x, q, c); x[0] += c[0] * 1e-30; x[5] += c[1] * 1e-30; x[10] += c[2] * 1e-30; x[15] += c[3] * 1e-30; }
  t[3] = clock64();
  // a positive definite tile for the factorization: x = column nn of (I * 20 + small)
  int fail = 0;
  for (int it = 0; it < reps; ++it) {
#pragma unroll
    for (int j = 0; j < 16; ++j) x[j] = (j == nn ? 20.0 : 0.0) + 0.01 * (j + nn) + 1e-30 * x[j];
    fail += tile_potrf(x, nn, ri);
  }
  t[4] = clock64();
  d4_t acc = {0, 0, 0, 0};
  for (int it = 0; it < reps; ++it) {
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) acc = mfma4(T[(4 * kc + q) * TS + nn], -c[kc], acc);
  }
  t[5] = clock64();
  double mi[16];
  for (int it = 0; it < reps; ++it) {
#pragma unroll
    for (int j = 0; j < 16; ++j) x[j] = (j == nn ? 20.0 : 0.0) + 0.01 * (j + nn) + 1e-30 * x[j];
    fail += tile_potrf_inv(x, nn, ri, mi);
  }
  t[6] = clock64();
  {
    double ma[4], ua[4];
    tile_inv_operands(T, T, q, nn, ma, ua);
    d4_t bt[1] = {c};
    for (int it = 0; it < reps; ++it) tile_solve_mfma<1>(bt, 1, ma, ua);
    t[7] = clock64();
    acc[0] += bt[0][0] + mi[3];
    d4_t b2[2] = {c, acc};
    for (int it = 0; it < reps; ++it) tile_solve_mfma<2>(b2, 2, ma, ua);
    acc[1] += b2[0][1] + b2[1][2];
  }
  const long long t8 = clock64();
  double s = acc[0] + acc[1] + acc[2] + acc[3] + fail;
  for (int j = 0; j < 16; ++j) s += x[j] + ri[j];
  out[lane] = s + c[0] + c[1] + c[2] + c[3];
  if (lane == 0) { for (int i = 0; i < 7; ++i) cyc[i] = (t[i + 1] - t[i]) / reps; cyc[7] = (t8 - t[7]) / reps; }
}
// shader clock against the constant 100 MHz counter, one wavefront spinning on dependent FMAs
__global__ __launch_bounds__(64) void k_clock(double* out, long long* cyc, int iters) {
  double x = threadIdx.x * 1e-3;
  const long long w0 = wall_clock64(), t0 = clock64();
  for (int it = 0; it < iters; ++it) x = fma(x, 1.0000001, 1e-9);
  const long long t1 = clock64(), w1 = wall_clock64();
  out[threadIdx.x] = x;
  if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = w1 - w0; }
}
}  // namespace hyp

int main() {
  using namespace hyp;
  const int n = 128;
  std::vector<double> h(4096);
  for (int i = 0; i < 4096; ++i) h[i] = 0.3 + 0.001 * (i % 97);
  for (int j = 0; j < 16; ++j) { h[j * TS + j] = 2.0; h[TL + j] = 0.5; }
  double *din, *dout; long long* dc;
  CK(hipMalloc(&din, 4096 * 8)); CK(hipMalloc(&dout, 4096 * 8)); CK(hipMalloc(&dc, 64 * 8));
  CK(hipMemcpy(din, h.data(), 4096 * 8, hipMemcpyHostToDevice));
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k_blocks, dim3(1), dim3(64), 0, 0, din, dout, dc, 200);
    CK(hipDeviceSynchronize());
  }
  long long hc[8];
  CK(hipMemcpy(hc, dc, 8 * 8, hipMemcpyDeviceToHost));
  printf("cycles per call (one wavefront alone): gather %lld  subst %lld  scatter %lld  potrf16 %lld  4 dependent mfma (LDS operand) %lld\n", hc[0], hc[1], hc[2], hc[3], hc[4]);
  printf("  round 4: potrf16 with the inverse riding along %lld  |  solve as 3 x 4 MFMAs with the inverse: one tile %lld, two tiles %lld\n", hc[5], hc[6], hc[7]);

  for (int iters : {20000, 2000000}) {
    hipLaunchKernelGGL(k_clock, dim3(1), dim3(64), 0, 0, dout, dc, iters);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(hc, dc, 2 * 8, hipMemcpyDeviceToHost));
    printf("one wavefront, %d dependent v_fma_f64: %lld shader cycles in %.1f us -> %.2f GHz, %.1f cycles per dependent FMA\n", iters, hc[0], hc[1] / 100.0,
           hc[0] / (hc[1] * 10.0), (double)hc[0] / iters);
  }
  // whole kernels on an idle chip
  const int N = 5000;
  std::vector<double> A((size_t)n * n);
  for (int j = 0; j < n; ++j) for (int i = 0; i < n; ++i) A[(size_t)j * n + i] = (i == j ? n : 0.0) + 1.0 / (1 + abs(i - j));
  double* dA; int* dinfo;
  CK(hipMalloc(&dA, (size_t)NB * N * 8)); CK(hipMalloc(&dinfo, 64));
  CK(hipMemset(dinfo, 0, 64));
  double* dTinv; CK(hipMalloc(&dTinv, 2048 * 8)); CK(hipMemset(dTinv, 0, 2048 * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<double> P((size_t)NB * N);
  for (size_t i = 0; i < P.size(); ++i) P[i] = 0.1 + 1e-3 * (i % 1013);
  for (int j = 0; j < n; ++j) for (int i = 0; i < n; ++i) P[(size_t)j * NB + i] = A[(size_t)j * n + i];
  const int reps = 50;
  float ms;
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipMemcpy(dA, P.data(), P.size() * 8, hipMemcpyHostToDevice));
    CK(hipEventRecord(e0, 0));
    for (int it = 0; it < reps; ++it) potrf_diag_mfma_launch(0, 1, dA, NB, 0, n, 0, dinfo, 0, dTinv, 2048);   // (refactors its own output: still positive pivots)
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
  }
  printf("potrf_diag_mfma_kernel 128x128 alone: %.2f us per launch (back to back, includes the ~1.5 us launch boundary)\n", ms / reps * 1e3);
  long long st[64];
  CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamps), sizeof(st)));
  printf("  diag stamps (cycles): load %lld | step 0: potrf16+publish %lld, solve+publish %lld, update %lld | steps 1..7 %lld | store %lld\n",
         0LL, st[1] - st[0], st[2] - st[1], st[4] - st[2], st[5] - st[4], st[6] - st[5]);
  printf("  tiles4: kernel entry -> tiles loaded %lld cycles | eight block steps %lld | \n", st[0] - st[7], st[5] - st[0]);
  {
    long long ws[8][16];
    CK(hipMemcpyFromSymbol(ws, HIP_SYMBOL(g_wstamps), sizeof(ws)));
    printf("  tiles4, block step 2 per wavefront (cycles from the step's start on wavefront 2): start | owner: diag updated, gathered, factored, flagged | updates done | flag seen, solved | at barrier, past barrier\n");
    const long long t0 = ws[2][0];
    for (int w = 0; w < 4; ++w) {
      printf("   w%d:", w);
      for (int i = 0; i < 10; ++i) printf(" %6lld", ws[w][i] ? ws[w][i] - t0 : -1LL);
      printf("\n");
    }
  }
  for (int waves = 1; waves <= 4; waves *= 2) {
    char buf[8]; snprintf(buf, 8, "%d", waves); setenv("HYP_PANEL_WAVES", buf, 1);
  }
  CK(hipMemcpy(dA, P.data(), P.size() * 8, hipMemcpyHostToDevice));
  potrf_diag_mfma_launch(0, 1, dA, NB, 0, n, 0, dinfo, 0, dTinv, 2048);
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipEventRecord(e0, 0));
    for (int it = 0; it < reps; ++it) potrf_panel_mfma_launch(0, 1, dA, NB, 0, 0, N - NB, 0, potrf_tinv_on() ? dTinv : nullptr, 2048);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
  }
  printf("potrf_panel_mfma_kernel 128 x %d alone: %.2f us per launch\n", N - NB, ms / reps * 1e3);
  CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamps), sizeof(st)));
  printf("  panel stamps (cycles): loads + U11 staging %lld | rinv %lld | step 0 %lld | steps 1..7 %lld | store %lld\n",
         st[11] - st[10], st[12] - st[11], st[13] - st[12], st[14] - st[13], st[15] - st[14]);
  // accuracy of the hardware seeds
  {
    double worst_rcp = 0, worst_rsq = 0;
    (void)worst_rcp; (void)worst_rsq;
  }
  return 0;
}
