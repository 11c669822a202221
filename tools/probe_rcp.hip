#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const double* x, double* r0, double* r1, double* r2, double* q0, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double d = x[i];
  double r = __builtin_amdgcn_rcp(d);
  r0[i] = r;
  double e = fma(-d, r, 1.0);
  r1[i] = fma(r, e, r);               // one quadratic step
  r2[i] = fma(r, fma(e, e, e), r);    // one cubic step
  q0[i] = __builtin_amdgcn_rsq(d);
}
int main() {
  const int n = 1 << 20;
  std::vector<double> x(n);
  for (int i = 0; i < n; ++i) x[i] = ldexp(1.0 + (double)rand() / RAND_MAX, (rand() % 40) - 20);
  double *dx, *d0, *d1, *d2, *d3;
  hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8); hipMalloc(&d3, n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, d2, d3, n);
  std::vector<double> a(n), b(n), c(n), q(n);
  hipMemcpy(a.data(), d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), d1, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(c.data(), d2, n * 8, hipMemcpyDeviceToHost); hipMemcpy(q.data(), d3, n * 8, hipMemcpyDeviceToHost);
  double m0 = 0, m1 = 0, m2 = 0, mq = 0;
  for (int i = 0; i < n; ++i) {
    long double t = 1.0L / (long double)x[i];
    m0 = fmax(m0, (double)fabsl(((long double)a[i] - t) / t));
    m1 = fmax(m1, (double)fabsl(((long double)b[i] - t) / t));
    m2 = fmax(m2, (double)fabsl(((long double)c[i] - t) / t));
    long double s = 1.0L / sqrtl((long double)x[i]);
    mq = fmax(mq, (double)fabsl(((long double)q[i] - s) / s));
  }
  printf("max rel err: v_rcp_f64 %.3e, +1 quadratic step %.3e, +1 cubic step %.3e; v_rsq_f64 %.3e (eps = 2.2e-16)\n", m0, m1, m2, mq);
  return 0;
}
