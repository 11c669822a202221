// EpiNormSpectral on the device: (u, W), u >= sigma_1(W), W is d1 x d2 (column-stacked), d1 <= d2.
// Reference: /root/reference/src/Cones/epinormspectral.jl (real case; line ranges inline).
// Dual feasibility needs the nuclear norm of a d1 x d2 matrix (svdvals!, :125-132): computed by a
// one-sided Jacobi (Hestenes) iteration on the rows, one workgroup per row pair per round.
#include "cones.hpp"

namespace hyp {

static const double EPS = 2.220446049250313e-16;

__global__ void scaled_identity_kernel(int n, double* A, long lda, double val) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y;
  if (i < n) A[(long)j * lda + i] = (i == j) ? val : 0.0;
}
__global__ void trace_kernel(int n, const double* A, long lda, double* out) {   // single block
  __shared__ double red[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += A[(long)i * lda + i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0];
}
// out = sum_j ||V[:, j]||_2 over the m columns (length len) of V, one workgroup: a wavefront per column (fixed order within
// the column: lane-strided partial sums, butterfly), the column norms in `norms` (m doubles), their sum in index order by one
// tree.  Replaces one dot-product launch per column (50 launches of 4 us per nuclear norm at 50 x 100).
__global__ __launch_bounds__(256) void colnorm_sum_kernel(int len, int m, const double* __restrict__ V, double* __restrict__ norms, double* __restrict__ out) {
  __shared__ double red[256];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int j = w; j < m; j += 4) {
    const double* v = V + (long)j * len;
    double s = 0.0;
    for (int r = lane; r < len; r += 64) s = fma(v[r], v[r], s);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) norms[j] = sqrt(s);
  }
  __syncthreads();
  double s = 0.0;
  for (int i = threadIdx.x; i < m; i += 256) s += norms[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0];
}
// S_j = T_j + T_j' - 2 u a_u[j] I   (batched d1 x d1; a_u[j] = arr[0 + j * lda])
__global__ void symm_shift_kernel(int d1, const double* __restrict__ T, double* __restrict__ S, const double* __restrict__ arr, long lda,
                                  double u) {
  const int j = blockIdx.z;
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  const int c = blockIdx.y;
  if (r >= d1) return;
  const double* t = T + (long)j * d1 * d1;
  double v = t[(long)c * d1 + r] + t[(long)r * d1 + c];
  if (r == c) v -= 2.0 * u * arr[(long)j * lda];
  S[(long)j * d1 * d1 + (long)c * d1 + r] = v;
}
// out[0, j] = Huu * arr[0, j] + <HuW, arr[1:, j]>
__global__ __launch_bounds__(256) void hp_first_row_kernel(int dw, double Huu, const double* __restrict__ HuW, const double* __restrict__ arr,
                                                           long lda, double* __restrict__ prod, long ldp) {
  __shared__ double red[256];
  const int j = blockIdx.x;
  const double* a = arr + (long)j * lda;
  double s = 0.0;
  for (int i = threadIdx.x; i < dw; i += 256) s += HuW[i] * a[1 + i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) prod[(long)j * ldp] = Huu * a[0] + red[0];
}
// explicit Hessian, upper triangle (epinormspectral.jl:172-209): rows/cols 1.. indexed (j + i*d1), (l + k*d1)
__global__ void ens_hess_kernel(int d1, int d2, const double* __restrict__ Zi, const double* __restrict__ tau, const double* __restrict__ WtauI,
                                const double* __restrict__ HuW, double Huu, double* __restrict__ H, long ldh) {
  const int dw = d1 * d2;
  const int r = blockIdx.x * blockDim.x + threadIdx.x;   // 0 .. dw (0 = u row)
  const int c = blockIdx.y;                                // 0 .. dw
  if (r > dw || r > c) return;
  double v;
  if (r == 0) {
    v = (c == 0) ? Huu : HuW[c - 1];
  } else {
    const int rr = r - 1, cc = c - 1;
    const int j = rr % d1, i = rr / d1, l = cc % d1, k = cc / d1;
    v = 2.0 * (Zi[(long)j * d1 + l] * WtauI[(long)k * d2 + i] + tau[(long)i * d1 + l] * tau[(long)k * d1 + j]);
  }
  H[(long)c * ldh + r] = v;
}
// one-sided Jacobi round: workgroup b rotates columns (p, q) of V (len x m, ld = len) chosen by the
// round-robin tournament; flag[0] counts rotations above the threshold in this sweep
__global__ __launch_bounds__(256) void jacobi_round_kernel(int len, int m, int mm, int t, double* __restrict__ V, long ldv, int* __restrict__ flag) {
  __shared__ double red[3][256];
  const int i = blockIdx.x;
  int p, q;
  if (i == 0) { p = mm - 1; q = t; }
  else { p = (t + i) % (mm - 1); q = (t - i + (mm - 1)) % (mm - 1); }
  if (p >= m || q >= m) return;   // dummy player of an odd tournament
  double* vp = V + (long)p * ldv;
  double* vq = V + (long)q * ldv;
  double a = 0.0, b = 0.0, g = 0.0;
  for (int r = threadIdx.x; r < len; r += 256) {
    const double x = vp[r], y = vq[r];
    a += x * x; b += y * y; g += x * y;
  }
  red[0][threadIdx.x] = a; red[1][threadIdx.x] = b; red[2][threadIdx.x] = g;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) {
      red[0][threadIdx.x] += red[0][threadIdx.x + off];
      red[1][threadIdx.x] += red[1][threadIdx.x + off];
      red[2][threadIdx.x] += red[2][threadIdx.x + off];
    }
    __syncthreads();
  }
  a = red[0][0]; b = red[1][0]; g = red[2][0];
  if (fabs(g) <= 1e-15 * sqrt(a * b) || g == 0.0) return;
  if (threadIdx.x == 0) atomicAdd(flag, 1);
  const double zeta = (b - a) / (2.0 * g);
  const double tt = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
  const double cs = 1.0 / sqrt(1.0 + tt * tt), sn = cs * tt;
  for (int r = threadIdx.x; r < len; r += 256) {
    const double x = vp[r], y = vq[r];
    vp[r] = cs * x - sn * y;
    vq[r] = sn * x + cs * y;
  }
}

// the same round with the rotations accumulated: columns p, q of J (m x m) get the rotation applied to columns p, q of V, so
// that V_final = V_initial J with J orthogonal
__global__ __launch_bounds__(256) void jacobi_round_vec_kernel(int len, int m, int mm, int t, double* __restrict__ V, long ldv, double* __restrict__ J,
                                                               int* __restrict__ flag) {
  __shared__ double red[3][256];
  const int i = blockIdx.x;
  int p, q;
  if (i == 0) { p = mm - 1; q = t; }
  else { p = (t + i) % (mm - 1); q = (t - i + (mm - 1)) % (mm - 1); }
  if (p >= m || q >= m) return;
  double* vp = V + (long)p * ldv;
  double* vq = V + (long)q * ldv;
  double a = 0.0, b = 0.0, g = 0.0;
  for (int r = threadIdx.x; r < len; r += 256) {
    const double x = vp[r], y = vq[r];
    a += x * x; b += y * y; g += x * y;
  }
  red[0][threadIdx.x] = a; red[1][threadIdx.x] = b; red[2][threadIdx.x] = g;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) {
      red[0][threadIdx.x] += red[0][threadIdx.x + off];
      red[1][threadIdx.x] += red[1][threadIdx.x + off];
      red[2][threadIdx.x] += red[2][threadIdx.x + off];
    }
    __syncthreads();
  }
  a = red[0][0]; b = red[1][0]; g = red[2][0];
  if (fabs(g) <= 1e-15 * sqrt(a * b) || g == 0.0) return;
  if (threadIdx.x == 0) atomicAdd(flag, 1);
  const double zeta = (b - a) / (2.0 * g);
  const double tt = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
  const double cs = 1.0 / sqrt(1.0 + tt * tt), sn = cs * tt;
  for (int r = threadIdx.x; r < len; r += 256) {
    const double x = vp[r], y = vq[r];
    vp[r] = cs * x - sn * y;
    vq[r] = sn * x + cs * y;
  }
  double* jp = J + (long)p * m;
  double* jq = J + (long)q * m;
  for (int r = threadIdx.x; r < m; r += 256) {
    const double x = jp[r], y = jq[r];
    jp[r] = cs * x - sn * y;
    jq[r] = sn * x + cs * y;
  }
}

// The whole one-sided Jacobi iteration of a SMALL matrix in one launch: B (len x m) and, optionally, the accumulated
// rotations J (m x m) live in LDS, one workgroup sweeps the round-robin tournament (m - 1 rounds of m / 2 disjoint pairs, a
// barrier per round) until a full sweep rotates nothing.  Same pairing, rotation formula and stopping rule as the
// multi-launch rounds above, which took ~400 launches of 3 us per SVD at 50 x 100 (the line search of matrix completion
// spent its time there once the explicit Hessian was gone); beyond the LDS the multi-launch form remains.
// 32 lanes per pair, 1024 threads: 32 pairs at a time.  What a pair costs is latency (a dependent FP64 operation is 32 cycles,
// tools/probe_potrf.hip): row sums by DPP, and the rotation from v_rcp / v_rsq seeds with Newton steps instead of the
// division / square-root expansions (three divisions and two roots were ~2.5k cycles per pair).
template <int CTRL>
__device__ __forceinline__ double jl_dpp(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double jl_sum32(double v) {   // total over the 32 lanes of a half wavefront, in every lane
  v += jl_dpp<0xB1>(v);    // quad_perm [1, 0, 3, 2]
  v += jl_dpp<0x4E>(v);    // quad_perm [2, 3, 0, 1]
  v += jl_dpp<0x141>(v);   // row_half_mirror
  v += jl_dpp<0x140>(v);   // row_mirror
  return v + __shfl_xor(v, 16, 32);
}
__device__ __forceinline__ double jl_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  const double e = fma(-x, r, 1.0);
  return fma(r, fma(e, e, e), r);
}
__device__ __forceinline__ double jl_rsqrt(double x) {
  double r = __builtin_amdgcn_rsq(x);
  const double h = 0.5 * x;
  double e = fma(-h * r, r, 0.5);
  r = fma(r, e, r);
  e = fma(-h * r, r, 0.5);
  return fma(r, e, r);
}
// tol_rot: a pair is rotated while its cosine |g| / sqrt(a b) exceeds it; tol_big: a sweep whose rotated pairs all had a cosine
// <= tol_big is the last one (it leaves cosines of the order of m tol_big^2 behind); floor_rel: columns whose norm is below
// floor_rel ||B||_F are numerically zero and take no part (0: every column does).
__global__ __launch_bounds__(1024) void jacobi_lds_kernel(int len, int m, double* __restrict__ Vg, double* __restrict__ Jg, int max_sweeps, int load_j,
                                                          int* __restrict__ sweeps_out, double tol_rot, double tol_big, double floor_rel) {
  extern __shared__ __attribute__((aligned(16))) double jl_lds[];
  const int ldv = len | 1;                       // odd stride
  double* V = jl_lds;                            // m columns of length len
  double* J = jl_lds + (long)m * ldv;            // m columns of length m (only if Jg)
  const int ldj = m | 1;
  __shared__ int rotated;
  const int tid = threadIdx.x;
  for (long e = tid; e < (long)len * m; e += 1024) V[(e / len) * ldv + (e % len)] = Vg[e];
  if (Jg) for (long e = tid; e < (long)m * m; e += 1024) J[(e / m) * ldj + (e % m)] = load_j ? Jg[e] : (((e / m) == (e % m)) ? 1.0 : 0.0);   // (load_j: warm start, the rotations continue an earlier product)
  __syncthreads();
  double floor2 = 0.0;
  if (floor_rel > 0.0) {   // ||B||_F^2 (invariant under the rotations), fixed summation order
    __shared__ double fr[1024];
    double f = 0.0;
    for (long e = tid; e < (long)len * m; e += 1024) { const double x = V[(e / len) * ldv + (e % len)]; f = fma(x, x, f); }
    fr[tid] = f;
    __syncthreads();
    for (int off = 512; off > 0; off >>= 1) {
      if (tid < off) fr[tid] += fr[tid + off];
      __syncthreads();
    }
    floor2 = floor_rel * floor_rel * fr[0];
  }
  const int mm = (m % 2 == 0) ? m : m + 1;
  const int sub = tid & 31, grp = tid >> 5;      // 32 groups of 32 lanes
  for (int sweep = 0; sweep < max_sweeps && m > 1; ++sweep) {
    if (tid == 0) rotated = 0;
    __syncthreads();
    for (int t = 0; t < mm - 1; ++t) {
      for (int i = grp; i < mm / 2; i += 32) {
        int p, q;
        if (i == 0) { p = mm - 1; q = t; }
        else { p = (t + i) % (mm - 1); q = (t - i + (mm - 1)) % (mm - 1); }
        if (p >= m || q >= m) continue;          // dummy player of an odd tournament (uniform over the group)
        double* vp = V + (long)p * ldv;
        double* vq = V + (long)q * ldv;
        double a = 0.0, b = 0.0, g = 0.0;
        // columns of up to 128 entries stay in registers between the scalar products and the rotation (all LDS reads of the
        // pair issued at once, none repeated); same sums in the same order as the loop form
        const bool in_regs = (len <= 128);
        double xr[4], yr[4];
        if (in_regs) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int r = sub + 32 * k;
            xr[k] = (r < len) ? vp[r] : 0.0;
            yr[k] = (r < len) ? vq[r] : 0.0;
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) { a = fma(xr[k], xr[k], a); b = fma(yr[k], yr[k], b); g = fma(xr[k], yr[k], g); }
        } else {
          for (int r = sub; r < len; r += 32) {
            const double x = vp[r], y = vq[r];
            a = fma(x, x, a); b = fma(y, y, b); g = fma(x, y, g);
          }
        }
        a = jl_sum32(a); b = jl_sum32(b); g = jl_sum32(g);
        // |g| <= tol sqrt(a b) compared on the squares (the double-precision square root expands to ~15 dependent operations in
        // the middle of the round's critical path); outside the range where a b and g^2 are safely representable, as written
        const double ab = a * b, g2 = g * g;
        const bool sq_ok = ab > 1e-280 && ab < 1e280;
        const bool tiny = sq_ok ? (g2 <= tol_rot * tol_rot * ab) : (fabs(g) <= tol_rot * sqrt(ab));
        if (tiny || g == 0.0 || fmin(a, b) <= floor2) continue;
        // (outside the range where the seeds + Newton steps are safe -- g or zeta near the ends of the exponent range, which
        //  happens when columns of W are ~1e-150 at the end of a solve with a zero optimum -- the plain expansions are used)
        double cs, sn;
        const double ag = fabs(g), az = fabs(b - a);
        if (ag > 1e-140 && ag < 1e140 && az < 1e140 * ag) {
          const double zeta = (b - a) * 0.5 * jl_rcp(g);
          const double w1 = fma(zeta, zeta, 1.0);
          const double hyp1 = w1 * jl_rsqrt(w1);                    // sqrt(1 + zeta^2)
          const double tt = copysign(jl_rcp(fabs(zeta) + hyp1), zeta);
          cs = jl_rsqrt(fma(tt, tt, 1.0));
          sn = cs * tt;
        } else {
          const double zeta = (b - a) / (2.0 * g);
          const double tt = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
          cs = 1.0 / sqrt(1.0 + tt * tt);
          sn = cs * tt;
        }
        if (!(cs == cs) || !(sn == sn)) continue;
        // a sweep whose rotated pairs all had a small cosine leaves cosines of the order of its square behind (below the
        // rotation threshold): it is the last one -- no further sweep just to find nothing to rotate
        if (sub == 0 && (sq_ok ? (g2 > tol_big * tol_big * ab) : (ag > tol_big * sqrt(ab)))) rotated = 1;
        if (in_regs) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int r = sub + 32 * k;
            if (r < len) {
              vp[r] = cs * xr[k] - sn * yr[k];
              vq[r] = sn * xr[k] + cs * yr[k];
            }
          }
        } else {
          for (int r = sub; r < len; r += 32) {
            const double x = vp[r], y = vq[r];
            vp[r] = cs * x - sn * y;
            vq[r] = sn * x + cs * y;
          }
        }
        if (Jg) {
          double* jp = J + (long)p * ldj;
          double* jq = J + (long)q * ldj;
          for (int r = sub; r < m; r += 32) {
            const double x = jp[r], y = jq[r];
            jp[r] = cs * x - sn * y;
            jq[r] = sn * x + cs * y;
          }
        }
      }
      __syncthreads();
    }
    if (tid == 0 && sweeps_out) *sweeps_out = sweep + 1;
    if (rotated == 0) break;
    __syncthreads();
  }
  __syncthreads();
  for (long e = tid; e < (long)len * m; e += 1024) Vg[e] = V[(e / len) * ldv + (e % len)];
  if (Jg) for (long e = tid; e < (long)m * m; e += 1024) Jg[e] = J[(e / m) * ldj + (e % m)];
}
// bytes of LDS the one-launch form needs; 0 = does not fit
static size_t jacobi_lds_bytes(int len, int m, bool with_j) {
  const size_t b = ((size_t)m * (len | 1) + (with_j ? (size_t)m * (m | 1) : 0)) * sizeof(double);
  return b <= 150 * 1024 ? b : 0;
}
// values_only: the caller wants the SUM of the singular values (the nuclear norm of the dual feasibility test).  With cosines
// c_pq left between the columns, the Gram matrix is D (I + C) D: its trace is exact, simple eigenvalues move in second order and
// a cluster d^2 (1 +- c) contributes d (sqrt(1 + c) + sqrt(1 - c)) = d (2 - c^2 / 4) -- the sum of the column norms is the nuclear
// norm up to O(||C||_F^2) relatively, so cosines of 1e-10 (||C||_F^2 <= m^2 1e-20) are as good as 1e-15 there; and columns below
// eps ||B||_F are numerically zero, as for LAPACK's own singular values (absolute accuracy eps sigma_max): at a dual point close
// to the boundary of the cone most columns are such noise, and rotating noise against noise never settles (20 sweeps and more).
static bool jacobi_in_lds(Ctx& ctx, int len, int m, double* V, double* J, bool load_j = false, bool values_only = false) {
  static const bool on = [] { const char* e = getenv("HYP_JACOBI_LDS"); return !(e && e[0] == '0'); }();
  const size_t lds = jacobi_lds_bytes(len, m, J != nullptr);
  if (!on || lds == 0) return false;
  static bool attr_set = false;
  if (!attr_set) {
    HYP_CHECK(hipFuncSetAttribute((const void*)jacobi_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    attr_set = true;
  }
  static const bool dbg = [] { const char* e = getenv("HYP_JACOBI_DBG"); return e && e[0] == '1'; }();   // sweeps of every call on stderr
  int* sw = dbg ? reinterpret_cast<int*>(ctx.dscal.d() + 63) : nullptr;
  const double tol_rot = values_only ? 1e-10 : 1e-15, tol_big = values_only ? 1e-6 : 1e-8, floor_rel = values_only ? 2.220446049250313e-16 : 0.0;
  hipLaunchKernelGGL(jacobi_lds_kernel, dim3(1), dim3(1024), lds, ctx.stream, len, m, V, J, 60, load_j ? 1 : 0, sw, tol_rot, tol_big, floor_rel);
  HYP_CHECK(hipGetLastError());
  if (dbg) {
    ctx.d2h(ctx.h_info + 32, sw, sizeof(int));
    ctx.sync();
    fprintf(stderr, "[jacobi] %d x %d%s%s: %d sweeps\n", len, m, J ? " +J" : "", load_j ? " warm" : "", ctx.h_info[32]);
  }
  return true;
}

// column i of B (len x m): sigma_i = its norm, V1[:, i] = B[:, i] / sigma_i; counts exact zeros into flag
__global__ __launch_bounds__(256) void svd_finish_kernel(int len, const double* __restrict__ B, double* __restrict__ V1, double* __restrict__ sig,
                                                         int* __restrict__ flag) {
  __shared__ double red[256];
  const int i = blockIdx.x;
  const double* b = B + (long)i * len;
  double s = 0.0;
  for (int r = threadIdx.x; r < len; r += 256) s += b[r] * b[r];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  const double nr = sqrt(red[0]);
  if (threadIdx.x == 0) {
    sig[i] = nr;
    if (!(nr > 0.0)) atomicAdd(flag, 1);
  }
  const double inv = (nr > 0.0) ? 1.0 / nr : 0.0;
  for (int r = threadIdx.x; r < len; r += 256) V1[(long)i * len + r] = b[r] * inv;
}

// the d1 x d1 block of the closed-form inverse (one workgroup): R1 = U' R V1, r_u -> A1, a
//   pairs i != j : A1_ij = z_i z_j / 2 * (u^2 R1_ij - s_i s_j R1_ji) / (u^4 - s_i^2 s_j^2)
//   arrow        : c_i = 4 u s_i / z_i^2, d_i = 2 (u^2 + s_i^2) / z_i^2;
//                  a = (r_u + sum c_i R1_ii / d_i) / S,  S = Huu - sum c_i^2 / d_i = sum 2 / (u^2 + s_i^2) - (d1 - 1) / u^2
//                  (evaluated in the second, subtraction-free form); A1_ii = (R1_ii + c_i a) / d_i
__global__ __launch_bounds__(256) void ens_closed_block_kernel(int d1, double u, double Huu, const double* __restrict__ sig, const double* __restrict__ R1,
                                                               const double* __restrict__ ru, double* __restrict__ A1, double* __restrict__ a_out) {
  __shared__ double red[2][256];
  __shared__ double a_sh;
  const double u2 = u * u;
  double s0 = 0.0, s1 = 0.0;
  for (int i = threadIdx.x; i < d1; i += 256) {
    const double si = sig[i], zi = u2 - si * si;
    const double ci = 4.0 * u * si / (zi * zi), di = 2.0 * (u2 + si * si) / (zi * zi);
    s0 += ci * R1[(long)i * d1 + i] / di;
    // the arrow's Schur complement Huu - sum c_i^2 / d_i WITHOUT the subtraction: with Huu = sum_i d_i - (d1 - 1) / u^2 (the
    // barrier is -sum log(u^2 - s_i^2) + (d1 - 1) log u in these coordinates) and d_i^2 - c_i^2 = 4 / z_i^2, each term is
    // d_i - c_i^2 / d_i = 2 / (u^2 + s_i^2).  Formed as written it is a difference of two numbers of size 1 / z^2 whose
    // value is of size 1 / u^2: near the boundary every digit cancels (for d1 = 1 there is nothing else in the sum -- the
    // 1 x 1 and 1 x 2 cones of tests/test_hip_solver.py::test_edge_case_models_hip ended in SlowProgress / NumericalFailure).
    s1 += 2.0 / (u2 + si * si);
  }
  red[0][threadIdx.x] = s0; red[1][threadIdx.x] = s1;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) { red[0][threadIdx.x] += red[0][threadIdx.x + off]; red[1][threadIdx.x] += red[1][threadIdx.x + off]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    (void)Huu;
    a_sh = (ru[0] + red[0][0]) / (red[1][0] - (double)(d1 - 1) / u2);
    a_out[0] = a_sh;
  }
  __syncthreads();
  const double a = a_sh;
  for (long e = threadIdx.x; e < (long)d1 * d1; e += 256) {
    const int i = (int)(e % d1), j = (int)(e / d1);
    const double si = sig[i], sj = sig[j], zi = u2 - si * si, zj = u2 - sj * sj;
    double v;
    if (i == j) {
      const double ci = 4.0 * u * si / (zi * zi), di = 2.0 * (u2 + si * si) / (zi * zi);
      v = (R1[e] + ci * a) / di;
    } else {
      const double ss = si * sj;
      v = 0.5 * zi * zj * (u2 * R1[(long)j * d1 + i] - ss * R1[(long)i * d1 + j]) / (u2 * u2 - ss * ss);
    }
    A1[e] = v;
  }
}
// T (d1 x d2) <- z_i / 2 * (Rt - P)_ij : the part of the rotated right-hand side orthogonal to V1
__global__ void ens_closed_perp_kernel(int d1, int d2, double u, const double* __restrict__ sig, const double* __restrict__ Rt, const double* __restrict__ P,
                                       double* __restrict__ T) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)d1 * d2) return;
  const int i = (int)(e % d1);
  const double si = sig[i];
  T[e] = 0.5 * (u * u - si * si) * (Rt[e] - P[e]);
}
__global__ void ens_identity_bases_kernel(int d1, int d2, double* __restrict__ U, double* __restrict__ V1) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < (long)d1 * d1) U[e] = ((e % d1) == (e / d1)) ? 1.0 : 0.0;
  if (e < (long)d2 * d1) V1[e] = ((e % d2) == (e / d2)) ? 1.0 : 0.0;
}

static void mm(Ctx& c, bool transa, int M, int N, int K, const double* A, long lda, const double* B, long ldb, double* C, long ldc,
               double alpha, double beta, int batch = 1, long sA = 0, long sB = 0, long sC = 0) {
  GemmArgs g{};
  g.M = M; g.N = N; g.K = K; g.A = A; g.lda = lda; g.strideA = sA; g.B = B; g.ldb = ldb; g.strideB = sB; g.C = C; g.ldc = ldc; g.strideC = sC;
  g.alpha = alpha; g.beta = beta; g.batch = batch;
  gemm(c, transa, g);
}
static double read_scalar(Ctx& ctx, const double* d) {
  ctx.d2h(ctx.h_pinned, d, sizeof(double));
  ctx.sync();
  return ctx.h_pinned[0];
}
static int read_info(Ctx& ctx, const int* d_info) {
  ctx.d2h(ctx.h_info, d_info, sizeof(int));
  ctx.sync();
  return ctx.h_info[0];
}

EpiNormSpectralCone::EpiNormSpectralCone(Ctx& c, int d1_, int d2_, bool use_dual) : GenericHessCone(c, CONE_EPINORMSPECTRAL) {
  HYP_REQUIRE(1 <= d1_ && d1_ <= d2_, "EpiNormSpectral: 1 <= d1 <= d2");
  d1 = d1_; d2 = d2_;
  dim = 1 + d1 * d2;
  nu = d1 + 1;   // :97
  use_dual_barrier = use_dual;
  alloc_common();
  alloc_generic();
  const size_t b12 = (size_t)d1 * d2 * 8, b11 = (size_t)d1 * d1 * 8, b22 = (size_t)d2 * d2 * 8;
  W.alloc(b12); WT.alloc(b12); tau.alloc(b12); HuW.alloc(b12); Zitau.alloc(b12);
  t12a.alloc(b12); t12b.alloc(b12); t12c.alloc(b12); t12d.alloc(b12);
  Z.alloc(b11); Zfact.alloc(b11); Zi.alloc(b11); t11.alloc(b11);
  WtauI.alloc(b22); t22.alloc(b22); t22b.alloc(b22);
  Zdinv.alloc(dinv_elems(d1) * 8);
  Zinfo.alloc(64);
}

void EpiNormSpectralCone::set_initial_point(double* h) {   // :99-105
  for (int i = 0; i < dim; ++i) h[i] = 0.0;
  h[0] = sqrt((double)(d1 + 1));
}

void EpiNormSpectralCone::zsolve(double* X, long ldx, int nrhs) {   // ldiv!(fact_Z, X)
  trsm_work.ensure((size_t)NB * std::max(nrhs, 1) * 8);
  trsm_upper_left(ctx, d1, nrhs, Zfact.d(), d1, Zdinv.d(), true, X, ldx, trsm_work.d());
  trsm_upper_left(ctx, d1, nrhs, Zfact.d(), d1, Zdinv.d(), false, X, ldx, trsm_work.d());
}

bool EpiNormSpectralCone::update_feas() {   // :107-123
  u = read_scalar(ctx, point.d());
  if (u > EPS) {
    ctx.d2d(W.p, point.d() + 1, (size_t)d1 * d2 * 8);
    dev_transpose(ctx, d1, d2, W.d(), d1, WT.d(), d2, 1, 0, 0);
    hipLaunchKernelGGL(scaled_identity_kernel, dim3((d1 + 127) / 128, d1), dim3(128), 0, ctx.stream, d1, Z.d(), (long)d1, u * u);
    mm(ctx, false, d1, d1, d2, W.d(), d1, WT.d(), d2, Z.d(), d1, -1.0, 1.0);   // Z = u^2 I - W W'
    ctx.d2d(Zfact.p, Z.p, (size_t)d1 * d1 * 8);
    potrf_upper_batched(ctx, d1, Zfact.d(), d1, 0, 1, Zdinv.d(), Zinfo.i());
    is_feas_ = (read_info(ctx, Zinfo.i()) == 0);
    if (is_feas_) dev_zero_strict_lower(ctx, d1, Zfact.d(), d1, 1, 0);
  } else {
    is_feas_ = false;
  }
  feas_updated = true;
  return is_feas_;
}

double EpiNormSpectralCone::nuclear_norm(const double* d_mat) {
  nuclear_norm_launch(d_mat, ctx.dscal.d());
  return read_scalar(ctx, ctx.dscal.d());
}

// the device work of the nuclear norm on ctx.stream, the result left in *d_out (no host round trip where the one-launch
// decomposition applies)
void EpiNormSpectralCone::nuclear_norm_launch(const double* d_mat, double* d_out) {
  // rows of the d1 x d2 matrix = columns of its transpose V (d2 x d1): orthogonalise them pairwise
  double* V = t12a.d();
  dev_transpose(ctx, d1, d2, d_mat, d1, V, d2, 1, 0, 0);
  const int m = d1, mpad = (m % 2 == 0) ? m : m + 1;
  // warm start as in update_svd: the dual points of consecutive trials are close, the rotations of the last call (Jdual) nearly
  // orthogonalise the new matrix already.  Only where the one-launch LDS kernel applies (it carries the rotation product).
  static const bool warm_on = [] { const char* e = getenv("HYP_SVD_WARM"); return !(e && e[0] == '0'); }();
  bool done = false;
  if (warm_on && m > 1 && jacobi_lds_bytes(d2, m, true) != 0) {
    Jdual.ensure((size_t)d1 * d1 * 8);
    double* V2 = t12b.d();
    const bool warm = dual_prev_ok && dual_warm_count < 16;
    if (warm) {
      mm(ctx, false, d2, d1, d1, V, d2, Jdual.d(), d1, V2, d2, 1.0, 0.0);
      ++dual_warm_count;
    } else {
      dual_warm_count = 0;
    }
    done = jacobi_in_lds(ctx, d2, m, warm ? V2 : V, Jdual.d(), warm, true);
    if (done) {
      dual_prev_ok = true;
      if (warm) V = V2;
    }
  }
  if (m > 1 && !done && !jacobi_in_lds(ctx, d2, m, V, nullptr, false, true)) {
    for (int sweep = 0; sweep < 40; ++sweep) {
      ctx.zero(Zinfo.p, sizeof(int));
      for (int t = 0; t < mpad - 1; ++t)
        hipLaunchKernelGGL(jacobi_round_kernel, dim3(mpad / 2), dim3(256), 0, ctx.stream, d2, m, mpad, t, V, (long)d2, Zinfo.i());
      if (read_info(ctx, Zinfo.i()) == 0) break;
    }
  }
  // singular values = column norms; their sum in one launch
  hipLaunchKernelGGL(colnorm_sum_kernel, dim3(1), dim3(256), 0, ctx.stream, d2, m, V, tmpd.d(), d_out);
  HYP_CHECK(hipGetLastError());
}

bool EpiNormSpectralCone::is_dual_feas() {   // :125-132
  if (dual_cached) return dual_feas_;
  const double ud = read_scalar(ctx, dual_point.d());
  if (ud > EPS) return (ud - nuclear_norm(dual_point.d() + 1)) > EPS;
  return false;
}

// A primal-feasible line-search candidate needs two independent decompositions, each latency-bound on ONE workgroup: the dual
// point's for its nuclear norm (is_dual_feas; 0.4 - 1.5 ms at 50 x 100) and the primal point's behind the closed-form inverse
// Hessian, which the proximity test asks for (0.2 - 0.3 ms).  After the primal feasibility test (the reference's order: an
// infeasible candidate -- the first steps of the schedule usually are -- costs one small Cholesky and nothing else) the dual
// chain goes to the helper stream and the primal decomposition runs on the main stream underneath it; one synchronisation
// joins them.  (The primal decomposition is started on the assumption that the dual point is feasible; if it is not, it ran
// underneath the dual chain anyway.)
void EpiNormSpectralCone::prefetch_feas() {
  static const bool on = [] { const char* e = getenv("HYP_ENS_PREFETCH"); return !(e && e[0] == '0'); }();
  if (!on || dual_cached) return;
  if (!(feas_updated ? is_feas_ : update_feas())) return;   // (early_reject may have run the primal test already)
  hipEvent_t e0 = ctx.aux_event(2);
  HYP_CHECK(hipEventRecord(e0, ctx.stream));                 // (point / dual_point were loaded on the main stream)
  HYP_CHECK(hipStreamWaitEvent(ctx.stream2, e0, 0));
  {
    StreamSwap on_helper(ctx);
    double* res = ctx.dscal.d() + 40;
    nuclear_norm_launch(dual_point.d() + 1, res);
    ctx.d2h(ctx.h_pinned + 40, res, sizeof(double));
    ctx.d2h(ctx.h_pinned + 41, dual_point.d(), sizeof(double));
  }
  static const bool cf = [] { const char* e = getenv("HYP_ENS_CLOSED_INV"); return !(e && e[0] == '0'); }();
  if (cf) update_svd();
  hipEvent_t e1 = ctx.aux_event(3);
  HYP_CHECK(hipEventRecord(e1, ctx.stream2));
  HYP_CHECK(hipStreamWaitEvent(ctx.stream, e1, 0));
  ctx.sync();
  const double nn = ctx.h_pinned[40], ud = ctx.h_pinned[41];
  dual_feas_ = (ud > EPS) && ((ud - nn) > EPS);
  dual_cached = true;
}

// Before either decomposition is started for a candidate: the primal feasibility test (one small Cholesky), and if it passes a
// lower bound of the proximity value by Cauchy-Schwarz, <v, H^-1 v> >= <v, w>^2 / <w, H w>, with w = the closed-form inverse of
// the LAST decomposed point applied to v -- singular vectors and values of the iterate the search started from, or of the
// previous candidate, still sit in Usvd / sig / V1 -- and <w, H w> from the closed-form Hessian product at THIS point (it needs
// the Cholesky of the feasibility test, no decomposition).  A candidate whose bound exceeds the neighbourhood is rejected
// whatever its dual feasibility: neither the dual point's decomposition (0.4 - 1.5 ms at 50 x 100) nor the primal one is run.
bool EpiNormSpectralCone::early_reject(double irtmu, double bound2) {
  static const bool on = [] { const char* e = getenv("HYP_PROX_LB"); return !(e && e[0] == '0'); }();
  static const bool cf = [] { const char* e = getenv("HYP_ENS_CLOSED_INV"); return !(e && e[0] == '0'); }();
  if (!(feas_updated ? is_feas_ : update_feas())) return true;      // search.jl:120-124: not in the cone
  if (!on || !cf || !svd_prev_ok || svd_updated || hess_fact_updated || d1 < 2) return false;
  const size_t vb = (size_t)dim * sizeof(double);
  const double* g = get_grad();
  ctx.d2d(vec1.p, g, vb);
  dev_axpby(ctx, dim, irtmu, dual_point.d(), 1.0, vec1.d());          // v
  closed_inv_apply(u_svd, vec2.d(), dim, vec1.d(), dim, 1);            // w
  prox_out.ensure(vb);
  hess_prod(prox_out.d(), dim, vec2.d(), dim, 1);                      // H w at this point
  double* ds = ctx.dscal.d() + 44;
  dev_dot(ctx, dim, vec1.d(), vec2.d(), ds);
  dev_dot(ctx, dim, vec2.d(), prox_out.d(), ds + 1);
  ctx.d2h(ctx.h_pinned + 44, ds, 2 * sizeof(double));
  ctx.sync();
  const double a = ctx.h_pinned[44], b = ctx.h_pinned[45];
  if (!(b > 0.0) || !(a == a) || !(b < INFINITY)) return false;
  return a * a / b > bound2 * (1.0 + 1e-9);
}

void EpiNormSpectralCone::update_grad() {   // :134-150
  ctx.kstat[7] += 1;
  ctx.d2d(tau.p, W.p, (size_t)d1 * d2 * 8);
  zsolve(tau.d(), d1, d2);                                              // tau = Z^-1 W
  // Zi = Z^-1 = U^-1 U^-T
  trtri_upper_batched(ctx, d1, Zfact.d(), d1, 0, Zdinv.d(), 0, t11.d(), d1, 0, 1);
  dev_transpose(ctx, d1, d1, t11.d(), d1, Z.d(), d1, 1, 0, 0);         // Z buffer reused for U^-T (Z itself is no longer needed)
  GemmArgs g{};
  g.M = d1; g.N = d1; g.K = d1; g.A = t11.d(); g.lda = d1; g.B = Z.d(); g.ldb = d1; g.C = Zi.d(); g.ldc = d1;
  g.alpha = 1; g.beta = 0; g.krange = KR_GE_M; g.batch = 1;
  gemm(ctx, false, g);
  hipLaunchKernelGGL(trace_kernel, dim3(1), dim3(256), 0, ctx.stream, d1, Zi.d(), (long)d1, ctx.dscal.d());
  const double trZi = read_scalar(ctx, ctx.dscal.d());
  g0_host = (-u * trZi) * 2.0 + (d1 - 1) / u;
  dev_scale_copy(ctx, d1 * d2, 2.0, tau.d(), grad.d() + 1);
  ctx.h2d(grad.p, &g0_host, sizeof(double));
  ctx.sync();
  grad_updated = true;
}

void EpiNormSpectralCone::update_hess_aux() {   // :152-170
  get_grad();
  ctx.d2d(Zitau.p, tau.p, (size_t)d1 * d2 * 8);
  zsolve(Zitau.d(), d1, d2);
  dev_scale_copy(ctx, d1 * d2, -4.0 * u, Zitau.d(), HuW.d());
  trZi2 = dot_host(d1 * d1, Zi.d(), Zi.d());
  Huu = 4.0 * u * u * trZi2 + (g0_host - 2.0 * (d1 - 1) / u) / u;
  hipLaunchKernelGGL(scaled_identity_kernel, dim3((d2 + 127) / 128, d2), dim3(128), 0, ctx.stream, d2, WtauI.d(), (long)d2, 1.0);
  mm(ctx, true, d2, d2, d1, W.d(), d1, tau.d(), d1, WtauI.d(), d2, 1.0, 1.0);   // I + W' tau
  hess_aux_updated = true;
}

void EpiNormSpectralCone::update_hess() {   // :172-209
  ensure_hess_storage(false);
  if (!hess_aux_updated) update_hess_aux();
  hipLaunchKernelGGL(ens_hess_kernel, dim3((dim + 127) / 128, dim), dim3(128), 0, ctx.stream, d1, d2, Zi.d(), tau.d(), WtauI.d(), HuW.d(), Huu,
                     H.d(), (long)dim);
  HYP_CHECK(hipGetLastError());
  dev_symmetrize_from_upper(ctx, dim, H.d(), dim, 1, 0);
  hess_updated = true;
}

void EpiNormSpectralCone::hess_prod(double* prod, long ldp, const double* arr, long lda, int ncols) {   // :211-239
  if (!hess_aux_updated) update_hess_aux();
  if (ncols <= 0) return;
  const int dw = d1 * d2;
  const long cap = 1L << 25;   // doubles per workspace
  int chunk = (int)std::min<long>(std::min(ncols, 32768), std::max<long>(1, cap / std::max((long)d1 * d1, (long)dw)));
  wsA.ensure((size_t)chunk * d1 * d1 * 8);
  wsB.ensure((size_t)chunk * d1 * d1 * 8);
  wsC.ensure((size_t)chunk * dw * 8);
  for (int c0 = 0; c0 < ncols; c0 += chunk) {
    const int nc = std::min(chunk, ncols - c0);
    const double* a = arr + (long)c0 * lda;
    double* p = prod + (long)c0 * ldp;
    hipLaunchKernelGGL(hp_first_row_kernel, dim3(nc), dim3(256), 0, ctx.stream, dw, Huu, HuW.d(), a, lda, p, ldp);
    // T_j = A_W_j W'
    mm(ctx, false, d1, d1, d2, a + 1, d1, WT.d(), d2, wsA.d(), d1, 1.0, 0.0, nc, lda, 0, (long)d1 * d1);
    hipLaunchKernelGGL(symm_shift_kernel, dim3((d1 + 63) / 64, d1, nc), dim3(64), 0, ctx.stream, d1, wsA.d(), wsB.d(), a, lda, u);
    // R_j = 2 S_j tau + 2 A_W_j
    HYP_CHECK(hipMemcpy2DAsync(wsC.p, (size_t)dw * 8, a + 1, (size_t)lda * 8, (size_t)dw * 8, nc, hipMemcpyDeviceToDevice, ctx.stream));
    mm(ctx, true, d1, d2, d1, wsB.d(), d1, tau.d(), d1, wsC.d(), d1, 2.0, 2.0, nc, (long)d1 * d1, 0, (long)dw);
    zsolve(wsC.d(), d1, nc * d2);
    HYP_CHECK(hipMemcpy2DAsync(p + 1, (size_t)ldp * 8, wsC.p, (size_t)dw * 8, (size_t)dw * 8, nc, hipMemcpyDeviceToDevice, ctx.stream));
  }
  HYP_CHECK(hipGetLastError());
}

const double* EpiNormSpectralCone::dder3(const double* d_dir) {   // :241-294
  if (!hess_aux_updated) update_hess_aux();
  const int dw = d1 * d2;
  const double u_dir = read_scalar(ctx, d_dir);
  const double* Wd = d_dir + 1;                       // W_dir (d1 x d2)
  double* d22b = t22b.d(); double* d22 = t22.d();
  double* b12 = t12b.d(); double* c12 = t12c.d(); double* dd12 = t12d.d(); double* a12 = t12a.d();
  double* d11 = t11.d();
  mm(ctx, true, d2, d2, d1, Wd, d1, tau.d(), d1, d22b, d2, 1.0, 0.0);          // d22b = W_dir' tau
  ctx.d2d(dd12, Wd, (size_t)dw * 8);
  zsolve(dd12, d1, d2);                                                         // dd = Z^-1 W_dir
  mm(ctx, false, d1, d2, d2, dd12, d1, WtauI.d(), d2, b12, d1, 1.0, 0.0);      // b = dd WtauI
  dev_transpose(ctx, d2, d2, d22b, d2, d22, d2, 1, 0, 0);                       // d22 <- d22b' (temporary)
  mm(ctx, false, d1, d2, d2, dd12, d1, d22, d2, c12, d1, 1.0, 0.0);            // c = dd d22b'
  mm(ctx, false, d1, d1, d2, dd12, d1, WT.d(), d2, d11, d1, 1.0, 0.0);         // d11 = dd W'
  mm(ctx, false, d2, d2, d2, d22b, d2, d22b, d2, d22, d2, 1.0, 0.0);           // d22 = d22b d22b
  mm(ctx, true, d2, d2, d1, Wd, d1, b12, d1, d22, d2, 1.0, 1.0);               // d22 += W_dir' b
  mm(ctx, false, d1, d2, d2, tau.d(), d1, d22, d2, dd12, d1, 1.0, 0.0);        // dd = tau d22
  mm(ctx, false, d1, d2, d2, c12, d1, WtauI.d(), d2, dd12, d1, 1.0, 1.0);      // dd += c WtauI
  mm(ctx, false, d1, d2, d2, b12, d1, d22b, d2, dd12, d1, 1.0, 1.0);           // dd += b d22b
  zsolve(b12, d1, d2);                                                          // b = Z^-1 b
  mm(ctx, false, d1, d2, d2, Zitau.d(), d1, d22b, d2, b12, d1, 1.0, 1.0);      // b += Zitau d22b
  dev_transpose(ctx, d1, d2, Wd, d1, a12, d2, 1, 0, 0);                         // a12 = W_dir' (d2 x d1)
  mm(ctx, false, d1, d1, d2, tau.d(), d1, a12, d2, d11, d1, 1.0, 1.0);         // d11 += tau W_dir'
  mm(ctx, false, d1, d2, d1, d11, d1, Zitau.d(), d1, b12, d1, 1.0, 1.0);       // b += d11 Zitau
  dev_axpby(ctx, dw, 0.0, b12, -2.0 * u, b12);                                  // b *= -2u
  const double const1 = 4.0 * u * u_dir * u;
  dev_scale_copy(ctx, dw, const1, Zitau.d(), c12);
  dev_axpby(ctx, dw, -u_dir, tau.d(), 1.0, c12);                                // c = const1 Zitau - u_dir tau
  zsolve(c12, d1, d2);
  dev_axpby(ctx, dw, 1.0, c12, 1.0, b12);                                       // b += c
  dev_axpby(ctx, dw, -2.0 * u_dir, b12, -2.0, dd12);                            // dd = -2 u_dir b - 2 dd
  ctx.d2d(dder3v.d() + 1, dd12, (size_t)dw * 8);
  // trZi3 = || L^-1 Zi ||_F^2, Z = L L' with L = U'
  ctx.d2d(d11, Zi.p, (size_t)d1 * d1 * 8);
  trsm_work.ensure((size_t)NB * d1 * 8);
  trsm_upper_left(ctx, d1, d1, Zfact.d(), d1, Zdinv.d(), true, d11, d1, trsm_work.d());
  const double trZi3 = dot_host(d1 * d1, d11, d11);
  dev_axpby(ctx, dw, 3.0, c12, 1.0, b12);                                       // b += 3 c
  const double wb = dot_host(dw, Wd, b12);
  const double r = u_dir / u;
  const double d0 = -wb - u * u_dir * (6.0 * trZi2 - 8.0 * u * trZi3 * u) * u_dir - (d1 - 1) * r * r / u;
  ctx.h2d(dder3v.p, &d0, sizeof(double));
  ctx.sync();
  return dder3v.d();
}

// ---------------------------------------------------------------------------------------------
// closed-form inverse Hessian (see cones.hpp).  Derivation, in the coordinates At = U' A [V1 V2], W = U S V1', z_i = u^2 - s_i^2,
// from hess_prod! (epinormspectral.jl:211-239: T = A W' + W A' - 2 u a I, out_W = 2 Z^-1 (T Z^-1 W + A), out_u = Huu a + <HuW, A>):
//   V2 part       out_ij = 2 At_ij / z_i
//   i != j < d1   [out_ij; out_ji] = 2 / (z_i z_j) [[u^2, s_i s_j], [s_i s_j, u^2]] [At_ij; At_ji]
//   diagonal, u   out_ii = d_i At_ii - c_i a,  out_u = Huu a - sum_i c_i At_ii,  c_i = 4 u s_i / z_i^2, d_i = 2 (u^2 + s_i^2) / z_i^2
// each of which inverts in closed form (the 2 x 2 determinant u^4 - s_i^2 s_j^2 and the arrow's Schur complement are positive in
// the interior of the cone).  U, s, V1 come from a one-sided Jacobi SVD of W' with accumulated rotations.
// ---------------------------------------------------------------------------------------------
bool EpiNormSpectralCone::update_svd() {
  if (svd_updated) return svd_ok;
  const size_t b11 = (size_t)d1 * d1 * 8, b21 = (size_t)d2 * d1 * 8;
  Usvd.ensure(b11); Jm.ensure(b11); V1.ensure(b21); V1T.ensure(b21); Bj.ensure(b21); sig.ensure((size_t)d1 * 8);
  // Warm start: W moves little between line-search trials and iterations, so W' J_prev (J_prev = the rotations accumulated by
  // the last decomposition) has nearly orthogonal columns already and two or three sweeps finish the job where a cold start
  // needs eight to ten (the decomposition is ~1 ms at 50 x 100, 40 % of config 3b's kernel time).  Any orthogonal start gives a
  // valid decomposition; every 16th one starts cold so that rounding in the accumulated J cannot pile up.
  static const bool warm_on = [] { const char* e = getenv("HYP_SVD_WARM"); return !(e && e[0] == '0'); }();
  bool warm = false;
  if (warm_on && svd_prev_ok && svd_warm_count < 16 && d1 > 1) {
    mm(ctx, false, d2, d1, d1, WT.d(), d2, Usvd.d(), d1, Bj.d(), d2, 1.0, 0.0);
    ctx.d2d(Jm.p, Usvd.p, b11);
    ++svd_warm_count;
    warm = true;
  } else {
    ctx.d2d(Bj.p, WT.p, b21);                                   // W' (d2 x d1): its columns are the rows of W
    dev_fill_identity(ctx, d1, Jm.d(), d1);
    svd_warm_count = 0;
  }
  const int m = d1, mm2 = (m % 2 == 0) ? m : m + 1;
  if (m > 1 && !jacobi_in_lds(ctx, d2, m, Bj.d(), Jm.d(), warm)) {
    for (int sweep = 0; sweep < 60; ++sweep) {
      ctx.zero(Zinfo.p, sizeof(int));
      for (int t = 0; t < mm2 - 1; ++t)
        hipLaunchKernelGGL(jacobi_round_vec_kernel, dim3(mm2 / 2), dim3(256), 0, ctx.stream, d2, m, mm2, t, Bj.d(), (long)d2, Jm.d(), Zinfo.i());
      if (read_info(ctx, Zinfo.i()) == 0) break;
    }
  }
  ctx.zero(Zinfo.p, sizeof(int));
  hipLaunchKernelGGL(svd_finish_kernel, dim3(d1), dim3(256), 0, ctx.stream, d2, Bj.d(), V1.d(), sig.d(), Zinfo.i());
  const int nzero = read_info(ctx, Zinfo.i());
  ctx.d2d(Usvd.p, Jm.p, b11);
  u_svd = u;
  svd_ok = true;
  svd_prev_ok = (nzero == 0);
  if (nzero == d1) {   // W = 0 (the initial point): any orthonormal bases do
    const long tot = std::max((long)d1 * d1, (long)d2 * d1);
    hipLaunchKernelGGL(ens_identity_bases_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx.stream, d1, d2, Usvd.d(), V1.d());
  } else if (nzero > 0) {
    svd_ok = false;    // some, not all, singular values vanish exactly: the generic path handles the point
  }
  dev_transpose(ctx, d2, d1, V1.d(), d2, V1T.d(), d1, 1, 0, 0);
  svd_updated = true;
  return svd_ok;
}

bool EpiNormSpectralCone::inv_hess_ready() {
  static const bool on = [] { const char* e = getenv("HYP_ENS_CLOSED_INV"); return !(e && e[0] == '0'); }();
  closed_inv = on;
  if (closed_inv && update_svd()) return true;
  return GenericHessCone::inv_hess_ready();
}

void EpiNormSpectralCone::inv_hess_prod(double* prod, long ldp, const double* arr, long lda, int ncols) {
  static const bool on = [] { const char* e = getenv("HYP_ENS_CLOSED_INV"); return !(e && e[0] == '0'); }();
  closed_inv = on;
  if (!closed_inv || hess_fact_updated || !update_svd()) {   // (a factorization that exists already is used: same operator)
    GenericHessCone::inv_hess_prod(prod, ldp, arr, lda, ncols);
    return;
  }
  if (!hess_aux_updated) update_hess_aux();
  closed_inv_apply(u, prod, ldp, arr, lda, ncols);
}

// the closed-form inverse with the decomposition currently held (Usvd, sig, V1 -- of THIS point after update_svd, of an earlier
// point before it: prox_lower_bound) and the epigraph variable u_used that belongs to it
void EpiNormSpectralCone::closed_inv_apply(double u_used, double* prod, long ldp, const double* arr, long lda, int ncols) {
  const double u = u_used;
  const int dw = d1 * d2;
  const size_t b11 = (size_t)d1 * d1 * 8, b12 = (size_t)dw * 8;
  cw1.ensure(b12); cw2.ensure(b11); cw3.ensure(b12); cw4.ensure(b11); cw5.ensure(b12);
  for (int j = 0; j < ncols; ++j) {
    const double* a = arr + (long)j * lda;
    double* p = prod + (long)j * ldp;
    double* Rt = cw1.d();      // U' R              (d1 x d2)
    double* R1 = cw2.d();      // Rt V1             (d1 x d1)
    double* P = cw3.d();       // R1 V1'            (d1 x d2)
    double* A1 = cw4.d();      // closed-form block (d1 x d1)
    double* T = cw5.d();       // z / 2 * (Rt - P), then + A1 V1' = At
    mm(ctx, true, d1, d2, d1, Usvd.d(), d1, a + 1, d1, Rt, d1, 1.0, 0.0);
    mm(ctx, false, d1, d1, d2, Rt, d1, V1.d(), d2, R1, d1, 1.0, 0.0);
    mm(ctx, false, d1, d2, d1, R1, d1, V1T.d(), d1, P, d1, 1.0, 0.0);
    hipLaunchKernelGGL(ens_closed_block_kernel, dim3(1), dim3(256), 0, ctx.stream, d1, u, Huu, sig.d(), R1, a, A1, p);
    hipLaunchKernelGGL(ens_closed_perp_kernel, dim3((unsigned)((dw + 255) / 256)), dim3(256), 0, ctx.stream, d1, d2, u, sig.d(), Rt, P, T);
    mm(ctx, false, d1, d2, d1, A1, d1, V1T.d(), d1, T, d1, 1.0, 1.0);
    mm(ctx, false, d1, d2, d1, Usvd.d(), d1, T, d1, p + 1, d1, 1.0, 0.0);
  }
  HYP_CHECK(hipGetLastError());
}

}  // namespace hyp
