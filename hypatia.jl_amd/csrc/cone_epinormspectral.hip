// EpiNormSpectral on the device: (u, W), u >= sigma_1(W), W is d1 x d2 (column-stacked), d1 <= d2.
// Reference: /root/reference/src/Cones/epinormspectral.jl (real case; line ranges inline).
// Dual feasibility needs the nuclear norm of a d1 x d2 matrix (svdvals!, :125-132): computed by a
// one-sided Jacobi (Hestenes) iteration on the rows, one workgroup per row pair per round.
#include "cones.hpp"
#include "tds_small.hpp"

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
// total over a group of GL = 32 / 16 lanes, in every lane of the group (16: the four DPP steps stay inside a row of 16 lanes)
template <int GL>
__device__ __forceinline__ double jl_sum(double v) {
  v += jl_dpp<0xB1>(v);
  v += jl_dpp<0x4E>(v);
  v += jl_dpp<0x141>(v);
  v += jl_dpp<0x140>(v);
  return GL == 32 ? v + __shfl_xor(v, 16, 32) : v;
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
//
// decide_u (values-only calls of the dual feasibility test, epinormspectral.jl:125-132: u - sum(svdvals(W)) > eps): the caller
// only wants to know on which side of *decide_u the nuclear norm lies.  In front of every sweep the state B = [b_1 ... b_m]
// (nuclear norm invariant under the rotations) gives two rigorous bounds from its column norms d_i and cosines c_pq:
//   ||B||_* <= sum_i d_i                          (B = sum_i b_i e_i', a sum of rank-one matrices of nuclear norm d_i)
//   ||B||_* >= ||B_S||_* >= <B_S, Q> / ||Q||_2 = sum_{i in S} d_i / sqrt(||I + C_S||_2) >= sum_{i in S} d_i / sqrt(1 + ||C_S||_F)
// for any subset S of the columns, Q = their normalised columns (S = the columns above 1e-8 of the largest: noise columns have
// arbitrary cosines).  Once both bounds are on one side of u by more than 1e-10 relatively the sweeps stop: the sum of the column
// norms the caller then forms decides as the converged value would (it is the upper bound; where the LOWER bound says "outside",
// so does it).  Closer than that to the boundary the sweeps run to the end as without the test.
// GL lanes per column pair, 32 pairs side by side (GL * 32 threads).  GL = 16 (HYP_JACOBI_GL=16: a wavefront carries four pairs
// through the rotation's scalar chain instead of two, 7 wavefronts instead of 13 for config 3b's 25 pairs) measured the same time
// per sweep as 32: a round is bound by its chain of dependent FP64 operations, not by instruction issue (EXPERIMENTS r05-22).
template <int GL>
__global__ __launch_bounds__(GL * 32) void jacobi_lds_kernel(int len, int m, double* __restrict__ Vg, double* __restrict__ Jg, int max_sweeps, int load_j,
                                                          int* __restrict__ sweeps_out, double tol_rot, double tol_big, double floor_rel,
                                                          const double* __restrict__ decide_u, double decide_eps) {
  extern __shared__ __attribute__((aligned(16))) double jl_lds[];
  const int ldv = len | 1;                       // odd stride
  double* V = jl_lds;                            // m columns of length len
  double* J = jl_lds + (long)m * ldv;            // m columns of length m (only if Jg)
  const int ldj = m | 1;
  __shared__ int rotated;
  const int tid = threadIdx.x;
  constexpr int NT = GL * 32, NR = 128 / GL;   // threads; entries per lane of a column held in registers (len <= 128)
  for (long e = tid; e < (long)len * m; e += NT) V[(e / len) * ldv + (e % len)] = Vg[e];
  if (Jg) for (long e = tid; e < (long)m * m; e += NT) J[(e / m) * ldj + (e % m)] = load_j ? Jg[e] : (((e / m) == (e % m)) ? 1.0 : 0.0);   // (load_j: warm start, the rotations continue an earlier product)
  __syncthreads();
  double floor2 = 0.0;
  __shared__ double fr[1024];   // (the static part has to stay small: 150 KB of the CU's 160 are the caller's to ask for dynamically)
  if (floor_rel > 0.0) {   // ||B||_F^2 (invariant under the rotations), fixed summation order
    double f = 0.0;
    for (long e = tid; e < (long)len * m; e += NT) { const double x = V[(e / len) * ldv + (e % len)]; f = fma(x, x, f); }
    fr[tid] = f;
    __syncthreads();
    for (int off = NT / 2; off > 0; off >>= 1) {
      if (tid < off) fr[tid] += fr[tid + off];
      __syncthreads();
    }
    floor2 = floor_rel * floor_rel * fr[0];
    __syncthreads();   // (fr is written again by the bounds pass)
  }
  const int mm = (m % 2 == 0) ? m : m + 1;
  const int sub = tid & (GL - 1), grp = tid / GL;   // 32 groups of GL lanes
  double* dc_n2 = fr;            // [256] (fr is free again once floor2 is formed)
  double* dc_part = fr + 256;    // [32]
  __shared__ int dc_done;
  const bool decide = (decide_u != nullptr) && m <= 256;
  const double u_dec = decide ? decide_u[0] : 0.0;
  if (sweeps_out && tid == 0) *sweeps_out = 0;
  for (int sweep = 0; sweep < max_sweeps && m > 1; ++sweep) {
    if (decide) {
      for (int i = grp; i < m; i += 32) {                       // squared column norms of the current state
        const double* vi = V + (long)i * ldv;
        double a = 0.0;
        for (int r = sub; r < len; r += GL) { const double x = vi[r]; a = fma(x, x, a); }
        a = jl_sum<GL>(a);
        if (sub == 0) dc_n2[i] = a;
      }
      __syncthreads();
      double amax = 0.0;
      for (int i = 0; i < m; ++i) amax = fmax(amax, dc_n2[i]);
      const double thr = 1e-16 * amax;                          // (squared norms: columns below 1e-8 of the largest stay out of S)
      double c2 = 0.0;                                          // this group's share of sum_{p < q in S} c_pq^2 (read-only pass: no barrier)
      for (int t = 0; t < mm - 1; ++t)
        for (int i = grp; i < mm / 2; i += 32) {
          int p, q;
          if (i == 0) { p = mm - 1; q = t; }
          else { p = (t + i) % (mm - 1); q = (t - i + (mm - 1)) % (mm - 1); }
          if (p >= m || q >= m) continue;
          const double ap = dc_n2[p], aq = dc_n2[q];
          if (!(ap > thr) || !(aq > thr)) continue;
          const double* vp = V + (long)p * ldv;
          const double* vq = V + (long)q * ldv;
          double g = 0.0;
          for (int r = sub; r < len; r += GL) g = fma(vp[r], vq[r], g);
          g = jl_sum<GL>(g);
          c2 += (g * g) / (ap * aq);
        }
      if (sub == 0) dc_part[grp] = c2;
      __syncthreads();
      if (tid == 0) {
        double s_up = 0.0, s_sub = 0.0, cf2 = 0.0;
        for (int i = 0; i < m; ++i) { const double d = sqrt(dc_n2[i]); s_up += d; if (dc_n2[i] > thr) s_sub += d; }
        for (int g2i = 0; g2i < 32; ++g2i) cf2 += dc_part[g2i];
        const double s_low = s_sub / sqrt(1.0 + sqrt(2.0 * cf2));
        const bool inside = (u_dec - s_up * (1.0 + 1e-10)) > decide_eps;
        const bool outside = !((u_dec - s_low * (1.0 - 1e-10)) > decide_eps);
        dc_done = (inside || outside) && (s_up == s_up) && (s_low == s_low) ? 1 : 0;
      }
      __syncthreads();
      if (dc_done) break;
    }
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
        double xr[NR], yr[NR];
        if (in_regs) {
#pragma unroll
          for (int k = 0; k < NR; ++k) {
            const int r = sub + GL * k;
            xr[k] = (r < len) ? vp[r] : 0.0;
            yr[k] = (r < len) ? vq[r] : 0.0;
          }
#pragma unroll
          for (int k = 0; k < NR; ++k) { a = fma(xr[k], xr[k], a); b = fma(yr[k], yr[k], b); g = fma(xr[k], yr[k], g); }
        } else {
          for (int r = sub; r < len; r += GL) {
            const double x = vp[r], y = vq[r];
            a = fma(x, x, a); b = fma(y, y, b); g = fma(x, y, g);
          }
        }
        a = jl_sum<GL>(a); b = jl_sum<GL>(b); g = jl_sum<GL>(g);
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
          for (int k = 0; k < NR; ++k) {
            const int r = sub + GL * k;
            if (r < len) {
              vp[r] = cs * xr[k] - sn * yr[k];
              vq[r] = sn * xr[k] + cs * yr[k];
            }
          }
        } else {
          for (int r = sub; r < len; r += GL) {
            const double x = vp[r], y = vq[r];
            vp[r] = cs * x - sn * y;
            vq[r] = sn * x + cs * y;
          }
        }
        if (Jg) {
          double* jp = J + (long)p * ldj;
          double* jq = J + (long)q * ldj;
          for (int r = sub; r < m; r += GL) {
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
  for (long e = tid; e < (long)len * m; e += NT) Vg[e] = V[(e / len) * ldv + (e % len)];
  if (Jg) for (long e = tid; e < (long)m * m; e += NT) Jg[e] = J[(e / m) * ldj + (e % m)];
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
static bool jacobi_in_lds(Ctx& ctx, int len, int m, double* V, double* J, bool load_j = false, bool values_only = false,
                          const double* d_decide_u = nullptr) {
  static const bool on = [] { const char* e = getenv("HYP_JACOBI_LDS"); return !(e && e[0] == '0'); }();
  const size_t lds = jacobi_lds_bytes(len, m, J != nullptr);
  if (!on || lds == 0) return false;
  static bool attr_set = false;
  if (!attr_set) {
    HYP_CHECK(hipFuncSetAttribute((const void*)jacobi_lds_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    HYP_CHECK(hipFuncSetAttribute((const void*)jacobi_lds_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    attr_set = true;
  }
  static const bool dbg = [] { const char* e = getenv("HYP_JACOBI_DBG"); return e && e[0] == '1'; }();   // sweeps of every call on stderr
  int* sw = dbg ? reinterpret_cast<int*>(ctx.dscal.d() + 63) : nullptr;
  const double tol_rot = values_only ? 1e-10 : 1e-15, tol_big = values_only ? 1e-6 : 1e-8, floor_rel = values_only ? 2.220446049250313e-16 : 0.0;
  static const bool decide_on = [] { const char* e = getenv("HYP_ENS_DUAL_DECIDE"); return !(e && e[0] == '0'); }();
  static const int gl_env = [] { const char* e = getenv("HYP_JACOBI_GL"); return e ? atoi(e) : 32; }();   // 16 / 32 lanes per column pair
  const int gl = (gl_env == 16) ? 16 : 32;   // (16 measured equal at 50 x 100, EXPERIMENTS r05-22: the round is bound by its dependent chain, not by issue)
  if (gl == 16)
    hipLaunchKernelGGL(jacobi_lds_kernel<16>, dim3(1), dim3(512), lds, ctx.stream, len, m, V, J, 60, load_j ? 1 : 0, sw, tol_rot, tol_big, floor_rel,
                       (decide_on && values_only) ? d_decide_u : nullptr, EPS);
  else
    hipLaunchKernelGGL(jacobi_lds_kernel<32>, dim3(1), dim3(1024), lds, ctx.stream, len, m, V, J, 60, load_j ? 1 : 0, sw, tol_rot, tol_big, floor_rel,
                       (decide_on && values_only) ? d_decide_u : nullptr, EPS);
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
// (body for any block size >= 256: the sums are those of 256 strided partial sums and their tree whatever the block size)
__device__ __forceinline__ void ens_closed_block_body(int d1, double u, const double* __restrict__ sig, const double* R1, const double* __restrict__ ru,
                                                      double* A1, double* __restrict__ a_out, double (*red)[256], double* a_sh) {
  const int tid = threadIdx.x;
  const double u2 = u * u;
  double s0 = 0.0, s1 = 0.0;
  if (tid < 256) {
    for (int i = tid; i < d1; i += 256) {
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
    red[0][tid] = s0; red[1][tid] = s1;
  }
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (tid < off) { red[0][tid] += red[0][tid + off]; red[1][tid] += red[1][tid + off]; }
    __syncthreads();
  }
  if (tid == 0) {
    *a_sh = (ru[0] + red[0][0]) / (red[1][0] - (double)(d1 - 1) / u2);
    a_out[0] = *a_sh;
  }
  __syncthreads();
  const double a = *a_sh;
  for (long e = tid; e < (long)d1 * d1; e += blockDim.x) {
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
__global__ __launch_bounds__(256) void ens_closed_block_kernel(int d1, double u, double Huu, const double* __restrict__ sig, const double* __restrict__ R1,
                                                               const double* __restrict__ ru, double* __restrict__ A1, double* __restrict__ a_out) {
  __shared__ double red[2][256];
  __shared__ double a_sh;
  (void)Huu;
  ens_closed_block_body(d1, u, sig, R1, ru, A1, a_out, red, &a_sh);
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

// ---------------------------------------------------------------------------------------------
// One-workgroup forms of the oracles for cones whose matrices fit one CU's LDS (d1 <= 64: config 3b's 50 x 100).  At that size
// every kernel of the chains above is a 5-20 us launch of one or two workgroups, and an oracle is 8-25 of them in a row (the
// feasibility test 8, the gradient with the Hessian's auxiliary matrices 22, a Hessian product 8, the closed-form inverse 7):
// ~900 launches and 7.4 ms per iteration at 0.012 of the MFMA peak.  Here an oracle is ONE launch of 512 threads: the small
// products on the matrix cores from LDS / L2 operands (wg_mm: one 16 x 16 tile per wavefront at a time, k ascending in steps of 4
// into one accumulator -- the order gemm_f64_kernel sums in), Z^-1 (.) by the register-resident wavefront program of
// tds_small.hpp on 16 columns per wavefront (the bits of zsolve), Z's Cholesky factor and its inverse in LDS.
// HYP_ENS_FUSED=0 restores the launch chains.  epinormspectral.jl:107-123 (feas), :134-170 (grad, hess aux), :211-239 (hess_prod).
// ---------------------------------------------------------------------------------------------
constexpr int EF_THREADS = 512;

// fs(m, n, sum_k fa(m, k) fb(k, n)) for the outputs m < M, n0 <= n < N; operands outside M / N / K read as zero.  The operand
// entries of 16 k-steps (K <= 64: all of them) are requested before the first MFMA: a tile costs one memory round trip per 64 of K,
// not one per unrolled group (operands in L2: 11 us per product of config 3b's sizes with four steps in flight, the launch
// chain's GEMM takes 8-11).
template <class FA, class FB, class FS>
__device__ __forceinline__ void wg_mm(int M, int N, int K, FA fa, FB fb, FS fs, int n0 = 0) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int tm = (M + 15) >> 4, tn = (N - n0 + 15) >> 4;
  for (int t = wave; t < tm * tn; t += nw) {
    const int mt = t % tm, nt = t / tm;
    const int m = mt * 16 + li, n = n0 + nt * 16 + li;
    const bool mok = m < M, nok = n < N;
    d4_t acc = (d4_t){0.0, 0.0, 0.0, 0.0};
    for (int kb = 0; kb < K; kb += 64) {
      double av[16], bv[16];
#pragma unroll
      for (int st = 0; st < 16; ++st) {
        const int k = kb + 4 * st + lk;
        av[st] = (mok && k < K) ? fa(m, k) : 0.0;
        bv[st] = (nok && k < K) ? fb(k, n) : 0.0;
      }
#pragma unroll
      for (int st = 0; st < 16; ++st)
        if (kb + 4 * st < K) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[st], bv[st], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int mo = mt * 16 + 4 * r + lk, no = n0 + nt * 16 + li;
      if (mo < M && no < N) fs(mo, no, acc[r]);
    }
  }
}

// Z^-1 applied NAPPLY times in a row to ncol columns (zsolve, then zsolve of the result): load(row, col) gives the right-hand
// side, store(stage, row, col, value) receives the result of application `stage`.  16 columns per wavefront in registers, the
// operand entries of the factor and its inverse staged in `ops` (TDS_LDS_DOUBLES of LDS) for all wavefronts.  Called by every
// thread of the workgroup (barriers inside).
template <int NAPPLY, class FL, class FS>
__device__ __forceinline__ void wg_zsolve(int d1, int ncol, const double* __restrict__ U, long ldu, const double* __restrict__ dinv, int refine, double* ops,
                                          FL load, FS store, int col0 = 0) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int q = lane >> 4, nn = lane & 15;
  for (int g0 = col0; g0 < ncol; g0 += nw * 16) {            // (uniform trip count; columns col0 .. ncol - 1)
    const int c0 = g0 + wave * 16;
    const bool active = c0 < ncol;
    const bool inb = c0 + nn < ncol;
    const int col = min(c0 + nn, ncol - 1);
    d4_t y[4], x[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * t + q + 4 * r;
        y[t][r] = (row < d1 && inb) ? load(min(row, d1 - 1), col) : 0.0;
      }
#pragma unroll
    for (int stage = 0; stage < NAPPLY; ++stage) {
      __syncthreads();
      tds_stage_ops<true>(U, ldu, dinv, d1, ops);
      __syncthreads();
      if (active) tds_apply_lds<true>(ops, d1, refine, y, x);     // forward: U'^-1
      __syncthreads();
      tds_stage_ops<false>(U, ldu, dinv, d1, ops);
      __syncthreads();
      if (active) tds_apply_lds<false>(ops, d1, refine, x, y);    // backward: U^-1
      if (inb) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * t + q + 4 * r;
            if (row < d1) store(stage, row, col, y[t][r]);
          }
      }
    }
  }
}

// sum over the workgroup of 256 strided partial sums (threads >= 256 pass 0) by the tree trace_kernel / hp_first_row_kernel use
__device__ __forceinline__ double wg_tree256(double v, double* red /* [256] */) {
  const int tid = threadIdx.x;
  __syncthreads();
  if (tid < 256) red[tid] = v;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (tid < off) red[tid] += red[tid + off];
    __syncthreads();
  }
  return red[0];
}

// update_feas (:107-123) in one launch: W, W' copied out of the point, Z = u^2 I - W W', its upper Cholesky factor in LDS
// (right-looking, two barriers per column), the factor's inverse by back substitution (16 lanes per column), both written in the
// layouts of potrf_upper_batched.  rec[0] = u, rec[1] = 0 / the 1-based index of the first non-positive pivot / -1 (u <= eps).
__global__ __launch_bounds__(EF_THREADS) void ens_feas_fused_kernel(int d1, int d2, const double* __restrict__ point, double* __restrict__ W,
                                                                    double* __restrict__ WT, double* __restrict__ Zfact, double* __restrict__ Zdinv,
                                                                    double* __restrict__ rec) {
  extern __shared__ __attribute__((aligned(16))) double ef_lds[];
  double* Zs = ef_lds;                     // d1 x d1, ld d1
  volatile double* Ds = ef_lds + (long)d1 * d1;
  const int tid = threadIdx.x;
  const double u = point[0];
  const double* Wp = point + 1;
  for (int e = tid; e < d1 * d2; e += EF_THREADS) {
    const double w = Wp[e];
    W[e] = w;
    WT[(e / d1) + (long)(e % d1) * d2] = w;
  }
  if (!(u > EPS)) {
    if (tid == 0) { rec[0] = u; rec[1] = -1.0; }
    return;
  }
  const double u2 = u * u;
  wg_mm(d1, d1, d2, [&](int m, int k) { return Wp[m + (long)k * d1]; }, [&](int k, int n) { return Wp[n + (long)k * d1]; },
        [&](int m, int n, double acc) {
          double v = -1.0 * acc;
          v += (m == n) ? u2 : 0.0;
          Zs[m + n * d1] = v;
        });
  __syncthreads();
  int fail = 0;
  for (int k = 0; k < d1; ++k) {
    const double piv = Zs[k + k * d1];
    if (!(piv > 0.0)) { fail = k + 1; break; }                // (uniform: every thread reads the same word)
    const double s = sqrt(piv);
    __syncthreads();                                           // (everyone has read the pivot before it is replaced)
    for (int j = k + tid; j < d1; j += EF_THREADS) Zs[k + j * d1] = (j == k) ? s : Zs[k + j * d1] / s;
    __syncthreads();
    const int w = d1 - k - 1;
    for (int idx = tid; idx < w * w; idx += EF_THREADS) {
      const int i = k + 1 + idx % w, j = k + 1 + idx / w;
      if (i <= j) Zs[i + j * d1] = fma(-Zs[k + i * d1], Zs[k + j * d1], Zs[i + j * d1]);
    }
    __syncthreads();
  }
  if (fail) {
    if (tid == 0) { rec[0] = u; rec[1] = (double)fail; }
    return;
  }
  // D = U^-1 (upper), column j by back substitution: d_jj = 1 / u_jj, d_ij = -(sum_{i < k <= j} u_ik d_kj) / u_ii
  for (int e = tid; e < d1 * d1; e += EF_THREADS) Ds[e] = 0.0;
  __syncthreads();
  {
    const int g = tid >> 4, l = tid & 15;
    for (int j = g; j < d1; j += EF_THREADS / 16) {
      if (l == 0) Ds[j + j * d1] = 1.0 / Zs[j + j * d1];
      for (int i = j - 1; i >= 0; --i) {
        double sacc = 0.0;
        for (int k = i + 1 + l; k <= j; k += 16) sacc = fma(Zs[i + k * d1], Ds[k + j * d1], sacc);
        sacc += __shfl_xor(sacc, 8, 16);
        sacc += __shfl_xor(sacc, 4, 16);
        sacc += __shfl_xor(sacc, 2, 16);
        sacc += __shfl_xor(sacc, 1, 16);
        if (l == 0) Ds[i + j * d1] = -sacc / Zs[i + i * d1];
      }
    }
  }
  __syncthreads();
  for (int e = tid; e < d1 * d1; e += EF_THREADS) {
    const int i = e % d1, j = e / d1;
    Zfact[e] = (i <= j) ? Zs[e] : 0.0;
    const double dij = Ds[e];
    Zdinv[i + (long)j * NB] = dij;                            // inv(U)
    Zdinv[(long)NB * NB + j + (long)i * NB] = dij;            // its transpose
  }
  if (tid == 0) { rec[0] = u; rec[1] = 0.0; }
}

// update_grad + update_hess_aux (:134-170) in one launch: tau = Z^-1 W, Zitau = Z^-1 tau, grad, HuW = -4 u Zitau, Zi = U^-1 U^-T,
// WtauI = I + W' tau; rec[2..5] = tr(Zi), <Zi, Zi>, g0, Huu
__global__ __launch_bounds__(EF_THREADS) void ens_grad_aux_fused_kernel(int d1, int d2, double u, const double* __restrict__ W,
                                                                        const double* __restrict__ U, const double* __restrict__ dinv, int refine,
                                                                        double* __restrict__ tau, double* __restrict__ Zi, double* __restrict__ grad,
                                                                        double* __restrict__ Zitau, double* __restrict__ HuW, double* __restrict__ WtauI,
                                                                        double* __restrict__ rec) {
  extern __shared__ __attribute__((aligned(16))) double ef_lds[];
  __shared__ double red[256];
  double* TauS = ef_lds;                   // d1 x d2
  double* ZiS = ef_lds + (long)d1 * d2;    // d1 x d1
  double* ops = ZiS + (long)d1 * d1;       // TDS_LDS_DOUBLES
  const int tid = threadIdx.x;
  const double c4u = -4.0 * u;
  // blockIdx.x = a share of the d2 columns (at most four 16-column groups: one per SIMD in the two solves): tau, Zitau, the gradient,
  // HuW and the columns of W' tau in WtauI depend on their own columns of W only; the first share forms Zi and the scalars as well
  const int groups = (d2 + 15) >> 4, gper = (groups + (int)gridDim.x - 1) / (int)gridDim.x;
  const int col0 = min(d2, (int)blockIdx.x * gper * 16), col1 = min(d2, ((int)blockIdx.x + 1) * gper * 16);
  if (col0 >= col1 && blockIdx.x > 0) return;
  wg_zsolve<2>(d1, col1, U, d1, dinv, refine, ops, [&](int row, int col) { return W[row + (long)col * d1]; },
               [&](int stage, int row, int col, double v) {
                 const long e = row + (long)col * d1;
                 if (stage == 0) { tau[e] = v; TauS[e] = v; grad[1 + e] = 2.0 * v; }
                 else { Zitau[e] = v; HuW[e] = c4u * v; }
               }, col0);
  __syncthreads();
  wg_mm(d2, col1, d1, [&](int m, int k) { return W[k + (long)m * d1]; }, [&](int k, int n) { return TauS[k + n * d1]; },
        [&](int m, int n, double acc) {
          double v = acc;
          v += (m == n) ? 1.0 : 0.0;
          WtauI[m + (long)n * d2] = v;
        }, col0);
  if (blockIdx.x > 0) return;
  wg_mm(d1, d1, d1, [&](int m, int k) { return (k >= m) ? dinv[m + (long)k * NB] : 0.0; },
        [&](int k, int n) { return (k >= n) ? dinv[n + (long)k * NB] : 0.0; },
        [&](int m, int n, double acc) { ZiS[m + n * d1] = acc; Zi[m + (long)n * d1] = acc; });
  __syncthreads();
  double tr = 0.0;
  if (tid < 256) for (int i = tid; i < d1; i += 256) tr += ZiS[i + i * d1];
  const double trZi = wg_tree256(tr, red);
  double s2 = 0.0;
  if (tid < 256) for (int e = tid; e < d1 * d1; e += 256) s2 = fma(ZiS[e], ZiS[e], s2);
  const double trZi2 = wg_tree256(s2, red);
  if (tid == 0) {
    // g0 = (-u trZi) 2 + (d1 - 1) / u ; Huu = 4 u u trZi2 + (g0 - 2 (d1 - 1) / u) / u   (as the host forms them, no contraction)
    const double dm1 = (double)(d1 - 1);
    const double g0 = __dadd_rn(__dmul_rn(__dmul_rn(-u, trZi), 2.0), __ddiv_rn(dm1, u));
    const double huu = __dadd_rn(__dmul_rn(__dmul_rn(__dmul_rn(4.0, u), u), trZi2),
                                 __ddiv_rn(__dadd_rn(g0, -__ddiv_rn(__dmul_rn(2.0, dm1), u)), u));
    grad[0] = g0;
    rec[2] = trZi; rec[3] = trZi2; rec[4] = g0; rec[5] = huu;
  }
}

// hess_prod (:211-239): out_u = Huu a_u + <HuW, A>; T = A W', S = T + T' - 2 u a_u I, out_W = Z^-1 (2 S tau + 2 A).
// blockIdx.x = column; blockIdx.y = a share of the column's d2 matrix columns: the solve with Z is the long part (370 MFMAs of 64
// cycles per 16 columns: 20 us for 100 columns on one CU's four matrix cores), so it is dealt out to gridDim.y workgroups of at
// most four 16-column groups each; every share forms T and S for itself (2 x 16 tiles) and its own columns of the right-hand side.
// A, W' and tau go through LDS (one coalesced pass each).
__global__ __launch_bounds__(EF_THREADS) void ens_hess_prod_fused_kernel(int d1, int d2, double u, double Huu, const double* __restrict__ HuW,
                                                                         const double* __restrict__ WT, const double* __restrict__ tau,
                                                                         const double* __restrict__ U, const double* __restrict__ dinv, int refine,
                                                                         const double* __restrict__ arr, long lda, double* __restrict__ prod, long ldp) {
  extern __shared__ __attribute__((aligned(16))) double ef_lds[];
  __shared__ double red[256];
  double* Ts = ef_lds;                          // d1 x d1
  double* Ss = Ts + (long)d1 * d1;              // d1 x d1
  double* Rs = Ss + (long)d1 * d1;              // d1 x d2: A, then the right-hand side in its place
  double* X = Rs + (long)d1 * d2;               // max(d1 d2, TDS_LDS_DOUBLES): W', then tau, then the staged operands of the solve
  const int tid = threadIdx.x, dw = d1 * d2;
  const int groups = (d2 + 15) >> 4, gper = (groups + (int)gridDim.y - 1) / (int)gridDim.y;
  const int col0 = min(d2, (int)blockIdx.y * gper * 16), col1 = min(d2, ((int)blockIdx.y + 1) * gper * 16);
  if (col0 >= col1 && blockIdx.y > 0) return;
  const double* a = arr + (long)blockIdx.x * lda;
  double* p = prod + (long)blockIdx.x * ldp;
  const double a0 = a[0];
  const double* A = a + 1;
  for (int e = tid; e < dw; e += EF_THREADS) { Rs[e] = A[e]; X[e] = WT[e]; }
  if (blockIdx.y == 0) {
    double s = 0.0;
    if (tid < 256) for (int i = tid; i < dw; i += 256) s += HuW[i] * A[i];
    const double hw = wg_tree256(s, red);
    if (tid == 0) p[0] = Huu * a0 + hw;
  }
  __syncthreads();
  wg_mm(d1, d1, d2, [&](int m, int k) { return Rs[m + k * d1]; }, [&](int k, int n) { return X[k + n * d2]; },
        [&](int m, int n, double acc) { Ts[m + n * d1] = acc; });
  __syncthreads();
  for (int e = tid; e < d1 * d1; e += EF_THREADS) {
    const int r = e % d1, c = e / d1;
    double v = Ts[c * d1 + r] + Ts[r * d1 + c];
    if (r == c) v -= 2.0 * u * a0;
    Ss[c * d1 + r] = v;
  }
  for (int e = col0 * d1 + tid; e < col1 * d1; e += EF_THREADS) X[e] = tau[e];
  __syncthreads();
  wg_mm(d1, col1, d1, [&](int m, int k) { return Ss[k + m * d1]; }, [&](int k, int n) { return X[k + n * d1]; },
        [&](int m, int n, double acc) {
          double v = 2.0 * acc;
          v += 2.0 * Rs[m + n * d1];
          Rs[m + n * d1] = v;
        }, col0);
  __syncthreads();
  wg_zsolve<1>(d1, col1, U, d1, dinv, refine, X, [&](int row, int col) { return Rs[row + col * d1]; },
               [&](int, int row, int col, double v) { p[1 + row + (long)col * d1] = v; }, col0);
}

// closed_inv_apply, one workgroup per column (see the derivation at update_svd): Rt = U' R, R1 = Rt V1, T = z / 2 (Rt - R1 V1') +
// A1 V1', out_W = U T
__global__ __launch_bounds__(EF_THREADS) void ens_closed_inv_fused_kernel(int d1, int d2, double u, const double* __restrict__ Usvd,
                                                                          const double* __restrict__ V1, const double* __restrict__ V1T,
                                                                          const double* __restrict__ sig, const double* __restrict__ arr, long lda,
                                                                          double* __restrict__ prod, long ldp) {
  extern __shared__ __attribute__((aligned(16))) double ef_lds[];
  __shared__ double red[2][256];
  __shared__ double a_sh;
  double* Rt = ef_lds;                          // d1 x d2
  double* T = Rt + (long)d1 * d2;               // d1 x d2
  double* R1 = T + (long)d1 * d2;               // d1 x d1
  double* A1 = R1 + (long)d1 * d1;              // d1 x d1
  const double* a = arr + (long)blockIdx.x * lda;
  double* p = prod + (long)blockIdx.x * ldp;
  const double* A = a + 1;
  wg_mm(d1, d2, d1, [&](int m, int k) { return Usvd[k + (long)m * d1]; }, [&](int k, int n) { return A[k + (long)n * d1]; },
        [&](int m, int n, double acc) { Rt[m + n * d1] = acc; });
  __syncthreads();
  wg_mm(d1, d1, d2, [&](int m, int k) { return Rt[m + k * d1]; }, [&](int k, int n) { return V1[k + (long)n * d2]; },
        [&](int m, int n, double acc) { R1[m + n * d1] = acc; });
  __syncthreads();
  ens_closed_block_body(d1, u, sig, R1, a, A1, p, red, &a_sh);
  // T = z_i / 2 (Rt - R1 V1')
  wg_mm(d1, d2, d1, [&](int m, int k) { return R1[m + k * d1]; }, [&](int k, int n) { return V1T[k + (long)n * d1]; },
        [&](int m, int n, double acc) {
          const double si = sig[m];
          T[m + n * d1] = 0.5 * (u * u - si * si) * (Rt[m + n * d1] - acc);
        });
  __syncthreads();
  // T += A1 V1'   (in place: every entry is read and written by the one lane that owns it)
  wg_mm(d1, d2, d1, [&](int m, int k) { return A1[m + k * d1]; }, [&](int k, int n) { return V1T[k + (long)n * d1]; },
        [&](int m, int n, double acc) {
          double v = acc;
          v += T[m + n * d1];
          T[m + n * d1] = v;
        });
  __syncthreads();
  wg_mm(d1, d2, d1, [&](int m, int k) { return Usvd[m + (long)k * d1]; }, [&](int k, int n) { return T[k + n * d1]; },
        [&](int m, int n, double acc) { p[1 + m + (long)n * d1] = acc; });
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

// the one-workgroup kernels apply where the cone's matrices fit: d1 <= 64 (the wavefront program of tds_small.hpp) and the
// largest LDS layout (closed-form inverse: two d1 x d2 and two d1 x d1 matrices) within 150 KB
static size_t ens_fused_lds_bytes(int d1, int d2, int which) {
  const size_t dd = (size_t)d1 * d1, dw = (size_t)d1 * d2;
  switch (which) {
    case 0: return 2 * dd * 8;                                   // feas: Z, D
    case 1: return (dw + dd + TDS_LDS_DOUBLES) * 8;              // grad + aux: tau, Zi, staged operands
    case 2: return (2 * dd + dw + std::max(dw, (size_t)TDS_LDS_DOUBLES)) * 8;   // hess_prod: T, S, A / R, W' / tau / staged operands
    default: return (2 * dw + 2 * dd) * 8;                       // closed inverse: Rt, T, R1, A1
  }
}
bool EpiNormSpectralCone::fused() {
  if (fused_checked) return fused_ok;
  static const bool on = [] { const char* e = getenv("HYP_ENS_FUSED"); return !(e && e[0] == '0'); }();
  size_t mx = 0;
  for (int w = 0; w < 4; ++w) mx = std::max(mx, ens_fused_lds_bytes(d1, d2, w));
  fused_ok = on && d1 <= 64 && mx <= 150 * 1024;
  if (fused_ok) {
    static bool attr_set = false;
    if (!attr_set) {
      HYP_CHECK(hipFuncSetAttribute((const void*)ens_feas_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
      HYP_CHECK(hipFuncSetAttribute((const void*)ens_grad_aux_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
      HYP_CHECK(hipFuncSetAttribute((const void*)ens_hess_prod_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
      HYP_CHECK(hipFuncSetAttribute((const void*)ens_closed_inv_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
      attr_set = true;
    }
    frec.alloc(8 * sizeof(double));
  }
  fused_checked = true;
  return fused_ok;
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
  if (fused()) {
    hipLaunchKernelGGL(ens_feas_fused_kernel, dim3(1), dim3(EF_THREADS), ens_fused_lds_bytes(d1, d2, 0), ctx.stream, d1, d2, point.d(), W.d(), WT.d(),
                       Zfact.d(), Zdinv.d(), frec.d());
    HYP_CHECK(hipGetLastError());
    ctx.d2h(ctx.h_pinned + 56, frec.d(), 2 * sizeof(double));
    ctx.sync();
    u = ctx.h_pinned[56];
    is_feas_ = (u > EPS) && (ctx.h_pinned[57] == 0.0);
    feas_updated = true;
    return is_feas_;
  }
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

double EpiNormSpectralCone::nuclear_norm(const double* d_mat, const double* d_decide_u) {
  nuclear_norm_launch(d_mat, ctx.dscal.d(), d_decide_u);
  return read_scalar(ctx, ctx.dscal.d());
}

// the device work of the nuclear norm on ctx.stream, the result left in *d_out (no host round trip where the one-launch
// decomposition applies)
void EpiNormSpectralCone::nuclear_norm_launch(const double* d_mat, double* d_out, const double* d_decide_u) {
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
    done = jacobi_in_lds(ctx, d2, m, warm ? V2 : V, Jdual.d(), warm, true, d_decide_u);
    if (done) {
      dual_prev_ok = true;
      if (warm) V = V2;
    }
  }
  if (m > 1 && !done && !jacobi_in_lds(ctx, d2, m, V, nullptr, false, true, d_decide_u)) {
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
  if (ud > EPS) return (ud - nuclear_norm(dual_point.d() + 1, dual_point.d())) > EPS;
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
    nuclear_norm_launch(dual_point.d() + 1, res, dual_point.d());
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
  if (fused()) {   // gradient and the Hessian's auxiliary matrices together (update_hess_aux then finds them)
    const int shares = std::max(1, ((d2 + 15) / 16 + 3) / 4);
    hipLaunchKernelGGL(ens_grad_aux_fused_kernel, dim3(shares), dim3(EF_THREADS), ens_fused_lds_bytes(d1, d2, 1), ctx.stream, d1, d2, u, W.d(), Zfact.d(),
                       Zdinv.d(), trsm_refine_steps(), tau.d(), Zi.d(), grad.d(), Zitau.d(), HuW.d(), WtauI.d(), frec.d());
    HYP_CHECK(hipGetLastError());
    ctx.d2h(ctx.h_pinned + 58, frec.d() + 2, 4 * sizeof(double));
    ctx.sync();
    trZi2 = ctx.h_pinned[59];
    g0_host = ctx.h_pinned[60];
    Huu = ctx.h_pinned[61];
    grad_updated = true;
    hess_aux_updated = true;
    return;
  }
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
  if (hess_aux_updated) return;   // (the one-workgroup gradient kernel forms the auxiliary matrices as well)
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
  if (fused() && ncols <= 4096) {   // (a few columns: directions, residuals, bounds; the explicit-Hessian callers come with thousands)
    const int shares = std::max(1, ((d2 + 15) / 16 + 3) / 4);   // at most four 16-column groups of the solve per workgroup
    hipLaunchKernelGGL(ens_hess_prod_fused_kernel, dim3(ncols, shares), dim3(EF_THREADS), ens_fused_lds_bytes(d1, d2, 2), ctx.stream, d1, d2, u, Huu, HuW.d(),
                       WT.d(), tau.d(), Zfact.d(), Zdinv.d(), trsm_refine_steps(), arr, lda, prod, ldp);
    HYP_CHECK(hipGetLastError());
    return;
  }
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
  if (fused() && ncols > 0 && ncols <= 4096) {
    hipLaunchKernelGGL(ens_closed_inv_fused_kernel, dim3(ncols), dim3(EF_THREADS), ens_fused_lds_bytes(d1, d2, 3), ctx.stream, d1, d2, u, Usvd.d(), V1.d(),
                       V1T.d(), sig.d(), arr, lda, prod, ldp);
    HYP_CHECK(hipGetLastError());
    return;
  }
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
