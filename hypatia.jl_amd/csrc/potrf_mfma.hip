// Critical-path kernels of the blocked Cholesky (dense.hip: potrf_upper_batched), second generation: the 128 x 128
// diagonal block factor and the 128 x m panel solve, both cut into 16 x 16 MFMA tiles.
//   dpotrf 'U' of the reference: /root/reference/src/linearalgebra/dense.jl:189-200 (posdef_fact!), called from
//   src/Solvers/systemsolvers/qrchol.jl:249-250 and src/Cones/possemideftri.jl:85,94.
//
// Why: the first generation (potrf_diag.hip) advanced one COLUMN per workgroup barrier (128 barriers + LDS round trips per
// diagonal block, 43 us) and one column per DPP step in the panel (26 us); 39 such pairs in a row are 2.7 of the 3.8 ms the
// n = 5000 factorization took (profiles/r01_cholesky_timeline.txt).  Here a block advances 16 rows per step:
//   * the 16 x 16 diagonal tile is factored inside ONE wavefront (column per lane, pivot row broadcast with v_readlane:
//     no barrier, no LDS round trip per column),
//   * the row panel right of it is solved by forward SUBSTITUTION against that tile (dtrsm's rounding; no inverse),
//     column per lane with the tile's rows broadcast from LDS,
//   * everything below is a rank-16 update on v_mfma_f64_16x16x4_f64, accumulators resident in registers for the whole
//     kernel (the block never round-trips through memory between steps).
// Tile register layout = the MFMA C/D layout: lane (q = lane >> 4, n = lane & 15), register r holds row q + 4 r of column n.
// With the panel tile of block step jb used as the MFMA's second operand, its register r IS k-chunk r of the operand (rows
// 4 r .. 4 r + 3 over q), so solved tiles feed the rank-16 update with no data movement; the first operand (the tile of U
// left of the diagonal, transposed by the MFMA's own operand convention) comes from LDS, tiles stored [k][m] with a row
// stride of 17 doubles (conflict-free for the column writes and for the operand reads).
// Deterministic: fixed tile ownership and summation order.
#include "hyp_internal.hpp"
#include <utility>

namespace hyp {

#ifdef HYP_PROBE   // tools/probe_potrf.hip: s_memtime stamps of the phases (lane 0 of wavefront 0 of workgroup 0)
__device__ long long g_stamps[64];
#define STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) g_stamps[i] = clock64(); } while (0)
__device__ long long g_wstamps[8][16];
#define WSTAMP(cond, i) do { if ((cond) && (threadIdx.x & 63) == 0 && blockIdx.x == 0) g_wstamps[threadIdx.x >> 6][i] = clock64(); } while (0)
#else
#define STAMP(i) do { } while (0)
#define WSTAMP(cond, i) do { } while (0)
#endif

namespace {

// two doubles stored together where the address is only known to be 8-byte aligned (lda and k0 are arbitrary)
struct __attribute__((packed, aligned(8))) d2_t { double a, b; };
constexpr int TS = 17;          // row stride of a 16 x 16 tile in LDS
constexpr int TL = 16 * TS;     // doubles per tile in LDS

// workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every outstanding GLOBAL store of the
// wavefront (s_waitcnt vmcnt(0)), which would put the write-back of finished tiles on the critical path of every step
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ d4_t mfma4(double a, double b, d4_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }

// value of lane L of each 16-lane row (DPP row_newbcast): the four rows of a wavefront hold identical copies wherever this is used
template <int L>
__device__ __forceinline__ double bcast16(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x150 + L, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x150 + L, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

// The same as ONE instruction: gfx90a+ executes 64-bit DPP with row_newbcast (v_mov_b64_dpp), and v_fmac_f64 takes the DPP operand
// directly -- acc += (value of lane L of the row's src) * m.  Round 4: the elimination step of tile_potrf was 8 vector instructions
// per updated entry (two zero-initialised 32-bit DPP moves, a copy, the sign flip, the FMA) and issue-bound inside the kernels
// (10 K cycles per tile where the pivot-to-pivot chain is 3 K); it is one v_fmac_f64_dpp now.  (s_nop: a DPP operand written by
// the preceding VALU instruction needs two wait states, and the hazard recogniser does not see inside inline assembly.)
template <int L>
__device__ __forceinline__ double mov_bcast16(double src) {
  double d;
  asm("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(d) : "v"(src), "n"(L));
  return d;
}
template <int L>
__device__ __forceinline__ void fmac_bcast16(double& acc, double src, double m) {
  asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(m), "n"(L));
}

// C/D layout -> every lane holds all 16 rows of its column (the four lanes of a column end up with identical copies)
__device__ __forceinline__ void tile_gather(const d4_t& c, int n, double (&x)[16]) {
#pragma unroll
  for (int qq = 0; qq < 4; ++qq)
#pragma unroll
    for (int r = 0; r < 4; ++r) x[4 * r + qq] = __shfl(c[r], qq * 16 + n, 64);
}
__device__ __forceinline__ void tile_scatter(const double (&x)[16], int q, d4_t& c) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    // (three selects; the empty asm keeps the compiler from turning them into a scratch array indexed by q)
    double v = x[4 * r];
    v = (q == 1) ? x[4 * r + 1] : v;
    asm volatile("" : "+v"(v));
    v = (q == 2) ? x[4 * r + 2] : v;
    asm volatile("" : "+v"(v));
    v = (q == 3) ? x[4 * r + 3] : v;
    c[r] = v;
  }
}

// x <- T^-T x for the upper triangular tile T (LDS, [j * TS + i]) by forward substitution, rinv[j] = 1 / T[j][j].
// The tile's rows are wave-uniform LDS reads (broadcast); row j + 1 is fetched while row j is applied -- written out by
// hand because the compiler otherwise waits for every single read in front of its FMA (136 serial LDS round trips).
__device__ __forceinline__ void tile_subst(double (&x)[16], const double* __restrict__ T, const double* __restrict__ rinv) {
  double rv[16], row[2][16];
#pragma unroll
  for (int j = 0; j < 16; ++j) rv[j] = rinv[j];
#pragma unroll
  for (int i = 1; i < 16; ++i) row[0][i] = T[i];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    if (j + 1 < 15) {
#pragma unroll
      for (int i = j + 2; i < 16; ++i) row[(j + 1) & 1][i] = T[(j + 1) * TS + i];
    }
    x[j] *= rv[j];
#pragma unroll
    for (int i = j + 1; i < 16; ++i) x[i] = fma(-row[j & 1][i], x[j], x[i]);
  }
}

// the same for two columns per lane at once: one read of the tile's row serves both, and the two dependent chains fill each
// other's 32-cycle latency shadows
__device__ __forceinline__ void tile_subst2(double (&x)[16], double (&y)[16], const double* __restrict__ T, const double* __restrict__ rinv) {
  double rv[16], row[2][16];
#pragma unroll
  for (int j = 0; j < 16; ++j) rv[j] = rinv[j];
#pragma unroll
  for (int i = 1; i < 16; ++i) row[0][i] = T[i];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    if (j + 1 < 15) {
#pragma unroll
      for (int i = j + 2; i < 16; ++i) row[(j + 1) & 1][i] = T[(j + 1) * TS + i];
    }
    x[j] *= rv[j];
    y[j] *= rv[j];
#pragma unroll
    for (int i = j + 1; i < 16; ++i) {
      x[i] = fma(-row[j & 1][i], x[j], x[i]);
      y[i] = fma(-row[j & 1][i], y[j], y[i]);
    }
  }
}

// the rows 4 q .. 4 q + 3 of a lane's column, for 32-byte contiguous stores (the four lanes of a column cover its 128 bytes)
__device__ __forceinline__ void tile_rows4(const double (&x)[16], int q, double (&v)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    double t = x[i];
    t = (q == 1) ? x[4 + i] : t;
    asm volatile("" : "+v"(t));
    t = (q == 2) ? x[8 + i] : t;
    asm volatile("" : "+v"(t));
    t = (q == 3) ? x[12 + i] : t;
    v[i] = t;
  }
}

// 1 / sqrt(d) and sqrt(d) to an ulp or two without the division / square-root expansions: v_rsq_f64 + two Newton
// steps, one correction of s (off the critical path: 16 independent chains interleave)
__device__ __forceinline__ void rsqrt_sqrt(double d, double& r, double& s) {
  r = __builtin_amdgcn_rsq(d);
  const double h = 0.5 * d;
  double e = fma(-h * r, r, 0.5);
  r = fma(r, e, r);
  e = fma(-h * r, r, 0.5);
  r = fma(r, e, r);
  s = d * r;
  const double t = fma(-s, s, d);
  s = fma(0.5 * r, t, s);
}

// Upper Cholesky of one 16 x 16 tile inside a wavefront.  Lane n (every 16-lane row holds a copy) owns column n, rows
// 0 .. n meaningful.  A dependent v_fma_f64 costs 32 cycles on gfx950 (tools/probe_potrf.hip), so what matters is the
// number of double-precision operations between one pivot and the next.  The elimination therefore runs on UNSCALED rows
// (the L D L' form): step j broadcasts the raw pivot d_j (DPP), refines 1 / d_j from v_rcp_f64 with one cubic step
// (e = 1 - d r; r += r (e + e^2): three operations), forms the lane's multiplier f = w_jn / d_j and applies
// a_in -= w_ji f with the raw row broadcast by DPP ahead of time -- six operations from pivot to pivot.  The rows are scaled
// by 1 / sqrt(d_j) afterwards, all sixteen chains side by side.  fail = 0 or the 1-based index of the first non-positive
// pivot; after a failure the remaining arithmetic runs on meaningless numbers (dpotrf stops there; callers only read info).
template <int J, int... Is>
__device__ __forceinline__ void tile_potrf_updates(double (&x)[16], double nf, std::integer_sequence<int, Is...>) {
  ((fmac_bcast16<J + 1 + Is>(x[J + 1 + Is], x[J], nf)), ...);
}
template <int J>
__device__ __forceinline__ void tile_potrf_step(double (&x)[16], double (&piv)[16]) {
  const double d = mov_bcast16<J>(x[J]);
  piv[J] = d;
  const double r = __builtin_amdgcn_rcp(d);   // (five dependent operations from pivot to pivot: see tile_potrf_step_inv)
  const double e = fma(-d, r, 1.0);
  const double nf0 = x[J] * -r;
  const double p = fma(e, e, e);
  const double nf = fma(nf0, p, nf0);
  tile_potrf_updates<J>(x, nf, std::make_integer_sequence<int, 15 - J>{});
}
template <int... Js>
__device__ __forceinline__ void tile_potrf_all(double (&x)[16], double (&piv)[16], std::integer_sequence<int, Js...>) {
  (tile_potrf_step<Js>(x, piv), ...);
}
// The same elimination with the tile's INVERSE riding along (round 4).  The row operations that take A to W (A = L W, L unit
// lower, L_iJ = w_Ji / d_J) take the identity to L^-1; lane n carries column n of that second block in xi and updates it with the
// SAME broadcast pivot-row entries: xi_i -= w_Ji (xi_J / d_J) -- one more multiplication and 15 - J more FMAs per step, none of
// them on the pivot-to-pivot chain (they sit in the 32-cycle shadows of the six dependent operations).  With U = D^-1/2 W:
// U^-T = D^-1/2 L^-1, so mi[i] = xi[i] / sqrt(d_i) is column n of M = U^-T (lower triangular).  M turns the substitution of
// every tile right of the diagonal -- 2.9 K cycles of FP64 FMAs issue-bound at 8 cycles, on the critical path of all eight block
// steps of a diagonal block and of every panel step -- into three 4-MFMA products (tile_solve_mfma).
// 1 / sqrt(d_j) for all sixteen pivots in every lane, and sqrt of the lane's own one.  Round 4: lane n computes ONE chain (its own
// pivot) and the sixteen values are exchanged by DPP broadcasts.  Before, every lane ran the sixteen Newton chains itself and the
// compiler emitted them one after the other -- sixteen chains of ten dependent FP64 operations at 32 cycles each, 5 K cycles between
// the last elimination step and the publication of the tile on the critical path of every block step
// (profiles/r04_probe_potrf.txt).  Same inputs, same operations: the values are bitwise the ones each lane used to compute.
template <int... Js>
__device__ __forceinline__ void tile_bcast_all(double r, double (&ri)[16], std::integer_sequence<int, Js...>) {
  ((ri[Js] = mov_bcast16<Js>(r)), ...);
}
__device__ __forceinline__ double tile_pivot_scales(const double (&piv)[16], int n, double (&ri)[16]) {
  double dn = piv[0];
#pragma unroll
  for (int j = 1; j < 16; ++j) dn = (n == j) ? piv[j] : dn;
  double rn, sqn;
  rsqrt_sqrt(dn, rn, sqn);
  tile_bcast_all(rn, ri, std::make_integer_sequence<int, 16>{});
  return sqn;
}

template <int J, int I>
__device__ __forceinline__ void tile_potrf_upd_inv(double (&x)[16], double (&xi)[16], double nf, double nf2) {
  fmac_bcast16<I>(x[I], x[J], nf);     // x_I -= w_JI f   (nf = -f: the same rounded value as fma(-w, f, x))
  fmac_bcast16<I>(xi[I], x[J], nf2);
}
template <int J, int... Is>
__device__ __forceinline__ void tile_potrf_updates_inv(double (&x)[16], double (&xi)[16], double f, double f2, std::integer_sequence<int, Is...>) {
  (tile_potrf_upd_inv<J, J + 1 + Is>(x, xi, f, f2), ...);
}
template <int J>
__device__ __forceinline__ void tile_potrf_step_inv(double (&x)[16], double (&xi)[16], double (&piv)[16]) {
  const double d = mov_bcast16<J>(x[J]);
  piv[J] = d;
  // -x_J / d and -xi_J / d from v_rcp_f64 (4.6e-8) and one cubic correction applied to the PRODUCTS: r0 (1 + e + e^2) with e = 1 - d r0
  // is 1 / d to 1.1e-16 (profiles/r02_probe_rcp.txt); the raw products run beside the two correction terms, so the chain from
  // pivot to pivot is rcp, e, e + e^2, the corrected multiplier, the update: five dependent operations (six when 1 / d was refined first)
  const double r = __builtin_amdgcn_rcp(d);
  const double e = fma(-d, r, 1.0);
  const double nf0 = x[J] * -r, nf20 = xi[J] * -r;
  const double p = fma(e, e, e);
  const double nf = fma(nf0, p, nf0);
  const double nf2 = fma(nf20, p, nf20);
  tile_potrf_updates_inv<J>(x, xi, nf, nf2, std::make_integer_sequence<int, 15 - J>{});
}
template <int... Js>
__device__ __forceinline__ void tile_potrf_all_inv(double (&x)[16], double (&xi)[16], double (&piv)[16], std::integer_sequence<int, Js...>) {
  (tile_potrf_step_inv<Js>(x, xi, piv), ...);
}
// x: column n of the tile (factor on return, as tile_potrf), mi: column n of U^-T
__device__ __forceinline__ int tile_potrf_inv(double (&x)[16], int n, double (&ri)[16], double (&mi)[16]) {
  double piv[16], xi[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) xi[j] = (j == n) ? 1.0 : 0.0;
  tile_potrf_all_inv(x, xi, piv, std::make_integer_sequence<int, 16>{});
  int fail = 0;
#pragma unroll
  for (int j = 15; j >= 0; --j) fail = !(piv[j] > 0.0) ? j + 1 : fail;
  const double sqn = tile_pivot_scales(piv, n, ri);
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    x[j] = (n == j) ? sqn : x[j] * ri[j];
    mi[j] = xi[j] * ri[j];   // (rows above the lane's own stay exactly zero: their multipliers are)
  }
  return fail;
}

// X = T^-T B for a 16 x 16 tile B in the MFMA C/D layout (register r = rows q + 4 r of column nn), T upper triangular in LDS
// ([k * TS + m]), M = T^-T in LDS ([i * TS + n]): X0 = M B, then ONE step of refinement against T itself, R = B - T' X0,
// X = X0 + M R -- the residual of the refined X is that of substitution as long as cond(T) eps << 1 (the same reasoning as the
// diagonal-block solves of the triangular sweeps, dense.hip: diag_solve).  ma / ua = this lane's operand entries
// M[nn][4 kc + q] and T[4 kc + q][nn], read once per block step and shared by all tiles of the step.
// (the factor tile is published as the owner holds it, with whatever the elimination left below the diagonal: masked HERE, four
//  selects per reader, instead of sixteen on the owner's critical path)
__device__ __forceinline__ void tile_inv_operands(const double* __restrict__ Mi, const double* __restrict__ Tt, int q, int nn, double (&ma)[4], double (&ua)[4]) {
#pragma unroll
  for (int kc = 0; kc < 4; ++kc) {
    ma[kc] = Mi[nn * TS + 4 * kc + q];
    const double u = Tt[(4 * kc + q) * TS + nn];
    ua[kc] = (4 * kc + q <= nn) ? u : 0.0;
  }
}
template <int N>
__device__ __forceinline__ void tile_solve_mfma(d4_t (&b)[N], int cnt, const double (&ma)[4], const double (&ua)[4]) {
  d4_t xs[N], rs[N];
#pragma unroll
  for (int i = 0; i < N; ++i) { xs[i] = (d4_t){0.0, 0.0, 0.0, 0.0}; rs[i] = b[i]; }
#pragma unroll
  for (int kc = 0; kc < 4; ++kc)
#pragma unroll
    for (int i = 0; i < N; ++i)
      if (i < cnt) xs[i] = mfma4(ma[kc], b[i][kc], xs[i]);
#pragma unroll
  for (int kc = 0; kc < 4; ++kc)
#pragma unroll
    for (int i = 0; i < N; ++i)
      if (i < cnt) rs[i] = mfma4(ua[kc], -xs[i][kc], rs[i]);
#pragma unroll
  for (int kc = 0; kc < 4; ++kc)
#pragma unroll
    for (int i = 0; i < N; ++i)
      if (i < cnt) xs[i] = mfma4(ma[kc], rs[i][kc], xs[i]);
#pragma unroll
  for (int i = 0; i < N; ++i)
    if (i < cnt) b[i] = xs[i];
}

__device__ __forceinline__ int tile_potrf(double (&x)[16], int n, double (&ri)[16]) {
  double piv[16];
  tile_potrf_all(x, piv, std::make_integer_sequence<int, 16>{});
  int fail = 0;
#pragma unroll
  for (int j = 15; j >= 0; --j) fail = !(piv[j] > 0.0) ? j + 1 : fail;
  const double sqn = tile_pivot_scales(piv, n, ri);
#pragma unroll
  for (int j = 0; j < 16; ++j) x[j] = (n == j) ? sqn : x[j] * ri[j];
  return fail;
}

}  // namespace

// =============================================================================================
// Diagonal block: A[k0 : k0 + nb, k0 : k0 + nb] (upper) <- its Cholesky factor, nb <= 128.  One workgroup of EIGHT
// wavefronts (two per SIMD); wavefront w owns tile column w (tiles (a, w), a <= w).  Block step jb:
//   1. wavefront jb factors the diagonal tile in registers and publishes it (LDS);           -- barrier --
//   2. wavefronts w > jb solve their tile (jb, w) against it, store it (final) and publish it; -- barrier --
//   3. wavefronts w > jb apply the rank-16 update to their tiles (a, w), jb < a <= w.
// Wavefront jb + 1 has ONE tile to update before it factors the next diagonal tile, so the critical path of a step is
// factor + one solve + one update; the other wavefronts' updates run underneath the next factorization.
// =============================================================================================
// the code of wavefront W, fully specialised (tile ownership, loop bounds and register indices are compile-time: with a
// run-time wavefront index every MFMA sat in its own exec-masked block behind a waited LDS read, ~500 cycles apiece).
// Wavefront W owns tile columns W and 7 - W (9 tiles each way).
template <int W>
__device__ __forceinline__ void potrf_diag_wave(double* __restrict__ Ab, long lda, int nb, int nbt, int lane, double* Dt, double* rinv, double* Pt,
                                                int* sfail) {
  constexpr int C0 = W, C1 = 7 - W;
  const int q = lane >> 4, nn = lane & 15;
  const int l0 = 16 * C0 + nn, l1 = 16 * C1 + nn;   // this lane's two columns of the block
  double* colp0 = Ab + (long)min(l0, nb - 1) * lda;
  double* colp1 = Ab + (long)min(l1, nb - 1) * lda;
  d4_t acc0[8], acc1[8];
  // loads: unconditional on clamped addresses (a predicated load is waited for individually), identity padding applied after
#pragma unroll
  for (int a = 0; a <= C0; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc0[a][r] = colp0[min(16 * a + q + 4 * r, nb - 1)];
#pragma unroll
  for (int a = 0; a <= C1; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc1[a][r] = colp1[min(16 * a + q + 4 * r, nb - 1)];
#pragma unroll
  for (int a = 0; a <= C0; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 16 * a + q + 4 * r;
      acc0[a][r] = (i < nb && l0 < nb && i <= l0) ? acc0[a][r] : (i == l0 ? 1.0 : 0.0);
    }
#pragma unroll
  for (int a = 0; a <= C1; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 16 * a + q + 4 * r;
      acc1[a][r] = (i < nb && l1 < nb && i <= l1) ? acc1[a][r] : (i == l1 ? 1.0 : 0.0);
    }
  if (W == 0 && lane == 0) *sfail = 0;   // (wavefront 0 also owns block step 0; later owners write after barriers)
  STAMP(0);
#pragma unroll
  for (int jb = 0; jb < 8; ++jb) {
    if (jb >= nbt) break;   // (identity padding needs no elimination)
    if (jb == 1) STAMP(4);
    WSTAMP(jb == 2, 0);
    // ---- 1. the owner of tile column jb factors the diagonal tile and publishes it
    const bool own0 = (C0 == jb), own1 = (C1 == jb);
    double xd[16];
    if (own0 || own1) {
      double ri[16];
      tile_gather(own0 ? acc0[jb] : acc1[jb], nn, xd);
      const int f = tile_potrf(xd, nn, ri);
      if (q == 0) {
#pragma unroll
        for (int j = 0; j < 16; ++j) Dt[j * TS + nn] = xd[j];
      }
      if (lane == 0) {
#pragma unroll
        for (int j = 0; j < 16; ++j) rinv[j] = ri[j];
        if (f && *sfail == 0) *sfail = 16 * jb + f;
      }
    }
    WSTAMP(jb == 2, 3);
    lds_barrier();
    WSTAMP(jb == 2, 4);
    if (jb == 0) STAMP(1);
    if (own0 || own1) {   // upper triangle of the diagonal tile -> memory (off the critical path: the others are solving)
      const int l = own0 ? l0 : l1;
      double* colp = own0 ? colp0 : colp1;
      if (q == 0 && l < nb) {
#pragma unroll
        for (int j = 0; j < 16; ++j)
          if (j <= nn) colp[16 * jb + j] = xd[j];
      }
    }
    // ---- 2. row panel: the owned tiles (jb, C0), (jb, C1) right of the diagonal are solved against it and published
    const bool m0 = (C0 > jb) && (16 * C0 < nb), m1 = (C1 > jb) && (16 * C1 < nb);
    double x0[16], x1[16];
    if (m0 && m1) {
      tile_gather(acc0[jb], nn, x0);
      tile_gather(acc1[jb], nn, x1);
      tile_subst2(x0, x1, Dt, rinv);
    } else if (m0) {
      tile_gather(acc0[jb], nn, x0);
      tile_subst(x0, Dt, rinv);
    } else if (m1) {
      tile_gather(acc1[jb], nn, x1);
      tile_subst(x1, Dt, rinv);
    }
    if (m0) {
      tile_scatter(x0, q, acc0[jb]);
#pragma unroll
      for (int r = 0; r < 4; ++r) Pt[C0 * TL + (4 * r + q) * TS + nn] = acc0[jb][r];
    }
    if (m1) {
      tile_scatter(x1, q, acc1[jb]);
#pragma unroll
      for (int r = 0; r < 4; ++r) Pt[C1 * TL + (4 * r + q) * TS + nn] = acc1[jb][r];
    }
    WSTAMP(jb == 2, 8);
    lds_barrier();
    WSTAMP(jb == 2, 9);
    if (jb == 0) STAMP(2);
    // ---- 3. rank-16 update of the owned tiles (a, C), jb < a <= C: k-chunks outside, tiles inside (independent accumulators
    //         back to back; a dependent MFMA waits for its predecessor).  The own column's diagonal tile needs no LDS: its
    //         registers ARE the first operand -- and it goes first: it is the next diagonal tile when C = jb + 1.
    if (m0 || m1) {
      constexpr int AMAX = (C0 > C1) ? C0 : C1;
      double op[4][8];
#pragma unroll
      for (int kc = 0; kc < 4; ++kc)
#pragma unroll
        for (int a = jb + 1; a <= AMAX; ++a) op[kc][a] = Pt[a * TL + (4 * kc + q) * TS + nn];
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) {
        if (m0) acc0[C0] = mfma4(acc0[jb][kc], -acc0[jb][kc], acc0[C0]);
        if (m1) acc1[C1] = mfma4(acc1[jb][kc], -acc1[jb][kc], acc1[C1]);
#pragma unroll
        for (int a = jb + 1; a <= AMAX; ++a) {
          if (m0 && a < C0) acc0[a] = mfma4(op[kc][a], -acc0[jb][kc], acc0[a]);
          if (m1 && a < C1) acc1[a] = mfma4(op[kc][a], -acc1[jb][kc], acc1[a]);
        }
      }
    }
    // the solved rows are final: 32 contiguous bytes per lane
    if (m0 && l0 < nb) {
      double v[4];
      tile_rows4(x0, q, v);
      double* dst = colp0 + 16 * jb + 4 * q;
      *reinterpret_cast<d2_t*>(dst) = (d2_t){v[0], v[1]};
      *reinterpret_cast<d2_t*>(dst + 2) = (d2_t){v[2], v[3]};
    }
    if (m1 && l1 < nb) {
      double v[4];
      tile_rows4(x1, q, v);
      double* dst = colp1 + 16 * jb + 4 * q;
      *reinterpret_cast<d2_t*>(dst) = (d2_t){v[0], v[1]};
      *reinterpret_cast<d2_t*>(dst + 2) = (d2_t){v[2], v[3]};
    }
    WSTAMP(jb == 2, 10);
  }
  STAMP(5);
}

__global__ __launch_bounds__(256) void potrf_diag_mfma_kernel(double* __restrict__ A, long lda, long strideA, int n, int k0, int* __restrict__ info) {
  extern __shared__ __attribute__((aligned(16))) double pm_lds[];
  double* Dt = pm_lds;              // factor of the current diagonal tile
  double* rinv = pm_lds + TL;       // 1 / its diagonal
  double* Pt = pm_lds + TL + 16;    // solved row panel of the current block step: tile b at Pt + b * TL, [k][m]
  int* sfail = reinterpret_cast<int*>(pm_lds + TL + 16 + 8 * TL);   // first failed pivot of the block (1-based), 0 = none
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nb = min(NB, n - k0);
  const int nbt = (nb + 15) >> 4;
  double* Ab = A + (long)blockIdx.x * strideA + (long)k0 * lda + k0;
  switch (w) {
    case 0: potrf_diag_wave<0>(Ab, lda, nb, nbt, lane, Dt, rinv, Pt, sfail); break;
    case 1: potrf_diag_wave<1>(Ab, lda, nb, nbt, lane, Dt, rinv, Pt, sfail); break;
    case 2: potrf_diag_wave<2>(Ab, lda, nb, nbt, lane, Dt, rinv, Pt, sfail); break;
    default: potrf_diag_wave<3>(Ab, lda, nb, nbt, lane, Dt, rinv, Pt, sfail); break;
  }
  lds_barrier();
  if (tid == 0 && *sfail && *sfail <= nb) atomicCAS(&info[blockIdx.x], 0, k0 + *sfail);
  STAMP(6);
}

// =============================================================================================
// Third form of the diagonal-block kernel (default): up to NT <= 16 tile columns in one workgroup, with look-ahead INSIDE the
// kernel.  Four wavefronts; tile column c belongs to wavefront c mod 4 (c mod 8 < 4) or 3 - c mod 4 (else), so each wavefront
// owns columns W, 7 - W, 8 + W, 15 - W: a balanced share of the triangle.  Block step JB of a wavefront:
//   A. (owner of column JB only) apply the previous step's rank-16 update to tile (JB, JB) -- both operands are its own
//      registers --, factor it, publish it and raise the step's flag (an LDS word; no barrier);
//   B. apply the previous step's update to the rest of the owned tiles, first operands from the row panel published in LDS;
//   C. wait for the flag; solve the owned tiles of row JB by substitution against the published diagonal tile, publish them;
//   -- ONE barrier --
// so the 16 x 16 factorization (4.7 K cycles, tools/probe_potrf.hip) runs underneath the other wavefronts' updates instead of
// in front of a barrier of its own.  The published row panel is double-buffered (a fast wavefront may publish step JB while a
// slow one still reads step JB - 1); the diagonal tile is not (it is rewritten only after the barrier that ends its use).
// Measured (tools/bench_potrf.py): n = 5000 3.62 -> 3.55 ms -- the per-step gain is smaller than the 4.7 K cycles moved off
// the barrier path because the owner's own solve and the substitution of the others now bound the step.  The template also
// takes NW = 8 wavefronts with NT = 16 tile columns (a whole matrix of side <= 256 in one launch); that form needs 17 tiles
// plus the substitution temporaries in the 256 registers left at two wavefronts per SIMD, spills, and was no faster than the
// blocked path (148 vs 150 us at side 200): not instantiated.
// =============================================================================================
// (NW = 8 wavefronts: column c belongs to wavefront c (c < 8) or 15 - c, two columns each -- the form for NT = 16, where four
// wavefronts would hold up to 34 tiles each and spill)
__device__ __forceinline__ constexpr int tile_owner(int NW, int c) { return NW == 4 ? ((c & 4) ? 3 - (c & 3) : (c & 3)) : ((c & 8) ? 7 - (c & 7) : (c & 7)); }
__device__ __forceinline__ constexpr int own_col(int NW, int W, int k) {   // ascending in k; 99 = none
  return NW == 4 ? (k == 0 ? W : k == 1 ? 7 - W : k == 2 ? 8 + W : 15 - W) : (k == 0 ? W : k == 1 ? 15 - W : 99);
}
__device__ __forceinline__ constexpr int own_idx(int NW, int c) { return NW == 4 ? (((c >> 3) << 1) | ((c >> 2) & 1)) : ((c >> 3) & 1); }
__device__ __forceinline__ constexpr int own_cnt(int NW, int W, int NT) {
  int n = 0;
  for (int k = 0; k < 4; ++k) n += own_col(NW, W, k) < NT ? 1 : 0;
  return n;
}
// first index k of W's columns with own_col(NW, W, k) > c (own_cnt if none)
__device__ __forceinline__ constexpr int own_first_above(int NW, int W, int NT, int c) {
  int k = 0;
  while (k < own_cnt(NW, W, NT) && own_col(NW, W, k) <= c) ++k;
  return k;
}

template <int NW, int W, int NT, int JB, bool INV>
__device__ __forceinline__ void potrf_tiles_step(d4_t (&acc)[4][NT], double* const (&colp)[4], const int (&lcol)[4], int nb, int lane, double* Dt,
                                                 double* rinv, double* Pt, int* sfail, int* dflag, double* Mi, double* __restrict__ tinv) {
  constexpr int NC = own_cnt(NW, W, NT);
  constexpr bool OWNER = tile_owner(NW, JB) == W;
  constexpr int KO = own_idx(NW, JB);
  constexpr int P = JB - 1;
  const int q = lane >> 4, nn = lane & 15;
  double xd[16];
  // ---- A. owner: the next diagonal tile
  if constexpr (OWNER) {
    if constexpr (JB > 0) {
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) acc[KO][JB] = mfma4(acc[KO][P][kc], -acc[KO][P][kc], acc[KO][JB]);
    }
    double ri[16];
    tile_gather(acc[KO][JB], nn, xd);
    int f;
    if constexpr (INV) {
      double mi[16];
      f = tile_potrf_inv(xd, nn, ri, mi);
      if (q == 1) {   // (the four 16-lane rows hold identical copies: row 0 publishes the factor, row 1 its inverse, row 2 keeps the inverse for the panel kernels)
#pragma unroll
        for (int j = 0; j < 16; ++j) Mi[j * TS + nn] = mi[j];
      }
      if (q == 2 && tinv != nullptr) {   // (column-major in the record: entry (i, n) at 16 n + i)
#pragma unroll
        for (int j = 0; j < 16; ++j) tinv[JB * 256 + nn * 16 + j] = mi[j];
      }
    } else {
      f = tile_potrf(xd, nn, ri);
    }
    if (q == 0) {   // (INV: the tile is an MFMA operand as a whole, so the rows below the diagonal -- never read by the substitution -- are zeroed)
#pragma unroll
      for (int j = 0; j < 16; ++j) Dt[j * TS + nn] = xd[j];
    }
    if (lane == 0) {
#pragma unroll
      for (int j = 0; j < 16; ++j) rinv[j] = ri[j];
      if (f && *sfail == 0) *sfail = 16 * JB + f;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) *reinterpret_cast<volatile int*>(dflag) = JB + 1;
  }
  // ---- B. the previous step's rank-16 update of the other owned tiles (a, C), P < a <= C
  if constexpr (JB > 0) {
    constexpr int K0 = own_first_above(NW, W, NT, P);
    if constexpr (K0 < NC) {
      constexpr int AMAX = own_col(NW, W, NC - 1);
      const double* Pp = Pt + (P & 1) * NT * TL;
      bool m[4] = {false, false, false, false};
      bool any = false;
#pragma unroll
      for (int k = K0; k < NC; ++k) {
        m[k] = (16 * own_col(NW, W, k) < nb) && !(OWNER && k == KO);
        any |= m[k];
      }
      if (any) {
        double op[2][NT];
#pragma unroll
        for (int a = JB; a < AMAX; ++a) op[0][a] = Pp[a * TL + q * TS + nn];
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
          if (kc < 3) {
#pragma unroll
            for (int a = JB; a < AMAX; ++a) op[(kc + 1) & 1][a] = Pp[a * TL + (4 * (kc + 1) + q) * TS + nn];
          }
#pragma unroll
          for (int k = K0; k < NC; ++k) {
            constexpr int dummy = 0; (void)dummy;
            const int C = own_col(NW, W, k);
            if (m[k]) {
              acc[k][C] = mfma4(acc[k][P][kc], -acc[k][P][kc], acc[k][C]);   // (own registers: goes first)
#pragma unroll
              for (int a = JB; a < NT; ++a)
                if (a < C) acc[k][a] = mfma4(op[kc & 1][a], -acc[k][P][kc], acc[k][a]);
            }
          }
        }
      }
    }
  }
  if constexpr (OWNER) {   // upper triangle of the diagonal tile -> memory
    if (q == 0 && lcol[KO] < nb) {
#pragma unroll
      for (int j = 0; j < 16; ++j)
        if (j <= nn) colp[KO][16 * JB + j] = xd[j];
    }
  }
  // ---- C. row panel of step JB: the owned tiles (JB, C), C > JB
  constexpr int K1 = own_first_above(NW, W, NT, JB);
  constexpr int NS = NC - K1;
  if constexpr (NS > 0) {
    int cnt = 0;
#pragma unroll
    for (int k = K1; k < NC; ++k) cnt += (16 * own_col(NW, W, k) < nb) ? 1 : 0;
    if (INV && cnt > 0) {   // the solves as products with the published inverse of the diagonal tile (tile_solve_mfma)
      if constexpr (!OWNER) {
        while (*reinterpret_cast<volatile int*>(dflag) <= JB) __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
      }
      double ma[4], ua[4];
      tile_inv_operands(Mi, Dt, q, nn, ma, ua);
      d4_t bt[NS];
#pragma unroll
      for (int i = 0; i < NS; ++i) bt[i] = acc[K1 + i][JB];
      tile_solve_mfma<NS>(bt, cnt, ma, ua);
      double* Pc = Pt + (JB & 1) * NT * TL;
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        if (i < cnt) {
          const int C = own_col(NW, W, K1 + i);
          acc[K1 + i][JB] = bt[i];
#pragma unroll
          for (int r = 0; r < 4; ++r) Pc[C * TL + (4 * r + q) * TS + nn] = bt[i][r];
          if (lcol[K1 + i] < nb) {   // the solved rows are final
            double* dst = colp[K1 + i] + 16 * JB + q;
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[4 * r] = bt[i][r];
          }
        }
      }
    } else if (cnt > 0) {
      double x[NS][16];
#pragma unroll
      for (int i = 0; i < NS; ++i)
        if (i < cnt) tile_gather(acc[K1 + i][JB], nn, x[i]);
      if constexpr (!OWNER) {
        while (*reinterpret_cast<volatile int*>(dflag) <= JB) __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
      }
      if constexpr (NS >= 2) {
        if (cnt >= 2) tile_subst2(x[0], x[1], Dt, rinv);
        else tile_subst(x[0], Dt, rinv);
      } else {
        tile_subst(x[0], Dt, rinv);
      }
      if constexpr (NS >= 4) {
        if (cnt == 4) tile_subst2(x[2], x[3], Dt, rinv);
      }
      if constexpr (NS >= 3) {
        if (cnt == 3) tile_subst(x[2], Dt, rinv);
      }
      double* Pc = Pt + (JB & 1) * NT * TL;
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        if (i < cnt) {
          const int C = own_col(NW, W, K1 + i);
          tile_scatter(x[i], q, acc[K1 + i][JB]);
#pragma unroll
          for (int r = 0; r < 4; ++r) Pc[C * TL + (4 * r + q) * TS + nn] = acc[K1 + i][JB][r];
          if (lcol[K1 + i] < nb) {   // the solved rows are final: 32 contiguous bytes per lane
            double v[4];
            tile_rows4(x[i], q, v);
            double* dst = colp[K1 + i] + 16 * JB + 4 * q;
            *reinterpret_cast<d2_t*>(dst) = (d2_t){v[0], v[1]};
            *reinterpret_cast<d2_t*>(dst + 2) = (d2_t){v[2], v[3]};
          }
        }
      }
    }
  }
  lds_barrier();
}

template <int NW, int W, int NT, bool INV, int... JBs>
__device__ __forceinline__ void potrf_tiles_steps(d4_t (&acc)[4][NT], double* const (&colp)[4], const int (&lcol)[4], int nb, int nbt, int lane,
                                                  double* Dt, double* rinv, double* Pt, int* sfail, int* dflag, double* Mi, double* tinv,
                                                  std::integer_sequence<int, JBs...>) {
  ((JBs < nbt ? potrf_tiles_step<NW, W, NT, JBs, INV>(acc, colp, lcol, nb, lane, Dt, rinv, Pt, sfail, dflag, Mi, tinv) : (void)0), ...);
}

template <int NW, int W, int NT, bool INV>
__device__ __forceinline__ void potrf_tiles_wave(double* __restrict__ Ab, long lda, int nb, int nbt, int lane, double* Dt, double* rinv, double* Pt,
                                                 int* sfail, int* dflag, double* Mi, double* tinv) {
  constexpr int NC = own_cnt(NW, W, NT);
  const int q = lane >> 4, nn = lane & 15;
  d4_t acc[4][NT];
  double* colp[4];
  int lcol[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    lcol[k] = 16 * own_col(NW, W, k) + nn;
    colp[k] = Ab + (long)min(lcol[k], nb - 1) * lda;
  }
  // loads: unconditional on clamped addresses (a predicated load is waited for individually), identity padding applied after
#pragma unroll
  for (int k = 0; k < NC; ++k)
#pragma unroll
    for (int a = 0; a < NT; ++a)
      if (a <= own_col(NW, W, k)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[k][a][r] = colp[k][min(16 * a + q + 4 * r, nb - 1)];
      }
#pragma unroll
  for (int k = 0; k < NC; ++k)
#pragma unroll
    for (int a = 0; a < NT; ++a)
      if (a <= own_col(NW, W, k)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = 16 * a + q + 4 * r;
          acc[k][a][r] = (i < nb && lcol[k] < nb && i <= lcol[k]) ? acc[k][a][r] : (i == lcol[k] ? 1.0 : 0.0);
        }
      }
  potrf_tiles_steps<NW, W, NT, INV>(acc, colp, lcol, nb, nbt, lane, Dt, rinv, Pt, sfail, dflag, Mi, tinv, std::make_integer_sequence<int, NT>{});
}

template <int NW, int NT, bool INV>
__global__ __launch_bounds__(64 * NW) void potrf_tiles_kernel(double* __restrict__ A, long lda, long strideA, int n, int k0, int* __restrict__ info,
                                                              double* __restrict__ tinv_base, long tinv_stride) {
  extern __shared__ __attribute__((aligned(16))) double pm_lds[];
  double* Dt = pm_lds;              // factor of the current diagonal tile
  double* rinv = pm_lds + TL;       // 1 / its diagonal
  double* Pt = pm_lds + TL + 16;    // solved row panels of two consecutive block steps: tile b of step s at Pt + ((s & 1) NT + b) TL, [k][m]
  int* sfail = reinterpret_cast<int*>(pm_lds + TL + 16 + 2 * NT * TL);   // first failed pivot of the block (1-based), 0 = none
  int* dflag = sfail + 1;                                                // number of diagonal tiles published so far
  double* Mi = pm_lds + TL + 16 + 2 * NT * TL + 2;                       // INV: the inverse (transposed) of the current diagonal tile, [i][n]
  double* tinv = tinv_base ? tinv_base + (long)blockIdx.x * tinv_stride : nullptr;   // INV: the NT tile inverses of this block, for the panel kernels
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nb = min(16 * NT, n - k0);
  const int nbt = (nb + 15) >> 4;
  double* Ab = A + (long)blockIdx.x * strideA + (long)k0 * lda + k0;
  if (tid == 0) { *sfail = 0; *dflag = 0; }
  lds_barrier();
  if constexpr (NW == 4) {
    switch (w) {
      case 0: potrf_tiles_wave<NW, 0, NT, INV>(Ab, lda, nb, nbt, lane, Dt, rinv, Pt, sfail, dflag, Mi, tinv); break;
      case 1: potrf_tiles_wave<NW, 1, NT, INV>(Ab, lda, nb, nbt, lane, Dt, rinv, Pt, sfail, dflag, Mi, tinv); break;
      case 2: potrf_tiles_wave<NW, 2, NT, INV>(Ab, lda, nb, nbt, lane, Dt, rinv, Pt, sfail, dflag, Mi, tinv); break;
      default: potrf_tiles_wave<NW, 3, NT, INV>(Ab, lda, nb, nbt, lane, Dt, rinv, Pt, sfail, dflag, Mi, tinv); break;
    }
  } else {
    switch (w) {
      case 0: potrf_tiles_wave<NW, 0, NT, INV>(Ab, lda, nb, nbt, lane, Dt, rinv, Pt, sfail, dflag, Mi, tinv); break;
      case 1: potrf_tiles_wave<NW, 1, NT, INV>(Ab, lda, nb, nbt, lane, Dt, rinv, Pt, sfail, dflag, Mi, tinv); break;
      case 2: potrf_tiles_wave<NW, 2, NT, INV>(Ab, lda, nb, nbt, lane, Dt, rinv, Pt, sfail, dflag, Mi, tinv); break;
      case 3: potrf_tiles_wave<NW, 3, NT, INV>(Ab, lda, nb, nbt, lane, Dt, rinv, Pt, sfail, dflag, Mi, tinv); break;
      case 4: potrf_tiles_wave<NW, 4, NT, INV>(Ab, lda, nb, nbt, lane, Dt, rinv, Pt, sfail, dflag, Mi, tinv); break;
      case 5: potrf_tiles_wave<NW, 5, NT, INV>(Ab, lda, nb, nbt, lane, Dt, rinv, Pt, sfail, dflag, Mi, tinv); break;
      case 6: potrf_tiles_wave<NW, 6, NT, INV>(Ab, lda, nb, nbt, lane, Dt, rinv, Pt, sfail, dflag, Mi, tinv); break;
      default: potrf_tiles_wave<NW, 7, NT, INV>(Ab, lda, nb, nbt, lane, Dt, rinv, Pt, sfail, dflag, Mi, tinv); break;
    }
  }
  if (tid == 0 && *sfail && *sfail <= nb) atomicCAS(&info[blockIdx.x], 0, k0 + *sfail);
}

// =============================================================================================
// Fourth form of the diagonal-block kernel (round 4, default with the tile inverses): what bounded a block step of the third
// form was the OWNER's own serial path -- factor the diagonal tile (4.7 K cycles), then the previous step's rank-16 update of all
// its other tiles (up to 36 FP64 MFMAs of 64 cycles), then its solves -- with the three other wavefronts waiting at the barrier
// (profiles/r04_probe_potrf.txt: 12 K cycles per step where factor + one solve are 6 K).  Here
//   * the owner of step JB applies the previous panel only to ROW JB of its tiles and defers the rest to step JB + 1, where it is
//     not the owner (tile column c belongs to wavefront 0,1,2,3,2,3,0,1: nobody owns two consecutive steps) and would otherwise
//     wait for the next factorization; the published row panels are triple-buffered so that the deferred update can still read
//     panel JB - 1 while panel JB + 1 is being published;
//   * the solves are products with the diagonal tile's inverse (tile_solve_mfma, three 4-MFMA products: 1.0 K cycles instead
//     of 2.9 K + gather + scatter).
// Per step the owner now runs: 4 MFMAs + gather + factor + publish | 4 MFMAs + one solve + publish; the others: their updates
// (underneath the factorization), flag, at most two solves.  Same arithmetic per entry as the third form with INV (the order of
// the rank-16 updates of a tile is unchanged: panel 0, 1, 2, ...), so the two agree to the last bit.
// =============================================================================================
__device__ __forceinline__ constexpr int t4_owner(int c) { return c < 4 ? c : (c + 2) & 3; }
__device__ __forceinline__ constexpr int t4_col1(int W) { return W < 2 ? W + 6 : W + 2; }   // the second column of wavefront W (the first is W)

// panel PP (its tiles in Pb, tile (PP, a) at a TL) applied to this wavefront's tiles (a, C), max(R0, PP + 1) <= a <= min(R1, C),
// for C = C0 (if DO0) and C1 (if DO1); the own column's diagonal tile takes both operands from registers
template <int W, int PP, int R0, int R1, bool DO0, bool DO1>
__device__ __forceinline__ void t4_apply(d4_t (&acc0)[8], d4_t (&acc1)[8], const double* __restrict__ Pb, int q, int nn, bool m0, bool m1) {
  constexpr int C0 = W, C1 = t4_col1(W);
  constexpr int A0 = (R0 > PP + 1) ? R0 : PP + 1;
  constexpr bool U0 = DO0 && C0 > PP && A0 <= C0 && A0 <= R1, U1 = DO1 && C1 > PP && A0 <= C1 && A0 <= R1;
  if constexpr (U0 || U1) {
    if (!((U0 && m0) || (U1 && m1))) return;
    constexpr int AHI = (R1 < 7) ? R1 : 7;
    double op[4][8];
#pragma unroll
    for (int kc = 0; kc < 4; ++kc)
#pragma unroll
      for (int a = A0; a <= AHI; ++a)
        if ((U0 && a < C0) || (U1 && a < C1)) op[kc][a] = Pb[a * TL + (4 * kc + q) * TS + nn];
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
      if constexpr (U0 && C0 <= R1) { if (m0) acc0[C0] = mfma4(acc0[PP][kc], -acc0[PP][kc], acc0[C0]); }
      if constexpr (U1 && C1 <= R1) { if (m1) acc1[C1] = mfma4(acc1[PP][kc], -acc1[PP][kc], acc1[C1]); }
#pragma unroll
      for (int a = A0; a <= AHI; ++a) {
        if constexpr (U0) { if (a < C0 && m0) acc0[a] = mfma4(op[kc][a], -acc0[PP][kc], acc0[a]); }
        if constexpr (U1) { if (a < C1 && m1) acc1[a] = mfma4(op[kc][a], -acc1[PP][kc], acc1[a]); }
      }
    }
  }
}

template <int W, int JB>
__device__ __forceinline__ void t4_step(d4_t (&acc0)[8], d4_t (&acc1)[8], double* colp0, double* colp1, int l0, int l1, int nb, int lane, double* Dt,
                                        double* rinv, double* Pt, int* sfail, int* dflag, double* Mi, double* __restrict__ tinv) {
  constexpr int C0 = W, C1 = t4_col1(W);
  constexpr bool OWN0 = (C0 == JB), OWN1 = (C1 == JB), OWNER = OWN0 || OWN1;
  constexpr bool WAS = JB >= 1 && (C0 == JB - 1 || C1 == JB - 1);
  const int q = lane >> 4, nn = lane & 15;
  const bool m0 = 16 * C0 < nb, m1 = 16 * C1 < nb;
  const double* P1 = Pt + ((JB + 2) % 3) * 8 * TL;   // panel JB - 1
  const double* P2 = Pt + ((JB + 1) % 3) * 8 * TL;   // panel JB - 2
  double xd[16], miown[16];
  int fown = 0;
  WSTAMP(JB == 2, 0);
  if constexpr (OWNER) {
    // ---- the diagonal tile: the previous panel's update (own registers), factor with the inverse riding along, publish, flag
    if constexpr (JB > 0) t4_apply<W, JB - 1, JB, JB, OWN0, OWN1>(acc0, acc1, P1, q, nn, m0, m1);
    double ri[16], mi[16];
    WSTAMP(JB == 2, 1);
    tile_gather(OWN0 ? acc0[JB] : acc1[JB], nn, xd);
    WSTAMP(JB == 2, 2);
    const int f = tile_potrf_inv(xd, nn, ri, mi);
    WSTAMP(JB == 2, 3);
    // (the four 16-lane rows hold identical copies: row 0 publishes the factor, row 1 its inverse; nothing else in front of the flag)
    if (q == 0) {
#pragma unroll
      for (int j = 0; j < 16; ++j) Dt[j * TS + nn] = xd[j];
    }
    if (q == 1) {
#pragma unroll
      for (int j = 0; j < 16; ++j) Mi[j * TS + nn] = mi[j];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) *reinterpret_cast<volatile int*>(dflag) = JB + 1;
    WSTAMP(JB == 2, 4);
    fown = f;
#pragma unroll
    for (int j = 0; j < 16; ++j) miown[j] = mi[j];
    // row JB of the other column (the rest of this panel's update is deferred to the next step)
    if constexpr (JB > 0) t4_apply<W, JB - 1, JB, JB, !OWN0, !OWN1>(acc0, acc1, P1, q, nn, m0, m1);
  } else {
    if constexpr (WAS && JB >= 2) t4_apply<W, JB - 2, JB, 7, true, true>(acc0, acc1, P2, q, nn, m0, m1);   // deferred when this wavefront owned step JB - 1
    if constexpr (JB >= 1) t4_apply<W, JB - 1, JB, 7, true, true>(acc0, acc1, P1, q, nn, m0, m1);
  }
  WSTAMP(JB == 2, 5);
  // ---- row panel of step JB: the owned tiles (JB, C), C > JB, as products with the published inverse
  constexpr bool S0 = C0 > JB, S1 = C1 > JB;
  if constexpr (S0 || S1) {
    const bool s0 = S0 && m0, s1 = S1 && m1;
    if (s0 || s1) {
      if constexpr (!OWNER) {
        while (*reinterpret_cast<volatile int*>(dflag) <= JB) __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
      }
      WSTAMP(JB == 2, 6);
      double ma[4], ua[4];
      tile_inv_operands(Mi, Dt, q, nn, ma, ua);
      double* Pc = Pt + (JB % 3) * 8 * TL;
      if (s0 && s1) {
        d4_t bt[2] = {acc0[JB], acc1[JB]};
        tile_solve_mfma<2>(bt, 2, ma, ua);
        acc0[JB] = bt[0];
        acc1[JB] = bt[1];
      } else {
        d4_t bt[1] = {s0 ? acc0[JB] : acc1[JB]};
        tile_solve_mfma<1>(bt, 1, ma, ua);
        if (s0) acc0[JB] = bt[0];
        else acc1[JB] = bt[0];
      }
      WSTAMP(JB == 2, 7);
      if (s0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) Pc[C0 * TL + (4 * r + q) * TS + nn] = acc0[JB][r];
      }
      if (s1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) Pc[C1 * TL + (4 * r + q) * TS + nn] = acc1[JB][r];
      }
    }
  }
  WSTAMP(JB == 2, 8);
  lds_barrier();
  WSTAMP(JB == 2, 9);
  // ---- everything that goes to memory, behind the barrier (nobody waits for it): the solved rows (final), the diagonal tile's
  //      upper triangle, its inverse for the panel kernels, the failure record
  if constexpr (S0) {
    if (m0 && l0 < nb) {
      double* dst = colp0 + 16 * JB + q;
#pragma unroll
      for (int r = 0; r < 4; ++r) dst[4 * r] = acc0[JB][r];
    }
  }
  if constexpr (S1) {
    if (m1 && l1 < nb) {
      double* dst = colp1 + 16 * JB + q;
#pragma unroll
      for (int r = 0; r < 4; ++r) dst[4 * r] = acc1[JB][r];
    }
  }
  if constexpr (OWNER) {   // all 64 lanes: lane (q, nn) stores rows 4 q .. 4 q + 3 of column nn (sixteen predicated stores by one row of lanes took 1 K cycles)
    const int l = OWN0 ? l0 : l1;
    double* colp = OWN0 ? colp0 : colp1;
    double v[4];
    tile_rows4(xd, q, v);
    if (l < nb) {
      double* dst = colp + 16 * JB + 4 * q;
      if (4 * q + 3 <= nn) {
        *reinterpret_cast<d2_t*>(dst) = (d2_t){v[0], v[1]};
        *reinterpret_cast<d2_t*>(dst + 2) = (d2_t){v[2], v[3]};
      } else if (4 * q <= nn) {   // the group that crosses the diagonal: entry by entry (nothing below the diagonal is written)
        dst[0] = v[0];
        if (4 * q + 1 <= nn) dst[1] = v[1];
        if (4 * q + 2 <= nn) dst[2] = v[2];
      }
    }
    if (tinv != nullptr) {   // column-major in the record: tile JB at 256 JB, entry (i, n) at 16 n + i
      tile_rows4(miown, q, v);
      double* dst = tinv + JB * 256 + 16 * nn + 4 * q;
      *reinterpret_cast<d2_t*>(dst) = (d2_t){v[0], v[1]};
      *reinterpret_cast<d2_t*>(dst + 2) = (d2_t){v[2], v[3]};
    }
    if (lane == 0 && fown && *sfail == 0) *sfail = 16 * JB + fown;
  }
}

template <int W, int... JBs>
__device__ __forceinline__ void t4_steps(d4_t (&acc0)[8], d4_t (&acc1)[8], double* colp0, double* colp1, int l0, int l1, int nb, int nbt, int lane,
                                         double* Dt, double* rinv, double* Pt, int* sfail, int* dflag, double* Mi, double* tinv,
                                         std::integer_sequence<int, JBs...>) {
  ((JBs < nbt ? t4_step<W, JBs>(acc0, acc1, colp0, colp1, l0, l1, nb, lane, Dt, rinv, Pt, sfail, dflag, Mi, tinv) : (void)0), ...);
}

// The previous block step's update of THIS diagonal block, folded into its factorization (HYP_POTRF_DIAGUPD, dense.hip): the block
// C = A[k0:k0+nb, k0:k0+nb] still lacks  C -= P' P,  P = the 128 x nb panel block right above it (solved by the previous step's panel
// kernel).  Each wavefront forms the products of its own tiles -- the panel's 16-row slices travel through the LDS ring, eight rounds
// of one barrier -- in accumulators that start at zero, in the k order of the GEMM that used to do it (16-row tiles ascending, MFMA
// chunks of 4 ascending), and subtracts them from the loaded block with one rounding: the bits of  gemm(alpha = -1, beta = 1).
template <int W>
__device__ __forceinline__ void t4_prev_update(d4_t (&acc0)[8], d4_t (&acc1)[8], const double* __restrict__ Pg, long lda, int nb, int lane, double* Pt) {
  constexpr int C0 = W, C1 = t4_col1(W);
  const int q = lane >> 4, nn = lane & 15;
  const double* pc0 = Pg + (long)min(16 * C0 + nn, nb - 1) * lda + q;
  const double* pc1 = Pg + (long)min(16 * C1 + nn, nb - 1) * lda + q;
#pragma unroll
  for (int a = 0; a <= C0; ++a) acc0[a] = (d4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int a = 0; a <= C1; ++a) acc1[a] = (d4_t){0.0, 0.0, 0.0, 0.0};
  d4_t n0, n1;
#pragma unroll
  for (int g = 0; g < 4; ++g) { n0[g] = pc0[4 * g]; n1[g] = pc1[4 * g]; }
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const d4_t p0 = n0, p1 = n1;
    if (r + 1 < 8) {
#pragma unroll
      for (int g = 0; g < 4; ++g) { n0[g] = pc0[16 * (r + 1) + 4 * g]; n1[g] = pc1[16 * (r + 1) + 4 * g]; }
    }
    double* Pc = Pt + (r % 3) * 8 * TL;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      Pc[C0 * TL + (4 * g + q) * TS + nn] = p0[g];
      Pc[C1 * TL + (4 * g + q) * TS + nn] = p1[g];
    }
    lds_barrier();
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
#pragma unroll
      for (int a = 0; a <= C0; ++a) acc0[a] = mfma4(Pc[a * TL + (4 * kc + q) * TS + nn], p0[kc], acc0[a]);
#pragma unroll
      for (int a = 0; a <= C1; ++a) acc1[a] = mfma4(Pc[a * TL + (4 * kc + q) * TS + nn], p1[kc], acc1[a]);
    }
  }
  lds_barrier();   // (the ring is the factorization's from here)
}

template <int W, bool HASP>
__device__ __forceinline__ void t4_wave(double* __restrict__ Ab, long lda, int nb, int nbt, int lane, double* Dt, double* rinv, double* Pt, int* sfail,
                                        int* dflag, double* Mi, double* tinv, const double* __restrict__ Pg) {
  constexpr int C0 = W, C1 = t4_col1(W);
  const int q = lane >> 4, nn = lane & 15;
  const int l0 = 16 * C0 + nn, l1 = 16 * C1 + nn;
  double* colp0 = Ab + (long)min(l0, nb - 1) * lda;
  double* colp1 = Ab + (long)min(l1, nb - 1) * lda;
  d4_t acc0[8], acc1[8];
  if constexpr (HASP) {
    t4_prev_update<W>(acc0, acc1, Pg, lda, nb, lane, Pt);
#pragma unroll
    for (int a = 0; a <= C0; ++a) {
      d4_t c;
#pragma unroll
      for (int r = 0; r < 4; ++r) c[r] = colp0[min(16 * a + q + 4 * r, nb - 1)];
#pragma unroll
      for (int r = 0; r < 4; ++r) acc0[a][r] = c[r] - acc0[a][r];
    }
#pragma unroll
    for (int a = 0; a <= C1; ++a) {
      d4_t c;
#pragma unroll
      for (int r = 0; r < 4; ++r) c[r] = colp1[min(16 * a + q + 4 * r, nb - 1)];
#pragma unroll
      for (int r = 0; r < 4; ++r) acc1[a][r] = c[r] - acc1[a][r];
    }
  } else {
  // loads: unconditional on clamped addresses (a predicated load is waited for individually), identity padding applied after
#pragma unroll
  for (int a = 0; a <= C0; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc0[a][r] = colp0[min(16 * a + q + 4 * r, nb - 1)];
#pragma unroll
  for (int a = 0; a <= C1; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc1[a][r] = colp1[min(16 * a + q + 4 * r, nb - 1)];
  }
#pragma unroll
  for (int a = 0; a <= C0; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 16 * a + q + 4 * r;
      acc0[a][r] = (i < nb && l0 < nb && i <= l0) ? acc0[a][r] : (i == l0 ? 1.0 : 0.0);
    }
#pragma unroll
  for (int a = 0; a <= C1; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 16 * a + q + 4 * r;
      acc1[a][r] = (i < nb && l1 < nb && i <= l1) ? acc1[a][r] : (i == l1 ? 1.0 : 0.0);
    }
  STAMP(0);
  t4_steps<W>(acc0, acc1, colp0, colp1, l0, l1, nb, nbt, lane, Dt, rinv, Pt, sfail, dflag, Mi, tinv, std::make_integer_sequence<int, 8>{});
  STAMP(5);
}

template <bool HASP>
__global__ __launch_bounds__(256) void potrf_tiles4_kernel(double* __restrict__ A, long lda, long strideA, int n, int k0, int* __restrict__ info,
                                                           double* __restrict__ tinv_base, long tinv_stride, int potrf_prio) {
  extern __shared__ __attribute__((aligned(16))) double pm_lds[];
  if (potrf_prio) __builtin_amdgcn_s_setprio(3);
  double* Dt = pm_lds;                   // factor of the current diagonal tile
  double* Mi = pm_lds + TL;              // its inverse, transposed: [i][n]
  double* rinv = pm_lds + 2 * TL;        // 1 / its diagonal
  double* Pt = pm_lds + 2 * TL + 16;     // solved row panels of three consecutive block steps: tile b of step s at Pt + ((s % 3) 8 + b) TL, [k][m]
  int* sfail = reinterpret_cast<int*>(pm_lds + 2 * TL + 16 + 3 * 8 * TL);
  int* dflag = sfail + 1;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nb = min(NB, n - k0);
  const int nbt = (nb + 15) >> 4;
  double* Ab = A + (long)blockIdx.x * strideA + (long)k0 * lda + k0;
  double* tinv = tinv_base ? tinv_base + (long)blockIdx.x * tinv_stride : nullptr;
  if (tid == 0) { *sfail = 0; *dflag = 0; }
  STAMP(7);
  lds_barrier();
  const double* Pg = HASP ? Ab - NB : nullptr;   // the panel block above: rows k0 - 128 .. k0 - 1 of the same columns
  switch (w) {
    case 0: t4_wave<0, HASP>(Ab, lda, nb, nbt, lane, Dt, rinv, Pt, sfail, dflag, Mi, tinv, Pg); break;
    case 1: t4_wave<1, HASP>(Ab, lda, nb, nbt, lane, Dt, rinv, Pt, sfail, dflag, Mi, tinv, Pg); break;
    case 2: t4_wave<2, HASP>(Ab, lda, nb, nbt, lane, Dt, rinv, Pt, sfail, dflag, Mi, tinv, Pg); break;
    default: t4_wave<3, HASP>(Ab, lda, nb, nbt, lane, Dt, rinv, Pt, sfail, dflag, Mi, tinv, Pg); break;
  }
  lds_barrier();   // (the failure record of the last step is written behind its barrier)
  if (tid == 0 && *sfail && *sfail <= nb) atomicCAS(&info[blockIdx.x], 0, k0 + *sfail);
}

// =============================================================================================
// Panel solve: A12 <- U11^-T A12 (dtrsm 'L','U','T','N'), U11 = the factored 128 x 128 diagonal block, A12 128 x mcols.
// One wavefront per 16 columns, WAVES wavefronts per workgroup share U11 in LDS (36 upper tiles, 77 KB).
// =============================================================================================
__device__ __forceinline__ constexpr int tix(int a, int b) { return a * 8 - a * (a - 1) / 2 + (b - a); }

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void potrf_panel_mfma_kernel(double* __restrict__ A, long lda, long strideA, int k0, int mcols, int coff,
                                                                     const double* __restrict__ tinv_base, long tinv_stride, int potrf_prio) {
  extern __shared__ __attribute__((aligned(16))) double pm_lds[];
  if (potrf_prio) __builtin_amdgcn_s_setprio(3);
  double* Ut = pm_lds;               // 36 tiles
  double* rinv = pm_lds + 36 * TL;   // 128
  double* Mt = pm_lds + 36 * TL + NB;   // tinv: the 8 inverses (transposed) of U11's diagonal tiles, tile j at j TL, [i][n]
  const double* tinv = tinv_base ? tinv_base + (long)blockIdx.y * tinv_stride : nullptr;
  constexpr int THREADS = 64 * WAVES;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, q = lane >> 4, nn = lane & 15;
  double* Ab = A + (long)blockIdx.y * strideA;
  const double* U11 = Ab + (long)k0 * lda + k0;
  double* A12 = Ab + (long)(k0 + NB + coff) * lda + k0;   // (coff: columns of the panel left to the next-block kernel)
  const int c0 = (blockIdx.x * WAVES + wv) * 16;
  const bool active = c0 < mcols;
  double* colp = A12 + (long)min(c0 + nn, mcols - 1) * lda;

  // this wavefront's 128 x 16 slab first (the longest latency), then U11 -> LDS: ALL loads of a half are issued before
  // the first LDS store waits for one (a load-store pair per tile row serialises on the memory latency)
  STAMP(10);
  d4_t acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[i][r] = colp[16 * i + q + 4 * r];
  {
    constexpr int PER = 256 / THREADS;   // elements of a tile per thread
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      double v[18][PER];
#pragma unroll
      for (int t = 0; t < 18; ++t) {
        const int ti = 18 * half + t;
        // tile index -> (a, b), row-major over the upper triangle
        int a = 0, rem = ti;
#pragma unroll
        for (int aa = 0; aa < 8; ++aa)
          if (rem >= 8 - aa && a == aa) { rem -= 8 - aa; a = aa + 1; }
        const int b = a + rem;
#pragma unroll
        for (int p = 0; p < PER; ++p) {
          const int e = tid + THREADS * p, k = e & 15, m = e >> 4;
          v[t][p] = U11[(long)(16 * b + m) * lda + 16 * a + k];
          if (a == b && k > m) v[t][p] = 0.0;   // (below the diagonal of a diagonal tile: whatever the matrix holds there; the tile is an MFMA operand as a whole)
        }
      }
#pragma unroll
      for (int t = 0; t < 18; ++t) {
        const int ti = 18 * half + t;
#pragma unroll
        for (int p = 0; p < PER; ++p) {
          const int e = tid + THREADS * p, k = e & 15, m = e >> 4;
          Ut[ti * TL + k * TS + m] = v[t][p];
        }
      }
    }
  }
  if (tinv != nullptr) {
    for (int e = tid; e < 8 * 256; e += THREADS) Mt[(e >> 8) * TL + (e & 15) * TS + ((e >> 4) & 15)] = tinv[e];   // record: (i, n) at 16 n + i; LDS: [i][n]
  }
  __syncthreads();
  STAMP(11);
  for (int t = tid; t < NB; t += THREADS) rinv[t] = 1.0 / Ut[tix(t >> 4, t >> 4) * TL + (t & 15) * TS + (t & 15)];
  __syncthreads();
  if (!active) return;
  STAMP(12);

  const bool inb = c0 + nn < mcols;
  if (tinv != nullptr) {   // the tile solves as products with the diagonal tiles' inverses (tile_solve_mfma); same update loop
#pragma unroll
    for (int jb = 0; jb < 8; ++jb) {
      double ma[4], ua[4];
      tile_inv_operands(Mt + jb * TL, Ut + tix(jb, jb) * TL, q, nn, ma, ua);
      d4_t bt[1] = {acc[jb]};
      tile_solve_mfma<1>(bt, 1, ma, ua);
      acc[jb] = bt[0];
      if (inb) {
        double* dst = colp + 16 * jb + q;
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[4 * r] = bt[0][r];
      }
      const d4_t c = acc[jb];
#pragma unroll
      for (int kc = 0; kc < 4; ++kc)
#pragma unroll
        for (int i = jb + 1; i < 8; ++i) acc[i] = mfma4(Ut[tix(jb, i) * TL + (4 * kc + q) * TS + nn], -c[kc], acc[i]);
    }
    return;
  }
#pragma unroll
  for (int jb = 0; jb < 8; ++jb) {
    if (jb == 1) STAMP(13);
    double x[16];
    tile_gather(acc[jb], nn, x);
    tile_subst(x, Ut + tix(jb, jb) * TL, rinv + 16 * jb);
    tile_scatter(x, q, acc[jb]);
    if (inb) {   // rows 16 jb .. 16 jb + 15 of this column are final: 32 contiguous bytes per lane
      double v[4];
      tile_rows4(x, q, v);
      double* dst = colp + 16 * jb + 4 * q;
      *reinterpret_cast<d2_t*>(dst) = (d2_t){v[0], v[1]};
      *reinterpret_cast<d2_t*>(dst + 2) = (d2_t){v[2], v[3]};
    }
    const d4_t c = acc[jb];
#pragma unroll
    for (int kc = 0; kc < 4; ++kc)
#pragma unroll
      for (int i = jb + 1; i < 8; ++i) acc[i] = mfma4(Ut[tix(jb, i) * TL + (4 * kc + q) * TS + nn], -c[kc], acc[i]);
  }
  STAMP(14);
  STAMP(15);
}

// ---------------------------------------------------------------------------------------------
// launchers (same contracts as potrf_diag_launch(factor only) / potrf_panel_solve_launch of potrf_diag.hip)
// ---------------------------------------------------------------------------------------------
// Round 5 (profiles/r05_probe_interference.txt): a latency-bound wavefront that shares a SIMD with a wavefront issuing FP64 MFMAs back
// to back is starved by the instruction arbiter (dependent v_fma_f64 56 - 1300x slower, LDS + barrier loops 50 - 240x); with wavefront
// priority 3 (s_setprio) it is served between the MFMAs (2.6 - 37x better).  HYP_POTRF_PRIO=1 raises the priority of the kernels on the
// factorization's critical path -- the diagonal block, the panel, the look-ahead update --, which run beside the helper stream's
// trailing update.  Measured: n = 5000 2.91 ms with it, 2.89 without (profiles/r05_potrf_prio.txt) -- the real trailing GEMM leaves
// the arbiter gaps a pure MFMA loop does not, and what slows these kernels down beside it is the MEMORY path (a dependent load takes
// 4 - 6x longer beside an HBM stream even from an otherwise idle CU), which no priority reaches.  Default off.
int potrf_hiprio() {
  static const int on = [] { const char* e = getenv("HYP_POTRF_PRIO"); return (e && atoi(e) == 1) ? 1 : 0; }();
  return on;
}
static bool potrf_la_on() {
  static const bool on = [] { const char* e = getenv("HYP_POTRF_LA"); return !(e && atoi(e) == 0); }();
  return on;
}
bool potrf_tinv_on() {
  static const bool on = [] { const char* e = getenv("HYP_POTRF_TINV"); return !(e && atoi(e) == 0); }();
  return on && potrf_la_on();
}
bool potrf_diag_prev_ok() {   // the diagonal-block kernel that can take the previous step's update of its block (prev_update below)
  static const bool defer = [] { const char* e = getenv("HYP_POTRF_DEFER"); return !(e && atoi(e) == 0); }();
  return potrf_la_on() && potrf_tinv_on() && defer;
}
void potrf_diag_mfma_launch(hipStream_t st, int batch, double* A, long lda, long strideA, int n, int k0, int* info, int own_cu_lds, double* tinv,
                            long tinv_stride, bool prev_update) {
  if (prev_update && !(potrf_diag_prev_ok() && k0 >= NB)) {
    fprintf(stderr, "potrf_diag_mfma_launch: prev_update needs the tiles4 kernel and a full panel block above\n");
    abort();
  }
  if (potrf_la_on()) {
    const size_t lds = (size_t)(TL + 16 + 2 * 8 * TL + 2 + TL) * sizeof(double);
    const size_t want = own_cu_lds > 0 ? std::max<size_t>(lds, (size_t)own_cu_lds) : lds;
    static const bool defer = [] { const char* e = getenv("HYP_POTRF_DEFER"); return !(e && atoi(e) == 0); }();
    if (potrf_tinv_on() && defer) {
      const size_t lds4 = (size_t)(2 * TL + 16 + 3 * 8 * TL + 2) * sizeof(double);
      const size_t l4 = own_cu_lds > 0 ? std::max<size_t>(lds4, (size_t)own_cu_lds) : lds4;
      if (prev_update) hipLaunchKernelGGL(potrf_tiles4_kernel<true>, dim3(batch), dim3(256), l4, st, A, lda, strideA, n, k0, info, tinv, tinv_stride, potrf_hiprio());
      else hipLaunchKernelGGL(potrf_tiles4_kernel<false>, dim3(batch), dim3(256), l4, st, A, lda, strideA, n, k0, info, tinv, tinv_stride, potrf_hiprio());
    } else if (potrf_tinv_on()) hipLaunchKernelGGL((potrf_tiles_kernel<4, 8, true>), dim3(batch), dim3(256), want, st, A, lda, strideA, n, k0, info, tinv, tinv_stride);
    else hipLaunchKernelGGL((potrf_tiles_kernel<4, 8, false>), dim3(batch), dim3(256), want, st, A, lda, strideA, n, k0, info, nullptr, 0L);
  } else {
    const size_t lds = (size_t)(TL + 16 + 8 * TL + 2) * sizeof(double);
    const size_t want = own_cu_lds > 0 ? std::max<size_t>(lds, (size_t)own_cu_lds) : lds;
    hipLaunchKernelGGL(potrf_diag_mfma_kernel, dim3(batch), dim3(256), want, st, A, lda, strideA, n, k0, info);
  }
  HYP_CHECK(hipGetLastError());
}
int potrf_diag_mfma_own_cu_lds() {
  const int want = 124 * 1024;
  hipError_t e = !potrf_la_on() ? hipFuncSetAttribute(reinterpret_cast<const void*>(&potrf_diag_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, want)
                 : potrf_tinv_on() ? ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(&potrf_tiles4_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, want),
                                      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&potrf_tiles4_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, want),
                                      hipFuncSetAttribute(reinterpret_cast<const void*>(&potrf_tiles_kernel<4, 8, true>), hipFuncAttributeMaxDynamicSharedMemorySize, want))
                                   : hipFuncSetAttribute(reinterpret_cast<const void*>(&potrf_tiles_kernel<4, 8, false>), hipFuncAttributeMaxDynamicSharedMemorySize, want);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return want;
}
void potrf_panel_mfma_launch(hipStream_t st, int batch, double* A, long lda, long strideA, int k0, int mcols, int coff, const double* tinv,
                             long tinv_stride) {
  if (mcols <= 0) return;
  const size_t lds = (size_t)(36 * TL + NB + 8 * TL) * sizeof(double);
  static const int waves = [] { const char* e = getenv("HYP_PANEL_WAVES"); const int v = e ? atoi(e) : 2; return (v == 1 || v == 2 || v == 4) ? v : 2; }();
  static bool attr_set = false;
  if (!attr_set) {
    HYP_CHECK(hipFuncSetAttribute((const void*)potrf_panel_mfma_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HYP_CHECK(hipFuncSetAttribute((const void*)potrf_panel_mfma_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HYP_CHECK(hipFuncSetAttribute((const void*)potrf_panel_mfma_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  const dim3 grid((mcols + 16 * waves - 1) / (16 * waves), batch);
  if (waves == 1) hipLaunchKernelGGL(potrf_panel_mfma_kernel<1>, grid, dim3(64), lds, st, A, lda, strideA, k0, mcols, coff, tinv, tinv_stride, potrf_hiprio());
  else if (waves == 2) hipLaunchKernelGGL(potrf_panel_mfma_kernel<2>, grid, dim3(128), lds, st, A, lda, strideA, k0, mcols, coff, tinv, tinv_stride, potrf_hiprio());
  else hipLaunchKernelGGL(potrf_panel_mfma_kernel<4>, grid, dim3(256), lds, st, A, lda, strideA, k0, mcols, coff, tinv, tinv_stride, potrf_hiprio());
  HYP_CHECK(hipGetLastError());
}

}  // namespace hyp
