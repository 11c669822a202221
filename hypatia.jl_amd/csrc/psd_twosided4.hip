// Fourth form of the two-sided product W_j = R' V_j R with an upper triangular R (PosSemidefTri's sqrt_hess_prod!,
// /root/reference/src/Cones/possemideftri.jl:161-177, on the q x n block of G: qrchol.jl:219-233) for sides of 9 .. 13 MFMA tiles
// (129 .. 208; config 2: 200): the intermediate Z_j = V_j R never leaves the CU's REGISTERS.
//
// What held the two-pass kernels (psd_ts3_kernel) at 0.42 MFMA-busy was measured in round 4 (profiles/r04_pmc_ts3.json): 55 - 74 MFMAs
// per wavefront in a 36 K-cycle life, a barrier per 16 MFMAs, and Z written to and read back from HBM (3.4 GB per config-2
// iteration).  At side 200 Z is 169 tiles of 16 x 16 = 346 KB: too large for the LDS (160 KB) but not for the register file (512 KB
// per CU).  Here eight wavefronts (two per SIMD, 256 registers each) hold block columns of Z as MFMA ACCUMULATORS, one block column
// (13 tiles = 104 registers) per wavefront: the D layout of v_mfma_f64_16x16x4 (lane l, register r: D[(l >> 4) + 4 r, l & 15]) is
// exactly the layout of its B operand for k-chunk r (lane l: B[l >> 4, l & 15]), so the second product reads the accumulators of the
// first as operands -- no shuffle, no LDS round trip.  The register file bounds a workgroup at 8 block columns, so a matrix is TWO work
// items: each takes a set of at most 8 block columns of Z (both gather V; the second finds it in the L2).  (First form: four
// wavefronts with two block columns each, 256 accumulation + 250 vector registers -- the same code with NW = 4: 2.08 against 2.02 ms.)
//
// R upper triangular: block column a of Z needs the K slices t <= a of V (Z[:, a] = sum_{k <= a} V[:, k] R[k, a]) and is final after
// step t = a; tile row c of the result W[c, a] = sum_{k <= c} R[k, c]' Z[k, a], a <= c, needs the block columns a <= c.  So ONE
// loop over t = 0 .. T - 1 does both: step t stages the K slice t of V (gathered from the svec column, off-diagonals / sqrt(2)), the
// row slice R[16 t .. , :] and the column slice R[:, 16 t ..] into the LDS; every wavefront then updates its block columns a >= t
// (13 x 4 MFMAs each) and computes the result tiles W[t, a] of its block columns a <= t ((t + 1) x 4 MFMAs each) -- a column is
// either still accumulating or already producing, so the SIMDs stay busy in every step (column sets and SIMD pairs below: exhaustive
// search for the shortest sum over t of the busiest SIMD, 0.79 - 0.83 of the even split).  Two barriers per step; operands of
// step t + 1 are in flight in registers while step t multiplies.  A workgroup walks the items i = blockIdx.x, + gridDim.x, ...;
// the first slices of its next item are requested during the last step of the current one.
// The result is the upper triangle, written as the packed svec column (off-diagonals * sqrt(2)) straight from the accumulators;
// the summation order of every entry is that of the two-pass kernels (k ascending).
#include <cstdio>
#include <type_traits>
#include <utility>
#include <vector>

#include "cones.hpp"
#include "gemm_f64.hpp"
#include "hyp_internal.hpp"

namespace hyp {

namespace {

// Fragment reads are VOLATILE loads from the LDS address space (round 6): left to itself the compiler pairs the reads of two k chunks into
// ds_read2_b64 -- half the rate of ds_read_b64 and banked modulo 32 dwords, where rows j and j + 8 of the 18-double row stride collide.  This
// kernel reads one fragment per one or two MFMAs: at that ratio the paired, conflicting reads keep the LDS busy for as long as the matrix
// cores (EXPERIMENTS r06-16).  -DHYP_TS4_READ2: the plain loads of rounds 4-5, for A/B builds.
#ifdef HYP_TS4_READ2
#define TS4_LDS_RD(P) (*(P))
#else
typedef const volatile double __attribute__((address_space(3))) ts4_lds_cv;
#define TS4_LDS_RD(P) (*(ts4_lds_cv*)(P))
#endif
constexpr int T4_LDK = 18;   // LDS row stride of the 16-wide slices (conflict-free 64-bit fragment reads, as psd_twosided.hip)

struct Ts4Args {
  int s, ncols;
  const double* A;     // svec columns
  long lda;
  const double* Rp;    // R zero-padded to 16 T x 16 T, Rp[c * LD + k] = R[k, c]
  double* C;           // svec columns of the result
  long ldc;
  unsigned long long* probe;   // HYP_TS4_PROBE: per workgroup and wavefront 8 cycle sums (see ts4_wave)
};

// the two column sets of a matrix and, per set and SIMD, the block columns of its two wavefronts W and W + 4 (-1: none)
template <int T> struct Ts4Plan;
template <> struct Ts4Plan<13> { static constexpr int cols[2][4][2] = {{{0, 5}, {3, 6}, {4, 7}, {8, 9}}, {{1, 2}, {10, -1}, {11, -1}, {12, -1}}}; };
template <> struct Ts4Plan<12> { static constexpr int cols[2][4][2] = {{{0, 4}, {1, 5}, {2, 6}, {3, 7}}, {{8, -1}, {9, -1}, {10, -1}, {11, -1}}}; };
template <> struct Ts4Plan<11> { static constexpr int cols[2][4][2] = {{{0, 3}, {1, 4}, {2, 5}, {10, -1}}, {{6, -1}, {7, -1}, {8, -1}, {9, -1}}}; };
template <> struct Ts4Plan<10> { static constexpr int cols[2][4][2] = {{{0, 2}, {1, 3}, {8, -1}, {9, -1}}, {{4, -1}, {5, -1}, {6, -1}, {7, -1}}}; };
template <> struct Ts4Plan<9> { static constexpr int cols[2][4][2] = {{{0, 1}, {6, -1}, {7, -1}, {8, -1}}, {{2, -1}, {3, -1}, {4, -1}, {5, -1}}}; };
// round 5: 6 .. 8 tiles (sides 81 .. 128) fit ONE column set -- at most eight block columns, one per wavefront; the second set is empty
// and not launched (sides <= 80 have psd_ts5_kernel: one wavefront holds all of Z)
template <> struct Ts4Plan<8> { static constexpr int cols[2][4][2] = {{{0, 7}, {1, 6}, {2, 5}, {3, 4}}, {{-1, -1}, {-1, -1}, {-1, -1}, {-1, -1}}}; };
template <> struct Ts4Plan<7> { static constexpr int cols[2][4][2] = {{{0, 6}, {1, 5}, {2, 4}, {3, -1}}, {{-1, -1}, {-1, -1}, {-1, -1}, {-1, -1}}}; };
template <> struct Ts4Plan<6> { static constexpr int cols[2][4][2] = {{{0, 3}, {1, 2}, {4, -1}, {5, -1}}, {{-1, -1}, {-1, -1}, {-1, -1}, {-1, -1}}}; };
template <int T> constexpr bool ts4_two_sets() { return Ts4Plan<T>::cols[1][0][0] >= 0 || Ts4Plan<T>::cols[1][1][0] >= 0; }

template <int T> constexpr int ts4_ldc() { return ((16 * T - 18 + 31) / 32) * 32 + 18; }   // = 18 mod 32, >= 16 T
template <int T> constexpr size_t ts4_lds_doubles() { return (size_t)2 * 16 * T * T4_LDK + (size_t)16 * ts4_ldc<T>(); }

__device__ __forceinline__ double ts4_div_rt2(double x) {   // x / sqrt(2) correctly rounded (see div_rt2 in psd_twosided.hip)
  const double d = 1.4142135623730951, c = 0.70710678118654746;
  const double q = x * c;
  const double r = fma(-q, d, x);
  return fma(r, c, q);
}

constexpr int ts4_max2(int a, int b) { return a > b ? a : b; }

template <class F, int... I>
__device__ __forceinline__ void ts4_steps(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}

// One wavefront's program: H = the column set, W = its index (compile time: which block columns it owns), the step loop fully
// unrolled through integral constants -- every register index, every activity test and every LDS offset is a constant.  (A first
// form kept t and the column indices in scalar registers and guarded the MFMAs with wavefront-uniform branches: a quarter of the
// code, but the accumulators then flow through phi copies at every guard -- 5600 v_mov_b64 and 3000 AGPR moves in the listing, 435
// registers spilled.)
// NW = 4: one wavefront per SIMD with two block columns (512 registers each).  NW = 8: two wavefronts per SIMD with ONE block column
// each (256 registers each; wavefronts W and W + 4 share a SIMD and take the two columns the NW = 4 plan gives to wavefront W): the
// staging, the barrier waits and the fragment latencies of one run under the MFMAs of the other.  The staged tiles are dealt out
// by parity to the two halves of the workgroup (HALF = W >> 2).
template <int T, int H, int NW, int W>
__device__ __forceinline__ void ts4_wave(const Ts4Args& p, double* __restrict__ Vs, double* __restrict__ Rr, double* __restrict__ Rc) {
  constexpr int LD = 16 * T;
  constexpr int LDC = ts4_ldc<T>();
  constexpr int NH = NW / 4, HALF = (NW == 8) ? (W >> 2) : 0;
  constexpr int A0 = (NW == 8) ? Ts4Plan<T>::cols[H][W & 3][W >> 2] : Ts4Plan<T>::cols[H][W & 3][0];
  constexpr int A1 = (NW == 8) ? -1 : Ts4Plan<T>::cols[H][W & 3][1];
  constexpr int AC[2] = {A0, A1};
  constexpr int NST = (T + NH - 1) / NH;     // staged tiles per thread
  constexpr int NC = (NW == 8) ? 1 : 2;     // block columns per wavefront
  // what the whole workgroup (all four wavefronts of this column set) needs of a step
  constexpr int SET_MAX = ts4_max2(ts4_max2(ts4_max2(Ts4Plan<T>::cols[H][0][0], Ts4Plan<T>::cols[H][0][1]), ts4_max2(Ts4Plan<T>::cols[H][1][0], Ts4Plan<T>::cols[H][1][1])),
                                   ts4_max2(ts4_max2(Ts4Plan<T>::cols[H][2][0], Ts4Plan<T>::cols[H][2][1]), ts4_max2(Ts4Plan<T>::cols[H][3][0], Ts4Plan<T>::cols[H][3][1])));
  const int tid = threadIdx.x, lane = tid & 63;
  const int fj = lane & 15, fq = lane >> 4;
  const int x0 = tid & 15, y0 = (tid & 255) >> 4;
  int x = x0, y = y0;   // (re-defined opaquely in every step: see there)
  const int s = p.s, sm1 = s - 1;
  const double* __restrict__ Rp = p.Rp;

  double vst[NST], rst[NST];   // staged slices of the coming step (one 16 x 16 tile per rep and thread element)
  d4_t Z[NC][T];
  const d4_t zero4 = (d4_t){0.0, 0.0, 0.0, 0.0};

  // (V's K slice and R's row slice are only read by the first product, i.e. while t <= SET_MAX)
// (index clamps only where a tile meets the edge -- the last tile row / K block; 24-bit multiplies: every index is below 2^15)
// (32-bit BYTE offsets from a uniform base: the form global_load takes as scalar base + vector offset -- with element indices the
//  compiler built a 64-bit address per load, two 64-bit vector operations each, and vector operations wait for the FP64 MFMAs)
#define TS4_TRI(v) (__umul24((unsigned)(v), (unsigned)(v) + 1u) >> 1)
#define TS4_AT(BASE, IDX) (*reinterpret_cast<const double*>(reinterpret_cast<const char*>(BASE) + (size_t)(unsigned)((IDX) * 8u)))
#define TS4_LOAD_REP(AJ, TT, rep)                                                                                           \
  if (NH == 1 || ((rep) & 1) == HALF) {                                                                                     \
    if ((rep) < (TT)) { /* m = 16 rep + x above the K block: vec[k (k + 1) / 2 + m], k = 16 t + y */                         \
      const int k = ((TT) == T - 1) ? min(16 * (TT) + y, sm1) : 16 * (TT) + y;                                              \
      if ((TT) <= SET_MAX) vst[(rep) / NH] = TS4_AT(AJ, TS4_TRI(k) + (unsigned)(16 * (rep) + x));                                  \
      rst[(rep) / NH] = TS4_AT(Rp, (unsigned)((16 * (TT) + y) * LD + 16 * (rep) + x)); /* column slice: R[16 rep + x, 16 t + y] */  \
    } else if ((rep) > (TT)) { /* m = 16 rep + y below: vec[m (m + 1) / 2 + k], k = 16 t + x */                              \
      const int m = ((rep) == T - 1) ? min(16 * (rep) + y, sm1) : 16 * (rep) + y;                                           \
      if ((TT) <= SET_MAX) {                                                                                                \
        vst[(rep) / NH] = TS4_AT(AJ, TS4_TRI(m) + (unsigned)(16 * (TT) + x));                                                      \
        rst[(rep) / NH] = TS4_AT(Rp, (unsigned)((16 * (rep) + y) * LD + 16 * (TT) + x)); /* row slice: R[16 t + x, 16 rep + y] */   \
      }                                                                                                                     \
    } else {                                                                                                                \
      const int m = ((TT) == T - 1) ? min(16 * (TT) + y, sm1) : 16 * (TT) + y;                                              \
      const int k = ((TT) == T - 1) ? min(16 * (TT) + x, sm1) : 16 * (TT) + x;                                              \
      const int lo = min(m, k), hi = max(m, k);                                                                             \
      if ((TT) <= SET_MAX) vst[(rep) / NH] = TS4_AT(AJ, TS4_TRI(hi) + (unsigned)lo);                                               \
      rst[(rep) / NH] = TS4_AT(Rp, (unsigned)((16 * (TT) + y) * LD + 16 * (TT) + x)); /* the diagonal block: member of both slices */ \
    }                                                                                                                       \
  }
#define TS4_LOAD_STAGE(JJ, TT)                                                                                              \
  {                                                                                                                         \
    const double* __restrict__ Aj_ = p.A + (JJ) * p.lda;                                                                    \
    _Pragma("unroll") for (int rep = 0; rep < T; ++rep) TS4_LOAD_REP(Aj_, TT, rep)                                          \
  }

  // HYP_TS4_PROBE: cycles per phase, summed over steps and matrices ([0] wait at the first barrier, [1] registers -> LDS, [2] requests
  // of the next slices, [3] second barrier, [4] first product, [5] second product + stores)
  unsigned long long pc[6] = {0, 0, 0, 0, 0, 0};
  unsigned long long tprev = 0;
#define TS4_STAMP(I)                                                             \
  if (p.probe) {                                                                 \
    const unsigned long long now = __builtin_amdgcn_s_memtime();                 \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                           \
    pc[I] += now - tprev;                                                        \
    tprev = now;                                                                 \
  }
  long j = blockIdx.x;
  if (j < p.ncols) TS4_LOAD_STAGE(j, 0)
  if (p.probe) { tprev = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
  for (; j < p.ncols; j += gridDim.x) {
#pragma unroll
    for (int ci = 0; ci < NC; ++ci)
#pragma unroll
      for (int m = 0; m < T; ++m) Z[ci][m] = zero4;
    double* __restrict__ Cj = p.C + j * p.ldc;
    auto step = [&](auto tc) __attribute__((always_inline)) {
      constexpr int t = decltype(tc)::value;
      // (the index arithmetic of a step does not depend on the matrix: left visible, the compiler hoists all T x T steps' worth of
      //  offsets and edge masks out of the matrix loop and keeps them in registers -- 2750 spilled)
      x = x0; y = y0;
      asm volatile("" : "+v"(x), "+v"(y));
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");        // every wavefront is done with the slices of step t - 1
      TS4_STAMP(0)
      // ---- staged registers -> LDS ----
#pragma unroll
      for (int rep = 0; rep < T; ++rep) {
        if (NH == 2 && (rep & 1) != HALF) continue;
        if (rep < t) {
          if constexpr (t <= SET_MAX) {
            double v = ts4_div_rt2(vst[rep / NH]);
            if (t == T - 1) v = (16 * t + y < s) ? v : 0.0;                    // (only the last tiles meet the edge)
            Vs[(16 * rep + x) * T4_LDK + y] = v;
          }
          Rc[y * LDC + 16 * rep + x] = rst[rep / NH];
        } else if (rep > t) {
          if constexpr (t <= SET_MAX) {
            double v = ts4_div_rt2(vst[rep / NH]);
            if (rep == T - 1) v = (16 * rep + y < s) ? v : 0.0;
            Vs[(16 * rep + y) * T4_LDK + x] = v;
            Rr[(16 * rep + y) * T4_LDK + x] = rst[rep / NH];
          }
        } else {
          if constexpr (t <= SET_MAX) {
            double v = ts4_div_rt2(vst[rep / NH]);
            v = (x == y) ? vst[rep / NH] : v;
            if (rep == T - 1) v = (16 * t + y < s && 16 * t + x < s) ? v : 0.0;
            Vs[(16 * t + y) * T4_LDK + x] = v;
            Rr[(16 * t + y) * T4_LDK + x] = rst[rep / NH];
          }
          Rc[y * LDC + 16 * t + x] = rst[rep / NH];
        }
      }
      TS4_STAMP(1)
      // the slices of the next step (of the next matrix after the last step; past the last matrix: the last one again, unused) are
      // requested piece by piece BETWEEN the MFMAs below: issued as one block they took 1600 cycles per step (HYP_TS4_PROBE), the
      // memory pipeline taking 26 requests of four wavefronts at its own pace with the matrix cores idle
      constexpr int TN = (t + 1 < T) ? t + 1 : 0;
      const long jn = (t + 1 < T) ? j : min(j + (long)gridDim.x, (long)p.ncols - 1);
      const double* __restrict__ An = p.A + jn * p.lda;
      TS4_STAMP(2)
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      TS4_STAMP(3)

      // ---- first product: Z[:, a] += V[:, K slice t] R[K slice t, a] for the block columns a >= t ----
      constexpr bool p1[2] = {t <= A0, t <= A1};
      if constexpr (A0 < 0 && A1 < 0) {   // (a wavefront without a block column in this column set only stages)
#pragma unroll
        for (int rep = 0; rep < T; ++rep) TS4_LOAD_REP(An, TN, rep)
      }
      if constexpr (p1[0] || p1[1]) {
        double bf[2][4];
#pragma unroll
        for (int ci = 0; ci < NC; ++ci)
          if (p1[ci]) {
            const double* bs = Rr + (16 * AC[ci] + fj) * T4_LDK + fq;
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) bf[ci][ch] = TS4_LDS_RD(bs + 4 * ch);
          }
        // Tile rows in pairs: consecutive MFMAs go to different accumulators even when only one block column is active (a dependent
        // FP64 MFMA waits ~95 cycles for its predecessor, an independent one issues after 64: tools/probe_potrf.hip).  The fragments
        // of the next pair are requested before the MFMAs of the current one; the scheduling barriers keep the compiler from
        // hoisting ALL rows' fragment reads to the top.
        constexpr int NP = (T + 1) / 2;
        double afn[2][4];
#pragma unroll
        for (int mm = 0; mm < 2; ++mm) {
          const double* as = Vs + (16 * mm + fj) * T4_LDK + fq;
#pragma unroll
          for (int ch = 0; ch < 4; ++ch) afn[mm][ch] = TS4_LDS_RD(as + 4 * ch);
        }
#pragma unroll
        for (int mp = 0; mp < NP; ++mp) {
          TS4_LOAD_REP(An, TN, 2 * mp)
          if (2 * mp + 1 < T) TS4_LOAD_REP(An, TN, 2 * mp + 1)
          double af[2][4];
#pragma unroll
          for (int mm = 0; mm < 2; ++mm)
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) af[mm][ch] = afn[mm][ch];
          if (mp + 1 < NP) {
#pragma unroll
            for (int mm = 0; mm < 2; ++mm) {
              if (2 * (mp + 1) + mm >= T) continue;
              const double* as = Vs + (16 * (2 * (mp + 1) + mm) + fj) * T4_LDK + fq;
#pragma unroll
              for (int ch = 0; ch < 4; ++ch) afn[mm][ch] = TS4_LDS_RD(as + 4 * ch);
            }
          }
#pragma unroll
          for (int ch = 0; ch < 4; ++ch) {
#pragma unroll
            for (int mm = 0; mm < 2; ++mm) {
              if (2 * mp + mm >= T) continue;
#pragma unroll
              for (int ci = 0; ci < NC; ++ci)
                if (p1[ci]) Z[ci][2 * mp + mm] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[mm][ch], bf[ci][ch], Z[ci][2 * mp + mm], 0, 0, 0);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }

      TS4_STAMP(4)
      // ---- second product: W[a, c = t] = sum_{k <= t} Z[k, a]' R[k, t] for the block columns a <= t (D[i = c, j = a]) ----
      constexpr bool p2[2] = {A0 >= 0 && t >= A0, A1 >= 0 && t >= A1};
      if constexpr (p2[0] || p2[1]) {
        // (two partial sums per tile, the k-chunks of even and of odd index: independent accumulators for consecutive MFMAs, see above;
        //  added at the end)
        d4_t Y[2][2] = {{zero4, zero4}, {zero4, zero4}};
        double rfn[4];
        {
          const double* rs = Rc + fj * LDC + fq;
#pragma unroll
          for (int ch = 0; ch < 4; ++ch) rfn[ch] = TS4_LDS_RD(rs + 4 * ch);
        }
#pragma unroll
        for (int kt = 0; kt <= t; ++kt) {
          if constexpr (!(p1[0] || p1[1])) {   // (no first product in this step: the requests ride here, T of them over t + 1 rounds)
#pragma unroll
            for (int rep = (kt * T) / (t + 1); rep < ((kt + 1) * T) / (t + 1); ++rep) TS4_LOAD_REP(An, TN, rep)
          }
          double rf[4];
#pragma unroll
          for (int ch = 0; ch < 4; ++ch) rf[ch] = rfn[ch];
          if (kt + 1 <= t) {
            const double* rs = Rc + fj * LDC + 16 * (kt + 1) + fq;
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) rfn[ch] = TS4_LDS_RD(rs + 4 * ch);
          }
#pragma unroll
          for (int ch = 0; ch < 4; ++ch) {
#pragma unroll
            for (int ci = 0; ci < NC; ++ci)
              if (p2[ci]) Y[ci][ch & 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(rf[ch], Z[ci][kt][ch], Y[ci][ch & 1], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        // register r of a result tile: column c = 16 t + fq + 4 r of W, row 16 a + fj.  Only the diagonal tile (a = t) has entries
        // below the diagonal to skip and diagonal entries to leave unscaled, only the last tile column can pass the edge: everywhere
        // else the stores are unconditional (vector instructions between the MFMAs are MFMA time lost)
#pragma unroll
        for (int ci = 0; ci < NC; ++ci) {
          if (!p2[ci]) continue;
          const int row = 16 * AC[ci] + fj;
          const int col0 = 16 * t + fq;
          unsigned off = (TS4_TRI(col0) + (unsigned)row) * 8u;   // tri(c + 4) = tri(c) + 4 c + 10
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int col = col0 + 4 * r;
            const double v = Y[ci][0][r] + Y[ci][1][r];
            double* dst = reinterpret_cast<double*>(reinterpret_cast<char*>(Cj) + (size_t)off);
            if (AC[ci] == t) {
              if (row <= col && (t < T - 1 || col < s)) *dst = (row == col) ? v : v * 1.4142135623730951;   // mat[i, j] * rt2 (arrayutilities.jl:176)
            } else {
              if (t < T - 1 || col < s) *dst = v * 1.4142135623730951;
            }
            off += (unsigned)(4 * col + 10) * 8u;
          }
        }
      }
      TS4_STAMP(5)
    };
    ts4_steps(step, std::make_integer_sequence<int, T>{});
  }
  if (p.probe && lane == 0) {
#pragma unroll
    for (int i = 0; i < 6; ++i) p.probe[((long)blockIdx.x * 8 + W) * 8 + i] = pc[i];
  }
#undef TS4_STAMP
#undef TS4_LOAD_STAGE
#undef TS4_LOAD_REP
#undef TS4_AT
#undef TS4_TRI
}

template <int T, int H, int NW>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(NW / 4, NW / 4))) void psd_ts4_kernel(Ts4Args p) {
  extern __shared__ __attribute__((aligned(16))) double ts4_lds[];
  double* Vs = ts4_lds;                             // [m][k16]   V[m, 16 t + k]
  double* Rr = Vs + 16 * T * T4_LDK;                // [a][k16]   R[16 t + k, a]
  double* Rc = Rr + 16 * T * T4_LDK;                // [c16][k]   R[k, 16 t + c]
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wave == 0) ts4_wave<T, H, NW, 0>(p, Vs, Rr, Rc);
  else if (wave == 1) ts4_wave<T, H, NW, 1>(p, Vs, Rr, Rc);
  else if (wave == 2) ts4_wave<T, H, NW, 2>(p, Vs, Rr, Rc);
  else if (wave == 3) ts4_wave<T, H, NW, 3>(p, Vs, Rr, Rc);
  else if constexpr (NW == 8) {
    if (wave == 4) ts4_wave<T, H, NW, 4>(p, Vs, Rr, Rc);
    else if (wave == 5) ts4_wave<T, H, NW, 5>(p, Vs, Rr, Rc);
    else if (wave == 6) ts4_wave<T, H, NW, 6>(p, Vs, Rr, Rc);
    else ts4_wave<T, H, NW, 7>(p, Vs, Rr, Rc);
  }
}

template <int T, int H, int NW>
void ts4_launch_set(Ctx& c, const Ts4Args& a, int grid, size_t lds) {
  static bool attr_set = false;
  if (!attr_set) {
    HYP_CHECK(hipFuncSetAttribute((const void*)psd_ts4_kernel<T, H, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  hipLaunchKernelGGL((psd_ts4_kernel<T, H, NW>), dim3(grid), dim3(64 * NW), lds, c.stream, a);
  HYP_CHECK(hipGetLastError());
}

template <int T, int NW>
void ts4_launch_nw(Ctx& c, Ts4Args a) {
  const size_t lds = ts4_lds_doubles<T>() * sizeof(double);
  static const int cus = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n > 0 ? n : 256;
  }();
  const int grid = std::min(a.ncols, cus);
  static const bool probe = [] { const char* e = getenv("HYP_TS4_PROBE"); return e && e[0] == '1'; }();
  if (!probe) {
    ts4_launch_set<T, 0, NW>(c, a, grid, lds);   // (the two column sets write disjoint tile rows of the result)
    if constexpr (ts4_two_sets<T>()) ts4_launch_set<T, 1, NW>(c, a, grid, lds);
    return;
  }
  DBuf pb((size_t)grid * 8 * 8 * sizeof(unsigned long long));
  std::vector<unsigned long long> h((size_t)grid * 64);
  for (int set = 0; set < (ts4_two_sets<T>() ? 2 : 1); ++set) {
    c.zero(pb.p, pb.bytes);
    a.probe = (unsigned long long*)pb.p;
    if (set == 0) ts4_launch_set<T, 0, NW>(c, a, grid, lds);
    else if constexpr (ts4_two_sets<T>()) ts4_launch_set<T, 1, NW>(c, a, grid, lds);
    c.d2h(h.data(), pb.p, pb.bytes);
    c.sync();
    for (int w = 0; w < NW; ++w) {
      double sum[6] = {0, 0, 0, 0, 0, 0};
      for (int g = 0; g < grid; ++g)
        for (int i = 0; i < 6; ++i) sum[i] += (double)h[((size_t)g * 8 + w) * 8 + i];
      fprintf(stderr, "[ts4 probe] T=%d set %d wavefront %d of %d, mean s_memtime ticks per workgroup: barrier-1 %.0f  regs->LDS %.0f  requests %.0f  barrier-2 %.0f  first product %.0f  second product + stores %.0f\n",
              T, set, w, NW, sum[0] / grid, sum[1] / grid, sum[2] / grid, sum[3] / grid, sum[4] / grid, sum[5] / grid);
    }
  }
}

template <int T>
void ts4_launch(Ctx& c, Ts4Args a) {
  // (two wavefronts per SIMD with one block column each; the form with one wavefront per SIMD and two block columns -- NW = 4, same
  //  code -- measured 2.08 against 1.98 ms on 5000 matrices of side 200 and is not instantiated: it doubles the compile time)
  ts4_launch_nw<T, 8>(c, a);
}

__global__ void ts4_pad_r_kernel(int s, int LD, const double* __restrict__ R, double* __restrict__ Rp) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= LD * LD) return;
  const int k = e % LD, c = e / LD;
  Rp[e] = (k < s && c < s && k <= c) ? R[(long)c * s + k] : 0.0;
}

}  // namespace

// true: the product was done here (upper triangular R, 9 .. 13 tiles, enough matrices to occupy the chip, not in place)
bool psd_two_sided_onchip(Ctx& c, int side, int ncols, const double* R, int rstruct, const double* arr, long lda, double* prod, long ldp) {
  static const bool on = [] { const char* e = getenv("HYP_TS4"); return !(e && e[0] == '0'); }();
  static const int min_cols = [] { const char* e = getenv("HYP_TS4_MIN"); return e ? atoi(e) : 192; }();
  const int T = (side + 15) / 16;
  static const int tmin = [] { const char* e = getenv("HYP_TS4_TMIN"); return e ? atoi(e) : 6; }();   // (9: sides 81 .. 128 back on the two-pass kernels, round 4)
  if (!on || rstruct != 1 || T < tmin || T < 6 || T > 13 || ncols < min_cols || arr == prod) return false;
  const int LD = 16 * T;
  const size_t LD2 = (size_t)LD * LD;
  c.ts_ws.ensure(LD2 * sizeof(double));
  double* Rp = c.ts_ws.d();
  hipLaunchKernelGGL(ts4_pad_r_kernel, dim3((unsigned)((LD2 + 255) / 256)), dim3(256), 0, c.stream, side, LD, R, Rp);
  Ts4Args a{};
  a.s = side; a.ncols = ncols; a.A = arr; a.lda = lda; a.Rp = Rp; a.C = prod; a.ldc = ldp;
  switch (T) {
    case 6: ts4_launch<6>(c, a); break;
    case 7: ts4_launch<7>(c, a); break;
    case 8: ts4_launch<8>(c, a); break;
    case 9: ts4_launch<9>(c, a); break;
    case 10: ts4_launch<10>(c, a); break;
    case 11: ts4_launch<11>(c, a); break;
    case 12: ts4_launch<12>(c, a); break;
    default: ts4_launch<13>(c, a); break;
  }
  return true;
}

}  // namespace hyp
