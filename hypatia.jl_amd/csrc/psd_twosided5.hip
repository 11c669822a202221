// Fifth form of the two-sided product W_j = R' V_j R with an upper triangular R (PosSemidefTri's sqrt_hess_prod!,
// /root/reference/src/Cones/possemideftri.jl:161-177, on the q x n block of G: qrchol.jl:219-233; the first half of hess_prod!,
// :126-142) for SMALL sides, 3 .. 6 MFMA tiles (33 .. 96; config 4: 64 cones of side 80): ONE WAVEFRONT PER MATRIX.
//
// psd_ts4_kernel (psd_twosided4.hip) keeps the intermediate Z_j = V_j R of a side-200 matrix in the accumulators of eight
// wavefronts.  At side 80 ALL 25 tiles of Z are 200 registers of one wavefront (one wavefront per SIMD: 512 registers), so a matrix
// needs no second wavefront at all: no barrier per matrix, no Z or V in LDS, no tile-row imbalance.  (psd_ts_small_kernel, the form
// this replaces for triangular R, gives a matrix to five wavefronts: V and Z go through LDS, three barriers per matrix, the second
// product's tile rows are 15 / 14 / 12 / 9 / 5 tile products long and ten wavefronts share four SIMDs: 12.5 ms per config-4
// iteration = 0.3 of the FP64 MFMA peak.)
//
//   * R, padded to 16 T x 16 T, is staged into LDS once per workgroup ([k][c], row stride = 16 mod 32 doubles) and serves both products.
//   * V_j is never staged: a lane loads its MFMA A-operand entries V[16 m + (l & 15), 16 t + 4 ch + (l >> 4)] straight from the packed
//     svec column (row-major triangle: 16 lanes = 128 contiguous bytes), off-diagonals / sqrt(2) with the correctly rounded
//     three-operation division of the other kernels.  Byte offsets are per-lane constants + instruction immediates: no address
//     arithmetic between the MFMAs.  The K slice t + 1 (the next matrix's slice 0 after the last step) is in flight during step t.
//   * Step t = 0 .. T - 1 (R upper triangular: block column a of Z is final after step a):
//       first product   Z[m, a] += V[m, t] R[t, a]            for all m, a >= t           (T (T - t) x 4 MFMAs)
//       second product  W[t, a]  = sum_{k <= t} R[k, t]' Z[k, a]   for a <= t             ((t + 1)^2 x 4 MFMAs)
//     The D layout of v_mfma_f64_16x16x4 (lane l, register r: D[(l >> 4) + 4 r, l & 15]) is the layout of its B operand for k-chunk r,
//     so the second product reads Z's accumulators as operands (psd_twosided4.hip:8-11).  Consecutive MFMAs go to different
//     accumulators (a dependent FP64 MFMA waits ~95 cycles, an independent one issues after 64).
//   * W's upper triangle goes from the accumulators to the packed column (off-diagonals * sqrt(2)); svec entries whose column lies in
//     tile t are written in step t and read only in steps <= t, so the product may be formed in place.
// Summation order of every entry: k ascending, as in the other forms (bitwise the same W as psd_ts_small_kernel's is not claimed:
// that kernel rounds Z to memory precision the same way -- both keep FP64 -- but accumulates pass 2 in the same order; the tests
// compare with the oracle, tests/test_hip_cones.py).
#include <type_traits>
#include <utility>

#include "cones.hpp"
#include "gemm_f64.hpp"
#include "hyp_internal.hpp"

namespace hyp {

namespace {

// (round 6) fragment reads as volatile LDS loads: ds_read_b64 instead of the compiler's half-rate ds_read2_b64 pairs (see psd_twosided4.hip);
// -DHYP_TS5_READ2: plain loads, for A/B builds
#ifdef HYP_TS5_READ2
#define TS5_LDS_RD(P) (*(P))
#else
typedef const volatile double __attribute__((address_space(3))) ts5_lds_cv;
#define TS5_LDS_RD(P) (*(ts5_lds_cv*)(P))
#endif
struct Ts5Args {
  int s, ncols;
  const double* A;   // svec columns
  long lda;
  const double* R;   // s x s col-major, upper triangular (zeros below the diagonal in memory)
  double* C;
  long ldc;
};

__device__ __forceinline__ double ts5_div_rt2(double x) {   // x / sqrt(2) correctly rounded (div_rt2, psd_twosided.hip)
  const double d = 1.4142135623730951, c = 0.70710678118654746;
  const double q = x * c;
  const double r = fma(-q, d, x);
  return fma(r, c, q);
}

template <class F, int... I>
__device__ __forceinline__ void ts5_for(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}

#define TS5_AT(BASE, OFF) (*reinterpret_cast<const double*>(reinterpret_cast<const char*>(BASE) + (size_t)(unsigned)(OFF)))

// RAW: the product is formed with sqrt(2) V_j -- the packed column's off-diagonal entries AS THEY ARE, its diagonal entries times
// sqrt(2) -- so that R' (sqrt(2) V) R = sqrt(2) W holds the packed result's off-diagonal entries as they are and its diagonal
// entries times sqrt(2): two multiplications per lane and step instead of a correctly rounded division for each of the 100 operand
// entries and a multiplication for each of the 50 result entries of a lane, vector work that sits in front of every step's MFMAs
// (8.3 -> see profiles/r05_ts5.txt).  Same products, same sums; the scalings round the diagonal terms instead of the off-diagonal ones.
template <int T, bool EDGE, bool RAW>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void psd_ts5_kernel(Ts5Args p) {
  constexpr int N = 16 * T;
  constexpr int S = (N % 32 == 16) ? N : N + 16;
  extern __shared__ __attribute__((aligned(16))) double ts5_lds[];
  double* __restrict__ Rs = ts5_lds;   // [k][c]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nn = lane & 15, q = lane >> 4;
  const int s = p.s, sm1 = s - 1;
  for (int e = tid; e < N * N; e += 256) {
    const int k = e % N, c = e / N;
    Rs[k * S + c] = (k < s && c < s && k <= c) ? p.R[(long)c * s + k] : 0.0;
  }
  __syncthreads();

  // ---- per-lane byte offsets into a packed column (tri(v) = v (v + 1) / 2; indices beyond the side are clamped, their values zeroed)
  auto tri = [](int v) { return (unsigned)(v * (v + 1) / 2); };
  unsigned rowq[T];        // entry (row 16 m + nn, column q): + 8 (16 t + 4 ch) reaches column 16 t + 4 ch + q, for tiles below the K slice (m > t)
  unsigned colq[T][4];     // entry (row nn, column 16 t + 4 ch + q): + 8 (16 m) reaches row 16 m + nn, for tiles above the K slice (m < t); the stores too
  // (the diagonal tile takes whichever of the two forms has the larger index as its column: a select per load)
#pragma unroll
  for (int m = 0; m < T; ++m) rowq[m] = (tri(EDGE ? min(16 * m + nn, sm1) : 16 * m + nn) + (unsigned)q) * 8u;
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) {
      const int k = EDGE ? min(16 * t + 4 * ch + q, sm1) : 16 * t + 4 * ch + q;
      colq[t][ch] = (tri(k) + (unsigned)nn) * 8u;
    }
  bool upper[4];   // entry (nn, 4 ch + q) of a diagonal tile lies on or above the diagonal
#pragma unroll
  for (int ch = 0; ch < 4; ++ch) upper[ch] = (4 * ch + q >= nn);
  const unsigned last_off = (tri(sm1) + (unsigned)sm1) * 8u;
  // edge masks (last tile only): rows 16 (T - 1) + nn and columns 16 (T - 1) + 4 ch + q inside the side
  const bool row_in = !EDGE || (16 * (T - 1) + nn < s);
  bool col_in[4];
#pragma unroll
  for (int ch = 0; ch < 4; ++ch) col_in[ch] = !EDGE || (16 * (T - 1) + 4 * ch + q < s);

  const d4_t zero4 = (d4_t){0.0, 0.0, 0.0, 0.0};
  d4_t Z[T][T];
  double vn[T][4];   // raw entries of the coming K slice

  // request the K slice TT of column AJ into vn
#define TS5_REQUEST(AJ, TT)                                                                                       \
  {                                                                                                               \
    _Pragma("unroll") for (int m = 0; m < T; ++m) {                                                               \
      _Pragma("unroll") for (int ch = 0; ch < 4; ++ch) {                                                          \
        if (m > (TT)) vn[m][ch] = TS5_AT(AJ, rowq[m] + 8u * (unsigned)(16 * (TT) + 4 * ch));                      \
        else if (m < (TT)) vn[m][ch] = TS5_AT(AJ, colq[TT][ch] + 8u * (unsigned)(16 * m));                        \
        else {                                                                                                    \
          unsigned o_ = upper[ch] ? colq[TT][ch] + 8u * (unsigned)(16 * (TT)) : rowq[m] + 8u * (unsigned)(16 * (TT) + 4 * ch); \
          if (EDGE && (TT) == T - 1) o_ = min(o_, last_off);                                                      \
          vn[m][ch] = TS5_AT(AJ, o_);                                                                             \
        }                                                                                                         \
      }                                                                                                           \
    }                                                                                                             \
  }

  const long stride = (long)gridDim.x * 4;
  long j = (long)blockIdx.x * 4 + wave;
  if (j < p.ncols) {
    const double* __restrict__ A0 = p.A + j * p.lda;
    TS5_REQUEST(A0, 0)
  }
  for (; j < p.ncols; j += stride) {
#pragma unroll
    for (int m = 0; m < T; ++m)
#pragma unroll
      for (int a = 0; a < T; ++a) Z[m][a] = zero4;
    double* __restrict__ Cj = p.C + j * p.ldc;
    const long jn = min(j + stride, (long)p.ncols - 1);   // (past the last matrix: its last column again, unused)
    auto step = [&](auto tc) __attribute__((always_inline)) {
      constexpr int t = decltype(tc)::value;
      // ---- the slice of this step: scaled operands; then the next slice goes into flight
      double vf[T][4];
#pragma unroll
      for (int m = 0; m < T; ++m)
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
          double v;
          if (RAW) {
            v = vn[m][ch];
            if (m == t) v = (nn == 4 * ch + q) ? v * 1.4142135623730951 : v;
          } else {
            v = ts5_div_rt2(vn[m][ch]);
            if (m == t) v = (nn == 4 * ch + q) ? vn[m][ch] : v;
          }
          if (EDGE && m == T - 1) v = row_in ? v : 0.0;
          if (EDGE && t == T - 1) v = col_in[ch] ? v : 0.0;
          vf[m][ch] = v;
        }
      {
        constexpr int TN = (t + 1 < T) ? t + 1 : 0;
        const double* __restrict__ An = p.A + ((t + 1 < T) ? j : jn) * p.lda;
        TS5_REQUEST(An, TN)
      }
      // ---- first product: Z[m, a] += V[m, K slice t] R[K slice t, a], a >= t
      // (the R fragments of k-chunk ch + 1 are requested before the MFMAs of chunk ch; the scheduling barriers keep the compiler from
      //  hoisting every chunk's reads to the top of the step)
      double bfn[T];
#pragma unroll
      for (int a = t; a < T; ++a) bfn[a] = TS5_LDS_RD(Rs + (16 * t + q) * S + 16 * a + nn);
#pragma unroll
      for (int ch = 0; ch < 4; ++ch) {
        double bf[T];
#pragma unroll
        for (int a = t; a < T; ++a) bf[a] = bfn[a];
        if (ch + 1 < 4) {
#pragma unroll
          for (int a = t; a < T; ++a) bfn[a] = TS5_LDS_RD(Rs + (16 * t + 4 * (ch + 1) + q) * S + 16 * a + nn);
        }
#pragma unroll
        for (int m = 0; m < T; ++m)
#pragma unroll
          for (int a = t; a < T; ++a) Z[m][a] = __builtin_amdgcn_mfma_f64_16x16x4f64(vf[m][ch], bf[a], Z[m][a], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      // ---- second product: W[t, a] = sum_{k <= t} R[k, t]' Z[k, a], a <= t (D[i = column of W in tile t, j = row of W in tile a])
      constexpr int NY = (t == 0) ? 2 : t + 1;   // (one result tile only: two partial sums, the even and the odd k-chunks)
      d4_t Y[NY];
#pragma unroll
      for (int a = 0; a < NY; ++a) Y[a] = zero4;
      double rfn = TS5_LDS_RD(Rs + q * S + 16 * t + nn);
#pragma unroll
      for (int kt = 0; kt <= t; ++kt)
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
          const double rf = rfn;
          if (4 * kt + ch + 1 < 4 * (t + 1)) rfn = TS5_LDS_RD(Rs + (4 * (4 * kt + ch + 1) + q) * S + 16 * t + nn);   // (chunk index 4 kt + ch: rows 4 (4 kt + ch) + q)
          if constexpr (t == 0) {
            Y[ch & 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(rf, Z[0][0][ch], Y[ch & 1], 0, 0, 0);
          } else {
#pragma unroll
            for (int a = 0; a <= t; ++a) Y[a] = __builtin_amdgcn_mfma_f64_16x16x4f64(rf, Z[kt][a][ch], Y[a], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      // register r of a result tile: column 16 t + q + 4 r of W, row 16 a + nn; packed entry tri(column) + row
#pragma unroll
      for (int a = 0; a <= t; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          double v;
          if constexpr (t == 0) v = Y[0][r] + Y[1][r];
          else v = Y[a][r];
          double* dst = reinterpret_cast<double*>(reinterpret_cast<char*>(Cj) + (size_t)(colq[t][r] + 8u * (unsigned)(16 * a)));
          const bool inside = !(EDGE && t == T - 1) || col_in[r];
          if (a == t) {
            if (RAW) {
              if (nn <= q + 4 * r && inside) *dst = (nn == q + 4 * r) ? v * 0.70710678118654746 : v;
            } else {
              if (nn <= q + 4 * r && inside) *dst = (nn == q + 4 * r) ? v : v * 1.4142135623730951;   // mat[i, j] * rt2 (arrayutilities.jl:176)
            }
          } else {
            if (inside) *dst = RAW ? v : v * 1.4142135623730951;
          }
        }
    };
    ts5_for(step, std::make_integer_sequence<int, T>{});
  }
#undef TS5_REQUEST
}

template <int T, bool EDGE, bool RAW>
void ts5_launch_e(Ctx& c, const Ts5Args& a) {
  constexpr int N = 16 * T;
  constexpr int S = (N % 32 == 16) ? N : N + 16;
  const size_t lds = (size_t)N * S * sizeof(double);
  static bool attr_set = false;
  if (!attr_set) {
    HYP_CHECK(hipFuncSetAttribute((const void*)psd_ts5_kernel<T, EDGE, RAW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  static const int cus = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n > 0 ? n : 256;
  }();
  const int grid = std::max(1, std::min((a.ncols + 3) / 4, cus));
  hipLaunchKernelGGL((psd_ts5_kernel<T, EDGE, RAW>), dim3(grid), dim3(256), lds, c.stream, a);
  HYP_CHECK(hipGetLastError());
}

template <int T>
void ts5_launch(Ctx& c, const Ts5Args& a) {
  static const bool raw = [] { const char* e = getenv("HYP_TS5_RAW"); return !(e && e[0] == '0'); }();
  if (raw) {
    if (a.s == 16 * T) ts5_launch_e<T, false, true>(c, a);
    else ts5_launch_e<T, true, true>(c, a);
  } else {
    if (a.s == 16 * T) ts5_launch_e<T, false, false>(c, a);
    else ts5_launch_e<T, true, false>(c, a);
  }
}

}  // namespace

// true: the product was done here (upper triangular R, 3 .. 6 tiles, enough matrices to give every SIMD one)
bool psd_two_sided_wave(Ctx& c, int side, int ncols, const double* R, int rstruct, const double* arr, long lda, double* prod, long ldp) {
  static const bool on = [] { const char* e = getenv("HYP_TS5"); return !(e && e[0] == '0'); }();
  static const int min_cols = [] { const char* e = getenv("HYP_TS5_MIN"); return e ? atoi(e) : 512; }();
  // (six tiles -- 36 accumulator tiles = 288 registers -- spill 0.6 - 1.6 KB per lane as compiled today: HYP_TS5_T6=1 to try them)
  static const int tmax = [] { const char* e = getenv("HYP_TS5_T6"); return (e && e[0] == '1') ? 6 : 5; }();
  const int T = (side + 15) / 16;
  if (!on || rstruct != 1 || T < 3 || T > tmax || ncols < min_cols) return false;
  Ts5Args a{};
  a.s = side; a.ncols = ncols; a.A = arr; a.lda = lda; a.R = R; a.C = prod; a.ldc = ldp;
  switch (T) {
    case 3: ts5_launch<3>(c, a); break;
    case 4: ts5_launch<4>(c, a); break;
    case 5: ts5_launch<5>(c, a); break;
    default: ts5_launch<6>(c, a); break;
  }
  return true;
}

}  // namespace hyp
