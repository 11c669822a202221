// Two-sided products W_j = R' V_j R on batches of svec columns.
//
// This is the body of PosSemidefTri's sqrt_hess_prod! / inv_sqrt_hess_prod! / hess_prod! / inv_hess_prod!
// (/root/reference/src/Cones/possemideftri.jl:126-195) applied to the q x n block of G in update_lhs_fact
// (src/Solvers/systemsolvers/qrchol.jl:219-233): for every column, svec_to_smat!, two triangular or
// symmetric products with the s x s cone matrix, smat_to_svec!.
//
// The generic GEMM (gemm_f64_kernel.hpp) handles it as two tall stacked products, but with 128 x 128
// tiles a side of 200 is padded to 256 and the triangular operand is only skipped tile by tile: 2.1x
// the useful MFMA work, plus separate unpack / pack passes over 1.6 GB.  Here the work is cut at the
// MFMA's own 16 x 16 granularity and the svec <-> smat conversions are fused into loader / epilogue:
//
//   pass 1  Z_j = V_j R      A side: V_j gathered straight from the svec column (off-diagonals / sqrt(2))
//   pass 2  W_j = Z_j' R     (= R' V_j R, V symmetric) A side: Z_j, K-contiguous; only tiles on or above
//                            the diagonal; epilogue writes the packed svec column (off-diagonals * sqrt(2))
//
// Both passes are one kernel: C[m, c] = sum_k A[m, k] R[k, c].  The T x T grid of 16 x 16 tiles of a
// matrix (T = ceil(s / 16)) is split into nb x nb workgroup tiles of at most 7 x 7 MFMA tiles (s = 200:
// 7 + 6), 256 threads each; wavefront w owns tile rows {w, w + 4} of the workgroup tile and all its
// columns, so every fragment read from LDS feeds 2 or 7 MFMAs.  K blocks of 16, register-staged double
// buffering as in the big GEMM; K ranges follow R's triangularity tile by tile.  The MFMA operands are
// swapped (R fragment as the row operand) so that 16 lanes hold 16 consecutive ROWS of a column of C:
// 128-byte stores into column-major Z and into the packed svec column alike.  The workgroups of one
// matrix are 8 apart in launch order = on the same XCD, sharing V_j / Z_j in its L2.
// Fixed summation order: bitwise reproducible.
#include "cones.hpp"
#include "gemm_f64.hpp"
#include "hyp_internal.hpp"

namespace hyp {

constexpr int TS_THREADS = 256;
constexpr int TS_LDK = 18;                    // LDS row stride (16 k + 2): conflict-free b64 fragment reads
// BT = MFMA tiles per workgroup-tile edge (template parameter).  Smaller tiles re-read operands from L2 more often but
// leave room for 3 - 4 resident workgroups per CU, which hides the staging / barrier latency of this short-K kernel:
// measured on 3000 matrices of side 200 (pass 1 / pass 2, us): BT 7: 1233 / 1103, BT 5: 1271 / 824, BT 4: 1100 / 977, BT 3: 1085 / 700.
// (round 2: K loop reordered -- store block kt + 1, request block kt + 2, multiply block kt, LDS-only barrier: 2500 matrices
//  960 / 588 -> 876 / 568 us; the same order in the big GEMM kernel, whose K steps are four times longer, lost 4 %)

struct TsArgs {
  int s, T, rstruct;          // side, ceil(side / 16), 0 full / 1 upper / 2 lower triangular R
  int nb, ncols;              // workgroup tiles per dimension; matrices
  const double* A;            // pass 1: svec columns (lda = column stride); pass 2: Z workspace (s*s per matrix)
  long lda;
  const double* R;            // s x s col-major, ld s
  double* C;                  // pass 1: Z workspace; pass 2: svec columns (ldc = column stride)
  long ldc;
};

// x / sqrt(2), correctly rounded like the IEEE division of the reference, in three FMA-class operations
// (q = x c; r = x - q d; q + r c with c = fl(1 / d): Markstein's division by a constant) instead of the
// ~15-instruction v_div sequence, seven times per thread and K block
__device__ __forceinline__ double div_rt2(double x) {
  const double d = 1.4142135623730951, c = 0.70710678118654746;
  const double q = x * c;
  const double r = fma(-q, d, x);
  return fma(r, c, q);
}

// first tile and tile count of block b when T tiles are split into nb nearly equal blocks
__device__ __forceinline__ void ts_block_range(int T, int nb, int b, int& t0, int& cnt) {
  const int base = T / nb, extra = T - base * nb;
  t0 = b * base + min(b, extra);
  cnt = base + (b < extra ? 1 : 0);
}

template <int PASS, int TS_BT, bool VEC = false>
__global__ __launch_bounds__(TS_THREADS, (TS_BT <= 4 ? 4 : (TS_BT <= 5 ? 3 : 2))) void psd_ts_kernel(TsArgs p) {
  constexpr int TS_WR = (TS_BT + 3) / 4;        // tile rows per wavefront: w, w + 4
  constexpr int TS_NREP = VEC ? 2 * ((TS_BT + 1) / 2) : TS_BT;   // staged elements per thread per operand per K block (16 BT * 16 / 256; VEC: in pairs)
  constexpr int TS_NREP2 = (TS_BT + 1) / 2;     // VEC: 16-byte loads, k pair 2 (tid & 7), row (tid >> 3) + 32 rep
  typedef double d2_t __attribute__((ext_vector_type(2)));
  constexpr int TS_OPSZ = 16 * TS_BT * TS_LDK;  // doubles per operand per buffer
  __shared__ double lds[2][2][TS_OPSZ];   // [buffer][A / B][row * LDK + k]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int s = p.s, T = p.T;

  // workgroup -> (matrix, row block, column block); the nb^2 workgroups of a matrix sit on one XCD
  const int nbb = p.nb * p.nb;
  const int g = blockIdx.x >> 3, xcd = blockIdx.x & 7;
  const long j = (long)(g / nbb) * 8 + xcd;
  if (j >= p.ncols) return;
  const int sub = g % nbb;
  int mt0, TR, ct0, TC;
  ts_block_range(T, p.nb, sub % p.nb, mt0, TR);
  ts_block_range(T, p.nb, sub / p.nb, ct0, TC);
  if (PASS == 2 && mt0 > ct0 + TC - 1) return;   // entirely below the diagonal

  const double* __restrict__ Aj = (PASS == 1) ? p.A + j * p.lda : p.A + j * (long)s * s;
  const double* __restrict__ R = p.R;

  // K blocks that can contribute to this workgroup tile
  int kt_lo = 0, kt_hi = T;
  if (p.rstruct == 1) kt_hi = min(T, ct0 + TC);   // upper R: R[k, c] = 0 for k > c
  else if (p.rstruct == 2) kt_lo = ct0;            // lower R: R[k, c] = 0 for k < c

  d4_t acc[TS_WR][TS_BT];
#pragma unroll
  for (int i = 0; i < TS_WR; ++i)
#pragma unroll
    for (int c = 0; c < TS_BT; ++c) acc[i][c] = (d4_t){0.0, 0.0, 0.0, 0.0};

  const int lk = tid & 15, lr = tid >> 4;   // staging: k within the block; row / column lr + 16 rep
  double ra[TS_NREP], rb[TS_NREP];
  const int sm1 = s - 1;
  const int arow0 = 16 * mt0, bcol0 = 16 * ct0;

  // Loads are unconditional on clamped indices and nothing is done with the values until store_tiles, after
  // the MFMAs of the current K block: any arithmetic (or select) on a loaded value right here makes the
  // compiler wait for the load -- or sink the load under the predicate -- and serialises the round trips.
  // VEC (even side, operands contiguous in k: R always, Z in pass 2): one 16-byte load per lane fetches a k pair, 8 lanes cover
  // a 128-byte line; (the svec gather of pass 1 stays 8 bytes per lane: above the diagonal consecutive k are a column apart)
  const int vk = 2 * (tid & 7), vr = tid >> 3;
  const bool vhalf = (TS_BT & 1) ? (vr < 16) : true;   // odd BT: the last pair of 32 rows is half used (wavefront-uniform)
  auto load_into = [&](int kt, double (&xa)[TS_NREP], double (&xb)[TS_NREP]) {
    const int kk = min(16 * kt + lk, sm1);
    const int kk2 = min(16 * kt + vk, s - 2);
    if (VEC) {
#pragma unroll
      for (int rep = 0; rep < TS_NREP2; ++rep) {
        if (rep == TS_NREP2 - 1 && !vhalf) break;
        const d2_t tb = *reinterpret_cast<const d2_t*>(R + min(bcol0 + vr + 32 * rep, sm1) * s + kk2);
        xb[2 * rep] = tb.x; xb[2 * rep + 1] = tb.y;
        if (PASS == 2) {
          const d2_t ta = *reinterpret_cast<const d2_t*>(Aj + min(arow0 + vr + 32 * rep, sm1) * s + kk2);
          xa[2 * rep] = ta.x; xa[2 * rep + 1] = ta.y;
        }
      }
      if (PASS == 2) return;
    }
#pragma unroll
    for (int rep = 0; rep < TS_BT; ++rep) {
      const int mm = min(arow0 + lr + 16 * rep, sm1);
      if (PASS == 1) {   // V[m, k] from the svec column: column-major upper triangle
        const int lo = min(mm, kk), hi = max(mm, kk);
        xa[rep] = Aj[hi * (hi + 1) / 2 + lo];   // (32-bit: s <= 2048)
      } else {           // Z[k, m]: column m of Z is contiguous in k
        xa[rep] = Aj[mm * s + kk];
      }
      if (!VEC) xb[rep] = R[min(bcol0 + lr + 16 * rep, sm1) * s + kk];   // R[k, c]
    }
  };
  auto store_from = [&](int buf, int kt, const double (&xa)[TS_NREP], const double (&xb)[TS_NREP]) {
    const int kp = 16 * kt + lk;
    const double kmask = (kp < s) ? 1.0 : 0.0;
    if (VEC) {
      const double vmask = (16 * kt + vk < s) ? 1.0 : 0.0;   // (even side: both of a pair or neither)
#pragma unroll
      for (int rep = 0; rep < TS_NREP2; ++rep) {
        if (rep == TS_NREP2 - 1 && !vhalf) break;
        const int row = vr + 32 * rep;
        const double cm = (bcol0 + row < s) ? vmask : 0.0;
        *reinterpret_cast<d2_t*>(&lds[buf][1][row * TS_LDK + vk]) = (d2_t){xb[2 * rep] * cm, xb[2 * rep + 1] * cm};
        if (PASS == 2) {
          const double am = (arow0 + row < s) ? vmask : 0.0;
          *reinterpret_cast<d2_t*>(&lds[buf][0][row * TS_LDK + vk]) = (d2_t){xa[2 * rep] * am, xa[2 * rep + 1] * am};
        }
      }
      if (PASS == 2) return;
    }
#pragma unroll
    for (int rep = 0; rep < TS_BT; ++rep) {
      const int m = arow0 + lr + 16 * rep, c = bcol0 + lr + 16 * rep;
      double va = xa[rep];
      if (PASS == 1) va = (m == kp) ? va : div_rt2(va);   // off-diagonals: vec[k] / rt2 (arrayutilities.jl:231)
      lds[buf][0][(lr + 16 * rep) * TS_LDK + lk] = va * ((m < s) ? kmask : 0.0);
      if (!VEC) lds[buf][1][(lr + 16 * rep) * TS_LDK + lk] = xb[rep] * ((c < s) ? kmask : 0.0);
    }
  };
  auto load_tiles = [&](int kt) { load_into(kt, ra, rb); };
  auto store_tiles = [&](int buf, int kt) { store_from(buf, kt, ra, rb); };

  const int fr = lane & 15, fk = lane >> 4;
  // Which of this wavefront's 2 x 7 tiles exist (bit 2 c + i): rows li = wave + 4 i < TR, columns c < TC,
  // pass 2 only on / above the diagonal.  Per K block the mask is narrowed by R's triangularity; the MFMA
  // loop then costs ONE scalar bit test + branch per tile (4 back-to-back MFMAs on its accumulator) --
  // evaluating the predicates per MFMA was ~20 scalar instructions each and held the pipe at 30 %.
  unsigned tmask = 0;
#pragma unroll
  for (int c = 0; c < TS_BT; ++c)
#pragma unroll
    for (int i = 0; i < TS_WR; ++i) {
      const int li = wave + 4 * i;
      bool ok = (li < TR) && (c < TC);
      if (PASS == 2) ok = ok && (mt0 + li <= ct0 + c);
      tmask |= (ok ? 1u : 0u) << (2 * c + i);
    }
  tmask = __builtin_amdgcn_readfirstlane(tmask);

  auto compute = [&](int kt, int buf) {
    unsigned need = tmask;
    if (p.rstruct == 1) {        // columns with ct >= kt: c >= kt - ct0
      const int cmin = max(0, kt - ct0);
      need = (cmin >= TS_BT) ? 0u : (need & (~0u << (2 * cmin)));
    } else if (p.rstruct == 2) { // columns with ct <= kt: c <= kt - ct0
      const int cmax = kt - ct0;
      need = (cmax < 0) ? 0u : ((cmax >= TS_BT - 1) ? need : (need & ((1u << (2 * cmax + 2)) - 1u)));
    }
    if (need == 0u) return;
    const double* as = lds[buf][0] + fr * TS_LDK + fk;
    const double* bs = lds[buf][1] + fr * TS_LDK + fk;
    double af[TS_WR][4], bf[TS_BT][4];
#pragma unroll
    for (int i = 0; i < TS_WR; ++i)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) af[i][q4] = as[(wave + 4 * i) * 16 * TS_LDK + 4 * q4];
#pragma unroll
    for (int c = 0; c < TS_BT; ++c)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) bf[c][q4] = bs[c * 16 * TS_LDK + 4 * q4];
#pragma unroll
    for (int c = 0; c < TS_BT; ++c)
#pragma unroll
      for (int i = 0; i < TS_WR; ++i) {
        if (!((need >> (2 * c + i)) & 1u)) continue;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) acc[i][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[c][q4], af[i][q4], acc[i][c], 0, 0, 0);
      }
  };

  // K loop.  Step kt: the operands of block kt + 1 (requested a whole step ago) go to the other LDS buffer, block kt + 2 is
  // requested, block kt is multiplied; the barrier at the end orders LDS traffic only (__syncthreads() would also wait for
  // the loads just requested: s_waitcnt vmcnt(0)).
  if (kt_hi > kt_lo) {
    load_tiles(kt_lo);
    store_tiles(0, kt_lo);
    if (kt_lo + 1 < kt_hi) load_tiles(kt_lo + 1);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    for (int kt = kt_lo; kt < kt_hi; ++kt) {
      const int buf = (kt - kt_lo) & 1;
      if (kt + 1 < kt_hi) store_tiles(buf ^ 1, kt + 1);
      if (kt + 2 < kt_hi) load_tiles(kt + 2);
      compute(kt, buf);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
  }

  // epilogue: register r of tile (i, c) is C[row = 16 (mt0 + li) + fr, col = 16 (ct0 + c) + fk + 4 r]
#pragma unroll
  for (int i = 0; i < TS_WR; ++i) {
    const int li = wave + 4 * i;
    if (li >= TR) continue;
    const int m = 16 * (mt0 + li) + fr;
#pragma unroll
    for (int c = 0; c < TS_BT; ++c) {
      if (c >= TC) continue;
      if (PASS == 2 && mt0 + li > ct0 + c) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int col = 16 * (ct0 + c) + fk + 4 * r;
        if (PASS == 1) {
          if (m < s && col < s) p.C[j * (long)s * s + (long)col * s + m] = acc[i][c][r];
        } else {
          if (m <= col && col < s) {
            const double v = acc[i][c][r];
            p.C[j * p.ldc + (long)col * (col + 1) / 2 + m] = (m == col) ? v : v * 1.4142135623730951;   // mat[i, j] * rt2 (arrayutilities.jl:176)
          }
        }
      }
    }
  }
}

// =============================================================================================
// Third form of the two passes (round 3; default for sides > 96): the same tiling with the VALU work of the K loop removed.
// Cycle stamps in the loop of the kernel above (4 resident workgroups, side 200) gave per K block and wavefront: waiting for the
// operands requested a block earlier 15 cycles, LDS staging (index clamps, 0 / 1 masks, the division by sqrt(2)) 1000 - 1500,
// address arithmetic + issue of the next loads 300 - 600, LDS fragments + MFMAs 1200 - 1700, barrier the rest -- nothing waits
// for memory, but a wavefront's ~100 vector instructions take as long as its 16 MFMAs: on gfx950 the FP64 MFMA runs on the
// SIMD's FP64 lanes (matrix peak = vector peak, 78.6 TFLOP/s) and vector instructions of the other wavefronts do not issue
// underneath it, so every VALU instruction in the loop is MFMA time lost ("the parts add up").  Here
//   * R is copied once per call into a zero-padded 16 T x 16 T square and Z is kept at that leading dimension (pass 1 writes
//     the padding as exact zeros): no index clamps and no masks on any k-contiguous operand, whose loads become
//     global_load_dwordx4 v, v_off, s[base] with a per-thread constant offset and a scalar base that advances per K block;
//   * the svec gather of pass 1 distinguishes, per K block and wavefront-uniformly, blocks left of the row block (offset =
//     constant + 16 kt), right of it (k (k + 1) / 2 once per block, one add per element) and the diagonal-crossing ones (the
//     general formula); row / k masks only in edge blocks;
//   * the K loop is unrolled by two so that LDS addresses are immediates.
// Measured at side 200, 5000 columns: 2.75 -> 2.59 ms for the two passes (pass 1 1.69 -> 1.53, pass 2 unchanged at 1.05: its
// loop was already light).  Tried on top and not kept: a PERSISTENT form (4 workgroups per CU walking a list of (matrix, block)
// items as one flattened sequence of K blocks, the next item's first operands requested during the current item's last MFMAs):
// 2.88 ms with the items dealt round robin with a prime stride (3.63 with stride 128, where a workgroup meets the same block of
// every matrix); rotating the wavefront-to-tile-row assignment between workgroups (a block of three tile rows idles one
// wavefront): +-0; two matrices per workgroup sharing the R operand (25 % fewer operand bytes, half the occupancy): 2.79 vs 2.75.
// =============================================================================================
__global__ void ts_pad_r_kernel(int s, int LD, const double* __restrict__ R, double* __restrict__ Rp) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= LD * LD) return;
  const int k = e % LD, c = e / LD;
  Rp[e] = (k < s && c < s) ? R[(long)c * s + k] : 0.0;
}

template <int PASS, int TS_BT>
__global__ __launch_bounds__(TS_THREADS, 4) void psd_ts3_kernel(TsArgs p) {
  static_assert(TS_BT <= 4, "one tile row per wavefront");
  constexpr int NREP2 = (TS_BT + 1) / 2;     // 16-byte loads: k pair 2 (tid & 7), row (tid >> 3) + 32 rep
  typedef double d2_t __attribute__((ext_vector_type(2)));
  constexpr int TS_OPSZ = 16 * TS_BT * TS_LDK;
  __shared__ double lds[2][2][TS_OPSZ];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int s = p.s, T = p.T, LD = 16 * T;
  const long LD2 = (long)LD * LD;
  // (pass 2 launches the blocks on or above the diagonal only: column by column, (0,0) (0,1) (1,1) (0,2) ...)
  const int nbb = (PASS == 1) ? p.nb * p.nb : p.nb * (p.nb + 1) / 2;
  const int g = blockIdx.x >> 3, xcd = blockIdx.x & 7;
  const long j = (long)(g / nbb) * 8 + xcd;
  if (j >= p.ncols) return;
  int sub = g % nbb;
  int rb, cb;
  if (PASS == 1) {
    rb = sub % p.nb; cb = sub / p.nb;
  } else {
    cb = 0;
    while (sub > cb) { sub -= cb + 1; ++cb; }
    rb = sub;
  }
  int mt0, TR, ct0, TC;
  ts_block_range(T, p.nb, rb, mt0, TR);
  ts_block_range(T, p.nb, cb, ct0, TC);
  if (PASS == 2 && mt0 > ct0 + TC - 1) return;
  const double* __restrict__ Aj = (PASS == 1) ? p.A + j * p.lda : p.A + j * LD2;
  int kt_lo = 0, kt_hi = T;
  if (p.rstruct == 1) kt_hi = min(T, ct0 + TC);
  else if (p.rstruct == 2) kt_lo = ct0;

  d4_t acc[TS_BT];
#pragma unroll
  for (int c = 0; c < TS_BT; ++c) acc[c] = (d4_t){0.0, 0.0, 0.0, 0.0};

  const int arow0 = 16 * mt0, bcol0 = 16 * ct0;
  const int vk = 2 * (tid & 7), vr = tid >> 3;
  const bool vhalf = (TS_BT & 1) ? (vr < 16) : true;   // odd BT: the last pair of 32 rows is half used (wavefront-uniform)
  unsigned offB[NREP2], offZ[NREP2];
  int ldsV[NREP2];
#pragma unroll
  for (int rep = 0; rep < NREP2; ++rep) {
    offB[rep] = (unsigned)(min(bcol0 + vr + 32 * rep, LD - 1) * LD + vk);
    offZ[rep] = (unsigned)(min(arow0 + vr + 32 * rep, LD - 1) * LD + vk);
    ldsV[rep] = (vr + 32 * rep) * TS_LDK + vk;
  }
  // pass 1 gather: row lr + 16 rep of the block, k = lk of the K block
  const int lk = tid & 15, lr = tid >> 4;
  const int sm1 = s - 1;
  int mrow[TS_BT];
  unsigned triM[TS_BT];
#pragma unroll
  for (int rep = 0; rep < TS_BT; ++rep) {
    mrow[rep] = min(arow0 + lr + 16 * rep, sm1);
    triM[rep] = (unsigned)(mrow[rep] * (mrow[rep] + 1) / 2 + lk);
  }
  const bool rows_edge = (arow0 + 16 * TS_BT > s);

  d2_t rbv[NREP2], rzv[NREP2];
  double ra[TS_BT];

  auto load_tiles = [&](int kt) {
    const double* __restrict__ Rk = p.R + 16 * kt;
#pragma unroll
    for (int rep = 0; rep < NREP2; ++rep) {
      if (rep == NREP2 - 1 && !vhalf) break;
      rbv[rep] = *reinterpret_cast<const d2_t*>(Rk + offB[rep]);
    }
    if (PASS == 2) {
      const double* __restrict__ Zk = Aj + 16 * kt;
#pragma unroll
      for (int rep = 0; rep < NREP2; ++rep) {
        if (rep == NREP2 - 1 && !vhalf) break;
        rzv[rep] = *reinterpret_cast<const d2_t*>(Zk + offZ[rep]);
      }
    } else {
      if (16 * kt + 15 < arow0) {            // every k of the block left of every row: V[m, k] = vec[m (m + 1) / 2 + k]
        const double* __restrict__ Ak = Aj + 16 * kt;
#pragma unroll
        for (int rep = 0; rep < TS_BT; ++rep) ra[rep] = Ak[triM[rep]];
      } else if (16 * kt >= arow0 + 16 * TS_BT) {   // right of every row: vec[k (k + 1) / 2 + m]
        const int kk = min(16 * kt + lk, sm1);
        const unsigned tk = (unsigned)(kk * (kk + 1) / 2);
#pragma unroll
        for (int rep = 0; rep < TS_BT; ++rep) ra[rep] = Aj[tk + (unsigned)mrow[rep]];
      } else {
        const int kk = min(16 * kt + lk, sm1);
#pragma unroll
        for (int rep = 0; rep < TS_BT; ++rep) {
          const int lo = min(mrow[rep], kk), hi = max(mrow[rep], kk);
          ra[rep] = Aj[(unsigned)(hi * (hi + 1) / 2 + lo)];
        }
      }
    }
  };
  auto store_tiles = [&](double (&L)[2][TS_OPSZ], int kt) {
#pragma unroll
    for (int rep = 0; rep < NREP2; ++rep) {
      if (rep == NREP2 - 1 && !vhalf) break;
      *reinterpret_cast<d2_t*>(&L[1][ldsV[rep]]) = rbv[rep];
      if (PASS == 2) *reinterpret_cast<d2_t*>(&L[0][ldsV[rep]]) = rzv[rep];
    }
    if (PASS == 1) {
      const bool diag_blk = !(16 * kt + 15 < arow0) && !(16 * kt >= arow0 + 16 * TS_BT);
      const bool k_edge = (16 * kt + 16 > s);
      const int kp = 16 * kt + lk;
#pragma unroll
      for (int rep = 0; rep < TS_BT; ++rep) {
        double va = div_rt2(ra[rep]);   // off-diagonals: vec[k] / rt2 (arrayutilities.jl:231)
        if (diag_blk) va = (arow0 + lr + 16 * rep == kp) ? ra[rep] : va;
        if (rows_edge || k_edge) va = (arow0 + lr + 16 * rep < s && kp < s) ? va : 0.0;
        L[0][(lr + 16 * rep) * TS_LDK + lk] = va;
      }
    }
  };

  const int fr = lane & 15, fk = lane >> 4;
  unsigned tmask = 0;   // bit c: this wavefront's tile (wave, c) exists
#pragma unroll
  for (int c = 0; c < TS_BT; ++c) {
    bool ok = (wave < TR) && (c < TC);
    if (PASS == 2) ok = ok && (mt0 + wave <= ct0 + c);
    tmask |= (ok ? 1u : 0u) << c;
  }
  tmask = __builtin_amdgcn_readfirstlane(tmask);

  auto compute = [&](int kt, const double (&L)[2][TS_OPSZ]) {
    unsigned need = tmask;
    if (p.rstruct == 1) {
      const int cmin = max(0, kt - ct0);
      need = (cmin >= TS_BT) ? 0u : (need & (~0u << cmin));
    } else if (p.rstruct == 2) {
      const int cmax = kt - ct0;
      need = (cmax < 0) ? 0u : ((cmax >= TS_BT - 1) ? need : (need & ((1u << (cmax + 1)) - 1u)));
    }
    if (need == 0u) return;
    const double* as = L[0] + (wave * 16 + fr) * TS_LDK + fk;
    const double* bs = L[1] + fr * TS_LDK + fk;
    double af[4], bf[TS_BT][4];
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) af[q4] = as[4 * q4];
#pragma unroll
    for (int c = 0; c < TS_BT; ++c)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) bf[c][q4] = bs[c * 16 * TS_LDK + 4 * q4];
    // (tile by tile, the four k-chunks of a tile back to back on its accumulator: with the k-chunks outside -- consecutive MFMAs on
    //  different accumulators -- pass 1 took 1.94 instead of 1.69 ms at 48 x 48 and spilled at 64 x 64)
#pragma unroll
    for (int c = 0; c < TS_BT; ++c) {
      if (!((need >> c) & 1u)) continue;
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[c][q4], af[q4], acc[c], 0, 0, 0);
    }
  };

  if (kt_hi > kt_lo) {
    load_tiles(kt_lo);
    store_tiles(lds[0], kt_lo);
    if (kt_lo + 1 < kt_hi) load_tiles(kt_lo + 1);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    for (int kt = kt_lo; kt < kt_hi; kt += 2) {
      if (kt + 1 < kt_hi) store_tiles(lds[1], kt + 1);
      if (kt + 2 < kt_hi) load_tiles(kt + 2);
      compute(kt, lds[0]);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (kt + 1 >= kt_hi) break;
      if (kt + 2 < kt_hi) store_tiles(lds[0], kt + 2);
      if (kt + 3 < kt_hi) load_tiles(kt + 3);
      compute(kt + 1, lds[1]);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
  }

  if (wave >= TR) return;
  const int m = 16 * (mt0 + wave) + fr;
#pragma unroll
  for (int c = 0; c < TS_BT; ++c) {
    if (c >= TC) continue;
    if (PASS == 2 && mt0 + wave > ct0 + c) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int col = 16 * (ct0 + c) + fk + 4 * r;
      if (PASS == 1) {
        p.C[j * LD2 + (long)col * LD + m] = acc[c][r];   // (rows / columns beyond s: exact zeros, the padding pass 2 relies on)
      } else {
        if (m <= col && col < s) {
          const double v = acc[c][r];
          p.C[j * p.ldc + (long)col * (col + 1) / 2 + m] = (m == col) ? v : v * 1.4142135623730951;   // mat[i, j] * rt2 (arrayutilities.jl:176)
        }
      }
    }
  }
}

// =============================================================================================
// Small sides (<= 96, i.e. T <= 6 tiles: config 4's side 80): the whole two-sided product of a matrix stays on one CU.
// One workgroup of T wavefronts per matrix; R (shared by every column of the cone) is staged in LDS once per workgroup, which
// then walks a contiguous range of columns: V_j gathered from the svec column into LDS (both triangles), wavefront w computes
// tile row w of Z = V R in registers and writes it over ITS rows of V (nobody else reads those rows in this pass), then tile
// row w of the upper triangle of W = Z' R, stored as the packed svec column.  No Z round trip through HBM (the two-pass
// kernel moves 3 s^2 doubles per matrix for s^2 of input + output), no per-K-tile barrier: three barriers per matrix; the
// next column's svec entries are in flight (registers) during the two passes.
// Measured at config 4 (64 cones of side 80, 5000 columns each): 21.5 -> 12.2 ms per iteration.  Two variants were tried and
// lost: four wavefronts with the T^2 tiles dealt out evenly and Z in a second LDS buffer (18.8 ms: every MFMA then needs two
// LDS operand reads of its own, the tile-row form shares the V / Z fragment over a whole row of tiles), and run-time tile
// coordinates instead of per-wavefront tile rows (31 ms: one basic block per MFMA).
// LDS strides: [k][c] operands (R, and Z in pass 2) want consecutive rows 32 dwords apart modulo 64 (S = 16 mod 32 doubles);
// the row-indexed operand V of pass 1 wants 2 S2 = 4 mod 32 dwords (S2 = 16 T + 2).
// =============================================================================================
// TEAMS = 2 (side 80, config 4): two teams of T wavefronts share the staged R and work on two matrices side by side, each in a VZ
// buffer of its own -- 51 + 2 x 52.5 KB of LDS instead of one workgroup of five wavefronts per CU (103.7 KB: a second one does
// not fit), so that the barriers, the gather and the operand latencies of one matrix run under the MFMAs of the other.
template <int T, int TEAMS>
__global__ __launch_bounds__(64 * T * TEAMS) void psd_ts_small_kernel(TsArgs p, int cols_per_wg) {
  constexpr int N = 16 * T;
  constexpr int S = (N % 32 == 16) ? N : N + 16;
  constexpr int S2 = N + 2;
  constexpr int THREADS = 64 * T;            // per team
  constexpr int ALL = THREADS * TEAMS;
  extern __shared__ __attribute__((aligned(16))) double ts_lds[];
  double* Rs = ts_lds;             // [k][c], N x S
  const int gtid = threadIdx.x, lane = gtid & 63;
  const int gw = __builtin_amdgcn_readfirstlane(gtid >> 6);
  const int team = gw / T, w = gw % T;
  const int tid = gtid - team * THREADS;     // within the team
  double* VZ = ts_lds + N * S + team * (N * S2);   // [m][k] (V), then Z over the same rows, N x S2 (one per team)
  const int q = lane >> 4, nn = lane & 15;
  const int s = p.s;
  const long d = (long)s * (s + 1) / 2;
  for (int e = gtid; e < N * N; e += ALL) {   // R -> LDS (zero padding beyond s; a triangular R has its zeros in memory)
    const int k = e % N, c = e / N;
    Rs[k * S + c] = (k < s && c < s) ? p.R[(long)c * s + k] : 0.0;
  }
  for (int e = tid; e < N * S2; e += THREADS) VZ[e] = 0.0;
  // team t takes the columns j0 + t, j0 + t + TEAMS, ...; both teams walk the same number of rounds (the barriers are the
  // workgroup's), a team without a column in the last round idles through it
  const long jb = (long)blockIdx.x * cols_per_wg;
  const long j1 = min((long)p.ncols, jb + cols_per_wg);
  const long j0 = jb + team;
  const long rounds = (j1 - jb + TEAMS - 1) / TEAMS;
  constexpr int PER = (N * (N + 1) / 2 + THREADS - 1) / THREADS;   // upper-triangle entries per thread
  // (i, j) of this thread's entries, the same for every matrix, as the two LDS offsets i S2 + j and j S2 + i (14 bits each:
  // N S2 <= 96 x 98), bit 28 = diagonal entry, bit 29 = valid -- one register per entry
  unsigned eo[PER];
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const long e = tid + (long)THREADS * u;
    int j = (int)((sqrt(8.0 * (double)e + 1.0) - 1.0) * 0.5);
    while ((long)(j + 1) * (j + 2) / 2 <= e) ++j;
    while ((long)j * (j + 1) / 2 > e) --j;
    const int i = (int)(e - (long)j * (j + 1) / 2);
    eo[u] = (e < d) ? ((unsigned)(i * S2 + j) | ((unsigned)(j * S2 + i) << 14) | ((i == j) ? (1u << 28) : 0u) | (1u << 29)) : 0u;
  }
  double vreg[PER];
  if (j0 < j1) {
    const double* __restrict__ col = p.A + j0 * p.lda;
#pragma unroll
    for (int u = 0; u < PER; ++u) vreg[u] = col[min((long)tid + (long)THREADS * u, d - 1)];
  }
  __syncthreads();
  for (long rd = 0; rd < rounds; ++rd) {
    const long j = j0 + rd * TEAMS;
    const bool live = j < j1;
    // ---- V_j -> LDS, both triangles, off-diagonals / sqrt(2) (arrayutilities.jl:231)
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      if (live && (eo[u] >> 29)) {
        const double v = ((eo[u] >> 28) & 1u) ? vreg[u] : div_rt2(vreg[u]);
        VZ[eo[u] & 0x3fffu] = v;
        VZ[(eo[u] >> 14) & 0x3fffu] = v;
      }
    }
    __syncthreads();
    if (j + TEAMS < j1) {   // next column's entries: in flight during the two passes
      const double* __restrict__ col = p.A + (j + TEAMS) * p.lda;
#pragma unroll
      for (int u = 0; u < PER; ++u) vreg[u] = col[min((long)tid + (long)THREADS * u, d - 1)];
    }
    // ---- pass 1: tile row w of Z = V R (k-chunks outside, column tiles inside: consecutive MFMAs go to different accumulators)
    d4_t acc[T];
#pragma unroll
    for (int ct = 0; ct < T; ++ct) acc[ct] = (d4_t){0.0, 0.0, 0.0, 0.0};
    if (live) {
#pragma unroll
    for (int kt = 0; kt < T; ++kt) {
      double af[4];
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) af[kc] = VZ[(16 * w + nn) * S2 + 16 * kt + 4 * kc + q];
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) {
#pragma unroll
        for (int ct = 0; ct < T; ++ct) {
          if ((p.rstruct == 1 && kt > ct) || (p.rstruct == 2 && kt < ct)) continue;   // R[kt, ct] is a zero tile
          acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[kc], Rs[(16 * kt + 4 * kc + q) * S + 16 * ct + nn], acc[ct], 0, 0, 0);
        }
      }
    }
    // Z tile row w over this wavefront's own rows of V (lane (q, nn), register r: row q + 4 r, column nn of the tile)
#pragma unroll
    for (int ct = 0; ct < T; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) VZ[(16 * w + q + 4 * r) * S2 + 16 * ct + nn] = acc[ct][r];
    }
    __syncthreads();
    // ---- pass 2: tile row w of the upper triangle of W = Z' R, computed transposed (D[c][m]) so that 16 lanes hold 16
    //      consecutive ROWS of a column of W = 128 contiguous bytes of the packed column
    if (live) {
    double* __restrict__ out = p.C + j * p.ldc;
#pragma unroll
    for (int ct = 0; ct < T; ++ct) acc[ct] = (d4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kt = 0; kt < T; ++kt) {
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) {
        const double zf = VZ[(16 * kt + 4 * kc + q) * S2 + 16 * w + nn];
#pragma unroll
        for (int ct = 0; ct < T; ++ct) {
          if (ct < w) continue;
          if ((p.rstruct == 1 && kt > ct) || (p.rstruct == 2 && kt < ct)) continue;
          acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(Rs[(16 * kt + 4 * kc + q) * S + 16 * ct + nn], zf, acc[ct], 0, 0, 0);
        }
      }
    }
    const int m = 16 * w + nn;
#pragma unroll
    for (int ct = 0; ct < T; ++ct) {
      if (ct < w) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = 16 * ct + q + 4 * r;
        if (m <= c && c < s) out[(long)c * (c + 1) / 2 + m] = (m == c) ? acc[ct][r] : acc[ct][r] * 1.4142135623730951;   // arrayutilities.jl:176
      }
    }
    }
    __syncthreads();   // (the next matrix overwrites VZ)
  }
}

template <int T, int TEAMS>
static void ts_small_launch_t(Ctx& c, TsArgs a) {
  constexpr int N = 16 * T;
  constexpr int S = (N % 32 == 16) ? N : N + 16;
  constexpr int S2 = N + 2;
  const size_t lds = (size_t)(N * S + TEAMS * N * S2) * sizeof(double);
  static bool attr_set = false;
  if (!attr_set) {
    HYP_CHECK(hipFuncSetAttribute((const void*)psd_ts_small_kernel<T, TEAMS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  // enough workgroups to fill the chip a few times over (R is re-staged per workgroup: keep the ranges long)
  const int per_cu = std::max(1, (int)(160 * 1024 / lds));
  const int want = 256 * per_cu * 2;
  const int cols_per_wg = std::max(4 * TEAMS, ((a.ncols + want - 1) / want + TEAMS - 1) / TEAMS * TEAMS);
  const int grid = (a.ncols + cols_per_wg - 1) / cols_per_wg;
  hipLaunchKernelGGL((psd_ts_small_kernel<T, TEAMS>), dim3(grid), dim3(64 * T * TEAMS), lds, c.stream, a, cols_per_wg);
}
template <int T>
static void ts_small_launch(Ctx& c, TsArgs a) {
  // two teams where one workgroup per CU is all that fits otherwise and two VZ buffers + R still do (side 80: 156 KB)
  static const bool teams_on = [] { const char* e = getenv("HYP_TS_TEAMS"); return !(e && e[0] == '0'); }();
  if constexpr (T == 5) {
    if (teams_on && a.ncols >= 64) { ts_small_launch_t<T, 2>(c, a); return; }
  }
  ts_small_launch_t<T, 1>(c, a);
}

bool psd_two_sided_fused_ok(int side) { return side >= 1 && side <= 2048; }

// prod[:, j] = svec(R' smat(arr[:, j]) R), j < ncols.  zws: ncols * side^2 doubles.  arr may alias prod.
template <int PASS>
static void ts_launch(Ctx& c, TsArgs a, int bt) {
  a.nb = (a.T + bt - 1) / bt;
  const int grid = ((a.ncols + 7) / 8) * a.nb * a.nb * 8;
  static const bool vec_on = [] { const char* e = getenv("HYP_TS_VEC"); return !(e && e[0] == '0'); }();
  const bool vec = vec_on && (a.s % 2 == 0) && ((reinterpret_cast<uintptr_t>(a.R) | (PASS == 2 ? reinterpret_cast<uintptr_t>(a.A) : 0)) % 16 == 0);
  if (vec && bt == 2) { hipLaunchKernelGGL((psd_ts_kernel<PASS, 2, true>), dim3(grid), dim3(TS_THREADS), 0, c.stream, a); return; }
  if (vec && bt == 3) { hipLaunchKernelGGL((psd_ts_kernel<PASS, 3, true>), dim3(grid), dim3(TS_THREADS), 0, c.stream, a); return; }
  if (vec && bt == 4) { hipLaunchKernelGGL((psd_ts_kernel<PASS, 4, true>), dim3(grid), dim3(TS_THREADS), 0, c.stream, a); return; }
  switch (bt) {
    case 2: hipLaunchKernelGGL((psd_ts_kernel<PASS, 2>), dim3(grid), dim3(TS_THREADS), 0, c.stream, a); break;
    case 3: hipLaunchKernelGGL((psd_ts_kernel<PASS, 3>), dim3(grid), dim3(TS_THREADS), 0, c.stream, a); break;
    case 4: hipLaunchKernelGGL((psd_ts_kernel<PASS, 4>), dim3(grid), dim3(TS_THREADS), 0, c.stream, a); break;
    case 5: hipLaunchKernelGGL((psd_ts_kernel<PASS, 5>), dim3(grid), dim3(TS_THREADS), 0, c.stream, a); break;
    default: hipLaunchKernelGGL((psd_ts_kernel<PASS, 7>), dim3(grid), dim3(TS_THREADS), 0, c.stream, a); break;
  }
}

void psd_two_sided_fused(Ctx& c, int side, int ncols, const double* R, int rstruct, const double* arr, long lda, double* prod, long ldp,
                         double* zws) {
  if (ncols <= 0) return;
  static const int bt1 = [] { const char* e = getenv("HYP_TS_BT1"); return e ? atoi(e) : 4; }();   // pass 1: 64 x 64 workgroup tiles
  static const int bt2 = [] { const char* e = getenv("HYP_TS_BT2"); return e ? atoi(e) : 3; }();   // pass 2 (upper triangle only): 48 x 48, finer triangular skipping and 5 resident workgroups
  TsArgs a{};
  a.s = side; a.T = (side + 15) / 16; a.rstruct = rstruct; a.R = R; a.ncols = ncols;
  static const bool small_on = [] { const char* e = getenv("HYP_TS_SMALL"); return !(e && e[0] == '0'); }();
  if (small_on && psd_two_sided_wave(c, side, ncols, R, rstruct, arr, lda, prod, ldp)) return;   // (round 5: one wavefront per matrix)
  if (small_on && a.T <= 6) {   // one kernel, everything of a matrix on one CU
    a.A = arr; a.lda = lda; a.C = prod; a.ldc = ldp;
    switch (a.T) {
      case 1: ts_small_launch<1>(c, a); break;
      case 2: ts_small_launch<2>(c, a); break;
      case 3: ts_small_launch<3>(c, a); break;
      case 4: ts_small_launch<4>(c, a); break;
      case 5: ts_small_launch<5>(c, a); break;
      default: ts_small_launch<6>(c, a); break;
    }
    HYP_CHECK(hipGetLastError());
    return;
  }
  // a handful of columns (the Hessian products of the directions and of the line search: 16 workgroups per matrix at 64 x 64,
  // one wavefront per SIMD walking 16 dependent MFMAs per K block) are cut into 32 x 32 workgroup tiles instead
  static const int few = [] { const char* e = getenv("HYP_TS_FEW"); return e ? atoi(e) : 8; }();
  const bool small_grid = ncols <= few;
  static const bool ts3_on = [] { const char* e = getenv("HYP_TS3"); return !(e && e[0] == '0'); }();
  // (products of a few columns keep the round-2 kernel: they are bound by latency, not by vector work, and the copy of R into its
  //  padded square -- a launch of its own, 36 times per config-2 iteration -- costs them more than the lighter loop returns:
  //  19.8 us per product against 19.1 + 4.5)
  if (ts3_on && !small_grid && (long)a.T * 16 * a.T * 16 * 2 < (1L << 31)) {   // (32-bit element offsets inside a matrix)
    const int LD = 16 * a.T;
    const long LD2 = (long)LD * LD;
    c.ts_ws.ensure(((size_t)ncols + 1) * LD2 * sizeof(double));
    double* Rp = c.ts_ws.d();
    double* Z = Rp + LD2;
    hipLaunchKernelGGL(ts_pad_r_kernel, dim3((unsigned)((LD2 + 255) / 256)), dim3(256), 0, c.stream, side, LD, R, Rp);
    a.R = Rp;
    const int b1 = small_grid ? 2 : bt1, b2 = small_grid ? 2 : bt2;
    a.A = arr; a.lda = lda; a.C = Z; a.ldc = 0;
    a.nb = (a.T + b1 - 1) / b1;
    unsigned grid = (unsigned)(((ncols + 7) / 8) * a.nb * a.nb * 8);
    if (b1 == 2) hipLaunchKernelGGL((psd_ts3_kernel<1, 2>), dim3(grid), dim3(TS_THREADS), 0, c.stream, a);
    else if (b1 == 3) hipLaunchKernelGGL((psd_ts3_kernel<1, 3>), dim3(grid), dim3(TS_THREADS), 0, c.stream, a);
    else hipLaunchKernelGGL((psd_ts3_kernel<1, 4>), dim3(grid), dim3(TS_THREADS), 0, c.stream, a);
    a.A = Z; a.lda = 0; a.C = prod; a.ldc = ldp;
    a.nb = (a.T + b2 - 1) / b2;
    grid = (unsigned)(((ncols + 7) / 8) * (a.nb * (a.nb + 1) / 2) * 8);
    if (b2 == 2) hipLaunchKernelGGL((psd_ts3_kernel<2, 2>), dim3(grid), dim3(TS_THREADS), 0, c.stream, a);
    else if (b2 == 3) hipLaunchKernelGGL((psd_ts3_kernel<2, 3>), dim3(grid), dim3(TS_THREADS), 0, c.stream, a);
    else hipLaunchKernelGGL((psd_ts3_kernel<2, 4>), dim3(grid), dim3(TS_THREADS), 0, c.stream, a);
    HYP_CHECK(hipGetLastError());
    return;
  }
  a.A = arr; a.lda = lda; a.C = zws; a.ldc = 0;
  ts_launch<1>(c, a, small_grid ? 2 : bt1);
  a.A = zws; a.lda = 0; a.C = prod; a.ldc = ldp;
  ts_launch<2>(c, a, small_grid ? 2 : bt2);
  HYP_CHECK(hipGetLastError());
}

}  // namespace hyp
