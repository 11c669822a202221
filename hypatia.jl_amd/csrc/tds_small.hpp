// The refined triangular solve of a diagonal block of at most 64 rows as a WAVEFRONT program (one wavefront per 16 right-hand
// sides, nothing shared): x0 = op(D) y with the stored inverse D of the block, then `refine` steps r = y - op(T) x, x += op(D) r
// against the factor T itself (substitution's backward error).  In the MFMA's C/D layout register r of row tile t IS k-chunk
// 4 t + r of the second operand (row q + 4 r of the tile sits in the lanes with lane >> 4 = q, which is where the operand wants
// k = 4 kk + q), so x, the residual and the correction never leave the registers -- no LDS, no barrier; the <= 40 + 40 operand
// entries per lane of D and T are requested up front.
// Used by trsm_diag_refined_small_kernel (dense.hip: ldiv! on cone matrices of side <= 64, /root/reference/src/Cones/
// epinormspectral.jl:141, 156, 224, 251 ...) and by the one-workgroup kernels of the spectral cone (ens_fused.hip), which run the
// same program on columns they hold in LDS: same operations in the same order, same bits.
#pragma once
#include "hyp_internal.hpp"

namespace hyp {

constexpr int TDS_NOPS = 40;                                    // 4 + 8 + 12 + 16 k-steps of a 4-tile triangle

template <bool LOWER>
__device__ __forceinline__ constexpr int tds_idx(int t, int kk) {   // position of (tile t, k-step kk) among the steps a triangle keeps
  int n = 0;
  for (int a = 0; a < 4; ++a)
    for (int k = 0; k < 16; ++k) {
      const bool keep = LOWER ? (k < 4 * a + 4) : (k >= 4 * a);
      if (a == t && k == kk) return keep ? n : -1;
      if (keep) ++n;
    }
  return -1;
}

// raw operand entries of this lane: dop = op(D) (the stored inverse: dinv_blk, its transpose NB * NB doubles behind it), top = op(T);
// TRANS: op = transpose (forward solve with the upper factor), else the backward solve.  Addresses clamped, always valid.
template <bool TRANS>
__device__ __forceinline__ void tds_load_ops(const double* __restrict__ T, long ldt, const double* __restrict__ dinv_blk, int nb,
                                             double (&dop)[TDS_NOPS], double (&top)[TDS_NOPS]) {
  constexpr bool LOWER = TRANS;
  const int lane = threadIdx.x & 63, q = lane >> 4, nn = lane & 15;
  const double* Dop = dinv_blk + (TRANS ? (long)NB * NB : 0);
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      const int id = tds_idx<LOWER>(t, kk);
      if (id >= 0) {
        const int rc = min(16 * t + nn, nb - 1), kc = min(4 * kk + q, nb - 1);   // (clamped: always a valid address)
        dop[id] = Dop[(long)kc * NB + rc];
        top[id] = TRANS ? T[(long)rc * ldt + kc] : T[(long)kc * ldt + rc];
      }
    }
}
// entries outside the block or the triangle -> 0; top negated (the residual is y - op(T) x)
template <bool TRANS>
__device__ __forceinline__ void tds_mask_ops(int nb, double (&dop)[TDS_NOPS], double (&top)[TDS_NOPS]) {
  constexpr bool LOWER = TRANS;
  const int lane = threadIdx.x & 63, q = lane >> 4, nn = lane & 15;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      const int id = tds_idx<LOWER>(t, kk);
      if (id >= 0) {
        const int r = 16 * t + nn, k = 4 * kk + q;
        const bool ok = (r < nb) && (k < nb) && (LOWER ? (k <= r) : (k >= r));
        dop[id] = ok ? dop[id] : 0.0;
        top[id] = ok ? -top[id] : 0.0;
      }
    }
}
// x = op(T)^-1 y for the 16 columns of this wavefront: y[t][r], x[t][r] = row 16 t + (lane >> 4) + 4 r of column lane & 15
template <bool TRANS>
__device__ __forceinline__ void tds_apply(const double (&dop)[TDS_NOPS], const double (&top)[TDS_NOPS], int nb, int refine, const d4_t (&y)[4],
                                          d4_t (&x)[4]) {
  constexpr bool LOWER = TRANS;
#pragma unroll
  for (int t = 0; t < 4; ++t) x[t] = (d4_t){0.0, 0.0, 0.0, 0.0};
  // out[t] += sum_kk op[t][kk] v[kk >> 2][kk & 3]: k-steps outside, tiles inside (four independent accumulator chains)
  auto product = [&](const double (&op)[TDS_NOPS], const d4_t (&v)[4], d4_t (&out)[4]) {
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      if (4 * kk >= nb) continue;                             // (wave-uniform: beyond the block)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int id = tds_idx<LOWER>(t, kk);
        if (id >= 0 && 16 * t < nb) out[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(op[id], v[kk >> 2][kk & 3], out[t], 0, 0, 0);
      }
    }
  };
  product(dop, y, x);                                         // x0 = op(D) y
  for (int it = 0; it < refine; ++it) {
    d4_t res[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) res[t] = y[t];
    product(top, x, res);                                     // r = y - op(T) x
    product(dop, res, x);                                     // x += op(D) r
  }
}

// ---- the same program with the operand entries in LDS (one-workgroup kernels, ens_fused.hip): every wavefront of a workgroup
// multiplies with the SAME masked entries of op(D) and op(T) -- they depend on (lane, step) only --, so the workgroup stages them
// once, [2 * TDS_NOPS][64] doubles (lane-contiguous: conflict-free 8-byte reads), and a wavefront keeps only its 16 columns in
// registers (~100 VGPRs instead of ~260: eight wavefronts per workgroup fit).  Same values, same order of operations, same bits.
constexpr int TDS_LDS_DOUBLES = 2 * TDS_NOPS * 64;
template <bool LOWER>
struct TdsTab {
  int t[TDS_NOPS], kk[TDS_NOPS];
  constexpr TdsTab() : t{}, kk{} {
    int n = 0;
    for (int a = 0; a < 4; ++a)
      for (int k = 0; k < 16; ++k)
        if (LOWER ? (k < 4 * a + 4) : (k >= 4 * a)) { t[n] = a; kk[n] = k; ++n; }
  }
};
template <bool TRANS>
__device__ __forceinline__ void tds_stage_ops(const double* __restrict__ T, long ldt, const double* __restrict__ dinv_blk, int nb, double* ops) {
  constexpr bool LOWER = TRANS;
  constexpr TdsTab<LOWER> tab{};
  const double* Dop = dinv_blk + (TRANS ? (long)NB * NB : 0);
  for (int idx = threadIdx.x; idx < TDS_NOPS * 64; idx += blockDim.x) {
    const int id = idx >> 6, lane = idx & 63, q = lane >> 4, nn = lane & 15;
    const int r = 16 * tab.t[id] + nn, k = 4 * tab.kk[id] + q;
    const bool ok = (r < nb) && (k < nb) && (LOWER ? (k <= r) : (k >= r));
    const int rc = min(r, nb - 1), kc = min(k, nb - 1);
    const double dv = Dop[(long)kc * NB + rc];
    const double tv = TRANS ? T[(long)rc * ldt + kc] : T[(long)kc * ldt + rc];
    ops[idx] = ok ? dv : 0.0;
    ops[TDS_NOPS * 64 + idx] = ok ? -tv : 0.0;
  }
}
template <bool TRANS>
__device__ __forceinline__ void tds_apply_lds(const double* ops, int nb, int refine, const d4_t (&y)[4], d4_t (&x)[4]) {
  constexpr bool LOWER = TRANS;
  const int lane = threadIdx.x & 63;
  const double* dop = ops + lane;
  const double* top = ops + TDS_NOPS * 64 + lane;
#pragma unroll
  for (int t = 0; t < 4; ++t) x[t] = (d4_t){0.0, 0.0, 0.0, 0.0};
  auto product = [&](const double* op, const d4_t (&v)[4], d4_t (&out)[4]) {
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      if (4 * kk >= nb) continue;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int id = tds_idx<LOWER>(t, kk);
        if (id >= 0 && 16 * t < nb) out[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(op[id * 64], v[kk >> 2][kk & 3], out[t], 0, 0, 0);
      }
    }
  };
  product(dop, y, x);
  for (int it = 0; it < refine; ++it) {
    d4_t res[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) res[t] = y[t];
    product(top, x, res);
    product(dop, res, x);
  }
}

}  // namespace hyp
