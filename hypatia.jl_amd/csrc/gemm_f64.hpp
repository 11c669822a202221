// FP64 MFMA GEMM family for gfx950 (MI355X / CDNA4).
//
// Replaces the BLAS-3 call sites of the reference hot path:
//   dsyrk('U','T')  src/linearalgebra/dense.jl:80-86 (outer_prod!), called from
//                   src/Solvers/systemsolvers/qrchol.jl:234 (Schur assembly)
//   dgemm           qrchol.jl:244-245 (lhs += G_k' (H G_k)), wsosinterpnonnegative.jl:104-107,142-145
//   dtrsm/dtrmm     as GEMMs against explicitly inverted triangular blocks (chol.hip, cone kernels)
//
// One kernel template.  C (col-major, M x N) = alpha * op(A) * B + beta * C with
//   B : K x N col-major (K contiguous),
//   op(A) = A^T with A : K x M col-major (TRANSA, K contiguous)  -- the "TN" form every
//           Julia column-major Gram product takes -- or op(A) = A, A : M x K col-major (NN).
// Block tile 128 x 128 x 16, 256 threads = 4 wavefronts (2 x 2), each wavefront owns a 64 x 64
// sub-tile = 4 x 4 v_mfma_f64_16x16x4_f64 accumulators (128 VGPRs).  The MFMA is issued with the
// B fragment as its row operand and the A fragment as its column operand, so the accumulator
// register r of lane l holds C[m = l&15, n = (l>>4) + 4r] of the 16 x 16 tile: 16 consecutive
// lanes store 128 contiguous bytes of a column of C (coalesced in the column-major output).
// LDS tiles are [row][k] with a row stride of 18 doubles: the 16-lane x 2-k footprint of one
// ds_read_b64 lane group then covers all 32 eight-byte banks exactly once.
// Summation order over k is fixed by (K tile, k step, MFMA k slot): results are deterministic
// and independent of grid size.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hyp {

typedef double d4_t __attribute__((ext_vector_type(4)));

enum GemmTri : int {
  GEMM_FULL = 0,        // all tiles
  GEMM_UPPER = 1,       // only tiles/elements with col >= row are computed and stored (syrk 'U')
  GEMM_UPPER_RECT = 2,  // same element rule for a rectangular C (M <= N, an upper trapezoid): full tile grid, tiles below the diagonal exit
};

// K-range restriction for triangular operands (tile-granular skipping of structural zeros).
enum GemmKRange : int {
  KR_ALL = 0,
  KR_LE_M = 1,   // op(A)[m,k] == 0 for k > m   (op(A) lower triangular)  -> k <= tile m end
  KR_GE_M = 2,   // op(A)[m,k] == 0 for k < m   (op(A) upper triangular)  -> k >= tile m start
  KR_LE_N = 3,   // B[k,n] == 0 for k > n       (B upper triangular)      -> k <= tile n end
  KR_GE_N = 4,   // B[k,n] == 0 for k < n       (B lower triangular)      -> k >= tile n start
};

struct GemmArgs {
  int M, N, K;
  const double* A; long lda; long strideA;
  const double* B; long ldb; long strideB;
  double* C; long ldc; long strideC;
  double alpha, beta;
  int tri;      // GemmTri
  int krange;   // GemmKRange
  int batch;
  int tiles_m, tiles_n;
  // composite row index of C (0 = off): row m is stored at (m / cm_blk) * cm_stride + (m % cm_blk).
  // Lets an M-stacked batch (rows = (matrix j, row a)) write each j into its own contiguous block.
  int cm_blk; long cm_stride;
  int tile_hint;   // 0 = automatic, 64 / 128 = force the block tile edge
  int epi;         // 0: C = alpha*acc + beta*C ; 1: C = (alpha*acc)^2 + beta*C (Hadamard square, WSOS Hessian)
  int tag;         // 1 = the Schur-complement syrk (own kernel symbol, so profiles can tell it apart)
  int hiprio;      // 1: the wavefronts raise their issue priority (s_setprio 3): a critical-path product beside a bulk one (potrf look-ahead)
  // split-K (set by the launcher): blockIdx.z = slice; slices write raw partial sums to `part`
  // (slice-major copies of C's layout, ldc = part_ld) and a second kernel adds them in slice order
  int tri_off;     // upper forms keep elements with row <= col + tri_off (0 except for off-diagonal strips)
  int splitk_req;  // requested split-K slices (0 / 1: the launcher decides)
  int splitk; int kchunk; double* part; long part_ld; long part_stride;
  int vec2;        // set by the launcher: operands are 16-byte aligned with even leading dimensions
  const int* tile_map;   // optional (tm, tn) per blockIdx.x: XCD-aware tile order (set by the launcher)
  // tail of the split-K launch (set by the launcher): the workgroups of the last, partly filled round -- slice splitk_base - 1
  // of the tiles at launch positions >= tail_first -- are cut into tail_q sub-slices of tail_chunk k each, which run as the
  // extra z layers splitk_base .. splitk - 1 (positions < tail_first leave at once there) and write partial slots of their own
  int splitk_base, tail_q, tail_first, tail_chunk;
  // fused split-K reduction of the Schur syrk (set by the launcher): the slices of a tile count their arrivals in tile_cnt[launch
  // position]; the last one to arrive adds the partial sums in slice order (the sums of splitk_reduce_kernel) and writes C
  int* tile_cnt;
};

// Launcher state that belongs to the caller's context (one per hyp_ctx, i.e. per device and stream pair): the split-K
// partial-sum workspace and the cached XCD-aware tile order of the Schur syrk.  Without it (nullptr) the launcher
// uses neither split-K nor a tile map.  Not shared between contexts; released by its owner.
struct GemmScratch {
  double* splitk_ws = nullptr;
  size_t splitk_ws_bytes = 0;
  int* tile_map = nullptr;     // (tm, tn) per launch position, then the inverse: launch position per tile tm + tn * T
  int tile_map_T = -1;
  int* tile_cnt = nullptr;     // arrival counters of the fused split-K reduction (zero between launches: the last arrival resets its tile's)
  long tile_cnt_n = 0;
  // the same two tables for upper TRAPEZOIDS (GEMM_UPPER_RECT: row groups of the overlapped exchange, column groups of the split
  // factorization), sixteen kept by (tiles_m, tiles_n, tri_off); the inverse is indexed tm + tn * tiles_m
  struct TrapMap { int tiles_m = 0, tiles_n = 0, tri_off = 0; long nblk = 0; int* map = nullptr; unsigned long used = 0; };
  TrapMap trap[16];
  unsigned long trap_clock = 0;
  void release() {
    for (TrapMap& t : trap) { if (t.map) (void)hipFree(t.map); t = TrapMap(); }
    if (splitk_ws) (void)hipFree(splitk_ws);
    if (tile_map) (void)hipFree(tile_map);
    if (tile_cnt) (void)hipFree(tile_cnt);
    splitk_ws = nullptr; splitk_ws_bytes = 0; tile_map = nullptr; tile_map_T = -1; tile_cnt = nullptr; tile_cnt_n = 0;
  }
};

// launch on `st`; returns the HIP launch status
hipError_t gemm_f64_launch(hipStream_t st, bool transa, GemmArgs a, GemmScratch* scratch = nullptr);
// block columns [c0, c1) of the upper Schur syrk described by `a` (c0 = 0: the leading block; c0 > 0: the rest, c1 = a.N)
hipError_t schur_syrk_cols(hipStream_t st, GemmArgs a, int c0, int c1, GemmScratch* scratch);
// the thin last columns (N mod 128 <= 32 of them) of the upper Schur syrk described by `a`, every row up to the diagonal, by the
// skinny-product kernel; *N0 = the first column it took (a.N: none -- the caller's tiles cover everything)
hipError_t schur_syrk_edge(hipStream_t st, const GemmArgs& a, int* N0);

}  // namespace hyp
