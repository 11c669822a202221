// Internal declarations shared by the translation units of libhypatia_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include <utility>
#include <stdexcept>
#include "gemm_f64.hpp"

namespace hyp {

struct HipError : std::runtime_error {
  int code;
  HipError(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

#define HYP_CHECK(expr)                                                                         \
  do {                                                                                          \
    hipError_t e_ = (expr);                                                                     \
    if (e_ != hipSuccess)                                                                       \
      throw hyp::HipError(-(int)e_ - 1000, std::string(#expr) + ": " + hipGetErrorString(e_) +  \
                                                " at " __FILE__ ":" + std::to_string(__LINE__)); \
  } while (0)

#define HYP_REQUIRE(cond, msg)                                                     \
  do {                                                                             \
    if (!(cond)) throw hyp::HipError(-1, std::string("bad argument: ") + (msg));   \
  } while (0)

// Device buffer of doubles (or raw bytes) owned by the library.
struct DBuf {
  void* p = nullptr;
  size_t bytes = 0;
  bool owned = true;   // false: a view into storage owned elsewhere (a group arena), never freed or regrown here
  DBuf() {}
  explicit DBuf(size_t nbytes) { alloc(nbytes); }
  DBuf(DBuf&& o) noexcept : p(o.p), bytes(o.bytes), owned(o.owned) { o.p = nullptr; o.bytes = 0; o.owned = true; }
  DBuf(const DBuf&) = delete;
  DBuf& operator=(const DBuf&) = delete;
  DBuf& operator=(DBuf&& o) noexcept {
    if (this != &o) { release(); p = o.p; bytes = o.bytes; owned = o.owned; o.p = nullptr; o.bytes = 0; o.owned = true; }
    return *this;
  }
  ~DBuf() { release(); }
  void alloc(size_t nbytes) {
    release();
    if (nbytes == 0) return;
    HYP_CHECK(hipMalloc(&p, nbytes));
    bytes = nbytes;
  }
  void ensure(size_t nbytes) {
    if (nbytes > bytes) alloc(nbytes);
  }
  void release() {
    if (p && owned) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
    owned = true;
  }
  void view(void* ptr, size_t nbytes) {   // drop the own allocation and look at [ptr, ptr + nbytes) instead
    release();
    p = ptr;
    bytes = nbytes;
    owned = false;
  }
  double* d() const { return (double*)p; }
  int* i() const { return (int*)p; }
};

constexpr int NB = 128;   // Cholesky / triangular-solve block size

// Context: one HIP device, one stream; owns scratch and the host<->device staging buffers.
struct Ctx {
  int device = 0;
  unsigned long cone_epoch = 0;   // bumped whenever a cone of this context takes a new point or forgets its data (SysSolver::prelaunch_sqrt_hess)
  hipStream_t stream = nullptr;
  std::string last_error;
  double timers[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  // HIP-event timings of the update_lhs phases, accumulated in ms: [0] sqrt-Hessian products,
  // [1] Schur syrk, [2] Cholesky, [3] number of update_lhs_fact calls, [4] syrk launches
  double kstat[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  hipStream_t stream2 = nullptr;            // helper stream: look-ahead trailing updates of the blocked Cholesky
  hipStream_t stream_primary = nullptr;     // the main stream's handle (c.stream unless a StreamSwap / LaneSwitch section is open)
  // A persistent kernel whose workgroups wait for each other (trsv_onelaunch.hip) must be the only one of its kind on the device: two
  // of them, from two contexts or two PROCESSES sharing a GPU, could each hold CUs the other's unscheduled workgroups need.  The first
  // context to take an advisory lock on the device (flock on a file named after its PCI bus id, held for the context's life) may
  // launch them; every other context on that device uses the launch chains.
  bool persistent_ok = false;
  int device_lock_fd = -1;
  unsigned* ol_abort_host = nullptr;   // pinned, device-visible: raised by a persistent kernel whose poll timed out (trsv_onelaunch.hip) ...
  unsigned* ol_abort_dev = nullptr;    // ... the same word as the device sees it
  void check_persistent_abort();       // throws (and switches the persistent kernels off) if the word is raised
  std::vector<hipEvent_t> ev_pool;          // ordering events between stream and stream2
  hipEvent_t pool_event(size_t i);
  hipEvent_t aux[4] = {nullptr, nullptr, nullptr, nullptr};   // fork / join events of the two-stream sections (not the look-ahead pool)
  hipEvent_t aux_event(int i);
  // Lanes 2, 3, ...: further streams, each with the per-stream scratch a launcher takes from the context, for sections made of
  // SEVERAL independent latency-bound chains (the K bases of a WSOS cone): lane 0 is the main stream, lane 1 the helper stream.
  struct Lane;
  std::vector<Lane*> lanes;
  Lane& lane(int i);
  static int max_lanes();   // HYP_LANES (default 3; 2 = the two streams only)
  GemmScratch gemm_scratch;   // split-K workspace + syrk tile order of the GEMM launcher (per context, never shared)
  GemmScratch gemm_scratch2;  // the same for launches on the helper stream (StreamSwap)
  bool gemv_one = false;   // gemv(): one-right-hand-side products through the multi-column kernels (set by SysSolver::sgemv for its call)
  DBuf scratch;       // general device scratch (gemv partial sums)
  DBuf dscal;         // 64 device doubles for scalar results (dots, counts)
  DBuf stage_a, stage_b;   // device staging for host-pointer entry points
  DBuf potrf_tinv2;        // (the same for factorizations on the helper stream, see StreamSwap)
  DBuf potrf_tinv;         // blocked Cholesky: inverses of the 16 x 16 diagonal tiles, per block step (potrf_upper_batched)
  DBuf work_tri;           // workspace of trtri_upper_batched
  DBuf ts_ws;              // PSD two-sided product: zero-padded copy of the factor + the padded intermediates Z_j (psd_twosided.hip)
  long bk_hybrid_count = 0, bk_guard_trims = 0, bk_plain_count = 0;   // fall-backs behind a failed Cholesky: hybrid / trimmed by the guard / plain (hyp_ctx_bk_stats)
  long plan_builds = 0, plan_builds_one_step = 0;                    // solve plans built / of them with ONE refinement step by the adaptive rule (hyp_ctx_plan_stats)
  int diag_own_cu_lds = -1;   // dynamic LDS that gives the critical-path diagonal-block kernel a CU of its own (-1: not asked yet, 0: refused)
  int trsv_sb = 1024;      // largest super-block of the one-right-hand-side triangular solves (HYP_TRSV_SB; 0 = per-128-block path)
  bool trsv_sb_forced = false;   // HYP_TRSV_SB given: that size, plan from 2 super-blocks on (the round-1 rule)
  // super-block size for an n x n factor, 0 = no plan (per-128-block solves).  About three super-blocks: n = 999 (config 3b) ->
  // 384 (measured: 256: 10.7, 384: 10.4, 512 / no plan: 11.4-11.6 ms per iteration), n >= 3072 -> 1024
  int trsv_plan_sb(int n) const {
    if (trsv_sb <= 0) return 0;
    if (trsv_sb_forced) return n >= 2 * trsv_sb ? trsv_sb : 0;
    if (n < 512) return 0;
    const int third = ((n + 2) / 3 + 127) / 128 * 128;
    return third < trsv_sb ? third : trsv_sb;
  }
  int* h_info = nullptr;    // pinned host word(s)
  static constexpr int H_INFO_FACT = 8192;   // the system solver's factorization info (read back asynchronously: nobody else's slot)
  double* h_pinned = nullptr;   // pinned host staging (small vectors / scalars)
  size_t h_pinned_n = 0;
  static constexpr int H_SC_N = 128;         // pinned mirror of the direction solves' device scalars (SysSolver::d_sc), behind the general staging
  double* h_sc() const { return h_pinned + h_pinned_n; }
  double* h_sc_dev = nullptr;   // the same block as the device addresses it (written by publish_scalars_kernel)
  double* h_stage = nullptr;    // growable pinned staging for large caller-owned vectors (see stage_host)
  size_t h_stage_n = 0;
  double* stage_host(size_t n_doubles);
  Ctx(int dev);
  ~Ctx();
  void sync() {
    HYP_CHECK(hipStreamSynchronize(stream));
    if (ol_abort_host && *ol_abort_host) check_persistent_abort();
  }
  void h2d(void* dst, const void* src, size_t bytes) {
    if (bytes) HYP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream));
  }
  void d2h(void* dst, const void* src, size_t bytes) {
    if (bytes) HYP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, stream));
  }
  void d2d(void* dst, const void* src, size_t bytes) {
    if (bytes) HYP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, stream));
  }
  void zero(void* dst, size_t bytes) {
    if (bytes) HYP_CHECK(hipMemsetAsync(dst, 0, bytes, stream));
  }
};

// Run a section on the helper stream: every launcher takes the stream from ctx.stream, so the two members are
// exchanged for the lifetime of the guard (restored on scope exit, also when a HIP error throws).
struct StreamSwap {
  Ctx& c;
  // (the GEMM launcher's split-K workspace follows the stream: two products in flight on the two streams must not share it)
  // (so does the blocked Cholesky's record of tile inverses: two factorizations may be in flight on the two streams)
  explicit StreamSwap(Ctx& ctx) : c(ctx) { std::swap(c.stream, c.stream2); std::swap(c.gemm_scratch, c.gemm_scratch2); std::swap(c.potrf_tinv, c.potrf_tinv2); }
  ~StreamSwap() { std::swap(c.stream, c.stream2); std::swap(c.gemm_scratch, c.gemm_scratch2); std::swap(c.potrf_tinv, c.potrf_tinv2); }
  StreamSwap(const StreamSwap&) = delete;
  StreamSwap& operator=(const StreamSwap&) = delete;
};

struct Ctx::Lane {
  hipStream_t s = nullptr;
  GemmScratch gs;
  DBuf tinv;
  hipEvent_t done = nullptr;
};

// Run a section on lane i (0: where we are, 1: the helper stream, >= 2: a stream of its own); join_lanes makes the main stream wait
// for everything queued on the lanes used since fork_lanes.
struct LaneSwitch {
  Ctx& c;
  int i;
  explicit LaneSwitch(Ctx& ctx, int lane) : c(ctx), i(lane) { flip(); }
  ~LaneSwitch() { flip(); }
  LaneSwitch(const LaneSwitch&) = delete;
  LaneSwitch& operator=(const LaneSwitch&) = delete;
 private:
  void flip() {
    if (i == 1) { std::swap(c.stream, c.stream2); std::swap(c.gemm_scratch, c.gemm_scratch2); std::swap(c.potrf_tinv, c.potrf_tinv2); }
    else if (i >= 2) { Ctx::Lane& L = c.lane(i); std::swap(c.stream, L.s); std::swap(c.gemm_scratch, L.gs); std::swap(c.potrf_tinv, L.tinv); }
  }
};
void fork_lanes(Ctx& c, int nlanes);   // lanes 1 .. nlanes - 1 wait for what is queued on the main stream now
void join_lanes(Ctx& c, int nlanes);   // the main stream waits for lanes 1 .. nlanes - 1

// ---------------------------------------------------------------------------------------------
// dense.hip : factorizations, triangular solves, level-1/2 kernels (all on ctx.stream, device ptrs)
// ---------------------------------------------------------------------------------------------

// Blocked upper Cholesky A = U'U in place (upper triangle of each n x n matrix of the batch).
// dinv receives, for every NB-diagonal block, inv(U_kk) followed by its transpose:
// layout [batch][block][2][NB*NB] col-major, ld NB (DINV_BLK doubles per block).
// d_info[b] = 0 or the 1-based index of the first non-positive pivot (LAPACK dpotrf convention).
// dinv == nullptr: factor only (feasibility checks); potrf_invert_diag_blocks produces the block inverses later.
void potrf_upper_batched(Ctx& c, int n, double* A, long lda, long strideA, int batch, double* dinv, int* d_info, int kb_stop = -1);
// (kb_stop >= 0: only block steps 0 .. kb_stop - 1, each with its whole trailing update: rows < NB kb_stop hold rows of the factor, the
//  trailing block its Schur complement -- the start of the hybrid factorization of BKFact::factor_from)
// The Schur complement C = A'A (A: K x n, ld lda; C n x n, ld n) and its factorization F = chol(C) in two column groups, the leading
// n1 x n1 block factored (queues `lane` + the helper stream) while the product forms the columns >= n1 (dense.hip)
bool potrf_split_ok(int n, int n1);
void schur_split_begin(Ctx& c, int n, int K, const double* A, long lda, double* C, double* F, int n1, int* d_info, hipStream_t lane,
                       hipEvent_t left_ready, hipEvent_t left_done);
void schur_split_finish(Ctx& c, int n, const double* C, double* F, int n1, double* dinv, int* d_info, hipEvent_t left_done);
void potrf_invert_diag_blocks(Ctx& c, int n, double* A /* factored */, long lda, long strideA, int batch, double* dinv,
                              long strideD = 0 /* doubles between the batch members' dinv; 0: dinv_elems(n) */);
constexpr long DINV_BLK = 2L * NB * NB;
inline size_t dinv_elems(int n) { return (size_t)((n + NB - 1) / NB) * DINV_BLK; }

// x <- U^-T x (trans = true) or U^-1 x (trans = false), U upper triangular n x n with inverted
// diagonal blocks dinv; nrhs right-hand sides, x col-major with leading dimension ldx.
void trsv_upper(Ctx& c, int n, const double* U, long ldu, const double* dinv, bool trans, double* x);
int trsm_refine_steps();   // HYP_TRSM_REFINE (default 2): refinement steps of the solves with inverted diagonal blocks
void trsm_upper_left(Ctx& c, int n, int nrhs, const double* U, long ldu, const double* dinv, bool trans, double* X,
                     long ldx, double* work /* NB x nrhs */);
// the forward solve (U'^-1) for a batch of equally sized factors: member b at U + b strideU, dinv + b strideD, X + b strideX
void trsm_upper_left_fwd_batched(Ctx& c, int n, int nrhs, const double* U, long ldu, long strideU, const double* dinv, long strideD,
                                 double* X, long ldx, long strideX, int batch);
// One-right-hand-side solves with a LARGE upper Cholesky factor (the potrs of qrchol.jl:68).  The
// factor is cut into super-blocks of sb rows; build() inverts the diagonal super-blocks and keeps a
// transposed copy of U, so that every step of solve() is a set of coalesced column dot products:
// x_b = Binv y_b followed by `refine` steps of fixed-precision iterative refinement against the
// factor itself (the result has substitution's backward error), then one rank-sb update of the rest.
struct TriSolvePlan {
  int n = 0, sb = 0, refine = 2;
  DBuf Binv, BinvT, UT, work, work2, work_n;
  bool ready(int n_) const { return n == n_ && n_ > 0; }
  void invalidate() { n = 0; }
  void build(Ctx& c, int n_, const double* U, long ldu, const double* dinv);
  // Round 5 (trsv_onelaunch.hip): the same sweeps -- the same sums, bitwise -- as ONE launch each, or one for both sweeps of a
  // Cholesky potrs; products hand their vectors over through an arena of self-flagging words.  HYP_TRSV_ONE_LAUNCH=0: the launch chains.
  DBuf ol_rounds[4], ol_arena;    // round tables per number of refinement steps (0 .. 3): the adaptive rule below switches between them
  bool ol_ok = false;
  bool ol_have[4] = {false, false, false, false};
  int ol_n = 0, ol_sb = 0, ol_set = 0;
  long ol_ldu = 0, ol_asz = 0;
  int ol_first[4][3] = {}, ol_n0[4][3] = {}, ol_n1[4][3] = {};
  void ol_prepare(Ctx& c, long ldu);
  // Round 5: the second refinement step is dropped where it cannot change a digit.  build() measures the quality of the inverted
  // super-blocks on two probe vectors, rho = max_b || v - T_b' (B_b' v) ||_inf; a solve x0 = B b then has relative error <= rho and
  // one refinement step leaves rho^2: with rho <= HYP_TRSV_ADAPT_TOL (1e-10; the probes may underestimate the norm by a factor)
  // the second step moves nothing above 1e-20.  refine_req = what was asked for (HYP_TRSV_REFINE, default 2), refine = what runs.
  int refine_req = 2;
  int owner_class = 0;            // 0: a system solver's factor (Schur matrix, SymIndef), 1: a cone's Hessian factor (HYP_TRSV_ADAPT selects)
  double rho = -1.0;              // the last build's measurement (-1: not measured)
  DBuf probe_ws;
  void measure_quality(Ctx& c, const double* U, long ldu);
  // only on the context's main stream: two such launches side by side on two streams could each hold CUs the other's wavefronts wait for
  bool ol_fit[4] = {false, false, false, false};   // per number of right-hand sides: the instance's 512 workgroups are resident at once on this device (ol_fits)
  static bool ol_fits(Ctx& c, int nr);
  bool ol_usable(const Ctx& c, long ldu, int nr) const { return ol_ok && nr >= 1 && nr <= 3 && ol_fit[nr] && c.persistent_ok && refine >= 0 && refine <= 3 && ol_have[refine] && ldu == ol_ldu && c.stream == c.stream_primary; }
  void ol_sweep(Ctx& c, const double* U, int which, double* x, long ldx, double* x3, int nr);
  // both sweeps, x <- (U'U)^-1 x on nr = 1, 2 or 3 columns (x3 != nullptr: the third column lives there instead of x + 2 ldx)
  void solve_both(Ctx& c, const double* U, long ldu, double* x, long ldx, int nr, double* x3 = nullptr);
  void solve(Ctx& c, const double* U, long ldu, bool trans, double* x);
  void solve_multi(Ctx& c, const double* U, long ldu, bool trans, double* x, long ldx, int nr);   // nr <= 2 right-hand sides
  void solve_multi3(Ctx& c, const double* U, long ldu, bool trans, double* x, long ldx, double* x3);   // the pair x[:, 0:2] and a third vector x3 together
  void solve_n(Ctx& c, const double* U, long ldu, bool trans, double* x, long ldx, int nc);   // nc <= 8 columns, one launch per product (not solve()'s bits)
};
// bunchkaufman.hip : symmetric indefinite factorization with rook pivoting, the reference's fallback of a failed
// Cholesky (symm_fact!, dense.jl:164-165; posdef_fact_copy!, dense.jl:194-215).  P A P' = U' D U with U unit upper
// triangular (written over the upper triangle of A, unit diagonal explicit) and D block diagonal with 1x1 / 2x2 blocks,
// so the solve is gather(P) -> U'^-1 -> D^-1 -> U^-1 -> scatter(P') around the same triangular solves as the Cholesky.
struct BKFact {
  int n = 0;
  int n_2x2 = 0;                  // number of 2x2 pivot blocks of the last factorization
  DBuf dd, de, blk, perm;         // diagonal / off-diagonal of D, block marks (0: 1x1, 1 / 2: rows of a 2x2), P as a gather map
  DBuf state, wl, tmp, tr;        // tr: n x n work matrix (the factorization runs on the transposed triangle)
  DBuf guard;                     // bk_after_failed_cholesky: per kept block step min pivot^2 / max row entry^2
  int k0_used = 0;                // column the last bk_after_failed_cholesky started its rook-pivoted elimination from (0: plain)
  int guard_trimmed = 0;          // block steps the growth guard refused to keep in that call
  // A: upper triangle in, U out.  dinv (dinv_elems(n) doubles, may be null) receives the inverted diagonal
  // blocks for trsv_upper / trsm_upper_left / TriSolvePlan.  Returns LAPACK's info: 0 or the 1-based index of the
  // first exactly singular pivot (issuccess(fact) = info == 0).  Synchronizes.
  int factor(Ctx& c, int n_, double* A, long lda, double* dinv);
  // Round 4.  The matrices a failed Cholesky hands over fail at their LAST pivots (config 5: 4841 - 4845 of 4845, every one of 20
  // fall-backs): the leading k0 columns have a perfectly good Cholesky elimination.  A holds rows 0 .. k0 - 1 of that factor
  // (potrf_upper_batched(..., kb_stop)) and the Schur complement behind them; the rows become rows of the unit factor (U_ij / U_ii,
  // D_ii = U_ii^2) and the rook-pivoted elimination runs on the trailing block only (its interchanges permute the columns of the
  // rows above, as dsytrf_rook's do).  P A P' = U' D U as from factor(): the same solves apply.  22 ms -> the trailing block's share.
  int factor_from(Ctx& c, int n_, double* A, long lda, double* dinv, int k0);
  double* gather(Ctx& c, const double* x, long ldx, int nr);          // tmp[:, r] = P x[:, r]; returns tmp (ld n)
  void dsolve(Ctx& c, double* y, long ldy, int nr);                   // y <- D^-1 y
  void scatter(Ctx& c, const double* y, double* x, long ldx, int nr); // x[:, r] = P' y[:, r] (y with ld n)
  void solve(Ctx& c, const double* U, long ldu, const double* dinv, double* x, long ldx, int nr, DBuf& trsm_work);
};
// symm_fact! behind a failed Cholesky (dense.jl:194-215, the second link of posdef_fact_copy!): A = the matrix again (upper triangle),
// chol_info = the 1-based pivot the Cholesky failed at (0: unknown / forced: the plain rook-pivoted factorization from column 0).
// With the failing pivot in block step kb >= 1 the first kb block steps are redone as Cholesky steps and only the trailing block is
// eliminated with rook pivoting (BKFact::factor_from; HYP_BK_HYBRID=0: always from column 0).  Returns BKFact's info.
// Round 5: a kept block step must pass the growth guard (pivot^2 >= n eps max|a_ii|, row entries^2 <= 16 max|a_ii|); the steps from the
// first offender on are not kept (the matrix is taken again from A_src, leading dimension ld_src: the caller's untouched copy).
int bk_after_failed_cholesky(Ctx& c, BKFact& bk, int n, double* A, long lda, double* dinv, int* d_info_scratch, int chol_info,
                             const double* A_src, long ld_src);
// Y[:, r] = alpha op(A) X[:, r] + beta Y[:, r] for r < nr <= 2: one pass over A serves all right-hand sides
void gemv_multi(Ctx& c, bool trans, int m, int n, int nr, double alpha, const double* A, long lda, const double* X, long ldx, double beta,
                double* Y, long ldy);
// both products of one matrix in one pass over it (directions_multi.hip): Yn = A Xn + beta_n Yn, Yt = A' Xt + beta_t Yt, nr = 1 or 2 columns
bool gemv_both_ok(int m, int n, const double* A, long lda);
void gemv_both(Ctx& c, int m, int n, int nr, const double* A, long lda, const double* Xn, long ldxn, double beta_n, double* Yn, long ldyn,
               const double* Xt, long ldxt, double beta_t, double* Yt, long ldyt);
// explicit inverse of an upper triangular matrix from its inverted diagonal blocks: Uinv (upper, full
// storage, strictly-lower part zero).  Used for the small cone matrices.
void trtri_upper_batched(Ctx& c, int n, const double* U, long ldu, long strideU, const double* dinv, long strideD,
                         double* Uinv, long ldi, long strideI, int batch, DBuf* ws_override = nullptr);

// y = alpha * op(A) x + beta * y, A m x n col-major.  Deterministic (fixed reduction tree).
void gemv(Ctx& c, bool trans, int m, int n, double alpha, const double* A, long lda, const double* x, double beta,
          double* y);

// level-1 helpers (device scalars are read back through ctx pinned memory by the callers)
void dev_dot(Ctx& c, int n, const double* x, const double* y, double* d_out);          // *d_out = <x,y>
struct DotSpecs {   // several dot products as one launch (dev_dots): *out[k] = <x[k], y[k]> over n[k] entries
  int count = 0;
  int n[8];
  const double* x[8];
  const double* y[8];
  double* out[8];
  void add(int n_, const double* x_, const double* y_, double* out_) { n[count] = n_; x[count] = x_; y[count] = y_; out[count] = out_; ++count; }
};
void dev_dots(Ctx& c, const DotSpecs& sp);
struct ZeroSlots {   // up to 16 single doubles set to zero by one launch (dev_zero_slots)
  int count = 0;
  double* p[16];
  void add(double* q) { p[count++] = q; }
};
void dev_zero_slots(Ctx& c, const ZeroSlots& z);
void dev_axpby(Ctx& c, int n, double a, const double* x, double b, double* y);          // y = a x + b y
void dev_scale_copy(Ctx& c, int n, double a, const double* x, double* y);               // y = a x
void dev_transpose(Ctx& c, int m, int n, const double* A, long lda, double* B, long ldb, int batch, long strideA,
                   long strideB);   // B = A' (B is n x m)
void dev_fill_identity(Ctx& c, int n, double* A, long lda);
void dev_symmetrize_from_upper(Ctx& c, int n, double* A, long lda, int batch, long stride);   // copytri!(A,'U')
void dev_zero_strict_lower(Ctx& c, int n, double* A, long lda, int batch, long stride);

// svec <-> smat for batches of columns: arr is (d x ncols, ld = ldarr); mats is ncols blocks of s x s
// (col-major, contiguous).  unpack fills BOTH triangles (symmetric); pack reads the upper triangle.
void svec_unpack(Ctx& c, int side, int ncols, const double* arr, long ldarr, double* mats);
// the same for ngroups x group columns: column (g, k) of the packed input at arr + g * ldg + k * ldarr, matrices written back to back
void svec_unpack_grouped(Ctx& c, int side, int ngroups, int group, const double* arr, long ldg, long ldarr, double* mats);
void svec_pack(Ctx& c, int side, int ncols, const double* mats, double* arr, long ldarr, double scale);

// gemm wrapper on ctx.stream
void gemm(Ctx& c, bool transa, GemmArgs a);

}  // namespace hyp
