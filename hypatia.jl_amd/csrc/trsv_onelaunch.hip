// One-launch triangular sweeps (round 5): the super-block solves of TriSolvePlan (dense.hip / directions_multi.hip) without a kernel
// boundary between their products.
//
// What it replaces.  TriSolvePlan::solve / solve_multi / solve_multi3 run one sweep (U' y = x or U x = y, the two halves of the
// potrs of /root/reference/src/Solvers/systemsolvers/qrchol.jl:66-69 and of every ldiv! on a cone's Hessian factor, Cones.jl:113-118)
// as a chain of column-product launches: per super-block t = B x_b, `refine` x (e = x_b - T t, t += B e), then the rank-sb update
// of the rest -- 29 dependent launches per sweep at n = 5000 (5 super-blocks of 1024 rows, two refinement steps), 58 per potrs, each
// 4.7-5.5 us although its work is a fraction of a microsecond (profiles/r04_iteration_timeline.txt).
//
// How.  ONE launch per sweep -- or one for both sweeps of a Cholesky potrs -- of 2 x OL_W resident wavefronts.  A wavefront computes
// the SAME column product the launch chain's wavefronts computed (the per-lane loads, the accumulators, the shuffle tree and the
// final fused multiply-add are those of coldot_batched_kernel / coldot2_batched_kernel / coldot3_batched_kernel), so every number
// -- and every iterate of a solve -- is bitwise what the launch chain gives (tests/test_hip_switches.py, HYP_TRSV_ONE_LAUNCH=0/1).
// What changes is the hand-over between dependent products: every vector a later product reads lives in an arena of 8-byte words
// initialised to a sentinel (all ones: a NaN no arithmetic produces here, results are canonicalised before publication); the producer
// publishes each entry with ONE relaxed agent-scope 8-byte store (write-through), the consumer polls with relaxed agent-scope loads
// until no word it needs is the sentinel -- the data is its own flag (guide: Guideline 16 form R2, "8-B agent atomics both sides").
// No fence, no flag, no barrier, no LDS.  In-place vectors (x -= ... of the rank-sb updates) become VERSIONS: one arena vector per
// update, so a word is written exactly once per launch.
//
// Work assignment.  Static: the sweep is a list of ROUNDS per wavefront class; in round r wavefront w of the class owns column
// col0 + w of that round's product (rounds hold at most OL_W columns).  Class 0 ("chain") runs the products on the critical path
// of a super-block -- the five diagonal-block products --, class 1 ("update") the rank-sb updates of the rest, whose first chunk
// (the next super-block's rows) comes first; the later chunks then stream from HBM underneath the next super-block's chain.
// A round's inputs are produced by earlier rounds of either class only, and all 2 x OL_W wavefronts are resident (512 workgroups of
// four wavefronts: two per CU), so the polls always terminate; a poll that sees no progress for two seconds traps (a HIP error at the
// caller's next synchronisation, never a hang).
//
// The matrix column of a round is requested BEFORE the poll: its latency (HBM for the updates, L2 for the diagonal blocks) overlaps
// the hand-over's.  The arena of a launch is cleared by the PREVIOUS launch of the same plan (two arena sets, alternating; stream
// order makes the other set idle), so no memset node sits between launches.
#include "hyp_internal.hpp"

#include <cstdio>
#include <cstdlib>
#include <vector>

namespace hyp {

namespace {

constexpr int OL_W = 1024;                       // wavefronts per class = columns per round
constexpr unsigned long long OL_SENT = ~0ULL;    // never-published marker (a NaN with every payload bit set)
constexpr unsigned long long OL_QNAN = 0x7FF8000000000000ULL;

typedef __attribute__((address_space(1))) unsigned long long gu64;

struct OlRound {     // one product out[j] = base[j] + alpha * sum_i M[i, j] v[i] over ncols columns (or a copy-in round: msel = 4)
  long moff;         // element offset of column 0 / row 0 of the operand inside the matrix msel names
  long ld;           // its leading dimension
  int msel;          // 0: U, 1: UT, 2: Binv, 3: BinvT, 4: none (copy the caller's vector into the arena)
  int m;             // rows of the product (<= 1024)
  int mode;          // 0: all rows, 1: rows <= column, 2: rows >= column
  int ncols, col0;   // this round's columns [col0, col0 + ncols) of the product, wavefront w takes col0 + w
  int v_off;         // arena offset of v[0]
  int base_off;      // arena offset of base[0] (indexed by column), -1: none
  int out_off;       // arena offset of out[0] (indexed by column)
  int x_off;         // caller-vector offset of out[0] for the plain copy of the result (final values), -1: none; copy-in: source offset
  float alpha;
  int pad_;
};

struct OlArgs {
  const double* U; const double* UT; const double* Binv; const double* BinvT;
  int nrounds0, nrounds1;    // rounds of class 0, then class 1, in the table passed beside this struct
  unsigned long long* arena; // this launch's set; right-hand side r at arena + r * asz
  unsigned long long* clear; // the other set (cleared here for the next launch), 3 * asz words
  long asz;
  double* x; long ldx; double* x3;   // the caller's vectors: columns 0, 1 at x, x + ldx; column 2 at x3
  int direct_poll;                   // HYP_TRSV_OL_POLL=1: no one-word poll in front of the staging loads (experiment)
  unsigned* abort_word;              // host-visible: raised by a poll that timed out (see OlWatch)
};

__device__ __forceinline__ unsigned long long ol_load(const unsigned long long* p) {
  return __hip_atomic_load((gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void ol_publish(unsigned long long* p, double v) {
  unsigned long long b = (unsigned long long)__double_as_longlong(v);
  if (v != v) b = OL_QNAN;
  __hip_atomic_store((gu64*)p, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ol_val(unsigned long long b) { return __longlong_as_double((long long)b); }

// A poll that sees no progress for ~2 s of the 100 MHz wall clock gives up WITHOUT killing the HIP context (round 6; it used to
// trap): it raises a word in host-visible pinned memory and its wavefront leaves the kernel; every other poll looks at that word
// each 1024 spins and leaves too.  The launch's results are void; the host finds the word raised at its next synchronisation
// (Ctx::sync), throws, and the context uses the launch chains from then on.
struct OlWatch {
  unsigned long long t0 = 0;
  unsigned spins = 0;
  __device__ __forceinline__ bool tick(unsigned* abort_word) {
    __builtin_amdgcn_s_sleep(1);
    if ((++spins & 1023u) == 0) {
      if (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) return true;
      const unsigned long long t = wall_clock64();
      if (t0 == 0) t0 = t;
      else if (t - t0 > 200000000ULL) {
        __hip_atomic_store(abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return true;
      }
    }
    return false;
  }
};

typedef unsigned int ol_u32x4 __attribute__((ext_vector_type(4)));

// (two workgroups per CU must fit: at most 256 registers per lane, at most 80 KB of LDS)
// A workgroup's four wavefronts take four neighbouring columns of a round.  The vector they all read is fetched ONCE per workgroup --
// 16-byte write-through-coherent loads, two per thread and right-hand side -- into LDS (two buffers, alternating by round: one barrier
// per round), and each wavefront then takes its 16 entries per lane from there.  (First form of this kernel: every wavefront loaded
// its own 16 x 8 bytes per lane and right-hand side from the arena -- 8 MB per hand-over and right-hand side chip-wide, 2.9 us of
// it per right-hand side, slower than the launch chain at n = 5000: profiles/r05_trsv_onelaunch.txt.)
template <int NR>
__global__ __launch_bounds__(256, 2) void trsv_onelaunch_kernel(const OlRound* __restrict__ rounds, OlArgs a) {
  __shared__ __attribute__((aligned(16))) double vl[2][NR][1024];
  __shared__ int okf[2][4];
  unsigned att = 0;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int cls = blockIdx.x & 1;
  const int wg = blockIdx.x >> 1;
  const int w = wg * 4 + wv;
  // clear the other arena set (plain stores: the kernel boundary publishes them)
  {
    const long total = 3 * a.asz;
    const long t0 = (long)blockIdx.x * 256 + tid, nth = (long)gridDim.x * 256;
    for (long i = t0; i < total; i += nth) a.clear[i] = OL_SENT;
  }
  const OlRound* __restrict__ rd = rounds + (cls ? a.nrounds0 : 0);   // (uniform, read-only: scalar loads)
  const int nrounds = cls ? a.nrounds1 : a.nrounds0;
  unsigned long long* ar[3] = {a.arena, a.arena + a.asz, a.arena + 2 * a.asz};
  double* xc[3] = {a.x, a.x + a.ldx, a.x3};
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.arena, 0, (int)(3 * a.asz * 8), 0x00020000);
  int buf = 0;
  for (int r = 0; r < nrounds; ++r) {
    const OlRound R = rd[r];
    if (wg * 4 >= R.ncols) continue;   // (the whole workgroup)
    if (R.msel == 4) {   // copy-in: arena[out_off + jj] = x[x_off + jj], 64 entries per wavefront and pass
      for (int jj = (R.col0 + w) * 64 + lane; jj < R.m; jj += OL_W * 64) {
#pragma unroll
        for (int q = 0; q < NR; ++q) ol_publish(ar[q] + R.out_off + jj, xc[q][R.x_off + jj]);
      }
      continue;
    }
    const bool act = (w < R.ncols);                 // (a ragged last workgroup: idle wavefronts still stage and meet the barrier)
    const int j = R.col0 + min(w, R.ncols - 1);
    const double* Mb = R.msel == 0 ? a.U : R.msel == 1 ? a.UT : R.msel == 2 ? a.Binv : a.BinvT;
    const int m = R.m;
    int i0 = 0, i1 = m;
    if (R.mode == 1) i1 = min(j + 1, m);
    else if (R.mode == 2) i0 = min(j, m);
    const double* col = Mb + R.moff + (long)j * R.ld;
    const int ilast = max(i1 - 1, 0);
    double av[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) av[k] = col[min(i0 + lane + 64 * k, ilast)];
    // the rows the workgroup's columns read, as 16-byte pairs from an even row on
    const int jA = R.col0 + wg * 4, jB = R.col0 + min(wg * 4 + 3, R.ncols - 1);
    int lo = 0, hi = m;
    if (R.mode == 1) hi = min(jB + 1, m);
    else if (R.mode == 2) lo = min(jA, m);
    const int lo2 = lo & ~1;
    // hand-over: first one word everybody agrees on, then everything
    OlWatch watch;
    if (!a.direct_poll) {
      const unsigned long long* key = ar[NR - 1] + R.v_off + (hi - 1);
      while (ol_load(key) == OL_SENT)
        if (watch.tick(a.abort_word)) return;
    }
    unsigned long long bb[NR];
    const int hl2 = (hi - 1) & ~1;   // the even row of the last pair (>= lo2)
    for (;;) {
      // every load first (pairs beyond the range are fetched from its last pair again: no branch between the requests) ...
      ol_u32x4 x[NR][2];
#pragma unroll
      for (int q = 0; q < NR; ++q) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const int rowc = min(lo2 + 2 * (tid + 256 * p), hl2);
          x[q][p] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)((q * a.asz + R.v_off + rowc) * 8), 0, 16);   // (aux 16: sc1)
        }
        bb[q] = (R.base_off >= 0) ? ol_load(ar[q] + R.base_off + j) : 0ULL;
      }
      // ... then the tests and the copies into LDS
      bool ok = true;
#pragma unroll
      for (int q = 0; q < NR; ++q) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const int rowc = min(lo2 + 2 * (tid + 256 * p), hl2);
          const unsigned long long h0 = ((unsigned long long)x[q][p].y << 32) | x[q][p].x, h1 = ((unsigned long long)x[q][p].w << 32) | x[q][p].z;
          ok &= (rowc < lo || h0 != OL_SENT) && (rowc + 1 >= hi || h1 != OL_SENT);
          *(ol_u32x4*)&vl[buf][q][rowc] = x[q][p];
        }
        ok &= (bb[q] != OL_SENT);
      }
      // one barrier: it publishes the staged vector and the four wavefronts' verdicts (two verdict slots, alternating by attempt)
      const int wok = __all(ok);
      if (lane == 0) okf[att & 1][wv] = wok;
      __syncthreads();
      const int all_ok = okf[att & 1][0] & okf[att & 1][1] & okf[att & 1][2] & okf[att & 1][3];
      ++att;
      if (all_ok) break;
      if (watch.tick(a.abort_word)) return;
    }
    const double* v0 = vl[buf][0];
    const double* v1 = vl[buf][NR > 1 ? 1 : 0];
    const double* v2 = vl[buf][NR - 1];
    buf ^= 1;
    // (explicit fused multiply-adds: the launch chain's kernels compile to nothing else, and a product the compiler shares between the
    //  two branches of a group -- it did, once, in the three-column form -- rounds twice)
    const double alpha = (double)R.alpha;
    double res[NR];
    if (NR == 1) {   // the sums of coldot_batched_kernel (dense.hip)
      double vv[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) vv[k] = v0[min(i0 + lane + 64 * k, ilast)];
      double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int i = i0 + lane + 256 * g;
        if (i + 192 < i1) {
          s0 = __builtin_fma(av[4 * g], vv[4 * g], s0);
          s1 = __builtin_fma(av[4 * g + 1], vv[4 * g + 1], s1);
          s2 = __builtin_fma(av[4 * g + 2], vv[4 * g + 2], s2);
          s3 = __builtin_fma(av[4 * g + 3], vv[4 * g + 3], s3);
        } else {
#pragma unroll
          for (int t = 0; t < 3; ++t)
            if (i + 64 * t < i1) s0 = __builtin_fma(av[4 * g + t], vv[4 * g + t], s0);
        }
      }
      double s = (s0 + s1) + (s2 + s3);
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
      const double b = (R.base_off >= 0) ? ol_val(bb[0]) : 0.0;
      res[0] = __builtin_fma(alpha, s, b);
    } else {         // columns 0, 1: coldot2_batched_kernel; column 2: coldot_batched_kernel (= coldot3_batched_kernel)
      double s0 = 0.0, s1 = 0.0, t0 = 0.0, t1 = 0.0;
      {
        double va[16], vb[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          const int i = min(i0 + lane + 64 * k, ilast);
          va[k] = v0[i];
          vb[k] = v1[i];
        }
#pragma unroll
        for (int g = 0; g < 8; ++g) {
          const int i = i0 + lane + 128 * g;
          if (i + 64 < i1) {
            s0 = __builtin_fma(av[2 * g], va[2 * g], s0);
            s1 = __builtin_fma(av[2 * g + 1], va[2 * g + 1], s1);
            t0 = __builtin_fma(av[2 * g], vb[2 * g], t0);
            t1 = __builtin_fma(av[2 * g + 1], vb[2 * g + 1], t1);
          } else if (i < i1) {
            s0 = __builtin_fma(av[2 * g], va[2 * g], s0);
            t0 = __builtin_fma(av[2 * g], vb[2 * g], t0);
          }
        }
      }
      double u0 = 0.0, u1 = 0.0, u2 = 0.0, u3 = 0.0;
      if (NR == 3) {
        double vc[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) vc[k] = v2[min(i0 + lane + 64 * k, ilast)];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int i = i0 + lane + 256 * g;
          if (i + 192 < i1) {
            u0 = __builtin_fma(av[4 * g], vc[4 * g], u0);
            u1 = __builtin_fma(av[4 * g + 1], vc[4 * g + 1], u1);
            u2 = __builtin_fma(av[4 * g + 2], vc[4 * g + 2], u2);
            u3 = __builtin_fma(av[4 * g + 3], vc[4 * g + 3], u3);
          } else {
#pragma unroll
            for (int t = 0; t < 3; ++t)
              if (i + 64 * t < i1) u0 = __builtin_fma(av[4 * g + t], vc[4 * g + t], u0);
          }
        }
      }
      double s = s0 + s1, t = t0 + t1, u = (u0 + u1) + (u2 + u3);
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        s += __shfl_down(s, off);
        t += __shfl_down(t, off);
        if (NR == 3) u += __shfl_down(u, off);
      }
      const bool hb = (R.base_off >= 0);
      res[0] = __builtin_fma(alpha, s, hb ? ol_val(bb[0]) : 0.0);
      res[1] = __builtin_fma(alpha, t, hb ? ol_val(bb[1]) : 0.0);
      if (NR == 3) res[NR - 1] = __builtin_fma(alpha, u, hb ? ol_val(bb[NR - 1]) : 0.0);
    }
    if (lane == 0 && act) {
#pragma unroll
      for (int q = 0; q < NR; ++q) {
        ol_publish(ar[q] + R.out_off + j, res[q]);
        if (R.x_off >= 0) xc[q][R.x_off + j] = res[q];
      }
    }
  }
}

}  // namespace

bool trsv_one_launch_on() {
  static const bool on = [] { const char* e = getenv("HYP_TRSV_ONE_LAUNCH"); return !(e && atoi(e) == 0); }();
  return on;
}

// The rounds of one sweep appended to the two classes' lists.  Arena layout of a sweep, offsets from `base`:
//   XV[s] (s = 0 .. nsb-1, n words each): the sweep's vector after s rank-sb updates (XV[0]: the input);  then per super-block
//   T1, E1, T2, E2, ... (sb words each, 2 * refine of them);  then XF (n words): the finished entries.
// `in_off` >= 0: the input already sits in the arena there (the forward sweep's XF when both sweeps share a launch) and replaces XV[0].
// (every vector starts at an even word: the kernel fetches it in 16-byte pairs)
static long ol_npad(int n) { return (long)((n + 1) & ~1); }
static long ol_sweep_words(int n, int sb, int refine) {
  const int nsb = (n + sb - 1) / sb;
  return (long)(nsb + 1) * ol_npad(n) + (long)2 * refine * nsb * sb;
}
static void ol_append_sweep(std::vector<OlRound>& chain, std::vector<OlRound>& upd, int n, int sb, int refine, long ldu, bool trans, long base,
                            long in_off, bool copy_out) {
  const int nsb = (n + sb - 1) / sb;
  const long np = ol_npad(n);
  auto XV = [&](int s) { return (s == 0 && in_off >= 0) ? in_off : base + (long)s * np; };
  const long tmp0 = base + (long)nsb * np;
  const long XF = tmp0 + (long)2 * refine * nsb * sb;
  if (in_off < 0) {   // copy-in: the caller's vector becomes XV[0] (so that the in-place result never meets a late reader of the input)
    OlRound c{};
    c.msel = 4; c.m = n; c.ncols = std::min(OL_W, (n + 63) / 64); c.col0 = 0; c.out_off = (int)XV(0); c.x_off = 0; c.base_off = -1;
    chain.push_back(c);
  }
  for (int s = 0; s < nsb; ++s) {
    const int b = trans ? s : nsb - 1 - s;
    const int r0 = b * sb, m = std::min(sb, n - r0);
    const long xb = XV(s) + r0;                                                    // this super-block's rows of the current version
    const long Dm = trans ? (long)r0 * ldu + r0 : (long)r0 * n + r0;               // the factor's diagonal block (in U / in UT)
    const long Bm = (long)b * sb * sb;                                             // its inverse (in Binv / BinvT)
    const int mode = trans ? 1 : 2;
    auto tmp = [&](int k) { return tmp0 + ((long)s * 2 * refine + k) * sb; };
    auto diag = [&](bool inv, long v, long bs, long out, float alpha, bool fin) {
      OlRound q{};
      q.msel = inv ? (trans ? 2 : 3) : (trans ? 0 : 1);
      q.moff = inv ? Bm : Dm;
      q.ld = inv ? sb : (trans ? ldu : n);
      q.m = m; q.mode = mode; q.ncols = m; q.col0 = 0;
      q.v_off = (int)v; q.base_off = (int)bs; q.out_off = (int)out; q.alpha = alpha;
      q.x_off = (fin && copy_out) ? r0 : -1;
      chain.push_back(q);
    };
    const long fin = XF + r0;
    if (refine == 0) {
      diag(true, xb, -1, fin, 1.0f, true);                                         // x_b = B x_b
    } else {
      diag(true, xb, -1, tmp(0), 1.0f, false);                                     // t = B x_b
      for (int it = 0; it < refine; ++it) {
        const long t = tmp(2 * it), e = tmp(2 * it + 1);
        diag(false, t, xb, e, -1.0f, false);                                       // e = x_b - T t
        const bool last = (it + 1 == refine);
        diag(true, e, t, last ? fin : tmp(2 * it + 2), 1.0f, last);                // t += B e
      }
    }
    // the rank-m update of the rest: version s + 1.  Forward: columns r0 + m .. n of U, rows r0 .. r0 + m; backward: columns 0 .. r0 of UT
    const int rest = trans ? n - (r0 + m) : r0;
    if (rest <= 0 || s + 1 >= nsb) continue;
    const int c0 = trans ? r0 + m : 0;                                             // first entry of the rest
    // (the next super-block's rows first: they are what its chain waits for)
    const int nb = trans ? s + 1 : nsb - 2 - s;
    const int nr0 = nb * sb, nm = std::min(sb, n - nr0);
    auto chunk = [&](int first, int count) {                                       // entries [first, first + count) of the vector
      for (int o = 0; o < count; o += OL_W) {
        OlRound q{};
        q.msel = trans ? 0 : 1;
        q.moff = trans ? (long)(first + o) * ldu + r0 : (long)(first + o) * n + r0;
        q.ld = trans ? ldu : n;
        q.m = m; q.mode = 0; q.ncols = std::min(OL_W, count - o); q.col0 = 0;
        q.v_off = (int)fin; q.base_off = (int)(XV(s) + first + o); q.out_off = (int)(XV(s + 1) + first + o); q.alpha = -1.0f; q.x_off = -1;
        upd.push_back(q);
      }
    };
    chunk(nr0, nm);
    if (trans) chunk(nr0 + nm, n - (nr0 + nm));
    else chunk(c0, nr0 - c0);
  }
}

bool TriSolvePlan::ol_fits(Ctx& c, int nr) {
  static int fits[16][4];   // per device and instance: 0 not asked, 1 fits, -1 does not
  const int dev = (c.device >= 0 && c.device < 16) ? c.device : 0;
  if (nr < 1 || nr > 3) return false;
  if (fits[dev][nr] == 0) {
    int cus = 0, per_cu = 0;
    hipError_t e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c.device);
    if (e == hipSuccess) {
      if (nr == 1) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, trsv_onelaunch_kernel<1>, 256, 0);
      else if (nr == 2) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, trsv_onelaunch_kernel<2>, 256, 0);
      else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, trsv_onelaunch_kernel<3>, 256, 0);
    }
    if (e != hipSuccess) (void)hipGetLastError();
    fits[dev][nr] = (e == hipSuccess && (long)per_cu * cus >= 2 * OL_W / 4) ? 1 : -1;
  }
  return fits[dev][nr] > 0;
}

void TriSolvePlan::ol_prepare(Ctx& c, long ldu) {
  ol_ok = false;
  if (!trsv_one_launch_on() || n <= 0 || sb <= 0 || sb > 1024 || refine < 0 || refine > 3) return;
  // (asked once: hipDeviceGetAttribute is a driver round trip -- 0.19 ms of idle GPU behind every plan build when it sat here per call,
  //  profiles/r05_iteration_timeline.txt against the final one)
  // All 2 OL_W / 4 = 512 four-wavefront workgroups of a launch must be RESIDENT at once (a wavefront polls words that other workgroups
  // publish).  Asked per instance of the kernel and per device, once (driver round trips: 0.19 ms of idle GPU behind every plan build when
  // the attribute query sat here per call, profiles/r05_iteration_timeline.txt): the occupancy the runtime computes from the instance's
  // registers, its static LDS (NR = 3: 48 KB) and the launch bounds, times the CU count of the CONTEXT's device.  An instance that does
  // not fit (a part with less LDS per CU, a register-allocation regression) takes the launch chains -- ol_fits() is asked per sweep.
  for (int r = 1; r <= 3; ++r) ol_fit[r] = ol_fits(c, r);
  if (!ol_fit[1] && !ol_fit[2] && !ol_fit[3]) return;
  // (the arena is laid out for the most refinement steps this plan can run: the per-right-hand-side stride and what a launch clears do
  //  not change when the adaptive rule moves between tables)
  const int rmax = std::max(std::max(refine, refine_req), 0);
  const long words_max = ol_sweep_words(n, sb, std::min(rmax, 3));
  if (2 * words_max > 0x3fffffffL) return;  // (arena offsets are ints)
  if (!(ol_n == n && ol_sb == sb && ol_ldu == ldu && ol_asz == 2 * words_max)) {
    for (bool& h : ol_have) h = false;
    const size_t bytes = ((size_t)2 * 3 * 2 * words_max + 16) * sizeof(unsigned long long);   // two sets x three right-hand sides x two sweeps (+ a pair read past the last vector)
    ol_arena.ensure(bytes);
    HYP_CHECK(hipMemsetAsync(ol_arena.p, 0xFF, ol_arena.bytes, c.stream));
    ol_set = 0;
    ol_asz = 2 * words_max;
    ol_n = n; ol_sb = sb; ol_ldu = ldu;
  }
  if (!ol_have[refine]) {
    // tables: [0] forward sweep alone, [1] backward sweep alone, [2] both sweeps in one launch (the backward sweep reads the forward one's XF)
    const long words = ol_sweep_words(n, sb, refine);
    std::vector<OlRound> all;
    for (int t = 0; t < 3; ++t) {
      std::vector<OlRound> chain, upd;
      if (t == 0) ol_append_sweep(chain, upd, n, sb, refine, ldu, true, 0, -1, true);
      else if (t == 1) ol_append_sweep(chain, upd, n, sb, refine, ldu, false, 0, -1, true);
      else {
        ol_append_sweep(chain, upd, n, sb, refine, ldu, true, 0, -1, false);
        const int nsb = (n + sb - 1) / sb;
        const long XF = (long)nsb * ol_npad(n) + (long)2 * refine * nsb * sb;
        ol_append_sweep(chain, upd, n, sb, refine, ldu, false, words, XF, true);
      }
      ol_first[refine][t] = (int)all.size();
      ol_n0[refine][t] = (int)chain.size();
      ol_n1[refine][t] = (int)upd.size();
      all.insert(all.end(), chain.begin(), chain.end());
      all.insert(all.end(), upd.begin(), upd.end());
    }
    ol_rounds[refine].ensure(all.size() * sizeof(OlRound));
    HYP_CHECK(hipMemcpyAsync(ol_rounds[refine].p, all.data(), all.size() * sizeof(OlRound), hipMemcpyHostToDevice, c.stream));
    HYP_CHECK(hipStreamSynchronize(c.stream));   // (`all` is pageable and dies here; once per (n, sb, refine, ldu))
    ol_have[refine] = true;
  }
  ol_ok = true;
}

// HYP_TRSV_OL_STATS=1: at exit, how many sweeps went through the one-launch kernel (by kind and right-hand sides)
static long ol_counts[3][4];
static void ol_report() {
  for (int w = 0; w < 3; ++w)
    for (int r = 1; r <= 3; ++r)
      if (ol_counts[w][r]) fprintf(stderr, "[trsv one-launch] %s, %d right-hand side(s): %ld launches\n", w == 0 ? "forward" : w == 1 ? "backward" : "both", r, ol_counts[w][r]);
}
void TriSolvePlan::ol_sweep(Ctx& c, const double* U, int which, double* x, long ldx, double* x3, int nr) {
  static const bool stats = [] { const char* e = getenv("HYP_TRSV_OL_STATS"); const bool on = e && e[0] == '1'; if (on) atexit(ol_report); return on; }();
  if (stats) ++ol_counts[which][nr];
  OlArgs a{};
  a.U = U; a.UT = UT.d(); a.Binv = Binv.d(); a.BinvT = BinvT.d();
  const OlRound* rounds = (const OlRound*)ol_rounds[refine].p + ol_first[refine][which];
  a.nrounds0 = ol_n0[refine][which]; a.nrounds1 = ol_n1[refine][which];
  unsigned long long* base = (unsigned long long*)ol_arena.p;
  a.arena = base + (long)ol_set * 3 * ol_asz;
  a.clear = base + (long)(1 - ol_set) * 3 * ol_asz;
  a.asz = ol_asz;
  a.x = x; a.ldx = ldx; a.x3 = x3 ? x3 : x + 2 * ldx;
  static const int direct = [] { const char* e = getenv("HYP_TRSV_OL_POLL"); return e ? atoi(e) : 0; }();
  a.direct_poll = direct;
  a.abort_word = c.ol_abort_dev;
  ol_set = 1 - ol_set;
  const dim3 grid(2 * OL_W / 4), blk(256);
  if (nr == 1) hipLaunchKernelGGL(trsv_onelaunch_kernel<1>, grid, blk, 0, c.stream, rounds, a);
  else if (nr == 2) hipLaunchKernelGGL(trsv_onelaunch_kernel<2>, grid, blk, 0, c.stream, rounds, a);
  else hipLaunchKernelGGL(trsv_onelaunch_kernel<3>, grid, blk, 0, c.stream, rounds, a);
  HYP_CHECK(hipGetLastError());
}

}  // namespace hyp
