// Kernel body of the FP64 MFMA GEMM family (see gemm_f64.hpp for the contract).  Included by
// gemm_f64.hip (the library instantiation) and by tools/probe_gemm.hip.
#pragma once
#include "gemm_f64.hpp"
#include <vector>

namespace hyp {

// tools/probe_gemm64.hip builds this file with -DHYP_GEMM_PROBE: every wavefront adds the shader cycles (s_memtime) it spent in the
// four parts of its K steps -- requesting the next tile, LDS fragment reads + MFMA issue, waiting for the requested tile + LDS stores,
// the barrier -- to gemm_probe_acc[0..3], its K steps to [4], itself to [5], its whole life to [6].  Not compiled into the library.
#ifdef HYP_GEMM_PROBE
__device__ unsigned long long gemm_probe_acc[8];
#define GEMM_PROBE_T(x) const unsigned long long x = __builtin_amdgcn_s_memtime()
#else
#define GEMM_PROBE_T(x)
#endif

constexpr int BK = 16, LDS_S = BK + 2;
// Fragment reads are VOLATILE loads from the LDS address space: the compiler otherwise pairs the reads of two k steps into
// ds_read2_b64, which the LDS serves at half the rate of ds_read_b64 and banks modulo 32 dwords -- rows fr and fr + 8 of this layout
// (36 dwords apart) then collide (the 0.40 bank-conflict share of profiles/r05_pmc_cfg5p_cfg3b.json); ds_read_b64 banks modulo 64,
// where the 16 rows x 2 k of a 32-lane group cover the 64 banks exactly once (MI355X_MICROARCH.md, LDS table)
typedef const volatile double __attribute__((address_space(3))) lds_cv_f64;
constexpr int GEMM_THREADS = 256;

// map a linear index over the upper triangle (tn >= tm) of a T x T tile grid to (tm, tn),
// column by column: idx = tn*(tn+1)/2 + tm.
__device__ __forceinline__ void upper_tile_from_linear(int idx, int& tm, int& tn) {
  int t = (int)((sqrt(8.0 * (double)idx + 1.0) - 1.0) * 0.5);
  while ((long)(t + 1) * (t + 2) / 2 <= idx) ++t;
  while ((long)t * (t + 1) / 2 > idx) --t;
  tn = t;
  tm = idx - t * (t + 1) / 2;
}

// TW = MFMA tiles per wavefront per dimension: 4 -> 128 x 128 block tile (big products),
//      2 -> 64 x 64 block tile (small matrices: 4x the workgroups, 1/4 the serial MFMA chain each)
template <bool TRANSA, int TW, int TAG>
__global__ __launch_bounds__(GEMM_THREADS, 2)
void gemm_f64_kernel(GemmArgs p) {
  constexpr int BM = 32 * TW, BN = 32 * TW;
  constexpr int REPS = 2 * TW;              // staged elements per thread per operand per K tile
  constexpr int WT = 16 * TW;               // wavefront sub-tile edge
  __shared__ double lds[2][2][BM * LDS_S];   // [buffer][A/B][row*LDS_S + k]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  if (p.hiprio) __builtin_amdgcn_s_setprio(3);

  int tm, tn;
  const int bz = blockIdx.y;   // batch index
  if (p.tile_map) {
    tm = p.tile_map[2 * blockIdx.x];
    tn = p.tile_map[2 * blockIdx.x + 1];
  } else if (p.tri == GEMM_UPPER) {
    upper_tile_from_linear(blockIdx.x, tm, tn);
  } else {
    tm = blockIdx.x % p.tiles_m;
    tn = blockIdx.x / p.tiles_m;
  }
  const int m0 = tm * BM, n0 = tn * BN;
  if (p.tri == GEMM_UPPER_RECT && m0 > n0 + BN - 1 + p.tri_off) return;

  const double* __restrict__ A = p.A + (long)bz * p.strideA;
  const double* __restrict__ B = p.B + (long)bz * p.strideB;
  double* __restrict__ C = p.C + (long)bz * p.strideC;

  // K range (split-K: this slice's chunk)
  int kbeg = 0, kend = p.K;
  if (p.splitk > 1) {
    int z = blockIdx.z, sub = -1;
    if (p.tail_q > 1) {
      if (z >= p.splitk_base) {
        if ((int)blockIdx.x < p.tail_first) return;
        sub = z - p.splitk_base + 1;
        z = p.splitk_base - 1;
      } else if (z == p.splitk_base - 1 && (int)blockIdx.x >= p.tail_first) {
        sub = 0;
      }
    }
    kbeg = z * p.kchunk;
    kend = min(p.K, kbeg + p.kchunk);
    if (sub >= 0) {
      kbeg += sub * p.tail_chunk;
      kend = min(kend, kbeg + p.tail_chunk);
    }
  }
  switch (p.krange) {
    case KR_LE_M: kend = min(p.K, m0 + BM); break;
    case KR_GE_M: kbeg = min(p.K, m0) & ~(BK - 1); break;
    case KR_LE_N: kend = min(p.K, n0 + BN); break;
    case KR_GE_N: kbeg = min(p.K, n0) & ~(BK - 1); break;
    default: break;
  }

  d4_t acc[TW][TW];
#pragma unroll
  for (int i = 0; i < TW; ++i)
#pragma unroll
    for (int j = 0; j < TW; ++j) acc[i][j] = (d4_t){0.0, 0.0, 0.0, 0.0};

  // staging registers: REPS doubles of the A tile, REPS of the B tile per thread
  double ra[REPS], rb[REPS];

  // TN loader: k = tid & 15, row = (tid >> 4) + 16 * rep  (16 lanes read 128 contiguous bytes)
  // NN loader (A only): row = tid % BM, k = tid / BM + (256 / BM) * rep (BM lanes read contiguous rows)
  const int lk = tid & 15, lr = tid >> 4;
  constexpr int NN_KSTEP = GEMM_THREADS / BM;
  const int nn_r = tid % BM, nn_k = tid / BM;

  // ---- fast path (TN, tile fully inside M x N, 16-byte aligned operands): unpredicated double2 loads,
  //      k pair = 2 * (tid & 7), row = (tid >> 3) + 32 * rep; 8 lanes read 128 contiguous bytes
  typedef double d2_t __attribute__((ext_vector_type(2)));
  constexpr int REPS2 = REPS / 2;
  const bool fast = TRANSA && p.vec2 && (m0 + BM <= p.M) && (n0 + BN <= p.N);
  const int vk = (tid & 7) * 2, vr = tid >> 3;
  const double* __restrict__ fa = A + (long)(m0 + vr) * p.lda + vk;
  const double* __restrict__ fb = B + (long)(n0 + vr) * p.ldb + vk;

  auto load_tiles_fast = [&](int k0) {   // (shares the staging registers ra / rb with the general loader)
#pragma unroll
    for (int rep = 0; rep < REPS2; ++rep) {
      const d2_t ta = *reinterpret_cast<const d2_t*>(fa + (long)(32 * rep) * p.lda + k0);
      const d2_t tb = *reinterpret_cast<const d2_t*>(fb + (long)(32 * rep) * p.ldb + k0);
      ra[2 * rep] = ta.x; ra[2 * rep + 1] = ta.y;
      rb[2 * rep] = tb.x; rb[2 * rep + 1] = tb.y;
    }
  };
  auto store_tiles_fast = [&](int buf) {
    double* As = lds[buf][0];
    double* Bs = lds[buf][1];
#pragma unroll
    for (int rep = 0; rep < REPS2; ++rep) {
      *reinterpret_cast<d2_t*>(As + (vr + 32 * rep) * LDS_S + vk) = (d2_t){ra[2 * rep], ra[2 * rep + 1]};
      *reinterpret_cast<d2_t*>(Bs + (vr + 32 * rep) * LDS_S + vk) = (d2_t){rb[2 * rep], rb[2 * rep + 1]};
    }
  };

  auto load_tiles = [&](int k0) {
    if (fast && k0 + BK <= kend) {
      load_tiles_fast(k0);
      return;
    }

    // General (edge / unaligned) loader.  Loads are unconditional on clamped indices; out-of-range elements are
    // zeroed by a MULTIPLY with a 0 / 1 mask in store_tiles, AFTER the MFMAs of the current tile.  (With a select
    // the compiler sinks each load under its predicate and waits for every single one; with the multiply right
    // here it waits for the loads before the MFMA block: either way the prefetch is lost.)
    const int klast = kend - 1;
    if (TRANSA) {
      const int kc = min(k0 + lk, klast);
#pragma unroll
      for (int rep = 0; rep < REPS; ++rep) ra[rep] = A[(long)min(m0 + lr + 16 * rep, p.M - 1) * p.lda + kc];
    } else {
      const int mc = min(m0 + nn_r, p.M - 1);
#pragma unroll
      for (int rep = 0; rep < REPS; ++rep) ra[rep] = A[(long)min(k0 + nn_k + NN_KSTEP * rep, klast) * p.lda + mc];
    }
    {
      const int kc = min(k0 + lk, klast);
#pragma unroll
      for (int rep = 0; rep < REPS; ++rep) rb[rep] = B[(long)min(n0 + lr + 16 * rep, p.N - 1) * p.ldb + kc];
    }
  };

  auto store_tiles = [&](int buf, int k0) {
    if (fast && k0 + BK <= kend) {
      store_tiles_fast(buf);
      return;
    }
    double* As = lds[buf][0];
    double* Bs = lds[buf][1];
    const double kmask = (k0 + lk < kend) ? 1.0 : 0.0;
    if (TRANSA) {
#pragma unroll
      for (int rep = 0; rep < REPS; ++rep) As[(lr + 16 * rep) * LDS_S + lk] = ra[rep] * ((m0 + lr + 16 * rep < p.M) ? kmask : 0.0);
    } else {
      const double mmask = (m0 + nn_r < p.M) ? 1.0 : 0.0;
#pragma unroll
      for (int rep = 0; rep < REPS; ++rep)
        As[nn_r * LDS_S + nn_k + NN_KSTEP * rep] = ra[rep] * ((k0 + nn_k + NN_KSTEP * rep < kend) ? mmask : 0.0);
    }
#pragma unroll
    for (int rep = 0; rep < REPS; ++rep) Bs[(lr + 16 * rep) * LDS_S + lk] = rb[rep] * ((n0 + lr + 16 * rep < p.N) ? kmask : 0.0);
  };

  const int fr = lane & 15, fk = lane >> 4;

  auto compute = [&](int buf) {
    const double* As = lds[buf][0] + (wm * WT + fr) * LDS_S + fk;
    const double* Bs = lds[buf][1] + (wn * WT + fr) * LDS_S + fk;
    // (a register-double-buffered version of these fragment reads was tried: at 2 workgroups per CU it
    //  spills past 256 VGPRs and loses 15 %; the second resident workgroup already covers the LDS latency)
#pragma unroll
    for (int kk = 0; kk < BK; kk += 4) {
      double af[TW], bf[TW];
#pragma unroll
#ifdef HYP_GEMM_READ2   // (the round-1..5 form, for A/B builds: plain loads, paired by the compiler into ds_read2_b64)
      for (int i = 0; i < TW; ++i) af[i] = As[i * 16 * LDS_S + kk];
#pragma unroll
      for (int j = 0; j < TW; ++j) bf[j] = Bs[j * 16 * LDS_S + kk];
#else
      for (int i = 0; i < TW; ++i) af[i] = *(lds_cv_f64*)(As + i * 16 * LDS_S + kk);
#pragma unroll
      for (int j = 0; j < TW; ++j) bf[j] = *(lds_cv_f64*)(Bs + j * 16 * LDS_S + kk);
#endif
#pragma unroll
      for (int j = 0; j < TW; ++j)
#pragma unroll
        for (int i = 0; i < TW; ++i)
          acc[j][i] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[j], af[i], acc[j][i], 0, 0, 0);
    }
  };

  const int nkt = (kend > kbeg) ? (kend - kbeg + BK - 1) / BK : 0;
#ifdef HYP_GEMM_PROBE
  unsigned long long pr_req = 0, pr_mma = 0, pr_sto = 0, pr_bar = 0;
  GEMM_PROBE_T(pr_birth);
#endif
  if (nkt > 0) {
    load_tiles(kbeg);
    store_tiles(0, kbeg);
    __syncthreads();
    for (int t = 0; t < nkt; ++t) {
      const int buf = t & 1;
      GEMM_PROBE_T(c0);
      if (t + 1 < nkt) load_tiles(kbeg + (t + 1) * BK);
      GEMM_PROBE_T(c1);
      compute(buf);
      GEMM_PROBE_T(c2);
      if (t + 1 < nkt) store_tiles(buf ^ 1, kbeg + (t + 1) * BK);
      GEMM_PROBE_T(c3);
      __syncthreads();
#ifdef HYP_GEMM_PROBE
      const unsigned long long c4 = __builtin_amdgcn_s_memtime();
      pr_req += c1 - c0; pr_mma += c2 - c1; pr_sto += c3 - c2; pr_bar += c4 - c3;
#endif
    }
  }
#ifdef HYP_GEMM_PROBE
  if (lane == 0) {
    atomicAdd(&gemm_probe_acc[0], pr_req); atomicAdd(&gemm_probe_acc[1], pr_mma); atomicAdd(&gemm_probe_acc[2], pr_sto);
    atomicAdd(&gemm_probe_acc[3], pr_bar); atomicAdd(&gemm_probe_acc[4], (unsigned long long)nkt); atomicAdd(&gemm_probe_acc[5], 1ull);
    atomicAdd(&gemm_probe_acc[6], __builtin_amdgcn_s_memtime() - pr_birth);
  }
#endif

  // epilogue: lane (fr, fk), accumulator register r of tile (j, i) is
  //   C[m0 + wm*WT + i*16 + fr, n0 + wn*WT + j*16 + fk + 4r]
  const bool upper = (p.tri != GEMM_FULL);
  if (p.splitk > 1) {   // raw partial sums; alpha / beta / epilogue are applied by splitk_reduce_kernel
    double* __restrict__ W = p.part + ((long)bz * p.splitk + blockIdx.z) * p.part_stride;   // [batch member][slice] (one matrix: bz = 0)
    if (TAG == 1 && p.tile_cnt != nullptr) {
      // Round 6: the reduction rides in this launch.  Partial sums leave as device-coherent (write-through) stores -- the slices of a
      // tile run on different XCDs, whose L2s do not see each other's dirty lines --; a workgroup waits for its stores, counts itself
      // in, and the LAST slice of a tile to arrive adds the partial sums in slice order -- the very sums of splitk_reduce_kernel -- from
      // device-coherent loads and writes C.  What that kernel did in 0.2 ms behind the product now hides under other tiles' MFMAs.
#pragma unroll
      for (int j = 0; j < TW; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = n0 + wn * WT + j * 16 + fk + 4 * r;
          if (n >= p.N) continue;
#pragma unroll
          for (int i = 0; i < TW; ++i) {
            const int m = m0 + wm * WT + i * 16 + fr;
            if (m >= p.M || (upper && m > n + p.tri_off)) continue;
            __hip_atomic_store(&W[(long)n * p.part_ld + m], acc[j][i][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __shared__ int last_in;
      __syncthreads();
      if (tid == 0) {
        const int expect = (p.tail_q > 1 && (int)blockIdx.x >= p.tail_first) ? p.splitk_base - 1 + p.tail_q : p.splitk_base;
        const int old = __hip_atomic_fetch_add(&p.tile_cnt[blockIdx.x], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_in = (old == expect - 1) ? 1 : 0;
        if (last_in) __hip_atomic_store(&p.tile_cnt[blockIdx.x], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (ready for the next launch)
      }
      __syncthreads();
      if (!last_in) return;
      const int S = p.splitk_base;
      const int S_extra = (p.tail_q > 1 && (int)blockIdx.x >= p.tail_first) ? p.tail_q - 1 : 0;
      const double* __restrict__ P0 = p.part + (long)bz * p.splitk * p.part_stride;
#pragma unroll
      for (int j = 0; j < TW; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = n0 + wn * WT + j * 16 + fk + 4 * r;
          if (n >= p.N) continue;
#pragma unroll
          for (int i = 0; i < TW; ++i) {
            const int m = m0 + wm * WT + i * 16 + fr;
            if (m >= p.M || (upper && m > n + p.tri_off)) continue;
            const long off = (long)n * p.part_ld + m;
            double s = 0.0;
            for (int z = 0; z < S + S_extra; ++z) s += __hip_atomic_load(&P0[(long)z * p.part_stride + off], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            double* cp = C + (long)n * p.ldc + m;
            *cp = p.alpha * s + (p.beta != 0.0 ? p.beta * (*cp) : 0.0);
          }
        }
      return;
    }
#pragma unroll
    for (int j = 0; j < TW; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wn * WT + j * 16 + fk + 4 * r;
        if (n >= p.N) continue;
#pragma unroll
        for (int i = 0; i < TW; ++i) {
          const int m = m0 + wm * WT + i * 16 + fr;
          if (m >= p.M || (upper && m > n + p.tri_off)) continue;
          W[(long)n * p.part_ld + m] = acc[j][i][r];
        }
      }
    return;
  }
  if (p.beta != 0.0 && p.cm_blk == 0) {
    // read-modify-write of C (rank-k updates of the factorizations).  Written naively, every element is a
    // load that must wait for the previous element's store (C may alias itself): 4 TW^2 serial round
    // trips per thread.  Instead each group of 4 TW old values is fetched at once, on clamped addresses so
    // the loads are unconditional, and pinned before the first store of the group.
#pragma unroll
    for (int j = 0; j < TW; ++j) {
      double cold[4][TW];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int nc = min(n0 + wn * WT + j * 16 + fk + 4 * r, p.N - 1);
#pragma unroll
        for (int i = 0; i < TW; ++i) cold[r][i] = C[(long)nc * p.ldc + min(m0 + wm * WT + i * 16 + fr, p.M - 1)];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < TW; ++i) asm volatile("" : "+v"(cold[r][i]));
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wn * WT + j * 16 + fk + 4 * r;
#pragma unroll
        for (int i = 0; i < TW; ++i) {
          const int m = m0 + wm * WT + i * 16 + fr;
          double v = p.alpha * acc[j][i][r];
          if (p.epi == 1) v = v * v;
          v += p.beta * cold[r][i];
          if (n < p.N && m < p.M && !(upper && m > n + p.tri_off)) C[(long)n * p.ldc + m] = v;
        }
      }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < TW; ++j) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + wn * WT + j * 16 + fk + 4 * r;
      if (n >= p.N) continue;
#pragma unroll
      for (int i = 0; i < TW; ++i) {
        const int m = m0 + wm * WT + i * 16 + fr;
        if (m >= p.M) continue;
        if (upper && m > n + p.tri_off) continue;
        const long moff = p.cm_blk ? (long)(m / p.cm_blk) * p.cm_stride + (m % p.cm_blk) : (long)m;
        double* cp = C + (long)n * p.ldc + moff;
        double v = p.alpha * acc[j][i][r];
        if (p.epi == 1) v = v * v;
        if (p.beta != 0.0) v += p.beta * (*cp);
        *cp = v;
      }
    }
  }
}

// C = alpha * (sum of the split-K slices, in slice order) + beta * C.  Tiles of the launch's tail (launch position >= tail_first,
// tile_rank = launch position per 128 x 128 tile) have S_extra more partial slots, the sub-slices of their last slice.
__global__ void splitk_reduce_kernel(int M, int N, int upper, int tri_off, int S, const double* __restrict__ part, long part_ld, long part_stride,
                                     double alpha, double beta, double* __restrict__ C, long ldc, int S_extra, int tail_first,
                                     const int* __restrict__ tile_rank, int T, long part_batch, long strideC) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = blockIdx.y;
  if (m >= M || (upper && m > n + tri_off)) return;
  part += (long)blockIdx.z * part_batch;   // (batched split-K, requested explicitly: GemmArgs::splitk_req with batch > 1)
  C += (long)blockIdx.z * strideC;
  double s = 0.0;
  for (int z = 0; z < S; ++z) s += part[(long)z * part_stride + (long)n * part_ld + m];
  if (S_extra > 0 && tile_rank[(m >> 7) + (n >> 7) * T] >= tail_first)
    for (int z = S; z < S + S_extra; ++z) s += part[(long)z * part_stride + (long)n * part_ld + m];
  double* cp = C + (long)n * ldc + m;
  *cp = alpha * s + (beta != 0.0 ? beta * (*cp) : 0.0);
}

// XCD-aware order of the upper tiles of a T x T grid.  Hardware workgroup b runs on XCD b % 8 (observed,
// used for speed only): give every XCD a contiguous run of a super-tile-major enumeration (8 x 8 tiles
// per super-tile), so that the ~64 tiles resident on one XCD at a time share 16 operand panels in its
// private L2 instead of touching up to 128 different ones.
static const int* upper_tile_map(GemmScratch& gs, int T, long nblk) {
  if (gs.tile_map_T == T) return gs.tile_map;
  std::vector<int> logical;
  logical.reserve(2 * nblk);
  const int ST = 8, nst = (T + ST - 1) / ST;
  for (int J = 0; J < nst; ++J)
    for (int I = 0; I <= J; ++I)
      for (int j = 0; j < ST; ++j)
        for (int i = 0; i < ST; ++i) {
          const int tm = I * ST + i, tn = J * ST + j;
          if (tm < T && tn < T && tm <= tn) { logical.push_back(tm); logical.push_back(tn); }
        }
  std::vector<int> hw(2 * nblk + (size_t)T * T, -1);
  const long q = nblk / 8, r = nblk % 8;   // XCD x owns logical [start_x, start_x + q + (x < r))
  for (long b = 0; b < nblk; ++b) {
    const long x = b % 8, k = b / 8;
    const long start = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    const long L = start + k;
    hw[2 * b] = logical[2 * L];
    hw[2 * b + 1] = logical[2 * L + 1];
    hw[2 * nblk + hw[2 * b] + (size_t)hw[2 * b + 1] * T] = (int)b;   // inverse: launch position of tile (tm, tn)
  }
  if (gs.tile_map) (void)hipFree(gs.tile_map);
  gs.tile_map = nullptr; gs.tile_map_T = -1;
  if (hipMalloc((void**)&gs.tile_map, hw.size() * sizeof(int)) != hipSuccess) { gs.tile_map = nullptr; return nullptr; }
  if (hipMemcpy(gs.tile_map, hw.data(), hw.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
  gs.tile_map_T = T;
  return gs.tile_map;
}

// The same order for the tiles of an upper trapezoid that do work (GEMM_UPPER_RECT: tile (tm, tn) of a tiles_m x tiles_n grid unless
// 128 tm > 128 tn + 127 + tri_off): launch positions exist only for those, in super-tile-major order dealt out to the XCDs in
// contiguous runs.  *nblk_out = their number (the launch's grid).  A table per shape, sixteen shapes kept (least recently used goes).
static const int* trap_tile_map(GemmScratch& gs, int tiles_m, int tiles_n, int tri_off, long* nblk_out) {
  for (GemmScratch::TrapMap& t : gs.trap)
    if (t.map && t.tiles_m == tiles_m && t.tiles_n == tiles_n && t.tri_off == tri_off) { t.used = ++gs.trap_clock; *nblk_out = t.nblk; return t.map; }
  GemmScratch::TrapMap* slot = &gs.trap[0];   // an empty slot, else the least recently used one
  for (GemmScratch::TrapMap& t : gs.trap) {
    if (!t.map) { slot = &t; break; }
    if (t.used < slot->used) slot = &t;
  }
  std::vector<int> logical;
  const int ST = 8, nsm = (tiles_m + ST - 1) / ST, nsn = (tiles_n + ST - 1) / ST;
  for (int J = 0; J < nsn; ++J)
    for (int I = 0; I < nsm; ++I)
      for (int j = 0; j < ST; ++j)
        for (int i = 0; i < ST; ++i) {
          const int tm = I * ST + i, tn = J * ST + j;
          if (tm < tiles_m && tn < tiles_n && !(128L * tm > 128L * tn + 127 + tri_off)) { logical.push_back(tm); logical.push_back(tn); }
        }
  const long nblk = (long)logical.size() / 2;
  std::vector<int> hw(2 * nblk + (size_t)tiles_m * tiles_n, -1);
  const long q = nblk / 8, r = nblk % 8;
  for (long b = 0; b < nblk; ++b) {
    const long x = b % 8, k = b / 8;
    const long start = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    const long L = start + k;
    hw[2 * b] = logical[2 * L];
    hw[2 * b + 1] = logical[2 * L + 1];
    hw[2 * nblk + hw[2 * b] + (size_t)hw[2 * b + 1] * tiles_m] = (int)b;
  }
  if (slot->map) { (void)hipDeviceSynchronize(); (void)hipFree(slot->map); }   // (a launch may still read the table that goes)
  *slot = GemmScratch::TrapMap();
  int* d = nullptr;
  if (hipMalloc((void**)&d, hw.size() * sizeof(int)) != hipSuccess) return nullptr;
  if (hipMemcpy(d, hw.data(), hw.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d); return nullptr; }
  slot->map = d; slot->tiles_m = tiles_m; slot->tiles_n = tiles_n; slot->tri_off = tri_off; slot->nblk = nblk; slot->used = ++gs.trap_clock;
  *nblk_out = nblk;
  return d;
}

// The thin last tile column of the Schur syrk (n = 5000 = 39 x 128 + 8): C[i, N0 + e] = alpha * <A[:, i], A[:, N0 + e]> + beta * C for
// the r <= 32 edge columns and every row i <= N0 + e.  As a GEMM with 64-wide tiles it took 0.31 ms at K = 20100 (MFMA tiles that
// are mostly padding behind a loader shaped for square tiles: 2.6 TB/s); it is n dot products per edge column over ONE pass
// through A, i.e. a multi-right-hand-side A'x with the edge columns as the x's: CB columns of A per workgroup, the R edge
// columns re-read from L2, 16-byte loads, fixed per-thread row assignment and reduction tree (bitwise reproducible).
template <int CB, int R>
__global__ __launch_bounds__(256) void syrk_edge_kernel(int K, int N, int N0, int e0, int r, const double* __restrict__ A, long lda, double alpha,
                                                        double beta, double* __restrict__ C, long ldc) {
  typedef double d2_t __attribute__((ext_vector_type(2)));
  __shared__ double red[CB][R][4];
  const int i0 = blockIdx.x * CB;
  const double* a[CB];
  const double* x[R];
#pragma unroll
  for (int c = 0; c < CB; ++c) a[c] = A + (long)min(i0 + c, N - 1) * lda;
#pragma unroll
  for (int e = 0; e < R; ++e) x[e] = A + (long)min(N0 + e0 + e, N - 1) * lda;
  // (NS = 2: even and odd rows of a pair in separate sums; CB = 8 has room for one sum per dot product only)
  constexpr int NS = (CB * R <= 32) ? 2 : 1;
  double s[CB][R][NS];
#pragma unroll
  for (int c = 0; c < CB; ++c)
#pragma unroll
    for (int e = 0; e < R; ++e)
#pragma unroll
      for (int h = 0; h < NS; ++h) s[c][e][h] = 0.0;
  const int nfull = K / 512;
  for (int b = 0; b < nfull; ++b) {
    const int k = 512 * b + 2 * threadIdx.x;
    d2_t av[CB], xv[R];
#pragma unroll
    for (int c = 0; c < CB; ++c) av[c] = *reinterpret_cast<const d2_t*>(a[c] + k);
#pragma unroll
    for (int e = 0; e < R; ++e) xv[e] = *reinterpret_cast<const d2_t*>(x[e] + k);
#pragma unroll
    for (int c = 0; c < CB; ++c)
#pragma unroll
      for (int e = 0; e < R; ++e) {
        s[c][e][0] = fma(av[c].x, xv[e].x, s[c][e][0]);
        s[c][e][NS - 1] = fma(av[c].y, xv[e].y, s[c][e][NS - 1]);
      }
  }
  for (int k = 512 * nfull + threadIdx.x; k < K; k += 256) {   // element-wise tail
    double xs[R];
#pragma unroll
    for (int e = 0; e < R; ++e) xs[e] = x[e][k];
#pragma unroll
    for (int c = 0; c < CB; ++c) {
      const double a0 = a[c][k];
#pragma unroll
      for (int e = 0; e < R; ++e) s[c][e][0] = fma(a0, xs[e], s[c][e][0]);
    }
  }
#pragma unroll
  for (int c = 0; c < CB; ++c)
#pragma unroll
    for (int e = 0; e < R; ++e) {
      double t = (NS == 2) ? s[c][e][0] + s[c][e][NS - 1] : s[c][e][0];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off);
      if ((threadIdx.x & 63) == 0) red[c][e][threadIdx.x >> 6] = t;
    }
  __syncthreads();
  if (threadIdx.x < CB * R) {
    const int c = threadIdx.x / R, e = threadIdx.x % R;
    const int i = i0 + c, col = N0 + e0 + e;
    if (i < N && e0 + e < r && i <= col) {
      const double t = (red[c][e][0] + red[c][e][1]) + (red[c][e][2] + red[c][e][3]);
      double* dst = C + (long)col * ldc + i;
      *dst = alpha * t + (beta != 0.0 ? beta * (*dst) : 0.0);
    }
  }
}

// the r <= 32 edge columns of the Schur syrk (columns N0 .. N0 + r - 1, every row up to the diagonal) by syrk_edge_kernel
static bool syrk_edge_ok(const GemmArgs& a) {
  static const bool edge_on = [] { const char* e = getenv("HYP_SYRK_EDGE"); return !(e && atoi(e) == 0); }();
  return edge_on && ((uintptr_t)a.A % 16 == 0) && (a.lda % 2 == 0);
}
static hipError_t syrk_edge_launch(hipStream_t st, const GemmArgs& a, int N0, int r) {
  // columns of A per workgroup: every workgroup re-reads the edge columns from L2 (R K doubles); eight measured better than
  // four at K = 207 360 (config 4: 1.66 vs 2.19 ms per launch) and at K = 20 100 (config 2: syrk phase 8.30 vs 8.36 ms)
  static const int cb_env = [] { const char* e = getenv("HYP_SYRK_EDGE_CB"); return e ? atoi(e) : 8; }();
  const int cb = cb_env;
  for (int e0 = 0; e0 < r; e0 += 8) {
    if (cb == 8) hipLaunchKernelGGL((syrk_edge_kernel<8, 8>), dim3((a.N + 7) / 8), dim3(256), 0, st, a.K, a.N, N0, e0, r, a.A, a.lda, a.alpha, a.beta, a.C, a.ldc);
    else hipLaunchKernelGGL((syrk_edge_kernel<4, 8>), dim3((a.N + 3) / 4), dim3(256), 0, st, a.K, a.N, N0, e0, r, a.A, a.lda, a.alpha, a.beta, a.C, a.ldc);
  }
  return hipGetLastError();
}

hipError_t gemm_f64_launch(hipStream_t st, bool transa, GemmArgs a, GemmScratch* gs) {
  if (a.M <= 0 || a.N <= 0 || a.batch <= 0) return hipSuccess;
  // Schur syrk with a thin last tile column (n = 5000 = 39 x 128 + 8): the 128-wide edge workgroup tiles
  // would do full work for r / 128 useful output (10 % of all tiles at n = 5000).  The r <= 32 edge
  // columns are computed separately as a skinny product with 64-wide tiles and deeper split-K.
  if (a.tag == 1 && transa && a.tri == GEMM_UPPER && a.A == a.B && a.lda == a.ldb && a.batch == 1 && a.epi == 0 &&
      a.krange == KR_ALL && a.M == a.N && a.N >= 1024 && a.N % 128 != 0 && a.N % 128 <= 32 && a.K >= 4096) {
    const int r = a.N % 128, N0 = a.N - r;
    GemmArgs m = a;
    m.M = m.N = N0;
    hipError_t e = gemm_f64_launch(st, transa, m, gs);
    if (e != hipSuccess) return e;
    GemmArgs s = a;                      // C[0:N, N0:N], rows <= columns
    s.tag = 0; s.M = a.N; s.N = r;
    s.B = a.A + (long)N0 * a.lda;
    s.C = a.C + (long)N0 * a.ldc;
    s.tri = GEMM_UPPER_RECT; s.tri_off = N0;
    s.tile_hint = 64; s.splitk_req = 8;
    if (syrk_edge_ok(a)) return syrk_edge_launch(st, a, N0, r);
    return gemm_f64_launch(st, transa, s, gs);
  }
  // tile choice: the 128 x 128 tile unless the product is too small to fill the chip with it
  const long t128 = (long)((a.M + 127) / 128) * ((a.N + 127) / 128) * a.batch;
  const bool small = (a.tile_hint == 64) || (a.tile_hint == 0 && t128 < 192);
  const int BT = small ? 64 : 128;
  a.tiles_m = (a.M + BT - 1) / BT;
  a.tiles_n = (a.N + BT - 1) / BT;
  long nblk;
  const int* trap_map = nullptr;   // Schur-type product on an upper trapezoid: only the tiles that do work, in the XCD-aware order
  static const bool trap_on = [] { const char* e = getenv("HYP_GEMM_TRAP_MAP"); return !(e && atoi(e) == 0); }();
  if (a.tri == GEMM_UPPER) {
    int T = a.tiles_n;   // square
    nblk = (long)T * (T + 1) / 2;
  } else {
    nblk = (long)a.tiles_m * a.tiles_n;
    if (trap_on && gs && a.tag == 1 && transa && a.tri == GEMM_UPPER_RECT && !small && a.batch == 1 && a.epi == 0 && a.krange == KR_ALL && nblk >= 64) {
      long real = 0;
      trap_map = trap_tile_map(*gs, a.tiles_m, a.tiles_n, a.tri_off, &real);
      if (trap_map) nblk = real;
    }
  }
  // split-K for the tall Schur syrk: pick the slice count that minimises the number of rounds of
  // 512 resident workgroups (2 per CU) per unit of work, so the last round is not mostly empty
  a.splitk = 1;
  if (gs && a.tag == 1 && !small && a.batch == 1 && a.epi == 0 && a.krange == KR_ALL && a.K >= 4096) {
    // cost in units of one full-K round of 512 workgroups: rounds(S) / S, plus the partial-sum traffic of S
    // slices (N^2 / 2 doubles written and read back per slice at ~3.5 TB/s against 2.6e-7 K s per round;
    // measured at n = 5000, q = 20100: S = 3 -> 8.71 ms, S = 5 -> 8.42 ms, S = 7 -> 8.45 ms)
    const double eps = (trap_map ? (double)nblk * 16384.0 * 4.6e-12 : (double)a.N * a.N * 2.3e-12) / (2.6e-7 * a.K);
    double best = 1e30;
    for (int S = 1; S <= 8; ++S) {
      if (a.K / S < 1024) break;
      double rounds = (double)((nblk * S + 511) / 512);
      // (round 6) the rounds as the launch below will run them, i.e. with its last round cut into sub-slices: a 225-tile row group of
      // the overlapped exchange is 0.88 of an even split with 2 slices in one round and 0.99 with 3 slices and a last round cut in 3;
      // config 2 (780 tiles, K = 20 100) goes from 7 slices to 5: 7.92 - 7.94 ms against 8.01 - 8.03 back to back
      // (profiles/r06_syrk_slices.txt).  HYP_SYRK_S_PLAIN=1: the choice without the cut.
      static const bool s_plain = [] { const char* e = getenv("HYP_SYRK_S_PLAIN"); return e && atoi(e) == 1; }();
      static const bool tail_sel = [] { const char* e = getenv("HYP_SYRK_TAIL"); return !(e && atoi(e) == 0); }();
      if (!s_plain && tail_sel && transa && (trap_map || a.tri == GEMM_UPPER) && nblk >= 64 && S > 1) {
        const long rem = (nblk * S) % 512;
        const int kc = (((a.K + S - 1) / S) + BK - 1) / BK * BK;
        if (rem > 0 && rem <= nblk) {
          double last = 1.0;
          for (int q = 2; q <= 4; ++q) {
            const double t = (double)((rem * q + 511) / 512) / q;
            if (t < last - 1e-9 && kc / q >= 256) last = t;
          }
          rounds = (double)((nblk * S) / 512) + last;
        }
      }
      const double cost = rounds / S + (S > 1 ? S * eps : 0.0);
      if (cost < best - 1e-9) { best = cost; a.splitk = S; }
    }
    static const int s_env = [] { const char* e = getenv("HYP_SYRK_S"); return e ? atoi(e) : 0; }();
    if (s_env > 0 && a.K / s_env >= 512) a.splitk = s_env;
  }
  // Gram-type products with a small result and a long K (the WSOS cone's L x L x U products of every feasibility check and
  // third-order term: 495 x 495 x 4845 is 36 upper tiles, 303 K steps each -- 1.5 ms of latency on 36 CUs): split K so that
  // about two workgroups per CU exist
  static const bool auto_split = [] { const char* e = getenv("HYP_GEMM_AUTOSPLIT"); return !(e && atoi(e) == 0); }();
  // (round 5: also products on 64-wide tiles with 128 .. 255 of them, the Schur syrk included -- config 3b's 999 x 999 x 5001 product was
  //  136 workgroups of 313 K steps, 312 us on half the chip; HYP_GEMM_AUTOSPLIT_MAX.  A Schur syrk of FEWER tiles stays unsplit: the
  //  1 x 1 x 4845 one of config 5 primal is part of a committed whole-solve fixture whose Cholesky failures -- rounding events at the
  //  last pivots -- the tests want to see, tests/test_hip_fullsize_trajectory.py)
  static const long split_max = [] { const char* e = getenv("HYP_GEMM_AUTOSPLIT_MAX"); return e ? atol(e) : 256L; }();
  if (auto_split && gs && a.splitk_req <= 1 && a.batch == 1 && a.epi == 0 && a.krange == KR_ALL && a.cm_blk == 0 && a.K >= 2048 &&
      ((a.tag != 1 && nblk < 128) || (small && nblk >= 128 && nblk < split_max))) {
    const int S = (int)std::min<long>(std::min<long>(16, 512 / nblk), a.K / 256);
    if (S > 1) a.splitk_req = S;
  }
  // (batch > 1: only on the caller's request -- the candidate screen of the WSOS cone, whose Gram products are a few dozen tiles
  //  per matrix with K = U; the partial sums are laid out [member][slice])
  if (gs && a.splitk_req > 1 && a.epi == 0 && a.krange == KR_ALL && a.cm_blk == 0 && a.K / a.splitk_req >= 256) a.splitk = a.splitk_req;
  a.splitk_base = a.splitk; a.tail_q = 1; a.tail_first = 0; a.tail_chunk = 0;
  if (a.splitk > 1) {
    a.kchunk = (((a.K + a.splitk - 1) / a.splitk) + BK - 1) / BK * BK;
    // The last round of 512 resident workgroups is only partly filled (config 2: 3900 = 7 x 512 + 316, so the launch takes 8
    // workgroup times for 7.6 of work).  Its workgroups are cut into q sub-slices of K: 316 x 3 = 948 short ones fill two
    // rounds of a third of the length -- 7.67 workgroup times.  Only for the Schur syrk with its tile order (the reduction
    // needs the launch position of a tile).  Measured: 8.26 -> 8.18 ms back to back (q = 2: 8.29, 4: 8.17, 6: 8.18; cutting
    // more than the last round's workgroups: slower) -- the workgroups do not run in lock-step rounds, most of the tail was
    // already filled.
    static const bool tail_on = [] { const char* e = getenv("HYP_SYRK_TAIL"); return !(e && atoi(e) == 0); }();
    if (tail_on && gs && a.tag == 1 && transa && (a.tri == GEMM_UPPER || trap_map) && !small && a.batch == 1 && nblk >= 64) {
      const long rem = (nblk * a.splitk) % 512;
      if (rem > 0 && rem <= nblk) {
        int best_q = 1;
        double best = 1.0;
        for (int q = 2; q <= 4; ++q) {
          const double t = (double)((rem * q + 511) / 512) / q;
          if (t < best - 1e-9 && a.kchunk / q >= 256) { best = t; best_q = q; }
        }
        if (best_q > 1) {
          a.tail_q = best_q;
          a.tail_first = (int)(nblk - rem);
          a.tail_chunk = ((a.kchunk + best_q - 1) / best_q + BK - 1) / BK * BK;
          a.splitk = a.splitk_base + best_q - 1;
        }
      }
    }
    a.part_ld = a.M;
    a.part_stride = (long)a.M * a.N;
    const size_t need = (size_t)a.splitk * a.part_stride * sizeof(double) * (size_t)a.batch;
    if (need > gs->splitk_ws_bytes) {
      if (gs->splitk_ws) (void)hipFree(gs->splitk_ws);
      gs->splitk_ws = nullptr; gs->splitk_ws_bytes = 0;
      hipError_t e = hipMalloc((void**)&gs->splitk_ws, need);
      if (e != hipSuccess) return e;
      gs->splitk_ws_bytes = need;
    }
    a.part = gs->splitk_ws;
  }
  a.vec2 = (transa && ((uintptr_t)a.A % 16 == 0) && ((uintptr_t)a.B % 16 == 0) && (a.lda % 2 == 0) && (a.ldb % 2 == 0) &&
            (a.strideA % 2 == 0) && (a.strideB % 2 == 0) && a.krange != KR_GE_M && a.krange != KR_GE_N) ? 1 : 0;
  a.tile_map = nullptr;
  if (gs && a.tag == 1 && a.tri == GEMM_UPPER && !small && a.batch == 1 && nblk >= 64) a.tile_map = upper_tile_map(*gs, a.tiles_n, nblk);
  if (trap_map) a.tile_map = trap_map;
  if (a.tail_q > 1 && !a.tile_map) { a.splitk = a.splitk_base; a.tail_q = 1; }
  dim3 grid((unsigned)nblk, (unsigned)a.batch, (unsigned)a.splitk);
  // (round 6, HYP_SYRK_FUSED_REDUCE=1) the Schur syrk's split-K reduction inside the product's launch: see the kernel's epilogue.
  // Bitwise the separate kernel's matrix, but SLOWER (config 2: syrk phase 8.59 - 8.63 ms against 8.44 - 8.46 with the reduction kernel's
  // 0.2 ms included, same box, alternating): 64 write-through 8-byte stores per lane instead of stores the L2 merges, and a tile's
  // 640 KB of partial sums fetched by ONE workgroup at its latency.  Default off (EXPERIMENTS.md r06-9).
  a.tile_cnt = nullptr;
  static const bool fuse_red = [] { const char* e = getenv("HYP_SYRK_FUSED_REDUCE"); return e && atoi(e) == 1; }();
  if (fuse_red && gs && a.splitk > 1 && a.tag == 1 && transa && !small && a.batch == 1 && a.tile_map) {
    if (gs->tile_cnt_n < nblk) {
      if (gs->tile_cnt) (void)hipFree(gs->tile_cnt);
      gs->tile_cnt = nullptr; gs->tile_cnt_n = 0;
      if (hipMalloc((void**)&gs->tile_cnt, (size_t)nblk * sizeof(int)) == hipSuccess && hipMemset(gs->tile_cnt, 0, (size_t)nblk * sizeof(int)) == hipSuccess)
        gs->tile_cnt_n = nblk;
    }
    if (gs->tile_cnt_n >= nblk) a.tile_cnt = gs->tile_cnt;
  }
  if (a.tag == 1 && transa && !small) {
    hipLaunchKernelGGL((gemm_f64_kernel<true, 4, 1>), grid, dim3(GEMM_THREADS), 0, st, a);
  } else if (transa) {
    if (small) hipLaunchKernelGGL((gemm_f64_kernel<true, 2, 0>), grid, dim3(GEMM_THREADS), 0, st, a);
    else hipLaunchKernelGGL((gemm_f64_kernel<true, 4, 0>), grid, dim3(GEMM_THREADS), 0, st, a);
  } else {
    if (small) hipLaunchKernelGGL((gemm_f64_kernel<false, 2, 0>), grid, dim3(GEMM_THREADS), 0, st, a);
    else hipLaunchKernelGGL((gemm_f64_kernel<false, 4, 0>), grid, dim3(GEMM_THREADS), 0, st, a);
  }
  if (a.splitk > 1 && a.tile_cnt == nullptr)
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((a.M + 255) / 256, a.N, a.batch), dim3(256), 0, st, a.M, a.N, a.tri != GEMM_FULL ? 1 : 0, a.tri_off,
                       a.splitk_base, a.part, a.part_ld, a.part_stride, a.alpha, a.beta, a.C, a.ldc, a.tail_q - 1, a.tail_first,
                       a.tile_map ? a.tile_map + 2 * nblk : nullptr, trap_map ? a.tiles_m : a.tiles_n, (long)a.splitk * a.part_stride, a.strideC);
  return hipGetLastError();
}

// Block columns [c0, c1) of the Schur syrk  C = A'A (upper; `a` = the whole product's arguments, tag 1): c0 = 0 is the syrk of the
// leading c1 columns, c0 > 0 (a multiple of 128, c1 = N) the trapezoid  C[0:N, c0:N], rows <= columns  -- for the factorization
// that starts on the leading block while the rest is still being formed (dense.hip: schur_split_begin).  The slice count of the
// trapezoid is chosen on the tiles that do work (the launch's grid is the full rectangle; tiles below the diagonal exit).
hipError_t schur_syrk_cols(hipStream_t st, GemmArgs a, int c0, int c1, GemmScratch* gs) {
  static const int s_env = [] { const char* e = getenv("HYP_CHOL_SPLIT_S"); return e ? atoi(e) : 0; }();
  if (c0 == 0) {
    GemmArgs l = a;
    l.M = l.N = c1;
    if (s_env > 0) l.splitk_req = s_env;
    return gemm_f64_launch(st, true, l, gs);
  }
  const int N = a.N;
  int r = N % 128, N0 = N - r;
  if (!(r > 0 && r <= 32 && a.K >= 4096 && syrk_edge_ok(a))) { r = 0; N0 = N; }
  GemmArgs t = a;
  t.M = N0; t.N = N0 - c0;
  t.B = a.A + (long)c0 * a.lda;
  t.C = a.C + (long)c0 * a.ldc;
  t.tri = GEMM_UPPER_RECT; t.tri_off = c0;
  if (t.N > 0) {
    const long tm = (t.M + 127) / 128, tn = (t.N + 127) / 128, tc0 = c0 / 128;
    long real = 0;
    for (long j = 0; j < tn; ++j) real += std::min(tm, tc0 + j + 1);
    int S = 1;
    double best = 1e30;
    const double eps = ((double)t.M * t.N * 4.6e-12) / (2.6e-7 * a.K);   // (the partial sums' traffic, as for the square product)
    for (int q = 1; q <= 8; ++q) {
      if (a.K / q < 1024) break;
      const double cost = (double)((real * q + 511) / 512) / q + (q > 1 ? q * eps : 0.0);
      if (cost < best - 1e-9) { best = cost; S = q; }
    }
    if (s_env > 0) S = s_env;
    t.splitk_req = S;
    hipError_t e = gemm_f64_launch(st, true, t, gs);
    if (e != hipSuccess) return e;
  }
  if (r > 0) return syrk_edge_launch(st, a, N0, r);
  return hipSuccess;
}

hipError_t schur_syrk_edge(hipStream_t st, const GemmArgs& a, int* N0) {
  const int r = a.N % 128;
  *N0 = a.N;
  if (!(r > 0 && r <= 32 && a.N >= 1024 && a.K >= 4096 && a.M == a.N && syrk_edge_ok(a))) return hipSuccess;
  *N0 = a.N - r;
  return syrk_edge_launch(st, a, *N0, r);
}

}  // namespace hyp
