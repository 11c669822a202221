#!/usr/bin/env python3
"""bench.py -- IPM iterations/sec of the MI355X hot path on BASELINE.json's configurations.

A "step" is ONE interior-point iteration of Hypatia's CombinedStepper (update_lhs: sqrt-Hessian
products + Schur syrk + Cholesky; four direction solves with refinement; line search).  Inputs are resident
in HBM when the timed region starts.  If the solver converges inside the timed region it is put back at its
initial iterate and keeps stepping: the call that found it converged and the restart are TIME inside the region, not
steps (`restarts_in_timed_region`) -- every counted step is a full iteration.

    python bench.py --gpus N --steps K --warmup W [--config 2|4|2w]
prints ONE JSON line (rank 0): metric/value/unit + roofline (dominant kernel = the FP64-MFMA Schur syrk,
timed with HIP events on the library stream) + cpu_baseline (N = 1: the numpy/scipy restatement in oracle/
timed on the host cores for a bounded number of iterations of the same instance, and a BLAS-3-only lower
bound of the same iteration on all host cores).

Workloads (--config; default "2" at N = 1, "4" at N > 1):
  2   configs[1], the headline: ONE PosSemidefTri cone of side 200 (q = 20100), dense random G (q x n,
      n = 5000), p = 0.  With N > 1 ("strong"): one cone's oracles, factorization and solves do not shard, so
      the model is replicated and the ranks split the K dimension of the Schur product (one all-reduce of
      the n x n partial sums per iteration) -- Amdahl-bound, DESIGN section 6.
  4   configs[3]: 64 x PosSemidefTri(side 80), q = 207 360, n = 5000 -- the FIXED instance at every N
      ("strong" scaling): the cones, with their rows of G / h / z / s, are partitioned over the ranks
      (64 / N per rank), each rank assembles the Schur sum over ITS cones, one RCCL all-reduce (sum, f64,
      n x n) per iteration inside the library, replicated factorization.  value = IPM iterations/s.  The
      same line carries "same_workload_1gpu" (the committed single-GPU figure of THIS instance: the driver's
      own N = 1 run is the headline config 2) and "headline_config2_kshard" (config 2 on the same ranks).
  2w  one PosSemidefTri(200) block of the headline configuration per rank ("weak" scaling; at N = 1 this IS config 2);
      value = blocks x iterations/s.  The N > 1 line of config 4 carries it as "weak_config2_block_per_gpu".
  3b | 5p | 5d  the other configurations (matrix completion / polymin primal / polymin dual) on one GPU: same schema,
      a step = one IPM iteration of the full solve.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

METRIC2 = "IPM iterations/sec (+ ms per KKT solve): dense n=%d, PosSemidefTri side %d (Float64, QRCholDense + CombinedStepper)"
METRIC4 = "IPM iterations/sec (+ ms per KKT solve): dense n=%d, %d x PosSemidefTri side %d, cones sharded over the GPUs (Float64, QRCholDense + CombinedStepper)"
METRIC = "IPM block-iterations/sec: dense n=%d, one PosSemidefTri(%d) block per GPU (Float64, QRCholDense + CombinedStepper)"   # --config 2w
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s achievable
FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X FP64 matrix spec; measured 77.8 with tools/probe_mfma.hip (profiles/)


def blas_threads(want):
    """host BLAS threads for a `threadpool_limits` block: never MORE than the pool was started with.  Under `torch.distributed.run`
    (OMP_NUM_THREADS=1 in every rank) OpenBLAS sizes its buffers for one thread, and raising the limit to 8 afterwards crashed
    scipy's threaded LAPACK calls (dgeqp3 / dorgqr of find_initial_y: SIGSEGV in rank 0 of `--config 5d --gpus 2`)."""
    try:
        from threadpoolctl import threadpool_info
        cur = max([int(d.get("num_threads", 1)) for d in threadpool_info() if d.get("user_api") == "blas"] or [1])
    except Exception:
        cur = 1
    return max(1, min(int(want), cur))


def algorithm_record():
    """which route through the library a bench line timed: the switches are read from the environment by the
    library itself, once per process; the defaults leave the reference's ORDER of operations in four places"""
    on = lambda name: os.environ.get(name, "1")[:1] != "0"
    return {"ens_closed_form_inverse": on("HYP_ENS_CLOSED_INV"), "prox_lower_bound": on("HYP_PROX_LB"),
            "side_by_side_candidate_evaluation": on("HYP_ENS_PREFETCH") and on("HYP_WSOS_PAR"),
            # models of equal PosSemidefTri cones: the rejecting tests of ALL remaining candidates of the schedule at once, the
            # survivor through the sequential test (same accepted step, same iterates: DESIGN.md section 7)
            "line_search_candidate_screen": on("HYP_SEARCH_SCREEN") and on("HYP_PROX_LB"),
            # one large WSOSInterpNonnegative cone (config 5): the next candidates through feasibility chains, gradient solves and the
            # proximity bound together; only rejections are taken from it (csrc/wsos_screen.hip; candidates per batch, 0 = off)
            "wsos_candidate_screen": int(os.environ.get("HYP_WSOS_SCREEN", "4")) if on("HYP_PROX_LB") else 0,
            "triangular_solve_refinement_steps": int(os.environ.get("HYP_TRSM_REFINE", "2")),
            # one-vector solves with a large factor: refinement steps asked for, and whether a plan whose inverted super-blocks
            # measure good enough (rho <= 1e-10: a second step would move nothing above 1e-20) runs one step (round 5; counts: "solve_plans")
            "superblock_solve_refinement_steps": int(os.environ.get("HYP_TRSV_REFINE", "2")), "superblock_solve_adaptive": {"0": "off", "1": "every plan", "2": "system-solver factors", "3": "cone Hessian factors"}.get(os.environ.get("HYP_TRSV_ADAPT", "3"), "?"),
            "triangular_sweeps_one_launch": os.environ.get("HYP_TRSV_ONE_LAUNCH", "2") != "0",
            # (the candidate screen needs the proximity bound: off with HYP_PROX_LB=0)
            # behind a failed Cholesky: the rook-pivoted elimination only from the failing pivot's block on (round 4; 0: the whole matrix)
            "fallback_keeps_cholesky_blocks": on("HYP_BK_HYBRID"),
            # EpiNormSpectral with d1 <= 64 (config 3b): an oracle as one launch of one workgroup; the dual feasibility test decided on
            # rigorous bounds of the nuclear norm in front of every Jacobi sweep (round 5)
            "ens_one_workgroup_oracles": on("HYP_ENS_FUSED"), "ens_dual_test_on_bounds": on("HYP_ENS_DUAL_DECIDE"),
            "reference_route": not (on("HYP_ENS_CLOSED_INV") or on("HYP_PROX_LB") or on("HYP_ENS_PREFETCH") or on("HYP_WSOS_PAR") or on("HYP_BK_HYBRID") or on("HYP_ENS_DUAL_DECIDE") or on("HYP_ENS_FUSED"))}


def plan_stats(lib, ctx):
    """solve plans built so far and how many of them the adaptive rule gave ONE refinement step (hyp_ctx_plan_stats)"""
    try:
        st = (ctypes.c_longlong * 2)()
        lib.hyp_ctx_plan_stats(ctx, st)
        return {"built": int(st[0]), "one_refinement_step": int(st[1])}
    except Exception:
        return None


def pmc_traffic(n, q):
    """HBM bytes per syrk launch from the committed PMC passes (FETCH_SIZE x 2 gfx950 correction + WRITE_SIZE); only
    valid for the configuration they were collected on, else None."""
    try:
        for rnd in ("r06", "r05", "r04"):   # (the newest committed passes)
            path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", rnd + "_pmc_summary.json")
            if not os.path.exists(path):
                continue
            with open(path) as f:
                rec = json.load(f)["syrk"]
            if rec["algorithmic_bytes_per_launch"] == q * n * 8 + n * (n + 1) // 2 * 8:
                return rec["hbm_bytes_per_launch"]
    except Exception:
        pass
    return None


def gen_instance(n, sides, seed):
    """configs[1] generator: G = randn(q, n)/sqrt(n), h = G x0 + svec(I), c = -G' svec(I)."""
    rng = np.random.default_rng(seed)
    dims = [s * (s + 1) // 2 for s in sides]
    q = sum(dims)
    G = np.asfortranarray(rng.standard_normal((q, n)) / np.sqrt(n))
    x0 = rng.standard_normal(n)
    e = np.zeros(q)
    off = 0
    for s, d in zip(sides, dims):
        k = 0
        for i in range(1, s + 1):
            e[off + k] = 1.0
            k += i + 1
        off += d
    h = G @ x0 + e
    c = -(G.T @ e)
    specs = [("possemideftri", d) for d in dims]
    return (c, np.zeros((0, n)), np.zeros(0), G, h, specs, dict(status="Optimal"))


def ref_1gpu(config):
    """the committed single-GPU figure of the SAME workload (profiles/r04_bench_cfg4_1gpu.json, `python bench.py --config 4`):
    the driver's own N = 1 run is the headline configuration (config 2), not this workload"""
    try:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r05_bench_cfg%s_1gpu.json" % config)
        if not os.path.exists(path):
            path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r04_bench_cfg%s_1gpu.json" % config)
        with open(path) as f:
            rec = json.loads(f.read().strip().splitlines()[-1])
        return {"iterations_per_s": rec["iterations_per_s"], "ms_per_step": rec["ms_per_step"], "source": "profiles/" + os.path.basename(path)}
    except Exception:
        return None


def cpu_blas3_bound(n, side, threads):
    """seconds of the three BLAS-3 pieces of one config-2 iteration on the host, all cores (see the call site)"""
    import scipy.linalg as sla
    from threadpoolctl import threadpool_limits
    rng = np.random.default_rng(0)
    dim = side * (side + 1) // 2
    Ui = np.triu(rng.standard_normal((side, side))) / side + np.eye(side)
    ncol = min(n, 1000)                                           # the two-sided products on 1000 of the n columns, scaled
    V = rng.standard_normal((ncol, side, side))
    HG = np.asfortranarray(rng.standard_normal((dim, n)))
    S = rng.standard_normal((n, n + 8)); S = np.asfortranarray(S @ S.T + n * np.eye(n))
    with threadpool_limits(limits=blas_threads(threads), user_api="blas"):
        t0 = time.perf_counter(); W = np.matmul(Ui.T, np.matmul(V, Ui)); t_ts = (time.perf_counter() - t0) * n / ncol
        t0 = time.perf_counter(); sla.blas.dsyrk(1.0, HG, trans=1, lower=0); t_syrk = time.perf_counter() - t0
        t0 = time.perf_counter(); sla.lapack.dpotrf(S, lower=0, overwrite_a=True); t_chol = time.perf_counter() - t0
    del W
    return {"s_per_iteration": t_ts + t_syrk + t_chol, "two_sided_products_s": t_ts, "dsyrk_s": t_syrk, "dpotrf_s": t_chol, "cores": threads,
            "what": "BLAS-3 only, one call each, all host cores; lower bound, not a full iteration"}


def svec_identity(side):
    e = np.zeros(side * (side + 1) // 2)
    k = 0
    for i in range(1, side + 1):
        e[k] = 1.0
        k += i + 1
    return e


def gen_block(n, side, k, seed):
    """rows of G of cone k of the multi-cone instances (configs[3] generator, SURVEY.md 8d: the configs[1] generator per block):
    G_k = randn(dim, n) / sqrt(n) from its own stream (seed, k), so that every rank can draw exactly its cones' rows and the
    instance does not depend on the number of ranks.  float32 draws (8.3 GB of normals at config 4), stored as float64."""
    dim = side * (side + 1) // 2
    rng = np.random.default_rng([seed, k])
    G_k = np.empty((dim, n), order="F")
    scale = 1.0 / np.sqrt(n)
    step = 512
    for j0 in range(0, n, step):
        j1 = min(n, j0 + step)
        G_k[:, j0:j1] = rng.standard_normal((j1 - j0, dim), dtype=np.float32).T * scale
    return G_k


def gen_polymin5(primal_form, seed, keep=None, nvars=4, halfdeg=8):
    """configs[4] generator (examples/polymin/native.jl:56-90 in both forms): interpolation points from twice as many random
    candidates on the box by pivoted QR, a quartic objective sampled at them.  keep = a recorded choice of the points (the pivot order of
    dgeqp3 can depend on the BLAS build; the full-size golden trajectories of tests/golden/ record theirs)."""
    from oracle import polyutils as pu      # (interpolation data = instance data)
    rng = np.random.default_rng(seed)
    U, pts, Ps = pu.interpolate_box([-1.0] * nvars, [1.0] * nvars, halfdeg, rng=rng, sample_factor=2, keep=keep)
    a = rng.uniform(-0.5, 0.5, nvars)
    vals = np.sum((pts - a) ** 2, axis=1) + (pts[:, 0] * pts[:, 1] - pts[:, 2] * pts[:, 3]) ** 2 + 0.3 * pts[:, 0] * pts[:, 2]
    if primal_form:
        inst = (np.array([-1.0]), np.zeros((0, 1)), np.zeros(0), np.ones((U, 1)), vals, [("wsosinterpnonnegative", U, Ps, False)], {})
    else:
        inst = (vals, np.ones((1, U)), np.array([1.0]), -np.eye(U), np.zeros(U), [("wsosinterpnonnegative", U, Ps, True)], {})
    return inst, U, Ps


def emit_json_line(out):
    """the ONE JSON line, guaranteed to be the last thing on stdout: RCCL prints a version banner through C stdio, which (block-
    buffered when stdout is a pipe) would otherwise surface at exit, after the line; whatever a library writes later goes nowhere"""
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    print(json.dumps(out), flush=True)
    try:
        devnull = os.open(os.devnull, os.O_WRONLY)
        os.dup2(devnull, 1)
    except Exception:
        pass


SITE_NAMES = ["schur_sum", "solve_Gtz", "solve_htz", "residual_fused", "residual_htz", "residual_norm", "constant_column", "screen_a", "screen_b",
              "trial_sums", "trial_closing", "residual_products", "host_requested", "screen_agreement", "schur_incl_pack", "other"]


def main_multi(args, world, rank, local_rank):
    """N > 1.  --config 4 (default): the fixed 64 x PosSemidefTri(80) instance, cones partitioned over the ranks (strong scaling);
    --config 2w: one PosSemidefTri(side) block per rank (weak scaling).  Schur matrices summed by one all-reduce per iteration."""
    import torch                      # before the HIP library: one HIP runtime per process (torch's)
    import torch.distributed as dist
    backend = os.environ.get("HYP_DIST_BACKEND", "nccl")
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank % torch.cuda.device_count())
    dist.init_process_group(backend=backend)
    import hypatia_jl_amd as H
    from hypatia_jl_amd import distributed as D
    comm = D.Comm(device="cuda")

    def one(args):
        """one cone-sharded workload on the ranks of this job: returns the record (rank 0) or None"""
        return _run_cone_sharded(args, world, rank, comm, H, D, torch)

    out = one(args)
    strong = (args.config == "4")
    if strong and not args.no_secondary:
        # weak-scaling record of the same job: one PosSemidefTri(200) block of the headline configuration per rank (at N = 1 this
        # IS config 2, the workload of the driver's N = 1 line), value = blocks x iterations/s
        import copy
        aw = copy.copy(args)
        aw.config, aw.steps, aw.warmup, aw.cpu_iters = "2w", 40, 3, 0
        try:
            w = one(aw)
            if rank == 0:
                out["weak_config2_block_per_gpu"] = {k: w[k] for k in ("metric", "value", "unit", "ms_per_step", "iterations_per_s", "steps", "scaling",
                                                                       "config", "phases_ms_per_step", "collectives_per_step")}
        except Exception as e:
            if rank == 0:
                out["weak_config2_block_per_gpu"] = {"error": "%s: %s" % (type(e).__name__, e)}
    return _finish_multi(args, world, rank, local_rank, comm, out, strong, dist)


def _run_cone_sharded(args, world, rank, comm, H, D, torch):
    lib_mod = H._lib
    n = args.n
    strong = (args.config == "4")
    side = 80 if strong else args.side
    ncones = 64 if strong else world
    share = int(os.environ.get("HYP_BENCH_RANK_SHARE", "1"))
    if strong and share > 1:   # (diagnostic, docs/MULTIGPU.md: the cones one rank of `share` holds, through the SHARDED driver at this world size)
        ncones = max(world, 64 // share)
    dim = side * (side + 1) // 2
    q = dim * ncones
    owners = D.partition_cones(ncones, world)
    mine = [k for k, o in enumerate(owners) if o == rank]
    x0 = np.random.default_rng(args.seed).standard_normal(n)
    e_k = svec_identity(side)
    t_gen = time.perf_counter()
    G_r = np.empty((dim * len(mine), n), order="F")
    h = np.zeros(q)
    c = np.zeros(n)
    for i, k in enumerate(mine):
        G_k = gen_block(n, side, k, args.seed)
        G_r[i * dim:(i + 1) * dim] = G_k
        h[k * dim:(k + 1) * dim] = G_k @ x0 + e_k
        c -= G_k.T @ e_k
    comm.allreduce(h)
    comm.allreduce(c)
    t_gen = time.perf_counter() - t_gen
    cones = [D.ShardedCone(comm, owners[k], H.PosSemidefTri(dim) if owners[k] == rank else None, dim, side) for k in range(ncones)]
    model = D.DistModel(comm, c, h, G_r, cones, owners)
    t_setup = time.perf_counter()
    from threadpoolctl import threadpool_limits
    host_threads = max(1, min(args.cpu_threads, (os.cpu_count() or 8) // world))   # untimed host set-up: the ranks share the host cores
    with threadpool_limits(limits=blas_threads(host_threads), user_api="blas"):
        solver = H.Solver(verbose=args.verbose and rank == 0, syssolver=D.DistQRCholDenseSystemSolver(comm))
        solver.load(model)
        solver.setup()
    t_setup = time.perf_counter() - t_setup
    lib, ctx = H._lib.lib(), H._lib.ctx()

    resets = [0]

    def step():   # ONE full IPM iteration: a converged solve goes back to its initial iterate and steps from there,
        while not solver.iterate():   # inside the timed region -- the convergence check that found it and the reset are overhead, not a step
            solver.reset_iterate()
            resets[0] += 1

    from hypatia_jl_amd.solvers import _blas_limit
    blas_cap = _blas_limit()     # (iterate() alone re-enters the host BLAS cap per call; hold it across the run)
    blas_cap.__enter__()
    for _ in range(args.warmup):
        step()
    resets[0] = 0
    lib.hyp_reset_timers(ctx)
    n_solves0 = solver.n_solves
    for f in ("upsys", "upfact", "uprhs", "getdir", "search"):
        setattr(solver, "time_" + f, 0.0)
    comm.barrier()
    torch.cuda.synchronize()
    n_coll0 = comm.n_collectives          # (setup -- the LSQR initial point, rescaling -- and warmup are not counted)
    cs0 = np.zeros(2)
    ct0, ch0 = np.zeros(16), np.zeros(16, dtype=np.int64)
    try:
        H._lib.check(lib.hyp_sys_comm_stats(solver.syssolver.local._h, H._lib.vec_ptr(cs0)), "hyp_sys_comm_stats")
        H._lib.check(lib.hyp_sys_comm_times(solver.syssolver.local._h, H._lib.vec_ptr(ct0)), "hyp_sys_comm_times")
        H._lib.check(lib.hyp_sys_comm_hist(solver.syssolver.local._h, ch0.ctypes.data_as(ctypes.c_void_p)), "hyp_sys_comm_hist")
    except Exception:
        pass
    if comm.hist is not None:
        comm.hist.clear()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    n_coll = comm.n_collectives - n_coll0
    coll_hist = dict(comm.hist) if comm.hist is not None else None
    comm.barrier()
    el = np.array([time.perf_counter() - t0])
    blas_cap.__exit__(None, None, None)
    comm.allreduce(el, "max")
    elapsed = float(el[0])
    ks = (ctypes.c_double * 8)()
    lib.hyp_get_kernel_stats(ctx, ks)
    cs = np.zeros(2)
    ct, chh = np.zeros(16), np.zeros(16, dtype=np.int64)
    try:
        H._lib.check(lib.hyp_sys_comm_stats(solver.syssolver.local._h, H._lib.vec_ptr(cs)), "hyp_sys_comm_stats")
        H._lib.check(lib.hyp_sys_comm_times(solver.syssolver.local._h, H._lib.vec_ptr(ct)), "hyp_sys_comm_times")
        H._lib.check(lib.hyp_sys_comm_hist(solver.syssolver.local._h, chh.ctypes.data_as(ctypes.c_void_p)), "hyp_sys_comm_hist")
    except Exception:
        pass
    # what every rank saw (the first N > 1 run must say where a shortfall comes from): per rank its phases, its Schur exchange -- the
    # collective alone and with the triangle's pack / unpack --, its small exchanges, and how long it waited at the end of the region
    steps_ = max(args.steps, 1)
    dct, dch = ct - ct0, chh - ch0
    small_ms = float(dct[1:14].sum() + dct[15])
    mine_row = np.array([ks[0] / steps_, ks[1] / steps_, ks[2] / steps_, solver.time_upsys / steps_ * 1e3, solver.time_getdir / steps_ * 1e3,
                         solver.time_search / steps_ * 1e3, dct[0] / steps_, dct[14] / steps_, small_ms / steps_, float(dch[0]) / steps_,
                         float(dch[1:14].sum() + dch[15]) / steps_, float(len(mine))])
    table = np.zeros((world, mine_row.size))
    table[rank] = mine_row
    comm.allreduce(table)
    if rank == 0:
        syrk_ms = ks[1] / max(ks[4], 1)
        q_local = dim * len(mine)
        syrk_flops = float(n) * n * q_local               # this rank's share of n^2 q
        achieved = syrk_flops / (syrk_ms * 1e-3) / 1e12 if syrk_ms > 0 else 0.0
        its = args.steps / elapsed
        in_lib = bool(getattr(solver.syssolver, "rccl_in_library", False))
        out = {
            "metric": (METRIC4 % (n, ncones, side)) if strong else (METRIC % (n, side)),
            "value": its if strong else world * its,
            "unit": "iterations/s" if strong else "block-iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "restarts_in_timed_region": resets[0],   # (a converged solve restarts from its initial iterate: time counted, no step counted)
            "ms_per_step": elapsed / args.steps * 1e3, "iterations_per_s": its,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": ("configs[3]: %d x PosSemidefTri side=%d (q=%d), dense random G q x n, n=%d, p=0; the fixed instance at every N, "
                                    "%d cones per rank" % (ncones, side, q, n, len(mine))
                                    + (" -- RANK-SHARE EMULATION (HYP_BENCH_RANK_SHARE=%s): only the cones one rank of that many holds, through the sharded "
                                       "driver; a different model, read its phases only" % os.environ["HYP_BENCH_RANK_SHARE"]
                                       if int(os.environ.get("HYP_BENCH_RANK_SHARE", "1")) > 1 else "")) if strong else
                                   ("configs[1] block per GPU: %d x PosSemidefTri side=%d (q=%d total), dense random G, n=%d, p=0" % (world, side, q, n)),
                       "n": n, "q": q, "seed": args.seed, "algorithm": algorithm_record(), "parallelism": "cone-shard x%d" % world,
                       "exchange": "RCCL all-reduce (sum, f64, n x n) of the Schur matrix per iteration + small per-solve / per-trial all-reduces, "
                                   + ("issued by the library on its own stream (hyp_sys_set_comm_rccl)" if in_lib else "through the torch.distributed callback")},
            "roofline": {"bound": "mfma", "kernel": "gemm_f64_kernel<true, 4, 1> + splitk_reduce (per-rank Schur syrk, upper)", "achieved": achieved,
                         "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP64_MFMA_PEAK_TFLOPS, "traffic": None,
                         "launch_ms": syrk_ms, "flops_per_launch": syrk_flops},
            "phases_ms_per_step": {"sqrt_hess_prod": ks[0] / args.steps, "syrk": ks[1] / args.steps, "cholesky": ks[2] / args.steps,
                                   "update_lhs": solver.time_upsys / args.steps * 1e3, "get_directions": solver.time_getdir / args.steps * 1e3,
                                   "search": solver.time_search / args.steps * 1e3},
            "kkt_solves_per_step": (solver.n_solves - n_solves0) / args.steps,
            "ms_per_kkt_solve": solver.time_getdir / max(solver.n_solves - n_solves0, 1) * 1e3,
            "collectives_per_step": (n_coll + cs[0] - cs0[0]) / max(args.steps, 1),
            "library_exchanges_per_step": (cs[0] - cs0[0]) / max(args.steps, 1),
            "library_exchange_MB_per_step": (cs[1] - cs0[1]) * 8e-6 / max(args.steps, 1),
            # device time of the exchanges on rank 0 (HIP events around each ncclAllReduce on the library's stream; host clock around the
            # callback on the torch transport), per iteration
            "schur_allreduce_ms": dct[0] / steps_,
            "schur_exchange_ms_incl_pack": dct[14] / steps_,
            "small_collectives_ms_per_step": small_ms / steps_,
            "small_collectives_ms_by_site": {SITE_NAMES[i]: dct[i] / steps_ for i in range(16) if i not in (0, 14) and dct[i] > 0},
            "per_rank_ms_per_step": {"columns": ["sqrt_hess_prod", "syrk", "cholesky", "update_lhs", "get_directions", "search", "schur_allreduce",
                                                 "schur_exchange_incl_pack", "small_collectives", "schur_exchanges", "small_exchanges", "cones"],
                                     "rows": [[round(float(v), 4) for v in row] for row in table]},
            "setup_s": t_setup, "instance_generation_s": t_gen,
            "same_workload_1gpu": ref_1gpu(args.config),
        }
        if coll_hist is not None:   # HYP_PROFILE=1: where the collectives of the timed region come from
            for key, cnt in sorted(coll_hist.items(), key=lambda kv: -kv[1]):
                print("collectives %-40s %8d doubles op %-5s : %.1f per step" % (key[0], key[1], key[2], cnt / max(args.steps, 1)), file=sys.stderr)
    else:
        out = None
    try:   # (the library's communicator of this solver goes before the next workload brings up its own)
        solver.syssolver.close()
    except Exception:
        pass
    solver = model = cones = G_r = None
    import gc
    gc.collect()
    return out


def _finish_multi(args, world, rank, local_rank, comm, out, strong, dist):
    if strong and not args.no_secondary:
        # secondary record of the same job: the HEADLINE workload (config 2, the one the N = 1 bench line is quoted on) on the
        # same ranks -- one cone, model replicated, K-panel shard of the Schur product
        import copy
        a2 = copy.copy(args)
        a2.config, a2.steps, a2.warmup, a2.cpu_iters = "2", args.secondary_steps, 3, 0
        try:
            sec = run_headline(a2, world, rank, local_rank, True, comm=comm)
            if rank == 0:
                out["headline_config2_kshard"] = {k: sec[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "scaling", "config", "roofline",
                                                                     "phases_ms_per_step", "kkt_solves_per_step", "ms_per_kkt_solve")}
        except Exception as e:   # the secondary record must never cost the primary line
            if rank == 0:
                out["headline_config2_kshard"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if rank == 0:
        emit_json_line(out)
    try:
        dist.destroy_process_group()
    except Exception:
        pass


def run_headline(args, world, rank, local_rank, multi, comm=None):
    """the headline workload (config 2; --config 4 on one GPU): measure and return the JSON record; with several ranks the model is
    replicated and the Schur product K-sharded (KShardQRCholDenseSystemSolver)"""
    own_group = False
    if multi and comm is None:   # config 2 on N GPUs: ONE cone -- the model is replicated, the ranks split the K dimension of the Schur product
        import torch                      # before the HIP library: one HIP runtime per process (torch's)
        import torch.distributed as dist
        if torch.cuda.is_available():
            torch.cuda.set_device(local_rank % torch.cuda.device_count())
        dist.init_process_group(backend=os.environ.get("HYP_DIST_BACKEND", "nccl"))
        own_group = True
    import hypatia_jl_amd as H
    if multi:
        from hypatia_jl_amd import distributed as D
        if comm is None:
            comm = D.Comm(device="cuda")
        args.cpu_iters = 0

    t_setup = time.perf_counter()
    from threadpoolctl import threadpool_limits
    if args.config == "4":      # the multi-GPU workload on one GPU: the same blocks from the same streams
        side4, nc4 = 80, 64
        # HYP_BENCH_RANK_SHARE=N (diagnostic, docs/MULTIGPU.md): only the first 64 / N blocks -- the cones ONE rank of N holds -- so that the
        # phases of this line are what a rank computes per iteration at N ranks, minus the exchanges.  A different (smaller) model: its
        # phase times are the measurement, not its iterations/s; the line says so in config.workload.
        share = int(os.environ.get("HYP_BENCH_RANK_SHARE", "1"))
        if share > 1:
            nc4 = max(1, 64 // share)
        dim4 = side4 * (side4 + 1) // 2
        G = np.empty((dim4 * nc4, args.n), order="F")
        for k in range(nc4):
            G[k * dim4:(k + 1) * dim4] = gen_block(args.n, side4, k, args.seed)
        x0 = np.random.default_rng(args.seed).standard_normal(args.n)
        e = np.tile(svec_identity(side4), nc4)
        with threadpool_limits(limits=blas_threads(args.cpu_threads), user_api="blas"):
            inst = (-(G.T @ e), np.zeros((0, args.n)), np.zeros(0), G, G @ x0 + e, [("possemideftri", dim4)] * nc4, dict(status="Optimal"))
        args.cpu_iters = 0      # (the CPU port needs ~20 s per iteration here; the baseline is quoted on the headline configuration)
    else:
        inst = gen_instance(args.n, [args.side], args.seed)
    q = inst[3].shape[0]
    with threadpool_limits(limits=blas_threads(args.cpu_threads), user_api="blas"):   # host preprocessing (rescale, QR for the initial x): untimed setup
        solver = H.Solver(verbose=args.verbose and rank == 0, init_use_indirect=(args.config == "4"),
                          syssolver=(D.KShardQRCholDenseSystemSolver(comm) if comm is not None else None))
        solver.load(H.make_model(inst))
        solver.setup()
    t_setup = time.perf_counter() - t_setup
    lib, ctx = H._lib.lib(), H._lib.ctx()

    resets = [0]

    def step():   # ONE full IPM iteration: a converged solve goes back to its initial iterate and steps from there,
        while not solver.iterate():   # inside the timed region -- the convergence check that found it and the reset are overhead, not a step
            solver.reset_iterate()
            resets[0] += 1

    from hypatia_jl_amd.solvers import _blas_limit
    blas_cap = _blas_limit()     # (iterate() alone re-enters the host BLAS cap per call; hold it across the run)
    blas_cap.__enter__()
    for _ in range(args.warmup):
        step()
    resets[0] = 0
    lib.hyp_reset_timers(ctx)
    if hasattr(lib, "reset"):
        lib.reset()
    n_solves0 = solver.n_solves
    n_trials0 = solver.stepper.searcher.n_trials
    screen_stats = getattr(solver.syssolver, "search_screen_stats", None)
    screens0 = screen_stats() if screen_stats is not None else (0, 0)
    for f in ("upsys", "upfact", "uprhs", "getdir", "search"):
        setattr(solver, "time_" + f, 0.0)
    lib.hyp_ctx_synchronize(ctx)      # (every C-ABI call is synchronous; this is the device-wide fence)
    if comm is not None:
        comm.barrier()
        comm.torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    lib.hyp_ctx_synchronize(ctx)
    if comm is not None:
        comm.torch.cuda.synchronize()
        comm.barrier()
    elapsed = time.perf_counter() - t0
    if comm is not None:
        el = np.array([elapsed])
        comm.allreduce(el, "max")
        elapsed = float(el[0])
    blas_cap.__exit__(None, None, None)   # (the CPU baseline below runs with the full host BLAS pool)

    ks = (ctypes.c_double * 8)()
    lib.hyp_get_kernel_stats(ctx, ks)
    syrk_ms = ks[1] / max(ks[4], 1)
    nmp = solver.model.n - solver.model.p
    syrk_flops = float(nmp) * nmp * q                     # algorithmic: n^2 q
    if comm is not None:
        r0, r1 = D.kshard_range(q, rank, world)
        syrk_flops = float(nmp) * nmp * (r1 - r0)         # this rank's K panel
    achieved = syrk_flops / (syrk_ms * 1e-3) / 1e12 if syrk_ms > 0 else 0.0
    ms_per_step = elapsed / args.steps * 1e3
    n_solves = solver.n_solves - n_solves0
    n_trials = solver.stepper.searcher.n_trials - n_trials0

    out = {
        "metric": (METRIC2 % (args.n, args.side)) if args.config == "2" else (METRIC4 % (args.n, 64, 80)),
        "value": args.steps / elapsed,
        "unit": "iterations/s",
        "iterations_per_s": args.steps / elapsed,
        "n_gpus": world if comm is not None else 1,
        "steps": args.steps,
        "restarts_in_timed_region": resets[0],   # (a converged solve restarts from its initial iterate: time counted, no step counted)
        # rounds 1-2 and the first part of round 3 counted the call that only FOUND the solve converged (and restarted it) as a
        # step -- one call in 14 at config 2, which made those figures ~7 % optimistic; the same quotient for comparison with them:
        "ms_per_step_if_restarts_counted_as_steps": elapsed / (args.steps + resets[0]) * 1e3,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "strong" if (args.config == "4" or comm is not None) else None,   # (one GPU, no comm: nothing scales)
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": ("configs[1]: single PosSemidefTri side=%d (q=%d), dense random G q x n, n=%d, p=0" % (args.side, q, args.n)) if args.config == "2"
                               else (("configs[3]: 64 x PosSemidefTri side=80 (q=%d), dense random G q x n, n=%d, p=0, all cones on one GPU" % (q, args.n))
                                     if int(os.environ.get("HYP_BENCH_RANK_SHARE", "1")) <= 1 else
                                     ("RANK-SHARE EMULATION (HYP_BENCH_RANK_SHARE=%s): the first %d of configs[3]'s 64 PosSemidefTri(80) blocks (q=%d), n=%d -- the cones one "
                                      "rank holds; a different model, read its phases_ms_per_step only" % (os.environ["HYP_BENCH_RANK_SHARE"], q // 3240, q, args.n))),
                   "n": args.n, "q": q, "seed": args.seed, "algorithm": algorithm_record(),
                   **({"parallelism": "K-panel shard of the Schur product x%d, model replicated; one RCCL all-reduce (sum, f64, n x n) per iteration" % world}
                      if comm is not None else {})},
        "roofline": {"bound": "mfma", "kernel": "gemm_f64_kernel<true, 4, 1> (Schur syrk, upper)", "achieved": achieved,
                     "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP64_MFMA_PEAK_TFLOPS,
                     "traffic": pmc_traffic(args.n, q), "traffic_unit": "bytes of HBM traffic per launch (rocprofv3 PMC, profiles/r06_pmc_summary.json; r05_ / r04_ if absent)",
                     "launch_ms": syrk_ms, "flops_per_launch": syrk_flops},
        "phases_ms_per_step": {"sqrt_hess_prod": ks[0] / args.steps, "syrk": ks[1] / args.steps, "cholesky": ks[2] / args.steps,
                               "update_lhs": solver.time_upsys / args.steps * 1e3, "get_directions": solver.time_getdir / args.steps * 1e3,
                               "update_rhs": solver.time_uprhs / args.steps * 1e3, "search": solver.time_search / args.steps * 1e3},
        "kkt_solves_per_step": n_solves / args.steps,
        "ms_per_kkt_solve": solver.time_getdir / max(n_solves, 1) * 1e3,
        "search_trials_per_step": n_trials / args.steps, "solve_plans": plan_stats(lib, ctx),
        # of those, the candidates the side-by-side screen rejected (batches of up to SCREEN_MAX = 18 = the whole schedule, one read-back each) -- the others
        # went through the sequential acceptance test
        "search_screens_per_step": ((screen_stats()[0] - screens0[0]) / args.steps) if screen_stats is not None else 0.0,
        "search_trials_screened_out_per_step": ((screen_stats()[1] - screens0[1]) / args.steps) if screen_stats is not None else 0.0,
        "setup_s": t_setup,
        # the reference's ten timers (Solvers.jl:86-96): set-up ones for the whole solve set-up, the others per timed step
        "hypatia_timers_s": {"rescale": getattr(solver, "time_rescale", 0.0), "initx": getattr(solver, "time_initx", 0.0),
                             "inity": getattr(solver, "time_inity", 0.0), "unproc": getattr(solver, "time_unproc", 0.0),
                             "loadsys": getattr(solver, "time_loadsys", 0.0),
                             "upsys_per_step": solver.time_upsys / args.steps, "upfact_per_step": solver.time_upfact / args.steps,
                             "uprhs_per_step": solver.time_uprhs / args.steps, "getdir_per_step": solver.time_getdir / args.steps,
                             "search_per_step": solver.time_search / args.steps},
    }

    # the other two MFMA-bound kernels of update_lhs_fact, by the same rule (algorithmic flops of SURVEY 8d over the HIP-event time of the
    # phase inside the timed region): the PSD cone's two-sided product on the q x n block (side^3 (1 + 2/3) flop per column: a
    # triangular factor on both sides, the upper triangle of the result) and the Cholesky of the Schur matrix (n^3 / 3)
    if args.config == "2" and comm is None and ks[0] > 0 and ks[2] > 0:
        ts_flops = (5.0 / 3.0) * float(args.side) ** 3 * args.n
        ts_ms, ch_ms = ks[0] / args.steps, ks[2] / args.steps
        out["roofline_sqrt_hess_prod"] = {"bound": "mfma", "kernel": "psd_ts4_kernel (U^-T V U^-1 on the %d columns of G; sides 129 .. 208, else the two-pass kernels)" % args.n,
                                          "achieved": ts_flops / ts_ms / 1e9, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                          "frac": ts_flops / ts_ms / 1e9 / FP64_MFMA_PEAK_TFLOPS, "launch_ms": ts_ms, "flops_per_launch": ts_flops}
        ch_flops = float(args.n) ** 3 / 3.0
        out["roofline_cholesky"] = {"bound": "latency (39 dependent block steps; DESIGN.md section 5)", "kernel": "potrf_tiles4_kernel + potrf_panel_mfma_kernel + look-ahead GEMMs",
                                    "achieved": ch_flops / ch_ms / 1e9, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                    "frac": ch_flops / ch_ms / 1e9 / FP64_MFMA_PEAK_TFLOPS, "launch_ms": ch_ms, "flops_per_launch": ch_flops}

    # the HBM-bound part of the path: the passes over the resident G that every KKT solve is made of
    # (qrchol.jl:51-53, 71-73), timed with HIP events after the timed region; algorithmic bytes per pass = q * n * 8
    try:
        g4 = np.zeros(4)
        L = H._lib
        L.check(lib.hyp_sys_bench_gemv(solver.syssolver._h, 20, L.vec_ptr(g4)), "bench_gemv")
        gb = q * args.n * 8.0
        out["roofline_solve"] = {"bound": "hbm", "kernel": "gemv_t_multi_kernel<2> (G' X, two right-hand sides per pass over G)",
                                 "achieved": gb / g4[0] / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gb / g4[0] / 1e6 / HBM_PEAK_GBS,
                                 "bytes_per_launch": gb, "launch_ms": g4[0],
                                 "other_passes_GBs": {"G X (2 rhs)": gb / g4[1] / 1e6, "G' x": gb / g4[2] / 1e6, "G x": gb / g4[3] / 1e6}}
    except Exception as e:   # measurement extra: never fails the bench line
        print("roofline_solve skipped: %r" % (e,), file=sys.stderr)

    if args.cpu_iters > 0:
        # CPU baseline: the oracle restatement ("port") on the host cores, a bounded number of iterations
        from oracle.build import make_model as omodel
        from oracle.solvers import Solver as OSolver
        # host BLAS pool of the baseline: 8 threads is the fastest setting measured for this port on the GPU box
        # (tools/cpu_baseline_threads.py: 1 -> 8.4 s, 8 -> 3.7 s, 32 -> 7.0 s, 256 -> 17.6 s per iteration; the
        # per-column loop of the PSD products is many small BLAS calls, which large pools slow down)
        from threadpoolctl import threadpool_limits
        with threadpool_limits(limits=blas_threads(args.cpu_threads), user_api="blas"):
            os_ = OSolver(verbose=False, iter_limit=args.cpu_iters)
            os_.load(omodel(inst))
            os_.solve()
        cpu_it_s = os_.num_iters / os_.iter_time if os_.iter_time > 0 else 0.0
        out["cpu_baseline"] = {"value": cpu_it_s, "unit": "iterations/s", "cores": args.cpu_threads, "host_cores": os.cpu_count(), "kind": "port",
                               "sample": "first %d IPM iterations of the same instance; numpy/scipy restatement: OpenBLAS (%d threads) for the syrk / "
                                         "Cholesky / gemv, the per-column dtrsm loop of the PSD products is sequential as in the reference "
                                         "(possemideftri.jl:168-174) and Python-bound here" % (os_.num_iters, args.cpu_threads),
                               "s_per_iteration": (os_.iter_time / max(os_.num_iters, 1))}
        out["speedup_vs_cpu_port"] = out["value"] / cpu_it_s if cpu_it_s > 0 else None
        # Second, best-effort CPU figure that is not Python-bound: only the three BLAS-3 pieces of one iteration, each as ONE
        # library call on all host cores -- the PSD block as two batched dgemm with U^-1 (instead of the reference's per-column
        # dtrsm loop), dsyrk for the Schur matrix, dpotrf -- a LOWER bound of what any CPU implementation of the iteration
        # needs on this host (directions, line search, pack / unpack not counted).  Hypatia.jl itself was not run (no Julia here).
        try:
            out["cpu_baseline"]["blas3_lower_bound"] = cpu_blas3_bound(args.n, args.side, os.cpu_count() or 8)
            lb = out["cpu_baseline"]["blas3_lower_bound"]["s_per_iteration"]
            out["speedup_vs_cpu_bracket"] = {"vs_port": out["speedup_vs_cpu_port"], "vs_blas3_lower_bound": (lb * out["value"]) if lb > 0 else None}
            # the ratio to quote: against a BLAS-3-only lower bound of the CPU iteration on ALL host cores; the ratio against the
            # Python-bound numpy port on 8 threads (speedup_vs_cpu_port) says more about the port than about either machine
            out["speedup_vs_cpu"] = {"value": out["speedup_vs_cpu_bracket"]["vs_blas3_lower_bound"], "against": "cpu_baseline.blas3_lower_bound",
                                     "secondary_vs_numpy_port": out["speedup_vs_cpu_port"]}
        except Exception as e:
            print("blas3 bound skipped: %r" % (e,), file=sys.stderr)
    if hasattr(lib, "report"):
        print(lib.report(), file=sys.stderr)
        print("host wall in timed region: %.1f ms/step" % ms_per_step, file=sys.stderr)
    solver = None   # (release the device memory before a following workload)
    if own_group:
        if rank == 0:
            emit_json_line(out)
        try:
            comm.dist.destroy_process_group()
        except Exception:
            pass
    return out


def main_other(args, world=1, rank=0, local_rank=0, multi=False):
    """--config 3b | 5p | 5d: the other BASELINE.json configurations on one GPU, same JSON schema.  A step is one IPM iteration
    of the full solve of the instance (the solve is repeated until `steps` iterations have been timed).
      3b  matrix completion, EpiNormSpectral 50 x 100 (dim 5001): the largest size the reference's algorithm admits
      5p  polymin, WSOSInterpNonnegative, 4 variables, half-degree 8 (U = 4845), primal form (the cone uses the dual barrier)
      5d  the same in dual form (n - p = 4844 unknowns in the Schur system, the "MFMA Hessian-product" form)
    --config 5p | 5d --gpus N (N > 1; BASELINE configs[4] "1 -> 8 GPU"): ONE WSOS cone, so the model and the iterate are replicated and the
    ranks split the K dimension (the cone's U rows) of the Schur product -- KShardQRCholDenseSystemSolver, one all-reduce of the n x n
    partial sums per iteration; the cone's oracles (feasibility chains, gradient, U x U Hessian and its Cholesky per accepted trial),
    the Schur factorization, the solves and the line search run replicated and bitwise identical on every rank.  "strong" scaling with
    an Amdahl bound the line states: only update_lhs's product divides by N.
    roofline: the blocked Cholesky of the cone's dim x dim Hessian (Cones.jl:239-251), the dominant kernel chain of every
    accepted line-search trial, timed in isolation with HIP events (hyp_bench_potrf); 3b with the closed-form inverse has no
    such factorization in its loop and reports the Schur-matrix Cholesky instead."""
    comm = None
    if multi:
        import torch                      # before the HIP library: one HIP runtime per process (torch's)
        import torch.distributed as dist
        if torch.cuda.is_available():
            torch.cuda.set_device(local_rank % torch.cuda.device_count())
        dist.init_process_group(backend=os.environ.get("HYP_DIST_BACKEND", "nccl"))
    import hypatia_jl_amd as H
    if multi:
        from hypatia_jl_amd import distributed as D
        comm = D.Comm(device="cuda")
    from oracle import instances as I       # instance generators only (data)
    from threadpoolctl import threadpool_limits
    t_setup = time.perf_counter()
    solver_opts = {}
    mk_solver = lambda **kw: H.Solver(syssolver=(D.KShardQRCholDenseSystemSolver(comm) if comm is not None else None), **kw)
    if args.config == "3b":
        inst = I.matrixcompletion(50, 100, seed=args.seed)
        work = "configs[2] at the largest size the reference admits: matrix completion, EpiNormSpectral 50 x 100 (dim 5001)"
    elif args.config == "3c":
        # configs[2] PAST the reference's limit: its generic inv_hess_prod! needs the explicit dim x dim Hessian and a Cholesky of
        # it per line-search trial (Cones.jl:113-118: 65 GB and 2.4e14 flop per trial at 300 x 300), the closed-form inverse
        # Hessian of this library needs neither -- what binds is HBM for the dense G.  Known positions drawn as the reference
        # draws them (examples/matrixcompletion/native.jl:29-43).  NOT the reference's algorithm for this cone.
        assert os.environ.get("HYP_ENS_CLOSED_INV", "1")[:1] != "0", "--config 3c needs the closed-form inverse Hessian"
        sd = args.mc_side
        inst = I.matrixcompletion(sd, sd, seed=args.seed, with_replacement=True)
        work = ("configs[2] beyond the reference's own size limit: matrix completion, EpiNormSpectral %d x %d (dim %d), closed-form inverse "
                "Hessian -- not the reference's algorithm for this cone" % (sd, sd, 1 + sd * sd))
        solver_opts = dict(init_use_indirect=True)   # (G has orthogonal columns: LSQR, the reference's init_use_indirect, ends in three steps)
    else:
        inst, U, Ps = gen_polymin5(args.config == "5p", args.seed)
        work = "configs[4]: polymin, WSOSInterpNonnegative, 4 variables, half-degree 8 (U = %d), %s form" % (U, "primal" if args.config == "5p" else "dual")
    t_setup = time.perf_counter() - t_setup
    steps = args.steps if args.steps is not None else 40
    lib, ctx = H._lib.lib(), H._lib.ctx()
    iters, loop_s, solves, trials, nsolve_runs = 0, 0.0, 0, 0, 0
    phases = dict(upsys=0.0, getdir=0.0, search=0.0)
    status = None
    with threadpool_limits(limits=blas_threads(args.cpu_threads), user_api="blas"):
        if args.config != "3c":
            warm = mk_solver(verbose=False, iter_limit=2, **solver_opts)          # untimed: first-touch allocations, kernel loading
            warm.load(H.make_model(inst)); warm.solve()
            if comm is not None:
                warm.syssolver.close()
        lib.hyp_reset_timers(ctx)                             # (the executed-work counters of hyp_get_kernel_stats start here)
        if args.config == "3c":
            steps = 1                                         # one whole solve
        up_s, comm_calls, comm_doubles = 0.0, 0.0, 0.0
        while iters < steps:
            s = mk_solver(verbose=args.verbose and rank == 0, **solver_opts)
            s.load(H.make_model(inst))
            if comm is not None:     # the timed region of a solve = its iteration loop, bracketed on every rank
                comm.barrier()
            s.solve()
            it_time = s.iter_time
            if comm is not None:
                v = np.array([it_time])
                comm.allreduce(v, "max")        # (MAX over ranks of the same replicated loop)
                it_time = float(v[0])
                cs = (ctypes.c_double * 2)()
                lib.hyp_sys_comm_stats(s.syssolver._h, cs)
                comm_calls += cs[0]; comm_doubles += cs[1]
                s.syssolver.close()
            iters += s.num_iters; loop_s += it_time; solves += s.n_solves; trials += s.stepper.searcher.n_trials
            up_s += s.time_upsys
            for k in phases:
                phases[k] += getattr(s, "time_" + k)
            status = s.status
            nsolve_runs += 1
    # ---- roofline of the ITERATION: algorithmic flops of what one iteration does / its wall time, next to the
    # same with only the work that was actually executed counted (the proximity lower bound skips Hessians the reference would
    # assemble and factor), and the dominant kernel of the committed trace of this very command (profiles/)
    ks = (ctypes.c_double * 8)()
    lib.hyp_get_kernel_stats(ctx, ks)
    n_upfact, n_hfact, n_bk, n_grad = ks[3], ks[5], ks[6], ks[7]
    nm = s.model.n - s.model.p
    cone = s.model.cones[0]
    dimc = cone.dimension()
    if args.config in ("3b", "3c"):
        d1, d2 = (50, 100) if args.config == "3b" else (args.mc_side, args.mc_side)
        # closed-form oracles: d1^2 d2-sized GEMMs per column; Schur assembly through the hess_prod branch (qrchol.jl:240-246):
        # H G (nm columns; per column A W', S tau and the two triangular sweeps of Z^-1 (.): 6 d1^2 d2, epinormspectral.jl:211-239)
        # + G'(H G) (nm^2 q) + Cholesky nm^3 / 3
        f_col = 6.0 * d1 * d1 * d2
        f_uplhs = nm * f_col + float(nm) ** 2 * s.model.q + float(nm) ** 3 / 3
        f_trial_ref = float(dimc) ** 3 / 3 + 2.0 * d1 * d1 * d2          # the reference's per-trial explicit-Hessian Cholesky (Cones.jl:113-118)
        f_trial_exec = 40.0 * d1 * d1 * d2                                # (estimate) Z, its Cholesky, tau, two Jacobi decompositions, the closed-form inverse on two columns
        f_solve = 4.0 * s.model.q * nm + 2.0 * nm * nm
        trace_kernel = ("launch-bound: ~900 kernels of ~10 us per iteration; largest share jacobi_lds_kernel (profiles/r05_cfg3b_kernel_stats.csv)"
                        if args.config == "3b" else "gemm_f64_kernel (G' (H G), n^2 q flop) and the blocked Cholesky of the n x n Schur matrix")
        alg = (f_uplhs * n_upfact + trials * f_trial_ref + solves * f_solve) / iters
        exe = (f_uplhs * n_upfact + trials * f_trial_exec + n_hfact * float(dimc) ** 3 / 3 + solves * f_solve) / iters
    else:
        Ls = [P.shape[1] for P in Ps]
        f_feas = sum(2.0 * U * L * L + L ** 3 / 3.0 for L in Ls)          # Lambda_k = P_k' diag(x) P_k and its Cholesky (wsosinterpnonnegative.jl:89-117)
        f_grad = sum(1.0 * L * L * U for L in Ls)                         # L_k^-1 P_k' (:119-133)
        f_hess = sum(1.0 * U * U * L for L in Ls)                         # (P_k Lambda_k^-1 P_k')^.2, upper triangle (:135-150)
        f_chol = float(U) ** 3 / 3                                        # Cones.jl:239-251
        f_trial = f_feas + f_grad + f_hess + f_chol                      # = 9.1e10 at U = 4845
        if args.config == "5d":
            f_uplhs = 1.0 * U * U * nm + float(nm) ** 2 * s.model.q + float(nm) ** 3 / 3    # triangular product U_H G (q = U), syrk, Cholesky
        else:
            f_uplhs = 2.0 * U * U + 1.0                                                     # n = 1: one inverse-Hessian product
        f_solve = 4.0 * s.model.q * max(nm, 1) + 2.0 * nm * nm + 4.0 * U * U
        trace_kernel = ("gemm_f64_kernel<true,2,0> (the cone's L x L x U and U x U x L Gram products), then potrf_tiles_kernel / bk_pivot_kernel "
                        "(profiles/r05_cfg%s_kernel_stats.csv)" % args.config)
        alg = (f_uplhs * n_upfact + trials * f_trial + solves * f_solve) / iters
        exe = (f_uplhs * n_upfact + trials * f_feas + n_grad * f_grad + n_hfact * (f_hess + f_chol) + solves * f_solve) / iters
    ms_it = loop_s / iters * 1e3
    ref_route = alg
    if args.config in ("3b", "3c"):
        # the reference route's per-trial explicit-Hessian Cholesky (dim^3 / 3) is work this path never does: counting it would put
        # the "fraction" above one; the line's frac is the EXECUTED work, the reference route's flops are reported beside it
        alg = exe
    achieved = alg / (ms_it * 1e-3) / 1e12
    executed = exe / (ms_it * 1e-3) / 1e12
    out = {
        "metric": "IPM iterations/sec (+ ms per KKT solve): " + work + " (Float64, QRCholDense + CombinedStepper)",
        "value": iters / loop_s, "unit": "iterations/s", "iterations_per_s": iters / loop_s, "n_gpus": 1, "steps": iters, "warmup": 2,
        "ms_per_step": ms_it, "higher_is_better": True, "scaling": ("strong" if comm is not None else None), "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": work, "n": int(s.model.n), "p": int(s.model.p), "q": int(s.model.q), "seed": args.seed, "solves_timed": nsolve_runs,
                   "final_status": status, "algorithm": algorithm_record()},
        "roofline": {"bound": "mfma", "kernel": "whole iteration; dominant kernels by the trace: " + trace_kernel,
                     "achieved": achieved, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP64_MFMA_PEAK_TFLOPS, "traffic": None,
                     "what": "reference-algorithm flops of one iteration (SURVEY 8(d): update_lhs + N_t trials + N_s solves, measured N_t and N_s) / "
                             "wall time per iteration; executed_* counts only the stages that actually ran (Hessians skipped on the proximity "
                             "lower bound are not counted)",
                     "flops_per_step": alg, "reference_route_flops_per_step": ref_route, "executed_flops_per_step": exe, "executed_achieved": executed,
                     "executed_frac": executed / FP64_MFMA_PEAK_TFLOPS,
                     "per_step": {"search_trials": trials / iters, "cone_gradients": n_grad / iters, "cone_hessian_factorizations": n_hfact / iters,
                                  "bunch_kaufman_factorizations": n_bk / iters, "schur_factorizations": n_upfact / iters}},
        "phases_ms_per_step": {k: v / iters * 1e3 for k, v in phases.items()},
        "kkt_solves_per_step": solves / iters, "ms_per_kkt_solve": phases["getdir"] / max(solves, 1) * 1e3,
        "search_trials_per_step": trials / iters, "setup_s": t_setup, "solve_plans": plan_stats(lib, ctx),
    }
    if comm is not None:
        # what an N-GPU run of ONE cone can and cannot divide: the Schur product of update_lhs is K-sharded over the
        # cone's U rows; everything else is replicated.  The Amdahl bound follows from this run's own phase times.
        out["n_gpus"] = world
        up_ms = up_s / iters * 1e3
        # the product's share of update_lhs: HIP-event time of the square-root-Hessian product + syrk at THIS N (hyp_get_kernel_stats [0], [1])
        shard_ms = min(up_ms, (ks[0] + ks[1]) / iters) if args.config == "5d" else 0.0
        out["config"]["parallelism"] = "k-shard x%d (model replicated, Schur product split over the cone's rows)" % world
        out["config"]["exchange"] = ("one all-reduce (sum, f64) of the n x n Schur upper triangle per iteration: %.1f calls and %.3g doubles per "
                                     "iteration" % (comm_calls / iters, comm_doubles / iters))
        out["k_shard"] = {
            "scales_with_n_gpus": "the Schur product of update_lhs (triangular product U_H G and G'(.), qrchol.jl:219-246): ~%.1f of %.1f ms per "
                                  "iteration at this N" % (shard_ms, ms_it),
            "replicated": "the cone's oracles (feasibility chains, gradient, U x U Hessian + Cholesky per accepted line-search trial, "
                          "wsosinterpnonnegative.jl:89-150, Cones.jl:239-251), the Schur Cholesky, the solves, the line search",
            "estimated_speedup_vs_1gpu_from_these_phases": (ms_it + shard_ms * (world - 1)) / ms_it if ms_it > 0 else None,
            "amdahl_bound_any_n": (ms_it + shard_ms * (world - 1)) / max(ms_it - shard_ms, 1e-9) if ms_it > 0 else None,
            "why_not_the_hessian": "block-columns of the U x U Hessian over N ranks save <= (1 - 1/N) 1.05 ms per trial and cost an all-gather of "
                                   "94 MB before the replicated Cholesky (DESIGN.md section 6)"}
    if hasattr(lib, "report"):   # HYP_PROFILE=1: wall time per C-ABI entry point
        print(lib.report(), file=sys.stderr)
    if rank == 0:
        emit_json_line(out)
    if comm is not None:
        try:
            comm.dist.destroy_process_group()
        except Exception:
            pass


def pin_near_gpu(local_rank, do_pin=True):
    """HYP_BENCH_PIN=1 (experiment, EXPERIMENTS.md r05-32): restrict this process to the host CPUs that sysfs lists as local to the GPU
    (`local_cpulist` of the local_rank-th AMD accelerator on the PCI bus) -- the launch-heavy configurations (config 5: ~1500 launches
    per iteration) depend on the host's distance to the device.  Returns what it did, for the bench line."""
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        buf = ctypes.create_string_buffer(64)
        if hip.hipDeviceGetPCIBusId(buf, 64, int(local_rank)) != 0:
            return {"pinned": False, "why": "hipDeviceGetPCIBusId failed"}
        bdf = buf.value.decode().lower()
        base = "/sys/bus/pci/devices/" + bdf
        if not os.path.exists(base + "/local_cpulist"):
            return {"pinned": False, "why": "no sysfs entry for " + bdf}
        txt = open(base + "/local_cpulist").read().strip()
        node = open(base + "/numa_node").read().strip() if os.path.exists(base + "/numa_node") else "?"
        cpus = set()
        for part in txt.split(","):
            if "-" in part:
                a, b = part.split("-"); cpus.update(range(int(a), int(b) + 1))
            elif part:
                cpus.add(int(part))
        before = len(os.sched_getaffinity(0))
        cpus &= os.sched_getaffinity(0)
        if not cpus:
            return {"pinned": False, "why": "empty local_cpulist", "device": os.path.basename(base)}
        try:
            here = int(open("/proc/self/stat").read().rsplit(")", 1)[1].split()[36])   # the CPU this thread last ran on
        except Exception:
            here = -1
        if do_pin == "far":   # (experiment: the CPUs of the OTHER node(s))
            far = os.sched_getaffinity(0) - cpus
            if far:
                os.sched_setaffinity(0, far)
        elif do_pin:
            os.sched_setaffinity(0, cpus)
        return {"pinned": do_pin if do_pin == "far" else bool(do_pin), "device": os.path.basename(base), "numa_node": node, "cpus": len(cpus), "cpus_before": before,
                "running_on_cpu": here, "cpu_is_local": here in cpus}
    except Exception as e:   # (sysfs layout differs: run unpinned)
        return {"pinned": False, "why": repr(e)}


def main():
    if os.environ.get("HYP_BENCH_PIN") in ("0", "1", "far"):   # 0: report the placement only, 1: pin, far: pin to the other node's CPUs
        mode = os.environ["HYP_BENCH_PIN"]
        print("[bench] host affinity:", pin_near_gpu(int(os.environ.get("LOCAL_RANK", "0")), "far" if mode == "far" else mode == "1"), file=sys.stderr)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed IPM iterations (default: 220 at config 2 = ~5 s of timed region; 30 at config 4)")
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", default=None, help="2 (headline, N = 1 default) | 4 (64 x PSD(80), strong scaling, N > 1 default) | 2w (one PSD block per rank) | 3b | 5p | 5d")
    ap.add_argument("--nvars", dest="n", type=int, default=5000)
    ap.add_argument("--psd-side", dest="side", type=int, default=200)
    ap.add_argument("--mc-side", dest="mc_side", type=int, default=200, help="--config 3c: side of the square matrix to complete (300: G = 29 GB)")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--cpu-iters", type=int, default=2, help="oracle iterations timed for cpu_baseline (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=8, help="host BLAS threads for the cpu_baseline leg and the host-side setup")
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="N > 1, config 4: skip the secondary record (config 2, K-panel shard)")
    ap.add_argument("--secondary-steps", type=int, default=100)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    multi = world > 1 or bool(os.environ.get("HYP_FORCE_DIST"))   # HYP_FORCE_DIST=1: exercise the RCCL path with a single rank
    if args.config is None:
        args.config = "4" if multi else "2"
    if args.config in ("3b", "3c", "5p", "5d"):
        if multi and args.config not in ("5p", "5d"):
            raise SystemExit("--config %s is a single-GPU line" % args.config)
        return main_other(args, world, rank, local_rank, multi)
    if args.config not in ("2", "4", "2w"):
        raise SystemExit("--config must be 2, 4, 2w, 3b, 3c, 5p or 5d")
    if args.steps is None:
        args.steps = 220 if args.config == "2" else 30
    if args.warmup is None:
        args.warmup = 5 if args.config == "2" else 2
    if multi and args.config != "2":
        return main_multi(args, world, rank, local_rank)
    out = run_headline(args, world, rank, local_rank, multi)
    if not multi:
        emit_json_line(out)



if __name__ == "__main__":
    main()
