#!/usr/bin/env python3
"""bench.py -- IPM iterations/sec of the MI355X hot path on BASELINE.json's headline configuration.

A "step" is ONE interior-point iteration of Hypatia's CombinedStepper (update_lhs: sqrt-Hessian
products + Schur syrk + Cholesky; four direction solves with refinement; line search) on the synthetic
dense instance of configs[1]: a single PosSemidefTri cone of side 200 (q = 20100) with a dense random
G (q x n, n = 5000), p = 0.  Inputs are resident in HBM when the timed region starts.  If the solver
converges inside the timed region it is put back at its initial iterate and keeps stepping.

    python bench.py --gpus N --steps K --warmup W
prints ONE JSON line (rank 0): metric/value/unit + roofline (dominant kernel = the FP64-MFMA Schur
syrk, timed with HIP events on the library stream) + cpu_baseline (the numpy/scipy restatement in
oracle/, timed on the host cores for a bounded number of iterations of the same instance).

N > 1: one process per GPU (torch.distributed / RCCL).  The cone products of the workload are sharded
one PosSemidefTri(side 200) block per rank ("weak" scaling: per-GPU work fixed); every iteration's
partial Schur matrices are summed with one all-reduce.  value = blocks processed per second over all
ranks = N * iterations/sec.
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

# one metric for every N (weak scaling: one PosSemidefTri block per GPU, n shared).  At N = 1 a block-iteration IS an
# IPM iteration of BASELINE.json's headline instance (configs[1]).
METRIC = "IPM block-iterations/sec: dense n=%d, one PosSemidefTri(%d) block per GPU (Float64, QRCholDense + CombinedStepper)"
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s achievable
FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X FP64 matrix spec; measured 77.8 with tools/probe_mfma.hip (profiles/)


def pmc_traffic(n, q):
    """HBM bytes per syrk launch from the committed PMC passes (FETCH_SIZE x 2 gfx950 correction + WRITE_SIZE); only
    valid for the configuration they were collected on, else None."""
    try:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_summary.json")
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


def main_multi(args, world, rank, local_rank):
    """N > 1: one PosSemidefTri(side) block per rank (weak scaling), Schur matrices summed by all-reduce."""
    import torch                      # before the HIP library: one HIP runtime per process (torch's)
    import torch.distributed as dist
    backend = os.environ.get("HYP_DIST_BACKEND", "nccl")
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank % torch.cuda.device_count())
    dist.init_process_group(backend=backend)
    import hypatia_jl_amd as H
    from hypatia_jl_amd import distributed as D
    comm = D.Comm(device="cuda")
    n, side = args.n, args.side
    dim = side * (side + 1) // 2
    q = dim * world
    # this rank's block of the instance (same generator as configs[1], one seed per block)
    rng = np.random.default_rng(args.seed + 1000 * rank)
    G_r = np.asfortranarray(rng.standard_normal((dim, n)) / np.sqrt(n))
    x0 = np.random.default_rng(args.seed).standard_normal(n)
    e_r = np.zeros(dim)
    k = 0
    for i in range(1, side + 1):
        e_r[k] = 1.0
        k += i + 1
    h = np.zeros(q)
    h[rank * dim:(rank + 1) * dim] = G_r @ x0 + e_r
    comm.allreduce(h)
    c = -(G_r.T @ e_r)
    comm.allreduce(c)
    owners = list(range(world))
    cones = [D.ShardedCone(comm, r, H.PosSemidefTri(dim) if r == rank else None, dim, side) for r in range(world)]
    model = D.DistModel(comm, c, h, G_r, cones, owners)
    t_setup = time.perf_counter()
    from threadpoolctl import threadpool_limits
    host_threads = max(1, min(args.cpu_threads, (os.cpu_count() or 8) // world))   # untimed host set-up: the ranks share the host cores
    with threadpool_limits(limits=host_threads, user_api="blas"):
        solver = H.Solver(verbose=args.verbose and rank == 0, syssolver=D.DistQRCholDenseSystemSolver(comm))
        solver.load(model)
        solver.setup()
    t_setup = time.perf_counter() - t_setup
    lib, ctx = H._lib.lib(), H._lib.ctx()

    def step():
        if not solver.iterate():
            solver.reset_iterate()

    from hypatia_jl_amd.solvers import _blas_limit
    blas_cap = _blas_limit()     # (iterate() alone re-enters the host BLAS cap per call; hold it across the run)
    blas_cap.__enter__()
    for _ in range(args.warmup):
        step()
    lib.hyp_reset_timers(ctx)
    n_solves0 = solver.n_solves
    for f in ("upsys", "upfact", "uprhs", "getdir", "search"):
        setattr(solver, "time_" + f, 0.0)
    comm.barrier()
    torch.cuda.synchronize()
    n_coll0 = comm.n_collectives          # (setup -- the LSQR initial point, rescaling -- and warmup are not counted)
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
    if rank == 0:
        syrk_ms = ks[1] / max(ks[4], 1)
        syrk_flops = float(n) * n * dim
        achieved = syrk_flops / (syrk_ms * 1e-3) / 1e12 if syrk_ms > 0 else 0.0
        out = {
            "metric": METRIC % (n, side),
            "value": world * args.steps / elapsed,
            "unit": "block-iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "iterations_per_s": args.steps / elapsed,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "configs[1] block per GPU: %d x PosSemidefTri side=%d (q=%d total), dense random G, n=%d, p=0; "
                                   "cones sharded one per rank, Schur all-reduce (sum, f64, n x n) per iteration" % (world, side, q, n),
                       "n": n, "q": q, "seed": args.seed, "parallelism": "cone-shard x%d" % world},
            "roofline": {"bound": "mfma", "kernel": "gemm_f64_kernel<true, 4, 1> + splitk_reduce (per-rank Schur syrk, upper)", "achieved": achieved,
                         "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP64_MFMA_PEAK_TFLOPS, "traffic": None,
                         "launch_ms": syrk_ms, "flops_per_launch": syrk_flops},
            "kkt_solves_per_step": (solver.n_solves - n_solves0) / args.steps,
            "collectives_per_step": n_coll / max(args.steps, 1),
            "setup_s": t_setup,
        }
        if coll_hist is not None:   # HYP_PROFILE=1: where the collectives of the timed region come from
            for key, cnt in sorted(coll_hist.items(), key=lambda kv: -kv[1]):
                print("collectives %-40s %8d doubles op %-5s : %.1f per step" % (key[0], key[1], key[2], cnt / max(args.steps, 1)), file=sys.stderr)
        print(json.dumps(out))
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--nvars", dest="n", type=int, default=5000)
    ap.add_argument("--psd-side", dest="side", type=int, default=200)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--cpu-iters", type=int, default=2, help="oracle iterations timed for cpu_baseline (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=8, help="host BLAS threads for the cpu_baseline leg and the host-side setup")
    ap.add_argument("--verbose", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or os.environ.get("HYP_FORCE_DIST"):   # HYP_FORCE_DIST=1: exercise the RCCL path with a single rank
        return main_multi(args, world, rank, local_rank)

    import hypatia_jl_amd as H

    inst = gen_instance(args.n, [args.side], args.seed)
    q = inst[3].shape[0]
    t_setup = time.perf_counter()
    from threadpoolctl import threadpool_limits
    with threadpool_limits(limits=args.cpu_threads, user_api="blas"):   # host preprocessing (rescale, QR for the initial x): untimed setup
        solver = H.Solver(verbose=args.verbose)
        solver.load(H.make_model(inst))
        solver.setup()
    t_setup = time.perf_counter() - t_setup
    lib, ctx = H._lib.lib(), H._lib.ctx()

    def step():
        if not solver.iterate():
            solver.reset_iterate()

    from hypatia_jl_amd.solvers import _blas_limit
    blas_cap = _blas_limit()     # (iterate() alone re-enters the host BLAS cap per call; hold it across the run)
    blas_cap.__enter__()
    for _ in range(args.warmup):
        step()
    lib.hyp_reset_timers(ctx)
    if hasattr(lib, "reset"):
        lib.reset()
    n_solves0 = solver.n_solves
    n_trials0 = solver.stepper.searcher.n_trials
    for f in ("upsys", "upfact", "uprhs", "getdir", "search"):
        setattr(solver, "time_" + f, 0.0)
    lib.hyp_ctx_synchronize(ctx)      # (every C-ABI call is synchronous; this is the device-wide fence)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    lib.hyp_ctx_synchronize(ctx)
    elapsed = time.perf_counter() - t0
    blas_cap.__exit__(None, None, None)   # (the CPU baseline below runs with the full host BLAS pool)

    ks = (ctypes.c_double * 8)()
    lib.hyp_get_kernel_stats(ctx, ks)
    syrk_ms = ks[1] / max(ks[4], 1)
    nmp = solver.model.n - solver.model.p
    syrk_flops = float(nmp) * nmp * q                     # algorithmic: n^2 q
    achieved = syrk_flops / (syrk_ms * 1e-3) / 1e12 if syrk_ms > 0 else 0.0
    ms_per_step = elapsed / args.steps * 1e3
    n_solves = solver.n_solves - n_solves0
    n_trials = solver.stepper.searcher.n_trials - n_trials0

    out = {
        "metric": METRIC % (args.n, args.side),
        "value": args.steps / elapsed,          # one block on one GPU: block-iterations/s = IPM iterations/s
        "unit": "block-iterations/s",
        "iterations_per_s": args.steps / elapsed,
        "n_gpus": 1,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "configs[1]: single PosSemidefTri side=%d (q=%d), dense random G q x n, n=%d, p=0" % (args.side, q, args.n),
                   "n": args.n, "q": q, "seed": args.seed},
        "roofline": {"bound": "mfma", "kernel": "gemm_f64_kernel<true, 4, 1> (Schur syrk, upper)", "achieved": achieved,
                     "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP64_MFMA_PEAK_TFLOPS,
                     "traffic": pmc_traffic(args.n, q), "traffic_unit": "bytes of HBM traffic per launch (rocprofv3 PMC, profiles/r01_pmc_summary.json)",
                     "launch_ms": syrk_ms, "flops_per_launch": syrk_flops},
        "phases_ms_per_step": {"sqrt_hess_prod": ks[0] / args.steps, "syrk": ks[1] / args.steps, "cholesky": ks[2] / args.steps,
                               "update_lhs": solver.time_upsys / args.steps * 1e3, "get_directions": solver.time_getdir / args.steps * 1e3,
                               "update_rhs": solver.time_uprhs / args.steps * 1e3, "search": solver.time_search / args.steps * 1e3},
        "kkt_solves_per_step": n_solves / args.steps,
        "ms_per_kkt_solve": solver.time_getdir / max(n_solves, 1) * 1e3,
        "search_trials_per_step": n_trials / args.steps,
        "setup_s": t_setup,
        # the reference's ten timers (Solvers.jl:86-96): set-up ones for the whole solve set-up, the others per timed step
        "hypatia_timers_s": {"rescale": getattr(solver, "time_rescale", 0.0), "initx": getattr(solver, "time_initx", 0.0),
                             "inity": getattr(solver, "time_inity", 0.0), "unproc": getattr(solver, "time_unproc", 0.0),
                             "loadsys": getattr(solver, "time_loadsys", 0.0),
                             "upsys_per_step": solver.time_upsys / args.steps, "upfact_per_step": solver.time_upfact / args.steps,
                             "uprhs_per_step": solver.time_uprhs / args.steps, "getdir_per_step": solver.time_getdir / args.steps,
                             "search_per_step": solver.time_search / args.steps},
    }

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
        with threadpool_limits(limits=args.cpu_threads, user_api="blas"):
            os_ = OSolver(verbose=False, iter_limit=args.cpu_iters)
            os_.load(omodel(inst))
            os_.solve()
        cpu_it_s = os_.num_iters / os_.iter_time if os_.iter_time > 0 else 0.0
        out["cpu_baseline"] = {"value": cpu_it_s, "unit": "block-iterations/s", "cores": args.cpu_threads, "host_cores": os.cpu_count(), "kind": "port",
                               "sample": "first %d IPM iterations of the same instance; numpy/scipy restatement: OpenBLAS (%d threads) for the syrk / "
                                         "Cholesky / gemv, the per-column dtrsm loop of the PSD products is sequential as in the reference "
                                         "(possemideftri.jl:168-174) and Python-bound here" % (os_.num_iters, args.cpu_threads),
                               "s_per_iteration": (os_.iter_time / max(os_.num_iters, 1))}
        out["speedup_vs_cpu_port"] = out["value"] / cpu_it_s if cpu_it_s > 0 else None
    if hasattr(lib, "report"):
        print(lib.report(), file=sys.stderr)
        print("host wall in timed region: %.1f ms/step" % ms_per_step, file=sys.stderr)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
