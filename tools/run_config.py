"""Run one of BASELINE.json's configurations at FULL size through the HIP path and report the iteration
table, per-phase times and the conic certificate residuals of reference test/nativeinstances.jl:58-65.
    python tools/run_config.py --config 2|3b|4|5p|5d [--iters K]
config 4 and 5 use the reference's `init_use_indirect` initial point (LSQR) instead of the pivoted QR of
G, which alone would take minutes on the host at these sizes."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="4")
ap.add_argument("--iters", type=int, default=1000)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--sample-factor", type=int, default=2)
ap.add_argument("--verbose", action="store_true")
args = ap.parse_args()

from oracle import instances as I     # instance generators only (data), the solve below is the HIP path
import hypatia_jl_amd as H

t0 = time.perf_counter()
indirect = False
if args.config == "2":
    inst = I.psd_blocks(5000, [200], seed=args.seed)
elif args.config == "4":
    inst = I.psd_blocks(5000, [80] * 64, seed=args.seed); indirect = True
elif args.config == "3b":
    inst = I.matrixcompletion(50, 100, seed=args.seed)
elif args.config in ("5p", "5d"):
    rng = np.random.default_rng(args.seed)
    from oracle import polyutils as pu
    U, pts, Ps = pu.interpolate_box([-1.0] * 4, [1.0] * 4, 8, rng=rng, sample_factor=args.sample_factor)
    # a genuine polynomial of degree 4 sampled at the interpolation points (random VALUES at the points would
    # define a wildly oscillating degree-16 interpolant: objective ~1e5 and a numerically hopeless instance)
    a = rng.uniform(-0.5, 0.5, 4)
    vals = np.sum((pts - a) ** 2, axis=1) + (pts[:, 0] * pts[:, 1] - pts[:, 2] * pts[:, 3]) ** 2 + 0.3 * pts[:, 0] * pts[:, 2]
    if args.config == "5p":
        inst = (np.array([-1.0]), np.zeros((0, 1)), np.zeros(0), np.ones((U, 1)), vals, [("wsosinterpnonnegative", U, Ps, False)], {})
    else:
        inst = (vals, np.ones((1, U)), np.array([1.0]), -np.eye(U), np.zeros(U), [("wsosinterpnonnegative", U, Ps, True)], {})
else:
    raise SystemExit("unknown config")
c, A, b, G, h = inst[:5]
print("instance built in %.1f s: n=%d p=%d q=%d cones=%d" % (time.perf_counter() - t0, len(c), len(b), len(h), len(inst[5])), flush=True)

t0 = time.perf_counter()
solver = H.Solver(verbose=args.verbose, iter_limit=args.iters, init_use_indirect=indirect)
solver.load(H.make_model(inst))
solver.solve()
wall = time.perf_counter() - t0
x, y, z, s = solver.get_x(), solver.get_y(), solver.get_z(), solver.get_s()
def relres(a, bb):
    return float(np.linalg.norm(a - bb) / (1 + np.linalg.norm(bb)))
rep = {
    "config": args.config, "status": solver.status, "iters": solver.num_iters, "wall_s": wall,
    "iter_loop_s": solver.iter_time, "ms_per_iter": solver.iter_time / max(solver.num_iters, 1) * 1e3,
    "p_obj": solver.primal_obj, "d_obj": solver.dual_obj,
    "res_Ax_b": relres(A @ x, b) if len(b) else 0.0, "res_Gx_s_h": relres(G @ x + s, h), "res_Gtz_Aty_c": relres(G.T @ z + (A.T @ y if len(b) else 0), -c),
    "s_dot_z": float(s @ z),
    "phases_s": {k: getattr(solver, "time_" + k) for k in ("rescale", "initx", "inity", "loadsys", "upsys", "upfact", "uprhs", "getdir", "search")},
    "kkt_solves": solver.n_solves, "search_trials": solver.stepper.searcher.n_trials,
}
print(json.dumps(rep))
