"""Iterate-level (trajectory) comparison of the HIP path with the CPU oracle for models over ANY of the device cones (test
helper).  A trajectory = one row per iterate: primal / dual objective, gap, x / z feasibility residuals, tau, kappa, mu and the
line-search step size that led to the iterate (the quantities of `print_iteration`, src/Solvers/Solvers.jl:603-619, plus
`stepper.prev_alpha`, steppers/combined.jl:100-118).

The library reads its route switches (`HYP_ENS_CLOSED_INV`, `HYP_PROX_LB`, ...) once per process, so the HIP trajectory of a
given route is produced by a child process (`python tests/trajectory_harness.py NAME`, JSON on stdout)."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

COLS = ("p_obj", "d_obj", "gap", "x_feas", "z_feas", "tau", "kap", "mu", "alpha")

# route = the switches that take the device path off the reference's order of operations
REFERENCE_ROUTE = {"HYP_ENS_CLOSED_INV": "0", "HYP_PROX_LB": "0", "HYP_ENS_PREFETCH": "0", "HYP_WSOS_PAR": "0", "HYP_BK_HYBRID": "0",
                   "HYP_ENS_DUAL_DECIDE": "0", "HYP_ENS_FUSED": "0"}
DEFAULT_ROUTE = {}


def mixed_instance(specs, seed, n=None, p=0):
    """strictly feasible model over the given cone specs (the construction of tests/fuzz_models.py with fixed specs): s0 = the
    cones' initial points, z0 = -grad there, h = G x0 + s0, c = -G'z0 - A'y0, b = A x0"""
    from oracle.build import make_cone
    rng = np.random.default_rng(seed)
    s0, z0 = [], []
    for sp in specs:
        cone = make_cone(sp)
        cone.setup_data()
        cone.reset_data()
        pt = np.zeros(cone.dimension())
        cone.set_initial_point(pt)
        cone.load_point(pt)
        assert cone.is_feas()
        g = -np.array(cone.get_grad())
        if cone.use_dual_barrier:
            pt, g = g, pt
        s0.append(pt)
        z0.append(g)
    s0, z0 = np.concatenate(s0), np.concatenate(z0)
    q = len(s0)
    n = n or max(2, q // 3)
    G = rng.standard_normal((q, n)) / np.sqrt(n)
    A = rng.standard_normal((p, n))
    x0, y0 = rng.standard_normal(n), rng.standard_normal(p)
    return (-(G.T @ z0) - A.T @ y0, A, A @ x0, G, G @ x0 + s0, specs, dict(status="Optimal"))


def _wsos_spec(nvars, halfdeg, use_dual):
    from oracle import polyutils as pu
    U, _, Ps = pu.interpolate_box([-1.0] * nvars, [1.0] * nvars, halfdeg, sample=False)
    return ("wsosinterpnonnegative", U, Ps, use_dual)


def _golden_keep(name, force=False):
    """the interpolation-point choice a committed trajectory was computed with (None: not recorded / not needed)"""
    if os.environ.get("HYP_GOLDEN_REGEN") and not force:
        return None
    for fn in ("trajectory_wsos.json", "trajectory_fullsize.json"):
        f = os.path.join(ROOT, "tests", "golden", fn)
        if os.path.exists(f):
            rec = json.load(open(f))["cases"].get(name)
            if rec is not None:
                return rec.get("interp_keep")
    return None


def fullsize_instance(name):
    """the BASELINE.json configurations AS BENCHMARKED (bench.py's generators) and mid sizes on the same generator:
      psdfull_<n>_<side>x<count>_<seed>   bench.gen_instance(n, [side] * count, seed): config 2 is psdfull_5000_200x1_1
      cfg4_<n>_<side>x<count>_<seed>      bench.gen_block per cone (config 4 = cfg4_5000_80x64_1: G = 8.3 GB)
      cfg5p_<seed> / cfg5d_<seed>         bench.gen_polymin5 (U = 4845), primal / dual form (cfg5pw_ / cfg5dw_: the same, whole solves)"""
    import bench
    kind, rest = name.split("_", 1)
    if kind == "psdfull":
        n, sc, seed = rest.split("_")
        side, count = sc.split("x")
        return bench.gen_instance(int(n), [int(side)] * int(count), int(seed))
    if kind == "cfg4":
        n, sc, seed = rest.split("_")
        n, seed = int(n), int(seed)
        side, count = (int(v) for v in sc.split("x"))
        dim = side * (side + 1) // 2
        x0 = np.random.default_rng(seed).standard_normal(n)
        e_k = bench.svec_identity(side)
        G = np.empty((dim * count, n), order="F")
        h = np.zeros(dim * count)
        c = np.zeros(n)
        for k in range(count):
            G_k = bench.gen_block(n, side, k, seed)
            G[k * dim:(k + 1) * dim] = G_k
            h[k * dim:(k + 1) * dim] = G_k @ x0 + e_k
            c -= G_k.T @ e_k
        return (c, np.zeros((0, n)), np.zeros(0), G, h, [("possemideftri", dim)] * count, dict(status="Optimal"))
    if kind in ("cfg5p", "cfg5d"):
        return bench.gen_polymin5(kind == "cfg5p", int(rest), keep=_golden_keep(name))[0]
    if kind in ("cfg5pw", "cfg5dw"):   # the SAME instances as cfg5p_<seed> / cfg5d_<seed> (their interpolation points), solved to the end
        return bench.gen_polymin5(kind == "cfg5pw", int(rest), keep=_golden_keep(kind[:-1] + "_" + rest, force=True))[0]
    raise KeyError(name)


def instance(name):
    """named instances of the trajectory tests: matrix completion (examples/matrixcompletion/native.jl:23-70), polymin in both
    forms (examples/polymin/native.jl:56-90), the reference's own EpiNormSpectral / WSOS known-answer instances, mixed models"""
    from oracle import instances as I
    if name.split("_")[0] in ("psdfull", "cfg4", "cfg5p", "cfg5d", "cfg5pw", "cfg5dw"):
        return fullsize_instance(name)
    if name.startswith("mc_"):                      # mc_<d1>x<d2>_<seed>
        dd, seed = name[3:].split("_")
        d1, d2 = dd.split("x")
        return I.matrixcompletion(int(d1), int(d2), seed=int(seed))
    if name.startswith("polymin_"):                 # polymin_<nvars>_<halfdeg>_<p|d>_<seed>
        nv, hd, form, seed = name[8:].split("_")
        return I.polymin(int(nv), int(hd), form == "p", seed=int(seed), keep=_golden_keep(name))
    if name == "mixed_psd_ens_wsos":
        return mixed_instance([("possemideftri", 36), ("epinormspectral", 4, 6, False), _wsos_spec(2, 2, False), ("nonnegative", 3)], seed=11)
    if name == "mixed_dual_barriers":
        return mixed_instance([("epinormspectral", 3, 5, True), ("possemideftri", 21), _wsos_spec(2, 2, True), ("epinormspectral", 2, 2, False)],
                              seed=12, p=2)
    if name == "mixed_two_wsos_two_ens":
        return mixed_instance([_wsos_spec(1, 4, False), ("epinormspectral", 5, 5, False), _wsos_spec(2, 3, True), ("epinormspectral", 2, 9, True),
                               ("possemideftri", 10)], seed=13, p=1)
    return I.KNOWN_ANSWER[name]()


def perturbed(inst, seed=99):
    """the same model with G and h moved by one ulp per entry: what the ORACLE does with it measures how far rounding alone
    carries a trajectory (zeros stay zeros, so structured models keep their structure)"""
    rng = np.random.default_rng(seed)
    eps = np.finfo(float).eps
    G2 = inst[3] * (1.0 + eps * rng.choice([-1.0, 1.0], size=inst[3].shape))
    h2 = inst[4] * (1.0 + eps * rng.choice([-1.0, 1.0], size=inst[4].shape))
    return inst[:3] + (G2, h2) + inst[5:]


PROBE_ROWS, PROBE_VECS, PROBE_SEED = 64, 4, 2024


def schur_probe(S_upper, seed=PROBE_SEED):
    """what the full-size fixtures keep of an nm x nm Schur matrix (qrchol.jl:201-257; 200 MB at n = 5000): a seeded 64 x 64
    sub-block, the diagonal, S V for four seeded probe vectors and the Frobenius norm.  S_upper: its upper triangle (syrk 'U')."""
    S = np.triu(S_upper)
    S = S + np.triu(S, 1).T
    nm = S.shape[0]
    rng = np.random.default_rng(seed)
    rows = np.sort(rng.choice(nm, size=min(PROBE_ROWS, nm), replace=False))
    cols = np.sort(rng.choice(nm, size=min(PROBE_ROWS, nm), replace=False))
    V = rng.standard_normal((nm, PROBE_VECS))
    return dict(sub=S[np.ix_(rows, cols)].copy(), diag=np.diag(S).copy(), SV=S @ V, fro=np.array([np.linalg.norm(S)]))


def run_trajectory(solver_cls, model, gate_log=None, probe_iters=(), probes=None, **opts):
    """probe_iters: iteration numbers at whose top the Schur matrix of the PREVIOUS update_lhs is probed into `probes`
    (1 = the matrix assembled at the initial iterate)"""
    rows = []
    s = solver_cls(**opts)
    if gate_log is not None:
        s.gate_log = gate_log

    def cb(sv):
        rows.append((sv.primal_obj, sv.dual_obj, sv.gap, sv.x_feas, sv.z_feas, sv.point.tau, sv.point.kap, sv.mu,
                     getattr(sv.stepper, "prev_alpha", 1.0)))
        if sv.num_iters in probe_iters:
            ss = sv.syssolver
            probes[sv.num_iters] = schur_probe(ss.get_lhs() if hasattr(ss, "get_lhs") else ss.lhs_sub)
    s.iter_callback = cb
    s.load(model)
    s.solve()
    return s, np.array(rows)


GATE = 1e-4          # steppers/common.jl:47, 105: the third-order term of a cone is used iff its dder3_viol < 1e-4
GATE_DECADES = 2.0   # a value within two decades of the threshold does not determine the decision (see gate_margins)


def gate_margins(gate_log, n_rows):
    """per iterate i: the smallest distance, in decades, of a cone's dder3_viol from the threshold over the two adjustment
    right-hand sides built AT iterate i (inf: no cone with a third-order oracle).  dder3_viol = |<dder3, point> - <dir, H dir>| /
    (sqrt(eps) + |<dir, H dir>|) is zero in exact arithmetic: what the reference thresholds is the rounding error of the cone's
    third-order oracle, so two correct implementations differ in it by factors of 3 to 20 (tools/diag_pair.py prints both sides)
    and whichever side of 1e-4 a value near 1e-4 falls on is not a property of the algorithm."""
    m = np.full(n_rows, np.inf)
    for it, _, _, v in gate_log:
        if it < n_rows:
            m[it] = min(m[it], abs(np.log10(max(v, 1e-300) / GATE)))
    return m


def oracle_trajectory(inst, probe_iters=(), **opts):
    from oracle.build import make_model as omodel
    from oracle.solvers import Solver as OSolver
    log, probes = [], {}
    s, t = run_trajectory(OSolver, omodel(inst), gate_log=log, probe_iters=probe_iters, probes=probes, **opts)
    return dict(status=s.status, iters=s.num_iters, p_obj=s.primal_obj, rows=t, gate=gate_margins(log, len(t)), probes=probes, x=s.get_x())


def hip_trajectory(name, route, timeout=1800, **opts):
    """HIP trajectory of a named instance under a route (env switches), in a process of its own"""
    env = dict(os.environ)
    for k in REFERENCE_ROUTE:
        env.pop(k, None)
    env.update(route)
    holds = any(getattr(m, "_ctx", None) is not None for n, m in list(sys.modules.items()) if n.endswith("_lib") and hasattr(m, "load_library"))
    if holds:   # (this process holds a context and with it possibly the device's persistent-kernel lock; it launches nothing meanwhile.
        env.setdefault("HYP_PERSISTENT", "1")   #  Without one, the child takes the lock the normal way: ADVICE r05)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), name, json.dumps(opts)], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    d["rows"] = np.array(d["rows"])
    d["probes"] = {int(k): {f: np.array(v) for f, v in p.items()} for k, p in d.get("probes", {}).items()}
    return d


def stable_prefix(O, Ps, gate=None, mu_alpha=1e-7):
    """number of leading iterates on which a comparison of step sizes means something: the oracle reproduces them on every
    1-ulp-perturbed copy of the model (Ps: list of trajectories), mu >= mu_alpha, and no acceptance test of a third-order term
    was decided within GATE_DECADES of its threshold at an EARLIER iterate (row i+1 is the step built at iterate i)"""
    if isinstance(Ps, np.ndarray):
        Ps = [Ps]
    kp = len(O)
    for P in Ps:
        k = min(len(O), len(P))
        stable = P[:k, 8] == O[:k, 8]
        kp = min(kp, k if stable.all() else int(np.argmin(stable)))
    kp = min(kp, int(np.sum(O[:kp, 7] >= mu_alpha)) + 1)
    if gate is not None:
        marginal = np.nonzero(np.asarray(gate)[:kp] < GATE_DECADES)[0]
        if marginal.size:
            kp = min(kp, int(marginal[0]) + 1)
    return kp


def compare(ht, ot, pt, mu_alpha=1e-7, mu_tight=1e-3, tight=1e-10, label=""):
    """the parity bar of tests/test_hip_solver.py::test_trajectory_parity_psd for any model.  ht / ot / pt = HIP, oracle,
    perturbed-oracle trajectories.  Returns a report dict; raises AssertionError with the first diverging iteration otherwise."""
    H, O = ht["rows"], ot["rows"]
    Ps = pt["rows"] if isinstance(pt["rows"], list) else [pt["rows"]]
    P = Ps[0]
    assert ht["status"] == ot["status"], (label, ht["status"], ot["status"])
    k = min(len(H), len(O), min(len(p_) for p_ in Ps))
    same = H[:k, 8] == O[:k, 8]
    kp = min(k, stable_prefix(O, Ps, ot.get("gate"), mu_alpha))
    if not same[:kp].all():
        i = int(np.argmin(same[:kp]))
        raise AssertionError("%s: step size differs at iterate %d where the oracle is stable: HIP alpha %.6g, oracle %.6g (mu %.3e); "
                             "alphas HIP %s / oracle %s" % (label, i, H[i, 8], O[i, 8], O[i, 7], H[:k, 8].tolist(), O[:k, 8].tolist()))
    worst = {}
    margin = {}   # per column: (largest deviation / its bar while mu >= mu_tight, largest deviation / 100x-sensitivity limit on the prefix): < 1 passes
    for col, cname in ((0, "p_obj"), (1, "d_obj"), (7, "mu"), (5, "tau"), (3, "x_feas"), (4, "z_feas")):
        # residual norms sit at rounding level once the iterate is feasible: compare them relative to the model's scale 1
        scale = np.abs(O[:kp, col]) + (1e-300 if col in (0, 1, 5, 7) else 1e-6)
        dev = np.abs(H[:kp, col] - O[:kp, col]) / scale
        floor = np.max([np.abs(p_[:kp, col] - O[:kp, col]) / scale for p_ in Ps], axis=0)
        well = O[:kp, 7] >= mu_tight
        if well.any():
            # 1e-10 while mu >= 1e-3 -- unless the ORACLE's own rows move by more than a third of that under 1-ulp perturbations of G and
            # h at that iterate (config 5 in its dual form at U = 4845: 4.4e-10 in p_obj at mu = 1.4e-3, tools/diag_traj.py): no
            # implementation can agree with one oracle run more closely than oracle runs agree with each other, the bar is then 3x that
            bar = np.maximum(tight, 3 * np.maximum.accumulate(floor))
            over = well & (dev >= bar)
            assert not over.any(), "%s: %s differs by %.3e (relative; bar %.3e) at iterate %d while mu >= %g" % (
                label, cname, dev[over].max(), bar[over][int(np.argmax(dev[over]))], int(np.nonzero(over)[0][0]), mu_tight)
        lim = 100 * np.maximum.accumulate(np.maximum(floor, 1e-13))
        assert np.all(dev <= lim), "%s: %s beyond 100x the oracle's own 1-ulp sensitivity: %s vs %s" % (label, cname, dev, lim)
        worst[cname] = float(dev[well].max()) if well.any() else 0.0
        margin[cname] = (float((dev[well] / bar[well]).max()) if well.any() else 0.0, float((dev / lim).max()) if len(dev) else 0.0,
                         float(np.maximum.accumulate(floor)[well].max()) if well.any() else 0.0)
    return dict(prefix=kp, iters_hip=len(H) - 1, iters_oracle=len(O) - 1, worst=worst, margin=margin)


if __name__ == "__main__":
    import hypatia_jl_amd as Hm
    nm = sys.argv[1]
    opts = json.loads(sys.argv[2]) if len(sys.argv) > 2 else {}
    probes = {}
    s, t = run_trajectory(Hm.Solver, Hm.make_model(instance(nm)), probe_iters=tuple(opts.pop("probe_iters", ())), probes=probes, **opts)
    screens = list(s.syssolver.search_screen_stats()) if hasattr(s.syssolver, "search_screen_stats") else [0, 0]
    import ctypes
    bk = (ctypes.c_longlong * 3)()
    Hm._lib.check(Hm._lib.lib().hyp_ctx_bk_stats(Hm._lib.ctx(), bk), "bk_stats")
    print(json.dumps(dict(status=s.status, iters=s.num_iters, p_obj=s.primal_obj, rows=t.tolist(), screens=screens,
                          trials=int(s.stepper.searcher.n_trials), bk_stats=[int(v) for v in bk],
                          probes={str(k): {f: v.tolist() for f, v in p.items()} for k, p in probes.items()})))
