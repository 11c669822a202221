"""GPU parity of the system solver and of whole solves against the CPU oracle, plus the reference's
known-answer instances (test/nativeinstances.jl) solved through the HIP path."""
import numpy as np
import pytest

from instance_harness import build_solve_check

pytestmark = pytest.mark.gpu

HIP_KINDS = ("nonnegative", "possemideftri", "epinormspectral", "wsosinterpnonnegative", "linmatrixineq", "doublynonnegativetri", "hyporootdettri", "hypoperlogdettri", "wsosinterppossemideftri")


def _hip_ok(inst):
    return all(s[0] in HIP_KINDS for s in inst[5])


def _instances():
    from oracle import instances as I
    return {k: v for k, v in I.KNOWN_ANSWER.items() if _hip_ok(v())}


def _all_names():
    from oracle import instances as I
    return sorted(I.KNOWN_ANSWER)


@pytest.mark.parametrize("name", _all_names())
@pytest.mark.parametrize("reduce", [True, False])
def test_known_answer_hip(name, reduce):
    import hypatia_jl_amd as H
    from oracle import instances as I
    inst = I.KNOWN_ANSWER[name]()
    solver = H.Solver(default_tol_relax=10, reduce=reduce)
    build_solve_check(solver, H.make_model(inst), inst)


def _more_names():
    from oracle import instances as I
    return sorted(I.MORE_NATIVE)


@pytest.mark.parametrize("name", _more_names())
def test_more_native_instances_hip(name):
    """further native instances of the reference through the HIP path (oracle/instances.py: MORE_NATIVE): dependent
    equalities / dependent columns (consistent1 Optimal, inconsistent1 PrimalInconsistent, inconsistent2 DualInconsistent:
    process.jl:64-365), an objective offset, the LSQR initial point without preprocessing (indirect1, on the device SymIndef
    solver), the rank-deficient hypograph instances at their own tolerance, a model with no variables at all"""
    import hypatia_jl_amd as H
    from oracle import instances as I
    inst = I.MORE_NATIVE[name]()
    opts = dict(default_tol_relax=10)
    opts.update(inst[6].get("solver_opts", {}))
    if opts.get("syssolver") == "symindef":
        opts["syssolver"] = H.SymIndefDenseSystemSolver()
    build_solve_check(H.Solver(**opts), H.make_model(inst), inst)


@pytest.mark.parametrize("seed", list(range(40)))
def test_random_mixed_cone_models_hip(seed):
    """random strictly feasible models over a random mix of the device cones (tests/fuzz_models.py): one to four cones of random
    kinds, sizes from dimension one, dual-barrier variants, zero to n - 1 equalities, also n > q.  The HIP solve must be Optimal with
    the full certificate, and agree with the oracle's optimum"""
    import hypatia_jl_amd as H
    from fuzz_models import random_model
    from oracle.build import make_cone as omake, make_model as omodel
    from oracle.solvers import Solver as OSolver
    inst = random_model(seed, omake)
    s = build_solve_check(H.Solver(default_tol_relax=10), H.make_model(inst), inst)
    o = OSolver(default_tol_relax=10)
    o.load(omodel(inst))
    o.solve()
    assert o.get_status() == "Optimal"
    assert abs(s.get_primal_obj() - o.get_primal_obj()) <= 1e-6 * (1 + abs(o.get_primal_obj())), (s.get_primal_obj(), o.get_primal_obj())
    assert abs(s.get_num_iters() - o.get_num_iters()) <= 2, (s.get_num_iters(), o.get_num_iters())


@pytest.mark.parametrize("seed", list(range(100, 112)))
def test_random_mixed_cone_models_larger_hip(seed):
    """the same with size ranges six times as wide (PSD sides to 41, spectral cones to 23 x 46, n to 48): past the 16-wide MFMA tiles
    and, for some draws, the single-block paths of the factorization kernels"""
    import hypatia_jl_amd as H
    from fuzz_models import random_model
    from oracle.build import make_cone as omake, make_model as omodel
    from oracle.solvers import Solver as OSolver
    inst = random_model(seed, omake, k=6)
    s = build_solve_check(H.Solver(default_tol_relax=10), H.make_model(inst), inst)
    o = OSolver(default_tol_relax=10)
    o.load(omodel(inst))
    o.solve()
    assert o.get_status() == "Optimal"
    assert abs(s.get_primal_obj() - o.get_primal_obj()) <= 1e-6 * (1 + abs(o.get_primal_obj())), (s.get_primal_obj(), o.get_primal_obj())


@pytest.mark.parametrize("seed", list(range(200, 216)))
@pytest.mark.parametrize("options", ["qrchol_noreduce", "symindef", "symindef_nopreprocess"])
def test_random_mixed_cone_models_other_option_sets_hip(seed, options):
    """the random models under the reference's other option sets (test/runnativetests.jl:80-86, 101-118): QRChol without the
    reduction, the device SymIndef solver with and without preprocessing -- independent routes to the same optimum"""
    import hypatia_jl_amd as H
    from fuzz_models import random_model
    from oracle.build import make_cone as omake
    inst = random_model(seed, omake, k=2)
    ref = build_solve_check(H.Solver(default_tol_relax=10), H.make_model(inst), inst)
    if options == "qrchol_noreduce":
        sv = H.Solver(default_tol_relax=10, reduce=False)
    elif options == "symindef":
        sv = H.Solver(default_tol_relax=10, reduce=False, syssolver=H.SymIndefDenseSystemSolver())
    else:
        sv = H.Solver(default_tol_relax=10, reduce=False, preprocess=False, syssolver=H.SymIndefDenseSystemSolver())
    got = build_solve_check(sv, H.make_model(inst), inst)
    assert abs(got.get_primal_obj() - ref.get_primal_obj()) <= 1e-6 * (1 + abs(ref.get_primal_obj()))


@pytest.mark.parametrize("seed,k", [(s, 1) for s in range(300, 340)] + [(s, 3) for s in range(400, 412)])
def test_random_models_over_all_device_cones_hip(seed, k):
    """the random models drawn from ALL cone kinds of the device path: the WSOS cones and the complex Hermitian variants included"""
    import hypatia_jl_amd as H
    from fuzz_models import random_model, ALL_KINDS
    from oracle.build import make_cone as omake, make_model as omodel
    from oracle.solvers import Solver as OSolver
    inst = random_model(seed, omake, k=k, kinds=ALL_KINDS)
    s = build_solve_check(H.Solver(default_tol_relax=10), H.make_model(inst), inst)
    o = OSolver(default_tol_relax=10)
    o.load(omodel(inst))
    o.solve()
    assert o.get_status() == "Optimal"
    assert abs(s.get_primal_obj() - o.get_primal_obj()) <= 1e-6 * (1 + abs(o.get_primal_obj())), (s.get_primal_obj(), o.get_primal_obj())


def _edge_names():
    from oracle import instances as I
    return sorted(I.EDGE_CASES)


@pytest.mark.parametrize("name", _edge_names())
@pytest.mark.parametrize("reduce", [True, False])
def test_edge_case_models_hip(name, reduce):
    """shapes at the ends of the ranges through the HIP path: no free variable left after the equalities, cones of dimension one,
    PSD sides 1 / 2 / 5 / 17 next to one that crosses the 128-wide blocks of the factorization kernels, a one-row spectral cone"""
    import hypatia_jl_amd as H
    from oracle import instances as I
    inst = I.EDGE_CASES[name]()
    build_solve_check(H.Solver(default_tol_relax=10, reduce=reduce), H.make_model(inst), inst)


@pytest.mark.parametrize("maker", ["linearopt", "linearopt_large", "nonnegative1", "nonnegative2", "nonnegative3"])
def test_lp_objective_matches_highs_hip(maker):
    """linear programs through the HIP path against an independent solver (scipy's HiGHS): config 1 of BASELINE.json
    (examples/linearopt, m = 50, n = 100), a larger one of the same family, and the reference's nonnegative1-3 constructions"""
    import hypatia_jl_amd as H
    from scipy.optimize import linprog
    from oracle import instances as I
    if maker == "linearopt":
        inst = I.linearopt(50, 100, seed=1)
    elif maker == "linearopt_large":
        inst = I.linearopt(300, 800, seed=2)
    else:
        inst = I.MORE_NATIVE[maker]()
    c, A, b, G, h = inst[:5]
    s = build_solve_check(H.Solver(default_tol_relax=10), H.make_model(inst), inst)
    r = linprog(c, A_ub=G, b_ub=h, A_eq=A, b_eq=b, bounds=(None, None), method="highs")
    assert r.status == 0, r.message
    ref = r.fun + inst[6].get("obj_offset", 0.0)
    assert abs(s.get_primal_obj() - ref) <= 1e-6 * (1 + abs(ref)), (s.get_primal_obj(), ref)


@pytest.mark.parametrize("name", ["possemideftri5", "possemideftri6", "possemideftri7"])
@pytest.mark.parametrize("reduce", [True, False])
def test_known_answer_hip_complex_psd(name, reduce):
    """the reference's complex Hermitian PosSemidefTri instances (test/nativeinstances.jl:382-437) through the HIP path"""
    import hypatia_jl_amd as H
    from oracle import instances as I
    inst = I.KNOWN_ANSWER_COMPLEX[name]()
    solver = H.Solver(default_tol_relax=10, reduce=reduce)
    build_solve_check(solver, H.make_model(inst), inst)


@pytest.mark.parametrize("name", ["linmatrixineq1_complex_side2", "linmatrixineq1_complex_side4", "linmatrixineq2_complex_cc",
                                  "linmatrixineq2_complex_rcr", "linmatrixineq2_complex_crr"])
def test_known_answer_hip_complex_linmatrixineq(name):
    """the complex members of the reference's linmatrixineq1 / linmatrixineq2 instances (test/nativeinstances.jl:696-745) through
    the HIP path"""
    import hypatia_jl_amd as H
    from oracle import instances as I
    inst = I.KNOWN_ANSWER_COMPLEX[name]()
    solver = H.Solver(default_tol_relax=10)
    build_solve_check(solver, H.make_model(inst), inst)


def _complex_ens_names():
    from oracle import instances as I
    return sorted(k for k in I.KNOWN_ANSWER_COMPLEX if k.startswith("epinormspectral"))


@pytest.mark.parametrize("name", _complex_ens_names())
def test_known_answer_hip_complex_epinormspectral(name):
    """the complex members of the reference's epinormspectral1 / 2 / 3 instances (test/nativeinstances.jl:1038-1125) through the
    HIP path"""
    import hypatia_jl_amd as H
    from oracle import instances as I
    inst = I.KNOWN_ANSWER_COMPLEX[name]()
    solver = H.Solver(default_tol_relax=10)
    build_solve_check(solver, H.make_model(inst), inst)


@pytest.mark.parametrize("name", ["hyporootdettri1_complex", "hyporootdettri2_complex", "hypoperlogdettri1_complex", "hypoperlogdettri2_complex",
                                  "hypoperlogdettri3_complex"])
def test_known_answer_hip_complex_hypograph_cones(name):
    """the complex members of the reference's hyporootdettri1 / 2 and hypoperlogdettri1 / 2 / 3 instances
    (test/nativeinstances.jl:1569-1760) through the HIP path"""
    import hypatia_jl_amd as H
    from oracle import instances as I
    inst = I.KNOWN_ANSWER_COMPLEX[name]()
    solver = H.Solver(default_tol_relax=10)
    build_solve_check(solver, H.make_model(inst), inst)


@pytest.mark.parametrize("name", ["wsosinterpnonnegative4", "wsosinterpnonnegative5"])
@pytest.mark.parametrize("reduce", [True, False])
def test_known_answer_hip_complex_wsos(name, reduce):
    """the reference's two complex WSOSInterpNonnegative instances (test/nativeinstances.jl:2345-2383: the minimum of 1 + |z|^2 over
    the unit disc / the bidisc, primal and dual form) through the HIP path, and the same optimum and iteration count as the oracle"""
    import hypatia_jl_amd as H
    from oracle import instances as I
    from oracle.build import make_model as omodel
    from oracle.solvers import Solver as OSolver
    inst = I.KNOWN_ANSWER_COMPLEX[name]()
    hs = build_solve_check(H.Solver(default_tol_relax=10, reduce=reduce), H.make_model(inst), inst)
    os_ = build_solve_check(OSolver(default_tol_relax=10, reduce=reduce), omodel(inst), inst)
    assert abs(hs.primal_obj - os_.primal_obj) <= 1e-6 * (1 + abs(os_.primal_obj))
    assert abs(hs.num_iters - os_.num_iters) <= 2


@pytest.mark.parametrize("path", ["fused", "unfused", "host_composed"])
def test_total_factorization_failure_ends_in_numerical_failure(path, monkeypatch):
    """every link of posdef_fact_copy! failed (HYP_FORCE_FACT_FAIL=1 reports that from hyp_sys_update_lhs_fact / _update_lhs /
    _step_directions): the reference warns (qrchol.jl:253-255) and its step ends in NumericalFailure (combined.jl:97-117) -- on
    every composition of the step, not in an exception from the constant-column solve"""
    import hypatia_jl_amd as H
    from oracle import instances as I
    monkeypatch.setenv("HYP_FORCE_FACT_FAIL", "1")
    if path == "unfused":
        monkeypatch.setenv("HYP_NO_FUSED_STEP", "1")
    if path == "host_composed":
        monkeypatch.setenv("HYP_NO_NATIVE", "1")
    s = H.Solver()
    s.load(H.make_model(I.psd_blocks(30, [6, 4], seed=1)))
    s.solve()
    assert s.status == "NumericalFailure"
    assert s.num_iters == 0


def _trajectory(solver_cls, model, **opts):
    rows = []
    s = solver_cls(**opts)
    s.iter_callback = lambda sv: rows.append((sv.primal_obj, sv.dual_obj, sv.gap, sv.x_feas, sv.z_feas, sv.point.tau,
                                              sv.point.kap, sv.mu, getattr(sv.stepper, "prev_alpha", 1.0)))
    s.load(model)
    s.solve()
    return s, np.array(rows)


@pytest.mark.parametrize("n,sides,seed", [(30, [6, 4], 1), (60, [10, 8, 3], 2), (150, [24, 17], 3),
                                          (50, [7, 7, 7, 7, 7], 4), (60, [5, 8, 8, 8, 8, 3], 5)])   # (runs of equal cones: group arena, batched inverses)
def test_trajectory_parity_psd(n, sides, seed):
    """Iterate-by-iterate parity with the CPU oracle.  Bar (trajectory_harness.compare): identical status and line-search step sizes on
    the prefix where the oracle's own trajectory survives 1-ulp perturbations of G and h; objective / mu / tau / residual norms agree to
    1e-10 relative while the iteration is well conditioned (mu >= 1e-3), and everywhere on that prefix to within 100x the oracle's own
    sensitivity to such a perturbation (the IPM amplifies rounding by ~1/mu near convergence; the reference stores no trajectories, so
    this restatement is the only comparison point).  The sensitivity is the worst of THREE perturbed draws: with one draw the bar
    at the last iterates is itself a single noisy sample, and a change of rounding anywhere in the library (round 4: the Cholesky's
    tile solves, same backward error -- tools/potrf_accuracy.py) moved one case across it."""
    import hypatia_jl_amd as H
    import trajectory_harness as T
    from oracle import instances as I
    inst = I.psd_blocks(n, sides, seed=seed)
    hs, ht = T.run_trajectory(H.Solver, H.make_model(inst))
    ot = T.oracle_trajectory(inst)
    pts = [T.oracle_trajectory(T.perturbed(inst, seed=99 + j))["rows"] for j in range(3)]
    rep = T.compare(dict(status=hs.status, rows=ht), ot, dict(rows=pts), label="psd_blocks(%d, %s, %d)" % (n, sides, seed))
    assert hs.status == ot["status"] == "Optimal"
    assert rep["prefix"] >= min(int(np.sum(ot["rows"][:, 7] >= 1e-7)), T.stable_prefix(ot["rows"], pts)), rep
    assert rep["prefix"] >= 8, rep
    assert abs(hs.num_iters - ot["iters"]) <= (0 if rep["prefix"] >= len(ot["rows"]) else 3)
    assert abs(hs.primal_obj - ot["p_obj"]) <= 1e-7 * (1 + abs(ot["p_obj"]))
    assert np.allclose(hs.get_x(), ot["x"], rtol=1e-5, atol=1e-7)


def test_system_solver_matches_oracle_single_update():
    """one update_lhs + one solve_subsystem3 at the initial point: Schur matrix and solution vs the oracle."""
    import hypatia_jl_amd as H
    from oracle import instances as I
    from oracle.build import make_model as omodel
    from oracle.solvers import Solver as OSolver
    inst = I.psd_blocks(300, [30, 11, 20], seed=7)
    hs = H.Solver(iter_limit=1)
    hs.load(H.make_model(inst)); hs.solve()
    os_ = OSolver(iter_limit=1)
    os_.load(omodel(inst)); os_.solve()
    Lh = np.triu(hs.syssolver.get_lhs())
    Lo = np.triu(os_.syssolver.lhs_sub)
    assert np.linalg.norm(Lh - Lo) / np.linalg.norm(Lo) < 1e-12
    assert np.linalg.norm(hs.syssolver.sol_const.vec - os_.syssolver.sol_const.vec) / np.linalg.norm(os_.syssolver.sol_const.vec) < 1e-9
    assert np.linalg.norm(hs.point.vec - os_.point.vec) / np.linalg.norm(os_.point.vec) < 1e-9


def test_linearopt_config1_hip():
    import hypatia_jl_amd as H
    from oracle import instances as I
    inst = I.linearopt(50, 100, seed=1)
    s = build_solve_check(H.Solver(default_tol_relax=10), H.make_model(inst), inst)
    assert s.model.n == 50 and s.model.p == 0 and s.model.q == 100


def test_small_polymin_and_matrixcompletion_hip():
    """configs 5 and 3b at small size through the HIP path (WSOS both forms, EpiNormSpectral)"""
    import hypatia_jl_amd as H
    from oracle import instances as I
    from oracle.build import make_model as omodel
    from oracle.solvers import Solver as OSolver
    for inst in (I.polymin(2, 3, True, seed=2), I.polymin(2, 3, False, seed=2), I.polymin(3, 2, False, seed=4), I.matrixcompletion(4, 6, seed=2)):
        hs = build_solve_check(H.Solver(default_tol_relax=10), H.make_model(inst), inst)
        os_ = build_solve_check(OSolver(default_tol_relax=10), omodel(inst), inst)
        assert abs(hs.primal_obj - os_.primal_obj) <= 1e-6 * (1 + abs(os_.primal_obj))
        assert abs(hs.num_iters - os_.num_iters) <= 2


@pytest.mark.parametrize("name,reduce", [("possemideftri2", True), ("possemideftri2", False), ("nonnegative4", False),
                                         ("epinormspectral2_primal", True), ("epinormspectral3_3x4_dual", True),
                                         ("wsosinterpnonnegative2", True)])
def test_device_get_directions_matches_host_composition_and_oracle(name, reduce):
    """hyp_sys_get_directions (6x6 solve + residual + refinement, all on the device) against (a) the same
    routine composed on the host from solve_subsystem3 / cone oracles / mul_G as the reference composes it
    (common.jl:15-182), and (b) the CPU oracle's get_directions, for the four right-hand sides of a step."""
    import hypatia_jl_amd as H
    from hypatia_jl_amd import solvers as HS
    from oracle import instances as I
    from oracle import solvers as OS
    from oracle.build import make_model as omodel
    inst = I.KNOWN_ANSWER[name]()
    hs = H.Solver(iter_limit=3, reduce=reduce)
    hs.load(H.make_model(inst)); hs.solve()          # a few iterations in: a generic interior point
    os_ = OS.Solver(iter_limit=3, reduce=reduce)
    os_.load(omodel(inst)); os_.solve()
    assert np.linalg.norm(hs.point.vec - os_.point.vec) <= 1e-8 * np.linalg.norm(os_.point.vec)
    st, sysv = hs.stepper, hs.syssolver
    assert sysv.native_directions
    sysv.update_lhs(hs)
    os_.syssolver.update_lhs(os_)
    for upd, oupd in ((HS.update_rhs_cent, OS.update_rhs_cent), (HS.update_rhs_pred, OS.update_rhs_pred)):
        upd(hs, st.rhs)
        sysv.native_directions = True
        hs.worst_dir_res = 0.0
        HS.get_directions(st, hs)
        d_dev = st.dir.vec.copy()
        res_dev = hs.worst_dir_res          # the residual the device routine reports (drives its refinement)
        HS.apply_lhs(st, hs)
        res_true = np.max(np.abs(st.temp.vec - st.rhs.vec))
        assert res_dev <= 10 * res_true + 1e-13 * (1 + np.max(np.abs(st.rhs.vec))), (name, res_dev, res_true)
        sysv.native_directions = False
        HS.get_directions(st, hs)
        d_host = st.dir.vec.copy()
        sysv.native_directions = True
        oupd(os_, os_.stepper.rhs)
        OS.get_directions(os_.stepper, os_)
        d_orc = os_.stepper.dir.vec
        scale = np.linalg.norm(d_orc)
        floor = 1e-15 * np.linalg.norm(hs.point.vec)     # (a nearly converged instance has directions of ~1e-6: rounding of the point itself)
        assert np.linalg.norm(d_dev - d_host) <= 1e-11 * scale + floor, name
        assert np.linalg.norm(d_dev - d_orc) <= 1e-7 * scale, name


@pytest.mark.parametrize("name,reduce", [("possemideftri2", True), ("possemideftri2", False), ("wsosinterpnonnegative2", True),
                                         ("epinormspectral3_3x4_dual", True)])
def test_paired_directions_match_single(name, reduce):
    """hyp_sys_get_directions2 (two right-hand sides per pass over G / the factor / the cone matrices) against two
    calls of the single-column routine, on the (cent, pred) pair of an interior iterate."""
    import hypatia_jl_amd as H
    from hypatia_jl_amd import solvers as HS
    from oracle import instances as I
    inst = I.KNOWN_ANSWER[name]()
    hs = H.Solver(iter_limit=3, reduce=reduce)
    hs.load(H.make_model(inst)); hs.solve()
    st, sysv = hs.stepper, hs.syssolver
    sysv.update_lhs(hs)
    singles = []
    for k, upd in enumerate((HS.update_rhs_cent, HS.update_rhs_pred)):
        upd(hs, st.rhs)
        st.rhs2[k] = st.rhs.vec
        HS.get_directions(st, hs)
        singles.append(st.dir.vec.copy())
    (ra, rb), ns = sysv.get_directions2_native(hs, st.dir2, st.rhs2)
    assert ns >= 2 and np.isfinite(ra) and np.isfinite(rb)
    for k in range(2):
        scale = np.linalg.norm(singles[k])
        # (1e-8: near-converged iterates have directions ~1e-6 while the intermediates of the 6x6 reduction are O(1))
        assert np.linalg.norm(st.dir2[k] - singles[k]) <= 1e-8 * scale, (name, k)


@pytest.mark.parametrize("name", ["possemideftri2", "epinormspectral3_3x4_dual", "wsosinterpnonnegative2", "nonnegative4"])
def test_fused_step_directions_match_unfused(name):
    """hyp_sys_step_directions (update_lhs + right-hand sides built on the device + two paired solves) against the same
    step composed from update_lhs, the host right-hand-side builders and hyp_sys_get_directions2."""
    import hypatia_jl_amd as H
    from hypatia_jl_amd import solvers as HS
    from oracle import instances as I
    inst = I.KNOWN_ANSWER[name]()
    hs = H.Solver(iter_limit=3)
    hs.load(H.make_model(inst)); hs.solve()
    if hs.model.p != 0:
        pytest.skip("fused step is the reduced-model path")
    st, sysv = hs.stepper, hs.syssolver
    hs.calc_convergence_params()
    hs.res_norm_cutoff = 1e-4 * max(hs.x_norm_res, hs.y_norm_res, hs.z_norm_res, hs.tau_feas)
    assert sysv.step_directions_native(hs, st)
    fused = [d.vec.copy() for d in (st.dir_cent, st.dir_pred, st.dir_centadj, st.dir_predadj)]
    # unfused composition at the same state
    sysv.update_lhs(hs)
    HS.update_rhs_cent(hs, st.rhs); st.rhs2[0] = st.rhs.vec
    HS.update_rhs_pred(hs, st.rhs); st.rhs2[1] = st.rhs.vec
    st._pair(hs, st.dir_cent, st.dir_pred)
    HS.update_rhs_centadj(hs, st.rhs, st.dir_cent); st.rhs2[0] = st.rhs.vec
    HS.update_rhs_predadj(hs, st.rhs, st.dir_pred); st.rhs2[1] = st.rhs.vec
    st._pair(hs, st.dir_centadj, st.dir_predadj)
    unf = [d.vec.copy() for d in (st.dir_cent, st.dir_pred, st.dir_centadj, st.dir_predadj)]
    for k in range(4):
        scale = np.linalg.norm(unf[k]) + 1e-300
        assert np.linalg.norm(fused[k] - unf[k]) <= 1e-8 * scale + 1e-14, (name, k, np.linalg.norm(fused[k] - unf[k]) / scale)


def test_native_search_alpha_matches_host_walk():
    """hyp_sys_search_alpha (candidates formed natively, whole schedule in one call) against the host walk of the
    alpha schedule through hyp_sys_check_cone_points, from the same iterate and directions."""
    import hypatia_jl_amd as H
    from hypatia_jl_amd import solvers as HS
    from oracle import instances as I
    inst = I.psd_blocks(60, [10, 8, 3], seed=2)
    hs = H.Solver(iter_limit=4)
    hs.load(H.make_model(inst)); hs.solve()
    st, sysv = hs.stepper, hs.syssolver
    hs.calc_convergence_params()
    hs.res_norm_cutoff = 1e-4 * max(hs.x_norm_res, hs.y_norm_res, hs.z_norm_res, hs.tau_feas)
    assert sysv.step_directions_native(hs, st)
    st.unadj_only = st.cent_only = False
    a_native = sysv.search_alpha_native(hs.model, hs.point, st, 1)
    prox_native = st.searcher.prox
    # host walk
    a_host = 0.0
    for alpha in st.searcher.alpha_sched:
        st.update_stepper_points(alpha, hs.point, True)
        if sysv.check_cone_points_native(hs.model, st.temp, st.searcher):
            a_host = alpha
            break
    assert a_native == a_host and a_native > 0
    assert abs(prox_native - st.searcher.prox) <= 1e-9 * (1 + prox_native)


def test_resident_search_alpha_same_candidate_and_error_paths(monkeypatch):
    """hyp_sys_search_alpha_resident (one PosSemidefTri cone: candidates formed on the device from what step_directions left
    there, screened side by side) against hyp_sys_search_alpha on the host vectors of the same step: same accepted index, the
    accepted candidate bit for bit, same proximity value, same number of candidates visited.  And its contract: an error
    (no crash, message from hyp_last_error) without a preceding step_directions and for a model the screen does not apply to."""
    import ctypes
    import hypatia_jl_amd as H
    from hypatia_jl_amd import _lib as L
    from oracle import instances as I
    inst = I.psd_blocks(40, [48], seed=5)
    # (the host-vector search of this comparison reads the z / s rows of the directions: the whole vectors are downloaded here, not
    #  only the x rows a stepper on the resident search needs -- hyp_sys_set_direction_rows, round 6)
    monkeypatch.setenv("HYP_DIRS_X_ONLY", "0")
    hs = H.Solver(iter_limit=3)
    hs.load(H.make_model(inst)); hs.solve()
    st, sysv = hs.stepper, hs.syssolver
    assert sysv._screen_ok()
    results = []
    irtmu = 1.0 / np.sqrt(hs.mu)
    for resident in (True, False):
        for k, cone in enumerate(hs.model.cones):       # the cones back at the iterate (a search leaves them at its candidate)
            cone.reset_data()
            cone.load_point(hs.point.primal_views[k], irtmu)
            cone.load_dual_point(hs.point.dual_views[k])
            assert cone.is_feas()
        hs.calc_convergence_params()
        hs.res_norm_cutoff = 1e-4 * max(hs.x_norm_res, hs.y_norm_res, hs.z_norm_res, hs.tau_feas)
        assert sysv.step_directions_native(hs, st)      # (same point both times: the same directions to the last bit)
        sysv._dirs_resident = resident
        st.unadj_only = st.cent_only = False
        n0 = st.searcher.n_trials
        a = sysv.search_alpha_native(hs.model, hs.point, st, 1)
        results.append((a, st.temp.ztsk.copy(), st.searcher.prox, st.searcher.n_trials - n0))
        sysv._dirs_resident = False
    (a1, c1, p1, t1), (a2, c2, p2, t2) = results
    assert a1 == a2 > 0 and t1 == t2
    assert np.array_equal(c1, c2)
    assert p1 == p2

    def resident_rc(solver):
        sv = solver.syssolver
        sc = np.ascontiguousarray(solver.stepper.searcher.alpha_sched, dtype=np.float64)
        cand = np.zeros(2 * solver.model.q + 2)
        idx, nt, nl = ctypes.c_int(-1), ctypes.c_int(0), ctypes.c_int(0)
        prox, irt = ctypes.c_double(0.0), ctypes.c_double(0.0)
        return L.lib().hyp_sys_search_alpha_resident(sv._h, 0, 0, L.vec_ptr(sc), len(sc), 0, 0.01, 0.99, 1, float(solver.model.nu + 1),
                                                     L.vec_ptr(cand), ctypes.byref(idx), ctypes.byref(prox), ctypes.byref(nt), ctypes.byref(nl),
                                                     ctypes.byref(irt))
    fresh = H.Solver(iter_limit=0)
    fresh.load(H.make_model(inst)); fresh.solve()          # system solver loaded, no step taken
    assert resident_rc(fresh) != 0
    two = H.Solver(iter_limit=2)
    two.load(H.make_model(I.psd_blocks(40, [40, 33], seed=6))); two.solve()   # two cones: the screen does not apply
    assert not two.syssolver._screen_ok()
    assert resident_rc(two) != 0


# ---------------------------------------------------------------------------------------------
# SymIndefDenseSystemSolver: the 3x3 symmetric indefinite form, Bunch-Kaufman on the device
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", sorted(_instances()))
@pytest.mark.parametrize("preprocess", [True, False])
def test_known_answer_hip_symindef(name, preprocess):
    """the reference's option sets for this solver: test/runnativetests.jl:80-86 (no preprocessing) and :101-118
    (reduce = false)"""
    import hypatia_jl_amd as H
    from oracle import instances as I
    if not preprocess and name in ("dimension1",):
        pytest.skip("needs preprocessing (dependent equalities)")
    inst = I.KNOWN_ANSWER[name]()
    solver = H.Solver(default_tol_relax=10, reduce=False, preprocess=preprocess, syssolver=H.SymIndefDenseSystemSolver())
    build_solve_check(solver, H.make_model(inst), inst)


def test_symindef_matches_oracle_and_qrchol():
    """left-hand side and first iterate against the oracle's SymIndefDense; same optimum as the QRChol path"""
    import hypatia_jl_amd as H
    from oracle import instances as I
    from oracle.build import make_model as omodel
    from oracle.solvers import Solver as OSolver, SymIndefDenseSystemSolver as OSym
    inst = I.psd_blocks(40, [9, 6], seed=5)
    hs = H.Solver(iter_limit=1, reduce=False, syssolver=H.SymIndefDenseSystemSolver())
    hs.load(H.make_model(inst)); hs.solve()
    os_ = OSolver(iter_limit=1, reduce=False, syssolver=OSym())
    os_.load(omodel(inst)); os_.solve()
    Lh, Lo = np.triu(hs.syssolver.get_lhs()), np.triu(os_.syssolver.lhs_sub)
    assert np.linalg.norm(Lh - Lo) / np.linalg.norm(Lo) < 1e-11
    assert np.linalg.norm(hs.point.vec - os_.point.vec) / np.linalg.norm(os_.point.vec) < 1e-8
    full = H.Solver(reduce=False, syssolver=H.SymIndefDenseSystemSolver())
    full.load(H.make_model(inst)); full.solve()
    ref = H.Solver()
    ref.load(H.make_model(inst)); ref.solve()
    assert full.status == ref.status == "Optimal"
    assert abs(full.primal_obj - ref.primal_obj) <= 1e-7 * (1 + abs(ref.primal_obj))
    assert np.allclose(full.get_x(), ref.get_x(), rtol=1e-5, atol=1e-7)
