"""Pin the CPU oracle's solver path with the reference's deterministic known-answer instances
(/root/reference/test/nativeinstances.jl; option sets of test/runnativetests.jl:13-18 and :101-118:
default options with default_tol_relax = 10, and QRCholDenseSystemSolver with reduce = false)."""
import numpy as np
import pytest

from oracle import instances as inst_mod
from oracle.build import make_model
from oracle.solvers import Solver
from instance_harness import build_solve_check


@pytest.mark.parametrize("name", sorted(inst_mod.KNOWN_ANSWER))
@pytest.mark.parametrize("reduce", [True, False])
def test_known_answer(name, reduce):
    inst = inst_mod.KNOWN_ANSWER[name]()
    solver = Solver(default_tol_relax=10, reduce=reduce)
    build_solve_check(solver, make_model(inst), inst)


@pytest.mark.parametrize("name", sorted(inst_mod.KNOWN_ANSWER))
@pytest.mark.parametrize("preprocess", [True, False])
def test_known_answer_symindef(name, preprocess):
    """SymIndefDenseSystemSolver (symindef.jl:203-271) with the option sets of test/runnativetests.jl:80-86 (no
    preprocessing) and :101-118 (reduce = false): an independent factorization (Bunch-Kaufman of the 3x3 system) of the
    same Newton systems."""
    from oracle.solvers import SymIndefDenseSystemSolver
    inst = inst_mod.KNOWN_ANSWER[name]()
    if not preprocess and name in ("dimension1",):
        pytest.skip("needs preprocessing (dependent equalities, test/runnativetests.jl inst_preproc)")
    solver = Solver(default_tol_relax=10, reduce=False, preprocess=preprocess, syssolver=SymIndefDenseSystemSolver())
    build_solve_check(solver, make_model(inst), inst)


def test_linearopt_config1():
    inst = inst_mod.linearopt(50, 100, seed=1)
    s = build_solve_check(Solver(default_tol_relax=10), make_model(inst), inst)
    assert s.model.n == 50 and s.model.p == 0 and s.model.q == 100   # reduced sizes, SURVEY 8d cfg 1


def test_small_psd_blocks():
    inst = inst_mod.psd_blocks(12, [4, 3, 5], seed=3)
    build_solve_check(Solver(default_tol_relax=10), make_model(inst), inst)


def test_small_polymin_both_forms():
    for use_primal in (True, False):
        inst = inst_mod.polymin(2, 2, use_primal, seed=2)
        build_solve_check(Solver(default_tol_relax=10), make_model(inst), inst)


def test_small_matrixcompletion():
    inst = inst_mod.matrixcompletion(3, 4, seed=2)
    build_solve_check(Solver(default_tol_relax=10), make_model(inst), inst)


@pytest.mark.parametrize("name", sorted(inst_mod.MORE_NATIVE))
def test_more_native_instances(name):
    """further native instances of the reference for the cones of the device path (oracle/instances.py: MORE_NATIVE): the
    preprocessing cases (consistent1, inconsistent1 / 2: Optimal, PrimalInconsistent, DualInconsistent), an objective
    offset, the matrix-free set-up, a rank-deficient hypograph instance at its own tolerance, a model without variables"""
    inst = inst_mod.MORE_NATIVE[name]()
    import inspect
    opts = dict(default_tol_relax=10)
    known = inspect.signature(Solver.__init__).parameters   # (the oracle has no LSQR initial point: it takes the QR one there)
    opts.update({k: v for k, v in inst[6].get("solver_opts", {}).items() if k in known})
    if opts.get("syssolver") == "symindef":
        from oracle.solvers import SymIndefDenseSystemSolver
        opts["syssolver"] = SymIndefDenseSystemSolver()
    build_solve_check(Solver(**opts), make_model(inst), inst)


def _highs_objective(inst):
    """the same linear program through an independent algorithm and code base: scipy's HiGHS (min c'x, A x = b, G x <= h)"""
    from scipy.optimize import linprog
    c, A, b, G, h = inst[:5]
    r = linprog(c, A_ub=G, b_ub=h, A_eq=A if A.shape[0] else None, b_eq=b if A.shape[0] else None, bounds=(None, None), method="highs")
    assert r.status == 0, r.message
    return r.fun


@pytest.mark.parametrize("maker", ["linearopt", "nonnegative1", "nonnegative2", "nonnegative3"])
def test_lp_objective_matches_highs(maker):
    """the restatement of the Julia driver sits on both sides of the trajectory-parity tests; for linear programs an
    independent solver exists in the image: the oracle's optimal value against HiGHS"""
    inst = inst_mod.linearopt(50, 100, seed=1) if maker == "linearopt" else inst_mod.MORE_NATIVE[maker]()
    s = build_solve_check(Solver(default_tol_relax=10), make_model(inst), inst)
    ref = _highs_objective(inst) + inst[6].get("obj_offset", 0.0)
    assert abs(s.get_primal_obj() - ref) <= 1e-6 * (1 + abs(ref)), (s.get_primal_obj(), ref)


@pytest.mark.parametrize("name", sorted(inst_mod.EDGE_CASES))
def test_edge_case_models(name):
    """shapes at the ends of the ranges (oracle/instances.py: EDGE_CASES): certified by residuals, gap and cone membership"""
    inst = inst_mod.EDGE_CASES[name]()
    build_solve_check(Solver(default_tol_relax=10), make_model(inst), inst)


@pytest.mark.parametrize("seed", list(range(0, 40, 2)))
def test_random_mixed_cone_models(seed):
    """random strictly feasible models over a random mix of cones (tests/fuzz_models.py) on the oracle: Optimal with the certificate"""
    from fuzz_models import random_model
    from oracle.build import make_cone
    inst = random_model(seed, make_cone)
    build_solve_check(Solver(default_tol_relax=10), make_model(inst), inst)
