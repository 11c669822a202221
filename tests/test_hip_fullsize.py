"""Config 2 at BASELINE.json's full size (n = 5000, one PosSemidefTri of side 200, q = 20100) on the GPU, checked through
size-independent properties (the CPU oracle needs ~4 s per iteration here, so it is not run):
Schur matrix = G' H G on probe vectors, factor / solve round trip, KKT residual of the stepper directions,
line-search invariants, and bitwise reproducibility of an iteration."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def solver():
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    import hypatia_jl_amd as H
    from threadpoolctl import threadpool_limits
    inst = bench.gen_instance(5000, [200], 1)
    with threadpool_limits(limits=8, user_api="blas"):
        s = H.Solver(verbose=False)
        s.load(H.make_model(inst))
        s.setup()
    for _ in range(3):
        assert s.iterate()
    return s


def test_schur_matrix_is_G_H_G_on_probes(solver):
    s = solver
    sysv, cone = s.syssolver, s.model.cones[0]
    sysv.update_lhs(s)
    lhs = np.triu(sysv.get_lhs())
    lhs = lhs + np.triu(lhs, 1).T
    assert np.all(np.diag(lhs) > 0)
    rng = np.random.default_rng(0)
    for _ in range(3):
        v = rng.standard_normal(s.model.n)
        Gv = sysv.mul_G(False, v)
        HGv = np.zeros_like(Gv)
        cone.hess_prod(HGv, Gv)
        ref = sysv.mul_G(True, HGv)
        got = lhs @ v
        assert np.linalg.norm(got - ref) <= 1e-10 * np.linalg.norm(ref)
        # sqrt form: v' lhs v = || sqrt_hess_prod(G v) ||^2
        w = np.zeros_like(Gv)
        cone.sqrt_hess_prod(w, Gv)
        assert abs(v @ got - w @ w) <= 1e-10 * (w @ w)


def test_factor_solve_round_trip(solver):
    import ctypes
    from hypatia_jl_amd import _lib as L
    s = solver
    sysv = s.syssolver
    sysv.update_lhs(s)
    lhs = np.triu(sysv.get_lhs())
    lhs = lhs + np.triu(lhs, 1).T
    rng = np.random.default_rng(1)
    x = rng.standard_normal(s.model.n)
    b = lhs @ x
    y = b.copy()
    L.check(L.lib().hyp_sys_potrs(sysv._h, L.vec_ptr(y)), "potrs")
    # backward error of the solve (the matrix is ill conditioned late in the IPM: compare residuals, not x)
    assert np.linalg.norm(lhs @ y - b) <= 1e-12 * (np.linalg.norm(lhs, 2) * np.linalg.norm(y) + np.linalg.norm(b))


def test_stepper_directions_satisfy_the_kkt_system(solver):
    from hypatia_jl_amd import solvers as HS
    s = solver
    st, sysv = s.stepper, s.syssolver
    sysv.update_lhs(s)
    HS.update_rhs_cent(s, st.rhs); st.rhs2[0] = st.rhs.vec
    HS.update_rhs_pred(s, st.rhs); st.rhs2[1] = st.rhs.vec
    (ra, rb), ns = sysv.get_directions2_native(s, st.dir2, st.rhs2)
    for k, rep in enumerate((ra, rb)):
        st.rhs.vec[:] = st.rhs2[k]
        st.dir.vec[:] = st.dir2[k]
        HS.apply_lhs(st, s)                       # host-composed K * dir through the per-call entry points
        true_res = np.max(np.abs(st.temp.vec - st.rhs.vec))
        scale = 1 + np.max(np.abs(st.rhs.vec))
        assert true_res <= 1e-9 * scale, (k, true_res)
        assert rep <= 10 * true_res + 1e-12 * scale


def test_accepted_step_stays_in_the_neighbourhood_and_is_reproducible(solver):
    s = solver
    mu0 = s.mu
    assert s.iterate()
    alpha = s.stepper.prev_alpha
    assert 0 < alpha <= 1
    assert 0 < s.mu < mu0                      # the combined direction reduces mu
    assert s.stepper.searcher.prox < 0.99      # search.jl:33 prox_bound
    cone = s.model.cones[0]
    assert cone.is_feas() and cone.is_dual_feas()
    # bitwise reproducibility: two runs of the same two iterations from the initial iterate
    runs = []
    for _ in range(2):
        s.reset_iterate()
        assert s.iterate() and s.iterate()
        runs.append(s.point.vec.copy())
    assert np.array_equal(runs[0], runs[1])


@pytest.mark.timeout(900)
def test_full_solve_certificate_and_cone_membership():
    """config 2 solved to the end at full size: status Optimal, the conic certificate of test/nativeinstances.jl:58-65 and --
    independent of every barrier code -- smat(s) and smat(z) positive semidefinite by their eigenvalues"""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    import hypatia_jl_amd as H
    from instance_harness import check_membership
    from threadpoolctl import threadpool_limits
    inst = bench.gen_instance(5000, [200], 1)
    c, A, b, G, h = inst[:5]
    with threadpool_limits(limits=8, user_api="blas"):
        s = H.Solver(verbose=False)
        s.load(H.make_model(inst))
        s.solve()
    assert s.status == "Optimal"
    x, z, sv = s.get_x(), s.get_z(), s.get_s()
    rel = lambda a, bb: np.linalg.norm(a - bb) / (1 + np.linalg.norm(bb))
    tol = 1e-6
    assert abs(s.primal_obj - s.dual_obj) <= tol * (1 + abs(s.primal_obj))
    assert abs(c @ x - s.primal_obj) <= tol * (1 + abs(s.primal_obj))
    assert rel(G @ x + sv, h) <= tol
    assert rel(G.T @ z, -c) <= tol
    assert abs(sv @ z) <= np.sqrt(tol) * (1 + abs(s.primal_obj))
    check_membership(inst[5], sv, z, tol)
