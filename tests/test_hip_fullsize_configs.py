"""The other BASELINE.json configurations at FULL size on the GPU (config 2 is tests/test_hip_fullsize.py), checked through
size-independent properties -- the CPU oracle would need minutes per iteration at these sizes:
  3b  matrix completion, EpiNormSpectral 50 x 100 (dim 5001): the largest size the reference's algorithm admits
  4   64 x PosSemidefTri(80), q = 207 360, n = 5000 (G = 8.3 GB resident)
  5   polymin, WSOSInterpNonnegative, 4 variables, half-degree 8 (U = 4845), primal and dual form
Properties: the conic certificate of test/nativeinstances.jl:58-65 on the full solve (3b, 5), the Schur matrix on probe
vectors and the KKT residual of the stepper directions (4), Hessian / inverse-Hessian identities of the big cones."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _certificate(solver, inst, tol):
    c, A, b, G, h = inst[:5]
    x, y, z, s = solver.get_x(), solver.get_y(), solver.get_z(), solver.get_s()
    rel = lambda a, bb: np.linalg.norm(a - bb) / (1 + np.linalg.norm(bb))
    assert solver.status == "Optimal"
    assert abs(solver.primal_obj - solver.dual_obj) <= tol * (1 + abs(solver.primal_obj))
    assert rel(G @ x + s, h) <= tol
    assert rel(G.T @ z + (A.T @ y if len(b) else 0.0), -c) <= tol
    if len(b):
        assert rel(A @ x, b) <= tol
    assert abs(s @ z) <= np.sqrt(tol) * (1 + abs(solver.primal_obj))
    from instance_harness import check_membership
    check_membership(inst[5], s, z, tol)     # s in K, z in K* by the cones' definitions (singular values, moment matrices)


@pytest.mark.timeout(600)
def test_config3b_matrix_completion_full_size():
    import hypatia_jl_amd as H
    from oracle import instances as I          # instance generator only (data)
    inst = I.matrixcompletion(50, 100, seed=1)
    model = H.make_model(inst)
    cone = model.cones[0]
    assert cone.dimension() == 5001
    # the cone three iterations in (well conditioned): closed-form hess_prod, then H^-1 through the explicit Hessian's factorization
    s = H.Solver(verbose=False)
    s.load(model)
    s.setup()
    for _ in range(3):
        assert s.iterate()
    rng = np.random.default_rng(0)
    V = np.asfortranarray(rng.standard_normal((5001, 3)))
    P, Q = np.zeros_like(V), np.zeros_like(V)
    assert cone.is_feas()
    cone.hess_prod(P, V)
    cone.inv_hess_prod(Q, P)
    assert np.linalg.norm(Q - V) <= 1e-8 * np.linalg.norm(V)
    g = np.array(cone.get_grad())
    Hg = np.zeros(5001)
    cone.inv_hess_prod(Hg, g)
    assert abs(Hg @ g - cone.get_nu()) <= 1e-8 * cone.get_nu()        # <g, H^-1 g> = nu (logarithmic homogeneity)
    # the whole solve
    s = H.Solver(verbose=False)
    s.load(H.make_model(inst))
    s.solve()
    _certificate(s, inst, 1e-6)


@pytest.mark.timeout(900)
def test_config3c_matrix_completion_beyond_the_reference_limit():
    """matrix completion with EpiNormSpectral at 150 x 150 (dim 22 501, n = 10 086): 4.5 times the dimension at which the
    reference's own route stops being feasible (an explicit 22 501^2 Hessian = 4 GB and its Cholesky per line-search trial,
    Cones.jl:113-118), solved through the closed-form inverse Hessian; the reference's sampling rule for the known entries
    (examples/matrixcompletion/native.jl:29-43).  Checked without any barrier code: the conic certificate of
    test/nativeinstances.jl:58-65, s in the spectral-norm cone and z in the nuclear-norm cone by their singular values, and the
    completed matrix agrees with the known entries."""
    import hypatia_jl_amd as H
    from oracle import instances as I          # instance generator only (data)
    d = 150
    inst = I.matrixcompletion(d, d, seed=1, with_replacement=True)
    assert abs((inst[3].shape[1] - 1) / (d * d) - 0.449) < 0.01           # 1 - exp(-0.8) of the entries known
    s = H.Solver(verbose=False, init_use_indirect=True)
    s.load(H.make_model(inst))
    s.solve()
    _certificate(s, inst, 1e-6)
    # the objective is the spectral norm of the completed matrix, whose known entries are the data
    h = inst[4]
    W = s.get_s()[1:].reshape((d, d), order="F")
    known = np.abs(inst[3][1:, 1:]).sum(axis=1) == 0
    assert np.allclose(W.reshape(-1, order="F")[known], h[1:][known], atol=1e-6)
    assert abs(np.linalg.svd(W, compute_uv=False)[0] - s.primal_obj) <= 1e-5 * (1 + s.primal_obj)


@pytest.mark.timeout(900)
def test_config5_polymin_primal_full_size():
    import hypatia_jl_amd as H
    from oracle import polyutils as pu         # interpolation basis (data)
    rng = np.random.default_rng(1)
    U, pts, Ps = pu.interpolate_box([-1.0] * 4, [1.0] * 4, 8, rng=rng, sample_factor=2)
    assert U == 4845 and [P.shape[1] for P in Ps] == [495, 330, 330, 330, 330]
    a = rng.uniform(-0.5, 0.5, 4)
    vals = np.sum((pts - a) ** 2, axis=1) + (pts[:, 0] * pts[:, 1] - pts[:, 2] * pts[:, 3]) ** 2 + 0.3 * pts[:, 0] * pts[:, 2]
    inst = (np.array([-1.0]), np.zeros((0, 1)), np.zeros(0), np.ones((U, 1)), vals, [("wsosinterpnonnegative", U, Ps, False)], {})
    s = H.Solver(verbose=False)
    s.load(H.make_model(inst))
    s.setup()
    for _ in range(3):
        assert s.iterate()
    cone = s.model.cones[0]
    assert cone.get_nu() == 495 + 4 * 330
    V = np.asfortranarray(rng.standard_normal((U, 2)))
    P, Q = np.zeros_like(V), np.zeros_like(V)
    assert cone.is_feas()
    cone.hess_prod(P, V)
    cone.inv_hess_prod(Q, P)
    assert np.linalg.norm(Q - V) <= 1e-8 * np.linalg.norm(V)
    g = np.array(cone.get_grad())
    Hg = np.zeros(U)
    cone.inv_hess_prod(Hg, g)
    assert abs(Hg @ g - cone.get_nu()) <= 1e-8 * cone.get_nu()
    s = H.Solver(verbose=False)
    s.load(H.make_model(inst))
    s.solve()
    _certificate(s, inst, 1e-6)
    # the optimum is a lower bound of the sampled polynomial
    assert -s.primal_obj <= vals.min() + 1e-6


@pytest.mark.timeout(900)
def test_config5_polymin_dual_full_size():
    """round 5: config 5 in the DUAL form (examples/polymin/native.jl:56-90 with use_primal = false: c = vals, A = ones(1, U), b = 1,
    G = -I, the cone itself (not its dual); after the reduction n = U - 1 = 4844 and the 4844 x 4844 Schur matrix is the "MFMA
    Hessian product" of BASELINE.json's configs[4]) solved to Optimal at U = 4845 and certified without barrier code: primal / dual
    residuals, zero gap, complementarity, and s in the moment cone by its definition -- every P_k' diag(s) P_k positive
    semidefinite (check_membership).  The optimum is the same lower bound the primal form finds (strong duality)."""
    import hypatia_jl_amd as H
    from oracle import polyutils as pu         # interpolation basis (data)
    rng = np.random.default_rng(1)
    U, pts, Ps = pu.interpolate_box([-1.0] * 4, [1.0] * 4, 8, rng=rng, sample_factor=2)
    assert U == 4845
    a = rng.uniform(-0.5, 0.5, 4)
    vals = np.sum((pts - a) ** 2, axis=1) + (pts[:, 0] * pts[:, 1] - pts[:, 2] * pts[:, 3]) ** 2 + 0.3 * pts[:, 0] * pts[:, 2]
    inst = (vals, np.ones((1, U)), np.array([1.0]), -np.eye(U), np.zeros(U), [("wsosinterpnonnegative", U, Ps, True)], {})
    s = H.Solver(verbose=False)
    s.load(H.make_model(inst))
    s.solve()
    _certificate(s, inst, 1e-6)
    # a probability-like measure on the points (x >= 0 in the moment cone, sum x = 1) whose mean value is the bound
    x = s.get_x()
    assert abs(x.sum() - 1.0) <= 1e-6 and abs(vals @ x - s.primal_obj) <= 1e-8 * (1 + abs(s.primal_obj))
    assert s.primal_obj <= vals.min() + 1e-6
    # the primal form of the same polynomial reaches the same value
    instp = (np.array([-1.0]), np.zeros((0, 1)), np.zeros(0), np.ones((U, 1)), vals, [("wsosinterpnonnegative", U, Ps, False)], {})
    sp = H.Solver(verbose=False)
    sp.load(H.make_model(instp))
    sp.solve()
    assert sp.status == "Optimal" and abs(-sp.primal_obj - s.primal_obj) <= 1e-6 * (1 + abs(s.primal_obj))


@pytest.mark.timeout(1200)
def test_config4_64_psd_blocks_full_size():
    import hypatia_jl_amd as H
    from hypatia_jl_amd import solvers as HS
    from oracle import instances as I
    inst = I.psd_blocks(5000, [80] * 64, seed=1, dtype=np.float32)      # (float32 draws: 8.3 GB of G generated twice as fast)
    s = H.Solver(verbose=False, init_use_indirect=True)
    s.load(H.make_model(inst))
    s.setup()
    assert s.model.q == 207360 and s.model.n == 5000 and len(s.model.cones) == 64
    for _ in range(2):
        assert s.iterate()
    sysv = s.syssolver
    sysv.update_lhs(s)
    lhs = np.triu(sysv.get_lhs())
    lhs = lhs + np.triu(lhs, 1).T
    rng = np.random.default_rng(0)
    v = rng.standard_normal(5000)
    Gv = sysv.mul_G(False, v)
    HGv = np.zeros_like(Gv)
    for cone, idx in zip(s.model.cones, s.model.cone_idxs):
        out = np.zeros(cone.dimension())
        cone.hess_prod(out, np.ascontiguousarray(Gv[idx]))
        HGv[idx] = out
    ref = sysv.mul_G(True, HGv)
    assert np.linalg.norm(lhs @ v - ref) <= 1e-10 * np.linalg.norm(ref)          # Schur matrix = sum_k G_k' H_k G_k
    st = s.stepper
    HS.update_rhs_cent(s, st.rhs); st.rhs2[0] = st.rhs.vec
    HS.update_rhs_pred(s, st.rhs); st.rhs2[1] = st.rhs.vec
    (ra, rb), ns = sysv.get_directions2_native(s, st.dir2, st.rhs2)
    for k in range(2):                                                             # KKT residual of the directions
        st.rhs.vec[:] = st.rhs2[k]
        st.dir.vec[:] = st.dir2[k]
        HS.apply_lhs(st, s)
        true_res = np.max(np.abs(st.temp.vec - st.rhs.vec))
        assert true_res <= 1e-9 * (1 + np.max(np.abs(st.rhs.vec)))
    mu0 = s.mu
    assert s.iterate() and 0 < s.mu < mu0 and s.stepper.searcher.prox < 0.99
