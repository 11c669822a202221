"""BASELINE.json configurations at their STATED size, oracle by oracle against the CPU restatement:
  configs[2]  EpiNormSpectral 500 x 500 (dim 250 001; "3a" of SURVEY 8d): update_feas, is_dual_feas, grad, hess_aux,
              hess_prod! on 256 columns, dder3 -- HIP vs oracle/cones.py:442-597 (epinormspectral.jl:107-294).  The explicit
              250 001^2 Hessian is never formed on either side.
  configs[4]  WSOSInterpNonnegative, 4 variables, half-degree 8 (U = 4845): grad, hess_prod!, hess_prod_slow!, dder3,
              inv_hess_prod! -- HIP vs oracle/cones.py:601-690 (wsosinterpnonnegative.jl:89-200, Cones.jl:101-118, 239-251);
              and the DUAL form of the polymin instance (n = 4844 after the reduction, "MFMA Hessian-product" form) end to end
              with the conic certificate of test/nativeinstances.jl:58-65.
  configs[1]  size: the device least-squares initial x (hyp_dense_lstsq_normal) against the reference's column-pivoted QR
              (process.jl:64-178) on the same 20100 x 5000 matrix."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-300)


@pytest.mark.timeout(1200)
def test_config3a_epinormspectral_500x500_oracle_by_oracle():
    import hypatia_jl_amd as H
    from oracle import cones as OC
    d1 = d2 = 500
    hc, oc = H.EpiNormSpectral(d1, d2), OC.EpiNormSpectral(d1, d2)
    dim = 1 + d1 * d2
    assert hc.dimension() == oc.dimension() == dim and hc.get_nu() == oc.get_nu() == d1 + 1
    for c in (hc, oc):
        c.setup_data()
    rng = np.random.default_rng(500)
    # W = 0.5 randn / ||.||_2, u = 1
    Wm = rng.standard_normal((d1, d2))
    Wm *= 0.5 / np.linalg.norm(Wm, 2)
    pt = np.concatenate([[1.0], Wm.reshape(-1, order="F")])
    # dual point: feasible for the nuclear-norm cone with a small margin, then a marginally infeasible one
    Dm = rng.standard_normal((d1, d2))
    nn = float(np.sum(np.linalg.svd(Dm, compute_uv=False)))
    for margin, expect in ((1e-9, True), (-1e-9, False)):
        dual = np.concatenate([[nn * (1 + margin)], Dm.reshape(-1, order="F")])
        for c in (hc, oc):
            c.load_dual_point(dual)
        assert oc.is_dual_feas() == expect
        assert hc.is_dual_feas() == expect
    for c in (hc, oc):
        c.reset_data()
        c.load_point(pt, 1.3)
        assert c.is_feas()
    assert np.array_equal(hc.point, oc.point)
    g_h, g_o = np.array(hc.get_grad()), np.array(oc.get_grad())
    assert rel(g_h, g_o) <= 1e-10
    assert abs(g_h[0] - g_o[0]) <= 1e-10 * abs(g_o[0])
    for c in (hc, oc):
        c.update_hess_aux()
    ncols = 256
    V = np.asfortranarray(rng.standard_normal((dim, ncols)))
    Ph = np.zeros((dim, ncols), order="F")
    Po = np.zeros((dim, ncols), order="F")
    hc.hess_prod(Ph, V)
    oc.hess_prod(Po, V)
    assert rel(Ph, Po) <= 1e-10
    assert rel(Ph[0], Po[0]) <= 1e-10                     # the u-row on its own (one entry per column among 250 001)
    worst = max(rel(Ph[:, j], Po[:, j]) for j in range(ncols))
    assert worst <= 1e-10, worst
    # one vector, through the single-column path
    ph1, po1 = np.zeros(dim), np.zeros(dim)
    hc.hess_prod(ph1, V[:, 0].copy())
    oc.hess_prod(po1, V[:, 0].copy())
    assert rel(ph1, po1) <= 1e-10
    # <pt, H pt> = nu (logarithmic homogeneity), on the HIP side alone
    hp = np.zeros(dim)
    hc.hess_prod(hp, np.array(hc.point))
    assert abs(hp @ hc.point - hc.get_nu()) <= 1e-9 * hc.get_nu()
    d = V[:, 1].copy() * 0.01
    d3h, d3o = np.array(hc.dder3(d)), np.array(oc.dder3(d))
    assert rel(d3h, d3o) <= 1e-10
    assert abs(d3h[0] - d3o[0]) <= 1e-9 * (abs(d3o[0]) + 1e-300)
    # an infeasible point of the same size: sigma_1(W) slightly above u
    bad = np.concatenate([[0.5 * (1 - 1e-9)], Wm.reshape(-1, order="F")])
    for c in (hc, oc):
        c.reset_data()
        c.load_point(bad)
    assert not oc.is_feas()
    assert not hc.is_feas()


def _polymin_data(seed=1):
    from oracle import polyutils as pu
    rng = np.random.default_rng(seed)
    U, pts, Ps = pu.interpolate_box([-1.0] * 4, [1.0] * 4, 8, rng=rng, sample_factor=2)
    assert U == 4845 and [P.shape[1] for P in Ps] == [495, 330, 330, 330, 330]
    a = rng.uniform(-0.5, 0.5, 4)
    vals = np.sum((pts - a) ** 2, axis=1) + (pts[:, 0] * pts[:, 1] - pts[:, 2] * pts[:, 3]) ** 2 + 0.3 * pts[:, 0] * pts[:, 2]
    return U, pts, Ps, vals, rng


@pytest.mark.timeout(1800)
def test_config5_wsos_u4845_oracle_by_oracle():
    import hypatia_jl_amd as H
    from oracle import cones as OC
    U, pts, Ps, vals, rng = _polymin_data()
    hc, oc = H.WSOSInterpNonnegative(U, Ps), OC.WSOSInterpNonnegative(U, Ps)
    assert hc.get_nu() == oc.get_nu() == 495 + 4 * 330 and hc.use_dual_barrier() == oc.use_dual_barrier() == True   # noqa: E712
    for c in (hc, oc):
        c.setup_data()
    pt = np.ones(U) + 0.1 * (2 * rng.random(U) - 1)
    for c in (hc, oc):
        c.reset_data()
        c.load_point(pt, 0.8)
        assert c.is_feas()
    g_h, g_o = np.array(hc.get_grad()), np.array(oc.get_grad())
    assert rel(g_h, g_o) <= 1e-10
    assert abs(g_h @ hc.point + hc.get_nu()) <= 1e-9 * hc.get_nu()
    V = np.asfortranarray(rng.standard_normal((U, 3)))
    for name, tol in (("hess_prod", 1e-9), ("inv_hess_prod", 1e-8)):
        Ph, Po = np.zeros((U, 3), order="F"), np.zeros((U, 3), order="F")
        getattr(hc, name)(Ph, V)
        getattr(oc, name)(Po, V)
        assert rel(Ph, Po) <= tol, (name, rel(Ph, Po))
    # explicit Hessian entries on a sample of rows (a consistently wrong Hessian would pass H^-1 H = I)
    Ho = oc.hess()
    rows = rng.choice(U, 8, replace=False)
    E = np.zeros((U, len(rows)), order="F")
    E[rows, np.arange(len(rows))] = 1.0
    Hcols = np.zeros_like(E)
    hc.hess_prod(Hcols, E)
    Hs = np.triu(Ho) + np.triu(Ho, 1).T
    assert rel(Hcols, Hs[:, rows]) <= 1e-10
    # the operator form (hess_prod_slow!, :152-175), forced on both sides
    hc.use_hess_prod_slow = True
    oc.use_hess_prod_slow = True
    oc.use_hess_prod_slow_updated = True
    Ph, Po = np.zeros((U, 2), order="F"), np.zeros((U, 2), order="F")
    hc.hess_prod_slow(Ph, V[:, :2])
    oc.hess_prod_slow(Po, np.asfortranarray(V[:, :2]))
    assert rel(Ph, Po) <= 1e-9
    d = V[:, 2].copy() * 0.05
    d3h, d3o = np.array(hc.dder3(d)), np.array(oc.dder3(d))
    assert rel(d3h, d3o) <= 1e-9
    # an infeasible point (negative weights)
    bad = pt.copy()
    bad[::7] = -0.5
    for c in (hc, oc):
        c.reset_data()
        c.load_point(bad)
    assert not oc.is_feas()
    assert not hc.is_feas()


@pytest.mark.timeout(1800)
def test_config5_polymin_dual_full_size():
    """polymin in dual form (examples/polymin/native.jl, use_primal = false): c = vals, A = ones(1, U), b = [1], G = -I,
    h = 0, WSOSInterpNonnegative(U, Ps, use_dual = true); after the reduction n = U - 1 = 4844 and G Q2 is dense."""
    import hypatia_jl_amd as H
    U, pts, Ps, vals, rng = _polymin_data()
    inst = (vals, np.ones((1, U)), np.array([1.0]), -np.eye(U), np.zeros(U), [("wsosinterpnonnegative", U, Ps, True)], {})
    # the reference's default tolerances stop at tol_feas = sqrt(eps); the certificate bar below is tighter than that,
    # so the solve is asked for it (Solvers.jl:173, 190-214)
    s = H.Solver(verbose=False, tol_feas=1e-10, tol_rel_opt=1e-9, tol_abs_opt=1e-10)
    s.load(H.make_model(inst))
    s.solve()
    assert s.status == "Optimal", s.status
    assert s.model.n - s.model.p == U - 1                                 # reduced: 4844 unknowns in the Schur system
    c, A, b, G, h = inst[:5]
    x, y, z, sv = s.get_x(), s.get_y(), s.get_z(), s.get_s()
    r = lambda a, bb: np.linalg.norm(a - bb) / (1 + np.linalg.norm(bb))
    assert r(G @ x + sv, h) <= 1e-8, r(G @ x + sv, h)
    assert r(G.T @ z + A.T @ y, -c) <= 1e-8
    assert r(A @ x, b) <= 1e-8
    assert abs(s.primal_obj - s.dual_obj) <= 1e-7 * (1 + abs(s.primal_obj))
    assert abs(sv @ z) <= 1e-6 * (1 + abs(s.primal_obj))
    # the dual form's optimum is the primal form's lower bound of the sampled polynomial
    assert s.primal_obj <= vals.min() + 1e-6
    # x is a probability-like measure on the interpolation points: nonnegative weights of the moment functional summing to one
    assert abs(np.sum(x) - 1.0) <= 1e-8


@pytest.mark.timeout(1200)
def test_config2_initial_x_device_shortcut_matches_pivoted_qr():
    """process.jl:64-178 at config-2 size (q = 20100, n = 5000): init_x = AG_fact \\ (h - init_s) from the column-pivoted QR
    on the host against the device normal-equations path the driver takes by default at this size."""
    import ctypes
    import scipy.linalg as sla
    import hypatia_jl_amd as H
    from hypatia_jl_amd import _lib as L
    from oracle import instances as I
    inst = I.psd_blocks(5000, [200], seed=3)
    G, h = inst[3], inst[4]
    q, n = G.shape
    init_s = np.zeros(q)
    H.PosSemidefTri(q).set_initial_point(init_s)
    rhs = h - init_s
    AGf = np.asfortranarray(G)
    xs, rc, info = np.zeros(n), ctypes.c_double(0.0), ctypes.c_int(-1)
    L.check(L.lib().hyp_dense_lstsq_normal(L.ctx(), q, n, AGf.ctypes.data_as(ctypes.c_void_p), q, L.vec_ptr(np.ascontiguousarray(rhs)),
                                           L.vec_ptr(xs), ctypes.byref(rc), ctypes.byref(info)), "hyp_dense_lstsq_normal")
    assert info.value == 0 and rc.value > 1e-3
    Qf, R, piv = sla.qr(G.copy(), mode="economic", pivoting=True, overwrite_a=True)
    assert int(np.sum(np.abs(np.diagonal(R)) > 1000 * np.finfo(float).eps)) == n     # get_rank_est (process.jl:373-382)
    xq = np.zeros(n)
    xq[piv] = sla.solve_triangular(R, Qf.T @ rhs, lower=False)
    assert rel(xs, xq) <= 1e-10, rel(xs, xq)
    # both are least-squares solutions: the normal-equations residual G'(G x - rhs) vanishes
    for xx in (xs, xq):
        assert np.linalg.norm(G.T @ (G @ xx - rhs)) <= 1e-9 * np.linalg.norm(G.T @ rhs)
    # a nearly rank-deficient matrix must be refused by the gate (the caller then runs the reference's QR)
    Gd = AGf[:4000, :1500].copy(order="F")
    Gd[:, 7] = Gd[:, 3] * (1 + 1e-13) + 1e-13 * Gd[:, 5]
    xs2 = np.zeros(1500)
    L.check(L.lib().hyp_dense_lstsq_normal(L.ctx(), 4000, 1500, Gd.ctypes.data_as(ctypes.c_void_p), 4000, L.vec_ptr(np.ascontiguousarray(rhs[:4000])),
                                           L.vec_ptr(xs2), ctypes.byref(rc), ctypes.byref(info)), "hyp_dense_lstsq_normal")
    assert info.value != 0 or rc.value <= 1e-3
