"""GPU parity of the HIP cones: (1) the reference's own oracle identities (test/cone.jl:23-114) run on
the HIP cone through the C-ABI, (2) oracle-vs-HIP agreement of every oracle on the same seeded inputs."""
import numpy as np
import pytest

from cone_harness import run_test_oracles

pytestmark = pytest.mark.gpu

TOL = 1e-11   # relative, per-oracle parity target (<= 1e-12 .. 1e-11 at these sizes)


def rel(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-300)


@pytest.mark.parametrize("d", [1, 2, 6, 300])
def test_nonnegative_identities(d):
    import hypatia_jl_amd as H
    run_test_oracles(H.Nonnegative(d), explicit_hess=(d <= 6))


@pytest.mark.parametrize("side", [1, 2, 3, 5, 12])
def test_possemideftri_identities(side):
    import hypatia_jl_amd as H
    run_test_oracles(H.PosSemidefTri(side * (side + 1) // 2), tol=1e4 * np.finfo(float).eps)


@pytest.mark.parametrize("side", [40, 130, 200])
def test_possemideftri_identities_matrix_free(side):
    import hypatia_jl_amd as H
    # (the reference's noise = 0.1 makes a side >= 150 matrix indefinite: scale it with 1/sqrt(side))
    run_test_oracles(H.PosSemidefTri(side * (side + 1) // 2), explicit_hess=False, tol=1e-9, noise=0.5 / np.sqrt(side))


def _pair(kind, *args):
    import hypatia_jl_amd as H
    from oracle import cones as oc
    if kind == "psd":
        return H.PosSemidefTri(*args), oc.PosSemidefTri(*args)
    return H.Nonnegative(*args), oc.Nonnegative(*args)


@pytest.mark.parametrize("kind,dim", [("psd", 6), ("psd", 210), ("psd", 8385), ("psd", 20100), ("nonneg", 50)])
def test_oracle_vs_hip(kind, dim):
    hc, oc = _pair(kind, dim)
    rng = np.random.default_rng(dim)
    for c in (hc, oc):
        c.setup_data()
    pt = np.zeros(dim)
    oc.set_initial_point(pt)
    pt2 = np.zeros(dim)
    hc.set_initial_point(pt2)
    assert np.array_equal(pt, pt2)
    pt += 0.1 * (2 * rng.random(dim) - 1) / np.sqrt(max(1, dim / 50))
    pt *= 0.3
    dual = pt.copy() + 0.05 * (2 * rng.random(dim) - 1) / np.sqrt(max(1, dim / 50))
    for c in (hc, oc):
        c.reset_data()
        c.load_point(pt, 1.7)
        c.load_dual_point(dual)
        assert c.is_feas()
        assert c.is_dual_feas()
    assert np.array_equal(hc.point, oc.point)
    g_h, g_o = np.array(hc.get_grad()), np.array(oc.get_grad())
    assert rel(g_h, g_o) < TOL
    ncols = 3
    V = np.asfortranarray(rng.standard_normal((dim, ncols)))
    for name in ("hess_prod", "inv_hess_prod", "sqrt_hess_prod", "inv_sqrt_hess_prod", "hess_prod_slow"):
        Ph = np.zeros((dim, ncols), order="F")
        Po = np.zeros((dim, ncols), order="F")
        getattr(hc, name)(Ph, V)
        getattr(oc, name)(Po, V)
        assert rel(Ph, Po) < TOL * 10, name
    # strided views (rows of a taller matrix), as the system solver passes them (qrchol.jl:162-165)
    big_in = np.asfortranarray(rng.standard_normal((dim + 5, ncols)))
    big_out_h = np.full((dim + 7, ncols), 7.0, order="F")
    big_out_o = np.full((dim + 7, ncols), 7.0, order="F")
    hc.hess_prod(big_out_h[3:3 + dim, :], big_in[2:2 + dim, :])
    oc.hess_prod(big_out_o[3:3 + dim, :], big_in[2:2 + dim, :])
    assert rel(big_out_h, big_out_o) < TOL * 10
    assert np.all(big_out_h[:3] == 7.0) and np.all(big_out_h[3 + dim:] == 7.0)
    d3h = np.array(hc.dder3(V[:, 0].copy()))
    d3o = np.array(oc.dder3(V[:, 0].copy()))
    assert rel(d3h, d3o) < TOL * 10
    assert hc.check_numerics() == oc.check_numerics()
    ph, po = hc.get_proxsqr(0.9, True), oc.get_proxsqr(0.9, True)
    assert abs(ph - po) <= 1e-9 * max(1.0, abs(po))


def test_infeasible_points_detected():
    import hypatia_jl_amd as H
    c = H.PosSemidefTri(6)
    c.load_point(np.array([1.0, 3.0, 1.0, 0.0, 0.0, 1.0]))   # [[1, 3/rt2], [3/rt2, 1]] indefinite
    c.reset_data()
    assert not c.is_feas()
    c.load_dual_point(np.array([1.0, 0.0, -1.0, 0.0, 0.0, 1.0]))
    assert not c.is_dual_feas()
    n = H.Nonnegative(4)
    n.load_point(np.array([1.0, 0.0, 1.0, 2.0]))
    n.reset_data()
    assert not n.is_feas()
