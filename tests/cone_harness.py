"""Cone oracle identity harness: restates test_oracles / test_barrier of the reference
(/root/reference/test/cone.jl:23-114, 117-160).  Implementation-independent, so the same harness
checks the CPU oracle cones and the HIP cones."""
import numpy as np

EPS = np.finfo(np.float64).eps


def perturb_scale(point, noise, scale, rng):   # test/cone.jl:236-249
    if noise != 0:
        point += 2 * noise * rng.random(point.shape[0]) - noise
    if scale != 1:
        point *= scale
    return point


def sym_from_upper(H):
    return np.triu(H) + np.triu(H, 1).T


def assert_close(a, b, tol, what=""):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    err = np.abs(a - b)
    bound = tol + tol * np.maximum(np.abs(a), np.abs(b))
    assert np.all(err <= bound), f"{what}: max err {err.max():.3e} (tol {tol:.1e})"


def run_test_oracles(cone, noise=0.1, scale=0.1, tol=1e3 * EPS, init_only=False, init_tol=None, seed=1,
                     explicit_hess=True):
    if init_tol is None:
        init_tol = tol
    rng = np.random.default_rng(seed)
    dim = cone.dimension()
    cone.setup_data()
    cone.reset_data()

    point = np.zeros(dim)
    cone.set_initial_point(point)
    cone.load_point(point)
    assert cone.is_feas()
    assert np.array_equal(np.asarray(cone.point), point)

    dual_point = -np.array(cone.get_grad())
    cone.load_dual_point(dual_point)
    assert cone.is_dual_feas()
    assert cone.get_proxsqr(1.0, True) <= 1
    assert cone.get_proxsqr(1.0, False) <= dim

    prod_vec = np.zeros(dim)
    assert_close(cone.hess_prod(prod_vec, point), dual_point, tol, "H*point = -grad at init")
    if np.isfinite(init_tol):
        assert_close(point, dual_point, init_tol, "centrality of initial point")
    if init_only:
        return

    perturb_scale(point, noise, scale, rng)
    perturb_scale(dual_point, noise, 1.0 / scale, rng)
    cone.reset_data()
    cone.load_point(point)
    assert cone.is_feas()
    cone.load_dual_point(dual_point)
    assert cone.is_dual_feas()

    nu = cone.get_nu()
    grad = np.array(cone.get_grad())
    assert_close(point @ grad, -nu, tol, "<point, grad> = -nu")

    assert_close(cone.hess_prod(prod_vec, point), -grad, tol, "hess_prod(point) = -grad")
    assert_close(cone.inv_hess_prod(prod_vec, grad), -point, tol, "inv_hess_prod(grad) = -point")

    if explicit_hess:
        hess = sym_from_upper(np.array(cone.hess()))
        inv_hess = sym_from_upper(np.array(cone.inv_hess()))
        assert_close(hess @ inv_hess, np.eye(dim), tol, "H * Hinv = I")
        assert_close(hess @ point, -grad, tol, "H * point = -grad")
        prod_mat = np.zeros((dim, dim), order="F")
        assert_close(cone.hess_prod(prod_mat, np.asfortranarray(inv_hess)), np.eye(dim), tol, "hess_prod(Hinv) = I")
        assert_close(cone.inv_hess_prod(prod_mat, np.asfortranarray(hess)), np.eye(dim), tol, "inv_hess_prod(H) = I")
    else:
        # matrix-free variants of the same identities
        V = np.asfortranarray(rng.standard_normal((dim, 3)))
        T = np.zeros_like(V)
        R = np.zeros_like(V)
        cone.hess_prod(T, V)
        cone.inv_hess_prod(R, T)
        assert_close(R, V, tol * 10, "inv_hess_prod(hess_prod(V)) = V")

    psi = dual_point + grad
    proxsqr = psi @ cone.inv_hess_prod(prod_vec, psi)
    assert_close(cone.get_proxsqr(1.0, False), proxsqr, tol, "proxsqr identity")

    if hasattr(cone, "use_hess_prod_slow") and explicit_hess:
        cone.update_use_hess_prod_slow()
        assert cone.use_hess_prod_slow_updated
        assert not cone.use_hess_prod_slow
        cone.use_hess_prod_slow = True
        prod_mat = np.zeros((dim, dim), order="F")
        assert_close(cone.hess_prod_slow(prod_mat, np.asfortranarray(inv_hess)), np.eye(dim), tol, "hess_prod_slow(Hinv) = I")

    if cone.use_sqrt_hess_oracles(dim + 1):
        if explicit_hess:
            prod_mat = np.zeros((dim, dim), order="F")
            pm2 = np.asfortranarray(np.array(cone.sqrt_hess_prod(prod_mat, np.asfortranarray(inv_hess))).T)
            assert_close(cone.sqrt_hess_prod(prod_mat, pm2), np.eye(dim), tol, "sqrt_hess_prod identity")
            pm2 = np.zeros((dim, dim), order="F")
            cone.inv_sqrt_hess_prod(pm2, np.asfortranarray(np.eye(dim)))
            assert_close(pm2.T @ pm2, inv_hess, tol, "inv_sqrt_hess_prod identity")
        else:
            V = np.asfortranarray(rng.standard_normal((dim, 3)))
            S = np.zeros_like(V)
            T = np.zeros_like(V)
            cone.sqrt_hess_prod(S, V)
            cone.hess_prod(T, V)
            assert_close(S.T @ S, V.T @ T, tol * 10, "sqrt' sqrt = H (quadratic form)")

    if cone.use_dder3():
        assert_close(-np.array(cone.dder3(point)), grad, tol, "dder3(point) = -grad")
        dirv = perturb_scale(np.zeros(dim), noise, 1.0, rng)
        d3 = np.array(cone.dder3(dirv))
        Hd = np.zeros(dim)
        cone.hess_prod(Hd, dirv)
        assert_close(d3 @ point, dirv @ Hd, tol, "<dder3(dir), point> = dir' H dir")


def run_test_barrier(cone, barrier, noise=0.1, scale=0.1, seed=1, tol=2e-5):
    """finite-difference version of test_barrier (test/cone.jl:117-160): the reference uses
    ForwardDiff; central differences of the closed-form barrier stand in here."""
    rng = np.random.default_rng(seed)
    dim = cone.dimension()
    cone.setup_data()
    point = np.zeros(dim)
    cone.set_initial_point(point)
    perturb_scale(point, noise, scale, rng)
    cone.reset_data()
    cone.load_point(point)
    assert cone.is_feas()
    grad = np.array(cone.get_grad())

    def fd_grad(f, x, h):
        g = np.zeros_like(x)
        for i in range(x.shape[0]):
            e = np.zeros_like(x)
            e[i] = h
            g[i] = (f(x + e) - f(x - e)) / (2 * h)
        return g

    h = 1e-6 * scale
    assert_close(grad, fd_grad(barrier, point, h), tol * max(1.0, np.abs(grad).max()), "grad vs FD")

    dirv = rng.standard_normal(dim) * scale
    Hd = np.zeros(dim)
    cone.hess_prod(Hd, dirv)

    def grad_at(x):
        cone.reset_data()
        cone.load_point(x)
        assert cone.is_feas()
        return np.array(cone.get_grad())

    t = 1e-5
    fd_hd = (grad_at(point + t * dirv) - grad_at(point - t * dirv)) / (2 * t)
    assert_close(Hd, fd_hd, tol * max(1.0, np.abs(Hd).max()), "hess_prod vs FD of grad")

    def hd_at(x):
        cone.reset_data()
        cone.load_point(x)
        assert cone.is_feas()
        cone.get_grad()
        out = np.zeros(dim)
        cone.hess_prod(out, dirv)
        return out.copy()

    t = 1e-4
    fd_third = (hd_at(point + t * dirv) - hd_at(point - t * dirv)) / (2 * t)
    cone.reset_data()
    cone.load_point(point)
    assert cone.is_feas()
    cone.get_grad()
    cone.update_hess_aux()
    d3 = np.array(cone.dder3(dirv))
    assert_close(-2 * d3, fd_third, 1e-4 * max(1.0, np.abs(d3).max()), "-2 dder3 vs FD third derivative")
