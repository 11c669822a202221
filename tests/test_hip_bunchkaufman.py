"""GPU parity of the device Bunch-Kaufman (rook) factorization and of the factorization chain
Cholesky -> Bunch-Kaufman -> diagonal shift + Bunch-Kaufman (posdef_fact_copy!, src/linearalgebra/dense.jl:194-215;
symm_fact!, :164-165) at its two call sites: the Schur matrix of the QRChol solver (qrchol.jl:249-250) and the explicit
Hessian of the generic cones (Cones.jl:239-251).  The comparison point is LAPACK dsytrf_rook -- the routine the
reference calls -- run forwards (uplo = 'L'): same pivot sequence, same D blocks, same multipliers."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

c_int, c_vp = ctypes.c_int, ctypes.c_void_p


@pytest.fixture(scope="module")
def hip():
    import hypatia_jl_amd as H
    return H._lib.lib(), H._lib.ctx(), H._lib


def fp(a):
    return a.ctypes.data_as(c_vp)


def device_sysv(hip, A, B):
    lib, ctx, L = hip
    n = A.shape[0]
    Ad = np.asfortranarray(np.triu(A) + np.tril(np.full((n, n), 7.5), -1))   # the strict lower triangle must not be read
    X = np.asfortranarray(B.reshape(n, -1).copy())
    nrhs = X.shape[1]
    perm, blk = np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32)
    d, e = np.zeros(n), np.zeros(n)
    info = c_int(-1)
    L.check(lib.hyp_dense_sysv_rook(ctx, n, fp(Ad), n, fp(X), nrhs, n, ctypes.byref(info), fp(perm), fp(blk), fp(d), fp(e)), "sysv_rook")
    return np.triu(Ad), X, info.value, perm, blk, d, e


def cases(n, rng):
    M = rng.standard_normal((n, n))
    A = M + M.T
    yield "indefinite", A
    B = A.copy()
    np.fill_diagonal(B, 0.0)
    yield "zero diagonal", B
    yield "posdef", M @ M.T + 0.1 * np.eye(n)


@pytest.mark.parametrize("n", [1, 2, 3, 5, 17, 64, 127, 128, 129, 300, 1100])
def test_device_rook_factorization_matches_lapack(hip, n):
    from oracle import linalg as la
    rng = np.random.default_rng(n)
    for name, A in cases(n, rng):
        B = rng.standard_normal((n, 3))
        U, X, info, perm, blk, d, e = device_sysv(hip, A, B)
        a, ipiv, linfo = la.sytrf_rook_lapack(A, "L")
        lperm, lblk, ld, le, Lf = la.decode_rook_lower(a, ipiv)
        assert info == linfo, name
        if linfo != 0:                                           # (n = 1 with a zero diagonal)
            continue
        assert np.array_equal(perm, lperm), name                 # identical pivot choices, step by step
        assert np.array_equal(blk, lblk), name
        scale = np.abs(A).max()
        assert np.abs(d - ld).max() <= 1e-10 * scale and np.abs(e - le).max() <= 1e-10 * scale, name
        assert np.abs(U - Lf.T).max() <= 1e-9, name              # U = L' (unit diagonal explicit, 2x2 off-diagonals zero)
        # P A P' = U' D U to rounding
        D = np.diag(d)
        for k in range(n - 1):
            D[k, k + 1] = D[k + 1, k] = e[k]
        assert np.abs(U.T @ D @ U - A[np.ix_(perm, perm)]).max() <= 1e-13 * max(n, 8) * scale, name
        # solve: backward error of dsytrs_rook's class
        berr = np.linalg.norm(A @ X - B) / (np.linalg.norm(A, 2) * np.linalg.norm(X) + np.linalg.norm(B))
        assert berr <= 1e-14 * max(n, 8), (name, berr)
        Xo = la.bk_rook(A).solve(B)                              # the oracle's uplo = 'U' path (what Julia runs)
        assert np.linalg.norm(X - Xo) <= 1e-11 * np.linalg.cond(A) * np.linalg.norm(Xo), name


def test_device_rook_reports_singular_pivot(hip):
    from oracle import linalg as la
    for A in (np.zeros((4, 4)), np.diag([1.0, 0.0, 2.0]), np.array([[1.0, 1.0], [1.0, 1.0]])):
        _, _, info, *_ = device_sysv(hip, A, np.ones(A.shape[0]))
        assert info == la.sytrf_rook_lapack(A, "L")[2] and info > 0


def _solver(n, sides, iters=2):
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    import hypatia_jl_amd as H
    s = H.Solver(verbose=False)
    s.load(H.make_model(bench.gen_instance(n, sides, 1)))
    s.setup()
    for _ in range(iters):
        assert s.iterate()
    return s


@pytest.mark.parametrize("n,sides", [(150, [20]), (2100, [70])])
def test_schur_factorization_chain(hip, n, sides):
    """An indefinite / singular matrix in place of the Schur matrix walks the chain of dense.jl:194-215 on the device;
    n = 2100 takes the super-block triangular solves (TriSolvePlan) around the block-diagonal solve."""
    lib, ctx, L = hip
    from oracle import linalg as la
    s = _solver(n, sides, iters=1)
    sysv = s.syssolver
    rng = np.random.default_rng(n)
    M = rng.standard_normal((n, n))
    A = np.asfortranarray(M + M.T)
    info, fb = c_int(-1), c_int(-1)
    L.check(lib.hyp_sys_set_lhs(sysv._h, fp(A)), "set_lhs")
    L.check(lib.hyp_sys_factor_lhs(sysv._h, ctypes.byref(info), ctypes.byref(fb)), "factor_lhs")
    assert (info.value, fb.value) == (0, 1)                     # Cholesky failed -> Bunch-Kaufman
    assert la.posdef_fact_copy(A).kind == "bk"
    b = rng.standard_normal(n)
    x = b.copy()
    L.check(lib.hyp_sys_potrs(sysv._h, L.vec_ptr(x)), "potrs")
    xo = la.posdef_fact_copy(A).solve(b)
    berr = lambda v: np.linalg.norm(A @ v - b) / (np.linalg.norm(A, 2) * np.linalg.norm(v) + np.linalg.norm(b))
    assert berr(x) <= 4 * berr(xo) + 1e-15, (berr(x), berr(xo))
    assert np.linalg.norm(x - xo) <= 1e-11 * np.linalg.cond(A) * np.linalg.norm(xo)
    # exactly singular: Bunch-Kaufman reports a zero pivot -> increase_diag! -> Bunch-Kaufman
    Z = np.zeros((n, n), order="F")
    L.check(lib.hyp_sys_set_lhs(sysv._h, fp(Z)), "set_lhs")
    L.check(lib.hyp_sys_factor_lhs(sysv._h, ctypes.byref(info), ctypes.byref(fb)), "factor_lhs")
    assert (info.value, fb.value) == (0, 2)
    x = b.copy()
    L.check(lib.hyp_sys_potrs(sysv._h, L.vec_ptr(x)), "potrs")
    assert np.allclose(x, la.posdef_fact_copy(Z).solve(b), rtol=1e-12)
    # a positive definite matrix stays on the Cholesky
    P = np.asfortranarray(M @ M.T + np.eye(n))
    L.check(lib.hyp_sys_set_lhs(sysv._h, fp(P)), "set_lhs")
    L.check(lib.hyp_sys_factor_lhs(sysv._h, ctypes.byref(info), ctypes.byref(fb)), "factor_lhs")
    assert (info.value, fb.value) == (0, 0)


@pytest.mark.parametrize("name", ["possemideftri2", "epinormspectral2_primal", "wsosinterpnonnegative2"])
def test_whole_solve_through_bunchkaufman(monkeypatch, name):
    """HYP_FORCE_BK=1 sends every factorization (Schur matrix and generic cone Hessians) down the Bunch-Kaufman branch: on
    these positive definite matrices it is a second, independent factorization of the same systems, so the solve must
    reach the same answer in the same number of iterations as the Cholesky path (and the oracle's known answer)."""
    import hypatia_jl_amd as H
    from instance_harness import build_solve_check
    from oracle import instances as I
    inst = I.KNOWN_ANSWER[name]()
    ref = build_solve_check(H.Solver(default_tol_relax=10), H.make_model(inst), inst)
    assert ref.syssolver.fallback_kind == 0
    monkeypatch.setenv("HYP_FORCE_BK", "1")
    got = build_solve_check(H.Solver(default_tol_relax=10), H.make_model(inst), inst)
    assert got.syssolver.fallback_kind == 1
    assert got.status == ref.status and abs(got.num_iters - ref.num_iters) <= 1
    # (two convergent runs stop within the solver's own tolerance of the optimum -- tol_rel_opt = 10 sqrt(eps) = 1.5e-7 here --
    #  not within rounding of each other: their last iterates differ by the factorizations' rounding)
    assert abs(got.primal_obj - ref.primal_obj) <= 1e-6 * (1 + abs(ref.primal_obj))


@pytest.mark.parametrize("kind,args", [("ens", (3, 4)), ("ens", (20, 33)), ("wsos", (2, 10))])
def test_generic_cone_inverse_hessian_through_bunchkaufman(monkeypatch, kind, args):
    """Cones.jl:113-118 with hess_fact::BunchKaufman (the branch after a failed Cholesky of the explicit Hessian): same
    products as the Cholesky branch and as the oracle; use_sqrt_hess_oracles is false for it (Cones.jl:189-195)."""
    from test_hip_cones import _generic_pair, rel
    monkeypatch.setenv("HYP_FORCE_BK", "1")
    hc, oc = _generic_pair(kind, *args)
    dim = hc.dimension()
    rng = np.random.default_rng(dim)
    pt = np.zeros(dim)
    oc.set_initial_point(pt)
    scale = 0.1 / np.sqrt(max(1.0, dim / 20.0))
    pt = pt + scale * (2 * rng.random(dim) - 1)
    dual = pt + 0.3 * scale * (2 * rng.random(dim) - 1)
    for c in (hc, oc):
        c.setup_data()
        c.reset_data()
        c.load_point(pt, 0.7)
        c.load_dual_point(dual)
        assert c.is_feas()
        c.get_grad()
    for ncols in (1, 3):
        V = np.asfortranarray(rng.standard_normal((dim, ncols)))
        Ph = np.zeros((dim, ncols), order="F")
        Po = np.zeros((dim, ncols), order="F")
        hc.inv_hess_prod(Ph, V)
        oc.inv_hess_prod(Po, V)
        assert rel(Ph, Po) < 1e-8
    assert not hc.use_sqrt_hess_oracles(dim)                     # hess_fact is not a Cholesky
    assert hc.check_numerics() == oc.check_numerics()
    ph, po = hc.get_proxsqr(0.9, True), oc.get_proxsqr(0.9, True)
    assert abs(ph - po) <= 1e-7 * max(1.0, abs(po))


def _late_failure_matrix(n, nbad, rng, cond=1e8):
    """symmetric, positive definite on its leading n - nbad columns, with nbad negative eigen-directions that a Cholesky meets at its
    LAST pivots -- the shape of the matrices that reach symm_fact! in the solver (cone Hessians at the end of a solve)"""
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    lam = np.logspace(0, -np.log10(cond), n)
    A = (Q * lam) @ Q.T
    A = 0.5 * (A + A.T)
    L = np.linalg.cholesky(A)
    D = np.ones(n)
    D[n - nbad:] = -1.0                      # A' = L D L': the same elimination, the last nbad pivots negative
    Ab = (L * D) @ L.T
    return 0.5 * (Ab + Ab.T)


@pytest.mark.parametrize("n,nbad", [(300, 3), (700, 5), (1153, 1), (1153, 40), (2250, 2), (200, 4), (640, 1), (256, 128), (384, 129), (385, 1), (129, 1)])
def test_hybrid_factorization_behind_a_late_cholesky_failure(hip, n, nbad):
    """round 4: behind a failed Cholesky the block steps in front of the failing pivot's block are kept and only the trailing block
    goes through the rook-pivoted elimination (BKFact::factor_from).  Against numpy's solve: LAPACK's backward error; bk_start says
    the hybrid path ran (failure beyond the first 128 columns) or not (n = 200: one block, and with HYP_BK_HYBRID=0 always 0)."""
    lib, ctx, L = hip
    rng = np.random.default_rng(n + nbad)
    A = _late_failure_matrix(n, nbad, rng)
    B = rng.standard_normal((n, 3))
    Ad = np.asfortranarray(np.triu(A) + np.tril(np.full((n, n), 7.5), -1))   # the strict lower triangle must not be read
    X = np.asfortranarray(B.copy())
    info, fb, start = c_int(-1), c_int(-1), c_int(-1)
    L.check(lib.hyp_dense_posdef_solve(ctx, n, fp(Ad), n, fp(X), 3, n, ctypes.byref(info), ctypes.byref(fb), ctypes.byref(start)), "posdef_solve")
    assert info.value == 0 and fb.value == 1
    fail_col = n - nbad                                    # 0-based column of the first negative pivot
    assert start.value == (fail_col // 128) * 128
    Xref = np.linalg.solve(A, B)
    berr = lambda Xc: np.linalg.norm(A @ Xc - B) / (np.linalg.norm(A, 2) * np.linalg.norm(Xc) + np.linalg.norm(B))
    assert berr(X) <= 10 * berr(Xref) + 1e-15, (berr(X), berr(Xref))
    assert np.linalg.norm(X - Xref) <= 1e-6 * np.linalg.norm(Xref)          # (condition 1e8)
    # the plain rook-pivoted factorization of the same matrix solves the same system
    _, X2, info2, *_ = device_sysv(hip, A, B)
    assert info2 == 0 and np.linalg.norm(X - X2) <= 1e-6 * np.linalg.norm(Xref)


def _growth_matrix(n, k, rng, eps=1e-13):
    """A = Lc Mid Lc' (Lc unit lower triangular, mild): the elimination meets pivots 1, ..., 1, eps (column k), 1, ..., 1 and at the
    very last column -1/eps: the Cholesky runs to its last pivot, yet row k of its factor has an entry 1/sqrt(eps) -- the
    [[eps, 1], [1, 0]] example of why an unpivoted elimination has no growth bound, inside a block the hybrid path would keep"""
    Mid = np.eye(n)
    Mid[k, k] = eps
    Mid[n - 1, n - 1] = 0.0
    Mid[k, n - 1] = Mid[n - 1, k] = 1.0
    Lc = np.tril(0.3 * rng.standard_normal((n, n)) / np.sqrt(n), -1) + np.eye(n)
    A = Lc @ Mid @ Lc.T
    return 0.5 * (A + A.T)


@pytest.mark.parametrize("n,k,start_expected", [(400, 200, 128), (400, 50, 0), (700, 300, 256), (1153, 1100, 1024)])
def test_hybrid_guard_refuses_steps_with_element_growth(hip, n, k, start_expected):
    """round 5: a leading pivot that is positive only just (1e-13 of the diagonal) inside a block step the hybrid path
    would keep unpivoted.  The growth guard of bk_after_failed_cholesky (pivot^2 >= n eps max|a_ii|, row entries^2 <= 16 max|a_ii|)
    keeps only the block steps in front of it -- bk_start says so -- and the solve has LAPACK's backward error: the matrix itself is
    perfectly conditioned (cond 3.3), it is the unpivoted elimination that would lose 13 digits (row k of the factor reaches 1/sqrt(eps));
    dsytrf_rook's growth is bounded (reference: symm_fact!, src/linearalgebra/dense.jl:164-165, 194-215)"""
    lib, ctx, L = hip
    rng = np.random.default_rng(n + k)
    A = _growth_matrix(n, k, rng)
    B = rng.standard_normal((n, 2))
    Ad = np.asfortranarray(np.triu(A) + np.tril(np.full((n, n), -3.25), -1))
    X = np.asfortranarray(B.copy())
    st0 = (ctypes.c_longlong * 3)()
    L.check(lib.hyp_ctx_bk_stats(ctx, st0), "bk_stats")
    info, fb, start = c_int(-1), c_int(-1), c_int(-1)
    L.check(lib.hyp_dense_posdef_solve(ctx, n, fp(Ad), n, fp(X), 2, n, ctypes.byref(info), ctypes.byref(fb), ctypes.byref(start)), "posdef_solve")
    assert info.value == 0 and fb.value == 1
    assert start.value == start_expected, start.value        # (unguarded it would be ((n - 1) // 128) * 128)
    st1 = (ctypes.c_longlong * 3)()
    L.check(lib.hyp_ctx_bk_stats(ctx, st1), "bk_stats")
    assert st1[1] == st0[1] + 1                              # the guard trimmed this call
    assert (st1[0] - st0[0], st1[2] - st0[2]) == ((1, 0) if start_expected > 0 else (0, 1))
    Xref = np.linalg.solve(A, B)
    berr = lambda Xc: np.linalg.norm(A @ Xc - B) / (np.linalg.norm(A, 2) * np.linalg.norm(Xc) + np.linalg.norm(B))
    assert berr(X) <= 10 * berr(Xref) + 1e-15, (berr(X), berr(Xref))


def test_posdef_solve_cholesky_branch(hip):
    lib, ctx, L = hip
    rng = np.random.default_rng(3)
    n = 500
    M = rng.standard_normal((n, n + 5))
    A = M @ M.T + np.eye(n)
    b = rng.standard_normal(n)
    Ad, x = np.asfortranarray(A.copy()), b.copy()
    info, fb, start = c_int(-1), c_int(-1), c_int(-1)
    L.check(lib.hyp_dense_posdef_solve(ctx, n, fp(Ad), n, fp(x), 1, n, ctypes.byref(info), ctypes.byref(fb), ctypes.byref(start)), "posdef_solve")
    assert info.value == 0 and fb.value == 0 and start.value == 0
    assert np.linalg.norm(x - np.linalg.solve(A, b)) <= 1e-9 * np.linalg.norm(b)
