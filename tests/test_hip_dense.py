"""GPU parity of the dense kernels (through the C-ABI) against numpy/LAPACK."""
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


@pytest.mark.parametrize("transa,M,N,K,upper", [(1, 300, 200, 100, 0), (0, 131, 77, 45, 0), (1, 260, 260, 513, 1), (1, 5, 3, 2, 0)])
def test_gemm(hip, transa, M, N, K, upper):
    lib, ctx, L = hip
    rng = np.random.default_rng(0)
    A = np.asfortranarray(rng.standard_normal((K, M) if transa else (M, K)))
    B = np.asfortranarray(rng.standard_normal((K, N)))
    C = np.asfortranarray(rng.standard_normal((M, N)))
    C0 = C.copy()
    L.check(lib.hyp_dense_gemm(ctx, transa, upper, M, N, K, 0.7, fp(A), A.shape[0], fp(B), K, -0.3, fp(C), M), "gemm")
    ref = 0.7 * ((A.T if transa else A) @ B) - 0.3 * C0
    if upper:
        iu = np.triu_indices(M)
        assert np.allclose(C[iu], ref[iu], rtol=1e-13, atol=1e-12)
        il = np.tril_indices(M, -1)
        assert np.array_equal(C[il], C0[il])
    else:
        assert np.allclose(C, ref, rtol=1e-13, atol=1e-12)


@pytest.mark.parametrize("n", [1, 7, 128, 129, 200, 515])
def test_potrf_and_posv(hip, n):
    lib, ctx, L = hip
    rng = np.random.default_rng(n)
    M = rng.standard_normal((n, n + 3))
    A = np.asfortranarray(M @ M.T + 0.5 * np.eye(n))
    b = rng.standard_normal(n)
    Uref = np.linalg.cholesky(A).T
    Ad = A.copy(order="F")
    info = c_int(-1)
    L.check(lib.hyp_dense_potrf(ctx, n, fp(Ad), n, ctypes.byref(info)), "potrf")
    assert info.value == 0
    assert np.allclose(np.triu(Ad), Uref, rtol=1e-11, atol=1e-12)
    # strictly lower triangle untouched (dpotrf 'U' semantics)
    assert np.array_equal(np.tril(Ad, -1), np.tril(A, -1))
    Ad = A.copy(order="F")
    x = b.copy()
    L.check(lib.hyp_dense_posv(ctx, n, fp(Ad), n, fp(x), ctypes.byref(info)), "posv")
    assert info.value == 0
    xref = np.linalg.solve(A, b)
    assert np.linalg.norm(x - xref) <= 1e-10 * np.linalg.norm(xref) * np.linalg.cond(A)


@pytest.mark.parametrize("n", [768, 1000, 1153, 2250])
def test_potrf_large_lookahead_form(hip, n):
    """n >= 6 blocks of 128: the look-ahead form (dense.hip: potrf_upper_batched) -- diagonal block, panel and block-row
    update on the main stream, the rest of each rank-128 update on the helper stream underneath the next block step.
    Ragged last blocks included."""
    lib, ctx, L = hip
    rng = np.random.default_rng(n)
    M = rng.standard_normal((n, n + 5))
    A = np.asfortranarray(M @ M.T + 0.5 * np.eye(n))
    Uref = np.linalg.cholesky(A).T
    Ad = A.copy(order="F")
    info = c_int(-1)
    L.check(lib.hyp_dense_potrf(ctx, n, fp(Ad), n, ctypes.byref(info)), "potrf")
    assert info.value == 0
    U = np.triu(Ad)
    assert np.allclose(U, Uref, rtol=1e-9, atol=1e-10)
    assert np.linalg.norm(U.T @ U - A) <= 1e-14 * n * np.linalg.norm(A)
    assert np.array_equal(np.tril(Ad, -1), np.tril(A, -1))
    # a second factorization right behind the first (the ordering events are reused): same bits
    Ad2 = A.copy(order="F")
    L.check(lib.hyp_dense_potrf(ctx, n, fp(Ad2), n, ctypes.byref(info)), "potrf")
    assert np.array_equal(Ad, Ad2)


@pytest.mark.parametrize("n", [127, 128, 129, 255, 257, 383, 385, 511, 512, 513, 640, 641, 767, 768, 769, 895, 1023, 1024, 1025, 1153, 1535, 1537, 2049, 3073, 3210])
def test_posv_across_the_block_and_plan_thresholds(hip, n):
    """sizes on either side of every switch of the factor / solve path: the 128-wide block, the 6-block look-ahead form (768), the
    super-block solve plan (512) and its super-block sizes (ceil(n / 3 / 128) 128, capped at 1024).  Bar: LAPACK's backward error"""
    import scipy.linalg as sla
    lib, ctx, L = hip
    rng = np.random.default_rng(n)
    M = rng.standard_normal((n, n + 2))
    A = np.asfortranarray(M @ M.T / n + 1e-3 * np.eye(n))
    b = rng.standard_normal(n)
    Ad, x, info = A.copy(order="F"), b.copy(), c_int(-1)
    L.check(lib.hyp_dense_posv(ctx, n, fp(Ad), n, fp(x), ctypes.byref(info)), "posv")
    assert info.value == 0
    xref = sla.cho_solve(sla.cho_factor(A), b)
    berr = lambda v: np.linalg.norm(A @ v - b) / (np.linalg.norm(A, 2) * np.linalg.norm(v) + np.linalg.norm(b))
    assert berr(x) <= 4 * berr(xref) + 1e-16, (n, berr(x), berr(xref))
    Uref = np.linalg.cholesky(A).T
    assert np.allclose(np.triu(Ad), Uref, rtol=1e-8, atol=1e-10)


def test_potrf_large_reports_failed_minor(hip):
    import scipy.linalg.lapack as lp
    lib, ctx, L = hip
    n = 1000
    rng = np.random.default_rng(5)
    M = rng.standard_normal((n, n))
    A = M @ M.T + np.eye(n)
    A[700, 700] = -1.0
    Ad = np.asfortranarray(A)
    info = c_int(0)
    L.check(lib.hyp_dense_potrf(ctx, n, fp(Ad), n, ctypes.byref(info)), "potrf")
    _, iref = lp.dpotrf(A, lower=0)
    assert info.value == iref == 701


@pytest.mark.parametrize("n,cond", [(2048, 1e2), (2500, 1e10), (3333, 1e13)])
def test_posv_superblock_solves(hip, n, cond):
    """n >= 2 * 1024: the one-RHS solves go through the inverted super-blocks + refinement against the
    factor (TriSolvePlan).  Bar: the backward error of LAPACK's substitution-based dpotrs."""
    import scipy.linalg as sla
    lib, ctx, L = hip
    rng = np.random.default_rng(n)
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    ev = np.logspace(0, -np.log10(cond), n)
    A = np.asfortranarray((Q * ev) @ Q.T)
    A = np.asfortranarray(0.5 * (A + A.T))
    b = rng.standard_normal(n)
    Ad, x, info = A.copy(order="F"), b.copy(), c_int(-1)
    L.check(lib.hyp_dense_posv(ctx, n, fp(Ad), n, fp(x), ctypes.byref(info)), "posv")
    assert info.value == 0
    xref = sla.cho_solve(sla.cho_factor(A), b)
    berr = lambda v: np.linalg.norm(A @ v - b) / (np.linalg.norm(A, 2) * np.linalg.norm(v) + np.linalg.norm(b))
    assert berr(x) <= 4 * berr(xref) + 1e-16, (berr(x), berr(xref))
    assert np.linalg.norm(x - xref) <= 1e-12 * cond * np.linalg.norm(xref)


@pytest.mark.parametrize("n,nrhs,cond", [(1, 3, 1.0), (17, 1, 1e6), (50, 100, 1e8), (64, 16, 1e10), (65, 17, 1e8), (128, 33, 1e10), (129, 5, 1e6),
                                         (200, 1000, 1e10), (495, 600, 1e8), (330, 4845, 1e6), (50, 100, 1e14), (128, 33, 1e15)])
def test_posv_multi_has_substitution_backward_error(hip, n, nrhs, cond):
    """the multi-column triangular sweeps (trsm_upper_left: inverted diagonal blocks + two refinement steps against the factor, the
    wave-local kernel for blocks of at most 64 rows, the 128-row kernel otherwise, partial last blocks, ragged column counts)
    against LAPACK's dpotrs on ill-conditioned matrices: column by column the backward error of substitution"""
    import scipy.linalg as sla
    lib, ctx, L = hip
    rng = np.random.default_rng(1000 * n + nrhs)
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    ev = np.logspace(0, -np.log10(cond), n) if n > 1 else np.ones(1)
    A = (Q * ev) @ Q.T
    A = np.asfortranarray(0.5 * (A + A.T))
    # half the columns random, half images of O(1) vectors.  Measured (1x MI355X): with the refinement the backward error equals
    # LAPACK's (1.0e-16 .. 1.8e-16 against 0.9e-16 .. 1.8e-16) at every conditioning tried; the plain products with the inverted
    # blocks (HYP_TRSM_REFINE=0) give 3e-16 .. 5e-16 at cond 1e8 .. 1e10 and 7e-16 .. 1.4e-15 at cond 1e14 .. 1e15 -- the last
    # two cases of this list fail without the refinement, the others pass either way
    B = np.asfortranarray(rng.standard_normal((n, nrhs)))
    B[:, ::2] = A @ rng.standard_normal((n, B[:, ::2].shape[1]))
    Ad, X, info = A.copy(order="F"), B.copy(order="F"), c_int(-1)
    L.check(lib.hyp_dense_posv_multi(ctx, n, fp(Ad), n, fp(X), nrhs, n, ctypes.byref(info)), "posv_multi")
    assert info.value == 0
    Xref = sla.cho_solve(sla.cho_factor(A), B)
    nA = np.linalg.norm(A, 2)
    berr = lambda V: np.max(np.linalg.norm(A @ V - B, axis=0) / (nA * np.linalg.norm(V, axis=0) + np.linalg.norm(B, axis=0)))
    assert berr(X) <= 4 * berr(Xref) + 2e-16, (berr(X), berr(Xref))
    assert np.linalg.norm(X - Xref) <= 1e-11 * cond * np.linalg.norm(Xref)


def test_superblock_solve_kernels_same_sums_with_and_without_batched_loads(tmp_path):
    """the dot-product kernels of the super-block solves exist in two forms (all loads of a lane issued at once, or four at a
    time: HYP_COLDOT_BATCH); the switch is read once per process, so each form runs in a process of its own -- bitwise equal"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for flag in ("0", "1"):
        out = str(tmp_path / ("trsv%s.npz" % flag))
        env = dict(os.environ, HYP_COLDOT_BATCH=flag, HYP_TRSV_ONE_LAUNCH="0")
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "bench_trsv.py"), "2300", "--dump", out], cwd=root, env=env,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(np.load(out)["x2300"])
    assert np.all(np.isfinite(outs[0])) and np.array_equal(outs[0], outs[1])


def test_one_launch_triangular_sweeps_same_bits_as_the_launch_chains(tmp_path):
    """round 5 (csrc/trsv_onelaunch.hip): both sweeps of a potrs as ONE persistent launch (default), one launch per sweep
    (HYP_TRSV_ONE_LAUNCH=1) and the chain of one launch per column product (=0) on one, two and three right-hand sides: bitwise equal.
    Sizes: n = 5000 (five super-blocks of 1024, a ragged last one), 4845, 2300 (three of 768), 999 (three of 384, ragged), 640
    (256 + 256 + 128), 513 (a last super-block of one row).  qrchol.jl:66-69, Cones.jl:113-118"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sizes = ["5000", "4845", "2300", "999", "640", "513"]
    outs = []
    for flag in ("0", "1", "2"):
        out = str(tmp_path / ("trsv_ol%s.npz" % flag))
        # (HYP_PERSISTENT=1: this pytest process may hold the device's persistent-kernel lock through a context of its own -- it launches
        #  nothing while the tool runs; HYP_TRSV_OL_STATS: the tool's process reports the one-launch sweeps it ran, so the comparison
        #  cannot pass on launch chains alone)
        env = dict(os.environ, HYP_TRSV_ONE_LAUNCH=flag, HYP_PERSISTENT="1", HYP_TRSV_OL_STATS="1")
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "bench_trsv.py")] + sizes + ["--dump", out], cwd=root, env=env,
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        ran = "[trsv one-launch]" in r.stderr
        assert ran == (flag != "0"), (flag, r.stderr[-600:])
        if flag == "2":
            assert "both, 3 right-hand side(s)" in r.stderr and "both, 1 right-hand side(s)" in r.stderr
        outs.append(np.load(out))
    for n in sizes:
        a, b, c = (o["x" + n] for o in outs)
        assert np.all(np.isfinite(a)) and np.abs(a).max() > 0
        assert np.array_equal(a, b) and np.array_equal(a, c), n


def test_potrf_reports_failed_minor(hip):
    lib, ctx, L = hip
    n = 200
    rng = np.random.default_rng(3)
    M = rng.standard_normal((n, n))
    A = M @ M.T + np.eye(n)
    A[150, 150] = -1.0     # leading minor 151 is not positive definite
    Ad = np.asfortranarray(A)
    info = c_int(0)
    L.check(lib.hyp_dense_potrf(ctx, n, fp(Ad), n, ctypes.byref(info)), "potrf")
    import scipy.linalg.lapack as lp
    _, iref = lp.dpotrf(A, lower=0)
    assert info.value == iref == 151


@pytest.mark.parametrize("trans,m,n", [(0, 1000, 300), (1, 1000, 300), (0, 3, 5), (1, 20100, 64)])
def test_gemv(hip, trans, m, n):
    lib, ctx, L = hip
    rng = np.random.default_rng(1)
    A = np.asfortranarray(rng.standard_normal((m, n)))
    x = rng.standard_normal(m if trans else n)
    y = rng.standard_normal(n if trans else m)
    y0 = y.copy()
    L.check(lib.hyp_dense_gemv(ctx, trans, m, n, 1.5, fp(A), m, fp(x), -0.5, fp(y)), "gemv")
    ref = 1.5 * ((A.T if trans else A) @ x) - 0.5 * y0
    assert np.allclose(y, ref, rtol=1e-12, atol=1e-11)


@pytest.mark.parametrize("m,n,nr,lda,fused", [(20100, 300, 2, 20100, 1), (20100, 300, 1, 20100, 1), (4845, 200, 2, 4848, 1),
                                               (1024, 64, 2, 1024, 1), (2051, 131, 1, 2052, 1), (5000, 70, 2, 5001, 0), (600, 90, 2, 600, 0)])
def test_gemv_both_products_in_one_pass(hip, m, n, nr, lda, fused):
    """A X and A' Z from one pass over A (the residual of a pair of directions, apply_lhs common.jl:79-121, and the residuals of
    calc_convergence_params, Solvers.jl:425-483): ragged row and column counts, a leading dimension beyond m, beta on both sides,
    and the shapes that fall back to the two one-sided products"""
    lib, ctx, L = hip
    rng = np.random.default_rng(m + n)
    A = np.zeros((lda, n), order="F")
    A[:m] = rng.standard_normal((m, n))
    A[m:] = np.nan                                   # rows beyond m are never read
    Xn = np.asfortranarray(rng.standard_normal((n, nr)))
    Xt = np.asfortranarray(rng.standard_normal((m, nr)))
    Yn = np.asfortranarray(rng.standard_normal((m, nr)))
    Yt = np.asfortranarray(rng.standard_normal((n, nr)))
    Yn0, Yt0 = Yn.copy(), Yt.copy()
    used = c_int(-1)
    L.check(lib.hyp_dense_gemv_both(ctx, m, n, nr, fp(A), lda, fp(Xn), 1.0, fp(Yn), fp(Xt), -0.5, fp(Yt), ctypes.byref(used)), "gemv_both")
    assert used.value == fused
    assert np.allclose(Yn, A[:m] @ Xn + Yn0, rtol=1e-12, atol=1e-11)
    assert np.allclose(Yt, A[:m].T @ Xt - 0.5 * Yt0, rtol=1e-12, atol=1e-11)
    # a second call gives the same bits (fixed assignment and reduction order)
    Yn2, Yt2 = Yn0.copy(order="F"), Yt0.copy(order="F")
    L.check(lib.hyp_dense_gemv_both(ctx, m, n, nr, fp(A), lda, fp(Xn), 1.0, fp(Yn2), fp(Xt), -0.5, fp(Yt2), None), "gemv_both")
    assert np.array_equal(Yn, Yn2) and np.array_equal(Yt, Yt2)


@pytest.mark.parametrize("N,K", [(1032, 70001), (1044, 66000)])
def test_syrk_thin_last_tile_column_long_k(hip, N, K):
    """the edge columns of the Schur syrk (n = 39 x 128 + 8 at config 2) at a K where the edge kernel takes eight columns of A per
    workgroup (config 4: K = 207 360), odd K (element-wise tail) and an edge of 20 columns (three passes): against numpy on the
    edge columns, the rest of the upper triangle spot-checked, the lower triangle untouched"""
    lib, ctx, L = hip
    rng = np.random.default_rng(N + K)
    A = np.asfortranarray(rng.standard_normal((K, N), dtype=np.float32).astype(np.float64))
    C = np.full((N, N), 3.0, order="F")
    L.check(lib.hyp_dense_syrk(ctx, N, K, fp(A), K, fp(C), N), "syrk")
    N0 = N - N % 128
    ref_edge = A.T @ A[:, N0:]
    for e in range(N - N0):
        col = N0 + e
        assert np.allclose(C[:col + 1, col], ref_edge[:col + 1, e], rtol=1e-12, atol=1e-9), e
        assert np.all(C[col + 1:, col] == 3.0)
    cols = [0, 1, 127, 128, 500, N0 - 1]
    ref = A.T @ A[:, cols]
    for t, col in enumerate(cols):
        assert np.allclose(C[:col + 1, col], ref[:col + 1, t], rtol=1e-12, atol=1e-9)
    C2 = np.full((N, N), 3.0, order="F")
    L.check(lib.hyp_dense_syrk(ctx, N, K, fp(A), K, fp(C2), N), "syrk")
    assert np.array_equal(C, C2)


@pytest.mark.parametrize("N,K", [(300, 5000), (130, 4100), (257, 900), (1032, 4500), (1300, 4100)])
def test_syrk_schur_path_with_splitk(hip, N, K):
    """the Schur-assembly syrk (split-K slices + ordered reduction when K is long)"""
    lib, ctx, L = hip
    rng = np.random.default_rng(N + K)
    A = np.asfortranarray(rng.standard_normal((K, N)))
    C = np.full((N, N), 3.0, order="F")
    L.check(lib.hyp_dense_syrk(ctx, N, K, fp(A), K, fp(C), N), "syrk")
    ref = A.T @ A
    iu = np.triu_indices(N)
    assert np.allclose(C[iu], ref[iu], rtol=1e-12, atol=1e-10)
    il = np.tril_indices(N, -1)
    assert np.all(C[il] == 3.0)
    # deterministic: bitwise identical on a second run
    C2 = np.full((N, N), 3.0, order="F")
    L.check(lib.hyp_dense_syrk(ctx, N, K, fp(A), K, fp(C2), N), "syrk")
    assert np.array_equal(C, C2)


def test_two_contexts_are_independent(hip):
    """distinct contexts must be independent.  Two contexts on the same device run the split-K Schur syrk and
    the rook solve alternately (each owns its split-K workspace, tile order and streams) and agree with numpy."""
    lib, ctx1, L = hip
    h2 = c_vp()
    assert lib.hyp_ctx_create(0, ctypes.byref(h2)) == 0
    try:
        rng = np.random.default_rng(11)
        n, q = 1100, 4200
        A1 = np.asfortranarray(rng.standard_normal((q, n)))
        A2 = np.asfortranarray(rng.standard_normal((q, n)))
        C1, C2 = np.zeros((n, n), order="F"), np.zeros((n, n), order="F")
        for _ in range(2):
            L.check(lib.hyp_dense_syrk(ctx1, n, q, fp(A1), q, fp(C1), n), "syrk ctx1")
            L.check(lib.hyp_dense_syrk(h2, n, q, fp(A2), q, fp(C2), n), "syrk ctx2")
        iu = np.triu_indices(n)
        assert np.allclose(C1[iu], (A1.T @ A1)[iu], rtol=1e-12, atol=1e-10)
        assert np.allclose(C2[iu], (A2.T @ A2)[iu], rtol=1e-12, atol=1e-10)
        M = rng.standard_normal((300, 300))
        S = np.asfortranarray(M + M.T)
        b = rng.standard_normal(300)
        for c in (ctx1, h2):
            Sd, x, info = S.copy(order="F"), b.copy(), c_int(-1)
            L.check(lib.hyp_dense_sysv_rook(c, 300, fp(Sd), 300, fp(x), 1, 300, ctypes.byref(info), None, None, None, None), "sysv")
            assert info.value == 0 and np.linalg.norm(S @ x - b) <= 1e-10 * np.linalg.norm(b) * np.linalg.cond(S)
    finally:
        assert lib.hyp_ctx_destroy(h2) == 0


@pytest.mark.parametrize("m,n", [(2500, 600), (9000, 2100)])
def test_lstsq_normal_matches_qr_and_estimates_conditioning(hip, m, n):
    """set-up helper of find_initial_x: Cholesky of A'A + one corrected semi-normal-equations step on the device"""
    lib, ctx, L = hip
    rng = np.random.default_rng(m)
    A = np.asfortranarray(rng.standard_normal((m, n)) / np.sqrt(n))
    b = rng.standard_normal(m)
    x, rc, info = np.zeros(n), ctypes.c_double(0), c_int(-1)
    L.check(lib.hyp_dense_lstsq_normal(ctx, m, n, fp(A), m, fp(b), fp(x), ctypes.byref(rc), ctypes.byref(info)), "lstsq_normal")
    xref = np.linalg.lstsq(A, b, rcond=None)[0]
    sv = np.linalg.svd(A, compute_uv=False)
    assert info.value == 0
    assert np.linalg.norm(x - xref) <= 1e-12 * np.linalg.norm(xref)
    assert 0.5 * sv[-1] / sv[0] <= rc.value <= 1.5 * sv[-1] / sv[0]
    # nearly dependent columns: the estimate says so (the caller then takes the reference's pivoted QR)
    A2 = A.copy(order="F")
    A2[:, -1] = A2[:, 0] + 1e-7 * A2[:, 1]
    L.check(lib.hyp_dense_lstsq_normal(ctx, m, n, fp(A2), m, fp(b), fp(x), ctypes.byref(rc), ctypes.byref(info)), "lstsq_normal")
    assert info.value != 0 or rc.value < 1e-5


def _potrf_variant(n, cond, env):
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "potrf_variant.py"), str(n)] + ([repr(cond)] if cond else []),
                       cwd=root, env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


@pytest.mark.parametrize("n,cond", [(100, 0), (200, 0), (1000, 0), (1153, 1e12)])
def test_potrf_forms_of_the_diagonal_block_kernel(n, cond):
    """round 4: the diagonal-block kernel's fourth form (tile inverses riding along with the 16 x 16 factorizations, solves as MFMA
    products with one refinement step, the owner's trailing updates deferred by one block step: potrf_mfma.hip).  The deferral moves
    work between block steps without changing any sum: HYP_POTRF_DEFER=0 (the third form with the same solves) must give the same
    bits.  The solves themselves are new arithmetic: against the substitution form (HYP_POTRF_TINV=0) the factor keeps LAPACK's
    backward error, also on a matrix of condition 1e12 (the switches are read once per process: one child process each)."""
    new = _potrf_variant(n, cond, {})
    nodefer = _potrf_variant(n, cond, {"HYP_POTRF_DEFER": "0"})
    subst = _potrf_variant(n, cond, {"HYP_POTRF_TINV": "0"})
    assert new["info"] == nodefer["info"] == subst["info"] == 0
    assert new["sha"] == nodefer["sha"]
    assert new["sha"] != subst["sha"]                     # (different rounding: the comparison below is not vacuous)
    assert new["berr"] <= 1.5 * subst["berr"] + 1e-19, (new["berr"], subst["berr"])
    assert new["berr"] <= 4e-16 / n ** 0.5                # || U'U - A || / (n || A ||): a few eps over n^1.5


@pytest.mark.parametrize("n,cond", [(1900, 0), (2250, 1e10), (5000, 0)])
def test_potrf_diag_update_in_the_diagonal_block_kernel(n, cond):
    """round 6: with look-ahead, the previous block step's update of the next diagonal block is formed inside that block's
    factorization kernel (potrf_mfma.hip: t4_prev_update) and the rest of the block row is updated on the helper stream
    (dense.hip: HYP_POTRF_DIAGUPD=1; measured and left off, EXPERIMENTS.md r06-18).  Same products in the same order as the GEMM that did it before: the same bits as
    HYP_POTRF_DIAGUPD=0, and as the factorization without look-ahead."""
    new = _potrf_variant(n, cond, {"HYP_POTRF_DIAGUPD": "1"})
    old = _potrf_variant(n, cond, {"HYP_POTRF_DIAGUPD": "0"})
    plain = _potrf_variant(n, cond, {"HYP_POTRF_LOOKAHEAD": "0"})
    assert new["info"] == old["info"] == plain["info"] == 0
    assert new["sha"] == old["sha"] == plain["sha"]
    assert new["berr"] <= 4e-16 / n ** 0.5
