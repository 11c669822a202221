"""GPU: the device column-pivoted QR (hyp_qrcp_*, csrc/qrcp.hip) against LAPACK dgeqp3 (scipy.linalg.qr(pivoting=True)), the
routine Julia's qr!(AG, ColumnNorm()) of the reference's find_initial_x calls (src/Solvers/process.jl:64-178, 373-382):
pivots, R, Q'b, the rank decision on rank-deficient inputs, and find_initial_x through it."""
import numpy as np
import pytest
import scipy.linalg as sla

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / (np.linalg.norm(b) + 1e-300)


@pytest.mark.parametrize("m,n", [(5, 1), (30, 7), (64, 64), (300, 200), (1000, 130), (2500, 700)])
def test_qrcp_matches_lapack_on_full_rank_matrices(m, n):
    from hypatia_jl_amd.solvers import DeviceQRCP
    rng = np.random.default_rng(m * 1000 + n)
    M = rng.standard_normal((m, n)) * rng.uniform(0.2, 5.0, n)[None, :]      # distinct column norms: a stable pivot order
    b = rng.standard_normal(m)
    f = DeviceQRCP(M, b)
    piv, R, rdiag, qtb = f.get()
    Qs, Rs, ps = sla.qr(M, mode="full", pivoting=True)
    assert np.array_equal(piv, ps)
    assert _rel(R, np.triu(Rs[:n, :])) <= 1e-12
    assert _rel(rdiag, np.diagonal(Rs)[:n]) <= 1e-12
    assert _rel(qtb[:n], (Qs.T @ b)[:n]) <= 1e-12
    assert abs(np.linalg.norm(qtb) - np.linalg.norm(b)) <= 1e-12 * np.linalg.norm(b)     # Q orthogonal: the whole of Q'b
    # Q (Q' x) = x and Q' applied separately = the rider column
    x = rng.standard_normal(m)
    assert _rel(f.apply_q(f.apply_q(x, True), False), x) <= 1e-12
    assert _rel(f.apply_q(b, True), qtb) <= 1e-12
    # least squares through it
    xs = np.zeros(n)
    xs[piv] = sla.solve_triangular(R[:n, :n], qtb[:n], lower=False)
    assert _rel(xs, np.linalg.lstsq(M, b, rcond=None)[0]) <= 1e-10


@pytest.mark.parametrize("scale", [1e-150, 1e-300])
def test_qrcp_on_tiny_columns(scale):
    """columns near the bottom of the exponent range.  At 1e-150 everything equals LAPACK's (pivots, R, Q'b).  At 1e-300 the
    SQUARED column norms the pivot search works on underflow (dgeqp3 keeps scaled norms), so the pivot order is the identity
    instead of LAPACK's -- but every reflector is still dlarfg's: its norm comes from dnrm2's scaled sum and |beta| < dlamch('S') /
    dlamch('E') ~ 2e-292 takes the rescaling loop, so Q stays orthogonal and A P = Q R holds relative to the matrix's own scale"""
    from hypatia_jl_amd.solvers import DeviceQRCP
    rng = np.random.default_rng(11)
    m, n = 60, 9
    M = rng.standard_normal((m, n)) * rng.uniform(0.5, 2.0, n)[None, :] * scale
    b = rng.standard_normal(m)
    f = DeviceQRCP(M, b)
    piv, R, rdiag, qtb = f.get()
    assert abs(np.linalg.norm(qtb) - np.linalg.norm(b)) <= 1e-12 * np.linalg.norm(b)
    x = rng.standard_normal(m)
    assert _rel(f.apply_q(f.apply_q(x, True), False), x) <= 1e-12
    for j in range(n):   # A P = Q R, column by column
        col = np.zeros(m)
        col[:n] = R[:n, j] / scale
        assert _rel(f.apply_q(col, False), M[:, piv[j]] / scale) <= 1e-11
    if scale > 1e-200:
        Qs, Rs, ps = sla.qr(M, mode="full", pivoting=True)
        assert np.array_equal(piv, ps)
        assert _rel(R / scale, np.triu(Rs[:n, :]) / scale) <= 1e-11
        assert _rel(qtb[:n], (Qs.T @ b)[:n]) <= 1e-11


@pytest.mark.parametrize("m,n,rank", [(40, 12, 7), (400, 150, 100), (1500, 300, 299)])
def test_qrcp_rank_decision_on_rank_deficient_matrices(m, n, rank):
    """get_rank_est (process.jl:373-382): number of |R_ii| above init_tol_qr = 1000 eps"""
    from hypatia_jl_amd.solvers import DeviceQRCP
    rng = np.random.default_rng(rank)
    B = rng.standard_normal((m, rank)) * rng.uniform(0.5, 3.0, rank)[None, :]
    M = B @ rng.standard_normal((rank, n))
    f = DeviceQRCP(M)
    piv, R, rdiag, _ = f.get()
    Rs, ps = sla.qr(M, mode="r", pivoting=True)
    tol = 1000 * np.finfo(float).eps
    scale = abs(rdiag[0])
    assert int(np.sum(np.abs(rdiag) > tol * scale)) == int(np.sum(np.abs(np.diagonal(Rs)) > tol * scale)) == rank
    # the leading (well-determined) part of the pivot order and of R agree with LAPACK's; the kept columns span the range
    lead = max(1, rank // 2)
    assert np.array_equal(piv[:lead], ps[:lead])
    assert _rel(np.abs(rdiag[:lead]), np.abs(np.diagonal(Rs)[:lead])) <= 1e-10
    kept = M[:, piv[:rank]]
    resid = M - kept @ np.linalg.lstsq(kept, M, rcond=None)[0]
    assert np.linalg.norm(resid) <= 1e-9 * np.linalg.norm(M)


def test_find_initial_x_through_the_device_qr_matches_the_host_path(monkeypatch):
    """the driver's find_initial_x (process.jl:64-178) with the factorization on the device vs LAPACK on the host: full-rank
    model, and a model with dependent dual equalities (columns of [A; G]) that the preprocessing must remove"""
    import hypatia_jl_amd as H
    from hypatia_jl_amd import solvers as HS
    from oracle import instances as I
    rng = np.random.default_rng(5)
    inst = I.psd_blocks(400, [40, 30], seed=2)              # q = 1285, n = 400: (p + q) n^2 = 2.1e8 -> force the device path below
    c, A, b, G, h, specs = inst[:6]
    for dependent in (False, True):
        Gm, cm = G.copy(), c.copy()
        if dependent:                                       # column 17 = combination of columns 3 and 5, c consistent with it
            Gm[:, 17] = 0.5 * Gm[:, 3] - 2.0 * Gm[:, 5]
            cm[17] = 0.5 * cm[3] - 2.0 * cm[5]
        res = {}
        for mode in ("device", "host"):
            s = H.Solver(verbose=False)
            s.load(H.make_model((cm, A, b, Gm, h, specs, {})))
            s.status = "SolveCalled"
            s.model = s.orig_model.copy()
            init_s = np.concatenate([np.zeros(d[1]) for d in specs])
            off = 0
            for cone in s.model.cones:
                cone.set_initial_point(init_s[off:off + cone.dimension()])
                off += cone.dimension()
            if mode == "device":
                AG = s.model.G.copy()
                x = HS._find_initial_x_device_qr(s, s.model, AG, s.model.h - init_s)
            else:
                monkeypatch.setenv("HYP_INITX_DEVICE", "0")
                x = HS.find_initial_x(s, init_s)
                monkeypatch.delenv("HYP_INITX_DEVICE")
            res[mode] = (x, s.model.n, np.sort(s.x_keep_idxs) if hasattr(s, "x_keep_idxs") else None, s.status)
        xd, nd, kd, sd = res["device"]
        xh, nh, kh, sh = res["host"]
        assert sd == sh == "SolveCalled"
        assert nd == nh == (399 if dependent else 400)
        if dependent:
            assert np.array_equal(kd, kh)
        assert _rel(xd, xh) <= 1e-9


@pytest.mark.timeout(900)
def test_qrcp_config2_size():
    """configs[1] size (20100 x 5000): full rank, and the least-squares x agrees with the normal-equations path of round 1"""
    import ctypes
    import time
    import hypatia_jl_amd as H
    from hypatia_jl_amd import _lib as L
    from hypatia_jl_amd.solvers import DeviceQRCP
    from oracle import instances as I
    inst = I.psd_blocks(5000, [200], seed=3)
    G, h = inst[3], inst[4]
    q, n = G.shape
    init_s = np.zeros(q)
    H.PosSemidefTri(q).set_initial_point(init_s)
    rhs = h - init_s
    t0 = time.perf_counter()
    f = DeviceQRCP(G, rhs)
    piv, R, rdiag, qtb = f.get()
    dt = time.perf_counter() - t0
    print("device dgeqp3 20100 x 5000: %.2f s (upload and read-back included)" % dt)
    assert int(np.sum(np.abs(rdiag) > 1000 * np.finfo(float).eps)) == n
    assert np.all(np.abs(rdiag[:-1]) >= np.abs(rdiag[1:]) * (1 - 1e-10))       # |R_ii| non-increasing (column pivoting)
    x = np.zeros(n)
    x[piv] = sla.solve_triangular(R[:n, :n], qtb[:n], lower=False)
    AGf = np.asfortranarray(G)
    xs, rc, info = np.zeros(n), ctypes.c_double(0.0), ctypes.c_int(-1)
    L.check(L.lib().hyp_dense_lstsq_normal(L.ctx(), q, n, AGf.ctypes.data_as(ctypes.c_void_p), q, L.vec_ptr(np.ascontiguousarray(rhs)),
                                           L.vec_ptr(xs), ctypes.byref(rc), ctypes.byref(info)), "hyp_dense_lstsq_normal")
    assert _rel(x, xs) <= 1e-10
    assert np.linalg.norm(G.T @ (G @ x - rhs)) <= 1e-9 * np.linalg.norm(G.T @ rhs)
    assert dt < 5.0
