"""Instances: the reference's deterministic known-answer tests restated as data, and the synthetic
generators for BASELINE.json's configs.  Test infrastructure (see oracle/__init__.py).

Known-answer instances follow /root/reference/test/nativeinstances.jl (line ranges per function).
Each `inst_*` returns (c, A, b, G, h, cone_specs, expect) where cone_specs is a list of tuples
("nonnegative", dim) | ("possemideftri", dim) | ("epinormspectral", d1, d2, use_dual) |
("wsosinterpnonnegative", U, Ps, use_dual) so both the oracle cones and the HIP cones can be built
from the same description, and `expect` holds the pinned answers.
"""
import numpy as np

from . import polyutils as pu
from . import arrayutil as au

RT2 = np.sqrt(2.0)
EPS = float(np.finfo(np.float64).eps)
RT3 = np.sqrt(3.0)
TEST_TOL = float(np.sqrt(np.sqrt(np.finfo(np.float64).eps)))   # test_tol(T), nativeinstances.jl:29


def dimension1():   # nativeinstances.jl:88-108
    return (np.array([-1.0, 0]), np.zeros((0, 2)), np.zeros(0), np.array([[1.0, 0]]), np.array([1.0]),
            [("nonnegative", 1)], dict(status="Optimal", primal_obj=-1.0, x=[1.0, 0.0]))


def primalinfeas1():   # :169-180
    return (np.array([1.0, 0]), np.array([[1.0, 1]]), np.array([-2.0]), -np.eye(2), np.zeros(2),
            [("nonnegative", 2)], dict(status="PrimalInfeasible"))


def nonnegative4():   # :295-310
    G = np.zeros((3, 2))
    G[0, 0], G[0, 1], G[1, 1], G[2, 1] = 1, -1, 1, -1
    return (np.array([-2.0, 0]), np.zeros((0, 2)), np.zeros(0), G, np.array([0.0, 2, 0]),
            [("nonnegative", 3)],
            dict(status="Optimal", primal_obj=-4.0, x=[2.0, 2.0], s=[0.0, 0, 2], z=[2.0, 2, 0]))


def possemideftri1():   # :312-325
    return (np.array([0.0, -1, 0]), np.array([[1.0, 0, 0], [0, 0, 1]]), np.array([0.5, 1]), -np.eye(3), np.zeros(3),
            [("possemideftri", 3)], dict(status="Optimal", primal_obj=-1.0, x_at={1: 1.0}))


def possemideftri2():   # :327-340
    return (np.array([0.0, -1, 0]), np.array([[1.0, 0, 1]]), np.array([0.0]), -np.eye(3), np.zeros(3),
            [("possemideftri", 3)], dict(status="Optimal", primal_obj=0.0, x_norm=0.0))


def possemideftri3(seed=1):   # :342-360 (property-based: objective = max eigenvalue of a random matrix)
    rng = np.random.default_rng(seed)
    m = rng.random((2, 2))
    m = np.triu(m) + np.triu(m, 1).T
    h = -np.array([m[0, 0], RT2 * m[0, 1], m[1, 1]])
    eig_max = float(np.max(np.linalg.eigvalsh(m)))
    return (np.array([1.0]), np.zeros((0, 1)), np.zeros(0), np.array([[-1.0], [0], [-1]]), h,
            [("possemideftri", 3)], dict(status="Optimal", primal_obj=eig_max, x=[eig_max]))


def possemideftri4(seed=1):   # :362-380
    rng = np.random.default_rng(seed)
    s = 3
    m = rng.random((s, s))
    m = np.triu(m) + np.triu(m, 1).T
    dim = 6
    jj, ii = np.tril_indices(s)
    c = -(m[ii, jj] * np.where(ii == jj, 1.0, RT2))
    A = (np.eye(s)[ii, jj] * 1.0).reshape(1, dim)
    return (c, A, np.array([1.0]), -np.eye(dim), np.zeros(dim), [("possemideftri", dim)],
            dict(status="Optimal", primal_obj=-float(np.max(np.linalg.eigvalsh(m)))))


def possemideftri8():   # :439-462
    G = np.zeros((15, 1))
    G[[0, 2, 5, 9, 14], 0] = -1
    h = np.zeros(15)
    h[[6, 7, 8, 10, 11, 12]] = RT2 * np.array([1.0, 1, 0, 1, -1, 1])
    inv6, rt2inv6, invrt6 = 1 / 6, RT2 / 6, 1 / (RT2 * RT3)
    return (np.array([1.0]), np.zeros((0, 1)), np.zeros(0), G, h, [("possemideftri", 15)],
            dict(status="Optimal", primal_obj=RT3,
                 s=[RT3, 0, RT3, 0, 0, RT3, RT2, RT2, 0, RT3, RT2, -RT2, RT2, 0, RT3],
                 z=[inv6, -rt2inv6, inv6, rt2inv6, -rt2inv6, inv6, 0, 0, 0, 0, -invrt6, invrt6, -invrt6, 0, 0.5]))


def possemideftri9():   # :464-491
    G = np.zeros((16, 10))
    for j in (1, 3, 6, 7, 9):
        G[0, j] = 0.5
    G[0, 0] = G[1, 1] = G[3, 3] = G[6, 6] = G[10, 7] = G[15, 9] = -1
    G[2, 2] = G[4, 4] = G[5, 5] = G[14, 8] = -RT2
    h = np.zeros(16)
    h[[7, 8, 9, 11, 12, 13]] = RT2 * np.array([1.0, 1, 0, 1, -1, 1])
    c = np.zeros(10)
    c[0] = 1
    invrt2, invrt3 = 1 / RT2, 1 / RT3
    invrt6 = invrt2 * invrt3
    return (c, np.zeros((0, 10)), np.zeros(0), G, h, [("nonnegative", 1), ("possemideftri", 15)],
            dict(status="Optimal", primal_obj=RT2 + RT3,
                 s=[0, invrt2 + invrt3, 1 - RT2 / RT3, invrt2 + invrt3, RT2 * invrt3, -RT2 * invrt3, invrt3, RT2, RT2, 0,
                    RT2, RT2, -RT2, RT2, 0, RT3],
                 z=[1, 0.5, 0, 0.5, 0, 0, 0.5, -0.5, -0.5, 0, 0.5, -invrt6, invrt6, -invrt6, 0, 0.5]))


def epinormspectral2(use_dual, seed=1):   # :1074-1103 (real case; property-based)
    rng = np.random.default_rng(seed)
    Xn, Xm = 3, 4
    dim = Xn * Xm
    mat = rng.random((Xn, Xm))
    c = -mat.reshape(-1, order="F")
    G = np.vstack([np.zeros((1, dim)), -np.eye(dim)])
    h = np.concatenate([[1.0], np.zeros(dim)])
    sv = np.linalg.svd(mat, compute_uv=False)
    obj = -sv[0] if use_dual else -np.sum(sv)
    return (c, np.zeros((0, dim)), np.zeros(0), G, h, [("epinormspectral", Xn, Xm, use_dual)],
            dict(status="Optimal", primal_obj=float(obj)))


def epinormspectral3(Xn, Xm, use_dual):   # :1105-1125 (real cases)
    dim = Xn * Xm
    return (-np.ones(dim), np.zeros((0, dim)), np.zeros(0), np.vstack([np.zeros((1, dim)), -np.eye(dim)]),
            np.zeros(dim + 1), [("epinormspectral", Xn, Xm, use_dual)],
            dict(status="Optimal", primal_obj=0.0, x_norm=0.0))


def epinormspectral4(use_dual):   # :1127-1155
    G = np.zeros((7, 1))
    G[0, 0] = -1
    h = np.array([0.0, 1, 1, 1, -1, 0, 1])
    invrt2, invrt3 = 1 / RT2, 1 / RT3
    if use_dual:
        exp = dict(status="Optimal", primal_obj=RT2 + RT3, s=[RT2 + RT3, 1, 1, 1, -1, 0, 1],
                   z=[1, -invrt2, -invrt3, -invrt2, invrt3, 0, -invrt3])
    else:
        exp = dict(status="Optimal", primal_obj=RT3, s=[RT3, 1, 1, 1, -1, 0, 1], z=[1, 0, -invrt3, 0, invrt3, 0, -invrt3])
    return (np.array([1.0]), np.zeros((0, 1)), np.zeros(0), G, h, [("epinormspectral", 2, 3, use_dual)], exp)


def wsosinterpnonnegative1():   # :2286-2304
    U, pts, Ps = pu.interpolate_box([0.0, 0.0], [1.0, 1.0], 2)
    x, y = pts[:, 0], pts[:, 1]
    h = x ** 4 + x ** 2 * y ** 2 + 4 * y ** 2 + 4
    return (np.array([-1.0]), np.zeros((0, 1)), np.zeros(0), np.ones((U, 1)), h,
            [("wsosinterpnonnegative", U, Ps, False)], dict(status="Optimal", primal_obj=-4.0, x=[4.0]))


def wsosinterpnonnegative2():   # :2306-2324
    U, pts, Ps = pu.interpolate_box([0.0, 0.0], [3.0, 3.0], 2)
    x, y = pts[:, 0], pts[:, 1]
    h = (x - 2) ** 2 + (x * y - 3) ** 2
    return (np.array([-1.0]), np.zeros((0, 1)), np.zeros(0), np.ones((U, 1)), h,
            [("wsosinterpnonnegative", U, Ps, False)], dict(status="Optimal", primal_obj=0.0, x=[0.0]))


def wsosinterpnonnegative3():   # :2326-2343
    U, pts, Ps = pu.interpolate_box([0.0, 0.0], [3.0, 3.0], 2)
    x, y = pts[:, 0], pts[:, 1]
    c = (x - 2) ** 2 + (x * y - 3) ** 2
    return (c, np.ones((1, U)), np.array([1.0]), -np.eye(U), np.zeros(U),
            [("wsosinterpnonnegative", U, Ps, True)], dict(status="Optimal", primal_obj=0.0))


def linmatrixineq1(side, seed=1):   # :696-719 (real case; property-based: objective = 2 / largest eigenvalue of A_1)
    rng = np.random.default_rng(seed)
    Ah = rng.random((side, side))
    A1 = Ah @ Ah.T + 2 * np.eye(side)
    A1 = 0.5 * (A1 + A1.T)
    vals, vecs = np.linalg.eigh(A1)
    v1 = vecs[:, -1]
    A2 = -np.outer(v1, v1)
    A2 = 0.5 * (A2 + A2.T)
    G = np.zeros((2, 1))
    G[0, 0] = -1.0
    return (np.array([1.0]), np.zeros((0, 1)), np.zeros(0), G, np.array([0.0, 2.0]), [("linmatrixineq", [A1, A2], False)],
            dict(status="Optimal", primal_obj=2 / vals[-1], s=[2 / vals[-1], 2.0]))


def linmatrixineq2(seed=1):   # :721-745 (the all-real member [T, T] of the list; only primal_obj < 0 is asserted)
    rng = np.random.default_rng(seed)
    As = []
    for _ in range(2):
        Ah = rng.random((3, 3))
        M = Ah @ Ah.T
        As.append(0.5 * (M + M.T))
    As[0] = As[0] + np.eye(3)
    G = np.vstack([np.zeros((1, 1)), -np.eye(1)])
    return (np.ones(1), np.zeros((0, 1)), np.zeros(0), G, np.array([1.0, 0.0]), [("linmatrixineq", As, False)],
            dict(status="Optimal", primal_obj_negative=True))


def linmatrixineq3():   # :747-789 (dense members; the sparse / Diagonal / I variants are the same matrices)
    As = [np.array([[1.0, 0.0], [0.0, 1.0]]), np.array([[1.0, 0.0], [0.0, -1.0]])]
    G = np.zeros((2, 1))
    G[0, 0] = -1.0
    return (np.array([1.0]), np.zeros((0, 1)), np.zeros(0), G, np.array([0.0, -1.0]), [("linmatrixineq", As, False)],
            dict(status="Optimal", primal_obj=1.0, s=[1.0, -1.0]))


def doublynonnegativetri1():   # :493-511 (the loop body resets use_dual = false: only the primal-barrier case runs)
    G = -np.eye(3)
    return (np.array([0.0, 1.0, 0.0]), np.array([[1.0, 0, 0], [0, 0, 1.0]]), np.ones(2), G, np.zeros(3),
            [("doublynonnegativetri", 3, False)], dict(status="Optimal", primal_obj=0.0, x=[1.0, 0.0, 1.0], s=[1.0, 0.0, 1.0]))


def doublynonnegativetri2():   # :513-526
    G = -np.eye(3)
    return (np.array([0.0, -1.0, 0.0]), np.array([[1.0, 0, 0], [0, 0, 1.0]]), np.array([1.0, 1.5]), G, np.array([-0.5, 0.0, -0.5]),
            [("doublynonnegativetri", 3, False)], dict(status="Optimal", primal_obj=-1.0, x_at={1: 1.0}))


def _rand_psd_svec(side, seed, scale=1.0, rank=None):
    rng = np.random.default_rng(seed)
    Mh = scale * rng.random((side, rank or side))
    M = Mh @ Mh.T
    v = np.zeros(side * (side + 1) // 2)
    au.smat_to_svec(v, np.asfortranarray(0.5 * (M + M.T)), au.RT2)
    return v


def _rootdet(v, side):
    m = np.zeros((side, side), order="F")
    au.svec_to_smat(m, np.asarray(v, dtype=float), au.RT2)
    m = np.triu(m) + np.triu(m, 1).T
    return np.linalg.det(m) ** (1.0 / side)


def hyporootdettri1(seed=1):   # :1569-1598 (real case; property-based: u = rootdet(W) at the optimum, primal and dual)
    side = 3
    dim = 1 + side * (side + 1) // 2
    G = np.zeros((dim, 1))
    G[0, 0] = -1.0
    h = np.zeros(dim)
    h[1:] = _rand_psd_svec(side, seed)

    def check(sv, approx):
        assert approx(sv.get_x()[0], -sv.get_primal_obj())
        s, z = sv.get_s(), sv.get_z()
        assert approx(_rootdet(s[1:], side), s[0])
        assert approx(_rootdet(z[1:] * side, side), -z[0])
    return (np.array([-1.0]), np.zeros((0, 1)), np.zeros(0), G, h, [("hyporootdettri", dim, False)], dict(status="Optimal", check=check))


def hyporootdettri2(seed=1):   # :1600-1629 (real case, dual cone)
    side = 4
    dim = 1 + side * (side + 1) // 2
    G = np.zeros((dim, 1))
    G[0, 0] = -1.0
    h = np.zeros(dim)
    h[1:] = _rand_psd_svec(side, seed)

    def check(sv, approx):
        assert approx(sv.get_x()[0], sv.get_primal_obj())
        s, z = sv.get_s(), sv.get_z()
        assert approx(_rootdet(s[1:] * side, side), -s[0])
        assert approx(_rootdet(z[1:], side), z[0])
    return (np.array([1.0]), np.zeros((0, 1)), np.zeros(0), G, h, [("hyporootdettri", dim, True)], dict(status="Optimal", check=check))


def hyporootdettri4():   # :1657-1674
    G = np.zeros((6, 4))
    G[0, 0] = G[1, 1] = G[3, 3] = -1.0
    G[2, 2] = -au.RT2
    G[4, 1] = G[5, 3] = 1.0
    return (np.array([-1.0, 0, 0, 0]), np.zeros((0, 4)), np.zeros(0), G, np.array([0.0, 0, 0, 0, 1, 1]),
            [("hyporootdettri", 4, False), ("nonnegative", 2)],
            dict(status="Optimal", primal_obj=-1.0, x=[1.0, 1, 0, 1], z=[-1.0, 0.5, 0, 0.5, 0.5, 0.5]))


def _logdet(v, side):
    m = np.zeros((side, side), order="F")
    au.svec_to_smat(m, np.asarray(v, dtype=float), au.RT2)
    m = np.triu(m) + np.triu(m, 1).T
    return np.linalg.slogdet(m)[1]


def hypoperlogdettri1(seed=1):   # :1797-1827 (real case; property-based)
    side = 4
    dim = 2 + side * (side + 1) // 2
    G = np.zeros((dim, 2))
    G[0, 0] = G[1, 1] = -1.0
    h = np.zeros(dim)
    rng = np.random.default_rng(seed)
    Mh = rng.random((side, side))
    M = Mh @ Mh.T + np.eye(side)
    au.smat_to_svec(h[2:], np.asfortranarray(0.5 * (M + M.T)), au.RT2)

    def check(sv, approx):
        x, s, z = sv.get_x(), sv.get_s(), sv.get_z()
        assert approx(x[0], -sv.get_primal_obj()) and approx(x[1], 1.0)
        assert approx(s[1] * _logdet(s[2:] / s[1], side), s[0])
        assert approx(z[0] * (_logdet(-z[2:] / z[0], side) + side), z[1])
    return (np.array([-1.0, 0.0]), np.array([[0.0, 1.0]]), np.array([1.0]), G, h, [("hypoperlogdettri", dim, False)],
            dict(status="Optimal", check=check))


def hypoperlogdettri2(seed=1):   # :1829-1859 (real case, dual cone)
    side = 2
    dim = 2 + side * (side + 1) // 2
    G = np.zeros((dim, 2))
    G[0, 0] = G[1, 1] = -1.0
    h = np.zeros(dim)
    h[2:] = _rand_psd_svec(side, seed)

    def check(sv, approx):
        x, s, z = sv.get_x(), sv.get_s(), sv.get_z()
        assert approx(x[1], sv.get_primal_obj()) and approx(x[0], -1.0)
        assert approx(s[0] * (_logdet(-s[2:] / s[0], side) + side), s[1])
        assert approx(z[1] * _logdet(z[2:] / z[1], side), z[0])
    return (np.array([0.0, 1.0]), np.array([[1.0, 0.0]]), np.array([-1.0]), G, h, [("hypoperlogdettri", dim, True)],
            dict(status="Optimal", check=check))


def hypoperlogdettri3(seed=1):   # :1861-1884 (real case)
    side = 3
    dim = 2 + side * (side + 1) // 2
    G = np.zeros((dim, 2))
    G[0, 0] = G[1, 1] = -1.0
    h = np.zeros(dim)
    h[2:] = _rand_psd_svec(side, seed)

    def check(sv, approx):
        x = sv.get_x()
        assert approx(x[0], -sv.get_primal_obj()) and approx(np.linalg.norm(x), 0.0)
    return (np.array([-1.0, 0.0]), np.array([[0.0, 1.0]]), np.array([0.0]), G, h, [("hypoperlogdettri", dim, False)],
            dict(status="Optimal", check=check))


def hypoperlogdettri4():   # :1886-1907
    A = np.zeros((1, 5))
    A[0, 1] = 1.0
    G = np.zeros((7, 5))
    G[0, 0] = G[1, 1] = G[2, 2] = G[4, 4] = -1.0
    G[3, 3] = -au.RT2
    G[5, 2] = G[6, 4] = 1.0
    return (np.array([-1.0, 0, 0, 0, 0]), A, np.array([1.0]), G, np.array([0.0, 0, 0, 0, 0, 1, 1]),
            [("hypoperlogdettri", 5, False), ("nonnegative", 2)],
            dict(status="Optimal", primal_obj=0.0, x=[0.0, 1, 1, 0, 1], y=[-2.0], z=[-1.0, -2, 1, 0, 1, 1, 1]))


def wsosinterppossemideftri1():   # :2385-2405: convexity parameter of (x + 1)^2 (x - 1)^2 on [-1, 1]
    U, pts, Ps = pu.interpolate_box([-1.0], [1.0], 1)
    x = pts[:, 0]
    h = 12 * x ** 2 - 4                      # second derivative of (x^2 - 1)^2
    return (np.array([-1.0]), np.zeros((0, 1)), np.zeros(0), np.ones((U, 1)), h,
            [("wsosinterppossemideftri", 1, U, Ps, False)], dict(status="Optimal", primal_obj=4.0, x=[-4.0]))


KNOWN_ANSWER = {
    "dimension1": dimension1, "primalinfeas1": primalinfeas1, "nonnegative4": nonnegative4,
    "possemideftri1": possemideftri1, "possemideftri2": possemideftri2, "possemideftri3": possemideftri3,
    "possemideftri4": possemideftri4, "possemideftri8": possemideftri8, "possemideftri9": possemideftri9,
    "epinormspectral2_primal": lambda: epinormspectral2(False), "epinormspectral2_dual": lambda: epinormspectral2(True),
    "epinormspectral3_1x1": lambda: epinormspectral3(1, 1, False), "epinormspectral3_1x3_dual": lambda: epinormspectral3(1, 3, True),
    "epinormspectral3_2x2": lambda: epinormspectral3(2, 2, False), "epinormspectral3_3x4_dual": lambda: epinormspectral3(3, 4, True),
    "epinormspectral4_primal": lambda: epinormspectral4(False), "epinormspectral4_dual": lambda: epinormspectral4(True),
    "wsosinterpnonnegative1": wsosinterpnonnegative1, "wsosinterpnonnegative2": wsosinterpnonnegative2,
    "wsosinterpnonnegative3": wsosinterpnonnegative3,
    "linmatrixineq1_side2": lambda: linmatrixineq1(2), "linmatrixineq1_side4": lambda: linmatrixineq1(4),
    "linmatrixineq2": linmatrixineq2, "linmatrixineq3": linmatrixineq3,
    "doublynonnegativetri1": doublynonnegativetri1, "doublynonnegativetri2": doublynonnegativetri2,
    "hyporootdettri1": hyporootdettri1, "hyporootdettri2": hyporootdettri2, "hyporootdettri4": hyporootdettri4,
    "hypoperlogdettri1": hypoperlogdettri1, "hypoperlogdettri2": hypoperlogdettri2, "hypoperlogdettri3": hypoperlogdettri3,
    "hypoperlogdettri4": hypoperlogdettri4,
    "wsosinterppossemideftri1": wsosinterppossemideftri1,
}


# ----------------------------------------------------------------------------------------------
# More of the reference's native instances for the cones of the device path: the ones whose data come from Julia's random
# stream (same construction, numpy's stream: their assertions are status-level or property-based), the preprocessing
# cases (dependent equalities / dependent columns: process.jl:64-365) and the ones with non-default options.  `expect` may
# carry "tol" (the instance's own tolerance), "obj_offset" and "solver_opts" next to the answers.  Not part of
# KNOWN_ANSWER (whose members have golden fixtures).
# ----------------------------------------------------------------------------------------------
def _randint(rng, lo, hi, shape):
    return rng.integers(lo, hi + 1, size=shape).astype(np.float64)


def consistent1(seed=1):   # :110-130 (dependent rows of A with a consistent b, dependent columns of [A; G] with a consistent c)
    rng = np.random.default_rng(seed)
    n, p, q = 30, 15, 30
    c = np.zeros(n)
    A = _randint(rng, -9, 9, (p, n))
    G = 10.0 * np.eye(q, n)
    r1, r2 = rng.random(), rng.random()
    A[10:15, :] = r1 * A[0:5, :] - r2 * A[5:10, :]
    b = A.sum(axis=1)
    r1, r2 = rng.random(), rng.random()
    A[:, 10:15] = r1 * A[:, 0:5] - r2 * A[:, 5:10]
    G[:, 10:15] = r1 * G[:, 0:5] - r2 * G[:, 5:10]
    c[10:15] = r1 * c[0:5] - r2 * c[5:10]
    return (c, A, b, G, np.zeros(q), [("nonnegative", q)], dict(status="Optimal", tol=10 * TEST_TOL))


def inconsistent1(seed=1):   # :132-148 (dependent rows of A, inconsistent b)
    rng = np.random.default_rng(seed)
    n, p, q = 30, 15, 30
    c = _randint(rng, 0, 9, n)
    A = _randint(rng, -9, 9, (p, n))
    b = rng.random(p)
    r1, r2 = rng.random(), rng.random()
    A[10:15, :] = r1 * A[0:5, :] - r2 * A[5:10, :]
    b[10:15] = 2 * (r1 * b[0:5] - r2 * b[5:10])
    return (c, A, b, -np.eye(q, n), np.zeros(q), [("nonnegative", q)], dict(status="PrimalInconsistent"))


def inconsistent2(seed=1):   # :150-167 (dependent columns of [A; G], inconsistent c)
    rng = np.random.default_rng(seed)
    n, p, q = 30, 15, 30
    c = _randint(rng, 0, 9, n)
    A = _randint(rng, -9, 9, (p, n))
    G = -np.eye(q, n)
    b = rng.random(p)
    r1, r2 = rng.random(), rng.random()
    A[:, 10:15] = r1 * A[:, 0:5] - r2 * A[:, 5:10]
    G[:, 10:15] = r1 * G[:, 0:5] - r2 * G[:, 5:10]
    c[10:15] = 2 * (r1 * c[0:5] - r2 * c[5:10])
    return (c, A, b, G, np.zeros(q), [("nonnegative", q)], dict(status="DualInconsistent"))


def nonnegative1(seed=1):   # :249-263 (obj_offset = 1)
    rng = np.random.default_rng(seed)
    n, p, q = 6, 3, 6
    c = _randint(rng, 0, 9, n)
    A = _randint(rng, -9, 9, (p, n))
    return (c, A, A.sum(axis=1), -np.eye(q, n), np.zeros(q), [("nonnegative", q)], dict(status="Optimal", obj_offset=1.0))


def nonnegative2(seed=1):   # :265-278
    rng = np.random.default_rng(seed)
    n, p, q = 5, 2, 10
    c = _randint(rng, 0, 9, n)
    A = _randint(rng, 1, 9, (p, n))
    G = rng.random((q, n)) - 2.0 * np.eye(q, n)
    return (c, A, A.sum(axis=1), G, G.sum(axis=1), [("nonnegative", q)], dict(status="Optimal", tol=2 * TEST_TOL))


def nonnegative3(seed=1):   # :280-293
    rng = np.random.default_rng(seed)
    n, p, q = 15, 6, 15
    c = _randint(rng, 0, 9, n)
    A = _randint(rng, -9, 9, (p, n))
    return (c, A, A.sum(axis=1), -np.eye(q), np.zeros(q), [("nonnegative", q)], dict(status="Optimal", tol=2 * TEST_TOL))


def indirect1(seed=1):   # :2592-2606 with the option set of test/runnativetests.jl:89-99: LSQR initial point, no preprocessing or
    # reduction, loose tolerances; the dense SymIndef system solver stands in for the matrix-free one (out of scope)
    rng = np.random.default_rng(seed)
    n, p = 3, 2
    c = _randint(rng, 0, 9, n)
    A = _randint(rng, -9, 9, (p, n))
    return (c, A, A.sum(axis=1), -np.eye(n), np.zeros(n), [("nonnegative", n)],
            dict(status="Optimal", tol=1e-3, obj_offset=1.0,
                 solver_opts=dict(init_use_indirect=True, preprocess=False, reduce=False, syssolver="symindef", tol_feas=1e-4, tol_rel_opt=1e-4,
                                  tol_abs_opt=1e-4, tol_infeas=1e-6)))


def doublynonnegativetri3():   # :528-541 (despite its name, a PosSemidefTri(3) instance)
    return (np.ones(3), np.array([[1.0, 0, 1]]), np.zeros(1), -np.eye(3), np.zeros(3), [("possemideftri", 3)],
            dict(status="Optimal", primal_obj=0.0, x_norm=0.0))


def epinormspectral1(use_dual, seed=1):   # :1038-1072, real member (property-based on the singular values of s and z)
    rng = np.random.default_rng(seed)
    Xn, Xm = 3, 4
    dim = Xn * Xm
    c = np.concatenate([[1.0], np.zeros(dim)])
    A = np.hstack([np.zeros((dim, 1)), np.eye(dim)])
    b = rng.random(dim)
    h = np.concatenate([[0.0], rng.random(dim)])

    def check(solver, approx):
        s, z = solver.get_s(), solver.get_z()
        psv = np.linalg.svd(s[1:].reshape((Xn, Xm), order="F"), compute_uv=False)
        dsv = np.linalg.svd(z[1:].reshape((Xn, Xm), order="F"), compute_uv=False)
        if use_dual:
            assert approx(np.sum(psv), s[0]) and approx(dsv[0], z[0])
        else:
            assert approx(psv[0], s[0]) and approx(np.sum(dsv), z[0])
    return (c, A, b, -np.eye(dim + 1), h, [("epinormspectral", Xn, Xm, use_dual)], dict(status="Optimal", check=check))


def hyporootdettri3(is_complex=False, seed=1):   # :1631-1657 (W of rank side - 1: the optimum is u = 0; tol = eps^0.15)
    rng = np.random.default_rng(seed)
    side = 3
    if is_complex:
        half = 0.2 * (rng.random((side, side - 1)) + 1j * rng.random((side, side - 1)))
        mat = half @ half.conj().T
        dim = 1 + side * side
        sv = _svec_c(mat)
        spec = ("hyporootdettri_complex", dim, False)
    else:
        half = 0.2 * rng.random((side, side - 1))
        mat = half @ half.T
        dim = 1 + side * (side + 1) // 2
        jj, ii = np.tril_indices(side)
        sv = mat[ii, jj] * np.where(ii == jj, 1.0, RT2)
        spec = ("hyporootdettri", dim, False)
    G = np.zeros((dim, 1))
    G[0, 0] = -1.0
    h = np.zeros(dim)
    h[1:] = sv
    return (np.array([-1.0]), np.zeros((0, 1)), np.zeros(0), G, h, [spec],
            dict(status="Optimal", primal_obj=0.0, x=[0.0], tol=float(np.finfo(np.float64).eps ** 0.15)))


def wsosinterppossemideftri2(seed=1):   # :2407-2427: convexity parameter of x1^4 - 3 x2^2 (Hessian diag(12 x1^2, -6)) on R^2
    U, pts, Ps = pu.interpolate_free(2, 1, np.random.default_rng(seed))
    G = np.concatenate([np.ones(U), np.zeros(U), np.ones(U)]).reshape(-1, 1)
    h = np.concatenate([12 * pts[:, 0] ** 2, np.zeros(U), -6.0 * np.ones(U)])
    return (np.array([-1.0]), np.zeros((0, 1)), np.zeros(0), G, h, [("wsosinterppossemideftri", 2, U, Ps, False)],
            dict(status="Optimal", primal_obj=6.0, x=[-6.0]))


def wsosinterppossemideftri3(seed=1):   # :2429-2450: feasibility of a fixed SOS matrix polynomial -- a model with NO variables
    U, pts, Ps = pu.interpolate_free(1, 3, np.random.default_rng(seed))
    x = pts[:, 0]
    m11, m12 = x + 2 * x ** 3, np.ones(U)
    m21, m22 = -x ** 2 + 2, 3 * x ** 2 - x + 1
    q11 = (m11 * m11 + m21 * m21) / 10
    q21 = (m12 * m11 + m22 * m21) / 10
    q22 = (m12 * m12 + m22 * m22) / 10
    h = np.concatenate([q11, RT2 * q21, q22])
    return (np.zeros(0), np.zeros((0, 0)), np.zeros(0), np.zeros((3 * U, 0)), h, [("wsosinterppossemideftri", 2, U, Ps, False)],
            dict(status="Optimal", primal_obj=0.0))


MORE_NATIVE = {
    "consistent1": consistent1, "inconsistent1": inconsistent1, "inconsistent2": inconsistent2,
    "nonnegative1": nonnegative1, "nonnegative2": nonnegative2, "nonnegative3": nonnegative3, "indirect1": indirect1,
    "doublynonnegativetri3": doublynonnegativetri3,
    "epinormspectral1_primal": lambda: epinormspectral1(False), "epinormspectral1_dual": lambda: epinormspectral1(True),
    "hyporootdettri3": lambda: hyporootdettri3(False), "hyporootdettri3_complex": lambda: hyporootdettri3(True),
    "wsosinterppossemideftri2": wsosinterppossemideftri2, "wsosinterppossemideftri3": wsosinterppossemideftri3,
}


# ----------------------------------------------------------------------------------------------
# Edge cases of our own (not from the reference's test set): shapes at the ends of the ranges the device code blocks by --
# no free variables left after the equalities, cones of dimension one, a PSD side that crosses the 128-wide block size of
# the factorization kernels next to tiny ones, a one-row spectral cone.  Answers are certified by the instance harness
# (residuals, gap, cone membership by definition); the oracle and the HIP path must both pass.
# ----------------------------------------------------------------------------------------------
def edge_p_equals_n(seed=0):
    rng = np.random.default_rng(seed)
    n = 4
    A = rng.standard_normal((n, n)) + 3 * np.eye(n)
    x0 = rng.random(n) + 0.5
    return (rng.random(n), A, A @ x0, -np.eye(n), np.zeros(n), [("nonnegative", n)], dict(status="Optimal", x=list(x0)))


def edge_tiny_cones():
    c = np.array([1.0, 1.0, 1.0, 0.5])
    return (c, np.array([[1.0, 1, 1, 1]]), np.array([2.0]), -np.eye(4), np.zeros(4),
            [("possemideftri", 1), ("nonnegative", 1), ("epinormspectral", 1, 1, False)], dict(status="Optimal", primal_obj=1.5))


def edge_one_row_spectral(use_dual, seed=1):   # EpiNormSpectral(1, 7): the spectral norm of a row is its 2-norm, the nuclear norm too
    rng = np.random.default_rng(seed)
    w = rng.standard_normal(7)
    G = np.zeros((8, 1))
    G[0, 0] = -1.0
    h = np.concatenate([[0.0], w])
    return (np.array([1.0]), np.zeros((0, 1)), np.zeros(0), G, h, [("epinormspectral", 1, 7, use_dual)],
            dict(status="Optimal", primal_obj=float(np.linalg.norm(w))))


def edge_psd_ragged(seed=4):
    return psd_blocks(24, [1, 2, 5, 17, 130], seed=seed)


def edge_infeasible(which):
    """one-cone models without a solution: the certificate statuses on the cones of the device path"""
    if which == "psd_unbounded":        # min -tr X over X >= 0
        return (np.array([-1.0, 0, -1.0]), np.zeros((0, 3)), np.zeros(0), -np.eye(3), np.zeros(3), [("possemideftri", 3)], dict(status="DualInfeasible"))
    if which == "spectral_infeasible":  # u = -1 with u >= ||W||
        return (np.zeros(5), np.array([[1.0, 0, 0, 0, 0]]), np.array([-1.0]), -np.eye(5), np.zeros(5), [("epinormspectral", 2, 2, False)],
                dict(status="PrimalInfeasible"))
    if which == "spectral_unbounded":   # min -u
        return (np.array([-1.0, 0, 0, 0, 0]), np.zeros((0, 5)), np.zeros(0), -np.eye(5), np.zeros(5), [("epinormspectral", 2, 2, False)],
                dict(status="DualInfeasible"))
    if which == "rootdet_unbounded":    # max u under u <= rootdet(W), W free
        return (np.array([-1.0, 0, 0, 0]), np.zeros((0, 4)), np.zeros(0), -np.eye(4), np.zeros(4), [("hyporootdettri", 4, False)],
                dict(status="DualInfeasible"))
    if which == "wsos_infeasible":      # the constant -1 is not a nonnegative polynomial
        U, pts, Ps = pu.interpolate_box([-1.0], [1.0], 2)
        return (np.zeros(1), np.zeros((0, 1)), np.zeros(0), np.zeros((U, 1)), -np.ones(U), [("wsosinterpnonnegative", U, Ps, False)],
                dict(status="PrimalInfeasible"))
    raise ValueError(which)


EDGE_CASES = {
    "psd_unbounded": lambda: edge_infeasible("psd_unbounded"), "spectral_infeasible": lambda: edge_infeasible("spectral_infeasible"),
    "spectral_unbounded": lambda: edge_infeasible("spectral_unbounded"), "rootdet_unbounded": lambda: edge_infeasible("rootdet_unbounded"),
    "wsos_infeasible": lambda: edge_infeasible("wsos_infeasible"),
    "p_equals_n": edge_p_equals_n, "tiny_cones": edge_tiny_cones,
    "one_row_spectral_primal": lambda: edge_one_row_spectral(False), "one_row_spectral_dual": lambda: edge_one_row_spectral(True),
    "psd_ragged_sides": edge_psd_ragged,
}


# ----------------------------------------------------------------------------------------------
# synthetic generators for the BASELINE.json configs
# ----------------------------------------------------------------------------------------------
def svec_identity(side):
    v = np.zeros(side * (side + 1) // 2)
    k = 0
    for i in range(1, side + 1):
        v[k] = 1
        k += i + 1
    return v


def psd_blocks(n, sides, seed=1, dtype=np.float64):
    """configs 2 and 4: product of PosSemidefTri(side_k) cones, dense random G (q x n), p = 0.
    G = randn(q, n)/sqrt(n); h = G x0 + svec(I); c = -G' svec(I)  => strictly feasible primal/dual pair."""
    rng = np.random.default_rng(seed)
    dims = [s * (s + 1) // 2 for s in sides]
    q = sum(dims)
    if dtype == np.float32:   # (draws in single precision, stored in double: half the generation time of the 8.3 GB of config 4)
        G = np.asfortranarray(rng.standard_normal((q, n), dtype=np.float32), dtype=np.float64)
        G /= np.sqrt(n)
    else:
        G = np.asfortranarray(rng.standard_normal((q, n)) / np.sqrt(n))
    x0 = rng.standard_normal(n)
    e = np.concatenate([svec_identity(s) for s in sides])
    h = G @ x0 + e
    c = -(G.T @ e)
    specs = [("possemideftri", d) for d in dims]
    return (c, np.zeros((0, n)), np.zeros(0), G, h, specs, dict(status="Optimal"))


def linearopt(m=50, n=100, seed=1):
    """config 1: examples/linearopt/native.jl:15-30 with nz_frac = 1 (dense): A = 10 rand(m, n), b = A 1,
    c = rand(n), G = -I, h = 0, Nonnegative(n)."""
    rng = np.random.default_rng(seed)
    A = 10 * rng.random((m, n))
    b = A @ np.ones(n)
    c = rng.random(n)
    return (c, A, b, -np.eye(n), np.zeros(n), [("nonnegative", n)], dict(status="Optimal"))


def polymin(nvars, halfdeg, use_primal, seed=1, keep=None):
    """config 5: examples/polymin/native.jl:56-90 (real, WSOS formulation), random_interp_data
    (examples/polymin/data_real.jl:23-33) on the box [-1, 1]^n.  keep: a recorded choice of interpolation points
    (polyutils.choose_interp_pts) for fixtures that must rebuild the same model on another machine."""
    rng = np.random.default_rng(seed)
    U, pts, Ps = pu.interpolate_box([-1.0] * nvars, [1.0] * nvars, halfdeg, rng=rng, keep=keep)
    vals = rng.standard_normal(U)
    if use_primal:
        return (np.array([-1.0]), np.zeros((0, 1)), np.zeros(0), np.ones((U, 1)), vals,
                [("wsosinterpnonnegative", U, Ps, False)], dict(status="Optimal"))
    return (vals, np.ones((1, U)), np.array([1.0]), -np.eye(U), np.zeros(U),
            [("wsosinterpnonnegative", U, Ps, True)], dict(status="Optimal"))


# ---- predefined polynomials with known minima over boxes: examples/polymin/data_real.jl:36-152 (real_poly_data).  Each entry:
# (number of variables, f on the columns of a point matrix, box lower bounds, box upper bounds, true_obj, line range of its branch).
# The domains are NOT the unit box (butcher, caprasse, goldsteinprice, heart, lotkavolterra, reactiondiffusion, rosenbrock,
# schwefel): they exercise the shifted / scaled interpolation of interp_box (src/PolyUtils/realinterp.jl:84-106).  The ball and
# ellipsoid variants of the same file are not restated.
def _goldsteinprice(x):
    return ((1 + (x[0] + x[1] + 1) ** 2 * (19 - 14 * x[0] + 3 * x[0] ** 2 - 14 * x[1] + 6 * x[0] * x[1] + 3 * x[1] ** 2))
            * (30 + (2 * x[0] - 3 * x[1]) ** 2 * (18 - 32 * x[0] + 12 * x[0] ** 2 + 48 * x[1] - 36 * x[0] * x[1] + 27 * x[1] ** 2)))


REAL_POLY = {
    "butcher": (6, lambda x: x[5] * x[1] ** 2 + x[4] * x[2] ** 2 - x[0] * x[3] ** 2 + x[3] ** 3 + x[3] ** 2 - 1 / 3 * x[0] + 4 / 3 * x[3],
                [-1, -0.1, -0.1, -1, -0.1, -0.1], [0, 0.9, 0.5, -0.1, -0.05, -0.03], -1.4393333333, "38-43"),
    "caprasse": (4, lambda x: (-x[0] * x[2] ** 3 + 4 * x[1] * x[2] ** 2 * x[3] + 4 * x[0] * x[2] * x[3] ** 2 + 2 * x[1] * x[3] ** 3
                               + 4 * x[0] * x[2] + 4 * x[2] ** 2 - 10 * x[1] * x[3] - 10 * x[3] ** 2 + 2),
                 [-0.5] * 4, [0.5] * 4, -3.1800966258, "44-49"),
    "goldsteinprice": (2, _goldsteinprice, [-2] * 2, [2] * 2, 3.0, "50-55"),
    "heart": (8, lambda x: (x[0] * x[5] ** 3 - 3 * x[0] * x[5] * x[6] ** 2 + x[2] * x[6] ** 3 - 3 * x[2] * x[6] * x[5] ** 2 + x[1] * x[4] ** 3
                            - 3 * x[1] * x[4] * x[7] ** 2 + x[3] * x[7] ** 3 - 3 * x[3] * x[7] * x[4] ** 2 + 0.9563453),
              [-0.1, 0.4, -0.7, -0.7, 0.1, -0.1, -0.3, -1.1], [0.4, 1, -0.4, 0.4, 0.2, 0.2, 1.1, -0.3], -1.36775, "70-76"),
    "lotkavolterra": (4, lambda x: x[0] * (x[1] ** 2 + x[2] ** 2 + x[3] ** 2 - 1.1) + 1, [-2] * 4, [2] * 4, -20.8, "77-81"),
    "magnetism7": (7, lambda x: (x[0] ** 2 + 2 * x[1] ** 2 + 2 * x[2] ** 2 + 2 * x[3] ** 2 + 2 * x[4] ** 2 + 2 * x[5] ** 2 + 2 * x[6] ** 2 - x[0]),
                   [-1] * 7, [1] * 7, -0.25, "82-86"),
    "motzkin": (2, lambda x: 1 - 48 * x[0] ** 2 * x[1] ** 2 + 64 * x[0] ** 2 * x[1] ** 4 + 64 * x[0] ** 4 * x[1] ** 2, [-1] * 2, [1] * 2, 0.0, "92-96"),
    "reactiondiffusion": (3, lambda x: -x[0] + 2 * x[1] - x[2] - 0.835634534 * x[1] * (1 + x[1]), [-5] * 3, [5] * 3, -36.71269068, "110-114"),
    "robinson": (2, lambda x: (1 + x[0] ** 6 + x[1] ** 6 - x[0] ** 4 * x[1] ** 2 + x[0] ** 4 - x[0] ** 2 * x[1] ** 4 + x[1] ** 4 - x[0] ** 2
                               + x[1] ** 2 + 3 * x[0] ** 2 * x[1] ** 2), [-1] * 2, [1] * 2, 0.814814, "115-120"),
    "rosenbrock": (2, lambda x: (1 - x[0]) ** 2 + 100 * (x[0] ** 2 - x[1]) ** 2, [-5] * 2, [10] * 2, 0.0, "127-131"),
    "schwefel": (3, lambda x: (x[0] - x[1] ** 2) ** 2 + (x[1] - 1) ** 2 + (x[0] - x[2] ** 2) ** 2 + (x[2] - 1) ** 2, [-10] * 3, [10] * 3, 0.0, "137-141"),
}

# the box-domain members of examples/polymin/native_test.jl's "minimal" and "fast" lists: (name, halfdeg, use_primal, use_wsos)
REAL_POLY_INSTANCES = [
    ("butcher", 2, True, True), ("caprasse", 4, True, True), ("goldsteinprice", 7, True, True), ("heart", 2, True, True),
    ("lotkavolterra", 3, True, True), ("magnetism7", 2, True, True), ("motzkin", 3, True, True), ("reactiondiffusion", 4, True, True),
    ("robinson", 8, True, True), ("rosenbrock", 5, True, True), ("schwefel", 2, True, True),
    ("lotkavolterra", 3, False, True), ("motzkin", 3, False, True), ("schwefel", 2, False, True),
    ("lotkavolterra", 3, False, False), ("motzkin", 3, False, False),
]


def polymin_named(name, halfdeg, use_primal, use_wsos=True, seed=1):
    """examples/polymin/native.jl:26-34 (a predefined polynomial: get_interp_data, data_real.jl:10-20) and :56-112 (build_real: the
    WSOS formulation in primal or dual form, and the dual PSD formulation with one PosSemidefTri / Nonnegative block per basis
    matrix).  expect: Optimal and primal_obj = +-true_obj at the example's own tolerance eps^0.1 (native.jl:136-144).  Boxes in 7
    or more variables take the sampling branch of interpolate (realinterp.jl:23-25): the candidates come from numpy's generator
    instead of Julia's -- the optimum does not depend on the points."""
    nv, fn, lo, up, true_obj, _ = REAL_POLY[name]
    rng = np.random.default_rng(seed)
    U, pts, Ps = pu.interpolate_box([float(v) for v in lo], [float(v) for v in up], halfdeg, rng=rng)
    vals = np.array([float(fn(pts[j, :])) for j in range(U)])
    tol = EPS ** 0.1
    expect = dict(status="Optimal", primal_obj=(-1.0 if use_primal else 1.0) * true_obj, tol=tol, true_obj=true_obj)
    if use_primal:
        assert use_wsos, "primal psd formulation is not implemented (native.jl:52-54)"
        return (np.array([-1.0]), np.zeros((0, 1)), np.zeros(0), np.ones((U, 1)), vals,
                [("wsosinterpnonnegative", U, Ps, False)], expect)
    if use_wsos:
        return (vals, np.ones((1, U)), np.array([1.0]), -np.eye(U), np.zeros(U), [("wsosinterpnonnegative", U, Ps, True)], expect)
    # dual PSD formulation (:82-108): rows of G are the scaled lower triangles of -P_k[u, :]' P_k[u, :], one column per point
    specs, blocks = [], []
    nonneg = 0
    for Pk in Ps:
        Lk = Pk.shape[1]
        dk = Lk * (Lk + 1) // 2
        if dk == 1:
            nonneg += 1
        else:
            if nonneg > 0:
                specs.append(("nonnegative", nonneg))
            specs.append(("possemideftri", dk))
        Gk = np.zeros((dk, U))
        l = 0
        for i in range(Lk):
            for j in range(i + 1):
                Gk[l, :] = -Pk[:, i] * Pk[:, j] * (1.0 if i == j else RT2)   # (scale_svec!: off-diagonals times sqrt(2))
                l += 1
        blocks.append(Gk)
    # (the reference pushes a pending Nonnegative cone BEFORE the next PSD cone and once more at the end, without resetting the
    #  count -- every polynomial of the lists has L_k > 1 for all k, so the count stays 0)
    assert nonneg == 0, "a one-dimensional basis block: outside the instances of native_test.jl"
    G = np.vstack(blocks)
    return (vals, np.ones((1, U)), np.array([1.0]), G, np.zeros(G.shape[0]), specs, expect)


def matrixcompletion(d1, d2, seed=1, known_frac=0.8, with_replacement=False):
    """config 3b: examples/matrixcompletion/native.jl:23-70 shape, spectral-norm objective with
    EpiNormSpectral(d1, d2): minimize u s.t. (u, W) in cone, known entries of W fixed.
    with_replacement: the reference's own sampling rule (:29-43): round(0.8 d1 d2) positions drawn WITH replacement and values
    uniform in [-1, 1], so that a fraction 1 - exp(-0.8) = 0.551 of the entries ends up known (n = 1 + 0.449 d1 d2)."""
    rng = np.random.default_rng(seed)
    if with_replacement:
        nk = int(round(d1 * d2 * known_frac))
        rows, cols = rng.integers(0, d1, nk), rng.integers(0, d2, nk)
        kv = 2 * rng.random(nk) - 1
        mask = np.zeros((d1, d2), dtype=bool)
        vals = np.zeros((d1, d2))
        vals[rows, cols] = kv          # (a repeated position keeps its last draw, as the reference's loop does)
        mask[rows, cols] = True
    else:
        mask = rng.random((d1, d2)) < known_frac
        vals = rng.standard_normal((d1, d2))
    unknown = np.argwhere(~mask.reshape(-1, order="F")).ravel()
    nvar = 1 + unknown.shape[0]
    dim = 1 + d1 * d2
    G = np.zeros((dim, nvar), order="F")
    G[0, 0] = -1
    G[1 + unknown, 1 + np.arange(unknown.shape[0])] = -1
    h = np.zeros(dim)
    h[1:] = np.where(mask.reshape(-1, order="F"), vals.reshape(-1, order="F"), 0.0)
    c = np.zeros(nvar)
    c[0] = 1
    return (c, np.zeros((0, nvar)), np.zeros(0), G, h, [("epinormspectral", d1, d2, False)], dict(status="Optimal"))


# ---- complex Hermitian PosSemidefTri (SURVEY 8(f) rank 3 "complex Hermitian variants"): oracle-only so far, kept out of
# KNOWN_ANSWER (whose members all run through the device path and have golden fixtures)
def _svec_c(m):
    from .cones_complex import smat_to_svec_c
    return smat_to_svec_c(np.zeros(m.shape[0] ** 2), m, RT2)


def _rand_herm_c(side, rng):   # Hermitian(rand(Complex, side, side), :U): entries in the unit square, real diagonal
    m = rng.random((side, side)) + 1j * rng.random((side, side))
    u = np.triu(m, 1)
    return np.diag(np.diag(m).real) + u + u.conj().T


def possemideftri5():   # :382-397
    return (np.array([1.0, 0, 0, 1]), np.array([[0.0, 0, 1, 0]]), np.array([1.0]), -np.eye(4), np.zeros(4),
            [("possemideftri_complex", 4)], dict(status="Optimal", primal_obj=RT2, x=[1 / RT2, 0, 1, 1 / RT2]))


def possemideftri6(seed=1):   # :399-416 (objective = largest eigenvalue of a random Hermitian matrix)
    m = _rand_herm_c(2, np.random.default_rng(seed))
    eig_max = float(np.max(np.linalg.eigvalsh(m)))
    return (np.array([1.0]), np.zeros((0, 1)), np.zeros(0), np.array([[-1.0], [0], [0], [-1]]), -_svec_c(m),
            [("possemideftri_complex", 4)], dict(status="Optimal", primal_obj=eig_max, x=[eig_max]))


def possemideftri7(seed=1):   # :418-437
    side = 3
    m = _rand_herm_c(side, np.random.default_rng(seed))
    dim = side * side
    eig_max = float(np.max(np.linalg.eigvalsh(m)))
    return (-_svec_c(m), _svec_c(np.eye(side, dtype=complex)).reshape(1, dim), np.array([1.0]), -np.eye(dim), np.zeros(dim),
            [("possemideftri_complex", dim)], dict(status="Optimal", primal_obj=-eig_max))


def _rvec_c(m):
    from .cones_complex import cvec_to_rvec
    return cvec_to_rvec(np.zeros(2 * m.size), m)


def epinormspectral1_complex(use_dual, seed=1):   # :1038-1072, complex member (property-based on the singular values of s and z)
    from .cones_complex import rvec_to_cmat
    rng = np.random.default_rng(seed)
    Xn, Xm = 3, 4
    dim = 2 * Xn * Xm
    c = np.concatenate([[1.0], np.zeros(dim)])
    A = np.hstack([np.zeros((dim, 1)), np.eye(dim)])
    b = rng.random(dim)
    h = np.concatenate([[0.0], rng.random(dim)])

    def check(solver, approx):
        s, z = solver.get_s(), solver.get_z()
        psv = np.linalg.svd(rvec_to_cmat(s[1:], Xn, Xm), compute_uv=False)
        dsv = np.linalg.svd(rvec_to_cmat(z[1:], Xn, Xm), compute_uv=False)
        if use_dual:
            assert approx(np.sum(psv), s[0]) and approx(dsv[0], z[0])
        else:
            assert approx(psv[0], s[0]) and approx(np.sum(dsv), z[0])
    return (c, A, b, -np.eye(dim + 1), h, [("epinormspectral_complex", Xn, Xm, use_dual)], dict(status="Optimal", check=check))


def epinormspectral2_complex(use_dual, seed=1):   # :1074-1103, complex member
    rng = np.random.default_rng(seed)
    Xn, Xm = 3, 4
    dim = 2 * Xn * Xm
    mat = rng.random((Xn, Xm)) + 1j * rng.random((Xn, Xm))
    sv = np.linalg.svd(mat, compute_uv=False)
    obj = -sv[0] if use_dual else -np.sum(sv)
    return (-_rvec_c(mat), np.zeros((0, dim)), np.zeros(0), np.vstack([np.zeros((1, dim)), -np.eye(dim)]),
            np.concatenate([[1.0], np.zeros(dim)]), [("epinormspectral_complex", Xn, Xm, use_dual)],
            dict(status="Optimal", primal_obj=float(obj)))


def epinormspectral3_complex(Xn, Xm, use_dual):   # :1105-1125, complex members
    dim = 2 * Xn * Xm
    return (-np.ones(dim), np.zeros((0, dim)), np.zeros(0), np.vstack([np.zeros((1, dim)), -np.eye(dim)]),
            np.zeros(dim + 1), [("epinormspectral_complex", Xn, Xm, use_dual)],
            dict(status="Optimal", primal_obj=0.0, x_norm=0.0))


def _rand_c(rng, shape, is_complex):
    m = rng.random(shape)
    return m + 1j * rng.random(shape) if is_complex else m.astype(complex)


def linmatrixineq1_complex(side, seed=1):   # :696-719, complex members (objective = 2 / largest eigenvalue of A_1)
    rng = np.random.default_rng(seed)
    Ah = _rand_c(rng, (side, side), True)
    A1 = Ah @ Ah.conj().T + 2 * np.eye(side)
    A1 = 0.5 * (A1 + A1.conj().T)
    vals, vecs = np.linalg.eigh(A1)
    v1 = vecs[:, -1]
    A2 = -np.outer(v1, v1.conj())
    A2 = 0.5 * (A2 + A2.conj().T)
    G = np.zeros((2, 1))
    G[0, 0] = -1.0
    return (np.array([1.0]), np.zeros((0, 1)), np.zeros(0), G, np.array([0.0, 2.0]), [("linmatrixineq_complex", [A1, A2], False)],
            dict(status="Optimal", primal_obj=2 / vals[-1], s=[2 / vals[-1], 2.0]))


def linmatrixineq2_complex(kinds, seed=1):   # :721-745, the members with complex matrices (kinds: True = complex)
    rng = np.random.default_rng(seed)
    dim = len(kinds)
    As = []
    for is_c in kinds:
        Ah = _rand_c(rng, (3, 3), is_c)
        M = Ah @ Ah.conj().T
        As.append(0.5 * (M + M.conj().T))
    As[0] = As[0] + np.eye(3)
    G = np.vstack([np.zeros((1, dim - 1)), -np.eye(dim - 1)])
    h = np.zeros(dim)
    h[0] = 1.0
    return (np.ones(dim - 1), np.zeros((0, dim - 1)), np.zeros(0), G, h, [("linmatrixineq_complex", As, False)],
            dict(status="Optimal", primal_obj_negative=True))


def _rootdet_c(v, side):
    from .cones_complex import svec_to_smat_c, herm_from_upper
    m = np.zeros((side, side), dtype=complex)
    svec_to_smat_c(m, np.asarray(v, dtype=float), RT2)
    return float(np.linalg.det(herm_from_upper(m)).real ** (1.0 / side))


def _rand_psd_svec_c(side, seed, scale=1.0, rank=None):
    rng = np.random.default_rng(seed)
    Mh = scale * _rand_c(rng, (side, rank or side), True)
    M = Mh @ Mh.conj().T
    return _svec_c(0.5 * (M + M.conj().T))


def hyporootdettri1_complex(seed=1):   # :1569-1598, complex member
    side = 3
    dim = 1 + side * side
    G = np.zeros((dim, 1))
    G[0, 0] = -1.0
    h = np.zeros(dim)
    h[1:] = _rand_psd_svec_c(side, seed)

    def check(sv, approx):
        assert approx(sv.get_x()[0], -sv.get_primal_obj())
        s, z = sv.get_s(), sv.get_z()
        assert approx(_rootdet_c(s[1:], side), s[0])
        assert approx(_rootdet_c(z[1:] * side, side), -z[0])
    return (np.array([-1.0]), np.zeros((0, 1)), np.zeros(0), G, h, [("hyporootdettri_complex", dim, False)],
            dict(status="Optimal", check=check))


def hyporootdettri2_complex(seed=1):   # :1600-1629, complex member (dual cone)
    side = 4
    dim = 1 + side * side
    G = np.zeros((dim, 1))
    G[0, 0] = -1.0
    h = np.zeros(dim)
    h[1:] = _rand_psd_svec_c(side, seed)

    def check(sv, approx):
        assert approx(sv.get_x()[0], sv.get_primal_obj())
        s, z = sv.get_s(), sv.get_z()
        assert approx(_rootdet_c(s[1:] * side, side), -s[0])
        assert approx(_rootdet_c(z[1:], side), z[0])
    return (np.array([1.0]), np.zeros((0, 1)), np.zeros(0), G, h, [("hyporootdettri_complex", dim, True)],
            dict(status="Optimal", check=check))


def _logdet_c(v, side):
    from .cones_complex import svec_to_smat_c, herm_from_upper
    m = np.zeros((side, side), dtype=complex)
    svec_to_smat_c(m, np.asarray(v, dtype=float), RT2)
    return float(np.linalg.slogdet(herm_from_upper(m))[1])


def hypoperlogdettri1_complex(seed=1):   # :1797-1827, complex member
    side = 4
    dim = 2 + side * side
    G = np.zeros((dim, 2))
    G[0, 0] = G[1, 1] = -1.0
    h = np.zeros(dim)
    rng = np.random.default_rng(seed)
    Mh = _rand_c(rng, (side, side), True)
    M = Mh @ Mh.conj().T + np.eye(side)
    h[2:] = _svec_c(0.5 * (M + M.conj().T))

    def check(sv, approx):
        x, s, z = sv.get_x(), sv.get_s(), sv.get_z()
        assert approx(x[0], -sv.get_primal_obj()) and approx(x[1], 1.0)
        assert approx(s[1] * _logdet_c(s[2:] / s[1], side), s[0])
        assert approx(z[0] * (_logdet_c(-z[2:] / z[0], side) + side), z[1])
    return (np.array([-1.0, 0.0]), np.array([[0.0, 1.0]]), np.array([1.0]), G, h, [("hypoperlogdettri_complex", dim, False)],
            dict(status="Optimal", check=check))


def hypoperlogdettri2_complex(seed=1):   # :1829-1859, complex member (dual cone)
    side = 2
    dim = 2 + side * side
    G = np.zeros((dim, 2))
    G[0, 0] = G[1, 1] = -1.0
    h = np.zeros(dim)
    h[2:] = _rand_psd_svec_c(side, seed)

    def check(sv, approx):
        x, s, z = sv.get_x(), sv.get_s(), sv.get_z()
        assert approx(x[1], sv.get_primal_obj()) and approx(x[0], -1.0)
        assert approx(s[0] * (_logdet_c(-s[2:] / s[0], side) + side), s[1])
        assert approx(z[1] * _logdet_c(z[2:] / z[1], side), z[0])
    return (np.array([0.0, 1.0]), np.array([[1.0, 0.0]]), np.array([-1.0]), G, h, [("hypoperlogdettri_complex", dim, True)],
            dict(status="Optimal", check=check))


def hypoperlogdettri3_complex(seed=1):   # :1861-1884, complex member
    side = 3
    dim = 2 + side * side
    G = np.zeros((dim, 2))
    G[0, 0] = G[1, 1] = -1.0
    h = np.zeros(dim)
    h[2:] = _rand_psd_svec_c(side, seed)

    def check(sv, approx):
        x = sv.get_x()
        assert approx(x[0], -sv.get_primal_obj()) and approx(np.linalg.norm(x), 0.0)
    return (np.array([-1.0, 0.0]), np.array([[0.0, 1.0]]), np.array([0.0]), G, h, [("hypoperlogdettri_complex", dim, False)],
            dict(status="Optimal", check=check))


def wsosinterpnonnegative4(seed=1):   # :2345-2363: min of 1 + |z|^2 over the unit disc = 1, complex WSOS certificate
    rng = np.random.default_rng(seed)
    gs = [lambda z: 1.0 - float(np.sum(np.abs(z) ** 2))]
    points, Ps = pu.interpolate_complex(1, 2, gs, [1], rng=rng)
    U = len(points)
    hvals = np.array([1.0 + float(np.sum(np.abs(z) ** 2)) for z in points])
    return (np.array([-1.0]), np.zeros((0, 1)), np.zeros(0), np.ones((U, 1)), hvals,
            [("wsosinterpnonnegative_complex", U, Ps, False)], dict(status="Optimal", primal_obj=-1.0))


def wsosinterpnonnegative5(seed=1):   # :2365-2383: the same minimum over the bidisc, dual form
    rng = np.random.default_rng(seed)
    gs = [lambda z: 1.0 - abs(z[0]) ** 2, lambda z: 1.0 - abs(z[1]) ** 2]
    points, Ps = pu.interpolate_complex(2, 2, gs, [1, 1], rng=rng)
    U = len(points)
    cvals = np.array([1.0 + float(np.sum(np.abs(z) ** 2)) for z in points])
    return (cvals, np.ones((1, U)), np.array([1.0]), -np.eye(U), np.zeros(U),
            [("wsosinterpnonnegative_complex", U, Ps, True)], dict(status="Optimal", primal_obj=1.0))


KNOWN_ANSWER_COMPLEX = {
    "hypoperlogdettri1_complex": hypoperlogdettri1_complex, "hypoperlogdettri2_complex": hypoperlogdettri2_complex,
    "hypoperlogdettri3_complex": hypoperlogdettri3_complex,
    "hyporootdettri1_complex": hyporootdettri1_complex, "hyporootdettri2_complex": hyporootdettri2_complex,
    "linmatrixineq1_complex_side2": lambda: linmatrixineq1_complex(2), "linmatrixineq1_complex_side4": lambda: linmatrixineq1_complex(4),
    "linmatrixineq2_complex_cc": lambda: linmatrixineq2_complex([True, True]),
    "linmatrixineq2_complex_rcr": lambda: linmatrixineq2_complex([False, True, False]),
    "linmatrixineq2_complex_crr": lambda: linmatrixineq2_complex([True, False, False]),
    "possemideftri5": possemideftri5, "possemideftri6": possemideftri6, "possemideftri7": possemideftri7,
    "epinormspectral1_complex_primal": lambda: epinormspectral1_complex(False),
    "epinormspectral1_complex_dual": lambda: epinormspectral1_complex(True),
    "epinormspectral2_complex_primal": lambda: epinormspectral2_complex(False),
    "epinormspectral2_complex_dual": lambda: epinormspectral2_complex(True),
    "epinormspectral3_complex_1x1": lambda: epinormspectral3_complex(1, 1, False),
    "epinormspectral3_complex_1x3_dual": lambda: epinormspectral3_complex(1, 3, True),
    "epinormspectral3_complex_2x2": lambda: epinormspectral3_complex(2, 2, False),
    "epinormspectral3_complex_3x4_dual": lambda: epinormspectral3_complex(3, 4, True),
    "wsosinterpnonnegative4": wsosinterpnonnegative4, "wsosinterpnonnegative5": wsosinterpnonnegative5,
}
