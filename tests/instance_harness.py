"""build_solve_check of the reference (test/nativeinstances.jl:32-86) restated; shared by the oracle
and HIP-path instance tests."""
import numpy as np

EPS = np.finfo(np.float64).eps
TEST_TOL = np.sqrt(np.sqrt(EPS))   # test_tol(T), nativeinstances.jl:29


def approx(a, b, tol):
    a = np.asarray(a, dtype=float)
    b = np.asarray(b, dtype=float)
    return np.all(np.abs(a - b) <= tol + tol * np.maximum(np.abs(a), np.abs(b)))


def _smat(v):
    """svec -> symmetric matrix (columns of the upper triangle, off-diagonal entries scaled by sqrt 2: arrayutilities.jl)"""
    side = int(round((np.sqrt(8 * len(v) + 1) - 1) / 2))
    M = np.zeros((side, side))
    k = 0
    for j in range(side):
        for i in range(j + 1):
            M[i, j] = M[j, i] = v[k] if i == j else v[k] / np.sqrt(2.0)
            k += 1
    return M


def check_membership(specs, s, z, tol):
    """Solver-independent half of the optimality certificate: s in K and z in K* by the cones' DEFINITIONS (eigenvalues,
    singular values), not by any barrier code.  With the residual checks of build_solve_check (primal / dual feasibility, zero
    gap) this certifies the optimum whatever route the iterates took.  Cones without a cheap definition-level test are skipped
    (for WSOSInterpNonnegative only the moment-matrix side is tested)."""
    off = 0
    for spec in specs:
        kind = spec[0]
        dim = {"nonnegative": lambda: spec[1], "possemideftri": lambda: spec[1], "epinormspectral": lambda: 1 + spec[1] * spec[2],
               "wsosinterpnonnegative": lambda: spec[1]}.get(kind)
        if dim is None:
            return            # (offsets of later cones would need this cone's dimension: stop at the first unknown kind)
        dim = dim()
        sk, zk = s[off:off + dim], z[off:off + dim]
        off += dim
        sc = tol * max(1.0, float(np.max(np.abs(sk))), float(np.max(np.abs(zk))))
        if kind == "nonnegative":
            assert sk.min() >= -sc and zk.min() >= -sc
        elif kind == "possemideftri":
            assert np.linalg.eigvalsh(_smat(sk)).min() >= -sc and np.linalg.eigvalsh(_smat(zk)).min() >= -sc
        elif kind == "epinormspectral":
            d1, d2, use_dual = spec[1], spec[2], spec[3]
            prim, dual = (zk, sk) if use_dual else (sk, zk)      # prim in the spectral-norm cone, dual in the nuclear-norm cone
            assert prim[0] >= np.linalg.svd(prim[1:].reshape((d1, d2), order="F"), compute_uv=False)[0] - sc
            assert dual[0] >= np.linalg.svd(dual[1:].reshape((d1, d2), order="F"), compute_uv=False).sum() - sc
        elif kind == "wsosinterpnonnegative":
            Ps, use_dual = spec[2], spec[3]
            mom = sk if use_dual else zk                          # the side that lives in the dual (moment) cone
            for P in Ps:
                assert np.linalg.eigvalsh(P.T @ (mom[:, None] * P)).min() >= -sc * max(1.0, float(np.abs(P).max()) ** 2)


def build_solve_check(solver, model, inst, tol=TEST_TOL):
    c, A, b, G, h = inst[:5]
    expect = inst[6]
    tol = expect.get("tol", tol)                 # (instances with a tolerance of their own)
    offset = expect.get("obj_offset", 0.0)
    solver.load(model)
    solver.solve()
    status = solver.get_status()
    assert status == expect["status"], f"status {status} != {expect['status']}"
    p_obj, d_obj = solver.get_primal_obj(), solver.get_dual_obj()
    x, y, z, s = solver.get_x(), solver.get_y(), solver.get_z(), solver.get_s()
    rt_tol = np.sqrt(tol)
    if status == "Optimal":
        assert approx(p_obj, d_obj, tol)
        assert approx(c @ x + offset, p_obj, tol)
        assert approx(-(b @ y) - h @ z + offset, d_obj, tol)
        assert approx(A @ x, b, tol)
        assert approx(G @ x + s, h, tol)
        assert approx(G.T @ z + A.T @ y, -c, tol)
        assert approx(s @ z, 0.0, rt_tol)
        check_membership(inst[5], s, z, tol)
    elif status == "PrimalInfeasible":
        assert approx(-(b @ y) - h @ z, d_obj, tol)
        assert approx(G.T @ z, -A.T @ y, rt_tol)
    elif status == "DualInfeasible":
        assert approx(c @ x, p_obj, tol)
        assert approx(G @ x, -s, rt_tol)
        assert approx(A @ x, np.zeros(len(y)), rt_tol)
    if "primal_obj" in expect:
        assert approx(p_obj, expect["primal_obj"], tol), (p_obj, expect["primal_obj"])
    if "check" in expect:   # property-based expectations of the reference's test (a callable of the solver)
        expect["check"](solver, lambda a, b: approx(a, b, tol))
    if expect.get("primal_obj_negative"):
        assert p_obj < 0, p_obj
    if "x" in expect:
        assert approx(x, expect["x"], tol), x
    if "x_at" in expect:
        for i, v in expect["x_at"].items():
            assert approx(x[i], v, tol)
    if "x_norm" in expect:
        assert approx(np.linalg.norm(x), expect["x_norm"], tol)
    if "y" in expect:
        assert approx(y, expect["y"], tol), y
    if "s" in expect:
        assert approx(s, expect["s"], tol), s
    if "z" in expect:
        assert approx(z, expect["z"], tol), z
    return solver
