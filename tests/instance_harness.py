"""build_solve_check of the reference (test/nativeinstances.jl:32-86) restated; shared by the oracle
and HIP-path instance tests."""
import numpy as np

EPS = np.finfo(np.float64).eps
TEST_TOL = np.sqrt(np.sqrt(EPS))   # test_tol(T), nativeinstances.jl:29


def approx(a, b, tol):
    a = np.asarray(a, dtype=float)
    b = np.asarray(b, dtype=float)
    return np.all(np.abs(a - b) <= tol + tol * np.maximum(np.abs(a), np.abs(b)))


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
