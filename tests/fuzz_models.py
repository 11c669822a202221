"""Random strictly feasible conic models over a mix of the device cones (test helper): s0 = the cones' initial points, z0 = -grad
there (interior of the dual cones), h = G x0 + s0, c = -G' z0 - A' y0, b = A x0, so that primal and dual are strictly feasible and
the solve must end Optimal.  Cone kinds and sizes are drawn from the seed; k scales the size ranges."""
import numpy as np


def _spec(kind, rng, k=1):
    if kind == "nonnegative":
        return ("nonnegative", int(rng.integers(1, 6 * k)))
    if kind == "possemideftri":
        s = int(rng.integers(1, 7 * k))
        return ("possemideftri", s * (s + 1) // 2)
    if kind == "epinormspectral":
        d1 = int(rng.integers(1, 4 * k))
        return ("epinormspectral", d1, d1 + int(rng.integers(0, 4 * k)), bool(rng.integers(0, 2)))
    if kind == "doublynonnegativetri":
        s = int(rng.integers(1, 5 * k))
        return ("doublynonnegativetri", s * (s + 1) // 2, bool(rng.integers(0, 2)))
    if kind == "hyporootdettri":
        s = int(rng.integers(1, 5 * k))
        return ("hyporootdettri", 1 + s * (s + 1) // 2, bool(rng.integers(0, 2)))
    if kind == "hypoperlogdettri":
        s = int(rng.integers(1, 5 * k))
        return ("hypoperlogdettri", 2 + s * (s + 1) // 2, bool(rng.integers(0, 2)))
    if kind == "linmatrixineq":
        side = int(rng.integers(2, 5 * k))
        m = int(rng.integers(2, min(4, side * (side + 1) // 2) + 1))   # (linmatrixineq.jl:56: dim <= svec length)
        As = []
        for i in range(m):
            M = rng.standard_normal((side, side))
            As.append(0.5 * (M + M.T) + (side + 1.0) * np.eye(side) * (1.0 if i == 0 else 0.0))
        return ("linmatrixineq", As, bool(rng.integers(0, 2)))
    if kind == "wsosinterpnonnegative":
        from oracle import polyutils as pu
        nv, hd = int(rng.integers(1, 3)), int(rng.integers(1, 3))
        U, _, Ps = pu.interpolate_box([-1.0] * nv, [1.0] * nv, hd, sample=False)
        return ("wsosinterpnonnegative", U, Ps, bool(rng.integers(0, 2)))
    if kind == "wsosinterppossemideftri":
        from oracle import polyutils as pu
        U, _, Ps = pu.interpolate_box([-1.0], [1.0], int(rng.integers(1, 3)), sample=False)
        return ("wsosinterppossemideftri", int(rng.integers(1, 4)), U, Ps, bool(rng.integers(0, 2)))
    if kind == "possemideftri_complex":
        s = int(rng.integers(1, 5 * k))
        return ("possemideftri_complex", s * s)
    if kind == "epinormspectral_complex":
        d1 = int(rng.integers(1, 3 * k))
        return ("epinormspectral_complex", d1, d1 + int(rng.integers(0, 3 * k)), bool(rng.integers(0, 2)))
    if kind == "hyporootdettri_complex":
        s = int(rng.integers(1, 4 * k))
        return ("hyporootdettri_complex", 1 + s * s, bool(rng.integers(0, 2)))
    if kind == "hypoperlogdettri_complex":
        s = int(rng.integers(1, 4 * k))
        return ("hypoperlogdettri_complex", 2 + s * s, bool(rng.integers(0, 2)))
    raise ValueError(kind)


ALL_KINDS = ["nonnegative", "possemideftri", "epinormspectral", "doublynonnegativetri", "hyporootdettri", "hypoperlogdettri", "linmatrixineq",
             "wsosinterpnonnegative", "wsosinterppossemideftri", "possemideftri_complex", "epinormspectral_complex", "hyporootdettri_complex",
             "hypoperlogdettri_complex"]
KINDS = ["nonnegative", "possemideftri", "epinormspectral", "doublynonnegativetri", "hyporootdettri", "hypoperlogdettri", "linmatrixineq"]


def random_model(seed, make_cone, k=1, kinds=None):
    """-> instance tuple (c, A, b, G, h, specs, expect); make_cone builds a cone object (oracle or HIP) from a spec"""
    rng = np.random.default_rng(seed)
    ncones = int(rng.integers(1, 5))
    kinds = kinds or KINDS
    specs = [_spec(kinds[int(rng.integers(0, len(kinds)))], rng, k) for _ in range(ncones)]
    s0, z0 = [], []
    for sp in specs:
        cone = make_cone(sp)
        cone.setup_data()
        cone.reset_data()
        p = np.zeros(cone.dimension())
        cone.set_initial_point(p)
        cone.load_point(p)
        assert cone.is_feas()
        g = -np.array(cone.get_grad())
        if cone.use_dual_barrier() if callable(getattr(cone, "use_dual_barrier", None)) else getattr(cone, "use_dual_barrier", False):
            p, g = g, p                       # the barrier lives on the dual cone: the roles of s and z swap
        s0.append(p)
        z0.append(g)
    s0, z0 = np.concatenate(s0), np.concatenate(z0)
    q = len(s0)
    n = int(rng.integers(1, max(2, min(q, 8 * k)) + 1))
    p = int(rng.integers(0, n))               # p < n
    G = rng.standard_normal((q, n))
    A = rng.standard_normal((p, n))
    x0, y0 = rng.standard_normal(n), rng.standard_normal(p)
    return (-(G.T @ z0) - A.T @ y0, A, A @ x0, G, G @ x0 + s0, specs, dict(status="Optimal"))
