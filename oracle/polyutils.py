"""Polynomial interpolation utilities -- oracle restatement (test infrastructure; oracle/__init__.py).

Setup-time only (produces the `Ps` matrices of WSOSInterpNonnegative).  Follows
/root/reference/src/PolyUtils/:
  realinterp.jl   interpolate :11-46, interp_sample :52-68, interp_box :84-118, cheb2_pts :121,
                  calc_univariate_chebyshev :123-165, cheb2_data :167-206, padua_data :208-277,
                  approxfekete_data :281-314, make_wsos_arrays :316-331, n_deg_exponents :333,
                  choose_interp_pts :335-371, make_chebyshev_vandermonde :374-397
  realdomains.jl  BoxDomain sample/degree/weights :69-101
Quadrature weights (get_quadr) are not restated: the hot path never uses them.
"""
import itertools
from math import comb

import numpy as np
from scipy.linalg import qr


def get_L(n, d):
    return comb(n + d, n)


def get_U(n, d):
    return comb(n + 2 * d, n)


def prod_consec(n, d, j=0):
    out = 1
    for v in range(2 * d + 1 + j, 2 * d + n + 1):
        out *= v
    return out


def cheb2_pts(k):
    return np.array([-np.cos(np.pi * j / (k - 1)) for j in range(k)])


def calc_univariate_chebyshev(pts_i, d):
    u = np.zeros((pts_i.shape[0], d + 1))
    u[:, 0] = 1
    u[:, 1] = pts_i
    for t in range(2, d + 1):
        u[:, t] = 2 * pts_i * u[:, t - 1] - u[:, t - 2]
    return u


def multiexponents(n, t):
    """Combinatorics.multiexponents(n, t): exponent vectors in with-replacement-combination order."""
    out = []
    for c in itertools.combinations_with_replacement(range(n), t):
        xp = [0] * n
        for i in c:
            xp[i] += 1
        out.append(tuple(xp))
    return out


def n_deg_exponents(n, deg):
    return [xp for t in range(deg + 1) for xp in multiexponents(n, t)]


def make_chebyshev_vandermonde(pts, deg):
    n = pts.shape[1]
    expos = n_deg_exponents(n, deg)
    u = [calc_univariate_chebyshev(pts[:, i], max(deg, 1)) for i in range(n)]
    V = np.zeros((pts.shape[0], len(expos)))
    for col, xp in enumerate(expos):
        v = u[0][:, xp[0]].copy()
        for j in range(1, n):
            v *= u[j][:, xp[j]]
        V[:, col] = v
    return V


LAST_KEEP = None   # the candidate indices the last pivoted QR kept (fixtures record them: see choose_interp_pts)


def choose_interp_pts(cand_pts, d, keep=None):
    """keep = a recorded choice of candidates: the pivot order of LAPACK's dgeqp3 depends on the BLAS build when column norms
    nearly tie, so a fixture that must describe the SAME model on another machine stores the choice instead of redoing it"""
    global LAST_KEEP
    n = cand_pts.shape[1]
    U = get_U(n, d)
    V = make_chebyshev_vandermonde(cand_pts, 2 * d)
    if keep is None:
        # F = qr!(Array(V'), ColumnNorm()); keep_pts = F.p[1:U]
        piv = qr(V.T, mode="r", pivoting=True)[1]
        keep = piv[:U]
    keep = np.asarray(keep, dtype=np.int64)
    assert keep.shape[0] == U
    LAST_KEEP = keep.copy()
    return V[keep, :], keep


def make_wsos_arrays(dom_degree, cand_pts, d, keep=None):
    n = cand_pts.shape[1]
    V, keep = choose_interp_pts(cand_pts, d, keep)
    pts = cand_pts[keep, :]
    P0 = V[:, : get_L(n, d)]
    Lsub = get_L(n, (2 * d - dom_degree) // 2)
    P0sub = P0[:, :Lsub]
    return pts, P0, P0sub


def cheb2_data(d):
    U = get_U(1, d)
    pts = cheb2_pts(U).reshape(-1, 1)
    P0 = make_chebyshev_vandermonde(pts, d)
    return U, pts, P0, P0[:, : get_L(1, d - 1)]


def padua_data(d):
    U = get_U(2, d)
    cheba = cheb2_pts(2 * d + 1)
    chebb = cheb2_pts(2 * d + 2)
    pts = np.zeros((U, 2))
    j = 0
    for a in range(2 * d + 1):
        for b in range(2 * d + 2):
            if (a + b) % 2 == 0:
                pts[j, 0] = -cheba[a]
                pts[U - 1 - j, 1] = -chebb[2 * d + 1 - b]
                j += 1
    P0 = make_chebyshev_vandermonde(pts, d)
    return U, pts, P0, P0[:, : get_L(2, d - 1)]


def approxfekete_data(n, d, keep=None):
    npts = prod_consec(n, d)
    cand = np.zeros((npts, n))
    for j in range(1, n + 1):
        ig = prod_consec(n, d, j)
        cs = cheb2_pts(2 * d + j)
        i = 0
        l = 0
        while True:
            cand[i:i + ig, j - 1] = cs[l]
            i += ig
            l += 1
            if l >= 2 * d + j:
                if i >= npts:
                    break
                l = 0
    pts, P0, P0sub = make_wsos_arrays(2, cand, d, keep)
    return pts.shape[0], pts, P0, P0sub


def interp_box_unit(n, d, keep=None):
    if n == 1:
        return cheb2_data(d)
    if n == 2:
        return padua_data(d)
    return approxfekete_data(n, d, keep)


def interpolate_box(l, u, d, sample=None, rng=None, sample_factor=0, keep=None):
    """interpolate(BoxDomain(l, u), d): returns (U, pts, Ps).  realinterp.jl:11-46."""
    l = np.asarray(l, dtype=np.float64)
    u = np.asarray(u, dtype=np.float64)
    n = l.shape[0]
    U = get_U(n, d)
    if sample is None:
        sample = (n >= 7) or (prod_consec(n, d) > 35000)
    if sample:
        if sample_factor <= 0:
            sample_factor = 10 if U <= 12000 else 5 if U <= 15000 else 2 if U <= 22000 else 1
        rng = rng if rng is not None else np.random.default_rng(1)
        # BoxDomain sample: realdomains.jl:87-95
        cand = (rng.random((U * sample_factor, n)) - 0.5) * (u - l)[None, :] + 0.5 * (u + l)[None, :]
        pts, P0, P0sub = make_wsos_arrays(2, cand, d, keep)
        g = [(pts[:, i] - l[i]) * (u[i] - pts[:, i]) for i in range(n)]   # weights :98-101
        Ps = [P0] + [np.sqrt(gi)[:, None] * P0sub for gi in g]
        return U, pts, Ps
    U2, pts, P0, P0sub = interp_box_unit(n, d, keep)
    pscale = 0.5 * (u - l)
    pshift = 0.5 * (u + l)
    Ps = [P0] + [(np.sqrt(1 - pts[:, j] ** 2) * pscale[j])[:, None] * P0sub for j in range(n)]
    trpts = pts * pscale[None, :] + pshift[None, :]
    return U2, trpts, Ps


def interpolate_free(n, d, rng=None, sample_factor=10):
    """interpolate(FreeDomain(n), d): returns (U, pts, Ps) with Ps = [P0].  A domain that is not a box always takes the
    sampling branch (realinterp.jl:23-25, 49-67): candidates uniform in [-1, 1]^n (realdomains.jl:57-58), points chosen
    by the pivoted QR of the Chebyshev Vandermonde matrix, no weight polynomials (realdomains.jl:62)."""
    U = get_U(n, d)
    rng = rng if rng is not None else np.random.default_rng(1)
    cand = 2.0 * rng.random((U * sample_factor, n)) - 1.0
    pts, P0, _ = make_wsos_arrays(0, cand, d)
    return U, pts, [P0]


def interpolate_complex(halfdeg, n, gs, g_halfdegs, rng=None, sample_factor=10, keep=None):
    """interpolate(Complex{T}, halfdeg, n, gs, g_halfdegs): src/PolyUtils/complex.jl:13-72 (use_qr = false).  Returns (points, Ps):
    U = L^2 points in C^n chosen from sample_factor * U random points of the domain {z in the unit box : g(z) > 0 for g in gs} by
    the pivoted QR of the transposed Vandermonde matrix of the basis z^a conj(z)^b, and the complex bases P0 = the monomial
    columns, P_i = Diagonal(sqrt(g_i(points))) P0[:, 1:L_i].  The random draws come from numpy's generator instead of Julia's
    (the instances built on it assert optimal values that do not depend on the points)."""
    global LAST_KEEP
    rng = rng if rng is not None else np.random.default_rng(1)
    L = comb(n + halfdeg, n)
    U = L * L
    L_basis = [a for t in range(halfdeg + 1) for a in multiexponents(n, t)]

    def mon_pow(z, ex):
        out = 1.0 + 0.0j
        for i, e in enumerate(ex):
            out *= z[i] ** e
        return out

    num_samples = sample_factor * U
    samples = []
    while len(samples) < num_samples:                                   # :37-47
        z = np.array([complex(2 * rng.random() - 1, 2 * rng.random() - 1) for _ in range(n)])
        if all(g(z) > 0 for g in gs):
            samples.append(z)
    # V[s, l * L + k] = z^L_basis[k] * conj(z)^L_basis[l]  (:28-31: l outer, k inner)
    V = np.zeros((num_samples, U), dtype=complex)
    for si, z in enumerate(samples):
        zp = np.array([mon_pow(z, a) for a in L_basis])
        V[si, :] = np.outer(np.conj(zp), zp).reshape(-1)               # row-major (l, k) -> l * L + k
    if keep is None:
        piv = qr(V.T, mode="r", pivoting=True)[1]                       # qr(Matrix(transpose(V)), ColumnNorm()), :52
        keep = piv[:U]
    keep = np.asarray(keep, dtype=np.int64)
    LAST_KEEP = keep.copy()
    points = [samples[i] for i in keep]
    V = V[keep, :]
    P0 = V[:, :L]                                                       # :57
    Ps = [np.asfortranarray(P0)]
    for g, ghd in zip(gs, g_halfdegs):                                  # :62-69
        gi = np.array([g(z) for z in points], dtype=float)
        Li = comb(n + halfdeg - ghd, n)
        Ps.append(np.asfortranarray(np.sqrt(gi)[:, None] * P0[:, :Li]))
    return points, Ps
