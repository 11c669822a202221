"""Pin the CPU oracle cones with the reference's own cone tests
(/root/reference/test/cone.jl:23-114 identities at 1e3*eps; closed-form barriers :327-330, 342-346,
505-513, 764-768; sizes from :321-325, 336-340, 499-503, 757-762)."""
import numpy as np
import pytest

from oracle import cones as oc
from oracle import arrayutil as au
from oracle import polyutils as pu
from cone_harness import run_test_oracles, run_test_barrier


@pytest.mark.parametrize("d", [1, 2, 6])
def test_nonnegative_oracles(d):
    run_test_oracles(oc.Nonnegative(d))


def test_nonnegative_barrier():
    run_test_barrier(oc.Nonnegative(3), lambda s: -np.sum(np.log(s)))


@pytest.mark.parametrize("side", [1, 2, 3, 5])
def test_possemideftri_oracles(side):
    run_test_oracles(oc.PosSemidefTri(au.svec_length(side)))


def _smat_full(s, side):
    m = np.zeros((side, side))
    au.svec_to_smat(m, s)
    return np.triu(m) + np.triu(m, 1).T


def test_possemideftri_barrier():
    side = 3
    run_test_barrier(oc.PosSemidefTri(au.svec_length(side)), lambda s: -np.linalg.slogdet(_smat_full(s, side))[1])


@pytest.mark.parametrize("d1,d2", [(1, 1), (1, 2), (2, 2), (2, 4), (3, 4)])
def test_epinormspectral_oracles(d1, d2):
    run_test_oracles(oc.EpiNormSpectral(d1, d2))


def test_epinormspectral_barrier():
    d1, d2 = 2, 3

    def barrier(s):
        u = s[0]
        W = s[1:].reshape(d1, d2, order="F")
        return -np.linalg.slogdet(u * u * np.eye(d1) - W @ W.T)[1] + (d1 - 1) * np.log(u)

    run_test_barrier(oc.EpiNormSpectral(d1, d2), barrier)


@pytest.mark.parametrize("nvars,halfdeg", [(1, 1), (1, 3), (2, 1), (2, 2), (3, 1)])
def test_wsosinterpnonnegative_oracles(nvars, halfdeg):
    U, _, Ps = pu.interpolate_box([-1.0] * nvars, [1.0] * nvars, halfdeg, sample=False)
    run_test_oracles(oc.WSOSInterpNonnegative(U, Ps), init_tol=np.inf)


def test_wsosinterpnonnegative_barrier():
    U, _, Ps = pu.interpolate_box([-1.0, -1.0], [1.0, 1.0], 1, sample=False)
    run_test_barrier(oc.WSOSInterpNonnegative(U, Ps),
                     lambda s: -sum(np.linalg.slogdet(P.T @ (s[:, None] * P))[1] for P in Ps))


def test_svec_roundtrip_and_order():
    # arrayutilities.jl:163-181, 218-236: column-major upper triangle, sqrt(2) off-diagonals
    side = 4
    m = np.arange(16, dtype=float).reshape(4, 4)
    m = m + m.T
    v = np.zeros(au.svec_length(side))
    au.smat_to_svec(v, m)
    assert v[0] == m[0, 0] and np.isclose(v[1], m[0, 1] * np.sqrt(2)) and v[2] == m[1, 1]
    assert np.isclose(v[au.svec_idx(3, 1)], m[1, 3] * np.sqrt(2))
    m2 = np.zeros((4, 4))
    au.svec_to_smat(m2, v)
    assert np.allclose(np.triu(m2), np.triu(m))
    # svec is an isometry: <svec(A), svec(B)> = tr(AB)
    b = np.random.default_rng(0).standard_normal((4, 4)); b = b + b.T
    vb = np.zeros(10); au.smat_to_svec(vb, b)
    assert np.isclose(v @ vb, np.trace(m @ b))


def _rand_syms(side, count, rng):   # test/cone.jl:280-289 (rand_herms, real members)
    Ah = rng.standard_normal((side, side))
    As = [Ah @ Ah.T + np.eye(side)]
    for _ in range(count - 1):
        M = rng.standard_normal((side, side))
        As.append(np.triu(M) + np.triu(M, 1).T)
    return [0.5 * (A + A.T) for A in As]


@pytest.mark.parametrize("side,count", [(2, 2), (3, 2), (4, 2), (3, 3), (4, 3)])
def test_linmatrixineq_oracles(side, count):   # test/cone.jl:423-429
    rng = np.random.default_rng(side * 10 + count)
    run_test_oracles(oc.LinMatrixIneq(_rand_syms(side, count, rng)), noise=1e-2, init_tol=np.inf)


def test_linmatrixineq_barrier():   # test/cone.jl:431-436
    rng = np.random.default_rng(1)
    Ps = _rand_syms(2, 2, rng)
    run_test_barrier(oc.LinMatrixIneq(Ps), lambda s: -np.linalg.slogdet(sum(s[i] * Ps[i] for i in range(len(Ps))))[1])


@pytest.mark.parametrize("side", [1, 2, 5])
def test_doublynonnegativetri_oracles(side):   # test/cone.jl:353-361
    run_test_oracles(oc.DoublyNonnegativeTri(au.svec_length(side)), init_tol=np.sqrt(np.finfo(float).eps))


@pytest.mark.parametrize("side", [10, 20])
def test_doublynonnegativetri_initial_point(side):
    run_test_oracles(oc.DoublyNonnegativeTri(au.svec_length(side)), init_tol=np.sqrt(np.finfo(float).eps), init_only=True)


def test_doublynonnegativetri_barrier():   # test/cone.jl:363-371
    side = 3
    cone = oc.DoublyNonnegativeTri(au.svec_length(side))
    od = cone.offdiag_idxs
    run_test_barrier(cone, lambda s: -np.linalg.slogdet(_smat_full(s, side))[1] - np.sum(np.log(s[od])))


@pytest.mark.parametrize("side", [1, 2, 4])
def test_hyporootdettri_oracles(side):   # test/cone.jl:606-610
    run_test_oracles(oc.HypoRootdetTri(1 + au.svec_length(side)))


def test_hyporootdettri_barrier():   # test/cone.jl:612-620
    side = 3

    def barrier(s):
        logdet = np.linalg.slogdet(_smat_full(s[1:], side))[1]
        return -np.log(np.exp(logdet / side) - s[0]) - logdet
    run_test_barrier(oc.HypoRootdetTri(1 + au.svec_length(side)), barrier)


@pytest.mark.parametrize("side", [1, 2, 4])
def test_hypoperlogdettri_oracles(side):   # test/cone.jl:648-655
    run_test_oracles(oc.HypoPerLogdetTri(2 + au.svec_length(side)), init_tol=1e-4)


@pytest.mark.parametrize("side", [8, 12])
def test_hypoperlogdettri_initial_point(side):
    run_test_oracles(oc.HypoPerLogdetTri(2 + au.svec_length(side)), init_tol=1e-1, init_only=True)


def test_hypoperlogdettri_barrier():   # test/cone.jl:657-665
    side = 3

    def barrier(s):
        u, v = s[0], s[1]
        W = _smat_full(s[2:], side)
        return -np.log(v * np.linalg.slogdet(W / v)[1] - u) - np.log(v) - np.linalg.slogdet(W)[1]
    run_test_barrier(oc.HypoPerLogdetTri(2 + au.svec_length(side)), barrier)


@pytest.mark.parametrize("nvars,halfdeg,R", [(1, 1, 1), (1, 1, 4), (2, 2, 1), (3, 1, 2)])
def test_wsosinterppossemideftri_oracles(nvars, halfdeg, R):   # test/cone.jl:775-780
    U, _, Ps = pu.interpolate_box([-1.0] * nvars, [1.0] * nvars, halfdeg, sample=False)
    run_test_oracles(oc.WSOSInterpPosSemidefTri(R, U, Ps), init_tol=np.inf)


def test_wsosinterppossemideftri_barrier():   # test/cone.jl:782-803
    U, _, Ps = pu.interpolate_box([-1.0], [1.0], 1, sample=False)
    R = 3
    cone = oc.WSOSInterpPosSemidefTri(R, U, Ps)

    def barrier(s):
        return -sum(np.linalg.slogdet(cone._block_matrix(s, Pk))[1] for Pk in Ps)
    run_test_barrier(cone, barrier)


@pytest.mark.parametrize("side", [3, 8, 17])
def test_possemideftri_proximity_identity_of_the_candidate_screen(side):
    """the device's side-by-side candidate screen evaluates PosSemidefTri's proximity value in an inverse-free
    form: with smat(s) = U'U and v = z / sqrt(mu) + g(s / sqrt(mu)),  <v, H^-1 v> = || U Z U' / mu - I ||_F^2  (U the factor of
    the UNSCALED primal point; = ||W||^2 / mu^2 - 2 tr W / mu + side for W = U Z U').  Here against the oracle's get_proxsqr
    (Cones.jl:294-310) on points near and far from the central path."""
    rng = np.random.default_rng(side)
    dim = au.svec_length(side)
    for spread in (0.05, 0.5, 3.0):
        A = rng.standard_normal((side, side))
        S = A @ A.T / side + np.eye(side)
        w, V = np.linalg.eigh(S)
        mu = 0.37
        Z = (V * (mu / w * np.exp(spread * rng.standard_normal(side)))) @ V.T     # Z S / mu = I up to the spread
        Z = (Z + Z.T) / 2
        s, z = np.zeros(dim), np.zeros(dim)
        au.smat_to_svec(s, np.triu(S))
        au.smat_to_svec(z, np.triu(Z))
        irtmu = 1.0 / np.sqrt(mu)
        cone = oc.PosSemidefTri(dim)
        cone.setup_data()
        cone.load_point(s, irtmu)
        cone.load_dual_point(z)
        cone.reset_data()
        assert cone.is_feas()
        ref = cone.get_proxsqr(irtmu, True)
        U = np.linalg.cholesky(S).T                                               # S = U'U
        W = U @ Z @ U.T
        direct = np.linalg.norm(W / mu - np.eye(side), "fro") ** 2
        expanded = np.sum(W * W) / mu ** 2 - 2.0 * np.trace(W) / mu + side
        assert abs(direct - ref) <= 1e-9 * (1 + ref), (side, spread, direct, ref)
        assert abs(expanded - ref) <= 1e-9 * (1 + ref), (side, spread, expanded, ref)
