"""CPU: the complex Hermitian PosSemidefTri oracle (oracle/cones_complex.py) against the reference's own cone tests
(test/cone.jl:335-346: test_oracles for sides 1, 2, 3, 5 and test_barrier with -logdet) and the defining identities of the
complex vectorisation (arrayutilities.jl): round trip, inner products, the Kronecker matrix as the operator D -> M D M."""
import numpy as np
import pytest

from oracle import arrayutil as au
from oracle import cones_complex as occ
from cone_harness import run_test_oracles, run_test_barrier


def _rand_herm(side, rng):
    a = rng.standard_normal((side, side)) + 1j * rng.standard_normal((side, side))
    return a + a.conj().T


def _full(s, side):
    m = np.zeros((side, side), dtype=complex)
    occ.svec_to_smat_c(m, s)
    return occ.herm_from_upper(m)


@pytest.mark.parametrize("side", [1, 2, 3, 5])
def test_possemideftri_complex_oracles(side):   # test/cone.jl:336-340
    run_test_oracles(occ.PosSemidefTriComplex(occ.svec_length_c(side)))


def test_possemideftri_complex_barrier():   # test/cone.jl:342-346
    side = 3
    run_test_barrier(occ.PosSemidefTriComplex(occ.svec_length_c(side)), lambda s: -np.linalg.slogdet(_full(s, side))[1])


@pytest.mark.parametrize("side", [1, 2, 4])
def test_complex_svec_roundtrip_and_inner_product(side):
    rng = np.random.default_rng(side)
    A, B = _rand_herm(side, rng), _rand_herm(side, rng)
    a, b = np.zeros(side * side), np.zeros(side * side)
    occ.smat_to_svec_c(a, A)
    occ.smat_to_svec_c(b, B)
    assert np.allclose(_full(a, side), A, atol=1e-14)
    assert abs(a @ b - np.trace(A @ B).real) < 1e-12 * max(1.0, abs(a @ b))   # the scaling makes svec an isometry
    # (re, -im) of the upper-triangle entry: the second entry of column 1 is -sqrt(2) * imag(A[0, 1])
    if side >= 2:
        assert abs(a[1] - au.RT2 * A[0, 1].real) < 1e-14 and abs(a[2] + au.RT2 * A[0, 1].imag) < 1e-14


@pytest.mark.parametrize("side", [1, 2, 3, 4])
def test_complex_symm_kron_is_the_two_sided_product(side):
    rng = np.random.default_rng(10 + side)
    M = _rand_herm(side, rng)
    dim = side * side
    K = np.zeros((dim, dim))
    occ.symm_kron_c(K, M)
    assert np.allclose(K, K.T)
    for _ in range(3):
        D = _rand_herm(side, rng)
        d, out = np.zeros(dim), np.zeros(dim)
        occ.smat_to_svec_c(d, D)
        occ.smat_to_svec_c(out, M @ D @ M)
        assert np.allclose(K @ d, out, rtol=1e-12, atol=1e-12)


def test_complex_initial_point_is_identity():
    side = 4
    c = occ.PosSemidefTriComplex(side * side)
    c.setup_data()
    p = np.zeros(side * side)
    c.set_initial_point(p)
    assert np.allclose(_full(p, side), np.eye(side))


def _complex_names():
    from oracle import instances as I
    return sorted(I.KNOWN_ANSWER_COMPLEX)


@pytest.mark.parametrize("name", _complex_names())
@pytest.mark.parametrize("reduce", [True, False])
def test_complex_known_answer(name, reduce):   # test/nativeinstances.jl:382-437, :1038-1125 (complex members) through the oracle's solver
    from oracle import instances as I
    from oracle.build import make_model
    from oracle.solvers import Solver
    from instance_harness import build_solve_check
    inst = I.KNOWN_ANSWER_COMPLEX[name]()
    build_solve_check(Solver(default_tol_relax=10, reduce=reduce), make_model(inst), inst)


@pytest.mark.parametrize("d1,d2", [(1, 1), (1, 2), (2, 2), (2, 4), (3, 4)])
def test_epinormspectral_complex_oracles(d1, d2):   # test/cone.jl (EpiNormSpectral{T, R} test_oracles: the same dimension pairs)
    run_test_oracles(occ.EpiNormSpectralComplex(d1, d2))


def test_epinormspectral_complex_barrier():
    d1, d2 = 2, 3

    def barrier(s):
        u = s[0]
        W = occ.rvec_to_cmat(s[1:], d1, d2)
        return -np.linalg.slogdet(u * u * np.eye(d1) - W @ W.conj().T)[1] + (d1 - 1) * np.log(u)

    run_test_barrier(occ.EpiNormSpectralComplex(d1, d2), barrier)


def _rand_herms(side, count, rng):   # test/cone.jl:280-289 (rand_herms, complex members)
    Ah = rng.standard_normal((side, side)) + 1j * rng.standard_normal((side, side))
    As = [Ah @ Ah.conj().T + np.eye(side)]
    for _ in range(count - 1):
        As.append(_rand_herm(side, rng))
    return [0.5 * (A + A.conj().T) for A in As]


@pytest.mark.parametrize("side,count", [(2, 2), (3, 2), (4, 2), (3, 3), (4, 3)])
def test_linmatrixineq_complex_oracles(side, count):   # test/cone.jl:423-429, complex Hermitian members
    rng = np.random.default_rng(side * 10 + count)
    run_test_oracles(occ.LinMatrixIneqComplex(_rand_herms(side, count, rng)), noise=1e-2, init_tol=np.inf)


def test_linmatrixineq_complex_barrier():   # test/cone.jl:431-436
    rng = np.random.default_rng(1)
    Ps = _rand_herms(2, 2, rng)
    run_test_barrier(occ.LinMatrixIneqComplex(Ps), lambda s: -np.linalg.slogdet(sum(s[i] * Ps[i] for i in range(len(Ps))))[1])


@pytest.mark.parametrize("side", [1, 2, 4])
def test_hyporootdettri_complex_oracles(side):   # test/cone.jl:606-610, complex members
    run_test_oracles(occ.HypoRootdetTriComplex(1 + side * side))


def test_hyporootdettri_complex_barrier():   # test/cone.jl:612-620
    side = 3

    def barrier(s):
        sign, logdet = np.linalg.slogdet(_full(s[1:], side))
        return -np.log(np.exp(logdet / side) - s[0]) - logdet

    run_test_barrier(occ.HypoRootdetTriComplex(1 + side * side), barrier)


@pytest.mark.parametrize("side", [1, 2, 4])
def test_hypoperlogdettri_complex_oracles(side):   # test/cone.jl:648-655, complex members
    run_test_oracles(occ.HypoPerLogdetTriComplex(2 + side * side), init_tol=1e-4)   # the central-ray table is a fit


def test_hypoperlogdettri_complex_barrier():   # test/cone.jl:657-665
    side = 3

    def barrier(s):
        u, v = s[0], s[1]
        W = _full(s[2:], side)
        return -np.log(v * np.linalg.slogdet(W / v)[1] - u) - np.log(v) - np.linalg.slogdet(W)[1]

    run_test_barrier(occ.HypoPerLogdetTriComplex(2 + side * side), barrier)


def _rand_interp_complex(num_vars, halfdeg):   # test/cone.jl:306-315: complex Ps on the unit ball (numpy's draws instead of Julia's)
    from oracle import polyutils as pu
    gs = [lambda z: 1.0 - float(np.sum(np.abs(z) ** 2))]
    points, Ps = pu.interpolate_complex(halfdeg, num_vars, gs, [1], rng=np.random.default_rng(1))
    return len(points), Ps


@pytest.mark.parametrize("num_vars,halfdeg", [(1, 1), (1, 3), (2, 1), (2, 2), (3, 1)])
def test_wsosinterpnonnegative_complex_oracles(num_vars, halfdeg):   # test/cone.jl:757-762 with R = Complex
    U, Ps = _rand_interp_complex(num_vars, halfdeg)
    assert U == Ps[0].shape[1] ** 2                                  # U = L^2 (PolyUtils/complex.jl:26-27)
    run_test_oracles(occ.WSOSInterpNonnegativeComplex(U, Ps), init_tol=np.inf)


def test_wsosinterpnonnegative_complex_barrier():   # test/cone.jl:764-768 with R = Complex
    U, Ps = _rand_interp_complex(2, 1)
    run_test_barrier(occ.WSOSInterpNonnegativeComplex(U, Ps),
                     lambda s: -sum(np.linalg.slogdet(P.conj().T @ (s[:, None] * P))[1] for P in Ps))


def test_complex_interpolation_bases():
    """PolyUtils/complex.jl:13-72: U = L^2 points inside the domain, P0 = the monomial columns z^a (first column all ones), weighted
    bases sqrt(g_i) P0[:, 1:L_i]; Lambda(1) = P' P is Hermitian positive definite (a unisolvent point set)"""
    from oracle import polyutils as pu
    gs = [lambda z: 1.0 - abs(z[0]) ** 2, lambda z: 1.0 - abs(z[1]) ** 2]
    pts, Ps = pu.interpolate_complex(2, 2, gs, [1, 1], rng=np.random.default_rng(3))
    assert len(pts) == 36 and [P.shape for P in Ps] == [(36, 6), (36, 3), (36, 3)]
    assert all(g(z) > 0 for z in pts for g in gs)
    assert np.allclose(Ps[0][:, 0], 1.0)
    assert np.allclose(Ps[0][:, 1], [z[0] for z in pts]) and np.allclose(Ps[0][:, 2], [z[1] for z in pts])   # multiexponents order
    assert np.allclose(Ps[1], np.sqrt([gs[0](z) for z in pts])[:, None] * Ps[0][:, :3])
    for P in Ps:
        lam = P.conj().T @ P
        assert np.allclose(lam, lam.conj().T) and np.all(np.linalg.eigvalsh(lam) > 0)
