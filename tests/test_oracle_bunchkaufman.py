"""The oracle's forward restatement of the rook-pivoted symmetric indefinite factorization (the reference's
fallback of a failed Cholesky: bunchkaufman!(A, true), src/linearalgebra/dense.jl:164-165, 194-215) pinned
against LAPACK dsytrf_rook itself -- the routine Julia calls.  With uplo = 'L' LAPACK eliminates forwards,
the order the device kernel uses, so pivot sequence, D blocks and L can be compared entry by entry."""
import numpy as np
import pytest

from oracle import linalg as la

pytestmark = pytest.mark.skipif(la._rook_sym("dsytrf_rook") is None, reason="LAPACK build without dsytrf_rook")


def sym_cases(n, rng):
    M = rng.standard_normal((n, n))
    A = M + M.T
    yield "indefinite", A
    B = A.copy()
    np.fill_diagonal(B, 0.0)
    yield "zero diagonal (2x2 pivots)", B
    yield "posdef", M @ M.T + 0.1 * np.eye(n)
    C = A.copy()
    C[np.abs(C) < 1.0] = 0.0
    yield "sparse pattern", C + C.T


def dmat(d, e):
    n = len(d)
    D = np.diag(d)
    for k in range(n - 1):
        D[k, k + 1] = D[k + 1, k] = e[k]
    return D


@pytest.mark.parametrize("n", [1, 2, 3, 4, 7, 33, 64, 65, 150])
def test_forward_rook_restatement_matches_lapack(n):
    rng = np.random.default_rng(100 + n)
    for name, A in sym_cases(n, rng):
        a, ipiv, info = la.sytrf_rook_lapack(A, "L")
        perm, blk, d, e, L = la.decode_rook_lower(a, ipiv)
        p2, b2, d2, e2, L2, i2 = la.ldl_rook_forward(A)
        assert info == i2, name
        if info != 0:
            continue
        assert np.array_equal(perm, p2) and np.array_equal(blk, b2), name
        scale = np.abs(A).max()
        assert np.abs(d - d2).max() <= 1e-11 * scale and np.abs(e - e2).max() <= 1e-11 * scale, name
        assert np.abs(L - L2).max() <= 1e-10, name
        assert np.abs(L2).max() <= 1.0 / (1.0 - (1 + np.sqrt(17)) / 8) + 1e-12   # rook pivoting bounds the multipliers
        P = A[np.ix_(p2, p2)]
        assert np.abs(L2 @ dmat(d2, e2) @ L2.T - P).max() <= 1e-13 * n * scale, name


def test_singular_pivot_is_reported_like_lapack():
    for A in (np.zeros((4, 4)), np.diag([1.0, 0.0, 2.0]), np.array([[1.0, 1.0], [1.0, 1.0]])):
        _, _, info = la.sytrf_rook_lapack(A, "L")
        assert la.ldl_rook_forward(A)[5] == info and info > 0


def test_posdef_fact_copy_chain():
    """dense.jl:194-215: Cholesky, else Bunch-Kaufman, else diagonal shift + Bunch-Kaufman."""
    rng = np.random.default_rng(0)
    M = rng.standard_normal((20, 20))
    assert la.posdef_fact_copy(M @ M.T + np.eye(20)).kind == "chol"
    A = M + M.T
    f = la.posdef_fact_copy(A)
    assert f.kind == "bk" and f.success
    b = rng.standard_normal(20)
    assert np.allclose(A @ f.solve(b), b, atol=1e-10)
    f = la.posdef_fact_copy(np.zeros((5, 5)))           # BK finds a zero pivot -> shift -> BK of 1000 eps (1 + 1e-5) I
    assert f.kind == "bk" and f.success
    assert not la.posdef_fact_copy(np.zeros((5, 5)), try_shift=False).success
