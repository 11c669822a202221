"""Dense factorization helpers -- oracle restatement (test infrastructure; see oracle/__init__.py).

Follows /root/reference/src/linearalgebra/dense.jl:
  inv_fact! (Cholesky -> potri)                 :15-22
  outer_prod! (syrk 'U','T')                    :80-86
  increase_diag!                                :106-113
  symm_fact! = bunchkaufman!(A, true) (rook)    :164-165
  posdef_fact_copy! (chol -> BK -> shift + BK)  :194-215
LAPACK comes from scipy's bundled OpenBLAS; dsytrf_rook/dsytrs_rook are reached through ctypes
because scipy.linalg.lapack does not wrap them.
"""
import ctypes
import glob
import os

import numpy as np
import scipy
import scipy.linalg as sla
from scipy.linalg import lapack, blas

EPS = np.finfo(np.float64).eps


def _find_openblas():
    base = os.path.join(os.path.dirname(os.path.dirname(scipy.__file__)), "scipy.libs")
    libs = glob.glob(os.path.join(base, "libscipy_openblas*.so*"))
    if not libs:
        return None
    try:
        return ctypes.CDLL(libs[0])
    except OSError:
        return None


_OB = _find_openblas()


def _rook_sym(name):
    if _OB is None:
        return None
    for cand in ("scipy_" + name + "_", name + "_", "scipy_" + name + "_64_"):
        try:
            return getattr(_OB, cand)
        except AttributeError:
            continue
    return None


class CholFact:
    """Julia `Cholesky` with uplo = 'U' (A = U'U); `info` as LAPACK (0 = success)."""

    kind = "chol"

    def __init__(self, factors, info):
        self.factors = factors
        self.info = int(info)

    @property
    def success(self):
        return self.info == 0

    @property
    def U(self):
        return np.triu(self.factors)

    def solve(self, b):
        x, info = lapack.dpotrs(self.factors, b, lower=0)
        assert info == 0
        return x


class BKFact:
    """Julia `BunchKaufman` with rook pivoting, uplo 'U' (dsytrf_rook / dsytrs_rook)."""

    kind = "bk"

    def __init__(self, LD, ipiv, info):
        self.LD = LD
        self.ipiv = ipiv
        self.info = int(info)

    @property
    def success(self):
        return self.info == 0

    def solve(self, b):
        fn = _rook_sym("dsytrs_rook")
        n = self.LD.shape[0]
        x = np.array(b, dtype=np.float64, order="F", copy=True)
        if x.ndim == 1:
            x = x.reshape(n, 1, order="F")
        nrhs = x.shape[1]
        info = ctypes.c_int(0)
        if fn is not None:
            fn(ctypes.c_char_p(b"U"), ctypes.byref(ctypes.c_int(n)), ctypes.byref(ctypes.c_int(nrhs)),
               self.LD.ctypes.data_as(ctypes.c_void_p), ctypes.byref(ctypes.c_int(n)),
               self.ipiv.ctypes.data_as(ctypes.c_void_p), x.ctypes.data_as(ctypes.c_void_p),
               ctypes.byref(ctypes.c_int(n)), ctypes.byref(info), ctypes.c_size_t(1))
        else:  # standard (non-rook) BK fallback
            x, inf2 = lapack.dsytrs(self.LD, self.ipiv, x, lower=0)
            info.value = inf2
        assert info.value == 0
        return x.reshape(b.shape) if np.ndim(b) == 1 else x


def chol_upper(mat):
    """cholesky!(Hermitian(mat, :U), check = false) on a copy; returns CholFact."""
    c, info = lapack.dpotrf(mat, lower=0, clean=0, overwrite_a=0)
    return CholFact(c, info)


def bk_rook(mat):
    """bunchkaufman!(Symmetric(mat, :U), true, check = false) on a copy (dense.jl:164-165)."""
    n = mat.shape[0]
    a = np.array(mat, dtype=np.float64, order="F", copy=True)
    fn = _rook_sym("dsytrf_rook")
    if fn is None:
        ld, ipiv, info = lapack.dsytrf(a, lower=0)
        return BKFact(np.asfortranarray(ld), ipiv.astype(np.int32), info)
    ipiv = np.zeros(n, dtype=np.int32)
    lwork = max(1, 64 * n)
    work = np.zeros(lwork)
    info = ctypes.c_int(0)
    fn(ctypes.c_char_p(b"U"), ctypes.byref(ctypes.c_int(n)), a.ctypes.data_as(ctypes.c_void_p),
       ctypes.byref(ctypes.c_int(n)), ipiv.ctypes.data_as(ctypes.c_void_p),
       work.ctypes.data_as(ctypes.c_void_p), ctypes.byref(ctypes.c_int(lwork)), ctypes.byref(info),
       ctypes.c_size_t(1))
    return BKFact(a, ipiv, info.value)


def increase_diag(A):
    """dense.jl:106-113."""
    d = np.diagonal(A).copy()
    np.fill_diagonal(A, (1 + 1e-5) * np.maximum(d, 1000 * EPS))
    return A


def posdef_fact_copy(mat, try_shift=True):
    """dense.jl:194-215: Cholesky, else Bunch-Kaufman (rook), else diagonal shift + BK.

    `mat` is read through its upper triangle only (Symmetric(:U)).
    """
    fact = chol_upper(mat)
    if not fact.success:
        full = np.triu(mat) + np.triu(mat, 1).T
        fact = bk_rook(full)
        if try_shift and not fact.success:
            full = np.triu(mat) + np.triu(mat, 1).T
            increase_diag(full)
            fact = bk_rook(full)
    return fact


def inv_fact_chol(fact):
    """inv_fact!(mat, fact::Cholesky) = potri: upper triangle of the inverse (dense.jl:15-22)."""
    inv, info = lapack.dpotri(fact.factors, lower=0)
    assert info == 0
    return inv


def outer_prod(A):
    """outer_prod!(A, B, true, false) = syrk('U','T'): upper triangle of A'A (dense.jl:80-86)."""
    return blas.dsyrk(1.0, A, trans=1, lower=0)
