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


def sytrf_rook_lapack(mat, uplo="U"):
    """Raw LAPACK dsytrf_rook on a copy: (factors, ipiv (1-based, LAPACK convention), info)."""
    n = mat.shape[0]
    a = np.array(mat, dtype=np.float64, order="F", copy=True)
    fn = _rook_sym("dsytrf_rook")
    assert fn is not None, "this LAPACK build has no dsytrf_rook"
    ipiv = np.zeros(n, dtype=np.int32)
    lwork = max(1, 64 * n)
    work = np.zeros(lwork)
    info = ctypes.c_int(0)
    fn(ctypes.c_char_p(uplo.encode()), ctypes.byref(ctypes.c_int(n)), a.ctypes.data_as(ctypes.c_void_p),
       ctypes.byref(ctypes.c_int(n)), ipiv.ctypes.data_as(ctypes.c_void_p),
       work.ctypes.data_as(ctypes.c_void_p), ctypes.byref(ctypes.c_int(lwork)), ctypes.byref(info),
       ctypes.c_size_t(1))
    return a, ipiv, info.value


def decode_rook_lower(a, ipiv):
    """LAPACK dsytrf_rook(uplo = 'L') output -> (perm, blk, d, e, L): P A P' = L D L' with perm[i] the original
    index in position i, blk[i] = 0 (1x1) or 1 / 2 (rows of a 2x2 block), D = diag(d) + offdiag(e), L unit lower.
    With uplo = 'L' LAPACK eliminates forwards (columns 1, 2, ...), the order of the device factorization."""
    n = a.shape[0]
    perm = np.arange(n)
    blk = np.zeros(n, dtype=np.int32)
    d = np.diag(a).copy()
    e = np.zeros(n)
    L = np.tril(a, -1) + np.eye(n)

    def swap(r1, r2, k):   # LAPACK stores L as a product P1 L1 P2 L2 ...: later interchanges act on the earlier columns
        if r1 != r2:
            perm[[r1, r2]] = perm[[r2, r1]]
            L[[r1, r2], :k] = L[[r2, r1], :k]

    k = 0
    while k < n:
        if ipiv[k] > 0:
            swap(k, ipiv[k] - 1, k)
            k += 1
        else:
            swap(k, -ipiv[k] - 1, k)
            swap(k + 1, -ipiv[k + 1] - 1, k)
            blk[k], blk[k + 1] = 1, 2
            e[k] = a[k + 1, k]
            L[k + 1, k] = 0.0
            k += 2
    return perm, blk, d, e, L


def ldl_rook_forward(mat):
    """Unblocked restatement of the rook-pivoted symmetric indefinite factorization (the algorithm behind
    bunchkaufman!(A, true), dense.jl:164-165 = LAPACK dsytf2_rook), eliminating forwards.  Reads the UPPER
    triangle of `mat`.  Returns (perm, blk, d, e, L, info) with P A P' = L D L' as in decode_rook_lower."""
    n = mat.shape[0]
    A = np.triu(mat) + np.triu(mat, 1).T
    A = np.array(A, dtype=np.float64)
    alpha = (1.0 + np.sqrt(17.0)) / 8.0
    perm = np.arange(n)
    blk = np.zeros(n, dtype=np.int32)
    d = np.zeros(n)
    e = np.zeros(n)
    L = np.eye(n)
    info = 0

    def swap(a_, b_):
        if a_ == b_:
            return
        A[[a_, b_], :] = A[[b_, a_], :]
        A[:, [a_, b_]] = A[:, [b_, a_]]
        L[[a_, b_], :a_] = L[[b_, a_], :a_]
        perm[[a_, b_]] = perm[[b_, a_]]

    k = 0
    while k < n:
        absakk = abs(A[k, k])
        if k + 1 < n:
            col = np.abs(A[k + 1:, k])
            imax = k + 1 + int(np.argmax(col))
            colmax = col[imax - k - 1]
        else:
            imax, colmax = k, 0.0
        kstep, kp, p, skip = 1, k, k, False
        if max(absakk, colmax) == 0.0 or np.isnan(absakk):
            skip = True
            if info == 0:
                info = k + 1
        elif not (absakk < alpha * colmax):
            kp = k
        else:
            while True:
                row = np.abs(A[imax, k:]).copy()
                row[imax - k] = -1.0
                jmax = k + int(np.argmax(row))
                rowmax = max(row[jmax - k], 0.0)
                if not (abs(A[imax, imax]) < alpha * rowmax):
                    kp, kstep = imax, 1
                    break
                if p == jmax or rowmax <= colmax:
                    kp, kstep = imax, 2
                    break
                p, colmax, imax = imax, rowmax, jmax
        if kstep == 2 and p != k:
            swap(k, p)
        kk = k + kstep - 1
        if kp != kk:
            swap(kk, kp)
        if skip:
            d[k] = A[k, k]
        elif kstep == 1:
            d[k] = A[k, k]
            w = A[k + 1:, k].copy()
            # dsytf2_rook / dlasyf_rook: scale by the reciprocal of the pivot when |pivot| >= sfmin (R1 = ONE / A(K,K), DSCAL)
            l = w * (1.0 / d[k]) if abs(d[k]) >= np.finfo(np.float64).tiny else w / d[k]
            A[k + 1:, k + 1:] -= np.outer(l, w)
            L[k + 1:, k] = l
        else:
            d11, d12, d22 = A[k, k], A[k + 1, k], A[k + 1, k + 1]
            d[k], d[k + 1], e[k] = d11, d22, d12
            blk[k], blk[k + 1] = 1, 2
            D11, D22 = d22 / d12, d11 / d12
            T = 1.0 / (D11 * D22 - 1.0)
            w1 = A[k + 2:, k].copy()
            w2 = A[k + 2:, k + 1].copy()
            l1 = T * (D11 * w1 - w2) / d12
            l2 = T * (D22 * w2 - w1) / d12
            A[k + 2:, k + 2:] -= np.outer(l1, w1) + np.outer(l2, w2)
            L[k + 2:, k] = l1
            L[k + 2:, k + 1] = l2
        k += kstep
    return perm, blk, d, e, L, info


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


def symm_fact_copy(mat):
    """dense.jl:170-184: Bunch-Kaufman (rook) of Symmetric(mat, :U); on failure increase_diag! and again."""
    full = np.triu(mat) + np.triu(mat, 1).T
    fact = bk_rook(full)
    if not fact.success:
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
