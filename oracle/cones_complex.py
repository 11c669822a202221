"""TEST INFRASTRUCTURE (CPU oracle), not the product: complex Hermitian variant of PosSemidefTri, the first item of
SURVEY 8(f) rank 3 that the device path does not cover yet ("complex Hermitian variants").  Restates
reference src/Cones/possemideftri.jl:9-207 for R = Complex{Float64} and the complex vectorisation helpers of
src/Cones/arrayutilities.jl:13,81,103-108 (lengths), :188-210 (smat_to_svec!), :240-262 (svec_to_smat!), :308-352 (symm_kron!),
:366-383 (spectral_kron_element!).  Parity pinned by the reference's own oracle identities (test/cone.jl: logdet barrier finite
differences, H*point = -grad, H^-1 H = I, dder3 against the second-order difference) in tests/test_oracle_cones_complex.py; the
reference is Julia and cannot be run here.

The svec of a Hermitian matrix holds, column by column over the upper triangle, the real diagonal entries and for i < j the
pair (re, -im) of sqrt(2)*mat[i, j], i.e. (re, im) of the lower-triangle entry."""
import numpy as np
import scipy.linalg as sla

from . import arrayutil as au
from .cones import Cone, _cols


def svec_length_c(side):   # arrayutilities.jl:81
    return side * side


def svec_side_c(length):   # arrayutilities.jl:103-108
    side = int(round(np.sqrt(length)))
    assert side * side == length
    return side


def smat_to_svec_c(vec, mat, rt2=au.RT2):   # arrayutilities.jl:188-210
    k = 0
    m = mat.shape[0]
    for j in range(m):
        for i in range(j + 1):
            if i == j:
                vec[k] = mat[i, j].real
                k += 1
            else:
                ck = mat[i, j] * rt2
                vec[k] = ck.real
                vec[k + 1] = -ck.imag
                k += 2
    assert k == len(vec)
    return vec


def svec_to_smat_c(mat, vec, rt2=au.RT2):   # arrayutilities.jl:240-262 (upper triangle only)
    k = 0
    m = mat.shape[0]
    for j in range(m):
        for i in range(j + 1):
            if i == j:
                mat[i, j] = vec[k]
                k += 1
            else:
                mat[i, j] = complex(vec[k], -vec[k + 1]) / rt2
                k += 2
    assert k == len(vec)
    return mat


def herm_from_upper(mat):   # copytri!(mat, 'U', true)
    u = np.triu(mat, 1)
    return np.asfortranarray(np.diag(np.diag(mat).real) + u + u.conj().T)


def symm_kron_c(skr, mat, rt2=au.RT2):   # arrayutilities.jl:308-352; upper triangle of skr, mirrored at the end (Symmetric(., :U))
    side = mat.shape[0]
    col = 0
    for l in range(side):
        for k in range(l):
            row = 0
            for j in range(side):
                for i in range(j):
                    a = mat[i, k] * mat[l, j]
                    b = mat[j, k] * mat[l, i]
                    apb, amb = a + b, a - b   # :366-383
                    skr[row, col] = apb.real
                    skr[row + 1, col] = -amb.imag
                    skr[row, col + 1] = apb.imag
                    skr[row + 1, col + 1] = amb.real
                    row += 2
                c = rt2 * mat[j, k] * mat[l, j]
                skr[row, col] = c.real
                skr[row, col + 1] = c.imag
                row += 1
                if row > col:
                    break
            col += 2
        row = 0
        for j in range(side):
            for i in range(j):
                c = rt2 * mat[i, l] * mat[l, j]
                skr[row, col] = c.real
                skr[row + 1, col] = -c.imag
                row += 2
            skr[row, col] = abs(mat[j, l]) ** 2
            row += 1
            if row > col:
                break
        col += 1
    iu = np.triu_indices(skr.shape[0], 1)
    skr[(iu[1], iu[0])] = skr[iu]
    return skr


class PosSemidefTriComplex(Cone):
    """possemideftri.jl:9-207 with R = Complex{Float64}: Hermitian positive definite matrices, barrier -logdet."""

    def __init__(self, dim):
        assert dim >= 1
        self.dim = dim
        self.rt2 = au.RT2
        self.side = svec_side_c(dim)

    def reset_data(self):   # :51-52
        self.feas_updated = self.grad_updated = self.hess_updated = self.inv_hess_updated = False
        self.hess_fact_updated = False

    def use_sqrt_hess_oracles(self, arr_dim):   # :54
        return True

    def setup_extra_data(self):   # :56-65
        s = self.side
        self.mat = np.zeros((s, s), dtype=complex, order="F")
        self.inv_mat = np.zeros((s, s), dtype=complex, order="F")
        self.U = None

    def get_nu(self):   # :67
        return self.side

    def set_initial_point(self, arr):   # :69-78 (increment 2 i + 1 between diagonal entries)
        arr[:] = 0
        k = 0
        for i in range(1, self.side + 1):
            arr[k] = 1
            k += 2 * i + 1
        return arr

    def _chol(self, upper):
        full = herm_from_upper(upper)
        try:
            return sla.cholesky(full, lower=False)
        except sla.LinAlgError:
            return None

    def update_feas(self):   # :80-90
        assert not self.feas_updated
        svec_to_smat_c(self.mat, self.point, self.rt2)
        self.U = self._chol(self.mat)
        self.is_feas_ = self.U is not None
        self.feas_updated = True
        return self.is_feas_

    def is_dual_feas(self):   # :92-95
        m = np.zeros((self.side, self.side), dtype=complex)
        svec_to_smat_c(m, self.dual_point, self.rt2)
        return self._chol(m) is not None

    def update_grad(self):   # :97-107
        assert self.is_feas_
        Ui = sla.solve_triangular(self.U, np.eye(self.side), lower=False)
        self.inv_mat[:] = Ui @ Ui.conj().T
        smat_to_svec_c(self.grad, self.inv_mat, self.rt2)
        self.grad *= -1
        self.mat[:] = herm_from_upper(self.mat)
        self.grad_updated = True
        return self.grad

    def update_hess(self):   # :109-116
        assert self.grad_updated
        self.hess_ = np.zeros((self.dim, self.dim))
        symm_kron_c(self.hess_, herm_from_upper(self.inv_mat), self.rt2)
        self.hess_updated = True
        return self.hess_

    def update_inv_hess(self):   # :118-124
        assert self.is_feas()
        self.inv_hess_ = np.zeros((self.dim, self.dim))
        symm_kron_c(self.inv_hess_, herm_from_upper(self.mat), self.rt2)
        self.inv_hess_updated = True
        return self.inv_hess_

    def _unpack(self, col):
        m = np.zeros((self.side, self.side), dtype=complex, order="F")
        svec_to_smat_c(m, col, self.rt2)
        return herm_from_upper(m)

    def _two_sided(self, prod, arr, left, right):
        P, A = _cols(prod), _cols(arr)
        for i in range(A.shape[1]):
            smat_to_svec_c(P[:, i], left(right(self._unpack(A[:, i]))), self.rt2)
        return prod

    def hess_prod(self, prod, arr):   # :126-142   X^-1 V X^-1 through the Cholesky factor
        assert self.is_feas()
        U = self.U
        rdiv = lambda V: sla.cho_solve((U, False), V.conj().T).conj().T
        ldiv = lambda V: sla.cho_solve((U, False), V)
        return self._two_sided(prod, arr, ldiv, rdiv)

    def inv_hess_prod(self, prod, arr):   # :144-159   X V X
        assert self.is_feas()
        X = herm_from_upper(self.mat)
        return self._two_sided(prod, arr, lambda V: X @ V, lambda V: V @ X)

    def sqrt_hess_prod(self, prod, arr):   # :161-177   U^-H V U^-1
        assert self.is_feas()
        U = self.U
        rdiv = lambda V: sla.solve_triangular(U, V.conj().T, trans="C", lower=False).conj().T      # V U^-1
        ldiv = lambda V: sla.solve_triangular(U, V, trans="C", lower=False)                          # U^-H (.)
        return self._two_sided(prod, arr, ldiv, rdiv)

    def inv_sqrt_hess_prod(self, prod, arr):   # :179-195   U V U^H
        assert self.is_feas()
        U = self.U
        return self._two_sided(prod, arr, lambda V: U @ V, lambda V: V @ U.conj().T)

    def dder3(self, dir):   # :197-207   X^-1 D X^-1 D X^-1
        assert self.grad_updated
        U = self.U
        S = sla.cho_solve((U, False), self._unpack(dir))                          # X^-1 D
        S = sla.solve_triangular(U, S.conj().T, trans="C", lower=False).conj().T   # (.) U^-1
        smat_to_svec_c(self.dder3_, S @ S.conj().T, self.rt2)
        return self.dder3_
