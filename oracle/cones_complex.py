"""TEST INFRASTRUCTURE (CPU oracle), not the product: complex variants of PosSemidefTri, EpiNormSpectral, LinMatrixIneq, HypoRootdetTri, HypoPerLogdetTri and
WSOSInterpNonnegative (src/Cones/wsosinterpnonnegative.jl:15-200 with complex bases), the "complex Hermitian variants" of SURVEY 8(f) rank 3.  Restates
reference src/Cones/possemideftri.jl:9-207, src/Cones/epinormspectral.jl:13-294 (R = Complex{Float64}), src/Cones/linmatrixineq.jl:9-159 (Hermitian members) and the complex
vectorisation helpers of
src/Cones/arrayutilities.jl:13,81,103-108 (lengths), :188-210 (smat_to_svec!), :240-262 (svec_to_smat!), :308-352 (symm_kron!),
:366-383 (spectral_kron_element!).  Parity pinned by the reference's own oracle identities (test/cone.jl: logdet barrier finite
differences, H*point = -grad, H^-1 H = I, dder3 against the second-order difference) in tests/test_oracle_cones_complex.py; the
reference is Julia and cannot be run here.

The svec of a Hermitian matrix holds, column by column over the upper triangle, the real diagonal entries and for i < j the
pair (re, -im) of sqrt(2)*mat[i, j], i.e. (re, im) of the lower-triangle entry."""
import numpy as np
import scipy.linalg as sla
from scipy.linalg import lapack

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


# ----------------------------------------------------------------------------------------------
def cvec_to_rvec(rvec, cmat):   # arrayutilities.jl:30-45 (vec_copyto!, complex -> real): (re, im) pairs in column-major order
    flat = np.asarray(cmat).reshape(-1, order="F")
    rvec[0::2] = flat.real
    rvec[1::2] = flat.imag
    return rvec


def rvec_to_cmat(rvec, d1, d2):   # arrayutilities.jl:47-60 (vec_copyto!, real -> complex)
    return np.asfortranarray((rvec[0::2] + 1j * rvec[1::2]).reshape(d1, d2, order="F"))


class EpiNormSpectralComplex(Cone):
    """epinormspectral.jl:13-294 with R = Complex{Float64}: (u, W), u >= sigma_1(W), W complex d1 x d2 (d1 <= d2) held as
    interleaved (re, im) pairs; barrier -logdet(u^2 I - W W^H) + (d1 - 1) log u."""

    def __init__(self, d1, d2, use_dual=False):
        assert 1 <= d1 <= d2
        self.use_dual_barrier_ = use_dual
        self.d1, self.d2 = d1, d2
        self.dim = 1 + 2 * d1 * d2

    def reset_data(self):   # :70-72
        self.feas_updated = self.grad_updated = self.hess_updated = self.inv_hess_updated = False
        self.hess_aux_updated = self.hess_fact_updated = False

    def setup_extra_data(self):   # :75-95
        self.W = self.tau = self.Zi = self.U = None

    def get_nu(self):   # :97
        return self.d1 + 1

    def set_initial_point(self, arr):   # :99-105
        arr[:] = 0
        arr[0] = np.sqrt(self.get_nu())
        return arr

    def _solve(self, M):   # ldiv!(fact_Z, M)
        return sla.cho_solve((self.U, False), M)

    def update_feas(self):   # :107-123
        assert not self.feas_updated
        u = self.point[0]
        self.is_feas_ = False
        if u > np.finfo(float).eps:
            self.W = rvec_to_cmat(self.point[1:], self.d1, self.d2)
            Z = u * u * np.eye(self.d1) - self.W @ self.W.conj().T
            try:
                self.U = sla.cholesky((Z + Z.conj().T) / 2, lower=False)
                self.is_feas_ = True
            except sla.LinAlgError:
                self.is_feas_ = False
        self.feas_updated = True
        return self.is_feas_

    def is_dual_feas(self):   # :125-132
        u = self.dual_point[0]
        if u > np.finfo(float).eps:
            W = rvec_to_cmat(self.dual_point[1:], self.d1, self.d2)
            return bool(u - np.sum(sla.svdvals(W)) > np.finfo(float).eps)
        return False

    def update_grad(self):   # :134-150
        assert self.is_feas_
        u = self.point[0]
        self.tau = self._solve(self.W)
        self.Zi = self._solve(np.eye(self.d1, dtype=complex))
        self.grad[0] = -u * np.trace(self.Zi).real
        cvec_to_rvec(self.grad[1:], self.tau)
        self.grad *= 2
        self.grad[0] += (self.d1 - 1) / u
        self.grad_updated = True
        return self.grad

    def update_hess_aux(self):   # :152-170
        assert self.grad_updated
        u = self.point[0]
        self.Zitau = self._solve(self.tau)
        self.HuW = -4 * u * self.Zitau
        self.trZi2 = float(np.sum(np.abs(self.Zi) ** 2))
        self.Huu = 4 * u * u * self.trZi2 + (self.grad[0] - 2 * (self.d1 - 1) / u) / u
        self.WtauI = np.eye(self.d2) + self.W.conj().T @ self.tau
        self.hess_aux_updated = True

    def update_hess(self):   # :172-209 (upper triangle, 2 x 2 real blocks of spectral_kron_element!, arrayutilities.jl:366-383)
        if not self.hess_aux_updated:
            self.update_hess_aux()
        d1, d2 = self.d1, self.d2
        Zi, tau, WtauI = self.Zi, self.tau, self.WtauI
        H = np.zeros((self.dim, self.dim))
        r = 1
        for i in range(d2):
            for j in range(d1):
                c = r
                for k in range(i, d2):
                    for l in range(j if i == k else 0, d1):
                        a = Zi[l, j] * WtauI[i, k]
                        b = tau[l, i] * tau[j, k]
                        apb, amb = a + b, a - b
                        H[r, c] = apb.real
                        H[r + 1, c] = -amb.imag
                        H[r, c + 1] = apb.imag
                        H[r + 1, c + 1] = amb.real
                        c += 2
                r += 2
        H *= 2
        cvec_to_rvec(H[0, 1:], self.HuW)
        H[0, 0] = self.Huu
        self.hess_ = np.triu(H)
        self.hess_updated = True
        return self.hess_

    def hess_prod(self, prod, arr):   # :211-239
        if not self.hess_aux_updated:
            self.update_hess_aux()
        u = self.point[0]
        W = self.W
        P, A = _cols(prod), _cols(arr)
        for j in range(A.shape[1]):
            a1 = A[0, j]
            AW = rvec_to_cmat(A[1:, j], self.d1, self.d2)
            P[0, j] = self.Huu * a1 + np.vdot(self.HuW, AW).real
            T = AW @ W.conj().T
            T = T + T.conj().T
            T[np.diag_indices(self.d1)] -= 2 * u * a1
            R = self._solve(2 * (T @ self.tau) + 2 * AW)
            cvec_to_rvec(P[1:, j], R)
        return prod

    def dder3(self, dir):   # :241-294
        assert self.hess_aux_updated
        u = self.point[0]
        W = self.W
        u_dir = dir[0]
        W_dir = rvec_to_cmat(dir[1:], self.d1, self.d2)
        Zi, tau, Zitau, WtauI = self.Zi, self.tau, self.Zitau, self.WtauI
        H = lambda M: M.conj().T

        d2d2b = H(W_dir) @ tau
        d1d2d = self._solve(W_dir)
        d1d2b = d1d2d @ WtauI
        d1d2c = d1d2d @ H(d2d2b)
        d1d1 = d1d2d @ H(W)
        d2d2 = d2d2b @ d2d2b

        d2d2 = d2d2 + H(W_dir) @ d1d2b
        d1d2d = tau @ d2d2
        d1d2d = d1d2d + d1d2c @ WtauI
        d1d2d = d1d2d + d1d2b @ d2d2b

        d1d2b = self._solve(d1d2b)
        d1d2b = d1d2b + Zitau @ d2d2b

        d1d1 = d1d1 + tau @ H(W_dir)
        d1d2b = d1d2b + d1d1 @ Zitau
        d1d2b = d1d2b * (-2 * u)

        const1 = 4 * u * u_dir * u
        d1d2c = self._solve(const1 * Zitau - u_dir * tau)
        d1d2b = d1d2b + d1d2c

        d1d2d = -2 * u_dir * d1d2b - 2 * d1d2d
        cvec_to_rvec(self.dder3_[1:], d1d2d)

        LiZi = sla.solve_triangular(self.U, Zi, trans="C", lower=False)   # fact_Z.L \ Zi with Z = U^H U
        trZi3 = float(np.sum(np.abs(LiZi) ** 2))
        d1d2b = d1d2b + 3 * d1d2c
        self.dder3_[0] = (-np.vdot(W_dir, d1d2b).real - u * u_dir * (6 * self.trZi2 - 8 * u * trZi3 * u) * u_dir
                          - (self.d1 - 1) * (u_dir / u) ** 2 / u)
        return self.dder3_


# ----------------------------------------------------------------------------------------------
class LinMatrixIneqComplex(Cone):
    """linmatrixineq.jl:9-159 with complex Hermitian (or mixed real / complex) members `As`: the cone vector stays real
    (one weight per matrix); barrier -logdet(sum_i w_i A_i)."""

    def __init__(self, As, use_dual=False):
        As = [np.array(A, dtype=complex) for A in As]
        self.dim = len(As)
        assert self.dim > 1                                    # :42
        self.side = As[0].shape[0]
        for A in As:
            assert A.shape == (self.side, self.side) and np.allclose(A, A.conj().T, rtol=0, atol=0)   # :44-53 ishermitian
        assert self.side * self.side >= self.dim
        assert np.all(np.linalg.eigvalsh(As[0]) > 0)           # :57
        self.use_dual_barrier_ = bool(use_dual)
        self.As = As
        self.nu = self.side                                    # :72

    def reset_data(self):   # :68-70
        self.feas_updated = self.grad_updated = self.hess_updated = self.inv_hess_updated = False
        self.hess_fact_updated = False
        self.use_hess_prod_slow = self.use_hess_prod_slow_updated = False

    def get_nu(self):
        return self.nu

    def set_initial_point(self, arr):   # :74-81
        arr[:] = 0.0
        arr[0] = 1.0
        return arr

    def update_feas(self):   # :87-96
        assert not self.feas_updated
        sumA = sum(w * A for w, A in zip(self.point, self.As))
        try:
            self.L = sla.cholesky(sumA, lower=True)
            self.is_feas_ = True
        except sla.LinAlgError:
            self.is_feas_ = False
        self.feas_updated = True
        return self.is_feas_

    def update_grad(self):   # :98-109   Hermitian(L \ (L \ A_i)', :U)
        assert self.is_feas_
        self.sumAinvAs = []
        for i, A in enumerate(self.As):
            T = sla.solve_triangular(self.L, A, lower=True)
            M = sla.solve_triangular(self.L, T.conj().T, lower=True)
            self.sumAinvAs.append(herm_from_upper(M))
            self.grad[i] = -np.trace(self.sumAinvAs[-1]).real
        self.grad_updated = True
        return self.grad

    def update_hess(self):   # :111-123 (upper triangle): real(dot(M_i, M_j')) = Re tr(M_i M_j)
        assert self.grad_updated
        H = np.zeros((self.dim, self.dim))
        for i in range(self.dim):
            for j in range(i, self.dim):
                H[i, j] = np.vdot(self.sumAinvAs[i], self.sumAinvAs[j].conj().T).real
        self.hess_ = H
        self.hess_updated = True
        return self.hess_

    def hess_prod_slow(self, prod, arr):   # :125-144
        if not self.use_hess_prod_slow_updated:
            self.update_use_hess_prod_slow()
        assert self.hess_updated
        if not self.use_hess_prod_slow:
            return self.hess_prod(prod, arr)
        assert self.grad_updated
        P, A = _cols(prod), _cols(arr)
        for j in range(A.shape[1]):
            j_mat = sum(A[i, j] * self.sumAinvAs[i] for i in range(self.dim))
            for i in range(self.dim):
                P[i, j] = np.vdot(j_mat, self.sumAinvAs[i]).real
        return prod

    def dder3(self, dir):   # :146-159
        assert self.grad_updated
        dir_mat = sum(d * M for d, M in zip(dir, self.sumAinvAs))
        Z = dir_mat @ dir_mat.conj().T
        for i in range(self.dim):
            self.dder3_[i] = np.vdot(Z, self.sumAinvAs[i]).real
        return self.dder3_


# ----------------------------------------------------------------------------------------------
class WSOSInterpNonnegativeComplex(Cone):
    """wsosinterpnonnegative.jl:15-200 with R = Complex{T}: real-valued Hermitian polynomials, complex bases `Ps`
    (PolyUtils/complex.jl:13-72), REAL cone vector of U interpolant values.  The barrier is for the DUAL cone:
    use_dual_barrier = !use_dual (:58).  Lambda_k = P_k' Diag(pt) P_k is Hermitian."""

    def __init__(self, U, Ps, use_dual=False):
        for Pk in Ps:
            assert Pk.shape[0] == U                            # :54-56
        self.use_dual_barrier_ = not use_dual                  # :58
        self.dim = U
        self.Ps = [np.asfortranarray(Pk, dtype=complex) for Pk in Ps]
        self.nu = sum(Pk.shape[1] for Pk in Ps)                # :61

    def reset_data(self):   # :66-68
        self.feas_updated = self.grad_updated = self.hess_updated = self.inv_hess_updated = False
        self.hess_fact_updated = False
        self.use_hess_prod_slow = self.use_hess_prod_slow_updated = False

    def setup_extra_data(self):   # :70-85
        K = len(self.Ps)
        self.LamF = [None] * K
        self.LamFLP = [None] * K

    def get_nu(self):
        return self.nu

    def set_initial_point(self, arr):   # :87
        arr[:] = 1.0
        return arr

    def update_feas(self):   # :89-117 (the Ps_order timing sort only changes evaluation order)
        assert not self.feas_updated
        self.is_feas_ = True
        for k, Pk in enumerate(self.Ps):
            LUk = Pk.conj().T * self.point[None, :]            # Pk' * Diagonal(point)
            LLk = LUk @ Pk
            c, info = lapack.zpotrf(LLk, lower=1, clean=1)     # cholesky!(Hermitian(LLk, :L), check = false)
            self.LamF[k] = c
            if info != 0:
                self.is_feas_ = False
                break
        self.feas_updated = True
        return self.is_feas_

    def update_grad(self):   # :119-133
        assert self.is_feas_
        self.grad[:] = 0
        for k, Pk in enumerate(self.Ps):
            LFLP = sla.solve_triangular(self.LamF[k], Pk.conj().T, lower=True)   # ldiv!(LamFLP_k, LamF_k.L, P_k')
            self.LamFLP[k] = LFLP
            self.grad -= np.sum(np.abs(LFLP) ** 2, axis=0)                        # sum(abs2, LamFLP_k[:, j])
        self.grad_updated = True
        return self.grad

    def update_hess(self):   # :135-150 (upper triangle)
        assert self.grad_updated
        H = np.zeros((self.dim, self.dim))
        for k in range(len(self.Ps)):
            UU = self.LamFLP[k].conj().T @ self.LamFLP[k]      # outer_prod!(LamFLP_k, UU, true, false)
            H += np.triu(np.abs(UU) ** 2)                      # abs2(UU[i, j])
        self.hess_ = H
        self.hess_updated = True
        return self.hess_

    def _partial_lambda(self, dir, LFLP):   # :190-200
        LU = LFLP * dir[None, :]
        LL = LU @ LFLP.conj().T
        LLh = np.triu(LL) + np.triu(LL, 1).conj().T            # Hermitian(LLk) reads the upper triangle
        np.fill_diagonal(LLh, LLh.diagonal().real)
        return LLh @ LFLP

    def hess_prod_slow(self, prod, arr):   # :152-175
        if not self.use_hess_prod_slow_updated:
            self.update_use_hess_prod_slow()
        assert self.hess_updated
        if not self.use_hess_prod_slow:
            return self.hess_prod(prod, arr)
        assert self.grad_updated
        P, A = _cols(prod), _cols(arr)
        P[:] = 0
        for k in range(len(self.Ps)):
            LFLP = self.LamFLP[k]
            for j in range(A.shape[1]):
                LU = self._partial_lambda(A[:, j], LFLP)
                P[:, j] += np.sum(LFLP.conj() * LU, axis=0).real   # real(dot(LamFLP_k[:, i], LU_k[:, i]))
        return prod

    def dder3(self, dir):   # :177-188
        assert self.grad_updated
        self.dder3_[:] = 0
        for k in range(len(self.Ps)):
            LU = self._partial_lambda(dir, self.LamFLP[k])
            self.dder3_ += np.sum(np.abs(LU) ** 2, axis=0)
        return self.dder3_


# ----------------------------------------------------------------------------------------------
class _CholC:
    """upper Cholesky factor of a Hermitian matrix (fact.U)"""
    def __init__(self, U):
        self.factors = U


def _hermitian_hooks(cls_name, base, doc):
    """complex Hermitian variant of a PSD-family oracle cone whose matrix handling goes through the representation hooks
    (oracle/cones.py HypoRootdetTri: _side_of, _diag_step, _smat_full, _full, _to_svec, _kron, _chol, _inv_from_chol, _tr, _fro2,
    _two_sided_chol, _two_sided_chol_back)"""

    class _C(base):
        def _side_of(self, length):
            return svec_side_c(length)

        def _diag_step(self, i):
            return 2 * i + 1

        def setup_extra_data(self):
            base.setup_extra_data(self)
            d = self.d
            self.mat = np.zeros((d, d), dtype=complex, order="F")
            self.Wi = np.zeros((d, d), dtype=complex, order="F")

        def _smat_full(self, v):
            m = np.zeros((self.d, self.d), dtype=complex, order="F")
            svec_to_smat_c(m, v, self.rt2)
            return herm_from_upper(m)

        @staticmethod
        def _full(upper):
            return herm_from_upper(np.asarray(upper))

        def _to_svec(self, out, mat):
            return smat_to_svec_c(out, mat, self.rt2)

        def _kron(self, out, mat):
            return symm_kron_c(out, mat, self.rt2)

        def _chol(self, v):
            try:
                U = sla.cholesky(self._smat_full(v), lower=False)
            except sla.LinAlgError:
                return None, 0.0
            return _CholC(U), 2 * np.sum(np.log(np.diag(U).real))

        def _inv_from_chol(self, f):
            Ui = sla.solve_triangular(f.factors, np.eye(self.d), lower=False)
            return np.asfortranarray(Ui @ Ui.conj().T)

        @staticmethod
        def _tr(m):
            return np.trace(m).real

        @staticmethod
        def _fro2(m):
            return float(np.sum(np.abs(m) ** 2))

        def _two_sided_chol(self, R):   # U^-H R U^-1
            U = self.fact_W.factors
            T = sla.solve_triangular(U, np.asarray(R).conj().T, trans="C", lower=False).conj().T
            return sla.solve_triangular(U, T, trans="C", lower=False)

        def _two_sided_chol_back(self, S):   # U^-1 S U^-H
            U = self.fact_W.factors
            T = sla.solve_triangular(U, np.asarray(S).conj().T, lower=False).conj().T
            return sla.solve_triangular(U, T, lower=False)

    _C.__name__ = _C.__qualname__ = cls_name
    _C.__doc__ = doc
    return _C


from .cones import HypoRootdetTri as _HypoRootdetTri   # noqa: E402

HypoRootdetTriComplex = _hermitian_hooks(
    "HypoRootdetTriComplex", _HypoRootdetTri,
    "hyporootdettri.jl:9-324 with R = Complex{Float64}: (u, w), u <= det(smat(w))^(1/d), w the complex svec of a Hermitian matrix.")

from .cones import HypoPerLogdetTri as _HypoPerLogdetTri   # noqa: E402

HypoPerLogdetTriComplex = _hermitian_hooks(
    "HypoPerLogdetTriComplex", _HypoPerLogdetTri,
    "hypoperlogdettri.jl:9-368 with R = Complex{Float64}: (u, v, w), u <= v logdet(smat(w) / v), w the complex svec of a Hermitian matrix.")
