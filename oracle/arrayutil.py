"""svec/smat utilities -- oracle restatement (test infrastructure; see oracle/__init__.py).

Follows /root/reference/src/Cones/arrayutilities.jl:
  svec_length / svec_side        :71-96
  svec_idx                       :115-120
  smat_to_svec! (real)           :163-181
  svec_to_smat! (real)           :218-236
  symm_kron! (real)              :268-306
svec order is the column-major upper triangle, (i <= j) -> j(j+1)/2 + i (0-based), off-diagonals
scaled by sqrt(2).
"""
import numpy as np

RT2 = np.sqrt(2.0)


def svec_length(side):
    return side * (side + 1) // 2


def svec_side(length):
    side = (int(np.sqrt(1 + 8 * length))) // 2
    while side * (side + 1) < 2 * length:
        side += 1
    while side * (side + 1) > 2 * length:
        side -= 1
    assert side * (side + 1) == 2 * length
    return side


def svec_idx(row, col):
    """0-based index of element (row, col) in the svec (arrayutilities.jl:115-120, 1-based there)."""
    if row < col:
        row, col = col, row
    return row * (row + 1) // 2 + col


def smat_to_svec(vec, mat, rt2=RT2):
    """vec <- svec(upper triangle of mat)  (arrayutilities.jl:163-181)."""
    side = mat.shape[0]
    # column-major upper triangle: for j in 0..side-1, for i in 0..j
    jj, ii = np.tril_indices(side)   # (jj >= ii), ordered by jj then ii
    v = mat[ii, jj] * np.where(ii == jj, 1.0, rt2)
    vec[:] = v
    return vec


def svec_to_smat(mat, vec, rt2=RT2):
    """upper triangle of mat <- smat(vec)  (arrayutilities.jl:218-236).  Lower triangle untouched."""
    side = mat.shape[0]
    jj, ii = np.tril_indices(side)
    mat[ii, jj] = np.where(ii == jj, vec, vec / rt2)
    return mat


def copytri_upper(mat):
    """LinearAlgebra.copytri!(mat, 'U', true): mirror upper triangle into the lower one."""
    iu = np.triu_indices(mat.shape[0], 1)
    mat[iu[1], iu[0]] = mat[iu]
    return mat


def symm_kron(skr, mat, rt2=RT2):
    """Upper triangle of the symmetric Kronecker product (arrayutilities.jl:268-306).

    skr[svec(i,j), svec(k,l)] for the operator V -> mat V mat in svec coordinates.
    Plain loops: used only for small sides in tests (explicit hess / inv_hess).
    """
    side = mat.shape[0]
    col_idx = 0
    for l in range(side):
        for k in range(l):
            row_idx = 0
            done = False
            for j in range(side):
                for i in range(j):
                    skr[row_idx, col_idx] = mat[i, k] * mat[j, l] + mat[i, l] * mat[j, k]
                    row_idx += 1
                skr[row_idx, col_idx] = rt2 * mat[j, k] * mat[j, l]
                row_idx += 1
                if row_idx > col_idx:
                    done = True
                    break
            col_idx += 1
        row_idx = 0
        for j in range(side):
            for i in range(j):
                skr[row_idx, col_idx] = rt2 * mat[i, l] * mat[j, l]
                row_idx += 1
            skr[row_idx, col_idx] = mat[j, l] ** 2
            row_idx += 1
            if row_idx > col_idx:
                break
        col_idx += 1
    return skr
