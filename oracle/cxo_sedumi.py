"""ORACLE (test infrastructure only -- never imported by conex_amd/ or the timed region of bench.py):
numpy restatement of the reference's MATLAB preprocessing for SeDuMi-format problems, function by
function, 1-based MATLAB indices turned 0-based:

    CleanLinear                   interfaces/matlab/util/CleanLinear.m:1-30
    coneBase.Symmetrize           interfaces/matlab/util/coneBase.m:127-190
    BuildMask + SubspaceClosureCoordDisjointSupport   util/BuildMask.m:1-85
    BinaryPsdCompletion + conncomp                    util/BinaryPsdCompletion.m:1-62
    coneBase.SubMatToIndx         util/coneBase.m:254-266
    blkdiagPrg                    util/blkdiagPrg.m:17-38
    ExtractConstraintMatrices     util/ExtractConstraintMatrices.m:1-48

Pinned by the reference's own literal test interfaces/matlab/test/test_extract_constraints.m:1-38
(tests/test_sedumi_frontend.py) -- the rest of the MATLAB tests draw their data from rand().
Dense numpy on purpose (small test problems): clarity over speed."""
import numpy as np


def clean_linear(A, b):
    keep = np.flatnonzero(np.any(np.c_[A, b] != 0, axis=1))     # CleanLinear.m:19
    return A[keep], b[keep], keep


def lower_upper(Ks):
    """indxL / indxU of coneBase.CalcIndicesLU (:127-165): entry pairs (i, j), (j, i), i >= j."""
    iL, iU, off = [], [], 0
    for n in Ks:
        temp = off + np.arange(n * n).reshape(n, n).T            # temp(i, j) = offset + i + j n
        for j in range(n):
            for i in range(j, n):
                iL.append(temp[i, j])
                iU.append(temp[j, i])
        off += n * n
    return np.array(iL), np.array(iU)


def symmetrize(A, Ks):
    A = np.array(A, dtype=float, copy=True)
    iL, iU = lower_upper(Ks)
    As = (A[:, iL] + A[:, iU]) / 2                               # coneBase.m:185
    A[:, iL] = As
    A[:, iU] = As
    return A


def subspace_closure(M, A, b):                                   # BuildMask.m:64-85
    M = np.any(A[b != 0, :] != 0, axis=0) | M
    nnz = M.sum()
    while True:
        tau = np.any(A[:, M] != 0, axis=1)
        M = np.any(A[tau, :] != 0, axis=0)
        if nnz != M.sum():
            nnz = M.sum()
        else:
            return M


def conncomp(Adj):                                               # BinaryPsdCompletion.m:20-62
    N = Adj.shape[0]
    Adj = Adj.copy()
    Adj[np.arange(N), np.arange(N)] = 0
    Adj = Adj + Adj.T
    seen = np.zeros(N, dtype=bool)
    members = []
    for n in range(N):
        if not seen[n]:
            members.append([n])
            seen[n] = True
            ptr = 0
            while ptr < len(members[-1]):
                nbrs = np.flatnonzero(Adj[:, members[-1][ptr]])
                new = nbrs[~seen[nbrs]]
                seen[new] = True
                members[-1].extend(new.tolist())
                ptr += 1
    sizes = np.array([len(mm) for mm in members])
    order = np.argsort(sizes, kind="stable")                      # MATLAB sort: stable
    return [members[k] for k in order]


def binary_psd_completion(M):                                    # BinaryPsdCompletion.m:1-17
    r = np.unique(np.nonzero(M)[0])
    if r.size == 0:
        return M, []
    cliques = []
    for comp in conncomp(M[np.ix_(r, r)].astype(int)):
        cl = r[comp]
        M[np.ix_(cl, cl)] = True
        cliques.append(cl)
    return M, cliques


def build_mask(A, b, c, Ks):                                     # BuildMask.m:1-61
    M = np.asarray(c != 0).ravel()
    nnz = M.sum()
    while True:
        M = subspace_closure(M, A, b)
        cliques, off = [], 0
        for n in Ks:
            blk, cl = binary_psd_completion(M[off:off + n * n].reshape(n, n).T.copy())
            M[off:off + n * n] = blk.T.ravel()
            cliques.append(cl)
            off += n * n
        if nnz == M.sum():
            break
        nnz = M.sum()
    indx, Kr, off = [], [], 0
    for n, cl_i in zip(Ks, cliques):
        for cl in cl_i:
            t = np.zeros((n, n), dtype=bool)
            t[np.ix_(cl, cl)] = True
            indx.extend((np.flatnonzero(t.T.ravel()) + off).tolist())   # find(t(:)): column-major
            Kr.append(len(cl))
        off += n * n
    indx = np.array(indx, dtype=int)
    return A[:, indx], c[indx], Kr, indx


def extract_constraint_matrices(A, affine, Ks):                 # ExtractConstraintMatrices.m:1-48
    out, off = [], 0
    for n in Ks:
        sub = A[:, off:off + n * n]
        var = np.flatnonzero(np.any(sub != 0, axis=1))            # unique(matrix_row)
        mats = np.stack([sub[v].reshape(n, n).T for v in var], axis=2) if var.size else np.zeros((n, n, 0))
        out.append({"order": n, "variables": var, "matrices": mats,
                    "affine": np.asarray(affine[off:off + n * n]).reshape(n, n).T})
        off += n * n
    return out


def preprocess(A, b, c, Ks, blkdiag=True):
    """conex.m:3-31 up to the point where the program is built."""
    A = np.asarray(A, dtype=float)
    b = np.asarray(b, dtype=float).ravel()
    c = np.asarray(c, dtype=float).ravel()
    A, b, keep1 = clean_linear(A, b)                             # conex.m:3
    A = symmetrize(A, Ks)                                        # conex.m:6
    c = symmetrize(c[None, :], Ks)[0]
    if not blkdiag:
        return {"kept_rows": keep1, "kept_cols": np.arange(A.shape[1]), "b": b,
                "blocks": extract_constraint_matrices(A, c, Ks)}
    Ar, cr, Kr, indx = build_mask(A, b, c, Ks)                   # blkdiagPrg.m:26-27
    Ar, br, keep2 = clean_linear(Ar, b)                          # blkdiagPrg.m:29
    return {"kept_rows": keep1[keep2], "kept_cols": indx, "b": br,
            "blocks": extract_constraint_matrices(Ar, cr, Kr)}
