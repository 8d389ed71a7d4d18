"""SeDuMi-format front end: ``x, y, info = conex(A, b, c, K)`` as the reference's MATLAB entry point
interfaces/matlab/conex.m:2-82 has it, on ctypes over ``libconex.so``'s ``CONEX_SolveSedumi`` /
``CONEX_SedumiPreprocess`` (include/conex_sedumi.h).  All arithmetic of the solve runs the HIP path;
the preprocessing (support closure + block splitting, util/BuildMask.m) is host C++ in the library.

``A`` is m x N (dense array or anything with ``.tocoo()``), ``K`` a dict / object with ``s`` (the PSD
orders); the problem is  maximize b'y  s.t.  c - A'y in K  (SeDuMi's dual form)."""
import ctypes as C

import numpy as np

from . import kkt


class SedumiOptions(C.Structure):
    _fields_ = [("blkdiag", C.c_int), ("errors", C.c_int), ("verbose", C.c_int)]


class SedumiInfo(C.Structure):
    _fields_ = [("solved", C.c_int), ("pinf", C.c_int), ("dinf", C.c_int), ("cpusec", C.c_double),
                ("errors", C.c_double * 2), ("num_blocks", C.c_int), ("num_rows_kept", C.c_int)]


_LP, _DP = C.POINTER(C.c_long), C.POINTER(C.c_double)
_PROBLEM = [C.c_long, C.c_long, C.c_long, _LP, _LP, _DP, _DP, _DP, C.c_int, _LP]


def _lib():
    L = kkt.load_library()
    if not getattr(L, "_sedumi_ready", False):
        L.CONEX_SolveSedumi.restype = C.c_int
        L.CONEX_SolveSedumi.argtypes = _PROBLEM + [C.POINTER(SedumiOptions), _DP, _DP, C.POINTER(SedumiInfo)]
        L.CONEX_SedumiPreprocess.restype = C.c_void_p
        L.CONEX_SedumiPreprocess.argtypes = _PROBLEM + [C.c_int]
        L.CONEX_SedumiFree.restype = None
        L.CONEX_SedumiFree.argtypes = [C.c_void_p]
        for name, res, args in (("CONEX_SedumiNumBlocks", C.c_int, [C.c_void_p]),
                                ("CONEX_SedumiKeptRows", C.c_long, [C.c_void_p, _LP]),
                                ("CONEX_SedumiKeptColumns", C.c_long, [C.c_void_p, _LP]),
                                ("CONEX_SedumiReducedB", C.c_long, [C.c_void_p, _DP]),
                                ("CONEX_SedumiBlockOrder", C.c_int, [C.c_void_p, C.c_int]),
                                ("CONEX_SedumiBlockNumVariables", C.c_int, [C.c_void_p, C.c_int]),
                                ("CONEX_SedumiBlockData", C.c_int, [C.c_void_p, C.c_int, _LP, _DP, _DP])):
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        L._sedumi_ready = True
    return L


def _orders(K):
    s = K["s"] if isinstance(K, dict) else K.s
    return np.ascontiguousarray(np.atleast_1d(np.asarray(s)).ravel(), dtype=np.int64)


def _problem(A, b, c, K):
    if hasattr(A, "tocoo"):
        coo = A.tocoo()
        m, N = coo.shape
        row, col, val = coo.row, coo.col, coo.data
    else:
        A = np.asarray(A, dtype=np.float64)
        m, N = A.shape
        row, col = np.nonzero(A)
        val = A[row, col]
    row = np.ascontiguousarray(row, dtype=np.int64)
    col = np.ascontiguousarray(col, dtype=np.int64)
    val = np.ascontiguousarray(val, dtype=np.float64)
    b = np.ascontiguousarray(np.asarray(b, dtype=np.float64).ravel())
    c = np.ascontiguousarray(np.asarray(c, dtype=np.float64).ravel())
    Ks = _orders(K)
    if b.size != m or c.size != N:
        raise ValueError("b / c do not match the shape of A")
    keep = (row, col, val, b, c, Ks)
    args = [m, N, len(val), row.ctypes.data_as(_LP), col.ctypes.data_as(_LP), val.ctypes.data_as(_DP),
            b.ctypes.data_as(_DP), c.ctypes.data_as(_DP), len(Ks), Ks.ctypes.data_as(_LP)]
    return args, keep, m, N


def conex(A, b, c, K, blkdiag=-1, errors=False):
    """-> (x, y, info): x the primal (N,), y the dual (m,), info a :class:`SedumiInfo`."""
    L = _lib()
    args, keep, m, N = _problem(A, b, c, K)
    opt = SedumiOptions(int(blkdiag), int(bool(errors)), 0)
    info = SedumiInfo()
    x, y = np.zeros(N), np.zeros(m)
    rc = L.CONEX_SolveSedumi(*args, C.byref(opt), x.ctypes.data_as(_DP), y.ctypes.data_as(_DP), C.byref(info))
    if rc != 0:
        raise ValueError("CONEX_SolveSedumi rejected the problem (see stderr)")
    return x, y, info


def preprocess(A, b, c, K, blkdiag=1):
    """The preprocessing alone -> dict(kept_rows, kept_cols, b, blocks=[dict(order, variables,
    matrices (n, n, nv), affine (n, n))])."""
    L = _lib()
    args, keep, m, N = _problem(A, b, c, K)
    h = L.CONEX_SedumiPreprocess(*args, int(blkdiag))
    if not h:
        raise ValueError("CONEX_SedumiPreprocess rejected the problem (see stderr)")
    try:
        def vec(fn, dtype, ptr):
            out = np.zeros(fn(h, None), dtype=dtype)
            fn(h, out.ctypes.data_as(ptr))
            return out
        out = {"kept_rows": vec(L.CONEX_SedumiKeptRows, np.int64, _LP),
               "kept_cols": vec(L.CONEX_SedumiKeptColumns, np.int64, _LP),
               "b": vec(L.CONEX_SedumiReducedB, np.float64, _DP), "blocks": []}
        for i in range(L.CONEX_SedumiNumBlocks(h)):
            n, nv = L.CONEX_SedumiBlockOrder(h, i), L.CONEX_SedumiBlockNumVariables(h, i)
            var = np.zeros(nv, dtype=np.int64)
            mats, aff = np.zeros(n * n * nv), np.zeros(n * n)
            L.CONEX_SedumiBlockData(h, i, var.ctypes.data_as(_LP), mats.ctypes.data_as(_DP), aff.ctypes.data_as(_DP))
            out["blocks"].append({"order": n, "variables": var,
                                  "matrices": mats.reshape(nv, n, n).transpose(2, 1, 0).copy(),  # [:, :, k] column-major
                                  "affine": aff.reshape(n, n).T.copy()})
        return out
    finally:
        L.CONEX_SedumiFree(h)
