"""Python-3 front end with the surface of the reference's ``ConexProgram.Conex`` class
(interfaces/python/ConexProgram.py:58-277, Python 2 over a SWIG module), on ctypes over
``libconex.so``'s ``CONEX_*`` ABI (include/conex.h) -- so every solve runs the HIP path.

Conventions kept from the reference wrapper:

* matrix inequalities are passed as an ``n x n x m`` array, ``A[:, :, i]`` multiplying variable
  ``i`` (the SWIG typemap takes Fortran-ordered 3-D input, conex.i:17-29), and the constraint is
  ``c - sum_i y_i A[:, :, i]  >= 0``;
* ``Maximize(b)`` / ``Solve()`` return a :class:`Solution` with ``y`` and ``status`` (1 = solved,
  the polarity of cone_program.cc:532);
* builder failures raise (the reference raises ``NameError``; here :class:`ConexError`, a subclass);
* ``DefaultConfiguration`` applies the wrapper's own overrides (ConexProgram.py:115-125) on top of
  ``CONEX_SetDefaultOptions``.

Arrays are plain ``numpy.ndarray`` (the reference uses ``numpy.matrix``).
"""
import ctypes as C

import numpy as np

from . import capi


class ConexError(NameError):
    """Raised where the reference wrapper raises NameError."""


class Errors:
    """Optimality measures of :meth:`Conex.ComputeErrors` (ConexProgram.py:11-15, 244-277)."""

    def __init__(self):
        self.Ax_minus_b = 0.0
        self.x_dot_s = 0.0
        self.min_eig_S = []
        self.min_eig_X = []


class Solution:
    def __init__(self):
        self.err = Errors()
        self.x = []
        self.y = []
        self.s = []
        self.status = 0


class LMIOperator:
    """y -> sum_i y[var_i] A[:, :, i]  and its adjoint  X -> (tr(A[:, :, i] X))_i."""

    def __init__(self, matrices, num_vars=None, variables=None):
        self.matrices = np.asarray(matrices, dtype=np.float64)
        m_local = self.matrices.shape[2]
        self.variables = list(range(m_local)) if variables is None else [int(v) for v in variables]
        if len(self.variables) != m_local:
            raise ConexError("Invalid LMI")
        self.m = m_local if num_vars is None else int(num_vars)

    def apply(self, y):
        y = np.asarray(y, dtype=np.float64).ravel()
        out = np.zeros(self.matrices.shape[:2])
        for i, var in enumerate(self.variables):
            out += self.matrices[:, :, i] * y[var]
        return out

    def adjoint(self, X):
        out = np.zeros(self.m)
        for i, var in enumerate(self.variables):
            out[var] += np.trace(self.matrices[:, :, i] @ np.asarray(X))
        return out


class _LinearOperator:
    def __init__(self, A):
        self.A = np.asarray(A, dtype=np.float64)
        self.m = self.A.shape[1]

    def apply(self, y):
        return self.A @ np.asarray(y, dtype=np.float64).ravel()

    def adjoint(self, x):
        return self.A.T @ np.asarray(x, dtype=np.float64).ravel()


def _f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


class Conex:
    def __init__(self, m=-1):
        self._L = capi.api()
        self.a = self._L.CONEX_CreateConeProgram()
        if m >= 0 and self._L.CONEX_SetNumberOfVariables(self.a, int(m)) != 0:
            raise ConexError("Failed to set the number of variables.")
        self.num_constraints = 0
        self.A = []   # operators of the constraints added through the Add* calls
        self.c = []   # their affine terms (shape = shape of the dual variable)
        self.m = m

    def __del__(self):
        if getattr(self, "a", None):
            self._L.CONEX_DeleteConeProgram(self.a)
            self.a = None

    # ---- options and statistics
    def DefaultConfiguration(self):
        cfg = capi.default_config()
        cfg.inv_sqrt_mu_max = 1000
        cfg.maximum_mu = 1e20
        cfg.max_iterations = 100
        cfg.final_centering_steps = 1
        cfg.prepare_dual_variables = 1
        cfg.infeasibility_threshold = 1e8
        cfg.divergence_upper_bound = 1
        return cfg

    def GetIterationNumberStats(self, num):
        st = capi.IterationStats()
        self._L.CONEX_GetIterationStats(self.a, C.byref(st), int(num))
        return st

    def GetIterationStats(self):
        last = self.GetIterationNumberStats(-1).iteration_number
        return [self.GetIterationNumberStats(i) for i in range(last + 1)]

    # ---- whole-constraint builders
    def AddLinearInequality(self, A, c):
        A = np.asarray(A, dtype=np.float64)
        c = np.asarray(c, dtype=np.float64).ravel()
        Af = capi.colmajor(A)
        if self._L.CONEX_AddDenseLinearConstraint(self.a, capi.dp(Af), A.shape[0], A.shape[1],
                                                  capi.dp(_f64(c)), len(c)) < 0:
            raise ConexError("Failed to add constraint.")
        self.m = A.shape[1]
        self.A.append(_LinearOperator(A))
        self.c.append(c.copy())
        self.num_constraints += 1

    def AddLinearInequalities(self, A, lb, ub):
        A = np.asarray(A, dtype=np.float64)
        lb = _f64(np.asarray(lb, dtype=np.float64).ravel())
        ub = _f64(np.asarray(ub, dtype=np.float64).ravel())
        Af = capi.colmajor(A)
        self._L.CONEX_AddLinearInequalities(self.a, capi.dp(Af), A.shape[0], A.shape[1], capi.dp(lb), len(lb),
                                            capi.dp(ub), len(ub))
        self.A.append(_LinearOperator(A))
        self.c.append(ub.copy())
        self.num_constraints += 1

    def AddQuadraticCost(self, P):
        P = np.asarray(P, dtype=np.float64)
        if P.shape[0] != self.m or P.shape[1] != self.m:
            raise ConexError("Cost matrix dimension does not match number of variables.")
        Pf = capi.colmajor(P)
        self._L.CONEX_AddQuadraticCost(self.a, capi.dp(Pf), P.shape[0], P.shape[1])

    def _lmi_buffers(self, A, c):
        A = np.asarray(A, dtype=np.float64)
        c = np.asarray(c, dtype=np.float64)
        if A.ndim != 3 or A.shape[0] != A.shape[1] or c.shape != A.shape[:2]:
            raise ConexError("Invalid LMI")
        Af = np.ascontiguousarray(np.transpose(A, (2, 1, 0))).ravel()  # [i][col][row]: column-major n x n x m
        return A, c, Af, capi.colmajor(c)

    def AddDenseLinearMatrixInequality(self, A, c):
        A, c, Af, cf = self._lmi_buffers(A, c)
        n, m = A.shape[0], A.shape[2]
        if self._L.CONEX_AddDenseLMIConstraint(self.a, capi.dp(Af), n, n, m, capi.dp(cf), n, n) < 0:
            raise ConexError("Failed to add constraint.")
        self.n, self.m = n, m
        self.A.append(LMIOperator(A))
        self.c.append(c.copy())
        self.num_constraints += 1

    def AddSparseLinearMatrixInequality(self, A, c, variables):
        variables = np.asarray(variables).astype(np.int64).ravel()
        if len(variables) and int(variables.max()) + 1 > self.m:
            raise ConexError("Invalid sparse LMI. %d != %d" % (self.m, int(variables.max()) + 1))
        A, c, Af, cf = self._lmi_buffers(A, c)
        n, m = A.shape[0], A.shape[2]
        v = np.ascontiguousarray(variables, dtype=np.int64)
        if self._L.CONEX_AddSparseLMIConstraint(self.a, capi.dp(Af), n, n, m, capi.dp(cf), n, n,
                                                v.ctypes.data_as(C.POINTER(C.c_long)), len(v)) < 0:
            raise ConexError("Failed to add constraint.")
        self.A.append(LMIOperator(A, self.m, variables))
        self.c.append(c.copy())
        self.num_constraints += 1

    # ---- entry-by-entry builders
    def _new(self, fn, *args, what="constraint"):
        cid = C.c_int()
        if fn(self.a, *args, C.byref(cid)) != 0:
            raise ConexError("Failed to add %s." % what)
        self.num_constraints += 1
        return cid.value

    def NewLinearMatrixInequality(self, order, hyper_complex_dim):
        cid = self._new(self._L.CONEX_NewLinearMatrixInequality, int(order), int(hyper_complex_dim))
        self.c.append(np.zeros((order, order)))
        return cid

    def NewLorentzConeConstraint(self, order):
        return self._new(self._L.CONEX_NewLorentzConeConstraint, int(order))

    def NewLinearInequality(self, num_rows):
        return self._new(self._L.CONEX_NewLinearInequality, int(num_rows))

    def NewQuadraticCost(self):
        return self._new(self._L.CONEX_NewQuadraticCost, what="quadratic cost")

    def UpdateQuadraticCostMatrix(self, cost_id, value, row, col):
        if self._L.CONEX_UpdateQuadraticCostMatrix(self.a, int(cost_id), float(value), int(row), int(col)) != 0:
            raise ConexError("Failed to update quadratic cost.")

    def UpdateLinearOperator(self, constraint, value, variable, row, col=0, hyper_complex_dim=0):
        if self._L.CONEX_UpdateLinearOperator(self.a, int(constraint), float(value), int(variable), int(row),
                                              int(col), int(hyper_complex_dim)) != 0:
            raise ConexError("Failed to update operator.")

    def UpdateAffineTerm(self, constraint, value, row, col=0, hyper_complex_dim=0):
        if self._L.CONEX_UpdateAffineTerm(self.a, int(constraint), float(value), int(row), int(col),
                                          int(hyper_complex_dim)) != 0:
            raise ConexError("Failed to update affine term.")

    # ---- solve
    def Maximize(self, b, config=None):
        cfg = config if config is not None else self.DefaultConfiguration()
        b = _f64(np.asarray(b, dtype=np.float64).ravel())
        if len(b) != self.m:
            raise ConexError("Cost vector dimension does not match number of variables.")
        sol = Solution()
        sol.y = np.ones(self.m)
        sol.status = self._L.CONEX_Maximize(self.a, capi.dp(b), len(b), C.byref(cfg), capi.dp(sol.y), len(sol.y))
        return sol

    def Solve(self, config=None):
        cfg = config if config is not None else self.DefaultConfiguration()
        cfg.enable_line_search = 1
        cfg.enable_rescaling = 0
        sol = Solution()
        sol.y = np.ones(self.m)
        sol.status = self._L.CONEX_Solve(self.a, C.byref(cfg), capi.dp(sol.y), len(sol.y))
        return sol

    def GetDualVariables(self):
        xs = []
        for i in range(self.num_constraints):
            rows = self._L.CONEX_GetDualVariableSize(self.a, i)
            shape = np.shape(self.c[i]) if i < len(self.c) else (rows,)
            if len(shape) == 2:
                buf = np.zeros(shape[0] * shape[1])
                self._L.CONEX_GetDualVariable(self.a, i, capi.dp(buf), shape[0], shape[1])
                xs.append(buf.reshape(shape[1], shape[0]).T.copy())
            else:
                buf = np.zeros(shape[0])
                self._L.CONEX_GetDualVariable(self.a, i, capi.dp(buf), shape[0], 1)
                xs.append(buf)
        return xs

    def ComputeErrors(self, y, xa, b):
        """Slacks s_i = c_i - A_i y and the optimality measures |b - sum_i A_i^* x_i|, <x, s>, the
        smallest eigenvalue (entry) of every s_i and x_i, for constraints added through Add*."""
        b = np.asarray(b, dtype=np.float64).ravel()
        err = Errors()
        slacks = []
        Ax = np.zeros(len(b))
        for op, c, x in zip(self.A, self.c, xa):
            s = np.asarray(c) - op.apply(y)
            slacks.append(s)
            Ax += op.adjoint(x)
            if s.ndim == 1:
                err.x_dot_s += float(s @ np.asarray(x).ravel())
                err.min_eig_S.append(float(s.min()))
                err.min_eig_X.append(float(np.asarray(x).min()))
            else:
                err.x_dot_s += float(np.trace(s @ x))
                err.min_eig_S.append(float(np.linalg.eigvalsh(0.5 * (s + s.T)).min()))
                err.min_eig_X.append(float(np.linalg.eigvalsh(0.5 * (x + np.asarray(x).T)).min()))
        err.Ax_minus_b = float(np.linalg.norm(b - Ax))
        return slacks, err
