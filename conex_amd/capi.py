"""ctypes binding of the outer CONEX_* C-ABI (include/conex.h): the role SWIG's generated
``conex`` module (interfaces/python/conex.i) plays in the reference.  :mod:`conex_amd.program`
builds the reference's Python class on top of it; the tests use it directly."""
import ctypes as C

import numpy as np

from .kkt import load_library

c_double_p = C.POINTER(C.c_double)


class SolverConfiguration(C.Structure):
    """interfaces/conex.h:10-30 field order."""
    _fields_ = [
        ("prepare_dual_variables", C.c_int),
        ("initialization_mode", C.c_int),
        ("inv_sqrt_mu_max", C.c_double),
        ("minimum_mu", C.c_double),
        ("maximum_mu", C.c_double),
        ("divergence_upper_bound", C.c_double),
        ("enable_line_search", C.c_int),
        ("dinf_upper_bound", C.c_double),
        ("final_centering_steps", C.c_int),
        ("final_centering_tolerance", C.c_double),
        ("initial_centering_steps_warmstart", C.c_int),
        ("initial_centering_steps_coldstart", C.c_int),
        ("warmstart_abort_threshold", C.c_double),
        ("max_iterations", C.c_int),
        ("iterative_refinement_iterations", C.c_int),
        ("infeasibility_threshold", C.c_double),
        ("kkt_error_tolerance", C.c_double),
        ("enable_rescaling", C.c_int),
        ("kkt_solver", C.c_int),
    ]


class IterationStats(C.Structure):
    _fields_ = [("mu", C.c_double), ("iteration_number", C.c_int)]


def api():
    L = load_library()
    vp, ci, cd = C.c_void_p, C.c_int, C.c_double
    ip = C.POINTER(C.c_int)
    sig = {
        "CONEX_CreateConeProgram": (vp, []),
        "CONEX_DeleteConeProgram": (None, [vp]),
        "CONEX_AddDenseLinearConstraint": (ci, [vp, c_double_p, ci, ci, c_double_p, ci]),
        "CONEX_AddLinearInequalities": (ci, [vp, c_double_p, ci, ci, c_double_p, ci, c_double_p, ci]),
        "CONEX_AddQuadraticCost": (ci, [vp, c_double_p, ci, ci]),
        "CONEX_AddDenseLMIConstraint": (ci, [vp, c_double_p, ci, ci, ci, c_double_p, ci, ci]),
        "CONEX_AddSparseLMIConstraint": (ci, [vp, c_double_p, ci, ci, ci, c_double_p, ci, ci,
                                              C.POINTER(C.c_long), ci]),
        "CONEX_Maximize": (ci, [vp, c_double_p, ci, C.POINTER(SolverConfiguration), c_double_p, ci]),
        "CONEX_Solve": (ci, [vp, C.POINTER(SolverConfiguration), c_double_p, ci]),
        "CONEX_GetDualVariable": (None, [vp, ci, c_double_p, ci, ci]),
        "CONEX_GetDualVariableSize": (ci, [vp, ci]),
        "CONEX_SetDefaultOptions": (None, [C.POINTER(SolverConfiguration)]),
        "CONEX_GetIterationStats": (None, [vp, C.POINTER(IterationStats), ci]),
        "CONEX_UpdateLinearOperator": (ci, [vp, ci, cd, ci, ci, ci, ci]),
        "CONEX_NewLinearMatrixInequality": (ci, [vp, ci, ci, ip]),
        "CONEX_UpdateAffineTerm": (ci, [vp, ci, cd, ci, ci, ci]),
        "CONEX_NewLorentzConeConstraint": (ci, [vp, ci, ip]),
        "CONEX_NewLinearInequality": (ci, [vp, ci, ip]),
        "CONEX_NewQuadraticCost": (ci, [vp, ip]),
        "CONEX_UpdateQuadraticCostMatrix": (ci, [vp, ci, cd, ci, ci]),
        "CONEX_SetNumberOfVariables": (ci, [vp, ci]),
        # not in conex.h (the reference reaches these cones through its C++ API only)
        "CONEX_HIP_AddQuadraticConstraint": (ci, [vp, c_double_p, ci, c_double_p, ci, ci, c_double_p, ci,
                                                  C.POINTER(C.c_long), ci]),
        "CONEX_HIP_AddQuadraticCostEpigraph": (ci, [vp, c_double_p, ci, C.POINTER(C.c_long), C.c_long]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    return L


def default_config():
    cfg = SolverConfiguration()
    api().CONEX_SetDefaultOptions(C.byref(cfg))
    return cfg


def colmajor(a):
    a = np.asarray(a, dtype=np.float64)
    if a.ndim == 2:
        return np.ascontiguousarray(a.T).ravel()
    if a.ndim == 3:
        return np.ascontiguousarray(np.transpose(a, (0, 2, 1))).ravel()
    return np.ascontiguousarray(a).ravel()


def dp(a):
    return a.ctypes.data_as(c_double_p)
