"""CPU-only checks of the drop-in boundary: the shared library loads, exports every symbol the
headers under include/ declare, and reproduces the status codes the reference's interface tests
pin (interfaces/test/interface_test.cc:5-120, interface_test_soc.cc:5-65).  No compute calls."""
import ctypes as C
import os
import re

import pytest

import conex_api as ca
from conex_amd import LIB_PATH, load_library

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SUCCESS, FAILURE = 0, 1


def _declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b((?:CONEX|cxk)_[A-Za-z_0-9]+)\s*\(", text)))


@pytest.mark.parametrize("header", ["conex.h", "conex_kkt_hip.h"])
def test_library_exports_every_declared_symbol(header):
    L = load_library()
    names = _declared(header)
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/{header} but not exported by {LIB_PATH}"


def test_conex_h_declares_the_reference_abi():
    """the 21 functions of the reference's interfaces/conex.h:41-99"""
    expected = """CONEX_CreateConeProgram CONEX_DeleteConeProgram CONEX_AddDenseLinearConstraint
    CONEX_AddLinearInequalities CONEX_AddQuadraticCost CONEX_AddDenseLMIConstraint
    CONEX_AddSparseLMIConstraint CONEX_Maximize CONEX_Solve CONEX_GetDualVariable
    CONEX_GetDualVariableSize CONEX_SetDefaultOptions CONEX_GetIterationStats
    CONEX_UpdateLinearOperator CONEX_NewLinearMatrixInequality CONEX_UpdateAffineTerm
    CONEX_NewLorentzConeConstraint CONEX_NewLinearInequality CONEX_NewQuadraticCost
    CONEX_UpdateQuadraticCostMatrix CONEX_SetNumberOfVariables""".split()
    assert sorted(expected) == _declared("conex.h")


def test_default_options_match_reference_defaults():
    cfg = ca.default_config()  # cone_program.h:17-38
    assert (cfg.prepare_dual_variables, cfg.initialization_mode) == (0, 0)
    assert cfg.inv_sqrt_mu_max == 1000 and cfg.minimum_mu == 1e-15 and cfg.maximum_mu == 1e4
    assert cfg.divergence_upper_bound == 1 and cfg.enable_line_search == 0
    assert cfg.dinf_upper_bound == 1 and cfg.final_centering_steps == 5
    assert cfg.final_centering_tolerance == .01 and cfg.warmstart_abort_threshold == 2
    assert cfg.max_iterations == 25 and cfg.infeasibility_threshold == 1e5
    assert cfg.kkt_error_tolerance == 1e10 and cfg.enable_rescaling == 1
    assert cfg.kkt_solver == 0 and cfg.iterative_refinement_iterations == 0


def test_add_lmi_status_codes():  # interface_test.cc:5-33
    L = ca.api()
    p = L.CONEX_CreateConeProgram()
    cid = C.c_int(0)
    assert L.CONEX_NewLinearMatrixInequality(p, 2, 2, C.byref(cid)) == SUCCESS and cid.value == 0
    assert L.CONEX_NewLinearMatrixInequality(p, 2, 4, C.byref(cid)) == SUCCESS and cid.value == 1
    assert L.CONEX_NewLinearMatrixInequality(None, 2, 2, C.byref(cid)) == FAILURE
    assert L.CONEX_NewLinearMatrixInequality(p, 2, 3, C.byref(cid)) == FAILURE
    assert L.CONEX_NewLinearMatrixInequality(p, 0, 2, C.byref(cid)) == FAILURE
    assert L.CONEX_NewLinearMatrixInequality(p, 4, 8, C.byref(cid)) == FAILURE  # octonion order <= 3
    assert L.CONEX_NewLinearMatrixInequality(p, 3, 8, C.byref(cid)) == SUCCESS  # octonions: order <= 3
    L.CONEX_DeleteConeProgram(p)


def test_update_lmi_status_codes():  # interface_test.cc:35-85
    L = ca.api()
    p = L.CONEX_CreateConeProgram()
    c1, c2 = C.c_int(0), C.c_int(0)
    order, h = 2, 2
    assert L.CONEX_NewLinearMatrixInequality(p, order, h, C.byref(c1)) == SUCCESS
    assert L.CONEX_NewLinearMatrixInequality(p, order, h, C.byref(c2)) == SUCCESS
    assert L.CONEX_UpdateLinearOperator(p, c1.value, .3, 2, order - 1, order - 2, h - 1) == SUCCESS
    assert L.CONEX_UpdateLinearOperator(p, c1.value, .3, 2, order - 1, order - 2, h) == FAILURE
    assert L.CONEX_UpdateLinearOperator(p, c1.value, .3, 2, order, order - 2, h - 1) == FAILURE
    assert L.CONEX_UpdateLinearOperator(p, c1.value, .3, 2, order - 1, order, h - 1) == FAILURE
    assert L.CONEX_UpdateAffineTerm(p, c1.value, .3, order - 1, order - 2, h) == FAILURE
    assert L.CONEX_UpdateAffineTerm(p, c1.value, .3, order - 1, order - 2, h - 1) == SUCCESS
    # diagonal of a skew-symmetric (imaginary) part
    assert L.CONEX_UpdateAffineTerm(p, c1.value, .3, 0, 0, h - 1) == FAILURE
    assert L.CONEX_UpdateLinearOperator(p, 7, .3, 0, 0, 0, 0) == FAILURE  # invalid constraint
    L.CONEX_DeleteConeProgram(p)


def test_set_number_of_variables_once():  # interface_test.cc:87-95
    L = ca.api()
    p = L.CONEX_CreateConeProgram()
    assert L.CONEX_SetNumberOfVariables(p, 4) == SUCCESS
    assert L.CONEX_SetNumberOfVariables(p, 4) == FAILURE
    assert L.CONEX_SetNumberOfVariables(None, 4) == FAILURE
    L.CONEX_DeleteConeProgram(p)


def test_update_quadratic_cost_status_codes():  # interface_test.cc:97-119
    L = ca.api()
    p = L.CONEX_CreateConeProgram()
    n = 4
    assert L.CONEX_SetNumberOfVariables(p, n) == SUCCESS
    cid = C.c_int(0)
    assert L.CONEX_NewQuadraticCost(p, C.byref(cid)) == SUCCESS
    for i in range(n):
        for j in range(n):
            assert L.CONEX_UpdateQuadraticCostMatrix(p, cid.value, float(i * n + j), i, j) == SUCCESS
    assert L.CONEX_UpdateQuadraticCostMatrix(p, cid.value, 1.0, n, n) == FAILURE
    L.CONEX_DeleteConeProgram(p)


def test_lorentz_cone_status_codes():  # interface_test_soc.cc:5-65
    L = ca.api()
    p = L.CONEX_CreateConeProgram()
    cid = C.c_int(0)
    assert L.CONEX_NewLorentzConeConstraint(p, 2, C.byref(cid)) == SUCCESS and cid.value == 0
    assert L.CONEX_NewLorentzConeConstraint(p, 2, C.byref(cid)) == SUCCESS and cid.value == 1
    assert L.CONEX_NewLorentzConeConstraint(None, 2, C.byref(cid)) == FAILURE
    assert L.CONEX_NewLorentzConeConstraint(p, 0, C.byref(cid)) == FAILURE
    order = 2
    assert L.CONEX_UpdateLinearOperator(p, 0, .3, 2, order - 1, 0, 0) == SUCCESS
    assert L.CONEX_UpdateLinearOperator(p, 0, .3, 2, order - 1, 0, 1) == FAILURE
    assert L.CONEX_UpdateLinearOperator(p, 0, .3, 2, -1, 0, 0) == FAILURE
    assert L.CONEX_UpdateLinearOperator(p, 0, .3, 2, order - 1, 1, 0) == FAILURE
    assert L.CONEX_UpdateAffineTerm(p, 0, .3, 0, 0, 0) == SUCCESS
    assert L.CONEX_UpdateAffineTerm(p, 0, .3, 2, 0, 0) == SUCCESS
    assert L.CONEX_UpdateAffineTerm(p, 0, .3, 2, 0, 1) == FAILURE
    assert L.CONEX_UpdateAffineTerm(p, 0, .3, 2, 1, 0) == FAILURE
    L.CONEX_DeleteConeProgram(p)


def test_numeric_path_fails_loudly_without_gpu():
    """No CPU fallback: a host-only context refuses numeric calls."""
    import numpy as np
    from conex_amd import KktContext
    from conex_amd.kkt import KktError
    k = KktContext(2, device=-1)
    k.add_static(np.eye(2), [0, 1])
    k.initialize()
    with pytest.raises(KktError):
        k.assemble()


def _normalised_header(path):
    """(prototypes, struct fields, #define constants) of a C header with comments, parameter
    names and white space removed: `int CONEX_X(void* p, const double* A, int Ar)` becomes
    ('CONEX_X', 'int', ('void*', 'const double*', 'int'))."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    defines = dict(re.findall(r"^\s*#define\s+(CONEX_[A-Z_0-9]+)\s+(\S+)\s*$", text, flags=re.M))
    text = re.sub(r"^\s*#.*$", " ", text, flags=re.M)
    text = re.sub(r'extern\s+"C"\s*\{', " ", text)

    def ctype(decl):
        decl = re.sub(r"\s+", " ", decl.strip())
        m = re.match(r"^(.*?[\s\*])([A-Za-z_][A-Za-z_0-9]*)$", decl)   # drop the parameter name
        if m and m.group(1).strip() not in ("", "const", "unsigned", "long", "struct"):
            decl = m.group(1)
        return re.sub(r"\s*\*\s*", "*", decl.strip())

    structs = {}
    for body, name in re.findall(r"typedef\s+struct[^{]*\{(.*?)\}\s*([A-Za-z_0-9]+)\s*;", text, flags=re.S):
        fields = []
        for f in body.split(";"):
            f = f.strip()
            if f:
                m = re.match(r"^(.*?[\s\*])([A-Za-z_][A-Za-z_0-9]*)$", re.sub(r"\s+", " ", f))
                fields.append((re.sub(r"\s*\*\s*", "*", m.group(1).strip()), m.group(2)))
        structs[name] = fields
    text = re.sub(r"typedef\s+struct[^{]*\{.*?\}\s*[A-Za-z_0-9]+\s*;", " ", text, flags=re.S)
    protos = {}
    for ret, name, params in re.findall(r"([A-Za-z_][A-Za-z_0-9\s\*]*?)\s*\b(CONEX_[A-Za-z_0-9]+)\s*\(([^)]*)\)\s*;",
                                        text, flags=re.S):
        ret = re.sub(r"\b(CONEX_API|extern)\b", " ", ret)
        ps = tuple(ctype(p) for p in params.split(",") if p.strip() and p.strip() != "void")
        protos[name] = (re.sub(r"\s*\*\s*", "*", re.sub(r"\s+", " ", ret.strip())), ps)
    return protos, structs, defines


REFERENCE_HEADER = "/root/reference/interfaces/conex.h"


@pytest.mark.skipif(not os.path.exists(REFERENCE_HEADER),
                    reason="the reference tree is only present in the build container")
def test_conex_h_equals_the_reference_header_prototype_by_prototype():
    """interfaces/conex.h:7-99 against include/conex.h: return types, argument TYPES in order, the
    field list (types, names, order) of CONEX_SolverConfiguration and the status constants.  The
    symbol-name test above cannot see a swapped argument or a reordered field; this one does."""
    ours = _normalised_header(os.path.join(ROOT, "include", "conex.h"))
    ref = _normalised_header(REFERENCE_HEADER)
    assert len(ref[0]) == 21 and "CONEX_SolverConfiguration" in ref[1]
    assert sorted(ours[0]) == sorted(ref[0])
    for name, sig in ref[0].items():
        assert ours[0][name] == sig, f"{name}: {ours[0][name]} != reference {sig}"
    assert ours[1]["CONEX_SolverConfiguration"] == ref[1]["CONEX_SolverConfiguration"]
    assert len(ref[1]["CONEX_SolverConfiguration"]) == 19
    for name, val in ref[2].items():
        assert ours[2].get(name) == val, name


def test_ctypes_configuration_mirror_has_the_header_field_order():
    """tests/conex_api.py's ctypes mirror of CONEX_SolverConfiguration (what every C-ABI test passes
    to CONEX_Maximize) lists the fields of include/conex.h in order and with matching C types."""
    _, structs, _ = _normalised_header(os.path.join(ROOT, "include", "conex.h"))
    fields = structs["CONEX_SolverConfiguration"]
    mirror = ca.Config._fields_ if hasattr(ca, "Config") else ca.SolverConfiguration._fields_
    assert [n for _, n in fields] == [n for n, _ in mirror]
    want = {"int": C.c_int, "double": C.c_double}
    for (ctype_, name), (_, py) in zip(fields, mirror):
        assert want[ctype_] is py, name
