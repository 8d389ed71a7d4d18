"""ctypes binding of the CPU oracle (oracle/libconex_oracle.so).

TEST INFRASTRUCTURE ONLY: the product package (conex_amd/) never imports this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_LIB = None

c_int_p = C.POINTER(C.c_int)
c_long_p = C.POINTER(C.c_long)
c_double_p = C.POINTER(C.c_double)


class Config(C.Structure):
    """cone_program.h:17-38 (oracle field order, see conex_oracle.h)."""
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
        ("infeasibility_threshold", C.c_double),
        ("kkt_error_tolerance", C.c_double),
        ("kkt_solver", C.c_int),
        ("enable_rescaling", C.c_int),
        ("iterative_refinement_iterations", C.c_int),
    ]


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "libconex_oracle.so"],
                          stdout=subprocess.DEVNULL)


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.path.join(ORACLE_DIR, "libconex_oracle.so")
    if not os.path.exists(path):
        build()
    L = C.CDLL(path)
    vp = C.c_void_p
    sig = {
        "cxo_program_new": (vp, [C.c_int]),
        "cxo_program_free": (None, [vp]),
        "cxo_add_lmi": (C.c_int, [vp, C.c_int, C.c_int, c_double_p, c_double_p, c_int_p]),
        "cxo_add_linear": (C.c_int, [vp, C.c_int, C.c_int, c_double_p, c_double_p, c_int_p]),
        "cxo_add_soc": (C.c_int, [vp, C.c_int, C.c_int, c_double_p, c_double_p, c_int_p]),
        "cxo_add_quadratic": (C.c_int, [vp, C.c_int, C.c_int, c_double_p, c_double_p, c_double_p, c_int_p]),
        "cxo_add_static": (C.c_int, [vp, C.c_int, c_double_p, c_int_p]),
        "cxo_add_equality": (C.c_int, [vp, C.c_int, C.c_int, c_double_p, c_double_p, c_int_p]),
        "cxo_add_hermitian": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, c_double_p, c_double_p,
                                        c_int_p]),
        "cxo_hc_multiply": (None, [C.c_int, C.c_int, C.c_int, C.c_int, c_double_p, c_double_p,
                                   c_double_p]),
        "cxo_hc_exponential_map": (None, [C.c_int, C.c_int, c_double_p, c_double_p]),
        "cxo_hc_quadratic_representation": (None, [C.c_int, C.c_int, c_double_p, c_double_p, c_double_p]),
        "cxo_hc_trace_inner_product": (C.c_double, [C.c_int, C.c_int, c_double_p, c_double_p]),
        "cxo_hc_geodesic_update_scaled": (None, [C.c_int, C.c_int, c_double_p, c_double_p, c_double_p]),
        "cxo_hc_approximate_eigenvalues": (C.c_int, [C.c_int, C.c_int, c_double_p, c_double_p,
                                                     c_double_p, C.c_int, c_double_p]),
        "cxo_hc_random": (C.c_double, [C.c_ulong, C.c_ulong, C.c_ulong]),
        "cxo_num_constraints": (C.c_int, [vp]),
        "cxo_initialize": (C.c_int, [vp]),
        "cxo_system_size": (C.c_int, [vp]),
        "cxo_get_order": (C.c_int, [vp, c_int_p]),
        "cxo_get_permutation": (C.c_int, [vp, c_int_p, c_int_p]),
        "cxo_get_list": (C.c_int, [vp, C.c_int, C.c_int, c_int_p]),
        "cxo_get_supernode_sizes": (C.c_int, [vp, c_int_p]),
        "cxo_slab_size": (C.c_long, [vp]),
        "cxo_get_block_offsets": (C.c_int, [vp, c_long_p, c_long_p]),
        "cxo_get_ss_index": (C.c_int, [vp, C.c_int, c_long_p]),
        "cxo_set_identity": (None, [vp]),
        "cxo_dual_size": (C.c_int, [vp, C.c_int]),
        "cxo_get_W": (None, [vp, C.c_int, c_double_p]),
        "cxo_set_W": (None, [vp, C.c_int, c_double_p]),
        "cxo_assemble": (None, [vp]),
        "cxo_get_slab": (None, [vp, c_double_p]),
        "cxo_get_constraint_schur": (None, [vp, C.c_int, c_double_p, c_double_p, c_double_p,
                                            c_double_p]),
        "cxo_get_residuals": (None, [vp, c_double_p, c_double_p, c_double_p]),
        "cxo_factor": (C.c_int, [vp]),
        "cxo_solve_inplace": (None, [vp, c_double_p]),
        "cxo_set_refinement": (None, [vp, C.c_int]),
        "cxo_kkt_matrix": (None, [vp, c_double_p]),
        "cxo_prepare_step": (None, [vp, C.c_int, C.c_double, C.c_double, c_double_p, c_double_p]),
        "cxo_take_step": (None, [vp, C.c_int, C.c_double, C.c_double]),
        "cxo_weighted_slack_eigenvalues": (None, [vp, c_double_p, C.c_double, c_double_p]),
        "cxo_kkt_solve": (C.c_int, [vp, c_double_p, C.c_double, C.c_double, C.c_double,
                                    c_double_p]),
        "cxo_solve": (C.c_int, [vp, c_double_p, C.POINTER(Config), c_double_p]),
        "cxo_num_iterations": (C.c_int, [vp]),
        "cxo_iteration_mu": (C.c_double, [vp, C.c_int]),
        "cxo_get_dual_variable": (None, [vp, C.c_int, c_double_p]),
        "cxo_set_verbose": (None, [C.c_int]),
        "cxo_set_strict_direct_update": (None, [C.c_int]),
        "cxo_default_config": (None, [C.POINTER(Config)]),
        "cxo_path_in_tree": (C.c_int, [C.c_int, C.c_int, C.c_int, c_int_p, c_int_p, c_int_p]),
        "cxo_pick_clique_order": (C.c_int, [C.c_int, c_int_p, c_int_p, C.c_int, c_int_p, c_int_p,
                                            c_int_p, c_int_p, c_int_p]),
        "cxo_pade": (None, [C.c_int, c_double_p, c_double_p]),
        "cxo_lanczos_asym": (C.c_int, [C.c_int, c_double_p, c_double_p, c_double_p, C.c_int,
                                       c_double_p]),
        "cxo_lanczos_sym": (C.c_int, [C.c_int, c_double_p, c_double_p, C.c_int, c_double_p]),
        "cxo_jacobi": (C.c_int, [C.c_int, c_double_p, c_double_p, c_double_p, C.c_int,
                                 c_double_p]),
        "cxo_tridiag_eigs": (C.c_int, [C.c_int, c_double_p, c_double_p, c_double_p]),
        "cxo_divergence_upper_bound_inverse": (C.c_double, [C.c_double, c_double_p]),
        "cxo_divergence_upper_bound": (C.c_double, [C.c_double, c_double_p]),
        "cxo_ws_new": (vp, [C.c_int, c_int_p, c_int_p, c_int_p]),
        "cxo_ws_free": (None, [vp]),
        "cxo_ws_N": (C.c_int, [vp]),
        "cxo_ws_slab_size": (C.c_long, [vp]),
        "cxo_ws_slab": (c_double_p, [vp]),
        "cxo_ws_offsets": (None, [vp, c_long_p, c_long_p]),
        "cxo_ws_ss_index": (C.c_int, [vp, C.c_int, c_long_p]),
        "cxo_ws_cholesky": (C.c_int, [vp]),
        "cxo_ws_forward": (None, [vp, c_double_p]),
        "cxo_ws_backward": (None, [vp, c_double_p]),
        "cxo_ws_to_dense": (None, [vp, c_double_p]),
        "cxo_ws_ldlt": (C.c_int, [vp]),
        "cxo_ws_solve_ldlt": (None, [vp, c_double_p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _LIB = L
    return L


def dp(a):
    return a.ctypes.data_as(c_double_p)


def ip(a):
    return a.ctypes.data_as(c_int_p)


def lp(a):
    return a.ctypes.data_as(c_long_p)


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def colmajor(a):
    """flat column-major copy of a 2-D (or stack of 2-D) array."""
    a = np.asarray(a, dtype=np.float64)
    if a.ndim == 2:
        return np.ascontiguousarray(a.T).ravel()
    if a.ndim == 3:  # (m, n, n): each matrix column-major
        return np.ascontiguousarray(np.transpose(a, (0, 2, 1))).ravel()
    return np.ascontiguousarray(a).ravel()


def planes_colmajor(a):
    """(..., n, n) stacks of real planes -> flat buffer with every plane column-major."""
    a = np.asarray(a, dtype=np.float64)
    return np.ascontiguousarray(np.swapaxes(a, -1, -2)).ravel()


def flatten_lists(lists):
    ptr = np.zeros(len(lists) + 1, dtype=np.int32)
    for i, l in enumerate(lists):
        ptr[i + 1] = ptr[i] + len(l)
    idx = np.array([x for l in lists for x in l], dtype=np.int32)
    if idx.size == 0:
        idx = np.zeros(1, dtype=np.int32)
    return ptr, idx


class Program:
    """Thin OO veneer over cxo_* mirroring conex::Program (cone_program.h:99-233)."""

    def __init__(self, num_vars):
        self.L = lib()
        self.h = self.L.cxo_program_new(num_vars)
        self.num_vars = num_vars
        self.cons = []  # (type, n, m)

    def __del__(self):
        try:
            self.L.cxo_program_free(self.h)
        except Exception:
            pass

    @staticmethod
    def _vars(vars_):
        if vars_ is None:
            return None, None
        v = np.ascontiguousarray(vars_, dtype=np.int32)
        return v, ip(v)

    def add_lmi(self, A, Cm, vars_=None):
        A = np.asarray(A, dtype=np.float64)
        m, n = A.shape[0], A.shape[1]
        a = colmajor(A)
        c = colmajor(np.asarray(Cm, dtype=np.float64))
        v, vp_ = self._vars(vars_)
        r = self.L.cxo_add_lmi(self.h, n, m, dp(a), dp(c), vp_)
        if r >= 0:
            self.cons.append(("lmi", n, m))
        return r

    def add_hermitian(self, A, Cm, vars_=None):
        """A: (m, d, n, n) real planes (plane 0 symmetric, the others skew), Cm: (d, n, n)."""
        A = np.asarray(A, dtype=np.float64)
        m, d, n = A.shape[0], A.shape[1], A.shape[2]
        a = planes_colmajor(A)
        c = planes_colmajor(np.asarray(Cm, dtype=np.float64))
        v, vp_ = self._vars(vars_)
        r = self.L.cxo_add_hermitian(self.h, n, d, m, dp(a), dp(c), vp_)
        if r >= 0:
            self.cons.append(("herm", n, m, d))
        return r

    def add_equality(self, A, b, vars_=None):
        """A y[vars] = b (EqualityConstraints{A, b}); appends A.shape[0] multipliers."""
        A = np.asarray(A, dtype=np.float64)
        r_, m = A.shape
        a = colmajor(A)
        bb = f64(np.asarray(b).ravel())
        v, vp_ = self._vars(vars_)
        r = self.L.cxo_add_equality(self.h, r_, m, dp(a), dp(bb), vp_)
        if r >= 0:
            self.cons.append(("eq", r_, m + r_))
        return r

    def add_linear(self, A, c, vars_=None):
        A = np.asarray(A, dtype=np.float64)
        r_, m = A.shape
        a = colmajor(A)
        cc = f64(np.asarray(c).ravel())
        v, vp_ = self._vars(vars_)
        r = self.L.cxo_add_linear(self.h, r_, m, dp(a), dp(cc), vp_)
        if r >= 0:
            self.cons.append(("linear", r_, m))
        return r

    def add_soc(self, A, c, vars_=None):
        A = np.asarray(A, dtype=np.float64)
        n1, m = A.shape
        a = colmajor(A)
        cc = f64(np.asarray(c).ravel())
        v, vp_ = self._vars(vars_)
        r = self.L.cxo_add_soc(self.h, n1 - 1, m, dp(a), dp(cc), vp_)
        if r >= 0:
            self.cons.append(("soc", n1 - 1, m))
        return r

    def add_quadratic(self, Q, A, c, vars_=None):
        A = np.asarray(A, dtype=np.float64)
        n1, m = A.shape
        a = colmajor(A)
        cc = f64(np.asarray(c).ravel())
        q = None if Q is None else colmajor(np.asarray(Q, dtype=np.float64))
        v, vp_ = self._vars(vars_)
        r = self.L.cxo_add_quadratic(self.h, n1 - 1, m, None if q is None else dp(q), dp(a), dp(cc), vp_)
        if r >= 0:
            self.cons.append(("quad", n1 - 1, m))
        return r

    def add_static(self, G, vars_):
        G = np.asarray(G, dtype=np.float64)
        m = G.shape[0]
        g = colmajor(G)
        v, vp_ = self._vars(vars_)
        r = self.L.cxo_add_static(self.h, m, dp(g), vp_)
        if r >= 0:
            self.cons.append(("static", 0, m))
        return r

    def initialize(self):
        return self.L.cxo_initialize(self.h)

    # --- symbolic ---
    @property
    def K(self):
        return self.L.cxo_num_constraints(self.h)

    @property
    def N(self):
        return self.L.cxo_system_size(self.h)

    def order(self):
        o = np.zeros(self.K, dtype=np.int32)
        self.L.cxo_get_order(self.h, ip(o))
        return o

    def permutation(self):
        nv = self.num_vars
        p = np.zeros(max(nv, self.N) + 1, dtype=np.int32)
        q = np.zeros(max(nv, self.N) + 1, dtype=np.int32)
        n = self.L.cxo_get_permutation(self.h, ip(p), ip(q))
        return p[:n], q[:n]

    def get_list(self, which, e):
        n = self.L.cxo_get_list(self.h, which, e, None)
        out = np.zeros(max(n, 1), dtype=np.int32)
        self.L.cxo_get_list(self.h, which, e, ip(out))
        return out[:n]

    def supernode_sizes(self):
        o = np.zeros(self.K, dtype=np.int32)
        self.L.cxo_get_supernode_sizes(self.h, ip(o))
        return o

    def slab_size(self):
        return self.L.cxo_slab_size(self.h)

    def block_offsets(self):
        d = np.zeros(self.K, dtype=np.int64)
        o = np.zeros(self.K, dtype=np.int64)
        self.L.cxo_get_block_offsets(self.h, lp(d), lp(o))
        return d, o

    def ss_index(self, e):
        n = self.L.cxo_get_ss_index(self.h, e, None)
        out = np.zeros(max(n, 1), dtype=np.int64)
        self.L.cxo_get_ss_index(self.h, e, lp(out))
        return out[:n]

    # --- numeric ---
    def set_identity(self):
        self.L.cxo_set_identity(self.h)

    def get_W(self, i):
        n = self.L.cxo_dual_size(self.h, i)
        w = np.zeros(max(n, 1))
        self.L.cxo_get_W(self.h, i, dp(w))
        if self.cons[i][0] == "herm":   # (d, n, n) logical planes
            _, order, _, d = self.cons[i]
            return np.transpose(w[:n].reshape(d, order, order), (0, 2, 1)).copy()
        return w[:n]

    def set_W(self, i, w):
        if self.cons[i][0] == "herm":
            w = planes_colmajor(np.asarray(w, dtype=np.float64))
        else:
            w = f64(np.asarray(w).ravel())
        self.L.cxo_set_W(self.h, i, dp(w))

    def assemble(self):
        self.L.cxo_assemble(self.h)

    def slab(self):
        s = np.zeros(self.slab_size())
        self.L.cxo_get_slab(self.h, dp(s))
        return s

    def constraint_schur(self, i):
        m = self.cons[i][2]
        G = np.zeros(m * m)
        AW = np.zeros(m)
        AQc = np.zeros(m)
        sc = np.zeros(2)
        self.L.cxo_get_constraint_schur(self.h, i, dp(G), dp(AW), dp(AQc), dp(sc))
        return G.reshape(m, m).T.copy(), AW, AQc, sc

    def residuals(self):
        N = self.N
        AW = np.zeros(N)
        AQc = np.zeros(N)
        sc = np.zeros(2)
        self.L.cxo_get_residuals(self.h, dp(AW), dp(AQc), dp(sc))
        return AW, AQc, sc

    def factor(self):
        return self.L.cxo_factor(self.h)

    def solve_inplace(self, y):
        y = f64(y).copy()
        self.L.cxo_solve_inplace(self.h, dp(y))
        return y

    def set_refinement(self, iterations):
        self.L.cxo_set_refinement(self.h, int(iterations))

    def kkt_matrix(self):
        N = self.N
        out = np.zeros(N * N)
        self.L.cxo_kkt_matrix(self.h, dp(out))
        return out.reshape(N, N).T.copy()

    def prepare_step(self, y, c_weight, e_weight=1.0, affine=0):
        y = f64(y)
        info = np.zeros(2)
        self.L.cxo_prepare_step(self.h, affine, c_weight, e_weight, dp(y), dp(info))
        return info

    def take_step(self, step_size, e_weight=1.0, affine=0):
        self.L.cxo_take_step(self.h, affine, e_weight, step_size)

    def weighted_slack_eigenvalues(self, y, c_weight):
        y = f64(y)
        out = np.zeros(4)
        self.L.cxo_weighted_slack_eigenvalues(self.h, dp(y), c_weight, dp(out))
        return out

    def kkt_solve(self, b, inv_sqrt_mu, b_scaling=1.0, c_scaling=1.0):
        b = f64(b)
        y = np.zeros(self.N)
        ok = self.L.cxo_kkt_solve(self.h, dp(b), inv_sqrt_mu, b_scaling, c_scaling, dp(y))
        return ok, y

    def solve(self, b, cfg=None):
        if cfg is None:
            cfg = default_config()
        b = f64(np.asarray(b).ravel())
        y = np.zeros(max(self.num_vars, 1))
        ok = self.L.cxo_solve(self.h, dp(b), C.byref(cfg), dp(y))
        return ok, y

    def dual_variable(self, i):
        n = self.L.cxo_dual_size(self.h, i)
        w = np.zeros(max(n, 1))
        self.L.cxo_get_dual_variable(self.h, i, dp(w))
        return w[:n]

    def iteration_mu(self, i):
        return self.L.cxo_iteration_mu(self.h, int(i))

    def num_iterations(self):
        return self.L.cxo_num_iterations(self.h)


def default_config():
    cfg = Config()
    lib().cxo_default_config(C.byref(cfg))
    return cfg


def pick_clique_order(cliques, root):
    L = lib()
    K = len(cliques)
    ptr, idx = flatten_lists(cliques)
    nv = max(max(c) for c in cliques) + 1
    order = np.zeros(K, dtype=np.int32)
    sn_ptr = np.zeros(K + 1, dtype=np.int32)
    sep_ptr = np.zeros(K + 1, dtype=np.int32)
    sn_idx = np.zeros(nv * K + 1, dtype=np.int32)
    sep_idx = np.zeros(nv * K + 1, dtype=np.int32)
    L.cxo_pick_clique_order(K, ip(ptr), ip(idx), root, ip(order), ip(sn_ptr), ip(sn_idx),
                            ip(sep_ptr), ip(sep_idx))
    sn = [list(sn_idx[sn_ptr[i]:sn_ptr[i + 1]]) for i in range(K)]
    sep = [list(sep_idx[sep_ptr[i]:sep_ptr[i + 1]]) for i in range(K)]
    return list(order), sn, sep


class Workspace:
    """Raw TriangularMatrixWorkspace (triangular_matrix_workspace.h) over given path."""

    def __init__(self, path, supernode_size):
        self.L = lib()
        self.K = len(path)
        ptr, idx = flatten_lists(path)
        sz = np.ascontiguousarray(supernode_size, dtype=np.int32)
        self.h = self.L.cxo_ws_new(self.K, ip(ptr), ip(idx), ip(sz))
        self.N = self.L.cxo_ws_N(self.h)
        n = self.L.cxo_ws_slab_size(self.h)
        self.slab = np.ctypeslib.as_array(self.L.cxo_ws_slab(self.h), shape=(max(n, 1),))[:n]
        self.diag_off = np.zeros(self.K, dtype=np.int64)
        self.offd_off = np.zeros(self.K, dtype=np.int64)
        self.L.cxo_ws_offsets(self.h, lp(self.diag_off), lp(self.offd_off))
        self.supernode_size = list(supernode_size)
        self.path = [list(p) for p in path]

    def __del__(self):
        try:
            self.L.cxo_ws_free(self.h)
        except Exception:
            pass

    def ss_index(self, e):
        n = self.L.cxo_ws_ss_index(self.h, e, None)
        out = np.zeros(max(n, 1), dtype=np.int64)
        self.L.cxo_ws_ss_index(self.h, e, lp(out))
        return out[:n]

    def diag(self, e):
        ns = self.supernode_size[e]
        return self.slab[self.diag_off[e]:self.diag_off[e] + ns * ns].reshape(ns, ns).T

    def offd(self, e):
        ns = self.supernode_size[e]
        s = len(self.path[e]) - ns
        return self.slab[self.offd_off[e]:self.offd_off[e] + ns * s].reshape(s, ns).T

    def cholesky(self):
        return self.L.cxo_ws_cholesky(self.h)

    def forward(self, y):
        y = f64(y).copy()
        self.L.cxo_ws_forward(self.h, dp(y))
        return y

    def backward(self, y):
        y = f64(y).copy()
        self.L.cxo_ws_backward(self.h, dp(y))
        return y

    def ldlt(self):
        return self.L.cxo_ws_ldlt(self.h)

    def solve_ldlt(self, y):
        y = f64(y).copy()
        self.L.cxo_ws_solve_ldlt(self.h, dp(y))
        return y

    def to_dense(self):
        out = np.zeros(self.N * self.N)
        self.L.cxo_ws_to_dense(self.h, dp(out))
        return out.reshape(self.N, self.N).T.copy()
