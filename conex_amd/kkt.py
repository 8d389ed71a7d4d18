"""ctypes binding of the cxk_* C-ABI (include/conex_kkt_hip.h).

Mirrors conex::Program / SupernodalKKTSolver method names (cone_program.h:99-233,
kkt_solver.h:16-65) so parity tests read like the reference's own tests.
"""
import ctypes as C
import os

import numpy as np

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libconex.so")
_LIB = None

c_int_p = C.POINTER(C.c_int)
c_long_p = C.POINTER(C.c_long)
c_double_p = C.POINTER(C.c_double)


def _dp(a):
    return a.ctypes.data_as(c_double_p)


def _ip(a):
    return a.ctypes.data_as(c_int_p)


def _lp(a):
    return a.ctypes.data_as(c_long_p)


_SIGNATURES = {
    "cxk_create": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    "cxk_destroy": (None, [C.c_void_p]),
    "cxk_last_error": (C.c_char_p, [C.c_void_p]),
    "cxk_add_lmi": (C.c_int, [C.c_void_p, C.c_int, C.c_int, c_double_p, c_double_p, c_int_p]),
    "cxk_add_linear": (C.c_int, [C.c_void_p, C.c_int, C.c_int, c_double_p, c_double_p, c_int_p]),
    "cxk_add_soc": (C.c_int, [C.c_void_p, C.c_int, C.c_int, c_double_p, c_double_p, c_int_p]),
    "cxk_add_quadratic": (C.c_int, [C.c_void_p, C.c_int, C.c_int, c_double_p, c_double_p, c_double_p, c_int_p]),
    "cxk_add_equality": (C.c_int, [C.c_void_p, C.c_int, C.c_int, c_double_p, c_double_p, c_int_p]),
    "cxk_factor_regularized": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "cxk_add_hermitian": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, c_double_p, c_double_p,
                                    c_int_p]),
    "cxk_add_static": (C.c_int, [C.c_void_p, C.c_int, c_double_p, c_int_p]),
    "cxk_num_constraints": (C.c_int, [C.c_void_p]),
    "cxk_set_shard": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "cxk_finalize": (C.c_int, [C.c_void_p]),
    "cxk_system_size": (C.c_int, [C.c_void_p]),
    "cxk_get_order": (C.c_int, [C.c_void_p, c_int_p]),
    "cxk_get_permutation": (C.c_int, [C.c_void_p, c_int_p, c_int_p]),
    "cxk_get_list": (C.c_int, [C.c_void_p, C.c_int, C.c_int, c_int_p]),
    "cxk_get_supernode_sizes": (C.c_int, [C.c_void_p, c_int_p]),
    "cxk_slab_size": (C.c_long, [C.c_void_p]),
    "cxk_get_block_offsets": (C.c_int, [C.c_void_p, c_long_p, c_long_p]),
    "cxk_get_ss_index": (C.c_int, [C.c_void_p, C.c_int, c_long_p]),
    "cxk_num_levels": (C.c_int, [C.c_void_p]),
    "cxk_set_identity": (C.c_int, [C.c_void_p]),
    "cxk_dual_size": (C.c_int, [C.c_void_p, C.c_int]),
    "cxk_get_W": (C.c_int, [C.c_void_p, C.c_int, c_double_p]),
    "cxk_set_W": (C.c_int, [C.c_void_p, C.c_int, c_double_p]),
    "cxk_assemble": (C.c_int, [C.c_void_p]),
    "cxk_factor": (C.c_int, [C.c_void_p, c_int_p]),
    "cxk_set_cost": (C.c_int, [C.c_void_p, c_double_p]),
    "cxk_newton_direction": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_double]),
    "cxk_solve_rhs": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_double]),
    "cxk_step_scalars": (C.c_int, [C.c_void_p, c_double_p]),
    "cxk_kkt_solve_async": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_double]),
    "cxk_sync": (C.c_int, [C.c_void_p, c_int_p]),
    "cxk_solve_inplace": (C.c_int, [C.c_void_p, c_double_p]),
    "cxk_get_y": (C.c_int, [C.c_void_p, c_double_p]),
    "cxk_set_y": (C.c_int, [C.c_void_p, c_double_p]),
    "cxk_prepare_step": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_double, c_double_p]),
    "cxk_take_step": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_double]),
    "cxk_prepare_take_step": (C.c_int, [C.c_void_p, C.c_double, C.c_double, c_double_p, C.POINTER(C.c_int)]),
    "cxk_weighted_slack_eigenvalues": (C.c_int, [C.c_void_p, C.c_double, c_double_p]),
    "cxk_get_slab": (C.c_int, [C.c_void_p, c_double_p]),
    "cxk_set_slab": (C.c_int, [C.c_void_p, c_double_p]),
    "cxk_get_constraint_schur": (C.c_int, [C.c_void_p, C.c_int, c_double_p, c_double_p,
                                           c_double_p, c_double_p]),
    "cxk_get_residuals": (C.c_int, [C.c_void_p, c_double_p, c_double_p, c_double_p]),
    "cxk_exchange_buffer": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), c_long_p]),
    "cxk_exchange_download": (C.c_int, [C.c_void_p, c_double_p]),
    "cxk_exchange_upload": (C.c_int, [C.c_void_p, c_double_p]),
    "cxk_kkt_local_async": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_double]),
    "cxk_kkt_finish_async": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_double]),
    "cxk_owns_constraint": (C.c_int, [C.c_void_p, C.c_int]),
    "cxk_get_valid_variables": (C.c_int, [C.c_void_p, C.POINTER(C.c_ubyte)]),
    "cxk_shard_info": (C.c_int, [C.c_void_p, c_int_p, c_int_p, c_long_p]),
    "cxk_assemble_local": (C.c_int, [C.c_void_p]),
    "cxk_finish_assemble": (C.c_int, [C.c_void_p]),
    "cxk_assembly_work": (C.c_int, [C.c_void_p, c_double_p, c_double_p]),
    "cxk_count_sparse_lmi": (C.c_int, [C.c_void_p]),
    "cxk_count_lmi_kernel": (C.c_int, [C.c_void_p, C.c_int]),
    "cxk_fused_assembly": (C.c_int, [C.c_void_p]),
    "cxk_fused_tree": (C.c_int, [C.c_void_p]),
    "cxk_comm_init_rccl_solo": (C.c_int, [C.c_void_p]),
    "cxk_set_reference_identity": (C.c_int, [C.c_void_p, C.c_int]),
    "cxk_set_solver_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "cxk_phase_timers": (C.c_int, [C.c_void_p, C.c_int]),
    "cxk_phase_mark": (C.c_int, [C.c_void_p, C.c_int]),
    "cxk_phase_read": (C.c_int, [C.c_void_p, c_double_p, C.c_int]),
    "cxk_comm_unique_id": (C.c_int, [C.c_void_p]),
    "cxk_comm_init_rccl": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "cxk_comm_set_allreduce": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "cxk_comm_selftest": (C.c_int, [C.c_void_p, C.c_int]),
    "cxk_comm_count": (C.c_int, [C.c_void_p]),
    "cxk_dense_top_columns": (C.c_int, [C.c_void_p]),
    "cxk_factor_async": (C.c_int, [C.c_void_p]),
    "cxk_factor_solve_async": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_double]),
    "cxk_factor_direction_async": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_double]),
    "cxk_factor_status": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "cxk_step_scalars_async": (C.c_int, [C.c_void_p]),
    "cxk_device_mu_supported": (C.c_int, [C.c_void_p]),
    "cxk_triple_supported": (C.c_int, [C.c_void_p]),
    "cxk_factor_solve_triple_async": (C.c_int, [C.c_void_p, C.c_double, C.c_double]),
    "cxk_select_mu_async": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_double, C.c_double,
                                      C.c_double]),
    "cxk_newton_direction_device_mu": (C.c_int, [C.c_void_p, C.c_double, C.c_double]),
    "cxk_prepare_take_step_device_mu": (C.c_int, [C.c_void_p, C.c_double, C.c_double, c_double_p,
                                                  C.POINTER(C.c_int), c_double_p]),
    "cxk_kernel_time": (C.c_int, [C.c_void_p, C.c_int, c_double_p]),
    "cxk_enable_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "cxk_kernel_clock": (C.c_int, [C.c_void_p, C.c_int, C.c_int, c_double_p]),
    "cxk_set_chain_segments": (C.c_int, [C.c_void_p, C.c_int]),
    "cxk_chain_segments": (C.c_int, [C.c_void_p]),
    "cxk_fused_tree_timed_out": (C.c_int, [C.c_void_p]),
    "cxk_debug_force_fused_timeout": (C.c_int, [C.c_void_p]),
    "cxk_set_iterative_refinement": (C.c_int, [C.c_void_p, C.c_int]),
}


def load_library():
    """Load libconex.so; raises (loudly) when the HIP library has not been built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build the HIP library first (python __graft_entry__.py or "
            "make -C conex_amd/csrc). There is no CPU fallback for the KKT path.")
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(L, name)  # AttributeError if an ABI symbol is missing
        fn.restype = res
        fn.argtypes = args
    _LIB = L
    return L


def _colmajor(a):
    a = np.asarray(a, dtype=np.float64)
    if a.ndim == 2:
        return np.ascontiguousarray(a.T).ravel()
    if a.ndim == 3:
        return np.ascontiguousarray(np.transpose(a, (0, 2, 1))).ravel()
    return np.ascontiguousarray(a).ravel()


class KktError(RuntimeError):
    pass


class KktContext:
    """Device-resident cone program (one HIP device, one stream)."""

    def __init__(self, num_vars, device=0, stream=None):
        self.L = load_library()
        h = C.c_void_p()
        rc = self.L.cxk_create(num_vars, device, C.c_void_p(stream) if stream else None,
                               C.byref(h))
        if rc != 0:
            raise KktError(f"cxk_create failed for device {device} (no usable HIP device?)")
        self.h = h
        self.num_vars = num_vars
        self.cons = []

    def close(self):
        if getattr(self, "h", None):
            self.L.cxk_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise KktError(f"{what}: {self.L.cxk_last_error(self.h).decode()}")

    @staticmethod
    def _vars(v):
        if v is None:
            return None, None
        a = np.ascontiguousarray(v, dtype=np.int32)
        return a, _ip(a)

    # ---- construction
    def add_lmi(self, A, Cm, vars_=None):
        A = np.asarray(A, dtype=np.float64)
        m, n = A.shape[0], A.shape[1]
        a, c = _colmajor(A), _colmajor(np.asarray(Cm, dtype=np.float64))
        keep, vp = self._vars(vars_)
        r = self.L.cxk_add_lmi(self.h, n, m, _dp(a), _dp(c), vp)
        if r >= 0:
            self.cons.append(("lmi", n, m))
        return r

    def add_linear(self, A, c, vars_=None):
        A = np.asarray(A, dtype=np.float64)
        rows, m = A.shape
        a = _colmajor(A)
        cc = np.ascontiguousarray(np.asarray(c, dtype=np.float64).ravel())
        keep, vp = self._vars(vars_)
        r = self.L.cxk_add_linear(self.h, rows, m, _dp(a), _dp(cc), vp)
        if r >= 0:
            self.cons.append(("linear", rows, m))
        return r

    def add_hermitian(self, A, Cm, vars_=None):
        """Hermitian PSD over R / C / H / O: A (m, d, n, n) real planes, Cm (d, n, n); d in {1, 2, 4, 8}
        (octonions, d = 8: order at most 3)."""
        A = np.asarray(A, dtype=np.float64)
        m, d, n = A.shape[0], A.shape[1], A.shape[2]
        a = np.ascontiguousarray(np.swapaxes(A, -1, -2)).ravel()
        c = np.ascontiguousarray(np.swapaxes(np.asarray(Cm, dtype=np.float64), -1, -2)).ravel()
        keep, vp = self._vars(vars_)
        r = self.L.cxk_add_hermitian(self.h, n, d, m, _dp(a), _dp(c), vp)
        if r >= 0:
            self.cons.append(("herm", n, m, d))
        return r

    def add_equality(self, A, b, vars_=None):
        """A y[vars] = b; appends A.shape[0] multipliers and switches the solver to LDLT."""
        A = np.asarray(A, dtype=np.float64)
        rows, m = A.shape
        a = _colmajor(A)
        bb = np.ascontiguousarray(np.asarray(b, dtype=np.float64).ravel())
        keep, vp = self._vars(vars_)
        r = self.L.cxk_add_equality(self.h, rows, m, _dp(a), _dp(bb), vp)
        if r >= 0:
            self.cons.append(("eq", rows, m + rows))
        return r

    def factor_regularized(self):
        f = C.c_int(0)
        self._check(self.L.cxk_factor_regularized(self.h, C.byref(f)), "cxk_factor_regularized")
        return f.value

    def add_soc(self, A, c, vars_=None):
        A = np.asarray(A, dtype=np.float64)
        n1, m = A.shape
        a = _colmajor(A)
        cc = np.ascontiguousarray(np.asarray(c, dtype=np.float64).ravel())
        keep, vp = self._vars(vars_)
        r = self.L.cxk_add_soc(self.h, n1 - 1, m, _dp(a), _dp(cc), vp)
        if r >= 0:
            self.cons.append(("soc", n1 - 1, m))
        return r

    def add_quadratic(self, Q, A, c, vars_=None):
        """QuadraticConstraint(Q, A, c): Q (n, n) or None (identity), A (n + 1, m), c (n + 1,)."""
        A = np.asarray(A, dtype=np.float64)
        n1, m = A.shape
        a = _colmajor(A)
        cc = np.ascontiguousarray(np.asarray(c, dtype=np.float64).ravel())
        q = None if Q is None else _colmajor(np.asarray(Q, dtype=np.float64))
        keep, vp = self._vars(vars_)
        r = self.L.cxk_add_quadratic(self.h, n1 - 1, m, None if q is None else _dp(q), _dp(a), _dp(cc), vp)
        if r >= 0:
            self.cons.append(("quad", n1 - 1, m))
        return r

    def add_static(self, G, vars_):
        G = np.asarray(G, dtype=np.float64)
        m = G.shape[0]
        g = _colmajor(G)
        keep, vp = self._vars(vars_)
        r = self.L.cxk_add_static(self.h, m, _dp(g), vp)
        if r >= 0:
            self.cons.append(("static", 0, m))
        return r

    def set_shard(self, rank, world):
        self._check(self.L.cxk_set_shard(self.h, rank, world), "cxk_set_shard")

    def initialize(self):
        self._check(self.L.cxk_finalize(self.h), "cxk_finalize")
        return 1

    # ---- symbolic
    @property
    def K(self):
        return self.L.cxk_num_constraints(self.h)

    @property
    def N(self):
        return self.L.cxk_system_size(self.h)

    def order(self):
        o = np.zeros(self.K, dtype=np.int32)
        self.L.cxk_get_order(self.h, _ip(o))
        return o

    def permutation(self):
        n = max(self.num_vars, self.N) + 1
        p = np.zeros(n, dtype=np.int32)
        q = np.zeros(n, dtype=np.int32)
        k = self.L.cxk_get_permutation(self.h, _ip(p), _ip(q))
        return p[:k], q[:k]

    def get_list(self, which, e):
        n = self.L.cxk_get_list(self.h, which, e, None)
        out = np.zeros(max(n, 1), dtype=np.int32)
        self.L.cxk_get_list(self.h, which, e, _ip(out))
        return out[:n]

    def supernode_sizes(self):
        o = np.zeros(self.K, dtype=np.int32)
        self.L.cxk_get_supernode_sizes(self.h, _ip(o))
        return o

    def slab_size(self):
        return self.L.cxk_slab_size(self.h)

    def block_offsets(self):
        d = np.zeros(self.K, dtype=np.int64)
        o = np.zeros(self.K, dtype=np.int64)
        self.L.cxk_get_block_offsets(self.h, _lp(d), _lp(o))
        return d, o

    def ss_index(self, e):
        n = self.L.cxk_get_ss_index(self.h, e, None)
        out = np.zeros(max(n, 1), dtype=np.int64)
        self.L.cxk_get_ss_index(self.h, e, _lp(out))
        return out[:n]

    def num_levels(self):
        return self.L.cxk_num_levels(self.h)

    # ---- numeric
    def set_identity(self):
        self._check(self.L.cxk_set_identity(self.h), "cxk_set_identity")

    def get_W(self, i):
        n = self.L.cxk_dual_size(self.h, i)
        w = np.zeros(max(n, 1))
        self._check(self.L.cxk_get_W(self.h, i, _dp(w)), "cxk_get_W")
        if self.cons[i][0] == "herm":   # (d, n, n) logical planes
            _, order, _, d = self.cons[i]
            return np.transpose(w[:n].reshape(d, order, order), (0, 2, 1)).copy()
        return w[:n]

    def set_W(self, i, w):
        if self.cons[i][0] == "herm":
            w = np.ascontiguousarray(np.swapaxes(np.asarray(w, dtype=np.float64), -1, -2)).ravel()
        else:
            w = np.ascontiguousarray(np.asarray(w, dtype=np.float64).ravel())
        self._check(self.L.cxk_set_W(self.h, i, _dp(w)), "cxk_set_W")

    def assemble(self):
        self._check(self.L.cxk_assemble(self.h), "cxk_assemble")

    def slab(self):
        s = np.zeros(self.slab_size())
        self._check(self.L.cxk_get_slab(self.h, _dp(s)), "cxk_get_slab")
        return s

    def set_slab(self, s):
        s = np.ascontiguousarray(s, dtype=np.float64)
        self._check(self.L.cxk_set_slab(self.h, _dp(s)), "cxk_set_slab")

    def constraint_schur(self, i):
        m = self.cons[i][2]
        G = np.zeros(m * m)
        AW = np.zeros(m)
        AQc = np.zeros(m)
        sc = np.zeros(2)
        self._check(self.L.cxk_get_constraint_schur(self.h, i, _dp(G), _dp(AW), _dp(AQc), _dp(sc)),
                    "cxk_get_constraint_schur")
        return G.reshape(m, m).T.copy(), AW, AQc, sc

    def residuals(self):
        N = self.N
        AW = np.zeros(N)
        AQc = np.zeros(N)
        sc = np.zeros(2)
        self._check(self.L.cxk_get_residuals(self.h, _dp(AW), _dp(AQc), _dp(sc)),
                    "cxk_get_residuals")
        return AW, AQc, sc

    def set_refinement(self, iterations):
        """SetIterativeRefinementIterations: takes effect at the next factorization."""
        self._check(self.L.cxk_set_iterative_refinement(self.h, int(iterations)),
                    "cxk_set_iterative_refinement")

    def factor(self):
        ok = C.c_int(0)
        self._check(self.L.cxk_factor(self.h, C.byref(ok)), "cxk_factor")
        return ok.value

    def solve_inplace(self, y):
        y = np.ascontiguousarray(y, dtype=np.float64).copy()
        self._check(self.L.cxk_solve_inplace(self.h, _dp(y)), "cxk_solve_inplace")
        return y

    def set_cost(self, b):
        b = np.ascontiguousarray(np.asarray(b, dtype=np.float64).ravel())
        self._check(self.L.cxk_set_cost(self.h, _dp(b)), "cxk_set_cost")

    def newton_direction(self, inv_sqrt_mu, b_scaling=1.0, c_scaling=1.0):
        self._check(self.L.cxk_newton_direction(self.h, inv_sqrt_mu, b_scaling, c_scaling),
                    "cxk_newton_direction")

    def factor_solve_async(self, cb, cq, cw):
        """Factor and solve y <- K^-1 (cb b + cq AQc + cw AW) in one upward pass (cxk_factor_solve_async)."""
        self._check(self.L.cxk_factor_solve_async(self.h, cb, cq, cw), "cxk_factor_solve_async")

    def factor_direction_async(self, inv_sqrt_mu, b_scaling=1.0, c_scaling=1.0):
        self._check(self.L.cxk_factor_direction_async(self.h, inv_sqrt_mu, b_scaling, c_scaling),
                    "cxk_factor_direction_async")

    def solve_rhs(self, cb, cq, cw):
        self._check(self.L.cxk_solve_rhs(self.h, cb, cq, cw), "cxk_solve_rhs")

    def step_scalars(self):
        out = np.zeros(6)
        self._check(self.L.cxk_step_scalars(self.h, _dp(out)), "cxk_step_scalars")
        return out

    def kkt_solve_async(self, inv_sqrt_mu, b_scaling=1.0, c_scaling=1.0):
        self._check(self.L.cxk_kkt_solve_async(self.h, inv_sqrt_mu, b_scaling, c_scaling),
                    "cxk_kkt_solve_async")

    def sync(self):
        ok = C.c_int(0)
        self._check(self.L.cxk_sync(self.h, C.byref(ok)), "cxk_sync")
        return ok.value

    def kkt_solve(self, b, inv_sqrt_mu, b_scaling=1.0, c_scaling=1.0):
        self.set_cost(b)
        self.kkt_solve_async(inv_sqrt_mu, b_scaling, c_scaling)
        ok = self.sync()
        return ok, self.get_y()

    def get_y(self):
        y = np.zeros(self.N)
        self._check(self.L.cxk_get_y(self.h, _dp(y)), "cxk_get_y")
        return y

    def set_y(self, y):
        y = np.ascontiguousarray(y, dtype=np.float64)
        self._check(self.L.cxk_set_y(self.h, _dp(y)), "cxk_set_y")

    def prepare_step(self, y, c_weight, e_weight=1.0, affine=0):
        if y is not None:
            self.set_y(y)
        info = np.zeros(2)
        self._check(self.L.cxk_prepare_step(self.h, affine, c_weight, e_weight, _dp(info)),
                    "cxk_prepare_step")
        return info

    def prepare_take_step(self, y, c_weight, e_weight=1.0):
        """PrepareStep + TakeStep with the step length min(1, 2 / norminfd^2) evaluated on the device
        (cxk_prepare_take_step) -> (normsqrd, norminfd, took)."""
        if y is not None:
            self.set_y(y)
        info = np.zeros(2)
        took = C.c_int(0)
        self._check(self.L.cxk_prepare_take_step(self.h, c_weight, e_weight, _dp(info), C.byref(took)),
                    "cxk_prepare_take_step")
        return info[0], info[1], bool(took.value)

    def take_step(self, step_size, e_weight=1.0, affine=0):
        self._check(self.L.cxk_take_step(self.h, affine, e_weight, step_size), "cxk_take_step")

    def weighted_slack_eigenvalues(self, y, c_weight):
        if y is not None:
            self.set_y(y)
        out = np.zeros(4)
        self._check(self.L.cxk_weighted_slack_eigenvalues(self.h, c_weight, _dp(out)),
                    "cxk_weighted_slack_eigenvalues")
        return out

    # ---- multi-GPU exchange / accounting
    def exchange_buffer(self):
        p = C.c_void_p()
        n = C.c_long()
        self._check(self.L.cxk_exchange_buffer(self.h, C.byref(p), C.byref(n)),
                    "cxk_exchange_buffer")
        return p.value, n.value

    def kkt_local_async(self, inv_sqrt_mu, b_scaling=1.0, c_scaling=1.0):
        self._check(self.L.cxk_kkt_local_async(self.h, inv_sqrt_mu, b_scaling, c_scaling),
                    "cxk_kkt_local_async")

    def kkt_finish_async(self, inv_sqrt_mu, b_scaling=1.0, c_scaling=1.0):
        self._check(self.L.cxk_kkt_finish_async(self.h, inv_sqrt_mu, b_scaling, c_scaling),
                    "cxk_kkt_finish_async")

    def set_solver_mode(self, mode):
        """0 / 1 supernodal LLT / LDLT (by structure), 2 dense column-pivoted QR (cxk_set_solver_mode)."""
        self._check(self.L.cxk_set_solver_mode(self.h, mode), "cxk_set_solver_mode")

    def phase_timers(self, on=True):
        self._check(self.L.cxk_phase_timers(self.h, int(bool(on))), "cxk_phase_timers")

    def phase_mark(self, phase):
        self._check(self.L.cxk_phase_mark(self.h, phase), "cxk_phase_mark")

    def phase_read(self, reset=False):
        out = np.zeros(5)
        self._check(self.L.cxk_phase_read(self.h, _dp(out), int(bool(reset))), "cxk_phase_read")
        return out

    # ---- collectives of a sharded context (cxk_comm_*)
    @staticmethod
    def comm_unique_id():
        """ncclGetUniqueId: 128 bytes made by ONE rank and handed to all (cxk_comm_unique_id)."""
        buf = C.create_string_buffer(128)
        if load_library().cxk_comm_unique_id(buf) != 0:
            raise KktError("cxk_comm_unique_id failed (librccl.so missing?)")
        return buf.raw

    def comm_init_rccl(self, unique_id, rank, world):
        """RCCL communicator for this context's device; before initialize() it also sets the shard."""
        assert len(unique_id) == 128
        self._check(self.L.cxk_comm_init_rccl(self.h, C.c_char_p(unique_id), rank, world), "cxk_comm_init_rccl")

    def comm_init_rccl_solo(self):
        """Diagnostic: one-rank RCCL communicator under a larger virtual shard world (cxk_comm_init_rccl_solo)."""
        self._check(self.L.cxk_comm_init_rccl_solo(self.h), "cxk_comm_init_rccl_solo")

    def comm_count(self):
        """Ranks the attached RCCL communicator holds (ncclCommCount); 0 without one."""
        return int(self.L.cxk_comm_count(self.h))

    def comm_selftest(self, count=1000):
        self._check(self.L.cxk_comm_selftest(self.h, count), "cxk_comm_selftest")

    def comm_set_allreduce(self, host_allreduce):
        """Another transport / tests: host_allreduce(array, op) -> array reduces HOST copies across the
        ranks (op 0 sum, 1 max, 2 min); the wrapper moves the device buffer to the host and back."""
        hip = C.CDLL("libamdhip64.so")
        fn_t = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_void_p)

        def _cb(user, dev, count, op, stream):
            try:
                buf = np.empty(count)
                if hip.hipStreamSynchronize(C.c_void_p(stream)) != 0:
                    return 1
                if hip.hipMemcpy(buf.ctypes.data_as(C.c_void_p), C.c_void_p(dev), C.c_size_t(8 * count), 2) != 0:
                    return 1
                out = np.ascontiguousarray(host_allreduce(buf, op), dtype=np.float64)
                if hip.hipMemcpy(C.c_void_p(dev), out.ctypes.data_as(C.c_void_p), C.c_size_t(8 * count), 1) != 0:
                    return 1
                return 0
            except Exception:  # an exception must not cross the C boundary
                import traceback
                traceback.print_exc()
                return 1
        self._coll_cb = fn_t(_cb)  # keeps the trampoline alive as long as the context
        self._check(self.L.cxk_comm_set_allreduce(self.h, C.cast(self._coll_cb, C.c_void_p), None),
                    "cxk_comm_set_allreduce")

    def owns(self, i):
        return bool(self.L.cxk_owns_constraint(self.h, i))

    def valid_variables(self):
        m = np.zeros(self.N, dtype=np.uint8)
        self._check(self.L.cxk_get_valid_variables(self.h, m.ctypes.data_as(C.POINTER(C.c_ubyte))),
                    "cxk_get_valid_variables")
        return m.astype(bool)

    def shard_info(self):
        cut, nlev, cnt = C.c_int(), C.c_int(), C.c_long()
        self._check(self.L.cxk_shard_info(self.h, C.byref(cut), C.byref(nlev), C.byref(cnt)),
                    "cxk_shard_info")
        return cut.value, nlev.value, cnt.value

    def exchange_download(self):
        _, count = self.exchange_buffer()
        out = np.zeros(count)
        self._check(self.L.cxk_exchange_download(self.h, _dp(out)), "cxk_exchange_download")
        return out

    def exchange_upload(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        self._check(self.L.cxk_exchange_upload(self.h, _dp(x)), "cxk_exchange_upload")

    def exchange_tensor(self, torch):
        """Zero-copy torch view of the device exchange buffer (for torch.distributed.all_reduce)."""
        ptr, count = self.exchange_buffer()

        class _Arr:
            __cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (ptr, False),
                                        "version": 2}
        return torch.as_tensor(_Arr(), device="cuda")

    def assemble_local(self):
        self._check(self.L.cxk_assemble_local(self.h), "cxk_assemble_local")

    def finish_assemble(self):
        self._check(self.L.cxk_finish_assemble(self.h), "cxk_finish_assemble")

    def dense_top_columns(self):
        """Columns factored by the dense top kernel (cxk_dense_top_columns); 0 when it is not used."""
        return self.L.cxk_dense_top_columns(self.h)

    def count_sparse_lmi(self):
        """Constraints on the sparse-LMI evaluation path (cxk_count_sparse_lmi)."""
        return self.L.cxk_count_sparse_lmi(self.h)

    def fused_assembly(self):
        """True when the assembly rides in the first factor level's launch (cxk_fused_assembly)."""
        return bool(self.L.cxk_fused_assembly(self.h))

    def fused_tree(self):
        """True when a KKT solve runs as one launch over the whole elimination tree (cxk_fused_tree)."""
        return bool(self.L.cxk_fused_tree(self.h))

    def count_lmi_kernel(self, which):
        """Constraints whose Schur block comes from kernel `which` (cxk_count_lmi_kernel): 0 literal,
        1 DPP + MFMA rows, 2 persistent MFMA, 3 GEMM pipeline, 4 sparse."""
        return self.L.cxk_count_lmi_kernel(self.h, which)

    def assembly_work(self):
        b = C.c_double()
        f = C.c_double()
        self.L.cxk_assembly_work(self.h, C.byref(b), C.byref(f))
        return b.value, f.value

    def enable_timing(self, on=True):
        self.L.cxk_enable_timing(self.h, int(on))

    def kernel_time(self, reset=True):
        ms = C.c_double()
        n = self.L.cxk_kernel_time(self.h, int(reset), C.byref(ms))
        return n, ms.value

    CLOCKS = {"assembly": 0, "tree": 1, "solve": 2, "query": 3, "prepare": 4, "take": 5}

    def kernel_clock(self, which, reset=True):
        """(samples, average ms) of a kernel clock slot (cxk_kernel_clock; names in CLOCKS)."""
        ms = C.c_double()
        n = self.L.cxk_kernel_clock(self.h, self.CLOCKS[which] if isinstance(which, str) else int(which),
                                    int(reset), C.byref(ms))
        return n, ms.value

    def set_chain_segments(self, segments):
        """0: the reference's elimination order; P >= 2: a chain-shaped tree in P segments (before initialize)."""
        self._check(self.L.cxk_set_chain_segments(self.h, int(segments)), "cxk_set_chain_segments")

    def chain_segments(self):
        return self.L.cxk_chain_segments(self.h)

    def fused_tree_timed_out(self):
        return bool(self.L.cxk_fused_tree_timed_out(self.h))

    def debug_force_fused_timeout(self):
        self._check(self.L.cxk_debug_force_fused_timeout(self.h), "cxk_debug_force_fused_timeout")


def gemm_f64(A, B, C=None, ta=False, tb=False, alpha=1.0, beta=0.0, lower_only=False, splits=1,
             reps=1, device=0):
    """Batched C = alpha op(A) op(B) + beta C through cxk_gemm_f64 (fp64 MFMA kernel).

    A: (batch, M, K) logical operand op(A); B: (batch, K, N); returns (C, avg_ms).  The packed
    column-major buffers the C-ABI expects are built here (ta / tb choose the stored layout)."""
    import ctypes as C_
    L = load_library()
    A = np.asarray(A, dtype=np.float64)
    B = np.asarray(B, dtype=np.float64)
    batch, M, K = A.shape
    N = B.shape[2]
    # stored A: M x K column-major (ta False) or K x M column-major (ta True)
    Ap = np.ascontiguousarray(A if ta else A.transpose(0, 2, 1))     # col-major(MxK) == C-order(KxM)
    Bp = np.ascontiguousarray(B.transpose(0, 2, 1) if not tb else B)  # stored K x N col-major, or N x K
    Cp = np.zeros((batch, N, M)) if C is None else np.ascontiguousarray(
        np.asarray(C, dtype=np.float64).transpose(0, 2, 1))
    ms = C_.c_double(0.0)
    L.cxk_gemm_f64.restype = C_.c_int
    L.cxk_gemm_f64.argtypes = [C_.c_int] * 7 + [C_.c_void_p] * 3 + [C_.c_double] * 2 + [C_.c_int] * 3 + [
        C_.POINTER(C_.c_double)]
    rc = L.cxk_gemm_f64(device, int(ta), int(tb), M, N, K, batch, Ap.ctypes.data, Bp.ctypes.data,
                        Cp.ctypes.data, alpha, beta, int(lower_only), splits, reps, C_.byref(ms))
    if rc != 0:
        raise RuntimeError("cxk_gemm_f64 failed")
    return Cp.transpose(0, 2, 1).copy(), ms.value
