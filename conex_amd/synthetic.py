"""Seeded synthetic cone programs of the BASELINE.json shapes.

Data only (numpy); used by tests/ and bench.py to feed the same inputs to the HIP path and
to the CPU oracle.  Values are uniform in [-1, 1] to mirror Eigen's MatrixXd::Random used by
the reference's tests (conex/test/test_util.cc:19,67-73).
"""
import numpy as np

SEED = 20201


def random_sym(rng, n):
    R = rng.uniform(-1.0, 1.0, (n, n))
    return 0.5 * (R + R.T)


def tree_cliques(K, branching=8, clique_size=20, overlap=5):
    """C4 structure: b-ary tree of cliques; child shares `overlap` of its parent's own variables."""
    fresh = clique_size - overlap
    groups = max(1, fresh // overlap)
    cliques = [list(range(clique_size))]
    for c in range(1, K):
        p = (c - 1) // branching
        start_p = 0 if p == 0 else clique_size + fresh * (p - 1)
        g = (c - 1) % groups
        shared = [start_p + overlap * g + t for t in range(overlap)]
        own = [clique_size + fresh * (c - 1) + t for t in range(fresh)]
        cliques.append(shared + own)
    num_vars = clique_size + fresh * (K - 1)
    return cliques, num_vars


def chain_cliques(K, clique_size=10, overlap=2):
    """C3 structure: clique k = {(s-o)k .. (s-o)k + s-1}."""
    step = clique_size - overlap
    cliques = [list(range(step * k, step * k + clique_size)) for k in range(K)]
    return cliques, step * (K - 1) + clique_size


def lmi_problem(K=1000, n=20, m=20, branching=8, overlap=5, seed=SEED):
    """Chordal SDP: K dense LMIs of order n over m variables each, C = I.

    Returns dict(A: (K, m, n, n), C: (K, n, n), cliques, num_vars, b).
    b is the scatter of 1/2 tr(A_ci) (GetFeasibleObjective at W = I, cone_program.cc:535-545).
    """
    rng = np.random.default_rng(seed)
    cliques, num_vars = tree_cliques(K, branching, m, overlap)
    A = rng.uniform(-1.0, 1.0, (K, m, n, n))
    A = 0.5 * (A + np.transpose(A, (0, 1, 3, 2)))
    Cm = np.broadcast_to(np.eye(n), (K, n, n)).copy()
    b = np.zeros(num_vars)
    for c in range(K):
        b[cliques[c]] += 0.5 * np.trace(A[c], axis1=1, axis2=2)
    return dict(A=A, C=Cm, cliques=cliques, num_vars=num_vars, b=b, n=n, m=m)


def sparsify(prob, density, seed=SEED + 7):
    """Zero all but about `density` of the entries of every A_i of an lmi_problem / hermitian_problem
    (symmetric pattern, the diagonal of the first plane kept with the same probability plus one
    guaranteed diagonal entry so that tr A_i != 0); b is recomputed.  Mirrors programs built entry
    by entry through CONEX_UpdateLinearOperator (hermitian_psd.cc:249-275)."""
    rng = np.random.default_rng(seed)
    A = prob["A"].copy()
    herm = A.ndim == 5
    K, m, n = A.shape[0], A.shape[1], A.shape[-1]
    for c in range(K):
        for i in range(m):
            keep = np.triu(rng.uniform(size=(n, n)) < density)
            keep[rng.integers(n), rng.integers(n)] = True
            keep = np.triu(keep) | np.triu(keep).T | np.tril(keep) | np.tril(keep).T
            d = rng.integers(n)
            keep[d, d] = True
            A[c, i] = A[c, i] * keep          # broadcasts over the planes of a Hermitian matrix
            diag = A[c, i, 0] if herm else A[c, i]
            if diag[d, d] == 0.0:
                diag[d, d] = 0.5
    out = dict(prob)
    out["A"] = A
    b = np.zeros(prob["num_vars"])
    for c in range(K):
        tr = np.trace(A[c, :, 0], axis1=1, axis2=2) if herm else np.trace(A[c], axis1=1, axis2=2)
        b[prob["cliques"][c]] += 0.5 * tr
    out["b"] = b
    return out


def maxcut_problem(n, edges_per_node=3, seed=SEED + 9):
    """Max-cut relaxation  max -sum y  s.t.  Diag(y) - Q >= 0  (Q = graph Laplacian / 4) as ONE LMI
    of order n over n variables: A_i = -e_i e_i^T (one nonzero each), C = -Q.  The Schur complement
    is a dense n x n matrix: one supernode of n columns."""
    rng = np.random.default_rng(seed)
    Q = np.zeros((n, n))
    for _ in range(edges_per_node * n):
        i, j = rng.integers(n, size=2)
        if i != j:
            Q[i, j] = Q[j, i] = rng.uniform(0.2, 1.0)
    Q = 0.25 * (np.diag(Q.sum(axis=1)) - Q)
    A = np.zeros((1, n, n, n))
    for i in range(n):
        A[0, i, i, i] = -1.0
    return dict(A=A, C=-Q[None], cliques=[list(range(n))], num_vars=n, b=-np.ones(n), n=n, m=n, Q=Q)


def scaling_points(K, n, seed=SEED + 1, scale=0.3):
    """W_c = expm(scale * sym(R)): symmetric positive definite, W != I."""
    rng = np.random.default_rng(seed)
    W = np.empty((K, n, n))
    for c in range(K):
        S = random_sym(rng, n) * scale
        lam, Q = np.linalg.eigh(S)
        W[c] = (Q * np.exp(lam)) @ Q.T
        W[c] = 0.5 * (W[c] + W[c].T)
    return W


def lp_problem(rows=20, num_vars=10, seed=SEED):
    """C1: one dense linear inequality block (test_lp.cc:19-36)."""
    rng = np.random.default_rng(seed)
    A = rng.uniform(-1, 1, (rows, num_vars))
    c = np.abs(rng.uniform(-1, 1, rows))
    x0 = np.abs(rng.uniform(-1, 1, rows))
    x0 *= 0.01 / np.linalg.norm(x0)
    return dict(A=A, c=c, b=A.T @ x0)


def soc_problem(K=5000, dim=10, m=10, overlap=2, seed=SEED, tree=0):
    """C3: K second-order cones in R^{dim+1}; chain overlap (the reference-style C3), or with
    tree > 0 a `tree`-ary clique tree (SURVEY 8d's variant with tree parallelism)."""
    rng = np.random.default_rng(seed)
    if tree:
        cliques, num_vars = tree_cliques(K, tree, m, overlap)
    else:
        cliques, num_vars = chain_cliques(K, m, overlap)
    A = rng.uniform(-1, 1, (K, dim + 1, m))
    c = np.zeros((K, dim + 1))
    c[:, 0] = 1.0
    b = np.zeros(num_vars)
    for k in range(K):
        b[cliques[k]] += A[k].T @ c[k]
    return dict(A=A, c=c, cliques=cliques, num_vars=num_vars, b=b)


def soc_scaling_points(K, dim, seed=SEED + 2):
    rng = np.random.default_rng(seed)
    W = np.zeros((K, dim + 1))
    W[:, 1:] = rng.uniform(-0.3, 0.3, (K, dim))
    W[:, 0] = np.linalg.norm(W[:, 1:], axis=1) + rng.uniform(0.5, 1.5, K)
    return W


def random_hermitian(rng, d, n):
    """d real planes of a Hermitian matrix over R / C / H: plane 0 symmetric, the others skew
    (the reference builds them as T::Random + ConjugateTranspose, hermitian_psd_test.cc:36-40)."""
    R = rng.uniform(-1.0, 1.0, (d, n, n))
    H = np.empty_like(R)
    H[0] = R[0] + R[0].T
    for p in range(1, d):
        H[p] = R[p] - R[p].T
    return H


def hermitian_problem(K=6, n=5, d=2, m=6, branching=2, overlap=2, seed=SEED):
    """K Hermitian PSD constraints of order n over R (d=1), C (d=2) or H (d=4), clique tree as in
    lmi_problem, C = identity, b = scatter of 1/2 Re tr(A_ci)."""
    rng = np.random.default_rng(seed)
    cliques, num_vars = tree_cliques(K, branching, m, overlap)
    A = np.empty((K, m, d, n, n))
    for c in range(K):
        for i in range(m):
            A[c, i] = random_hermitian(rng, d, n)
    Cm = np.zeros((K, d, n, n))
    Cm[:, 0] = np.eye(n)
    b = np.zeros(num_vars)
    for c in range(K):
        b[cliques[c]] += 0.5 * np.trace(A[c, :, 0], axis1=1, axis2=2)
    return dict(A=A, C=Cm, cliques=cliques, num_vars=num_vars, b=b, n=n, m=m, d=d)


def hermitian_scaling_points(K, n, d, seed=SEED + 3, scale=0.2):
    """Hermitian positive definite W = (I + s H)^2 as d real planes."""
    rng = np.random.default_rng(seed)
    W = np.zeros((K, d, n, n))
    for c in range(K):
        H = random_hermitian(rng, d, n) * (scale / 2)
        H[0] += np.eye(n)
        W[c] = hc_multiply(H, H)
        W[c, 0] = 0.5 * (W[c, 0] + W[c, 0].T)
        for p in range(1, d):
            W[c, p] = 0.5 * (W[c, p] - W[c, p].T)
    return W


# sign table of the reference's hyper-complex product (jordan_matrix_algebra.cc:103-124; the algebra
# of dimension d uses the top-left d x d corner): plane (i ^ j) receives  sign[i][j] * X_i Y_j
HC_SIGN = np.array([[1, 1, 1, 1, 1, 1, 1, 1], [1, -1, -1, 1, -1, 1, 1, -1], [1, 1, -1, -1, -1, -1, 1, 1],
                    [1, -1, 1, -1, -1, 1, -1, 1], [1, 1, 1, 1, -1, -1, -1, -1], [1, -1, 1, -1, 1, -1, 1, -1],
                    [1, -1, -1, 1, 1, -1, -1, 1], [1, 1, -1, -1, 1, 1, -1, -1]])


def hc_multiply(X, Y):
    d = X.shape[0]
    Z = np.zeros((d, X.shape[1], Y.shape[2]))
    for i in range(d):
        for j in range(d):
            Z[i ^ j] += HC_SIGN[i, j] * (X[i] @ Y[j])
    return Z


def mixed_problem(K=4600, herm_every=(8, 23), branching=8, overlap=4, seed=SEED):
    """C5 shape: a clique tree mixing complex Hermitian PSD cones of order 12 over 24
    variables with second-order cones of dimension 10 over 10 variables; node c is Hermitian when
    c % herm_every[1] < herm_every[0] (about 1600 of 4600), every child shares `overlap` of its
    parent's own variables.  Returns dict(kinds, A, C, cliques, num_vars, b)."""
    rng = np.random.default_rng(seed)
    kinds = ["herm" if c % herm_every[1] < herm_every[0] else "soc" for c in range(K)]
    size = [24 if k == "herm" else 10 for k in kinds]
    start = [0] * K                       # first own (fresh) variable of node c
    cliques = [list(range(size[0]))]
    nxt = size[0]
    for c in range(1, K):
        p = (c - 1) // branching
        fresh_p = size[p] - (overlap if p > 0 else 0)
        own_p = start[p] if p > 0 else 0
        groups = max(1, fresh_p // overlap)
        g = (c - 1) % groups
        shared = [own_p + overlap * g + t for t in range(overlap)]
        fresh = size[c] - overlap
        start[c] = nxt
        cliques.append(shared + list(range(nxt, nxt + fresh)))
        nxt += fresh
    A, Cm = [], []
    b = np.zeros(nxt)
    for c in range(K):
        if kinds[c] == "herm":
            a = np.stack([random_hermitian(rng, 2, 12) for _ in range(24)])
            cm = np.zeros((2, 12, 12))
            cm[0] = np.eye(12)
            b[cliques[c]] += 0.5 * np.trace(a[:, 0], axis1=1, axis2=2)
        else:
            a = rng.uniform(-1, 1, (11, 10))
            cm = np.zeros(11)
            cm[0] = 1.0
            b[cliques[c]] += a.T @ cm
        A.append(a)
        Cm.append(cm)
    return dict(kinds=kinds, A=A, C=Cm, cliques=cliques, num_vars=nxt, b=b)


def mixed_scaling_points(prob, seed=SEED + 5):
    rng = np.random.default_rng(seed)
    W = []
    for k in prob["kinds"]:
        if k == "herm":
            H = random_hermitian(rng, 2, 12) * 0.1
            H[0] += np.eye(12)
            w = hc_multiply(H, H)
            w[0] = 0.5 * (w[0] + w[0].T)
            w[1] = 0.5 * (w[1] - w[1].T)
        else:
            w = np.zeros(11)
            w[1:] = rng.uniform(-0.3, 0.3, 10)
            w[0] = np.linalg.norm(w[1:]) + rng.uniform(0.5, 1.5)
        W.append(w)
    return W


def build(ctx_cls, prob, kind="lmi", **kw):
    """Instantiate `ctx_cls(num_vars, **kw)` (oracle Program or KktContext) from a problem dict."""
    if kind == "lmi":
        p = ctx_cls(prob["num_vars"], **kw)
        for c, cl in enumerate(prob["cliques"]):
            assert p.add_lmi(prob["A"][c], prob["C"][c], cl) == c
    elif kind == "herm":
        p = ctx_cls(prob["num_vars"], **kw)
        for c, cl in enumerate(prob["cliques"]):
            assert p.add_hermitian(prob["A"][c], prob["C"][c], cl) == c
    elif kind == "mixed":
        p = ctx_cls(prob["num_vars"], **kw)
        for c, cl in enumerate(prob["cliques"]):
            add = p.add_hermitian if prob["kinds"][c] == "herm" else p.add_soc
            assert add(prob["A"][c], prob["C"][c], cl) == c
    elif kind == "soc":
        p = ctx_cls(prob["num_vars"], **kw)
        for c, cl in enumerate(prob["cliques"]):
            assert p.add_soc(prob["A"][c], prob["c"][c], cl) == c
    elif kind == "lp":
        p = ctx_cls(prob["A"].shape[1], **kw)
        assert p.add_linear(prob["A"], prob["c"]) == 0
    else:
        raise ValueError(kind)
    p.initialize()
    return p
