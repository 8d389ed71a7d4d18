// SeDuMi-format front end (include/conex_sedumi.h): the reference's MATLAB pipeline
// interfaces/matlab/conex.m + util/{CleanLinear, coneBase, ConexPreprocess, blkdiagPrg, BuildMask,
// BinaryPsdCompletion, ExtractConstraintMatrices}.m on the host, in front of the CONEX_* C-ABI.
// Index conventions: everything here is 0-based; a PSD block of order n occupies n^2 consecutive
// columns, column-major (entry (i, j) at offset + i + j n), as coneBase.m:86-94 lays them out.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <numeric>
#include <vector>

#include "../../include/conex.h"
#include "../../include/conex_sedumi.h"

namespace {

struct Entry {
  long col;
  double val;
};
using Row = std::vector<Entry>;  // sorted by column, no explicit zeros

struct Problem {
  long N = 0;
  std::vector<Row> rows;
  std::vector<double> b, c;
  std::vector<long> Ks;
};

struct Block {
  int order = 0;
  std::vector<long> variables;  // indices into the solver's variables (kept rows), ascending
  std::vector<double> matrices; // order x order x variables.size(), column-major
  std::vector<double> affine;   // order x order
};

struct Prepared {
  long m = 0, N = 0;
  std::vector<long> kept_rows;     // input rows behind the solver's variables (both CleanLinear passes)
  std::vector<long> kept_cols;     // blkdiagPrg.indx: input columns, in the reduced order
  std::vector<double> b;           // reduced cost
  std::vector<double> c_sym;       // symmetrized c, input numbering
  std::vector<Block> blocks;
  bool dense_single = false;       // conex.m's else-branch: one dense LMI over all kept rows
};

bool Fail(const char* msg) {
  fprintf(stderr, "conex_sedumi: %s\n", msg);
  return false;
}

// sparse(i, j, v): duplicates add up, zeros vanish
bool BuildRows(long m, long N, long nnz, const long* ri, const long* ci, const double* v, std::vector<Row>* out) {
  std::vector<std::map<long, double>> acc((size_t)m);
  for (long k = 0; k < nnz; k++) {
    if (ri[k] < 0 || ri[k] >= m || ci[k] < 0 || ci[k] >= N) return Fail("a triplet of A is out of range");
    acc[(size_t)ri[k]][ci[k]] += v[k];
  }
  out->assign((size_t)m, Row());
  for (long r = 0; r < m; r++)
    for (auto& kv : acc[(size_t)r])
      if (kv.second != 0.0) (*out)[(size_t)r].push_back(Entry{kv.first, kv.second});
  return true;
}

// CleanLinear.m:19-29 (useQR = 0): the rows of [A, b] that are not all zero
std::vector<long> CleanLinear(const std::vector<Row>& rows, const std::vector<double>& b) {
  std::vector<long> keep;
  for (size_t r = 0; r < rows.size(); r++)
    if (!rows[r].empty() || b[r] != 0.0) keep.push_back((long)r);
  return keep;
}

// coneBase.m:180-190: (i, j) and (j, i) of every PSD block both become their average
void SymmetrizeRow(Row* row, const std::vector<long>& Ks) {
  std::map<long, double> v;
  for (const Entry& e : *row) v[e.col] = e.val;
  std::map<long, double> out;
  long off = 0;
  size_t bi = 0;
  for (const Entry& e : *row) {
    while (bi < Ks.size() && e.col >= off + Ks[bi] * Ks[bi]) off += Ks[bi] * Ks[bi], bi++;
    if (bi >= Ks.size()) {
      out[e.col] = e.val;
      continue;
    }
    const long n = Ks[bi], p = e.col - off, i = p % n, j = p / n;
    const long mate = off + j + i * n;
    auto it = v.find(mate);
    const double avg = (e.val + (it == v.end() ? 0.0 : it->second)) / 2;
    out[e.col] = avg;
    out[mate] = avg;
  }
  row->clear();
  for (auto& kv : out)
    if (kv.second != 0.0) row->push_back(Entry{kv.first, kv.second});
}

void SymmetrizeDense(std::vector<double>* c, const std::vector<long>& Ks) {
  long off = 0;
  for (long n : Ks) {
    for (long j = 0; j < n; j++)
      for (long i = j + 1; i < n; i++) {
        const double avg = ((*c)[off + i + j * n] + (*c)[off + j + i * n]) / 2;
        (*c)[off + i + j * n] = (*c)[off + j + i * n] = avg;
      }
    off += n * n;
  }
}

long Count(const std::vector<char>& M) { return (long)std::count(M.begin(), M.end(), (char)1); }

// BuildMask.m:64-85
void SubspaceClosure(std::vector<char>* M, const std::vector<Row>& rows, const std::vector<double>& b) {
  for (size_t r = 0; r < rows.size(); r++)
    if (b[r] != 0.0)
      for (const Entry& e : rows[r]) (*M)[(size_t)e.col] = 1;  // the stuff we must pass through
  long nnz = Count(*M);
  for (;;) {
    std::vector<char> next(M->size(), 0);
    for (const Row& row : rows) {
      bool tau = false;  // a row at least partially passed through ...
      for (const Entry& e : row) tau = tau || (*M)[(size_t)e.col];
      if (tau)
        for (const Entry& e : row) next[(size_t)e.col] = 1;  // ... is passed through whole
    }
    *M = next;
    const long now = Count(*M);
    if (now == nnz) break;
    nnz = now;
  }
}

// BinaryPsdCompletion.m:1-17 with its conncomp (:20-62): the connected components of the support of
// one PSD block, smallest first (MATLAB's sort is stable: ties keep their order of discovery), every
// component completed to a full diagonal block of the mask.
std::vector<std::vector<long>> BinaryPsdCompletion(char* M, long n) {
  std::vector<long> r;
  for (long i = 0; i < n; i++) {
    bool any = false;
    for (long j = 0; j < n && !any; j++) any = M[i + j * n];
    if (any) r.push_back(i);
  }
  std::vector<std::vector<long>> members;
  if (r.empty()) return members;
  const long R = (long)r.size();
  auto adj = [&](long a, long b2) { return a != b2 && (M[r[a] + r[b2] * n] || M[r[b2] + r[a] * n]); };
  std::vector<char> seen((size_t)R, 0);
  for (long s = 0; s < R; s++) {
    if (seen[(size_t)s]) continue;
    members.emplace_back(1, s);
    seen[(size_t)s] = 1;
    for (size_t ptr = 0; ptr < members.back().size(); ptr++) {
      const long cur = members.back()[ptr];
      for (long q = 0; q < R; q++)
        if (!seen[(size_t)q] && adj(q, cur)) {
          seen[(size_t)q] = 1;
          members.back().push_back(q);
        }
    }
  }
  std::stable_sort(members.begin(), members.end(),
                   [](const std::vector<long>& a, const std::vector<long>& b2) { return a.size() < b2.size(); });
  for (auto& comp : members) {
    for (long& v : comp) v = r[(size_t)v];
    for (long a : comp)
      for (long b2 : comp) M[a + b2 * n] = 1;
  }
  return members;
}

// ExtractConstraintMatrices.m:1-48 on rows already restricted / renumbered to `cols` columns
void Extract(const std::vector<Row>& rows, const std::vector<double>& c, const std::vector<long>& Ks,
             std::vector<Block>* out) {
  out->clear();
  long off = 0;
  for (long n : Ks) {
    Block B;
    B.order = (int)n;
    B.affine.assign(c.begin() + off, c.begin() + off + n * n);
    std::vector<std::vector<double>> mats;
    for (size_t r = 0; r < rows.size(); r++) {
      std::vector<double> mat;
      for (const Entry& e : rows[r])
        if (e.col >= off && e.col < off + n * n) {
          if (mat.empty()) mat.assign((size_t)(n * n), 0.0);
          mat[(size_t)(e.col - off)] = e.val;
        }
      if (!mat.empty()) {
        B.variables.push_back((long)r);
        B.matrices.insert(B.matrices.end(), mat.begin(), mat.end());
      }
    }
    out->push_back(std::move(B));
    off += n * n;
  }
}

bool Prepare(long m, long N, long nnz, const long* ri, const long* ci, const double* v, const double* b,
             const double* c, int num_psd, const long* Ks_in, int blkdiag, Prepared* P) {
  if (m < 1 || N < 1 || nnz < 0 || !b || !c || num_psd < 1 || !Ks_in) return Fail("invalid arguments");
  std::vector<long> Ks(Ks_in, Ks_in + num_psd);
  long total = 0;
  for (long n : Ks) {
    if (n < 1) return Fail("K.s holds a non-positive order");
    total += n * n;
  }
  if (total != N) return Fail("A has a column count other than sum(K.s.^2): only K.s is supported (conex.m:7-15)");
  std::vector<Row> rows;
  if (!BuildRows(m, N, nnz, ri, ci, v, &rows)) return false;
  std::vector<double> bv(b, b + m), cv(c, c + N);
  // conex.m:3-6
  std::vector<long> keep1 = CleanLinear(rows, bv);
  {
    std::vector<Row> r2;
    std::vector<double> b2;
    for (long r : keep1) {
      r2.push_back(rows[(size_t)r]);
      b2.push_back(bv[(size_t)r]);
    }
    rows.swap(r2);
    bv.swap(b2);
  }
  for (Row& row : rows) SymmetrizeRow(&row, Ks);
  SymmetrizeDense(&cv, Ks);
  P->m = m;
  P->N = N;
  P->c_sym = cv;
  if (blkdiag < 0) blkdiag = num_psd > 1;  // conex.m:20
  if (!blkdiag) {
    P->kept_rows = keep1;
    P->kept_cols.resize((size_t)N);
    std::iota(P->kept_cols.begin(), P->kept_cols.end(), 0L);
    P->b = bv;
    Extract(rows, cv, Ks, &P->blocks);
    // conex.m:46-50: one block, no preprocessing: a dense LMI over every kept row
    P->dense_single = num_psd == 1;
    if (P->dense_single) {
      Block& B = P->blocks[0];
      const long n = Ks[0];
      B.variables.resize(rows.size());
      std::iota(B.variables.begin(), B.variables.end(), 0L);
      B.matrices.assign((size_t)(n * n) * rows.size(), 0.0);
      for (size_t r = 0; r < rows.size(); r++)
        for (const Entry& e : rows[r]) B.matrices[r * (size_t)(n * n) + (size_t)e.col] = e.val;
    }
    for (const Block& B : P->blocks)
      if (B.variables.empty()) return Fail("a PSD block carries no variable");
    return true;
  }
  // ---- BuildMask.m:1-61
  std::vector<char> M((size_t)N, 0);
  for (long j = 0; j < N; j++) M[(size_t)j] = cv[(size_t)j] != 0.0;
  long nnzM = Count(M);
  std::vector<std::vector<std::vector<long>>> cliques(Ks.size());
  for (;;) {
    SubspaceClosure(&M, rows, bv);
    long off = 0;
    for (size_t i = 0; i < Ks.size(); i++) {
      cliques[i] = BinaryPsdCompletion(M.data() + off, Ks[i]);
      off += Ks[i] * Ks[i];
    }
    const long now = Count(M);
    if (now == nnzM) break;
    nnzM = now;
  }
  std::vector<long> indx, Kr;
  {
    long off = 0;
    for (size_t i = 0; i < Ks.size(); i++) {
      const long n = Ks[i];
      for (auto clique : cliques[i]) {
        std::sort(clique.begin(), clique.end());  // coneBase.SubMatToIndx (:254-266): find() of the masked block
        for (long col : clique)
          for (long row : clique) indx.push_back(off + row + col * n);
        Kr.push_back((long)clique.size());
      }
      off += n * n;
    }
  }
  if (Kr.empty()) return Fail("the support of the problem is empty");
  std::vector<long> newcol((size_t)N, -1);
  for (size_t k = 0; k < indx.size(); k++) newcol[(size_t)indx[k]] = (long)k;
  std::vector<Row> rr(rows.size());
  for (size_t r = 0; r < rows.size(); r++) {
    for (const Entry& e : rows[r])
      if (newcol[(size_t)e.col] >= 0) rr[r].push_back(Entry{newcol[(size_t)e.col], e.val});
    std::sort(rr[r].begin(), rr[r].end(), [](const Entry& a, const Entry& b2) { return a.col < b2.col; });
  }
  std::vector<double> cr(indx.size());
  for (size_t k = 0; k < indx.size(); k++) cr[k] = cv[(size_t)indx[k]];
  // blkdiagPrg.m:29
  const std::vector<long> keep2 = CleanLinear(rr, bv);
  std::vector<Row> r3;
  for (long r : keep2) {
    r3.push_back(rr[(size_t)r]);
    P->b.push_back(bv[(size_t)r]);
    P->kept_rows.push_back(keep1[(size_t)r]);
  }
  P->kept_cols = indx;
  Extract(r3, cr, Kr, &P->blocks);
  for (const Block& B : P->blocks)
    if (B.variables.empty()) return Fail("a block of the reduced problem carries no variable");
  return true;
}

}  // namespace

extern "C" {

void* CONEX_SedumiPreprocess(long m, long N, long nnz, const long* A_row, const long* A_col, const double* A_val,
                             const double* b, const double* c, int num_psd, const long* Ks, int blkdiag) {
  Prepared* P = new Prepared();
  if (!Prepare(m, N, nnz, A_row, A_col, A_val, b, c, num_psd, Ks, blkdiag, P)) {
    delete P;
    return nullptr;
  }
  return P;
}
void CONEX_SedumiFree(void* handle) { delete static_cast<Prepared*>(handle); }
int CONEX_SedumiNumBlocks(const void* handle) { return handle ? (int)static_cast<const Prepared*>(handle)->blocks.size() : 0; }
long CONEX_SedumiKeptRows(const void* handle, long* rows) {
  if (!handle) return 0;
  const auto& v = static_cast<const Prepared*>(handle)->kept_rows;
  if (rows) std::copy(v.begin(), v.end(), rows);
  return (long)v.size();
}
long CONEX_SedumiKeptColumns(const void* handle, long* cols) {
  if (!handle) return 0;
  const auto& v = static_cast<const Prepared*>(handle)->kept_cols;
  if (cols) std::copy(v.begin(), v.end(), cols);
  return (long)v.size();
}
long CONEX_SedumiReducedB(const void* handle, double* b) {
  if (!handle) return 0;
  const auto& v = static_cast<const Prepared*>(handle)->b;
  if (b) std::copy(v.begin(), v.end(), b);
  return (long)v.size();
}
int CONEX_SedumiBlockOrder(const void* handle, int block) {
  const Prepared* P = static_cast<const Prepared*>(handle);
  return P && block >= 0 && block < (int)P->blocks.size() ? P->blocks[(size_t)block].order : -1;
}
int CONEX_SedumiBlockNumVariables(const void* handle, int block) {
  const Prepared* P = static_cast<const Prepared*>(handle);
  return P && block >= 0 && block < (int)P->blocks.size() ? (int)P->blocks[(size_t)block].variables.size() : -1;
}
int CONEX_SedumiBlockData(const void* handle, int block, long* variables, double* matrices, double* affine) {
  const Prepared* P = static_cast<const Prepared*>(handle);
  if (!P || block < 0 || block >= (int)P->blocks.size()) return CONEX_FAILURE;
  const Block& B = P->blocks[(size_t)block];
  if (variables) std::copy(B.variables.begin(), B.variables.end(), variables);
  if (matrices) std::copy(B.matrices.begin(), B.matrices.end(), matrices);
  if (affine) std::copy(B.affine.begin(), B.affine.end(), affine);
  return CONEX_SUCCESS;
}

int CONEX_SolveSedumi(long m, long N, long nnz, const long* A_row, const long* A_col, const double* A_val,
                      const double* b, const double* c, int num_psd, const long* Ks,
                      const CONEX_SedumiOptions* opt, double* x, double* y, CONEX_SedumiInfo* info) {
  if (!x || !y) return CONEX_FAILURE;
  Prepared P;
  if (!Prepare(m, N, nnz, A_row, A_col, A_val, b, c, num_psd, Ks, opt ? opt->blkdiag : -1, &P)) return CONEX_FAILURE;
  const int nvars = (int)P.kept_rows.size();
  if (nvars < 1) {
    Fail("no row of A is left");
    return CONEX_FAILURE;
  }
  void* prog = CONEX_CreateConeProgram();
  if (!prog) return CONEX_FAILURE;
  int rc = CONEX_SetNumberOfVariables(prog, nvars);
  std::vector<int> ids;
  for (const Block& B : P.blocks) {
    if (rc != CONEX_SUCCESS) break;
    const int n = B.order, nv = (int)B.variables.size();
    int id;
    if (P.dense_single)  // conex.m:46-50
      id = CONEX_AddDenseLMIConstraint(prog, B.matrices.data(), n, n, nv, B.affine.data(), n, n);
    else                 // conex.m:38-45
      id = CONEX_AddSparseLMIConstraint(prog, B.matrices.data(), n, n, nv, B.affine.data(), n, n, B.variables.data(), nv);
    if (id < 0) rc = CONEX_FAILURE;
    ids.push_back(id);
  }
  if (rc != CONEX_SUCCESS) {
    CONEX_DeleteConeProgram(prog);
    return CONEX_FAILURE;
  }
  // conex.m:52-59 (`max_iteration`, sic, is taken as max_iterations)
  CONEX_SolverConfiguration cfg;
  CONEX_SetDefaultOptions(&cfg);
  cfg.prepare_dual_variables = 1;
  cfg.inv_sqrt_mu_max = 1000;
  cfg.infeasibility_threshold = 1e3;
  cfg.max_iterations = 25;
  cfg.divergence_upper_bound = 1;
  cfg.final_centering_steps = 5;
  std::vector<double> yr((size_t)nvars, 0.0);
  const auto t0 = std::chrono::steady_clock::now();
  const int solved = CONEX_Maximize(prog, P.b.data(), nvars, &cfg, yr.data(), nvars);
  const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  // ConexPostProcess (ConexPreprocess.m:35-56) / conex.m:67-76: x block by block into the kept
  // columns, y back through both CleanLinear maps
  std::fill(x, x + N, 0.0);
  std::fill(y, y + m, 0.0);
  size_t at = 0;
  for (size_t i = 0; i < P.blocks.size(); i++) {
    const int n = P.blocks[i].order;
    std::vector<double> X((size_t)n * n, 0.0);
    (void)CONEX_GetDualVariable(prog, ids[i], X.data(), n, n);
    for (size_t k = 0; k < X.size(); k++) x[P.kept_cols[at + k]] = X[k];
    at += X.size();
  }
  for (int k = 0; k < nvars; k++) y[P.kept_rows[(size_t)k]] = yr[(size_t)k];
  CONEX_DeleteConeProgram(prog);
  if (info) {
    memset(info, 0, sizeof(*info));
    info->solved = solved;
    info->pinf = info->dinf = !solved;
    info->cpusec = secs;
    info->num_blocks = (int)P.blocks.size();
    info->num_rows_kept = nvars;
    if (opt && opt->errors) {
      double cx = 0, by = 0;
      for (long j = 0; j < N; j++) cx += P.c_sym[(size_t)j] * x[j];
      for (long r = 0; r < m; r++) by += b[r] * y[r];
      info->errors[0] = std::fabs(cx - by);
      info->errors[1] = cx - by;
    }
  }
  return CONEX_SUCCESS;
}

}  // extern "C"
