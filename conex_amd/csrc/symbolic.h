// Host-side symbolic analysis for the supernodal KKT solver.
//
// Produces exactly the integer structures the reference computes in
//   conex/kkt_solver.cc:70-131      (GetRootNode, is_empty, RelabelCliques)
//   conex/supernodal_solver.cc:376-431 (GetData, SupernodesToData)
//   conex/clique_ordering.cc:111-333   (PickCliqueOrder, FillIn)
//   conex/triangular_matrix_workspace.cc:37-159 (block layout, S_S, intersections)
// but with an inverted variable->clique index instead of the reference's
// K x K table of intersection vectors, and slab offsets instead of double*.
// Equality of every output with the literal restatement in oracle/ is
// asserted by tests/test_symbolic_parity.py.
#pragma once
#include <cstdint>
#include <vector>

namespace cxk {

using IntList = std::vector<int>;
using IntLists = std::vector<IntList>;

struct RootedTree {
  std::vector<int> parent;
  std::vector<int> height;
};

// clique_ordering.cc:307-333. `valid_leaf` may be empty. Outputs are indexed by the
// ORIGINAL clique index (as in the reference).
void PickCliqueOrder(const IntLists& cliques_sorted, const std::vector<int>& valid_leaf, int root,
                     std::vector<int>* order, IntLists* supernodes, IntLists* separators,
                     RootedTree* tree);

// tree_utils.cc:11-25
IntList PathInTree(int x, int y, const std::vector<int>& parent, const std::vector<int>& depth);

// MatrixData (supernodal_solver.h:18-29) after RelabelCliques (kkt_solver.cc:47-68).
struct MatrixData {
  int K = 0;
  int N = 0;
  int num_vars = 0;
  std::vector<int> clique_order;      // elimination position -> original clique id
  IntLists cliques;                   // permuted labels: supernode then separators
  IntLists supernodes_orig;           // original labels
  IntLists separators_orig;           // original labels, ordered by permuted label
  IntLists supernodes_pos;            // position inside owning constraint, -1 = fill-in
  IntLists separators_pos;
  std::vector<int> supernode_size;
  std::vector<int> permutation;       // original var -> eliminated position
  std::vector<int> permutation_inverse;
};

int GetRootNode(const IntLists& cliques, const IntLists& dual_vars);
MatrixData Analyze(const IntLists& cliques, const IntLists& dual_vars);

// A chain-shaped elimination tree (every step updates the next one only: BASELINE config 3 as the
// reference's own tests arrange it, 5000 second-order cones, 10 000 strictly dependent block steps)
// has no parallelism in the reference's order.  The order is part of the parity contract for what the
// library REPORTS (order / supernodes / separators / permutation stay the reference's); the
// factorization itself may run in any order that gives the same solution.  SegmentChain cuts the
// chain into `segments` pieces: the variables that carry an update across a cut (the separator of the
// step before it) are DEFERRED -- eliminated after both neighbouring pieces, together with the last
// step of the piece in front of the cut -- so that the pieces become independent subtrees of equal
// depth (each step then carries its piece's deferred variables as extra separator rows: structural
// fill, the "spikes" of a partitioned tridiagonal solver), and the deferred sets themselves merge in
// nested-dissection order, log2(segments) levels deep: K / segments + log2(segments) dependent levels
// instead of K, every one of them swept for all pieces at once by the ordinary level kernels.  Same
// matrix, another elimination order: the Newton direction is the same to rounding (tests: <= 1e-10
// against the oracle, which eliminates in the reference's order).
// Returns false (out untouched) when the structure is not a plain chain.
bool IsChain(const MatrixData& ref);
bool SegmentChain(const MatrixData& ref, const IntLists& cliques, const IntLists& dual_vars, int segments,
                  MatrixData* out);

// Block layout of the supernodal slab (triangular_matrix_workspace.cc).
struct Layout {
  int K = 0;
  int N = 0;
  std::vector<int> supernode_size;
  std::vector<int> supernode_start;     // first permuted label of each supernode
  IntLists separators;                  // permuted labels
  std::vector<int64_t> diag_off;        // n_s x n_s, col-major
  std::vector<int64_t> offd_off;        // n_s x s,  col-major
  int64_t slab_size = 0;
  std::vector<int> var_to_sn;
  std::vector<int> var_to_pos;
  std::vector<std::vector<int64_t>> ss_index;  // seperator_diagonal as slab offsets
};

Layout BuildLayout(const MatrixData& md);
int64_t LookupAddress(const Layout& L, int r, int c);

}  // namespace cxk
