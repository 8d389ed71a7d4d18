// extern "C" door onto the one reference translation unit that builds without
// Eigen (conex/tree_utils.cc).  Compiled only into oracle/_ref (never shipped,
// never linked by the product); used by tests to cross-check the oracle's
// PathInTree restatement against the real reference code.
#include <vector>

#include "conex/tree_utils.h"

extern "C" int ref_path_in_tree(int x, int y, int n, const int* parent, const int* depth,
                                int* out) {
  std::vector<int> p(parent, parent + n), d(depth, depth + n);
  std::vector<int> path = conex::PathInTree(x, y, p, d);
  for (size_t i = 0; i < path.size(); i++) out[i] = path[i];
  return static_cast<int>(path.size());
}
