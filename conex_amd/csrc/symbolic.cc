// Host-side symbolic analysis; see symbolic.h for the reference map.
#include "symbolic.h"

#include <algorithm>
#include <numeric>
#include <stdexcept>

namespace cxk {

namespace {

int GetMax(const IntLists& cliques) {
  int mx = cliques.at(0).at(0);
  for (const auto& c : cliques)
    for (int v : c) mx = std::max(mx, v);
  return mx;
}

IntList Intersect(const IntList& a, const IntList& b) {
  IntList out;
  std::set_intersection(a.begin(), a.end(), b.begin(), b.end(), std::back_inserter(out));
  return out;
}

// Greedy max-intersection DFS over the clique graph (clique_ordering.cc:111-201).
// Tie rules kept verbatim: weight >= running max (starting at 1) collects *all*
// arg-max unvisited neighbours in increasing index; a clique that cannot be a
// leaf stops the fan-out early; discovery order is reversed at the end.
int OrderHelper(const IntLists& cliques, const std::vector<int>& valid_leaf, int root_in,
                const IntLists& var_to_cliques, IntLists* separators, std::vector<int>* order,
                RootedTree* tree) {
  const int n = static_cast<int>(cliques.size());
  const bool leaf_rule = !valid_leaf.empty();
  std::vector<char> visited(n, 0);
  std::vector<int> count(n, 0);
  std::vector<int> touched;
  std::vector<int> stack;
  std::vector<int> argmax;
  std::vector<int> degree_weight(n, 0);

  stack.push_back(root_in < 0 ? 0 : root_in);
  order->clear();
  order->reserve(n);

  while (static_cast<int>(order->size()) < n) {
    const int active = stack.back();
    if (!visited[active]) {
      order->push_back(active);
      visited[active] = 1;
      tree->parent[active] = active;
      tree->height[active] = 0;
    }

    touched.clear();
    for (int v : cliques[active]) {
      for (int i : var_to_cliques[v]) {
        if (i == active) continue;
        if (count[i]++ == 0) touched.push_back(i);
      }
    }

    size_t max_weight = 1;
    argmax.clear();
    auto consider = [&](int i) {
      if (i == active || visited[i]) return;
      size_t w = static_cast<size_t>(count[i]);
      if (leaf_rule) {
        if (!valid_leaf[i]) w += 10000;
        if (!valid_leaf[active]) w += 10000;
      }
      if (w >= max_weight) {
        if (w > max_weight) {
          argmax.clear();
          max_weight = w;
        }
        argmax.push_back(i);
      }
    };
    if (leaf_rule) {
      for (int i = 0; i < n; i++) consider(i);
    } else {
      std::sort(touched.begin(), touched.end());
      for (int i : touched) consider(i);
    }

    for (int e : argmax) {
      (*separators)[e] = Intersect(cliques[active], cliques[e]);
      stack.push_back(e);
      order->push_back(e);
      visited[e] = 1;
      const int sz = static_cast<int>((*separators)[e].size());
      degree_weight[active] += sz;
      degree_weight[e] += sz;
      tree->parent[e] = active;
      tree->height[e] = tree->height[active] + 1;
      if (leaf_rule && !valid_leaf[e]) break;
    }
    for (int i : touched) count[i] = 0;

    if (argmax.empty()) {
      stack.pop_back();
      if (stack.empty()) {
        int node = -1;
        for (int i = 0; i < n; i++)
          if (!visited[i]) {
            node = i;
            break;
          }
        if (node == -1) break;
        stack.push_back(node);
      }
    }
  }

  int root_node = 0;
  for (int i = 1; i < n; i++)
    if (degree_weight[i] > degree_weight[root_node]) root_node = i;
  std::reverse(order->begin(), order->end());
  return root_node;
}

// Running-intersection repair (clique_ordering.cc:261-305).
void FillIn(const RootedTree& tree, int num_variables, const std::vector<int>& order,
            IntLists* supernodes, IntLists* separators) {
  const int num_cliques = static_cast<int>(order.size());
  std::vector<int> eliminated(num_variables, num_cliques + 1);
  for (int i = 0; i < num_cliques; i++) {
    for (int v : (*supernodes)[order[i]]) {
      if (eliminated[v] < num_cliques) {
        IntList path = PathInTree(order[i], eliminated[v], tree.parent, tree.height);
        for (size_t j = 0; j + 1 < path.size(); j++) {
          IntList& sep = (*separators)[path[j]];
          auto it = std::lower_bound(sep.begin(), sep.end(), v);
          if (it == sep.end() || *it != v) sep.insert(it, v);
        }
        eliminated[v] = path.back();
      } else {
        eliminated[v] = order[i];
      }
    }
  }
  supernodes->assign(num_cliques, IntList());
  for (int v = 0; v < num_variables; v++)
    if (eliminated[v] < num_cliques) (*supernodes)[eliminated[v]].push_back(v);
  for (auto& s : *separators) std::sort(s.begin(), s.end());
  for (auto& s : *supernodes) std::sort(s.begin(), s.end());
}

}  // namespace

IntList PathInTree(int x, int y, const std::vector<int>& parent, const std::vector<int>& depth) {
  IntList path;
  while (x != y) {
    if (depth[x] < depth[y]) {
      path.push_back(y);
      y = parent.at(y);
    } else {
      path.push_back(x);
      x = parent.at(x);
    }
  }
  path.push_back(x);
  return path;
}

void PickCliqueOrder(const IntLists& cliques_sorted, const std::vector<int>& valid_leaf, int root,
                     std::vector<int>* order, IntLists* supernodes, IntLists* separators,
                     RootedTree* tree) {
  const int n = static_cast<int>(cliques_sorted.size());
  const int num_vars = GetMax(cliques_sorted) + 1;
  IntLists var_to_cliques(num_vars);
  for (int i = 0; i < n; i++)
    for (int v : cliques_sorted[i]) var_to_cliques[v].push_back(i);

  tree->parent.assign(n, 0);
  tree->height.assign(n, 0);
  separators->assign(n, IntList());
  int better_root =
      OrderHelper(cliques_sorted, valid_leaf, root, var_to_cliques, separators, order, tree);
  if (root == -1) {
    separators->assign(n, IntList());
    tree->parent.assign(n, 0);
    tree->height.assign(n, 0);
    OrderHelper(cliques_sorted, valid_leaf, better_root, var_to_cliques, separators, order, tree);
  }
  supernodes->assign(n, IntList());
  for (int e : *order) {
    std::set_difference(cliques_sorted[e].begin(), cliques_sorted[e].end(),
                        (*separators)[e].begin(), (*separators)[e].end(),
                        std::back_inserter((*supernodes)[e]));
  }
  FillIn(*tree, num_vars, *order, supernodes, separators);
}

int GetRootNode(const IntLists& vars, const IntLists& dual_vars) {
  int arg_max = 0;
  size_t mx = dual_vars.empty() ? 0 : dual_vars.at(0).size();
  for (size_t i = 1; i < dual_vars.size(); i++)
    if (dual_vars[i].size() > mx) {
      arg_max = static_cast<int>(i);
      mx = dual_vars[i].size();
    }
  if (mx > 0) return arg_max;
  arg_max = 0;
  mx = vars.at(0).size();
  for (size_t i = 1; i < vars.size(); i++)
    if (vars[i].size() > mx) {
      arg_max = static_cast<int>(i);
      mx = vars[i].size();
    }
  return arg_max;
}

MatrixData Analyze(const IntLists& cliques, const IntLists& dual_vars_in) {
  const int K = static_cast<int>(cliques.size());
  IntLists dual_vars = dual_vars_in;
  dual_vars.resize(K);
  std::vector<int> valid_leaf(K);
  for (int i = 0; i < K; i++) valid_leaf[i] = dual_vars[i].empty();
  const int root = GetRootNode(cliques, dual_vars);

  IntLists sorted = cliques;
  for (auto& c : sorted) std::sort(c.begin(), c.end());
  std::vector<int> order;
  IntLists supernodes, separators;
  RootedTree tree;
  PickCliqueOrder(sorted, valid_leaf, root, &order, &supernodes, &separators, &tree);

  // SupernodesToData (supernodal_solver.cc:389-431)
  MatrixData d;
  d.K = K;
  d.num_vars = GetMax(cliques) + 1;
  d.clique_order = order;
  d.permutation.assign(d.num_vars, 0);
  d.permutation_inverse.assign(d.num_vars, 0);
  int pos = 0;
  for (int e : order)
    for (int v : supernodes[e]) {
      d.permutation_inverse[pos] = v;
      d.permutation[v] = pos;
      pos++;
    }
  d.supernode_size.resize(K);
  d.cliques.resize(K);
  d.supernodes_orig.resize(K);
  d.separators_orig.resize(K);
  d.supernodes_pos.resize(K);
  d.separators_pos.resize(K);
  for (int k = 0; k < K; k++) {
    const int e = order[k];
    IntList temp;
    for (int v : separators[e]) temp.push_back(d.permutation[v]);
    std::sort(temp.begin(), temp.end());
    d.supernodes_orig[k] = supernodes[e];
    for (int t : temp) d.separators_orig[k].push_back(d.permutation_inverse[t]);
    for (int v : supernodes[e]) d.cliques[k].push_back(d.permutation[v]);
    for (int t : temp) d.cliques[k].push_back(t);
    d.supernode_size[k] = static_cast<int>(supernodes[e].size());
  }
  d.N = std::accumulate(d.supernode_size.begin(), d.supernode_size.end(), 0);

  // RelabelCliques (kkt_solver.cc:47-68): original label -> position in the constraint's
  // variable list (primal variables, then its multipliers); -1 marks fill-in.
  for (int k = 0; k < K; k++) {
    const int j = order[k];
    IntList labels(cliques[j].begin(), cliques[j].end() - dual_vars[j].size());
    labels.insert(labels.end(), dual_vars[j].begin(), dual_vars[j].end());
    auto position = [&](int v) {
      auto it = std::find(labels.begin(), labels.end(), v);
      return it == labels.end() ? -1 : static_cast<int>(it - labels.begin());
    };
    for (int v : d.supernodes_orig[k]) d.supernodes_pos[k].push_back(position(v));
    for (int v : d.separators_orig[k]) d.separators_pos[k].push_back(position(v));
  }
  return d;
}

// ---- chain-shaped trees: segment-parallel elimination order (symbolic.h)
bool IsChain(const MatrixData& ref) {
  const int K = ref.K;
  if (K < 2) return false;
  std::vector<int> step_of(ref.N, -1);
  for (int k = 0; k < K; k++) {
    if (ref.supernode_size[k] <= 0) return false;
    for (int i = 0; i < ref.supernode_size[k]; i++) step_of[ref.cliques[k][i]] = k;
  }
  for (int k = 0; k + 1 < K; k++) {
    const IntList& c = ref.cliques[k];
    if ((int)c.size() == ref.supernode_size[k]) return false;  // a second root: not one chain
    int parent = K;
    for (size_t i = ref.supernode_size[k]; i < c.size(); i++) parent = std::min(parent, step_of[c[i]]);
    if (parent != k + 1) return false;
  }
  return true;
}

bool SegmentChain(const MatrixData& ref, const IntLists& cliques, const IntLists& dual_vars, int segments,
                  MatrixData* out) {
  const int K = ref.K;
  if (segments < 2 || K < 2 * segments || !IsChain(ref)) return false;
  for (const IntList& dv : dual_vars)
    if (!dv.empty()) return false;  // (the LDLT path pivots inside the reference's blocks: left alone)
  const int nv = ref.num_vars;
  // owner (reference step) of every variable
  std::vector<int> owner(nv, -1);
  for (int k = 0; k < K; k++)
    for (int v : ref.supernodes_orig[k]) owner[v] = k;
  // Cuts.  The variables the step before a cut updates (its separator) are DEFERRED: they leave their
  // supernode and are eliminated with the HOST of the cut -- the last step of the segment in front of
  // it, whose own variables they are coupled to anyway -- after both neighbouring segments.  A step
  // keeps at least one variable of its own, a host hosts one cut.
  std::vector<char> deferred(nv, 0), is_host(K, 0);
  std::vector<int> cut_host;            // per accepted cut: its host step
  std::vector<IntList> cut_vars;        // ... and its deferred variables, in the reference's order
  int prev_cut = 0;
  for (int p = 1; p < segments; p++) {
    const int a = (int)((int64_t)p * K / segments);  // the cut lies between steps a - 1 and a
    if (a - prev_cut < 2 || a > K - 2) continue;      // (segments of at least two steps)
    IntList cand;
    for (int v : ref.separators_orig[a - 1])
      if (!deferred[v]) cand.push_back(v);
    if (cand.empty()) continue;
    bool fits = true;
    for (int v : cand) {
      int kept = 0, leaving = 0;
      for (int u : ref.supernodes_orig[owner[v]]) kept += !deferred[u];
      for (int u : cand) leaving += owner[u] == owner[v];
      if (kept - leaving < 1) fits = false;
    }
    if (!fits) continue;
    std::sort(cand.begin(), cand.end(), [&](int x, int y) { return ref.permutation[x] < ref.permutation[y]; });
    for (int v : cand) deferred[v] = 1;
    cut_host.push_back(a - 1);
    cut_vars.push_back(cand);
    is_host[a - 1] = 1;
    prev_cut = a;
  }
  const int C = (int)cut_host.size();
  if (C == 0) return false;
  // New order of the steps: every step that hosts nothing in the reference's order (the segments are
  // independent of each other now), then the hosts in nested-dissection order over the cuts -- the cut
  // in the middle of a range last, so that the interfaces merge pairwise, log2(segments) levels deep,
  // instead of forming one dense block.
  std::vector<int> seq;  // new position -> reference step
  for (int k = 0; k < K; k++)
    if (!is_host[k]) seq.push_back(k);
  {
    // post-order of the bisection tree over cuts [0, C), iteratively
    struct Range { int lo, hi; bool emit; };
    std::vector<Range> stack;
    stack.push_back({0, C - 1, false});
    while (!stack.empty()) {
      const Range r = stack.back();
      stack.pop_back();
      if (r.lo > r.hi) continue;
      const int mid = (r.lo + r.hi) / 2;
      if (r.emit) {
        seq.push_back(cut_host[mid]);
        continue;
      }
      stack.push_back({r.lo, r.hi, true});
      stack.push_back({mid + 1, r.hi, false});
      stack.push_back({r.lo, mid - 1, false});
    }
  }
  if ((int)seq.size() != K) return false;
  std::vector<int> newpos(K, -1);
  for (int k = 0; k < K; k++) newpos[seq[k]] = k;
  std::vector<int> host_cut(K, -1);
  for (int c = 0; c < C; c++) host_cut[cut_host[c]] = c;
  // supernodes: a step's own variables without the deferred ones; a host's, then its cut's
  IntLists sn(K), sep(K);
  std::vector<int> step_of(nv, -1);
  for (int k = 0; k < K; k++) {
    const int r = seq[k];
    for (int v : ref.supernodes_orig[r])
      if (!deferred[v]) sn[k].push_back(v);
    if (host_cut[r] >= 0)
      for (int v : cut_vars[host_cut[r]]) sn[k].push_back(v);
    if (sn[k].empty()) return false;
    for (int v : sn[k]) step_of[v] = k;
  }
  // separators by symbolic elimination: a step's structure is its constraint's variables and the
  // separators of the steps whose first update lands in it, minus what it eliminates itself
  std::vector<std::vector<int>> children(K);
  std::vector<char> mark(nv, 0);
  for (int k = 0; k < K; k++) {
    std::vector<int> st;
    auto add = [&](int v) {
      if (!mark[v]) {
        mark[v] = 1;
        st.push_back(v);
      }
    };
    const int con = ref.clique_order[seq[k]];
    for (int v : cliques[con]) add(v);
    for (int j : children[k])
      for (int v : sep[j]) add(v);
    int parent = K;
    bool valid = true;
    for (int v : st) {
      mark[v] = 0;
      if (step_of[v] == k) continue;
      if (step_of[v] < k) valid = false;  // (an order that eliminates a variable before its last update)
      sep[k].push_back(v);
      parent = std::min(parent, step_of[v]);
    }
    if (!valid) return false;
    if (parent < K) children[parent].push_back(k);
  }
  // MatrixData as Analyze builds it (SupernodesToData + RelabelCliques) for the new order
  MatrixData d;
  d.K = K;
  d.num_vars = nv;
  d.clique_order.resize(K);
  for (int k = 0; k < K; k++) d.clique_order[k] = ref.clique_order[seq[k]];
  d.permutation.assign(nv, 0);
  d.permutation_inverse.assign(nv, 0);
  int pos = 0;
  for (int k = 0; k < K; k++)
    for (int v : sn[k]) {
      d.permutation_inverse[pos] = v;
      d.permutation[v] = pos;
      pos++;
    }
  if (pos != ref.N) return false;
  d.N = ref.N;
  d.supernode_size.resize(K);
  d.cliques.resize(K);
  d.supernodes_orig.resize(K);
  d.separators_orig.resize(K);
  d.supernodes_pos.resize(K);
  d.separators_pos.resize(K);
  for (int k = 0; k < K; k++) {
    IntList temp;
    for (int v : sep[k]) temp.push_back(d.permutation[v]);
    std::sort(temp.begin(), temp.end());
    d.supernodes_orig[k] = sn[k];
    for (int t : temp) d.separators_orig[k].push_back(d.permutation_inverse[t]);
    for (int v : sn[k]) d.cliques[k].push_back(d.permutation[v]);
    for (int t : temp) d.cliques[k].push_back(t);
    d.supernode_size[k] = (int)sn[k].size();
    const IntList& labels = cliques[d.clique_order[k]];
    auto position = [&](int v) {
      auto it = std::find(labels.begin(), labels.end(), v);
      return it == labels.end() ? -1 : static_cast<int>(it - labels.begin());
    };
    for (int v : d.supernodes_orig[k]) d.supernodes_pos[k].push_back(position(v));
    for (int v : d.separators_orig[k]) d.separators_pos[k].push_back(position(v));
  }
  *out = d;
  return true;
}

namespace {
int64_t Pad4(int64_t n) { return (n + 3) & ~int64_t(3); }
}  // namespace

int64_t LookupAddress(const Layout& L, int r, int c) {
  const int node = L.var_to_sn[c];
  const int node_r = L.var_to_sn[r];
  const int j = L.var_to_pos[c];
  const int ns = L.supernode_size[node];
  if (node == node_r) return L.diag_off[node] + int64_t(j) * ns + L.var_to_pos[r];
  const IntList& sep = L.separators[node];
  for (size_t cnt = 0; cnt < sep.size(); cnt++)
    if (sep[cnt] == r) return L.offd_off[node] + int64_t(cnt) * ns + j;
  throw std::runtime_error("Specified entry of sparse matrix is not accessible.");
}

Layout BuildLayout(const MatrixData& md) {
  Layout L;
  L.K = md.K;
  L.N = md.N;
  L.supernode_size = md.supernode_size;
  L.supernode_start.resize(md.K);
  L.separators.resize(md.K);
  L.var_to_sn.resize(md.N);
  L.var_to_pos.resize(md.N);
  L.diag_off.resize(md.K);
  L.offd_off.resize(md.K);
  int var = 0;
  int64_t off = 0;
  for (int e = 0; e < md.K; e++) {
    const int ns = md.supernode_size[e];
    L.supernode_start[e] = var;
    for (int i = 0; i < ns; i++) {
      L.var_to_sn[var] = e;
      L.var_to_pos[var] = i;
      var++;
    }
    L.separators[e].assign(md.cliques[e].begin() + ns, md.cliques[e].end());
    L.diag_off[e] = off;
    off += Pad4(int64_t(ns) * ns);
    L.offd_off[e] = off;
    off += Pad4(int64_t(ns) * int64_t(L.separators[e].size()));
  }
  L.slab_size = off;
  L.ss_index.resize(md.K);
  for (int e = 0; e < md.K; e++) {
    const IntList& s = L.separators[e];
    for (size_t j = 0; j < s.size(); j++)
      for (size_t i = j; i < s.size(); i++) L.ss_index[e].push_back(LookupAddress(L, s[i], s[j]));
  }
  return L;
}

}  // namespace cxk
