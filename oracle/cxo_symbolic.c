/*
 * conex oracle (TEST INFRASTRUCTURE ONLY) -- symbolic analysis.
 *
 * Literal restatement of the reference's clique-tree elimination ordering:
 *   conex/clique_ordering.cc   (PickCliqueOrderHelper, GetCliqueEliminationOrder, FillIn)
 *   conex/tree_utils.cc        (PathInTree)
 *   conex/supernodal_solver.cc (Sort, IntersectionOfSorted, UnionOfSorted, GetData,
 *                               SupernodesToData)
 *   conex/kkt_solver.cc        (GetRootNode, is_empty, RelabelCliques)
 * Deliberately keeps the reference's O(K^2) scan and its quirks (the
 * "recompute if cached intersection is empty" rule, ">= max_weight" ties, the
 * early break for non-leaf cliques, the root re-pick for root == -1).
 */
#include <stdlib.h>
#include <string.h>

#include "cxo_internal.h"

/* ---------------- ivec ---------------- */
void iv_init(ivec* v) {
  v->d = NULL;
  v->n = 0;
  v->cap = 0;
}
void iv_free(ivec* v) {
  free(v->d);
  v->d = NULL;
  v->n = v->cap = 0;
}
void iv_clear(ivec* v) { v->n = 0; }
void iv_push(ivec* v, int x) {
  if (v->n == v->cap) {
    v->cap = v->cap ? 2 * v->cap : 4;
    v->d = (int*)realloc(v->d, sizeof(int) * (size_t)v->cap);
  }
  v->d[v->n++] = x;
}
void iv_copy(ivec* dst, const ivec* src) {
  iv_clear(dst);
  for (int i = 0; i < src->n; i++) iv_push(dst, src->d[i]);
}
static int cmp_int(const void* a, const void* b) {
  int x = *(const int*)a, y = *(const int*)b;
  return (x > y) - (x < y);
}
void iv_sort(ivec* v) {
  if (v->n > 1) qsort(v->d, (size_t)v->n, sizeof(int), cmp_int);
}
void iv_intersection(const ivec* a, const ivec* b, ivec* out) {
  iv_clear(out);
  int i = 0, j = 0;
  while (i < a->n && j < b->n) {
    if (a->d[i] < b->d[j]) {
      i++;
    } else if (b->d[j] < a->d[i]) {
      j++;
    } else {
      iv_push(out, a->d[i]);
      i++;
      j++;
    }
  }
}
void iv_union(const ivec* a, const ivec* b, ivec* out) {
  ivec tmp;
  iv_init(&tmp);
  int i = 0, j = 0;
  while (i < a->n && j < b->n) {
    if (a->d[i] < b->d[j]) {
      iv_push(&tmp, a->d[i++]);
    } else if (b->d[j] < a->d[i]) {
      iv_push(&tmp, b->d[j++]);
    } else {
      iv_push(&tmp, a->d[i]);
      i++;
      j++;
    }
  }
  while (i < a->n) iv_push(&tmp, a->d[i++]);
  while (j < b->n) iv_push(&tmp, b->d[j++]);
  iv_copy(out, &tmp);
  iv_free(&tmp);
}
void iv_difference(const ivec* a, const ivec* b, ivec* out) {
  iv_clear(out);
  int i = 0, j = 0;
  while (i < a->n) {
    if (j >= b->n || a->d[i] < b->d[j]) {
      iv_push(out, a->d[i++]);
    } else if (b->d[j] < a->d[i]) {
      j++;
    } else {
      i++;
      j++;
    }
  }
}
ivec* ivs_new(int n) {
  ivec* v = (ivec*)malloc(sizeof(ivec) * (size_t)(n > 0 ? n : 1));
  for (int i = 0; i < n; i++) iv_init(&v[i]);
  return v;
}
void ivs_free(ivec* v, int n) {
  if (!v) return;
  for (int i = 0; i < n; i++) iv_free(&v[i]);
  free(v);
}

/* ---------------- tree_utils.cc:11-25 ---------------- */
void cxo_path_in_tree_iv(int x, int y, const int* parent, const int* depth, ivec* path) {
  iv_clear(path);
  while (x != y) {
    if (depth[x] < depth[y]) {
      iv_push(path, y);
      y = parent[y];
    } else {
      iv_push(path, x);
      x = parent[x];
    }
  }
  iv_push(path, x);
}

/* ---------------- clique_ordering.cc ---------------- */
static int get_max(int K, const ivec* cliques) { /* :16-26 */
  int max = cliques[0].d[0];
  for (int c = 0; c < K; c++)
    for (int i = 0; i < cliques[c].n; i++)
      if (cliques[c].d[i] > max) max = cliques[c].d[i];
  return max;
}

static size_t linear_index(int i, int j, int n) { /* :28-34 */
  return (i > j) ? (size_t)j * n + i : (size_t)i * n + j;
}

typedef struct {
  int n;
  ivec* data; /* n*n */
} symmat;

/* Weight::get_weight :92-108 */
static size_t get_weight(symmat* inter, const ivec* cliques_sorted, const int* valid_leaf,
                         int active, int i) {
  ivec* e = &inter->data[linear_index(active, i, inter->n)];
  if (e->n == 0) iv_intersection(&cliques_sorted[active], &cliques_sorted[i], e);
  size_t weight = (size_t)e->n;
  if (valid_leaf) {
    if (!valid_leaf[i]) weight += 10000;
    if (!valid_leaf[active]) weight += 10000;
  }
  return weight;
}

/* PickCliqueOrderHelper :111-201 */
static int pick_helper(int K, const ivec* cliques_sorted, const int* valid_leaf, int root_in,
                       symmat* inter, ivec* separators, ivec* order, int* parent, int* height) {
  int n = K;
  int* visited = (int*)calloc((size_t)n, sizeof(int));
  ivec stack, argmax, edges_a, edges_b;
  iv_init(&stack);
  iv_init(&argmax);
  iv_init(&edges_a);
  iv_init(&edges_b);
  int root = root_in < 0 ? 0 : root_in;
  iv_push(&stack, root);
  iv_clear(order);

  while (order->n < n) {
    int active = stack.d[stack.n - 1];
    if (visited[active] == 0) {
      iv_push(order, active);
      visited[active] = 1;
      parent[active] = active;
      height[active] = 0;
    }
    size_t max_weight = 1;
    iv_clear(&argmax);
    for (int i = 0; i < n; i++) {
      if (i == active) continue;
      size_t w = get_weight(inter, cliques_sorted, valid_leaf, active, i);
      if (w >= max_weight && !visited[i]) {
        if (w > max_weight) {
          iv_clear(&argmax);
          max_weight = w;
        }
        iv_push(&argmax, i);
      }
    }
    for (int k = 0; k < argmax.n; k++) {
      int e = argmax.d[k];
      iv_copy(&separators[e], &inter->data[linear_index(active, e, inter->n)]);
      iv_push(&stack, e);
      iv_push(order, e);
      visited[e] = 1;
      iv_push(&edges_a, active);
      iv_push(&edges_b, e);
      parent[e] = active;
      height[e] = height[active] + 1;
      if (valid_leaf && !valid_leaf[e]) break;
    }
    if (argmax.n == 0) {
      stack.n--;
      if (stack.n == 0) {
        int node = -1;
        for (int i = 0; i < n; i++)
          if (visited[i] == 0) {
            node = i;
            break;
          }
        if (node == -1) break;
        iv_push(&stack, node);
      }
    }
  }

  /* GetMaxWeightedDegreeNode :62-75 + max_element (first max) */
  int* weights = (int*)calloc((size_t)n, sizeof(int));
  for (int k = 0; k < edges_a.n; k++) {
    int sz = inter->data[linear_index(edges_a.d[k], edges_b.d[k], inter->n)].n;
    weights[edges_a.d[k]] += sz;
    weights[edges_b.d[k]] += sz;
  }
  int root_node = 0;
  for (int i = 1; i < n; i++)
    if (weights[i] > weights[root_node]) root_node = i;

  /* std::reverse(order) */
  for (int i = 0, j = order->n - 1; i < j; i++, j--) {
    int t = order->d[i];
    order->d[i] = order->d[j];
    order->d[j] = t;
  }
  free(weights);
  free(visited);
  iv_free(&stack);
  iv_free(&argmax);
  iv_free(&edges_a);
  iv_free(&edges_b);
  return root_node;
}

/* GetCliqueEliminationOrder :203-240 */
static void get_clique_elimination_order(int K, const ivec* cliques_sorted, const int* valid_leaf,
                                         int root, ivec* order, ivec* supernodes,
                                         ivec* separators, int* parent, int* height) {
  symmat inter;
  inter.n = K;
  inter.data = ivs_new(K * K);
  for (int i = 0; i < K; i++) iv_clear(&separators[i]);
  int better_root =
      pick_helper(K, cliques_sorted, valid_leaf, root, &inter, separators, order, parent, height);
  if (root == -1) {
    for (int i = 0; i < K; i++) iv_clear(&separators[i]);
    /* RootedTree tree_i(n): value-initialised to zero */
    for (int i = 0; i < K; i++) {
      parent[i] = 0;
      height[i] = 0;
    }
    pick_helper(K, cliques_sorted, valid_leaf, better_root, &inter, separators, order, parent,
                height);
  }
  for (int k = 0; k < order->n; k++) {
    int e = order->d[k];
    iv_difference(&cliques_sorted[e], &separators[e], &supernodes[e]);
  }
  ivs_free(inter.data, K * K);
}

/* FillIn :261-305 */
static void fill_in(int K, const int* parent, const int* height, int num_variables,
                    const ivec* order, ivec* supernodes, ivec* separators) {
  int* eliminated = (int*)malloc(sizeof(int) * (size_t)num_variables);
  int num_cliques = order->n;
  for (int i = 0; i < num_variables; i++) eliminated[i] = num_cliques + 1;
  ivec path, single;
  iv_init(&path);
  iv_init(&single);
  for (int i = 0; i < order->n; i++) {
    const ivec* sn = &supernodes[order->d[i]];
    for (int q = 0; q < sn->n; q++) {
      int v = sn->d[q];
      if (eliminated[v] < num_cliques) {
        cxo_path_in_tree_iv(order->d[i], eliminated[v], parent, height, &path);
        for (int j = 0; j < path.n - 1; j++) {
          int e = path.d[j];
          iv_clear(&single);
          iv_push(&single, v);
          iv_union(&separators[e], &single, &separators[e]);
        }
        eliminated[v] = path.d[path.n - 1];
      } else {
        eliminated[v] = order->d[i];
      }
    }
  }
  for (int i = 0; i < K; i++) iv_clear(&supernodes[i]);
  for (int i = 0; i < num_variables; i++)
    if (eliminated[i] < num_cliques) iv_push(&supernodes[eliminated[i]], i);
  for (int i = 0; i < K; i++) {
    iv_sort(&separators[i]);
    iv_sort(&supernodes[i]);
  }
  free(eliminated);
  iv_free(&path);
  iv_free(&single);
}

/* PickCliqueOrder :307-333 (post_order output is test-only and omitted) */
void cxo_pick_clique_order_iv(int K, const ivec* cliques_sorted, const int* valid_leaf, int root,
                              int* order_out, ivec* supernodes, ivec* separators,
                              int* tree_parent, int* tree_height) {
  ivec order;
  iv_init(&order);
  int* parent = tree_parent ? tree_parent : (int*)calloc((size_t)K, sizeof(int));
  int* height = tree_height ? tree_height : (int*)calloc((size_t)K, sizeof(int));
  if (tree_parent) memset(parent, 0, sizeof(int) * (size_t)K);
  if (tree_height) memset(height, 0, sizeof(int) * (size_t)K);
  get_clique_elimination_order(K, cliques_sorted, valid_leaf, root, &order, supernodes,
                               separators, parent, height);
  int num_vars = get_max(K, cliques_sorted) + 1;
  fill_in(K, parent, height, num_vars, &order, supernodes, separators);
  for (int i = 0; i < K; i++) order_out[i] = order.d[i];
  iv_free(&order);
  if (!tree_parent) free(parent);
  if (!tree_height) free(height);
}

/* ---------------- kkt_solver.cc:70-94 ---------------- */
int cxo_get_root_node(int K, const ivec* vars, const ivec* dual_vars) {
  int arg_max = 0;
  int max = dual_vars ? dual_vars[0].n : 0;
  if (dual_vars) {
    for (int i = 1; i < K; i++)
      if (dual_vars[i].n > max) {
        arg_max = i;
        max = dual_vars[i].n;
      }
  }
  if (max > 0) return arg_max;
  arg_max = 0;
  max = vars[0].n;
  for (int i = 1; i < K; i++)
    if (vars[i].n > max) {
      arg_max = i;
      max = vars[i].n;
    }
  return arg_max;
}

/* ---------------- supernodal_solver.cc:389-431 ---------------- */
static cxo_matrix_data* supernodes_to_data(int K, int num_vars, const int* order,
                                           const ivec* supernodes, const ivec* separators) {
  cxo_matrix_data* d = (cxo_matrix_data*)calloc(1, sizeof(cxo_matrix_data));
  d->K = K;
  d->num_vars = num_vars;
  d->clique_order = (int*)malloc(sizeof(int) * (size_t)K);
  memcpy(d->clique_order, order, sizeof(int) * (size_t)K);
  d->permutation = (int*)calloc((size_t)num_vars, sizeof(int));
  d->permutation_inverse = (int*)calloc((size_t)num_vars, sizeof(int));
  int i = 0;
  for (int k = 0; k < K; k++) {
    int e = order[k];
    for (int q = 0; q < supernodes[e].n; q++) {
      int v = supernodes[e].d[q];
      d->permutation_inverse[i] = v;
      d->permutation[v] = i;
      i++;
    }
  }
  d->supernode_size = (int*)malloc(sizeof(int) * (size_t)K);
  d->cliques = ivs_new(K);
  d->supernodes_orig = ivs_new(K);
  d->separators_orig = ivs_new(K);
  d->supernodes_pos = ivs_new(K);
  d->separators_pos = ivs_new(K);
  d->N = 0;
  ivec temp;
  iv_init(&temp);
  for (int k = 0; k < K; k++) {
    int e = order[k];
    /* temp = sort(Relabel(separators[e], permutation)); sep = Relabel(temp, permutation_inverse) */
    iv_clear(&temp);
    for (int q = 0; q < separators[e].n; q++) iv_push(&temp, d->permutation[separators[e].d[q]]);
    iv_sort(&temp);
    iv_copy(&d->supernodes_orig[k], &supernodes[e]);
    iv_clear(&d->separators_orig[k]);
    for (int q = 0; q < temp.n; q++)
      iv_push(&d->separators_orig[k], d->permutation_inverse[temp.d[q]]);
    iv_clear(&d->cliques[k]);
    for (int q = 0; q < supernodes[e].n; q++)
      iv_push(&d->cliques[k], d->permutation[supernodes[e].d[q]]);
    for (int q = 0; q < temp.n; q++) iv_push(&d->cliques[k], temp.d[q]);
    d->supernode_size[k] = supernodes[e].n;
    d->N += supernodes[e].n;
  }
  iv_free(&temp);
  return d;
}

/* kkt_solver.cc:11-68: ConcatFirstN / ReplaceWithPosition / RelabelCliques */
static void relabel_cliques(cxo_matrix_data* d, const ivec* cliques, const ivec* dual_vars) {
  ivec labels;
  iv_init(&labels);
  for (int e = 0; e < d->K; e++) {
    int j = d->clique_order[e];
    iv_clear(&labels);
    int nd = dual_vars ? dual_vars[j].n : 0;
    for (int i = 0; i < cliques[j].n - nd; i++) iv_push(&labels, cliques[j].d[i]);
    for (int i = 0; i < nd; i++) iv_push(&labels, dual_vars[j].d[i]);
    for (int pass = 0; pass < 2; pass++) {
      const ivec* src = pass == 0 ? &d->supernodes_orig[e] : &d->separators_orig[e];
      ivec* dst = pass == 0 ? &d->supernodes_pos[e] : &d->separators_pos[e];
      iv_clear(dst);
      for (int q = 0; q < src->n; q++) {
        int pos = -1;
        for (int t = 0; t < labels.n; t++)
          if (labels.d[t] == src->d[q]) {
            pos = t;
            break;
          }
        iv_push(dst, pos);
      }
    }
  }
  iv_free(&labels);
}

cxo_matrix_data* cxo_matrix_data_build(int K, const ivec* cliques, const ivec* dual_vars) {
  /* is_empty(dual_vars) kkt_solver.cc:96-102 */
  int* valid_leaf = (int*)malloc(sizeof(int) * (size_t)K);
  for (int i = 0; i < K; i++) valid_leaf[i] = dual_vars ? (dual_vars[i].n == 0) : 1;
  int root = cxo_get_root_node(K, cliques, dual_vars);
  /* GetData supernodal_solver.cc:376-387 */
  ivec* sorted = ivs_new(K);
  for (int i = 0; i < K; i++) {
    iv_copy(&sorted[i], &cliques[i]);
    iv_sort(&sorted[i]);
  }
  int* order = (int*)malloc(sizeof(int) * (size_t)K);
  ivec* supernodes = ivs_new(K);
  ivec* separators = ivs_new(K);
  int* parent = (int*)calloc((size_t)K, sizeof(int));
  int* height = (int*)calloc((size_t)K, sizeof(int));
  cxo_pick_clique_order_iv(K, sorted, valid_leaf, root, order, supernodes, separators, parent,
                           height);
  cxo_matrix_data* d = supernodes_to_data(K, get_max(K, cliques) + 1, order, supernodes, separators);
  d->pc_supernodes = supernodes;
  d->pc_separators = separators;
  d->tree_parent = parent;
  d->tree_height = height;
  relabel_cliques(d, cliques, dual_vars);
  ivs_free(sorted, K);
  free(order);
  free(valid_leaf);
  return d;
}

cxo_matrix_data* cxo_matrix_data_from_supernodes(int K, const ivec* cliques, int num_vars,
                                                 const int* order, const ivec* supernodes,
                                                 const ivec* separators) {
  cxo_matrix_data* d = supernodes_to_data(K, num_vars, order, supernodes, separators);
  relabel_cliques(d, cliques, NULL);
  return d;
}

void cxo_matrix_data_free(cxo_matrix_data* d) {
  if (!d) return;
  free(d->clique_order);
  ivs_free(d->cliques, d->K);
  ivs_free(d->supernodes_orig, d->K);
  ivs_free(d->separators_orig, d->K);
  ivs_free(d->supernodes_pos, d->K);
  ivs_free(d->separators_pos, d->K);
  ivs_free(d->pc_supernodes, d->K);
  ivs_free(d->pc_separators, d->K);
  free(d->supernode_size);
  free(d->permutation);
  free(d->permutation_inverse);
  free(d->tree_parent);
  free(d->tree_height);
  free(d);
}
