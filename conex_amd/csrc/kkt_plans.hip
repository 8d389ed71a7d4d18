// Symbolic plans of a context (host code): dependency levels of the elimination tree and the
// partition of a sharded context, then every index table the kernels read -- gather lists, pull
// lists in consumer order, level segments, the chain at the top, paired downward levels, the
// records of the whole-tree launch, the dense top.  Split off kkt_context.hip (launches + C-ABI).
#define CXK_DEVICE_FUNCTIONS_ONLY  // kernels_kkt.hip.h: types and templates only (the plain kernels live in kkt_context.hip)
#include "kkt_internal.h"
#include "big_chol.h"

namespace cxk_host {

// ---------------------------------------------------------------- tree structure + partition
// Dependency levels of the supernodal elimination tree and (world > 1) the split into a
// replicated top T (all levels >= cut_level) and per-rank subtrees.
void ComputeTreeStructure(cxk_context* ctx) {
  const Layout& L = ctx->lay;
  const int K = ctx->md.K;
  ctx->t_ns.assign(K, 0);
  ctx->t_nsep.assign(K, 0);
  ctx->t_start.assign(K, 0);
  ctx->t_level.assign(K, 0);
  ctx->t_parent.assign(K, -1);
  for (int e = 0; e < K; e++) {
    ctx->t_ns[e] = L.supernode_size[e];
    ctx->t_nsep[e] = (int)L.separators[e].size();
    ctx->t_start[e] = L.supernode_start[e];
    if (ctx->t_nsep[e] > 0) ctx->t_parent[e] = L.var_to_sn[L.separators[e][0]];
  }
  // a supernode sits one level above every supernode that updates it (children have smaller
  // elimination index, so one ascending pass suffices)
  for (int i = 0; i < K; i++) {
    if (ctx->t_ns[i] == 0) continue;
    for (int v : L.separators[i]) {
      const int p = L.var_to_sn[v];
      if (ctx->t_level[p] < ctx->t_level[i] + 1) ctx->t_level[p] = ctx->t_level[i] + 1;
    }
  }
  ctx->nlev = 0;
  for (int e = 0; e < K; e++)
    if (ctx->t_ns[e] > 0) ctx->nlev = std::max(ctx->nlev, ctx->t_level[e] + 1);
}

double ConstraintWork(const ConstraintRec& c) {
  const double n = c.n, m = c.m;
  switch (c.type) {
    case CXK_LMI: return 4 * n * n * n * (m + 1) + n * n * m * m;
    case CXK_LINEAR: return n * m * m;
    case CXK_SOC: return (n + 1) * m * m;
    case CXK_QUAD: return (n + 1) * m * m;
    case CXK_OCT: return 8 * 64 * 8 * n * n * n * (m + 1) + 8 * n * n * m * m;
    default: return m * m;
  }
}

void PartitionTree(cxk_context* ctx) {
  const int K = ctx->md.K, G = ctx->world;
  ctx->sn_top.assign(K, 0);
  ctx->sn_mine.assign(K, 1);
  ctx->owned.assign(K, 1);
  ctx->cut_level = ctx->nlev;
  ctx->var_valid.assign(ctx->md.N, 1);
  ctx->n_xs = 0;
  ctx->n_xv = 0;
  if (G <= 1) return;
  // Choose the cut: the highest level (smallest replicated top, smallest exchange) whose
  // longest-processing-time assignment of subtrees is balanced within 15 % of the ideal;
  // if no level achieves that, the best-balanced one.
  std::vector<int> root_of(K, -1), owner_of_root(K, 0);
  double total = 0;
  for (int e = 0; e < K; e++) total += ConstraintWork(ctx->cons[ctx->md.clique_order[e]]);
  auto try_cut = [&](int cut, std::vector<int>* roots_out, std::vector<int>* root_of_out,
                     std::vector<int>* owner_out) -> double {
    std::vector<int> ro(K, -1);
    std::vector<double> weight(K, 0.0);
    auto is_top = [&](int e) { return ctx->t_ns[e] > 0 && ctx->t_level[e] >= cut; };
    for (int e = K - 1; e >= 0; e--) {  // parents have larger elimination index
      if (is_top(e)) continue;
      const int par = ctx->t_parent[e];
      ro[e] = (par < 0 || is_top(par)) ? e : ro[par];
    }
    double top_work = 0;
    for (int e = 0; e < K; e++) {
      const double w = ConstraintWork(ctx->cons[ctx->md.clique_order[e]]);
      if (is_top(e))
        top_work += w;
      else
        weight[ro[e]] += w;
    }
    std::vector<int> roots;
    for (int e = 0; e < K; e++)
      if (!is_top(e) && ro[e] == e) roots.push_back(e);
    std::stable_sort(roots.begin(), roots.end(), [&](int a, int b) { return weight[a] > weight[b]; });
    std::vector<double> load(G, top_work / G);
    std::vector<int> owner(K, 0);
    for (int r : roots) {  // longest processing time first, ties to the lowest rank
      int best = 0;
      for (int g = 1; g < G; g++)
        if (load[g] < load[best]) best = g;
      owner[r] = best;
      load[best] += weight[r];
    }
    if (roots_out) *roots_out = roots;
    if (root_of_out) *root_of_out = ro;
    if (owner_out) *owner_out = owner;
    return *std::max_element(load.begin(), load.end());
  };
  int cut = ctx->nlev;
  if (ctx->nlev > 1) {
    int best_cut = std::max(ctx->nlev - 1, 1);
    double best_load = -1;
    for (int c = std::max(ctx->nlev - 1, 1); c >= 1; c--) {
      const double mx = try_cut(c, nullptr, nullptr, nullptr);
      if (best_load < 0 || mx < best_load * 0.999) {
        best_load = mx;
        best_cut = c;
      }
      if (mx <= 1.15 * total / G) {
        best_cut = c;
        break;
      }
    }
    cut = best_cut;
  }
  ctx->cut_level = cut;
  std::vector<int> roots;
  try_cut(cut, &roots, &root_of, &owner_of_root);
  for (int e = 0; e < K; e++) ctx->sn_top[e] = ctx->t_ns[e] > 0 && ctx->t_level[e] >= cut;
  int rr = 0;
  for (int e = 0; e < K; e++) {
    const int i = ctx->md.clique_order[e];
    if (ctx->sn_top[e]) {
      ctx->sn_mine[e] = 1;                             // T is factored by every rank
      ctx->owned[i] = (rr++ % G) == ctx->rank;          // its constraints are dealt round-robin
    } else {
      const bool mine = owner_of_root[root_of[e]] == ctx->rank;
      ctx->sn_mine[e] = mine;
      ctx->owned[i] = mine;
    }
  }
  for (int p = 0; p < ctx->md.N; p++) ctx->var_valid[p] = ctx->sn_mine[ctx->lay.var_to_sn[p]];
  for (int e = 0; e < K; e++)
    if (ctx->sn_top[e]) {
      const int64_t n = ctx->t_ns[e];
      ctx->n_xs += n * (n + 1) / 2 + n * ctx->t_nsep[e];
      ctx->n_xv += (int)n;
    }
}

// ---------------------------------------------------------------- plan building
int BuildPlans(cxk_context* ctx) {
  const MatrixData& md = ctx->md;
  const Layout& L = ctx->lay;
  const int K = md.K, N = md.N;
  const std::vector<int>& ns = ctx->t_ns;
  const std::vector<int>& nsep = ctx->t_nsep;
  const std::vector<int>& start = ctx->t_start;
  const bool sharded = ctx->world > 1;
  auto block_wanted = [&](int e) { return !sharded || ctx->sn_mine[e]; };

  // ---- assembly gather (UpdateBlocks order: elimination index descending).  In sharded mode
  // only the blocks this rank factors are written; sources of constraints owned elsewhere are
  // dropped, which leaves PARTIAL sums in the top blocks (completed by the exchange).
  std::vector<int> entry_of(L.slab_size, -1);
  std::vector<int64_t> dst;
  std::vector<std::vector<int64_t>> srcs;
  auto entry = [&](int64_t off) -> std::vector<int64_t>& {
    if (entry_of[off] < 0) {
      entry_of[off] = static_cast<int>(dst.size());
      dst.push_back(off);
      srcs.emplace_back();
    }
    return srcs[entry_of[off]];
  };
  const bool quirks = ctx->reference_identity > 0;
  for (int e = K - 1; e >= 0; e--) {
    const int i = md.clique_order[e];
    const int m = ctx->cons[i].m;
    const int64_t base = ctx->g_off[i];
    const bool mine = ctx->owned[i];
    auto coeff = [&](int a, int b) -> int64_t {  // GetCoeff supernodal_assembler.cc:59-70
      if (a < 0 || b < 0 || !mine) return -1;
      return a >= b ? base + (int64_t)b * m + a : base + (int64_t)a * m + b;
    };
    const IntList& r = md.supernodes_pos[e];
    const IntList& s = md.separators_pos[e];
    const int nse = (int)r.size(), nsp = (int)s.size();
    // The reference's direct_update test (BindDiagonalBlock, supernodal_assembler.cc:72-91) passes
    // on a supernode whose positions are -1, 0, .., m-2 (a fill-in variable in front) and then
    // writes G one row/column off and drops the separator terms -- a defect.
    // Reproduced as written by default (reference identity); cxk_set_reference_identity(ctx, 0) /
    // CXK_REFERENCE_QUIRKS=0 scatters every block by position instead.
    bool misplaced = false;
    if (quirks && nse > 0 && m == nse && r[0] != 0) {
      misplaced = true;
      for (int q = 1; q < nse; q++) misplaced = misplaced && r[q] > r[q - 1];
    }
    if (block_wanted(e)) {
      for (int j = 0; j < nse; j++)  // SetLowerTri
        for (int i2 = j; i2 < nse; i2++) {
          auto& v = entry(L.diag_off[e] + (int64_t)j * nse + i2);
          v.clear();
          v.push_back(misplaced ? coeff(i2, j) : coeff(r[i2], r[j]));
        }
      if (nse > 0)
        for (int j = 0; j < nsp; j++)  // Set
          for (int i2 = 0; i2 < nse; i2++) {
            auto& v = entry(L.offd_off[e] + (int64_t)j * nse + i2);
            v.clear();
            v.push_back(misplaced ? (int64_t)-1 : coeff(r[i2], s[j]));
          }
    }
    if (misplaced) continue;  // UpdateBlocks returns before Scatter on a direct update
    int cnt = 0;
    for (int j = 0; j < nsp; j++)  // Scatter
      for (int i2 = j; i2 < nsp; i2++) {
        const int64_t off = L.ss_index[e][cnt++];
        const int owner_sn = L.var_to_sn[L.separators[e][j]];
        if (block_wanted(owner_sn)) entry(off).push_back(coeff(s[i2], s[j]));
      }
  }
  std::vector<int> as_ptr(dst.size() + 1, 0);
  std::vector<int64_t> as_src;
  for (size_t t = 0; t < dst.size(); t++) {
    for (int64_t q : srcs[t]) as_src.push_back(q);
    as_ptr[t + 1] = (int)as_src.size();
  }
  std::vector<GatherRec> h_as_rec;
  ctx->as_T = (int64_t)dst.size();
  CXK_TRY(ctx->as_dst.upload(dst));
  CXK_TRY(ctx->as_ptr.upload(as_ptr));
  CXK_TRY(ctx->as_src.upload(as_src));
  {
    std::vector<GatherRec> recs(dst.size());
    for (size_t t = 0; t < dst.size(); t++) {
      const int len = as_ptr[t + 1] - as_ptr[t];
      recs[t].dst = dst[t];
      recs[t].first = len > 0 ? as_src[as_ptr[t]] : -1;
      recs[t].beg = as_ptr[t] + 1;
      recs[t].extra = len > 0 ? len - 1 : 0;
    }
    CXK_TRY(ctx->as_rec.upload(recs));
    h_as_rec = recs;
  }

  // ---- residual gather (constraint order); variables of foreign subtrees are skipped
  std::vector<std::vector<int64_t>> per(N);
  std::vector<ResidRec> h_rs_rec;
  {
    for (int i = 0; i < (int)ctx->cons.size(); i++) {
      if (!ctx->owned[i]) continue;
      for (int q = 0; q < (int)ctx->cliques[i].size(); q++)
        per[md.permutation[ctx->cliques[i][q]]].push_back(ctx->r_off[i] + q);
    }
    std::vector<int> ptr(N + 1, 0);
    std::vector<int64_t> src;
    for (int p = 0; p < N; p++) {
      for (int64_t q : per[p]) src.push_back(q);
      ptr[p + 1] = (int)src.size();
    }
    CXK_TRY(ctx->rs_ptr.upload(ptr));
    CXK_TRY(ctx->rs_src.upload(src));
    std::vector<ResidRec> recs(N);
    for (int p = 0; p < N; p++) {
      const int len = ptr[p + 1] - ptr[p];
      recs[p].first = len > 0 ? src[ptr[p]] : -1;
      recs[p].beg = ptr[p] + 1;
      recs[p].extra = len > 0 ? len - 1 : 0;
    }
    CXK_TRY(ctx->rs_rec.upload(recs));
    h_rs_rec = recs;
  }

  // ---- clique variables in permuted numbering
  {
    std::vector<int> ptr(ctx->cons.size() + 1, 0), perm;
    for (size_t i = 0; i < ctx->cons.size(); i++) {
      for (int v : ctx->cliques[i]) perm.push_back(md.permutation[v]);
      ptr[i + 1] = (int)perm.size();
    }
    CXK_TRY(ctx->cl_ptr.upload(ptr));
    CXK_TRY(ctx->cl_perm.upload(perm));
  }

  // ---- published-update slots: s(s+1)/2 Schur values and s forward values per supernode
  std::vector<int> h_tg_ptr, h_fs_ptr, h_bs_ptr, h_bs_c, h_bs_row;  // host copies for the per-supernode records
  std::vector<int64_t> h_pt_dst;                                     // slab targets of the pre-contributed updates
  std::vector<int64_t> upd_off(K, 0);
  std::vector<int> updb_off(K, 0);
  int64_t upd_total = 0;
  int updb_total = 0;
  for (int i = 0; i < K; i++) {
    upd_off[i] = upd_total;
    updb_off[i] = updb_total;
    if (ns[i] > 0) {
      upd_total += (int64_t)nsep[i] * (nsep[i] + 1) / 2;
      updb_total += nsep[i];
    }
  }
  // pull lists.  A target inside a subtree only has children of the same subtree.  A target in
  // the top T pulls its T children during the T sweep; the updates of THIS rank's subtrees are
  // folded in before the exchange (pre-reduce lists pt_* / pf_*).
  std::vector<int> tgt_of(L.slab_size, -1);
  std::vector<std::vector<int>> tg_of_sn(K);
  std::vector<int64_t> tg_dst_all;
  std::vector<std::vector<int64_t>> contrib, pre_contrib;
  std::vector<std::vector<int>> fs(N), pre_fs(N);
  for (int i = 0; i < K; i++) {
    if (ns[i] == 0 || nsep[i] == 0) continue;
    if (sharded && !ctx->sn_mine[i]) continue;  // foreign subtree: its updates arrive by exchange
    const IntList& s = L.separators[i];
    int cnt = 0;
    for (int k = 0; k < nsep[i]; k++) {
      const int p = L.var_to_sn[s[k]];
      const bool pre = sharded && ctx->sn_top[p] && !ctx->sn_top[i];
      for (int j = k; j < nsep[i]; j++) {
        const int64_t off = L.ss_index[i][cnt];
        if (tgt_of[off] < 0) {
          tgt_of[off] = (int)tg_dst_all.size();
          tg_dst_all.push_back(off);
          contrib.emplace_back();
          pre_contrib.emplace_back();
          tg_of_sn[p].push_back(tgt_of[off]);
        }
        (pre ? pre_contrib : contrib)[tgt_of[off]].push_back(upd_off[i] + cnt);
        cnt++;
      }
      (pre ? pre_fs : fs)[s[k]].push_back(updb_off[i] + k);
    }
  }
  // Slots.  A published value is written straight to the place its (single) consumer reads it
  // from: the contributions of target t of supernode p occupy  upd[ubase_p + t_local * m_p + i],
  // i = position in the reference's accumulation order, m_p = longest list of p (unused slots
  // stay 0.0: subtracting them is exact).  The consumer can therefore issue every load as soon
  // as it knows its record -- no index lists on the critical path.  Publishers look their slot up
  // in pub_dst (indexed by the child-side numbering upd_off[i] + t).  Pre-reduce contributions
  // (subtree -> top, sharded runs) get plain list slots after the dense region.
  std::vector<int64_t> h_ubase(K, 0);
  std::vector<int> h_m(K, 0), h_fbase(K, 0), h_mf(K, 0);
  std::vector<int> pub_dst((size_t)upd_total, -1), pubb_dst((size_t)updb_total, -1);
  int64_t slots = 0;
  int slotsb = 0;
  {
    std::vector<int> tg_ptr(K + 1, 0), tr_ptr, tg_loc, tg_reg, pt_ptr;
    std::vector<int64_t> tr_src, pt_dst, pt_src;
    tr_ptr.push_back(0);
    pt_ptr.push_back(0);
    for (int p = 0; p < K; p++) {
      size_t m = 0;
      for (int t : tg_of_sn[p]) m = std::max(m, contrib[t].size());
      h_ubase[p] = slots;
      h_m[p] = (int)m;
      int tl = 0;
      for (int t : tg_of_sn[p]) {
        const int64_t off = tg_dst_all[t];
        const int64_t dsz = (int64_t)ns[p] * ns[p];
        if (!contrib[t].empty()) {
          tg_loc.push_back(off >= L.diag_off[p] && off < L.diag_off[p] + dsz
                               ? (int)(off - L.diag_off[p])
                               : (int)(dsz + off - L.offd_off[p]));
          {
            // the same entry in the register-shaped image of FactorSupernodeLean: 64 * column + lane
            const int nsm = RegisterShape(ns[p], nsep[p]) >> 8, loc = tg_loc.back();
            const int col = loc < dsz ? loc / ns[p] : (int)(loc - dsz) % ns[p];
            const int ln = loc < dsz ? loc % ns[p] : nsm + (int)(loc - dsz) / ns[p];
            tg_reg.push_back(nsm > 0 ? 64 * col + ln : 0);
          }
          for (size_t i = 0; i < contrib[t].size(); i++) {
            const int64_t slot = slots + (int64_t)tl * (int64_t)m + (int64_t)i;
            pub_dst[contrib[t][i]] = (int)slot;
            tr_src.push_back(slot);
          }
          tr_ptr.push_back((int)tr_src.size());
          tl++;
        }
      }
      slots += (int64_t)tl * (int64_t)m;
      tg_ptr[p + 1] = (int)tg_loc.size();
    }
    for (int p = 0; p < K; p++)
      for (int t : tg_of_sn[p])
        if (!pre_contrib[t].empty()) {
          pt_dst.push_back(tg_dst_all[t]);
          for (int64_t q : pre_contrib[t]) {
            pub_dst[q] = (int)slots;
            pt_src.push_back(slots++);
          }
          pt_ptr.push_back((int)pt_src.size());
        }
    CXK_DEMAND(slots < (int64_t)INT32_MAX, "published-update slots exceed 32-bit indexing");
    CXK_TRY(ctx->tg_ptr.upload(tg_ptr));
    h_tg_ptr = tg_ptr;
    CXK_TRY(ctx->tg_loc.upload(tg_loc));
    tg_reg.resize(tg_reg.size() + kPullPad, 0);  // FactorSupernodeLean loads unconditionally (clamped)
    CXK_TRY(ctx->tg_reg.upload(tg_reg));
    CXK_TRY(ctx->tr_ptr.upload(tr_ptr));
    CXK_TRY(ctx->tr_src.upload(tr_src));
    CXK_TRY(ctx->pt_dst.upload(pt_dst));
    h_pt_dst = pt_dst;
    CXK_TRY(ctx->pt_ptr.upload(pt_ptr));
    CXK_TRY(ctx->pt_src.upload(pt_src));
  }
  std::vector<std::vector<int>> pre_fs_slots(N);
  {
    std::vector<int> fs_ptr(N + 1, 0), fs_src;
    for (int e = 0; e < K; e++) {
      size_t m = 0;
      for (int r = 0; r < ns[e]; r++) m = std::max(m, fs[start[e] + r].size());
      h_fbase[e] = slotsb;
      h_mf[e] = (int)m;
      slotsb += ns[e] * (int)m;
    }
    for (int e = 0; e < K; e++)
      for (int r = 0; r < ns[e]; r++) {
        const int p = start[e] + r;
        for (size_t i = 0; i < fs[p].size(); i++) pubb_dst[fs[p][i]] = h_fbase[e] + r * h_mf[e] + (int)i;
      }
    for (int p = 0; p < N; p++) {
      for (int q : fs[p]) fs_src.push_back(pubb_dst[q]);
      fs_ptr[p + 1] = (int)fs_src.size();
    }
    for (int p = 0; p < N; p++)
      for (int q : pre_fs[p]) {
        pubb_dst[q] = slotsb;
        pre_fs_slots[p].push_back(slotsb++);
      }
    CXK_TRY(ctx->fs_ptr.upload(fs_ptr));
    h_fs_ptr = fs_ptr;
    CXK_TRY(ctx->fs_src.upload(fs_src));
  }
  // values nobody on this rank consumes land in one dump slot at the end
  for (int& d : pub_dst)
    if (d < 0) d = (int)slots;
  for (int& d : pubb_dst)
    if (d < 0) d = slotsb;
  CXK_TRY(ctx->upd_off.upload(upd_off));
  CXK_TRY(ctx->updb_off.upload(updb_off));
  pub_dst.resize(pub_dst.size() + kPullPad, (int)slots);  // padding read (never used) by clamped loads
  pubb_dst.resize(pubb_dst.size() + kPullPad, slotsb);
  CXK_TRY(ctx->pub_dst.upload(pub_dst));
  CXK_TRY(ctx->pubb_dst.upload(pubb_dst));
  CXK_TRY(ctx->upd.alloc((size_t)slots + 1 + kPullPad, true));   // unused slots subtract 0.0
  CXK_TRY(ctx->updb.alloc((size_t)slotsb + 2 + kPullPad, true));  // + dump slot + a slot that stays 0.0

  // ---- exchange layout: [T slab entries | AW_T | AQc_T | fwd_T | <w,c> <c,Qc> fail]
  std::vector<int64_t> xs, xs_base(K, -1);  // (xs_base / xv_base: where a top supernode's entries / variables start)
  std::vector<int> xv, xv_base(K, -1);
  if (sharded) {
    std::vector<int> pf_ptr, pf_src;
    pf_ptr.push_back(0);
    for (int e = 0; e < K; e++) {
      if (!ctx->sn_top[e]) continue;
      xs_base[e] = (int64_t)xs.size();
      xv_base[e] = (int)xv.size();
      for (int j = 0; j < ns[e]; j++)
        for (int i2 = j; i2 < ns[e]; i2++) xs.push_back(L.diag_off[e] + (int64_t)j * ns[e] + i2);
      for (int64_t q = 0; q < (int64_t)ns[e] * nsep[e]; q++) xs.push_back(L.offd_off[e] + q);
      for (int r = 0; r < ns[e]; r++) {
        const int p = start[e] + r;
        xv.push_back(p);
        for (int q : pre_fs_slots[p]) pf_src.push_back(q);
        pf_ptr.push_back((int)pf_src.size());
      }
    }
    CXK_DEMAND(ctx->n_xs == (int64_t)xs.size() && ctx->n_xv == (int)xv.size(),
               "internal error: exchange layout mismatch");
    CXK_TRY(ctx->xs_off.upload(xs));
    {
      // exchange_pack folds this rank's own Schur updates into its partial top entries on the
      // way out: entry i of the exchange -> its list of published values (pt_ptr), or -1
      std::map<int64_t, int> list_of;
      for (size_t t = 0; t < h_pt_dst.size(); t++) list_of[h_pt_dst[t]] = (int)t;
      std::vector<int> xs_pt(xs.size() + 1, -1);
      size_t found = 0;
      for (size_t i = 0; i < xs.size(); i++) {
        auto it = list_of.find(xs[i]);
        if (it != list_of.end()) {
          xs_pt[i] = it->second;
          found++;
        }
      }
      CXK_DEMAND(found == h_pt_dst.size(), "internal error: a pre-contributed update targets an entry outside the exchange");
      CXK_TRY(ctx->xs_pt.upload(xs_pt));
    }
    CXK_TRY(ctx->xv_idx.upload(xv));
    CXK_TRY(ctx->pf_ptr.upload(pf_ptr));
    CXK_TRY(ctx->pf_src.upload(pf_src));
    CXK_TRY(ctx->xbuf.alloc((size_t)ctx->n_xs + 3 * (size_t)ctx->n_xv + 4));
    std::vector<unsigned char> count(N, 0);
    for (int p = 0; p < N; p++) {
      const int e = L.var_to_sn[p];
      count[p] = ctx->sn_top[e] ? (ctx->rank == 0) : (ctx->sn_mine[e] != 0);
    }
    CXK_TRY(ctx->d_count_mask.upload(count));
    CXK_TRY(ctx->shard_tmp.alloc(std::max<size_t>((size_t)N, 2 * ctx->cons.size())));
  }

  {
    // backward accumulation order: ancestors descending, columns ascending within one ancestor
    std::vector<int> bs_ptr(K + 1, 0), bs_c, bs_row;
    for (int j = 0; j < K; j++) {
      if (ns[j] > 0) {
        const IntList& s = L.separators[j];
        int hi = nsep[j];
        while (hi > 0) {
          const int anc = L.var_to_sn[s[hi - 1]];
          int lo = hi - 1;
          while (lo > 0 && L.var_to_sn[s[lo - 1]] == anc) lo--;
          for (int c = lo; c < hi; c++) {
            bs_c.push_back(c);
            bs_row.push_back(s[c]);
          }
          hi = lo;
        }
      }
      bs_ptr[j + 1] = (int)bs_c.size();
    }
    CXK_TRY(ctx->bs_ptr.upload(bs_ptr));
    h_bs_ptr = bs_ptr;
    h_bs_c = bs_c;
    h_bs_row = bs_row;
    CXK_TRY(ctx->bs_c.upload(bs_c));
    CXK_TRY(ctx->bs_row.upload(bs_row));
  }
  // level lists: supernodes with at least one column that this rank factors
  const int nlev = ctx->nlev;
  ctx->level_ptr.assign(nlev + 1, 0);
  ctx->level_sn.clear();
  ctx->chol_lds = 8;
  ctx->level_big.assign(nlev, 0);
  ctx->level_nh.assign(nlev, 0);
  size_t big_ws = 0;
  auto panel_bytes = [&](int e) {
    return sizeof(double) * ((size_t)ns[e] * ns[e] + (size_t)ns[e] * nsep[e] + 3 * (size_t)ns[e] + 2);
  };
  // (register shape, dense pulls, inline separator list) of a supernode: see cxk_context::LevelSeg
  auto seg_fast = [&](int e) {
    return h_tg_ptr[e + 1] - h_tg_ptr[e] <= kFastTargets && h_m[e] <= kFastSlots && h_mf[e] <= kFastSlots;
  };
  auto seg_inline = [&](int e) { return h_bs_ptr[e + 1] - h_bs_ptr[e] <= 8 && N < (1 << 26); };
  auto seg_key = [&](int e) { return std::make_tuple(RegisterShape(ns[e], nsep[e]), seg_fast(e), seg_inline(e)); };
  for (int l = 0; l < nlev; l++) {
    std::vector<int> huge;
    for (int e = 0; e < K; e++)
      if (ns[e] > 0 && ctx->t_level[e] == l && block_wanted(e)) {
        if (ns[e] > 32 || nsep[e] > 16) ctx->level_big[l] = 1;
        if (panel_bytes(e) > kLdsLimit) {
          huge.push_back(e);
          big_ws = std::max(big_ws, (size_t)nsep[e] * nsep[e] + nsep[e] + 1);
          if (ctx->use_ldlt) big_ws = std::max(big_ws, panel_bytes(e) / sizeof(double) + 2);  // the panel image of the LDLT kernel
          continue;
        }
        ctx->level_sn.push_back(e);
        ctx->chol_lds = std::max(ctx->chol_lds, panel_bytes(e));
        // register-shaped LDS image of FactorSupernodeLean: 64 lanes x NSMAX columns
        ctx->chol_lds = std::max(ctx->chol_lds, sizeof(double) * 64 * (size_t)(RegisterShape(ns[e], nsep[e]) >> 8));
      }
    ctx->level_nh[l] = (int)ctx->level_sn.size() - ctx->level_ptr[l];
    // segments: supernodes of a level are independent, so their order inside it is free
    std::stable_sort(ctx->level_sn.begin() + ctx->level_ptr[l], ctx->level_sn.end(),
                     [&](int a, int b) { return seg_key(a) < seg_key(b); });
    for (int e : huge) ctx->level_sn.push_back(e);
    ctx->level_ptr[l + 1] = (int)ctx->level_sn.size();
  }
  // Inside a segment, supernodes that read the solution of the same supernode of the level above
  // sit next to each other (tree_backward_pair hands them to one workgroup); top-down, so that
  // the level above already has its final order.
  {
    std::vector<int> pos_of(K, -1), key(K, 0);
    for (int l = nlev - 2; l >= 0; l--) {
      for (int pos = ctx->level_ptr[l + 1]; pos < ctx->level_ptr[l + 2]; pos++) pos_of[ctx->level_sn[pos]] = pos;
      const auto first = ctx->level_sn.begin() + ctx->level_ptr[l], last = first + ctx->level_nh[l];
      for (auto it = first; it != last; ++it) {
        int best = INT_MAX;
        for (int v : L.separators[*it]) {
          const int p = L.var_to_sn[v];
          if (ctx->t_level[p] == l + 1 && pos_of[p] >= 0) best = std::min(best, pos_of[p]);
        }
        key[*it] = best;
      }
      std::stable_sort(first, last, [&](int a, int b) {
        return std::make_tuple(seg_key(a), key[a]) < std::make_tuple(seg_key(b), key[b]);
      });
    }
  }
  if (big_ws > 0) {
    CXK_DEMAND(!sharded, "supernodes beyond LDS are single-GPU for now");
    CXK_TRY(ctx->big_ws.alloc(big_ws));
    // (CXK_NO_BIG_DATAFLOW=1: the host-driven panel loop, for comparison)
    if (!getenv("CXK_NO_BIG_DATAFLOW")) CXK_TRY(ctx->big_flags.alloc(2 * kBigCholMaxBlocks, true));
  }
  CXK_DEMAND(ctx->chol_lds <= kLdsLimit,
             "internal error: a supernode routed to the LDS kernels does not fit LDS");
  CXK_TRY(ctx->d_level_sn.upload(ctx->level_sn));
  CXK_TRY(ctx->d_level_ptr.upload(ctx->level_ptr));
  std::vector<SnRec> h_recs;
  {
    std::vector<SnRec> recs(ctx->level_sn.size());
    for (size_t pos = 0; pos < recs.size(); pos++) {
      const int e = ctx->level_sn[pos];
      SnRec& r = recs[pos];
      r.p = e;
      r.ns = ns[e];
      r.nsep = nsep[e];
      r.start = start[e];
      r.tg_beg = h_tg_ptr[e];
      r.tg_end = h_tg_ptr[e + 1];
      r.bs_beg = h_bs_ptr[e];
      r.bs_end = h_bs_ptr[e + 1];
      r.diag_off = L.diag_off[e];
      r.offd_off = L.offd_off[e];
      r.upd_off = upd_off[e];
      r.updb_off = updb_off[e];
      r.ubase = h_ubase[e];
      r.m = h_m[e];
      r.fbase = h_fbase[e];
      r.mf = h_mf[e];
      r.nsep_inline = 0;
      const int cnt = r.bs_end - r.bs_beg;
      if (cnt <= 8 && N < (1 << 26)) {
        r.nsep_inline = cnt;
        for (int q = 0; q < cnt; q++) r.sep[q] = h_bs_row[r.bs_beg + q] | (h_bs_c[r.bs_beg + q] << 26);
      }
    }
    CXK_TRY(ctx->p_rec.upload(recs));
    h_recs = recs;
    ctx->h_recs = recs;
    ctx->level_segs.assign(nlev, {});
    ctx->level_lean.assign(nlev, 0);
    for (int l = 0; l < nlev; l++) {
      const int first = ctx->level_ptr[l], last = first + ctx->level_nh[l];
      bool all = ctx->level_nh[l] == ctx->level_ptr[l + 1] - ctx->level_ptr[l] && last > first;
      for (int pos = first; pos < last; pos++) {
        const int e = ctx->level_sn[pos];
        auto& segs = ctx->level_segs[l];
        const int sh = RegisterShape(ns[e], nsep[e]);
        const bool fast = seg_fast(e), inl = seg_inline(e);
        if (segs.empty() || segs.back().shape != sh || segs.back().fast != fast || segs.back().inl != inl) {
          cxk_context::LevelSeg sg;
          sg.begin = pos;
          sg.shape = sh;
          sg.fast = fast;
          sg.inl = inl;
          segs.push_back(sg);
        }
        segs.back().end = pos + 1;
        all = all && sh > 0 && fast && inl;
      }
      ctx->level_lean[l] = all;
      if (getenv("CXK_DEBUG_LEVELS")) {
        fprintf(stderr, "level %d:", l);
        for (auto& sg : ctx->level_segs[l])
          fprintf(stderr, " [%d x <%d,%d>%s%s]", sg.end - sg.begin, sg.shape >> 8, sg.shape & 255, sg.fast ? " fast" : "", sg.inl ? " inline" : "");
        fprintf(stderr, " + %d beyond LDS\n", ctx->level_ptr[l + 1] - ctx->level_ptr[l] - ctx->level_nh[l]);
      }
    }
  }
  // narrow top of the tree: trailing levels that hold few supernodes are swept by one workgroup
  // (levels separated by a workgroup barrier instead of a kernel boundary); never below the cut
  {
    int top = nlev;
    ctx->no_lean = getenv("CXK_NO_LEAN") != nullptr;
    ctx->no_ranges = getenv("CXK_NO_RANGES") != nullptr;
    while (top > 0 && ctx->level_ptr[top] - ctx->level_ptr[top - 1] <= 8 && !ctx->level_big[top - 1]) top--;
    if (sharded) top = std::max(top, ctx->cut_level);
    if (ctx->use_ldlt) top = nlev;  // LDLT sweeps run level by level, one workgroup per supernode
    // A short top whose levels all have a shape-specialised kernel is swept level by level as
    // well: the lean per-level launches (one memory round trip per step) measured faster than the
    // generic one-workgroup sweep (C4: 24 -> 2 x (6.0 + 3.6) us).  Long narrow tops (chains) keep
    // the one-workgroup sweep: there a kernel boundary per step would dominate.
    if (!ctx->no_lean && !getenv("CXK_KEEP_TOP") && top < nlev && nlev - top <= kSplitTopLevels) {
      bool all = true;
      for (int l = top; l < nlev; l++) all = all && ctx->level_lean[l];
      if (all) top = nlev;
    }
    // A narrow top that is a pure chain -- one lean supernode per level, at most two shapes -- goes to
    // the chain kernel whatever its length (one launch of one wavefront, records prefetched, no
    // workgroup barriers): BASELINE config 3 as the reference arranges it is 5000 such levels.
    if (!ctx->use_ldlt && !ctx->no_lean && !getenv("CXK_NO_CHAIN") && !getenv("CXK_KEEP_TOP") && top < nlev) {
      int c0 = nlev, sa = 0, sb = 0;
      const int floor_level = sharded ? ctx->cut_level : 0;
      while (c0 > floor_level) {
        const int l = c0 - 1;
        if (ctx->level_ptr[l + 1] - ctx->level_ptr[l] != 1 || !ctx->level_lean[l]) break;
        const int sh = ctx->level_segs[l][0].shape;
        if (sa == 0 || sh == sa) {
          sa = sh;
        } else if (sb == 0 || sh == sb) {
          sb = sh;
        } else {
          break;
        }
        c0--;
      }
      if (c0 <= top && nlev - c0 >= 2 && ChainPairCompiled(sa, sb)) top = nlev;
    }
    ctx->top_level = top;
    // chain at the top (single GPU, Cholesky, top swept level by level)
    ctx->chain_level = nlev;
    if (!ctx->use_ldlt && !ctx->no_lean && top == nlev && !getenv("CXK_NO_CHAIN")) {
      int c0 = nlev, sa = 0, sb = 0;
      const int floor_level = sharded ? ctx->cut_level : 0;  // the chain stays inside the replicated top
      while (c0 > floor_level && nlev - c0 < kChainMaxLevels) {
        const int l = c0 - 1;
        if (ctx->level_ptr[l + 1] - ctx->level_ptr[l] != 1 || !ctx->level_lean[l]) break;
        const int sh = ctx->level_segs[l][0].shape;
        if (sa == 0 || sh == sa) {
          sa = sh;
        } else if (sb == 0 || sh == sb) {
          sb = sh;
        } else {
          break;
        }
        c0--;
      }
      if (sb != 0 && sb < sa) std::swap(sa, sb);
      if (nlev - c0 >= 2 && ChainPairCompiled(sa, sb)) {
        ctx->chain_level = c0;
        ctx->chain_a = sa;
        ctx->chain_b = sb == 0 ? sa : sb;
      }
    }
    // pairs of downward levels, from the leaves up: both one lean segment, every lower supernode
    // reads at most one supernode of the upper level, and those that read the same one are consecutive
    ctx->back_pairs.clear();
    ctx->back_pairs.resize(nlev);
    if (!ctx->use_ldlt && !ctx->no_lean && top == nlev && !getenv("CXK_NO_BACK_PAIRS")) {
      const int up_end = ctx->chain_level < nlev ? ctx->chain_level : nlev;
      auto plain = [&](int l) {
        return ctx->level_lean[l] && !ctx->level_big[l] && ctx->level_segs[l].size() == 1 &&
               ctx->level_nh[l] == ctx->level_ptr[l + 1] - ctx->level_ptr[l];
      };
      std::vector<int> pos_of(K, -1);
      for (int l = 0; l + 1 < up_end;) {
        bool ok = plain(l) && plain(l + 1);
        std::vector<BackPairEntry> tab;
        if (ok) {
          for (int pos = ctx->level_ptr[l + 1]; pos < ctx->level_ptr[l + 2]; pos++) pos_of[ctx->level_sn[pos]] = pos;
          std::vector<int> dep(ctx->level_ptr[l + 1] - ctx->level_ptr[l], -1);
          for (int pos = ctx->level_ptr[l]; pos < ctx->level_ptr[l + 1] && ok; pos++) {
            int d = -1;
            for (int v : L.separators[ctx->level_sn[pos]]) {
              const int p = L.var_to_sn[v];
              if (ctx->t_level[p] != l + 1) continue;
              if (pos_of[p] < 0 || (d >= 0 && d != pos_of[p])) ok = false;
              d = pos_of[p];
            }
            dep[pos - ctx->level_ptr[l]] = d;
          }
          // runs of equal dependence; a parent's children must form ONE run
          std::vector<char> seen(ctx->level_ptr[l + 2] - ctx->level_ptr[l + 1], 0);
          for (int i = 0; i < (int)dep.size() && ok;) {
            int j = i;
            while (j < (int)dep.size() && dep[j] == dep[i]) j++;
            if (dep[i] >= 0) {
              char& sn = seen[dep[i] - ctx->level_ptr[l + 1]];
              if (sn) ok = false;
              sn = 1;
              tab.push_back(BackPairEntry{dep[i], ctx->level_ptr[l] + i, j - i, 0});
            } else {
              for (int q = i; q < j; q += 8) tab.push_back(BackPairEntry{-1, ctx->level_ptr[l] + q, std::min(8, j - q), 0});
            }
            i = j;
          }
          // a supernode of the upper level nobody below reads (cannot happen by the definition of a
          // level; kept for safety): solved by a workgroup without children
          for (size_t q = 0; q < seen.size() && ok; q++)
            if (!seen[q]) tab.push_back(BackPairEntry{ctx->level_ptr[l + 1] + (int)q, ctx->level_ptr[l], 0, 0});
        }
        if (ok) {
          auto bp = std::make_unique<cxk_context::BackPair>();
          bp->nwg = (int)tab.size();
          bp->shape_p = ctx->level_segs[l + 1][0].shape;
          bp->shape_c = ctx->level_segs[l][0].shape;
          CXK_TRY(bp->tab.upload(tab));
          if (getenv("CXK_DEBUG_LEVELS")) fprintf(stderr, "backward pair: levels %d + %d in %d workgroups\n", l + 1, l, bp->nwg);
          ctx->back_pairs[l + 1] = std::move(bp);
          l += 2;
        } else {
          l += 1;
        }
      }
    }
  }
  // ---- solve-only sweeps: does every forward launch run a lean kernel (then the right-hand side is
  // formed inside them, RhsIn)?  Levels below the chain must be all-lean single launches, the
  // rest must be the chain (no one-workgroup top, no supernode beyond LDS).
  {
    bool all = !sharded && !ctx->use_ldlt && !ctx->no_lean && ctx->top_level == nlev;
    const int up_end = ctx->chain_level < nlev ? ctx->chain_level : nlev;
    for (int l = 0; l < up_end && all; l++) {
      all = ctx->level_lean[l] && !ctx->level_big[l] &&
            ctx->level_nh[l] == ctx->level_ptr[l + 1] - ctx->level_ptr[l] && ctx->level_segs[l].size() <= 2;
    }
    ctx->forward_all_lean = all && nlev >= 1;
  }

  // ---- assembly folded into the first factor level.  Taken when level 0 is ONE segment of a
  // register shape with dense pulls, launched on its own (not part of a chain / dense top), and
  // every supernode in it is a leaf whose panel entries and right-hand-side rows have exactly one
  // source each, all in the Schur block of its own constraint, at positions pos[row] -- the
  // leaves of a clique tree.  Those supernodes then read G(max(pos_r, pos_c), min(..)) themselves
  // and the gather lists shrink to what the levels above need.
  ctx->fused_asm = false;
  if ((!sharded || ctx->cut_level >= 1) && !ctx->use_ldlt && !ctx->no_lean && !getenv("CXK_NO_FUSED_ASM") && nlev >= 2 &&
      (ctx->level_segs[0].size() == 1 || ctx->level_segs[0].size() == 2) && ctx->level_lean[0] &&
      ctx->top_level >= 1 && ctx->chain_level >= 1 && ctx->level_segs[0][0].shape != 0 &&
      4 * ctx->chol_lds <= kLdsLimit) {
    const int first = ctx->level_ptr[0], cnt0 = ctx->level_nh[0];
    std::vector<AsmRec> arecs(cnt0);
    std::vector<char> slab_own(h_as_rec.size(), 0), var_own(N, 0);
    bool ok = cnt0 > 0 && cnt0 == ctx->level_ptr[1] - first;
    for (int q = 0; q < cnt0 && ok; q++) {
      const int e = ctx->level_sn[first + q];
      const int i = md.clique_order[e];
      const int m = ctx->cons[i].m;
      const IntList& r = md.supernodes_pos[e];
      const IntList& sp = md.separators_pos[e];
      const int nse = (int)r.size(), nsp = (int)sp.size();
      ok = nse == ns[e] && nsp == nsep[e] && nse + nsp <= 72 && m <= 255 && ctx->owned[i] &&
           h_tg_ptr[e + 1] == h_tg_ptr[e] && h_mf[e] == 0;
      AsmRec& ar = arecs[q];
      memset(&ar, 0, sizeof(ar));
      ar.g_off = ctx->g_off[i];
      ar.r_off = ctx->r_off[i];
      ar.m = m;
      for (int a = 0; a < nse && ok; a++) {
        ok = r[a] >= 0 && r[a] < m;
        ar.pos[a] = (unsigned char)r[a];
      }
      for (int a = 0; a < nsp && ok; a++) {
        ok = sp[a] >= 0 && sp[a] < m;
        ar.pos[nse + a] = (unsigned char)sp[a];
      }
      auto single = [&](int64_t off, int pa, int pb) {  // the slab entry has the one source G(pa, pb)
        const int t = entry_of[off];
        if (t < 0) return false;
        const GatherRec& g = h_as_rec[t];
        const int hi = std::max(pa, pb), lo = std::min(pa, pb);
        if (g.extra != 0 || g.first != ar.g_off + hi + (int64_t)lo * m) return false;
        slab_own[t] = 1;
        return true;
      };
      for (int j = 0; j < nse && ok; j++)
        for (int i2 = j; i2 < nse && ok; i2++) ok = single(L.diag_off[e] + (int64_t)j * nse + i2, r[i2], r[j]);
      for (int j = 0; j < nsp && ok; j++)
        for (int i2 = 0; i2 < nse && ok; i2++) ok = single(L.offd_off[e] + (int64_t)j * nse + i2, r[i2], sp[j]);
      for (int a = 0; a < nse && ok; a++) {
        const int pvar = start[e] + a;
        ok = per[pvar].size() == 1 && per[pvar][0] == ar.r_off + r[a];
        var_own[pvar] = 1;
      }
    }
    if (ok) {
      std::vector<GatherRec> g2;
      for (size_t t = 0; t < h_as_rec.size(); t++)
        if (!slab_own[t]) g2.push_back(h_as_rec[t]);
      std::vector<ResidRec> r2;
      std::vector<int> v2;
      for (int pvar = 0; pvar < N; pvar++)
        if (!var_own[pvar]) {
          r2.push_back(h_rs_rec[pvar]);
          v2.push_back(pvar);
        }
      ctx->as_T2 = (int64_t)g2.size();
      ctx->rs_N2 = (int)v2.size();
      if (g2.empty()) g2.push_back(GatherRec{0, -1, 0, 0});
      if (v2.empty()) {
        r2.push_back(ResidRec{-1, 0, 0});
        v2.push_back(0);
      }
      CXK_TRY(ctx->asm_rec.upload(arecs));
      CXK_TRY(ctx->as_rec2.upload(g2));
      CXK_TRY(ctx->rs_rec2.upload(r2));
      CXK_TRY(ctx->rs_var2.upload(v2));
      ctx->fused_asm = true;
    }
  }
  // ---- the whole tree in one launch (tree_fused.hip).  Taken when every supernode has a register
  // kernel (at most two shapes) with dense pulls and an inline separator list, sits alone in its
  // constraint's Schur block at non-negative positions (no fill-in rows), its entries take their
  // first source from that block, the lists of further sources fit the dense slots, and the grid
  // is resident at once (the way back down waits for HIGHER positions).
  ctx->fused_tree = false;
  ctx->fused_sweep = false;
  ctx->fused_shard = false;
  // (sharded contexts: the own subtrees up to the cut and, behind the exchange, the replicated top and
  // the way back down -- two launches, tree_fused.h; CXK_NO_FUSED_SHARD=1 keeps the level kernels there)
  if ((!sharded || (!getenv("CXK_NO_FUSED_SHARD") && ctx->cut_level >= 1 && ctx->cut_level < nlev)) && !ctx->use_ldlt &&
      !ctx->no_lean && !getenv("CXK_NO_FUSED_TREE") && nlev >= 1 && N < (1 << 26)) {
    const int cnt_all = (int)ctx->level_sn.size();
    const int cnt_up = sharded ? ctx->level_ptr[ctx->cut_level] : cnt_all;  // positions below the cut
    bool ok = cnt_all > 0 && cnt_all == ctx->level_ptr[nlev];
    const char* why = ok ? nullptr : "a supernode without columns / beyond LDS";
    auto note = [&](const char* msg) {
      if (!ok && !why) why = msg;
    };
    int sa = 0, sb = 0;
    // a program that is ONE dense supernode of 33 .. 64 columns (BASELINE config 2: 50): the wide
    // instances of the same launch (tree_fused.hip, ElimWide)
    const bool wide_single = ok && !sharded && cnt_all == 1 && K >= 1 && ns[ctx->level_sn[0]] > 32 && ns[ctx->level_sn[0]] <= 64 &&
                             nsep[ctx->level_sn[0]] == 0 && !getenv("CXK_NO_FUSED_WIDE");
    if (wide_single) {
      sa = sb = ((ns[ctx->level_sn[0]] + 7) / 8 * 8) << 8;
    } else {
      // at most two register shapes; a shape without separator columns <N, 0> runs on <N, S> where
      // the tree has one (same rows per lane: the pull locations tg_reg are the same)
      std::vector<int> shapes;
      for (int l = 0; l < nlev && ok; l++) {
        if (ctx->level_ptr[l + 1] == ctx->level_ptr[l]) continue;  // (a level this rank has no supernode on)
        ok = !ctx->level_big[l] && ctx->level_nh[l] == ctx->level_ptr[l + 1] - ctx->level_ptr[l];
        // (the whole-tree kernels take pull lists of any length up to kFusedMaxSlots, kFastSlots at a
        // time: a level needs a register shape and inline separator lists, not the level kernels' "fast")
        for (auto& sg : ctx->level_segs[l]) {
          ok = ok && sg.shape != 0 && sg.inl;
          if (std::find(shapes.begin(), shapes.end(), sg.shape) == shapes.end()) shapes.push_back(sg.shape);
        }
        for (int pos = ctx->level_ptr[l]; pos < ctx->level_ptr[l + 1] && ok; pos++) {
          const int e = ctx->level_sn[pos];
          ok = h_tg_ptr[e + 1] - h_tg_ptr[e] <= kFastTargets && h_m[e] <= kFusedMaxSlots && h_mf[e] <= kFusedMaxSlots;
        }
      }
      for (size_t i = 0; i < shapes.size(); i++)
        if ((shapes[i] & 255) == 0)
          for (size_t j = 0; j < shapes.size(); j++)
            if (j != i && shapes[i] >= 0 && (shapes[j] >> 8) == (shapes[i] >> 8) && (shapes[j] & 255) > 0) {
              shapes[i] = -1;
              break;
            }
      shapes.erase(std::remove(shapes.begin(), shapes.end(), -1), shapes.end());
      std::sort(shapes.begin(), shapes.end());
      note("a level without a register shape, inline separator lists or within the slot limits");
      ok = ok && !shapes.empty() && shapes.size() <= 2;
      note("more than two register shapes");
      if (ok) {
        sa = shapes[0];
        sb = shapes.back();
      }
    }
    // (a tree that is one long chain keeps the chain kernel: one wavefront, no hand-offs)
    ok = ok && (nlev <= 64 || cnt_all >= 4 * nlev);
    note("a long chain");
    ok = ok && FusedTreeCompiled(sa, sb);
    note("no instance for the pair of shapes");
    std::vector<int> recs((size_t)cnt_all * kFusedRecWords, 0), xreg;
    std::vector<long long> xsrc, rsrc;
    for (int pos = 0; pos < cnt_all && ok; pos++) {
      const int e = ctx->level_sn[pos];
      const int i = md.clique_order[e];
      const int m = ctx->cons[i].m;
      const IntList& r = md.supernodes_pos[e];
      const IntList& sp = md.separators_pos[e];
      const int nse = (int)r.size(), nsp = (int)sp.size();
      const int nsm = wide_single ? sa >> 8 : RegisterShape(ns[e], nsep[e]) >> 8;
      int* w = recs.data() + (size_t)pos * kFusedRecWords;
      if (pos >= cnt_up) {
        // the replicated top of a sharded context: the panel comes from the exchange buffer
        ok = ctx->sn_top[e] && nsm > 0 && xs_base[e] >= 0 && ctx->n_xs < (int64_t)INT32_MAX;
        note("a top supernode without a register kernel");
        if (!ok) break;
        memcpy(w, &h_recs[pos], sizeof(SnRec));
        w[32] = (int)(xs_base[e] & 0xffffffffll);
        w[33] = (int)(xs_base[e] >> 32);
        w[34] = xv_base[e];
        w[63] = ctx->t_level[e];
        continue;
      }
      ok = nse == ns[e] && nsp == nsep[e] && nse + nsp <= 72 && m <= 254 && ctx->owned[i] && nsm > 0;
      note("a supernode that is not its constraint's own block");
      if (!ok) break;
      memcpy(w, &h_recs[pos], sizeof(SnRec));
      AsmRec ar;
      memset(&ar, 0, sizeof(ar));
      ar.g_off = ctx->g_off[i];
      ar.r_off = ctx->r_off[i];
      ar.m = m;
      for (int a = 0; a < nse && ok; a++) {
        ok = r[a] >= 0 && r[a] < m;
        ar.pos[a] = (unsigned char)r[a];
      }
      note("a fill-in variable among the supernode's own (position -1)");
      // separator rows the constraint does not contain (structural fill: the deferred variables of a
      // segmented chain): position 255, entries that start as zeros (AsmRec::pad_ flags the record)
      for (int a = 0; a < nsp && ok; a++) {
        ok = sp[a] < m;
        ar.pos[nse + a] = sp[a] < 0 ? (unsigned char)255 : (unsigned char)sp[a];
        if (sp[a] < 0) ar.pad_ = 1;
      }
      if (!ok) break;
      memcpy(w + 32, &ar, sizeof(AsmRec));
      // entries with further sources, in the order of the panel (columns of the diagonal block, then
      // the off block): (image location, sources)
      std::vector<std::pair<int, std::vector<int64_t>>> extra;
      auto visit = [&](int64_t off, int pa, int pb, int reg) {
        const int t = entry_of[off];
        if (t < 0) return false;
        const GatherRec& g = h_as_rec[t];
        const int hi = std::max(pa, pb), lo = std::min(pa, pb);
        if (pa < 0 || pb < 0) {  // a fill-in row: no source in the own block
          if (g.first >= 0) return false;
        } else if (g.first != ar.g_off + hi + (int64_t)lo * m) {
          return false;
        }
        if (g.extra > 0) {
          extra.emplace_back(reg, std::vector<int64_t>(as_src.begin() + g.beg, as_src.begin() + g.beg + g.extra));
        }
        return true;
      };
      for (int j = 0; j < nse && ok; j++)
        for (int i2 = j; i2 < nse && ok; i2++) ok = visit(L.diag_off[e] + (int64_t)j * nse + i2, r[i2], r[j], 64 * j + i2);
      for (int j = 0; j < nsp && ok; j++)
        for (int i2 = 0; i2 < nse && ok; i2++) ok = visit(L.offd_off[e] + (int64_t)j * nse + i2, r[i2], sp[j], 64 * i2 + nsm + j);
      note("an entry whose first source is not the own block");
      size_t mx = 0;
      for (auto& x : extra) mx = std::max(mx, x.second.size());
      ok = ok && extra.size() <= (size_t)kFusedExtraTargets && mx <= (size_t)kFusedExtraMax;
      note("too many entries with further sources / too many sources");
      // variables that several constraints share: all their sources, in the gather's order
      size_t mr = 0;
      for (int a = 0; a < nse && ok; a++) {
        const auto& lst = per[start[e] + a];
        if (lst.size() == 1)
          ok = lst[0] == ar.r_off + r[a];
        else
          ok = !lst.empty() && std::find(lst.begin(), lst.end(), ar.r_off + r[a]) != lst.end();
        if (lst.size() > 1) mr = std::max(mr, lst.size());
      }
      note("a variable whose sources do not include the own constraint");
      ok = ok && mr <= (size_t)kFusedExtraMax;
      note("a variable shared by more than 64 constraints");
      if (!ok) break;
      const int64_t xbase = (int64_t)xsrc.size();
      w[56] = (int)xreg.size();
      w[57] = (int)extra.size();
      w[58] = (int)mx;
      w[59] = (int)rsrc.size();
      w[60] = (int)mr;
      w[61] = (int)(xbase & 0xffffffffll);
      w[62] = (int)(xbase >> 32);
      w[63] = ctx->t_level[e];
      for (auto& x : extra) {
        xreg.push_back(x.first);
        for (size_t q2 = 0; q2 < mx; q2++) xsrc.push_back(q2 < x.second.size() ? (long long)x.second[q2] : -1ll);
      }
      if (mr > 0)
        for (int a = 0; a < nse; a++) {
          const auto& lst = per[start[e] + a];
          for (size_t q2 = 0; q2 < mr; q2++) rsrc.push_back(lst.size() > 1 && q2 < lst.size() ? (long long)lst[q2] : -1ll);
        }
      ok = rsrc.size() < (size_t)INT32_MAX && xreg.size() < (size_t)INT32_MAX;
    }
    // a supernode's published values go to the supernodes that own its separator variables: all of
    // them must sit at higher positions (waits go to lower positions on the way up)
    std::vector<int> pub;
    const size_t us = (size_t)slots + 1 + kPullPad, ubs = (size_t)slotsb + 2 + kPullPad;
    if (ok) {
      std::vector<int> pos_of(K, -1);
      for (int pos = 0; pos < cnt_all; pos++) pos_of[ctx->level_sn[pos]] = pos;
      for (int pos = 0; pos < cnt_all && ok; pos++) {
        const int e = ctx->level_sn[pos];
        for (int v : L.separators[e]) ok = ok && pos_of[L.var_to_sn[v]] > pos;
        note("a consumer at a lower position");
      }
      ok = ok && us + ubs + 8 < (size_t)INT32_MAX;
      for (int pos = 0; pos < cnt_all && ok; pos++) {
        const int e = ctx->level_sn[pos];
        int* w = recs.data() + (size_t)pos * kFusedRecWords;
        w[21] = w[22] = 0;
        w[23] = (int)pub.size();
        for (int64_t t = 0; t < (int64_t)nsep[e] * (nsep[e] + 1) / 2; t++) pub.push_back(pub_dst[(size_t)(upd_off[e] + t)]);
        for (int c = 0; c < nsep[e]; c++) pub.push_back((int)us + pubb_dst[(size_t)(updb_off[e] + c)]);
      }
    }
    bool split = false;
    if (ok) {
      // residency: every workgroup (one wavefront each, one more for the scalars) at once, with a
      // CU's worth of margin per slot count the occupancy query may overstate; a larger tree takes
      // the way up and the way down as two launches (tree_fused.h, FusedTreeMode)
      const int occ = FusedTreeOccupancy(sa, sb, sharded);
      ok = occ >= 2;
      note("occupancy query failed");
      split = (int64_t)cnt_all + 1 > (int64_t)(occ - 1) * ctx->cus || getenv("CXK_FUSED_SPLIT") != nullptr;
      if (sharded) {
        // only the top has to be resident at once (tree_fused.h, kFusedShardTop)
        ok = ok && (int64_t)(cnt_all - cnt_up) + 1 <= (int64_t)(occ - 1) * ctx->cus;
        note("the replicated top exceeds the resident wavefronts");
        split = false;
      }
    }
    if (ok && sharded) {
      // pack tables: per exchange entry / top variable the record of its own-rank sources
      std::vector<GatherRec> xg(xs.size() + 1, GatherRec{0, -1, 0, 0});
      for (size_t t = 0; t < xs.size() && ok; t++) {
        const int et = entry_of[xs[t]];
        ok = et >= 0;
        if (ok) xg[t] = h_as_rec[et];
      }
      note("a top entry without a gather record");
      std::vector<ResidRec> xr(xv.size() + 1, ResidRec{-1, 0, 0});
      for (size_t j = 0; j < xv.size(); j++) xr[j] = h_rs_rec[xv[j]];
      if (ok) {
        CXK_TRY(ctx->fx_xg.upload(xg));
        CXK_TRY(ctx->fx_xr.upload(xr));
        CXK_TRY(ctx->fx_done.alloc(64 * 16, true));  // 64 counters, 128 bytes apart
        ctx->fx_done_target = 0;
      }
    }
    if (ok) {
      xreg.resize(xreg.size() + kPullPad, 0);
      xsrc.resize(xsrc.size() + kPullPad * kFusedExtraMax, -1ll);
      rsrc.resize(rsrc.size() + 64 * kFusedExtraMax, -1ll);
      pub.resize(pub.size() + 64, (int)slots);
      // hand-off slots: every slot with a producer starts as the sentinel in BOTH sets, the rest 0.0
      const double sent = [] {
        double d;
        const unsigned long long bits = kFusedSentinel;
        memcpy(&d, &bits, sizeof(d));
        return d;
      }();
      // (forward-value slots three times over: right-hand sides 1 and 2 of a launch with three, kFusedTriple)
      const size_t hs = us + 3 * ubs + 8;
      ctx->fx_updb_base = (long long)us;
      ctx->fx_fwd_stride = (long long)ubs;
      ctx->fx_hand_init.assign(2 * hs, 0.0);
      for (size_t t = 0; t + 64 < pub.size(); t++) {
        const int d = pub[t];
        if (d != (int)slots && d != (int)us + slotsb) {
          ctx->fx_hand_init[d] = ctx->fx_hand_init[hs + d] = sent;
          if ((size_t)d >= us)
            for (size_t q = 1; q < 3; q++) ctx->fx_hand_init[d + q * ubs] = ctx->fx_hand_init[hs + d + q * ubs] = sent;
        }
      }
      if (!split && !sharded && !getenv("CXK_FUSED_LEVEL_ORDER")) {
        // Which workgroup takes which supernode.  Every wavefront of the launch is resident (no order is needed
        // for progress) and the dispatcher deals workgroups round-robin over the 8 XCDs (workgroup b on XCD
        // b mod 8: tools/xcc_placement_bench.hip -- a speed assumption only).  A hand-off between wavefronts of
        // one XCD is 0.1 - 0.3 us shorter than one across the fabric (MI355X_MICROARCH.md, handoff-1to1), and
        // the launch is nine hand-offs deep: the supernodes are dealt in depth-first order of the tree, an
        // eighth of them per XCD, so that a supernode mostly sits with its children; within an XCD the level
        // order stays (leaves first: they have the most to load).
        std::vector<int> pos_of(K, -1), parent(cnt_all, -1);
        for (int pos = 0; pos < cnt_all; pos++) pos_of[ctx->level_sn[pos]] = pos;
        std::vector<std::vector<int>> kids(cnt_all);
        std::vector<int> roots;
        for (int pos = 0; pos < cnt_all; pos++) {
          int par = INT32_MAX;
          for (int v : L.separators[ctx->level_sn[pos]]) par = std::min(par, pos_of[L.var_to_sn[v]]);
          if (par == INT32_MAX)
            roots.push_back(pos);
          else
            kids[par].push_back(pos);
        }
        std::vector<int> dfs, stack(roots.rbegin(), roots.rend());
        dfs.reserve(cnt_all);
        while (!stack.empty()) {
          const int u = stack.back();
          stack.pop_back();
          dfs.push_back(u);
          for (auto it = kids[u].rbegin(); it != kids[u].rend(); ++it) stack.push_back(*it);
        }
        if ((int)dfs.size() == cnt_all) {
          std::vector<int> perm(cnt_all, -1);  // workgroup -> position in level order
          size_t at = 0;
          for (int x = 0; x < 8; x++) {
            const int n = (cnt_all - x + 7) / 8;  // workgroups x, x + 8, ... below cnt_all
            std::vector<int> mine(dfs.begin() + at, dfs.begin() + at + n);
            at += n;
            std::sort(mine.begin(), mine.end());
            for (int i = 0; i < n; i++) perm[8 * i + x] = mine[i];
          }
          std::vector<int> dealt(recs.size());
          for (int b = 0; b < cnt_all; b++)
            memcpy(dealt.data() + (size_t)b * kFusedRecWords, recs.data() + (size_t)perm[b] * kFusedRecWords, sizeof(int) * kFusedRecWords);
          recs.swap(dealt);
        }
      }
      CXK_TRY(ctx->fx_rec.upload(recs));
      CXK_TRY(ctx->fx_xreg.upload(xreg));
      CXK_TRY(ctx->fx_xsrc.upload(xsrc));
      CXK_TRY(ctx->fx_rsrc.upload(rsrc));
      CXK_TRY(ctx->fx_pub.upload(pub));
      CXK_TRY(ctx->fx_hand.upload(ctx->fx_hand_init));
      CXK_TRY(ctx->fx_ysig.upload(std::vector<double>(6 * (size_t)N, sent)));  // two sets x three right-hand sides
      CXK_TRY(ctx->y3.alloc(3 * (size_t)N, true));
      ctx->fused_tgen = 0;
      ctx->y3_valid = false;
      if (!ctx->fx_flag) {
        CXK_TRY(hipHostMalloc(reinterpret_cast<void**>(&ctx->fx_flag), 64, hipHostMallocDefault));
        *ctx->fx_flag = 0.0;
      }
      ctx->fused_sa = sa;
      ctx->fused_sb = sb;
      ctx->fused_gen = 0;
      ctx->fused_tree = true;
      ctx->fused_split = split;
      ctx->fused_shard = sharded;
      ctx->fused_up = cnt_up;
      // (solve-only sweeps of a sharded context keep the level kernels and their own small exchange)
      ctx->fused_sweep = !sharded && getenv("CXK_NO_FUSED_SWEEP") == nullptr;
      if (getenv("CXK_DEBUG_LEVELS"))
        fprintf(stderr, "whole tree in %s: %d supernodes, shapes <%d,%d> <%d,%d>, %zu entries / %zu variables with further sources\n",
                sharded ? "two launches around the exchange (own subtrees up + pack, top + down)" : split ? "two launches (up, down)" : "one launch", cnt_all, sa >> 8, sa & 255, sb >> 8, sb & 255,
                xreg.size() - kPullPad, rsrc.size());
    } else if (getenv("CXK_DEBUG_LEVELS")) {
      fprintf(stderr, "whole-tree launch not taken: %s\n", why ? why : "(unnamed check)");
    }
  }
  // ---- the top as one dense T x T factorization (single GPU, Cholesky): tables for
  // tree_top_dense.  Rows = the variables of the top supernodes in elimination order.
  // Used where the supernode-by-supernode kernels are weak: when the last levels hold a mid-size
  // supernode (33..64 columns: otherwise a 256-thread workgroup with two barriers per column,
  // 62 us for C2's 50-column root).  The dense range [dense_level, nlev) starts at such a level;
  // for tops made of small supernodes (C4) the supernode-by-supernode top measured faster.
  ctx->top_dense.on = false;
  ctx->dense_level = nlev;
  if (!sharded && !ctx->use_ldlt && !getenv("CXK_NO_TOP_DENSE")) {
    int dt = -1;
    {
      int cols = 0, count = 0;
      bool clean = true;
      for (int l = nlev - 1; l >= 0 && clean; l--) {
        if (ctx->level_nh[l] != ctx->level_ptr[l + 1] - ctx->level_ptr[l]) break;  // a panel beyond LDS
        for (int pos = ctx->level_ptr[l]; pos < ctx->level_ptr[l + 1]; pos++) {
          cols += ns[ctx->level_sn[pos]];
          count++;
        }
        if (cols > kTopMaxCols || count > kTopMaxSn) break;
        if (ctx->level_big[l]) dt = l;
      }
    }
    std::vector<int> tsn;
    int T = 0;
    bool ok = dt >= 0;
    if (ok) {
      for (int pos = ctx->level_ptr[dt]; pos < ctx->level_ptr[nlev]; pos++) tsn.push_back(ctx->level_sn[pos]);
      std::sort(tsn.begin(), tsn.end());
      for (int e : tsn) T += ns[e];
    }
    if (ok && !tsn.empty() && (int)tsn.size() <= kTopMaxSn && T <= kTopMaxCols && T > 0) {
      TopDenseArgs& a = ctx->top_dense.args;
      a.nt = (int)tsn.size();
      a.T = T;
      std::vector<int> is_top(K, -1), vrow(N, -1);
      int row = 0, base = 0;
      for (int k = 0; k < a.nt; k++) {
        const int e = tsn[k];
        is_top[e] = k;
        a.ns[k] = ns[e];
        a.nsep[k] = nsep[e];
        a.start[k] = start[e];
        a.row0[k] = row;
        a.base[k] = base;
        a.diag_off[k] = L.diag_off[e];
        a.offd_off[k] = L.offd_off[e];
        for (int i2 = 0; i2 < ns[e]; i2++) vrow[start[e] + i2] = row + i2;
        row += ns[e];
        base += ns[e] * ns[e] + ns[e] * nsep[e];
      }
      for (int k = 0; k < a.nt && ok; k++)  // separators of the top stay inside the top
        for (int v : L.separators[tsn[k]])
          if (vrow[v] < 0) ok = false;
      if (ok && base <= kTopMaxCols * kTopMaxCols) {
        std::vector<int> off((size_t)T * T, -1);
        for (int k = 0; k < a.nt; k++) {
          const int e = tsn[k], n = ns[e];
          for (int jl = 0; jl < n; jl++) {
            const int j = a.row0[k] + jl;
            for (int rl = jl; rl < n; rl++) off[(size_t)(a.row0[k] + rl) * T + j] = a.base[k] + rl + jl * n;
            const IntList& sp = L.separators[e];
            for (int c = 0; c < (int)sp.size(); c++) off[(size_t)vrow[sp[c]] * T + j] = a.base[k] + n * n + jl + c * n;
          }
        }
        // Updates from below the top arrive through the supernodes' consumer-ordered slots; slots
        // fed from inside the top are never written in this mode (they hold 0.0).  Forward-solve
        // values use explicit fixed-width lists (the forward-only sweeps do write the inner slots).
        std::vector<int64_t> updb_off64(updb_off.begin(), updb_off.end());
        auto producer = [&](const std::vector<int64_t>& offs, int64_t q) {
          return (int)(std::upper_bound(offs.begin(), offs.end(), q) - offs.begin()) - 1;
        };
        int u_lds = 0, t_lds = 0;
        std::vector<int> rhs_src((size_t)T * kTopRhsSrc, slotsb + 1);  // slotsb + 1: never written, 0.0
        for (int k = 0; k < a.nt && ok; k++) {
          const int e = tsn[k];
          a.ubase[k] = (int)h_ubase[e];
          a.m[k] = h_m[e];
          a.tg_beg[k] = h_tg_ptr[e];
          a.ntg[k] = h_tg_ptr[e + 1] - h_tg_ptr[e];
          a.ubase_lds[k] = u_lds;
          a.tg_lds[k] = t_lds;
          u_lds += a.ntg[k] * a.m[k];
          t_lds += a.ntg[k];
          for (int i2 = 0; i2 < ns[e]; i2++) {
            int cnt2 = 0;
            for (int q : fs[start[e] + i2])
              if (is_top[producer(updb_off64, q)] < 0) {
                if (cnt2 == kTopRhsSrc) {
                  ok = false;
                  break;
                }
                rhs_src[(size_t)(a.row0[k] + i2) * kTopRhsSrc + cnt2++] = pubb_dst[q];
              }
          }
        }
        if (!ok || u_lds > kTopMaxImage || t_lds > kTopMaxImage) {
          ok = false;
        } else {
          CXK_TRY(ctx->top_dense.off.upload(off));
          CXK_TRY(ctx->top_dense.pl_src.upload(rhs_src));
          a.top_off = ctx->top_dense.off.p;
          a.rhs_src = ctx->top_dense.pl_src.p;
        }
        if (ok) CXK_TRY(RaiseTopDenseLimits());
        if (ok) {
          ctx->top_dense.on = true;
          ctx->dense_level = dt;
        }
      }
    }
  }
  // ---- level ranges below the top: merge consecutive levels into one launch when every
  // connected piece of the forest restricted to them fits one workgroup (<= 8 supernodes per
  // level: one wavefront each, and <= kRangeMaxRecs records for the LDS prefetch)
  ctx->ranges.clear();
  if (!sharded && !ctx->use_ldlt) {
    const int top = ctx->top_level;
    std::vector<int> pos_of(K, -1);
    for (size_t pos = 0; pos < ctx->level_sn.size(); pos++) pos_of[ctx->level_sn[pos]] = (int)pos;
    std::vector<int> uf(K);
    auto find = [&](int x) {
      while (uf[x] != x) x = uf[x] = uf[uf[x]];
      return x;
    };
    // pieces[root] -> per level list of supernodes; returns false when a piece is too wide
    auto build = [&](int lo, int hi, std::vector<std::vector<std::vector<int>>>* out) {
      for (int e = 0; e < K; e++) uf[e] = e;
      auto in = [&](int e) { return ns[e] > 0 && pos_of[e] >= 0 && ctx->t_level[e] >= lo && ctx->t_level[e] < hi; };
      for (int e = 0; e < K; e++) {
        if (!in(e)) continue;
        for (int v : L.separators[e]) {
          const int a = L.var_to_sn[v];
          if (in(a)) uf[find(e)] = find(a);
        }
      }
      std::map<int, int> index;
      out->clear();
      for (int pos = ctx->level_ptr[lo]; pos < ctx->level_ptr[hi]; pos++) {  // level order keeps lists sorted
        const int e = ctx->level_sn[pos];
        const int r = find(e);
        auto it = index.find(r);
        if (it == index.end()) {
          it = index.emplace(r, (int)out->size()).first;
          out->emplace_back(hi - lo);
        }
        (*out)[it->second][ctx->t_level[e] - lo].push_back(e);
      }
      for (auto& piece : *out) {
        size_t total = 0;
        for (auto& lev : piece) {
          if (lev.size() > 8) return false;
          total += lev.size();
        }
        if (total > (size_t)kRangeMaxRecs) return false;
      }
      return true;
    };
    std::vector<SnRec> rr;
    int lo = 0;
    while (lo < top) {
      int hi = lo + 1;
      std::vector<std::vector<std::vector<int>>> pieces, trial;
      bool any_big = ctx->level_big[lo];
      if (!any_big)
        while (hi < top && !ctx->level_big[hi] && hi - lo < 8 && build(lo, hi + 1, &trial)) {
          pieces.swap(trial);
          hi++;
        }
      bool all_lean = !ctx->no_lean;
      for (int l = lo; l < hi; l++) all_lean = all_lean && ctx->level_lean[l];
      if (all_lean) {  // every level has its shape-specialised backward kernel: faster than the merged sweep
        lo = hi;
        continue;
      }
      if (hi - lo > 1) {
        auto rg = std::make_unique<cxk_context::SweepRange>();
        rg->lo = lo;
        rg->hi = hi;
        rg->groups = (int)pieces.size();
        std::vector<int> tab;
        size_t widest = 1;
        for (auto& piece : pieces) {
          for (auto& lev : piece) {
            tab.push_back((int)rr.size());
            widest = std::max(widest, lev.size());
            for (int e : lev) rr.push_back(h_recs[pos_of[e]]);
          }
          tab.push_back((int)rr.size());
        }
        rg->waves = (int)widest;
        CXK_TRY(rg->wg_lev.upload(tab));
        ctx->ranges.push_back(std::move(rg));
      }
      lo = hi;
    }
    CXK_TRY(ctx->rec_r.upload(rr));
  }
  CXK_TRY(ctx->p_ns.upload(ns));
  CXK_TRY(ctx->p_nsep.upload(nsep));
  CXK_TRY(ctx->p_start.upload(start));
  CXK_TRY(ctx->p_diag.upload(L.diag_off));
  CXK_TRY(ctx->p_offd.upload(L.offd_off));
  FactorPlan& P = ctx->plan;
  P.rec = ctx->p_rec.p;
  P.ns = ctx->p_ns.p;
  P.nsep = ctx->p_nsep.p;
  P.start = ctx->p_start.p;
  P.diag_off = ctx->p_diag.p;
  P.offd_off = ctx->p_offd.p;
  P.upd_off = ctx->upd_off.p;
  P.updb_off = ctx->updb_off.p;
  P.tg_ptr = ctx->tg_ptr.p;
  P.tg_loc = ctx->tg_loc.p;
  P.tg_reg = ctx->tg_reg.p;
  P.tr_ptr = ctx->tr_ptr.p;
  P.tr_src = ctx->tr_src.p;
  P.fs_ptr = ctx->fs_ptr.p;
  P.fs_src = ctx->fs_src.p;
  P.upd = ctx->upd.p;
  P.pub_dst = ctx->pub_dst.p;
  P.pubb_dst = ctx->pubb_dst.p;
  P.updb = ctx->updb.p;
  P.bs_ptr = ctx->bs_ptr.p;
  P.bs_c = ctx->bs_c.p;
  P.bs_row = ctx->bs_row.p;
  return CXK_SUCCESS;
}

}  // namespace cxk_host
