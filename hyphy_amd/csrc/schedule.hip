// Schedule compiler of libhyphy_hip.so: turns the host's list of dirty nodes into the post-order programs the pruning
// kernels interpret (common.h: entry format), cuts full passes into chains / level-peeled fragments, finds the node a
// re-rooted schedule hangs the tree from, and decides the device's pattern order.  Host-only code: nothing here touches
// the device (the hyphy_hip_plan_* entry points expose it to CPU tests).
#include "partition.h"

namespace hyhip {

// views[0] = the partition's own tree (matrix slot of a node = its node code)
void init_plain_view(hyphy_hip_partition *p) {
  hyphy_hip_partition::View &v = p->views[0];
  v.L = (int)p->L;
  v.I = (int)p->I;
  v.parents = p->parents;
  v.children = p->children;
  v.leaf_has_ambig = p->leaf_has_ambig;
  v.slot.resize((size_t)(p->L + p->I));
  for (size_t n = 0; n < v.slot.size(); n++) v.slot[n] = (int)n;
}

// Build the post-order schedule for the nodes the host marked dirty.  update_nodes comes from
// DetermineNodesForUpdate (tree.cpp:3117-3331): dirty nodes, their ancestors and the direct
// children of every touched internal node.  We recompute every internal node that is the parent
// of a listed node (plus ancestors, defensively) from ALL its children; children whose
// conditionals were not recomputed in this call are read back from the persisted device copy.
// Append one *program* (the schedule of a connected set of touched internal nodes, ascending =
// post-order) to p->ops_host.  Children that are internal nodes outside `nodes` are read from the
// persisted copy in HBM (they were finalised by an earlier launch or are unchanged).  The program is
// padded to an even entry count plus two trailing no-ops (the device loop is unrolled by two and
// fetches entries two ahead).  Returns the LDS slot its last node was finalised into.
int emit_program(hyphy_hip_partition *p, const std::vector<int> &nodes, int *offset_out, int *n_out, bool handoff,
                 bool is_root_program) {
  const hyphy_hip_partition::View &v = p->vw();
  const int L = v.L, I = v.I;
  const int T = p->shards.empty() ? 1 : p->shards[0].T;
  const int G = p->nuc ? (p->nuc_leaf_pairs ? 2 : 1) : (T <= 2 ? 2 : 1);  // leaves per leaf-group entry (prune.hip)
  // A finished node whose parent is the next node of the program is read by that parent straight from
  // the exchange slot it was finalised into (slots 0/1 alternate with the finalisation count, so the
  // writer of the NEXT finalisation never touches it); otherwise it is parked in an LDS slot
  // (2..lds_slots(T)-1) until its parent comes up or — when the slots run out — re-read from HBM.
  std::vector<int> slot_of(I, -1);      // LDS slot holding internal node i (valid until consumed)
  std::vector<char> recomputed(I, 0);   // finalised earlier in THIS program
  const int n_slots = p->n_slots > 0 ? p->n_slots : lds_slots(T);
  std::vector<char> slot_busy(n_slots, 0);
  std::vector<int> last_entry(I, -1);   // index (in ops_host) of the OPF_LAST entry of a node finalised by this program
  const bool lazy = !p->sched_persist;
  const int np_flag = p->nuc ? OPF_NOPERSIST_NUC : OPF_NOPERSIST;
  const int off = (int)p->ops_host.size();
  int fin = 0, root_slot = 0;
  for (size_t ti = 0; ti < nodes.size(); ti++) {
    const int par = nodes[ti];
    std::vector<int> ch_filtered;
    if (par == p->emit_skip_par) {  // (re-rooted schedules: the given root no longer has the first node of rr_path below it)
      for (int c : v.children[par])
        if (c != p->emit_skip_child) ch_filtered.push_back(c);
    }
    const std::vector<int> &ch = par == p->emit_skip_par ? ch_filtered : v.children[par];
    std::vector<int4> entries;
    std::vector<int> release_after;
    auto internal_entry = [&](int c) {
      int4 op;
      op.y = par;
      op.z = v.slot[c];
      op.w = c - L;
      // (4-state kernel: only parking slots are LDS; the node finalised last is still in registers)
      const int sl = p->nuc ? (slot_of[c - L] >= 2 ? slot_of[c - L] : -1) : slot_of[c - L];
      if (sl >= 0) {
        op.x = OPK_INTERNAL | (sl << 24);
        if (sl >= 2) release_after.push_back(sl);  // reusable only after this parent's barrier
        slot_of[c - L] = -1;
      } else {
        op.x = OPK_INTERNAL_GLOBAL | (0xff << 24);
        const bool inregs = p->nuc && ti > 0 && nodes[ti - 1] == c - L;  // (4-state kernel: child still in registers)
        if (recomputed[c - L]) {
          op.x |= OPF_GSYNC;
          if (!inregs && last_entry[c - L] >= 0) p->ops_host[last_entry[c - L]].x &= ~np_flag;  // re-read below: must be stored
        }
        else if (handoff) op.x |= OPF_HANDOFF;  // root of a child fragment finished by another workgroup of this launch
        if (inregs) op.x |= OPF_INREGS;
      }
      entries.push_back(op);
    };
    // order: [child finalised by the previous entry] -> leaves (grouped) -> other internal children
    int first_internal = -1;
    if (ti > 0)
      for (int c : ch)
        if (c >= L && c - L == nodes[ti - 1]) first_internal = c;
    if (first_internal >= 0) internal_entry(first_internal);
    std::vector<int> leaves;
    for (int c : ch)
      if (c < L) leaves.push_back(c);
    // leaves in groups of G; a leaf that carries ambiguity codes (in this shard) forms a group of its own
    // (its tiles may need a full matrix product instead of the column gather)
    for (size_t k = 0; k < leaves.size();) {
      int nl = 1;
      const bool amb0 = v.leaf_has_ambig[leaves[k]];
      if (!amb0 && G > 1 && k + 1 < leaves.size() && !v.leaf_has_ambig[leaves[k + 1]]) nl = 2;
      const unsigned l0 = (unsigned)leaves[k], l1 = nl > 1 ? (unsigned)leaves[k + 1] : l0;
      int4 op;
      op.x = OPK_LEAF | (amb0 ? OPF_AMBIG : 0) | (nl << 8) | (0xff << 24);
      op.y = par;
      op.z = (int)(l0 | (l1 << 16));
      op.w = 0;
      entries.push_back(op);
      k += nl;
    }
    for (int c : ch)
      if (c >= L && c != first_internal) internal_entry(c);
    // destination slot of the finished node
    int dst = fin & 1;
    const bool next_consumes = ti + 1 < nodes.size() && v.parents[L + par] == nodes[ti + 1];
    {
      if (!next_consumes && ti + 1 < nodes.size()) {
        dst = -1;
        for (int sidx = 2; sidx < n_slots; sidx++)
          if (!slot_busy[sidx]) {
            dst = sidx;
            break;
          }
        if (dst >= 0) {
          slot_busy[dst] = 1;
          slot_of[par] = dst;
        } else {
          dst = fin & 1;  // no parking slot free: the consumer will re-read the persisted copy
        }
      } else {
        slot_of[par] = dst;  // consumed by the very next parent from the exchange slot (or: last node)
      }
    }
    entries.back().x |= OPF_LAST | ((fin & 1) ? OPF_PARITY : 0) | (dst << 16);
    if (handoff && !is_root_program && ti + 1 == nodes.size()) entries.back().x |= OPF_PUBLISH;  // fragment root
    // lazy persistence: skip the store of this node unless it is the root of a fragment (read by another
    // program) — a later consumer through the persisted copy clears the flag again
    if (lazy && (is_root_program || ti + 1 < nodes.size())) entries.back().x |= np_flag;
    last_entry[par] = (int)p->ops_host.size() + (int)entries.size() - 1;
    for (const int4 &e : entries) p->ops_host.push_back(e);
    for (int sidx : release_after) slot_busy[sidx] = 0;
    recomputed[par] = 1;
    root_slot = dst;
    fin++;
  }
  int4 nop;
  nop.x = OPK_LEAF | (0xff << 24);
  nop.y = 0;
  nop.z = 0;
  nop.w = 0;
  if ((p->ops_host.size() - off) & 1) p->ops_host.push_back(nop);
  *offset_out = off;
  *n_out = (int)p->ops_host.size() - off;
  p->ops_host.push_back(nop);
  p->ops_host.push_back(nop);
  return root_slot;
}

namespace {

// Build the device schedule for the nodes the host marked dirty.  update_nodes comes from
// DetermineNodesForUpdate (tree.cpp:3117-3331): dirty nodes, their ancestors and the direct
// children of every touched internal node.  We recompute every internal node that is the parent
// of a listed node (plus ancestors, defensively) from ALL its children; children whose
// conditionals were not recomputed in this call are read back from the persisted device copy.
//
// Full evaluations are cut into LEVELS of independent subtree fragments ("forest scheduling"): the
// fragments of one level run concurrently as separate workgroups (grid.z), levels are separate
// launches, fragment roots are handed up through the persisted copy in HBM.  With one workgroup per
// 16-pattern tile walking the whole tree, 10k codons give only 624 workgroups for 768 resident
// slots (and 78 per GPU when sharded 8 ways): cutting the tree multiplies the workgroup count.
void build_schedule_impl(hyphy_hip_partition *p, const int64_t *update_nodes, int64_t n_update, bool full) {
  const hyphy_hip_partition::View &v = p->vw();  // (update_nodes: node codes of the view)
  const int L = v.L, I = v.I;
  std::vector<char> touched(I, 0);
  if (full) {
    std::fill(touched.begin(), touched.end(), 1);
  } else {
    for (int64_t k = 0; k < n_update; k++) {
      int64_t n = update_nodes[k];
      if (n < 0 || n >= L + I) continue;
      int64_t par = v.parents[n];
      while (par >= 0 && !touched[par]) {
        touched[par] = 1;
        par = v.parents[L + par];
      }
    }
  }
  p->ops_host.clear();
  p->programs.clear();
  p->levels.clear();
  std::vector<int> touched_list;
  for (int par = 0; par < I; par++)
    if (touched[par]) touched_list.push_back(par);
  if (touched_list.empty()) return;

  // fragment size: aim at >= ~6 workgroups per CU over the whole launch sequence
  int max_frag = I;
  if (full && !p->nuc && !p->shards.empty()) {
    const Shard &s0 = p->shards[0];
    const long wgs = std::max(1, s0.ntiles / std::max(1, s0.T)) * (long)std::max<int64_t>(1, p->batch_classes);
    const long target = 6L * s0.cus;
    if (const char *e = getenv("HYPHY_HIP_FRAGMENT")) max_frag = std::max(1, atoi(e));
    else if (wgs < target) max_frag = (int)std::max<long>(4, (long)I * wgs / target);
  }
  p->chain = false;
  p->rr_active = false;
  p->jn_host.clear();
  // ---- chain schedules (wave-per-tile kernel, full passes) -------------------------------------------------
  // Bottom subtrees of at most `m` internal nodes become SOURCE programs (walked serially by one wave, exactly like a
  // fragment); every node above them is a TRUNK node, reached by chains: a wave that finishes node c computes the
  // edge product towards the parent and arrives there, the last arriver finalises the parent and goes on (prune.hip).
  // The critical path of a tile is then the height of the tree (not the size of its largest fragment), and the grid is
  // dispatched source-major with the sources sorted by their distance to the root, so that every tile's critical path
  // starts first and the short chains that join near the root fill the end of the launch.
  // (HYPHY_HIP_CUT=levels or an explicit HYPHY_HIP_FRAGMENT keep the level-peeled fragments; HYPHY_HIP_CHAIN_M sets m)
  const bool want_levels = (getenv("HYPHY_HIP_CUT") && !strcmp(getenv("HYPHY_HIP_CUT"), "levels")) ||
                           (getenv("HYPHY_HIP_FRAGMENT") && !getenv("HYPHY_HIP_CHAIN_M")) ||
                           (p->chain_m_forced < 0 && !getenv("HYPHY_HIP_CHAIN_M"));
  if (full && !p->nuc && p->variant >= 1 && !p->shards.empty() && !want_levels) {
    const Shard &s0 = p->shards[0];
    // Topology the schedule is built on: the given one, or (re-rooted schedules) the same unrooted tree hung from rr_path.back();
    // the edges of rr_path are then reversed.  rpar = parent, order = children before parents, on_path = index along rr_path.
    const bool lazy_full = !p->sched_persist;
    const int rr_env = getenv("HYPHY_HIP_REROOT") ? atoi(getenv("HYPHY_HIP_REROOT")) : -1;  // (1: always, 0: never, unset: the tuner decides)
    const bool rr = !p->rr_path.empty() && (rr_env == 1 || (rr_env != 0 && p->rr_use)) && lazy_full && p->pin_node < 0 &&
                    p->batch_classes <= 1 && p->C == 1;
    std::vector<int> rpar(I, -1), on_path(I, -1), order;
    for (int n = 0; n < I - 1; n++) rpar[n] = (int)v.parents[L + n];
    int root_idx = I - 1;
    if (rr) {
      const std::vector<int> &a = p->rr_path;
      for (size_t i = 0; i + 1 < a.size(); i++) rpar[a[i]] = a[i + 1];
      rpar[a.back()] = -1;
      root_idx = a.back();
      for (size_t i = 0; i < a.size(); i++) on_path[a[i]] = (int)i;
    }
    std::vector<int> size(I, 1), height(I, 1), to_root(I, 0);
    std::vector<std::vector<int>> ich(I);
    for (int n = 0; n < I; n++)
      if (rpar[n] >= 0) ich[rpar[n]].push_back(n);
    {
      std::vector<std::pair<int, size_t>> stack(1, std::make_pair(root_idx, (size_t)0));
      while (!stack.empty()) {
        std::pair<int, size_t> &t = stack.back();
        if (t.second < ich[t.first].size()) {
          const int c = ich[t.first][t.second++];
          stack.push_back(std::make_pair(c, (size_t)0));
        } else {
          order.push_back(t.first);
          stack.pop_back();
        }
      }
    }
    for (int n : order)
      for (int c : ich[n]) {
        size[n] += size[c];
        height[n] = std::max(height[n], height[c] + 1);
      }
    for (size_t k = order.size(); k-- > 0;)
      if (rpar[order[k]] >= 0) to_root[order[k]] = to_root[rpar[order[k]]] + 1;
    auto count_sources = [&](int m) {
      int k = 0;
      for (int n = 0; n < I; n++)
        if (size[n] <= m && on_path[n] <= 0 && (rpar[n] < 0 || size[rpar[n]] > m || on_path[rpar[n]] > 0)) k++;
      return k;
    };
    const long wgs = std::max(1, s0.ntiles) * (long)std::max<int64_t>(1, p->batch_classes);
    const long target = 24L * s0.cus;  // >= 3 rounds of the 8 resident waves per CU
    int m = 1;
    if (const char *e = getenv("HYPHY_HIP_CHAIN_M")) m = std::max(1, atoi(e));
    else if (p->chain_m_forced > 0) m = p->chain_m_forced;
    else {
      if (wgs >= target) m = I;  // enough tiles: one wave walks the whole tree
      else
        for (int t = 2; t <= 8; t++)
          if ((long)count_sources(t) * wgs >= target) m = t;
    }
    // What the join table can describe (the kernels decode jn[n].x = parent | image slot << 16 with "negative = root", and
    // jn[n].y = arrivals needed | child sum << 8): parents below 2^16, image slots — the twins of a re-rooted schedule sit
    // behind the branch cache's, at B + I + 2 + k — below 2^15, at most 255 internal children per node.  A tree beyond that
    // (upwards of ~10 000 taxa, or a star of > 255 subtrees) is walked by the cuts that need no join table.
    {
      bool ok = I <= 65535 && (long)p->B + p->I + 2 + kMaxTwin < 32768;
      for (int n = 0; n < I && ok; n++)
        if (ich[n].size() > 255) ok = false;
      if (!ok) m = I;
    }
    if (m < I) {
      struct Src { int root, prio; };
      std::vector<Src> srcs;
      std::vector<char> in_source(I, 0);
      for (size_t k = order.size(); k-- > 0;) {  // parents before children; the nodes of rr_path above the given root stay trunk nodes
        const int n = order[k], par = rpar[n];
        if (par >= 0 && in_source[par]) in_source[n] = 1;
        else if (size[n] <= m && on_path[n] <= 0) {  // (on_path == 0: the given root — a source like any other if it is small)
          in_source[n] = 1;
          srcs.push_back({n, to_root[n] + height[n]});
        }
      }
      std::stable_sort(srcs.begin(), srcs.end(), [](const Src &x, const Src &y) { return x.prio > y.prio || (x.prio == y.prio && x.root < y.root); });
      // Tiny sources next to the root would be dispatched last and, arriving last at their joins, carry the serial
      // remainder of the trunk while the chip drains: dispatched FIRST they deposit and retire in a few microseconds,
      // and the long chains that arrive later go on with the trunk (HYPHY_HIP_TINY_FIRST = largest source size moved up).
      {
        const int tiny = getenv("HYPHY_HIP_TINY_FIRST") ? atoi(getenv("HYPHY_HIP_TINY_FIRST")) : 0;
        if (tiny > 0) std::stable_partition(srcs.begin(), srcs.end(), [&](const Src &x) { return size[x.root] <= tiny; });
      }
      p->jn_host.assign(I, make_int4(-1, 0, 0, 0));
      for (const Src &sr : srcs) {
        std::vector<int> nodes;  // the subtree below sr.root, ascending = post-order
        std::vector<int> stack(1, sr.root);
        while (!stack.empty()) {
          const int n = stack.back();
          stack.pop_back();
          nodes.push_back(n);
          for (int c : ich[n]) stack.push_back(c);
        }
        std::sort(nodes.begin(), nodes.end());
        int off, n;
        p->emit_skip_par = rr ? p->rr_path[0] : -1;
        p->emit_skip_child = rr ? L + p->rr_path[1] : -1;
        const int rs = emit_program(p, nodes, &off, &n, false, true);
        p->emit_skip_par = p->emit_skip_child = -1;
        hyphy_hip_partition::Prog pr{off, n};
        pr.parent = sr.root == root_idx ? -1 : 0;
        if (p->variant == 2) pr.parent = rs;  // (row-split workgroups: the exchange slot the source's root ends up in)
        pr.need = sr.root;  // (chain schedules: w = the source's root node)
        p->programs.push_back(pr);
        if (sr.root == root_idx) p->root_slot = rs;
      }
      const bool lazy = !p->sched_persist;
      for (int n = 0; n < I; n++) {
        int sum = 0;
        if (ich[n].size() == 2) sum = ich[n][0] + ich[n][1];  // (read by a wave that finds its ONE sibling already arrived)
        // x: parent | image slot of the edge above n << 16: the node's own branch L + n, or — reversed edges of a re-rooted
        // schedule — the transposed twin of the branch of the NEXT node on the path (expm.hip; slot behind the branch cache's)
        const int slot = (rr && on_path[n] >= 0) ? (int)(p->B + (p->I + 2) + on_path[n]) : v.slot[L + n];
        int4 j = make_int4(rpar[n] < 0 ? -1 : (rpar[n] | (slot << 16)), (int)ich[n].size() | (sum << 8), 0, 0);
        if (!in_source[n]) {  // trunk node: its leaf groups, one OPK_DEP entry per internal child, finalisation flags
          j.z = (int)p->ops_host.size();
          std::vector<int> leaves;
          for (int c : v.children[n])
            if (c < L) leaves.push_back(c);
          for (size_t k = 0; k < leaves.size();) {
            int nl = 1;
            const bool amb0 = v.leaf_has_ambig[leaves[k]];
            if (!amb0 && k + 1 < leaves.size() && !v.leaf_has_ambig[leaves[k + 1]]) nl = 2;
            const unsigned l0 = (unsigned)leaves[k], l1 = nl > 1 ? (unsigned)leaves[k + 1] : l0;
            p->ops_host.push_back(make_int4(OPK_LEAF | (amb0 ? OPF_AMBIG : 0) | (nl << 8) | (0xff << 24), n, (int)(l0 | (l1 << 16)), 0));
            k += nl;
          }
          for (int c : ich[n]) p->ops_host.push_back(make_int4(OPK_DEP | (0xff << 24), n, v.slot[L + c], c));
          p->ops_host.back().x |= OPF_LAST | (lazy ? OPF_NOPERSIST : 0);
          j.w = (int)p->ops_host.size() - j.z;
        }
        p->jn_host[n] = j;
      }
      p->ops_host.push_back(make_int4(OPK_LEAF | (0xff << 24), 0, 0, 0));  // (the interpreter reads one entry ahead)
      p->levels.push_back({0, (int)p->programs.size()});
      p->chain = true;
      p->rr_active = rr;
      if (getenv("HYPHY_HIP_VERBOSE")) {
        fprintf(stderr, "[hyphy_hip] chain schedule: m = %d, %zu sources (root node / distance):", m, srcs.size());
        for (const Src &sr : srcs) fprintf(stderr, " %d/%d", sr.root, sr.prio);
        fprintf(stderr, "\n");
      }
      return;
    }
  }
  if (max_frag >= I || !full) {  // one program (also: every partial update)
    int off, n;
    p->root_slot = emit_program(p, touched_list, &off, &n);
    p->programs.push_back({off, n});
    p->levels.push_back({0, 1});
    return;
  }
  // peel levels: a level's fragments are the maximal subtrees (in what is left of the tree) with at
  // most max_frag internal nodes; their roots become HBM-resident inputs of the next level.
  // Wave-per-tile kernel: ONE launch; the fragments are chained on the device — the workgroup that
  // completes the last child fragment of a program (per tile) goes on to run that program itself
  // (arrival counters, prune.hip), so the levels below only define the cut, not launches.
  const bool chained = p->variant == 1;
  std::vector<int> prog_of(I, -1);
  std::vector<char> done(I, 0);
  std::vector<int> size(I, 0);
  for (;;) {
    for (int n = 0; n < I; n++) {  // post-order: children before parents
      if (done[n]) { size[n] = 0; continue; }
      int sz = 1;
      for (int c : v.children[n])
        if (c >= L) sz += size[c - L];
      size[n] = sz;
    }
    const int root = I - 1;
    const int first_prog = (int)p->programs.size();
    std::vector<std::vector<int>> frags;
    if (size[root] <= max_frag) {
      std::vector<int> rest;
      for (int n = 0; n < I; n++)
        if (!done[n]) rest.push_back(n);
      frags.push_back(rest);
    } else {
      // fragment roots: size <= max_frag while the parent's is larger
      std::vector<int> frag_root(I, -1);
      for (int n = I - 1; n >= 0; n--) {  // parents before children
        if (done[n]) continue;
        const int par = (int)v.parents[L + n];
        if (par >= 0 && !done[par] && frag_root[par] >= 0) frag_root[n] = frag_root[par];
        else if (size[n] <= max_frag) frag_root[n] = n;
      }
      std::vector<int> index(I, -1);
      for (int n = 0; n < I; n++) {
        if (done[n] || frag_root[n] < 0) continue;
        if (index[frag_root[n]] < 0) {
          index[frag_root[n]] = (int)frags.size();
          frags.push_back(std::vector<int>());
        }
        frags[index[frag_root[n]]].push_back(n);
      }
    }
    // longest fragments first: the grid is dispatched in program order within a tile, and a tile's chained
    // parent program can only start after its slowest child
    if (chained)
      std::stable_sort(frags.begin(), frags.end(),
                       [](const std::vector<int> &x, const std::vector<int> &y) { return x.size() > y.size(); });
    bool finished = false;
    for (const std::vector<int> &f : frags) {
      int off, n;
      const int rs = emit_program(p, f, &off, &n, chained, f.back() == root);
      p->programs.push_back({off, n});
      for (int nd : f) prog_of[nd] = (int)p->programs.size() - 1;
      for (int nd : f) done[nd] = 1;
      if (f.back() == root) {
        p->root_slot = rs;
        finished = true;
      }
    }
    p->levels.push_back({first_prog, (int)frags.size()});
    if (getenv("HYPHY_HIP_VERBOSE")) {
      fprintf(stderr, "[hyphy_hip] level %zu: %zu fragment(s), internal nodes:", p->levels.size() - 1, frags.size());
      for (const std::vector<int> &f : frags) fprintf(stderr, " %zu", f.size());
      fprintf(stderr, "\n");
    }
    if (finished) break;
  }
  if (chained) {
    for (size_t k = 0; k < p->programs.size(); k++) {
      // the fragment root is the parent (y) of the last OPF_LAST entry of the program
      int froot = -1;
      for (int e = 0; e < p->programs[k].n; e++) {
        const int4 &op = p->ops_host[p->programs[k].off + e];
        if (op.x & OPF_LAST) froot = op.y;
      }
      const int par_node = froot >= 0 ? (int)v.parents[L + froot] : -1;
      if (par_node >= 0) {
        p->programs[k].parent = prog_of[par_node];
        p->programs[p->programs[k].parent].need++;
      }
    }
    const auto l0 = p->levels[0];
    p->levels.clear();
    p->levels.push_back(l0);  // one launch: grid.z = the leaf fragments; the rest is reached by chaining
  }
}

// Rescaling tests only where they are needed (wave-per-tile kernel, full passes).  A rescale multiplies by an exact power
// of 2^64, so WHERE a node is tested does not change any mantissa — only underflow has to be excluded.  A node whose
// internal children were all tested (their per-pattern sums are >= 2^-64 after the test) and that has at most four children
// cannot fall below 2^-256 times the spread of a conditional vector, hundreds of binary orders above the denormals: its
// own test is skipped (OPF_NOSCALE) and its parent tests again.  The root is always tested.  Saves the cross-lane sum,
// the ballot and their latency in front of the next product at every other level (HYPHY_HIP_SCALE_THIN=0: test everywhere).
void thin_rescale_tests(hyphy_hip_partition *p) {
  static const bool on = !(getenv("HYPHY_HIP_SCALE_THIN") && atoi(getenv("HYPHY_HIP_SCALE_THIN")) == 0);
  if (!on) return;
  const hyphy_hip_partition::View &v = p->vw();
  const int L = v.L, I = v.I;
  std::vector<char> tested(I, 1);
  for (int n = 0; n < I; n++) {  // children before parents
    bool kids_tested = true;
    for (int c : v.children[n])
      if (c >= L && !tested[c - L]) kids_tested = false;
    // (mode 1: the class-table kernel tests every compressed node, so a generalised leaf counts like a tested child;
    //  pinned states only exist in mode 0)
    tested[n] = (n == I - 1 || !kids_tested || v.children[n].size() > 4 || (p->mode == 0 && n == p->pin_node - L)) ? 1 : 0;
  }
  if (p->rr_active)  // (re-rooted schedule: the nodes whose children differ from the given topology — and the new root — always test)
    for (int n : p->rr_path) tested[n] = 1;
  for (int4 &op : p->ops_host)
    if ((op.x & OPF_LAST) && op.y >= 0 && op.y < I && !tested[op.y]) op.x |= OPF_NOSCALE;
}

}  // namespace

void build_schedule(hyphy_hip_partition *p, const int64_t *update_nodes, int64_t n_update, bool full) {
  build_schedule_impl(p, update_nodes, n_update, full);
  if (full && !p->nuc && p->variant == 1 && !p->shards.empty() && p->shards[0].T == 1) thin_rescale_tests(p);
}

// Pattern order on the device.  A leaf edge is a per-site column gather from the leaf branch's matrix (prune.hip: leaf_gather),
// and the texture-address path coalesces the lanes of a quad that read the same cache lines: 16 sites with 16 different
// states cost 4 000 cycles per gather under load, sites that share their state in runs of >= 4 cost 1 070
// (tools/ubench/glds_probe.hip).  Patterns are therefore kept sorted — by their most frequent state first (conserved sites
// of the same codon become neighbours), then lexicographically by leaf — which takes the distinct states per (leaf, tile)
// from 14.2 to 3.9 on the headline alignment.  The order is internal: every per-pattern input and output of the C-ABI is
// translated through `perm` (gather_sites, download_partials, set_pinned_states, site fits).
// The internal node that minimises the tree's height (edges to the farthest leaf) when the tree is hung from it, and the path
// to it from the given root (rr_path, see hyphy_hip_partition).  Topology only; ties keep the given root.
void reroot_path(hyphy_hip_partition *p) {
  const hyphy_hip_partition::View &vw = p->vw();
  const int L = vw.L, I = vw.I, N = L + I;
  p->rr_path.clear();
  p->rr_cands.clear();
  if (I < 4) return;
  std::vector<std::vector<int>> adj(N);
  for (int n = 0; n < N - 1; n++) {
    const int par = L + (int)vw.parents[n];
    adj[n].push_back(par);
    adj[par].push_back(n);
  }
  // distances (in edges) from `r` to every node, BFS parents in `from`; returns the farthest LEAF
  auto sweep = [&](int r, std::vector<int> &dist, std::vector<int> &from) {
    dist.assign(N, -1);
    from.assign(N, -1);
    std::vector<int> queue(1, r);
    dist[r] = 0;
    int far = -1;
    for (size_t h = 0; h < queue.size(); h++) {
      const int n = queue[h];
      if (n < L && (far < 0 || dist[n] > dist[far])) far = n;
      for (int m : adj[n])
        if (dist[m] < 0) {
          dist[m] = dist[n] + 1;
          from[m] = n;
          queue.push_back(m);
        }
    }
    return far;
  };
  // The nodes of least eccentricity over the leaves are the middle of a longest leaf-to-leaf path (two sweeps).
  const int root = N - 1;
  std::vector<int> d0, f0, du, fu;
  const int u = sweep(root, d0, f0);
  const int root_height = d0[u];
  const int v = sweep(u, du, fu);
  const int D = du[v];
  (void)root_height;  // (a given root that is itself a centre node keeps the other centre node as a candidate: the tuner times both)
  std::vector<int> mids;
  for (int n = v; n != u; n = fu[n])  // walk v -> u: the one or two middle nodes
    if ((du[n] == D / 2 || du[n] == (D + 1) / 2) && n >= L && n != root) mids.push_back(n);
  std::sort(mids.begin(), mids.end(), [&](int x, int y) { return d0[x] < d0[y]; });  // closest to the given root first
  p->rr_cands.clear();
  for (int best : mids) {
    std::vector<int> up;  // best -> ... -> root
    for (int n = best; n != root; n = L + (int)vw.parents[n]) up.push_back(n - L);
    up.push_back(I - 1);
    if ((int)up.size() - 1 > kMaxTwin) continue;
    p->rr_cands.push_back(std::vector<int>(up.rbegin(), up.rend()));
  }
  if (!p->rr_cands.empty()) p->rr_path = p->rr_cands[0];
}

void sort_patterns(hyphy_hip_partition *p, const int64_t *leaf_codes, int64_t L, int64_t S) {
  if (S < 32) return;
  std::vector<int64_t> major(S, 0);
  {
    std::vector<int> cnt;
    for (int64_t k = 0; k < S; k++) {
      cnt.assign((size_t)p->D, 0);
      int best = 0;
      for (int64_t l = 0; l < L; l++) {
        const int64_t c = leaf_codes[l * S + k];
        if (c >= 0 && c < p->D && ++cnt[(size_t)c] > cnt[(size_t)best]) best = (int)c;
      }
      major[k] = best;
    }
  }
  p->perm.resize(S);
  for (int64_t k = 0; k < S; k++) p->perm[k] = k;
  std::sort(p->perm.begin(), p->perm.end(), [&](int64_t a, int64_t b) {
    if (major[a] != major[b]) return major[a] < major[b];
    for (int64_t l = 0; l < L; l++) {
      const int64_t ca = leaf_codes[l * S + a], cb = leaf_codes[l * S + b];
      if (ca != cb) return ca < cb;
    }
    return a < b;
  });
}

}  // namespace hyhip

using namespace hyhip;

extern "C" {

/* Host-only planning helpers (no device needed): what hyphy_hip_create decides from the topology and the leaf table alone. */
int64_t hyphy_hip_plan_reroot(int64_t L, int64_t I, const int64_t *flat_parents, int64_t candidate, int64_t *path_out, int64_t cap) {
  if (L < 2 || I < 1 || !flat_parents) return -1;
  hyphy_hip_partition tmp;
  tmp.L = L;
  tmp.I = I;
  tmp.parents.assign(flat_parents, flat_parents + L + I);
  for (int64_t n = 0; n < L + I - 1; n++)
    if (flat_parents[n] < 0 || flat_parents[n] >= I) return -1;
  tmp.children.assign(I, std::vector<int>());
  for (int64_t n = 0; n < L + I - 1; n++) tmp.children[flat_parents[n]].push_back((int)n);
  tmp.leaf_has_ambig.assign(L, 0);
  init_plain_view(&tmp);
  reroot_path(&tmp);
  if (candidate < 0 || candidate >= (int64_t)tmp.rr_cands.size()) return 0;
  const std::vector<int> &path = tmp.rr_cands[(size_t)candidate];
  for (size_t k = 0; k < path.size() && (int64_t)k < cap; k++)
    if (path_out) path_out[k] = path[k];
  return (int64_t)path.size();
}

int hyphy_hip_plan_pattern_order(int64_t D, int64_t L, int64_t S, const int64_t *leaf_codes, int64_t *order_out) {
  if (D < 2 || L < 1 || S < 1 || !leaf_codes || !order_out) return -1;
  hyphy_hip_partition tmp;
  tmp.D = D;
  tmp.L = L;
  tmp.S = S;
  sort_patterns(&tmp, leaf_codes, L, S);
  for (int64_t k = 0; k < S; k++) order_out[k] = tmp.perm.empty() ? k : tmp.perm[k];
  return 0;
}

int hyphy_hip_plan_schedule(int64_t L, int64_t I, const int64_t *flat_parents, int64_t kernel, int64_t chain_m, int64_t ntiles,
                            int64_t reroot, int64_t *info_out) {
  if (L < 2 || I < 1 || !flat_parents || !info_out || kernel < 0 || kernel > 2 || ntiles < 1) return -1;
  hyphy_hip_partition tmp;
  tmp.D = 61; tmp.L = L; tmp.I = I; tmp.C = 1; tmp.B = L + I - 1; tmp.NW = 4; tmp.DP = 64;
  tmp.parents.assign(flat_parents, flat_parents + L + I);
  tmp.children.assign(I, std::vector<int>());
  for (int64_t n = 0; n < L + I - 1; n++) {
    const int64_t par = flat_parents[n];
    if (par < 0 || par >= I || (n >= L && par <= n - L)) return -1;
    tmp.children[par].push_back((int)n);
  }
  tmp.leaf_has_ambig.assign(L, 0);
  init_plain_view(&tmp);
  tmp.shards.resize(1);
  tmp.shards[0].T = 1;
  tmp.shards[0].ntiles = (int)ntiles;
  tmp.shards[0].S_pad = (int)ntiles * 16;
  tmp.shards[0].cus = 256;
  tmp.variant = (int)kernel;
  tmp.n_slots = kernel == 1 ? tmp.n_slots_wave : lds_slots(1);
  tmp.chain_m_forced = (int)chain_m;
  tmp.sched_persist = false;  // a steady-state (lazy) full pass
  if (reroot) {
    reroot_path(&tmp);
    tmp.rr_use = true;
  }
  build_schedule(&tmp, nullptr, 0, true);
  int64_t max_slot = 0, max_need = 0, bad = 0, trunk = 0;
  if (tmp.chain) {
    // decode every record exactly as prune.hip does and hold it against the topology the schedule was built on
    std::vector<int> rpar(I, -1);
    for (int64_t n = 0; n + 1 < I; n++) rpar[n] = (int)flat_parents[L + n];
    if (tmp.rr_active) {
      for (size_t i = 0; i + 1 < tmp.rr_path.size(); i++) rpar[tmp.rr_path[i]] = tmp.rr_path[i + 1];
      rpar[tmp.rr_path.back()] = -1;
    }
    std::vector<int> kids(I, 0);
    for (int64_t n = 0; n < I; n++)
      if (rpar[n] >= 0) kids[rpar[n]]++;
    for (int64_t n = 0; n < I; n++) {
      const int4 j = tmp.jn_host[n];
      if (rpar[n] < 0) {
        if (j.x >= 0) bad++;
      } else {
        if (j.x < 0 || (j.x & 0xffff) != rpar[n]) bad++;
        max_slot = std::max<int64_t>(max_slot, j.x >> 16);
      }
      if ((j.y & 0xff) != kids[n]) bad++;
      max_need = std::max<int64_t>(max_need, j.y & 0xff);
      if (j.w > 0) trunk++;
    }
  }
  info_out[0] = tmp.chain ? 1 : 0;
  info_out[1] = (int64_t)tmp.programs.size();
  info_out[2] = (int64_t)tmp.ops_host.size();
  info_out[3] = max_slot;
  info_out[4] = max_need;
  info_out[5] = bad;
  info_out[6] = tmp.rr_active ? 1 : 0;
  info_out[7] = trunk;
  tmp.shards.clear();
  return 0;
}

const char *hyphy_hip_schedule_info(const hyphy_hip_partition *p) {
  if (!p) return "";
  static thread_local std::string out;
  char buf[96];
  snprintf(buf, sizeof buf, " [current: %s%s, %zu program(s)]", p->chain ? "chain" : "levels", p->rr_active ? ", re-rooted" : "",
           p->programs.size());
  out = p->tune_report + buf;
  if (!p->rep_report.empty()) out += " {" + p->rep_report + "}";
  if (!p->shards.empty()) {
    snprintf(buf, sizeof buf, " [device memory, first shard: %.1f MB]", p->shards[0].dev_bytes / 1e6);
    out += buf;
  }
  return out.c_str();
}

}  // extern "C"
