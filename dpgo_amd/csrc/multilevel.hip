// multilevel.hip -- the multilevel preconditioner: hierarchy set-up and V-cycle launches (stands in for src/PoseGraph.cpp:598-613, src/QuadraticProblem.cpp:56-69).
#include "host.h"

namespace dpgo_host {

// ---------------------------------------------------------------------------------------------------------
// Multilevel preconditioner: hierarchy setup (symbolic on the host once per block pattern, numeric on the device
// for every new set of Q values) and the per-iteration launches.  DESIGN.md section 5.
int ml_tile(int b, int split) { return (64 / (b * split)) * kWaves; }
// tiles of the additive preconditioner's layout (4 lane groups per pose, one tile = one aggregate per workgroup)
int additive_tile(const dpgo_problem_s* p) { return ml_tile(p->b, 4); }
int ml_level_split(int n) { return n < 40000 ? 4 : 1; }

// Aggregate sizes per coarsening.  Every k must divide the workgroup tile of its level (fused restriction); the
// coarsest operator is a dense inverse of at most kMlDense unknowns (Infinity-Cache resident), kMlDenseMax if that is
// what it takes to get there in one coarsening; otherwise one more level.
constexpr int kMlDense = 3200, kMlDenseMax = 6400;
// Two-level hierarchies use GRAPH aggregates (a single negative entry -S: breadth-first-grown aggregates of at most S
// poses, ml_graph_aggregates) whenever one coarsening with S <= kMlGraphMax reaches a dense level of about 2 500
// unknowns: compact aggregates need 40-60 % of the Hessian-vector products that index runs of the same size need
// (DESIGN.md section 5), and the dense level can then be small.  DPGO_ML_GRAPH=0: index runs as before.
constexpr int kMlGraphMax = 512, kMlGraphUnknownsPerPose = 1600;
int ml_default_graph_size(int n, int b) {
  if (options().ml_graph == 0) return 0;
  if (options().ml_graph_size >= 2) return options().ml_graph_size;  // experiments: force the size
  const long long S = std::max<long long>(4, ((long long)n * b + kMlGraphUnknownsPerPose - 1) / kMlGraphUnknownsPerPose);
  return S <= kMlGraphMax ? (int)S : 0;
}
// Blocks whose plain greedy growth would use aggregates of >= kMlMergeFrom poses (n (d+1) >= ~100 000 unknowns: >= 25 600
// poses in 3-D) grow them to S = ceil(n (d+1) / 2 200) instead and MERGE the growth's fragments up to 3 S / 2
// (ml_merge_small_aggregates): the aggregates come out uniform (mean ~ S instead of ~0.55 S with a tail of fragments), the
// same coarse-space quality needs a quarter fewer of them -- 100k poses: 546 aggregates / 70 products to |rgrad| < 1e-2
// against 732 / 77, a dense level of 38 MB instead of 69 MB; 25k: 536 / 87 against 589 / 93 (oracle, round 4).  Round 5:
// from a plain growth size of 12 on (n (d+1) > ~17 600 unknowns: >= 4 400 poses in 3-D) -- torus3D 38 -> 35 products
// (exact factor: 22; S = 6 / cap 9: 29 with 814 aggregates, S = 4 / cap 6: 28 with 1 215 aggregates = a dense level of 4 860
// unknowns, 189 MB, whose stream costs what the ten products save), 6 250-pose grid 50 -> 42, 12 500-pose slab 140 -> ~105
// (oracle, tools/balanced_aggregates_experiment.py ... vcycle).  Smaller blocks keep the plain growth (sphere2500: 364 aggregates / 32 products against 417 / 49 merged;
// their hierarchies are what the committed vectors pin).
constexpr int kMlMergeFrom = 12, kMlMergedUnknownsPerPose = 2200;
std::vector<int> ml_default_ks(int n, int b, int split0) {
  if (const int S = ml_default_graph_size(n, b)) {
    const bool forced = options().ml_graph_size >= 2;
    if (!forced && S >= kMlMergeFrom) {
      const int Sm = (int)(((long long)n * b + kMlMergedUnknownsPerPose - 1) / kMlMergedUnknownsPerPose);
      const int cap = Sm + Sm / 2;
      if (cap <= kMlGraphMax) return std::vector<int>{-Sm, -cap};
    }
    return std::vector<int>{-S};
  }
  std::vector<int> ks;
  int cur = n, split = split0;
  for (int guard = 0; guard < 16; ++guard) {
    const int P = ml_tile(b, split);
    int pick = 0;
    for (int limit : {kMlDense, kMlDenseMax}) {
      for (int k = 4; k <= P && !pick; ++k)
        if (P % k == 0 && (long long)((cur + k - 1) / k) * b <= limit) pick = k;
      if (pick) break;
    }
    if (pick) {
      ks.push_back(pick);
      return ks;
    }
    int k = 2;
    for (int c = 2; c <= 8; ++c)
      if (P % c == 0) k = c;
    ks.push_back(k);
    cur = (cur + k - 1) / k;
    split = ml_level_split(cur);
  }
  return ks;
}

void ml_free(dpgo_problem_s* p) {
  p->ml_additive_layout = false;
  for (auto& L : p->ml) {
    free_bsr(L.A);
    free_bsr(L.AP);
    void* ptrs[] = {L.slot_row, L.dinv, L.Pb, L.r, L.x1, L.x, L.res1, L.lab, L.agg_ptr, L.agg_mem, L.parent, L.pslot, L.tbuf, L.tile_perm, L.mem_pos,
                    L.seg_info, L.seg_ptr, L.Pb32, L.AP32, L.x1f, L.res1f};
    for (void* q : ptrs)
      if (q) (void)hipFree(q);
  }
  p->ml.clear();
  void* ptrs[] = {p->ml_dense, p->ml_W, p->ml_Rx, p->ml_dense32, p->ml_packed, p->ml_pd, p->ml_pt, p->ml_chunks,
                  p->ml_chunk_first};
  for (void* q : ptrs)
    if (q) (void)hipFree(q);
  p->ml_dense = p->ml_W = p->ml_Rx = nullptr;
  p->ml_dense32 = nullptr;
  p->ml_packed = p->ml_pd = p->ml_pt = nullptr;
  p->ml_chunks = nullptr;
  p->ml_chunk_first = nullptr;
  p->ml_nchunks = 0;
  p->ml_lda = 0;
  p->ml_symbolic = p->ml_ready = false;
  p->ml_ops32_ready = false;
}

// Large blocks grow (and merge) their aggregates independently inside `chunks` contiguous index ranges -- [n c / chunks,
// n (c + 1) / chunks): a seed's search does not leave its range, a fragment joins a neighbour of its own range -- one host
// thread each; the ranges' aggregates are numbered one range after the other.  This is the RULE (restated in the oracle:
// amg_growth_chunks / the lo, hi arguments of amg_graph_aggregates and amg_merge_small_aggregates), not a schedule: the
// result does not depend on the number of threads that execute it.  100k poses: growth + merge 3.7 -> 0.6 ms.
int ml_growth_chunks(int n) {
  const int forced = options().ml_growth_chunks;
  if (forced > 0) return std::max(1, std::min(forced, std::max(1, n / 64)));
  return n >= 65536 ? 8 : 1;
}
namespace {
// growth inside [lo, hi): lab (aggregate ids local to the range, from 0), parent, pslot written at the range's indices;
// mem / ptr local to the range
int grow_range(const std::vector<int32_t>& rowptr, const std::vector<int32_t>& colidx, int lo, int hi, int S,
               std::vector<int32_t>& lab, std::vector<int32_t>& ptr, std::vector<int32_t>& mem,
               std::vector<int32_t>& parent, std::vector<int32_t>& pslot) {
  mem.clear();
  mem.reserve(hi - lo);
  ptr.assign(1, 0);
  int na = 0;
  for (int s = lo; s < hi; ++s) {
    if (lab[s] >= 0) continue;
    const size_t first = mem.size();
    lab[s] = na;
    mem.push_back(s);
    for (size_t head = first; head < mem.size() && (int)(mem.size() - first) < S; ++head) {
      const int u = mem[head];
      for (int t = rowptr[u]; t < rowptr[u + 1] && (int)(mem.size() - first) < S; ++t) {
        const int v = colidx[t];
        if (v < lo || v >= hi || lab[v] >= 0) continue;
        lab[v] = na;
        parent[v] = u;
        pslot[v] = t;
        mem.push_back(v);
      }
    }
    ptr.push_back((int32_t)mem.size());
    ++na;
  }
  return na;
}

// merge inside [lo, hi) (lab: ids local to the range); in place; returns the number of aggregates of the range
int merge_range(const std::vector<int32_t>& rowptr, const std::vector<int32_t>& colidx, int lo, int hi, int S, int cap,
                std::vector<int32_t>& lab, std::vector<int32_t>& ptr, std::vector<int32_t>& mem,
                std::vector<int32_t>& parent, std::vector<int32_t>& pslot) {
  const int na = (int)ptr.size() - 1;
  const int n = hi - lo;
  std::vector<std::vector<int32_t>> members(na);
  for (int a = 0; a < na; ++a) members[a].assign(mem.begin() + ptr[a], mem.begin() + ptr[a + 1]);
  std::vector<int> cnt(na, 0);
  std::vector<int> touched;
  for (bool changed = true; changed;) {
    changed = false;
    for (int a = 0; a < na; ++a) {
      if (members[a].empty() || 2 * (int)members[a].size() > S) continue;
      touched.clear();
      for (int i : members[a])
        for (int t = rowptr[i]; t < rowptr[i + 1]; ++t) {
          const int v = colidx[t];
          if (v < lo || v >= hi) continue;
          const int c = lab[v];
          if (c == a) continue;
          if (cnt[c]++ == 0) touched.push_back(c);
        }
      std::sort(touched.begin(), touched.end());
      int best = -1, best_n = 0;
      for (int c : touched) {
        if ((int)(members[c].size() + members[a].size()) <= cap && cnt[c] > best_n) best = c, best_n = cnt[c];
        cnt[c] = 0;
      }
      if (best >= 0) {
        members[best].insert(members[best].end(), members[a].begin(), members[a].end());
        for (int i : members[a]) lab[i] = best;
        members[a].clear();
        changed = true;
      }
    }
  }
  // Members of the surviving aggregates in index order (one counting pass over the poses instead of a sort per aggregate),
  // `grew` = absorbed at least one other aggregate.
  std::vector<char> grew(na, 0);
  for (int a = 0; a < na; ++a)
    if (!members[a].empty() && (int)members[a].size() != ptr[a + 1] - ptr[a]) grew[a] = 1;
  std::vector<int32_t> first_member(na, -1);
  for (int i = hi - 1; i >= lo; --i) first_member[lab[i]] = i;
  std::vector<int> alive;
  for (int a = 0; a < na; ++a)
    if (!members[a].empty()) alive.push_back(a);
  std::sort(alive.begin(), alive.end(), [&](int x, int y) { return first_member[x] < first_member[y]; });
  std::vector<int32_t> sorted_mem, sorted_ptr(na + 1, 0);
  {  // only the aggregates that grew need their members sorted (the roots of the search below)
    for (int a = 0; a < na; ++a) sorted_ptr[a + 1] = sorted_ptr[a] + (grew[a] ? (int32_t)members[a].size() : 0);
    sorted_mem.resize(sorted_ptr[na]);
    std::vector<int32_t> fill(sorted_ptr.begin(), sorted_ptr.end() - 1);
    for (int i = lo; i < hi; ++i)
      if (grew[lab[i]]) sorted_mem[fill[lab[i]]++] = i;
  }
  std::vector<int32_t> new_lab(n, -1), new_mem;  // (new_lab indexed by pose - lo)
  std::vector<int32_t> old_parent(parent.begin() + lo, parent.begin() + hi), old_pslot(pslot.begin() + lo, pslot.begin() + hi);
  std::fill(parent.begin() + lo, parent.begin() + hi, -1);
  std::fill(pslot.begin() + lo, pslot.begin() + hi, 0);
  new_mem.reserve(n);
  std::vector<int32_t> new_ptr(1, 0);
  for (size_t k = 0; k < alive.size(); ++k) {
    const int a = alive[k];
    if (!grew[a]) {
      // untouched by the merge: the growth's own search started from the aggregate's smallest member (seeds are taken in
      // index order) and claimed exactly these poses in exactly the order the search below would -- keep its tree
      for (int m = ptr[a]; m < ptr[a + 1]; ++m) {
        const int v = mem[m];
        new_lab[v - lo] = (int32_t)k;
        parent[v] = old_parent[v - lo];
        pslot[v] = old_pslot[v - lo];
        new_mem.push_back(v);
      }
      new_ptr.push_back((int32_t)new_mem.size());
      continue;
    }
    // (a merged aggregate is connected by construction, so the search from its smallest member reaches everything; should
    // the pattern not be symmetric, the members it misses become further roots in index order)
    for (int q = sorted_ptr[a]; q < sorted_ptr[a + 1]; ++q) {
      const int root = sorted_mem[q];
      if (new_lab[root - lo] >= 0) continue;
      size_t head = new_mem.size();
      new_lab[root - lo] = (int32_t)k;
      new_mem.push_back(root);
      for (; head < new_mem.size(); ++head) {
        const int u = new_mem[head];
        for (int t = rowptr[u]; t < rowptr[u + 1]; ++t) {
          const int v = colidx[t];
          if (v < lo || v >= hi || lab[v] != a || new_lab[v - lo] >= 0) continue;
          new_lab[v - lo] = (int32_t)k;
          parent[v] = u;
          pslot[v] = t;
          new_mem.push_back(v);
        }
      }
    }
    new_ptr.push_back((int32_t)new_mem.size());
  }
  mem.swap(new_mem);
  ptr.swap(new_ptr);
  std::copy(new_lab.begin(), new_lab.end(), lab.begin() + lo);
  return (int)alive.size();
}
}  // namespace

template <class F>
void parallel_ranges(int n, int chunks, F&& fn);
int setup_threads();

// Symbolic setup: level sizes, block patterns of the Galerkin operators, buffers.
// Graph aggregates of at most S nodes, grown greedily: seeds in index order; a seed's aggregate takes unassigned nodes in
// breadth-first order (queue; a node's neighbours in the order of its block row) until it holds S.  lab = aggregate of
// every node, mem / ptr = members in discovery order, parent / pslot = the breadth-first tree (slot of block
// (parent, node) in the pattern).  cap > 0: followed by the merge of the fragments (below), both inside the index ranges of
// ml_growth_chunks.  Restated in oracle/dpgo_oracle.py (amg_graph_aggregates, amg_merge_small_aggregates).
//
// The merge: the greedy growth leaves fragments (pockets between full aggregates); where an aggregate is a WORKGROUP of the
// one-launch solve (additive preconditioner) every fragment costs a whole workgroup.  Passes over the aggregates in index
// order until nothing changes: an aggregate of at most S / 2 nodes joins the neighbouring aggregate (one it shares a block
// with) it has the most blocks in common with among those that still have room (sizes add up to at most `cap`; ties: the
// lower index).  Afterwards the aggregates are renumbered in the order of their smallest member and every aggregate's
// breadth-first tree is rebuilt from that member (neighbours in block-row order).
int ml_grow_and_merge(const std::vector<int32_t>& rowptr, const std::vector<int32_t>& colidx, int n, int S, int cap,
                      bool do_grow, std::vector<int32_t>& lab, std::vector<int32_t>& ptr, std::vector<int32_t>& mem,
                      std::vector<int32_t>& parent, std::vector<int32_t>& pslot, int chunks) {
  chunks = std::max(1, std::min(chunks, std::max(1, n)));
  if (do_grow) {
    lab.assign(n, -1);
    parent.assign(n, -1);
    pslot.assign(n, 0);
  }
  struct Part {
    std::vector<int32_t> ptr, mem;
    int na = 0;
  };
  std::vector<Part> parts(chunks);
  auto bound = [&](int c) { return (int)((long long)n * c / chunks); };
  if (!do_grow) {
    // the caller's aggregates (ml_graph_aggregates' output: numbered range after range, none across a range boundary) are
    // split by range, ids local to the range; anything else is merged as ONE range
    const int na_in = (int)ptr.size() - 1;
    std::vector<int> first_agg(chunks + 1, na_in);
    bool split_ok = chunks > 1;
    if (split_ok) {
      int c = 0;
      first_agg[0] = 0;
      for (int a = 0; a < na_in && split_ok; ++a) {
        if (ptr[a + 1] <= ptr[a]) { split_ok = false; break; }
        int lo_m = n, hi_m = -1;
        for (int m = ptr[a]; m < ptr[a + 1]; ++m) lo_m = std::min(lo_m, (int)mem[m]), hi_m = std::max(hi_m, (int)mem[m]);
        while (c + 1 < chunks && lo_m >= bound(c + 1)) first_agg[++c] = a;
        if (lo_m < bound(c) || hi_m >= bound(c + 1)) split_ok = false;
      }
      while (c + 1 < chunks) first_agg[++c] = na_in;
      first_agg[chunks] = na_in;
    }
    if (!split_ok) {
      chunks = 1;
      parts.resize(1);
      parts[0].ptr = ptr;
      parts[0].mem = mem;
      parts[0].na = na_in;
    } else {
      for (int c = 0; c < chunks; ++c) {
        const int a0 = first_agg[c], a1 = first_agg[c + 1];
        Part& P = parts[c];
        P.na = a1 - a0;
        P.ptr.assign(1, 0);
        for (int a = a0; a < a1; ++a) P.ptr.push_back(ptr[a + 1] - ptr[a0]);
        P.mem.assign(mem.begin() + ptr[a0], mem.begin() + ptr[a1]);
        if (a0 > 0)
          for (int i = bound(c); i < bound(c + 1); ++i) lab[i] -= a0;
      }
    }
  }
  const bool timing_ = options().setup_timing > 1;
  const auto tA = std::chrono::steady_clock::now();
  std::vector<double> tch(chunks, 0.0);
  parallel_ranges(chunks, std::min(chunks, setup_threads()), [&](int, int c0, int c1) {
    for (int c = c0; c < c1; ++c) {
      const auto t0 = std::chrono::steady_clock::now();
      Part& P = parts[c];
      const int lo = bound(c), hi = bound(c + 1);
      if (do_grow) P.na = grow_range(rowptr, colidx, lo, hi, S, lab, P.ptr, P.mem, parent, pslot);
      if (cap > 0) P.na = merge_range(rowptr, colidx, lo, hi, S, cap, lab, P.ptr, P.mem, parent, pslot);
      tch[c] = 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
  });
  const auto tB = std::chrono::steady_clock::now();
  // the ranges' aggregates one after the other
  std::vector<int> agg_off(chunks + 1, 0);
  std::vector<size_t> mem_off(chunks + 1, 0);
  for (int c = 0; c < chunks; ++c) agg_off[c + 1] = agg_off[c] + parts[c].na, mem_off[c + 1] = mem_off[c] + parts[c].mem.size();
  const int na = agg_off[chunks];
  mem.resize(mem_off[chunks]);
  ptr.assign((size_t)na + 1, 0);
  parallel_ranges(chunks, std::min(chunks, setup_threads()), [&](int, int c0, int c1) {
    for (int c = c0; c < c1; ++c) {
      const Part& P = parts[c];
      if (agg_off[c] > 0)
        for (int i = bound(c); i < bound(c + 1); ++i) lab[i] += agg_off[c];
      for (int a = 0; a < P.na; ++a) ptr[(size_t)agg_off[c] + a + 1] = (int32_t)(mem_off[c] + P.ptr[a + 1]);
      if (!P.mem.empty()) std::memcpy(mem.data() + mem_off[c], P.mem.data(), sizeof(int32_t) * P.mem.size());
    }
  });
  if (timing_) {
    std::fprintf(stderr, "dpgo_hip: grow / merge (%d ranges): parallel section %.3f ms, join %.3f ms; per range:", chunks,
                 1e3 * std::chrono::duration<double>(tB - tA).count(),
                 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - tB).count());
    for (double t : tch) std::fprintf(stderr, " %.3f", t);
    std::fprintf(stderr, "\n");
  }
  return na;
}
int ml_graph_aggregates(const std::vector<int32_t>& rowptr, const std::vector<int32_t>& colidx, int n, int S,
                        std::vector<int32_t>& lab, std::vector<int32_t>& ptr, std::vector<int32_t>& mem,
                        std::vector<int32_t>& parent, std::vector<int32_t>& pslot) {
  return ml_grow_and_merge(rowptr, colidx, n, S, 0, true, lab, ptr, mem, parent, pslot, ml_growth_chunks(n));
}
int ml_merge_small_aggregates(const std::vector<int32_t>& rowptr, const std::vector<int32_t>& colidx, int n, int S, int cap,
                              std::vector<int32_t>& lab, std::vector<int32_t>& ptr, std::vector<int32_t>& mem,
                              std::vector<int32_t>& parent, std::vector<int32_t>& pslot) {
  return ml_grow_and_merge(rowptr, colidx, n, S, cap, false, lab, ptr, mem, parent, pslot, ml_growth_chunks(n));
}

// Host threads of the symbolic set-up (DPGO_SETUP_THREADS; the results do not depend on the count: every parallel section
// below works on disjoint row ranges and is joined in range order).
int setup_threads() {
  const int want = options().setup_threads;
  if (want > 0) return std::min(want, 64);
  const unsigned hw = std::thread::hardware_concurrency();
  return (int)std::max(1u, std::min(8u, hw ? hw : 1u));
}
// fn(chunk, begin, end) over `chunks` contiguous ranges of [0, n) on the process's worker threads (TaskPool, host.h)
template <class F>
void parallel_ranges(int n, int chunks, F&& fn) {
  chunks = std::max(1, std::min(chunks, n > 0 ? n : 1));
  TaskPool::get().run(chunks, [&](int c) {
    fn(c, (int)((long long)n * c / chunks), (int)((long long)n * (c + 1) / chunks));
  }, setup_threads());
}
// rows [lo, hi) of a row-wise pattern built by `row(i, out)` (appends row i's sorted columns) into per-range pieces, joined
// in range order: rowptr / colidx identical to the serial loop
template <class F>
void pattern_by_ranges(int n, int threads, int grain, size_t reserve_hint, F&& row, std::vector<int32_t>& rowptr_out,
                       std::vector<int32_t>& col_out) {
  const int chunks = std::max(1, std::min(threads, n / grain + 1));
  std::vector<std::vector<int32_t>> cols(chunks), cnt(chunks);
  parallel_ranges(n, chunks, [&](int c, int lo, int hi) {
    auto& cc = cols[c];
    auto& nn = cnt[c];
    cc.reserve(reserve_hint / chunks + 16);
    nn.resize(hi - lo);
    for (int i = lo; i < hi; ++i) {
      const size_t first = cc.size();
      row(c, i, cc);
      nn[i - lo] = (int32_t)(cc.size() - first);
    }
  });
  rowptr_out.assign(n + 1, 0);
  size_t total = 0;
  for (auto& cc : cols) total += cc.size();
  col_out.resize(total);
  std::vector<size_t> base(chunks + 1, 0);
  for (int c = 0; c < chunks; ++c) base[c + 1] = base[c] + cols[c].size();
  parallel_ranges(chunks, chunks, [&](int, int c0, int c1) {
    for (int c = c0; c < c1; ++c) {
      const int lo = (int)((long long)n * c / chunks);
      if (!cols[c].empty()) std::memcpy(col_out.data() + base[c], cols[c].data(), sizeof(int32_t) * cols[c].size());
      size_t at = base[c];
      for (size_t k = 0; k < cnt[c].size(); ++k) {
        at += cnt[c][k];
        rowptr_out[lo + k + 1] = (int32_t)at;
      }
    }
  });
}

// Pattern of A P for labelled aggregates: row i holds the sorted, distinct labels of its block columns.  (Inserting into
// the sorted row instead of sort + unique was measured: no gain, the labels' gather is what it costs.)  Row ranges in parallel.
void ml_ap_pattern(const std::vector<int32_t>& rowptr, const std::vector<int32_t>& colidx, const std::vector<int32_t>& lab,
                   int n, std::vector<int32_t>& arow, std::vector<int32_t>& acol) {
  pattern_by_ranges(n, setup_threads(), 1024, colidx.size(), [&](int, int i, std::vector<int32_t>& out) {
    const size_t first = out.size();
    for (int t = rowptr[i]; t < rowptr[i + 1]; ++t) out.push_back(lab[colidx[t]]);
    std::sort(out.begin() + first, out.end());
    out.erase(std::unique(out.begin() + first, out.end()), out.end());
  }, arow, acol);
}

// Runs of equal labels inside chunks of G consecutive poses (what one wave of the level-0 kernels adds up before it
// writes): seg_info[first pose of a run] = 32 * slot + length (-1 elsewhere), the slots ordered by aggregate and, inside
// an aggregate, by pose (a counting sort over the runs), seg_ptr = the aggregates' slot ranges.  Returns the run count.
int ml_run_table(const std::vector<int32_t>& lab, int n, int na, int G, std::vector<int32_t>& seg_info,
                 std::vector<int32_t>& seg_ptr) {
  seg_info.assign(n, -1);
  seg_ptr.assign(na + 1, 0);
  int nruns = 0;
  for (int i = 0; i < n;) {
    int j = i + 1;
    while (j < n && j % G != 0 && lab[j] == lab[i]) ++j;
    seg_info[i] = j - i;  // (length for now)
    seg_ptr[lab[i] + 1] += 1;
    ++nruns;
    i = j;
  }
  for (int a = 0; a < na; ++a) seg_ptr[a + 1] += seg_ptr[a];
  std::vector<int32_t> next(seg_ptr.begin(), seg_ptr.end() - 1);
  for (int i = 0; i < n;) {
    const int len = seg_info[i];
    seg_info[i] = len + 32 * next[lab[i]]++;
    i += len;
  }
  return nruns;
}

// ks_in: aggregate sizes per coarsening; {-S}: two levels, graph aggregates of at most S poses; {-S, -cap}: the same with
// the fragments of the greedy growth merged up to `cap` poses (ml_merge_small_aggregates).  perm_tile > 0 (graph
// aggregates): also build the (aggregate, slot) -> pose table of the additive preconditioner's persistent layout with
// that many slots per aggregate.
int ml_symbolic_setup(dpgo_problem_s* p, const std::vector<int>& ks_in, int perm_tile) {
  // DPGO_SETUP_TIMING=1: host section times on stderr (where the once-per-pattern cost of the hierarchy goes)
  const bool timing = options().setup_timing != 0;
  auto t_last = std::chrono::steady_clock::now();
  const auto t_first = t_last;
  auto lap = [&](const char* what) {
    if (!timing) return;
    const auto t = std::chrono::steady_clock::now();
    std::fprintf(stderr, "dpgo_hip: symbolic set-up: %-44s %7.3f ms\n", what, 1e3 * std::chrono::duration<double>(t - t_last).count());
    t_last = t;
  };
  ml_free(p);
  lap("previous hierarchy freed");
  if ((int)p->h_rowptr.size() != p->n + 1) return fail(DPGO_ERR_STATE, "multilevel: Q's block pattern is not set");
  const int b = p->b, bb = b * b;
  const size_t tb = sizeof(double) * p->T;
  std::vector<int32_t> rowptr = p->h_rowptr, colidx = p->h_colidx;
  int cur = p->n;
  // a single negative entry -S: two levels, graph aggregates of at most S poses; two negative entries: merged up to -ks[1]
  const bool merged = ks_in.size() == 2 && ks_in[0] < 0 && ks_in[1] < 0;
  const bool graph = (ks_in.size() == 1 && ks_in[0] < 0) || merged;
  std::vector<int> ks = ks_in;
  if (merged) ks.pop_back();
  if (graph) ks[0] = -ks_in[0];
  const int merge_cap = merged ? -ks_in[1] : 0;
  if (merged && merge_cap < ks[0]) return fail(DPGO_ERR_INVALID, "multilevel: the merge bound is at least the growth size");
  for (int k : ks)
    if (k < 0) return fail(DPGO_ERR_INVALID, "multilevel: graph aggregates (a negative size) make a two-level hierarchy");
  p->ml.resize(ks.size() + 1);
  for (size_t l = 0; l <= ks.size(); ++l) {
    auto& L = p->ml[l];
    L.n = cur;
    L.split = (l == 0) ? p->split : ml_level_split(cur);
    L.k = (l < ks.size()) ? ks[l] : 0;
    if (l == 0 && graph) {
      if (L.k < 2) return fail(DPGO_ERR_INVALID, "multilevel: graph aggregates hold at least 2 poses");
      std::vector<int32_t> lab, ptr, mem, parent, pslot;
      int na;
      // the level-0 buffers whose sizes are known up front are allocated by a helper thread while the aggregates grow
      hipError_t alloc_err = hipSuccess;
      const int dev_ = p->device;
      const int nthreads = setup_threads();
      JobGuard alloc_job{TaskPool::get().submit(1, [&, dev_](int) {
        hipError_t e = hipSetDevice(dev_);
        if (e == hipSuccess) e = hipMalloc(&L.tbuf, tb * cur);
        if (e == hipSuccess) e = hipMalloc(&L.res1, tb * cur);
        if (e == hipSuccess) e = hipMalloc(&L.Pb, sizeof(double) * (size_t)cur * bb);
        if (e == hipSuccess) e = hipMalloc(&L.x1, tb * cur);
        if (e == hipSuccess) e = hipMalloc(&L.x, tb * cur);
        alloc_err = e;
      }, nthreads)};
      if (p->add_plan_known && p->add_agg.S == L.k && p->add_agg.cap == merge_cap && (int)p->add_agg.lab.size() == cur) {
        const auto& A = p->add_agg;  // (the additive plan of this pattern was found with exactly these aggregates)
        lab = A.lab, ptr = A.ptr, mem = A.mem, parent = A.parent, pslot = A.pslot;
        na = (int)ptr.size() - 1;
      } else {
        na = ml_graph_aggregates(rowptr, colidx, cur, L.k, lab, ptr, mem, parent, pslot);
        lap("greedy growth of the aggregates");
        if (merge_cap) na = ml_merge_small_aggregates(rowptr, colidx, cur, L.k, merge_cap, lab, ptr, mem, parent, pslot);
        lap("merge of the growth's fragments");
      }
      L.graph = true;
      L.merge_cap = merge_cap;
      // the patterns that follow from the labels -- A P (rows in parallel) and the dense level's operator (aggregates in
      // parallel) -- are built by worker threads while this one uploads the labels and builds the run and tile tables
      std::vector<int32_t> arow, acol, crow, ccol;
      JobGuard pattern_job{TaskPool::get().submit(2, [&](int which) {
        if (which == 0) {  // pattern of A P: the aggregates every row's block columns fall into
          ml_ap_pattern(rowptr, colidx, lab, cur, arow, acol);
          return;
        }
        // pattern of the dense level's operator: the aggregates of the block columns of every member's row
        const int chunks = std::max(1, std::min(std::max(1, nthreads / 2), na / 16 + 1));
        std::vector<std::vector<int32_t>> marks(chunks);
        pattern_by_ranges(na, chunks, 16, colidx.size() / 8, [&](int c, int a, std::vector<int32_t>& out) {
          auto& mark = marks[c];
          if (mark.empty()) mark.assign(na, -1);
          const size_t first = out.size();
          for (int m = ptr[a]; m < ptr[a + 1]; ++m) {
            const int i = mem[m];
            for (int t = rowptr[i]; t < rowptr[i + 1]; ++t) {
              const int cl = lab[colidx[t]];
              if (mark[cl] != a) {
                mark[cl] = a;
                out.push_back(cl);
              }
            }
          }
          std::sort(out.begin() + first, out.end());
        }, crow, ccol);
      }, nthreads)};
      CHK(upload(&L.lab, lab.data(), lab.size(), p->stream));
      CHK(upload(&L.agg_ptr, ptr.data(), ptr.size(), p->stream));
      CHK(upload(&L.agg_mem, mem.data(), mem.size(), p->stream));
      CHK(upload(&L.parent, parent.data(), parent.size(), p->stream));
      CHK(upload(&L.pslot, pslot.data(), pslot.size(), p->stream));
      std::vector<int32_t> mpos(cur);
      for (int m = 0; m < cur; ++m) mpos[mem[m]] = m;
      CHK(upload(&L.mem_pos, mpos.data(), mpos.size(), p->stream));
      lap("labels, members, trees uploaded");
      std::vector<int32_t> seg_info, seg_ptr;
      L.nseg = ml_run_table(lab, cur, na, 64 / (b * L.split), seg_info, seg_ptr);
      CHK(upload(&L.seg_info, seg_info.data(), seg_info.size(), p->stream));
      CHK(upload(&L.seg_ptr, seg_ptr.data(), seg_ptr.size(), p->stream));
      std::vector<int32_t> tperm;
      // the layout of the additive preconditioner's persistent kernel: aggregate = workgroup tile of `perm_tile` slots
      if (!perm_tile && !merge_cap && L.k == additive_tile(p)) perm_tile = L.k;
      if (perm_tile) {
        if (std::max(L.k, merge_cap) > perm_tile) return fail(DPGO_ERR_INVALID, "multilevel: aggregates larger than the tile");
        tperm.assign((size_t)na * perm_tile, -1);
        for (int a = 0; a < na; ++a)
          for (int m = ptr[a]; m < ptr[a + 1]; ++m) tperm[(size_t)a * perm_tile + (m - ptr[a])] = mem[m];
        CHK(upload(&L.tile_perm, tperm.data(), tperm.size(), p->stream));
        L.perm_tile = perm_tile;
      }
      lap("run table, tile table");
      TaskPool::get().wait(pattern_job.job);
      lap("patterns of A P and of the dense level joined");
      CHK(upload_bsr(L.AP, cur, na, (int)acol.size(), b, arow.data(), acol.data(), nullptr, p->stream));
      lap("A P allocated, pattern uploaded");
      TaskPool::get().wait(alloc_job.job);
      if (alloc_err != hipSuccess) return fail(DPGO_ERR_HIP, std::string("hipMalloc (level-0 buffers): ") + hipGetErrorString(alloc_err));
      lap("level-0 vectors joined");
      HIPC(hipStreamSynchronize(p->stream));  // the host vectors go out of scope
      lap("stream synchronised");
      rowptr.swap(crow);
      colidx.swap(ccol);
      cur = na;
      continue;
    }
    if (L.k) {
      if (L.k < 2 || ml_tile(b, L.split) % L.k)
        return fail(DPGO_ERR_INVALID, "multilevel: aggregate size must divide the workgroup tile of its level (" +
                                          std::to_string(ml_tile(b, L.split)) + " nodes)");
    }
    if (l > 0) {
      CHK(upload_bsr(L.A, cur, cur, (int)colidx.size(), b, rowptr.data(), colidx.data(), nullptr, p->stream));
      std::vector<int32_t> srow(colidx.size());
      for (int i = 0; i < cur; ++i)
        for (int t = rowptr[i]; t < rowptr[i + 1]; ++t) srow[t] = i;
      CHK(upload(&L.slot_row, srow.data(), srow.size(), p->stream));
      HIPC(hipStreamSynchronize(p->stream));  // srow goes out of scope at the end of this block
      HIPC(hipMalloc(&L.r, tb * cur));
      if (!L.k) HIPC(hipMalloc(&L.x, tb * cur));  // dense level: its solution, read by the level above
    }
    if (l == 0 && L.k) {  // pattern of A P: the aggregates the block columns of every row fall into
      const int k = L.k;
      std::vector<int32_t> arow(cur + 1, 0), acol;
      acol.reserve(colidx.size());
      for (int i = 0; i < cur; ++i) {
        const size_t first = acol.size();
        for (int t = rowptr[i]; t < rowptr[i + 1]; ++t) acol.push_back(colidx[t] / k);
        std::sort(acol.begin() + first, acol.end());
        acol.erase(std::unique(acol.begin() + first, acol.end()), acol.end());
        arow[i + 1] = (int32_t)acol.size();
      }
      CHK(upload_bsr(L.AP, cur, (cur + k - 1) / k, (int)acol.size(), b, arow.data(), acol.data(), nullptr, p->stream));
      HIPC(hipMalloc(&L.res1, tb * cur));
    }
    if (L.k) {
      if (l > 0) HIPC(hipMalloc(&L.dinv, sizeof(double) * (size_t)cur * bb));
      HIPC(hipMalloc(&L.Pb, sizeof(double) * (size_t)cur * bb));
      HIPC(hipMalloc(&L.x1, tb * cur));
      HIPC(hipMalloc(&L.x, tb * cur));
      // pattern of the next level: block columns j / k of the rows of every aggregate
      const int k = L.k, nc = (cur + k - 1) / k;
      std::vector<int32_t> crow(nc + 1, 0), ccol;
      std::vector<int32_t> mark(nc, -1);
      for (int a = 0; a < nc; ++a) {
        const size_t first = ccol.size();
        for (int i = a * k; i < std::min(cur, a * k + k); ++i)
          for (int t = rowptr[i]; t < rowptr[i + 1]; ++t) {
            const int c = colidx[t] / k;
            if (mark[c] != a) {
              mark[c] = a;
              ccol.push_back(c);
            }
          }
        std::sort(ccol.begin() + first, ccol.end());
        crow[a + 1] = (int32_t)ccol.size();
      }
      rowptr.swap(crow);
      colidx.swap(ccol);
      cur = nc;
    }
  }
  const int N = cur * b;
  if (N > 16384) return fail(DPGO_ERR_INVALID, "multilevel: dense coarsest operator too large (" + std::to_string(N) +
                                                   " unknowns): use more levels / larger aggregates");
  p->ml_lda = ((N + kNB - 1) / kNB) * kNB;
  // + 8 rows: the apply kernel reads (and discards) the rows of a ghost node behind a ragged last node group
  HIPC(hipMalloc(&p->ml_dense, sizeof(double) * (size_t)p->ml_lda * (p->ml_lda + 8)));
  HIPC(hipMalloc(&p->ml_dense32, sizeof(float) * (size_t)p->ml_lda * (p->ml_lda + 8)));
  if (p->ml.size() == 2) {  // two levels: the packed lower triangle and the bookkeeping of k_dense_sym_apply
    const int nT = p->ml_lda / kNB;
    int chunk = kDenseChunk;
    if (options().dense_chunk > 0) chunk = options().dense_chunk;  // tuning knob
    std::vector<DenseChunk> chunks;
    std::vector<int> first(nT + 1, 0);
    for (int I = 0; I < nT; ++I) {
      first[I] = (int)chunks.size();
      for (int J0 = 0; J0 <= I; J0 += chunk) chunks.push_back(DenseChunk{I, J0, std::min(chunk, I + 1 - J0), 0});
    }
    first[nT] = (int)chunks.size();
    p->ml_nchunks = (int)chunks.size();
    CHK(upload(&p->ml_chunks, chunks.data(), chunks.size(), p->stream));
    CHK(upload(&p->ml_chunk_first, first.data(), first.size(), p->stream));
    HIPC(hipMalloc(&p->ml_packed, sizeof(double) * (size_t)nT * (nT + 1) / 2 * kNB * kNB));
    HIPC(hipMalloc(&p->ml_pd, sizeof(double) * (size_t)p->ml_nchunks * kNB * p->r));
    HIPC(hipMalloc(&p->ml_pt, sizeof(double) * (size_t)nT * p->ml_lda * p->r));
    HIPC(hipStreamSynchronize(p->stream));  // the host vectors go out of scope
  }
  HIPC(hipMalloc(&p->ml_W, sizeof(double) * (size_t)p->ml_lda * kNB));
  HIPC(hipMalloc(&p->ml_Rx, sizeof(double) * (size_t)p->ml_lda * kNB));
  HIPC(hipStreamSynchronize(p->stream));
  lap("coarser levels, dense level allocated");
  if (timing)
    std::fprintf(stderr, "dpgo_hip: symbolic set-up: %-44s %7.3f ms\n", "total",
                 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t_first).count());
  p->ml_symbolic = true;
  return DPGO_OK;
}

int flat_grid(size_t items) {
  size_t g = (items + kBlock - 1) / kBlock;
  if (g < 1) g = 1;
  return g < (size_t)kMaxGrid ? (int)g : kMaxGrid;
}

bool gj_use_mfma() {
  return options().gj_mfma != 0;
}

// In-place inverse of the dense SPD lda x lda array M (lda a multiple of 64); W, Rx: lda x 64 panels.
// (Round 6, both measured at 2 184 unknowns = 35 block steps and NOT kept: a look-ahead schedule -- the tiles of block row /
// column kb + 1 first, the rest of the rank-64 update in one launch with the next panel, W / Rx double-buffered, bitwise the
// same result -- 2.80 ms against 2.65 ms: the panel's latency-bound pivot loop runs slower beside 600 update workgroups than
// the launch it saves; the 64 x 64 pivot block by four block steps of 16 pivots -- one wavefront inverting the 16 x 16 block
// with wave barriers only, rank-16 updates on 4 x 4 register tiles, reciprocal by v_rcp_f64 + two Newton steps -- 2.96 ms:
// a pivot's dependent chain (LDS line, reciprocal, quad shuffle, LDS write, barrier) costs the same 0.7 us in a wavefront as
// in a workgroup.)
int dense_spd_inverse(hipStream_t s, double* M, int lda, double* W, double* Rx, bool mfma) {
  const int nt = lda / kNB;
  for (int kb = 0; kb < nt; ++kb) {
    hipLaunchKernelGGL(k_sweep_panel, dim3(nt), dim3(kBlock), 0, s, M, lda, kb, W, Rx);
    if (mfma)
      hipLaunchKernelGGL(k_sweep_update<true>, dim3(nt, nt), dim3(kBlock), 0, s, M, lda, kb, W, Rx);
    else
      hipLaunchKernelGGL(k_sweep_update<false>, dim3(nt, nt), dim3(kBlock), 0, s, M, lda, kb, W, Rx);
  }
  hipLaunchKernelGGL(k_sweep_finish, dim3(nt, nt), dim3(kBlock), 0, s, M, lda);
  HIPC(hipGetLastError());
  return DPGO_OK;
}

template <int D>
int ml_numeric_setup_d(dpgo_problem_s* p) {
  const int nl = (int)p->ml.size();
  long long stride = 1;
  for (int l = 0; l + 1 < nl; ++l) {
    auto& L = p->ml[l];
    auto& C = p->ml[l + 1];
    const long long span = stride * L.k;
    // (wave-parallel forms of the two setup kernels that walked an aggregate's members with ONE thread; DPGO_ML_SETUP_SERIAL=1
    // restores them)
    const bool serial = options().ml_setup_serial != 0;
    auto wave_grid = [](int items) { return std::max(1, std::min(kMaxGrid, (items + kWaves - 1) / kWaves)); };
    if (L.graph && !serial)
      hipLaunchKernelGGL(k_ml_build_P_tree_wave<D>, dim3(wave_grid(C.n)), dim3(kBlock), 0, p->stream, p->Q.dev(), L.agg_ptr,
                         L.agg_mem, L.parent, L.pslot, L.mem_pos, L.Pb, C.n);
    else if (L.graph)
      hipLaunchKernelGGL(k_ml_build_P_tree<D>, dim3(flat_grid(C.n)), dim3(kBlock), 0, p->stream, p->Q.dev(), L.agg_ptr,
                         L.agg_mem, L.parent, L.pslot, L.Pb, C.n);
    else
      hipLaunchKernelGGL(k_ml_build_P<D>, dim3(flat_grid(C.n)), dim3(kBlock), 0, p->stream, p->Q.dev(), p->n, (int)stride,
                         (int)span, L.Pb, C.n);
    const BsrDev A = (l == 0) ? p->Q.dev() : L.A.dev();
    const bool have_ap = l == 0 && L.AP.vals;
    if (have_ap)  // A P first: the Galerkin operator of a two-level hierarchy is its restriction
      hipLaunchKernelGGL(k_ml_build_AP<D>, dim3(flat_grid(L.n)), dim3(kBlock), 0, p->stream, p->Q.dev(), p->ml_shift, L.Pb,
                         L.agg(), L.n, L.AP.dev(), L.AP.vals);
    if (have_ap && !serial)
      hipLaunchKernelGGL(k_ml_galerkin_ap<D>, dim3(wave_grid(C.A.nnzb)), dim3(kBlock), 0, p->stream, L.AP.dev(), L.Pb,
                         L.agg(), L.agg_ptr, L.agg_mem, L.n, C.slot_row, C.A.colidx, C.A.vals, C.A.nnzb);
    else
      hipLaunchKernelGGL(k_ml_galerkin<D>, dim3(flat_grid(C.A.nnzb)), dim3(kBlock), 0, p->stream, A,
                         (l == 0) ? p->ml_shift : 0.0, L.Pb, L.agg(), L.agg_ptr, L.agg_mem, L.n, C.slot_row, C.A.colidx,
                         C.A.vals, C.A.nnzb);
    if (C.k)  // smoother of the next level (level 0 uses the handle's block-Jacobi factors)
      hipLaunchKernelGGL(k_build_dinv<D>, dim3(flat_grid(C.n)), dim3(kBlock), 0, p->stream, C.A.dev(), 0.0, C.dinv, C.n);
    stride = span;
  }
  HIPC(hipGetLastError());
  auto& Lc = p->ml.back();
  const int lda = p->ml_lda, N = Lc.n * p->b;
  HIPC(hipMemsetAsync(p->ml_dense, 0, sizeof(double) * (size_t)lda * (lda + 8), p->stream));
  hipLaunchKernelGGL(k_dense_pad_identity, dim3(1), dim3(kBlock), 0, p->stream, p->ml_dense, lda, N);
  hipLaunchKernelGGL(k_ml_dense_assemble<D>, dim3(flat_grid(Lc.A.nnzb)), dim3(kBlock), 0, p->stream, Lc.A.dev(),
                     Lc.slot_row, p->ml_dense, lda, Lc.A.nnzb);
  HIPC(hipGetLastError());
  CHK(dense_spd_inverse(p->stream, p->ml_dense, lda, p->ml_W, p->ml_Rx, gj_use_mfma()));
  if (p->ml_packed) {
    const int nT = lda / kNB;
    hipLaunchKernelGGL(k_dense_pack_lower, dim3(nT, nT), dim3(kBlock), 0, p->stream, p->ml_dense, lda, p->ml_packed);
    HIPC(hipGetLastError());
  }
  {  // the fp32 storage of the inverse: what the cycle streams when the dense level is kept in fp32 -- by request
     // (ml_coarse_bits == 32: the fp64 array then keeps the SAME rounded values, so that what dpgo_problem_multilevel_get
     // returns is what the cycle applies) or together with the rest of the cycle's storage (coarse32_active: the fp64
     // array stays exact, the cycle applies its rounding)
    const size_t total = (size_t)lda * (lda + 8);
    if (p->ml_coarse_bits == 32)
      hipLaunchKernelGGL(k_dense_round_f32, dim3(flat_grid(total)), dim3(kBlock), 0, p->stream, p->ml_dense, p->ml_dense32,
                         total);
    else
      hipLaunchKernelGGL(k_copy_f32, dim3(flat_grid(total)), dim3(kBlock), 0, p->stream, p->ml_dense, p->ml_dense32, total);
    HIPC(hipGetLastError());
  }
  return DPGO_OK;
}

// Numeric setup for the CURRENT values of Q (device only; redone after every re-weighting).
int ml_numeric_setup(dpgo_problem_s* p) {
  if (!p->ml_symbolic) return fail(DPGO_ERR_STATE, "multilevel: symbolic setup missing");
  CHK(build_dinv(p, p->ml_shift));
  if (p->d == 2)
    CHK(ml_numeric_setup_d<2>(p));
  else
    CHK(ml_numeric_setup_d<3>(p));
  p->ml_ready = true;
  p->ml_ops32_ready = false;  // (the fp32 copies of A P and the prolongation follow at the next solve that streams them)
  return DPGO_OK;
}

// The hierarchy's shape in the form ml_symbolic_setup takes it.
std::vector<int> ml_current_ks(const dpgo_problem_s* p) {
  std::vector<int> ks;
  for (size_t l = 0; l + 1 < p->ml.size(); ++l) ks.push_back(p->ml[l].graph ? -p->ml[l].k : p->ml[l].k);
  if (p->ml.size() == 2 && p->ml[0].graph && p->ml[0].merge_cap) ks.push_back(-p->ml[0].merge_cap);
  return ks;
}

// Layout of the additive preconditioner inside the one-launch solve (k_rtr_persist<..., ADD>): ONE aggregate per workgroup,
// at most kPersistMax = 256 of them.  Host only, once per block pattern:
//   1. graph aggregates of at most one 4-lane-group tile (16 poses in 3-D), plain greedy growth, while there are <= 256
//      (blocks up to ~3 500 poses: the lowest-latency layout);
//   2. else one pose per (d+1) lanes (tile = 64 poses in 3-D): graph aggregates grown to S poses, fragments merged up to
//      min(tile, 3 S / 2), with the smallest S (from ceil(n / 230) in steps of an eighth) that leaves <= 256 aggregates
//      (12 500-pose slab: S = 55, 230 aggregates; 6 250-pose grid: S = 28, 221) -- blocks up to ~14 000 poses;
//   3. without graph aggregates (DPGO_ML_GRAPH=0): index runs of one tile.
const dpgo_problem_s::AddPlan& additive_plan(dpgo_problem_s* p) {
  if (p->add_plan_known) return p->add_plan;
  p->add_plan = dpgo_problem_s::AddPlan();
  p->add_plan_known = true;
  p->add_agg = dpgo_problem_s::AggCache();
  if (p->split != 4 || (int)p->h_rowptr.size() != p->n + 1) return p->add_plan;
  const int P4 = ml_tile(p->b, 4), P1 = ml_tile(p->b, 1), n = p->n;
  const bool graph_ok = options().ml_graph != 0;
  if (graph_ok) {
    auto& A = p->add_agg;
    auto &lab = A.lab, &ptr = A.ptr, &mem = A.mem, &parent = A.parent, &pslot = A.pslot;
    if ((long long)n <= (long long)kPersistMax * P4) {
      const int na = ml_graph_aggregates(p->h_rowptr, p->h_colidx, n, P4, lab, ptr, mem, parent, pslot);
      if (na <= kPersistMax) {
        p->add_plan = dpgo_problem_s::AddPlan{4, P4, P4, 0, na, true};
        A.S = P4, A.cap = 0;
        return p->add_plan;
      }
    }
    if ((long long)n <= (long long)kPersistMax * P1) {
      // A handle that is solved next to other handles of the device (dpgo_optimize_device_many: persist_share > 1 when the
      // plan is first asked for) aims at HALF the chip -- every aggregate is a workgroup that owns a CU for the whole solve,
      // so two such solves run side by side instead of taking turns; the product count is a weak function of the aggregate
      // size, the time of an iteration is not a function of how full the tiles are.
      const int want = (p->persist_share > 1 && (long long)n * 10 <= (long long)(kPersistMax / 2) * P1 * 8) ? kPersistMax / 2 : kPersistMax;
      for (int S = std::max(8, (n + (want * 9) / 10 - 1) / ((want * 9) / 10)); S <= P1; S += std::max(2, S / 8)) {
        const int cap = std::min(P1, S + S / 2);
        ml_graph_aggregates(p->h_rowptr, p->h_colidx, n, S, lab, ptr, mem, parent, pslot);
        const int na = ml_merge_small_aggregates(p->h_rowptr, p->h_colidx, n, S, cap, lab, ptr, mem, parent, pslot);
        if (na <= want) {
          p->add_plan = dpgo_problem_s::AddPlan{1, P1, S, cap, na, true};
          A.S = S, A.cap = cap;
          return p->add_plan;
        }
      }
      if (want < kPersistMax) {  // (no growth size reaches half the chip: the whole-chip plan)
        for (int S = std::max(8, (n + 229) / 230); S <= P1; S += std::max(2, S / 8)) {
          const int cap = std::min(P1, S + S / 2);
          ml_graph_aggregates(p->h_rowptr, p->h_colidx, n, S, lab, ptr, mem, parent, pslot);
          const int na = ml_merge_small_aggregates(p->h_rowptr, p->h_colidx, n, S, cap, lab, ptr, mem, parent, pslot);
          if (na <= kPersistMax) {
            p->add_plan = dpgo_problem_s::AddPlan{1, P1, S, cap, na, true};
            A.S = S, A.cap = cap;
            return p->add_plan;
          }
        }
      }
    }
    p->add_agg = dpgo_problem_s::AggCache();
  }
  if ((n + P4 - 1) / P4 <= kPersistMax)
    p->add_plan = dpgo_problem_s::AddPlan{4, P4, P4, 0, (n + P4 - 1) / P4, false};
  else if ((n + P1 - 1) / P1 <= kPersistMax)
    p->add_plan = dpgo_problem_s::AddPlan{1, P1, P1, 0, (n + P1 - 1) / P1, false};
  return p->add_plan;
}

// lane groups per pose of the additive layout the CURRENT two-level hierarchy fits (0: none)
int additive_split_of(const dpgo_problem_s* p) {
  if (!p->ml_symbolic || p->ml.size() != 2 || p->split != 4 || p->ml[1].n > kPersistMax) return 0;
  const auto& L = p->ml[0];
  const int P4 = ml_tile(p->b, 4), P1 = ml_tile(p->b, 1);
  const int tile = L.graph ? (L.tile_perm ? L.perm_tile : 0) : L.k;
  return tile == P4 ? 4 : (tile == P1 ? 1 : 0);
}

// Make the hierarchy match the handle's Q (lazily, like the reference's constructPreconditioner inside the first
// PreConditioner call, src/PoseGraph.cpp:582-586).
int ml_ensure(dpgo_problem_s* p, double shift, bool additive) {
  // the additive preconditioner needs ONE aggregate per workgroup tile of its persistent layout (two levels); a hierarchy
  // the caller set up explicitly is kept if it has that shape, the default one is replaced by the handle's plan
  // (additive_plan) and put back when the V-cycle is asked for again
  if (additive && !additive_split_of(p)) {
    const auto& plan = additive_plan(p);
    if (!plan.split) return fail(DPGO_ERR_UNSUPPORTED, "additive preconditioner: the block does not fit 256 aggregates of one workgroup tile");
    std::vector<int> ks{plan.graph ? -plan.S : plan.S};
    if (plan.graph && plan.cap) ks.push_back(-plan.cap);
    CHK(ml_symbolic_setup(p, ks, plan.graph ? plan.tile : 0));
    if (!additive_split_of(p)) return fail(DPGO_ERR_STATE, "additive preconditioner: hierarchy does not match its plan");
    p->ml_additive_layout = true;
    p->ml_user_ks = false;
  } else if (!additive && p->ml_additive_layout && !p->ml_user_ks) {
    CHK(ml_symbolic_setup(p, ml_default_ks(p->n, p->b, p->split)));
    p->ml_additive_layout = false;
  }
  // level 0 smooths with the handle's shared block-Jacobi factors: a block-Jacobi solve with another shift in between
  // has overwritten them, so they are re-derived for THIS shift even when the hierarchy itself is current (no-op otherwise)
  if (p->ml_ready && p->ml_shift == shift) return build_dinv(p, shift);
  if (!p->ml_symbolic) CHK(ml_symbolic_setup(p, ml_default_ks(p->n, p->b, p->split)));
  p->ml_shift = shift;
  return ml_numeric_setup(p);
}

// The fp32 operator copies of the cycle (dpgo_problem_s::ml_operator_bits): (re)built from the fp64 originals whenever those
// changed -- after the hierarchy's numeric set-up, after the symmetric copy of Q was refreshed -- by the first solve that
// wants them (call after resolve_tcg_storage and ml_ensure; never inside a recorded iteration).
int ml_ops32_ensure(dpgo_problem_s* p) {
  if (!p->ml_ops32_wanted() || p->ml_ops32_ready) return DPGO_OK;
  auto& L0 = p->ml[0];
  const size_t bb = (size_t)p->b * p->b;
  const size_t nu = (size_t)p->sym.nu * bb, nap = (size_t)L0.AP.nnzb * bb, npb = (size_t)L0.n * bb;
  if (!p->sym.uvalsT32) HIPC(hipMalloc(&p->sym.uvalsT32, sizeof(float) * nu));
  if (!L0.AP32) HIPC(hipMalloc(&L0.AP32, sizeof(float) * nap));
  if (!L0.Pb32) HIPC(hipMalloc(&L0.Pb32, sizeof(float) * npb));
  if (!L0.x1f) HIPC(hipMalloc(&L0.x1f, sizeof(float) * (size_t)L0.n * p->T));
  if (!L0.res1f) HIPC(hipMalloc(&L0.res1f, sizeof(float) * (size_t)L0.n * p->T));
  auto copy = [&](const double* in, float* out, size_t total) {
    hipLaunchKernelGGL(k_copy_f32, dim3(flat_grid(total)), dim3(kBlock), 0, p->stream, in, out, total);
  };
  copy(p->sym.uvalsT, p->sym.uvalsT32, nu);
  copy(L0.AP.vals, L0.AP32, nap);
  copy(L0.Pb, L0.Pb32, npb);
  HIPC(hipGetLastError());
  p->ml_ops32_ready = true;
  return DPGO_OK;
}

// Dense level + prolongation.  Large coarsest levels: two nodes per workgroup (halves the right-hand-side loads per
// matrix byte); balanced rounds: every workgroup takes the same number of node groups (a ragged last round would leave
// most of the chip idle while the dense inverse streams).
int persist_capacity(int device);  // (two resident slots per CU; below)
int launch_coarse_prolong(dpgo_problem_s* p, const dpgo_problem_s::MlLevel& L, const dpgo_problem_s::MlLevel& C,
                          const DevState* gate, double* xc_out) {
  const bool f32 = p->coarse32_active();
  // nodes per workgroup (the right-hand side is read once per workgroup): 732 nodes: 1 -> 2: 19.7 -> 17.1 us, 4: 18.1;
  // three (fp64 storage) where that brings the level down to one workgroup per CU in one round: 546 nodes: 2 -> 3:
  // 273 -> 182 workgroups, 13.2 -> 11.6 us, the 100k bench step 4.45 -> 4.33 ms
  const int cus = persist_capacity(p->device) / 2;
  int nodes = C.n >= 512 ? 2 : 1;
  if (nodes == 2 && !f32 && (C.n + 1) / 2 > cus && (C.n + 2) / 3 <= cus) nodes = 3;
  if (const int v = options().coarse_nodes)  // tuning knob
    nodes = (v == 4 || v == 2 || (v == 3 && !f32)) ? v : 1;
  const int groups = (C.n + nodes - 1) / nodes;
  int cap = kMaxGrid;
  if (options().coarse_grid > 0) cap = options().coarse_grid;  // tuning knob
  const int rounds = (groups + cap - 1) / cap;
  const int gc = std::max(1, (groups + rounds - 1) / rounds);
  // non-temporal loads of the inverse whenever the loop's working set does not fit the Infinity Cache (kernel comment)
  int hint = p->beyond_cache();
  if (options().coarse_nt >= 0) hint = options().coarse_nt != 0;  // tuning knob
#define COARSE_LAUNCH(NODES, MT, MPTR)                                                                               \
  hipLaunchKernelGGL((k_ml_coarse_prolong<D, R, NODES, MT>), dim3(gc), dim3(kBlock), 0, p->stream, MPTR, p->ml_lda,   \
                     reinterpret_cast<const MT*>(C.r), L.x1, L.Pb, L.k, L.x, gate, L.n, C.n, xc_out, hint)
  DISPATCH(p->d, p->r, {
    if (nodes == 4 && f32)
      COARSE_LAUNCH(4, float, p->ml_dense32);
    else if (nodes == 4)
      COARSE_LAUNCH(4, double, p->ml_dense);
    else if (nodes == 3)
      COARSE_LAUNCH(3, double, p->ml_dense);
    else if (nodes == 2 && f32)
      COARSE_LAUNCH(2, float, p->ml_dense32);
    else if (nodes == 2)
      COARSE_LAUNCH(2, double, p->ml_dense);
    else if (f32)
      COARSE_LAUNCH(1, float, p->ml_dense32);
    else
      COARSE_LAUNCH(1, double, p->ml_dense);
  });
#undef COARSE_LAUNCH
  HIPC(hipGetLastError());
  return DPGO_OK;
}

// Dense level from the packed lower triangle: xc = A_c^-1 rc into C.x (two launches: partial products, fixed-order sums)
int launch_dense_sym(dpgo_problem_s* p, const dpgo_problem_s::MlLevel& C, const DevState* gate) {
  const int N = C.n * p->b, nT = p->ml_lda / kNB;
  switch (p->r) {
#define DENSE_SYM_CASE(RR)                                                                                              \
  case RR:                                                                                                              \
    hipLaunchKernelGGL((k_dense_sym_apply<RR>), dim3(p->ml_nchunks), dim3(kBlock), 0, p->stream, p->ml_packed,          \
                       p->ml_chunks, C.r, N, p->ml_lda, p->ml_pd, p->ml_pt, gate);                                       \
    hipLaunchKernelGGL((k_dense_sym_finish<RR>), dim3(nT, 4), dim3(kBlock), 0, p->stream, p->ml_pd, p->ml_pt,            \
                       p->ml_chunk_first, nT, N, p->ml_lda, C.x, gate);                                                  \
    break;
    DENSE_SYM_CASE(2)
    DENSE_SYM_CASE(3)
    DENSE_SYM_CASE(4)
    DENSE_SYM_CASE(5)
    DENSE_SYM_CASE(6)
#undef DENSE_SYM_CASE
    default:
      return fail(DPGO_ERR_UNSUPPORTED, "unsupported r");
  }
  HIPC(hipGetLastError());
  return DPGO_OK;
}

// Level-0 restriction of the cycle: rc = P^T (r - A x1) into ml[1].r (+ the residual itself for k_ml_post_ap).  Graph
// aggregates: the restriction kernel adds P_i^T res_i up over every run of same-aggregate poses inside a wave's chunk and
// writes one partial sum per run, k_ml_agg_sum adds an aggregate's partial sums up.
int launch_ml_restrict0(dpgo_problem_s* p, const double* r, const DevState* gate, int g0, bool stop_check) {
  auto& L = p->ml[0];
  auto& C = p->ml[1];
  // inside the tCG loop (not after its first update): tCG's residual test one kernel early (TcgStopCheck, multilevel.h);
  // the <r,r> partial sums are the ones k_tcg_update wrote, one per workgroup of ITS grid
  TcgStopCheck stop;
  if (stop_check && gate) {
    stop.state = const_cast<DevState*>(gate);
    stop.pin = p->pB();
    stop.nb = p->grid_u(true);  // (k_tcg_update_span's multilevel-mode instance wrote them)
    stop.hflag = p->hflag;
    stop.gen = p->launch_gen();
  }
  float* rc32 = (C.k == 0 && p->coarse32_active()) ? reinterpret_cast<float*>(C.r) : (float*)nullptr;
  double* res_out = p->ml_use_ap() ? L.res1 : nullptr;
  const double* dnext = C.k ? C.dinv : (const double*)nullptr;
  if (p->ml_vec32_active()) {  // ... its fp32 copy (and the prolongation's), the cycle's internal vectors in fp32 as well
    DISPATCH(p->d, p->r, hipLaunchKernelGGL((k_ml_restrict<D, R, 1, BsrSymDev32, float, float>), dim3(g0), dim3(kBlock), 0,
                                            p->stream, p->sym.dev32(), L.x1f, r, L.Pb32, p->ml_shift, L.k, C.r, rc32, dnext,
                                            p->ml_omega, C.x1, gate, L.n, L.res1f, L.tbuf, L.seg_info, stop));
  } else if (p->ml_ops32_active()) {  // (A/B: fp32 operator copies, fp64 vectors -- DPGO_ML_VECTOR_BITS=64)
    DISPATCH(p->d, p->r, hipLaunchKernelGGL((k_ml_restrict<D, R, 1, BsrSymDev32, float, double>), dim3(g0), dim3(kBlock), 0,
                                            p->stream, p->sym.dev32(), L.x1, r, L.Pb32, p->ml_shift, L.k, C.r, rc32, dnext,
                                            p->ml_omega, C.x1, gate, L.n, res_out, L.tbuf, L.seg_info, stop));
  } else if (p->tcg_sym) {  // level 0 reads Q: the symmetric copy when the tCG-step kernel does
    DISPATCH(p->d, p->r, hipLaunchKernelGGL((k_ml_restrict<D, R, 1, BsrSymDev>), dim3(g0), dim3(kBlock), 0, p->stream,
                                            p->sym.dev(), L.x1, r, L.Pb, p->ml_shift, L.k, C.r, rc32, dnext, p->ml_omega,
                                            C.x1, gate, L.n, res_out, L.tbuf, L.seg_info, stop));
  } else {
    DISPATCH(p->d, p->r, LAUNCH_SPLIT(p, k_ml_restrict, g0, p->Q.dev(), L.x1, r, L.Pb, p->ml_shift, L.k, C.r, rc32, dnext,
                                      p->ml_omega, C.x1, gate, L.n, res_out, L.tbuf, L.seg_info, stop));
  }
  if (L.graph)
    DISPATCH(p->d, p->r, hipLaunchKernelGGL((k_ml_agg_sum<D, R>), dim3(std::min(C.n, kMaxGrid)), dim3(kBlock), 0, p->stream,
                                            L.tbuf, L.seg_ptr, C.n, C.r, rc32, gate));
  HIPC(hipGetLastError());
  return DPGO_OK;
}

// Level-0 post-smoothing of a two-level hierarchy through A P (k_ml_post_ap).
int launch_ml_post_ap(dpgo_problem_s* p, const double* Xdev, const double* r, double* z, double* pout, const DevState* gate) {
  auto& L0 = p->ml[0];
  if (p->ml_ops32_active()) {
    const BsrDev32 ap32{L0.AP.rowptr, L0.AP.colidx, L0.AP32};
    if (p->ml_vec32_active()) {
      DISPATCH(p->d, p->r, hipLaunchKernelGGL((k_ml_post_ap<D, R, 1, float, float>), dim3(p->grid_post()), dim3(kBlock), 0,
                                              p->stream, ap32, Xdev, r, L0.res1f, p->ml[1].x, L0.Pb32, L0.agg(), p->dinv,
                                              p->ml_omega, z, pout, gate, p->n));
    } else {
      DISPATCH(p->d, p->r, hipLaunchKernelGGL((k_ml_post_ap<D, R, 1, float, double>), dim3(p->grid_post()), dim3(kBlock), 0,
                                              p->stream, ap32, Xdev, r, L0.res1, p->ml[1].x, L0.Pb32, L0.agg(), p->dinv,
                                              p->ml_omega, z, pout, gate, p->n));
    }
    HIPC(hipGetLastError());
    return DPGO_OK;
  }
  DISPATCH(p->d, p->r, LAUNCH_SPLIT(p, k_ml_post_ap, p->grid_post(), L0.AP.dev(), Xdev, r, L0.res1, p->ml[1].x, L0.Pb,
                                    L0.agg(), p->dinv, p->ml_omega, z, pout, gate, p->n));
  HIPC(hipGetLastError());
  return DPGO_OK;
}

// The launches of one cycle after the pre-smoothing step of level 0 (x1 = w Dinv r is in ml[0].x1):
// z = proj_X(M^-1 r); partial sums <r,r>, <z,r> into `pout` (may be NULL).  `gate`: state record for early exit.
int launch_ml_tail(dpgo_problem_s* p, const double* Xdev, const double* r, double* z, double* pout,
                   const DevState* gate, bool stop_check) {
  const int nl = (int)p->ml.size();
  // the dense level reads its right-hand side in the precision its inverse is stored in (same buffer)
  auto rc32_of = [&](const dpgo_problem_s::MlLevel& C) {
    return (C.k == 0 && p->coarse32_active()) ? reinterpret_cast<float*>(C.r) : (float*)nullptr;
  };
  auto A_of = [&](int l) { return l == 0 ? p->Q.dev() : p->ml[l].A.dev(); };
  auto r_of = [&](int l) { return l == 0 ? r : (const double*)p->ml[l].r; };
  int g0 = p->grid_restrict();  // grid of the level-0 launches: restriction first, post-smoothing later
  auto grid_of = [&](const dpgo_problem_s::MlLevel& L) {
    if (&L == &p->ml[0]) return g0;
    const int P = ml_tile(p->b, L.split);
    return std::max(1, std::min(kMaxGrid, (L.n + P - 1) / P));
  };
#define ML_SPLIT_LAUNCH(L, KERNEL, ...)                                                                  \
  do {                                                                                                   \
    const int g_ = grid_of(L);                                                                           \
    if ((L).split == 4)                                                                                  \
      hipLaunchKernelGGL((KERNEL<D, R, 4>), dim3(g_), dim3(kBlock), 0, p->stream, __VA_ARGS__);          \
    else if ((L).split == 2)                                                                             \
      hipLaunchKernelGGL((KERNEL<D, R, 2>), dim3(g_), dim3(kBlock), 0, p->stream, __VA_ARGS__);          \
    else                                                                                                 \
      hipLaunchKernelGGL((KERNEL<D, R, 1>), dim3(g_), dim3(kBlock), 0, p->stream, __VA_ARGS__);          \
  } while (0)
  const bool ap = p->ml_use_ap();  // two levels: the residual after pre-smoothing is kept, the dense level hands over xc
  CHK(launch_ml_restrict0(p, r, gate, g0, stop_check));
  for (int l = 1; l + 1 < nl; ++l) {  // down
    auto& L = p->ml[l];
    auto& C = p->ml[l + 1];
    DISPATCH(p->d, p->r, ML_SPLIT_LAUNCH(L, k_ml_restrict, A_of(l), L.x1, r_of(l), L.Pb, 0.0, L.k, C.r, rc32_of(C),
                                         C.k ? C.dinv : (const double*)nullptr, p->ml_omega, C.x1, gate, L.n,
                                         (double*)nullptr, (double*)nullptr, (const int32_t*)nullptr));
  }
  {  // dense level (+ prolongation unless the level above does it itself)
    auto& L = p->ml[nl - 2];
    auto& C = p->ml[nl - 1];
    if (p->ml_use_dense_sym())
      CHK(launch_dense_sym(p, C, gate));
    else
      CHK(launch_coarse_prolong(p, L, C, gate, ap ? C.x : nullptr));
  }
  g0 = p->grid_post();
  if (ap) return launch_ml_post_ap(p, Xdev, r, z, pout, gate);
  for (int l = nl - 2; l >= 1; --l) {  // up
    auto& L = p->ml[l];
    auto& F = p->ml[l - 1];
    DISPATCH(p->d, p->r, ML_SPLIT_LAUNCH(L, k_ml_post_mid, L.A.dev(), L.x, L.r, L.dinv, p->ml_omega, F.x1, F.Pb, F.k, F.x,
                                         F.n, gate, L.n));
  }
  if (p->tcg_sym) {
    DISPATCH(p->d, p->r, hipLaunchKernelGGL((k_ml_post<D, R, 1, BsrSymDev>), dim3(g0), dim3(kBlock), 0, p->stream,
                                            p->sym.dev(), Xdev, p->ml[0].x, r, p->dinv, p->ml_omega, p->ml_shift, z, pout,
                                            gate, p->n));
  } else {
    DISPATCH(p->d, p->r, ML_SPLIT_LAUNCH(p->ml[0], k_ml_post, p->Q.dev(), Xdev, p->ml[0].x, r, p->dinv, p->ml_omega,
                                         p->ml_shift, z, pout, gate, p->n));
  }
#undef ML_SPLIT_LAUNCH
  HIPC(hipGetLastError());
  return DPGO_OK;
}

// Stand-alone application z = proj_X(M^-1 v) (QuadraticProblem::PreConditioner outside the tCG loop).
int launch_ml_apply(dpgo_problem_s* p, const double* Xdev, const double* v, double* z) {
  DISPATCH(p->d, p->r, hipLaunchKernelGGL((k_ml_presmooth<D, R>), dim3(p->grid()), dim3(kBlock), 0, p->stream, v, p->dinv,
                                          p->ml_omega, p->ml[0].x1, (const DevState*)nullptr, p->n));
  HIPC(hipGetLastError());
  // (outside the tCG loop the cycle reads the fp64 originals: its pre-smoothed iterate was written in fp64 just above)
  p->ml_ops32_suspend = true;
  const int rc = launch_ml_tail(p, Xdev, v, z, nullptr, nullptr);
  p->ml_ops32_suspend = false;
  return rc;
}

}  // namespace dpgo_host

extern "C" {


int dpgo_multilevel_default_ks(int n, int d, int* ks, int* nks) {
  if (n <= 0 || d < 2 || d > 3 || !nks) return fail(DPGO_ERR_INVALID, "bad arguments");
  int split = (n < 40000) ? 4 : 1;
  if (const int v = options().split)
    if (v == 1 || v == 2 || v == 4) split = v;
  const std::vector<int> v = ml_default_ks(n, d + 1, split);
  if (ks)
    for (size_t l = 0; l < v.size() && (int)l < *nks; ++l) ks[l] = v[l];
  *nks = (int)v.size();
  return DPGO_OK;
}


int dpgo_multilevel_graph_aggregates(int n, const int32_t* rowptr, const int32_t* colidx, int max_size, int32_t* label,
                                     int32_t* parent, int* n_aggregates) {
  if (n <= 0 || !rowptr || !colidx || max_size < 2 || !label) return fail(DPGO_ERR_INVALID, "bad arguments");
  const std::vector<int32_t> rp(rowptr, rowptr + n + 1), ci(colidx, colidx + rowptr[n]);
  for (int32_t c : ci)
    if (c < 0 || c >= n) return fail(DPGO_ERR_INVALID, "block column out of range");
  std::vector<int32_t> lab, ptr, mem, par, pslot;
  const int na = ml_graph_aggregates(rp, ci, n, max_size, lab, ptr, mem, par, pslot);
  std::copy(lab.begin(), lab.end(), label);
  if (parent) std::copy(par.begin(), par.end(), parent);
  if (n_aggregates) *n_aggregates = na;
  return DPGO_OK;
}


int dpgo_multilevel_merged_aggregates(int n, const int32_t* rowptr, const int32_t* colidx, int max_size, int merge_cap,
                                      int32_t* label, int32_t* parent, int* n_aggregates) {
  if (n <= 0 || !rowptr || !colidx || max_size < 2 || merge_cap < max_size || !label) return fail(DPGO_ERR_INVALID, "bad arguments");
  const std::vector<int32_t> rp(rowptr, rowptr + n + 1), ci(colidx, colidx + rowptr[n]);
  for (int32_t c : ci)
    if (c < 0 || c >= n) return fail(DPGO_ERR_INVALID, "block column out of range");
  std::vector<int32_t> lab, ptr, mem, par, pslot;
  ml_graph_aggregates(rp, ci, n, max_size, lab, ptr, mem, par, pslot);
  const int na = ml_merge_small_aggregates(rp, ci, n, max_size, merge_cap, lab, ptr, mem, par, pslot);
  std::copy(lab.begin(), lab.end(), label);
  if (parent) std::copy(par.begin(), par.end(), parent);
  if (n_aggregates) *n_aggregates = na;
  return DPGO_OK;
}


int dpgo_problem_additive_plan(dpgo_problem_t p, int* lane_groups, int* tile, int* growth, int* merge_cap, int* aggregates,
                               int* graph) {
  if (!p) return fail(DPGO_ERR_INVALID, "null handle");
  if ((int)p->h_rowptr.size() != p->n + 1) return fail(DPGO_ERR_STATE, "Q's block pattern is not set");
  const auto& plan = additive_plan(p);
  if (lane_groups) *lane_groups = plan.split;
  if (tile) *tile = plan.tile;
  if (growth) *growth = plan.S;
  if (merge_cap) *merge_cap = plan.cap;
  if (aggregates) *aggregates = plan.na;
  if (graph) *graph = plan.graph ? 1 : 0;
  return DPGO_OK;
}


int dpgo_problem_setup_multilevel(dpgo_problem_t p, int nks, const int* ks, double omega, double shift) {
  CHK(check_ready(p));
  if (nks < 0 || nks > 8 || (nks > 0 && !ks) || !(omega > 0.0) || !(shift >= 0.0))
    return fail(DPGO_ERR_INVALID, "bad multilevel arguments");
  std::vector<int> v = nks > 0 ? std::vector<int>(ks, ks + nks) : ml_default_ks(p->n, p->b, p->split);
  const bool same = p->ml_symbolic && ml_current_ks(p) == v;
  if (!same) {
    // graph aggregates with merged fragments that fit a workgroup tile of the one-launch solve also get that layout's
    // (aggregate, slot) table, so that an explicit hierarchy of this shape serves precond = additive as well
    int perm_tile = 0;
    if (v.size() == 2 && v[0] < 0 && v[1] < 0 && p->split == 4)
      perm_tile = -v[1] <= ml_tile(p->b, 4) ? ml_tile(p->b, 4) : (-v[1] <= ml_tile(p->b, 1) ? ml_tile(p->b, 1) : 0);
    CHK(ml_symbolic_setup(p, v, perm_tile));
  }
  p->ml_user_ks = nks > 0;
  p->ml_additive_layout = false;
  p->ml_omega = omega;
  p->ml_shift = shift;
  CHK(ml_numeric_setup(p));
  HIPC(hipStreamSynchronize(p->stream));
  return DPGO_OK;
}


int dpgo_problem_multilevel_path(dpgo_problem_t p, int* flags) {
  if (!p || !flags) return fail(DPGO_ERR_INVALID, "null handle / pointer");
  if (!p->ml_symbolic) return fail(DPGO_ERR_STATE, "multilevel hierarchy not set up");
  *flags = (p->ml_use_ap() ? DPGO_ML_PATH_AP : 0) | (p->ml_use_dense_sym() ? DPGO_ML_PATH_PACKED_DENSE : 0);
  return DPGO_OK;
}


int dpgo_problem_multilevel_coarse_bits(dpgo_problem_t p, int* bits) {
  if (!p || !bits) return fail(DPGO_ERR_INVALID, "null handle / pointer");
  if (*bits < 0) {
    *bits = p->ml_coarse_bits;
    return DPGO_OK;
  }
  if (*bits != 32 && *bits != 64) return fail(DPGO_ERR_INVALID, "the coarsest inverse is stored in 32 or 64 bits");
  if (*bits != p->ml_coarse_bits) {
    p->ml_coarse_bits = *bits;
    p->ml_ready = false;  // the stored inverse is rebuilt at the next use
  }
  return DPGO_OK;
}


int dpgo_problem_multilevel_operator_bits(dpgo_problem_t p, int* bits, int* active) {
  if (!p || !bits) return fail(DPGO_ERR_INVALID, "null handle / pointer");
  if (*bits >= 0) {
    if (*bits != 32 && *bits != 64) return fail(DPGO_ERR_INVALID, "the cycle's operator copies are stored in 32 or 64 bits");
    p->ml_operator_bits = *bits;
  }
  *bits = p->ml_operator_bits;
  // (what the last solve's cycle streamed: 1 operator copies, 2 the cycle's internal vectors, 4 the dense level -- in fp32)
  if (active) *active = (p->ml_ops32_active() ? 1 : 0) | (p->ml_vec32_active() ? 2 : 0) | (p->coarse32_active() ? 4 : 0);
  return DPGO_OK;
}

int dpgo_problem_multilevel_info(dpgo_problem_t p, int* nlevels, int* sizes, int* ks, int* nnzb) {
  if (!p) return fail(DPGO_ERR_INVALID, "null handle");
  if (!p->ml_symbolic) return fail(DPGO_ERR_STATE, "multilevel hierarchy not set up");
  const int cap = nlevels ? *nlevels : 0;
  for (int l = 0; l < (int)p->ml.size() && l < cap; ++l) {
    if (sizes) sizes[l] = p->ml[l].n;
    if (ks) ks[l] = p->ml[l].graph ? -p->ml[l].k : p->ml[l].k;  // negative: graph aggregates of at most that many poses
    // (graph aggregates whose fragments were merged: the LAST level's entry, otherwise 0, carries -merge bound)
    if (ks && l > 0 && l + 1 == (int)p->ml.size() && p->ml[0].graph && p->ml[0].merge_cap) ks[l] = -p->ml[0].merge_cap;
    if (nnzb) nnzb[l] = (l == 0) ? p->Q.nnzb : p->ml[l].A.nnzb;
  }
  if (nlevels) *nlevels = (int)p->ml.size();
  return DPGO_OK;
}


int dpgo_problem_multilevel_get(dpgo_problem_t p, int level, int what, void* out_host) {
  CHK(check_ready(p));
  if (!p->ml_ready) return fail(DPGO_ERR_STATE, "multilevel hierarchy not built");
  if (!out_host || level < 0 || level >= (int)p->ml.size()) return fail(DPGO_ERR_INVALID, "bad level / null pointer");
  auto& L = p->ml[level];
  const int bb = p->b * p->b;
  const void* src = nullptr;
  size_t bytes = 0;
  switch (what) {
    case DPGO_ML_P_BLOCKS:
      src = L.Pb, bytes = sizeof(double) * (size_t)L.n * bb;
      break;
    case DPGO_ML_A_ROWPTR:
      src = L.A.rowptr, bytes = sizeof(int32_t) * ((size_t)L.n + 1);
      break;
    case DPGO_ML_A_COLIDX:
      src = L.A.colidx, bytes = sizeof(int32_t) * (size_t)L.A.nnzb;
      break;
    case DPGO_ML_A_VALUES:
      src = L.A.vals, bytes = sizeof(double) * (size_t)L.A.nnzb * bb;
      break;
    case DPGO_ML_AGG_LABELS:
      src = L.graph ? L.lab : nullptr, bytes = sizeof(int32_t) * (size_t)L.n;
      break;
    case DPGO_ML_AP_NNZB: {
      if (!L.AP.vals) return fail(DPGO_ERR_INVALID, "this level does not hold that item");
      *static_cast<int32_t*>(out_host) = L.AP.nnzb;
      return DPGO_OK;
    }
    case DPGO_ML_RESTRICT_PARTIALS: {
      if (!L.graph) return fail(DPGO_ERR_INVALID, "this level does not hold that item");
      *static_cast<int32_t*>(out_host) = L.nseg;
      return DPGO_OK;
    }
    case DPGO_ML_DENSE_INVERSE: {
      if (level + 1 != (int)p->ml.size()) return fail(DPGO_ERR_INVALID, "the dense inverse belongs to the last level");
      const int N = L.n * p->b;  // the N x N corner of the padded lda x lda array
      HIPC(hipMemcpy2DAsync(out_host, sizeof(double) * N, p->ml_dense, sizeof(double) * p->ml_lda, sizeof(double) * N, N,
                            hipMemcpyDeviceToHost, p->stream));
      HIPC(hipStreamSynchronize(p->stream));
      return DPGO_OK;
    }
    default:
      return fail(DPGO_ERR_INVALID, "unknown item");
  }
  if (!src) return fail(DPGO_ERR_INVALID, "this level does not hold that item");
  HIPC(hipMemcpyAsync(out_host, src, bytes, hipMemcpyDeviceToHost, p->stream));
  HIPC(hipStreamSynchronize(p->stream));
  return DPGO_OK;
}


int dpgo_dense_spd_inverse(int N, const double* A_host, double* Ainv_host, int device, int use_mfma) {
  if (N <= 0 || N > 16384 || !A_host || !Ainv_host) return fail(DPGO_ERR_INVALID, "bad arguments");
  int cnt = 0;
  CHK(dpgo_device_count(&cnt));
  if (cnt <= 0) return fail(DPGO_ERR_HIP, "no HIP device (this library has no CPU fallback)");
  if (device < 0 || device >= cnt) return fail(DPGO_ERR_INVALID, "device index out of range");
  HIPC(hipSetDevice(device));
  const int lda = ((N + kNB - 1) / kNB) * kNB;
  TmpDev tmp;
  double *M = nullptr, *W = nullptr, *Rx = nullptr;
  CHK(tmp.alloc(&M, sizeof(double) * (size_t)lda * lda));
  CHK(tmp.alloc(&W, sizeof(double) * (size_t)lda * kNB));
  CHK(tmp.alloc(&Rx, sizeof(double) * (size_t)lda * kNB));
  HIPC(hipMemset(M, 0, sizeof(double) * (size_t)lda * lda));
  HIPC(hipMemcpy2D(M, sizeof(double) * lda, A_host, sizeof(double) * N, sizeof(double) * N, N, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_dense_pad_identity, dim3(1), dim3(kBlock), 0, (hipStream_t) nullptr, M, lda, N);
  CHK(dense_spd_inverse(nullptr, M, lda, W, Rx, use_mfma != 0));
  HIPC(hipMemcpy2D(Ainv_host, sizeof(double) * N, M, sizeof(double) * lda, sizeof(double) * N, N, hipMemcpyDeviceToHost));
  return DPGO_OK;
}
#ifdef DPGO_TIMELINE
int dpgo_debug_timeline_cycle(long long* out /* [2][3][64]: restriction, post-smoothing */) {
  HIPC(hipDeviceSynchronize());
  HIPC(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tl_restrict), sizeof(long long) * 3 * 64));
  HIPC(hipMemcpyFromSymbol(out + 3 * 64, HIP_SYMBOL(g_tl_post), sizeof(long long) * 3 * 64));
  return DPGO_OK;
}
#endif

}  // extern "C"
