// host_build.cpp -- host-side construction of the agent-local data matrices (no GPU code).
//
// dpgo_build_Q_bsr   : PoseGraph::constructQ (reference src/PoseGraph.cpp:381-491) with
//                      constructConnectionLaplacianSE (src/DPGO_utils.cpp:272-344), emitted directly
//                      as block-CSR (the reference goes through a scalar A*Omega*A^T sparse product).
// dpgo_build_G_coupling: the linear-term operator of PoseGraph::constructG (src/PoseGraph.cpp:493-580):
//                      G = G0 + Xnbr * C with C the inter-agent off-diagonal Laplacian blocks.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "../../include/dpgo_hip.h"

namespace {

struct Blk {
  double v[16];
  Blk() { std::memset(v, 0, sizeof(v)); }
};

// T = [R t; 0 1] (b x b), om = diag(w*kappa (x d), w*tau)   (src/DPGO_utils.cpp:307-329)
struct EdgeMats {
  double T[4][4];
  double om[4];
};

EdgeMats edge_mats(int d, const double* R, const double* t, double kappa, double tau, double w) {
  EdgeMats E;
  std::memset(&E, 0, sizeof(E));
  for (int p = 0; p < d; ++p) {
    for (int q = 0; q < d; ++q) E.T[p][q] = R[p * d + q];
    E.T[p][d] = t[p];
    E.om[p] = w * kappa;
  }
  E.T[d][d] = 1.0;
  E.om[d] = w * tau;
  return E;
}

using Key = std::pair<int32_t, int32_t>;

void add_block(std::map<Key, Blk>& M, int b, int i, int j, const double (*A)[4], double sign) {
  Blk& blk = M[Key(i, j)];
  for (int p = 0; p < b; ++p)
    for (int q = 0; q < b; ++q) blk.v[p * b + q] += sign * A[p][q];
}

int emit(const std::map<Key, Blk>& M, int n, int b, int* nnzb_out, int32_t* rowptr, int32_t* colidx, double* vals) {
  *nnzb_out = (int)M.size();
  if (!rowptr || !colidx || !vals) return DPGO_OK;
  std::fill(rowptr, rowptr + n + 1, 0);
  int t = 0;
  for (const auto& kv : M) {
    rowptr[kv.first.first + 1] += 1;
    colidx[t] = kv.first.second;
    std::memcpy(vals + (size_t)t * b * b, kv.second.v, sizeof(double) * b * b);
    ++t;
  }
  for (int i = 0; i < n; ++i) rowptr[i + 1] += rowptr[i];
  return DPGO_OK;
}

}  // namespace

extern "C" {

int dpgo_build_Q_bsr(int my_id, int d, int n, int m, const int32_t* r1, const int32_t* p1, const int32_t* r2,
                     const int32_t* p2, const double* R, const double* t, const double* kappa, const double* tau,
                     const double* weight, int n_priors, const int32_t* prior_idx, double prior_kappa,
                     double prior_tau, int* nnzb_out, int32_t* rowptr, int32_t* colidx, double* vals) {
  if (!nnzb_out || n <= 0 || m < 0 || (d != 2 && d != 3)) return DPGO_ERR_INVALID;
  if (m > 0 && (!r1 || !p1 || !r2 || !p2 || !R || !t || !kappa || !tau || !weight)) return DPGO_ERR_INVALID;
  const int b = d + 1;
  std::map<Key, Blk> M;
  for (int i = 0; i < n; ++i) M[Key(i, i)];  // explicit diagonal block for every pose (PoseGraph.cpp:470-485)
  for (int e = 0; e < m; ++e) {
    const EdgeMats E = edge_mats(d, R + (size_t)e * d * d, t + (size_t)e * d, kappa[e], tau[e], weight[e]);
    double TO[4][4] = {}, TOT[4][4] = {}, OM[4][4] = {}, TOt[4][4] = {};
    for (int p = 0; p < b; ++p)
      for (int q = 0; q < b; ++q) TO[p][q] = E.T[p][q] * E.om[q];
    for (int p = 0; p < b; ++p)
      for (int q = 0; q < b; ++q) {
        double s = 0.0;
        for (int k = 0; k < b; ++k) s += TO[p][k] * E.T[q][k];
        TOT[p][q] = s;
        TOt[p][q] = TO[q][p];
      }
    for (int p = 0; p < b; ++p) OM[p][p] = E.om[p];
    const bool mine1 = r1[e] == my_id, mine2 = r2[e] == my_id;
    if (mine1 && mine2) {  // private edge (odometry or private loop closure)
      const int i = p1[e], j = p2[e];
      if (i < 0 || i >= n || j < 0 || j >= n || i == j) return DPGO_ERR_INVALID;
      add_block(M, b, i, i, TOT, 1.0);
      add_block(M, b, j, j, OM, 1.0);
      add_block(M, b, i, j, TO, -1.0);
      add_block(M, b, j, i, TOt, -1.0);
    } else if (mine1) {  // outgoing shared edge: Q[p1,p1] += T Om T^T  (PoseGraph.cpp:431-434)
      if (p1[e] < 0 || p1[e] >= n) return DPGO_ERR_INVALID;
      add_block(M, b, p1[e], p1[e], TOT, 1.0);
    } else if (mine2) {  // incoming shared edge: Q[p2,p2] += Om     (PoseGraph.cpp:455-457)
      if (p2[e] < 0 || p2[e] >= n) return DPGO_ERR_INVALID;
      add_block(M, b, p2[e], p2[e], OM, 1.0);
    }  // irrelevant edges are ignored (PoseGraph.cpp:66-69)
  }
  for (int k = 0; k < n_priors; ++k) {  // PoseGraph.cpp:461-468
    const int idx = prior_idx[k];
    if (idx < 0 || idx >= n) return DPGO_ERR_INVALID;
    double P[4][4] = {};
    for (int p = 0; p < d; ++p) P[p][p] = prior_kappa;
    P[d][d] = prior_tau;
    add_block(M, b, idx, idx, P, 1.0);
  }
  return emit(M, n, b, nnzb_out, rowptr, colidx, vals);
}

int dpgo_build_G_coupling(int my_id, int d, int n, int m, const int32_t* r1, const int32_t* p1, const int32_t* r2,
                          const int32_t* p2, const double* R, const double* t, const double* kappa, const double* tau,
                          const double* weight, const int32_t* slot_of_edge, int* nnzb_out, int32_t* rowptr,
                          int32_t* colidx, double* vals) {
  if (!nnzb_out || n <= 0 || m < 0 || (d != 2 && d != 3)) return DPGO_ERR_INVALID;
  if (m > 0 && (!r1 || !p1 || !r2 || !p2 || !R || !t || !kappa || !tau || !weight || !slot_of_edge))
    return DPGO_ERR_INVALID;
  const int b = d + 1;
  std::map<Key, Blk> M;
  for (int e = 0; e < m; ++e) {
    const bool mine1 = r1[e] == my_id, mine2 = r2[e] == my_id;
    if (mine1 == mine2) continue;  // private or irrelevant
    const int slot = slot_of_edge[e];
    if (slot < 0) continue;  // inactive neighbour (PoseGraph.cpp:521-526)
    const EdgeMats E = edge_mats(d, R + (size_t)e * d * d, t + (size_t)e * d, kappa[e], tau[e], weight[e]);
    double TO[4][4] = {}, TOt[4][4] = {};
    for (int p = 0; p < b; ++p)
      for (int q = 0; q < b; ++q) TO[p][q] = E.T[p][q] * E.om[q];
    for (int p = 0; p < b; ++p)
      for (int q = 0; q < b; ++q) TOt[p][q] = TO[q][p];
    if (mine1) {
      // outgoing: G[:,p1] += -X_j Om T^T (PoseGraph.cpp:533-537) == X_j * Q[j,i], row-block i holds
      // Q[i,j] = -T Om
      if (p1[e] < 0 || p1[e] >= n) return DPGO_ERR_INVALID;
      add_block(M, b, p1[e], slot, TO, -1.0);
    } else {
      // incoming: G[:,p2] += -X_i T Om (PoseGraph.cpp:558-562); row-block p2 holds Q[j,i] = -Om T^T
      if (p2[e] < 0 || p2[e] >= n) return DPGO_ERR_INVALID;
      add_block(M, b, p2[e], slot, TOt, -1.0);
    }
  }
  return emit(M, n, b, nnzb_out, rowptr, colidx, vals);
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------
// Pose order for HBM-bound blocks.  The block-SpMM family gathers the tiles of a pose's graph neighbours and, on the
// symmetric storage, the blocks stored with them; a workgroup tile walk gives each of the 8 XCDs one contiguous eighth of
// the poses, whose ~96 resident workgroups sweep that range together.  What is re-used (a pose's tile by its 6 lattice
// neighbours, an upper block by its lower reference) is served by the XCD's 4 MiB L2 only if the two uses are close in
// the sweep: in the odometry ("snake") order of a 50 x 50 x 40 lattice the neighbours of a pose are up to 2 500 rows away
// and the kernels fetch 1.18-1.38 x the bytes they store (PMC).  dpgo_locality_order keeps the `nparts` contiguous chunks
// of the index range (chunk boundaries at multiples of `align` poses: the XCDs' shares) and, INSIDE every chunk, keeps
// RUNS of `run` consecutive poses together (consecutive poses are odometry neighbours: a wave's 16 poses gather
// contiguous memory, which a pose-by-pose renumbering destroys -- measured: plain reverse Cuthill-McKee made the kernels
// 6 % SLOWER) and orders the runs by reverse Cuthill-McKee over the graph of runs -- breadth-first levels from a
// pseudo-peripheral run, neighbours by (degree, index).  run = 1: plain reverse Cuthill-McKee of the poses.
// Pure host code, deterministic.  new_index[i] = position of caller pose i.
int dpgo_locality_order_runs(int n, const int32_t* rowptr, const int32_t* colidx, int nparts, int align, int run,
                             int32_t* new_index) {
  if (n <= 0 || !rowptr || !colidx || !new_index || nparts < 1 || align < 1 || run < 1) return DPGO_ERR_INVALID;
  for (int i = 0; i < n; ++i) {
    if (rowptr[i + 1] < rowptr[i]) return DPGO_ERR_INVALID;
    for (int t = rowptr[i]; t < rowptr[i + 1]; ++t)
      if (colidx[t] < 0 || colidx[t] >= n) return DPGO_ERR_INVALID;
  }
  // chunk boundaries: k n / nparts rounded down to a multiple of `align`
  std::vector<int> bound(nparts + 1, 0);
  for (int k = 1; k < nparts; ++k) {
    long long b = ((long long)n * k / nparts) / align * align;
    bound[k] = (int)std::max<long long>(bound[k - 1], std::min<long long>(b, n));
  }
  bound[nparts] = n;
  int pos = 0;
  for (int k = 0; k < nparts; ++k) {
    const int c0 = bound[k], c1 = bound[k + 1];
    if (c1 <= c0) continue;
    // the graph of runs: run s = poses [c0 + s run, c0 + (s + 1) run) of the chunk; adjacency = runs joined by an edge
    const int ns = (c1 - c0 + run - 1) / run;
    std::vector<std::vector<int>> adj(ns);
    for (int i = c0; i < c1; ++i) {
      const int si = (i - c0) / run;
      for (int t = rowptr[i]; t < rowptr[i + 1]; ++t) {
        const int j = colidx[t];
        if (j < c0 || j >= c1) continue;
        const int sj = (j - c0) / run;
        if (sj != si) adj[si].push_back(sj);
      }
    }
    std::vector<int> deg(ns), level(ns, -1), order, queue, nb;
    std::vector<char> seen(ns, 0);
    for (int s = 0; s < ns; ++s) {
      std::sort(adj[s].begin(), adj[s].end());
      adj[s].erase(std::unique(adj[s].begin(), adj[s].end()), adj[s].end());
      deg[s] = (int)adj[s].size();
    }
    // breadth-first search over the unvisited runs from `root`; returns the run of the deepest level with the smallest
    // degree (a pseudo-peripheral candidate) and leaves the visit order in `queue`
    auto bfs = [&](int root, bool commit) {
      queue.clear();
      queue.push_back(root);
      level[root] = 0;
      for (size_t head = 0; head < queue.size(); ++head) {
        const int u = queue[head];
        nb.clear();
        for (int v : adj[u]) {
          if (seen[v] || level[v] >= 0) continue;
          level[v] = level[u] + 1;
          nb.push_back(v);
        }
        std::sort(nb.begin(), nb.end(), [&](int a, int b) { return deg[a] != deg[b] ? deg[a] < deg[b] : a < b; });
        queue.insert(queue.end(), nb.begin(), nb.end());
      }
      const int deepest = level[queue.back()];
      int far = queue.back();
      for (size_t q = queue.size(); q-- > 0 && level[queue[q]] == deepest;)
        if (deg[queue[q]] < deg[far] || (deg[queue[q]] == deg[far] && queue[q] < far)) far = queue[q];
      if (commit)
        for (int u : queue) seen[u] = 1;
      for (int u : queue) level[u] = -1;
      return far;
    };
    for (int s = 0; s < ns; ++s) {
      if (seen[s]) continue;
      int root = bfs(s, false);  // two sweeps towards the periphery of this component
      root = bfs(root, false);
      bfs(root, true);
      order.insert(order.end(), queue.begin(), queue.end());
    }
    if ((int)order.size() != ns) return DPGO_ERR_STATE;
    std::reverse(order.begin(), order.end());  // reverse Cuthill-McKee
    for (int s : order)
      for (int i = c0 + s * run; i < std::min(c1, c0 + (s + 1) * run); ++i) new_index[i] = pos++;
  }
  return pos == n ? DPGO_OK : DPGO_ERR_STATE;
}

int dpgo_locality_order(int n, const int32_t* rowptr, const int32_t* colidx, int nparts, int align, int32_t* new_index) {
  return dpgo_locality_order_runs(n, rowptr, colidx, nparts, align, 16, new_index);
}
