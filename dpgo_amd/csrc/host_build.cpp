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
// dpgo_build_multilevel: host setup of the optional two-level preconditioner (the analogue of
// PoseGraph::constructPreconditioner, reference src/PoseGraph.cpp:598-613, which factors Q + 0.1 I on the host).
// Same construction as dpgo_amd/multilevel.py / the oracle's amg_prolongation_blocks:
//   aggregates of k consecutive poses; Pb_i = G(root -> i)^T with G composed along the odometry chain, the relative
//   pose T = [R t; 0 1] of edge i -> i+1 read off Q_{i,i+1} = -T Om = -[w kappa R, w tau t; 0, w tau]
//   (src/DPGO_utils.cpp:307-329); Ac = P^T (Q + shift I) P accumulated densely; AcInv by Cholesky.
namespace {

// in-place inverse of a dense SPD matrix (row-major N x N): A = L L^T, Linv, A^-1 = Linv^T Linv
int spd_inverse(std::vector<double>& A, int N) {
  // Cholesky, lower triangle in place (row-oriented inner products: contiguous rows)
  for (int i = 0; i < N; ++i) {
    double* ai = &A[(size_t)i * N];
    for (int j = 0; j <= i; ++j) {
      const double* aj = &A[(size_t)j * N];
      double s = ai[j];
      for (int k = 0; k < j; ++k) s -= ai[k] * aj[k];
      if (j < i) {
        ai[j] = s / aj[j];
      } else {
        if (!(s > 0.0)) return 1;
        ai[i] = std::sqrt(s);
      }
    }
  }
  // Linv (lower) into W, column by column through forward substitution on unit vectors, row-major rows of W
  std::vector<double> W((size_t)N * N, 0.0);
  for (int i = 0; i < N; ++i) {
    const double* li = &A[(size_t)i * N];
    double* wi = &W[(size_t)i * N];
    // row i of Linv: w_i = (e_i - sum_{k<i} L_ik w_k) / L_ii
    for (int k = 0; k < i; ++k) {
      const double lik = li[k];
      if (lik != 0.0) {
        const double* wk = &W[(size_t)k * N];
        for (int c = 0; c <= k; ++c) wi[c] -= lik * wk[c];
      }
    }
    wi[i] += 1.0;
    const double inv = 1.0 / li[i];
    for (int c = 0; c <= i; ++c) wi[c] *= inv;
  }
  // A^-1 = Linv^T Linv: (A^-1)_{pq} = sum_{i >= max(p,q)} W_ip W_iq; accumulate row-wise rank-1 updates
  std::fill(A.begin(), A.end(), 0.0);
  for (int i = 0; i < N; ++i) {
    const double* wi = &W[(size_t)i * N];
    for (int p = 0; p <= i; ++p) {
      const double wp = wi[p];
      if (wp == 0.0) continue;
      double* ap = &A[(size_t)p * N];
      for (int q = 0; q <= i; ++q) ap[q] += wp * wi[q];
    }
  }
  return 0;
}

}  // namespace

extern "C" int dpgo_multilevel_default_k(int n, int d) {
  const int b = d + 1;
  int k = 4;
  while (((n + k - 1) / k) * b > 3200) k *= 2;
  if (k > 16 && ((n + 15) / 16) * b <= 8192) k = 16;
  return k;
}

extern "C" int dpgo_build_multilevel(int d, int n, const int32_t* rowptr, const int32_t* colidx, const double* vals,
                                     double shift, int k, double* P_blocks, double* AcInv) {
  if (d < 2 || d > 3 || n <= 0 || !rowptr || !colidx || !vals || !P_blocks || !AcInv || k < 2) return DPGO_ERR_INVALID;
  const int b = d + 1, bb = b * b, nc = (n + k - 1) / k, N = nc * b;
  // ---- prolongation blocks
  double G[16];
  auto set_identity = [&](double* M) {
    std::memset(M, 0, sizeof(double) * bb);
    for (int q = 0; q < b; ++q) M[q * b + q] = 1.0;
  };
  set_identity(G);
  for (int i = 0; i < n; ++i) {
    if (i % k == 0) {
      set_identity(G);
    } else {
      const double* blk = nullptr;
      for (int t = rowptr[i - 1]; t < rowptr[i]; ++t)
        if (colidx[t] == i) blk = vals + (size_t)t * bb;
      bool ok = blk && -blk[d * b + d] > 0.0;
      double wk = 0.0;
      if (ok) {
        for (int p = 0; p < d; ++p) wk += blk[p * b + 0] * blk[p * b + 0];
        wk = std::sqrt(wk);
        ok = wk > 0.0;
      }
      if (ok) {
        double T[16], Gn[16];
        set_identity(T);
        const double wt = -blk[d * b + d];
        for (int p = 0; p < d; ++p) {
          for (int q = 0; q < d; ++q) T[p * b + q] = -blk[p * b + q] / wk;
          T[p * b + d] = -blk[p * b + d] / wt;
        }
        for (int p = 0; p < b; ++p)
          for (int q = 0; q < b; ++q) {
            double s = 0.0;
            for (int m = 0; m < b; ++m) s += G[p * b + m] * T[m * b + q];
            Gn[p * b + q] = s;
          }
        std::memcpy(G, Gn, sizeof(double) * bb);
      } else {
        set_identity(G);
      }
    }
    for (int p = 0; p < b; ++p)
      for (int q = 0; q < b; ++q) P_blocks[(size_t)i * bb + p * b + q] = G[q * b + p];  // G^T
  }
  // ---- Ac = P^T (Q + shift I) P, dense
  std::vector<double> Ac((size_t)N * N, 0.0);
  for (int i = 0; i < n; ++i) {
    const double* Pi = P_blocks + (size_t)i * bb;
    const int ai = i / k;
    for (int t = rowptr[i]; t < rowptr[i + 1]; ++t) {
      const int j = colidx[t], aj = j / k;
      const double* Pj = P_blocks + (size_t)j * bb;
      double A[16], AP[16];
      std::memcpy(A, vals + (size_t)t * bb, sizeof(double) * bb);
      if (j == i)
        for (int q = 0; q < b; ++q) A[q * b + q] += shift;
      for (int p = 0; p < b; ++p)
        for (int q = 0; q < b; ++q) {
          double s = 0.0;
          for (int m = 0; m < b; ++m) s += A[p * b + m] * Pj[m * b + q];
          AP[p * b + q] = s;
        }
      for (int p = 0; p < b; ++p)
        for (int q = 0; q < b; ++q) {
          double s = 0.0;
          for (int m = 0; m < b; ++m) s += Pi[m * b + p] * AP[m * b + q];
          Ac[(size_t)(ai * b + p) * N + aj * b + q] += s;
        }
    }
  }
  for (int p = 0; p < N; ++p)  // symmetrise (round-off)
    for (int q = 0; q < p; ++q) {
      const double s = 0.5 * (Ac[(size_t)p * N + q] + Ac[(size_t)q * N + p]);
      Ac[(size_t)p * N + q] = Ac[(size_t)q * N + p] = s;
    }
  if (spd_inverse(Ac, N)) return DPGO_ERR_STATE;
  std::memcpy(AcInv, Ac.data(), sizeof(double) * (size_t)N * N);
  return DPGO_OK;
}
