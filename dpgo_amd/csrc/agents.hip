// agents.hip -- callers either side of the local solve: GNC re-weighting (src/DPGO_robust.cpp, src/PGOAgent.cpp:997-1142), initial guesses (src/DPGO_solver.cpp:220-303), manifold operations (src/manifold/*.cpp), public-pose exchange plans.
#include "host.h"

namespace dpgo_host {

int free_edges(dpgo_problem_s* p) {
  void* ptrs[] = {p->e_p1,  p->e_p2,    p->c_ptr,  p->c_edge,   p->e_R,    p->e_t,    p->e_kappa,
                  p->e_tau, p->e_w,     p->e_rsq,  p->q_base,   p->e_fixed, p->c_kind, p->e_counts,
                  p->e_role, p->e_slot, p->g_ptr,  p->g_edge,   p->g_kind, p->c_base};
  for (void* q : ptrs)
    if (q) (void)hipFree(q);
  p->e_role = p->g_kind = nullptr;
  p->e_slot = p->g_ptr = p->g_edge = nullptr;
  p->c_base = nullptr;
  p->n_shared_edges = 0;
  p->e_p1 = p->e_p2 = p->c_ptr = p->c_edge = nullptr;
  p->e_R = p->e_t = p->e_kappa = p->e_tau = p->e_w = p->e_rsq = p->q_base = nullptr;
  p->e_fixed = p->c_kind = nullptr;
  p->e_counts = nullptr;
  p->em = 0;
  return DPGO_OK;
}
int rebuild_vals(dpgo_problem_s* p, int nnzb, const int32_t* cptr, const int32_t* cedge, const uint8_t* ckind,
                 const double* base, double sign, double* out) {
  if (nnzb <= 0) return DPGO_OK;
  const int g = std::max(1, std::min(kMaxGrid, (nnzb + kBlock - 1) / kBlock));
  if (p->d == 2)
    hipLaunchKernelGGL(k_rebuild_Q<2>, dim3(g), dim3(kBlock), 0, p->stream, p->edges(), cptr, cedge, ckind, base,
                       sign, out, nnzb);
  else
    hipLaunchKernelGGL(k_rebuild_Q<3>, dim3(g), dim3(kBlock), 0, p->stream, p->edges(), cptr, cedge, ckind, base,
                       sign, out, nnzb);
  HIPC(hipGetLastError());
  return DPGO_OK;
}
int rebuild_Q_from_weights(dpgo_problem_s* p, const double* base, double sign, double* out) {
  return rebuild_vals(p, p->Q.nnzb, p->c_ptr, p->c_edge, p->c_kind, base, sign, out);
}
int rebuild_C_from_weights(dpgo_problem_s* p, const double* base, double sign, double* out) {
  if (!p->g_ptr) return DPGO_OK;
  return rebuild_vals(p, p->C.nnzb, p->g_ptr, p->g_edge, p->g_kind, base, sign, out);
}
int refresh_after_weights(dpgo_problem_s* p) {
  CHK(rebuild_Q_from_weights(p, p->q_base, 1.0, p->Q.vals));
  CHK(rebuild_C_from_weights(p, p->c_base, 1.0, p->C.vals));  // G itself is refreshed by the next update_G call
  p->ml_ready = false;
  p->auto_decided = false;
  p->sym.ready = p->tcg_sym = false;
  const double s = p->dinv_shift > 0 ? p->dinv_shift : 1e-1;
  p->dinv_shift = -1.0;  // clearQuadraticMatrix also drops the preconditioner (src/PoseGraph.cpp:352-355)
  return build_dinv(p, s);
}
}  // namespace dpgo_host

extern "C" {


int dpgo_problem_set_reweightable_edges_ex(dpgo_problem_t p, int m, const int32_t* p1, const int32_t* p2,
                                           const uint8_t* role, const int32_t* slot_in, const double* R,
                                           const double* t, const double* kappa, const double* tau,
                                           const double* weight, const uint8_t* fixed_weight) {
  CHK(check_ready(p));
  if (m < 0 || (m > 0 && (!p1 || !p2 || !R || !t || !kappa || !tau || !weight || !fixed_weight)))
    return fail(DPGO_ERR_INVALID, "null edge arrays");
  if (role && !slot_in) return fail(DPGO_ERR_INVALID, "roles given without neighbour slots");
  const int n = p->n, d = p->d, nnzb = p->Q.nnzb;
  // host copy of the patterns to locate the blocks each edge contributes to
  std::vector<int32_t> rowptr(n + 1), colidx(nnzb);
  HIPC(hipMemcpy(rowptr.data(), p->Q.rowptr, sizeof(int32_t) * (n + 1), hipMemcpyDeviceToHost));
  HIPC(hipMemcpy(colidx.data(), p->Q.colidx, sizeof(int32_t) * nnzb, hipMemcpyDeviceToHost));
  int n_shared = 0;
  for (int e = 0; e < m; ++e)
    if (role && role[e]) ++n_shared;
  std::vector<int32_t> crow, ccol;
  const int cnnz = (n_shared > 0) ? p->C.nnzb : 0;
  if (n_shared > 0) {
    if (!p->C.rowptr) return fail(DPGO_ERR_STATE, "shared re-weightable edges need the G coupling first");
    crow.resize(n + 1);
    ccol.resize(cnnz > 0 ? cnnz : 1);
    HIPC(hipMemcpy(crow.data(), p->C.rowptr, sizeof(int32_t) * (n + 1), hipMemcpyDeviceToHost));
    if (cnnz > 0) HIPC(hipMemcpy(ccol.data(), p->C.colidx, sizeof(int32_t) * cnnz, hipMemcpyDeviceToHost));
  }
  auto find = [](const std::vector<int32_t>& rp, const std::vector<int32_t>& ci, int i, int j) -> int {
    const int32_t* b = ci.data() + rp[i];
    const int32_t* e = ci.data() + rp[i + 1];
    const int32_t* it = std::lower_bound(b, e, (int32_t)j);
    return (it != e && *it == j) ? (int)(it - ci.data()) : -1;
  };
  std::vector<std::vector<std::pair<int, uint8_t>>> lists(nnzb), glists(cnnz);
  std::vector<uint8_t> role_v(m > 0 ? m : 1, 0);
  std::vector<int32_t> slot_v(m > 0 ? m : 1, 0);
  for (int e = 0; e < m; ++e) {
    const int i = p1[e], j = p2[e];
    const int ro = role ? role[e] : 0;
    role_v[e] = (uint8_t)ro;
    if (ro == 0) {
      if (i < 0 || i >= n || j < 0 || j >= n || i == j) return fail(DPGO_ERR_INVALID, "edge endpoint out of range");
      const int sii = find(rowptr, colidx, i, i), sjj = find(rowptr, colidx, j, j), sij = find(rowptr, colidx, i, j),
                sji = find(rowptr, colidx, j, i);
      if (sii < 0 || sjj < 0 || sij < 0 || sji < 0)
        return fail(DPGO_ERR_STATE, "edge does not fit the block pattern of Q");
      lists[sii].push_back({e, 0});
      lists[sjj].push_back({e, 1});
      lists[sij].push_back({e, 2});
      lists[sji].push_back({e, 3});
    } else if (ro == 1 || ro == 2) {
      // outgoing: Q_ii += T Om T^T, C(i, slot) = -T Om; incoming: Q_jj += Om, C(j, slot) = -Om T^T
      // (PoseGraph::constructQ :462-486, constructG :533-562)
      const int mine = (ro == 1) ? i : j;
      const int sl = slot_in[e];
      if (mine < 0 || mine >= n || sl < 0 || sl >= p->C.ncols)
        return fail(DPGO_ERR_INVALID, "shared edge endpoint / neighbour slot out of range");
      slot_v[e] = sl;
      const int sd = find(rowptr, colidx, mine, mine), sc = find(crow, ccol, mine, sl);
      if (sd < 0 || sc < 0) return fail(DPGO_ERR_STATE, "shared edge does not fit the pattern of Q / the G coupling");
      lists[sd].push_back({e, (uint8_t)(ro == 1 ? 0 : 1)});
      glists[sc].push_back({e, (uint8_t)(ro == 1 ? 2 : 3)});
    } else {
      return fail(DPGO_ERR_INVALID, "edge role must be 0, 1 or 2");
    }
  }
  auto flatten = [](const std::vector<std::vector<std::pair<int, uint8_t>>>& L, std::vector<int32_t>& ptr,
                    std::vector<int32_t>& edge, std::vector<uint8_t>& kind) {
    ptr.assign(L.size() + 1, 0);
    for (size_t s = 0; s < L.size(); ++s) {
      for (auto& pr : L[s]) {
        edge.push_back(pr.first);
        kind.push_back(pr.second);
      }
      ptr[s + 1] = (int32_t)edge.size();
    }
  };
  std::vector<int32_t> cptr, cedge, gptr, gedge;
  std::vector<uint8_t> ckind, gkind;
  flatten(lists, cptr, cedge, ckind);
  CHK(free_edges(p));
  p->em = m;
  p->n_shared_edges = n_shared;
  CHK(upload(&p->e_p1, p1, (size_t)m, p->stream));
  CHK(upload(&p->e_p2, p2, (size_t)m, p->stream));
  CHK(upload(&p->e_role, role_v.data(), (size_t)m, p->stream));
  CHK(upload(&p->e_slot, slot_v.data(), (size_t)m, p->stream));
  CHK(upload(&p->e_R, R, (size_t)m * d * d, p->stream));
  CHK(upload(&p->e_t, t, (size_t)m * d, p->stream));
  CHK(upload(&p->e_kappa, kappa, (size_t)m, p->stream));
  CHK(upload(&p->e_tau, tau, (size_t)m, p->stream));
  CHK(upload(&p->e_w, weight, (size_t)m, p->stream));
  CHK(upload(&p->e_fixed, fixed_weight, (size_t)m, p->stream));
  CHK(upload(&p->c_ptr, cptr.data(), cptr.size(), p->stream));
  CHK(upload(&p->c_edge, cedge.data(), cedge.size(), p->stream));
  CHK(upload(&p->c_kind, ckind.data(), ckind.size(), p->stream));
  HIPC(hipMalloc(&p->e_rsq, sizeof(double) * (m > 0 ? m : 1)));
  HIPC(hipMalloc(&p->e_counts, sizeof(int) * 4));
  HIPC(hipMalloc(&p->q_base, sizeof(double) * (size_t)nnzb * p->b * p->b));
  // base = Q(current weights) - sum of the listed edges' contributions at those weights
  CHK(rebuild_Q_from_weights(p, p->Q.vals, -1.0, p->q_base));
  if (n_shared > 0 && cnnz > 0) {
    flatten(glists, gptr, gedge, gkind);
    CHK(upload(&p->g_ptr, gptr.data(), gptr.size(), p->stream));
    CHK(upload(&p->g_edge, gedge.data(), gedge.size(), p->stream));
    CHK(upload(&p->g_kind, gkind.data(), gkind.size(), p->stream));
    HIPC(hipMalloc(&p->c_base, sizeof(double) * (size_t)cnnz * p->b * p->b));
    CHK(rebuild_C_from_weights(p, p->C.vals, -1.0, p->c_base));
  }
  HIPC(hipStreamSynchronize(p->stream));
  return DPGO_OK;
}


int dpgo_problem_set_reweightable_edges(dpgo_problem_t p, int m, const int32_t* p1, const int32_t* p2, const double* R,
                                        const double* t, const double* kappa, const double* tau, const double* weight,
                                        const uint8_t* fixed_weight) {
  return dpgo_problem_set_reweightable_edges_ex(p, m, p1, p2, nullptr, nullptr, R, t, kappa, tau, weight,
                                                fixed_weight);
}


int dpgo_problem_gnc_reweight_device(dpgo_problem_t p, const double* X_dev, const double* nbr_tiles_dev, double mu,
                                     double barc, double w_tol, int update, int counts[3], double* max_rsq) {
  CHK(check_ready(p));
  if (!p->e_w) return fail(DPGO_ERR_STATE, "re-weightable edges not set");
  if (!X_dev) return fail(DPGO_ERR_INVALID, "null X");
  if (p->n_shared_edges > 0 && !nbr_tiles_dev) return fail(DPGO_ERR_INVALID, "shared edges need the neighbour tiles");
  if (update && !(mu > 0.0)) return fail(DPGO_ERR_INVALID, "GNC mu must be positive");
  HIPC(hipMemsetAsync(p->e_counts, 0, sizeof(int) * 4, p->stream));
  const int g = std::max(1, std::min(kMaxGrid, (p->em + kBlock - 1) / kBlock));
  DISPATCH(p->d, p->r, hipLaunchKernelGGL((k_edge_weights<D, R>), dim3(g), dim3(kBlock), 0, p->stream, p->edges(), X_dev,
                                          nbr_tiles_dev, mu, barc, w_tol, update, p->e_counts));
  HIPC(hipGetLastError());
  if (update) CHK(refresh_after_weights(p));
  int h[4] = {0, 0, 0, 0};
  HIPC(hipMemcpyAsync(h, p->e_counts, sizeof(int) * 4, hipMemcpyDeviceToHost, p->stream));
  std::vector<double> rs;
  if (max_rsq) {
    rs.resize(p->em > 0 ? p->em : 1, 0.0);
    if (p->em > 0)
      HIPC(hipMemcpyAsync(rs.data(), p->e_rsq, sizeof(double) * p->em, hipMemcpyDeviceToHost, p->stream));
  }
  HIPC(hipStreamSynchronize(p->stream));
  if (counts) {
    counts[0] = h[0];
    counts[1] = h[1];
    counts[2] = h[2];
  }
  if (max_rsq) {
    double mx = 0.0;
    for (int e = 0; e < p->em; ++e) mx = std::max(mx, rs[e]);
    *max_rsq = mx;
  }
  return DPGO_OK;
}


int dpgo_problem_gnc_reweight(dpgo_problem_t p, const double* X_host, double mu, double barc, double w_tol, int update,
                              int counts[3], double* max_rsq) {
  CHK(check_ready(p));
  if (!X_host) return fail(DPGO_ERR_INVALID, "null X");
  if (p->n_shared_edges > 0) return fail(DPGO_ERR_STATE, "shared edges need the device flavour (neighbour tiles)");
  CHK(h2d(p, p->x2, X_host));
  return dpgo_problem_gnc_reweight_device(p, p->x2, nullptr, mu, barc, w_tol, update, counts, max_rsq);
}


int dpgo_problem_set_edge_weights(dpgo_problem_t p, const double* weight_host) {
  CHK(check_ready(p));
  if (!p->e_w) return fail(DPGO_ERR_STATE, "re-weightable edges not set");
  if (!weight_host) return fail(DPGO_ERR_INVALID, "null weights");
  if (p->em > 0) HIPC(hipMemcpyAsync(p->e_w, weight_host, sizeof(double) * p->em, hipMemcpyHostToDevice, p->stream));
  CHK(refresh_after_weights(p));
  HIPC(hipStreamSynchronize(p->stream));
  return DPGO_OK;
}


int dpgo_problem_get_edge_weights(dpgo_problem_t p, double* weight_host, double* rsq_host) {
  CHK(check_ready(p));
  if (!p->e_w) return fail(DPGO_ERR_STATE, "re-weightable edges not set");
  if (weight_host && p->em > 0)
    HIPC(hipMemcpyAsync(weight_host, p->e_w, sizeof(double) * p->em, hipMemcpyDeviceToHost, p->stream));
  if (rsq_host && p->em > 0)
    HIPC(hipMemcpyAsync(rsq_host, p->e_rsq, sizeof(double) * p->em, hipMemcpyDeviceToHost, p->stream));
  HIPC(hipStreamSynchronize(p->stream));
  return DPGO_OK;
}


// ---- initial guesses ----
namespace {
// Masked PCG: solve  mask A mask x = rhs  (rhs already masked) for the tiles x; A = the handle's Q.  Host-driven
// (two tiny read-backs per iteration): initialisation runs once per problem, outside the hot path.
struct InitBufs {
  double *x, *r, *z, *p, *Ap, *diag, *partial;
};
int init_dot(dpgo_problem_s* h, const double* a, const double* b, InitBufs& w, size_t total, int g, double* out) {
  hipLaunchKernelGGL(k_init_dot, dim3(g), dim3(kBlock), 0, h->stream, a, b, w.partial, total);
  HIPC(hipGetLastError());
  std::vector<double> host(g);
  HIPC(hipMemcpyAsync(host.data(), w.partial, sizeof(double) * g, hipMemcpyDeviceToHost, h->stream));
  HIPC(hipStreamSynchronize(h->stream));
  double s = 0.0;
  for (double v : host) s += v;
  *out = s;
  return DPGO_OK;
}
int init_pcg(dpgo_problem_s* h, InitBufs& w, const double* rhs, int mode, double tol, int max_iter, int* iters) {
  const int T = h->T, R = h->r, D = h->d;
  const size_t total = (size_t)h->n * T;
  const int g = std::max(1, std::min(kMaxGrid, (int)((total + kBlock - 1) / kBlock)));
  auto axpby = [&](double a, const double* x, double b, double* y) -> int {
    hipLaunchKernelGGL(k_init_axpby, dim3(g), dim3(kBlock), 0, h->stream, a, x, b, y, total, T, R, D, mode);
    HIPC(hipGetLastError());
    return DPGO_OK;
  };
  auto apply = [&](const double* v, double* out) -> int {  // out = mask(A v), v masked
    CHK(launch_spmm(h, h->Q, v, nullptr, out));
    return axpby(1.0, out, 0.0, out);
  };
  auto precond = [&](const double* r, double* z) -> int {
    hipLaunchKernelGGL(k_init_jacobi, dim3(g), dim3(kBlock), 0, h->stream, r, w.diag, z, total, T, R, D, mode);
    HIPC(hipGetLastError());
    return DPGO_OK;
  };
  HIPC(hipMemsetAsync(w.x, 0, sizeof(double) * total, h->stream));
  CHK(axpby(1.0, rhs, 0.0, w.r));
  CHK(precond(w.r, w.z));
  CHK(axpby(1.0, w.z, 0.0, w.p));
  double rz = 0.0, r0 = 0.0;
  CHK(init_dot(h, w.r, w.z, w, total, g, &rz));
  CHK(init_dot(h, w.r, w.r, w, total, g, &r0));
  *iters = 0;
  if (!(r0 > 0.0)) return DPGO_OK;
  double best = r0;
  for (int it = 0; it < max_iter; ++it) {
    CHK(apply(w.p, w.Ap));
    double pAp = 0.0;
    CHK(init_dot(h, w.p, w.Ap, w, total, g, &pAp));
    if (!(pAp > 0.0)) break;
    const double alpha = rz / pAp;
    CHK(axpby(alpha, w.p, 1.0, w.x));
    CHK(axpby(-alpha, w.Ap, 1.0, w.r));
    CHK(precond(w.r, w.z));
    double rz_new = 0.0, rr = 0.0;
    CHK(init_dot(h, w.r, w.z, w, total, g, &rz_new));
    CHK(init_dot(h, w.r, w.r, w, total, g, &rr));
    *iters = it + 1;
    best = std::min(best, rr);
    if (rr <= tol * tol * r0) break;
    CHK(axpby(1.0, w.z, rz_new / rz, w.p));
    rz = rz_new;
  }
  return DPGO_OK;
}
}  // namespace


int dpgo_chordal_initialization(int d, int n, int m, const int32_t* p1, const int32_t* p2, const double* R,
                                const double* t, const double* kappa, const double* tau, double tol, int max_iter,
                                double* T_host, int iters_out[2], int device) {
  if ((d != 2 && d != 3) || n <= 0 || m < 0 || !T_host || (m > 0 && (!p1 || !p2 || !R || !t || !kappa || !tau)))
    return fail(DPGO_ERR_INVALID, "bad arguments");
  if (!(tol > 0.0)) tol = 1e-13;
  if (max_iter <= 0) max_iter = (int)std::min<long long>(20ll * n + 100, 200000);
  const int b = d + 1, r = d;  // tiles [n][d+1][d]: the rank-d "lifted" problem IS the SE(d) problem
  std::vector<int32_t> zero(std::max(m, 1), 0);
  std::vector<double> ones(std::max(m, 1), 1.0), tau0(std::max(m, 1), 0.0);
  dpgo_problem_t hq[2] = {nullptr, nullptr};  // [0]: rotation-only connection Laplacian (tau = 0), [1]: Q
  struct Cleanup {
    dpgo_problem_t* h;
    ~Cleanup() {
      dpgo_problem_destroy(h[0]);
      dpgo_problem_destroy(h[1]);
    }
  } cleanup{hq};
  for (int which = 0; which < 2; ++which) {
    const double* tw = which == 0 ? tau0.data() : tau;
    int nnzb = 0;
    int rc = dpgo_build_Q_bsr(0, d, n, m, zero.data(), p1, zero.data(), p2, R, t, kappa, tw, ones.data(), 0, nullptr, 0.0,
                              0.0, &nnzb, nullptr, nullptr, nullptr);
    if (rc != DPGO_OK) return fail(rc, "chordal initialisation: measurement index out of range");
    std::vector<int32_t> rowptr(n + 1), colidx(nnzb);
    std::vector<double> vals((size_t)nnzb * b * b);
    rc = dpgo_build_Q_bsr(0, d, n, m, zero.data(), p1, zero.data(), p2, R, t, kappa, tw, ones.data(), 0, nullptr, 0.0, 0.0,
                          &nnzb, rowptr.data(), colidx.data(), vals.data());
    if (rc != DPGO_OK) return fail(rc, "chordal initialisation: could not build the data matrix");
    CHK(dpgo_problem_create(&hq[which], r, d, n, device));
    CHK(dpgo_problem_set_Q_bsr(hq[which], nnzb, rowptr.data(), colidx.data(), vals.data()));
  }
  dpgo_problem_s* hr = hq[0];
  dpgo_problem_s* ht = hq[1];
  const size_t total = (size_t)n * hr->T;
  TmpDev tmp;
  InitBufs w{};
  double *rhs = nullptr, *V = nullptr, *Tr = nullptr;
  for (double** v : {&w.x, &w.r, &w.z, &w.p, &w.Ap, &rhs, &V, &Tr}) CHK(tmp.alloc(v, sizeof(double) * total));
  CHK(tmp.alloc(&w.diag, sizeof(double) * (size_t)n * b));
  CHK(tmp.alloc(&w.partial, sizeof(double) * kMaxGrid));
  const int gflat = std::max(1, std::min(kMaxGrid, (n + kBlock - 1) / kBlock));
  const int gtot = std::max(1, std::min(kMaxGrid, (int)((total + kBlock - 1) / kBlock)));
  int it_rot = 0, it_tr = 0;
  // ---- rotations: minimise sum kappa |R_j - R_i R_ij|^2, R_0 = I.  With E0 = tile 0 = [I | 0]:  L (E0 + x) = 0 on
  // the free rows  =>  mask L mask x = -mask(L E0)
  std::vector<double> e0(hr->T, 0.0);
  for (int c = 0; c < d; ++c) e0[(size_t)c * r + c] = 1.0;
  HIPC(hipMemsetAsync(V, 0, sizeof(double) * total, hr->stream));
  HIPC(hipStreamSynchronize(hr->stream));  // (the host copy below comes from pageable memory: keep it strictly after)
  HIPC(hipMemcpyAsync(V, e0.data(), sizeof(double) * hr->T, hipMemcpyHostToDevice, hr->stream));
  HIPC(hipStreamSynchronize(hr->stream));
  CHK(launch_spmm(hr, hr->Q, V, nullptr, rhs));
  hipLaunchKernelGGL(k_init_axpby, dim3(gtot), dim3(kBlock), 0, hr->stream, -1.0, rhs, 0.0, rhs, total, hr->T, r, d, 0);
  if (d == 2)
    hipLaunchKernelGGL(k_init_diag<2>, dim3(gflat), dim3(kBlock), 0, hr->stream, hr->Q.dev(), w.diag, n);
  else
    hipLaunchKernelGGL(k_init_diag<3>, dim3(gflat), dim3(kBlock), 0, hr->stream, hr->Q.dev(), w.diag, n);
  HIPC(hipGetLastError());
  CHK(init_pcg(hr, w, rhs, 0, tol, max_iter, &it_rot));
  // V = E0 + x, then every block to SO(d) (projectToRotationGroup, src/DPGO_utils.cpp:464-478): the rounding kernel with
  // the identity as anchor
  hipLaunchKernelGGL(k_init_axpby, dim3(gtot), dim3(kBlock), 0, hr->stream, 1.0, w.x, 0.0, w.x, total, hr->T, r, d, 0);
  HIPC(hipMemcpyAsync(V, w.x, sizeof(double) * total, hipMemcpyDeviceToDevice, hr->stream));
  HIPC(hipStreamSynchronize(hr->stream));  // (as above)
  HIPC(hipMemcpyAsync(V, e0.data(), sizeof(double) * hr->T, hipMemcpyHostToDevice, hr->stream));
  HIPC(hipStreamSynchronize(hr->stream));
  CHK(dpgo_round_trajectory_device(r, d, n, V, e0.data(), Tr, hr->stream));
  HIPC(hipStreamSynchronize(hr->stream));
  // ---- translations: minimise sum tau |t_j - t_i - R_i t_ij|^2, t_0 = 0: the translation columns of Q [R | t] = 0
  CHK(launch_spmm(ht, ht->Q, Tr, nullptr, rhs));
  hipLaunchKernelGGL(k_init_axpby, dim3(gtot), dim3(kBlock), 0, ht->stream, -1.0, rhs, 0.0, rhs, total, ht->T, r, d, 1);
  if (d == 2)
    hipLaunchKernelGGL(k_init_diag<2>, dim3(gflat), dim3(kBlock), 0, ht->stream, ht->Q.dev(), w.diag, n);
  else
    hipLaunchKernelGGL(k_init_diag<3>, dim3(gflat), dim3(kBlock), 0, ht->stream, ht->Q.dev(), w.diag, n);
  HIPC(hipGetLastError());
  CHK(init_pcg(ht, w, rhs, 1, tol, max_iter, &it_tr));
  // T = [R | t]: rotation columns from Tr, translation column from the solve (pose 0: zero)
  hipLaunchKernelGGL(k_init_axpby, dim3(gtot), dim3(kBlock), 0, ht->stream, 1.0, w.x, 0.0, w.x, total, ht->T, r, d, 1);
  hipLaunchKernelGGL(k_axpby_plain, dim3(gtot), dim3(kBlock), 0, ht->stream, 1.0, w.x, 1.0, Tr, total);
  HIPC(hipGetLastError());
  HIPC(hipMemcpyAsync(T_host, Tr, sizeof(double) * total, hipMemcpyDeviceToHost, ht->stream));
  HIPC(hipStreamSynchronize(ht->stream));
  if (iters_out) {
    iters_out[0] = it_rot;
    iters_out[1] = it_tr;
  }
  return DPGO_OK;
}


int dpgo_odometry_initialization(int d, int n, int m, const int32_t* p1, const int32_t* p2, const double* R,
                                 const double* t, double* T_host) {
  if ((d != 2 && d != 3) || n <= 0 || m < 0 || !T_host || (m > 0 && (!p1 || !p2 || !R || !t)))
    return fail(DPGO_ERR_INVALID, "bad arguments");
  const int b = d + 1;
  std::vector<int> edge_of(n, -1);  // odometry edge leaving pose i (i -> i + 1)
  for (int e = 0; e < m; ++e)
    if (p1[e] >= 0 && p1[e] + 1 == p2[e] && p2[e] < n && edge_of[p1[e]] < 0) edge_of[p1[e]] = e;
  std::memset(T_host, 0, sizeof(double) * (size_t)n * b * d);
  for (int c = 0; c < d; ++c) T_host[(size_t)c * d + c] = 1.0;  // tile 0 = [I | 0]
  for (int dst = 1; dst < n; ++dst) {
    const int e = edge_of[dst - 1];
    if (e < 0) return fail(DPGO_ERR_INVALID, "odometry initialisation: no odometry edge " + std::to_string(dst - 1) +
                                                 " -> " + std::to_string(dst));  // reference: CHECK(m.p1 == src)
    const double* Ts = T_host + (size_t)(dst - 1) * b * d;  // tile [c][row]: R(row, c) at c*d + row, t(row) at d*d + row
    double* Td = T_host + (size_t)dst * b * d;
    const double* Re = R + (size_t)e * d * d;  // R[e][row][col]
    const double* te = t + (size_t)e * d;
    for (int row = 0; row < d; ++row) {
      for (int c = 0; c < d; ++c) {
        double s = 0.0;
        for (int k = 0; k < d; ++k) s += Ts[(size_t)k * d + row] * Re[k * d + c];  // (R_src R_e)(row, c)
        Td[(size_t)c * d + row] = s;
      }
      double s = Ts[(size_t)d * d + row];
      for (int k = 0; k < d; ++k) s += Ts[(size_t)k * d + row] * te[k];  // t_src + R_src t_e
      Td[(size_t)d * d + row] = s;
    }
  }
  return DPGO_OK;
}


// ---- manifold ----
namespace {
int manifold_args(int r, int d, int n, int device) {
  if (n <= 0 || r < d || d < 2 || d > 3) return fail(DPGO_ERR_INVALID, "need n > 0, r >= d, d in {2,3}");
  if (!supported(d, r)) return fail(DPGO_ERR_UNSUPPORTED, "(d, r) not compiled in");
  int cnt = 0;
  CHK(dpgo_device_count(&cnt));
  if (cnt <= 0) return fail(DPGO_ERR_HIP, "no HIP device (this library has no CPU fallback)");
  if (device < 0 || device >= cnt) return fail(DPGO_ERR_INVALID, "device index out of range");
  HIPC(hipSetDevice(device));
  return DPGO_OK;
}
int tiles_grid(int d, int n) {
  const int P = (64 / (d + 1)) * kWaves;
  int t = (n + P - 1) / P;
  if (t < 1) t = 1;
  return t < kMaxGrid ? t : kMaxGrid;
}
}  // namespace


int dpgo_manifold_project_device(int r, int d, int n, const double* M_dev, double* out_dev, void* stream) {
  return dpgo_axpby_project_device(r, d, n, 1.0, M_dev, 0.0, nullptr, 0.0, nullptr, 1, out_dev, stream);
}


int dpgo_axpby_project_device(int r, int d, int n, double a, const double* A_dev, double b, const double* B_dev,
                              double c, const double* C_dev, int project, double* out_dev, void* stream) {
  if (!A_dev || !out_dev) return fail(DPGO_ERR_INVALID, "null pointer");
  if (n <= 0) return fail(DPGO_ERR_INVALID, "n <= 0");
  DISPATCH(d, r, hipLaunchKernelGGL((k_axpby_project<D, R>), dim3(tiles_grid(d, n)), dim3(kBlock), 0,
                                    (hipStream_t)stream, a, A_dev, b, B_dev, c, C_dev, project, out_dev, n));
  HIPC(hipGetLastError());
  return DPGO_OK;
}


int dpgo_round_trajectory_device(int r, int d, int n, const double* X_dev, const double* anchor_host, double* T_dev,
                                 void* stream) {
  if (!X_dev || !T_dev) return fail(DPGO_ERR_INVALID, "null pointer");
  if (n <= 0) return fail(DPGO_ERR_INVALID, "n <= 0");
  if (!dpgo_supported(d, r)) return fail(DPGO_ERR_UNSUPPORTED, "unsupported (d, r)");
  AnchorArg an;
  std::memset(&an, 0, sizeof(an));
  an.use = anchor_host ? 1 : 0;
  if (anchor_host) std::memcpy(an.v, anchor_host, sizeof(double) * (size_t)(d + 1) * r);
  int g = (n + kBlock - 1) / kBlock;
  if (g > kMaxGrid) g = kMaxGrid;
  DISPATCH(d, r, hipLaunchKernelGGL((k_round<D, R>), dim3(g), dim3(kBlock), 0, (hipStream_t)stream, X_dev, an, T_dev, n));
  HIPC(hipGetLastError());
  return DPGO_OK;
}


int dpgo_round_trajectory(int r, int d, int n, const double* X_host, const double* anchor_host, double* T_host,
                          int device) {
  if (!X_host || !T_host) return fail(DPGO_ERR_INVALID, "null pointer");
  CHK(manifold_args(r, d, n, device));
  TmpDev tmp;
  const size_t xb = sizeof(double) * (size_t)n * (d + 1) * r, tb = sizeof(double) * (size_t)n * (d + 1) * d;
  double *X = nullptr, *T = nullptr;
  CHK(tmp.alloc(&X, xb));
  CHK(tmp.alloc(&T, tb));
  HIPC(hipMemcpy(X, X_host, xb, hipMemcpyHostToDevice));
  CHK(dpgo_round_trajectory_device(r, d, n, X, anchor_host, T, nullptr));
  HIPC(hipMemcpy(T_host, T, tb, hipMemcpyDeviceToHost));
  return DPGO_OK;
}


int dpgo_gather_tiles_device(int r, int d, const double* src_dev, const int32_t* idx_dev, int count, double* dst_dev,
                             void* stream) {
  if (count == 0) return DPGO_OK;
  if (!src_dev || !idx_dev || !dst_dev || count < 0) return fail(DPGO_ERR_INVALID, "bad arguments");
  size_t total = (size_t)count * (d + 1) * r;
  int g = (int)((total + kBlock - 1) / kBlock);
  if (g > kMaxGrid) g = kMaxGrid;
  DISPATCH(d, r, hipLaunchKernelGGL((k_gather_tiles<D, R>), dim3(g), dim3(kBlock), 0, (hipStream_t)stream, src_dev,
                                    idx_dev, count, dst_dev));
  HIPC(hipGetLastError());
  return DPGO_OK;
}


int dpgo_permute_tiles_device(int r, int d, int n, const int32_t* new_index_dev, const double* in_dev, double* out_dev,
                              int forward, void* stream) {
  if (n == 0) return DPGO_OK;
  if (!new_index_dev || !in_dev || !out_dev || n < 0 || in_dev == out_dev) return fail(DPGO_ERR_INVALID, "bad arguments");
  size_t total = (size_t)n * (d + 1) * r;
  int g = (int)((total + kBlock - 1) / kBlock);
  if (g > kMaxGrid) g = kMaxGrid;
  if (forward) {
    DISPATCH(d, r, hipLaunchKernelGGL((k_scatter_tiles<D, R>), dim3(g), dim3(kBlock), 0, (hipStream_t)stream, in_dev,
                                      new_index_dev, n, out_dev));
  } else {
    DISPATCH(d, r, hipLaunchKernelGGL((k_gather_tiles<D, R>), dim3(g), dim3(kBlock), 0, (hipStream_t)stream, in_dev,
                                      new_index_dev, n, out_dev));
  }
  HIPC(hipGetLastError());
  return DPGO_OK;
}


struct dpgo_exchange_plan_s {
  int device = 0, T = 0, nmsg = 0, total = 0;
  void *src = nullptr, *idx = nullptr, *dst = nullptr, *first = nullptr;
};


int dpgo_exchange_plan_create(dpgo_exchange_plan_t* out, int r, int d, int nmsg, const double* const* src_dev,
                              const int32_t* const* idx_dev, const int* count, double* const* dst_dev, int device) {
  if (!out || nmsg <= 0 || !src_dev || !idx_dev || !count || !dst_dev || !supported(d, r))
    return fail(DPGO_ERR_INVALID, "bad exchange plan arguments");
  *out = nullptr;
  std::vector<int32_t> first(nmsg + 1, 0);
  for (int m = 0; m < nmsg; ++m) {
    if (count[m] < 0 || (count[m] > 0 && (!src_dev[m] || !idx_dev[m] || !dst_dev[m]))) return fail(DPGO_ERR_INVALID, "bad message");
    first[m + 1] = first[m] + count[m];
  }
  HIPC(hipSetDevice(device));
  auto* pl = new dpgo_exchange_plan_s();
  pl->device = device;
  pl->T = (d + 1) * r;
  pl->nmsg = nmsg;
  pl->total = first[nmsg];
  int rc = [&]() -> int {
    HIPC(hipMalloc(&pl->src, sizeof(void*) * nmsg));
    HIPC(hipMalloc(&pl->idx, sizeof(void*) * nmsg));
    HIPC(hipMalloc(&pl->dst, sizeof(void*) * nmsg));
    HIPC(hipMalloc(&pl->first, sizeof(int32_t) * (nmsg + 1)));
    HIPC(hipMemcpy(pl->src, src_dev, sizeof(void*) * nmsg, hipMemcpyHostToDevice));
    HIPC(hipMemcpy(pl->idx, idx_dev, sizeof(void*) * nmsg, hipMemcpyHostToDevice));
    HIPC(hipMemcpy(pl->dst, dst_dev, sizeof(void*) * nmsg, hipMemcpyHostToDevice));
    HIPC(hipMemcpy(pl->first, first.data(), sizeof(int32_t) * (nmsg + 1), hipMemcpyHostToDevice));
    return DPGO_OK;
  }();
  if (rc != DPGO_OK) {
    dpgo_exchange_plan_destroy(pl);
    return rc;
  }
  *out = pl;
  return DPGO_OK;
}


int dpgo_exchange_plan_run(dpgo_exchange_plan_t pl, void* stream) {
  if (!pl) return fail(DPGO_ERR_INVALID, "null exchange plan");
  if (pl->total == 0) return DPGO_OK;
  HIPC(hipSetDevice(pl->device));
  const ExchangeTable tb{(const double* const*)pl->src, (const int32_t* const*)pl->idx, (double* const*)pl->dst,
                         (const int32_t*)pl->first, pl->nmsg};
  const int g = std::max(1, std::min(kMaxGrid, (pl->total + kBlock / 4 - 1) / (kBlock / 4)));
  switch (pl->T) {
#define CASE_T(TT) case TT: hipLaunchKernelGGL((k_gather_tiles_batched<TT>), dim3(g), dim3(kBlock), 0, (hipStream_t)stream, tb); break;
    CASE_T(6) CASE_T(9) CASE_T(12) CASE_T(15) CASE_T(16) CASE_T(20) CASE_T(24)
#undef CASE_T
    default: return fail(DPGO_ERR_UNSUPPORTED, "unsupported (d, r)");
  }
  HIPC(hipGetLastError());
  return DPGO_OK;
}


// ---- ordering words of the peer-store transport (kernels/agent.h: k_flags_write / k_flags_wait) ----
static int flag_table(int n, unsigned long long* const* words_dev, const unsigned long long* values, int first, FlagTable* t) {
  t->n = std::min(kFlagCap, n - first);
  for (int k = 0; k < t->n; ++k) {
    if (!words_dev[first + k]) return fail(DPGO_ERR_INVALID, "null ordering word");
    t->p[k] = words_dev[first + k];
    t->v[k] = values[first + k];
  }
  return DPGO_OK;
}
int dpgo_flags_write_device(int n, unsigned long long* const* words_dev, const unsigned long long* values, void* stream) {
  if (n < 0 || (n > 0 && (!words_dev || !values))) return fail(DPGO_ERR_INVALID, "bad ordering-word arguments");
  for (int first = 0; first < n; first += kFlagCap) {
    FlagTable t;
    CHK(flag_table(n, words_dev, values, first, &t));
    hipLaunchKernelGGL(k_flags_write, dim3(1), dim3(64), 0, (hipStream_t)stream, t);
  }
  HIPC(hipGetLastError());
  return DPGO_OK;
}
namespace {
int flags_wait(int n, unsigned long long* const* words_dev, const unsigned long long* values, long long timeout_ms,
               unsigned long long* err_word, void* stream) {
  for (int first = 0; first < n; first += kFlagCap) {
    FlagTable t;
    CHK(flag_table(n, words_dev, values, first, &t));
    hipLaunchKernelGGL(k_flags_wait, dim3(1), dim3(64), 0, (hipStream_t)stream, t, timeout_ms * 100000LL, err_word);
  }
  HIPC(hipGetLastError());
  return DPGO_OK;
}
}  // namespace
int dpgo_flags_wait_device(int n, unsigned long long* const* words_dev, const unsigned long long* values, int timeout_ms,
                           void* stream) {
  if (n < 0 || (n > 0 && (!words_dev || !values)) || timeout_ms <= 0) return fail(DPGO_ERR_INVALID, "bad ordering-word arguments");
  return flags_wait(n, words_dev, values, timeout_ms, nullptr, stream);
}
int dpgo_flags_wait_device_checked(int n, unsigned long long* const* words_dev, const unsigned long long* values,
                                   long long timeout_ms, unsigned long long* err_word, void* stream) {
  if (n < 0 || (n > 0 && (!words_dev || !values)) || timeout_ms < 0 || (timeout_ms > 0 && !err_word))
    return fail(DPGO_ERR_INVALID, "bad ordering-word arguments");
  return flags_wait(n, words_dev, values, timeout_ms, err_word, stream);
}

int dpgo_exchange_plan_destroy(dpgo_exchange_plan_t pl) {
  if (!pl) return DPGO_OK;
  for (void* q : {pl->src, pl->idx, pl->dst, pl->first})
    if (q) (void)hipFree(q);
  delete pl;
  return DPGO_OK;
}


int dpgo_max_translation_distance_device(int r, int d, int n, const double* X_dev, const double* Xprev_dev,
                                         double* out_dev, double* out_host, void* stream) {
  if (!X_dev || !Xprev_dev || !out_dev || n <= 0) return fail(DPGO_ERR_INVALID, "bad arguments");
  hipStream_t s = (hipStream_t)stream;
  HIPC(hipMemsetAsync(out_dev, 0, sizeof(double), s));
  const int g = std::max(1, std::min(kMaxGrid, (n + kBlock - 1) / kBlock));
  DISPATCH(d, r, hipLaunchKernelGGL((k_max_translation_distance<D, R>), dim3(g), dim3(kBlock), 0, s, X_dev, Xprev_dev, n,
                                    reinterpret_cast<unsigned long long*>(out_dev)));
  HIPC(hipGetLastError());
  if (out_host) {
    HIPC(hipMemcpyAsync(out_host, out_dev, sizeof(double), hipMemcpyDeviceToHost, s));
    HIPC(hipStreamSynchronize(s));
  }
  return DPGO_OK;
}


int dpgo_manifold_project(int r, int d, int n, const double* M, double* out, int device) {
  if (!M || !out) return fail(DPGO_ERR_INVALID, "null pointer");
  CHK(manifold_args(r, d, n, device));
  TmpDev tmp;
  const size_t vb = sizeof(double) * (size_t)n * (d + 1) * r;
  double *a = nullptr, *o = nullptr;
  CHK(tmp.alloc(&a, vb));
  CHK(tmp.alloc(&o, vb));
  HIPC(hipMemcpy(a, M, vb, hipMemcpyHostToDevice));
  CHK(dpgo_manifold_project_device(r, d, n, a, o, nullptr));
  HIPC(hipMemcpy(out, o, vb, hipMemcpyDeviceToHost));
  return DPGO_OK;
}


int dpgo_manifold_tangent_project(int r, int d, int n, const double* X, const double* V, double* out, int device) {
  if (!X || !V || !out) return fail(DPGO_ERR_INVALID, "null pointer");
  CHK(manifold_args(r, d, n, device));
  TmpDev tmp;
  const size_t vb = sizeof(double) * (size_t)n * (d + 1) * r;
  double *x = nullptr, *v = nullptr, *o = nullptr;
  CHK(tmp.alloc(&x, vb));
  CHK(tmp.alloc(&v, vb));
  CHK(tmp.alloc(&o, vb));
  HIPC(hipMemcpy(x, X, vb, hipMemcpyHostToDevice));
  HIPC(hipMemcpy(v, V, vb, hipMemcpyHostToDevice));
  DISPATCH(d, r, hipLaunchKernelGGL((k_precond<D, R>), dim3(tiles_grid(d, n)), dim3(kBlock), 0, (hipStream_t) nullptr,
                                    x, v, (const double*)nullptr, o, n));
  HIPC(hipGetLastError());
  HIPC(hipMemcpy(out, o, vb, hipMemcpyDeviceToHost));
  return DPGO_OK;
}


int dpgo_manifold_retract(int r, int d, int n, const double* X, const double* eta, double scale, double* out,
                          int device) {
  if (!X || !eta || !out) return fail(DPGO_ERR_INVALID, "null pointer");
  CHK(manifold_args(r, d, n, device));
  TmpDev tmp;
  const size_t vb = sizeof(double) * (size_t)n * (d + 1) * r;
  double *x = nullptr, *v = nullptr, *o = nullptr;
  CHK(tmp.alloc(&x, vb));
  CHK(tmp.alloc(&v, vb));
  CHK(tmp.alloc(&o, vb));
  HIPC(hipMemcpy(x, X, vb, hipMemcpyHostToDevice));
  HIPC(hipMemcpy(v, eta, vb, hipMemcpyHostToDevice));
  DISPATCH(d, r, hipLaunchKernelGGL((k_retract<D, R>), dim3(tiles_grid(d, n)), dim3(kBlock), 0, (hipStream_t) nullptr,
                                    x, v, scale, o, (const DevState*)nullptr, n));
  HIPC(hipGetLastError());
  HIPC(hipMemcpy(out, o, vb, hipMemcpyDeviceToHost));
  return DPGO_OK;
}

}  // extern "C"
