// problem.hip -- handle lifecycle, data matrices and QuadraticProblem evaluations (src/QuadraticProblem.cpp, src/PoseGraph.cpp:381-580).
#include "host.h"

namespace dpgo_host {

int set_device(dpgo_problem_s* p) {
  HIPC(hipSetDevice(p->device));
  return DPGO_OK;
}

int upload_bsr(Bsr& m, int nrows, int ncols, int nnzb, int b, const int32_t* rowptr, const int32_t* colidx,
               const double* vals, hipStream_t s) {
  CHK(free_bsr(m));
  m.nrows = nrows;
  m.ncols = ncols;
  m.nnzb = nnzb;
  HIPC(hipMalloc(&m.rowptr, sizeof(int32_t) * (nrows + 1)));
  HIPC(hipMalloc(&m.colidx, sizeof(int32_t) * (nnzb > 0 ? nnzb : 1)));
  HIPC(hipMalloc(&m.vals, sizeof(double) * (size_t)(nnzb > 0 ? nnzb : 1) * b * b));
  HIPC(hipMemcpyAsync(m.rowptr, rowptr, sizeof(int32_t) * (nrows + 1), hipMemcpyHostToDevice, s));
  if (nnzb > 0) {
    HIPC(hipMemcpyAsync(m.colidx, colidx, sizeof(int32_t) * nnzb, hipMemcpyHostToDevice, s));
    if (vals) HIPC(hipMemcpyAsync(m.vals, vals, sizeof(double) * (size_t)nnzb * b * b, hipMemcpyHostToDevice, s));
  }
  HIPC(hipStreamSynchronize(s));
  return DPGO_OK;
}


int validate_bsr(int nrows, int ncols, int nnzb, const int32_t* rowptr, const int32_t* colidx, bool need_diag) {
  if (!rowptr || nnzb < 0 || (nnzb > 0 && !colidx)) return fail(DPGO_ERR_INVALID, "null BSR arrays");
  if (rowptr[0] != 0 || rowptr[nrows] != nnzb) return fail(DPGO_ERR_INVALID, "BSR rowptr does not span nnzb");
  for (int i = 0; i < nrows; ++i) {
    if (rowptr[i + 1] < rowptr[i]) return fail(DPGO_ERR_INVALID, "BSR rowptr not monotone");
    bool diag = false;
    for (int t = rowptr[i]; t < rowptr[i + 1]; ++t) {
      if (colidx[t] < 0 || colidx[t] >= ncols) return fail(DPGO_ERR_INVALID, "BSR column index out of range");
      if (t > rowptr[i] && colidx[t] <= colidx[t - 1])
        return fail(DPGO_ERR_INVALID, "BSR column indices must be sorted and unique within a block row");
      if (colidx[t] == i) diag = true;
    }
    if (need_diag && !diag) return fail(DPGO_ERR_INVALID, "BSR block row without diagonal block");
  }
  return DPGO_OK;
}

int build_dinv(dpgo_problem_s* p, double shift) {
  if (!p->Q.vals) return fail(DPGO_ERR_STATE, "Q not set");
  if (p->dinv_shift == shift) return DPGO_OK;
  const int g = (p->n + kBlock - 1) / kBlock;
  if (p->d == 2)
    hipLaunchKernelGGL(k_build_dinv<2>, dim3(g), dim3(kBlock), 0, p->stream, p->Q.dev(), shift, p->dinv, p->n);
  else
    hipLaunchKernelGGL(k_build_dinv<3>, dim3(g), dim3(kBlock), 0, p->stream, p->Q.dev(), shift, p->dinv, p->n);
  HIPC(hipGetLastError());
  p->dinv_shift = shift;
  return DPGO_OK;
}

int poll_state(dpgo_problem_s* p) {
  HIPC(hipMemcpyAsync(p->hstate, p->dstate + p->cur, sizeof(DevState), hipMemcpyDeviceToHost, p->stream));
  HIPC(hipStreamSynchronize(p->stream));
  return DPGO_OK;
}

int push_state(dpgo_problem_s* p) {
  HIPC(hipMemcpyAsync(p->dstate + 0, p->hstate, sizeof(DevState), hipMemcpyHostToDevice, p->stream));
  HIPC(hipMemcpyAsync(p->dstate + 1, p->hstate, sizeof(DevState), hipMemcpyHostToDevice, p->stream));
  HIPC(hipStreamSynchronize(p->stream));
  return DPGO_OK;
}

// ---- symmetric copy of Q (plain SpMM on cold blocks) ----
void sym_free(dpgo_problem_s* p) {
  auto& S = p->sym;
  void* ptrs[] = {S.urow, S.ucol, S.usrc, S.lrow, S.lcol, S.lslot, S.lsrc, S.uvalsT, S.flag, S.uvalsT32, S.tord};
  for (void* q : ptrs)
    if (q) (void)hipFree(q);
  S = dpgo_problem_s::SymQ();
  p->tcg_sym = false;
}


// pattern arrays from the host copy of Q's block pattern; a pattern that is not structurally symmetric leaves
// pattern_ok = false (the plain kernel stays in use), it is not an error
int sym_symbolic_setup(dpgo_problem_s* p) {
  sym_free(p);
  auto& S = p->sym;
  S.symbolic = true;
  const int n = p->n;
  if ((int)p->h_rowptr.size() != n + 1) return DPGO_OK;
  const auto& rp = p->h_rowptr;
  const auto& ci = p->h_colidx;
  std::vector<int32_t> urow(n + 1, 0), lrow(n + 1, 0), ucol, usrc, lcol, lslot, lsrc;
  ucol.reserve(ci.size() / 2 + n);
  for (int i = 0; i < n; ++i) {
    for (int t = rp[i]; t < rp[i + 1]; ++t)
      if (ci[t] >= i) {
        ucol.push_back(ci[t]);
        usrc.push_back(t);
      }
    urow[i + 1] = (int32_t)ucol.size();
  }
  for (int i = 0; i < n; ++i) {
    for (int t = rp[i]; t < rp[i + 1]; ++t) {
      const int j = ci[t];
      if (j >= i) break;  // columns are sorted
      const auto b0 = ucol.begin() + urow[j], b1 = ucol.begin() + urow[j + 1];
      const auto it = std::lower_bound(b0, b1, (int32_t)i);
      if (it == b1 || *it != i) return DPGO_OK;  // block (i, j) without block (j, i)
      lcol.push_back(j);
      lslot.push_back((int32_t)(it - ucol.begin()));
      lsrc.push_back(t);
    }
    lrow[i + 1] = (int32_t)lcol.size();
  }
  if (ucol.size() + lcol.size() != ci.size()) return DPGO_OK;
  if (2 * lcol.size() + (size_t)n != ci.size()) return DPGO_OK;  // an upper block without its lower one
  S.nu = (int)ucol.size();
  S.nl = (int)lcol.size();
  CHK(sym_upload(&S.urow, urow, p->stream));
  CHK(sym_upload(&S.ucol, ucol, p->stream));
  CHK(sym_upload(&S.usrc, usrc, p->stream));
  CHK(sym_upload(&S.lrow, lrow, p->stream));
  CHK(sym_upload(&S.lcol, lcol, p->stream));
  CHK(sym_upload(&S.lslot, lslot, p->stream));
  CHK(sym_upload(&S.lsrc, lsrc, p->stream));
  HIPC(hipMalloc(&S.uvalsT, sizeof(double) * (size_t)std::max(1, S.nu) * p->b * p->b));
  HIPC(hipMalloc(&S.flag, sizeof(int)));
  if (options().tile_walk != 0) {
    // The walk over the workgroup tiles of the symmetric-storage kernels (BsrSymDevT::tord).  tile_iter gives XCD x the
    // contiguous eighth [T x / 8, T (x + 1) / 8) of the T tiles and its workgroups walk it with a stride of their count, so
    // the tiles one XCD has in flight together are ~96 CONSECUTIVE positions of its eighth.  In index order those are 2.5
    // lattice layers of the 100k grid: the z-neighbours (+-2 500 poses = 39 tiles) of 40 % of the rows belong to another
    // round -- their z tiles and the upper blocks their lower references point to are fetched again (PMC: 1.18x the
    // stored bytes, rounds 2-4).  Breadth-first order over the tile graph inside each eighth makes consecutive positions
    // graph neighbours (on the lattice: all layers of a strip of rows before the next strip); the pose order -- the data
    // layout, the odometry runs the span loads rely on -- is untouched (round 4's pose renumbering lost exactly there).
    const int P = (64 / p->b) * kWaves, T = (n + P - 1) / P;
    if (T >= 16) {
      std::vector<std::vector<int32_t>> adj(T);
      for (int i = 0; i < n; ++i) {
        const int ti = i / P;
        for (int t = rp[i]; t < rp[i + 1]; ++t) {
          const int tj = ci[t] / P;
          if (tj != ti && (adj[ti].empty() || adj[ti].back() != tj)) adj[ti].push_back(tj);
        }
      }
      for (auto& a : adj) {
        std::sort(a.begin(), a.end());
        a.erase(std::unique(a.begin(), a.end()), a.end());
      }
      std::vector<int32_t> order;
      order.reserve(T);
      std::vector<char> seen(T, 0);
      for (int x = 0; x < 8; ++x) {
        const int lo = (int)(((long long)T * x) >> 3), hi = (int)(((long long)T * (x + 1)) >> 3);
        for (int seed = lo; seed < hi; ++seed) {  // (components in index order; one on a connected share)
          if (seen[seed]) continue;
          size_t head = order.size();
          order.push_back(seed);
          seen[seed] = 1;
          while (head < order.size()) {
            const int u = order[head++];
            for (int v : adj[u])
              if (v >= lo && v < hi && !seen[v]) {
                seen[v] = 1;
                order.push_back(v);
              }
          }
        }
      }
      if ((int)order.size() == T) {
        CHK(sym_upload(&S.tord, order, p->stream));
        HIPC(hipStreamSynchronize(p->stream));  // (`order` goes out of scope)
      }
    }
  }
  HIPC(hipStreamSynchronize(p->stream));  // the host vectors go out of scope
  S.pattern_ok = true;
  return DPGO_OK;
}

// true when the symmetric copy is usable for Q's current values (refreshes it when they changed)
int sym_ensure(dpgo_problem_s* p, bool* usable) {
  *usable = false;
  auto& S = p->sym;
  if (!S.symbolic) CHK(sym_symbolic_setup(p));
  if (!S.pattern_ok) return DPGO_OK;
  if (!S.ready) {
    const size_t total = (size_t)S.nu * p->b * p->b;
    const int g = (int)std::min<size_t>(kMaxGrid, (total + kBlock - 1) / kBlock);
    HIPC(hipMemsetAsync(S.flag, 0, sizeof(int), p->stream));
    if (p->d == 2) {
      hipLaunchKernelGGL(k_sym_refresh<2>, dim3(g), dim3(kBlock), 0, p->stream, p->Q.vals, S.usrc, S.uvalsT, S.nu);
      hipLaunchKernelGGL(k_sym_check<2>, dim3(g), dim3(kBlock), 0, p->stream, p->Q.vals, S.lsrc, S.lslot, S.uvalsT, S.nl,
                         S.flag);
    } else {
      hipLaunchKernelGGL(k_sym_refresh<3>, dim3(g), dim3(kBlock), 0, p->stream, p->Q.vals, S.usrc, S.uvalsT, S.nu);
      hipLaunchKernelGGL(k_sym_check<3>, dim3(g), dim3(kBlock), 0, p->stream, p->Q.vals, S.lsrc, S.lslot, S.uvalsT, S.nl,
                         S.flag);
    }
    HIPC(hipGetLastError());
    int bad = 0;
    HIPC(hipMemcpyAsync(&bad, S.flag, sizeof(int), hipMemcpyDeviceToHost, p->stream));
    HIPC(hipStreamSynchronize(p->stream));
    S.values_ok = (bad == 0);
    S.ready = true;
    p->ml_ops32_ready = false;  // (the fp32 copy of these values is rebuilt by the next solve that streams it)
  }
  *usable = S.values_ok;
  return DPGO_OK;
}

// ---- kernel launch helpers (templated on D, R through DISPATCH) ----
int launch_spmm_sym(dpgo_problem_s* p, const BsrSymDev& M, const double* V, const double* Gadd, double* OUT) {
  const int g = p->grid_spmm_sym();
  DISPATCH(p->d, p->r,
           {
             if (p->want_stream_nt())
               hipLaunchKernelGGL((k_spmm_sym<D, R, 1>), dim3(g), dim3(kBlock), 0, p->stream, M, V, Gadd, OUT, p->n);
             else
               hipLaunchKernelGGL((k_spmm_sym<D, R, 0>), dim3(g), dim3(kBlock), 0, p->stream, M, V, Gadd, OUT, p->n);
           });
  HIPC(hipGetLastError());
  return DPGO_OK;
}

int launch_spmm(dpgo_problem_s* p, const Bsr& M, const double* V, const double* Gadd, double* OUT, int nrows) {
  if (&M == &p->Q && nrows < 0 && p->sym_wanted()) {
    bool usable = false;
    CHK(sym_ensure(p, &usable));
    if (usable) return launch_spmm_sym(p, p->sym.dev(), V, Gadd, OUT);
  }
  const int rows = nrows >= 0 ? nrows : p->n;
  int g = p->grid_spmm();
  if (nrows >= 0) {  // rectangular operator with its own row count (restriction)
    const int P = (64 / (p->b * p->split)) * kWaves;
    g = std::max(1, std::min(kMaxGrid, (rows + P - 1) / P));
  }
  DISPATCH(p->d, p->r, {
    if (p->want_stream_nt() && p->split == 1 && &M == &p->Q)
      hipLaunchKernelGGL((k_spmm<D, R, 1, 1>), dim3(g), dim3(kBlock), 0, p->stream, M.dev(), V, Gadd, OUT, rows);
    else
      LAUNCH_SPLIT(p, k_spmm, g, M.dev(), V, Gadd, OUT, rows);
  });
  HIPC(hipGetLastError());
  return DPGO_OK;
}

bool outer_sym_enabled() {  // tuning knob: DPGO_OUTER_SYM=0 keeps the outer iteration on the plain copy of Q
  return options().outer_sym != 0;
}
// `sym`: inside a solve whose tCG-step kernel reads the symmetric copy of Q (p->tcg_sym, valid for the duration of the
// solve) the gradient and the rho-test Hessian read it too: 7 us less per launch at 100k poses with cold operands, and the
// outer iteration no longer streams the 91 MB of the plain copy through the Infinity Cache the tCG loop lives in.
int launch_grad(dpgo_problem_s* p, const double* X, double* RG, double* S, double* EG,
                const DevState* st, bool sym) {
  const double* Gm = p->has_G ? p->G : nullptr;
  if (sym && p->split == 1) {
    p->nb_grad = p->grid_outer_sym();
    DISPATCH(p->d, p->r, hipLaunchKernelGGL((k_grad<D, R, 1, BsrSymDev>), dim3(p->nb_grad), dim3(kBlock), 0, p->stream,
                                            p->sym.dev(), X, Gm, RG, S, EG, p->pE(), st, p->n));
  } else {
    p->nb_grad = p->grid_s();
    DISPATCH(p->d, p->r, LAUNCH_SPLIT(p, k_grad, p->nb_grad, p->Q.dev(), X, Gm, RG, S, EG, p->pE(), st, p->n));
  }
  HIPC(hipGetLastError());
  return DPGO_OK;
}

int launch_hess(dpgo_problem_s* p, const double* X, const double* S, const double* V, const double* Gdot,
                double* HV, double* partials, const DevState* st, int check_tcg, bool sym) {
  if (sym && p->split == 1) {
    p->nb_hess = p->grid_outer_sym();
    DISPATCH(p->d, p->r, hipLaunchKernelGGL((k_hess<D, R, 1, BsrSymDev>), dim3(p->nb_hess), dim3(kBlock), 0, p->stream,
                                            p->sym.dev(), X, S, V, Gdot, HV, partials, st, check_tcg, p->n));
  } else {
    p->nb_hess = p->grid_s();
    DISPATCH(p->d, p->r,
             LAUNCH_SPLIT(p, k_hess, p->nb_hess, p->Q.dev(), X, S, V, Gdot, HV, partials, st, check_tcg, p->n));
  }
  HIPC(hipGetLastError());
  return DPGO_OK;
}

int launch_retract(dpgo_problem_s* p, const double* X, const double* eta, double scale, double* X2,
                   const DevState* st) {
  // (no partial sums, 50 VGPRs: not bound to the update kernel's grid -- at most DPGO_GRID_RETRACT workgroups, 1 024 by default; 512 / 1 024 / 1 563 measured 259.4 / 261.4 / 260.0 it/s)
  const int tiles_r = std::max(1, (p->n + (64 / p->b) * kWaves - 1) / ((64 / p->b) * kWaves));
  const int gr = std::min(tiles_r, options().grid_retract > 0 ? options().grid_retract : kMaxGrid);
  DISPATCH(p->d, p->r, hipLaunchKernelGGL((k_retract<D, R>), dim3(gr), dim3(kBlock), 0, p->stream, X, eta,
                                          scale, X2, st, p->n));
  HIPC(hipGetLastError());
  return DPGO_OK;
}

int launch_rtr_update(dpgo_problem_s* p) {
  // (the partial sums of the last k_grad / k_hess launch: their grids, launch_grad / launch_hess)
  const int ge = p->nb_grad > 0 ? p->nb_grad : p->grid_s(), gh = p->nb_hess > 0 ? p->nb_hess : p->grid_s();
  DISPATCH(p->d, p->r,
           hipLaunchKernelGGL((k_rtr_update<D, R>), dim3(p->grid_flat()), dim3(kBlock), 0, p->stream, p->x1, p->x2,
                              p->g1, p->g2, p->S1, p->S2, p->pE(), ge, p->pH(), gh, p->dstate + p->cur,
                              p->dstate + (p->cur ^ 1), p->n));
  HIPC(hipGetLastError());
  p->cur ^= 1;
  return DPGO_OK;
}

int launch_precond(dpgo_problem_s* p, const double* X, const double* V, const double* dinv, double* Z) {
  DISPATCH(p->d, p->r, hipLaunchKernelGGL((k_precond<D, R>), dim3(p->grid()), dim3(kBlock), 0, p->stream, X, V,
                                          dinv, Z, p->n));
  HIPC(hipGetLastError());
  return DPGO_OK;
}

int launch_rtr_begin(dpgo_problem_s* p, double tol, double Delta0, double Dmax, int max_inner, int tiny) {
  hipLaunchKernelGGL(k_rtr_begin, dim3(1), dim3(kBlock), 0, p->stream, p->pE(), p->nb_grad > 0 ? p->nb_grad : p->grid_s(), p->dstate, tol, Delta0,
                     Dmax, max_inner, tiny);
  HIPC(hipGetLastError());
  p->cur = 0;
  return DPGO_OK;
}

int check_ready(dpgo_problem_s* p) {
  if (!p) return fail(DPGO_ERR_INVALID, "null handle");
  if (!p->Q.vals) return fail(DPGO_ERR_STATE, "quadratic matrix Q not set");
  return set_device(p);
}

int h2d(dpgo_problem_s* p, double* dst, const double* src) {
  HIPC(hipMemcpyAsync(dst, src, p->vec_bytes(), hipMemcpyHostToDevice, p->stream));
  return DPGO_OK;
}
int d2h(dpgo_problem_s* p, double* dst, const double* src) {
  HIPC(hipMemcpyAsync(dst, src, p->vec_bytes(), hipMemcpyDeviceToHost, p->stream));
  HIPC(hipStreamSynchronize(p->stream));
  return DPGO_OK;
}

}  // namespace dpgo_host

extern "C" {


const char* dpgo_version(void) { return "dpgo_hip 0.1 (gfx950)"; }
const char* dpgo_last_error(void) { return g_err.c_str(); }
void dpgo_set_last_error(const char* msg) { g_err = msg ? msg : ""; }  // for the other translation units of the library

int dpgo_device_count(int* count) {
  if (!count) return fail(DPGO_ERR_INVALID, "null count");
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  if (e != hipSuccess) {
    *count = 0;
    return fail(DPGO_ERR_HIP, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
  }
  *count = c;
  return DPGO_OK;
}


void dpgo_ropt_params_default(dpgo_ropt_params* p) {
  if (!p) return;
  p->method = DPGO_METHOD_RTR;
  p->verbose = 0;
  p->gradnorm_tol = 1e-2;
  p->RGD_stepsize = 1e-3;
  p->RGD_use_preconditioner = 1;
  p->RTR_iterations = 3;
  p->RTR_tCG_iterations = 50;
  p->RTR_initial_radius = 100.0;
  p->precond = DPGO_PRECOND_AUTO;
  p->precond_shift = 1e-1;
  p->accept_tiny_decrease = 1;
  p->tcg_poll_interval = 0;
  p->time_bound_s = 5.0;
}


int dpgo_supported(int d, int r) { return supported(d, r) ? 1 : 0; }

int dpgo_problem_create(dpgo_problem_t* out, int r, int d, int n, int device) {
  if (!out) return fail(DPGO_ERR_INVALID, "null out");
  *out = nullptr;
  if (n <= 0 || r < d || d < 2 || d > 3) return fail(DPGO_ERR_INVALID, "need n > 0, r >= d, d in {2,3}");
  if (!supported(d, r)) return fail(DPGO_ERR_UNSUPPORTED, "(d, r) not compiled in");
  int cnt = 0;
  CHK(dpgo_device_count(&cnt));
  if (cnt <= 0) return fail(DPGO_ERR_HIP, "no HIP device (this library has no CPU fallback)");
  if (device < 0 || device >= cnt) return fail(DPGO_ERR_INVALID, "device index out of range");
  TaskPool::get().warm(setup_threads());  // (the set-up code's worker threads exist before their first use)
  auto* p = new dpgo_problem_s();
  p->r = r;
  p->d = d;
  p->n = n;
  p->b = d + 1;
  p->T = p->b * r;
  p->device = device;
  // small blocks are latency-bound: spread each row over 4 lane groups
  p->split = (n < 40000) ? 4 : 1;
  if (const int v = options().split)
    if (v == 1 || v == 2 || v == 4) p->split = v;
  if (options().ml_operator_bits == 64) p->ml_operator_bits = 64;
  int rc = [&]() -> int {
    HIPC(hipSetDevice(device));
    CHK(tune_launch_caps(p));
    HIPC(hipStreamCreateWithFlags(&p->own_stream, hipStreamNonBlocking));
    p->stream = p->own_stream;
    const size_t vb = p->vec_bytes();
    double** vecs[] = {&p->x1, &p->x2, &p->g1, &p->g2, &p->eta, &p->delta, &p->Hd, &p->rr, &p->z, &p->G, &p->G0};
    for (auto v : vecs) {
      HIPC(hipMalloc(v, vb));
      HIPC(hipMemsetAsync(*v, 0, vb, p->stream));
    }
    HIPC(hipMalloc(&p->S1, sizeof(double) * (size_t)n * d * d));
    HIPC(hipMalloc(&p->S2, sizeof(double) * (size_t)n * d * d));
    HIPC(hipMalloc(&p->dinv, sizeof(double) * (size_t)n * p->b * p->b));
    HIPC(hipMalloc(&p->partials, sizeof(double) * 5 * kPartialCap * kNP));
    HIPC(hipMemsetAsync(p->partials, 0, sizeof(double) * 5 * kPartialCap * kNP, p->stream));
    HIPC(hipMalloc(&p->dstate, sizeof(DevState) * 2));
    // (host-coherent, device-visible: the one-launch solve's commit kernel writes its report straight into them)
    HIPC(hipHostMalloc(&p->hstate, sizeof(DevState), hipHostMallocCoherent | hipHostMallocMapped));
    HIPC(hipMalloc(&p->pctrl, sizeof(PersistCtrl)));
    HIPC(hipMalloc(&p->pgran, sizeof(unsigned long long) * kGranWords));
    HIPC(hipMemsetAsync(p->pgran, 0, sizeof(unsigned long long) * kGranWords, p->stream));
    HIPC(hipHostMalloc(&p->hctrl, sizeof(PersistCtrl), hipHostMallocCoherent | hipHostMallocMapped));
    CHK(tune_persist(p));
    HIPC(hipHostMalloc(&p->hflag, 64, hipHostMallocCoherent | hipHostMallocMapped));
    *p->hflag = 0ull;
    HIPC(hipStreamSynchronize(p->stream));
    return DPGO_OK;
  }();
  if (rc != DPGO_OK) {
    dpgo_problem_destroy(p);
    return rc;
  }
  *out = p;
  return DPGO_OK;
}


// ---- the library's switches and what a handle currently runs (plain text, for logs and tests) ----
static int copy_text(const std::string& s, char* out, int capacity) {
  if (!out || capacity <= 0) return fail(DPGO_ERR_INVALID, "null / empty output buffer");
  const size_t k = std::min(s.size(), (size_t)capacity - 1);
  std::memcpy(out, s.data(), k);
  out[k] = 0;
  return DPGO_OK;
}
int dpgo_describe_options(char* out, int capacity) { return copy_text(options_describe(), out, capacity); }
int dpgo_options_reload(void) {
  (void)options();
  options_read(options_storage());
  return DPGO_OK;
}
int dpgo_problem_describe(dpgo_problem_t p, char* out, int capacity) {
  if (!p) return fail(DPGO_ERR_INVALID, "null handle");
  std::string s = "dpgo_hip problem: d=" + std::to_string(p->d) + " r=" + std::to_string(p->r) + " n=" + std::to_string(p->n) +
                  " nnzb=" + std::to_string(p->Q.nnzb) + " device=" + std::to_string(p->device) + "\n";
  s += "  lane groups per pose " + std::to_string(p->split) + "; launch caps update/hess/hess_sym/restrict/post " +
       std::to_string(p->cap_u) + "/" + std::to_string(p->cap_h) + "/" + std::to_string(p->cap_hs) + "/" +
       std::to_string(p->cap_restrict) + "/" + std::to_string(p->cap_post) + "\n";
  s += std::string("  tCG-step storage: ") + (p->tcg_sym ? "symmetric" : "plain") + (p->stream_nt ? ", non-temporal single-use operands" : "") +
       "; coupling " + ((p->has_G || p->C.nnzb > 0) ? "yes" : "no") + "\n";
  s += std::string("  one-launch solve: ") + ((p->persist && !p->persist_failed_once) ? "on" : "off") + ", last geometry " +
       std::to_string(p->persist_wgs) + " workgroups x " + std::to_string(p->persist_split) + " lane groups x " +
       std::to_string(p->persist_mt) + " tiles" + (p->persist_add ? " (additive)" : "") + "\n";
  s += std::string("  auto: ") + (p->auto_decided ? (p->auto_ml ? "multilevel" : "block-Jacobi") : "undecided") + ", cost-rule state " +
       std::to_string(p->auto_cost.state) + ", block-Jacobi units " + std::to_string(p->auto_cost.jac_units) + ", switches " +
       std::to_string(p->auto_cost.switches) + ", hand-backs " + std::to_string(p->auto_cost.backoff) + "\n";
  s += "  hierarchy:";
  if (!p->ml_symbolic) s += " none";
  for (size_t l = 0; l < p->ml.size(); ++l)
    s += " [" + std::to_string(p->ml[l].n) + " nodes" + (p->ml[l].k ? ", k=" + std::to_string(p->ml[l].graph ? -p->ml[l].k : p->ml[l].k) : ", dense") + "]";
  s += std::string(p->ml_ready ? " built" : " not built") + ", dense level fp" + std::to_string(p->coarse32_active() ? 32 : 64) +
       (p->ml_additive_layout ? ", additive layout" : "") + ", level-0 operator copies of the cycle fp" +
       std::to_string(p->ml_operator_bits) + (p->ml_ops32_active() ? " (in use)" : "") + "\n";
  s += std::string("  iteration graphs: ") + (p->iter_graph_failed ? "unavailable" : (p->iter_graph[0].exec || p->iter_graph[1].exec ? "captured" : "none yet")) + "\n";
  s += "options:\n" + options_describe();
  return copy_text(s, out, capacity);
}

int dpgo_problem_destroy(dpgo_problem_t p) {
  if (!p) return DPGO_OK;
  (void)hipSetDevice(p->device);
  if (p->own_stream) (void)hipStreamSynchronize(p->own_stream);
  free_bsr(p->Q);
  free_bsr(p->C);
  ml_free(p);
  sym_free(p);
  free_edges(p);
  double* vecs[] = {p->x1, p->x2, p->g1, p->g2, p->eta, p->delta, p->Hd, p->rr, p->z, p->G, p->G0,
                    p->S1, p->S2, p->dinv, p->partials};
  for (auto v : vecs)
    if (v) (void)hipFree(v);
  if (p->dstate) (void)hipFree(p->dstate);
  if (p->hstate) (void)hipHostFree(p->hstate);
  if (p->hflag) (void)hipHostFree(p->hflag);
  if (p->pctrl) (void)hipFree(p->pctrl);
  if (p->pgran) (void)hipFree(p->pgran);
  if (p->hctrl) (void)hipHostFree(p->hctrl);
  for (auto& g : p->iter_graph)
    if (g.exec) (void)hipGraphExecDestroy(g.exec);
  if (p->own_stream) (void)hipStreamDestroy(p->own_stream);
  delete p;
  return DPGO_OK;
}


int dpgo_problem_set_stream(dpgo_problem_t p, void* hip_stream) {
  if (!p) return fail(DPGO_ERR_INVALID, "null handle");
  CHK(set_device(p));
  HIPC(hipStreamSynchronize(p->stream));
  p->stream = (hipStream_t)hip_stream;  // NULL = the default (null) stream
  return DPGO_OK;
}


int dpgo_problem_use_own_stream(dpgo_problem_t p) {
  if (!p) return fail(DPGO_ERR_INVALID, "null handle");
  CHK(set_device(p));
  HIPC(hipStreamSynchronize(p->stream));
  p->stream = p->own_stream;
  return DPGO_OK;
}


int dpgo_problem_dims(dpgo_problem_t p, int* r, int* d, int* n, int* nnzb) {
  if (!p) return fail(DPGO_ERR_INVALID, "null handle");
  if (r) *r = p->r;
  if (d) *d = p->d;
  if (n) *n = p->n;
  if (nnzb) *nnzb = p->Q.nnzb;
  return DPGO_OK;
}


int dpgo_problem_set_Q_bsr(dpgo_problem_t p, int nnzb, const int32_t* rowptr, const int32_t* colidx,
                           const double* vals) {
  if (!p) return fail(DPGO_ERR_INVALID, "null handle");
  if (!vals) return fail(DPGO_ERR_INVALID, "null vals");
  CHK(validate_bsr(p->n, p->n, nnzb, rowptr, colidx, true));
  CHK(set_device(p));
  // registered re-weightable edges index into the old pattern: drop them (the caller re-registers)
  if (p->e_w) CHK(free_edges(p));
  CHK(upload_bsr(p->Q, p->n, p->n, nnzb, p->b, rowptr, colidx, vals, p->stream));
  const bool same_pattern = (int)p->h_rowptr.size() == p->n + 1 && (int)p->h_colidx.size() == nnzb &&
                            std::equal(rowptr, rowptr + p->n + 1, p->h_rowptr.begin()) &&
                            std::equal(colidx, colidx + nnzb, p->h_colidx.begin());
  if (!same_pattern) {
    p->h_rowptr.assign(rowptr, rowptr + p->n + 1);
    p->h_colidx.assign(colidx, colidx + nnzb);
    p->add_plan_known = false;
    if (p->ml_user_ks && p->ml_symbolic) {  // keep the caller's aggregate sizes across a pattern change
      const int perm_tile = p->ml[0].perm_tile;
      CHK(ml_symbolic_setup(p, ml_current_ks(p), perm_tile));
    } else {
      ml_free(p);
    }
    sym_free(p);
  }
  p->sym.ready = p->tcg_sym = false;
  p->ml_ready = false;  // the hierarchy's values belong to the old Q: rebuilt on the device at the next use
  p->auto_decided = false;
  p->dinv_shift = -1.0;
  CHK(build_dinv(p, 1e-1));  // src/PoseGraph.cpp:603
  HIPC(hipStreamSynchronize(p->stream));
  return DPGO_OK;
}


int dpgo_problem_set_Q_csr(dpgo_problem_t p, const int32_t* outer, const int32_t* inner, const double* values) {
  if (!p) return fail(DPGO_ERR_INVALID, "null handle");
  if (!outer || !inner || !values) return fail(DPGO_ERR_INVALID, "null CSR arrays");
  const int b = p->b, n = p->n;
  std::vector<int32_t> rowptr(n + 1, 0), colidx;
  std::vector<double> vals;
  std::vector<int> slot(n, -1);  // block column -> position in the current block row
  for (int i = 0; i < n; ++i) {
    const int first = (int)colidx.size();
    std::vector<int> cols;
    for (int rr = 0; rr < b; ++rr) {
      const int row = i * b + rr;
      if (outer[row + 1] < outer[row]) return fail(DPGO_ERR_INVALID, "CSR outer index not monotone");
      for (int t = outer[row]; t < outer[row + 1]; ++t) {
        const int c = inner[t];
        if (c < 0 || c >= n * b) return fail(DPGO_ERR_INVALID, "CSR column index out of range");
        const int j = c / b;
        if (slot[j] < 0) {
          slot[j] = 1;
          cols.push_back(j);
        }
      }
    }
    std::sort(cols.begin(), cols.end());
    for (size_t k = 0; k < cols.size(); ++k) slot[cols[k]] = first + (int)k;
    colidx.insert(colidx.end(), cols.begin(), cols.end());
    vals.resize(colidx.size() * (size_t)b * b, 0.0);
    for (int rr = 0; rr < b; ++rr) {
      const int row = i * b + rr;
      for (int t = outer[row]; t < outer[row + 1]; ++t) {
        const int c = inner[t];
        vals[(size_t)slot[c / b] * b * b + rr * b + (c % b)] += values[t];
      }
    }
    for (int j : cols) slot[j] = -1;
    rowptr[i + 1] = (int32_t)colidx.size();
  }
  return dpgo_problem_set_Q_bsr(p, (int)colidx.size(), rowptr.data(), colidx.data(), vals.data());
}


int dpgo_problem_update_Q_values(dpgo_problem_t p, const double* vals) {
  CHK(check_ready(p));
  if (!vals) return fail(DPGO_ERR_INVALID, "null vals");
  HIPC(hipMemcpyAsync(p->Q.vals, vals, sizeof(double) * (size_t)p->Q.nnzb * p->b * p->b, hipMemcpyHostToDevice,
                      p->stream));
  // registered re-weightable edges: the constant part of Q is whatever the new values hold beyond the listed
  // edges' contributions at the current weights
  if (p->e_w) CHK(rebuild_Q_from_weights(p, p->Q.vals, -1.0, p->q_base));
  const double s = p->dinv_shift > 0 ? p->dinv_shift : 1e-1;
  p->dinv_shift = -1.0;  // PoseGraph::clearQuadraticMatrix also drops the preconditioner (src/PoseGraph.cpp:352-355)
  p->ml_ready = false;
  p->auto_decided = false;
  p->sym.ready = p->tcg_sym = false;
  CHK(build_dinv(p, s));
  HIPC(hipStreamSynchronize(p->stream));
  return DPGO_OK;
}


int dpgo_problem_get_Q_values(dpgo_problem_t p, double* vals_host) {
  CHK(check_ready(p));
  if (!vals_host) return fail(DPGO_ERR_INVALID, "null vals");
  HIPC(hipMemcpyAsync(vals_host, p->Q.vals, sizeof(double) * (size_t)p->Q.nnzb * p->b * p->b, hipMemcpyDeviceToHost,
                      p->stream));
  HIPC(hipStreamSynchronize(p->stream));
  return DPGO_OK;
}


int dpgo_problem_set_G(dpgo_problem_t p, const double* G_host) {
  if (!p) return fail(DPGO_ERR_INVALID, "null handle");
  CHK(set_device(p));
  if (!G_host) {
    p->has_G = false;
    return DPGO_OK;
  }
  CHK(h2d(p, p->G, G_host));
  HIPC(hipStreamSynchronize(p->stream));
  p->has_G = true;
  return DPGO_OK;
}


int dpgo_problem_set_G_device(dpgo_problem_t p, const double* G_dev) {
  if (!p) return fail(DPGO_ERR_INVALID, "null handle");
  CHK(set_device(p));
  if (!G_dev) {
    p->has_G = false;
    return DPGO_OK;
  }
  HIPC(hipMemcpyAsync(p->G, G_dev, p->vec_bytes(), hipMemcpyDeviceToDevice, p->stream));
  p->has_G = true;
  return DPGO_OK;
}


int dpgo_problem_set_G_coupling(dpgo_problem_t p, int ncols, int nnzb, const int32_t* rowptr, const int32_t* colidx,
                                const double* vals, const double* G0_host) {
  if (!p) return fail(DPGO_ERR_INVALID, "null handle");
  if (ncols < 0) return fail(DPGO_ERR_INVALID, "ncols < 0");
  if (nnzb > 0 && !vals) return fail(DPGO_ERR_INVALID, "null vals");
  CHK(validate_bsr(p->n, ncols, nnzb, rowptr, colidx, false));
  CHK(set_device(p));
  // shared re-weightable edges index into the old coupling pattern: drop them (the caller re-registers)
  if (p->n_shared_edges > 0) CHK(free_edges(p));
  CHK(upload_bsr(p->C, p->n, ncols, nnzb, p->b, rowptr, colidx, vals, p->stream));
  if (G0_host) {
    CHK(h2d(p, p->G0, G0_host));
  } else {
    HIPC(hipMemsetAsync(p->G0, 0, p->vec_bytes(), p->stream));
  }
  HIPC(hipStreamSynchronize(p->stream));
  return DPGO_OK;
}


int dpgo_problem_update_G_from_neighbors_device(dpgo_problem_t p, const double* nbr_tiles_dev) {
  if (!p) return fail(DPGO_ERR_INVALID, "null handle");
  if (!p->C.rowptr) return fail(DPGO_ERR_STATE, "G coupling not set");
  if (!nbr_tiles_dev && p->C.nnzb > 0) return fail(DPGO_ERR_INVALID, "null neighbour tiles");
  CHK(set_device(p));
  CHK(launch_spmm(p, p->C, nbr_tiles_dev, p->G0, p->G));
  p->has_G = true;
  return DPGO_OK;
}


// ---- QuadraticProblem methods (host pointers) ----
static int eval_common(dpgo_problem_t p, const double* X) {
  CHK(check_ready(p));
  if (!X) return fail(DPGO_ERR_INVALID, "null X");
  CHK(h2d(p, p->x2, X));
  return DPGO_OK;
}


int dpgo_problem_f(dpgo_problem_t p, const double* X, double* f) {
  if (!f) return fail(DPGO_ERR_INVALID, "null out");
  CHK(eval_common(p, X));
  CHK(launch_grad(p, p->x2, nullptr, nullptr, nullptr));
  CHK(launch_rtr_begin(p, 0.0, 1.0, 1.0, 0, 0));
  CHK(poll_state(p));
  *f = p->hstate->f1;
  return DPGO_OK;
}


int dpgo_problem_euc_grad(dpgo_problem_t p, const double* X, double* EG) {
  if (!EG) return fail(DPGO_ERR_INVALID, "null out");
  CHK(eval_common(p, X));
  CHK(launch_spmm(p, p->Q, p->x2, p->has_G ? p->G : nullptr, p->g2));
  return d2h(p, EG, p->g2);
}


int dpgo_problem_euc_hess(dpgo_problem_t p, const double* V, double* HV) {
  if (!HV) return fail(DPGO_ERR_INVALID, "null out");
  CHK(eval_common(p, V));
  CHK(launch_spmm(p, p->Q, p->x2, nullptr, p->g2));
  return d2h(p, HV, p->g2);
}


int dpgo_problem_rie_grad(dpgo_problem_t p, const double* X, double* RG) {
  if (!RG) return fail(DPGO_ERR_INVALID, "null out");
  CHK(eval_common(p, X));
  CHK(launch_grad(p, p->x2, p->g2, nullptr, nullptr));
  return d2h(p, RG, p->g2);
}


int dpgo_problem_rie_grad_norm(dpgo_problem_t p, const double* X, double* gn) {
  if (!gn) return fail(DPGO_ERR_INVALID, "null out");
  CHK(eval_common(p, X));
  CHK(launch_grad(p, p->x2, nullptr, nullptr, nullptr));
  CHK(launch_rtr_begin(p, 0.0, 1.0, 1.0, 0, 0));
  CHK(poll_state(p));
  *gn = p->hstate->ngf;
  return DPGO_OK;
}


int dpgo_problem_rie_hess(dpgo_problem_t p, const double* X, const double* V, double* HV) {
  if (!HV || !V) return fail(DPGO_ERR_INVALID, "null pointer");
  CHK(eval_common(p, X));
  CHK(h2d(p, p->eta, V));
  CHK(launch_grad(p, p->x2, nullptr, p->S2, nullptr));
  CHK(launch_hess(p, p->x2, p->S2, p->eta, nullptr, p->g2, p->pH(), nullptr, 0));
  return d2h(p, HV, p->g2);
}


int dpgo_problem_precondition(dpgo_problem_t p, int precond, double shift, const double* X, const double* V,
                              double* Z) {
  if (!Z || !V) return fail(DPGO_ERR_INVALID, "null pointer");
  CHK(eval_common(p, X));
  CHK(h2d(p, p->eta, V));
  const double* dinv = nullptr;
  if (precond == DPGO_PRECOND_AUTO) {
    p->auto_decide();
    precond = p->auto_ml ? DPGO_PRECOND_MULTILEVEL : DPGO_PRECOND_BLOCK_JACOBI;
  }
  if (precond == DPGO_PRECOND_BLOCK_JACOBI) {
    CHK(build_dinv(p, shift));
    dinv = p->dinv;
  } else if (precond == DPGO_PRECOND_MULTILEVEL) {
    CHK(ml_ensure(p, shift));
    CHK(ml_ops32_ensure(p));
    CHK(launch_ml_apply(p, p->x2, p->eta, p->g2));
    return d2h(p, Z, p->g2);
  } else if (precond == DPGO_PRECOND_ADDITIVE) {
    return fail(DPGO_ERR_UNSUPPORTED, "the additive preconditioner exists inside the persistent tCG kernel only");
  } else if (precond != DPGO_PRECOND_NONE) {
    return fail(DPGO_ERR_INVALID, "unknown preconditioner");
  }
  CHK(launch_precond(p, p->x2, p->eta, dinv, p->g2));
  return d2h(p, Z, p->g2);
}


int dpgo_spmm_device(dpgo_problem_t p, const double* V_dev, double* OUT_dev, int add_G) {
  CHK(check_ready(p));
  if (!V_dev || !OUT_dev) return fail(DPGO_ERR_INVALID, "null pointer");
  return launch_spmm(p, p->Q, V_dev, (add_G && p->has_G) ? p->G : nullptr, OUT_dev);
}


int dpgo_problem_eval_device(dpgo_problem_t p, const double* X_dev, double* f, double* gradnorm) {
  CHK(check_ready(p));
  if (!X_dev) return fail(DPGO_ERR_INVALID, "null pointer");
  CHK(launch_grad(p, X_dev, nullptr, nullptr, nullptr));
  CHK(launch_rtr_begin(p, 0.0, 1.0, 1.0, 0, 0));
  CHK(poll_state(p));
  if (f) *f = p->hstate->f1;
  if (gradnorm) *gradnorm = p->hstate->ngf;
  return DPGO_OK;
}


int dpgo_problem_eval_terms_device(dpgo_problem_t p, const double* X_dev, double* xqx, double* xg, double* g2) {
  CHK(check_ready(p));
  if (!X_dev) return fail(DPGO_ERR_INVALID, "null pointer");
  CHK(launch_grad(p, X_dev, nullptr, nullptr, nullptr));
  CHK(launch_rtr_begin(p, 0.0, 1.0, 1.0, 0, 0));
  CHK(poll_state(p));
  if (xqx) *xqx = p->hstate->xqx;
  if (xg) *xg = p->hstate->xg;
  if (g2) *g2 = p->hstate->ngf * p->hstate->ngf;
  return DPGO_OK;
}


int dpgo_problem_set_spmm_variant(dpgo_problem_t p, int variant, int* in_use) {
  CHK(check_ready(p));
  if (variant != DPGO_SPMM_AUTO && variant != DPGO_SPMM_PLAIN && variant != DPGO_SPMM_SYMMETRIC)
    return fail(DPGO_ERR_INVALID, "unknown product storage");
  p->spmm_variant = variant;
  bool usable = false;
  if (p->sym_wanted()) CHK(sym_ensure(p, &usable));
  if (in_use) *in_use = usable ? DPGO_SPMM_SYMMETRIC : DPGO_SPMM_PLAIN;
  return DPGO_OK;
}


int dpgo_device_malloc(void** out, size_t bytes, int device) {
  if (!out) return fail(DPGO_ERR_INVALID, "null out");
  *out = nullptr;
  int cnt = 0;
  CHK(dpgo_device_count(&cnt));
  if (cnt <= 0) return fail(DPGO_ERR_HIP, "no HIP device (this library has no CPU fallback)");
  if (device < 0 || device >= cnt) return fail(DPGO_ERR_INVALID, "device index out of range");
  HIPC(hipSetDevice(device));
  HIPC(hipMalloc(out, bytes > 0 ? bytes : 1));
  return DPGO_OK;
}


int dpgo_device_free(void* p) {
  if (p) HIPC(hipFree(p));
  return DPGO_OK;
}


int dpgo_device_memcpy(void* dst, const void* src, size_t bytes, int kind, void* stream) {
  if (bytes == 0) return DPGO_OK;
  if (!dst || !src) return fail(DPGO_ERR_INVALID, "null pointer");
  hipMemcpyKind k;
  switch (kind) {
    case DPGO_COPY_H2D: k = hipMemcpyHostToDevice; break;
    case DPGO_COPY_D2H: k = hipMemcpyDeviceToHost; break;
    case DPGO_COPY_D2D: k = hipMemcpyDeviceToDevice; break;
    default: return fail(DPGO_ERR_INVALID, "unknown copy kind");
  }
  HIPC(hipMemcpyAsync(dst, src, bytes, k, (hipStream_t)stream));
  if (kind == DPGO_COPY_D2H) HIPC(hipStreamSynchronize((hipStream_t)stream));
  return DPGO_OK;
}


int dpgo_device_synchronize(void* stream) {
  HIPC(hipStreamSynchronize((hipStream_t)stream));
  return DPGO_OK;
}

}  // extern "C"
