/* dpgo_oracle_c.c -- plain-C restatement of the per-agent local solve of mit-acl/dpgo.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * call this file (through oracle/c_oracle.py).  It exists for two reasons:
 *   1. a second, independently written restatement of the algorithm (the first is oracle/dpgo_oracle.py) -- the two
 *      are checked against each other in tests/test_oracle.py, which pins the oracle's internal consistency;
 *   2. a fair single-core CPU baseline for bench.py (`gcc -O3 -march=x86-64-v3`, one thread like the reference:
 *      ENABLE_OPENMP is OFF in the reference's CMakeLists.txt:55), instead of timing NumPy overheads.
 * Parity status: "parity unpinned" at trajectory level against the reference binary (Eigen / ROPTLIB / SuiteSparse
 * are absent from this image); pinned by the reference's known-answer tests through the
 * Python oracle and by agreement with it.
 *
 * What is restated (reference file:line):
 *   f, EucGrad, EucHessianEta, RieGrad, PreConditioner     src/QuadraticProblem.cpp:29-83
 *   optimize / trustRegion / gradientDescent               src/QuadraticOptimizer.cpp:26-137
 *   ROPTLIB RTRNewton + tCG_TR, Stiefel (params set 3) x Euclidean  -- third party, restated from the published
 *   algorithm (SURVEY.md section 8 a8 / a9 / c'): Euclidean metric, qf retraction, projection W - Y sym(Y^T W),
 *   Riemannian Hessian proj(VQ - V_rot sym(Y^T EG)), accept rho > 0.1, shrink x0.25 below 0.25, grow x2 above 0.75
 *   on boundary / negative curvature, Delta_max = 5 Delta_0, tCG stop |r| <= |r0| min(|r0|^theta, kappa).
 *   Preconditioner: block-Jacobi (Q_ii + shift I)^-1 as on the device (the reference's exact CHOLMOD solve is not
 *   restated here; the Python oracle has it).
 *
 * Layout: X is the reference's r x (d+1)n column-major matrix = n pose tiles of (d+1) x r doubles with r contiguous
 * ("tiles [n][d+1][r]"); Q is block-CSR with (d+1)x(d+1) row-major blocks (scipy bsr_matrix), int32 indices.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MAXB 4
#define MAXR 8

typedef struct {
  int n, d, r, b, T;
  const int32_t *rowptr, *colidx;
  const double* vals;
  const double* G; /* may be NULL */
  double* dinv;    /* n * b * b, or NULL (no preconditioner) */
} Problem;

typedef struct {
  int method;            /* 0 RTR, 1 RGD */
  double gradnorm_tol, RGD_stepsize;
  int RGD_use_preconditioner, RTR_iterations, RTR_tCG_iterations;
  double RTR_initial_radius;
  int precond;           /* 0 none, 1 block-Jacobi */
  double precond_shift;
  int accept_tiny_decrease;
  int hess_recurrence;   /* bit 0 -- 0: H applied to delta (reference arithmetic); 1: H delta' = beta H delta - H z.
                          * bit 1 -- every full-vector sum is formed tile-wise (partial sums over 64 poses, then the partials
                          * in order) like the device's per-workgroup partials, instead of one running sum: a second
                          * summation ORDER of the same arithmetic, used to measure how far two orders drift apart. */
} CParams;

typedef struct {
  int success;
  double fInit, gradNormInit, fOpt, gradNormOpt;
  int tCGStatus, tcg_iterations, rtr_iterations, n_spmm;
} CResult;

enum { TCG_NEGCURV = 0, TCG_EXCREGION = 1, TCG_LCON = 2, TCG_SCON = 3, TCG_MAXITER = 4 };

static int g_spmm = 0;
static int g_tiled_sums = 0;
#define SUM_TILE 64 /* poses per partial sum in tile-wise mode */

/* OUT = V Q (+ G): tile i of OUT, row c: sum_t sum_k Q_t[c][k] * V_j[k][:]   (Q symmetric: (VQ)^T = Q V^T).
 * The body is instantiated with compile-time block / rank sizes for the shapes the benchmarks use, so that the
 * compiler can unroll and vectorise it (a fair single-core baseline), plus a generic fallback. */
#define SPMM_BODY(B_, R_)                                                        \
  do {                                                                           \
    const int T_ = (B_) * (R_);                                                  \
    for (int i = 0; i < p->n; ++i) {                                             \
      double acc[MAXB * MAXR];                                                   \
      for (int e = 0; e < T_; ++e) acc[e] = add ? add[(size_t)i * T_ + e] : 0.0; \
      for (int t = p->rowptr[i]; t < p->rowptr[i + 1]; ++t) {                    \
        const double* restrict q = p->vals + (size_t)t * (B_) * (B_);            \
        const double* restrict x = V + (size_t)p->colidx[t] * T_;                \
        for (int c = 0; c < (B_); ++c)                                           \
          for (int k = 0; k < (B_); ++k) {                                       \
            const double qv = q[c * (B_) + k];                                   \
            for (int a = 0; a < (R_); ++a) acc[c * (R_) + a] += qv * x[k * (R_) + a]; \
          }                                                                      \
      }                                                                          \
      memcpy(OUT + (size_t)i * T_, acc, sizeof(double) * T_);                    \
    }                                                                            \
  } while (0)

static void spmm(const Problem* p, const double* V, const double* add, double* OUT) {
  const int b = p->b, r = p->r;
  g_spmm++;
  if (b == 4 && r == 5) SPMM_BODY(4, 5);
  else if (b == 4 && r == 3) SPMM_BODY(4, 3);
  else if (b == 4 && r == 4) SPMM_BODY(4, 4);
  else if (b == 4 && r == 6) SPMM_BODY(4, 6);
  else if (b == 3 && r == 2) SPMM_BODY(3, 2);
  else if (b == 3 && r == 3) SPMM_BODY(3, 3);
  else SPMM_BODY(b, r);
}

static double dot(const Problem* p, const double* a, const double* b) {
  double s = 0.0;
  const size_t N = (size_t)p->n * p->T;
  if (g_tiled_sums) {
    const size_t chunk = (size_t)SUM_TILE * p->T;
    for (size_t k0 = 0; k0 < N; k0 += chunk) {
      double part = 0.0;
      const size_t k1 = k0 + chunk < N ? k0 + chunk : N;
      for (size_t k = k0; k < k1; ++k) part += a[k] * b[k];
      s += part;
    }
    return s;
  }
  for (size_t k = 0; k < N; ++k) s += a[k] * b[k];
  return s;
}

/* S_i = sym(Y_i^T W_i) (d x d, row-major), Y = rotation rows of the tile */
static void sym_ytw(const Problem* p, const double* X, const double* W, double* S) {
  const int d = p->d, r = p->r, T = p->T;
  for (int i = 0; i < p->n; ++i) {
    const double *y = X + (size_t)i * T, *w = W + (size_t)i * T;
    for (int a = 0; a < d; ++a)
      for (int c = 0; c < d; ++c) {
        double s1 = 0.0, s2 = 0.0;
        for (int k = 0; k < r; ++k) {
          s1 += y[a * r + k] * w[c * r + k];
          s2 += w[a * r + k] * y[c * r + k];
        }
        S[(size_t)i * d * d + a * d + c] = 0.5 * (s1 + s2);
      }
  }
}

/* ROPTLIB Stiefel::ExtrProjection per pose: W_rot - Y sym(Y^T W_rot); translation row untouched.  In place OK. */
static void tangent_project(const Problem* p, const double* X, const double* W, double* OUT) {
  const int d = p->d, r = p->r, T = p->T;
  for (int i = 0; i < p->n; ++i) {
    const double *y = X + (size_t)i * T, *w = W + (size_t)i * T;
    double S[MAXB * MAXB], o[MAXB * MAXR];
    for (int a = 0; a < d; ++a)
      for (int c = 0; c < d; ++c) {
        double s1 = 0.0, s2 = 0.0;
        for (int k = 0; k < r; ++k) {
          s1 += y[a * r + k] * w[c * r + k];
          s2 += w[a * r + k] * y[c * r + k];
        }
        S[a * d + c] = 0.5 * (s1 + s2);
      }
    for (int c = 0; c < d; ++c)
      for (int k = 0; k < r; ++k) {
        double v = w[c * r + k];
        for (int a = 0; a < d; ++a) v -= y[a * r + k] * S[a * d + c];
        o[c * r + k] = v;
      }
    for (int k = 0; k < r; ++k) o[d * r + k] = w[d * r + k];
    memcpy(OUT + (size_t)i * T, o, sizeof(double) * T);
  }
}

/* Riemannian Hessian: proj_X( V Q - V_rot S ) with S = sym(Y^T EG_rot) cached at the iterate */
static void rie_hess(const Problem* p, const double* X, const double* S, const double* V, double* OUT) {
  const int d = p->d, r = p->r, T = p->T;
  spmm(p, V, NULL, OUT);
  for (int i = 0; i < p->n; ++i) {
    double* h = OUT + (size_t)i * T;
    const double* v = V + (size_t)i * T;
    const double* s = S + (size_t)i * d * d;
    for (int c = 0; c < d; ++c)
      for (int a = 0; a < d; ++a)
        for (int k = 0; k < r; ++k) h[c * r + k] -= v[a * r + k] * s[a * d + c];
  }
  tangent_project(p, X, OUT, OUT);
}

/* QuadraticProblem::PreConditioner with the block-Jacobi factor: Z_i = Dinv_i V_i, then tangent projection */
static void precondition(const Problem* p, const double* X, const double* V, double* Z) {
  const int b = p->b, r = p->r, T = p->T;
  if (p->dinv) {
    for (int i = 0; i < p->n; ++i) {
      const double* v = V + (size_t)i * T;
      const double* D = p->dinv + (size_t)i * b * b;
      double o[MAXB * MAXR];
      for (int c = 0; c < b; ++c)
        for (int k = 0; k < r; ++k) {
          double s = 0.0;
          for (int a = 0; a < b; ++a) s += D[c * b + a] * v[a * r + k];
          o[c * r + k] = s;
        }
      memcpy(Z + (size_t)i * T, o, sizeof(double) * T);
    }
    tangent_project(p, X, Z, Z);
  } else {
    tangent_project(p, X, V, Z);
  }
}

/* qf retraction: modified Gram-Schmidt of the rotation rows of X + eta (diag(R) > 0); translation p + eta */
static void qf_retract(const Problem* p, const double* X, const double* eta, double* OUT) {
  const int d = p->d, r = p->r, T = p->T;
  for (int i = 0; i < p->n; ++i) {
    double o[MAXB * MAXR];
    for (int e = 0; e < T; ++e) o[e] = X[(size_t)i * T + e] + eta[(size_t)i * T + e];
    for (int k = 0; k < d; ++k) {
      double* v = o + k * r;
      for (int l = 0; l < k; ++l) {
        double s = 0.0;
        for (int a = 0; a < r; ++a) s += o[l * r + a] * v[a];
        for (int a = 0; a < r; ++a) v[a] -= s * o[l * r + a];
      }
      double nrm = 0.0;
      for (int a = 0; a < r; ++a) nrm += v[a] * v[a];
      nrm = sqrt(nrm);
      for (int a = 0; a < r; ++a) v[a] /= nrm;
    }
    memcpy(OUT + (size_t)i * T, o, sizeof(double) * T);
  }
}

static double cost(const Problem* p, const double* X, double* work) {
  const size_t N = (size_t)p->n * p->T;
  spmm(p, X, NULL, work);
  double s = 0.0, g = 0.0;
  if (g_tiled_sums) {
    s = dot(p, work, X);
    if (p->G) g = dot(p, X, p->G);
    return 0.5 * s + g;
  }
  for (size_t k = 0; k < N; ++k) s += work[k] * X[k];
  if (p->G)
    for (size_t k = 0; k < N; ++k) g += X[k] * p->G[k];
  return 0.5 * s + g;
}

/* inverse of a small SPD matrix (Gauss-Jordan with partial pivoting) */
static int inv_small(int b, const double* A, double* Ainv) {
  double M[MAXB][2 * MAXB];
  for (int i = 0; i < b; ++i)
    for (int j = 0; j < b; ++j) {
      M[i][j] = A[i * b + j];
      M[i][b + j] = (i == j) ? 1.0 : 0.0;
    }
  for (int c = 0; c < b; ++c) {
    int piv = c;
    for (int i = c + 1; i < b; ++i)
      if (fabs(M[i][c]) > fabs(M[piv][c])) piv = i;
    if (M[piv][c] == 0.0) return 1;
    if (piv != c)
      for (int j = 0; j < 2 * b; ++j) {
        double tmp = M[c][j];
        M[c][j] = M[piv][j];
        M[piv][j] = tmp;
      }
    const double inv = 1.0 / M[c][c];
    for (int j = 0; j < 2 * b; ++j) M[c][j] *= inv;
    for (int i = 0; i < b; ++i)
      if (i != c) {
        const double f = M[i][c];
        for (int j = 0; j < 2 * b; ++j) M[i][j] -= f * M[c][j];
      }
  }
  for (int i = 0; i < b; ++i)
    for (int j = 0; j < b; ++j) Ainv[i * b + j] = M[i][b + j];
  return 0;
}

static int build_dinv(Problem* p, double shift) {
  const int b = p->b;
  p->dinv = (double*)malloc(sizeof(double) * (size_t)p->n * b * b);
  if (!p->dinv) return 1;
  for (int i = 0; i < p->n; ++i) {
    double D[MAXB * MAXB];
    int found = 0;
    for (int t = p->rowptr[i]; t < p->rowptr[i + 1]; ++t)
      if (p->colidx[t] == i) {
        memcpy(D, p->vals + (size_t)t * b * b, sizeof(double) * b * b);
        found = 1;
      }
    if (!found) memset(D, 0, sizeof(D));
    for (int k = 0; k < b; ++k) D[k * b + k] += shift;
    if (inv_small(b, D, p->dinv + (size_t)i * b * b)) return 1;
  }
  return 0;
}

/* ROPTLIB SolversTR::tCG_TR, eta0 = 0, theta = 1, kappa = 0.1, Min_Inner_Iter = 0 */
static int tcg(const Problem* p, const double* X, const double* g, const double* S, double Delta, int max_inner,
               int hess_recurrence, double* eta, double** wk, int* n_hess_out, int* inner_out) {
  const size_t N = (size_t)p->n * p->T;
  double *r = wk[0], *z = wk[1], *delta = wk[2], *Hd = wk[3], *Hz = wk[4];
  const double theta = 1.0, kappa = 0.1;
  memcpy(r, g, sizeof(double) * N);
  memset(eta, 0, sizeof(double) * N);
  double e_Pe = 0.0, r_r = dot(p, r, r);
  const double norm_r0 = sqrt(r_r);
  precondition(p, X, r, z);
  double z_r = dot(p, z, r), d_Pd = z_r, e_Pd = 0.0, beta = 0.0;
  for (size_t k = 0; k < N; ++k) delta[k] = -z[k];
  int status = TCG_MAXITER, j = 0, n_hess = 0;
  while (j < max_inner) {
    if (hess_recurrence) {
      rie_hess(p, X, S, z, Hz);
      if (j == 0)
        for (size_t k = 0; k < N; ++k) Hd[k] = -Hz[k];
      else
        for (size_t k = 0; k < N; ++k) Hd[k] = beta * Hd[k] - Hz[k];
    } else {
      rie_hess(p, X, S, delta, Hd);
    }
    n_hess++;
    const double d_Hd = dot(p, delta, Hd);
    const double alpha = (d_Hd != 0.0) ? z_r / d_Hd : INFINITY;
    const double e_Pe_new = e_Pe + 2.0 * alpha * e_Pd + alpha * alpha * d_Pd;
    if (d_Hd <= 0.0 || e_Pe_new >= Delta * Delta) {
      const double tau = (-e_Pd + sqrt(e_Pd * e_Pd + d_Pd * (Delta * Delta - e_Pe))) / d_Pd;
      for (size_t k = 0; k < N; ++k) eta[k] += tau * delta[k];
      status = (d_Hd < 0.0) ? TCG_NEGCURV : TCG_EXCREGION;
      break;
    }
    e_Pe = e_Pe_new;
    for (size_t k = 0; k < N; ++k) {
      eta[k] += alpha * delta[k];
      r[k] += alpha * Hd[k];
    }
    r_r = dot(p, r, r);
    const double norm_r = sqrt(r_r);
    const double pw = pow(norm_r0, theta);
    if (norm_r <= norm_r0 * (pw < kappa ? pw : kappa)) {
      status = (kappa < pw) ? TCG_LCON : TCG_SCON;
      break;
    }
    precondition(p, X, r, z);
    const double zold_rold = z_r;
    z_r = dot(p, z, r);
    beta = z_r / zold_rold;
    for (size_t k = 0; k < N; ++k) delta[k] = beta * delta[k] - z[k];
    e_Pd = beta * (e_Pd + alpha * d_Pd);
    d_Pd = z_r + beta * beta * d_Pd;
    j++;
  }
  *n_hess_out = n_hess;
  *inner_out = j;
  return status;
}

/* f, Riemannian gradient and S at X; returns |rgrad| */
static double grad_at(const Problem* p, const double* X, double* EG, double* S, double* RG, double* f) {
  const size_t N = (size_t)p->n * p->T;
  spmm(p, X, p->G, EG);
  double s = 0.0, g = 0.0;
  if (p->G) {
    for (size_t k = 0; k < N; ++k) {
      s += (EG[k] - p->G[k]) * X[k];
      g += X[k] * p->G[k];
    }
  } else {
    for (size_t k = 0; k < N; ++k) s += EG[k] * X[k];
  }
  *f = 0.5 * s + g;
  sym_ytw(p, X, EG, S);
  tangent_project(p, X, EG, RG);
  return sqrt(dot(p, RG, RG));
}

/* ROPTLIB SolversTR::Run; returns accepted_last */
static int run_rtr(const Problem* p, const CParams* prm, double* x1, double Delta0, double Delta_max, int max_iter,
                   CResult* res, double** wk) {
  const size_t N = (size_t)p->n * p->T;
  double *EG = wk[5], *g1 = wk[6], *eta = wk[7], *x2 = wk[8], *Heta = wk[9], *S = wk[10];
  const double sqeps = sqrt(2.220446049250313e-16);
  double f1;
  double ngf = grad_at(p, x1, EG, S, g1, &f1);
  double Delta = Delta0;
  int it = 0, accepted_last = 0, status = TCG_MAXITER;
  int isstop = ngf < prm->gradnorm_tol;
  while (!isstop && it < max_iter) {
    int n_hess = 0, inner = 0;
    status = tcg(p, x1, g1, S, Delta, prm->RTR_tCG_iterations, prm->hess_recurrence & 1, eta, wk, &n_hess, &inner);
    res->tcg_iterations += n_hess;
    qf_retract(p, x1, eta, x2);
    const double f2 = cost(p, x2, Heta);
    rie_hess(p, x1, S, eta, Heta);
    double den = 0.0;
    for (size_t k = 0; k < N; ++k) den += eta[k] * (g1[k] + 0.5 * Heta[k]);
    const double rho = (f1 - f2) / (-den);
    if (rho > 0.75) {
      if (status == TCG_EXCREGION || status == TCG_NEGCURV) Delta *= 2.0;
      if (Delta > Delta_max) Delta = Delta_max;
    } else if (rho < 0.25) {
      Delta *= 0.25;
    }
    const int accept = (rho > 0.1) ||
                       (prm->accept_tiny_decrease && fabs(f1 - f2) / (fabs(f1) + 1.0) < sqeps && f2 < f1);
    if (accept) {
      memcpy(x1, x2, sizeof(double) * N);
      ngf = grad_at(p, x1, EG, S, g1, &f1);
      isstop = ngf < prm->gradnorm_tol;
    }
    accepted_last = accept;
    it++;
  }
  res->rtr_iterations += it;
  res->tCGStatus = status;
  return accepted_last;
}

/* ------------------------------------------------------------------ exported entry points */

int dpgo_c_spmm(int n, int d, int r, const int32_t* rowptr, const int32_t* colidx, const double* vals, const double* V,
                double* OUT, int reps) {
  Problem p = {n, d, r, d + 1, (d + 1) * r, rowptr, colidx, vals, NULL, NULL};
  if (d + 1 > MAXB || r > MAXR) return 1;
  for (int k = 0; k < (reps > 0 ? reps : 1); ++k) spmm(&p, V, NULL, OUT);
  return 0;
}

/* f, |rgrad|, rgrad (optional), Riemannian Hessian of V (optional), preconditioned V (optional) at X */
int dpgo_c_eval(int n, int d, int r, const int32_t* rowptr, const int32_t* colidx, const double* vals, const double* G,
                const double* X, const double* V, double shift, double* f, double* gradnorm, double* RG, double* HV,
                double* PV) {
  if (d + 1 > MAXB || r > MAXR) return 1;
  Problem p = {n, d, r, d + 1, (d + 1) * r, rowptr, colidx, vals, G, NULL};
  const size_t N = (size_t)n * p.T;
  double* EG = (double*)malloc(sizeof(double) * N);
  double* rg = (double*)malloc(sizeof(double) * N);
  double* S = (double*)malloc(sizeof(double) * (size_t)n * d * d);
  if (!EG || !rg || !S) return 2;
  *gradnorm = grad_at(&p, X, EG, S, rg, f);
  if (RG) memcpy(RG, rg, sizeof(double) * N);
  if (HV && V) rie_hess(&p, X, S, V, HV);
  if (PV && V) {
    if (build_dinv(&p, shift)) return 3;
    precondition(&p, X, V, PV);
    free(p.dinv);
  }
  free(EG);
  free(rg);
  free(S);
  return 0;
}

/* QuadraticOptimizer::optimize (src/QuadraticOptimizer.cpp:26-48) */
int dpgo_c_optimize(int n, int d, int r, const int32_t* rowptr, const int32_t* colidx, const double* vals,
                    const double* G, const CParams* prm, const double* X0, double* Xopt, CResult* res) {
  if (d + 1 > MAXB || r > MAXR || n <= 0) return 1;
  Problem p = {n, d, r, d + 1, (d + 1) * r, rowptr, colidx, vals, G, NULL};
  const size_t N = (size_t)n * p.T;
  memset(res, 0, sizeof(*res));
  res->tCGStatus = TCG_MAXITER;
  g_spmm = 0;
  g_tiled_sums = (prm->hess_recurrence & 2) ? 1 : 0;
  if (prm->precond == 1 && build_dinv(&p, prm->precond_shift)) return 3;
  double* wk[11];
  for (int k = 0; k < 11; ++k) {
    wk[k] = (double*)malloc(sizeof(double) * (k == 10 ? (size_t)n * d * d : N));
    if (!wk[k]) return 2;
  }
  double* x = (double*)malloc(sizeof(double) * N);
  if (!x) return 2;
  memcpy(x, X0, sizeof(double) * N);
  double f;
  res->gradNormInit = grad_at(&p, x, wk[5], wk[10], wk[6], &f);
  res->fInit = f;
  if (prm->method == 0) { /* trustRegion(): :50-108 */
    if (!(res->gradNormInit < prm->gradnorm_tol)) {
      if (prm->RTR_iterations == 1) {
        double radius = prm->RTR_initial_radius;
        int total = 0;
        for (;;) {
          memcpy(x, X0, sizeof(double) * N);
          const int acc = run_rtr(&p, prm, x, radius, radius, 1, res, wk);
          if (acc) break;
          if (total > 10) {
            memcpy(x, X0, sizeof(double) * N);
            break;
          }
          radius /= 4.0;
          total++;
        }
      } else {
        run_rtr(&p, prm, x, prm->RTR_initial_radius, 5.0 * prm->RTR_initial_radius, prm->RTR_iterations, res, wk);
      }
    }
  } else { /* gradientDescent(): :110-137 */
    double *g = wk[6], *step = wk[7];
    if (prm->RGD_use_preconditioner) {
      precondition(&p, x, g, wk[8]);
      g = wk[8];
    }
    for (size_t k = 0; k < N; ++k) step[k] = -prm->RGD_stepsize * g[k];
    qf_retract(&p, x, step, wk[9]);
    memcpy(x, wk[9], sizeof(double) * N);
  }
  res->gradNormOpt = grad_at(&p, x, wk[5], wk[10], wk[6], &f);
  res->fOpt = f;
  res->success = 1;
  res->n_spmm = g_spmm;
  memcpy(Xopt, x, sizeof(double) * N);
  for (int k = 0; k < 11; ++k) free(wk[k]);
  free(x);
  if (p.dinv) free(p.dinv);
  return 0;
}
