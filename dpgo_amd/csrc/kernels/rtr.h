// kernels/rtr.h -- trust-region acceptance test and state transitions (tail of ROPTLIB SolversTR::Run).
// Part of kernels.h (included inside namespace dpgo, in this order: common.h, problem.h, tcg.h, persist.h, multilevel.h, dense.h, manifold.h, rtr.h, agent.h, init.h).
#pragma once

// ================================================================ K7c: RTR acceptance test
// ROPTLIB SolversTR::Run, tail of one outer iteration: rho = (f1 - f2) / -(<eta,g> + 0.5 <eta,H eta>),
// radius update, acceptance (rho > 0.1, or the tiny-decrease clause), and on acceptance
// x1 <- x2, g1 <- g2, S1 <- S2.
// pe: k_grad partials at x2;  ph: k_hess partials for V = eta, Gdot = g1.
template <int D, int R>
__global__ __launch_bounds__(kBlock) void k_rtr_update(double* __restrict__ x1, const double* __restrict__ x2,
                                                       double* __restrict__ g1, const double* __restrict__ g2,
                                                       double* __restrict__ S1, const double* __restrict__ S2,
                                                       const double* __restrict__ pe, int nb_e,
                                                       const double* __restrict__ ph, int nb_h,
                                                       const DevState* __restrict__ sin, DevState* __restrict__ sout,
                                                       int n) {
  using GEO = Geo<D, R>;
  __shared__ double red[kWaves * kNP];
  DevState st;
  load_state(st, sin);
  if (st.rtr_stop) {
    if (blockIdx.x == 0 && threadIdx.x == 0) store_state(sout, st);
    return;
  }
  double e3[3], h2[2];
  load_partials<3>(pe, nb_e, e3, red);
  load_partials<2>(ph, nb_h, h2, red);
  const double f2 = 0.5 * e3[0] + e3[1];
  const double ngf2 = sqrt(e3[2]);
  const double eta_Heta = h2[0], eta_g = h2[1];
  const double rho = (st.f1 - f2) / (-(eta_g + 0.5 * eta_Heta));
  if (rho > 0.75) {
    if (st.tcg_status == TCG_EXCREGION || st.tcg_status == TCG_NEGCURV) st.Delta *= 2.0;
    if (st.Delta > st.Delta_max) st.Delta = st.Delta_max;
  } else if (rho < 0.25) {
    st.Delta *= 0.25;
  }
  const double sqeps = 1.4901161193847656e-08;  // sqrt(DBL_EPSILON)
  bool accept = rho > 0.1;
  if (!accept && st.accept_tiny) accept = (fabs(st.f1 - f2) / (fabs(st.f1) + 1.0) < sqeps) && (f2 < st.f1);
  st.f2 = f2;
  st.rho = rho;
  st.accepted_last = accept ? 1 : 0;
  st.outer_iter += 1;
  if (accept) {
    st.f1 = f2;
    st.ngf = ngf2;
    st.n_accept += 1;
    st.rtr_stop = (ngf2 < st.tol) ? 1 : 0;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) store_state(sout, st);
  if (!accept) return;
  const size_t total = (size_t)n * GEO::T;
  const size_t stride = (size_t)gridDim.x * kBlock;
  for (size_t e = (size_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += stride) {
    x1[e] = x2[e];
    g1[e] = g2[e];
  }
  const size_t totS = (size_t)n * D * D;
  for (size_t e = (size_t)blockIdx.x * kBlock + threadIdx.x; e < totS; e += stride) S1[e] = S2[e];
}

// RTR start: f1, |g1| from k_grad partials at x1; initial radius; stop test.
static __global__ void k_rtr_begin(const double* __restrict__ pe, int nb_e, DevState* __restrict__ s0, double tol,
                            double Delta0, double Delta_max, int max_inner, int accept_tiny) {
  __shared__ double red[kWaves * kNP];
  double e3[3];
  load_partials<3>(pe, nb_e, e3, red);
  if (threadIdx.x == 0) {
    DevState st;
    st.f1 = 0.5 * e3[0] + e3[1];
    st.ngf = sqrt(e3[2]);
    st.Delta = Delta0;
    st.Delta_max = Delta_max;
    st.tol = tol;
    st.f2 = st.f1;
    st.rho = 0.0;
    st.fInit = st.f1;
    st.gnInit = st.ngf;
    st.xqx = e3[0];
    st.xg = e3[1];
    st.outer_iter = 0;
    st.rtr_stop = (st.ngf < tol) ? 1 : 0;
    st.accepted_last = 0;
    st.n_accept = 0;
    st.accept_tiny = accept_tiny;
    st.pad0 = 0;
    st.z_r = st.d_Pd = st.e_Pd = st.e_Pe = st.norm_r0 = st.alpha = 0.0;
    st.theta = 1.0;   // ROPTLIB RTRNewton default (SURVEY 8c' item 4)
    st.kappa = 0.1;
    st.tcg_j = 0;
    st.tcg_done = 0;
    st.tcg_status = TCG_MAXITER;
    st.max_inner = max_inner;
    st.n_hess = 0;
    st.min_inner = 0;
    store_state(s0, st);
    store_state(s0 + 1, st);
  }
}
