// Reference known-answer tests re-stated against the C++ mirror (include/dpgo_hip.hpp):
//   tests/testTriangleGraph.cpp:57   (noise-free triangle; 1e-4)
//   tests/testPGO.cpp:131-190        (testPrior; 1e-6)
//   tests/testUtils.cpp:40-54        (LiftedSEManifold::project; 1e-5)
// plus the CSR hand-over (dpgo_problem_set_Q_csr) a reference-side PoseGraph would use.
// Exit code 0 = pass, 77 = no HIP device (the library has no CPU fallback), 1 = failure.
#include <cstdio>
#include <fstream>
#include <cstdlib>

#include "dpgo_hip.hpp"

using namespace dpgo_hip;

static Matrix mul(const Matrix& A, const Matrix& B) {
  Matrix C(A.rows(), B.cols());
  for (size_t i = 0; i < A.rows(); ++i)
    for (size_t j = 0; j < B.cols(); ++j) {
      double s = 0;
      for (size_t k = 0; k < A.cols(); ++k) s += A(i, k) * B(k, j);
      C(i, j) = s;
    }
  return C;
}
static Matrix transpose(const Matrix& A) {
  Matrix C(A.cols(), A.rows());
  for (size_t i = 0; i < A.rows(); ++i)
    for (size_t j = 0; j < A.cols(); ++j) C(j, i) = A(i, j);
  return C;
}
static Matrix se3(const double (&v)[12]) {  // row-major 3x4 literal -> 4x4
  Matrix T = Matrix::Identity(4, 4);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 4; ++j) T(i, j) = v[i * 4 + j];
  return T;
}
static Matrix se3_inv(const Matrix& T) {
  // general inverse (Eigen's Tw0.inverse() in the reference test): the 4-digit rotation literals are not
  // exactly orthogonal, so R^T is NOT the inverse
  Matrix R = T.block(0, 0, 3, 3), t = T.block(0, 3, 3, 1), Ri(3, 3);
  const double det = R(0, 0) * (R(1, 1) * R(2, 2) - R(1, 2) * R(2, 1)) - R(0, 1) * (R(1, 0) * R(2, 2) - R(1, 2) * R(2, 0)) +
                     R(0, 2) * (R(1, 0) * R(2, 1) - R(1, 1) * R(2, 0));
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      const int i1 = (j + 1) % 3, i2 = (j + 2) % 3, j1 = (i + 1) % 3, j2 = (i + 2) % 3;
      Ri(i, j) = (R(i1, j1) * R(i2, j2) - R(i1, j2) * R(i2, j1)) / det;
    }
  Matrix Ti = Matrix::Identity(4, 4);
  Ti.setBlock(0, 0, Ri);
  Matrix mt = mul(Ri, t);
  for (int i = 0; i < 3; ++i) Ti(i, 3) = -mt(i, 0);
  return Ti;
}
static RelativeSEMeasurement between(size_t i, size_t j, const Matrix& Ti, const Matrix& Tj) {
  Matrix dT = mul(se3_inv(Ti), Tj);
  return RelativeSEMeasurement(0, 0, i, j, dT.block(0, 0, 3, 3), dT.block(0, 3, 3, 1), 1.0, 1.0);
}
#define REQUIRE(cond)                                              \
  do {                                                             \
    if (!(cond)) {                                                 \
      std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
      return 1;                                                    \
    }                                                              \
  } while (0)

static int run() {
  const unsigned d = 3, r = 3;
  // ---------------- triangle graph (testTriangleGraph.cpp:16-57)
  const double w1[12] = {0.1436, 0.7406, 0.6564, 1, -0.8179, -0.2845, 0.5000, 1, 0.5571, -0.6087, 0.5649, 1};
  const double w2[12] = {-0.4069, -0.4150, -0.8138, 2, 0.4049, 0.7166, -0.5679, 2, 0.8188, -0.5606, -0.1236, 2};
  Matrix Tw0 = Matrix::Identity(4, 4), Tw1 = se3(w1), Tw2 = se3(w2);
  std::vector<RelativeSEMeasurement> ms = {between(0, 1, Tw0, Tw1), between(1, 2, Tw1, Tw2), between(0, 2, Tw0, Tw2)};
  auto pg = std::make_shared<PoseGraph>(0, r, d);
  pg->setMeasurements(ms);
  REQUIRE(pg->n() == 3);
  QuadraticProblem problem(pg);
  // odometryInitialization (src/DPGO_solver.cpp:271-303)
  Matrix T0(d, 3 * (d + 1));
  Matrix cur = Matrix::Identity(4, 4);
  for (int i = 0; i < 3; ++i) {
    if (i > 0) {
      Matrix dT = Matrix::Identity(4, 4);
      dT.setBlock(0, 0, ms[i - 1].R);
      dT.setBlock(0, 3, ms[i - 1].t);
      cur = mul(cur, dT);
    }
    T0.setBlock(0, i * 4, cur.block(0, 0, 3, 4));
  }
  {  // the library's initial guesses: odometryInitialization equals the hand composition above; solvePGO without an
     // initial guess runs the chordal initialisation on the device and lands at the same optimum (noiseless triangle)
    Matrix Todo = odometryInitialization({ms[0], ms[1]});
    double dmax = 0;
    for (size_t q = 0; q < T0.rows() * T0.cols(); ++q) dmax = std::max(dmax, std::fabs(Todo.data()[q] - T0.data()[q]));
    REQUIRE(dmax <= 1e-14);
    Matrix Tch = chordalInitialization(ms);
    REQUIRE(Tch.rows() == 3 && Tch.cols() == 12);
    ROptParameters pp;
    pp.gradnorm_tol = 1e-9;
    pp.RTR_iterations = 20;
    Matrix Ts = solvePGO(ms, pp);  // T0 = nullptr
    const Matrix* Twp[3] = {&Tw0, &Tw1, &Tw2};
    Matrix R0s = transpose(Ts.block(0, 0, 3, 3));
    double e2 = 0;
    for (int i = 0; i < 3; ++i) {  // trajectory in the frame of pose 0 vs Ttrue
      Matrix Ri = mul(R0s, Ts.block(0, i * 4, 3, 3));
      Matrix dt(3, 1);
      for (int k = 0; k < 3; ++k) dt(k, 0) = Ts(k, i * 4 + 3) - Ts(k, 3);
      Matrix ti = mul(R0s, dt);
      for (int a = 0; a < 3; ++a) {
        for (int b = 0; b < 3; ++b) e2 += std::pow(Ri(a, b) - (*Twp[i])(a, b), 2);
        e2 += std::pow(ti(a, 0) - (*Twp[i])(a, 3), 2);
      }
    }
    std::printf("solvePGO from the chordal initialisation: |Ttrue - T| = %.3e\n", std::sqrt(e2));
    REQUIRE(std::sqrt(e2) <= 1e-4);
  }
  // solve to a tight tolerance so the optimizer really runs (a correct solver lands within 9.1e-5 of
  // Ttrue: the margin of the reference's 1e-4 is consumed by the 4-digit literals)
  QuadraticOptimizer optimizer(&problem);
  optimizer.setGradientNormTolerance(1e-9);
  optimizer.setRTRIterations(20);
  Matrix Topt = optimizer.optimize(T0);
  REQUIRE(optimizer.getOptResult().success);
  // trajectory in the frame of pose 0 vs Ttrue
  Matrix R0t = transpose(Topt.block(0, 0, 3, 3));
  double err2 = 0;
  const Matrix* Tw[3] = {&Tw0, &Tw1, &Tw2};
  for (int i = 0; i < 3; ++i) {
    Matrix Ri = mul(R0t, Topt.block(0, i * 4, 3, 3));
    Matrix dt(3, 1);
    for (int k = 0; k < 3; ++k) dt(k, 0) = Topt(k, i * 4 + 3) - Topt(k, 3);
    Matrix ti = mul(R0t, dt);
    for (int a = 0; a < 3; ++a) {
      for (int b = 0; b < 3; ++b) err2 += std::pow(Ri(a, b) - (*Tw[i])(a, b), 2);
      err2 += std::pow(ti(a, 0) - (*Tw[i])(a, 3), 2);
    }
  }
  std::printf("triangle: |Ttrue - T| = %.3e, f %.3e -> %.3e\n", std::sqrt(err2), optimizer.getOptResult().fInit,
              optimizer.getOptResult().fOpt);
  REQUIRE(std::sqrt(err2) <= 1e-4);

  // ---------------- CSR hand-over: scalar CSR of Q with the structural zeros dropped
  {
    const auto& Q = pg->quadraticMatrix();
    const int b = d + 1, n = 3;
    std::vector<int32_t> outer(n * b + 1, 0), inner;
    std::vector<double> vals;
    for (int i = 0; i < n; ++i)
      for (int rr = 0; rr < b; ++rr) {
        for (int t = Q.rowptr[i]; t < Q.rowptr[i + 1]; ++t)
          for (int cc = 0; cc < b; ++cc) {
            const double v = Q.vals[(size_t)t * b * b + rr * b + cc];
            if (v != 0.0) {
              inner.push_back(Q.colidx[t] * b + cc);
              vals.push_back(v);
            }
          }
        outer[i * b + rr + 1] = (int32_t)inner.size();
      }
    const double f_bsr = problem.f(T0);
    check(dpgo_problem_set_Q_csr(problem.handle(), outer.data(), inner.data(), vals.data()));
    const double f_csr = problem.f(T0);
    std::printf("csr hand-over: f %.15g vs %.15g\n", f_csr, f_bsr);
    REQUIRE(std::fabs(f_csr - f_bsr) <= 1e-13 * std::fabs(f_bsr) + 1e-15);
  }

  // ---------------- prior (testPGO.cpp:131-190)
  {
    RelativeSEMeasurement m(0, 0, 0, 1, Matrix::Identity(3, 3), Matrix::Zero(3, 1), 10000, 100);
    m.fixedWeight = true;
    auto pg2 = std::make_shared<PoseGraph>(0, 3, 3);
    pg2->setMeasurements({m});
    // prior rotation: the literal of testPGO.cpp:158-160 projected to SO(3) -- use the C ABI's polar projection
    Matrix lit(3, 4);
    const double pr[9] = {0.7236, 0.1817, 0.6658, -0.6100, 0.6198, 0.4938, -0.3230, -0.7634, 0.5594};
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) lit(i, j) = pr[i * 3 + j];
    Matrix prior = LiftedSEManifold(3, 3, 1).project(lit);  // det > 0 for this literal: polar factor = SO(3) projection
    pg2->setPrior(1, prior);
    QuadraticProblem prob2(pg2);
    ROptParameters params;
    params.RTR_iterations = 50;
    params.RTR_tCG_iterations = 500;
    params.gradnorm_tol = 1e-5;
    QuadraticOptimizer opt2(&prob2, params);
    Matrix T = Matrix::Zero(3, 8);
    T.setBlock(0, 0, Matrix::Identity(3, 3));
    T.setBlock(0, 4, Matrix::Identity(3, 3));
    Matrix Tq = opt2.optimize(T);
    double e0 = 0, e1 = 0;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 4; ++j) {
        e0 += std::pow(Tq(i, j) - prior(i, j), 2);
        e1 += std::pow(Tq(i, 4 + j) - prior(i, j), 2);
      }
    std::printf("prior: errors %.3e %.3e\n", std::sqrt(e0), std::sqrt(e1));
    REQUIRE(std::sqrt(e0) < 1e-6 && std::sqrt(e1) < 1e-6);
  }

  // ---------------- LiftedSEManifold::project (testUtils.cpp:40-54)
  {
    const int dd = 3, rr = 5, nn = 100;
    Matrix M(rr, (dd + 1) * nn);
    unsigned s = 12345;
    for (size_t k = 0; k < M.rows() * M.cols(); ++k) {
      s = s * 1664525u + 1013904223u;
      M.data()[k] = (double)(s >> 8) / (1u << 23) - 1.0;
    }
    Matrix X = LiftedSEManifold(rr, dd, nn).project(M);
    for (int i = 0; i < nn; ++i) {
      Matrix Y = X.block(0, i * (dd + 1), rr, dd);
      Matrix D = mul(transpose(Y), Y);
      for (int a = 0; a < dd; ++a) D(a, a) -= 1.0;
      REQUIRE(D.norm() <= 1e-5);
    }
    std::printf("project: ok\n");
  }
  // ---------------- chi2inv / RobustCost::computeErrorThresholdAtQuantile (tests/testUtils.cpp:56-71 restated on table
  // values: the reference checks CDF(chi2inv(0.95, 4)) = 0.95 by sampling)
  REQUIRE(std::fabs(chi2inv(0.95, 4) - 9.487729036781154) < 1e-11);
  REQUIRE(std::fabs(chi2inv(0.9, 6) - 10.644640675668422) < 1e-11);
  REQUIRE(std::fabs(chi2inv(0.99, 6) - 16.811893829770927) < 1e-11);
  REQUIRE(std::fabs(RobustCost::computeErrorThresholdAtQuantile(0.9, 3) - 3.2626125537164876) < 1e-12);
  REQUIRE(RobustCost::computeErrorThresholdAtQuantile(1.0, 3) == 1e5);
  std::printf("chi2inv: ok\n");

  // ---------------- rounding (PGOAgent::getTrajectoryInLocalFrame, src/PGOAgent.cpp:718-738): the triangle
  // solution rounded in the frame of pose 0 is Ttrue again; with pose 1 as the global anchor, pose 1 is identity
  {
    Matrix Tl = roundTrajectory(Topt, d);
    double e2 = 0;
    for (int i = 0; i < 3; ++i)
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 4; ++b) e2 += std::pow(Tl(a, i * 4 + b) - (*Tw[i])(a, b), 2);
    std::printf("rounding: |Ttrue - round(T)| = %.3e\n", std::sqrt(e2));
    REQUIRE(std::sqrt(e2) <= 1e-4);
    Matrix anchor = Topt.block(0, 4, 3, 4);
    Matrix Tg = roundTrajectory(Topt, d, &anchor);
    double ea = 0;
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 4; ++b) ea += std::pow(Tg(a, 4 + b) - (a == b ? 1.0 : 0.0), 2);
    REQUIRE(std::sqrt(ea) <= 1e-12);
  }

  // ---------------- robust PGO (testPGO.cpp:193-271): 4-pose chain, one inlier and one outlier loop closure;
  // GNC-TLS with barc = 7, tol 1e-1, 50 RTR iterations drives the weights to 1 and 0
  {
    const int nn = 4;
    const double kap = 10000, ta = 100;
    auto rot = [](double ax, double ay, double az, double ang) {  // Rodrigues
      const double nrm = std::sqrt(ax * ax + ay * ay + az * az);
      ax /= nrm, ay /= nrm, az /= nrm;
      const double c = std::cos(ang), s_ = std::sin(ang), C = 1 - c;
      Matrix Rm(3, 3);
      Rm(0, 0) = c + ax * ax * C, Rm(0, 1) = ax * ay * C - az * s_, Rm(0, 2) = ax * az * C + ay * s_;
      Rm(1, 0) = ay * ax * C + az * s_, Rm(1, 1) = c + ay * ay * C, Rm(1, 2) = ay * az * C - ax * s_;
      Rm(2, 0) = az * ax * C - ay * s_, Rm(2, 1) = az * ay * C + ax * s_, Rm(2, 2) = c + az * az * C;
      return Rm;
    };
    std::vector<Matrix> Tg;
    const double axes[4][4] = {{1, 2, 3, 0.7}, {-2, 1, 0.5, 2.1}, {0.3, -1, 2, 1.3}, {2, 2, -1, 2.9}};
    for (int i = 0; i < nn; ++i) {
      Matrix Ti = Matrix::Identity(4, 4);
      Ti.setBlock(0, 0, rot(axes[i][0], axes[i][1], axes[i][2], axes[i][3]));
      for (int k = 0; k < 3; ++k) Ti(k, 3) = i;
      Tg.push_back(Ti);
    }
    std::vector<RelativeSEMeasurement> meas;
    for (int i = 0; i + 1 < nn; ++i) {
      RelativeSEMeasurement m = between(i, i + 1, Tg[i], Tg[i + 1]);
      m.kappa = kap, m.tau = ta, m.fixedWeight = true;
      meas.push_back(m);
    }
    RelativeSEMeasurement inl = between(0, 3, Tg[0], Tg[3]);
    inl.kappa = kap, inl.tau = ta, inl.fixedWeight = false;
    meas.push_back(inl);
    RelativeSEMeasurement outl(0, 0, 1, 3, rot(-1, 0.2, 0.4, 2.4), Matrix::Zero(3, 1), kap, ta);
    outl.fixedWeight = false;
    meas.push_back(outl);
    // odometryInitialization (src/DPGO_solver.cpp:271-303)
    Matrix TOdom(3, 4 * nn), acc = Matrix::Identity(4, 4);
    for (int i = 0; i < nn; ++i) {
      if (i > 0) {
        Matrix dT = Matrix::Identity(4, 4);
        dT.setBlock(0, 0, meas[i - 1].R);
        dT.setBlock(0, 3, meas[i - 1].t);
        acc = mul(acc, dT);
      }
      TOdom.setBlock(0, i * 4, acc.block(0, 0, 3, 4));
    }
    solveRobustPGOParams rp;
    rp.opt_params.gradnorm_tol = 1e-1;
    rp.opt_params.RTR_iterations = 50;
    rp.robust_params.GNCBarc = 7.0;
    auto mutable_measurements = meas;
    Matrix Tr = solveRobustPGO(mutable_measurements, rp, &TOdom);
    REQUIRE(Tr.cols() == (size_t)4 * nn);
    for (const auto& m : mutable_measurements) {
      if (m.fixedWeight) {
        REQUIRE(m.weight == 1.0);
      } else if (m.p1 == 0 && m.p2 == 3) {
        std::printf("robust: inlier weight %.3e\n", m.weight);
        REQUIRE(std::fabs(m.weight - 1) <= 1e-6);
      } else if (m.p1 == 1 && m.p2 == 3) {
        std::printf("robust: outlier weight %.3e\n", m.weight);
        REQUIRE(std::fabs(m.weight) <= 1e-6);
      }
    }
    // RobustCost::weight known values (GNC-TLS, src/DPGO_robust.cpp:80-92)
    RobustCostParameters cp;
    cp.costType = RobustCostParameters::Type::GNC_TLS;
    cp.GNCInitMu = 1.0;
    cp.GNCBarc = 2.0;
    RobustCost rc(cp);
    REQUIRE(rc.weight(1.0) == 1.0 && rc.weight(3.0) == 0.0);  // rSq <= mu/(mu+1) barc^2 = 2 ; rSq >= 8
    REQUIRE(std::fabs(rc.weight(2.0) - (std::sqrt(4.0 * 2.0 / 4.0) - 1.0)) < 1e-15);
  }

  // ---------------- the preconditioners through the C++ mirror: block-Jacobi and an explicitly set-up multilevel
  // hierarchy reach the same optimum as the default (= multilevel, built by the first solve) on the triangle graph
  // (3 poses -> one aggregate: the coarse solve is exact on the kernel modes)
  for (int pc : {DPGO_PRECOND_BLOCK_JACOBI, DPGO_PRECOND_MULTILEVEL}) {
    if (pc == DPGO_PRECOND_MULTILEVEL) {
      REQUIRE(problem.multilevelCoarseBits() == 64);  // storage defaults: full precision, plain block-CSR on a small block
      bool ops32 = true;
      REQUIRE(problem.multilevelOperatorBits(-1, &ops32) == 32 && !ops32);  // (fp32 operator copies: HBM-bound blocks only)
      REQUIRE(problem.autoInfo().state == 0 && problem.autoInfo().switches == 0);
      REQUIRE(problem.describe().find("DPGO_TILE_WALK=") != std::string::npos);
      REQUIRE(problem.setSpmmVariant(DPGO_SPMM_SYMMETRIC) == DPGO_SPMM_PLAIN);  // needs blocks of >= 40 000 poses
      REQUIRE(problem.setSpmmVariant(DPGO_SPMM_AUTO) == DPGO_SPMM_PLAIN);
      REQUIRE(problem.setupMultilevel({2}) == 2);
      REQUIRE(problem.multilevelPath() == DPGO_ML_PATH_AP);  // two levels, tiny dense level: row-streaming dense kernel
    }
    ROptParameters pm;
    REQUIRE(pm.precond == DPGO_PRECOND_AUTO);
    pm.gradnorm_tol = 1e-9;
    pm.RTR_iterations = 20;
    pm.precond = pc;
    QuadraticOptimizer om(&problem, pm);
    Matrix Tm = om.optimize(T0);
    REQUIRE(om.getOptResult().success);
    std::printf("precond %d: f %.3e -> %.3e (default precond: %.3e)\n", pc, om.getOptResult().fInit,
                om.getOptResult().fOpt, optimizer.getOptResult().fOpt);
    REQUIRE(om.getOptResult().fOpt <= om.getOptResult().fInit * (1 + 1e-12) + 1e-12);
    REQUIRE(std::fabs(om.getOptResult().fOpt - optimizer.getOptResult().fOpt) <= 1e-9);
    double dm = 0;
    for (size_t q = 0; q < Tm.rows() * Tm.cols(); ++q) dm = std::max(dm, std::fabs(Tm.data()[q] - Topt.data()[q]));
    REQUIRE(dm <= 1e-5);
  }

  // ---------------- the solve in two halves (dpgo_optimize_device_begin / _end) and the additive plan through the mirror:
  // same result as the one-call device solve, the halves refuse to be called out of order
  {
    const auto plan = problem.additivePlan();
    REQUIRE(plan.lane_groups == 4 && plan.tile == 16 && plan.aggregates == 1 && plan.graph == 1);  // 3 poses: one aggregate
    ROptParameters pm;
    pm.precond = DPGO_PRECOND_BLOCK_JACOBI;
    QuadraticOptimizer oa(&problem, pm);
    const size_t bytes = sizeof(double) * T0.rows() * T0.cols();
    double *Xa = nullptr, *Xb = nullptr;
    check(dpgo_device_malloc((void**)&Xa, bytes, 0));
    check(dpgo_device_malloc((void**)&Xb, bytes, 0));
    check(dpgo_device_memcpy(Xa, T0.data(), bytes, DPGO_COPY_H2D, nullptr));
    check(dpgo_device_memcpy(Xb, T0.data(), bytes, DPGO_COPY_H2D, nullptr));
    const ROPTResult whole = oa.optimizeDevice(Xa);
    bool threw = false;
    try {
      oa.optimizeDeviceEnd();
    } catch (const Error&) {
      threw = true;  // nothing in flight
    }
    REQUIRE(threw);
    oa.optimizeDeviceBegin(Xb);
    threw = false;
    try {
      oa.optimizeDeviceBegin(Xb);
    } catch (const Error&) {
      threw = true;  // one solve per handle
    }
    REQUIRE(threw);
    const ROPTResult halves = oa.optimizeDeviceEnd();
    REQUIRE(halves.success && halves.fOpt == whole.fOpt && halves.tcgIterations == whole.tcgIterations);
    Matrix A(T0.rows(), T0.cols()), Bm(T0.rows(), T0.cols());
    check(dpgo_device_memcpy(A.data(), Xa, bytes, DPGO_COPY_D2H, nullptr));
    check(dpgo_device_memcpy(Bm.data(), Xb, bytes, DPGO_COPY_D2H, nullptr));
    for (size_t q = 0; q < A.rows() * A.cols(); ++q) REQUIRE(A.data()[q] == Bm.data()[q]);
    check(dpgo_device_free(Xa));
    check(dpgo_device_free(Xb));
    std::printf("begin/end: ok\n");
  }

  // ---------------- LiftedSEVariable / LiftedSEVector (tests/testEigenMap.cpp:12-36: the flat storage is the matrix)
  {
    LiftedSEVariable var(5, 3, 7);
    Matrix X = var.getData();
    REQUIRE(X.rows() == 5 && X.cols() == 28);
    for (unsigned i = 0; i < 7; ++i)
      for (unsigned a = 0; a < 5; ++a)
        for (unsigned c = 0; c < 4; ++c) REQUIRE(X(a, i * 4 + c) == ((c < 3 && a == c) ? 1.0 : 0.0));
    Matrix Yi(5, 3), ti(5, 1);
    for (unsigned a = 0; a < 5; ++a) {
      for (unsigned c = 0; c < 3; ++c) Yi(a, c) = 10.0 * a + c;
      ti(a, 0) = -1.0 - a;
    }
    var.rotation(2) = Yi;       // writable views (Eigen::Ref in the reference)
    var.translation(2) = ti;
    var.pose(6)(4, 3) = 42.0;
    const LiftedSEVariable& cv = var;
    REQUIRE(cv.pose(2)(3, 1) == 31.0 && cv.pose(2)(2, 3) == -3.0 && cv.translation(6)(4, 0) == 42.0);
    REQUIRE(var.data()[(2 * 4 + 1) * 5 + 3] == 31.0);  // column-major r x (d+1)n, pose tiles consecutive
    LiftedSEVariable copy(var);
    copy.pose(0)(0, 0) = 7.0;
    REQUIRE(var.getData()(0, 0) == 1.0 && copy.getData()(0, 0) == 7.0);
    LiftedSEVector vec(5, 3, 7);
    REQUIRE(vec.getData().norm() == 0.0);
    vec.setData(var.getData());
    REQUIRE(vec.getData()(3, 9) == 31.0);
    bool threw = false;
    try {
      var.setData(Matrix(5, 27));
    } catch (const Error& e) {
      threw = e.code == DPGO_ERR_INVALID;
    }
    REQUIRE(threw);
    // the variable's storage feeds the device path unchanged
    LiftedSEManifold man(5, 3, 7);
    Matrix P = man.project(var.getData());
    REQUIRE(P.rows() == 5 && P.cols() == 28);
    std::printf("lifted variable: ok\n");
  }

  // ---------------- error behaviour: shape mismatch is reported, not aborted
  try {
    problem.f(Matrix(2, 5));
    REQUIRE(false);
  } catch (const Error& e) {
    REQUIRE(e.code == DPGO_ERR_INVALID);
  }
  return 0;
}

// ---------------- the reference demo's schedule through the C++ PGOAgent mirror (examples/MultiRobotExample.cpp:170-255):
// greedy block selection + Nesterov acceleration with restarts, public poses exchanged as PoseDicts.  The scenario file
// (written by tests/test_cxx_shim.py from the Python driver's run on the same GPU) holds the measurements, the
// partition, the initial iterate and the expected selection sequence / final cost.
static int run_greedy_scenario(const char* path) {
  std::ifstream in(path);
  REQUIRE(in.good());
  unsigned d, r, robots, n, m, restart;
  in >> d >> r >> robots >> n >> m >> restart;
  std::vector<RelativeSEMeasurement> all(m);
  for (auto& e : all) {
    int fixed;
    in >> e.r1 >> e.p1 >> e.r2 >> e.p2 >> e.kappa >> e.tau >> fixed;
    e.fixedWeight = fixed != 0;
    e.R = Matrix(d, d);
    e.t = Matrix(d, 1);
    for (unsigned p = 0; p < d; ++p)
      for (unsigned q = 0; q < d; ++q) in >> e.R(p, q);
    for (unsigned p = 0; p < d; ++p) in >> e.t(p, 0);
  }
  std::vector<unsigned> start(robots), end(robots);
  for (unsigned a = 0; a < robots; ++a) in >> start[a] >> end[a];
  Matrix X0(r, (size_t)(d + 1) * n);
  for (size_t q = 0; q < X0.rows() * X0.cols(); ++q) in >> X0.data()[q];
  unsigned K;
  in >> K;
  std::vector<unsigned> want(K);
  for (auto& v : want) in >> v;
  double want_cost, want_gn;
  in >> want_cost >> want_gn;
  REQUIRE(in.good());

  PGOAgentParameters prm(d, r, robots, ROptParameters(), /*acceleration=*/true, restart);
  std::vector<std::unique_ptr<PGOAgent>> agents;
  for (unsigned a = 0; a < robots; ++a) {
    agents.emplace_back(new PGOAgent(a, prm));
    agents[a]->setMeasurements(all);  // PoseGraph keeps the edges that touch robot a
    REQUIRE(agents[a]->num_poses() == end[a] - start[a]);
    agents[a]->setX(X0.block(0, (size_t)start[a] * (d + 1), r, (size_t)(end[a] - start[a]) * (d + 1)));
  }
  auto exchange_to = [&](unsigned q, bool aux) {
    for (unsigned a = 0; a < robots; ++a) {
      if (a == q) continue;
      PoseDict dict;
      if (aux)
        agents[a]->getAuxSharedPoseDict(dict);
      else
        agents[a]->getSharedPoseDict(dict);
      if (aux)
        agents[q]->updateAuxNeighborPoses(a, dict);
      else
        agents[q]->updateNeighborPoses(a, dict);
    }
  };
  unsigned selected = 0;
  std::vector<unsigned> order;
  double cost = 0, gn = 0;
  for (unsigned it = 0; it < 1000; ++it) {
    for (unsigned a = 0; a < robots; ++a)
      if (a != selected) agents[a]->iterate(false);
    exchange_to(selected, false);
    exchange_to(selected, true);
    agents[selected]->iterate(true);
    for (unsigned q = 0; q < robots; ++q) exchange_to(q, false);
    std::vector<double> g2(robots);
    cost = 0;
    double gsum = 0;
    for (unsigned a = 0; a < robots; ++a) {
      double hc = 0;
      agents[a]->localTerms(&hc, &g2[a]);
      cost += 2.0 * hc;
      gsum += g2[a];
    }
    gn = std::sqrt(gsum);
    order.push_back(selected);
    if (gn < 0.1) break;  // examples/MultiRobotExample.cpp:229
    selected = (unsigned)(std::max_element(g2.begin(), g2.end()) - g2.begin());
  }
  std::printf("greedy: %zu iterations, cost %.8f, gradnorm %.4f (Python driver: %u, %.8f, %.4f)\n", order.size(), cost, gn,
              K, want_cost, want_gn);
  REQUIRE(order.size() == K);
  for (unsigned k = 0; k < K; ++k) REQUIRE(order[k] == want[k]);
  REQUIRE(std::fabs(cost - want_cost) <= 1e-9 * std::fabs(want_cost));

  // status and the termination vote (PGOAgent::getStatus / setNeighborStatus / shouldTerminate, src/PGOAgent.cpp:399-420,
  // 846-878): an agent that has optimised carries its iteration number and the relative change of its last update; the
  // vote needs every robot's status, INITIALIZED and ready
  bool all_ready = true;
  for (unsigned a = 0; a < robots; ++a) {
    const PGOAgentStatus st = agents[a]->getStatus();
    REQUIRE(st.agentID == a && st.state == PGOAgentState::INITIALIZED && st.relativeChange >= 0.0);
    REQUIRE(st.iterationNumber <= agents[a]->iteration_number());
    REQUIRE(st.readyToTerminate == (st.iterationNumber > 0 && st.relativeChange <= prm.relChangeTol));
    all_ready = all_ready && st.readyToTerminate;
  }
  REQUIRE(!agents[0]->shouldTerminate());  // no neighbour status received yet
  for (unsigned q = 0; q < robots; ++q)
    for (unsigned a = 0; a < robots; ++a)
      if (a != q) agents[q]->setNeighborStatus(agents[a]->getStatus());
  for (unsigned q = 0; q < robots; ++q) REQUIRE(agents[q]->shouldTerminate() == all_ready);
  const PGOAgentStatus last = agents[order.back()]->getStatus();
  std::printf("status of robot %u: iteration %u, relative change %.3e, ready %d; team vote %d\n", order.back(),
              last.iterationNumber, last.relativeChange, (int)last.readyToTerminate, (int)all_ready);
  REQUIRE(last.iterationNumber == agents[order.back()]->iteration_number() && last.relativeChange > 0.0);
  return 0;
}

// Host-only: PoseGraph::setNeighborActive / useInactiveNeighbors (src/PoseGraph.cpp:199-207, 632-634) -- the shared edges
// with an inactive neighbour leave Q and G (:425-430, :527-532); needs no device.
static int check_inactive_neighbours() {
  Matrix I3 = Matrix::Identity(3, 3), t = Matrix::Zero(3, 1);
  t(0, 0) = 1.0;
  std::vector<RelativeSEMeasurement> ms = {RelativeSEMeasurement(1, 1, 0, 1, I3, t, 2.0, 3.0),   // odometry of robot 1
                                           RelativeSEMeasurement(0, 1, 4, 0, I3, t, 5.0, 7.0),   // incoming from robot 0
                                           RelativeSEMeasurement(1, 2, 1, 6, I3, t, 11.0, 13.0)};  // outgoing to robot 2
  PoseGraph pg(1, 3, 3), without(1, 3, 3);
  pg.setMeasurements(ms);
  without.setMeasurements({ms[0], ms[1]});
  REQUIRE(pg.hasNeighbor(0) && pg.hasNeighbor(2) && !pg.hasNeighbor(3) && pg.isNeighborActive(2));
  const auto full = pg.quadraticMatrix().vals;
  const unsigned long v0 = pg.qVersion();
  pg.setNeighborActive(3, false);  // not a neighbour: ignored
  pg.setNeighborActive(2, true);   // unchanged: nothing dropped
  REQUIRE(pg.qVersion() == v0);
  pg.setNeighborActive(2, false);
  REQUIRE(pg.qVersion() > v0 && !pg.isNeighborActive(2) && pg.activeNeighborIDs() == std::set<unsigned>({0}));
  const auto& q1 = pg.quadraticMatrix().vals;
  const auto& q2 = without.quadraticMatrix().vals;
  REQUIRE(q1.size() == q2.size() && q1.size() == full.size());
  double dmax = 0.0, dfull = 0.0;
  for (size_t k = 0; k < q1.size(); ++k) {
    dmax = std::max(dmax, std::fabs(q1[k] - q2[k]));
    dfull = std::max(dfull, std::fabs(q1[k] - full[k]));
  }
  REQUIRE(dmax == 0.0 && dfull > 1.0);
  Matrix X0(3, 4);
  for (unsigned a = 0; a < 3; ++a) X0(a, a) = 1.0;
  pg.setNeighborPoses({{PoseGraph::PoseID(0, 4), X0}});  // robot 2's pose is NOT required any more
  without.setNeighborPoses({{PoseGraph::PoseID(0, 4), X0}});
  const Matrix &G1 = pg.linearMatrix(), &G2 = without.linearMatrix();
  for (size_t k = 0; k < (size_t)G1.rows() * G1.cols(); ++k) REQUIRE(G1.data()[k] == G2.data()[k]);
  pg.setNeighborActive(2, true);
  pg.setNeighborPoses({{PoseGraph::PoseID(0, 4), X0}});
  bool threw = false;
  try {
    pg.linearMatrix();
  } catch (const Error&) {
    threw = true;  // "Missing active neighbor pose" (:523-526)
  }
  REQUIRE(threw);
  const auto& q3 = pg.quadraticMatrix().vals;
  for (size_t k = 0; k < q3.size(); ++k) REQUIRE(q3[k] == full[k]);
  std::printf("inactive neighbours: ok\n");
  return 0;
}

int main(int argc, char** argv) {
  if (check_inactive_neighbours() != 0) return 1;
  int count = 0;
  if (dpgo_device_count(&count) != DPGO_OK || count < 1) {
    // still exercise the failure path: creation must fail with DPGO_ERR_HIP, never fall back
    try {
      auto pg = std::make_shared<PoseGraph>(0, 3, 3);
      Matrix I3 = Matrix::Identity(3, 3), z = Matrix::Zero(3, 1);
      pg->setMeasurements({RelativeSEMeasurement(0, 0, 0, 1, I3, z, 1, 1)});
      QuadraticProblem p(pg);
      std::printf("unexpected: problem created without a device\n");
      return 1;
    } catch (const Error& e) {
      std::printf("no HIP device: %s\n", e.what());
      return e.code == DPGO_ERR_HIP ? 77 : 1;
    }
  }
  try {
    const int rc = run();
    if (rc != 0 || argc < 2) return rc;
    return run_greedy_scenario(argv[1]);
  } catch (const std::exception& e) {
    std::printf("exception: %s\n", e.what());
    return 1;
  }
}
