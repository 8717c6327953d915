// bench_probes.hip -- kernel timing probes behind bench.py (HIP events on the stream of the solver; rotating operand sets for HBM-only rates).
#include "host.h"

extern "C" {


int dpgo_bench_spmm(dpgo_problem_t p, int reps, int warmup, double* avg_ms) {
  CHK(check_ready(p));
  if (reps <= 0 || !avg_ms) return fail(DPGO_ERR_INVALID, "bad arguments");
  hipEvent_t e0, e1;
  HIPC(hipEventCreate(&e0));
  HIPC(hipEventCreate(&e1));
  for (int i = 0; i < warmup; ++i) CHK(launch_spmm(p, p->Q, p->x1, nullptr, p->x2));
  HIPC(hipEventRecord(e0, p->stream));
  for (int i = 0; i < reps; ++i) CHK(launch_spmm(p, p->Q, p->x1, nullptr, p->x2));
  HIPC(hipEventRecord(e1, p->stream));
  HIPC(hipEventSynchronize(e1));
  float ms = 0.f;
  HIPC(hipEventElapsedTime(&ms, e0, e1));
  HIPC(hipEventDestroy(e0));
  HIPC(hipEventDestroy(e1));
  *avg_ms = (double)ms / reps;
  return DPGO_OK;
}


int dpgo_bench_hess_rotating(dpgo_problem_t p, int nsets, int reps, int warmup, double* avg_ms) {
  CHK(check_ready(p));
  if (nsets < 1 || nsets > 512 || reps <= 0 || !avg_ms) return fail(DPGO_ERR_INVALID, "bad arguments");
  std::memset(p->hstate, 0, sizeof(DevState));  // as dpgo_bench_hess: a state without early exits
  p->hstate->z_r = 1.0;
  p->hstate->theta = 1.0;
  p->hstate->kappa = -1.0;
  p->hstate->max_inner = 1 << 30;
  p->hstate->min_inner = 1 << 30;  // the convergence test is never evaluated, whatever the partial sums hold
  CHK(push_state(p));
  // every operand of the tCG-step kernel gets nsets private copies; the handle's pointers are swapped per launch
  const size_t vbytes = sizeof(double) * (size_t)p->Q.nnzb * p->b * p->b;
  const size_t cbytes = sizeof(int32_t) * (size_t)p->Q.nnzb;
  const size_t sbytes = sizeof(double) * (size_t)p->n * p->d * p->d;
  CHK(resolve_tcg_storage(p));
  const bool symq = p->tcg_sym;  // the kernel reads the symmetric copy: that is what rotates
  auto& SY = p->sym;
  struct Set {
    double *vals = nullptr, *x1 = nullptr, *S1 = nullptr, *z = nullptr, *delta = nullptr, *Hd = nullptr;
    int32_t* colidx = nullptr;
    double* uv = nullptr;
    int32_t *uc = nullptr, *lc = nullptr, *ls = nullptr;
  };
  std::vector<Set> sets(nsets);
  const Set orig{p->Q.vals, p->x1, p->S1, p->z, p->delta, p->Hd, p->Q.colidx, SY.uvalsT, SY.ucol, SY.lcol, SY.lslot};
  bool ok = true;
  auto dup = [&](auto** dst, const void* src, size_t bytes) {
    if (!ok) return;
    if (hipMalloc(dst, bytes) != hipSuccess) {
      ok = false;
      return;
    }
    (void)hipMemcpyAsync(*dst, src, bytes, hipMemcpyDeviceToDevice, p->stream);
  };
  for (auto& st : sets) {
    if (symq) {
      dup(&st.uv, orig.uv, sizeof(double) * (size_t)SY.nu * p->b * p->b);
      dup(&st.uc, orig.uc, sizeof(int32_t) * (size_t)SY.nu);
      dup(&st.lc, orig.lc, sizeof(int32_t) * (size_t)std::max(1, SY.nl));
      dup(&st.ls, orig.ls, sizeof(int32_t) * (size_t)std::max(1, SY.nl));
    } else {
      dup(&st.vals, orig.vals, vbytes);
      dup(&st.colidx, orig.colidx, cbytes);
    }
    dup(&st.x1, orig.x1, p->vec_bytes());
    dup(&st.S1, orig.S1, sbytes);
    dup(&st.z, orig.z, p->vec_bytes());
    dup(&st.delta, orig.delta, p->vec_bytes());
    dup(&st.Hd, orig.Hd, p->vec_bytes());
  }
  auto use = [&](const Set& st) {
    if (symq) {
      SY.uvalsT = st.uv;
      SY.ucol = st.uc;
      SY.lcol = st.lc;
      SY.lslot = st.ls;
    } else {
      p->Q.vals = st.vals;
      p->Q.colidx = st.colidx;
    }
    p->x1 = st.x1;
    p->S1 = st.S1;
    p->z = st.z;
    p->delta = st.delta;
    p->Hd = st.Hd;
  };
  int rc = ok ? DPGO_OK : fail(DPGO_ERR_HIP, "hipMalloc failed for the rotating buffer sets");
  auto launch = [&](int i) -> int {
    use(sets[i % nsets]);
    return launch_tcg_hess_with(p, p->dstate, p->dstate + 1, 0, nullptr, 0u);
  };
  float ms = 0.f;
  if (rc == DPGO_OK) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int i = 0; i < warmup && rc == DPGO_OK; ++i) rc = launch(i);
    (void)hipEventRecord(e0, p->stream);
    for (int i = 0; i < reps && rc == DPGO_OK; ++i) rc = launch(i);
    (void)hipEventRecord(e1, p->stream);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
  }
  use(orig);
  (void)hipStreamSynchronize(p->stream);
  for (auto& st : sets) {
    void* ptrs[] = {st.vals, st.colidx, st.x1, st.S1, st.z, st.delta, st.Hd, st.uv, st.uc, st.lc, st.ls};
    for (void* q : ptrs)
      if (q) (void)hipFree(q);
  }
  if (rc != DPGO_OK) return rc;
  *avg_ms = (double)ms / reps;
  return DPGO_OK;
}


namespace {
// rotating copies of the symmetric storage (values, column indices, references; the row pointers are shared)
int bench_spmm_sym_rotating(dpgo_problem_s* p, int nsets, int reps, int warmup, double* avg_ms, double* set_bytes) {
  const auto& S = p->sym;
  const size_t vbytes = sizeof(double) * (size_t)S.nu * p->b * p->b;
  struct Set {
    double *v = nullptr, *x = nullptr, *o = nullptr;
    int32_t *uc = nullptr, *lc = nullptr, *ls = nullptr;
  };
  std::vector<Set> sets(nsets);
  int rc = DPGO_OK;
  auto cleanup = [&]() {
    for (auto& st : sets) {
      void* ptrs[] = {st.v, st.x, st.o, st.uc, st.lc, st.ls};
      for (void* q : ptrs)
        if (q) (void)hipFree(q);
    }
  };
  for (auto& st : sets) {
    if (hipMalloc(&st.v, vbytes) != hipSuccess || hipMalloc(&st.x, p->vec_bytes()) != hipSuccess ||
        hipMalloc(&st.o, p->vec_bytes()) != hipSuccess || hipMalloc(&st.uc, sizeof(int32_t) * S.nu) != hipSuccess ||
        hipMalloc(&st.lc, sizeof(int32_t) * std::max(1, S.nl)) != hipSuccess ||
        hipMalloc(&st.ls, sizeof(int32_t) * std::max(1, S.nl)) != hipSuccess) {
      rc = fail(DPGO_ERR_HIP, "hipMalloc failed for the rotating buffer sets");
      break;
    }
    (void)hipMemcpyAsync(st.v, S.uvalsT, vbytes, hipMemcpyDeviceToDevice, p->stream);
    (void)hipMemcpyAsync(st.uc, S.ucol, sizeof(int32_t) * S.nu, hipMemcpyDeviceToDevice, p->stream);
    (void)hipMemcpyAsync(st.lc, S.lcol, sizeof(int32_t) * S.nl, hipMemcpyDeviceToDevice, p->stream);
    (void)hipMemcpyAsync(st.ls, S.lslot, sizeof(int32_t) * S.nl, hipMemcpyDeviceToDevice, p->stream);
    (void)hipMemcpyAsync(st.x, p->x1, p->vec_bytes(), hipMemcpyDeviceToDevice, p->stream);
  }
  if (rc != DPGO_OK) {
    cleanup();
    return rc;
  }
  auto launch = [&](int i) {
    const Set& st = sets[i % nsets];
    return launch_spmm_sym(p, BsrSymDev{S.urow, st.uc, st.v, S.lrow, st.lc, st.ls, S.tord}, st.x, nullptr, st.o);
  };
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int i = 0; i < warmup && rc == DPGO_OK; ++i) rc = launch(i);
  (void)hipEventRecord(e0, p->stream);
  for (int i = 0; i < reps && rc == DPGO_OK; ++i) rc = launch(i);
  (void)hipEventRecord(e1, p->stream);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  cleanup();
  if (rc != DPGO_OK) return rc;
  *avg_ms = (double)ms / reps;
  if (set_bytes)
    *set_bytes = (double)(vbytes + sizeof(int32_t) * ((size_t)S.nu + 2 * (size_t)S.nl) + 2 * p->vec_bytes());
  return DPGO_OK;
}
}  // namespace


int dpgo_bench_spmm_rotating(dpgo_problem_t p, int nsets, int reps, int warmup, double* avg_ms,
                             double* set_bytes) {
  CHK(check_ready(p));
  if (nsets < 1 || nsets > 512 || reps <= 0 || !avg_ms) return fail(DPGO_ERR_INVALID, "bad arguments");
  if (p->sym_wanted()) {
    bool usable = false;
    CHK(sym_ensure(p, &usable));
    if (usable) return bench_spmm_sym_rotating(p, nsets, reps, warmup, avg_ms, set_bytes);
  }
  // nsets private copies of (Q values, block columns, X, OUT): cycling through them makes every launch read
  // data that left the 256 MB Infinity Cache ("rotate >= 3 buffer sets > 256 MB total")
  const size_t vbytes = sizeof(double) * (size_t)p->Q.nnzb * p->b * p->b;
  const size_t cbytes = sizeof(int32_t) * (size_t)p->Q.nnzb;
  std::vector<Bsr> mats(nsets);
  std::vector<double*> xs(nsets, nullptr), outs(nsets, nullptr);
  int rc = DPGO_OK;
  auto cleanup = [&]() {
    for (int k = 0; k < nsets; ++k) {
      if (mats[k].vals) (void)hipFree(mats[k].vals);
      if (mats[k].colidx) (void)hipFree(mats[k].colidx);
      if (xs[k]) (void)hipFree(xs[k]);
      if (outs[k]) (void)hipFree(outs[k]);
    }
  };
  for (int k = 0; k < nsets && rc == DPGO_OK; ++k) {
    mats[k] = p->Q;  // shares rowptr (0.4 MB)
    mats[k].vals = nullptr;
    mats[k].colidx = nullptr;
    if (hipMalloc(&mats[k].vals, vbytes) != hipSuccess || hipMalloc(&mats[k].colidx, cbytes) != hipSuccess ||
        hipMalloc(&xs[k], p->vec_bytes()) != hipSuccess || hipMalloc(&outs[k], p->vec_bytes()) != hipSuccess) {
      rc = fail(DPGO_ERR_HIP, "hipMalloc failed for the rotating buffer sets");
      break;
    }
    (void)hipMemcpyAsync(mats[k].vals, p->Q.vals, vbytes, hipMemcpyDeviceToDevice, p->stream);
    (void)hipMemcpyAsync(mats[k].colidx, p->Q.colidx, cbytes, hipMemcpyDeviceToDevice, p->stream);
    (void)hipMemcpyAsync(xs[k], p->x1, p->vec_bytes(), hipMemcpyDeviceToDevice, p->stream);
  }
  if (rc != DPGO_OK) {
    cleanup();
    return rc;
  }
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int i = 0; i < warmup && rc == DPGO_OK; ++i) rc = launch_spmm(p, mats[i % nsets], xs[i % nsets], nullptr, outs[i % nsets]);
  (void)hipEventRecord(e0, p->stream);
  for (int i = 0; i < reps && rc == DPGO_OK; ++i) rc = launch_spmm(p, mats[i % nsets], xs[i % nsets], nullptr, outs[i % nsets]);
  (void)hipEventRecord(e1, p->stream);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  cleanup();
  if (rc != DPGO_OK) return rc;
  *avg_ms = (double)ms / reps;
  if (set_bytes) *set_bytes = (double)(vbytes + cbytes + 2 * p->vec_bytes());
  return DPGO_OK;
}


int dpgo_bench_hess(dpgo_problem_t p, int reps, int warmup, double* avg_ms) {
  CHK(check_ready(p));
  if (reps <= 0 || !avg_ms) return fail(DPGO_ERR_INVALID, "bad arguments");
  CHK(resolve_tcg_storage(p));
  // a state in which the tCG-step kernel never takes an early exit
  std::memset(p->hstate, 0, sizeof(DevState));
  p->hstate->z_r = 1.0;
  p->hstate->theta = 1.0;
  p->hstate->kappa = -1.0;  // convergence test can never fire
  p->hstate->max_inner = 1 << 30;
  p->hstate->min_inner = 1 << 30;  // the convergence test is never evaluated, whatever the partial sums hold
  CHK(push_state(p));
  auto launch = [&]() -> int {
    return launch_tcg_hess_with(p, p->dstate, p->dstate + 1, 0, nullptr, 0u);
  };
  hipEvent_t e0, e1;
  HIPC(hipEventCreate(&e0));
  HIPC(hipEventCreate(&e1));
  for (int i = 0; i < warmup; ++i) CHK(launch());
  HIPC(hipEventRecord(e0, p->stream));
  for (int i = 0; i < reps; ++i) CHK(launch());
  HIPC(hipEventRecord(e1, p->stream));
  HIPC(hipEventSynchronize(e1));
  float ms = 0.f;
  HIPC(hipEventElapsedTime(&ms, e0, e1));
  HIPC(hipEventDestroy(e0));
  HIPC(hipEventDestroy(e1));
  *avg_ms = (double)ms / reps;
  return DPGO_OK;
}


int dpgo_bench_solve(dpgo_problem_t p, const dpgo_ropt_params* params, const double* X0_dev, int reps, int warmup,
                     double* avg_ms, double* avg_products, int* persistent) {
  CHK(check_ready(p));
  if (reps <= 0 || !params || !X0_dev || !avg_ms) return fail(DPGO_ERR_INVALID, "bad arguments");
  // every repetition solves from the same iterate (copied in outside the event pair)
  hipEvent_t e0, e1;
  HIPC(hipEventCreate(&e0));
  HIPC(hipEventCreate(&e1));
  double total = 0.0, products = 0.0;
  bool all_persistent = true;
  int rc = DPGO_OK;
  for (int i = 0; i < warmup + reps && rc == DPGO_OK; ++i) {
    rc = [&]() -> int {
      HIPC(hipMemcpyAsync(p->x1, X0_dev, p->vec_bytes(), hipMemcpyDeviceToDevice, p->stream));
      dpgo_ropt_result res;
      HIPC(hipEventRecord(e0, p->stream));
      CHK(run_optimize(p, params, &res));
      HIPC(hipEventRecord(e1, p->stream));
      HIPC(hipEventSynchronize(e1));
      float ms = 0.f;
      HIPC(hipEventElapsedTime(&ms, e0, e1));
      if (i >= warmup) {
        total += ms;
        products += res.tcg_iterations;
        all_persistent = all_persistent && p->hctrl && p->hctrl->members > 0 && !p->persist_failed_once;
      }
      return DPGO_OK;
    }();
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (rc != DPGO_OK) return rc;
  *avg_ms = total / reps;
  if (avg_products) *avg_products = products / reps;
  if (persistent) *persistent = all_persistent ? 1 : 0;
  return DPGO_OK;
}


int dpgo_bench_iteration_kernels(dpgo_problem_t p, int reps, int warmup, double out_ms[5]) {
  CHK(check_ready(p));
  if (reps <= 0 || !out_ms) return fail(DPGO_ERR_INVALID, "bad arguments");
  for (int q = 0; q < 5; ++q) out_ms[q] = 0.0;
  CHK(resolve_tcg_storage(p));
  // a state in which no kernel takes an early exit (as dpgo_bench_hess); alpha = z_r / d_Hd stays finite
  std::memset(p->hstate, 0, sizeof(DevState));
  p->hstate->z_r = 1.0;
  p->hstate->theta = 1.0;
  p->hstate->kappa = -1.0;
  p->hstate->max_inner = 1 << 30;
  p->hstate->min_inner = 1 << 30;  // the convergence test is never evaluated, whatever the partial sums hold
  p->hstate->Delta = 1e300;
  CHK(push_state(p));
  CHK(build_dinv(p, p->ml_ready ? p->ml_shift : 1e-1));
  {  // <delta, H delta> partials of a "previous k_tcg_hess": positive, so that the update kernel takes its regular path
    std::vector<double> ones((size_t)kPartialCap * kNP, 1.0);
    HIPC(hipMemcpyAsync(p->pA(), ones.data(), sizeof(double) * ones.size(), hipMemcpyHostToDevice, p->stream));
    HIPC(hipStreamSynchronize(p->stream));
  }
  hipEvent_t e0, e1;
  HIPC(hipEventCreate(&e0));
  HIPC(hipEventCreate(&e1));
  auto timed = [&](auto&& launch, double* out) -> int {
    for (int i = 0; i < warmup; ++i) CHK(launch());
    HIPC(hipEventRecord(e0, p->stream));
    for (int i = 0; i < reps; ++i) CHK(launch());
    HIPC(hipEventRecord(e1, p->stream));
    HIPC(hipEventSynchronize(e1));
    float ms = 0.f;
    HIPC(hipEventElapsedTime(&ms, e0, e1));
    *out = (double)ms / reps;
    return DPGO_OK;
  };
  const bool ml = p->ml_ready;
  int rc = timed([&]() -> int {
    const int cur = p->cur;
    int r2 = launch_tcg_update(p, p->dinv, 0, ml ? p->ml[0].x1 : nullptr, ml ? p->ml_omega : 0.0);
    p->cur = cur;  // keep reading the pushed state
    return r2;
  }, &out_ms[0]);
  if (rc == DPGO_OK && ml) {
    const int nl = (int)p->ml.size();
    auto& L0 = p->ml[0];
    const bool ap = p->ml_use_ap();
    rc = timed([&]() -> int { return launch_ml_restrict0(p, p->rr, nullptr, p->grid_restrict()); }, &out_ms[1]);
    if (rc == DPGO_OK) rc = timed([&]() -> int {
      auto& L = p->ml[nl - 2];
      auto& Cc = p->ml[nl - 1];
      if (p->ml_use_dense_sym()) return launch_dense_sym(p, Cc, nullptr);
      return launch_coarse_prolong(p, L, Cc, nullptr, ap ? Cc.x : nullptr);
    }, &out_ms[2]);
    if (rc == DPGO_OK) rc = timed([&]() -> int {
      if (ap) {
        return launch_ml_post_ap(p, p->x1, p->rr, p->z, p->pB(), nullptr);
      } else if (p->tcg_sym) {
        DISPATCH(p->d, p->r, hipLaunchKernelGGL((k_ml_post<D, R, 1, BsrSymDev>), dim3(p->grid_post()), dim3(kBlock), 0,
                                                p->stream, p->sym.dev(), p->x1, L0.x, p->rr, p->dinv, p->ml_omega,
                                                p->ml_shift, p->z, p->pB(), (const DevState*)nullptr, p->n));
      } else {
        DISPATCH(p->d, p->r, LAUNCH_SPLIT(p, k_ml_post, p->grid_post(), p->Q.dev(), p->x1, L0.x, p->rr, p->dinv, p->ml_omega,
                                          p->ml_shift, p->z, p->pB(), (const DevState*)nullptr, p->n));
      }
      HIPC(hipGetLastError());
      return DPGO_OK;
    }, &out_ms[3]);
    if (rc == DPGO_OK)
      rc = timed([&]() -> int { return launch_ml_tail(p, p->x1, p->rr, p->z, p->pB(), nullptr); }, &out_ms[4]);
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return rc;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------
// Test probe of the one-launch solve's communication primitives (kernels/persist.h, kernels/common.h), without a solve
// around them: `steps` chip-wide reductions of K = 2 partial sums, each carrying a payload of PAY doubles per workgroup
// (chip_allreduce<2, PAY>), and the wavefront reduce-scatter the additive preconditioner's coarse solve uses
// (wave_reduce_rows).  tests/test_parity_gpu.py::test_in_kernel_reduction_primitives.
namespace dpgo {
template <int PAY>
__global__ __launch_bounds__(kBlock, 1) void k_probe_allreduce(unsigned long long* gran, unsigned salt, int steps,
                                                               const double* __restrict__ in, const double* __restrict__ pay_in,
                                                               double* __restrict__ sums, double* __restrict__ pay_out,
                                                               double* __restrict__ rows_out, int* error, int poll) {
  constexpr int TP = (PAY + 3) / 4 * 4;
  __shared__ double red[2 * 2 * kWaves * kGranVals];
  __shared__ double tw_s[kWaves][PAY];
  __shared__ int ok_s;
  const int rank = blockIdx.x, members = gridDim.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) ok_s = 1;
  unsigned step = 0;
  for (int s = 0; s < steps; ++s) {
    double part[2], got[PAY];
    part[0] = in[((size_t)rank * kBlock + threadIdx.x) * 2 + 0] * (s + 1);
    part[1] = in[((size_t)rank * kBlock + threadIdx.x) * 2 + 1] - s;
    if (lane < PAY) tw_s[wave][lane] = pay_in[((size_t)rank * kWaves + wave) * PAY + lane] + s;
    if (!chip_allreduce<2, PAY>(gran, rank, members, salt, step, part, red, error, &ok_s, poll, &tw_s[0][0], got)) return;
    if (threadIdx.x == 0) {
      sums[((size_t)rank * steps + s) * 2 + 0] = part[0];
      sums[((size_t)rank * steps + s) * 2 + 1] = part[1];
    }
    if ((int)threadIdx.x < members) {
#pragma unroll
      for (int e = 0; e < PAY; ++e) pay_out[(((size_t)rank * steps + s) * members + threadIdx.x) * PAY + e] = got[e];
    }
    __syncthreads();  // (tw_s is rewritten by the next step)
  }
  // reduce-scatter of PAY values over every wavefront: value e of thread t = in[t][0] * (e + 1) + in[t][1]
  double v[TP], rs[TP / 4];
#pragma unroll
  for (int e = 0; e < TP; ++e)
    v[e] = e < PAY ? in[((size_t)rank * kBlock + threadIdx.x) * 2] * (e + 1) + in[((size_t)rank * kBlock + threadIdx.x) * 2 + 1] : 0.0;
  wave_reduce_rows<TP>(v, rs);
  if ((threadIdx.x & 15) == 15) {
    const int q = wave_rows_value(lane);
#pragma unroll
    for (int j = 0; j < TP / 4; ++j)
      if (4 * j + q < PAY) rows_out[((size_t)rank * kWaves + wave) * PAY + 4 * j + q] = rs[j];
  }
}
}  // namespace dpgo

extern "C" int dpgo_debug_reduction_primitives(int workgroups, int pay, int steps, const double* in_dev, const double* pay_in_dev,
                                               double* sums_dev, double* pay_out_dev, double* rows_out_dev) {
  using namespace dpgo;
  if (workgroups < 1 || workgroups > kPersistMax || steps < 1 || !in_dev || !pay_in_dev || !sums_dev || !pay_out_dev || !rows_out_dev)
    return fail(DPGO_ERR_INVALID, "bad arguments");
  int device = 0, cus = 0;
  HIPC(hipGetDevice(&device));
  HIPC(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
  if (workgroups > cus) return fail(DPGO_ERR_INVALID, "more workgroups than the device holds at once");
  unsigned long long* gran = nullptr;
  int* error = nullptr;
  HIPC(hipMalloc(&gran, sizeof(unsigned long long) * kGranWords));
  HIPC(hipMalloc(&error, sizeof(int)));
  HIPC(hipMemset(gran, 0, sizeof(unsigned long long) * kGranWords));
  HIPC(hipMemset(error, 0, sizeof(int)));
  const unsigned salt = 5u << 20;
  const int poll = (kPollFirstSleep << 8) | kPollSleep;
  switch (pay) {
#define CASE_(P)                                                                                                         \
  case P:                                                                                                                \
    hipLaunchKernelGGL((k_probe_allreduce<P>), dim3(workgroups), dim3(kBlock), 0, nullptr, gran, salt, steps, in_dev,   \
                       pay_in_dev, sums_dev, pay_out_dev, rows_out_dev, error, poll);                                    \
    break;
    CASE_(6) CASE_(9) CASE_(15) CASE_(20) CASE_(24)
#undef CASE_
    default:
      (void)hipFree(gran);
      (void)hipFree(error);
      return fail(DPGO_ERR_UNSUPPORTED, "payload size: one of 6, 9, 15, 20, 24");
  }
  HIPC(hipGetLastError());
  HIPC(hipDeviceSynchronize());
  int herr = 0;
  HIPC(hipMemcpy(&herr, error, sizeof(int), hipMemcpyDeviceToHost));
  (void)hipFree(gran);
  (void)hipFree(error);
  if (herr) return fail(DPGO_ERR_HIP, "a reduction timed out");
  return DPGO_OK;
}
