// solve.hip -- QuadraticOptimizer::optimize on the device (src/QuadraticOptimizer.cpp, ROPTLIB RTRNewton / tCG_TR): tCG launches, the one-launch solve, preconditioner selection.
#include "host.h"

namespace dpgo_host {

int launch_tcg_update(dpgo_problem_s* p, const double* dinv, int first, double* z_out,
                      double ml_omega) {
  const bool ml_mode = ml_omega > 0.0;
  const int g = p->grid_u(ml_mode);
  double* zt = z_out ? z_out : p->z;
  // (the pre-smoothed iterate of a cycle that keeps its internal vectors in fp32 goes to that buffer instead)
  float* z32 = (ml_omega > 0.0 && !p->ml.empty() && z_out == p->ml[0].x1 && p->ml_vec32_active()) ? p->ml[0].x1f : nullptr;
  DISPATCH(p->d, p->r, {
    if constexpr (Span<D, R, 1>::kOk) {
      if (ml_mode)  // (the instance compiled for the multilevel mode: 3 waves per SIMD)
        hipLaunchKernelGGL((k_tcg_update_span<D, R, 1>), dim3(g), dim3(kBlock), 0, p->stream, p->x1, p->g1, dinv, p->delta,
                           p->Hd, p->eta, p->rr, zt, p->pA(), p->grid_s(), p->pB(), p->dstate + p->cur,
                           p->dstate + (p->cur ^ 1), first, p->n, p->hflag, p->launch_gen(), ml_omega, z32);
      else
        hipLaunchKernelGGL((k_tcg_update_span<D, R, 0>), dim3(g), dim3(kBlock), 0, p->stream, p->x1, p->g1, dinv, p->delta,
                           p->Hd, p->eta, p->rr, zt, p->pA(), p->grid_s(), p->pB(), p->dstate + p->cur,
                           p->dstate + (p->cur ^ 1), first, p->n, p->hflag, p->launch_gen(), ml_omega, z32);
    } else {
      // (odd tile size: the generic kernel has no fp32 output -- resolve_tcg_storage keeps such blocks off the symmetric
      // storage, hence off the cycle's fp32 vectors; a state that says otherwise is refused instead of dropping z32)
      if (z32) return fail(DPGO_ERR_STATE, "fp32 cycle vectors need the span kernels (even pose tile size)");
      hipLaunchKernelGGL((k_tcg_update<D, R>), dim3(g), dim3(kBlock), 0, p->stream, p->x1, p->g1, dinv, p->delta,
                         p->Hd, p->eta, p->rr, zt, p->pA(), p->grid_s(), p->pB(), p->dstate + p->cur,
                         p->dstate + (p->cur ^ 1), first, p->n, p->hflag, p->launch_gen(), ml_omega);
    }
  });
  HIPC(hipGetLastError());
  p->cur ^= 1;
  return DPGO_OK;
}

// fused direction update + Riemannian Hessian-vector product (one tCG step)
// the tCG-step kernel: span variant whenever the pose tile size is even (all 3-D cases)
#define LAUNCH_TCG_HESS(p, SIN, SOUT, FIRST, HFLAG, GEN)                                                          \
  do {                                                                                                            \
    if constexpr (Span<D, R, 1>::kOk) {                                                                           \
      if ((p)->tcg_sym && options().hess_dma == 1)                                                                \
        hipLaunchKernelGGL((k_tcg_hess_sym_dma<D, R, 1, 2, 4, 1>), dim3((p)->grid_s()), dim3(kBlock), 0,          \
                           (p)->stream, (p)->sym.dev(), (p)->x1, (p)->S1, (p)->z, (p)->delta, (p)->Hd, (p)->pB(), \
                           (p)->nb_zr(), (p)->pA(), SIN, SOUT, FIRST, (p)->n, HFLAG, GEN);                         \
      else if ((p)->tcg_sym && options().hess_dma == 2)                                                           \
        hipLaunchKernelGGL((k_tcg_hess_sym_dma<D, R, 1, 3, 2, 0>), dim3((p)->grid_s()), dim3(kBlock), 0,          \
                           (p)->stream, (p)->sym.dev(), (p)->x1, (p)->S1, (p)->z, (p)->delta, (p)->Hd, (p)->pB(), \
                           (p)->nb_zr(), (p)->pA(), SIN, SOUT, FIRST, (p)->n, HFLAG, GEN);                         \
      else if ((p)->tcg_sym && (p)->stream_nt)                                                                    \
        hipLaunchKernelGGL((k_tcg_hess_sym<D, R, 1>), dim3((p)->grid_s()), dim3(kBlock), 0, (p)->stream,          \
                           (p)->sym.dev(), (p)->x1, (p)->S1, (p)->z, (p)->delta, (p)->Hd, (p)->pB(), (p)->nb_zr(), \
                           (p)->pA(), SIN, SOUT, FIRST, (p)->n, HFLAG, GEN);                                      \
      else if ((p)->tcg_sym)                                                                                      \
        hipLaunchKernelGGL((k_tcg_hess_sym<D, R, 0>), dim3((p)->grid_s()), dim3(kBlock), 0, (p)->stream,          \
                           (p)->sym.dev(), (p)->x1, (p)->S1, (p)->z, (p)->delta, (p)->Hd, (p)->pB(), (p)->nb_zr(), \
                           (p)->pA(), SIN, SOUT, FIRST, (p)->n, HFLAG, GEN);                                      \
      else if ((p)->stream_nt && (p)->split == 1)                                                                 \
        hipLaunchKernelGGL((k_tcg_hess_span<D, R, 1, 1>), dim3((p)->grid_s()), dim3(kBlock), 0, (p)->stream,      \
                           (p)->Q.dev(), (p)->x1, (p)->S1, (p)->z, (p)->delta, (p)->Hd, (p)->pB(), (p)->nb_zr(),   \
                           (p)->pA(), SIN, SOUT, FIRST, (p)->n, HFLAG, GEN);                                      \
      else                                                                                                        \
      LAUNCH_SPLIT(p, k_tcg_hess_span, (p)->grid_s(), (p)->Q.dev(), (p)->x1, (p)->S1, (p)->z, (p)->delta, (p)->Hd, \
                   (p)->pB(), (p)->nb_zr(), (p)->pA(), SIN, SOUT, FIRST, (p)->n, HFLAG, GEN);                   \
    } else                                                                                                        \
      LAUNCH_SPLIT(p, k_tcg_hess, (p)->grid_s(), (p)->Q.dev(), (p)->x1, (p)->S1, (p)->z, (p)->delta, (p)->Hd,      \
                   (p)->pB(), (p)->nb_zr(), (p)->pA(), SIN, SOUT, FIRST, (p)->n, HFLAG, GEN);                   \
  } while (0)

int launch_tcg_hess_with(dpgo_problem_s* p, const DevState* sin, DevState* sout, int first, unsigned long long* hflag,
                         unsigned gen) {  // (the state slots and the progress word of the caller's choice: kernel probes)
  DISPATCH(p->d, p->r, LAUNCH_TCG_HESS(p, sin, sout, first, hflag, gen));
  HIPC(hipGetLastError());
  return DPGO_OK;
}
int launch_tcg_hess(dpgo_problem_s* p, int first) {
  CHK(launch_tcg_hess_with(p, p->dstate + p->cur, p->dstate + (p->cur ^ 1), first, p->hflag, p->launch_gen()));
  p->cur ^= 1;
  return DPGO_OK;
}

// which storage of Q the tCG-step kernel of the coming launches reads (Q does not change inside a solve)
int resolve_tcg_storage(dpgo_problem_s* p) {
  p->tcg_sym = false;
  p->stream_nt = p->want_stream_nt();
  bool span = false;
  DISPATCH(p->d, p->r, { span = Span<D, R, 1>::kOk; });
  if (!span || !p->sym_wanted()) return DPGO_OK;
  bool usable = false;
  CHK(sym_ensure(p, &usable));
  p->tcg_sym = usable;
  return DPGO_OK;
}

// ---------------------------------------------------------------------------------------------------------
// One-launch solve (kernels/persist.h, k_rtr_persist): one launch runs QuadraticOptimizer::optimize whole.
//
// Residency.  Every workgroup of such a launch waits for all the others, so all of them must be resident at once.  The
// grid is therefore sized against a per-device count of resident slots shared by all handles of the process (one slot =
// one 256-thread workgroup; capacity = two per CU: every variant of the kernel is compiled for two workgroups per CU
// -- registers, LDS --, whatever else runs), reserved for the duration of the solve.  A handle that cannot reserve runs the
// multi-launch scheme.  Other processes are not covered: every in-kernel spin is bounded, a time-out poisons the state
// record (rtr_stop = kPersistPoison) and leaves the caller's iterate untouched; run_optimize then runs the solve with the
// multi-launch scheme.
constexpr int kMaxDevices = 64;
std::atomic<int> g_warnings{0};  // warnings printed to stderr so far (dpgo_warning_count)
std::atomic<int> g_persist_used[kMaxDevices];
std::atomic<int> g_persist_cap[kMaxDevices];  // 0 = not yet queried

int persist_capacity(int device) {
  int cap = g_persist_cap[device % kMaxDevices].load();
  if (cap == 0) {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0) cus = 1;
    cap = 2 * cus;
    g_persist_cap[device % kMaxDevices].store(cap);
  }
  return cap;
}
bool persist_reserve(dpgo_problem_s* p, int slots, int limit) {
  auto& used = g_persist_used[p->device % kMaxDevices];
  int cur = used.load();
  while (cur + slots <= limit)
    if (used.compare_exchange_weak(cur, cur + slots)) {
      p->persist_reserved = slots;
      return true;
    }
  return false;
}
void persist_release(dpgo_problem_s* p) {
  if (p->persist_reserved > 0) g_persist_used[p->device % kMaxDevices].fetch_sub(p->persist_reserved);
  p->persist_reserved = 0;
}

// Geometry of a launch: lane groups per pose (SPLIT), tiles per workgroup (MT), workgroups.  The smallest-latency layout
// whose grid fits the handle's share of the resident slots: 4 lane groups per pose (short gather chains) while the tiles
// fit, otherwise one pose per (d+1) lanes, with up to 2 tiles per workgroup.
bool additive_available(dpgo_problem_s* p) {
  return p->persist && !p->persist_failed_once && additive_plan(p).split != 0;
}
// `free_slots`: what may be reserved.  Alone on the device (share = 1): the lowest-latency layout that fits (4 lane groups
// per pose while the tiles fit, then one pose per (d+1) lanes).  Sharing the device with `share` concurrently solved
// agents: the lowest-latency layout of which `share` copies fit side by side; if there is none, the most compact one
// (the solves then take turns).
PersistGeo persist_geometry(const dpgo_problem_s* p, int free_slots, int share, bool additive) {
  if (additive) {  // fixed by the hierarchy (after ml_ensure); one workgroup per CU (the rows of the coarse inverse live in its LDS)
    const int sp = additive_split_of(p);
    if (!sp) return PersistGeo();
    PersistGeo g{sp, 1, p->ml[1].n, 0};
    g.slots = g.wgs * persist_slots_per_wg(sp, 1, true);
    if (g.wgs > kPersistMax || g.slots > free_slots) return PersistGeo();
    return g;
  }
  const int env_split = options().persist_split, env_mt = options().persist_mt;
  const int cand[4][2] = {{4, 1}, {4, 2}, {1, 1}, {1, 2}};
  PersistGeo compact;
  for (auto& c : cand) {
    if (env_split && c[0] != env_split) continue;
    if (env_mt && c[1] != env_mt) continue;
    const int P = (64 / (p->b * c[0])) * kWaves;
    const int tiles = std::max(1, (p->n + P - 1) / P);
    const int wgs = (tiles + c[1] - 1) / c[1];
    const int slots = wgs * persist_slots_per_wg(c[0], c[1]);
    if (wgs > kPersistMax || slots > free_slots) continue;
    const PersistGeo g{c[0], c[1], wgs, slots};
    if ((long long)slots * std::max(1, share) <= free_slots) return g;  // everybody fits at once
    if (compact.wgs == 0 || slots < compact.slots) compact = g;
  }
  return compact;
}

// Enqueues the persistent launch of a WHOLE solve (k_rtr_persist; no host wait).  *used = false: not launched (no geometry
// / no free slots) -- the caller runs the multi-launch scheme.
int launch_rtr_persistent(dpgo_problem_s* p, const dpgo_ropt_params* prm, const double* dinv, bool* used, bool additive) {
  *used = false;
  p->gen += 1;
  if (p->persist_stream_ordered) {
    // dpgo_optimize_device_begin: the caller enqueues this handle's solves and everything between them on ONE stream, so
    // no two of its one-launch solves are ever resident together -- nothing to reserve (a reservation could only be
    // released by the collecting call, long after the kernel has left the chip)
    const PersistGeo g = persist_geometry(p, persist_capacity(p->device), 1, additive);
    if (g.wgs <= 0) return DPGO_OK;
    p->persist_split = g.split;
    p->persist_mt = g.mt;
    p->persist_wgs = g.wgs;
    p->persist_add = additive;
  } else if (p->persist_reserved == 0) {
    // Alone on the device: what is free now, first come first served.  Sharing it with other concurrently solved agents:
    // the most compact layout, at most 4/5 of the slots in use at once (a CU that holds a persistent workgroup has no
    // registers left for anything else, and every agent's other kernels -- gradient, retraction, rho test -- need
    // somewhere to run: packing the chip full made a 16-agent sweep slower), and a solve that finds no room WAITS for
    // another one to finish (a solve is well under a millisecond) instead of taking the slow path.
    const int cap = persist_capacity(p->device);
    auto& used = g_persist_used[p->device % kMaxDevices];
    PersistGeo g;
    if (p->persist_share <= 1) {
      g = persist_geometry(p, cap - used.load(), 1, additive);
      if (g.wgs <= 0 || !persist_reserve(p, g.slots, cap)) return DPGO_OK;
    } else {
      // (the additive form's grid is fixed by its hierarchy -- one workgroup per aggregate, up to the whole chip: such solves
      // take turns)
      const int limit = additive ? cap : cap - cap / 5;
      g = persist_geometry(p, limit, p->persist_share, additive);
      if (g.wgs <= 0) return DPGO_OK;
      const auto t0 = std::chrono::steady_clock::now();
      while (!persist_reserve(p, g.slots, limit)) {
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 0.05) return DPGO_OK;
        std::this_thread::yield();
      }
    }
    p->persist_split = g.split;
    p->persist_mt = g.mt;
    p->persist_wgs = g.wgs;
    p->persist_add = additive;
  }
  // (the granules' epochs are salted per launch, so what earlier launches left in the table never matches; the table is
  // cleared before a salt can repeat -- every 2047 generations of this handle -- and at creation)
  // (the control block: every field but `error` is assigned by the launch, and `error` is zero unless a launch of this
  // handle timed out -- cleared at creation and behind such a launch only, not by a fill kernel in front of every solve)
  if (p->pctrl_dirty) {
    HIPC(hipMemsetAsync(p->pctrl, 0, sizeof(PersistCtrl), p->stream));
    p->pctrl_dirty = false;
  }
  if (p->gen - p->gran_cleared_at >= 0x7ffu) {  // (the multi-launch scheme advances `gen` too: count, do not test bits)
    HIPC(hipMemsetAsync(p->pgran, 0, sizeof(unsigned long long) * kGranWords, p->stream));
    p->gran_cleared_at = p->gen;
  }
  const unsigned salt = ((p->gen & 0x7ffu) + 1u) << 20;  // never 0; the in-launch step counter fills the low 20 bits
  // granule sweeps of the in-kernel all-reduce: wait before the first one (a granule needs ~1 us to cross the chip and
  // the slowest of more workgroups arrives later; sweeping earlier only loads the fabric: sphere2500 11.9 -> 7.9 us per
  // iteration, 12.5k slab 15.3 -> 12.1), back off between sweeps.  DPGO_POLL_FIRST / DPGO_POLL_SLEEP override.
  const int env_first = options().poll_first, env_sleep = options().poll_sleep;
  // (whole-solve kernel, run r4j, us per product at first = 16 / 24 / 32 / 44 / 56: sphere2500, 157 workgroups of 4 lane
  // groups per pose, 7.3 / 6.5 / 7.0 / 7.6 / 8.3; 6 250 poses, 196 workgroups of the same layout with two tiles, 10.9 / 9.9 /
  // 9.9 / 10.5 / 11.1; 12.5k slab, one pose per (d+1) lanes, 12.0 / 10.9 / 10.7 / 10.5 / 10.4)
  const int first = env_first >= 0 ? std::min(255, env_first)
                                   : (p->persist_split == 4 ? (p->persist_wgs <= 160 ? kPollFirstSleep : 30) : 32);
  // (one pose per (d+1) lanes: 44 until round 6; with the phases that round shortened -- own tiles from LDS, 16-byte cells --
  // the sweep over 12 / 20 / 28 / 36 / 44 / 52 has its best at 28-36: slab additive 17.1 / 15.7 / 14.6 / 14.7 / 15.0 / 15.3,
  // block-Jacobi 10.5 / 9.9 / 9.6 / 9.6 / 9.7 / 9.7, torus3D additive 17.1 / 15.5 / 15.0 / 14.8 / 15.2 / 15.7 us per product;
  // tools/r6/poll_first_sweep.sh)
  const int between = env_sleep >= 0 ? std::min(255, env_sleep) : kPollSleep;
  // (a reduction that also carries the additive preconditioner's payload -- 2 (d+1) r more granules per participant --
  // completes later: its first sweep waits longer; 0 in the argument = as the plain one)
  const int env_first_pay = options().poll_first_pay;
  const int first_pay = env_first_pay >= 0 ? std::min(255, env_first_pay) : kPollFirstPaySleep;
  const int poll = (first_pay << 16) | (first << 8) | between;

  AddDev add{};
  size_t lds = 0;
  if (additive) {
    auto& L0 = p->ml[0];
    auto& C = p->ml[1];
    add = AddDev{L0.Pb, p->ml_dense, p->ml_lda, C.n, C.r, 1.0, L0.graph ? L0.tile_perm : nullptr,
                 L0.graph ? L0.lab : nullptr, L0.graph ? L0.mem_pos : nullptr, L0.graph ? L0.agg_ptr : nullptr};
    lds = sizeof(double) * (size_t)p->b * C.n * p->b;  // (d+1) rows of the inverse
  }
  const RtrArgs ra{prm->gradnorm_tol, prm->RTR_initial_radius, 5.0 * prm->RTR_initial_radius, prm->RTR_tCG_iterations,
                   prm->RTR_iterations, prm->accept_tiny_decrease};
  const double* Glin = p->has_G ? p->G : nullptr;
  // (static + dynamic LDS of the additive instances can exceed 64 KB: the launch attribute is raised to the largest size
  // any handle of the process has asked of that instantiation on that device)
#define PERSIST_LAUNCH(SP, MT_, ADD_, LDS_)                                                                           \
  do {                                                                                                                \
    if ((LDS_) > 0) { /* the attribute belongs to the instantiation and the device: only ever raised */              \
      static std::atomic<int> hw_[kMaxDevices];                                                                       \
      auto& h_ = hw_[p->device % kMaxDevices];                                                                        \
      if ((int)(LDS_) > h_.load()) {                                                                                  \
        HIPC(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rtr_persist<D, R, SP, MT_, ADD_>),                   \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LDS_)));                           \
        int cur_ = h_.load();                                                                                         \
        while (cur_ < (int)(LDS_) && !h_.compare_exchange_weak(cur_, (int)(LDS_))) {}                                 \
      }                                                                                                               \
    }                                                                                                                 \
    hipLaunchKernelGGL((k_rtr_persist<D, R, SP, MT_, ADD_>), dim3(p->persist_wgs), dim3(kBlock), LDS_, p->stream,     \
                       p->Q.dev(), p->x1, Glin, dinv, p->x2, p->eta, p->z, p->pgran, salt, p->dstate, p->pctrl, p->n, \
                       p->hflag, p->gen, poll, ra, add);                                                              \
  } while (0)
  DISPATCH(p->d, p->r, {
    if (additive && p->persist_split == 4) PERSIST_LAUNCH(4, 1, true, lds);
    else if (additive) PERSIST_LAUNCH(1, 1, true, lds);
    else if (p->persist_split == 4 && p->persist_mt == 1) PERSIST_LAUNCH(4, 1, false, 0);
    else if (p->persist_split == 4) PERSIST_LAUNCH(4, 2, false, 0);
    else if (p->persist_mt == 1) PERSIST_LAUNCH(1, 1, false, 0);
    else PERSIST_LAUNCH(1, 2, false, 0);
  });
#undef PERSIST_LAUNCH
  {  // the iterate reaches the caller's X only if the launch completed on every participant (k_persist_commit)
    const size_t count = (size_t)p->n * p->T;
    const int grid = (int)std::min<size_t>(1024, (count + kBlock - 1) / kBlock);
    // (p->cur = 0 below: dstate[0] is the record the solve leaves; the commit kernel also writes it and the control block
    // into the host-coherent copies the caller reads after synchronising)
    hipLaunchKernelGGL(k_persist_commit, dim3(grid), dim3(kBlock), 0, p->stream, p->dstate, p->pctrl, p->x2, p->x1, count,
                       p->hstate, p->hctrl);
  }
  HIPC(hipGetLastError());
  p->cur = 0;
  *used = true;
  return DPGO_OK;
}

void persist_report(dpgo_problem_s* p) {  // (hctrl has been read back with the state record)
  if (!options().persist_verbose) return;
  const double it = std::max<double>(1.0, (double)p->hctrl->ticks[4]);
  std::fprintf(stderr,
               "dpgo_hip: persistent tCG: %u workgroups (%d lane groups per pose, %d tiles each)%s, %u iterations; per "
               "iteration (us): Hessian phase %.2f, all-reduce %.2f, update phase %.2f, all-reduce %.2f; per solve (us): "
               "set-up + initial statistics %.1f, first updates %.1f, retraction / trial point / rho test %.1f\n",
               p->hctrl->members, p->persist_split, p->persist_mt, p->hctrl->error ? " TIMED OUT" : "", p->hctrl->iters,
               0.01 * (double)p->hctrl->ticks[0] / it, 0.01 * (double)p->hctrl->ticks[1] / it,
               0.01 * (double)p->hctrl->ticks[2] / it, 0.01 * (double)p->hctrl->ticks[3] / it,
               0.01 * (double)p->hctrl->ticks[5], 0.01 * (double)p->hctrl->ticks[6], 0.01 * (double)p->hctrl->ticks[7]);
}

// ---- one steady tCG iteration as an instantiated hipGraph (dpgo_problem_s::IterGraph) ----
// Everything the launches of an iteration read from the handle, hashed: a captured graph is valid exactly while this
// value is unchanged (a buffer that was freed and came back at the same address with the same sizes is the same launch).
struct KeyHash {
  unsigned long long h = 1469598103934665603ull;
  void add(const void* ptr) { mix((unsigned long long)(uintptr_t)ptr); }
  void add(long long v) { mix((unsigned long long)v); }
  void add(double v) {
    unsigned long long u;
    std::memcpy(&u, &v, sizeof(u));
    mix(u);
  }
  void mix(unsigned long long v) {
    for (int k = 0; k < 8; ++k) {
      h ^= (v >> (8 * k)) & 0xffull;
      h *= 1099511628211ull;
    }
  }
};
void key_bsr(KeyHash& k, const Bsr& m) {
  k.add((long long)m.nrows), k.add((long long)m.ncols), k.add((long long)m.nnzb);
  k.add(m.rowptr), k.add(m.colidx), k.add(m.vals);
}
unsigned long long iter_graph_key(const dpgo_problem_s* p, const double* dinv, bool ml, bool early_stop) {
  KeyHash k;
  k.add((long long)p->d), k.add((long long)p->r), k.add((long long)p->n), k.add((long long)p->split), k.add((long long)p->cur);
  k.add((long long)p->tcg_sym), k.add((long long)p->stream_nt), k.add((long long)ml), k.add((long long)early_stop);
  k.add((long long)p->grid()), k.add((long long)p->grid_u(true)), k.add((long long)p->grid_s()), k.add((long long)p->grid_restrict()), k.add((long long)p->grid_post());
  k.add((long long)p->zr_from_post), k.add((long long)p->nb_zr()), k.add((long long)p->device);
  key_bsr(k, p->Q);
  const auto& y = p->sym;
  k.add(y.urow), k.add(y.ucol), k.add(y.uvalsT), k.add(y.lrow), k.add(y.lcol), k.add(y.lslot), k.add(y.tord);
  const void* vecs[] = {p->x1, p->g1, p->S1, p->z, p->delta, p->Hd, p->eta, p->rr, dinv, p->dinv, p->partials, p->dstate, p->hflag};
  for (auto v : vecs) k.add(v);
  if (!ml) return k.h;
  k.add(p->ml_omega), k.add(p->ml_shift), k.add((long long)p->ml_coarse_bits), k.add((long long)p->ml_lda);
  k.add((long long)p->ml_use_ap()), k.add((long long)p->ml_use_dense_sym()), k.add((long long)p->beyond_cache());
  k.add(p->ml_dense), k.add(p->ml_dense32), k.add(p->ml_packed), k.add(p->ml_pd), k.add(p->ml_pt), k.add(p->ml_chunks);
  k.add(p->ml_chunk_first), k.add((long long)p->ml_nchunks), k.add((long long)p->ml.size());
  k.add((long long)p->ml_ops32_active()), k.add((long long)p->ml_vec32_active()), k.add((long long)p->coarse32_active());
  k.add(p->sym.uvalsT32);
  for (const auto& L : p->ml) {
    k.add((long long)L.n), k.add((long long)L.k), k.add((long long)L.split), k.add((long long)L.graph), k.add((long long)L.nseg);
    key_bsr(k, L.A), key_bsr(k, L.AP);
    const void* ptrs[] = {L.slot_row, L.dinv, L.Pb, L.r, L.x1, L.x, L.res1, L.lab, L.agg_ptr, L.agg_mem, L.parent, L.pslot,
                          L.mem_pos, L.seg_info, L.seg_ptr, L.tile_perm, L.tbuf, L.Pb32, L.AP32, L.x1f, L.res1f};
    for (auto v : ptrs) k.add(v);
  }
  return k.h;
}

// One ROPTLIB SolversTR::Run outer iteration: tCG + retraction + rho test.  State stays on the device; the host
// feeds tCG-step kernels just-in-time (or polls the state every `tcg_poll_interval` inner iterations).
int rtr_outer_iteration(dpgo_problem_s* p, const dpgo_ropt_params* prm, const double* dinv, Counters& cnt,
                        bool poll_at_end) {
  p->gen += 1;
  const bool add = prm->precond == DPGO_PRECOND_ADDITIVE;
  // (additive: where the persistent kernel cannot run -- no free slots, an earlier time-out -- the V-cycle on the same
  // two-level hierarchy takes over)
  const bool ml = prm->precond == DPGO_PRECOND_MULTILEVEL || add;
  p->zr_from_post = ml;
  if (add) cnt.vcycle_for_additive = true;
  // multilevel: the update kernel writes the pre-smoothing step of level 0 instead of the block-Jacobi z; the cycle's
  // last kernel produces z and the partial sums <r,r>, <z,r>
  auto update = [&](int first) -> int {
    if (ml) {
      CHK(launch_tcg_update(p, dinv, first, p->ml[0].x1, p->ml_omega));
      const bool early_stop = options().ml_early_stop != 0;
      return launch_ml_tail(p, p->x1, p->rr, p->z, p->pB(), p->dstate + p->cur, early_stop && !first);
    }
    return launch_tcg_update(p, dinv, first);
  };
  CHK(update(1));
  const int max_inner = prm->RTR_tCG_iterations;
  // Steady iterations (j >= 1) of the just-in-time feed are replayed from an instantiated hipGraph: the launches of one
  // iteration recorded once (stream capture on the handle's private stream -- the caller's may be the legacy default
  // stream, which cannot be captured --, generation 0 = "the one in the state record") and launched into the handle's
  // stream.  Same kernels, same arguments, same order: bit-identical iterates.  OPT-IN (DPGO_ITER_GRAPH=1): measured on the
  // 100k-pose grid (round 5, two interleaved repetitions) a replayed iteration is SLOWER than its six stream launches --
  // 158.2 / 157.1 against 150.6 / 152.0 us per product -- although the boundary between two kernels INSIDE one graph is half
  // a stream boundary (tools/launch_lab.hip: 1.6 against 3.4 us): every hipGraphLaunch of this six-node graph costs more
  // than the five boundaries it shortens.
  const bool graph_wanted = options().iter_graph != 0, early_stop_ = options().ml_early_stop != 0;
  const bool use_graph = graph_wanted && !p->iter_graph_failed && prm->tcg_poll_interval <= 0 && max_inner > 1 && p->own_stream;
  auto replay = [&]() -> int {  // one steady iteration; DPGO_OK with *launched = false: the caller launches directly
    auto& g = p->iter_graph[p->cur & 1];
    const unsigned long long key = iter_graph_key(p, dinv, ml, early_stop_);
    if (!g.exec || g.key != key) {
      if (g.exec) (void)hipGraphExecDestroy(g.exec);
      g.exec = nullptr;
      hipStream_t own = p->own_stream, keep = p->stream;
      const int cur0 = p->cur;
      if (keep == own) HIPC(hipStreamSynchronize(own));  // (recording starts on an idle stream)
      if (hipStreamBeginCapture(own, hipStreamCaptureModeThreadLocal) != hipSuccess) {
        (void)hipGetLastError();
        p->iter_graph_failed = true;
        return DPGO_ERR_UNSUPPORTED;
      }
      p->stream = own;
      p->capturing = true;
      int rc = launch_tcg_hess(p, 0);
      if (rc == DPGO_OK) rc = update(0);
      p->capturing = false;
      p->stream = keep;
      p->cur = cur0;
      hipGraph_t graph = nullptr;
      const hipError_t e1 = hipStreamEndCapture(own, &graph);
      hipError_t e2 = hipSuccess;
      if (rc == DPGO_OK && e1 == hipSuccess && graph) e2 = hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0);
      if (graph) (void)hipGraphDestroy(graph);
      if (rc != DPGO_OK || e1 != hipSuccess || e2 != hipSuccess || !g.exec) {
        (void)hipGetLastError();
        g.exec = nullptr;
        p->iter_graph_failed = true;
        return DPGO_ERR_UNSUPPORTED;
      }
      g.key = key;
    }
    if (hipGraphLaunch(g.exec, p->stream) != hipSuccess) {
      (void)hipGetLastError();
      p->iter_graph_failed = true;
      return DPGO_ERR_UNSUPPORTED;
    }
    return DPGO_OK;
  };
  auto step = [&](int j) -> int {
    if (j >= 1 && use_graph && !p->iter_graph_failed) {
      if (replay() == DPGO_OK) return DPGO_OK;  // (otherwise: nothing of this iteration has been enqueued)
      if (options().persist_verbose) std::fprintf(stderr, "dpgo_hip: hipGraph replay of a tCG iteration unavailable; plain launches\n");
    }
    CHK(launch_tcg_hess(p, j == 0 ? 1 : 0));
    return update(0);
  };
  if (max_inner <= 0) CHK(launch_tcg_hess(p, 1));  // only finalises the tCG state (eta = 0)
  bool done = false;
  if (prm->tcg_poll_interval > 0) {
    // polling mode: enqueue `poll` iterations, then synchronise and read the state back
    const int poll = prm->tcg_poll_interval;
    int j = 0;
    while (j < max_inner) {
      const int chunk = (max_inner - j) < poll ? (max_inner - j) : poll;
      for (int c = 0; c < chunk; ++c) CHK(step(j + c));
      j += chunk;
      CHK(poll_state(p));
      if (p->hstate->tcg_done || p->hstate->rtr_stop) {
        done = true;
        break;
      }
    }
  } else {
    // just-in-time feed: stay kAhead iterations ahead of the progress word the device publishes into
    // host-coherent memory; no synchronisation, no copy, at most kAhead wasted (early-exit) iterations
    // a multilevel iteration is 5+ launches: waste fewer of them after tCG stops.  (2 is the minimum: iteration j's count
    // is published by the prologue of iteration j+1's Hessian-step kernel, so one iteration ahead never sees progress --
    // tried in round 4, the watchdog fires.)
    int kAhead = ml ? 2 : 4;
    if (options().tcg_ahead > 0) kAhead = std::max(2, options().tcg_ahead);  // tuning knob (>= 2, see above)
    int enq = 0, last_j = -1;
    auto t_progress = std::chrono::steady_clock::now();
    while (true) {
      const unsigned long long w = __atomic_load_n(p->hflag, __ATOMIC_ACQUIRE);
      int dev_j = 0;
      if ((unsigned)(w >> 32) == p->gen) {
        dev_j = (int)((w >> 8) & 0xFFFFFFu);
        if (w & 3ull) {
          done = true;
          if (w & 2ull) p->saw_rtr_stop = true;
          break;
        }
      }
      if (enq >= max_inner) break;
      if (dev_j != last_j) {  // the watchdog measures time WITHOUT progress, not time since the loop started
        last_j = dev_j;
        t_progress = std::chrono::steady_clock::now();
      }
      if (enq < dev_j + kAhead) {
        CHK(step(enq));
        enq += 1;
      } else {
        __builtin_ia32_pause();
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_progress).count() > 60.0) {
          HIPC(hipStreamSynchronize(p->stream));  // surfaces a device fault instead of spinning forever
          return fail(DPGO_ERR_HIP, "tCG progress word did not advance for 60 s");
        }
      }
    }
  }
  if (!done) CHK(launch_tcg_hess(p, 0));  // max_inner iterations enqueued, not (yet known to be) finished: last prologue
  if (p->saw_rtr_stop) return DPGO_OK;  // the previous outer iteration already met the stop test
  CHK(launch_retract(p, p->x1, p->eta, 1.0, p->x2, p->dstate + p->cur));
  CHK(launch_grad(p, p->x2, p->g2, p->S2, nullptr, p->dstate + p->cur, p->tcg_sym && outer_sym_enabled()));
  cnt.spmm += 1;
  CHK(launch_hess(p, p->x1, p->S1, p->eta, p->g1, p->Hd, p->pH(), p->dstate + p->cur, 0, p->tcg_sym && outer_sym_enabled()));
  cnt.spmm += 1;
  CHK(launch_rtr_update(p));
  if (poll_at_end) CHK(poll_state(p));
  return DPGO_OK;
}

// DPGO_PRECOND_AUTO after a solve: what the next one runs (dpgo_hip.h).  `used` = the preconditioner the solve resolved to,
// `products` = its Hessian-vector products.
//   * a block without coupling to other agents never hands back (its cheap early calls end on the trust-region boundary
//     after a few products whatever the preconditioner, and the switch back and forth cost the 100k grid 37.6 against
//     29.3 ms);
//   * a coupled block the additive one-launch solve cannot hold: hysteresis on the share of the tCG budget a solve used
//     (the V-cycle is ~3x a block-Jacobi iteration: it pays when the budget binds);
//   * a coupled block the additive solve CAN hold (<= ~14 000 poses): the cost rule.  Q is constant across RBCD sweeps, so
//     the hierarchy is paid once: when the block-Jacobi solves since the last change of Q have cost as much as one set-up
//     (kAutoSetupUnits: 2.8-3.0 ms against 9.9-10.6 us per block-Jacobi product on 6 250 / 12 500 poses), the next solve
//     runs additive on trial; it stays while its products x the additive unit cost (19 us) stay below the reference
//     block-Jacobi solve's x the block-Jacobi unit cost, and hands back otherwise (the hierarchy is kept: the next trial
//     is free but waits twice as long).
//     The unit costs are those of a solve that has the device to itself.  A handle solved NEXT TO others of the device
//     (dpgo_optimize_device_many, persist_share > 1) is charged for the part of the chip its launch blocks instead: the
//     additive form owns one CU per aggregate (up to the whole chip: such solves take turns), block-Jacobi's compact
//     layout a quarter to a half of it, so there a product costs [us] x max(resident slots / slots in use at once, 1 / share)
//     x share -- the same figure as alone whenever every concurrently solved handle fits at once.
// (round 6: an additive iteration is two chip-wide reductions, as a block-Jacobi one -- 11.0 against 9.6 us on a 12 500-pose
// slab, 8.7 against 6.6 on sphere2500; it was 15.7 / 10.6 with three: 18 units)
constexpr int kAutoUnitsJacobi = 10, kAutoUnitsAdditive = 13, kAutoSetupUnits = 2800, kAutoMinProducts = 6;
int auto_units_jacobi(dpgo_problem_s* p) {
  const int share = std::max(1, p->persist_share);
  if (share == 1 || !p->persist) return kAutoUnitsJacobi;
  const int cap = persist_capacity(p->device), limit = cap - cap / 5;  // (what launch_rtr_persistent lets such solves use)
  const PersistGeo g = persist_geometry(p, limit, share, false);
  if (g.wgs <= 0) return kAutoUnitsJacobi;
  const double part = std::max((double)g.slots / limit, 1.0 / share);
  // (two tiles per workgroup cost 1.25x per product with 4 lane groups per pose; with one pose per (d+1) lanes -- the
  // 12 500-pose slabs -- that variant keeps 512 registers and spills: measured under sharing 22 us per product against
  // 9.4 alone, i.e. two such solves side by side gain nothing over one after the other: 2.1x.  Round 6, loop-back sweeps
  // 8 x 12 500: forced block-Jacobi 3.81 ms, forced additive 3.05 ms, the rule with 1.25x here 3.72 ms -- a mix.)
  const double two_tiles = g.mt == 2 ? (g.split == 1 ? 2.1 : 1.25) : 1.0;
  return std::max(1, (int)std::lround(kAutoUnitsJacobi * two_tiles * part * share));
}
int auto_units_additive(dpgo_problem_s* p) {  // (after additive_available(p): the plan exists)
  const int share = std::max(1, p->persist_share);
  if (share == 1) return kAutoUnitsAdditive;
  const int cap = persist_capacity(p->device);
  const double part = std::max((double)(p->add_plan.na * persist_slots_per_wg(p->add_plan.split, 1, true)) / cap, 1.0 / share);
  return std::max(1, (int)std::lround(kAutoUnitsAdditive * part * share));
}
void auto_update(dpgo_problem_s* p, const dpgo_ropt_params* prm, int used, int products) {
  const int budget = std::max(1, prm->RTR_iterations) * std::max(1, prm->RTR_tCG_iterations);
  const bool coupled = p->has_G || p->C.nnzb > 0;
  auto& a = p->auto_cost;
  a.last_used = used;
  a.last_products = products;
  if (!coupled) {
    if (!p->auto_ml && 2 * products >= budget) p->auto_ml = true;
    return;
  }
  const bool cost_rule = options().auto_cost_rule != 0;
  if (!p->auto_ml) {  // the solve ran block-Jacobi
    a.state = 0;
    a.uj = auto_units_jacobi(p);
    a.jac_units += (long long)kAutoUnitsJacobi * products;  // (the set-up is wall time: paid back in solo units)
    const bool binds = 2 * products >= budget;
    const bool paid = cost_rule && products >= kAutoMinProducts && a.jac_units >= ((long long)kAutoSetupUnits << a.backoff);
    if (binds || paid) {
      // (the plan -- host aggregation, once per block pattern -- is only looked for when the rule wants it)
      const bool add = cost_rule && !p->ml_user_ks && additive_available(p);
      if (add) a.ua = auto_units_additive(p);
      // a trial that cannot win is not run: even at kAutoMinProducts the additive solve would cost more than this one did
      const bool hopeless = add && !binds && (long long)a.ua * kAutoMinProducts >= (long long)a.uj * products;
      if (hopeless) {
        a.jac_units = 0;
        a.backoff = std::min(a.backoff + 1, 6);
      } else if (binds || add) {
        p->auto_ml = true;
        if (add) {
          a.state = 1;
          a.ref = products;
          a.switches += 1;
        }
      }
    }
    return;
  }
  if (a.state == 0) {  // a multilevel choice outside the cost rule (V-cycle blocks, dpgo_problem_auto_state): budget hysteresis
    if (10 * products <= budget) p->auto_ml = false;
    return;
  }
  a.ua = auto_units_additive(p);
  // on trial: strictly cheaper than the reference solve; once accepted: handed back only when 15 % dearer (the two are
  // within a few per cent of each other on interior blocks of a chain partition -- no flapping)
  const long long cost = (long long)a.ua * products * 100, ref = (long long)a.uj * a.ref * (a.state == 2 ? 115 : 100);
  if (cost < ref) {
    a.state = 2;
  } else {  // no cheaper than block-Jacobi on this block in this phase of the run: hand back, try again later
    p->auto_ml = false;
    a.state = 0;
    a.jac_units = 0;
    a.backoff = std::min(a.backoff + 1, 6);
  }
}

// phase: RUN_FULL = the whole solve, synchronously.  RUN_BEGIN = enqueue only: if the solve is a one-launch solve
// (k_rtr_persist) the call returns with the launch, its commit kernel and the read-backs in flight (p->pending.launched);
// otherwise the solve runs to completion right here.  RUN_END = collect what RUN_BEGIN left in flight (waits for the
// stream, reads the state record, falls back to the multi-launch scheme after a time-out exactly as the synchronous call).
int run_optimize(dpgo_problem_s* p, const dpgo_ropt_params* prm, dpgo_ropt_result* res, int phase) {
  // X is in p->x1 on entry and on exit.  src/QuadraticOptimizer.cpp:26-48.
  auto t0 = std::chrono::steady_clock::now();
  Counters cnt;
  std::memset(res, 0, sizeof(*res));
  res->tCGStatus = DPGO_TCG_MAXITER;
  const bool resume = phase == RUN_END;
  if (p->hctrl && !resume) std::memset(p->hctrl, 0, sizeof(PersistCtrl));
  struct SlotGuard {  // the resident-slot reservation of the persistent kernel lives as long as the solve
    dpgo_problem_s* p;
    bool armed;
    ~SlotGuard() {
      if (armed) persist_release(p);
    }
  } slot_guard{p, true};
  dpgo_ropt_params resolved = resume ? p->pending.resolved : *prm;  // DPGO_PRECOND_AUTO -> what this handle currently runs
  if (!resume) {
    if (prm->precond == DPGO_PRECOND_AUTO) p->auto_decide();
    if (prm->precond == DPGO_PRECOND_AUTO)  // (the multilevel choice: the additive form wherever its persistent kernel runs)
      resolved.precond = (p->auto_ml && prm->method == DPGO_METHOD_RTR)
                             ? ((additive_available(p) && !p->ml_user_ks) ? DPGO_PRECOND_ADDITIVE : DPGO_PRECOND_MULTILEVEL)
                             : DPGO_PRECOND_BLOCK_JACOBI;
  }
  const bool is_auto = resume ? p->pending.is_auto : prm->precond == DPGO_PRECOND_AUTO;
  if (resume) t0 = p->pending.t0;
  prm = &resolved;
  const double* dinv = nullptr;
  if (resume) {
    dinv = p->pending.dinv;
  } else
  if (prm->precond == DPGO_PRECOND_BLOCK_JACOBI) {
    CHK(build_dinv(p, prm->precond_shift));
    dinv = p->dinv;
  } else if (prm->precond == DPGO_PRECOND_MULTILEVEL) {
    // built lazily for the current Q, like the reference's factor (src/PoseGraph.cpp:582-586)
    CHK(ml_ensure(p, prm->precond_shift));
    dinv = p->dinv;  // the smoother's block-Jacobi factors (same shift)
  } else if (prm->precond == DPGO_PRECOND_ADDITIVE) {
    if (prm->method != DPGO_METHOD_RTR) return fail(DPGO_ERR_UNSUPPORTED, "the additive preconditioner exists inside the tCG loop only");
    if (!additive_split_of(p) && !additive_plan(p).split)
      return fail(DPGO_ERR_UNSUPPORTED, "additive preconditioner: block too large (at most 256 aggregates of one workgroup tile, " +
                                            std::to_string(ml_tile(p->b, 1)) + " poses)");
    CHK(ml_ensure(p, prm->precond_shift, /*additive=*/true));
    dinv = p->dinv;
  } else if (prm->precond != DPGO_PRECOND_NONE) {
    return fail(DPGO_ERR_INVALID, "unknown preconditioner");
  }
  if (!resume) {
    p->loop_extra_bytes = 0;
    if (prm->precond == DPGO_PRECOND_MULTILEVEL && !p->ml.empty()) {
      const size_t nd = (size_t)p->ml_lda;
      size_t bytes = nd * nd * (size_t)(p->ml_coarse_bits / 8) / (p->ml_use_dense_sym() ? 2 : 1);
      const auto& L0 = p->ml[0];
      bytes += (size_t)L0.AP.nnzb * (sizeof(double) * p->b * p->b + sizeof(int32_t)) + sizeof(double) * (size_t)p->n * p->b * p->b;
      bytes += 3 * p->vec_bytes();
      p->loop_extra_bytes = bytes;
    }
    CHK(resolve_tcg_storage(p));
    if (prm->precond == DPGO_PRECOND_MULTILEVEL) CHK(ml_ops32_ensure(p));  // (opt-in fp32 operator copies of the cycle)
  }
  // ---- blocks in the latency regime: the whole solve is ONE persistent launch (k_rtr_persist) and one read-back.  The
  // single-iteration radius-shrink mode (:80-99) and the polling mode keep the multi-launch scheme.
  const bool add = prm->precond == DPGO_PRECOND_ADDITIVE;
  // (the in-kernel all-reduce tags its granules with salt | step, the step counter in the low 20 bits: a solve whose
  // parameters allow more reductions than that -- at most 3 per tCG iteration + 4 per outer iteration + 1 -- keeps the
  // multi-launch scheme)
  const bool epochs_fit = (long long)std::max(1, prm->RTR_iterations) * (3LL * std::max(0, prm->RTR_tCG_iterations) + 4) + 1 < (1LL << 20);
  if (resume || (prm->method == DPGO_METHOD_RTR && prm->RTR_iterations != 1 && prm->tcg_poll_interval <= 0 && p->persist && epochs_fit &&
                 !p->persist_failed_once && (add || prm->precond == DPGO_PRECOND_BLOCK_JACOBI || prm->precond == DPGO_PRECOND_NONE))) {
    bool used = resume;
    if (!resume) CHK(launch_rtr_persistent(p, prm, dinv, &used, add));
    if (used) {
      if (!resume) {
        // (no read-back commands: k_persist_commit has written the state record and the control block into hstate / hctrl)
        if (phase == RUN_BEGIN) {  // everything of the solve is enqueued: the caller collects it with RUN_END
          auto& pd = p->pending;
          pd.launched = true;
          pd.resolved = resolved;
          pd.is_auto = is_auto;
          pd.dinv = dinv;
          pd.t0 = t0;
          slot_guard.armed = false;  // (the reservation is released by the collecting call)
          return DPGO_OK;
        }
      }
      HIPC(hipStreamSynchronize(p->stream));
      persist_report(p);
      // (k_persist_commit, which ran behind the solve, saw the same two words: with a poisoned record OR a raised time-out
      // flag -- some participant gave up, however late -- the caller's iterate has not been touched)
      if (p->hstate->rtr_stop != kPersistPoison && !p->hctrl->error) {
        const DevState& h = *p->hstate;
        res->fInit = h.fInit;
        res->gradNormInit = h.gnInit;
        res->fOpt = h.f1;
        res->gradNormOpt = h.ngf;
        res->tCGStatus = h.outer_iter > 0 ? h.tcg_status : DPGO_TCG_MAXITER;
        res->rtr_iterations = h.outer_iter;
        res->rtr_accepted = h.n_accept;
        res->latest_step_accepted = h.accepted_last;
        res->tcg_iterations = h.n_hess;
        res->precond_used = prm->precond;
        res->spmm_count = 1 + 2 * h.outer_iter + h.n_hess;
        if (is_auto) auto_update(p, prm, prm->precond, h.n_hess);
        res->success = 1;  // :44
        res->elapsedMs = 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        return DPGO_OK;
      }
      // a time-out (the launch's workgroups were not all resident: another process on the device): the caller's iterate is
      // untouched; this handle stops using the kernel and the solve runs on the multi-launch scheme
      p->persist_failed_once = true;
      p->pctrl_dirty = true;
      persist_release(p);
      if (options().persist_verbose)
        std::fprintf(stderr, "dpgo_hip: persistent solve timed out; this handle continues with the multi-launch scheme\n");
    }
  }
  // statistics before optimisation (:28-29) -- one fused pass: f, rgrad, S
  CHK(launch_grad(p, p->x1, p->g1, p->S1, nullptr, nullptr, prm->method == DPGO_METHOD_RTR && p->tcg_sym && outer_sym_enabled()));
  cnt.spmm += 1;
  CHK(launch_rtr_begin(p, prm->gradnorm_tol, prm->RTR_initial_radius, 5.0 * prm->RTR_initial_radius,
                       prm->RTR_tCG_iterations, prm->accept_tiny_decrease));
  // The initial statistics are read back together with the final ones when the solve is fed just-in-time (one
  // synchronisation per call instead of two): an iterate that already meets the tolerance (:57-59) makes the first tCG
  // launch publish rtr_stop, which ends the loop below before anything is changed.
  const bool deferred = prm->method == DPGO_METHOD_RTR && prm->RTR_iterations != 1 && prm->tcg_poll_interval <= 0;
  if (!deferred) {
    CHK(poll_state(p));
    res->fInit = p->hstate->fInit;
    res->gradNormInit = p->hstate->gnInit;
  } else {
    p->hstate->rtr_stop = 0;
  }
  int n_hess_total = 0;
  int shrink_tries = 0;

  if (prm->method == DPGO_METHOD_RTR) {
    // trustRegion(): src/QuadraticOptimizer.cpp:50-108
    if (!p->hstate->rtr_stop) {  // :57-59 early-out
      if (prm->RTR_iterations == 1) {  // :80-99 shrink the radius until the step is accepted
        double radius = prm->RTR_initial_radius;
        int total_steps = 0;
        while (true) {
          shrink_tries += 1;
          p->hstate->Delta = radius;
          p->hstate->Delta_max = radius;
          p->hstate->outer_iter = 0;
          CHK(push_state(p));
          p->saw_rtr_stop = false;
          CHK(rtr_outer_iteration(p, prm, dinv, cnt, true));
          if (p->hstate->accepted_last) break;
          if (total_steps > 10) break;  // "Too many RTR rejections. Returning initial guess." (x1 untouched)
          radius /= 4.0;
          total_steps++;
        }
      } else {
        const bool polling = prm->tcg_poll_interval > 0;
        p->saw_rtr_stop = false;
        for (int it = 0; it < prm->RTR_iterations; ++it) {
          CHK(rtr_outer_iteration(p, prm, dinv, cnt, polling));
          if (polling ? (p->hstate->rtr_stop != 0) : p->saw_rtr_stop) break;
          const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
          if (prm->time_bound_s > 0 && el > prm->time_bound_s) break;  // Solver.TimeBound (:78)
        }
        CHK(poll_state(p));
        if (deferred) {
          res->fInit = p->hstate->fInit;
          res->gradNormInit = p->hstate->gnInit;
        }
      }
      res->tCGStatus = p->hstate->tcg_status;
      res->rtr_iterations = (prm->RTR_iterations == 1) ? shrink_tries : p->hstate->outer_iter;
      res->rtr_accepted = p->hstate->n_accept;
      res->latest_step_accepted = p->hstate->accepted_last;
      n_hess_total = p->hstate->n_hess;
    }
    res->fOpt = p->hstate->f1;
    res->gradNormOpt = p->hstate->ngf;
  } else if (prm->method == DPGO_METHOD_RGD) {
    // gradientDescent(): src/QuadraticOptimizer.cpp:110-137 (one fixed-step preconditioned step)
    const double* step = p->g1;
    if (prm->RGD_use_preconditioner) {
      if (prm->precond == DPGO_PRECOND_MULTILEVEL)
        CHK(launch_ml_apply(p, p->x1, p->g1, p->z));
      else
        CHK(launch_precond(p, p->x1, p->g1, dinv, p->z));
      step = p->z;
    }
    CHK(launch_retract(p, p->x1, step, -prm->RGD_stepsize, p->x2, nullptr));
    HIPC(hipMemcpyAsync(p->x1, p->x2, p->vec_bytes(), hipMemcpyDeviceToDevice, p->stream));
    CHK(launch_grad(p, p->x1, p->g1, p->S1, nullptr));
    cnt.spmm += 1;
    CHK(launch_rtr_begin(p, prm->gradnorm_tol, prm->RTR_initial_radius, 5.0 * prm->RTR_initial_radius,
                         prm->RTR_tCG_iterations, prm->accept_tiny_decrease));
    CHK(poll_state(p));
    res->fOpt = p->hstate->f1;
    res->gradNormOpt = p->hstate->ngf;
  } else {
    return fail(DPGO_ERR_INVALID, "unknown method");
  }
  res->tcg_iterations = n_hess_total;
  res->precond_used = (prm->precond == DPGO_PRECOND_ADDITIVE && cnt.vcycle_for_additive) ? DPGO_PRECOND_MULTILEVEL : prm->precond;
  if (is_auto && prm->method == DPGO_METHOD_RTR) auto_update(p, prm, res->precond_used, n_hess_total);
  res->spmm_count = cnt.spmm + n_hess_total;
  res->success = 1;  // :44 (set unconditionally after a solve)
  res->elapsedMs = 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return DPGO_OK;
}

// The two kernels of the tCG loop run as persistent grids: one workgroup per resident slot (occupancy x CUs).
// More workgroups than slots only add prologues (state record + partial-sum reduction) and a ragged second
// round: 100k poses, same box: caps 1024/1024 -> 70.7 us per tCG iteration, 512/768 (= the resident counts of
// k_tcg_update / k_tcg_hess at 203 / 164 VGPRs) -> 64.7 us.  The other kernels keep the family's cap.
template <class K>
int resident_blocks(K kernel, int* out) {
  int per_cu = 0;
  HIPC(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kBlock, 0));
  int dev = 0, cus = 0;
  HIPC(hipGetDevice(&dev));
  HIPC(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  *out = std::max(1, std::min(kPartialCap, per_cu * cus));
  return DPGO_OK;
}
int tune_launch_caps(dpgo_problem_s* p) {
  DISPATCH(p->d, p->r, {
    if constexpr (Span<D, R, 1>::kOk) {
      CHK(resident_blocks((k_tcg_update_span<D, R, 0>), &p->cap_u));
      CHK(resident_blocks((k_tcg_update_span<D, R, 1>), &p->cap_u_ml));
      if (p->split == 4)
        CHK(resident_blocks(k_tcg_hess_span<D, R, 4>, &p->cap_h));
      else if (p->split == 2)
        CHK(resident_blocks(k_tcg_hess_span<D, R, 2>, &p->cap_h));
      else
        CHK(resident_blocks(k_tcg_hess_span<D, R, 1>, &p->cap_h));
      if (options().hess_dma == 1)
        CHK(resident_blocks((k_tcg_hess_sym_dma<D, R, 1, 2, 4, 1>), &p->cap_hs));
      else if (options().hess_dma == 2)
        CHK(resident_blocks((k_tcg_hess_sym_dma<D, R, 1, 3, 2, 0>), &p->cap_hs));
      else
        CHK(resident_blocks(k_tcg_hess_sym<D, R, 1>, &p->cap_hs));
    } else {
      CHK(resident_blocks(k_tcg_update<D, R>, &p->cap_u));
      p->cap_u_ml = p->cap_u;  // (one kernel for both modes)
      if (p->split == 4)
        CHK(resident_blocks(k_tcg_hess<D, R, 4>, &p->cap_h));
      else if (p->split == 2)
        CHK(resident_blocks(k_tcg_hess<D, R, 2>, &p->cap_h));
      else
        CHK(resident_blocks(k_tcg_hess<D, R, 1>, &p->cap_h));
    }
  });
  DISPATCH(p->d, p->r, {
    if (p->split == 4) {
      CHK(resident_blocks(k_ml_restrict<D, R, 4, BsrDev>, &p->cap_restrict));
      CHK(resident_blocks(k_ml_post_ap<D, R, 4>, &p->cap_post));
    } else if (p->split == 2) {
      CHK(resident_blocks(k_ml_restrict<D, R, 2, BsrDev>, &p->cap_restrict));
      CHK(resident_blocks(k_ml_post_ap<D, R, 2>, &p->cap_post));
    } else {
      // one pose per D+1 lanes: the smallest count over the variants a cycle may launch (plain / symmetric storage, fp64 /
      // fp32 copies) -- a grid sized for a variant with more resident workgroups than the launched one leaves part of it
      // waiting for a slot
      int c = 0;
      CHK(resident_blocks(k_ml_restrict<D, R, 1, BsrDev>, &p->cap_restrict));
      CHK(resident_blocks(k_ml_restrict<D, R, 1, BsrSymDev, double, double>, &c));
      p->cap_restrict = std::min(p->cap_restrict, c);
      CHK(resident_blocks(k_ml_restrict<D, R, 1, BsrSymDev32, float, float>, &c));
      p->cap_restrict = std::min(p->cap_restrict, c);
      CHK(resident_blocks(k_ml_restrict<D, R, 1, BsrSymDev32, float, double>, &c));
      p->cap_restrict = std::min(p->cap_restrict, c);
      CHK(resident_blocks(k_ml_post_ap<D, R, 1>, &p->cap_post));
      CHK(resident_blocks(k_ml_post_ap<D, R, 1, float, float>, &c));
      p->cap_post = std::min(p->cap_post, c);
      CHK(resident_blocks(k_ml_post_ap<D, R, 1, float, double>, &c));
      p->cap_post = std::min(p->cap_post, c);
    }
  });
  if (options().grid_ml > 0) p->cap_restrict = p->cap_post = std::min(kPartialCap, options().grid_ml);
  // tuning knobs (any value up to the partial-sum capacity is valid)
  if (options().grid_update > 0) p->cap_u = p->cap_u_ml = std::min(kPartialCap, options().grid_update);
  if (options().grid_hess > 0) p->cap_h = std::min(kPartialCap, options().grid_hess);
  if (options().grid_hess_sym > 0) p->cap_hs = std::min(kPartialCap, options().grid_hess_sym);
  DISPATCH(p->d, p->r, {
    int c = kMaxGrid;
    CHK(resident_blocks((k_spmm_sym<D, R, 1>), &c));
    p->cap_spmm_sym = std::min(kMaxGrid, c);
    int cg = kMaxGrid, ch = kMaxGrid;
    CHK(resident_blocks((k_grad<D, R, 1, BsrSymDev>), &cg));
    CHK(resident_blocks((k_hess<D, R, 1, BsrSymDev>), &ch));
    p->cap_outer_sym = std::min(kPartialCap, std::min(cg, ch));
  });
  if (options().grid_outer_sym > 0) p->cap_outer_sym = std::min(kPartialCap, options().grid_outer_sym);
  if (options().grid_spmm_sym > 0) p->cap_spmm_sym = std::min(4096, options().grid_spmm_sym);
  return DPGO_OK;
}

// One-launch solve: on by size -- every block the kernel can hold (two 64-pose tiles on each of 256 workgroups: 32 768 poses
// in 3-D; measured per Hessian-vector product against the multi-launch scheme: 625 poses 6.6 / 13.7 us, sphere2500 6.5 /
// 15.4, 6 250 9.9 / 17.2, 12.5k slab 10.5 / 19.7, 25k 18.7 / 26.5).  DPGO_PERSIST_MAX_POSES lowers the limit, DPGO_PERSIST=0/1
// overrides.
int tune_persist(dpgo_problem_s* p) {
  const int max_poses = options().persist_max_poses > 0 ? options().persist_max_poses : 1 << 30;
  const bool fits = persist_geometry(p, persist_capacity(p->device)).wgs > 0;
  bool on = fits && p->n <= max_poses;
  if (options().persist >= 0) on = fits && options().persist != 0;
  p->persist = on;
  return DPGO_OK;
}

}  // namespace dpgo_host

extern "C" {


int dpgo_problem_auto_state(dpgo_problem_t p, int* use_multilevel) {
  if (!p || !use_multilevel) return fail(DPGO_ERR_INVALID, "null pointer");
  if (*use_multilevel >= 0) {
    p->auto_ml = *use_multilevel != 0;
    p->auto_decided = true;
    p->auto_cost = dpgo_problem_s::AutoCost();  // (a choice made from outside is followed by the budget hysteresis)
  } else {
    if (*use_multilevel == -2) p->auto_decided = false;  // back to the decision a fresh handle takes for this problem
    p->auto_decide();
  }
  *use_multilevel = p->auto_ml ? 1 : 0;
  return DPGO_OK;
}


int dpgo_problem_auto_info(dpgo_problem_t p, int* state, long long* jacobi_units, int* reference_products, int* switches,
                           int* backoff, int* units_jacobi, int* units_additive) {
  if (!p) return fail(DPGO_ERR_INVALID, "null handle");
  const auto& a = p->auto_cost;
  if (units_jacobi) *units_jacobi = a.uj;
  if (units_additive) *units_additive = a.ua;
  if (state) *state = a.state;
  if (jacobi_units) *jacobi_units = a.jac_units;
  if (reference_products) *reference_products = a.ref;
  if (switches) *switches = a.switches;
  if (backoff) *backoff = a.backoff;
  return DPGO_OK;
}


int dpgo_auto_rule_constants(int* units_jacobi, int* units_additive, int* setup_units, int* min_products) {
  if (units_jacobi) *units_jacobi = kAutoUnitsJacobi;
  if (units_additive) *units_additive = kAutoUnitsAdditive;
  if (setup_units) *setup_units = kAutoSetupUnits;
  if (min_products) *min_products = kAutoMinProducts;
  return DPGO_OK;
}


int dpgo_optimize(dpgo_problem_t p, const dpgo_ropt_params* params, const double* X0, double* Xopt,
                  dpgo_ropt_result* result) {
  CHK(check_ready(p));
  if (!params || !X0 || !Xopt || !result) return fail(DPGO_ERR_INVALID, "null pointer");
  CHK(h2d(p, p->x1, X0));
  CHK(run_optimize(p, params, result));
  return d2h(p, Xopt, p->x1);
}


int dpgo_optimize_device(dpgo_problem_t p, const dpgo_ropt_params* params, double* X_dev, dpgo_ropt_result* result) {
  CHK(check_ready(p));
  if (!params || !X_dev || !result) return fail(DPGO_ERR_INVALID, "null pointer");
  // the caller's buffer IS the iterate for the duration of the call (no copies in or out): accepted steps are written
  // into it by k_rtr_update, rejected ones leave it untouched
  double* own = p->x1;
  p->x1 = X_dev;
  const int rc = run_optimize(p, params, result);
  p->x1 = own;
  if (rc != DPGO_OK) return rc;
  HIPC(hipStreamSynchronize(p->stream));
  return DPGO_OK;
}


int dpgo_warning_count(void) { return g_warnings.load(); }

int dpgo_optimize_device_begin(dpgo_problem_t p, const dpgo_ropt_params* params, double* X_dev,
                               const double* nbr_tiles_dev) {
  CHK(check_ready(p));
  if (!params || !X_dev) return fail(DPGO_ERR_INVALID, "null pointer");
  if (p->pending.active) return fail(DPGO_ERR_STATE, "a solve of this handle is already in flight (dpgo_optimize_device_end)");
  if (nbr_tiles_dev) {  // PGOAgent::updateX: G from the neighbours' public poses first (same stream)
    if (!p->C.rowptr) return fail(DPGO_ERR_STATE, "G coupling not set");
    CHK(launch_spmm(p, p->C, nbr_tiles_dev, p->G0, p->G));
    p->has_G = true;
  }
  auto& pd = p->pending;
  pd = dpgo_problem_s::Pending();
  pd.own_x1 = p->x1;
  p->x1 = X_dev;
  p->persist_stream_ordered = true;
  dpgo_ropt_result tmp;
  const int rc = run_optimize(p, params, &tmp, RUN_BEGIN);
  p->persist_stream_ordered = false;
  if (rc != DPGO_OK || !pd.launched) {  // failed, or the solve is not a one-launch solve and has run to completion
    p->x1 = pd.own_x1;
    if (rc != DPGO_OK) return rc;
    pd.result = tmp;
  }
  pd.active = true;
  return DPGO_OK;
}


int dpgo_optimize_device_end(dpgo_problem_t p, dpgo_ropt_result* result) {
  if (!p || !result) return fail(DPGO_ERR_INVALID, "null pointer");
  CHK(set_device(p));
  auto& pd = p->pending;
  if (!pd.active) return fail(DPGO_ERR_STATE, "no solve in flight (dpgo_optimize_device_begin)");
  pd.active = false;
  if (!pd.launched) {
    *result = pd.result;
    return DPGO_OK;
  }
  pd.launched = false;
  const int rc = run_optimize(p, &pd.resolved, result, RUN_END);
  p->x1 = pd.own_x1;
  if (rc != DPGO_OK) return rc;
  HIPC(hipStreamSynchronize(p->stream));
  return DPGO_OK;
}


// ---- several agents of one process updated concurrently (same-colour agents of a parallel RBCD sweep) ----
namespace {
// Host threads that feed the just-in-time tCG loops of several handles at once: one worker per concurrently solved handle,
// created on first use and kept (a solve is a few milliseconds; creating threads per sweep would show).
class FeedPool {
 public:
  static FeedPool& get() {
    static FeedPool* pool = new FeedPool();  // never destroyed: workers may outlive static destructors
    return *pool;
  }
  // runs job(0..count-1): job(0) on the calling thread, the others on workers; returns when all are done
  void run(int count, const std::function<void(int)>& job) {
    std::unique_lock<std::mutex> call(call_mu_);  // one batch at a time
    {
      std::lock_guard<std::mutex> lk(mu_);
      while ((int)workers_.size() < count - 1) {
        const int id = (int)workers_.size();
        workers_.emplace_back([this, id] { loop(id); });
        workers_.back().detach();
      }
      job_ = &job;
      count_ = count;
      pending_ = count - 1;
      epoch_ += 1;
    }
    cv_.notify_all();
    job(0);
    std::unique_lock<std::mutex> lk(mu_);
    done_.wait(lk, [this] { return pending_ == 0; });
    job_ = nullptr;
  }

 private:
  void loop(int id) {
    unsigned long long seen = 0;
    while (true) {
      const std::function<void(int)>* job = nullptr;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return epoch_ != seen; });
        seen = epoch_;
        if (id + 1 < count_) job = job_;
      }
      if (job) {
        (*job)(id + 1);
        std::lock_guard<std::mutex> lk(mu_);
        if (--pending_ == 0) done_.notify_all();
      }
    }
  }
  std::mutex call_mu_, mu_;
  std::condition_variable cv_, done_;
  std::vector<std::thread> workers_;
  const std::function<void(int)>* job_ = nullptr;
  int count_ = 0, pending_ = 0;
  unsigned long long epoch_ = 0;
};

// Runs body(k) for every handle on its OWN stream, ordered after `after_stream`; the handles' previous streams are
// restored afterwards.  Returns the first failure (its message becomes this thread's dpgo_last_error).
static int run_many(int count, const dpgo_problem_t* handles, void* after_stream, const std::function<int(int)>& body) {
  if (count <= 0) return DPGO_OK;
  if (!handles) return fail(DPGO_ERR_INVALID, "null handle array");
  for (int k = 0; k < count; ++k) {
    CHK(check_ready(handles[k]));
    if (handles[k]->device != handles[0]->device) return fail(DPGO_ERR_INVALID, "handles on different devices");
    for (int q = 0; q < k; ++q)
      if (handles[q] == handles[k]) return fail(DPGO_ERR_INVALID, "a handle appears twice");
  }
  {  // ROCm maps HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4): more concurrently solved handles than
     // that still complete, but queue behind each other -- say so once instead of silently serialising
    static std::atomic<bool> warned{false};
    const char* e = std::getenv("GPU_MAX_HW_QUEUES");
    const int queues = (e && std::atoi(e) > 0) ? std::atoi(e) : 4;
    if (count > queues && !warned.exchange(true)) {
      g_warnings.fetch_add(1);
      std::fprintf(stderr,
                   "dpgo_hip: warning: %d handles are solved concurrently but GPU_MAX_HW_QUEUES is %d: their streams share %d "
                   "hardware queues and partly serialise; set GPU_MAX_HW_QUEUES >= %d in the environment before the HIP runtime "
                   "initialises (bench.py does)\n",
                   count, queues, queues, count);
    }
  }
  std::vector<hipStream_t> prev(count);
  for (int k = 0; k < count; ++k) prev[k] = handles[k]->stream;  // (all of them first: restored below whatever fails)
  hipEvent_t ev = nullptr;
  HIPC(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  hipError_t e = hipEventRecord(ev, (hipStream_t)after_stream);
  for (int k = 0; k < count && e == hipSuccess; ++k) {
    if (prev[k] != handles[k]->own_stream) e = hipStreamSynchronize(prev[k]);  // earlier work of the handle itself
    handles[k]->stream = handles[k]->own_stream;
    handles[k]->persist_share = count;  // persistent tCG launches: prefer the layout with the fewest resident slots
    if (e == hipSuccess) e = hipStreamWaitEvent(handles[k]->own_stream, ev, 0);
  }
  std::vector<int> rc(count, DPGO_OK);
  std::vector<std::string> msg(count);
  if (e == hipSuccess) {
    FeedPool::get().run(count, [&](int k) {
      int r = (hipSetDevice(handles[k]->device) == hipSuccess) ? body(k) : fail(DPGO_ERR_HIP, "hipSetDevice failed");
      if (r == DPGO_OK && hipStreamSynchronize(handles[k]->stream) != hipSuccess)
        r = fail(DPGO_ERR_HIP, "hipStreamSynchronize failed");
      rc[k] = r;
      if (r != DPGO_OK) msg[k] = g_err;  // thread-local message of the worker
    });
  }
  for (int k = 0; k < count; ++k) {
    handles[k]->stream = prev[k];
    handles[k]->persist_share = 1;
  }
  (void)hipEventDestroy(ev);
  if (e != hipSuccess) return fail(DPGO_ERR_HIP, std::string("stream ordering of the concurrent update: ") + hipGetErrorString(e));
  for (int k = 0; k < count; ++k)
    if (rc[k] != DPGO_OK) return fail(rc[k], "handle " + std::to_string(k) + ": " + msg[k]);
  return DPGO_OK;
}
}  // namespace


int dpgo_optimize_device_many(int count, const dpgo_problem_t* handles, const dpgo_ropt_params* params,
                              double* const* X_dev, const double* const* nbr_tiles_dev, void* after_stream,
                              dpgo_ropt_result* results) {
  if (count <= 0) return DPGO_OK;
  if (!params || !X_dev || !results) return fail(DPGO_ERR_INVALID, "null pointer");
  for (int k = 0; k < count; ++k)
    if (!X_dev[k]) return fail(DPGO_ERR_INVALID, "null iterate");
  return run_many(count, handles, after_stream, [&](int k) -> int {
    dpgo_problem_s* p = handles[k];
    if (nbr_tiles_dev && nbr_tiles_dev[k]) {  // PGOAgent::updateX: G from the neighbours' public poses first
      if (!p->C.rowptr) return fail(DPGO_ERR_STATE, "G coupling not set");
      CHK(launch_spmm(p, p->C, nbr_tiles_dev[k], p->G0, p->G));
      p->has_G = true;
    }
    double* own = p->x1;
    p->x1 = X_dev[k];
    const int rc = run_optimize(p, params, &results[k]);
    p->x1 = own;
    return rc;
  });
}


int dpgo_problem_eval_terms_device_many(int count, const dpgo_problem_t* handles, const double* const* X_dev,
                                        const double* const* nbr_tiles_dev, void* after_stream, double* terms) {
  if (count <= 0) return DPGO_OK;
  if (!X_dev || !terms) return fail(DPGO_ERR_INVALID, "null pointer");
  for (int k = 0; k < count; ++k)
    if (!X_dev[k]) return fail(DPGO_ERR_INVALID, "null iterate");
  return run_many(count, handles, after_stream, [&](int k) -> int {
    dpgo_problem_s* p = handles[k];
    if (nbr_tiles_dev && nbr_tiles_dev[k]) {
      if (!p->C.rowptr) return fail(DPGO_ERR_STATE, "G coupling not set");
      CHK(launch_spmm(p, p->C, nbr_tiles_dev[k], p->G0, p->G));
      p->has_G = true;
    }
    CHK(launch_grad(p, X_dev[k], nullptr, nullptr, nullptr));
    CHK(launch_rtr_begin(p, 0.0, 1.0, 1.0, 0, 0));
    CHK(poll_state(p));
    terms[3 * k + 0] = p->hstate->xqx;
    terms[3 * k + 1] = p->hstate->xg;
    terms[3 * k + 2] = p->hstate->ngf * p->hstate->ngf;
    return DPGO_OK;
  });
}


#ifdef DPGO_TIMELINE
int dpgo_debug_timeline(long long* out /* [2][16] */) {
  HIPC(hipDeviceSynchronize());
  HIPC(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_timeline), sizeof(long long) * 32));
  return DPGO_OK;
}
int dpgo_debug_timeline_tiles(long long* out /* [3][64] */) {
  HIPC(hipDeviceSynchronize());
  HIPC(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tl_tiles), sizeof(long long) * 3 * 64));
  return DPGO_OK;
}
#endif


int dpgo_problem_persistent_info(dpgo_problem_t p, int* enabled, int* workgroups, int* last_members, int* last_iterations,
                                 int* last_layout) {
  if (!p) return fail(DPGO_ERR_INVALID, "null handle");
  if (enabled) *enabled = (p->persist && !p->persist_failed_once) ? 1 : 0;
  if (workgroups) *workgroups = p->persist_wgs;
  if (last_members) *last_members = p->hctrl ? (int)p->hctrl->members : 0;
  if (last_iterations) *last_iterations = p->hctrl ? (int)p->hctrl->iters : 0;
  if (last_layout) *last_layout = (p->hctrl && p->hctrl->members) ? p->persist_split * 16 + p->persist_mt : 0;
  return DPGO_OK;
}


int dpgo_problem_persistent_phases(dpgo_problem_t p, double us_per_iteration[4], int* iterations) {
  if (!p || !us_per_iteration) return fail(DPGO_ERR_INVALID, "null handle / pointer");
  const double it = p->hctrl ? (double)p->hctrl->ticks[4] : 0.0;
  for (int k = 0; k < 4; ++k)  // (100 MHz wall-clock ticks of participant 0, summed over the iterations after the first)
    us_per_iteration[k] = (p->hctrl && it > 0.0 && p->hctrl->members) ? 0.01 * (double)p->hctrl->ticks[k] / it : 0.0;
  if (iterations) *iterations = (p->hctrl && p->hctrl->members) ? (int)p->hctrl->iters : 0;
  return DPGO_OK;
}


int dpgo_problem_set_persistent(dpgo_problem_t p, int enable) {
  if (!p) return fail(DPGO_ERR_INVALID, "null handle");
  if (enable && persist_geometry(p, persist_capacity(p->device)).wgs <= 0)
    return fail(DPGO_ERR_UNSUPPORTED, "block too large for the persistent tCG kernel (at most 2 tiles on each of 256 workgroups)");
  p->persist = enable != 0;
  if (enable) p->persist_failed_once = false;
  return DPGO_OK;
}


int dpgo_problem_tcg_kernel_info(dpgo_problem_t p, int* symmetric, int* split, int* stream_nt) {
  CHK(check_ready(p));
  CHK(resolve_tcg_storage(p));
  if (symmetric) *symmetric = p->tcg_sym ? 1 : 0;
  if (split) *split = p->tcg_sym ? 1 : p->split;
  if (stream_nt) *stream_nt = (p->stream_nt && (p->tcg_sym || p->split == 1)) ? 1 : 0;
  return DPGO_OK;
}

}  // extern "C"
