#!/usr/bin/env python
"""bench.py -- RBCD iterations/sec of the MI355X-native local solver + Q*X roofline.

    python bench.py --gpus N --steps K --warmup W [--workload grid100k|sphere2500|grid:NXxNYxNZ]

One "step" = one RBCD iteration from a fixed, settled iterate (identical full work every step; see main()):
every agent runs QuadraticOptimizer::optimize once (RTR, 3 outer
iterations x <=50 tCG, Delta0 = 100, tol 1e-2: the reference defaults, include/DPGO/DPGO_types.h:53-61)
on its block, with the library's default preconditioner selection ("auto").  N = 1: a single agent owns the whole graph.
N > 1: the graph is cut into N contiguous blocks (examples/MultiRobotExample.cpp:71-88), one agent per
GPU / process; agents of one colour update in parallel, then the other colour (two-colour RBCD),
with the public-pose exchange over RCCL point-to-point.  Total work is fixed as N grows ("strong").

Workload at N = 1: the synthetic 100k-pose 3-D grid of BASELINE.json (configs[3], the configuration the
HBM-roofline target is quoted on; it fits one GPU).  Inputs are resident in HBM before the timed region.

Prints ONE JSON line (rank 0).  `roofline` describes the tCG-step kernel THE TIMED LOOP LAUNCHES (k_tcg_hess_span on the
plain block-CSR arrays for every BASELINE configuration; k_tcg_hess_sym where the size switch selects the symmetric
storage): `frac` is its HBM-only figure (every operand cycling through > 256 MB of private copies), `warm`
the back-to-back one the Infinity-Cache-resident loop sees; the other storage's figures stand beside it
(`symmetric_storage` / `plain_storage`, with the fraction on the bytes that storage really moves);
`roofline.kernels` lists the other kernels of one preconditioned tCG iteration.  `products_per_step` and
`time_to_tolerance_ms` are top-level: "it/s" alone does not say how much a step does; `hierarchy_setup_ms` is the
once-per-Q cost of the preconditioner those solves ran (`time_to_tolerance_incl_setup_ms` = both).  `cpu_baseline` (rank
0, N = 1 only) = the reference configuration of the workload on the host cores (the graph cut into 8 agents, one core
each, exact sparse factor of Q_a + 0.1 I, ONE two-colour sweep from the SETTLED iterate every timed step restores;
`factorisation_seconds` beside it) and, in `gpu_same_work`, this GPU running exactly that: same 8 blocks, same iterate,
same one sweep, the preconditioner selection in its steady state, cost and gradient norm after it printed for both,
`hierarchy_setup_ms` per block; the same pair from the initial iterate under `initial_iterate`; beside it the 1-core C
port of the device algorithm on the timed single-agent step.
"""
import argparse
import json
import os
import sys
import time

# Agents that share a GPU are solved concurrently, one HIP stream each; the ROCm runtime maps streams onto 4 hardware
# queues by default, so more than 4 same-colour agents would queue behind each other.  Must be set before the runtime
# initialises (i.e. before torch / the library are imported).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200,
                    help="timed steps (default 200: > 1 s of timed region at the 100k workload's 5 ms per step)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--settle", type=int, default=5,
                    help="untimed RBCD iterations from the initial guess before the benchmark state is frozen")
    ap.add_argument("--workload", default="grid100k")
    ap.add_argument("--rank", type=int, default=5, help="relaxation rank r")
    ap.add_argument("--precond", default="auto", choices=["auto", "jacobi", "multilevel", "additive"],
                    help="tCG preconditioner: auto (library default, include/dpgo_hip.h DPGO_PRECOND_AUTO), or one of "
                         "the others all the time")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the sphere2500 side measurement (`also` field)")
    ap.add_argument("--cpu-budget-s", type=float, default=25.0)
    ap.add_argument("--spmm-reps", type=int, default=200)
    ap.add_argument("--agents-per-gpu", type=int, default=0, help="0 = auto (1 if one GPU, else 2; 8 with --loopback)")
    ap.add_argument("--sequential", action="store_true",
                    help="diagnostic: the agents a process hosts are solved one after the other (DPGO_SEQUENTIAL_SWEEP=1) -- "
                         "the sum of the solo solve times, from which a many-GPU sweep time can be predicted")
    ap.add_argument("--loopback", action="store_true",
                    help="single GPU, several agents: every public-pose exchange and reduction travels through a 1-rank "
                         "RCCL communicator owned by the solver library (the N > 1 data path on one device)")
    ap.add_argument("--transport", default="rccl", choices=["rccl", "ipc"],
                    help="N > 1: how the public poses travel.  rccl (default): grouped ncclSend / ncclRecv on the solver's "
                         "stream.  ipc: the peer-store transport of dpgo_amd/ipc.py -- receivers' neighbour buffers mapped "
                         "through hipIpc handles, senders' pack kernel writes straight into them; processes of one node, gloo "
                         "for rendezvous and reductions (works with several ranks on ONE device, which RCCL refuses)")
    ap.add_argument("--dist", action="store_true",
                    help="run the N > 1 code path whatever the world size: torch.distributed process group (RCCL), the "
                         "library's communicator taken from it, barriers and all-reduces of the timing protocol, 2 agents "
                         "per GPU.  With WORLD_SIZE = 1 (python -m torch.distributed.run --nproc-per-node 1 bench.py "
                         "--gpus 1 --dist) the exchanges are self send / recv through the 1-rank communicator: every line "
                         "of the multi-GPU branch executes on one device")
    return ap.parse_args()


def make_workload(name, r):
    """Returns (dataset, n, X0 tiles [n, d+1, r], description)."""
    import dpgo_amd
    from dpgo_amd import synthetic
    if name == "grid100k":
        name = "grid:50x50x40"
    if name.startswith("grid:"):
        nx, ny, nz = (int(v) for v in name[5:].split("x"))
        meas, n, Ttrue = synthetic.synthetic_grid(nx, ny, nz, seed=0)
        X0 = synthetic.lift_tiles(synthetic.perturbed_truth(Ttrue, seed=2), r)
        return meas, n, X0, "synthetic 3-D grid %dx%dx%d (%d poses, %d edges), init = perturbed truth" % (
            nx, ny, nz, n, len(meas))
    if name in ("sphere2500", "torus3D"):
        from dpgo_amd.initialization import chordal_initialization
        meas, n = dpgo_amd.read_g2o_file(os.path.join(ROOT, "data", name + ".g2o"))
        X0 = synthetic.lift_tiles(chordal_initialization(meas, n), r)
        return meas, n, X0, "%s.g2o (%d poses, %d edges), chordal init" % (name, n, len(meas))
    raise SystemExit("unknown workload %r" % name)


def spmm_bytes(n, nnzb, d, r):
    """Algorithmic bytes of one Q*X block-SpMM: BSR values + int32 column indices +
    row pointers + one read of X + one write of OUT."""
    b = d + 1
    return nnzb * (8 * b * b + 4) + 4 * (n + 1) + 2 * 8 * r * b * n


def hess_bytes(n, nnzb, d, r):
    """k_tcg_hess = the SpMM (Q, indices, one read of z) + fused epilogue: reads the iterate X, the
    cached S = sym(Y^T EG) blocks, delta and H delta; writes delta and H delta."""
    b = d + 1
    v = 8 * r * b * n
    return nnzb * (8 * b * b + 4) + 4 * (n + 1) + v + (v + 8 * d * d * n) + 2 * v + 2 * v


def pmc_bytes(pmc, prefix):
    """HBM-side bytes per launch of the kernel whose name starts with `prefix`, from a committed rocprofv3 PMC summary
    (profiles/r*_pmc_fetch_write.json: separate FETCH_SIZE and WRITE_SIZE passes of this same command, KB): FETCH_SIZE x 2
    (the gfx950 correction, calibrated on k_retract / k_rtr_update whose byte counts are exact) + WRITE_SIZE; "max" = the
    full (non-early-exit) launches.  Several instantiations under one prefix: the one launched most often.  Returns
    (bytes, key) or (None, None)."""
    if not pmc:
        return None, None
    fs, ws = pmc.get("FETCH_SIZE_KB", {}), pmc.get("WRITE_SIZE_KB", {})
    keys = [k for k in fs if k.startswith(prefix) and k in ws]
    if not keys:
        return None, None
    key = max(keys, key=lambda k: fs[k].get("calls", 0))
    return (2.0 * fs[key]["max"] + ws[key]["max"]) * 1024.0, key


def kitti_with_outliers(k=25, seed=11):
    """BASELINE configs[4]'s input: kitti_00.g2o (2-D, 4 541 poses) + k injected outlier loop closures (random rotation,
    translation in [-5, 5]^2, 300-600 poses apart, the median precisions of the data set's loop closures) -- the generator of
    tests/test_parity_gpu.py::_kitti_with_outliers on the product's own measurement type."""
    import dpgo_amd
    from dpgo_amd.measurements import RelativeSEMeasurements
    om, n = dpgo_amd.read_g2o_file(os.path.join(ROOT, "data", "kitti_00.g2o"))
    rng = np.random.default_rng(seed)
    taken = set(zip(om.p1.tolist(), om.p2.tolist()))
    p1, p2 = [], []
    while len(p1) < k:
        a = int(rng.integers(0, n - 600))
        b = int(a + rng.integers(300, 600))
        if (a, b) not in taken:
            taken.add((a, b))
            p1.append(a)
            p2.append(b)
    th = rng.uniform(-np.pi, np.pi, k)
    Rk = np.stack([np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]]) for a in th])
    lc = np.nonzero(~om.fixedWeight)[0]
    z = np.zeros(k, dtype=np.int64)
    out = RelativeSEMeasurements(2, z, np.array(p1), z.copy(), np.array(p2), Rk, rng.uniform(-5, 5, (k, 2)),
                                 np.full(k, np.median(om.kappa[lc])), np.full(k, np.median(om.tau[lc])), np.ones(k),
                                 np.zeros(k, dtype=bool))
    return om, RelativeSEMeasurements.concatenate([om, out]), n


def kitti_gnc_gpu(r, device, robots=4, k=25, inner_sweeps=2):
    """BASELINE configs[4] timed: kitti_00 + 25 outliers cut into 4 agents on this GPU, GNC-TLS with the reference's
    schedule (include/DPGO/DPGO_robust.h:49-53: barc 5, mu x 1.4), 2 coloured sweeps between weight updates.  A weight
    update (PGOAgent::updateMeasurementWeights, src/PGOAgent.cpp:1104-1142) = public-pose exchange + residuals and weights
    (k_edge_weights) + Q values, coupling blocks and preconditioner values rebuilt on the device (k_rebuild_Q; the block
    pattern is fixed) + the two scalar reductions; timed between device synchronisations.  The run is done twice from the
    same initial iterate: the first contains every first-use set-up, the second is what is reported."""
    import torch
    import dpgo_amd
    from dpgo_amd import synthetic
    from dpgo_amd.agent import DeviceAgent, ExchangePlan, RBCDCluster, build_pose_graphs
    from dpgo_amd.initialization import chordal_initialization
    from dpgo_amd.robust import DistributedGNC, RobustCostParameters
    om, meas, n = kitti_with_outliers(k)
    X0 = synthetic.lift_tiles(chordal_initialization(om, n), r)
    ranges, graphs = build_pose_graphs(meas, n, robots, r)
    plan = ExchangePlan(graphs)
    agents = {a: DeviceAgent(graphs, plan, a, X0[ranges[a][0]:ranges[a][1]], dpgo_amd.ROptParameters(), device=device)
              for a in range(robots)}
    cluster = RBCDCluster(plan, agents)
    gnc = DistributedGNC(cluster, RobustCostParameters("GNC_TLS", GNCMaxNumIters=80, GNCBarc=5.0, GNCMuStep=1.4),
                         inner_sweeps=inner_sweeps)
    t_rw, t_sw = [], []
    rw, sw = gnc._reweight_all, gnc._sweeps

    def timed(fn, sink):
        def wrapped(*a, **kw):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = fn(*a, **kw)
            torch.cuda.synchronize()
            sink.append(time.perf_counter() - t0)
            return out
        return wrapped

    gnc._reweight_all, gnc._sweeps = timed(rw, t_rw), timed(sw, t_sw)
    runs = []
    for _ in range(2):
        for a, ag in agents.items():
            ag.set_iterate(X0[ranges[a][0]:ranges[a][1]])
        del t_rw[:], t_sw[:]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        info = gnc.run()
        torch.cuda.synchronize()
        runs.append(dict(total_ms=1e3 * (time.perf_counter() - t0), updates=info["updates"],
                         # (the first _reweight_all only evaluates the residuals for mu_0: update=False)
                         ms_per_weight_update=1e3 * float(np.median(t_rw[1:])) if len(t_rw) > 1 else None,
                         ms_per_inner_block=1e3 * float(np.median(t_sw)), inner_blocks=len(t_sw),
                         cost_2f=info["cost"], gradnorm=info["gradnorm"], last=info["history"][-1] if info["history"] else None))
    first, rep = runs
    rep = dict(rep)
    rep.update(workload="kitti_00.g2o + %d outlier loop closures, %d agents, r = %d, GNC-TLS barc 5, mu x 1.4, %d sweeps "
                        "between weight updates" % (k, robots, r, inner_sweeps),
               ms_per_gnc_outer_iteration=(rep["ms_per_weight_update"] or 0.0) + rep["ms_per_inner_block"],
               first_run_total_ms=first["total_ms"],
               preconditioners=sorted({ag.last_result.precond_used for ag in agents.values() if ag.last_result}))
    return rep


def kitti_gnc_cpu(r, budget_updates=6, robots=4, k=25, inner_sweeps=2):
    """The same schedule on the host with the oracle (reference configuration: exact (Q_a + 0.1 I)^-1), bounded to
    `budget_updates` weight updates; seconds per GNC outer iteration (weight update + inner sweeps) = total / updates."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import dpgo_oracle as O
    om_p, meas_p, n = kitti_with_outliers(k)

    def conv(m):
        return O.Measurements(m.d, m.r1.astype(np.int64), m.p1.astype(np.int64), m.r2.astype(np.int64),
                              m.p2.astype(np.int64), m.R, m.t, m.kappa, m.tau, m.weight.copy(), m.fixedWeight.copy())
    X0 = O.lift(O.chordal_initialization(conv(om_p), n), r)
    t0 = time.perf_counter()
    _, info = O.multi_agent_gnc(conv(meas_p), n, robots, r, X0, inner_sweeps=inner_sweeps, barc=5.0, mu_step=1.4,
                                max_updates=budget_updates, precond="exact")
    el = time.perf_counter() - t0
    blocks = info["updates"] + 1  # (the block before mu_0 and one after every update but possibly the last)
    return dict(seconds=el, updates=info["updates"], seconds_per_gnc_outer_iteration=el / max(blocks, 1),
                sample="oracle, %d agents sequentially on one core, exact sparse factor, first %d weight updates of the "
                       "schedule" % (robots, info["updates"]))


def cpu_baseline(meas_p, n, X_state, r, budget_s, precond="jacobi"):
    """CPU restatement ("port") timed on the SAME step the GPU is timed on: one RBCD iteration (RTR 3 x <=50 tCG,
    block-Jacobi, H-direction recurrence) from the settled iterate, one thread like the reference (ENABLE_OPENMP
    OFF).  Preferred: the plain-C oracle (oracle/dpgo_oracle_c.c, gcc -O3 -march=x86-64-v3); fallback: the NumPy/SciPy
    oracle.  If one full step does not fit the budget the tCG cap is lowered and the time scaled."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import dpgo_oracle as O
    d = meas_p.d
    om = O.Measurements(d, meas_p.r1.astype(np.int64), meas_p.p1.astype(np.int64), meas_p.r2.astype(np.int64),
                        meas_p.p2.astype(np.int64), meas_p.R, meas_p.t, meas_p.kappa, meas_p.tau, meas_p.weight,
                        meas_p.fixedWeight)
    Q = O.construct_Q(n, d, om)
    X = np.ascontiguousarray(X_state)
    mode = True  # same tCG arithmetic as the device (H-direction recurrence)
    max_inner = 50
    CO = None
    if mode is True and precond == "jacobi":
        try:
            import c_oracle as CO
            CO.load()
        except Exception as exc:  # noqa: BLE001 -- no gcc / no prebuilt library: fall back to NumPy
            sys.stderr.write("bench.py: C oracle unavailable (%r); timing the NumPy oracle\n" % (exc,))
            CO = None
    if CO is not None:
        t0 = time.perf_counter()
        CO.spmm(Q, X, reps=3)
        spmm_s = (time.perf_counter() - t0) / 3
        per_tcg = 1.6 * spmm_s  # one tCG iteration ~ one SpMM + O(n) passes
        inner = max_inner if 3 * max_inner * per_tcg * 1.2 <= budget_s else max(2, int(budget_s / 3 / per_tcg / 1.2))
        t0 = time.perf_counter()
        _, res = CO.optimize(Q, None, X, RTR_tCG_iterations=inner, hess_recurrence=True)
        el = time.perf_counter() - t0
        iters, what = res.tcg_iterations, "plain-C oracle (gcc -O3 -march=x86-64-v3, single thread)"
        extra = dict(fOpt=res.fOpt if inner == max_inner else None, spmm_ms_1core=1e3 * spmm_s,
                     spmm_GBs_1core=spmm_bytes(n, len(Q.colidx), d, r) / spmm_s / 1e9, host_cores=os.cpu_count())
    else:
        prob = O.QuadraticProblem(Q, None, r, d, precond="jacobi" if precond == "jacobi" else "amg")
        EG = prob.euc_grad(X)
        S = prob.sym_ytg(X, EG)
        g = O.tangent_project(X, EG, d)
        t0 = time.perf_counter()
        reps = 0
        while reps < 2 or (time.perf_counter() - t0 < 1.0 and reps < 50):
            prob.rie_hess(X, S, g)
            prob.precondition(X, g)
            reps += 1
        per_tcg = (time.perf_counter() - t0) / reps
        inner = max_inner if 3 * max_inner * per_tcg * 1.2 <= budget_s else max(2, int(budget_s / 3 / per_tcg / 1.2))
        opt = O.QuadraticOptimizer(prob, O.ROptParameters(RTR_tCG_iterations=inner), hess_recurrence=mode)
        t0 = time.perf_counter()
        opt.optimize(X)
        el = time.perf_counter() - t0
        iters, what = opt.result.tcg_iters, "NumPy/SciPy oracle (single thread)"
        extra = dict(fOpt=opt.result.fOpt if inner == max_inner else None, host_cores=os.cpu_count())
    scale = max_inner / inner
    return dict(value=1.0 / (el * scale), unit="it/s", cores=1, kind="port",
                sample="1 RBCD iteration from the same settled iterate, %d tCG Hessian-vector products in %.1f s%s; "
                       "%s, the device path's algorithm with its block-Jacobi preconditioner" % (
                           iters, el,
                           "" if inner == max_inner else " (tCG capped at %d of 50, time scaled x%.2f)" % (inner, scale),
                           what),
                tcg_iterations=iters, seconds=el, **extra)


def _ref_agent_solve(task):
    """Worker of cpu_baseline_reference (one process = one core = one agent, as the reference's one thread per agent):
    PGOAgent::updateX on the CPU oracle with the reference's exact preconditioner; the sparse factor is built lazily
    inside the first solve (src/PoseGraph.cpp:582-586) and timed separately."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import dpgo_oracle as O
    try:
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(limits=1)  # one thread per agent (ENABLE_OPENMP OFF, CMakeLists.txt:55)
    except Exception:  # noqa: BLE001
        limiter = None
    Qa, G, r, d, Xa = task
    prob = O.QuadraticProblem(Qa, G, r, d, precond="exact")
    t0 = time.perf_counter()
    prob.precondition(Xa, np.zeros_like(Xa))  # forces the factorisation
    t_fact = time.perf_counter() - t0
    opt = O.QuadraticOptimizer(prob, O.ROptParameters())
    t0 = time.perf_counter()
    Xn = opt.optimize(Xa)
    t_solve = time.perf_counter() - t0
    del limiter
    return Xn, opt.result.tcg_iters, t_fact, t_solve


def cpu_baseline_reference(meas_p, n, X_tiles, r, num_agents=8):
    """The reference's own configuration of this workload (BASELINE configs[3]: the 100k-pose grid, 8 agents) on the
    host cores: contiguous blocks (examples/MultiRobotExample.cpp:71-88), one process per agent, local solve =
    RTR 3 x <=50 tCG with the EXACT sparse factor of Q_a + 0.1 I (SciPy SuperLU standing in for CHOLMOD), one
    two-colour sweep from the given iterate.  A single agent owning all 100k poses -- the GPU's N = 1 step -- is out of
    reach of a sparse direct factor on a 3-D grid, so the sample is the partitioned problem.
    Returns a dict; time per sweep = sum over the colour phases of the slowest agent."""
    import multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import dpgo_oracle as O
    d = meas_p.d
    om = O.Measurements(d, meas_p.r1.astype(np.int64), meas_p.p1.astype(np.int64), meas_p.r2.astype(np.int64),
                        meas_p.p2.astype(np.int64), meas_p.R, meas_p.t, meas_p.kappa, meas_p.tau, meas_p.weight,
                        meas_p.fixedWeight)
    ranges, per = O.partition_contiguous(om, n, num_agents)
    X = np.ascontiguousarray(X_tiles).copy()
    info = []
    for a in range(num_agents):
        s, e = ranges[a]
        priv = O.Measurements.concat([per[a]["odometry"], per[a]["private"]])
        sh = per[a]["shared"]
        need = sorted({(int(sh.r2[k]), int(sh.p2[k])) if sh.r1[k] == a else (int(sh.r1[k]), int(sh.p1[k]))
                       for k in range(sh.m)})
        info.append(dict(Q=O.construct_Q(e - s, d, priv, sh, my_id=a), shared=sh, need=need,
                         adj=sorted({rob for rob, _ in need})))
    colour = [-1] * num_agents
    for a in range(num_agents):
        used = {colour[q] for q in info[a]["adj"] if colour[q] >= 0}
        colour[a] = min(c for c in range(num_agents) if c not in used)
    cores = min(num_agents, os.cpu_count() or 1)
    t_sweep, t_fact, products, per_agent = 0.0, 0.0, 0, [0] * num_agents
    with mp.get_context("fork").Pool(cores) as pool:
        for c in range(max(colour) + 1):
            ids = [a for a in range(num_agents) if colour[a] == c]
            tasks = []
            for a in ids:
                s, e = ranges[a]
                nbr = {(rob, fr): X[ranges[rob][0] + fr] for rob, fr in info[a]["need"]}
                tasks.append((info[a]["Q"], O.construct_G(e - s, d, r, info[a]["shared"], a, nbr), r, d, X[s:e]))
            out = pool.map(_ref_agent_solve, tasks)
            for a, (Xn, its, tf, ts) in zip(ids, out):
                X[ranges[a][0]:ranges[a][1]] = Xn
                products += its
                per_agent[a] = its
            t_sweep += max(ts for _, _, _, ts in out)
            t_fact = max(t_fact, max(tf for _, _, tf, _ in out))
    central = O.QuadraticProblem(O.construct_Q(n, d, om), None, r, d, precond="none")
    return dict(value=1.0 / t_sweep, unit="it/s", cores=cores, kind="port",
                sample="reference configuration of this workload: %d agents x %d poses, one core each, exact sparse "
                       "factor of Q_a + 0.1 I (SciPy SuperLU for CHOLMOD), RTR 3x<=50 tCG; ONE two-colour sweep "
                       "(1 it = every agent updates once) from the given iterate: %.2f s + %.2f s once "
                       "for the factorisations (inside the first solve, src/PoseGraph.cpp:582-586); NumPy/SciPy "
                       "oracle" % (num_agents, n // num_agents, t_sweep, t_fact),
                seconds_per_sweep=t_sweep, factorisation_seconds=t_fact, tcg_iterations=products,
                tcg_iterations_per_agent=per_agent, cost_2f_after=2 * central.f(X), gradnorm_after=central.rie_grad_norm(X), host_cores=os.cpu_count())


def gpu_same_decomposition(meas, n, X0, r, num_agents, precond, device, reset_auto=True):
    """What cpu_baseline_reference times, on this GPU: the same `num_agents` contiguous blocks, the same initial iterate,
    ONE two-colour sweep (every agent updates once, RTR 3 x <= 50 tCG, the library's default preconditioner selection),
    agents of a colour solved concurrently.  Cost and gradient norm of the central problem after the sweep are returned
    so that the two measurements can be compared as work, not only as time."""
    import torch
    import dpgo_amd
    from dpgo_amd.agent import DeviceAgent, ExchangePlan, RBCDCluster, build_pose_graphs
    ranges, graphs = build_pose_graphs(meas, n, num_agents, r)
    plan = ExchangePlan(graphs)
    params = dpgo_amd.ROptParameters(precond=precond)
    agents = {a: DeviceAgent(graphs, plan, a, X0[ranges[a][0]:ranges[a][1]], params, device=device)
              for a in range(num_agents)}
    cluster = RBCDCluster(plan, agents)
    for ag in agents.values():
        ag.snapshot()
    cluster.sweep()  # untimed: first-use setup (block-Jacobi factors, launch caches)
    selection_sweeps = 0
    if precond == "auto" and not reset_auto:  # the selection's steady state from this iterate (as in main(); untimed)
        def undecided():  # (as in main(): agents of the cost rule that have not been through a trial yet)
            n_open = 0
            for ag in agents.values():
                info = ag.problem.autoInfo()
                small = ag.n * (ag.d + 1) <= 65536 and ag.has_neighbours  # (256 aggregates of one 64-lane-group tile)
                if small and (info["state"] == "trial" or (info["state"] == "jacobi" and info["backoff"] == 0)):
                    n_open += 1
            return n_open
        while undecided() and selection_sweeps < 40:
            for ag in agents.values():
                ag.restore()
            cluster.sweep()
            selection_sweeps += 1
    best, products = None, 0
    for _ in range(3):
        for ag in agents.values():
            ag.restore()
            if precond == "auto" and reset_auto:
                ag.problem.autoState("reset")  # (a fresh block of a multi-agent problem starts on block-Jacobi)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        cluster.sweep()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if best is None or el < best:
            best = el
        products = sum(ag.last_result.tcg_iterations for ag in agents.values())
    f, g = cluster.central_cost_and_gradnorm()
    # the once-per-Q cost of the blocks' preconditioner (the pair of cpu_baseline.factorisation_seconds): the hierarchy of
    # the additive one-launch solve -- what `auto` builds for a coupled block of this size once block-Jacobi has cost as
    # much -- timed on a fresh handle of every block: first call = block pattern (host) + values (device), second = values
    setup = None
    try:
        setup = dict(first_ms=[], values_only_ms=[], aggregates=[])
        for a in range(num_agents):
            pr = dpgo_amd.QuadraticProblem(graphs[a], device=device, host_linear_term=False)
            pl = pr.additivePlan()
            ks = pl["ks"] if pl["lane_groups"] else None
            for key in ("first_ms", "values_only_ms"):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                info = pr.setupMultilevel(ks)
                setup[key].append(1e3 * (time.perf_counter() - t0))
            setup["aggregates"].append(info["sizes"][-1])
            del pr
        setup["what"] = ("additive two-level hierarchy (merged graph aggregates, one per workgroup)" if ks is not None
                         else "default multilevel hierarchy")
    except Exception as exc:  # noqa: BLE001
        setup = {"error": repr(exc)}
    return dict(value=1.0 / best, unit="it/s", seconds_per_sweep=best, agents=num_agents,
                hierarchy_setup_ms=setup, selection_sweeps=selection_sweeps, auto_rule=[agents[a].problem.autoInfo() for a in range(num_agents)] if precond == "auto" else None,
                poses_per_agent=n // num_agents, tcg_iterations=products,
                tcg_iterations_per_agent=[agents[a].last_result.tcg_iterations for a in range(num_agents)],
                cost_2f_after=2 * f, gradnorm_after=g,
                preconditioners=sorted({ag.last_result.precond_used for ag in agents.values()}),
                sample="this GPU on the reference configuration: the same %d blocks, the same initial iterate, ONE "
                       "two-colour sweep, same-colour agents solved concurrently (best of 3)" % num_agents)


def secondary_single_agent(workload, r, precond, steps, warmup, settle):
    """The same fixed-work measurement for a second, small workload (single agent, single GPU): BASELINE's metric is
    quoted on sphere2500 as well as on the 100k grid.  Returns a small dict for the `also` field of the JSON line."""
    import torch
    import dpgo_amd
    from dpgo_amd.agent import DeviceAgent, ExchangePlan, build_pose_graphs
    meas, n, X0, desc = make_workload(workload, r)
    ranges, graphs = build_pose_graphs(meas, n, 1, r)
    ag = DeviceAgent(graphs, ExchangePlan(graphs), 0, X0, dpgo_amd.ROptParameters(precond=precond))
    states, works, gns = [], [], []
    for _ in range(settle + 1):
        states.append(ag.X.clone())
        res = ag.update()
        works.append(res.tcg_iterations)
        gns.append(res.gradNormInit)
    k = max(i for i in range(len(works)) if works[i] >= 0.5 * works[0])
    ag.X.copy_(states[k])
    ag.snapshot()
    for _ in range(warmup):
        ag.restore()
        ag.update()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tcg = 0
    for _ in range(steps):
        ag.restore()
        res = ag.update()
        tcg += res.tcg_iterations
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ph = ag.problem.persistentPhases()  # (the last timed solve: in-kernel split of a one-launch solve, zeros otherwise)
    return dict(workload=desc, precond=precond, precond_used=res.precond_used, it_per_s=steps / el,
                ms_per_step=1e3 * el / steps, tcg_iterations_per_step=tcg / steps,
                us_per_product=1e6 * el / max(tcg, 1), settle_iterations=k, gradnorm_before_step=gns[k],
                gradnorm_after_step=res.gradNormOpt,
                in_kernel_us_per_iteration=(dict(hessian_phase=ph["hessian"], all_reduce_after_hessian=ph["reduce_after_hessian"],
                                                 update_phase=ph["update"], reductions_after_update=ph["reduce_after_update"],
                                                 total=ph["hessian"] + ph["reduce_after_hessian"] + ph["update"] + ph["reduce_after_update"],
                                                 note="participant 0 of the one-launch solve, 100 MHz wall clock, iterations after the "
                                                      "first; us_per_product above = wall time of the whole step (launch, "
                                                      "outer iterations, read-back) over its products")
                                            if ph["iterations"] > 0 else None))


def time_to_tolerance(workload, r, precond, tol=1e-2, max_calls=12, coarse_bits=None, operator_bits=None):
    """Products-to-tolerance and time-to-gradnorm: QuadraticOptimizer::optimize (reference defaults) called from the
    initial guess until |rgrad| < tol (the local solver's own tolerance), single agent, single GPU."""
    import torch
    import dpgo_amd
    from dpgo_amd.agent import DeviceAgent, ExchangePlan, build_pose_graphs
    meas, n, X0, desc = make_workload(workload, r)
    ranges, graphs = build_pose_graphs(meas, n, 1, r)
    ag = DeviceAgent(graphs, ExchangePlan(graphs), 0, X0, dpgo_amd.ROptParameters(precond=precond))
    ag.update()  # untimed warm-up
    if coarse_bits is not None:
        ag.problem.multilevelCoarseBits(coarse_bits)
    if operator_bits is not None:
        ag.problem.multilevelOperatorBits(operator_bits)
    if precond == "multilevel":
        ag.problem.setupMultilevel()  # the hierarchy is a one-time cost per Q (the reference factors inside its first solve)
    ag.problem.autoState("reset")     # "auto" starts where a fresh handle starts
    ag.set_iterate(X0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    products, calls, gn, used = 0, 0, None, []
    for _ in range(max_calls):
        res = ag.update()
        used.append(res.precond_used)
        products += res.tcg_iterations
        calls += 1
        gn = res.gradNormOpt
        if gn < tol:
            break
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    # the once-per-Q cost the loop above does not contain (the reference factors inside its first timed solve,
    # src/PoseGraph.cpp:582-586): the hierarchy of the preconditioner the calls ran, built on a FRESH handle -- block
    # pattern on the host + values on the device (first_ms), values only (what a change of Q's values costs)
    setup_first = setup_values = None
    ml_used = [u for u in used if u in ("multilevel", "additive")]
    if ml_used:
        try:
            pr = dpgo_amd.QuadraticProblem(graphs[0], host_linear_term=False)
            ks = None
            if ml_used[0] == "additive":
                pl = pr.additivePlan()
                ks = pl["ks"] if pl["lane_groups"] else None
            if coarse_bits is not None:
                pr.multilevelCoarseBits(coarse_bits)
            ts = []
            for _ in range(2):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                pr.setupMultilevel(ks)
                ts.append(1e3 * (time.perf_counter() - t1))
            setup_first, setup_values = ts
            del pr
        except Exception as exc:  # noqa: BLE001
            sys.stderr.write("bench.py: hierarchy set-up timing failed: %r\n" % (exc,))
    return dict(products=products, rbcd_iterations=calls, ms=1e3 * el, gradnorm=gn, reached=bool(gn < tol),
                us_per_product=1e6 * el / max(products, 1), preconditioners=used,
                cycle_operator_copy_bits=(32 if ag.problem.multilevelOperatorBits()["active"] else 64) if ml_used else None,
                hierarchy_setup_ms=setup_first, hierarchy_values_only_ms=setup_values,
                ms_incl_setup=(1e3 * el + setup_first) if setup_first is not None else 1e3 * el)


LINE_LIMIT = 6144  # bytes: the driver's record keeps the last 8 KB of stdout


def _rnd(v, sig=5):
    """Floats rounded to `sig` significant digits (the last line is a summary; bench_detail.json keeps full precision)."""
    if isinstance(v, float):
        return float("%.*g" % (sig, v)) if v == v and abs(v) != float("inf") else None
    if isinstance(v, dict):
        return {k: _rnd(x, sig) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_rnd(x, sig) for x in v]
    return v


def compact_line(out):
    """The driver's line from the full record `out`: headline keys, `config` in a few short strings, `roofline` (headline
    kernel + its PMC traffic + the fraction on the bytes it streams + the SpMM north_star names + one entry per kernel of
    the iteration + the whole iteration), `cpu_baseline` (the reference-configuration pair) and BASELINE configs[4]'s
    timing.  Guaranteed <= LINE_LIMIT bytes: optional groups are dropped from the end of the priority list until it fits."""
    rf, cb, cfg = out.get("roofline") or {}, out.get("cpu_baseline") or None, out.get("config") or {}
    q = out.get("quality") or {}
    head = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                                    "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "products_per_step",
                                    "gradnorm_before_step", "gradnorm_after_step", "time_to_tolerance_ms",
                                    "products_to_tolerance", "hierarchy_setup_ms", "hierarchy_values_only_ms",
                                    "time_to_tolerance_incl_setup_ms")}
    head["us_per_product"] = q.get("us_per_tcg_iteration_rank0")
    ml = rf.get("multilevel") or {}
    config = {"workload": cfg.get("workload"), "agents": cfg.get("agents"), "agents_per_gpu": cfg.get("agents_per_gpu"),
              "r": cfg.get("r"), "d": cfg.get("d"),
              "local_solver": "RTR 3x<=50 tCG, Delta0=100, tol=1e-2 (reference defaults), precond=%s, ran %s" % (
                  cfg.get("precond"), "+".join(cfg.get("precond_used_in_timed_steps") or [])),
              "cycle_storage": cfg.get("cycle_storage_short"), "schedule": (cfg.get("schedule") or "")[:120],
              "transport": cfg.get("transport"), "exchange_ms_per_step": q.get("exchange_ms_per_step_rank0"),
              "poses_per_agent": cfg.get("poses_per_agent"), "nnzb_per_agent": cfg.get("nnzb_per_agent")}
    roof = {k: rf.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "traffic_live",
                                   "bytes_per_launch", "avg_launch_us", "storage")}
    roof["kernel"] = (rf.get("kernel") or "").split(" (")[0]
    roof["what"] = "one tCG step (Q*z block-SpMM + Riemannian-Hessian epilogue + recurrences); achieved/frac: ALGORITHMIC " \
                   "bytes (full fp64 Q, SURVEY 8d) / rotating-operand time; frac_own_bytes: the bytes this storage streams"
    roof["streamed_bytes"] = rf.get("stored_bytes_per_launch")
    roof["frac_own_bytes"] = rf.get("frac_own_bytes")
    if rf.get("traffic") and rf.get("avg_launch_us"):
        roof["frac_traffic"] = rf["traffic"] / rf["avg_launch_us"] / 1e3 / HBM_PEAK_GBS
    if rf.get("warm"):
        roof["warm_us"], roof["warm_frac"] = rf["warm"].get("avg_launch_us"), rf["warm"].get("frac")
    for key in ("products_per_launch", "us_per_product", "effective_algorithmic_frac"):  # (one-launch solves)
        if rf.get(key) is not None:
            roof[key] = rf[key]
    sp = rf.get("spmm_symmetric") if rf.get("spmm_storage_selected") == "symmetric" and rf.get("spmm_symmetric") else rf.get("spmm_only")
    if sp:
        roof["spmm"] = {"kernel": (sp.get("kernel") or "").split(" (")[0], "bytes_per_launch": sp.get("bytes_per_launch"),
                        "avg_launch_us": sp.get("avg_launch_us"), "frac": sp.get("frac"),
                        "frac_own_bytes": sp.get("frac_own_bytes", sp.get("frac")), "traffic": sp.get("traffic")}
    roof["kernels"] = [{"kernel": k.get("kernel"), "us": k.get("avg_launch_us"), "streamed_bytes": k.get("streamed_bytes"),
                        "algorithmic_bytes_fp64": k.get("algorithmic_bytes_fp64"), "achieved": k.get("achieved"),
                        "frac": k.get("frac"), "traffic": k.get("traffic"), "frac_traffic": k.get("frac_traffic")}
                       for k in (rf.get("kernels") or [])]
    it = rf.get("iteration")
    roof["iteration"] = {k: it.get(k) for k in ("launches", "bytes", "traffic", "us_per_product", "achieved", "frac",
                                                "frac_traffic", "kernel_us_sum")} if it else None
    if ml:
        roof["hierarchy"] = {"sizes": ml.get("sizes"), "ks": ml.get("ks"), "coarse_inverse_bits": ml.get("coarse_inverse_bits"),
                             "cycle_operator_copy_bits": ml.get("cycle_operator_copy_bits")}
    cpu = None
    if cb:
        g = cb.get("gpu_same_work") or {}
        hs = (g.get("hierarchy_setup_ms") or {})
        cpu = {k: cb.get(k) for k in ("value", "unit", "cores", "kind", "seconds_per_sweep", "factorisation_seconds",
                                      "tcg_iterations", "host_cores", "gradnorm_after")}
        cpu["sample"] = (cb.get("sample_short") or cb.get("sample") or "")[:200]
        if g:
            cpu["gpu_same_work"] = {"seconds_per_sweep": g.get("seconds_per_sweep"), "tcg_iterations": g.get("tcg_iterations"),
                                    "preconditioners": g.get("preconditioners"), "gradnorm_after": g.get("gradnorm_after"),
                                    # (one timing per block on a fresh handle: the median is the figure, the max shows outliers)
                                    "hierarchy_setup_ms_median": sorted(hs["first_ms"])[len(hs["first_ms"]) // 2] if hs.get("first_ms") else None,
                                    "hierarchy_setup_ms_max": max(hs["first_ms"]) if hs.get("first_ms") else None}
    kit = out.get("kitti_gnc")
    kitti = None
    if kit and "error" not in kit:
        kitti = {k: kit.get(k) for k in ("ms_per_weight_update", "ms_per_inner_block", "ms_per_gnc_outer_iteration", "updates",
                                         "total_ms", "first_run_total_ms", "cost_2f", "preconditioners")}
        kitti["workload"] = "kitti_00 + 25 outliers, 4 agents, GNC-TLS barc 5, mu x1.4, 2 sweeps per update"
        if kit.get("cpu"):
            kitti["cpu_seconds_per_gnc_outer_iteration"] = kit["cpu"].get("seconds_per_gnc_outer_iteration")
            kitti["cpu_sample"] = (kit["cpu"].get("sample") or "")[:120]
    elif kit:
        kitti = {"error": str(kit.get("error"))[:160]}
    tt = {}
    for key, v in (q.get("to_tolerance") or {}).items():
        if isinstance(v, dict) and "products" in v and key.split("/")[-1] in ("auto", "jacobi"):
            tt[key] = [v.get("products"), v.get("ms"), v.get("reached")]
    also = {}
    for key, v in (out.get("also") or {}).items():
        if isinstance(v, dict) and "it_per_s" in v and key == "sphere2500":
            also[key] = {"it_per_s": v["it_per_s"], "us_per_product": v.get("us_per_product"), "precond_used": v.get("precond_used"),
                         "in_kernel_us_per_iteration": (v.get("in_kernel_us_per_iteration") or {}).get("total")}
    line = dict(head)
    line.update(config=config, roofline=roof, cpu_baseline=cpu, kitti_gnc=kitti,
                to_tolerance_products_ms_reached=tt or None, also=also or None,
                detail="full record: bench_detail.json and the stdout line prefixed 'DETAIL '")
    line = _rnd(line)
    text = json.dumps(line, separators=(",", ":"))
    for drop in ("also", "to_tolerance_products_ms_reached", ("roofline", "hierarchy"), ("roofline", "what"),
                 ("config", "schedule"), ("roofline", "traffic_source"), ("cpu_baseline", "sample"), "kitti_gnc"):
        if len(text) <= LINE_LIMIT:
            break
        if isinstance(drop, tuple):
            if isinstance(line.get(drop[0]), dict):
                line[drop[0]].pop(drop[1], None)
        else:
            line.pop(drop, None)
        text = json.dumps(line, separators=(",", ":"))
    if len(text) > LINE_LIMIT:
        raise RuntimeError("bench line is %d bytes (> %d)" % (len(text), LINE_LIMIT))
    return text


def main():
    args = parse_args()
    if args.sequential:
        os.environ["DPGO_SEQUENTIAL_SWEEP"] = "1"
    import torch
    import torch.distributed as dist
    import dpgo_amd
    from dpgo_amd.agent import DeviceAgent, ExchangePlan, RBCDCluster, build_pose_graphs

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch with torch.distributed.run for --gpus > 1")
    # the multi-GPU branch: N > 1 ranks, or forced at any world size (--dist / DPGO_BENCH_DIST=1) so that it can be
    # executed -- and is tested -- on a single device
    use_dist = world > 1 or args.dist or os.environ.get("DPGO_BENCH_DIST", "0") == "1"
    if use_dist and "MASTER_ADDR" not in os.environ:  # (python bench.py --dist without a launcher)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    loopback = args.loopback or (use_dist and world == 1)
    if dpgo_amd.device_count() < 1:
        raise SystemExit("bench.py needs a HIP device: the product path has no CPU fallback")
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    backend = None
    comm = None
    if use_dist:
        # "nccl" IS RCCL on ROCm.  DPGO_DIST_BACKEND=gloo (host-staged exchange) lets the N > 1 path be exercised
        # on a single-GPU box with all ranks sharing device 0; it is also the fallback if RCCL cannot initialise.
        backend = "gloo" if args.transport == "ipc" else os.environ.get("DPGO_DIST_BACKEND", "nccl")
        if args.transport == "ipc" and world == 1:
            raise SystemExit("--transport ipc needs at least two processes (torch.distributed.run --nproc-per-node 2)")
        if backend == "nccl":
            # no silent degradation: if RCCL cannot initialise the run FAILS (non-zero exit) -- a host-staged number
            # must never stand in for the xGMI one
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
            t = torch.ones(1, device="cuda")
            dist.all_reduce(t)  # create torch's communicator eagerly (barriers of the timing protocol)
            torch.cuda.synchronize()
            from dpgo_amd.comm import DeviceComm
            comm = DeviceComm.from_torch_distributed(dev_index)  # the data path's own communicator (C ABI dpgo_comm_*)
        else:
            dist.init_process_group(backend)
    elif loopback:
        from dpgo_amd.comm import DeviceComm, unique_id
        comm = DeviceComm(1, 0, unique_id(), dev_index)  # 1-rank RCCL communicator: self send / recv, all-reduce

    r = args.rank
    meas, n, X0, desc = make_workload(args.workload, r)
    d = meas.d
    # agents: 1 for a single GPU (one agent owns the whole graph, BASELINE configs[1] style);
    # for N > 1 GPUs two agents per GPU by default -- consecutive blocks of a chain / ring partition
    # alternate colours, so every GPU hosts one agent of each colour and works in BOTH colour phases
    apg = args.agents_per_gpu if args.agents_per_gpu > 0 else (2 if use_dist else (8 if args.loopback else 1))
    num_agents = world * apg
    ranges, graphs = build_pose_graphs(meas, n, num_agents, r)
    params = dpgo_amd.ROptParameters(precond=args.precond)  # reference defaults + block-Jacobi (or multilevel)
    plan = ExchangePlan(graphs)
    my_ids = list(range(rank * apg, (rank + 1) * apg))
    agents = {a: DeviceAgent(graphs, plan, a, X0[ranges[a][0]:ranges[a][1]], params, device=dev_index)
              for a in my_ids}
    cluster = RBCDCluster(plan, agents, rank, world, agents_per_rank=apg, comm=comm, loopback=loopback)
    if use_dist and args.transport == "ipc":
        cluster.enable_peer_store()
    big = max(my_ids, key=lambda a: graphs[a].n())  # the agent whose kernels are profiled below
    agent = agents[big]
    nnzb_local = len(graphs[big].quadraticMatrix()[1])
    n_local = graphs[big].n()

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    # Settle into the working regime (not timed), then make every step identical work: each warm-up / timed
    # step restores the settled iterate and runs ONE full RBCD iteration from it.  (Consecutive iterations
    # of a single agent converge within ~10 steps, after which optimize() is a no-op -- timing those would
    # inflate the rate and make it depend on K.)
    f0, g0 = cluster.central_cost_and_gradnorm()
    trajectory = [(2 * f0, g0)]
    # `settle` sweeps plus one probe sweep are run; the benchmark state is the iterate before the LAST sweep that
    # still did at least half the Hessian-vector products of the first one.  Large problems: that is the state
    # after all `settle` sweeps (the probe itself is heavy); small problems that converge during the settle
    # phase: the last iterate from which a sweep is real work.
    states, works = [], []
    for k in range(args.settle + 1):
        states.append({a: ag.X.clone() for a, ag in agents.items()})
        cluster.sweep()
        work = torch.tensor([float(sum(a.last_result.tcg_iterations for a in agents.values() if a.last_result))],
                            dtype=torch.float64, device="cpu" if (use_dist and cluster.stage) else "cuda")
        if use_dist:
            dist.all_reduce(work)
        works.append(float(work.item()))
        if k < args.settle:
            f, g = cluster.central_cost_and_gradnorm()
            trajectory.append((2 * f, g))
    settled = max(k for k in range(len(works)) if works[k] >= 0.5 * works[0])
    for a, ag in agents.items():
        ag.X.copy_(states[settled][a])
    trajectory = trajectory[:settled + 1]
    del states
    for a in agents.values():
        a.snapshot()
    # Steady state of the preconditioner selection (untimed): `auto` decides per handle from the solves themselves (coupled
    # blocks: the cost rule of dpgo_hip.h pays the hierarchy once the block-Jacobi solves have cost as much) -- sweeps from
    # the benchmark's iterate until no agent's selection state changed in two consecutive sweeps, so that the timed steps
    # run what a long RBCD run runs and contain no set-up.
    selection_sweeps = 0
    if args.precond == "auto" and num_agents > 1:
        def undecided():
            # an agent still on block-Jacobi that has never been through a trial (nor skipped a hopeless one) will switch
            # once its block-Jacobi work has paid one set-up: not a steady state yet; blocks outside the cost rule
            # (no coupling, or too large for the additive form: they follow the tCG-budget hysteresis) are steady at once
            n_open = 0
            for ag in agents.values():
                info = ag.problem.autoInfo()
                small = ag.n * (ag.d + 1) <= 65536 and ag.has_neighbours  # (256 aggregates of one 64-lane-group tile)
                if small and (info["state"] == "trial" or (info["state"] == "jacobi" and info["backoff"] == 0)):
                    n_open += 1
            return n_open
        while selection_sweeps < 40:
            open_ = torch.tensor([float(undecided())], dtype=torch.float64,
                                 device="cpu" if (use_dist and cluster.stage) else "cuda")
            if use_dist:  # every rank runs the same number of sweeps (they contain the exchanges)
                dist.all_reduce(open_, op=dist.ReduceOp.MAX)
            if float(open_.item()) == 0.0:
                break
            for a in agents.values():
                a.restore()
            cluster.sweep()
            selection_sweeps += 1

    # device time of the public-pose exchanges, from event pairs on the stream they are enqueued on -- no host wait is
    # added to the path being timed (pack kernel -> RCCL batch / device copies -> consumed by the coupling SpMM)
    ex_events = []
    _exchange = cluster.exchange

    def timed_exchange(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _exchange(*a, **k)
        e1.record()
        ex_events.append((e0, e1))

    if num_agents > 1:
        cluster.exchange = timed_exchange

    def step():
        for a in agents.values():
            a.restore()
        cluster.sweep()

    for _ in range(args.warmup):
        step()
    barrier()
    del ex_events[:]
    t0 = time.perf_counter()
    tcg_total = 0
    used_precond = set()
    for _ in range(args.steps):
        step()
        tcg_total += sum(a.last_result.tcg_iterations for a in agents.values() if a.last_result)
        used_precond |= {a.last_result.precond_used for a in agents.values() if a.last_result}
    barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if cluster.stage else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    exchange_ms = sum(e0.elapsed_time(e1) for e0, e1 in ex_events)
    cluster.exchange = _exchange
    # (what the timed steps' cycles streamed: asked NOW -- the probes below set the hierarchy up again)
    ops_state = agent.problem.multilevelOperatorBits()
    ops_active = bool(ops_state["active"])
    f1, g1 = cluster.central_cost_and_gradnorm()

    # ---- dominant-kernel roofline, measured live with HIP events on the solver's stream ----
    lib = dpgo_amd.lib.load()
    import ctypes as C
    ms_spmm = C.c_double(0.0)
    agent.problem.setSpmmVariant("plain")
    dpgo_amd.lib.check(lib.dpgo_bench_spmm(agent.problem.handle, args.spmm_reps, 10, C.byref(ms_spmm)))
    hb = hess_bytes(n_local, nnzb_local, d, r)
    sb = spmm_bytes(n_local, nnzb_local, d, r)
    # the same SpMM cycling through enough private operand sets to exceed the 256 MB Infinity Cache
    # (">= 3 buffer sets > 256 MB total"): the rate that can only come from HBM
    ms_rot, set_bytes = C.c_double(0.0), C.c_double(0.0)
    set_b = nnzb_local * (8 * (d + 1) ** 2 + 4) + 2 * 8 * r * (d + 1) * n_local
    nsets = int(min(512, max(3, -(-3 * 256 * 2 ** 20 // max(set_b, 1)) // 2 + 1)))  # >= 1.5 x 256 MB in total
    dpgo_amd.lib.check(lib.dpgo_bench_spmm_rotating(agent.problem.handle, nsets, args.spmm_reps, 10,
                                                    C.byref(ms_rot), C.byref(set_bytes)))
    # the same product on the symmetric storage (upper blocks only, outer-product gather): what blocks beyond the Infinity
    # Cache's size select by themselves (DPGO_SPMM_AUTO), forced here so that the 100k workload's rotating figure shows it
    spmm_sym = None
    if agent.problem.setSpmmVariant("symmetric") == "symmetric":
        ms_srot, ms_swarm, sset = C.c_double(0.0), C.c_double(0.0), C.c_double(0.0)
        dpgo_amd.lib.check(lib.dpgo_bench_spmm_rotating(agent.problem.handle, 2 * nsets, args.spmm_reps, 10,
                                                        C.byref(ms_srot), C.byref(sset)))
        dpgo_amd.lib.check(lib.dpgo_bench_spmm(agent.problem.handle, args.spmm_reps, 10, C.byref(ms_swarm)))
        spmm_sym = dict(kernel="k_spmm_sym<%d,%d> (plain Q*X on symmetric storage)" % (d, r),
                        bytes_per_launch=sb, avg_launch_us=ms_srot.value * 1e3,
                        achieved=sb / (ms_srot.value * 1e-3) / 1e9, frac=sb / (ms_srot.value * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        stored_bytes_per_launch=sset.value, buffer_sets=2 * nsets, total_MB=2 * nsets * sset.value / 1e6,
                        achieved_own_bytes=sset.value / (ms_srot.value * 1e-3) / 1e9,
                        frac_own_bytes=sset.value / (ms_srot.value * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        warm=dict(avg_launch_us=ms_swarm.value * 1e3, achieved=sb / (ms_swarm.value * 1e-3) / 1e9,
                                  frac=sb / (ms_swarm.value * 1e-3) / 1e9 / HBM_PEAK_GBS),
                        note="rates are the product's ALGORITHMIC bytes (full Q) over its time, comparable with "
                             "spmm_only; this storage moves stored_bytes_per_launch")
    # ---- the tCG-step kernel: both storages, HBM-only (rotating) and back-to-back; the HEADLINE is the kernel the timed
    # loop launched (`in_use`: what the library's size switch selects for this block)
    in_use = agent.problem.setSpmmVariant("auto")
    hsets = max(3, nsets // 2 + 1)  # a tCG-step operand set is ~1.7x an SpMM set
    span = ((d + 1) * r) % 2 == 0
    names = {}
    hess = {}
    for variant in ("plain", "symmetric"):
        if agent.problem.setSpmmVariant(variant) != variant:
            continue  # (the symmetric storage needs one pose per d+1 lanes: blocks >= 40 000 poses)
        ki = agent.problem.tcgKernelInfo()  # the instance the library launches for this block (size rules, dpgo_hip.h)
        names[variant] = ("k_tcg_hess_sym<%d,%d,%d>" % (d, r, ki["stream_nt"]) if ki["symmetric"] else
                          "k_tcg_hess_span<%d,%d,%d,%d>" % (d, r, ki["split"], ki["stream_nt"]) if span else
                          "k_tcg_hess<%d,%d,%d>" % (d, r, ki["split"]))
        cold, warm_ = C.c_double(0.0), C.c_double(0.0)
        dpgo_amd.lib.check(lib.dpgo_bench_hess_rotating(agent.problem.handle, hsets + (variant == "symmetric"),
                                                        args.spmm_reps, 10, C.byref(cold)))
        dpgo_amd.lib.check(lib.dpgo_bench_hess(agent.problem.handle, args.spmm_reps, 10, C.byref(warm_)))
        hess[variant] = dict(kernel=names[variant], cold_us=cold.value * 1e3, warm_us=warm_.value * 1e3)
    agent.problem.setSpmmVariant("auto")
    b_ = d + 1
    qb_ = nnzb_local * (8 * b_ * b_ + 4) + 4 * (n_local + 1)  # Q with its indices
    nu_ = (nnzb_local + n_local) // 2  # stored upper blocks (diagonal + one of every off-diagonal pair)
    hb_sym_own = (nu_ * (8 * b_ * b_ + 4) + (nnzb_local - nu_) * 8 + 2 * 4 * (n_local + 1)
                  + hb - (nnzb_local * (8 * b_ * b_ + 4) + 4 * (n_local + 1)))

    def hess_entry(variant):
        h = hess[variant]
        e = dict(kernel=h["kernel"], bytes_per_launch=hb, avg_launch_us=h["cold_us"],
                 achieved=hb / h["cold_us"] / 1e3, frac=hb / h["cold_us"] / 1e3 / HBM_PEAK_GBS,
                 warm=dict(avg_launch_us=h["warm_us"], achieved=hb / h["warm_us"] / 1e3,
                           frac=hb / h["warm_us"] / 1e3 / HBM_PEAK_GBS))
        if variant == "symmetric":  # effective rate above (full-Q bytes over its time); the bytes it really moves:
            e.update(stored_bytes_per_launch=hb_sym_own, achieved_own_bytes=hb_sym_own / h["cold_us"] / 1e3,
                     frac_own_bytes=hb_sym_own / h["cold_us"] / 1e3 / HBM_PEAK_GBS,
                     note="achieved / frac = the step's ALGORITHMIC bytes (full Q, SURVEY 8d) over this kernel's time, "
                          "comparable with the plain kernel; frac_own_bytes = the bytes this storage moves")
        return e

    timed = in_use if in_use in hess else "plain"
    other = "symmetric" if timed == "plain" else "plain"
    ms_hess = C.c_double(hess[timed]["warm_us"] * 1e-3)
    ms_hrot = C.c_double(hess[timed]["cold_us"] * 1e-3)
    kname = hess[timed]["kernel"]
    ach = hb / (ms_hess.value * 1e-3) / 1e9
    traffic = None
    traffic_src = None
    import glob
    pmc = None
    pmc_files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_fetch_write.json")))
    if world == 1 and args.workload == "grid100k" and r == 5 and pmc_files:
        # HBM-side bytes per launch from rocprofv3 PMC passes of this same command (separate FETCH_SIZE and
        # WRITE_SIZE passes, tools/profile_round.sh; pmc_bytes() above).  The newest committed summary is used -- a
        # constant of the committed profile, not a measurement of this run (traffic_source says which file).
        pmc = json.load(open(pmc_files[-1]))
        traffic, key = pmc_bytes(pmc, kname.split("<")[0] + "<")
        if traffic is not None:
            traffic_src = "%s [%s]" % (os.path.relpath(pmc_files[-1], ROOT), key)
    # HBM figure first (SURVEY 8d protocol: every operand of the launch cycles through > 256 MB of private copies, so
    # the 256 MB Infinity Cache cannot serve it); `warm` = back-to-back launches on the solver's own buffers, which is
    # what the tCG loop sees for blocks whose working set fits that cache (all of BASELINE's configurations)
    ach_rot = hb / (ms_hrot.value * 1e-3) / 1e9
    vec = 8 * r * b_ * n_local
    ms_it = (C.c_double * 5)()
    dpgo_amd.lib.check(lib.dpgo_bench_iteration_kernels(agent.problem.handle, args.spmm_reps, 10, ms_it))
    # Every kernel of one preconditioned tCG iteration gets TWO byte counts: `algorithmic_bytes_fp64` (every operand at
    # 8 bytes, the full Q: comparable across storages) and `streamed_bytes` (what THIS configuration's launch moves: upper
    # blocks only on the symmetric storage, fp32 operator copies and cycle vectors at 4 bytes).  achieved / frac are on
    # the streamed bytes; `traffic` = the committed PMC summary's HBM-side bytes of that kernel (pmc_bytes()).
    # (block-Jacobi mode reads X for the tangent projection of z: 8 pose vectors; the multilevel pre-smoothing step is not
    # projected: 7 -- the probe runs the mode the hierarchy state selects, dpgo_bench_iteration_kernels)
    vb = 4 if ops_active else 8             # bytes per stored value of the cycle's level-0 operator copies
    xb = 4 if ops_state["vectors"] else 8   # bytes per entry of the two cycle-internal vectors
    pbb = 8 * b_ * b_ * n_local             # one (d+1)^2 block per pose, fp64 (smoother factors; prolongation at vb)
    upd_vecs = 8 if args.precond == "jacobi" else 7

    def update_bytes(xb_):
        # block-Jacobi mode: 8 pose vectors + the factors; multilevel mode: reads eta, delta, H delta, r and the smoother
        # factors, writes eta, r and the pre-smoothed iterate (in the cycle's vector storage)
        return (upd_vecs * vec + pbb) if args.precond == "jacobi" else (6 * vec + pbb + vec * xb_ // 8)

    kernels = [dict(kernel="k_tcg_update_span" if span else "k_tcg_update", pmc_prefix="k_tcg_update",
                    what="eta, r updates, pre-smoothing / block-Jacobi, <r,r>",
                    algorithmic_bytes_fp64=update_bytes(8), streamed_bytes=update_bytes(xb), avg_launch_us=ms_it[0] * 1e3)]
    ml_info = None
    if args.precond != "jacobi":
        ml_info = agent.problem.setupMultilevel()  # (auto may not have built it yet)
        qb = nnzb_local * (8 * b_ * b_ + 4) + 4 * (n_local + 1)
        Nc = ml_info["sizes"][-1] * b_
        cbits = 32 if ops_state["dense"] else agent.problem.multilevelCoarseBits()  # (what the timed steps streamed)
        ml_info["coarse_inverse_bits"] = cbits
        # storage of the level-0 operator copies the cycle streams (symmetric Q in the restriction, A P in the
        # post-smoothing, prolongation blocks): fp32 copies by default on blocks that run the symmetric storage; every
        # product and sum is fp64.
        ml_info["cycle_operator_copy_bits"] = 32 if ops_active else 64
        path = agent.problem.multilevelPath()
        ml_info["path"] = path
        two = len(ml_info["sizes"]) == 2
        nnzb_ap = None
        graph = ml_info["ks"][0] < 0  # graph aggregates: members anywhere, summed by k_ml_agg_sum
        if path["ap"]:  # blocks of A P: the distinct aggregates the block columns of every row fall into
            nnzb_ap = int(agent.problem.multilevelGet(0, "ap_nnzb")[0])
        partials = int(agent.problem.multilevelGet(0, "restrict_partials")[0]) if graph else 0

        # The byte model of a cycle kernel is ONE function of the storage: (vb_, xb_, cb_) = bytes per operator value /
        # cycle-vector entry / dense-level entry, sym_ = Q walked on the symmetric storage.  streamed_bytes = the model at
        # this configuration's storage, algorithmic_bytes_fp64 = the same model at 8 / 8 / 8 bytes and the full Q.
        def dense_bytes(cb_):
            if path["packed_dense"]:
                nt_ = -(-Nc // 64)
                return 8 * 64 * 64 * nt_ * (nt_ + 1) // 2 + 2 * 8 * r * Nc + 2 * 8 * r * 64 * nt_ * nt_ // 2
            # the inverse and the restricted residual it multiplies at cb_; the coarse solution out (A P path) in fp64;
            # without A P the prolongation happens here (x1, P in; x out)
            return cb_ * Nc * Nc + cb_ * r * Nc + (8 * r * Nc if path["ap"] else 0) + (2 * vec + pbb if two and not path["ap"] else 0)

        def post_bytes(vb_, xb_):
            if path["ap"]:
                # A P and the prolongation blocks at vb_, the kept residual at xb_; X, r, smoother factors, the aggregate
                # labels and the output z in fp64 / int32; the coarse solution once
                return (nnzb_ap * (vb_ * b_ * b_ + 4) + 4 * (n_local + 1) + 3 * vec + vec * xb_ // 8 + pbb + pbb * vb_ // 8
                        + 4 * n_local + 8 * r * Nc)
            return qb + 4 * vec + pbb

        def restrict_bytes(vb_, xb_, cb_, sym_):
            # Q (this storage) + the pre-smoothed iterate (own tiles; its gathered tiles are re-reads) and the kept residual
            # at xb_ + r in fp64 + prolongation blocks at vb_ + run table + partial sums out and in (graph aggregates: one
            # per run of same-aggregate poses inside a wave's chunk) + the restricted residual at cb_
            q_ = (nu_ * (vb_ * b_ * b_ + 4) + (nnzb_local - nu_) * 8 + 2 * 4 * (n_local + 1)) if sym_ else qb
            return (q_ + vec * xb_ // 8 + vec + pbb * vb_ // 8 + (vec * xb_ // 8 if path["ap"] else 0)
                    + (4 * n_local if graph else 0) + 2 * 8 * r * b_ * partials + cb_ * r * Nc)

        if path["packed_dense"]:
            dense = dict(kernel="k_dense_sym_apply", pmc_prefix="k_dense_sym_apply",
                         what="k_ml_coarse_prolong -> k_dense_sym_apply + k_dense_sym_finish (packed lower triangle of the "
                              "inverse of %d unknowns, fp64, matrix cores)" % Nc)
        else:
            dense = dict(kernel="k_ml_coarse_prolong", pmc_prefix="k_ml_coarse_prolong",
                         what="dense inverse of %d unknowns stored in fp%d, fp64 arithmetic%s"
                              % (Nc, cbits, "" if path["ap"] else ", + prolongation"))
        dense.update(algorithmic_bytes_fp64=dense_bytes(8), streamed_bytes=dense_bytes(cbits // 8), avg_launch_us=ms_it[2] * 1e3)
        if path["ap"]:
            post = dict(kernel="k_ml_post_ap", pmc_prefix="k_ml_post_ap",
                        what="post-smoothing through A P and the coarse solution, prolongation, projection, <r,r>, <z,r>")
        else:
            post = dict(kernel="k_ml_post", pmc_prefix="k_ml_post<",
                        what="post-smoothing in the SpMM epilogue, projection, <r,r>, <z,r>")
        post.update(algorithmic_bytes_fp64=post_bytes(8, 8), streamed_bytes=post_bytes(vb, xb), avg_launch_us=ms_it[3] * 1e3)
        kernels += [
            dict(kernel="k_ml_restrict" + (" + k_ml_agg_sum" if graph else ""), pmc_prefix="k_ml_restrict",
                 pmc_prefix2="k_ml_agg_sum" if graph else None,
                 what="level 0: r - A x1 in one pass over Q, P^T, aggregate sums%s" % (", residual kept" if path["ap"] else ""),
                 algorithmic_bytes_fp64=restrict_bytes(8, 8, 8, False),
                 streamed_bytes=restrict_bytes(vb, xb, cbits // 8, timed == "symmetric"), avg_launch_us=ms_it[1] * 1e3),
            dense, post]
    for k_ in kernels:
        k_["achieved"] = k_["streamed_bytes"] / max(k_["avg_launch_us"], 1e-9) / 1e3
        k_["frac"] = k_["achieved"] / HBM_PEAK_GBS
        k_["frac_fp64_equivalent"] = k_["algorithmic_bytes_fp64"] / max(k_["avg_launch_us"], 1e-9) / 1e3 / HBM_PEAK_GBS
        t_, key_ = pmc_bytes(pmc, k_.pop("pmc_prefix"))
        p2_ = k_.pop("pmc_prefix2", None)
        if t_ is not None and p2_:
            t2_, _ = pmc_bytes(pmc, p2_)
            t_ += t2_ or 0.0
        k_["traffic"] = t_
        k_["frac_traffic"] = (t_ / max(k_["avg_launch_us"], 1e-9) / 1e3 / HBM_PEAK_GBS) if t_ is not None else None
    roofline = dict(bound="hbm",
                    kernel="%s (one tCG step: Q*z block-SpMM + Riemannian-Hessian epilogue + in-place direction / "
                           "H-direction recurrences)" % kname,
                    achieved=ach_rot, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach_rot / HBM_PEAK_GBS, traffic=traffic,
                    traffic_source=traffic_src, traffic_live=False,  # (a constant of the committed PMC summary, not of this run)
                    bytes_per_launch=hb, avg_launch_us=ms_hrot.value * 1e3,
                    protocol="HIP events over %d launches of the kernel the timed loop launches (storage selected: %s), "
                             "every operand rotating through %d private sets (> 256 MB in total): HBM-only rate"
                             % (args.spmm_reps, timed, hsets),
                    **{("%s_storage" % other): (hess_entry(other) if other in hess else None)},
                    warm=dict(avg_launch_us=ms_hess.value * 1e3, achieved=ach, frac=ach / HBM_PEAK_GBS,
                              protocol="back-to-back launches on the solver's own buffers (Infinity-Cache resident "
                                       "working set, what the tCG loop sees)"),
                    spmm_only=dict(kernel="k_spmm<%d,%d> (plain Q*X)" % (d, r), bytes_per_launch=sb,
                                   avg_launch_us=ms_rot.value * 1e3,
                                   achieved=sb / (ms_rot.value * 1e-3) / 1e9,
                                   frac=sb / (ms_rot.value * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                   buffer_sets=nsets, total_MB=nsets * set_bytes.value / 1e6,
                                   warm=dict(avg_launch_us=ms_spmm.value * 1e3,
                                             achieved=sb / (ms_spmm.value * 1e-3) / 1e9,
                                             frac=sb / (ms_spmm.value * 1e-3) / 1e9 / HBM_PEAK_GBS)),
                    spmm_symmetric=spmm_sym, spmm_storage_selected=in_use,
                    kernels=kernels, cycle_tail_us=ms_it[4] * 1e3, multilevel=ml_info)
    # the bytes the TIMED kernel's storage really moves, whichever storage that is (the plain arrays move exactly the
    # algorithmic bytes; the symmetric storage stores the upper blocks only)
    own = hb_sym_own if timed == "symmetric" else hb
    for e_, pre_ in ((roofline["spmm_only"], "k_spmm<"), (spmm_sym, "k_spmm_sym<")):
        if e_ is not None:
            e_["traffic"], _ = pmc_bytes(pmc, pre_)
    roofline.update(storage=timed, stored_bytes_per_launch=own, achieved_own_bytes=own / (ms_hrot.value * 1e-3) / 1e9,
                    frac_own_bytes=own / (ms_hrot.value * 1e-3) / 1e9 / HBM_PEAK_GBS)
    # One whole preconditioned tCG iteration of the TIMED loop: the bytes its launches stream (the Hessian-step kernel's own
    # bytes + the kernels above) over the loop's wall time per product (outer-iteration kernels and launch gaps included),
    # and the same with the committed PMC bytes.
    us_pp = 1e6 * elapsed / max(tcg_total, 1)
    it_bytes = own + sum(k_["streamed_bytes"] for k_ in kernels)
    it_traffic = (traffic + sum(k_["traffic"] for k_ in kernels)) if (traffic is not None and all(
        k_["traffic"] is not None for k_ in kernels)) else None
    roofline["iteration"] = dict(
        launches=1 + len(kernels) + sum(1 for k_ in kernels if " + " in k_["kernel"]),
        bytes=it_bytes, algorithmic_bytes_fp64=hb + sum(k_["algorithmic_bytes_fp64"] for k_ in kernels),
        traffic=it_traffic, us_per_product=us_pp, achieved=it_bytes / us_pp / 1e3, frac=it_bytes / us_pp / 1e3 / HBM_PEAK_GBS,
        frac_traffic=(it_traffic / us_pp / 1e3 / HBM_PEAK_GBS) if it_traffic is not None else None,
        kernel_us_sum=ms_hess.value * 1e3 + sum(k_["avg_launch_us"] for k_ in kernels),
        note="bytes = streamed bytes of the iteration's launches (Hessian step on its own storage + roofline.kernels); "
             "us_per_product = timed loop's wall time / its Hessian-vector products")

    # Blocks in the latency regime: the timed loop launched ONE kernel per solve (k_rtr_persist: the whole RTR solve, Q
    # resident in registers, vectors in LDS, products synchronised by an in-kernel all-reduce) -- that launch is the
    # dominant kernel of the step, and the headline figures are its: algorithmic bytes = the products it ran x the
    # tCG-step bytes above, duration = HIP events on the solver's stream around the launch (+ its 80-byte memset).
    pinfo = agent.problem.persistentInfo()
    if pinfo.get("enabled") and pinfo.get("last_members", 0) > 0:
        from dpgo_amd.solver import bench_solve
        bs = bench_solve(agent.optimizer, agent._snap, reps=20, warmup=3)
        if bs["persistent"] and bs["products"] > 0:
            solve_bytes = bs["products"] * hb
            ach_p = solve_bytes / (bs["ms"] * 1e-3) / 1e9
            roofline["multi_launch_kernel"] = dict(kernel=roofline["kernel"], achieved=roofline["achieved"],
                                                   frac=roofline["frac"], avg_launch_us=roofline["avg_launch_us"],
                                                   bytes_per_launch=hb, warm=roofline.pop("warm"),
                                                   note="the tCG-step kernel of the multi-launch scheme (blocks beyond the "
                                                        "persistent kernel's size); NOT what the timed loop ran")
            # HBM traffic of such a launch, modelled: Q, its indices, the block-Jacobi factors and G are read once, X is read
            # and written once; every tCG vector lives in registers / LDS, the exchanged vectors (z, x2, eta) are
            # write-through stores + agent-scope gathers that the Infinity Cache serves.  The kernel is bound by the latency
            # of its chip-wide reductions, so `frac` (HBM) is small by construction; the figure comparable with the
            # multi-launch kernel's is effective_algorithmic_GBs = products x the tCG step's algorithmic bytes / time.
            hbm_model = qb_ + 4 * vec + 8 * b_ * b_ * n_local
            ach_h = hbm_model / (bs["ms"] * 1e-3) / 1e9
            roofline["iteration"] = None  # (one launch per solve: products_per_launch / us_per_product below)
            roofline.update(
                kernel="k_rtr_persist<%d,%d,%d,%d> (a whole RTR solve in one launch: %d workgroups, Q in registers, iterates "
                       "in LDS, in-kernel all-reduces)" % (d, r, pinfo.get("last_split", 0), pinfo.get("last_tiles", 0),
                                                           pinfo["last_members"]),
                achieved=ach_h, frac=ach_h / HBM_PEAK_GBS, bytes_per_launch=hbm_model, avg_launch_us=bs["ms"] * 1e3,
                products_per_launch=bs["products"], us_per_product=bs["ms"] * 1e3 / bs["products"], traffic=None,
                traffic_source=None, effective_algorithmic_GBs=ach_p, effective_algorithmic_frac=ach_p / HBM_PEAK_GBS,
                effective_algorithmic_bytes=solve_bytes, storage="registers", stored_bytes_per_launch=hbm_model,
                achieved_own_bytes=ach_h, frac_own_bytes=ach_h / HBM_PEAK_GBS,
                protocol="HIP events on the solver's stream around one solve = one launch (+ its commit kernel), %d "
                         "repetitions from the benchmark's iterate; achieved / frac = MODELLED HBM bytes of the launch (Q, "
                         "indices, factors, G, X in and out) over its time: latency-bound by its chip-wide reductions (2-3 "
                         "per product), not by HBM; effective_algorithmic_* = products x the tCG step's algorithmic bytes "
                         "(SURVEY 8d) over the same time, the figure comparable with multi_launch_kernel" % 20)

    cpu = None
    jac_step = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        agent.restore()
        X_state = agent.iterate_in_caller_order().cpu().numpy()
        try:  # the device's block-Jacobi step from the same iterate: what the 1-core port below restates
            Xj = agent.X.clone()
            rj = dpgo_amd.QuadraticOptimizer(agent.problem, dpgo_amd.ROptParameters(precond="jacobi")).optimizeDevice(Xj)
            jac_step = dict(fOpt=rj.fOpt, tcg_iterations=rj.tcg_iterations)
        except Exception as exc:  # noqa: BLE001
            sys.stderr.write("bench.py: device block-Jacobi comparison step failed: %r\n" % (exc,))
        try:  # reference configuration (exact factor, 8 agents on 8 cores), from the benchmark's initial iterate
            cpu = cpu_baseline_reference(meas, n, X0, r)
        except Exception as exc:  # noqa: BLE001 -- report instead of losing the GPU measurement
            sys.stderr.write("bench.py: cpu_baseline (reference configuration) failed: %r\n" % (exc,))
        if cpu is not None:
            try:  # the same decomposition, iterate and stop on this GPU: the like-for-like pair of cpu_baseline.value
                same = gpu_same_decomposition(meas, n, X0, r, 8, args.precond, dev_index)
                cpu["gpu_same_work"] = same
                cpu["gpu_over_cpu_same_work"] = same["value"] / cpu["value"]
            except Exception as exc:  # noqa: BLE001
                sys.stderr.write("bench.py: gpu_same_work failed: %r\n" % (exc,))
        if cpu is not None:
            # the same pair from the SETTLED iterate the headline step starts from, cut into the same 8 blocks: there the tCG
            # budget (not the trust-region boundary after a step or two) ends the local solves -- the hot loop proper
            try:
                X_set = np.ascontiguousarray(np.concatenate([agents[a].in_caller_order(agents[a]._snap).cpu().numpy() for a in sorted(agents)], axis=0))
                settled_cpu = cpu_baseline_reference(meas, n, X_set, r)
                settled_cpu["sample_short"] = ("reference configuration: 8 agents x %d poses, one core each, exact sparse factor "
                                               "of Q_a+0.1I (SciPy SuperLU for CHOLMOD), RTR 3x<=50 tCG, ONE two-colour sweep from "
                                               "the SETTLED iterate" % (n // 8))
                settled_cpu["sample"] = "as cpu_baseline.sample, from the benchmark's SETTLED iterate (the state every " \
                                        "timed step restores): " + settled_cpu["sample"]
                try:
                    sg = gpu_same_decomposition(meas, n, X_set, r, 8, args.precond, dev_index, reset_auto=False)
                    sg["sample"] += "; from the settled iterate, the preconditioner selection in its steady state"
                    settled_cpu["gpu_same_work"] = sg
                    settled_cpu["gpu_over_cpu_same_work"] = sg["value"] / settled_cpu["value"]
                except Exception as exc:  # noqa: BLE001
                    sys.stderr.write("bench.py: gpu_same_work (settled) failed: %r\n" % (exc,))
                # the SETTLED pair is cpu_baseline.value (the hot loop proper: the tCG loop, not the launch overhead of a
                # handful of boundary-limited products, is what both sides spend their time in); the pair from the initial
                # iterate stays under its own key
                settled_cpu["initial_iterate"] = cpu
                cpu = settled_cpu
            except Exception as exc:  # noqa: BLE001
                sys.stderr.write("bench.py: cpu_baseline (settled iterate) failed: %r\n" % (exc,))
        try:  # the device algorithm on one core, same step as the GPU's (same settled iterate)
            port = cpu_baseline(meas, n, X_state, r, args.cpu_budget_s, "jacobi")
            if cpu is None:
                cpu = port
            else:
                cpu["single_agent_port"] = port
        except Exception as exc:  # noqa: BLE001
            sys.stderr.write("bench.py: cpu_baseline (port) failed: %r\n" % (exc,))

    also = None
    to_tol = None
    if rank == 0 and world == 1 and not args.no_secondary:
        # products-to-tolerance and time-to-gradnorm on this workload with both preconditioners, and the same for
        # sphere2500 (BASELINE configs[1]); then the fixed-work step rate of sphere2500
        to_tol = {}
        for wl in ([args.workload] + (["sphere2500", "torus3D"] if args.workload == "grid100k" else [])):
            for pc in ("auto", "multilevel", "additive", "jacobi"):
                try:  # a side measurement must never cost the main line
                    to_tol["%s/%s" % (wl, pc)] = time_to_tolerance(wl, r, pc)
                except Exception as exc:  # noqa: BLE001  ("additive" refuses blocks beyond 256 aggregates)
                    to_tol["%s/%s" % (wl, pc)] = {"error": repr(exc)}
            try:  # opt-in storage mode (NOT the headline configuration): dense level of the hierarchy kept in fp32
                to_tol["%s/multilevel+fp32_dense_level" % wl] = time_to_tolerance(wl, r, "multilevel", coarse_bits=32)
            except Exception as exc:  # noqa: BLE001
                to_tol["%s/multilevel+fp32_dense_level" % wl] = {"error": repr(exc)}
            try:  # the cycle streaming the fp64 originals of its level-0 operators instead of the default fp32 copies
                to_tol["%s/multilevel+fp64_cycle_operators" % wl] = time_to_tolerance(wl, r, "multilevel", operator_bits=64)
            except Exception as exc:  # noqa: BLE001
                to_tol["%s/multilevel+fp64_cycle_operators" % wl] = {"error": repr(exc)}
        if args.workload == "grid100k":
            also = {}
            for key, pc in (("sphere2500", "auto"), ("sphere2500_multilevel", "multilevel"), ("sphere2500_jacobi", "jacobi"),
                            ("sphere2500_additive", "additive")):
                try:
                    also[key] = secondary_single_agent("sphere2500", r, pc, args.steps, args.warmup, args.settle)
                except Exception as exc:  # noqa: BLE001
                    also[key] = {"error": repr(exc)}

    kitti = None
    if rank == 0 and world == 1 and not args.no_secondary and args.workload == "grid100k":
        try:  # BASELINE configs[4]: the re-weighted rebuild of Q each GNC outer iteration, timed
            kitti = kitti_gnc_gpu(r, dev_index)
            if not args.no_cpu_baseline:
                try:
                    kitti["cpu"] = kitti_gnc_cpu(r, budget_updates=10)
                except Exception as exc:  # noqa: BLE001
                    sys.stderr.write("bench.py: kitti_gnc CPU pair failed: %r\n" % (exc,))
        except Exception as exc:  # noqa: BLE001
            kitti = {"error": repr(exc)}

    if rank == 0:
        if cpu and cpu.get("single_agent_port", cpu).get("fOpt") is not None:
            port = cpu.get("single_agent_port", cpu)
            port["device_fOpt_same_step"] = jac_step["fOpt"] if jac_step else None
            if jac_step:
                port["rel_diff_fOpt_vs_device"] = abs(port["fOpt"] - jac_step["fOpt"]) / abs(port["fOpt"])
                port["device_tcg_iterations_same_step"] = jac_step["tcg_iterations"]
        tt_key = "%s/%s" % (args.workload, args.precond)
        tt_main = (to_tol or {}).get(tt_key) or {}
        out = {
            "metric": "rbcd_iterations_per_sec",
            "value": args.steps / elapsed,
            "unit": "it/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            # what a step is: Hessian-vector products (tCG iterations) of this rank's agents per step, and how far they
            # take the iterate; time from the initial guess to |rgrad| < 1e-2 with the same settings (single agent)
            "products_per_step": tcg_total / max(args.steps, 1),
            "gradnorm_before_step": trajectory[-1][1] if trajectory else None,
            "gradnorm_after_step": g1,
            "time_to_tolerance_ms": tt_main.get("ms") if tt_main.get("reached") else None,
            "products_to_tolerance": tt_main.get("products") if tt_main.get("reached") else None,
            # the once-per-Q cost (the reference's factorisation sits inside its first timed solve,
            # src/PoseGraph.cpp:582-586; cpu_baseline.factorisation_seconds is its pair): the hierarchy of the
            # preconditioner the solves above ran, on a fresh handle -- block pattern on the host + values on the device
            "hierarchy_setup_ms": tt_main.get("hierarchy_setup_ms"),
            "hierarchy_values_only_ms": tt_main.get("hierarchy_values_only_ms"),
            "time_to_tolerance_incl_setup_ms": tt_main.get("ms_incl_setup") if tt_main.get("reached") else None,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic" if args.workload.startswith("grid") else "g2o dataset shipped in data/",
            "config": {"workload": desc, "agents": num_agents, "agents_per_gpu": apg, "r": r, "d": d,
                       "local_solver": "RTR 3x<=50 tCG, Delta0=100, tol=1e-2 (reference defaults), precond = %s; %s" % (
                           {"jacobi": "block-Jacobi", "multilevel": "multilevel", "additive": "additive two-level",
                            "auto": "auto (library default: multilevel on a block without coupling; a coupled block "
                                    "starts on block-Jacobi and moves to the additive two-level one-launch solve once "
                                    "block-Jacobi has cost one hierarchy set-up, while that is cheaper -- dpgo_hip.h)"}[
                               args.precond],
                           ("time_to_tolerance_ms = %.3f (%d Hessian-vector products from the initial guess to |rgrad| < "
                            "1e-2, single agent, same settings)" % (tt_main["ms"], tt_main["products"]))
                           if tt_main.get("reached") else "time_to_tolerance_ms = not measured in this run"),
                       "time_to_tolerance_ms": tt_main.get("ms") if tt_main.get("reached") else None,
                       "cycle_storage": ("tCG vectors, the Hessian step, smoother factors and dense level fp64; the "
                                         "multilevel cycle streams fp%d copies of its level-0 operators (symmetric Q, A P, "
                                         "prolongation blocks), keeps its two internal vectors (pre-smoothed iterate, "
                                         "kept residual) in fp%d and its dense level (inverse, restricted residual) in "
                                         "fp%d; fp64 arithmetic throughout" % (
                                             ml_info["cycle_operator_copy_bits"], 32 if ops_state["vectors"] else 64,
                                             ml_info["coarse_inverse_bits"]))
                       if ml_info else None,
                       "cycle_storage_short": ("cycle operators fp%d, cycle vectors fp%d, dense level fp%d; tCG vectors, Hessian "
                                               "step and all arithmetic fp64" % (
                                                   ml_info["cycle_operator_copy_bits"], 32 if ops_state["vectors"] else 64,
                                                   ml_info["coarse_inverse_bits"])) if ml_info else None,
                       "precond": args.precond,
                       "hierarchy_setup_ms": tt_main.get("hierarchy_setup_ms"),
                       "time_to_tolerance_incl_setup_ms": tt_main.get("ms_incl_setup") if tt_main.get("reached") else None,
                       "products_per_step": tcg_total / max(args.steps, 1),
                       "precond_used_in_timed_steps": sorted(used_precond),
                       "selection_sweeps_before_timing": selection_sweeps,
                       "same_colour_agents": "sequential (diagnostic)" if args.sequential else "concurrent",
                       "schedule": "single agent" if num_agents == 1 else
                       "%d-colour parallel RBCD; 1 step = 1 sweep (every agent updates once; same-colour agents of a "
                       "GPU solved concurrently); public-pose exchange over %s" % (
                           plan.num_colours,
                           "RCCL p2p on the solver's stream (C ABI dpgo_comm_exchange)%s" % (
                               " through a 1-rank communicator (loop-back%s)" % (
                                   "; communicator taken from the torch.distributed process group" if use_dist else "")
                               if world == 1 else "") if comm
                           else ("device copies" if world == 1 else
                                 ("peer store: senders' pack kernels write into the receivers' hipIpc-mapped neighbour "
                                  "buffers, ordered %s (dpgo_amd/ipc.py)" % (
                                      "on the device by epoch words both sides map" if cluster.peer_store.device_ordered
                                      else "by two host barriers per exchange") if cluster.peer_store is not None else
                                  ("torch.distributed nccl p2p" if not cluster.stage else "gloo (host-staged)")))),
                       "dist_backend": backend, "transport": (args.transport if use_dist else None),
                       "poses_per_agent": n_local, "nnzb_per_agent": nnzb_local,
                       "pose_order": ("renumbered inside the agent for locality (reverse Cuthill-McKee inside each XCD's "
                                      "eighth, dpgo_locality_order); X0 in, iterates and trajectories out in the data "
                                      "set's own numbering") if agent.pose_order is not None else "as given"},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "also": also,
            "kitti_gnc": kitti,
            "quality": {"settle_iterations": settled,
                        "cost_2f_trajectory": [c for c, _ in trajectory],
                        "gradnorm_trajectory": [g for _, g in trajectory],
                        "cost_2f_after_step": 2 * f1, "gradnorm_after_step": g1,
                        "tcg_iterations_per_step_rank0": tcg_total / max(args.steps, 1),
                        "us_per_tcg_iteration_rank0": 1e6 * elapsed / max(tcg_total, 1),
                        "exchange_ms_per_step_rank0": exchange_ms / max(args.steps, 1),
                        "exchange_timing": "device time between event pairs around every exchange on its stream (no "
                                           "host synchronisation inside the timed loop)",
                        "to_tolerance": to_tol},
        }
        # Everything measured goes to bench_detail.json and to an EARLIER stdout line prefixed "DETAIL "; the LAST stdout
        # line is the driver's: one JSON object of at most 6 KB (compact_line).
        try:
            with open(os.path.join(ROOT, "bench_detail.json"), "w") as fh:
                json.dump(out, fh, indent=1)
        except OSError as exc:
            sys.stderr.write("bench.py: bench_detail.json not written: %r\n" % (exc,))
        final = ("DETAIL " + json.dumps(out), compact_line(out))
    else:
        final = None
    # The driver's line must be the LAST thing on stdout.  Libraries underneath print through C stdio (RCCL's version / path
    # banner), which is block-buffered on a pipe and would otherwise be flushed at process exit, BEHIND the line: every rank
    # flushes its C and Python buffers, the ranks meet, the process group is taken down, and only then rank 0 prints.
    import ctypes
    libc = ctypes.CDLL(None)
    sys.stdout.flush()
    libc.fflush(None)
    if use_dist:
        barrier()
        dist.destroy_process_group()
        libc.fflush(None)
    if final is not None:
        if use_dist and world > 1:
            time.sleep(0.2)  # (the other ranks' last flushes travel through the launcher's pipes)
        print(final[0])
        print(final[1])
        sys.stdout.flush()


if __name__ == "__main__":
    main()
