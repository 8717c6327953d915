"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same inputs.

Tolerances: fp64 kernels vs fp64 oracle; element-wise results must agree to 1e-11 relative
(different but equivalent summation orders), whole-solve results at matched settings to 1e-9,
and the final cost against the reference-configuration oracle (exact preconditioner) to the
1e-6 relative that BASELINE.json's north_star states.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import DATA, matrix_to_tiles, tiles_to_matrix, to_product_measurements, device_tcg_mode

pytestmark = pytest.mark.gpu
ROOT_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

RTOL_ELEM = 1e-11


def relerr(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-300))


def build_single_agent(oracle, name, r):
    import dpgo_amd
    om, n = oracle.read_g2o(os.path.join(DATA, name + ".g2o"))
    d = om.d
    Q = oracle.construct_Q(n, d, om)
    pg = dpgo_amd.PoseGraph(0, r, d)
    pg.setMeasurements(to_product_measurements(om))
    assert pg.n() == n
    prob = dpgo_amd.QuadraticProblem(pg)
    return om, n, d, Q, pg, prob


def random_point(oracle, n, d, r, seed):
    rng = np.random.default_rng(seed)
    M = rng.standard_normal((n, d + 1, r))
    return oracle.polar_project(M, d)


@pytest.mark.parametrize("name,r", [("tinyGrid3D", 5), ("smallGrid3D", 5), ("smallGrid3D", 3), ("sphere2500", 5),
                                    ("kitti_00", 5), ("kitti_00", 2), ("smallGrid3D", 4), ("smallGrid3D", 6)])
def test_problem_evaluations_match_oracle(oracle, name, r):
    """f, EucGrad, EucHessianEta, RieGrad, RieGradNorm, Riemannian Hessian, PreConditioner."""
    om, n, d, Q, pg, prob = build_single_agent(oracle, name, r)
    op = oracle.QuadraticProblem(Q, None, r, d, precond="jacobi")
    X = random_point(oracle, n, d, r, 1)
    V = np.random.default_rng(2).standard_normal((n, d + 1, r))
    Xm, Vm = tiles_to_matrix(X), tiles_to_matrix(V)
    assert abs(prob.f(Xm) - op.f(X)) <= 1e-12 * abs(op.f(X))
    assert relerr(matrix_to_tiles(prob.EucGrad(Xm), d), op.euc_grad(X)) < RTOL_ELEM
    assert relerr(matrix_to_tiles(prob.EucHessianEta(Xm, Vm), d), op.euc_hess(V)) < RTOL_ELEM
    assert relerr(matrix_to_tiles(prob.RieGrad(Xm), d), op.rie_grad(X)) < RTOL_ELEM
    assert abs(prob.RieGradNorm(Xm) - op.rie_grad_norm(X)) <= 1e-11 * op.rie_grad_norm(X)
    S = op.sym_ytg(X, op.euc_grad(X))
    Vt = oracle.tangent_project(X, V, d)
    assert relerr(matrix_to_tiles(prob.RieHessianEta(Xm, tiles_to_matrix(Vt)), d), op.rie_hess(X, S, Vt)) < RTOL_ELEM
    for pc in ("jacobi", "none"):
        op.precond = pc
        assert relerr(matrix_to_tiles(prob.PreConditioner(Xm, Vm, pc), d), op.precondition(X, V)) < RTOL_ELEM


def test_committed_golden_vectors():
    """The HIP path against the COMMITTED fixtures (tests/golden/, written by make_golden.py) with no oracle in the loop:
    Q*X, Riemannian gradient / Hessian, block-Jacobi preconditioner, qf retraction and polar projection on smallGrid3D
    (element-wise, 1e-11), the multilevel preconditioner (1e-9); f and |rgrad| at a seeded point on all five datasets; the
    RTR trace of one optimize with either preconditioner (iteration counts, cost)."""
    import json
    import dpgo_amd
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, "smallGrid3D_vectors.npz"))
    sc = json.load(open(os.path.join(GOLDEN, "golden_scalars.json")))
    meas, n = dpgo_amd.read_g2o_file(os.path.join(DATA, "smallGrid3D.g2o"))
    d, r = 3, 5
    pg = dpgo_amd.PoseGraph(0, r, d)
    pg.setMeasurements(meas)
    prob = dpgo_amd.QuadraticProblem(pg)
    X, V = tiles_to_matrix(g["X"]), tiles_to_matrix(g["V"])
    man = dpgo_amd.LiftedSEManifold(r, d, n)
    assert relerr(matrix_to_tiles(prob.PreConditioner(X, V, "multilevel"), d), g["precond_multilevel"]) < 1e-9
    for got, want in ((prob.EucHessianEta(X, X), "XQ"), (prob.RieGrad(X), "rgrad"), (prob.RieHessianEta(X, V), "rhess"),
                      (prob.PreConditioner(X, V, "jacobi"), "precond_jacobi"),
                      (man.Retraction(X, tiles_to_matrix(g["eta"])), "retract"), (man.project(tiles_to_matrix(g["M"])), "polar")):
        assert relerr(matrix_to_tiles(got, d), g[want]) < RTOL_ELEM, want
    for name, ref in sc["datasets"].items():
        m2, n2 = dpgo_amd.read_g2o_file(os.path.join(DATA, name + ".g2o"))
        assert (n2, len(m2), m2.d) == (ref["n"], ref["m"], ref["d"])
        pg2 = dpgo_amd.PoseGraph(0, 5, m2.d)
        pg2.setMeasurements(m2)
        assert len(pg2.quadraticMatrix()[1]) == ref["nnzb"]
        assert abs(pg2.quadraticMatrix()[2].sum() - ref["Q_sum"]) <= 1e-9 * ref["Q_abs_sum"]
        M = np.random.default_rng(ref["seed"]).standard_normal((n2, m2.d + 1, 5))
        Xr = dpgo_amd.LiftedSEManifold(5, m2.d, n2).project(tiles_to_matrix(M))  # the fixture's seeded point
        p2 = dpgo_amd.QuadraticProblem(pg2)
        # f is a cancellation-heavy sum when kappa is large (kitti_00): tolerance relative to sum |Q|
        assert abs(p2.f(Xr) - ref["f_random"]) <= 1e-10 * max(abs(ref["f_random"]), 1e-3 * ref["Q_abs_sum"]), name
        assert abs(p2.RieGradNorm(Xr) - ref["gradnorm_random"]) <= 1e-9 * ref["gradnorm_random"], name
    from dpgo_amd.initialization import chordal_initialization
    from dpgo_amd import synthetic
    tr = sc["smallGrid3D_rtr_trace_jacobi"]
    opt = dpgo_amd.QuadraticOptimizer(prob, dpgo_amd.ROptParameters(precond="jacobi"))
    opt.optimize(tiles_to_matrix(synthetic.lift_tiles(chordal_initialization(meas, n), r)))
    res = opt.getOptResult()
    assert (res.tcg_iterations, res.rtr_iterations) == (tr["tcg_iters"], tr["outer_iters"])
    assert abs(res.fInit - tr["fInit"]) <= 1e-8 * abs(tr["fInit"]) and abs(res.fOpt - tr["fOpt"]) <= 1e-8 * abs(tr["fOpt"])
    trm = sc["smallGrid3D_rtr_trace_multilevel"]  # the multilevel preconditioner (default hierarchy)
    optm = dpgo_amd.QuadraticOptimizer(prob, dpgo_amd.ROptParameters(precond="multilevel"))
    optm.optimize(tiles_to_matrix(synthetic.lift_tiles(chordal_initialization(meas, n), r)))
    resm = optm.getOptResult()
    assert prob.multilevelInfo()["ks"] == trm["ks"]
    assert (resm.tcg_iterations, resm.rtr_iterations) == (trm["tcg_iters"], trm["outer_iters"])
    assert abs(resm.fOpt - trm["fOpt"]) <= 1e-8 * abs(trm["fOpt"])


@pytest.mark.parametrize("d,r,n", [(3, 5, 1000), (3, 3, 17), (2, 2, 64), (2, 5, 333), (3, 6, 129), (3, 5, 1)])
def test_manifold_ops_match_oracle(oracle, d, r, n):
    """LiftedSEManifold::project (tests/testUtils.cpp:40-54 tolerance 1e-5 on Y^T Y = I; here also
    element-wise vs the SVD oracle), tangent projection, qf retraction."""
    import dpgo_amd
    rng = np.random.default_rng(5)
    M = rng.uniform(-1, 1, (n, d + 1, r))  # Matrix::Random, testUtils.cpp:45
    man = dpgo_amd.LiftedSEManifold(r, d, n)
    Xp = matrix_to_tiles(man.project(tiles_to_matrix(M)), d)
    Y = Xp[:, :d, :]
    assert np.abs(Y @ np.swapaxes(Y, 1, 2) - np.eye(d)).max() <= 1e-12
    assert relerr(Xp, oracle.polar_project(M, d)) < 1e-10
    X = oracle.polar_project(M, d)
    V = rng.standard_normal((n, d + 1, r))
    assert relerr(matrix_to_tiles(man.Projection(tiles_to_matrix(X), tiles_to_matrix(V)), d),
                  oracle.tangent_project(X, V, d)) < RTOL_ELEM
    eta = 0.3 * oracle.tangent_project(X, V, d)
    Xr = matrix_to_tiles(man.Retraction(tiles_to_matrix(X), tiles_to_matrix(eta)), d)
    assert relerr(Xr, oracle.qf_retract(X, eta, d)) < RTOL_ELEM
    Yr = Xr[:, :d, :]
    assert np.abs(Yr @ np.swapaxes(Yr, 1, 2) - np.eye(d)).max() <= 1e-12


@pytest.mark.parametrize("workgroups,pay", [(196, 20), (157, 20), (64, 9), (256, 24), (40, 15), (7, 6)])
def test_in_kernel_reduction_primitives(workgroups, pay):
    """The communication primitives of the one-launch solve on their own (dpgo_debug_reduction_primitives; no reference
    counterpart: they stand in for ROPTLIB's serial dot products inside tCG_TR).  chip_allreduce<2, PAY>: three chip-wide
    reductions of two sums over 256 threads x `workgroups`, each carrying a payload of PAY = (d+1) r doubles per workgroup
    -- the all-gather the additive preconditioner's restricted vectors ride on (round 6): every workgroup ends with the
    SAME bits, the sums agree with numpy, and thread t of every workgroup holds participant t's payload exactly (the
    per-wave parts added in wave order).  wave_reduce_rows: the reduce-scatter of PAY values over a wavefront
    (v_permlane32_swap / v_permlane16_swap pair folds + DPP row shifts) against numpy, every value in its place."""
    import torch
    import dpgo_amd.lib as L
    lib = L.load()
    if workgroups > torch.cuda.get_device_properties(0).multi_processor_count:
        pytest.skip("needs %d compute units" % workgroups)
    rng = np.random.default_rng(workgroups * 100 + pay)
    steps = 3
    a = rng.standard_normal((workgroups, 256, 2))
    pw = rng.standard_normal((workgroups, 4, pay))
    dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()  # noqa: E731
    a_d, pw_d = dev(a), dev(pw)
    sums = torch.zeros((workgroups, steps, 2), dtype=torch.float64, device="cuda")
    pout = torch.zeros((workgroups, steps, workgroups, pay), dtype=torch.float64, device="cuda")
    rows = torch.zeros((workgroups, 4, pay), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    L.check(lib.dpgo_debug_reduction_primitives(workgroups, pay, steps, L.ptr(a_d), L.ptr(pw_d), L.ptr(sums), L.ptr(pout),
                                                 L.ptr(rows)))
    sums, pout, rows = sums.cpu().numpy(), pout.cpu().numpy(), rows.cpu().numpy()
    for s in range(steps):
        want = np.array([(a[:, :, 0] * (s + 1)).sum(), (a[:, :, 1] - s).sum()])
        assert np.abs(sums[0, s] - want).max() <= 1e-11 * (np.abs(a).sum() + steps * a[:, :, 0].size)
        assert (sums[:, s] == sums[0, s]).all()  # identical bits in every workgroup
        p = pw + s
        gathered = ((p[:, 0] + p[:, 1]) + p[:, 2]) + p[:, 3]  # the waves in order
        assert (pout[:, s] == gathered[None]).all()
    e = np.arange(1, pay + 1)[None, None, :]
    v = a[:, :, 0:1] * e + a[:, :, 1:2]                       # [workgroups][256][pay]
    want_rows = v.reshape(workgroups, 4, 64, pay).sum(axis=2)
    assert np.abs(rows - want_rows).max() <= 1e-12 * np.abs(v).reshape(workgroups, 4, 64, pay).sum(axis=2).max()


@pytest.mark.parametrize("name,precond", [("sphere2500", "additive"), ("grid:25x25x10", "additive"), ("grid:25x25x10", "jacobi")])
def test_one_launch_solve_does_not_depend_on_timing(oracle, name, precond):
    """The in-kernel reductions of the one-launch solve add their operands in a fixed order whatever arrives first
    (kernels/persist.h, chip_allreduce: thread t sweeps participant t, fixed trees) -- also the one that carries the
    additive preconditioner's payload, and the gathers that take the workgroup's own tiles from LDS read what the memory
    copy holds.  So WHEN a sweep looks must not change a bit: three optimize calls with the polling knobs at their
    defaults, polling at once (DPGO_POLL_FIRST = DPGO_POLL_FIRST_PAY = 0, DPGO_POLL_SLEEP = 0) and late (90 / 120 / 7)
    leave bitwise the same iterates and result records.  (What ROPTLIB guarantees trivially with serial dot products.)"""
    import hashlib
    import torch
    import dpgo_amd
    r = 5
    if name.startswith("grid:"):
        om, n, Ttrue = oracle.synthetic_grid(*[int(v) for v in name[5:].split("x")], seed=0)
        X0 = oracle.lift(oracle.perturbed_truth(Ttrue, seed=2), r)
    else:
        om, n = oracle.read_g2o(os.path.join(DATA, name + ".g2o"))
        X0 = oracle.lift(oracle.chordal_initialization(om, n), r)
    d = om.d
    lib = dpgo_amd.lib.load()
    knobs = ("DPGO_POLL_FIRST", "DPGO_POLL_FIRST_PAY", "DPGO_POLL_SLEEP")
    saved = {k: os.environ.get(k) for k in knobs}
    digests = []
    try:
        for sw in ({}, {"DPGO_POLL_FIRST": "0", "DPGO_POLL_FIRST_PAY": "0", "DPGO_POLL_SLEEP": "0"},
                   {"DPGO_POLL_FIRST": "90", "DPGO_POLL_FIRST_PAY": "120", "DPGO_POLL_SLEEP": "7"}):
            for k in knobs:
                os.environ.pop(k, None)
            os.environ.update(sw)
            dpgo_amd.lib.check(lib.dpgo_options_reload())
            pg = dpgo_amd.PoseGraph(0, r, d)
            pg.setMeasurements(to_product_measurements(om))
            prob = dpgo_amd.QuadraticProblem(pg)
            opt = dpgo_amd.QuadraticOptimizer(prob, dpgo_amd.ROptParameters(precond=precond))
            Xd = torch.tensor(X0, device="cuda", dtype=torch.float64)
            h = hashlib.sha256()
            for call in range(3):
                res = opt.optimizeDevice(Xd)
                assert res.precond_used == precond and prob.persistentInfo()["last_members"] > 0, (sw, res)
                h.update(Xd.cpu().numpy().tobytes())
                h.update(repr((res.tcg_iterations, res.rtr_iterations, res.tCGStatus, res.fOpt, res.gradNormOpt)).encode())
            digests.append(h.hexdigest())
            del opt, prob
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        lib.dpgo_options_reload()
    assert digests[0] == digests[1] == digests[2], digests


def test_polar_projection_of_rank_deficient_blocks_is_finite(oracle):
    """LiftedSEManifold::project on blocks without full column rank (a zero block, a rank-1 block, a block with two
    parallel columns): the reference's JacobiSVD U V^T (src/DPGO_utils.cpp:480-486) stays finite there; so does the
    device's M (M^T M)^-1/2 (singular values clamped at 1e-14 sigma_max).  Full-rank blocks next to them are exact."""
    import dpgo_amd
    d, r, n = 3, 5, 40
    rng = np.random.default_rng(12)
    M = rng.standard_normal((n, d + 1, r))
    M[3, :d] = 0.0
    M[7, 1] = 2.0 * M[7, 0]
    M[7, 2] = -M[7, 0]
    M[11, 2] = M[11, 1]
    out = matrix_to_tiles(dpgo_amd.LiftedSEManifold(r, d, n).project(tiles_to_matrix(M)), d)
    assert np.isfinite(out).all()
    good = np.setdiff1d(np.arange(n), [3, 7, 11])
    assert relerr(out[good], oracle.polar_project(M, d)[good]) < 1e-10
    assert np.abs(out[:, d] - M[:, d]).max() == 0.0


@pytest.mark.parametrize("name,r,precond", [("smallGrid3D", 5, "jacobi"), ("smallGrid3D", 5, "none"),
                                            ("sphere2500", 5, "jacobi"), ("tinyGrid3D", 3, "jacobi"),
                                            ("kitti_00", 5, "jacobi"), ("torus3D", 5, "jacobi")])
def test_optimize_matches_oracle_at_matched_settings(oracle, name, r, precond):
    """One QuadraticOptimizer::optimize call with the reference defaults (RTR 3 x <=50 tCG,
    Delta0 = 100, tol 1e-2).  Same preconditioner and the same H-delta recurrence on both sides =>
    same trajectory (iteration counts, tCG status, iterate to 1e-7).  Against the oracle in the
    reference's own arithmetic (H applied to delta directly) the result must still agree closely."""
    import dpgo_amd
    om, n, d, Q, pg, prob = build_single_agent(oracle, name, r)
    if name == "tinyGrid3D":
        T = oracle.odometry_initialization(om.subset(np.nonzero(om.p1 + 1 == om.p2)[0]), n)
    else:
        T = oracle.chordal_initialization(om, n)
    X0 = oracle.lift(T, r)
    op = oracle.QuadraticProblem(Q, None, r, d, precond=precond)
    oopt = oracle.QuadraticOptimizer(op, oracle.ROptParameters(), hess_recurrence=device_tcg_mode(n, d, r))
    Xo = oopt.optimize(X0)
    gopt = dpgo_amd.QuadraticOptimizer(prob, dpgo_amd.ROptParameters(precond=precond))
    Xg = matrix_to_tiles(gopt.optimize(tiles_to_matrix(X0)), d)
    ro, rg = oopt.result, gopt.getOptResult()
    assert rg.success
    # f is a cancellation-heavy sum when kappa is large (kitti_00: kappa ~ 3e5, neighbouring poses
    # nearly equal): the fp64 error of either side scales with |X|^T |Q| |X|, not with |f|
    Xa = np.abs(X0).reshape(-1, r)
    scale = float((Xa * (abs(op.Qs) @ Xa)).sum())
    assert abs(rg.fInit - ro.fInit) <= 1e-14 * scale
    assert abs(rg.gradNormInit - ro.gradNormInit) <= 1e-10 * ro.gradNormInit
    assert rg.rtr_iterations == ro.outer_iters
    assert rg.tcg_iterations == ro.tcg_iters
    assert rg.tCGStatus == oracle.TCG_NAMES[ro.tCGStatus]
    assert abs(rg.fOpt - ro.fOpt) <= 1e-9 * abs(ro.fOpt) + 1e-14 * scale
    assert relerr(Xg, Xo) < 1e-7
    # and the problem object agrees with the optimizer's own statistics (QuadraticOptimizer.cpp:42-43)
    assert abs(prob.f(tiles_to_matrix(Xg)) - rg.fOpt) <= 1e-12 * abs(rg.fOpt) + 1e-14 * scale
    # reference arithmetic (no recurrence): same decisions, same result up to round-off amplification
    ref = oracle.QuadraticOptimizer(oracle.QuadraticProblem(Q, None, r, d, precond=precond), oracle.ROptParameters())
    Xr = ref.optimize(X0)
    assert ref.result.outer_iters == rg.rtr_iterations
    assert abs(rg.fOpt - ref.result.fOpt) <= 1e-7 * abs(ref.result.fOpt) + 1e-14 * scale
    assert relerr(Xg, Xr) < 1e-5


@pytest.mark.parametrize("name,ref2f,precond", [
    ("smallGrid3D", 1025.3980556263, "multilevel"), ("sphere2500", 1687.0058142808, "multilevel"),
    ("torus3D", 24227.0455583823, "multilevel"), ("smallGrid3D", 1025.3980556263, "jacobi"),
    ("sphere2500", 1687.0058142808, "jacobi"), ("torus3D", 24227.0455583823, "jacobi"),
    # kitti_00 (2-D, a 4 541-pose chain with 136 loop closures; BASELINE.md section 2: 2 f* = 125.6807087875): with the
    # multilevel cycle and with the default selection (block-Jacobi needs tens of thousands of products on a chain)
    ("kitti_00", 125.6807087875, "multilevel"), ("kitti_00", 125.6807087875, "auto")])
def test_final_cost_matches_reference_configuration(oracle, name, ref2f, precond):
    """north_star: final cost matches the reference CPU solver's on the same .g2o to 1e-6 relative.
    Both sides run RTR to a tight gradient norm from the chordal initialisation; the oracle uses the
    reference's exact (Q + 0.1 I)^-1 preconditioner, the device path its default (multilevel) or block-Jacobi."""
    import dpgo_amd
    r = 5
    om, n, d, Q, pg, prob = build_single_agent(oracle, name, r)
    X0 = oracle.lift(oracle.chordal_initialization(om, n), r)
    prm_o = oracle.ROptParameters(gradnorm_tol=1e-4, RTR_iterations=60, RTR_tCG_iterations=500)
    oopt = oracle.QuadraticOptimizer(oracle.QuadraticProblem(Q, None, r, d, precond="exact"), prm_o)
    oopt.optimize(X0)
    prm_g = dpgo_amd.ROptParameters(precond=precond, gradnorm_tol=1e-4, RTR_iterations=60, RTR_tCG_iterations=500,
                                    time_bound_s=120.0)
    gopt = dpgo_amd.QuadraticOptimizer(prob, prm_g)
    gopt.optimize(tiles_to_matrix(X0))
    fo, fg = oopt.result.fOpt, gopt.getOptResult().fOpt
    assert abs(fg - fo) <= 1e-6 * abs(fo)
    assert abs(2 * fg - ref2f) <= 1e-6 * ref2f  # literature optimum
    # (kitti_00: kappa up to 1e5 makes f a cancellation-heavy sum; below |rgrad| ~ 0.05 the rho test compares decreases
    # under the round-off of f and accept / reject is noise on either side -- the oracle's last step there has rho = 126)
    res = gopt.getOptResult()
    assert res.gradNormOpt < (1e-3 if name != "kitti_00" else 5e-2)


def test_final_cost_of_the_multi_agent_configuration_matches_reference(oracle):
    """BASELINE configs[2] as north_star states it: torus3D cut into 8 agents, final cost against the reference
    configuration to 1e-6 relative.  Reference side: the oracle's coloured RBCD with the EXACT (Q_a + 0.1 I)^-1
    preconditioner, 120 two-colour sweeps from the chordal initialisation -- generated once by tests/golden/make_golden.py
    and committed as scalars (2f = 24227.0719 after sweep 119; SURVEY 8e's probe: 24227.072, 1.1e-6 above the optimum
    24227.0456).  Device side: the same 8 blocks and 120 sweeps with the library's DEFAULT preconditioner selection."""
    import json
    import dpgo_amd
    from dpgo_amd.agent import DeviceAgent, ExchangePlan, RBCDCluster, build_pose_graphs
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_scalars.json")))
    ref = gold["torus3D_8_agents_120_sweeps_exact"]
    assert abs(ref["cost_2f"]["119"] - 24227.072) < 1e-3  # the figure BASELINE.md / SURVEY 8e quote
    r, robots = 5, 8
    om, n = oracle.read_g2o(os.path.join(DATA, "torus3D.g2o"))
    X0 = oracle.lift(oracle.chordal_initialization(om, n), r)
    ranges, graphs = build_pose_graphs(to_product_measurements(om), n, robots, r)
    plan = ExchangePlan(graphs)
    agents = {a: DeviceAgent(graphs, plan, a, X0[ranges[a][0]:ranges[a][1]], dpgo_amd.ROptParameters()) for a in range(robots)}
    cluster = RBCDCluster(plan, agents)
    used = set()
    for k in range(120):
        cluster.sweep()
        used |= {ag.last_result.precond_used for ag in agents.values()}
        if str(k) in ref["cost_2f"]:
            f, g = cluster.central_cost_and_gradnorm()
            # early sweeps: the two preconditioners take different local steps (1e-5 apart); the north_star bound applies
            # to the final cost
            assert abs(2 * f - ref["cost_2f"][str(k)]) <= (1e-6 if k == 119 else 1e-5) * ref["cost_2f"][str(k)], (k, 2 * f)
    assert abs(g - ref["gradnorm_after"]) <= 0.05 * ref["gradnorm_after"]
    assert abs(2 * f - ref["optimum_2f"]) <= 2e-6 * ref["optimum_2f"] and 2 * f > ref["optimum_2f"]
    assert used <= {"jacobi", "additive", "multilevel"} and "jacobi" in used


def test_feed_modes_and_single_iteration_radius_shrink(oracle):
    """The just-in-time kernel feed (default) and the polling feed run the same device arithmetic;
    RTR_iterations == 1 takes the radius-shrinking branch of trustRegion (src/QuadraticOptimizer.cpp:80-99)."""
    import dpgo_amd
    om, n, d, Q, pg, prob = build_single_agent(oracle, "sphere2500", 5)
    X0 = tiles_to_matrix(oracle.lift(oracle.chordal_initialization(om, n), 5))
    outs = []
    prob.setPersistent(False)  # the two feeds of the multi-launch scheme (a block this size otherwise solves in one launch)
    for poll in (0, 1, 8):
        opt = dpgo_amd.QuadraticOptimizer(prob, dpgo_amd.ROptParameters(precond="jacobi", tcg_poll_interval=poll))
        outs.append((opt.optimize(X0), opt.getOptResult()))
    for X, res in outs[1:]:
        assert np.array_equal(X, outs[0][0])
        assert (res.tcg_iterations, res.rtr_iterations, res.fOpt) == (outs[0][1].tcg_iterations,
                                                                      outs[0][1].rtr_iterations, outs[0][1].fOpt)
    # ... and the one-launch solve (k_rtr_persist): same algorithm, its own summation order
    prob.setPersistent(True)
    opt = dpgo_amd.QuadraticOptimizer(prob, dpgo_amd.ROptParameters(precond="jacobi"))
    Xp, resp = opt.optimize(X0), opt.getOptResult()
    assert prob.persistentInfo()["last_members"] > 0
    assert (resp.tcg_iterations, resp.rtr_iterations) == (outs[0][1].tcg_iterations, outs[0][1].rtr_iterations)
    assert abs(resp.fOpt - outs[0][1].fOpt) <= 1e-9 * abs(outs[0][1].fOpt)
    assert np.max(np.abs(Xp - outs[0][0])) < 1e-7
    # single-iteration mode: a tiny initial radius is accepted at once, result equals the oracle's
    prm_o = oracle.ROptParameters(RTR_iterations=1, RTR_initial_radius=1.0)
    oo = oracle.QuadraticOptimizer(oracle.QuadraticProblem(Q, None, 5, d, precond="jacobi"), prm_o, hess_recurrence=device_tcg_mode(n, d, 5))
    Xo = oo.optimize(matrix_to_tiles(X0, d))
    go = dpgo_amd.QuadraticOptimizer(prob, dpgo_amd.ROptParameters(precond="jacobi", RTR_iterations=1, RTR_initial_radius=1.0))
    Xg = matrix_to_tiles(go.optimize(X0), d)
    assert go.getOptResult().latest_step_accepted
    assert go.getOptResult().tCGStatus == oracle.TCG_NAMES[oo.result.tCGStatus] == "EXCREGION"
    assert relerr(Xg, Xo) < 1e-9


def test_triangle_graph_known_answer(oracle):
    """tests/testTriangleGraph.cpp:57 restated: noise-free 3-pose triangle, r = d = 3, kappa = tau = 1;
    chordal init + RTR returns Ttrue to 1e-4 (Frobenius)."""
    import dpgo_amd
    from test_oracle import triangle_graph
    om, Ttrue = triangle_graph(oracle)
    d = r = 3
    pg = dpgo_amd.PoseGraph(0, r, d)
    pg.setMeasurements(to_product_measurements(om))
    prob = dpgo_amd.QuadraticProblem(pg)
    T0 = oracle.chordal_initialization(om, 3)
    opt = dpgo_amd.QuadraticOptimizer(prob, dpgo_amd.ROptParameters(precond="jacobi"))
    Topt = opt.optimize(tiles_to_matrix(oracle.lift(T0, r)))
    # express in the frame of pose 0 (PGOAgent::getTrajectoryInLocalFrame)
    Tt = matrix_to_tiles(Topt, d)
    R0 = Tt[0, :d, :].T
    t0 = Tt[0, d, :]
    out = np.zeros((d, 3 * (d + 1)))
    for i in range(3):
        Ri = Tt[i, :d, :].T
        out[:, i * (d + 1):i * (d + 1) + d] = R0.T @ Ri
        out[:, i * (d + 1) + d] = R0.T @ (Tt[i, d, :] - t0)
    assert np.linalg.norm(out - Ttrue) <= 1e-4


def test_optimization_thread_and_line_graph_known_answers(oracle):
    """tests/testOptimizationThread.cpp:29-92 and tests/testLineGraph.cpp through the HIP path (DeviceAgent = the part of
    PGOAgent that drives it; device chordal initialisation, device rounding).  Triangle, d = r = 3: the trajectory in
    the local frame is Ttrue to 1e-4 after initialize() and after the optimisation loop (20 iterate() calls).  Line of
    five poses with one random translation repeated: the graph is a tree, so the initial guess is exact -- cost and
    gradient 0, the iterate unchanged, the rounded trajectory i * t (the reference asserts only the getters)."""
    import torch
    import dpgo_amd
    from dpgo_amd.agent import DeviceAgent, ExchangePlan, build_pose_graphs
    from dpgo_amd.initialization import chordal_initialization, odometry_initialization
    from dpgo_amd import synthetic
    from test_oracle import triangle_graph
    om, Ttrue = triangle_graph(oracle)
    pm = to_product_measurements(om)
    T0 = chordal_initialization(pm, 3)
    ranges, graphs = build_pose_graphs(pm, 3, 1, 3)
    agent = DeviceAgent(graphs, ExchangePlan(graphs), 0, synthetic.lift_tiles(T0, 3), dpgo_amd.ROptParameters(precond="jacobi"))

    def local(agent):
        T = agent.getTrajectoryInLocalFrame().cpu().numpy()  # tiles [n, d+1, d]
        return T.transpose(2, 0, 1).reshape(T.shape[2], -1)

    assert np.linalg.norm(local(agent) - Ttrue) <= 1e-4       # :80
    assert (agent.id, agent.n, agent.d, agent.r) == (0, 3, 3, 3)  # :85-88
    for _ in range(20):
        agent.iterate(True)
    assert np.linalg.norm(local(agent) - Ttrue) <= 1e-4       # :92
    rng = np.random.default_rng(3)
    t = rng.uniform(-1, 1, 3)
    z = np.zeros(4, dtype=np.int64)
    oml = oracle.Measurements(3, z, np.arange(4), z.copy(), np.arange(1, 5), np.tile(np.eye(3), (4, 1, 1)),
                              np.tile(t, (4, 1)), np.ones(4), np.ones(4), np.ones(4), np.ones(4, dtype=bool))
    pml = to_product_measurements(oml)
    for init in (chordal_initialization, odometry_initialization):
        X0 = synthetic.lift_tiles(init(pml, 5), 3)
        ranges, graphs = build_pose_graphs(pml, 5, 1, 3)
        ag = DeviceAgent(graphs, ExchangePlan(graphs), 0, X0, dpgo_amd.ROptParameters(precond="jacobi"))
        ag.iterate(True)
        res = ag.last_result
        assert (ag.id, ag.n, ag.d, ag.r) == (0, 5, 3, 3)      # what tests/testLineGraph.cpp asserts
        assert res.fInit < 1e-14 and res.gradNormInit < 1e-8 and np.abs(ag.X.cpu().numpy() - X0).max() < 1e-9
        T = ag.getTrajectoryInLocalFrame().cpu().numpy()
        for i in range(5):
            assert np.abs(T[i, :3] - np.eye(3)).max() < 1e-8 and np.abs(T[i, 3] - i * t).max() < 1e-8


def test_prior_known_answer(oracle):
    """tests/testPGO.cpp:131-190 (testPrior): 2-pose graph + prior on pose 1; RTR 50 x 500,
    tol 1e-5 => both poses equal the prior to 1e-6."""
    import dpgo_amd
    from test_oracle import prior_problem
    om, prior_pose, T0 = prior_problem(oracle)
    d = r = 3
    pg = dpgo_amd.PoseGraph(0, r, d)
    pg.setMeasurements(to_product_measurements(om))
    pg.setPrior(1, prior_pose)
    prob = dpgo_amd.QuadraticProblem(pg)
    prm = dpgo_amd.ROptParameters(precond="jacobi", RTR_iterations=50, RTR_tCG_iterations=500, gradnorm_tol=1e-5)
    opt = dpgo_amd.QuadraticOptimizer(prob, prm)
    X0 = tiles_to_matrix(oracle.lift(T0, r))
    assert np.linalg.norm(X0[:, 0:4] - prior_pose) > 1e-6
    Topt = opt.optimize(X0)
    assert np.linalg.norm(Topt[:, 0:4] - prior_pose) < 1e-6
    assert np.linalg.norm(Topt[:, 4:8] - prior_pose) < 1e-6


def test_rgd_step_matches_oracle(oracle):
    """QuadraticOptimizer::gradientDescent (src/QuadraticOptimizer.cpp:110-137)."""
    import dpgo_amd
    om, n, d, Q, pg, prob = build_single_agent(oracle, "smallGrid3D", 5)
    X0 = oracle.lift(oracle.chordal_initialization(om, n), 5)
    for use_pc in (True, False):
        op = oracle.QuadraticProblem(Q, None, 5, d, precond="jacobi")
        oo = oracle.QuadraticOptimizer(op, oracle.ROptParameters(method="RGD", RGD_use_preconditioner=use_pc))
        Xo = oo.optimize(X0)
        go = dpgo_amd.QuadraticOptimizer(prob, dpgo_amd.ROptParameters(precond="jacobi", method="RGD", RGD_use_preconditioner=use_pc))
        Xg = matrix_to_tiles(go.optimize(tiles_to_matrix(X0)), d)
        assert relerr(Xg, Xo) < RTOL_ELEM
        assert abs(go.getOptResult().fOpt - oo.result.fOpt) <= 1e-12 * abs(oo.result.fOpt)


def test_multi_agent_G_and_local_problem(oracle):
    """constructQ / constructG for an agent with shared edges (src/PoseGraph.cpp:381-580):
    the agent-local cost gradient equals the agent's block of the central Riemannian gradient
    (SURVEY 8c'), G built on the device from the neighbour tile buffer equals the oracle's."""
    import torch
    import dpgo_amd
    r = 5
    om, n = oracle.read_g2o(os.path.join(DATA, "smallGrid3D.g2o"))
    d = om.d
    ranges, per = oracle.partition_contiguous(om, n, 5)
    dataset, _ = dpgo_amd.read_g2o_file(os.path.join(DATA, "smallGrid3D.g2o"))
    ranges_p, per_p = dpgo_amd.partition_contiguous(dataset, n, 5)
    assert ranges_p == ranges
    X = random_point(oracle, n, d, r, 3)
    central = oracle.QuadraticProblem(oracle.construct_Q(n, d, om), None, r, d)
    RGc = central.rie_grad(X)
    for a in range(5):
        s, e = ranges[a]
        na = e - s
        priv = oracle.Measurements.concat([per[a]["odometry"], per[a]["private"]])
        Qa = oracle.construct_Q(na, d, priv, per[a]["shared"], my_id=a)
        pg = dpgo_amd.PoseGraph(a, r, d)
        pg.setMeasurements(per_p[a])
        assert pg.n() == na
        rp, ci, v = pg.quadraticMatrix()
        assert np.array_equal(rp, Qa.rowptr) and np.array_equal(ci, Qa.colidx)
        assert np.abs(v - Qa.vals).max() <= 1e-12 * np.abs(Qa.vals).max()
        nbr = {}
        for (rob, fr) in pg.neighborPoseIDs():
            nbr[(rob, fr)] = X[ranges[rob][0] + fr]  # tile [b, r]
        Ga = oracle.construct_G(na, d, r, per[a]["shared"], a, nbr)
        # host flavour: PoseGraph::setNeighborPoses + linearMatrix
        pg.setNeighborPoses({k: v_.T for k, v_ in nbr.items()})  # LiftedPose r x (d+1)
        assert relerr(matrix_to_tiles(pg.linearMatrix(), d), Ga) < 1e-13
        prob = dpgo_amd.QuadraticProblem(pg)
        Xa = X[s:e]
        assert relerr(matrix_to_tiles(prob.RieGrad(tiles_to_matrix(Xa)), d), RGc[s:e]) < 1e-10
        # device flavour: G = G0 + Xnbr * C from the tile buffer
        slots = prob.setCouplingFromPoseGraph()
        buf = torch.tensor(np.stack([nbr[sid] for sid in slots]), device="cuda", dtype=torch.float64).contiguous()
        prob.setStream(torch.cuda.current_stream().cuda_stream)
        prob.updateLinearMatrixFromNeighbors(buf)
        Xd = torch.tensor(np.ascontiguousarray(Xa), device="cuda", dtype=torch.float64)
        f_dev, gn_dev = prob.evalDevice(Xd)
        torch.cuda.synchronize()
        pa = oracle.QuadraticProblem(Qa, Ga, r, d)
        assert abs(f_dev - pa.f(Xa)) <= 1e-12 * abs(pa.f(Xa))
        assert abs(gn_dev - np.linalg.norm(RGc[s:e])) <= 1e-10 * np.linalg.norm(RGc[s:e])


def test_spmm_properties_at_full_size(oracle):
    """BASELINE.json config 4 size (100k-pose grid): size-independent properties of the Q*X kernel --
    linearity, symmetry <X, YQ> = <Y, XQ>, and agreement with the oracle's CSR product."""
    import torch
    import dpgo_amd
    meas, n, Ttrue = oracle.synthetic_grid(50, 50, 40, seed=0)
    d, r = 3, 5
    pg = dpgo_amd.PoseGraph(0, r, d)
    pg.setMeasurements(to_product_measurements(meas))
    assert pg.n() == 100000
    rp, ci, v = pg.quadraticMatrix()
    assert len(ci) == 687000  # nnzb = n + 2 * 293500
    prob = dpgo_amd.QuadraticProblem(pg)
    prob.setStream(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device="cuda").manual_seed(1)
    X = torch.randn((n, d + 1, r), device="cuda", dtype=torch.float64, generator=g)
    Y = torch.randn((n, d + 1, r), device="cuda", dtype=torch.float64, generator=g)
    XQ, YQ, ZQ = torch.empty_like(X), torch.empty_like(X), torch.empty_like(X)
    prob.spmmDevice(X, XQ)
    prob.spmmDevice(Y, YQ)
    Z = (2.5 * X - 0.75 * Y).contiguous()
    prob.spmmDevice(Z, ZQ)
    torch.cuda.synchronize()
    lin = (ZQ - (2.5 * XQ - 0.75 * YQ)).norm() / ZQ.norm()
    assert lin.item() < 1e-13
    sym = abs((X * YQ).sum() - (Y * XQ).sum()) / abs((X * YQ).sum())
    assert sym.item() < 1e-11
    Qs = oracle.BSR(n, d + 1, rp, ci, v).to_scipy().tocsr()
    ref = (Qs @ X.cpu().numpy().reshape(n * (d + 1), r)).reshape(n, d + 1, r)
    assert relerr(XQ.cpu().numpy(), ref) < 1e-13


def _grid2d_measurements(oracle, nx, ny, seed):
    """SE(2) measurements on an nx x ny lattice (snake odometry + the lattice's other edges), random relative poses."""
    rng = np.random.default_rng(seed)
    idx = np.arange(nx * ny).reshape(ny, nx)
    idx[1::2] = idx[1::2, ::-1].copy()  # boustrophedon numbering: consecutive indices are lattice neighbours
    pairs = set()
    for y in range(ny):
        for x in range(nx):
            if x + 1 < nx:
                pairs.add((min(idx[y, x], idx[y, x + 1]), max(idx[y, x], idx[y, x + 1])))
            if y + 1 < ny:
                pairs.add((min(idx[y, x], idx[y + 1, x]), max(idx[y, x], idx[y + 1, x])))
    pairs = np.array(sorted(pairs), dtype=np.int32)
    m = len(pairs)
    th = rng.uniform(-np.pi, np.pi, m)
    R = np.stack([np.stack([np.cos(th), -np.sin(th)], -1), np.stack([np.sin(th), np.cos(th)], -1)], -2)
    z = np.zeros(m, dtype=np.int32)
    return oracle.Measurements(2, z, pairs[:, 0].copy(), z.copy(), pairs[:, 1].copy(), R, rng.standard_normal((m, 2)),
                               rng.uniform(1.0, 50.0, m), rng.uniform(1.0, 50.0, m), np.ones(m),
                               np.zeros(m, dtype=bool)), nx * ny


@pytest.mark.parametrize("d,r,dims", [(3, 5, (40, 40, 25)), (3, 4, (41, 41, 24)), (3, 6, (41, 41, 24)), (2, 3, (250, 200)),
                                      (2, 2, (201, 203))])
def test_symmetric_storage_product_matches_plain_and_oracle(oracle, d, r, dims):
    """DPGO_SPMM_SYMMETRIC (upper blocks only, transposed, outer-product gather): same Q*V (+G) as the plain block-CSR
    kernel and the oracle's CSR product, also after Q's values change; refused (plain arrays read) for a Q whose lower
    blocks are not the transposes of the upper ones."""
    import torch
    import dpgo_amd
    if d == 3:  # (41 x 41 x 24 = 40 344 poses: ragged last workgroup tile)
        meas, n, _ = oracle.synthetic_grid(*dims, seed=3)
    else:
        meas, n = _grid2d_measurements(oracle, *dims, seed=4)
    pg = dpgo_amd.PoseGraph(0, r, d)
    pg.setMeasurements(to_product_measurements(meas))
    assert pg.n() == n and n >= 40000
    rp, ci, v = pg.quadraticMatrix()
    prob = dpgo_amd.QuadraticProblem(pg)
    prob.setStream(torch.cuda.current_stream().cuda_stream)
    assert prob.setSpmmVariant("auto") == "plain"  # below the size switch: the Infinity Cache holds the working set
    g = torch.Generator(device="cuda").manual_seed(7)
    X = torch.randn((n, d + 1, r), device="cuda", dtype=torch.float64, generator=g)
    plain, sym = torch.empty_like(X), torch.empty_like(X)
    prob.spmmDevice(X, plain)
    assert prob.setSpmmVariant("symmetric") == "symmetric"
    prob.spmmDevice(X, sym)
    torch.cuda.synchronize()
    Qs = oracle.BSR(n, d + 1, rp, ci, v).to_scipy().tocsr()
    ref = (Qs @ X.cpu().numpy().reshape(n * (d + 1), r)).reshape(n, d + 1, r)
    assert relerr(sym.cpu().numpy(), ref) < 1e-13
    assert relerr(sym.cpu().numpy(), plain.cpu().numpy()) < 1e-13
    # new values on the same pattern: the transposed copy follows
    v2 = np.asarray(v) * 1.5
    dpgo_amd.lib.check(prob._lib.dpgo_problem_update_Q_values(prob.handle, dpgo_amd.lib.ptr(np.ascontiguousarray(v2))))
    assert prob.setSpmmVariant("symmetric") == "symmetric"
    prob.spmmDevice(X, sym)
    torch.cuda.synchronize()
    assert relerr(sym.cpu().numpy(), 1.5 * ref) < 1e-13
    # a Q that is not symmetric in its values: the plain arrays are read, the product is still that matrix's
    v3 = np.array(v, copy=True).reshape(len(ci), d + 1, d + 1)
    row_of = np.repeat(np.arange(n), np.diff(rp))
    low = np.nonzero(np.asarray(ci) < row_of)[0][0]
    v3[low, 0, 1] += 1.0
    dpgo_amd.lib.check(prob._lib.dpgo_problem_update_Q_values(prob.handle, dpgo_amd.lib.ptr(np.ascontiguousarray(v3))))
    assert prob.setSpmmVariant("symmetric") == "plain"
    prob.spmmDevice(X, sym)
    torch.cuda.synchronize()
    Q3 = oracle.BSR(n, d + 1, rp, ci, v3).to_scipy().tocsr()
    ref3 = (Q3 @ X.cpu().numpy().reshape(n * (d + 1), r)).reshape(n, d + 1, r)
    assert relerr(sym.cpu().numpy(), ref3) < 1e-13


def test_symmetric_storage_needs_big_blocks(oracle):
    import dpgo_amd
    om, n, d, Q, pg, prob = build_single_agent(oracle, "sphere2500", 5)
    assert prob.setSpmmVariant("symmetric") == "plain"  # 4 lane groups per pose below 40 000 poses


@pytest.mark.parametrize("name,robots,sweeps,precond", [("smallGrid3D", 5, 4, "jacobi"), ("torus3D", 8, 10, "jacobi"),
                                                        ("smallGrid3D", 5, 4, "multilevel"),
                                                        ("torus3D", 8, 10, "multilevel"),
                                                        ("grid:50x50x40", 8, 2, "multilevel"),
                                                        ("torus3D", 8, 10, "additive"),
                                                        ("grid:50x50x40", 8, 2, "additive")])
def test_multi_agent_rbcd_on_one_gpu_matches_oracle(oracle, name, robots, sweeps, precond):
    """BASELINE configs[0] / configs[2] / configs[3] shape: N agents (one PGOAgent each in the reference's
    MultiRobotExample), here N DeviceAgents on one GPU exchanging public poses by device copies;
    coloured RBCD sweeps vs the oracle driver at matched settings (same preconditioner, same recurrence):
    smallGrid3D / 5, torus3D / 8 for ten sweeps, and the 100k-pose grid cut into 8 slabs of 12 500 poses.
    "additive": the coupled blocks (G from the neighbours) solved by the one-launch kernel with the additive two-level
    preconditioner -- torus3D's 625-pose blocks on 16-pose aggregates, the 12 500-pose slabs on merged graph aggregates of
    at most 64 poses (one workgroup each; the oracle builds every agent's hierarchy from that agent's additivePlan)."""
    import dpgo_amd
    from dpgo_amd.agent import DeviceAgent, ExchangePlan, RBCDCluster, build_pose_graphs
    r = 5
    if name.startswith("grid:"):
        om, n, Ttrue = oracle.synthetic_grid(*[int(v) for v in name[5:].split("x")], seed=0)
        X0 = oracle.lift(oracle.perturbed_truth(Ttrue, seed=2), r)
    else:
        om, n = oracle.read_g2o(os.path.join(DATA, name + ".g2o"))
        X0 = oracle.lift(oracle.chordal_initialization(om, n), r)
    d = om.d
    ranges, graphs = build_pose_graphs(to_product_measurements(om), n, robots, r)
    plan = ExchangePlan(graphs)
    agents = {a: DeviceAgent(graphs, plan, a, X0[ranges[a][0]:ranges[a][1]], dpgo_amd.ROptParameters(precond=precond))
              for a in range(robots)}
    amg_k = None
    if precond == "additive":
        plans = {a: agents[a].problem.additivePlan() for a in range(robots)}
        assert all(pl["lane_groups"] == (4 if name == "torus3D" else 1) and pl["graph"] for pl in plans.values()), plans
        amg_k = {a: plans[a]["ks"] for a in range(robots)}
    Xref, costs, gns = oracle.rbcd_coloured(om, n, robots, r, X0, sweeps, hess_recurrence=device_tcg_mode(n // robots, d, r),
                                            precond={"multilevel": "amg", "additive": "amg_additive"}.get(precond, precond),
                                            amg_k=amg_k)
    cluster = RBCDCluster(plan, agents)
    central = oracle.QuadraticProblem(oracle.construct_Q(n, d, om), None, r, d)
    f0, g0 = cluster.central_cost_and_gradnorm()
    assert abs(2 * f0 - 2 * central.f(X0)) <= 1e-10 * abs(2 * central.f(X0))
    assert abs(g0 - central.rie_grad_norm(X0)) <= 1e-9 * g0
    for k in range(sweeps):
        cluster.sweep()
        f, g = cluster.central_cost_and_gradnorm()
        assert abs(2 * f - costs[k]) <= 1e-9 * abs(costs[k])
        assert abs(g - gns[k]) <= 1e-6 * gns[k]
    X = np.concatenate([agents[a].X.cpu().numpy() for a in range(robots)], axis=0)
    assert relerr(X, Xref) < 1e-7
    assert costs[-1] < costs[0]
    if precond == "additive":  # every block's last solve ran inside the one-launch kernel on its plan's workgroups
        for a in range(robots):
            res, info = agents[a].optimizer.getOptResult(), agents[a].problem.persistentInfo()
            assert res.precond_used == "additive" and info["last_members"] == plans[a]["aggregates"], (a, res, info)
            assert agents[a].problem.multilevelInfo()["ks"] == plans[a]["ks"]


@pytest.mark.parametrize("name,robots,settle,sweeps", [("torus3D", 8, 0, 12), ("grid:50x50x40", 8, 4, 11)])
def test_auto_cost_rule_switches_coupled_blocks_to_additive(oracle, name, robots, settle, sweeps):
    """precond = "auto" on COUPLED blocks the additive one-launch solve can hold (include/dpgo_hip.h, DPGO_PRECOND_AUTO):
    every agent starts on block-Jacobi; once its block-Jacobi solves since Q last changed have cost as much as one
    hierarchy set-up (280 products' worth) the next solve runs the additive form on trial, and stays there while it is
    cheaper than the block-Jacobi solve it is measured against.  The run goes THROUGH the switch -- torus3D / 8 from the
    chordal guess, the 100k grid as 8 slabs of 12 500 poses from an iterate a few sweeps in (far from the optimum the
    slabs' solves end on the trust-region boundary after a handful of products and the rule rightly never fires) -- and
    is compared sweep by sweep with the oracle, which is TOLD which solve ran which preconditioner (precond_used) and
    builds every agent's hierarchy from that agent's additivePlan; the oracle's own restatement of the rule
    (AutoCostRule), fed with the oracle's product counts, must predict the device's choices."""
    import dpgo_amd
    from dpgo_amd.agent import DeviceAgent, ExchangePlan, RBCDCluster, build_pose_graphs
    r = 5
    if name.startswith("grid:"):
        om, n, Ttrue = oracle.synthetic_grid(*[int(v) for v in name[5:].split("x")], seed=0)
        X0 = oracle.lift(oracle.perturbed_truth(Ttrue, seed=2), r)
    else:
        om, n = oracle.read_g2o(os.path.join(DATA, name + ".g2o"))
        X0 = oracle.lift(oracle.chordal_initialization(om, n), r)
    d = om.d
    ranges, graphs = build_pose_graphs(to_product_measurements(om), n, robots, r)
    plan = ExchangePlan(graphs)
    agents = {a: DeviceAgent(graphs, plan, a, X0[ranges[a][0]:ranges[a][1]], dpgo_amd.ROptParameters())
              for a in range(robots)}
    assert all(ag.optimizer.params_.precond == "auto" for ag in agents.values())
    cluster = RBCDCluster(plan, agents)
    for _ in range(settle):  # (untimed part of the run: whatever it selects; the comparison starts from a fresh rule)
        cluster.sweep()
    for ag in agents.values():
        ag.problem.autoState("reset")
        assert ag.problem.autoState() is False and ag.problem.autoInfo()["state"] == "jacobi"
    Xstart = np.concatenate([agents[a].X.cpu().numpy() for a in range(robots)], axis=0)
    used, products, dev = {}, {}, []
    for k in range(sweeps):
        cluster.sweep()
        for a in range(robots):
            res = agents[a].last_result
            used[(a, k)], products[(a, k)] = res.precond_used, res.tcg_iterations
        f, g = cluster.central_cost_and_gradnorm()
        dev.append((2 * f, g))
    assert set(used.values()) <= {"jacobi", "additive"}, used
    switched = [a for a in range(robots) if any(used[(a, k)] == "additive" for k in range(sweeps))]
    assert switched, products  # the run reaches the switch
    plans = {a: agents[a].problem.additivePlan() for a in range(robots)}
    counts = []
    names = {"jacobi": "jacobi", "additive": "amg_additive"}
    Xref, costs, gns = oracle.rbcd_coloured(om, n, robots, r, Xstart, sweeps, hess_recurrence=device_tcg_mode(n // robots, d, r),
                                            precond="jacobi", amg_k={a: plans[a]["ks"] for a in range(robots)},
                                            schedule={key: names[v] for key, v in used.items()}, counts=counts)
    for k in range(sweeps):
        assert abs(dev[k][0] - costs[k]) <= 1e-9 * abs(costs[k]), k
        assert abs(dev[k][1] - gns[k]) <= 1e-6 * gns[k], k
    assert {(a, k): it for k, a, it in counts} == products
    X = np.concatenate([agents[a].X.cpu().numpy() for a in range(robots)], axis=0)
    assert relerr(X, Xref) < 1e-7
    for a in range(robots):  # the device followed the rule (this agent's unit costs: four same-colour agents share the device)
        info = agents[a].problem.autoInfo()
        rule = oracle.AutoCostRule(budget=150, units_jacobi=info["units_jacobi"], units_additive=info["units_additive"],
                                   setup_units=info["setup_units"], min_products=info["min_products"],
                                   units_jacobi_alone=info["units_jacobi_alone"])
        for k in range(sweeps):
            assert used[(a, k)] == rule.next(), (a, k, [products[(a, q)] for q in range(sweeps)])
            rule.record(products[(a, k)])
        got = agents[a].problem.autoInfo()
        assert (got["state"] != "jacobi") == (rule.next() == "additive") and got["switches"] == rule.switches, (a, got)
    for a in switched:  # an additive solve ran inside the one-launch kernel, on its plan's workgroups, hierarchy kept across G
        last = max(k for k in range(sweeps) if used[(a, k)] == "additive")
        assert agents[a].problem.multilevelInfo()["ks"] == plans[a]["ks"], a
        if last == sweeps - 1:
            assert agents[a].problem.persistentInfo()["last_members"] == plans[a]["aggregates"]


@pytest.mark.parametrize("storage", [pytest.param("plain", id="plain"),
                                     pytest.param("symmetric", id="symmetric-fp32_dense_level-oracle_applies_the_device_inverse")])
def test_whole_solve_at_full_size_matches_oracle(oracle, storage):
    """BASELINE.json config 4 at full size (100 000 poses, one agent): the first RBCD iterations of the bench's run --
    QuadraticOptimizer::optimize with the reference's default parameters from the perturbed-truth iterate, repeated --
    against the oracles at matched settings: block-Jacobi against the plain-C restatement, the default multilevel
    preconditioner (hierarchy [64], dense coarsest operator of 6 252 unknowns built on the device) against the NumPy
    one.  Every call starts from the oracle's current iterate (far from the optimum the trust-region boundary decides
    the steps and round-off differences between two implementations grow from call to call); per call: same RTR / tCG
    iteration counts, cost to 1e-9 (+ 1e-4 of the call's decrease), iterate to 1e-6 (1e-4 in calls that move far).
    storage = "symmetric": the same with every Q product of the tCG loop (k_tcg_hess_sym, level-0 restriction and
    post-smoothing) on the symmetric storage that blocks beyond the Infinity Cache's size select by themselves -- and,
    riding along, the opt-in fp32 storage of the dense level, for which the ORACLE IS HANDED THE DEVICE'S STORED INVERSE
    (checked to 1e-7 against its own first: two fp64 inverses that agree to 1e-12 round to neighbouring fp32 values in a
    few entries); that half of the comparison is therefore about the cycle, not about the inverse.
    The tolerances are round-off sensitivity, not slack of the device path: the plain-C restatement run twice with two
    summation orders drifts by the same amounts (tests/test_oracle.py::test_two_summation_orders_...)."""
    import torch
    import dpgo_amd
    import c_oracle as CO
    meas, n, Ttrue = oracle.synthetic_grid(50, 50, 40, seed=0)
    d, r = 3, 5
    X0 = oracle.lift(oracle.perturbed_truth(Ttrue, seed=2), r)
    Q = oracle.construct_Q(n, d, meas)
    pg = dpgo_amd.PoseGraph(0, r, d)
    pg.setMeasurements(to_product_measurements(meas))
    prob = dpgo_amd.QuadraticProblem(pg)
    assert prob.setSpmmVariant(storage) == storage
    for precond, calls in (("jacobi", 6), ("multilevel", 6)):
        opt = dpgo_amd.QuadraticOptimizer(prob, dpgo_amd.ROptParameters(precond=precond))
        Xd = torch.tensor(X0, device="cuda", dtype=torch.float64)
        if precond == "multilevel":
            bits = 32 if storage == "symmetric" else 64  # the opt-in fp32 storage of the dense level rides along
            info = prob.setupMultilevel(coarse_bits=bits)
            # (the default: two levels, graph aggregates grown to 182 poses with the fragments merged up to 273: 546 of them,
            # a dense level of about 2 200 unknowns)
            assert info["ks"] == [-182, -273] and info["sizes"][0] == 100000 and 400 <= info["sizes"][1] <= 900
            # (on the symmetric storage the cycle streams fp32 copies of its level-0 operators by default: mirrored)
            obits = prob.multilevelOperatorBits()["bits"] if storage == "symmetric" else 64
            assert obits == (32 if storage == "symmetric" else 64)
            op = oracle.QuadraticProblem(Q, None, r, d, precond="amg", amg_k=info["ks"], amg_coarse_bits=bits,
                                         amg_operator_bits=obits)
            if bits == 32:  # both sides run with the SAME stored inverse (see _hierarchy_check)
                inv = prob.multilevelGet(1, "inverse")
                assert relerr(inv, op.amg_setup()["AcInv"]) < 1e-7
                op.amg_setup()["AcInv"] = inv
        Xo = X0.copy()
        total = 0
        for it in range(calls):
            Xd.copy_(torch.tensor(Xo))
            res = opt.optimizeDevice(Xd)
            if precond == "jacobi":
                Xo, ro = CO.optimize(Q, None, Xo, hess_recurrence=True)
                want = (ro.tcg_iterations, ro.rtr_iterations, ro.fOpt)
            else:
                oo = oracle.QuadraticOptimizer(op, oracle.ROptParameters(), hess_recurrence=True)
                Xo = oo.optimize(Xo)
                want = (oo.result.tcg_iters, oo.result.outer_iters, oo.result.fOpt)
            assert (res.tcg_iterations, res.rtr_iterations) == want[:2], (precond, it)
            # 1e-9 of the cost, plus 1e-4 of the DECREASE the call achieved: far from the optimum a call is dozens of
            # CG steps on an ill-conditioned operator, whose round-off sensitivity two summation orders do not share
            dec = abs(res.fInit - res.fOpt)
            assert abs(res.fOpt - want[2]) <= 1e-9 * abs(want[2]) + 1e-4 * dec, (precond, it)
            # 150 CG steps on 400 000 unknowns leave 1.5e-7 between two summation orders in the iterate
            assert relerr(Xd.cpu().numpy(), Xo) < (1e-6 if dec < 1e-4 * abs(want[2]) else 1e-4), (precond, it)
            total += res.tcg_iterations
        if precond == "multilevel":
            ob = prob.multilevelOperatorBits()
            assert ob["active"] == ob["vectors"] == (storage == "symmetric") and ob["dense"] == (bits == 32)
        assert total > 40  # the calls reach the regime in which the tCG budget is actually used


def test_mixed_precision_default_reaches_the_reference_cost(oracle):
    """The bridge from the DEFAULT configuration of HBM-bound blocks -- the multilevel cycle streaming fp32 copies of its
    level-0 operators and keeping its two internal vectors in fp32 -- to the reference's exact fp64 operator, on a block
    where that storage is active: the synthetic 40 x 40 x 25 grid (40 000 poses, BASELINE configs[3]'s generator), single
    agent, symmetric storage, RTR to |rgrad| < 1e-4 from the perturbed-truth iterate.  Reference side: the oracle with the
    EXACT (Q + 0.1 I)^-1 preconditioner (src/QuadraticProblem.cpp:56-69) -- its sparse factor of the 160 000-unknown 3-D
    operator takes minutes, so the run is a committed fixture (tests/golden/make_golden_grid40k.py ->
    golden_scalars.json["grid40x40x25_exact"]).  Asserted: final cost within 1e-6 relative on all three pairs (default /
    fp64 cycle / reference), the two device runs within 2 products and one outer iteration of each other."""
    import json
    import torch
    import dpgo_amd
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_scalars.json")))
    ref = gold["grid40x40x25_exact"]
    # (the reference run used all of its 60 outer iterations and stopped at |rgrad| = 1.6e-4: 1e-13 of the cost away from
    # the optimum -- the comparison below is at 1e-6)
    assert ref["precond"] == "exact" and ref["gradNormOpt"] < 2e-4 and ref["n"] == 40000
    meas, n, Ttrue = oracle.synthetic_grid(40, 40, 25, seed=ref["seed_graph"])
    d, r = 3, ref["r"]
    X0 = oracle.lift(oracle.perturbed_truth(Ttrue, seed=ref["seed_iterate"]), r)
    pg = dpgo_amd.PoseGraph(0, r, d)
    pg.setMeasurements(to_product_measurements(meas))
    prob = dpgo_amd.QuadraticProblem(pg)
    assert abs(prob.f(tiles_to_matrix(X0)) - ref["fInit"]) <= 1e-12 * abs(ref["fInit"])  # the same problem, the same start
    assert prob.setSpmmVariant("symmetric") == "symmetric"
    prm = dpgo_amd.ROptParameters(precond="multilevel", gradnorm_tol=1e-4, RTR_iterations=100, RTR_tCG_iterations=500,
                                  time_bound_s=120.0)
    out = {}
    for bits in (32, 64):
        prob.multilevelOperatorBits(bits)
        opt = dpgo_amd.QuadraticOptimizer(prob, prm)
        Xd = torch.tensor(X0, device="cuda", dtype=torch.float64)
        res = opt.optimizeDevice(Xd)
        ob = prob.multilevelOperatorBits()
        assert ob["active"] == ob["vectors"] == (bits == 32), ob  # what the solve's cycles really streamed
        assert res.gradNormOpt < 2e-4
        out[bits] = res
    f32, f64, fo = out[32].fOpt, out[64].fOpt, ref["fOpt"]
    assert abs(f32 - fo) <= 1e-6 * abs(fo), (f32, fo)
    assert abs(f64 - fo) <= 1e-6 * abs(fo), (f64, fo)
    assert abs(f32 - f64) <= 1e-6 * abs(fo), (f32, f64)
    assert abs(out[32].tcg_iterations - out[64].tcg_iterations) <= 2, (out[32].tcg_iterations, out[64].tcg_iterations)
    assert abs(out[32].rtr_iterations - out[64].rtr_iterations) <= 1
    prob.multilevelOperatorBits(32)


def test_symmetric_storage_solve_2d_matches_oracle(oracle):
    """k_tcg_hess_sym for SE(2) (three lanes per pose: shuffle reduction instead of the quad butterfly): 50 000-pose
    lattice, r = 4, block-Jacobi, two calls against the plain-C restatement; and the same calls on the plain storage."""
    import torch
    import dpgo_amd
    import c_oracle as CO
    meas, n = _grid2d_measurements(oracle, 250, 200, seed=4)
    d, r = 2, 4
    Q = oracle.construct_Q(n, d, meas)
    pg = dpgo_amd.PoseGraph(0, r, d)
    pg.setMeasurements(to_product_measurements(meas))
    prob = dpgo_amd.QuadraticProblem(pg)
    X0 = random_point(oracle, n, d, r, 11)
    want = []
    Xo = X0.copy()
    for _ in range(2):
        Xo, ro = CO.optimize(Q, None, Xo, hess_recurrence=True)
        want.append((Xo.copy(), ro))
    for storage in ("symmetric", "plain"):
        assert prob.setSpmmVariant(storage) == storage
        opt = dpgo_amd.QuadraticOptimizer(prob, dpgo_amd.ROptParameters(precond="jacobi"))
        Xd = torch.tensor(X0, device="cuda", dtype=torch.float64)
        for it in range(2):
            if it:
                Xd.copy_(torch.tensor(want[it - 1][0]))
            res = opt.optimizeDevice(Xd)
            Xw, ro = want[it]
            assert (res.tcg_iterations, res.rtr_iterations) == (ro.tcg_iterations, ro.rtr_iterations), (storage, it)
            dec = abs(res.fInit - res.fOpt)
            assert abs(res.fOpt - ro.fOpt) <= 1e-9 * abs(ro.fOpt) + 1e-4 * dec, (storage, it)
            assert relerr(Xd.cpu().numpy(), Xw) < 1e-5, (storage, it)
    # ... and the multilevel cycle on the symmetric storage in 2-D (three lanes per pose, 12-double tiles): its fp32 storage
    # (operator copies and internal vectors, the default there) against the fp64 originals -- the same solve to round-off
    # of the preconditioner: same counts, cost to 1e-8, iterate to 1e-6
    assert prob.setSpmmVariant("symmetric") == "symmetric"
    outs = {}
    for bits in (32, 64):
        prob.multilevelOperatorBits(bits)
        opt = dpgo_amd.QuadraticOptimizer(prob, dpgo_amd.ROptParameters(precond="multilevel"))
        Xd = torch.tensor(X0, device="cuda", dtype=torch.float64)
        res = opt.optimizeDevice(Xd)
        ob = prob.multilevelOperatorBits()
        assert res.precond_used == "multilevel" and ob["active"] == ob["vectors"] == (bits == 32), (bits, ob)
        outs[bits] = (res, Xd.cpu().numpy())
    (r32, X32), (r64, X64) = outs[32], outs[64]
    assert (r32.tcg_iterations, r32.rtr_iterations, r32.tCGStatus) == (r64.tcg_iterations, r64.rtr_iterations, r64.tCGStatus)
    assert r32.fOpt < r32.fInit and abs(r32.fOpt - r64.fOpt) <= 1e-8 * abs(r64.fOpt) and relerr(X32, X64) < 1e-6


@pytest.mark.parametrize("name,r,precond,layout", [
    ("smallGrid3D", 5, "jacobi", None), ("sphere2500", 5, "jacobi", None), ("sphere2500", 5, "none", None),
    ("sphere2500", 3, "jacobi", None), ("tinyGrid3D", 5, "jacobi", None), ("smallGrid3D", 6, "none", None),
    ("torus3D", 5, "jacobi", None), ("kitti_00", 5, "jacobi", None), ("kitti_00", 4, "none", None),
    ("kitti_00", 3, "jacobi", None), ("kitti_00", 2, "jacobi", None),
    ("sphere2500", 5, "jacobi", (1, 1)), ("sphere2500", 5, "jacobi", (4, 2)), ("torus3D", 5, "jacobi", (1, 2)),
    ("kitti_00", 5, "jacobi", (1, 2)), ("kitti_00", 3, "jacobi", (1, 2)), ("kitti_00", 3, "none", (1, 1)),
    ("sphere2500", 3, "jacobi", (1, 2)), ("sphere2500", 6, "jacobi", (1, 2)), ("sphere2500", 6, "jacobi", (4, 2))])
def test_persistent_tcg_matches_oracle(oracle, name, r, precond, layout):
    """The persistent whole-chip tCG kernel (one launch per tCG run, the all-reduces of an iteration's dot products are
    its barriers; kernels/persist.h) against the oracle at matched settings, exactly as the two-kernel scheme is tested:
    same RTR / tCG iteration counts and status, iterate to 1e-7, cost to 1e-9 -- and it must really have run
    (participants > 0), in the default layout of the block's size and in forced layouts (lane groups per pose, tiles per
    workgroup; a subprocess, because the knobs are read once), in 3-D and 2-D."""
    if layout is not None:
        env = dict(os.environ, DPGO_PERSIST_SPLIT=str(layout[0]), DPGO_PERSIST_MT=str(layout[1]), DPGO_PERSIST="1")
        code = ("import sys; sys.path.insert(0, %r); import conftest, dpgo_oracle, test_parity_gpu as t; "
                "t._persistent_case(dpgo_oracle, %r, %d, %r, %r)" % (os.path.dirname(os.path.abspath(__file__)), name, r,
                                                                    precond, layout))
        p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-3000:]
        return
    _persistent_case(oracle, name, r, precond, None)


def _persistent_case(oracle, name, r, precond, layout):
    import dpgo_amd
    om, n, d, Q, pg, prob = build_single_agent(oracle, name, r)
    prob.setPersistent(True)
    X0 = oracle.lift(oracle.chordal_initialization(om, n), r)
    op = oracle.QuadraticProblem(Q, None, r, d, precond=precond)
    oo = oracle.QuadraticOptimizer(op, oracle.ROptParameters(), hess_recurrence=True)
    go = dpgo_amd.QuadraticOptimizer(prob, dpgo_amd.ROptParameters(precond=precond))
    Xo, Xg = X0, X0
    for call in range(2):
        Xo = oo.optimize(Xo)
        Xg = matrix_to_tiles(go.optimize(tiles_to_matrix(Xg)), d)
        rg = go.getOptResult()
        info = prob.persistentInfo()
        assert info["enabled"] == 1 and info["last_members"] >= 1 and info["last_members"] == info["workgroups"], info
        if layout is not None:
            assert (info["last_split"], info["last_tiles"]) == tuple(layout), info
        assert (rg.tcg_iterations, rg.rtr_iterations, rg.tCGStatus) == (oo.result.tcg_iters, oo.result.outer_iters,
                                                                         oracle.TCG_NAMES[oo.result.tCGStatus])
        assert relerr(Xg, Xo) < 1e-7
        Xa = np.abs(Xo).reshape(n * (d + 1), r)
        scale = float((Xa * (abs(op.Qs) @ Xa)).sum())
        assert abs(rg.fOpt - oo.result.fOpt) <= 1e-9 * abs(oo.result.fOpt) + 1e-14 * scale


@pytest.mark.parametrize("name,r", [("smallGrid3D", 5), ("sphere2500", 5), ("kitti_00", 5), ("tinyGrid3D", 5),
                                    ("sphere2500", 3), ("kitti_00", 3), ("smallGrid3D", 6), ("torus3D", 5),
                                    ("grid:25x25x10", 5), ("grid:50x50x5", 5), ("grid:25x25x10", 3),
                                    ("grid2d:100x80", 3), ("grid2d:100x80", 4)])
def test_additive_preconditioner_matches_oracle(oracle, name, r):
    """precond = "additive": z = proj_X(Dinv r + P A_c^-1 P^T r) on a two-level hierarchy with ONE aggregate per workgroup
    of the persistent kernel, which owns the aggregate's poses wherever their indices are -- graph aggregates of at most 16
    (3-D) / 20 (2-D) poses while 256 of them cover the block, beyond that (torus3D's 5 000 poses, the 6 250-pose grid
    block and the 12 500-pose slab of the 16- / 8-agent cuts of BASELINE configs[3]) aggregates of at most 64 poses whose
    fragments were merged (additivePlan) --, a whole preconditioned tCG iteration inside the persistent kernel (three
    in-kernel reductions; the restricted residual is the only extra exchange).  Against the oracle's restatement of the
    operator (precond = "amg_additive", same aggregates) at matched settings: same RTR / tCG iteration counts and status,
    iterate to 1e-7, cost to 1e-9, over three calls; the kernel must really have run; the hierarchy is the oracle's."""
    import dpgo_amd
    if name.startswith("grid"):
        if name.startswith("grid2d:"):  # SE(2) lattice, 8 000 poses: the 84-pose-tile layout of the 2-D instances
            om, n = _grid2d_measurements(oracle, *[int(v) for v in name[7:].split("x")], seed=4)
            X0 = oracle.lift(oracle.chordal_initialization(om, n), r)
        else:
            om, n, Ttrue = oracle.synthetic_grid(*[int(v) for v in name[5:].split("x")], seed=0)
            X0 = oracle.lift(oracle.perturbed_truth(Ttrue, seed=2), r)
        d = om.d
        Q = oracle.construct_Q(n, d, om)
        pg = dpgo_amd.PoseGraph(0, r, d)
        pg.setMeasurements(to_product_measurements(om))
        prob = dpgo_amd.QuadraticProblem(pg)
    else:
        om, n, d, Q, pg, prob = build_single_agent(oracle, name, r)
        X0 = oracle.lift(oracle.chordal_initialization(om, n), r)
    plan = prob.additivePlan()
    k = 16 if d == 3 else 20
    # sphere2500 160 aggregates, kitti_00 246: plain growth, four lane groups per pose; torus3D's 5 000 poses and the grid
    # blocks: one pose per (d+1) lanes, merged fragments
    assert plan["lane_groups"] == (1 if name == "torus3D" or name.startswith("grid") else 4), plan
    if plan["lane_groups"] == 4:
        assert (plan["ks"], plan["tile"]) == ([-k], k), plan
    else:
        assert plan["tile"] == (64 // (d + 1)) * 4 and plan["graph"] and len(plan["ks"]) == 2, plan  # 64 poses; 84 in 2-D
        assert -plan["ks"][1] == min(plan["tile"], plan["growth"] + plan["growth"] // 2)
    ks = plan["ks"]
    op = oracle.QuadraticProblem(Q, None, r, d, precond="amg_additive", amg_k=ks)
    na = op.amg_setup()["nc"]
    assert na == plan["aggregates"] <= 256
    if len(ks) == 2:  # the smallest growth size that fits: the one before leaves more than 256 aggregates
        S = plan["growth"]
        prev = [s_ for s_ in _additive_growth_sizes(n, plan["tile"]) if s_ < S]
        if prev:
            lab, ptr, mem, _, _ = oracle.amg_graph_aggregates(Q, prev[-1])
            cap = min(plan["tile"], prev[-1] + prev[-1] // 2)
            assert len(oracle.amg_merge_small_aggregates(Q, prev[-1], lab, ptr, mem, cap)[1]) - 1 > 256
    oo = oracle.QuadraticOptimizer(op, oracle.ROptParameters(), hess_recurrence=True)
    go = dpgo_amd.QuadraticOptimizer(prob, dpgo_amd.ROptParameters(precond="additive"))
    Xo, Xg = X0, X0
    # (the random-measurement SE(2) lattice is far from any optimum: every call ends on the trust-region boundary after
    # 4 ... 20 products through regions of negative curvature, so six calls, each from the ORACLE's previous iterate)
    resync = name.startswith("grid2d")
    for call in range(6 if resync else 3):
        if resync:
            Xg = Xo
        Xo = oo.optimize(Xo)
        Xg = matrix_to_tiles(go.optimize(tiles_to_matrix(Xg)), d)
        rg = go.getOptResult()
        info = prob.persistentInfo()
        assert rg.precond_used == "additive"
        if rg.gradNormInit >= 1e-2:  # (an iterate that already meets the tolerance leaves before any tCG launch)
            assert info["last_members"] == na and info["last_split"] == plan["lane_groups"], (rg, info)
        assert (rg.tcg_iterations, rg.rtr_iterations, rg.tCGStatus) == (oo.result.tcg_iters, oo.result.outer_iters,
                                                                         oracle.TCG_NAMES[oo.result.tCGStatus]), call
        assert relerr(Xg, Xo) < 1e-7
        Xa = np.abs(Xo).reshape(n * (d + 1), r)
        scale = float((Xa * (abs(op.Qs) @ Xa)).sum())
        assert abs(rg.fOpt - oo.result.fOpt) <= 1e-9 * abs(oo.result.fOpt) + 1e-14 * scale
    assert prob.multilevelInfo()["ks"] == ks
    _hierarchy_check(oracle, prob, op)


def _additive_growth_sizes(n, tile):
    """Growth sizes additive_plan (csrc/multilevel.hip) tries for a block beyond 256 aggregates of 16 poses, in order."""
    out, S = [], max(8, (n + 229) // 230)
    while S <= tile:
        out.append(S)
        S += max(2, S // 8)
    return out


def test_additive_preconditioner_selection_and_fallbacks(oracle):
    """auto on a small uncoupled block resolves its multilevel choice to the additive form (sphere2500: precond_used
    "additive"; torus3D's 5 000 poses: the 64-pose-tile layout); a block beyond 256 aggregates of one workgroup tile
    refuses "additive" explicitly and auto keeps the V-cycle there (an 18 000-pose grid); with the persistent kernel
    switched off "additive" runs the V-cycle on the same hierarchy and still converges to the same optimum."""
    import dpgo_amd
    from dpgo_amd.lib import DpgoError
    om, n, d, Q, pg, prob = build_single_agent(oracle, "sphere2500", 5)
    X0 = oracle.lift(oracle.chordal_initialization(om, n), 5)
    go = dpgo_amd.QuadraticOptimizer(prob, dpgo_amd.ROptParameters())
    X = X0
    for _ in range(8):
        X = matrix_to_tiles(go.optimize(tiles_to_matrix(X)), d)
        assert go.getOptResult().precond_used == "additive"
        if go.getOptResult().gradNormOpt < 1e-2:
            break
    f_add = go.getOptResult().fOpt
    assert go.getOptResult().gradNormOpt < 1e-2 and abs(2 * f_add - 1687.00581428) <= 1e-6 * 1687.0  # literature optimum
    prob.setPersistent(False)
    go2 = dpgo_amd.QuadraticOptimizer(prob, dpgo_amd.ROptParameters(precond="additive"))
    X = X0
    for _ in range(8):
        X = matrix_to_tiles(go2.optimize(tiles_to_matrix(X)), d)
        if go2.getOptResult().gradNormOpt < 1e-2:
            break
    assert prob.persistentInfo()["last_members"] == 0 and abs(go2.getOptResult().fOpt - f_add) <= 1e-7 * abs(f_add)
    mid = build_single_agent(oracle, "torus3D", 5)[-1]
    omt, nt = oracle.read_g2o(os.path.join(DATA, "torus3D.g2o"))
    Xt = tiles_to_matrix(oracle.lift(oracle.chordal_initialization(omt, nt), 5))
    gm = dpgo_amd.QuadraticOptimizer(mid, dpgo_amd.ROptParameters())
    gm.optimize(Xt)
    assert gm.getOptResult().precond_used == "additive" and mid.persistentInfo()["last_split"] == 1
    omb, nb, Tb = oracle.synthetic_grid(30, 30, 20, seed=0)
    pgb = dpgo_amd.PoseGraph(0, 5, 3)
    pgb.setMeasurements(to_product_measurements(omb))
    big = dpgo_amd.QuadraticProblem(pgb)
    assert big.additivePlan()["lane_groups"] == 0
    Xb = tiles_to_matrix(oracle.lift(oracle.perturbed_truth(Tb, seed=2), 5))
    with pytest.raises(DpgoError):
        dpgo_amd.QuadraticOptimizer(big, dpgo_amd.ROptParameters(precond="additive")).optimize(Xb)
    ga = dpgo_amd.QuadraticOptimizer(big, dpgo_amd.ROptParameters())
    ga.optimize(Xb)
    assert ga.getOptResult().precond_used == "multilevel"


def test_persistent_granule_table_survives_salt_wraparound(oracle):
    """The in-kernel all-reduce tags its granules with a per-launch salt (11 bits of the handle's generation counter) and the
    table is cleared only before a salt can repeat: 2 200 one-launch solves on one handle -- some through the multi-launch
    scheme in between, which advances the counter too -- all complete (no time-out, the kernel keeps running) and keep
    returning the converged iterate's statistics."""
    import dpgo_amd
    om, n, d, Q, pg, prob = build_single_agent(oracle, "smallGrid3D", 5)
    X = tiles_to_matrix(oracle.lift(oracle.chordal_initialization(om, n), 5))
    go = dpgo_amd.QuadraticOptimizer(prob, dpgo_amd.ROptParameters(precond="jacobi"))
    shrink = dpgo_amd.QuadraticOptimizer(prob, dpgo_amd.ROptParameters(precond="jacobi", RTR_iterations=1))  # multi-launch
    for _ in range(12):
        X = go.optimize(X)
    ref = go.getOptResult()
    assert ref.gradNormOpt < 1e-2 and prob.persistentInfo()["last_members"] > 0
    for k in range(2200):
        if k % 97 == 5:
            shrink.optimize(X)
            continue
        X2 = go.optimize(X)
        res = go.getOptResult()
        assert res.success and prob.persistentInfo()["last_members"] > 0, k
        assert abs(res.fOpt - ref.fOpt) <= 1e-12 * abs(ref.fOpt), k
    assert prob.persistentInfo()["enabled"] == 1  # (a time-out would have switched the kernel off for this handle)


def test_persistent_tcg_is_refused_beyond_its_capacity_and_follows_the_size_switch(oracle):
    """Blocks that need more than 2 tiles on each of 256 workgroups are refused (explicit request: error); the default
    is on by size: sphere2500 runs the persistent kernel without being asked, the single-iteration radius-shrink mode
    and an iterate that already meets the tolerance behave as on the two-kernel scheme."""
    import dpgo_amd
    from dpgo_amd.lib import DpgoError
    om, n, d, Q, pg, prob = build_single_agent(oracle, "sphere2500", 5)
    assert prob.persistentInfo()["enabled"] == 1
    X0 = oracle.lift(oracle.chordal_initialization(om, n), 5)
    two = build_single_agent(oracle, "sphere2500", 5)[-1]
    two.setPersistent(False)
    for prm in (dpgo_amd.ROptParameters(precond="jacobi", RTR_iterations=1, RTR_initial_radius=1e4),
                dpgo_amd.ROptParameters(precond="jacobi", gradnorm_tol=1e9),
                dpgo_amd.ROptParameters(precond="jacobi", RTR_tCG_iterations=0)):
        a = dpgo_amd.QuadraticOptimizer(prob, prm)
        b = dpgo_amd.QuadraticOptimizer(two, prm)
        Xa, Xb = a.optimize(tiles_to_matrix(X0)), b.optimize(tiles_to_matrix(X0))
        ra, rb = a.getOptResult(), b.getOptResult()
        assert (ra.tcg_iterations, ra.rtr_iterations, ra.tCGStatus, ra.latest_step_accepted) == (
            rb.tcg_iterations, rb.rtr_iterations, rb.tCGStatus, rb.latest_step_accepted)
        assert relerr(Xa, Xb) < 1e-9 and abs(ra.fOpt - rb.fOpt) <= 1e-10 * abs(rb.fOpt)
    assert two.persistentInfo()["last_members"] == 0
    from dpgo_amd import synthetic
    meas, nbig, _ = synthetic.synthetic_grid(50, 50, 16, seed=0)  # 40 000 poses: 625 tiles of 64 > 2 x 256
    pgb = dpgo_amd.PoseGraph(0, 5, 3)
    pgb.setMeasurements(meas)
    big = dpgo_amd.QuadraticProblem(pgb)
    assert big.persistentInfo()["enabled"] == 0
    with pytest.raises(DpgoError):
        big.setPersistent(True)


@pytest.mark.parametrize("name", ["tinyGrid3D", "smallGrid3D", "sphere2500", "torus3D", "kitti_00"])
def test_chordal_initialisation_matches_oracle(oracle, name):
    """dpgo_chordal_initialization (chordalInitialization, src/DPGO_solver.cpp:220-269, constructBMatrices /
    recoverTranslations, src/DPGO_utils.cpp:346-462): both least-squares problems solved on the device (Jacobi-PCG over
    the block-SpMM, SO(d) projection by the rounding kernel) against the oracle's direct sparse solves, on all five
    datasets; rotations are in SO(d); on noiseless measurements the relaxation returns the truth."""
    import dpgo_amd
    from dpgo_amd.initialization import chordal_initialization
    path = os.path.join(DATA, name + ".g2o")
    om, n = oracle.read_g2o(path)
    pm, _ = dpgo_amd.read_g2o_file(path)
    d = om.d
    (Tc, its), To = chordal_initialization(pm, n, return_iterations=True), oracle.chordal_initialization(om, n)
    assert Tc.shape == (n, d + 1, d) and min(its) >= 1
    assert np.abs(Tc[:, :d] - To[:, :d]).max() <= 1e-8  # rotations
    assert np.abs(Tc[:, d] - To[:, d]).max() <= 1e-7 * max(1.0, np.abs(To[:, d]).max())  # translations
    R = Tc[:, :d, :]
    assert np.abs(np.swapaxes(R, 1, 2) @ R - np.eye(d)).max() < 1e-10 and (np.linalg.det(R) > 0).all()
    Xc, Xo = oracle.lift(Tc, 5), oracle.lift(To, 5)
    P = oracle.QuadraticProblem(oracle.construct_Q(n, d, om), None, 5, d, precond="none")
    assert abs(P.f(Xc) - P.f(Xo)) <= 1e-7 * abs(P.f(Xo))  # same cost of the initial guess
    if n <= 200:  # noiseless measurements generated from Tc itself: the relaxation is exact
        Rg, tg = np.swapaxes(Tc[:, :d, :], 1, 2), Tc[:, d, :]
        exact = pm.select(np.arange(len(pm)))
        exact.R[:] = np.swapaxes(Rg[pm.p1], 1, 2) @ Rg[pm.p2]
        exact.t[:] = (np.swapaxes(Rg[pm.p1], 1, 2) @ (tg[pm.p2] - tg[pm.p1])[:, :, None])[:, :, 0]
        Te = chordal_initialization(exact, n)
        assert np.abs(np.swapaxes(Te[:, :d, :], 1, 2) - Rg[0].T @ Rg).max() < 1e-7
        assert np.abs(Te[:, d, :] - (tg - tg[0]) @ Rg[0]).max() < 1e-6 * max(1.0, np.abs(tg).max())


def test_rccl_transport_on_one_gpu(oracle):
    """The RCCL transport of the public-pose exchange on hardware (C ABI dpgo_comm_*): a 1-rank communicator owned by
    the solver library; every exchange of a 5-agent run travels as ONE grouped batch of self ncclSend / ncclRecv of
    packed pose tiles on the solver's stream (pack kernel -> RCCL -> coupling SpMM, no host wait), the reductions as
    RCCL all-reduces.  Same neighbour buffers, same sweeps (bit for bit), same central cost as with device copies
    (examples/MultiRobotExample.cpp:183-204,220-254 is what both replace)."""
    import torch
    import dpgo_amd
    from dpgo_amd.agent import DeviceAgent, ExchangePlan, RBCDCluster, build_pose_graphs
    from dpgo_amd.comm import DeviceComm, unique_id, SUM, MAX
    r, robots = 5, 5
    om, n = oracle.read_g2o(os.path.join(DATA, "smallGrid3D.g2o"))
    X0 = oracle.lift(oracle.chordal_initialization(om, n), r)
    comm = DeviceComm(1, 0, unique_id(), 0)
    t = torch.arange(6, dtype=torch.float64, device="cuda")
    comm.allreduce(t, SUM)
    comm.allreduce(t, MAX)
    comm.broadcast(t, 0)
    torch.cuda.synchronize()
    assert t.cpu().tolist() == [0.0, 1.0, 2.0, 3.0, 4.0, 5.0]
    runs = []
    for loop in (False, True):
        ranges, graphs = build_pose_graphs(to_product_measurements(om), n, robots, r)
        plan = ExchangePlan(graphs)
        agents = {a: DeviceAgent(graphs, plan, a, X0[ranges[a][0]:ranges[a][1]], dpgo_amd.ROptParameters())
                  for a in range(robots)}
        cluster = RBCDCluster(plan, agents, comm=comm if loop else None, loopback=loop)
        cluster.exchange(None)
        torch.cuda.synchronize()
        nbr0 = [agents[a].nbr.clone() for a in range(robots)]
        trace = [cluster.central_cost_and_gradnorm()]
        for _ in range(3):
            cluster.sweep()
            trace.append(cluster.central_cost_and_gradnorm())
        anchor = cluster.global_anchor()
        runs.append((nbr0, trace, [agents[a].X.clone() for a in range(robots)], anchor))
    for a in range(robots):
        assert torch.equal(runs[0][0][a], runs[1][0][a])  # what travelled through RCCL is what was packed
        assert torch.equal(runs[0][2][a], runs[1][2][a])  # identical sweeps
    assert runs[0][1] == runs[1][1] and np.array_equal(runs[0][3], runs[1][3])
    assert runs[1][1][-1][0] < runs[1][1][0][0]
    comm.close()


def test_concurrent_same_colour_agents_match_sequential_updates(oracle):
    """dpgo_optimize_device_many: the agents of a colour hosted by one GPU are solved CONCURRENTLY (own streams behind the
    exchange, one feeding thread each).  Every solve is a function of its own inputs only, so three sweeps of a 5-agent
    (smallGrid3D) and an 8-agent (torus3D) problem give the same iteration counts, iterates and central cost with and
    without the concurrency -- to round-off, not bit for bit: agents that share the device pick the most compact layout
    of the persistent tCG kernel, whose sums run in another order --, and so does the concurrent evaluation of the
    block terms."""
    import torch
    import dpgo_amd
    from dpgo_amd.agent import DeviceAgent, ExchangePlan, RBCDCluster, build_pose_graphs
    r = 5
    for name, robots in (("smallGrid3D", 5), ("torus3D", 8)):
        om, n = oracle.read_g2o(os.path.join(DATA, name + ".g2o"))
        X0 = oracle.lift(oracle.chordal_initialization(om, n), r)
        runs = []
        for concurrent in (False, True):
            ranges, graphs = build_pose_graphs(to_product_measurements(om), n, robots, r)
            plan = ExchangePlan(graphs)
            agents = {a: DeviceAgent(graphs, plan, a, X0[ranges[a][0]:ranges[a][1]], dpgo_amd.ROptParameters())
                      for a in range(robots)}
            cluster = RBCDCluster(plan, agents)
            cluster.concurrent = concurrent
            trace, counts = [cluster.central_cost_and_gradnorm()], []
            for _ in range(3):
                cluster.sweep()
                counts.append([(agents[a].last_result.tcg_iterations, agents[a].last_result.rtr_iterations,
                                agents[a].last_result.precond_used) for a in range(robots)])
                trace.append(cluster.central_cost_and_gradnorm())
            torch.cuda.synchronize()
            runs.append((trace, counts, [agents[a].X.clone() for a in range(robots)]))
        assert runs[0][1] == runs[1][1]
        for a in range(robots):
            assert relerr(runs[1][2][a].cpu().numpy(), runs[0][2][a].cpu().numpy()) < 1e-9
        for (f0, g0), (f1, g1) in zip(runs[0][0], runs[1][0]):
            assert abs(f0 - f1) <= 1e-10 * abs(f0) and abs(g0 - g1) <= 1e-7 * max(g0, 1e-3)
        assert runs[1][0][-1][0] < runs[1][0][0][0]


def test_stream_ordered_sweep_matches_phase_by_phase(oracle):
    """A process that hosts ONE agent per colour (two agents per GPU, the multi-GPU deployment) enqueues a whole sweep --
    exchange, solve, next colour's pack + exchange, solve -- on one stream through dpgo_optimize_device_begin / _end and
    reads the results back at the end (RBCDCluster.sweep): the iterates and results are those of the phase-by-phase sweep
    bit for bit (same kernels in the same order), for one-launch solves (2 x 6 250-pose blocks, block-Jacobi and the
    additive form via "auto") and for solves that cannot be enqueued (multilevel V-cycle: begin runs them to completion).
    The two halves refuse to be called out of order."""
    import dpgo_amd
    from dpgo_amd.agent import DeviceAgent, ExchangePlan, RBCDCluster, build_pose_graphs
    from dpgo_amd.lib import DpgoError
    r, robots = 5, 2
    om, n, Ttrue = oracle.synthetic_grid(25, 25, 20, seed=0)
    X0 = oracle.lift(oracle.perturbed_truth(Ttrue, seed=2), r)
    for precond in ("jacobi", "auto", "multilevel"):
        out = {}
        for ordered in (True, False):
            ranges, graphs = build_pose_graphs(to_product_measurements(om), n, robots, r)
            plan = ExchangePlan(graphs)
            agents = {a: DeviceAgent(graphs, plan, a, X0[ranges[a][0]:ranges[a][1]], dpgo_amd.ROptParameters(precond=precond))
                      for a in range(robots)}
            cluster = RBCDCluster(plan, agents)
            assert cluster._stream_ordered_sweep() is True
            if not ordered:
                cluster._so_sweep = False
            hist = []
            for _ in range(4):
                cluster.sweep()
                hist.append([(ag.last_result.tcg_iterations, ag.last_result.rtr_iterations, ag.last_result.precond_used,
                              ag.last_result.fOpt, ag.last_result.gradNormOpt) for ag in agents.values()])
            out[ordered] = (hist, [agents[a].X.cpu().numpy() for a in range(robots)], cluster.central_cost_and_gradnorm())
            if ordered and precond == "jacobi":
                ag = agents[0]
                with pytest.raises(DpgoError):
                    ag.optimizer.optimizeDeviceEnd()  # nothing in flight
                ag.update_begin()
                with pytest.raises(DpgoError):
                    ag.update_begin()  # one solve per handle
                ag.update_end()
                assert ag.problem.persistentInfo()["last_members"] > 0  # it WAS a one-launch solve
        assert out[True][0] == out[False][0], precond
        for Xa, Xb in zip(out[True][1], out[False][1]):
            assert np.array_equal(Xa, Xb), precond
        assert out[True][2] == out[False][2]
        if precond == "auto":  # coupled blocks: block-Jacobi first, the additive form once the budget binds
            used = [h[2] for sweep in out[True][0] for h in sweep]
            assert "jacobi" in used and set(used) <= {"jacobi", "additive"}


@pytest.mark.parametrize("workload,sweeps", [("smallGrid3D", 6), ("grid:20x20x10", 3)])
def test_two_processes_exchange_through_mapped_buffers_on_one_gpu(workload, sweeps):
    """A genuinely multi-PROCESS device exchange on one GPU (RCCL refuses two ranks on one device; IPC does not): two
    processes on device 0, two agents each (one per colour), gloo for the rendezvous and the small reductions, the
    receivers' neighbour tile buffers mapped into the senders through hipIpc handles and the senders' batched pack
    kernel (k_gather_tiles_batched) writing straight into them (dpgo_amd/ipc.py; SURVEY section 5's peer-store
    alternative).  The iterates after the sweeps are those of the same four agents in ONE process, bit for bit."""
    port = 29500 + (os.getpid() % 400) + (17 if workload.startswith("grid") else 0)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(os.path.dirname(os.path.abspath(__file__)), "ipc_worker.py"), workload,
           str(sweeps)]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("IPC_RESULT")]
    assert len(line) == 1, p.stdout[-2000:]
    kv = dict(tok.split("=", 1) for tok in line[0].split()[1:])
    assert kv["bit_identical"] == "1" and kv["costs_equal"] == "1" and kv["iterations_equal"] == "1", line[0]
    assert kv["decrease"] == "1" and float(kv["exchange_ms_per_sweep"]) > 0.0


def test_inactive_robot_leaves_the_team(oracle):
    """PGOAgent::setRobotActive(id, false) (src/PGOAgent.cpp:1173-1184) on the device path: robot 3 of smallGrid3D / 5 is
    switched off after two sweeps -- it stops updating, its neighbours (2 and 4) rebuild Q and the coupling blocks
    without the shared edges (values only, on the device), the sweeps go on -- and switched on again; every stage against
    the oracle's coloured RBCD with the same activity sets (same iterates to 1e-7, same cost)."""
    import dpgo_amd
    from dpgo_amd.agent import DeviceAgent, ExchangePlan, RBCDCluster, build_pose_graphs
    r, robots = 5, 5
    om, n = oracle.read_g2o(os.path.join(DATA, "smallGrid3D.g2o"))
    X0 = oracle.lift(oracle.chordal_initialization(om, n), r)
    ranges, graphs = build_pose_graphs(to_product_measurements(om), n, robots, r)
    plan = ExchangePlan(graphs)
    agents = {a: DeviceAgent(graphs, plan, a, X0[ranges[a][0]:ranges[a][1]], dpgo_amd.ROptParameters(precond="jacobi"))
              for a in range(robots)}
    cluster = RBCDCluster(plan, agents)
    mode = device_tcg_mode(n // robots, om.d, r)
    Xo = X0
    for inactive, sweeps in (((), 2), ((3,), 3), ((), 2)):
        cluster.set_robot_active(3, 3 not in inactive)
        assert agents[2].pg.isNeighborActive(3) == (3 not in inactive) and agents[0].isRobotActive(3) == (3 not in inactive)
        Xo, costs, gns = oracle.rbcd_coloured(om, n, robots, r, Xo, sweeps, hess_recurrence=mode, precond="jacobi",
                                              inactive=inactive)
        before = agents[3].X.clone()
        for _ in range(sweeps):
            cluster.sweep()
        X = np.concatenate([agents[a].X.cpu().numpy() for a in range(robots)], axis=0)
        assert relerr(X, Xo) < 1e-7, inactive
        if inactive:
            assert bool((agents[3].X == before).all())  # the inactive robot did not move
        else:  # (the central cost is assembled from the agents' LOCAL problems: with a robot off they leave edges out)
            f, g = cluster.central_cost_and_gradnorm()
            assert abs(2 * f - costs[-1]) <= 1e-9 * abs(costs[-1])


def test_inactive_neighbour_in_robust_mode_keeps_the_device_weights(oracle):
    """PGOAgent::setRobotActive in robust (GNC) mode -- ADVICE r4: the GNC weights live on the device, the host copy of
    the measurements is what Q is rebuilt from when a neighbour is switched off.  smallGrid3D / 5 with every loop closure
    registered as re-weightable: after a re-weighting that leaves weights strictly between 0 and 1, robot 3 is switched
    off and on again.  At every stage agent 2's device weights and Q values are those of the reference's data matrices
    (constructQ over the ACTIVE edges with the current weights, src/PoseGraph.cpp:381-491): the edges with robot 3 leave
    with weight 0, are not re-weighted while it is off (PGOAgent::updateMeasurementWeights walks activeLoopClosures()
    only, src/PGOAgent.cpp:1104-1118), and come back with the weights they left with; all other weights survive both
    switches."""
    import dpgo_amd
    from dpgo_amd.agent import DeviceAgent, ExchangePlan, RBCDCluster, build_pose_graphs
    r, robots, me, off = 5, 5, 2, 3
    om, n = oracle.read_g2o(os.path.join(DATA, "smallGrid3D.g2o"))
    d = om.d
    X0 = oracle.lift(oracle.chordal_initialization(om, n), r)
    ranges, graphs = build_pose_graphs(to_product_measurements(om), n, robots, r)
    plan = ExchangePlan(graphs)
    agents = {a: DeviceAgent(graphs, plan, a, X0[ranges[a][0]:ranges[a][1]], dpgo_amd.ROptParameters(precond="jacobi"))
              for a in range(robots)}
    cluster = RBCDCluster(plan, agents)
    for ag in agents.values():
        ag.problem.setReweightableEdges(include_shared=True)
    cluster.sweep()
    cluster.exchange(None)
    ag = agents[me]
    pg, prob = ag.pg, ag.problem
    idx = prob.reweightable_index
    m = pg.measurements()
    with_off = np.array([(m.r1[e] != m.r2[e]) and off in (int(m.r1[e]), int(m.r2[e])) for e in idx])
    assert with_off.any()

    def q_reference():  # Q of agent `me` from its host measurements as the reference's constructQ builds it
        pg._Q = None    # (the cached matrix predates the weights written back)
        _, _, vals = pg.quadraticMatrix()
        return np.asarray(vals).copy()

    def q_device():
        out = np.zeros_like(q_reference())
        dpgo_amd.lib.check(prob._lib.dpgo_problem_get_Q_values(prob.handle, dpgo_amd.lib.ptr(out)))
        return out

    # a re-weighting with a small mu: weights strictly inside (0, 1) exist afterwards
    mu = 0.05  # (GNC-TLS: residuals between mu / (mu + 1) barc^2 and (mu + 1) / mu barc^2 get intermediate weights)
    prob.gncReweightDevice(ag.X, ag.nbr, mu, 5.0, update=True)
    w1, _ = prob.getEdgeWeights()
    lc = ~pg.odometry_mask()[idx]
    assert np.all(w1[~lc] == 1.0) and (np.unique(np.round(w1[lc], 12)).size > 1 or np.any((w1[lc] > 0) & (w1[lc] < 1)))
    # --- robot 3 off: its edges leave (weight 0), everything else keeps the DEVICE weights
    cluster.set_robot_active(off, False)
    w2, _ = prob.getEdgeWeights()
    assert np.all(w2[with_off] == 0.0) and np.array_equal(w2[~with_off], w1[~with_off])
    assert np.array_equal(np.asarray(m.weight)[idx][~with_off], w1[~with_off])  # the host copy followed
    assert relerr(q_device(), q_reference()) < 1e-13
    # ... a re-weighting while it is off leaves those edges out
    prob.gncReweightDevice(ag.X, ag.nbr, 1.4 * mu, 5.0, update=True)
    w3, _ = prob.getEdgeWeights()
    assert np.all(w3[with_off] == 0.0)
    prob.pullEdgeWeights()
    assert relerr(q_device(), q_reference()) < 1e-13
    lam = np.linalg.eigvalsh(oracle.BSR(pg.n(), d + 1, *[np.asarray(v) for v in pg.quadraticMatrix()]).to_scipy().toarray())
    assert lam.min() > -1e-9 * lam.max()  # Q stays positive semidefinite
    # --- back on: the edges return with the weights they left with, the others with the latest ones
    cluster.set_robot_active(off, True)
    w4, _ = prob.getEdgeWeights()
    assert np.array_equal(w4[with_off], w1[with_off]) and np.array_equal(w4[~with_off], w3[~with_off])
    assert relerr(q_device(), q_reference()) < 1e-13
    cluster.sweep()  # the solves still run on the rebuilt matrices
    assert agents[me].last_result.success


def test_more_concurrent_handles_than_hardware_queues_warn_once_and_complete():
    """ROCm maps HIP streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default; bench.py raises it to 16 before the runtime
    starts).  A caller that solves 8 handles concurrently WITHOUT the variable is not refused and not silently serialised:
    the solves complete with the results of one-at-a-time solves and the library says so once on stderr
    (dpgo_warning_count)."""
    code = r"""
import os, sys
sys.path.insert(0, %r)
sys.path.insert(0, os.path.join(%r, "oracle"))
import numpy as np, torch
import dpgo_amd, dpgo_amd.lib as L
import dpgo_oracle as O
from dpgo_amd.solver import optimize_device_many
from conftest import to_product_measurements
om, n = O.read_g2o(os.path.join(%r, "data", "smallGrid3D.g2o"))
X0 = O.lift(O.chordal_initialization(om, n), 5)
probs, opts, Xs = [], [], []
for k in range(8):
    pg = dpgo_amd.PoseGraph(0, 5, 3)
    pg.setMeasurements(to_product_measurements(om))
    probs.append(dpgo_amd.QuadraticProblem(pg))
    opts.append(dpgo_amd.QuadraticOptimizer(probs[-1], dpgo_amd.ROptParameters(precond="jacobi")))
    Xs.append(torch.tensor(X0, device="cuda", dtype=torch.float64))
solo = opts[0].optimizeDevice(Xs[0].clone())
assert L.load().dpgo_warning_count() == 0
for rep in range(2):
    res = optimize_device_many(opts, Xs, [None] * 8, torch.cuda.current_stream().cuda_stream)
    if rep == 0:
        assert all(r.tcg_iterations == solo.tcg_iterations and abs(r.fOpt - solo.fOpt) <= 1e-12 * abs(solo.fOpt) for r in res)
assert L.load().dpgo_warning_count() == 1
print("QUEUES_OK")
""" % (ROOT_DIR, ROOT_DIR, ROOT_DIR)
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    env["PYTHONPATH"] = os.path.dirname(os.path.abspath(__file__)) + os.pathsep + env.get("PYTHONPATH", "")
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "QUEUES_OK" in p.stdout, (p.stdout[-500:], p.stderr[-2000:])
    assert p.stderr.count("GPU_MAX_HW_QUEUES is 4") == 1, p.stderr[-2000:]


def test_concurrent_update_with_different_parameters_solves_each_with_its_own(oracle):
    """dpgo_optimize_device_many takes ONE parameter record; the Python face solves optimizers that are configured
    differently one after the other, each with its own parameters, instead of silently applying the first one's."""
    import torch
    import dpgo_amd
    from dpgo_amd.solver import optimize_device_many
    om, n, d, Q, pg, prob = build_single_agent(oracle, "smallGrid3D", 5)
    _, _, _, _, pg2, prob2 = build_single_agent(oracle, "smallGrid3D", 5)
    X0 = oracle.lift(oracle.chordal_initialization(om, n), 5)
    oa = dpgo_amd.QuadraticOptimizer(prob, dpgo_amd.ROptParameters(precond="jacobi"))
    ob = dpgo_amd.QuadraticOptimizer(prob2, dpgo_amd.ROptParameters(precond="none", RTR_tCG_iterations=7))
    want = [o.optimizeDevice(torch.tensor(X0, device="cuda", dtype=torch.float64)) for o in (oa, ob)]
    assert want[0].tcg_iterations != want[1].tcg_iterations
    Xs = [torch.tensor(X0, device="cuda", dtype=torch.float64) for _ in range(2)]
    got = optimize_device_many([oa, ob], Xs, None, torch.cuda.current_stream().cuda_stream)
    for g, w in zip(got, want):
        assert (g.tcg_iterations, g.rtr_iterations, g.precond_used) == (w.tcg_iterations, w.rtr_iterations, w.precond_used)
        assert g.fOpt == w.fOpt


def test_external_stream_ordering_is_deterministic(oracle):
    """Regression: work of a handle bound to torch's current stream (the NULL / default stream) is ordered
    with torch ops on that stream -- restoring an iterate with tensor.copy_ and solving again gives the
    bit-identical result every time."""
    import torch
    import dpgo_amd
    from dpgo_amd.agent import DeviceAgent, ExchangePlan, build_pose_graphs
    om, n, Ttrue = oracle.synthetic_grid(20, 20, 10, seed=5)
    X0 = oracle.lift(oracle.perturbed_truth(Ttrue, seed=6), 5)
    ranges, graphs = build_pose_graphs(to_product_measurements(om), n, 1, 5)
    ag = DeviceAgent(graphs, ExchangePlan(graphs), 0, X0, dpgo_amd.ROptParameters(precond="jacobi"))
    for _ in range(2):
        ag.update()
    ag.snapshot()
    outs = []
    for _ in range(6):
        ag.restore()
        res = ag.update()
        outs.append((res.fInit, res.fOpt, res.tcg_iterations, ag.X.clone()))
    for o in outs[1:]:
        assert o[:3] == outs[0][:3]
        assert torch.equal(o[3], outs[0][3])


def test_greedy_accelerated_schedule_matches_oracle(oracle):
    """BASELINE configs[0]: smallGrid3D, 5 agents, r = 5 with the reference demo's schedule (greedy selection,
    Nesterov acceleration, restart every 30 iterations; examples/MultiRobotExample.cpp:170-255,
    src/PGOAgent.cpp:376-432).  Device agents (Y, V on the GPU, updateY / updateV through the polar-projection
    kernel) vs the oracle driver at matched settings: same selection sequence, same stop iteration, same cost.
    The oracle in the reference configuration (exact preconditioner) stops at iteration index 86 with
    2f = 1025.39835."""
    import dpgo_amd
    from dpgo_amd.agent import DeviceAgent, ExchangePlan, RBCDCluster, build_pose_graphs
    r, robots = 5, 5
    om, n = oracle.read_g2o(os.path.join(DATA, "smallGrid3D.g2o"))
    d = om.d
    X0 = oracle.lift(oracle.chordal_initialization(om, n), r)
    ref = oracle.multi_robot_example(om, n, robots, r, X0, precond="exact")
    assert ref["iterations"] == 87 and abs(ref["cost"] - 1025.39835) < 5e-6 and ref["gradnorm"] < 0.1
    want = oracle.multi_robot_example(om, n, robots, r, X0, precond="jacobi", hess_recurrence=device_tcg_mode(n // robots, om.d, r))
    ranges, graphs = build_pose_graphs(to_product_measurements(om), n, robots, r)
    plan = ExchangePlan(graphs)
    agents = {a: DeviceAgent(graphs, plan, a, X0[ranges[a][0]:ranges[a][1]], dpgo_amd.ROptParameters(precond="jacobi"))
              for a in range(robots)}
    for ag in agents.values():
        ag.enable_acceleration(robots)
    got = RBCDCluster(plan, agents).run_greedy()
    assert got["selected"] == want["selected"]
    assert got["iterations"] == want["iterations"]
    assert abs(got["cost"] - want["cost"]) <= 1e-9 * want["cost"]
    assert abs(got["gradnorm"] - want["gradnorm"]) <= 1e-5 * want["gradnorm"]
    assert sum(a.tcg_total for a in agents.values()) == want["tcg_total"]
    X = np.concatenate([agents[a].X.cpu().numpy() for a in range(robots)], axis=0)
    assert relerr(X, want["X"]) < 1e-7
    assert abs(got["cost"] - ref["cost"]) <= 1e-6 * ref["cost"]  # same optimum as the reference configuration
    # Two more runs on the SAME agents and ONE cluster: enable_acceleration() re-allocates Y and the auxiliary neighbour /
    # send buffers and the iterate is re-bound -- the cluster's cached exchange plans hold raw device addresses of the old
    # tensors and must be rebuilt (in between the allocator is pushed to hand the freed blocks to other data).
    import torch
    cluster = RBCDCluster(plan, agents)
    runs, gens, junk = [], [], []
    for rep in range(2):
        for a, ag in agents.items():
            ag.X = torch.tensor(np.ascontiguousarray(X0[ranges[a][0]:ranges[a][1]]), dtype=torch.float64, device="cuda")
            ag.iteration, ag.tcg_total = 0, 0
            ag.enable_acceleration(robots)
        junk.append([torch.full((3000,), float("nan"), dtype=torch.float64, device="cuda") for _ in range(64)])
        runs.append(cluster.run_greedy(max_iters=12))
        assert cluster._xplans, "the batched exchange ran"
        gens.append(cluster._xplans_gen)
    assert gens[0] != gens[1]
    for run in runs:
        assert run["selected"] == want["selected"][:len(run["selected"])] and len(run["selected"]) >= 12
    assert abs(runs[0]["cost"] - runs[1]["cost"]) <= 1e-12 * abs(runs[0]["cost"])


def test_agent_level_pose_renumbering_is_invisible_at_the_boundary(oracle):
    """The agent layer can renumber the poses INSIDE a block for locality (build_pose_graphs(reorder=True): runs of
    consecutive poses ordered by reverse Cuthill-McKee inside each XCD's eighth of the block, dpgo_locality_order; off by
    default -- measured, it does not pay on the lattice workloads) -- a renaming of the poses, so
    everything that crosses the agent's boundary must be what the un-renumbered agent produces: X0 in, iterates and
    rounded trajectories out in the caller's frame order, public poses exchanged between renumbered agents, the global
    anchor (the caller's pose 0).  One agent (40 x 40 x 25 lattice) and two agents (40 x 40 x 50) against the same agents
    with the renumbering switched off: same iteration counts, cost to 1e-10, iterate to 1e-7 (dot products sum in another
    order), trajectories to 1e-7; the cost of the returned iterate on the ORIGINAL graph is the cost the solver reports."""
    import dpgo_amd
    from dpgo_amd.agent import DeviceAgent, ExchangePlan, RBCDCluster, build_pose_graphs
    r = 5
    for dims, robots in (((40, 40, 25), 1), ((40, 40, 50), 2)):
        om, n, Ttrue = oracle.synthetic_grid(*dims, seed=0)
        X0 = oracle.lift(oracle.perturbed_truth(Ttrue, seed=2), r)
        runs = {}
        for reorder in (False, True):
            ranges, graphs = build_pose_graphs(to_product_measurements(om), n, robots, r, reorder=reorder)
            assert all((g.pose_order is not None) == bool(reorder) for g in graphs)
            if reorder:
                for g in graphs:
                    assert sorted(g.pose_order.tolist()) == list(range(g.n())) and not np.array_equal(g.pose_order, np.arange(g.n()))
            plan = ExchangePlan(graphs)
            agents = {a: DeviceAgent(graphs, plan, a, X0[ranges[a][0]:ranges[a][1]], dpgo_amd.ROptParameters(precond="jacobi"))
                      for a in range(robots)}
            cluster = RBCDCluster(plan, agents)
            f0, g0 = cluster.central_cost_and_gradnorm()
            for _ in range(2):
                cluster.sweep()
            f, g = cluster.central_cost_and_gradnorm()
            X = np.concatenate([agents[a].iterate_in_caller_order().cpu().numpy() for a in range(robots)], axis=0)
            loc = agents[0].getTrajectoryInLocalFrame().cpu().numpy()
            glob = np.concatenate([t.cpu().numpy() for _, t in sorted(cluster.trajectories_in_global_frame().items())], axis=0)
            runs[bool(reorder)] = dict(f0=f0, g0=g0, f=f, g=g, X=X, loc=loc, glob=glob,
                                       its=[(agents[a].last_result.tcg_iterations, agents[a].last_result.rtr_iterations)
                                            for a in range(robots)])
        a_, b_ = runs[False], runs[True]
        assert abs(a_["f0"] - b_["f0"]) <= 1e-12 * abs(a_["f0"]) and abs(a_["g0"] - b_["g0"]) <= 1e-10 * a_["g0"]
        assert a_["its"] == b_["its"], (a_["its"], b_["its"])
        assert abs(a_["f"] - b_["f"]) <= 1e-10 * abs(a_["f"]) and abs(a_["g"] - b_["g"]) <= 1e-6 * a_["g"]
        assert relerr(b_["X"], a_["X"]) < 1e-7 and relerr(b_["loc"], a_["loc"]) < 1e-7 and relerr(b_["glob"], a_["glob"]) < 1e-7
        central = oracle.QuadraticProblem(oracle.construct_Q(n, om.d, om), None, r, om.d)
        assert abs(central.f(b_["X"]) - b_["f"]) <= 1e-10 * abs(b_["f"]) and b_["f"] < b_["f0"]


def test_robust_pgo_known_answer_on_device(oracle):
    """tests/testPGO.cpp:193-271 (testRobustPGO) through the device path: solveRobustPGO with GNC-TLS
    (barc = 7, tol 1e-1, 50 RTR iterations, odometry start) classifies the inlier (w = 1) and the outlier
    (w = 0) within 1e-6; residual / weight kernel (K10) and the on-device rebuild of Q's values (K9)."""
    import dpgo_amd
    from dpgo_amd.robust import RobustCostParameters, solveRobustPGO, solveRobustPGOParams
    from test_oracle import robust_chain_problem
    om, n, T0 = robust_chain_problem(oracle)
    pm = to_product_measurements(om)
    prm = solveRobustPGOParams(opt_params=dpgo_amd.ROptParameters(precond="jacobi", gradnorm_tol=1e-1, RTR_iterations=50),
                               robust_params=RobustCostParameters("GNC_TLS", GNCBarc=7.0))
    T, info = solveRobustPGO(pm, n, prm, T0=T0)
    assert abs(pm.weight[3] - 1) <= 1e-6 and abs(pm.weight[4]) <= 1e-6
    assert np.all(pm.weight[:3] == 1.0)
    To, info_o = oracle.solve_robust_pgo(om, n, T0, oracle.ROptParameters(gradnorm_tol=1e-1, RTR_iterations=50),
                                         barc=7.0, precond="jacobi", hess_recurrence=device_tcg_mode(n, 3, 3))
    assert info["gnc_iterations"] == info_o["gnc_iterations"]
    assert abs(info["muInit"] - info_o["muInit"]) <= 1e-6 * abs(info_o["muInit"])
    assert relerr(T, To) < 1e-6


def test_gnc_reweighting_with_outliers_matches_oracle(oracle):
    """BASELINE configs[4] flavour (single agent): kitti_00 (2-D, EDGE_SE2) plus 25 synthetic outlier loop
    closures; GNC-TLS (barc = 5, mu step 1.4, reference defaults).  Device vs oracle at matched settings:
    the same edges are rejected, weights agree, Q rebuilt on the device equals Q rebuilt from the weights on
    the host, and every injected outlier ends with weight 0."""
    import dpgo_amd
    from dpgo_amd.robust import RobustCostParameters, solveRobustPGO, solveRobustPGOParams
    om, n = oracle.read_g2o(os.path.join(DATA, "kitti_00.g2o"))
    rng = np.random.default_rng(11)
    k = 25
    i = rng.integers(0, n - 600, k)
    j = i + rng.integers(300, 600, k)
    th = rng.uniform(-np.pi, np.pi, k)
    Rk = np.stack([np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]]) for a in th])
    lc = np.nonzero(~om.fixed)[0]
    kap, tau = np.median(om.kappa[lc]), np.median(om.tau[lc])
    z = np.zeros(k, dtype=np.int64)
    out = oracle.Measurements(2, z, i, z.copy(), j, Rk, rng.uniform(-5, 5, (k, 2)), np.full(k, kap), np.full(k, tau),
                              np.ones(k), np.zeros(k, dtype=bool))
    allm = oracle.Measurements.concat([om, out])
    T0 = oracle.chordal_initialization(om, n)  # initial guess from the clean graph (same on both sides)
    opt_o = oracle.ROptParameters(RTR_iterations=10, RTR_tCG_iterations=100)
    To, info_o = oracle.solve_robust_pgo(allm, n, T0, opt_o, barc=5.0, precond="jacobi", hess_recurrence=device_tcg_mode(n, 2, 2),
                                         max_iters=20)
    pm = to_product_measurements(oracle.Measurements.concat([om, out]))
    pm.weight[:] = 1.0
    prm = solveRobustPGOParams(opt_params=dpgo_amd.ROptParameters(precond="jacobi", RTR_iterations=10, RTR_tCG_iterations=100,
                                                                  time_bound_s=60.0),
                               robust_params=RobustCostParameters("GNC_TLS"))
    T, info = solveRobustPGO(pm, n, prm, T0=T0)
    assert info["gnc_iterations"] == info_o["gnc_iterations"]
    assert np.array_equal(pm.weight < 1e-8, allm.weight < 1e-8)
    assert np.abs(pm.weight - allm.weight).max() < 1e-5
    assert np.all(pm.weight[-k:] < 1e-8)  # every injected outlier is rejected
    assert np.all(pm.weight[om.fixed.nonzero()[0]] == 1.0)
    assert abs(info["fOpt"] - info_o["fOpt"]) <= 1e-6 * abs(info_o["fOpt"])
    # values-only rebuild (K9) == full host construction with the final weights
    pg = dpgo_amd.PoseGraph(0, 2, 2)
    pg.setMeasurements(pm)
    Qh = oracle.construct_Q(n, 2, allm)
    rp, ci, v = pg.quadraticMatrix()
    assert np.array_equal(ci, Qh.colidx) and np.abs(v - Qh.vals).max() <= 1e-9 * np.abs(Qh.vals).max()


def _random_graph(oracle, d, n, n_lc, hub_edges, seed):
    """Chain odometry + random loop closures + one hub pose with `hub_edges` extra edges (a block row far
    longer than the preloaded index window of the gather core)."""
    rng = np.random.default_rng(seed)
    if d == 3:
        q = rng.standard_normal((n, 4))
        Rg = oracle._quat_batch(q / np.linalg.norm(q, axis=1, keepdims=True))
    else:
        th = rng.uniform(-np.pi, np.pi, n)
        Rg = np.stack([np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]]) for a in th])
    tg = np.cumsum(rng.standard_normal((n, d)), axis=0)
    hub = n // 3
    pairs = [(i, i + 1) for i in range(n - 1)]
    seen = set(pairs)
    while len(pairs) < n - 1 + n_lc:
        i, j = sorted(rng.integers(0, n, 2))
        if j > i + 1 and (i, j) not in seen:
            seen.add((i, j)); pairs.append((int(i), int(j)))
    others = rng.permutation(np.setdiff1d(np.arange(n), [hub - 1, hub, hub + 1]))[:hub_edges]
    for o in others:
        e = (hub, int(o)) if rng.random() < 0.5 else (int(o), hub)  # both orientations, incl. p2 < p1
        if e not in seen and (e[1], e[0]) not in seen:
            seen.add(e); pairs.append(e)
    p1 = np.array([p[0] for p in pairs]); p2 = np.array([p[1] for p in pairs])
    m = len(pairs)
    R = np.swapaxes(Rg[p1], 1, 2) @ Rg[p2]
    t = (np.swapaxes(Rg[p1], 1, 2) @ (tg[p2] - tg[p1])[:, :, None])[:, :, 0] + 0.05 * rng.standard_normal((m, d))
    z = np.zeros(m, dtype=np.int64)
    om = oracle.Measurements(d, z, p1, z.copy(), p2, R, t, rng.uniform(5, 50, m), rng.uniform(5, 50, m), np.ones(m),
                             p1 + 1 == p2)
    T = np.zeros((n, d + 1, d))
    T[:, :d, :] = np.swapaxes(Rg, 1, 2)
    T[:, d, :] = tg
    return om, T, hub


@pytest.mark.parametrize("d,r,n,hub_edges", [(3, 5, 300, 60), (2, 4, 300, 60), (2, 5, 257, 40), (3, 3, 64, 30),
                                             (3, 5, 41000, 70)])
def test_high_degree_rows_and_ragged_sizes(oracle, d, r, n, hub_edges):
    """Rows longer than the preloaded index window (tail loops of the gather core), both split layouts
    (n < 40000: 4 lane groups per pose; n >= 40000: 1), 2-D and 3-D, span and generic vector paths (odd / even
    tile size), pose counts that are not multiples of the wave / workgroup tile."""
    import dpgo_amd
    om, T, hub = _random_graph(oracle, d, n, n // 2, hub_edges, seed=100 + n + d)
    Q = oracle.construct_Q(n, d, om)
    assert np.diff(Q.rowptr).max() >= hub_edges
    pg = dpgo_amd.PoseGraph(0, r, d)
    pg.setMeasurements(to_product_measurements(om))
    prob = dpgo_amd.QuadraticProblem(pg)
    op = oracle.QuadraticProblem(Q, None, r, d, precond="jacobi")
    rng = np.random.default_rng(1)
    X = oracle.polar_project(oracle.lift(T, r) + 0.1 * rng.standard_normal((n, d + 1, r)), d)
    V = oracle.tangent_project(X, rng.standard_normal((n, d + 1, r)), d)
    Xm, Vm = tiles_to_matrix(X), tiles_to_matrix(V)
    assert relerr(matrix_to_tiles(prob.EucHessianEta(Xm, Vm), d), op.euc_hess(V)) < RTOL_ELEM
    assert relerr(matrix_to_tiles(prob.RieGrad(Xm), d), op.rie_grad(X)) < RTOL_ELEM
    S = op.sym_ytg(X, op.euc_grad(X))
    assert relerr(matrix_to_tiles(prob.RieHessianEta(Xm, Vm), d), op.rie_hess(X, S, V)) < RTOL_ELEM
    if n <= 1000:
        oo = oracle.QuadraticOptimizer(op, oracle.ROptParameters(), hess_recurrence=device_tcg_mode(n, d, r))
        Xo = oo.optimize(X)
        go = dpgo_amd.QuadraticOptimizer(prob, dpgo_amd.ROptParameters(precond="jacobi"))
        Xg = matrix_to_tiles(go.optimize(Xm), d)
        rg = go.getOptResult()
        if rg.tcg_iterations == oo.result.tcg_iters:
            # random, badly scaled problems (a hub with dozens of edges, r = d) amplify round-off through the
            # trust-region boundary steps far more than the benchmark datasets do (those agree to 1e-7)
            # (measured on the r = d = 3 case: agreement 6e-16 after two outer iterations / 18 tCG steps, 1.6e-5
            # after the third, ill-conditioned, tCG run with identical iteration counts and statuses)
            assert relerr(Xg, Xo) < 1e-4
            assert abs(rg.fOpt - oo.result.fOpt) <= 1e-3 * abs(oo.result.fOpt)
            two = dpgo_amd.ROptParameters(precond="jacobi", RTR_iterations=2)
            o2 = oracle.QuadraticOptimizer(op, oracle.ROptParameters(RTR_iterations=2), hess_recurrence=device_tcg_mode(n, d, r))
            X2o = o2.optimize(X)
            X2g = matrix_to_tiles(dpgo_amd.QuadraticOptimizer(prob, two).optimize(Xm), d)
            assert relerr(X2g, X2o) < 1e-10
        else:
            # the tCG stopping test |r| <= |r0| min(|r0|, 0.1) is a discontinuous decision: on a random problem
            # round-off can move it by one iteration; both runs must still be the same descent
            assert abs(rg.tcg_iterations - oo.result.tcg_iters) <= 1 and rg.rtr_iterations == oo.result.outer_iters
            assert abs(rg.fOpt - oo.result.fOpt) <= 0.05 * abs(oo.result.fOpt) and rg.fOpt < 0.01 * rg.fInit
    else:
        go = dpgo_amd.QuadraticOptimizer(prob, dpgo_amd.ROptParameters(precond="jacobi", RTR_iterations=1, RTR_tCG_iterations=10))
        oo = oracle.QuadraticOptimizer(op, oracle.ROptParameters(RTR_iterations=1, RTR_tCG_iterations=10),
                                       hess_recurrence=device_tcg_mode(n, d, r))
        Xo = oo.optimize(X)
        Xg = matrix_to_tiles(go.optimize(Xm), d)
        assert relerr(Xg, Xo) < 1e-8


def test_single_pose_and_two_pose_graphs(oracle):
    """Smallest inputs: n = 2 (one edge), and an agent whose block is a single pose with only shared edges."""
    import dpgo_amd
    z = np.zeros(1, dtype=np.int64)
    Rm = oracle._quat_batch(np.array([[0.9, 0.1, -0.3, 0.2]]) / np.linalg.norm([0.9, 0.1, -0.3, 0.2]))
    om = oracle.Measurements(3, z, np.array([0]), z.copy(), np.array([1]), Rm, np.array([[1.0, 2.0, 3.0]]),
                             np.array([10.0]), np.array([4.0]), np.ones(1), np.array([True]))
    pg = dpgo_amd.PoseGraph(0, 5, 3)
    pg.setMeasurements(to_product_measurements(om))
    prob = dpgo_amd.QuadraticProblem(pg)
    X0 = oracle.lift(np.stack([np.vstack([np.eye(3), np.zeros((1, 3))])] * 2), 5)
    opt = dpgo_amd.QuadraticOptimizer(prob, dpgo_amd.ROptParameters(precond="jacobi", gradnorm_tol=1e-9, RTR_iterations=30))
    Xg = matrix_to_tiles(opt.optimize(tiles_to_matrix(X0)), 3)
    assert opt.getOptResult().fOpt < 1e-12  # one edge can be satisfied exactly
    # ... with every preconditioner (n = 2: one aggregate, a 4 x 4 dense level; the additive form's plan is one workgroup)
    assert prob.additivePlan()["aggregates"] == 1 and prob.additivePlan()["lane_groups"] == 4
    for pc in ("none", "multilevel", "additive", "auto"):
        o2 = dpgo_amd.QuadraticOptimizer(prob, dpgo_amd.ROptParameters(precond=pc, gradnorm_tol=1e-9, RTR_iterations=30))
        o2.optimize(tiles_to_matrix(X0))
        assert o2.getOptResult().fOpt < 1e-12 and o2.getOptResult().gradNormOpt < 1e-6, pc
    # single-pose agent: robot 1 owns one pose, linked to robot 0 by one shared edge
    shared = oracle.Measurements(3, z, np.array([1]), np.ones(1, dtype=np.int64), np.array([0]), Rm,
                                 np.array([[1.0, 2.0, 3.0]]), np.array([10.0]), np.array([4.0]), np.ones(1),
                                 np.array([False]))
    pg1 = dpgo_amd.PoseGraph(1, 5, 3)
    pg1.setMeasurements(to_product_measurements(shared))
    assert pg1.n() == 1 and len(pg1.quadraticMatrix()[1]) == 1
    nbr = {(0, 1): Xg[1]}
    pg1.setNeighborPoses({k: v.T for k, v in nbr.items()})
    prob1 = dpgo_amd.QuadraticProblem(pg1)
    Qa = oracle.construct_Q(1, 3, oracle.Measurements.empty(3), shared, my_id=1)
    Ga = oracle.construct_G(1, 3, 5, shared, 1, nbr)
    pa = oracle.QuadraticProblem(Qa, Ga, 5, 3)
    Xa = X0[:1]
    assert abs(prob1.f(tiles_to_matrix(Xa)) - pa.f(Xa)) <= 1e-12 * abs(pa.f(Xa))
    assert relerr(matrix_to_tiles(prob1.RieGrad(tiles_to_matrix(Xa)), 3), pa.rie_grad(Xa)) < RTOL_ELEM
    # the one-pose block solved with every preconditioner: the same optimum as the oracle's (block-Jacobi is exact on it)
    oo = oracle.QuadraticOptimizer(oracle.QuadraticProblem(Qa, Ga, 5, 3, precond="jacobi"),
                                   oracle.ROptParameters(gradnorm_tol=1e-9, RTR_iterations=30))
    oo.optimize(Xa)
    for pc in ("jacobi", "none", "multilevel", "additive", "auto"):
        o1 = dpgo_amd.QuadraticOptimizer(prob1, dpgo_amd.ROptParameters(precond=pc, gradnorm_tol=1e-9, RTR_iterations=30))
        o1.optimize(tiles_to_matrix(Xa))
        assert abs(o1.getOptResult().fOpt - oo.result.fOpt) <= 1e-9 * max(abs(oo.result.fOpt), 1.0), pc
        assert o1.getOptResult().gradNormOpt < 1e-6, pc


def _inject_outliers(oracle, om, n, k, seed):
    """k random loop closures (random rotation, translation in [-5, 5]^d) with median precisions, weight 1."""
    rng = np.random.default_rng(seed)
    d = om.d
    # no duplicate (src, dst) pairs: PoseGraph::addMeasurement drops those (src/PoseGraph.cpp:83-88)
    taken = set(zip(om.p1.tolist(), om.p2.tolist()))
    p1, p2 = [], []
    while len(p1) < k:
        a = int(rng.integers(0, n))
        b = int((a + rng.integers(5, n - 5)) % n)
        if (a, b) not in taken:
            taken.add((a, b))
            p1.append(a)
            p2.append(b)
    p1, p2 = np.array(p1), np.array(p2)
    Rs = np.stack([np.linalg.qr(rng.standard_normal((d, d)))[0] for _ in range(k)])
    for q in range(k):
        if np.linalg.det(Rs[q]) < 0:
            Rs[q][:, 0] *= -1
    z = np.zeros(k, dtype=np.int64)
    out = oracle.Measurements(d, z, p1, z.copy(), p2, Rs, rng.uniform(-5, 5, size=(k, d)),
                              np.full(k, np.median(om.kappa)), np.full(k, np.median(om.tau)), np.ones(k),
                              np.zeros(k, dtype=bool))
    return oracle.Measurements.concat([om, out])


def test_agent_status_and_termination_vote_match_oracle(oracle):
    """SURVEY 8f row 1, "status": PGOAgentStatus on the device path.  relativeChange =
    LiftedPoseArray::maxTranslationDistance(X, XPrev) (src/manifold/Poses.cpp:86-94) from the device kernel against the
    oracle on random iterates (3-D and 2-D; exact: a maximum, no summation order), then the coloured schedule driven by
    PGOAgent::shouldTerminate (src/PGOAgent.cpp:846-878) on smallGrid3D / 5 agents and torus3D / 8 agents: same
    stopping iteration, same iteration numbers and ready flags, relative changes to 1e-7, iterate to 1e-7, as the
    oracle's driver."""
    import torch
    import dpgo_amd
    from dpgo_amd import lib as L
    from dpgo_amd.agent import (DeviceAgent, ExchangePlan, PGOAgentParameters, RBCDCluster, build_pose_graphs)
    lib = L.load()
    rng = np.random.default_rng(5)
    for d, r, n in ((3, 5, 1000), (2, 3, 77), (3, 3, 1), (2, 5, 4097)):
        A, B = rng.standard_normal((n, d + 1, r)), rng.standard_normal((n, d + 1, r))
        B[n // 2, d] = A[n // 2, d] + 40.0 / np.sqrt(r)  # the maximum sits on one known pose
        Ad, Bd = torch.tensor(A, device="cuda"), torch.tensor(B, device="cuda")
        out_d = torch.zeros(1, dtype=torch.float64, device="cuda")
        import ctypes as C
        out_h = C.c_double(-1.0)
        L.check(lib.dpgo_max_translation_distance_device(r, d, n, L.ptr(Ad), L.ptr(Bd), L.ptr(out_d), C.byref(out_h), None))
        ref = oracle.max_translation_distance(A, B)
        assert abs(ref - 40.0) < 1e-12 and abs(out_h.value - ref) <= 1e-15 * ref and float(out_d.item()) == out_h.value
    r = 5
    for name, robots, tol in (("smallGrid3D", 5, 5e-2), ("torus3D", 8, 1e-2)):
        om, n = oracle.read_g2o(os.path.join(DATA, name + ".g2o"))
        X0 = oracle.lift(oracle.chordal_initialization(om, n), r)
        Xref, info_o = oracle.rbcd_coloured_until_terminated(
            om, n, robots, r, X0, oracle.AgentParameters(relChangeTol=tol, maxNumIters=80),
            hess_recurrence=device_tcg_mode(n // robots, om.d, r))
        ranges, graphs = build_pose_graphs(to_product_measurements(om), n, robots, r)
        plan = ExchangePlan(graphs)
        agents = {a: DeviceAgent(graphs, plan, a, X0[ranges[a][0]:ranges[a][1]], dpgo_amd.ROptParameters(precond="jacobi"))
                  for a in range(robots)}
        out = RBCDCluster(plan, agents).run_until_terminated(PGOAgentParameters(relChangeTol=tol, maxNumIters=80))
        assert out["iterations"] == info_o["iterations"] < 80, (name, out["iterations"], info_o["iterations"])
        for a in range(robots):
            st, so = out["statuses"][a], info_o["statuses"][a]
            assert (st.iterationNumber, st.readyToTerminate, st.state) == (so.iterationNumber, so.readyToTerminate, so.state)
            assert abs(st.relativeChange - so.relativeChange) <= 1e-7 * max(so.relativeChange, 1e-3)
        X = np.concatenate([agents[a].X.cpu().numpy() for a in range(robots)], axis=0)
        assert relerr(X, Xref) < 1e-7


def test_distributed_gnc_with_the_reference_weight_update_trigger(oracle):
    """DistributedGNC with agent_params: the weight updates follow PGOAgent::shouldUpdateMeasurementWeights
    (src/PGOAgent.cpp:997-1045: every agent readyToTerminate -- relative change <= relChangeTol, 5 before the first
    update; converged-weight ratio >= robustOptMinConvergenceRatio -- or robustOptInnerIters global iterations), not
    a fixed sweep count.  smallGrid3D + 10 outliers, 3 agents: same number of global iterations in every block, same
    classification history, every outlier rejected, as the oracle's driver with the same trigger."""
    import dpgo_amd
    from dpgo_amd.agent import DeviceAgent, ExchangePlan, PGOAgentParameters, RBCDCluster, build_pose_graphs
    from dpgo_amd.robust import DistributedGNC, RobustCostParameters
    r, robots, k = 5, 3, 10
    om, n = oracle.read_g2o(os.path.join(DATA, "smallGrid3D.g2o"))
    d = om.d
    allm = _inject_outliers(oracle, om, n, k, seed=7)
    X0 = oracle.lift(oracle.chordal_initialization(om, n), r)
    ref_meas = _inject_outliers(oracle, om, n, k, seed=7)
    ap = dict(relChangeTol=2e-2, robustOptInnerIters=12, robustOptMinConvergenceRatio=0.8)
    Xref, info_o = oracle.multi_agent_gnc(ref_meas, n, robots, r, X0, barc=5.0, mu_step=1.4, max_updates=40,
                                          hess_recurrence=device_tcg_mode(n // robots, d, r),
                                          agent_params=oracle.AgentParameters(robust=True, **ap))
    assert info_o["history"][-1]["undecided"] == 0
    assert min(info_o["inner_iterations"]) < 12 <= max(info_o["inner_iterations"])  # both clauses of the trigger fire
    ranges, graphs = build_pose_graphs(to_product_measurements(allm), n, robots, r)
    plan = ExchangePlan(graphs)
    agents = {a: DeviceAgent(graphs, plan, a, X0[ranges[a][0]:ranges[a][1]], dpgo_amd.ROptParameters(precond="jacobi"))
              for a in range(robots)}
    gnc = DistributedGNC(RBCDCluster(plan, agents),
                         RobustCostParameters("GNC_TLS", GNCMaxNumIters=40, GNCBarc=5.0, GNCMuStep=1.4),
                         agent_params=PGOAgentParameters(robust=True, **ap))
    info = gnc.run()
    assert info["inner_iterations"] == info_o["inner_iterations"], (info["inner_iterations"], info_o["inner_iterations"])
    assert info["updates"] == info_o["updates"]
    for h, ho in zip(info["history"], info_o["history"]):
        assert (h["inliers"], h["outliers"], h["undecided"]) == (ho["inliers"], ho["outliers"], ho["undecided"])
    assert abs(info["cost"] - info_o["cost"]) <= 1e-6 * info_o["cost"]
    X = np.concatenate([agents[a].X.cpu().numpy() for a in range(robots)], axis=0)
    assert relerr(X, Xref) < 1e-6
    assert np.all(ref_meas.weight[-k:] < 1e-8) and np.all(ref_meas.weight[:om.m] > 1 - 1e-8)


def test_distributed_gnc_matches_oracle(oracle):
    """BASELINE configs[4] (multi-agent GNC): smallGrid3D + 10 injected outlier loop closures, 3 agents on one
    GPU.  Private AND shared loop closures are re-weighted on the device (shared ones read the neighbour tile
    buffer; Q's diagonal terms, the coupling blocks and the preconditioner are rebuilt in place).  Against the
    oracle's synchronous protocol at matched settings: same number of weight updates, same muInit, same
    classification, weights within 1e-6, final cost within 1e-6 relative; every injected outlier is rejected,
    no inlier is, and the final cost is the optimum of the clean graph."""
    import dpgo_amd
    from dpgo_amd.agent import DeviceAgent, ExchangePlan, RBCDCluster, build_pose_graphs
    from dpgo_amd.robust import DistributedGNC, RobustCostParameters
    r, robots, k, sweeps = 5, 3, 10, 2
    om, n = oracle.read_g2o(os.path.join(DATA, "smallGrid3D.g2o"))
    d = om.d
    allm = _inject_outliers(oracle, om, n, k, seed=7)
    X0 = oracle.lift(oracle.chordal_initialization(om, n), r)
    ref_meas = _inject_outliers(oracle, om, n, k, seed=7)
    Xref, info_o = oracle.multi_agent_gnc(ref_meas, n, robots, r, X0, inner_sweeps=sweeps, barc=5.0, mu_step=1.4,
                                          max_updates=40, hess_recurrence=device_tcg_mode(n // robots, d, r))
    assert info_o["history"][-1]["undecided"] == 0  # the protocol terminated by classification

    pm = to_product_measurements(allm)
    ranges, graphs = build_pose_graphs(pm, n, robots, r)
    plan = ExchangePlan(graphs)
    agents = {a: DeviceAgent(graphs, plan, a, X0[ranges[a][0]:ranges[a][1]], dpgo_amd.ROptParameters(precond="jacobi"))
              for a in range(robots)}
    cluster = RBCDCluster(plan, agents)
    gnc = DistributedGNC(cluster, RobustCostParameters("GNC_TLS", GNCMaxNumIters=40, GNCBarc=5.0, GNCMuStep=1.4),
                         inner_sweeps=sweeps)
    info = gnc.run()
    assert info["updates"] == info_o["updates"], (info["history"], info_o["history"])
    assert abs(info["muInit"] - info_o["muInit"]) <= 1e-8 * info_o["muInit"]
    for h, ho in zip(info["history"], info_o["history"]):
        assert (h["inliers"], h["outliers"], h["undecided"]) == (ho["inliers"], ho["outliers"], ho["undecided"])
    assert abs(info["cost"] - info_o["cost"]) <= 1e-6 * info_o["cost"]
    X = np.concatenate([agents[a].X.cpu().numpy() for a in range(robots)], axis=0)
    assert relerr(X, Xref) < 1e-6

    # weights: map every agent's edges back to global edges through (robot, frame) keys
    key_to_global = {}
    per = n // robots
    rob = np.minimum(np.arange(n) // per, robots - 1)
    loc = np.arange(n) - rob * per
    for e in range(ref_meas.m):
        key_to_global[(rob[ref_meas.p1[e]], loc[ref_meas.p1[e]], rob[ref_meas.p2[e]], loc[ref_meas.p2[e]])] = e
    seen = np.zeros(ref_meas.m, dtype=bool)
    for a, (idx, w) in gnc.weights().items():
        m = graphs[a].measurements()
        for pos, wv in zip(idx, w):
            e = key_to_global[(int(m.r1[pos]), int(m.p1[pos]), int(m.r2[pos]), int(m.p2[pos]))]
            seen[e] = True
            assert abs(wv - ref_meas.weight[e]) <= 1e-6, (a, e, wv, ref_meas.weight[e])
    assert seen.all()
    assert np.all(ref_meas.weight[-k:] < 1e-8) and np.all(ref_meas.weight[:om.m] > 1 - 1e-8)

    # the coupling rebuilt on the device gives the same G as constructG with the final weights
    cluster.exchange(None)
    _, per_robot = oracle.partition_contiguous(ref_meas, n, robots)
    for a in range(robots):
        s, e = ranges[a]
        agents[a].problem.updateLinearMatrixFromNeighbors(agents[a].nbr)
        Gdev = matrix_to_tiles(agents[a].problem.EucGrad(tiles_to_matrix(np.zeros((e - s, d + 1, r)))), d)
        sh = per_robot[a]["shared"]
        nbr = {}
        for q in range(sh.m):
            rb, fr = (int(sh.r2[q]), int(sh.p2[q])) if sh.r1[q] == a else (int(sh.r1[q]), int(sh.p1[q]))
            nbr[(rb, fr)] = X[ranges[rb][0] + fr]
        Gref = oracle.construct_G(e - s, d, r, sh, a, nbr)
        assert np.abs(Gdev - Gref).max() <= 1e-10 * max(1.0, np.abs(Gref).max())
    # and the outliers do not move the optimum: the clean graph's cost at its 5-robot demo optimum
    assert abs(info["cost"] - 1025.398) < 0.05


@pytest.mark.parametrize("d,r", [(3, 5), (3, 3), (2, 3), (2, 2), (3, 6)])
def test_rounding_matches_oracle(oracle, d, r):
    """K12 vs the SVD-based restatement of getTrajectoryInLocalFrame / InGlobalFrame (src/PGOAgent.cpp:718-767):
    a noisy lifted trajectory (the rounding input at a nearly rank-d solution), some blocks reflected so that the
    det < 0 branch of projectToRotationGroup runs; host-pointer and device flavours."""
    import torch
    import dpgo_amd
    rng = np.random.default_rng(100 * d + r)
    n = 333
    Ylift, _ = np.linalg.qr(rng.standard_normal((r, d)))
    X = np.zeros((n, d + 1, r))
    for i in range(n):
        Q, _ = np.linalg.qr(rng.standard_normal((d, d)))
        if np.linalg.det(Q) < 0:
            Q[:, 0] *= -1
        if i % 17 == 5:
            Q[:, 1] *= -1  # a reflected block: the det < 0 branch of projectToRotationGroup
        X[i, :d] = (Ylift @ Q + 0.05 * rng.standard_normal((r, d))).T
        X[i, d] = 3 * rng.standard_normal(r)
    for anchor in (None, X[7].copy()):
        ref = oracle.round_trajectory(X, d, anchor)
        assert (np.linalg.det(ref[:, :d]) > 0).all()
        Tm = dpgo_amd.round_trajectory(tiles_to_matrix(X), r, d, None if anchor is None else anchor.T)
        got = np.ascontiguousarray(np.asfortranarray(Tm).T).reshape(n, d + 1, d)
        assert np.abs(got - ref).max() < 1e-11
        Xd = torch.tensor(X, dtype=torch.float64, device="cuda")
        got_d = dpgo_amd.round_trajectory_device(Xd, None if anchor is None else anchor.T).cpu().numpy()
        assert np.abs(got_d - ref).max() < 1e-11
        assert np.abs(np.linalg.det(got_d[:, :d]) - 1).max() < 1e-12


def test_end_to_end_g2o_to_trajectory(oracle, tmp_path):
    """.g2o -> chordal initialisation -> device solve at rank d -> rounding -> CSV: on
    smallGrid3D the rounded rank-3 solution reaches the literature optimum and survives the CSV round trip."""
    import dpgo_amd
    from dpgo_amd.robust import solvePGO
    meas, n = dpgo_amd.read_g2o_file(os.path.join(DATA, "smallGrid3D.g2o"))
    T = solvePGO(meas, n, dpgo_amd.ROptParameters(precond="jacobi", gradnorm_tol=1e-6, RTR_iterations=100, RTR_tCG_iterations=200))
    Tm = dpgo_amd.round_trajectory(tiles_to_matrix(T), 3, 3)
    tiles = matrix_to_tiles(Tm, 3)
    om, _ = oracle.read_g2o(os.path.join(DATA, "smallGrid3D.g2o"))
    cost = 2 * oracle.QuadraticProblem(oracle.construct_Q(n, 3, om), None, 3, 3).f(tiles)
    assert abs(cost - 1025.398) < 2e-3  # rank-3 optimum == certified global optimum for this dataset
    assert np.abs(tiles[0, :3] - np.eye(3)).max() < 1e-12 and np.abs(tiles[0, 3]).max() < 1e-12
    f = str(tmp_path / "traj.csv")
    assert dpgo_amd.log_trajectory(3, n, Tm, f)
    assert np.abs(dpgo_amd.load_trajectory(f) - Tm).max() < 1e-12


def test_example_scripts_run_end_to_end(tmp_path):
    """examples/ (counterparts of the reference's MultiRobotExample / SingleRobotExample): the demo schedule
    converges below the reference's stop threshold (gradnorm < 0.1, examples/MultiRobotExample.cpp:227-229) to the
    known optimum and the rounded trajectories are written in the PGOLogger CSV format."""
    import subprocess
    import sys
    import dpgo_amd
    root = os.path.dirname(DATA)
    out_dir = str(tmp_path / "traj")
    p = subprocess.run([sys.executable, os.path.join(root, "examples", "multi_robot_example.py"), "5",
                        os.path.join(DATA, "smallGrid3D.g2o"), "--out-dir", out_dir], capture_output=True, text=True,
                       stdin=subprocess.DEVNULL, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    iters = [ln for ln in p.stdout.splitlines() if ln.startswith("Iter = ")]
    assert iters and len(iters) < 1000
    last = dict(kv.split(" = ") for kv in iters[-1].split(" | "))
    assert float(last["gradnorm"]) < 0.1 and abs(float(last["cost"]) - 1025.4) < 0.05
    T0 = dpgo_amd.load_trajectory(os.path.join(out_dir, "robot0.csv"))
    assert T0.shape == (3, 4 * 25) and np.abs(T0[:, :4] - np.eye(3, 4)).max() < 1e-9  # anchor = robot 0, pose 0
    for a in range(1, 5):
        Ta = dpgo_amd.load_trajectory(os.path.join(out_dir, "robot%d.csv" % a))
        R = Ta[:, :3]
        assert np.abs(R.T @ R - np.eye(3)).max() < 1e-9 and abs(np.linalg.det(R) - 1) < 1e-9
    single = str(tmp_path / "single.csv")
    p = subprocess.run([sys.executable, os.path.join(root, "examples", "single_robot_example.py"),
                        os.path.join(DATA, "smallGrid3D.g2o"), "--out", single], capture_output=True, text=True,
                       stdin=subprocess.DEVNULL, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "wrote" in p.stdout and dpgo_amd.load_trajectory(single).shape == (3, 4 * 125)


def _kitti_with_outliers(oracle, om, n, k):
    """kitti_00 + k injected outlier loop closures (random rotation, translation in [-5, 5]^2, 300-600 poses apart)."""
    rng = np.random.default_rng(11)
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
    lc = np.nonzero(~om.fixed)[0]
    z = np.zeros(k, dtype=np.int64)
    out = oracle.Measurements(2, z, np.array(p1), z.copy(), np.array(p2), Rk, rng.uniform(-5, 5, (k, 2)),
                              np.full(k, np.median(om.kappa[lc])), np.full(k, np.median(om.tau[lc])), np.ones(k),
                              np.zeros(k, dtype=bool))
    return oracle.Measurements.concat([om, out])


def test_distributed_gnc_kitti_reference_schedule(oracle):
    """BASELINE configs[4] with the REFERENCE's robust-cost schedule (include/DPGO/DPGO_robust.h:49-53: GNC-TLS, barc = 5,
    mu step 1.4) at r = 5 with the library's default preconditioner selection: kitti_00 cut into 4 agents, 25 injected
    outlier loop closures, Q values / coupling / preconditioner rebuilt on the device after each of the ~43 weight
    updates.  Reference side: the oracle's distributed GNC in the reference configuration (exact (Q_a + 0.1 I)^-1).  The
    two sides take different local steps, so what must agree is what GNC decides: the initial mu, the number of weight
    updates (+-2) and the final classification of every edge (all 25 outliers rejected, all 136 original loop closures
    kept, nothing undecided).  Final cost (north_star, 1e-6): RBCD on a chain cut of kitti_00 is far from converged after
    the schedule (and after 300 more sweeps: 1.5 % above, measured with the oracle), so the problem with the device's
    FINAL weights is solved centrally on the device and compared with the oracle's optimum for ITS final weights."""
    import dpgo_amd
    from dpgo_amd.agent import DeviceAgent, ExchangePlan, RBCDCluster, build_pose_graphs
    from dpgo_amd.robust import DistributedGNC, RobustCostParameters
    r, robots, k, sweeps = 5, 4, 25, 2
    om, n = oracle.read_g2o(os.path.join(DATA, "kitti_00.g2o"))
    ref_meas = _kitti_with_outliers(oracle, om, n, k)
    X0 = oracle.lift(oracle.chordal_initialization(om, n), r)
    Xref, info_o = oracle.multi_agent_gnc(ref_meas, n, robots, r, X0, inner_sweeps=sweeps, barc=5.0, mu_step=1.4,
                                          max_updates=80, precond="exact")
    last_o = info_o["history"][-1]
    assert (last_o["inliers"], last_o["outliers"], last_o["undecided"]) == (136, k, 0) and 30 <= info_o["updates"] <= 60
    dev_meas = _kitti_with_outliers(oracle, om, n, k)
    ranges, graphs = build_pose_graphs(to_product_measurements(dev_meas), n, robots, r)
    plan = ExchangePlan(graphs)
    agents = {a: DeviceAgent(graphs, plan, a, X0[ranges[a][0]:ranges[a][1]], dpgo_amd.ROptParameters())
              for a in range(robots)}
    cluster = RBCDCluster(plan, agents)
    gnc = DistributedGNC(cluster, RobustCostParameters("GNC_TLS", GNCMaxNumIters=80, GNCBarc=5.0, GNCMuStep=1.4),
                         inner_sweeps=sweeps)
    info = gnc.run()
    last = info["history"][-1]
    assert (last["inliers"], last["outliers"], last["undecided"]) == (136, k, 0), info["history"]
    assert abs(info["updates"] - info_o["updates"]) <= 2, (info["updates"], info_o["updates"])
    # (mu_0 = barc^2 / (2 max residual^2 - barc^2) after two unconverged sweeps: the two preconditioners leave the worst
    # outlier's residual 1-2 % apart)
    assert abs(info["muInit"] - info_o["muInit"]) <= 0.1 * info_o["muInit"]
    # every edge's final weight, gathered from the agents that hold it: the oracle's classification, edge for edge
    per = n // robots
    rob = np.minimum(np.arange(n) // per, robots - 1)
    loc = np.arange(n) - rob * per
    key_of = {(int(rob[a]), int(loc[a]), int(rob[b]), int(loc[b])): e
              for e, (a, b) in enumerate(zip(ref_meas.p1, ref_meas.p2))}
    w_dev = np.ones(len(ref_meas.p1))
    seen = np.zeros(len(ref_meas.p1), dtype=bool)
    for a, (idx, w) in gnc.weights().items():
        m = graphs[a].measurements()
        for pos, wv in zip(idx, w):
            e = key_of[(int(m.r1[pos]), int(m.p1[pos]), int(m.r2[pos]), int(m.p2[pos]))]
            assert not seen[e] or abs(w_dev[e] - wv) <= 1e-12  # both endpoints of a shared edge agree
            w_dev[e], seen[e] = wv, True
    assert seen[~ref_meas.fixed].all()
    assert np.array_equal(w_dev > 0.5, ref_meas.weight > 0.5)
    assert np.all(w_dev[-k:] < 1e-8) and np.all(w_dev[:om.m] > 1 - 1e-8)
    # the distributed iterate is a descent sequence that has not converged (chain cut): above the optimum on both sides
    f_dist, _ = cluster.central_cost_and_gradnorm()
    # final cost: the weighted problem each side ended with, solved centrally to a tight tolerance
    Qo = oracle.construct_Q(n, 2, ref_meas)
    prm_o = oracle.ROptParameters(gradnorm_tol=1e-5, RTR_iterations=80, RTR_tCG_iterations=500)
    oc = oracle.QuadraticOptimizer(oracle.QuadraticProblem(Qo, None, r, 2, precond="exact"), prm_o)
    oc.optimize(Xref)
    dev_meas.weight[:] = w_dev
    pg = dpgo_amd.PoseGraph(0, r, 2)
    pg.setMeasurements(to_product_measurements(dev_meas))
    central = dpgo_amd.QuadraticProblem(pg)
    gc = dpgo_amd.QuadraticOptimizer(central, dpgo_amd.ROptParameters(gradnorm_tol=1e-4, RTR_iterations=80,
                                                                       RTR_tCG_iterations=500, time_bound_s=120.0))
    Xd = np.concatenate([agents[a].X.cpu().numpy() for a in range(robots)], axis=0)
    gc.optimize(tiles_to_matrix(Xd))
    fo, fg = oc.result.fOpt, gc.getOptResult().fOpt
    assert abs(fg - fo) <= 1e-6 * abs(fo), (fg, fo)
    assert abs(2 * fg - 125.6807087875) <= 1e-6 * 125.68  # all outliers gone, all originals kept: kitti_00's optimum
    assert 2 * f_dist > 2 * fg and 2 * f_dist < info_o["cost"] * 1.05


def test_distributed_gnc_kitti_four_agents(oracle):
    """BASELINE configs[4] itself: kitti_00 (2-D, EDGE_SE2) cut into 4 agents, 25 injected outlier loop closures,
    GNC-TLS with barc = 5.  r = 3 (tile size 9: the non-span kernels with a coupling term); a coarse mu schedule
    (x8 per update) keeps the CPU oracle at seconds.  Same classification history as the oracle, every injected
    outlier rejected, no original loop closure rejected, weights within 1e-6, final cost within 1e-6 relative."""
    import dpgo_amd
    from dpgo_amd.agent import DeviceAgent, ExchangePlan, RBCDCluster, build_pose_graphs
    from dpgo_amd.robust import DistributedGNC, RobustCostParameters
    r, robots, k, sweeps = 3, 4, 25, 2
    om, n = oracle.read_g2o(os.path.join(DATA, "kitti_00.g2o"))

    def with_outliers():
        return _kitti_with_outliers(oracle, om, n, k)

    ref_meas = with_outliers()
    X0 = oracle.lift(oracle.chordal_initialization(om, n), r)
    Xref, info_o = oracle.multi_agent_gnc(ref_meas, n, robots, r, X0, inner_sweeps=sweeps, barc=5.0, mu_step=8.0,
                                          max_updates=12, hess_recurrence=device_tcg_mode(n // robots, 2, r))
    assert info_o["history"][-1]["undecided"] == 0
    ranges, graphs = build_pose_graphs(to_product_measurements(with_outliers()), n, robots, r)
    plan = ExchangePlan(graphs)
    agents = {a: DeviceAgent(graphs, plan, a, X0[ranges[a][0]:ranges[a][1]], dpgo_amd.ROptParameters(precond="jacobi"))
              for a in range(robots)}
    gnc = DistributedGNC(RBCDCluster(plan, agents),
                         RobustCostParameters("GNC_TLS", GNCMaxNumIters=12, GNCBarc=5.0, GNCMuStep=8.0),
                         inner_sweeps=sweeps)
    info = gnc.run()
    assert info["updates"] == info_o["updates"], (info["history"], info_o["history"])
    assert abs(info["muInit"] - info_o["muInit"]) <= 1e-7 * info_o["muInit"]
    for h, ho in zip(info["history"], info_o["history"]):
        assert (h["inliers"], h["outliers"], h["undecided"]) == (ho["inliers"], ho["outliers"], ho["undecided"])
    assert abs(info["cost"] - info_o["cost"]) <= 1e-6 * info_o["cost"]
    assert np.all(ref_meas.weight[-k:] < 1e-8) and np.all(ref_meas.weight[:om.m] > 1 - 1e-8)
    # device weights: all injected outliers (the last k global edges) end at 0 on whichever agent(s) hold them
    per = n // robots
    rob = np.minimum(np.arange(n) // per, robots - 1)
    loc = np.arange(n) - rob * per
    outlier_keys = {(int(rob[a]), int(loc[a]), int(rob[b]), int(loc[b]))
                    for a, b in zip(ref_meas.p1[-k:], ref_meas.p2[-k:])}
    seen = set()
    for a, (idx, w) in gnc.weights().items():
        m = graphs[a].measurements()
        for pos, wv in zip(idx, w):
            key = (int(m.r1[pos]), int(m.p1[pos]), int(m.r2[pos]), int(m.p2[pos]))
            if key in outlier_keys:
                seen.add(key)
                assert wv < 1e-8
            elif not m.fixedWeight[pos]:
                assert wv > 1 - 1e-6
    assert seen == outlier_keys


def _hierarchy_check(oracle, prob, op, tol=1e-9):
    """Every piece of the device-built hierarchy against the oracle's (amg_setup): prolongation blocks, Galerkin
    operators (pattern and values) of every level, dense inverse of the coarsest one."""
    import scipy.sparse as sp
    info = prob.multilevelInfo()
    m = op.amg_setup()
    b = op.b
    assert info["ks"] == m["ks"] and info["sizes"] == [L["n"] for L in m["levels"]] + [m["nc"]]
    for l, L in enumerate(m["levels"]):
        assert relerr(prob.multilevelGet(l, "P"), L["Pb"]) < 1e-12
        if l > 0:
            rowptr, colidx, vals = (prob.multilevelGet(l, w) for w in ("rowptr", "colidx", "A"))
            Ad = sp.bsr_matrix((vals, colidx, rowptr), shape=(L["n"] * b, L["n"] * b)).toarray()
            Ao = L["A"].toarray()
            assert relerr(Ad, Ao) < 1e-12
            assert ((Ad != 0) | (Ao == 0)).all()  # the symbolic pattern covers every non-zero
    last = len(m["levels"])
    if last > 1 or True:
        rowptr, colidx, vals = (prob.multilevelGet(last, w) for w in ("rowptr", "colidx", "A"))
        Ad = sp.bsr_matrix((vals, colidx, rowptr), shape=(m["nc"] * b, m["nc"] * b)).toarray()
        assert relerr(Ad, m["Ac"]) < 1e-12
    inv = prob.multilevelGet(last, "inverse")
    assert prob.multilevelCoarseBits() == op.amg_coarse_bits
    if op.amg_coarse_bits == 64:
        assert relerr(inv @ m["Ac"], np.eye(m["nc"] * b)) < tol
    else:  # stored in fp32: what the caller reads back is what the cycle applies
        assert (inv.astype(np.float32).astype(np.float64) == inv).all()
    assert relerr(inv, m["AcInv"]) < 1e-7
    if op.amg_coarse_bits == 32:
        # two fp64 inverses that agree to 1e-12 round to neighbouring fp32 values in a few entries; the solve
        # comparisons that follow run both sides with the SAME stored operator (the one just checked)
        m["AcInv"] = inv


@pytest.mark.parametrize("name,r,ks,bits", [("smallGrid3D", 5, None, 32), ("sphere2500", 5, None, 32),
                                            ("kitti_00", 3, None, 32), ("torus3D", 4, None, 32),
                                            ("sphere2500", 5, [4, 4], 32), ("torus3D", 5, [2, 4, 8], 32),
                                            ("kitti_00", 2, [4, 5], 32), ("smallGrid3D", 3, [8, 2], 32),
                                            ("sphere2500", 5, None, 64), ("kitti_00", 3, None, 64),
                                            ("torus3D", 5, [2, 4, 8], 64)])
def test_multilevel_preconditioner_matches_oracle(oracle, name, r, ks, bits):
    """precond = "multilevel" (the default): the aggregation-multigrid V-cycle that stands in for the reference's exact
    solve of Q + 0.1 I (src/QuadraticProblem.cpp:56-69; factor: src/PoseGraph.cpp:598-613) against the oracle's
    restatement (`amg`), with the default hierarchy and with explicit 3- and 4-level ones: the device-built hierarchy
    piece by piece, one application to 1e-9, one optimize at matched settings with identical iteration counts and
    iterates to 1e-7, fewer Hessian-vector products / a smaller gradient than block-Jacobi.  bits = storage precision of
    the dense level (64 is the default, 32 an opt-in; every product and sum is fp64 either way)."""
    import dpgo_amd
    om, n, d, Q, pg, prob = build_single_agent(oracle, name, r)
    X0 = oracle.lift(oracle.chordal_initialization(om, n), r)
    assert prob.multilevelCoarseBits() == 64  # the default; 32 is opt-in
    info = prob.setupMultilevel(ks, coarse_bits=bits)
    if ks is None:
        assert info["ks"] == oracle.amg_default_ks(n, d + 1)
    op = oracle.QuadraticProblem(Q, None, r, d, precond="amg", amg_k=info["ks"], amg_coarse_bits=bits)
    _hierarchy_check(oracle, prob, op)
    V = oracle.tangent_project(X0, np.random.default_rng(4).standard_normal(X0.shape), d)
    Zd = matrix_to_tiles(prob.PreConditioner(tiles_to_matrix(X0), tiles_to_matrix(V), precond="multilevel"), d)
    assert relerr(Zd, op.precondition(X0, V)) < 1e-9
    oo = oracle.QuadraticOptimizer(op, oracle.ROptParameters(), hess_recurrence=True)
    Xo = oo.optimize(X0)
    go = dpgo_amd.QuadraticOptimizer(prob, dpgo_amd.ROptParameters(precond="multilevel"))
    Xg = matrix_to_tiles(go.optimize(tiles_to_matrix(X0)), d)
    rg = go.getOptResult()
    assert rg.precond_used == "multilevel"
    Xa = np.abs(Xo).reshape(n * (d + 1), r)  # f is a cancellation-heavy sum (kitti_00): error scales with |X|^T|Q||X|
    scale = float((Xa * (abs(op.Qs) @ Xa)).sum())
    if name == "kitti_00" and rg.tcg_iterations != oo.result.tcg_iters:
        # condition ~1e8: the 1e-11 difference between two dense inverses can move tCG's stopping test by one step
        assert abs(rg.tcg_iterations - oo.result.tcg_iters) <= 1 and rg.rtr_iterations == oo.result.outer_iters
        assert abs(rg.fOpt - oo.result.fOpt) <= 1e-6 * abs(oo.result.fOpt) + 1e-14 * scale
    else:
        assert (rg.tcg_iterations, rg.rtr_iterations) == (oo.result.tcg_iters, oo.result.outer_iters)
        # (kitti_00, condition ~1e8: the same 1e-11 between the two dense inverses shows as 8e-7 in the iterate)
        assert relerr(Xg, Xo) < (1e-5 if name == "kitti_00" else 1e-7)
        assert abs(rg.fOpt - oo.result.fOpt) <= 1e-9 * abs(oo.result.fOpt) + 1e-14 * scale
    if ks is None:
        gj = dpgo_amd.QuadraticOptimizer(prob, dpgo_amd.ROptParameters(precond="jacobi"))  # same handle
        gj.optimize(tiles_to_matrix(X0))
        rj = gj.getOptResult()
        assert rg.gradNormOpt <= rj.gradNormOpt * 1.0001 or rg.tcg_iterations < rj.tcg_iterations
        if name != "smallGrid3D":
            assert rg.gradNormOpt < 0.5 * rj.gradNormOpt  # one RBCD iteration gets much further


@pytest.mark.parametrize("name", ["sphere2500", "smallGrid3D"])
def test_default_preconditioner_selection_matches_oracle(oracle, name):
    """precond = "auto" (the default).  A block WITHOUT coupling to other agents starts on a multilevel preconditioner
    (the tCG budget, not the trust-region boundary, ends its solves) and stays there; a handle forced to block-Jacobi
    switches after a solve that used half of its budget.  The multilevel choice is the additive two-level form wherever
    its persistent kernel runs (<= 256 aggregates), the V-cycle otherwise.  The decision is a function of the problem:
    setting Q again resets it.  Whatever a call ran (ROPTResult.precond_used), it matches the oracle run with that
    preconditioner at matched settings."""
    import dpgo_amd
    om, n, d, Q, pg, prob = build_single_agent(oracle, name, 5)
    r = 5
    go = dpgo_amd.QuadraticOptimizer(prob, dpgo_amd.ROptParameters())
    assert go.params_.precond == "auto"
    Xo = Xg = oracle.lift(oracle.chordal_initialization(om, n), r)
    used = []
    ops = {"jacobi": oracle.QuadraticProblem(Q, None, r, d, precond="jacobi")}
    assert prob.autoState() is True  # single block, no coupling: multilevel from the first call
    prob.autoState(False)            # ... the hysteresis is followed from the block-Jacobi side
    for call in range(3):
        Xg = matrix_to_tiles(go.optimize(tiles_to_matrix(Xg)), d)
        rg = go.getOptResult()
        used.append(rg.precond_used)
        if rg.precond_used == "multilevel" and "multilevel" not in ops:
            ops["multilevel"] = oracle.QuadraticProblem(Q, None, r, d, precond="amg", amg_k=prob.multilevelInfo()["ks"])
        if rg.precond_used == "additive" and "additive" not in ops:
            ops["additive"] = oracle.QuadraticProblem(Q, None, r, d, precond="amg_additive",
                                                      amg_k=prob.multilevelInfo()["ks"])
        oo = oracle.QuadraticOptimizer(ops[rg.precond_used], oracle.ROptParameters(), hess_recurrence=True)
        Xo = oo.optimize(Xo)
        assert (rg.tcg_iterations, rg.rtr_iterations) == (oo.result.tcg_iters, oo.result.outer_iters), (call, used)
        assert relerr(Xg, Xo) < 1e-7
    if name == "sphere2500":  # (the multilevel choice of a block this small is the additive form)
        assert used == ["jacobi", "additive", "additive"]
    else:
        assert used[0] == "jacobi"
    # a new Q resets the decision (repeated runs reproduce)
    prob.autoState(False)
    rowptr, colidx, vals = pg.quadraticMatrix()
    dpgo_amd.lib.check(prob._lib.dpgo_problem_set_Q_bsr(prob.handle, len(colidx), dpgo_amd.lib.ptr(rowptr),
                                                        dpgo_amd.lib.ptr(colidx), dpgo_amd.lib.ptr(vals)))
    assert prob.autoState() is True


def test_multilevel_hierarchy_follows_Q_values(oracle):
    """The hierarchy belongs to Q's values: after a re-weighting (here: every loop closure at weight 0.5, then a loop
    closure switched off, which breaks no chain, then an ODOMETRY edge at weight 0, which does) the next solve rebuilds
    the values on the device -- prolongation blocks, Galerkin operators, dense inverse -- and matches the oracle
    hierarchy of the re-weighted graph.  RGD with the multilevel preconditioner runs too."""
    import dpgo_amd
    om, n, d, Q, pg, prob = build_single_agent(oracle, "smallGrid3D", 5)
    r = 5
    X0 = oracle.lift(oracle.chordal_initialization(om, n), r)
    opt = dpgo_amd.QuadraticOptimizer(prob, dpgo_amd.ROptParameters(precond="multilevel"))
    opt.optimize(tiles_to_matrix(X0))
    prob.setReweightableEdges()
    idx = prob.reweightable_index
    for step in range(3):
        w = np.full(len(idx), 0.5)
        if step >= 1:
            w[-1] = 0.0
        if step == 2:
            w[0] = 0.0
        prob.setEdgeWeights(w)
        Xg = matrix_to_tiles(opt.optimize(tiles_to_matrix(X0)), d)
        om2 = om.subset(np.arange(om.m))
        om2.weight = om.weight.copy()
        om2.weight[pg.kept_index[idx]] = w
        Q2 = oracle.construct_Q(n, d, om2)
        op = oracle.QuadraticProblem(Q2, None, r, d, precond="amg", amg_k=prob.multilevelInfo()["ks"])
        _hierarchy_check(oracle, prob, op)
        oo = oracle.QuadraticOptimizer(op, oracle.ROptParameters(), hess_recurrence=True)
        Xo = oo.optimize(X0)
        rg = opt.getOptResult()
        assert (rg.tcg_iterations, rg.rtr_iterations) == (oo.result.tcg_iters, oo.result.outer_iters)
        assert relerr(Xg, Xo) < 1e-7
    # one preconditioned RGD step (src/QuadraticOptimizer.cpp:110-137) with the multilevel operator
    go = dpgo_amd.QuadraticOptimizer(prob, dpgo_amd.ROptParameters(method="RGD", precond="multilevel"))
    Xg = matrix_to_tiles(go.optimize(tiles_to_matrix(X0)), d)
    oo = oracle.QuadraticOptimizer(op, oracle.ROptParameters(method="RGD"))
    assert relerr(Xg, oo.optimize(X0)) < 1e-9


@pytest.mark.parametrize("N", [1, 5, 64, 65, 200, 777])
@pytest.mark.parametrize("mfma", [0, 1])
def test_dense_spd_inverse(N, mfma):
    """The blocked Gauss-Jordan kernels that invert the coarsest operator (kernels/dense.h), with the rank-64 updates
    on plain FMAs and on the fp64 matrix cores: A^-1 A = I to 1e-11 on random SPD matrices (condition ~1e3), sizes
    around the 64-block edges."""
    import ctypes as C
    import dpgo_amd.lib as L
    rng = np.random.default_rng(N)
    B = rng.standard_normal((N, N))
    A = B @ B.T / N + 1e-2 * np.eye(N) + np.diag(rng.uniform(0, 1, N))
    A = 0.5 * (A + A.T)
    out = np.zeros_like(A)
    L.check(L.load().dpgo_dense_spd_inverse(N, L.ptr(np.ascontiguousarray(A)), L.ptr(out), 0, mfma))
    assert np.isfinite(out).all()
    assert relerr(out @ A, np.eye(N)) < 1e-11
    assert relerr(out, np.linalg.inv(A)) < 1e-10


@pytest.mark.parametrize("d,r,n,hub_edges,drop", [(3, 5, 300, 60, 0), (2, 4, 300, 40, 7), (2, 3, 257, 30, 5),
                                                  (3, 3, 67, 20, 3), (3, 5, 5001, 50, 11)])
def test_multilevel_on_random_graphs_with_broken_chains(oracle, d, r, n, hub_edges, drop):
    """precond = "multilevel" away from the benchmark datasets: a hub row, ragged sizes (n not a multiple of k), both
    tile parities, and odometry chains with missing links (the prolongation restarts at the identity there).  One
    application matches the oracle's cycle to 1e-8 -- THAT is the parity claim of this test.  The whole solve is only
    smoke-tested here: a descent with the same iteration counts as the oracle (+-1 tCG step on these badly scaled
    problems), costs compared to 2 % when the counts agree (one step more or less before the trust-region boundary is
    another, equally valid step; whole-solve parity lives in the tests on the data sets).  (Far from the optimum the
    trust-region boundary, measured in the preconditioner's norm, decides the step: no claim is made here about which
    preconditioner gets further in three outer iterations.)"""
    import dpgo_amd
    om, T, hub = _random_graph(oracle, d, n, n // 2, hub_edges, seed=900 + n + d)
    if drop:  # remove every `drop`-th odometry edge but keep the graph connected through the loop closures
        chain = np.nonzero(om.p1 + 1 == om.p2)[0]
        lost = chain[5::max(len(chain) // drop, 1)][:drop]
        keep = np.setdiff1d(np.arange(om.m), lost)
        extra_p1, extra_p2 = om.p1[lost] - 1, om.p2[lost]  # bridge i-1 -> i+1 with a consistent measurement
        om = om.subset(keep)
        Rg = np.swapaxes(T[:, :d, :], 1, 2)
        z = np.zeros(len(lost), dtype=np.int64)
        br = oracle.Measurements(d, z, extra_p1, z.copy(), extra_p2, np.swapaxes(Rg[extra_p1], 1, 2) @ Rg[extra_p2],
                                 (np.swapaxes(Rg[extra_p1], 1, 2) @ (T[extra_p2, d] - T[extra_p1, d])[:, :, None])[:, :, 0],
                                 np.full(len(lost), 20.0), np.full(len(lost), 20.0), np.ones(len(lost)),
                                 np.zeros(len(lost), dtype=bool))
        om = oracle.Measurements.concat([om, br])
    Q = oracle.construct_Q(n, d, om)
    pg = dpgo_amd.PoseGraph(0, r, d)
    pg.setMeasurements(to_product_measurements(om))
    prob = dpgo_amd.QuadraticProblem(pg)
    op = oracle.QuadraticProblem(Q, None, r, d, precond="amg")
    rng = np.random.default_rng(3)
    X = oracle.polar_project(oracle.lift(T, r) + 0.1 * rng.standard_normal((n, d + 1, r)), d)
    V = oracle.tangent_project(X, rng.standard_normal(X.shape), d)
    Zd = matrix_to_tiles(prob.PreConditioner(tiles_to_matrix(X), tiles_to_matrix(V), precond="multilevel"), d)
    assert np.isfinite(Zd).all() and relerr(Zd, op.precondition(X, V)) < 1e-9
    assert prob.multilevelInfo()["ks"] == op.amg_setup()["ks"]
    go = dpgo_amd.QuadraticOptimizer(prob, dpgo_amd.ROptParameters(precond="multilevel"))
    Xg = matrix_to_tiles(go.optimize(tiles_to_matrix(X)), d)
    rg = go.getOptResult()
    oo = oracle.QuadraticOptimizer(op, oracle.ROptParameters(), hess_recurrence=True)
    oo.optimize(X)
    assert rg.success and np.isfinite(Xg).all() and rg.fOpt < rg.fInit
    assert rg.rtr_iterations == oo.result.outer_iters and abs(rg.tcg_iterations - oo.result.tcg_iters) <= 1
    # same counts: the same path to round-off.  One tCG step more or less before the trust-region boundary (round-off
    # decides on these ill-conditioned random graphs) is another, equally valid step whose cost is not comparable (seen:
    # 560 on the device against 630); the operator itself is pinned above to 1e-9
    # (equal TOTALS do not imply equal steps either: with the hub graph of 300 poses the second outer iteration ends on the
    # boundary after 20 or 19 steps depending on the summation order of the Galerkin product -- round 4's wave-parallel
    # setup kernels: 43 products and 12 661 against the oracle's 43 and 12 770; round 3's: 42 and 12 900 -- so the costs
    # are compared to 2 % only; what pins the arithmetic is the 1e-9 on the operator above)
    if rg.tcg_iterations == oo.result.tcg_iters:
        assert abs(rg.fOpt - oo.result.fOpt) <= 2e-2 * abs(oo.result.fOpt)


SWITCH_SETS = [
    ({}, "baseline"),
    ({"DPGO_ML_EARLY_STOP": "0"}, "bitwise"),      # tCG's residual test back in the Hessian-step kernel's prologue
    ({"DPGO_ITER_GRAPH": "1"}, "bitwise"),         # steady tCG iterations replayed from an instantiated hipGraph
    ({"DPGO_SPMM_SYMMETRIC": "1", "DPGO_ML_OPERATOR_BITS": "64"}, "oracle"),  # the cycle streams the fp64 operators (sym. storage)
    ({"DPGO_SPMM_SYMMETRIC": "1", "DPGO_ML_VECTOR_BITS": "64"}, "oracle"),    # fp32 operator copies, fp64 vectors inside the cycle
    ({"DPGO_SPMM_SYMMETRIC": "1", "DPGO_ML_DENSE_BITS": "32"}, "oracle"),     # ... and the dense level in fp32 as well
    ({"DPGO_SPMM_SYMMETRIC": "1"}, "oracle"),      # symmetric storage of Q: k_tcg_hess_sym, level-0 restriction / post-smoothing
    ({"DPGO_SPMM_SYMMETRIC": "1", "DPGO_OUTER_SYM": "0", "DPGO_STREAM_NT": "1"}, "oracle"),  # outer iteration on the plain copy
    ({"DPGO_SPMM_SYMMETRIC": "1", "DPGO_STREAM_NT": "1", "DPGO_HESS_DMA": "1"}, "oracle"),  # k_tcg_hess_sym_dma: own tiles by LDS-DMA, double-buffered
    ({"DPGO_SPMM_SYMMETRIC": "1", "DPGO_STREAM_NT": "1", "DPGO_HESS_DMA": "2"}, "oracle"),  # ... single-buffered, 3 waves per SIMD, 2 blocks in flight
    ({"DPGO_SPMM_SYMMETRIC": "1", "DPGO_TILE_WALK": "0"}, "oracle"),  # symmetric-storage kernels walk their tiles in index order
    ({"DPGO_SPMM_SYMMETRIC": "0", "DPGO_STREAM_NT": "1"}, "oracle"),  # plain storage with non-temporal single-use operands
    ({"DPGO_SETUP_THREADS": "1"}, "bitwise"),      # the hierarchy's symbolic set-up on the calling thread alone
    ({"DPGO_SETUP_THREADS": "5", "DPGO_SETUP_PIN": "0"}, "bitwise"),  # ... on five unpinned threads
    ({"DPGO_ML_GROWTH_CHUNKS": "4"}, "oracle"),    # aggregates grown and merged inside 4 index ranges (the oracle reads the same variable)
    ({"DPGO_ML_GRAPH": "0"}, "oracle"),            # index-run hierarchy (k_ml_post_ap on runs, in-workgroup restriction sums)
    ({"DPGO_ML_GRAPH": "0", "DPGO_ML_AP": "0"}, "oracle"),  # ... post-smoothing gathers through Q (k_ml_post)
    ({"DPGO_ML_SETUP_SERIAL": "1", "DPGO_GJ_MFMA": "0"}, "oracle"),  # round-3 set-up kernels, FMA rank-64 updates
    ({"DPGO_ML_DENSE_SYM": "1"}, "oracle"),        # dense level from the packed lower triangle on the matrix cores
    ({"DPGO_COARSE_NODES": "1", "DPGO_COARSE_NT": "1"}, "oracle"),  # one node per workgroup, non-temporal inverse
]


@pytest.mark.parametrize("workload", ["smallGrid3D", "grid:40x40x25"])
def test_kernel_selecting_switches_match_oracle(oracle, workload):
    """Every environment switch that selects a different KERNEL or storage on the multi-launch multilevel path
    (csrc/host.h, DPGO_OPTIONS; the tested configuration used to be the default one only), flipped one set at a time on
    smallGrid3D and on a 40 000-pose grid (the smallest block that can run the symmetric storage): two
    QuadraticOptimizer::optimize calls with the multilevel preconditioner against the oracle told the device's
    hierarchy, each call from the oracle's iterate -- same tCG / RTR counts, cost 1e-9 (+ 1e-4 of the call's decrease),
    iterate 1e-6.  Switches that must not change a single bit (where the residual test sits, how the launches are
    enqueued) are also compared bitwise with the default run.  The library reads its switches once; the test reloads
    them (dpgo_options_reload) and builds a fresh handle per set.  DPGO_ASYNC_SWEEP lives in the Python agent layer:
    test_stream_ordered_sweep_matches_phase_by_phase."""
    import hashlib
    import torch
    import dpgo_amd
    r = 5
    if workload.startswith("grid:"):
        om, n, Ttrue = oracle.synthetic_grid(*[int(v) for v in workload[5:].split("x")], seed=0)
        X0 = oracle.lift(oracle.perturbed_truth(Ttrue, seed=2), r)
    else:
        om, n = oracle.read_g2o(os.path.join(DATA, workload + ".g2o"))
        X0 = oracle.lift(oracle.chordal_initialization(om, n), r)
    d = om.d
    Q = oracle.construct_Q(n, d, om)
    lib = dpgo_amd.lib.load()
    names = sorted({k for sw, _ in SWITCH_SETS for k in sw})
    saved = {k: os.environ.get(k) for k in names}
    want = {}      # hierarchy -> [(tcg, rtr, fOpt, Xo)] of the oracle's calls
    digests = {}
    try:
        for sw, mode in SWITCH_SETS:
            for k in names:
                os.environ.pop(k, None)
            os.environ.update(sw)
            dpgo_amd.lib.check(lib.dpgo_options_reload())
            text = dpgo_amd.lib.describe_options()
            assert all(("%s=%s [set]" % kv) in text for kv in sw.items()), (sw, text)
            pg = dpgo_amd.PoseGraph(0, r, d)
            pg.setMeasurements(to_product_measurements(om))
            prob = dpgo_amd.QuadraticProblem(pg)
            prob.setPersistent(False)
            opt = dpgo_amd.QuadraticOptimizer(prob, dpgo_amd.ROptParameters(precond="multilevel"))
            ks = tuple(prob.setupMultilevel()["ks"])
            # (the cycle of a block that runs the symmetric storage streams fp32 copies of its level-0 operators unless told
            # otherwise: the oracle mirrors the storage, amg_operator_bits)
            want_bits = 64 if sw.get("DPGO_ML_OPERATOR_BITS") == "64" else 32
            obits = want_bits if (sw.get("DPGO_SPMM_SYMMETRIC") == "1" and n >= 40000) else 64
            vbits = 64 if sw.get("DPGO_ML_VECTOR_BITS") == "64" else obits
            hier = list(ks)
            cbits = 32 if (obits == 32 and vbits == 32 and sw.get("DPGO_ML_DENSE_BITS") == "32") else 64
            ks = ks + ((obits, vbits, cbits) if obits == 32 else ())
            if "DPGO_ML_GROWTH_CHUNKS" in sw:  # (another aggregation rule under the same sizes: its own oracle rows)
                ks = ks + ("ranges", sw["DPGO_ML_GROWTH_CHUNKS"])
            if ks not in want:
                op = oracle.QuadraticProblem(Q, None, r, d, precond="amg", amg_k=hier, amg_operator_bits=obits,
                                             amg_vector_bits=vbits, amg_coarse_bits=cbits)
                rows, Xo = [], X0
                for call in range(2):
                    oo = oracle.QuadraticOptimizer(op, oracle.ROptParameters(), hess_recurrence=True)
                    Xn = oo.optimize(Xo)
                    rows.append((oo.result.tcg_iters, oo.result.outer_iters, oo.result.fOpt, Xo, Xn))
                    Xo = Xn
                want[ks] = rows
            Xd = torch.tensor(X0, device="cuda", dtype=torch.float64)
            h = hashlib.sha256()
            for call, (tcg, rtr, fopt, Xin, Xout) in enumerate(want[ks]):
                Xd.copy_(torch.tensor(Xin))
                res = opt.optimizeDevice(Xd)
                assert res.precond_used == "multilevel" and prob.persistentInfo()["last_members"] == 0, (sw, res)
                assert (res.tcg_iterations, res.rtr_iterations) == (tcg, rtr), (sw, call)
                dec = abs(res.fInit - res.fOpt)
                assert abs(res.fOpt - fopt) <= 1e-9 * abs(fopt) + 1e-4 * dec, (sw, call)
                assert relerr(Xd.cpu().numpy(), Xout) < 1e-6, (sw, call)
                h.update(Xd.cpu().numpy().tobytes())
                h.update(repr((res.tcg_iterations, res.rtr_iterations, res.tCGStatus, res.fOpt, res.gradNormOpt)).encode())
            digests[tuple(sorted(sw.items()))] = (mode, h.hexdigest())
            if "DPGO_SPMM_SYMMETRIC" in sw and n >= 40000:
                assert prob.tcgKernelInfo()["symmetric"] == int(sw["DPGO_SPMM_SYMMETRIC"]), sw
            assert prob.multilevelOperatorBits() == dict(bits=want_bits, active=(obits == 32), vectors=(obits == 32 and vbits == 32),
                                                         dense=(cbits == 32)), sw
            del opt, prob
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        lib.dpgo_options_reload()
    base = digests[()][1]
    for key, (mode, dig) in digests.items():
        if mode == "bitwise":
            assert dig == base, key


def _early_stop_case(name, r):
    """Helper of test_early_residual_test_changes_nothing (own process: the knob is read once): three multi-launch
    multilevel solves, prints a digest of the iterate and the solver's counters."""
    import hashlib
    import json

    import dpgo_oracle as oracle
    import dpgo_amd
    om, n, d, Q, pg, prob = build_single_agent(oracle, name, r)
    prob.setPersistent(False)
    X = tiles_to_matrix(oracle.lift(oracle.chordal_initialization(om, n), r))
    go = dpgo_amd.QuadraticOptimizer(prob, dpgo_amd.ROptParameters(precond="multilevel"))
    rows = []
    for call in range(3):
        X = go.optimize(X)
        rg = go.getOptResult()
        rows.append([rg.tcg_iterations, rg.rtr_iterations, rg.tCGStatus, repr(rg.fOpt), repr(rg.gradNormOpt)])
    print("DIGEST " + json.dumps([hashlib.sha256(np.ascontiguousarray(X).tobytes()).hexdigest(), rows]))


@pytest.mark.parametrize("name,r", [("sphere2500", 5), ("kitti_00", 3)])
def test_early_residual_test_changes_nothing(name, r):
    """Multilevel tCG evaluates the residual test in the restriction kernel, one kernel before the Hessian-step kernel's
    prologue would (TcgStopCheck, kernels/multilevel.h): a converged run skips the V-cycle on its final residual.  Nothing
    of the solve may depend on it: the iterate after three solves is BIT-identical with the test left where it was
    (DPGO_ML_EARLY_STOP=0), and so are the counters, the status and the cost."""
    out = {}
    for flag in ("0", "1"):
        env = dict(os.environ, DPGO_ML_EARLY_STOP=flag)
        code = ("import sys; sys.path.insert(0, %r); import conftest, test_parity_gpu as t; t._early_stop_case(%r, %d)" %
                (os.path.dirname(os.path.abspath(__file__)), name, r))
        p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-3000:]
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith("DIGEST ")]
        assert lines, p.stdout[-2000:]
        out[flag] = lines[-1]
    assert out["0"] == out["1"]
    assert any(st in out["1"] for st in ("LCON", "SCON")), out["1"]  # (a run that ended on the residual test is in the sample)
