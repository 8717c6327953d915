"""CPU-side checks of the C-ABI library and the host logic (no GPU compute calls)."""
import ctypes as C
import os
import re
import sys

import numpy as np
import pytest

from conftest import DATA, ROOT, to_product_measurements


def test_library_loads_and_exports_every_declared_symbol():
    """Every function declared in include/dpgo_hip.h is exported by libdpgo_hip.so and bound in
    dpgo_amd/lib.py."""
    import dpgo_amd.lib as L
    lib = L.load()
    hdr = open(os.path.join(ROOT, "include", "dpgo_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(dpgo_[a-zA-Z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 30
    for name in sorted(declared):
        assert hasattr(lib, name), "libdpgo_hip.so does not export %s" % name
        assert name in L.SIGNATURES, "%s not bound in dpgo_amd/lib.py" % name
    assert set(L.SIGNATURES) == declared
    assert lib.dpgo_version().startswith(b"dpgo_hip")


def test_params_default_mirror_reference():
    """ROptParameters defaults (include/DPGO/DPGO_types.h:53-61) and the preconditioner shift
    (src/PoseGraph.cpp:603)."""
    import dpgo_amd
    import dpgo_amd.lib as L
    c = L.RoptParamsC()
    L.load().dpgo_ropt_params_default(C.byref(c))
    assert (c.method, c.verbose, c.RGD_use_preconditioner, c.RTR_iterations, c.RTR_tCG_iterations) == (0, 0, 1, 3, 50)
    assert (c.gradnorm_tol, c.RGD_stepsize, c.RTR_initial_radius, c.precond_shift) == (1e-2, 1e-3, 100.0, 1e-1)
    assert c.time_bound_s == 5.0
    p = dpgo_amd.ROptParameters().to_c()
    for f, _ in L.RoptParamsC._fields_:
        assert getattr(p, f) == getattr(c, f), f
    assert L.load().dpgo_supported(3, 5) == 1 and L.load().dpgo_supported(2, 2) == 1
    assert L.load().dpgo_supported(3, 2) == 0


def test_no_cpu_fallback_without_device():
    """Without a HIP device the product path fails loudly (DPGO_ERR_HIP); it never computes on the CPU."""
    import dpgo_amd
    import dpgo_amd.lib as L
    if dpgo_amd.device_count() > 0:
        pytest.skip("a GPU is present")
    h = L._P()
    rc = L.load().dpgo_problem_create(C.byref(h), 5, 3, 10, 0)
    assert rc == L.ERR_HIP
    assert b"device" in L.load().dpgo_last_error().lower()
    M = np.zeros((5, 8), order="F")
    with pytest.raises(dpgo_amd.DpgoError):
        dpgo_amd.LiftedSEManifold(5, 3, 2).project(M)
    # the RCCL transport and the dense-inverse kernels refuse too (no host-staged or CPU stand-in)
    hc = L._P()
    assert L.load().dpgo_comm_create(C.byref(hc), 1, 0, b"\0" * 128, 0) == L.ERR_HIP
    A = np.eye(3)
    assert L.load().dpgo_dense_spd_inverse(3, L.ptr(A), L.ptr(A.copy()), 0, 1) == L.ERR_HIP


def test_invalid_arguments_are_reported_not_aborted():
    """Reference: glog CHECK aborts (src/PoseGraph.cpp:19, QuadraticProblem.cpp:30-31); C ABI: codes."""
    import dpgo_amd
    import dpgo_amd.lib as L
    h = L._P()
    assert L.load().dpgo_problem_create(C.byref(h), 2, 3, 10, 0) == L.ERR_INVALID  # r < d
    assert L.load().dpgo_problem_create(C.byref(h), 5, 3, 0, 0) == L.ERR_INVALID  # n = 0
    assert L.load().dpgo_problem_create(C.byref(h), 9, 3, 4, 0) in (L.ERR_UNSUPPORTED, L.ERR_HIP)
    with pytest.raises(ValueError):
        dpgo_amd.PoseGraph(0, 2, 3)
    nn = C.c_int(0)
    assert L.load().dpgo_build_Q_bsr(0, 4, 3, 0, *([None] * 9), 0, None, 1.0, 1.0, C.byref(nn), None, None, None) \
        == L.ERR_INVALID


@pytest.mark.parametrize("name", ["tinyGrid3D", "smallGrid3D", "sphere2500", "kitti_00"])
def test_g2o_reader_and_Q_builder_match_oracle(oracle, name):
    """read_g2o_file (src/DPGO_utils.cpp:113-257) and constructQ (src/PoseGraph.cpp:381-491):
    product host code vs oracle."""
    import dpgo_amd
    path = os.path.join(DATA, name + ".g2o")
    om, n = oracle.read_g2o(path)
    pm, n2 = dpgo_amd.read_g2o_file(path)
    assert n == n2 and len(pm) == om.m and pm.d == om.d
    assert np.array_equal(pm.p1, om.p1) and np.array_equal(pm.p2, om.p2)
    assert np.array_equal(pm.R, om.R) and np.array_equal(pm.t, om.t)
    assert np.allclose(pm.kappa, om.kappa, rtol=1e-15) and np.allclose(pm.tau, om.tau, rtol=1e-15)
    assert np.array_equal(pm.fixedWeight, om.fixed)
    pg = dpgo_amd.PoseGraph(0, 5, om.d)
    pg.setMeasurements(pm)
    rp, ci, v = pg.quadraticMatrix()
    Q = oracle.construct_Q(n, om.d, om)
    assert np.array_equal(rp, Q.rowptr) and np.array_equal(ci, Q.colidx)
    assert np.abs(v - Q.vals).max() <= 1e-13 * np.abs(Q.vals).max()
    # symmetric, every block row has its diagonal block
    S = oracle.BSR(n, om.d + 1, rp, ci, v).to_scipy().tocsr()
    assert abs(S - S.T).max() <= 1e-12 * abs(S).max()
    rows = np.repeat(np.arange(n), np.diff(rp))
    assert np.array_equal(np.unique(rows[rows == ci]), np.arange(n))


def test_partition_and_coupling_match_oracle(oracle):
    """examples/MultiRobotExample.cpp:71-119 partition; constructQ/constructG with shared edges and a prior."""
    import dpgo_amd
    path = os.path.join(DATA, "smallGrid3D.g2o")
    om, n = oracle.read_g2o(path)
    pm, _ = dpgo_amd.read_g2o_file(path)
    ranges, per = oracle.partition_contiguous(om, n, 5)
    ranges_p, per_p = dpgo_amd.partition_contiguous(pm, n, 5)
    assert ranges == ranges_p
    r, d = 5, 3
    X = oracle.polar_project(np.random.default_rng(0).standard_normal((n, d + 1, r)), d)
    for a in range(5):
        s, e = ranges[a]
        na = e - s
        pg = dpgo_amd.PoseGraph(a, r, d)
        pg.setMeasurements(per_p[a])
        nshared = per[a]["shared"].m
        assert len(pg.sharedLoopClosures()) == nshared and pg.n() == na
        prior = {2: X[s + 2]}
        pg.setPrior(2, X[s + 2].T)
        priv = oracle.Measurements.concat([per[a]["odometry"], per[a]["private"]])
        Qa = oracle.construct_Q(na, d, priv, per[a]["shared"], my_id=a, priors=prior)
        rp, ci, v = pg.quadraticMatrix()
        assert np.array_equal(rp, Qa.rowptr) and np.array_equal(ci, Qa.colidx)
        assert np.abs(v - Qa.vals).max() <= 1e-12 * np.abs(Qa.vals).max()
        nbr = {pid: X[ranges[pid[0]][0] + pid[1]] for pid in pg.neighborPoseIDs()}
        with pytest.raises(LookupError):  # missing active neighbour pose (src/PoseGraph.cpp:515-520)
            pg.linearMatrix()
        pg.setNeighborPoses({k: t.T for k, t in nbr.items()})
        G = pg.linearMatrix()
        Ga = oracle.construct_G(na, d, r, per[a]["shared"], a, nbr, priors=prior)
        Gt = np.ascontiguousarray(G.T).reshape(na, d + 1, r)
        assert np.abs(Gt - Ga).max() <= 1e-12 * np.abs(Ga).max()


def test_inactive_neighbours_leave_the_data_matrices(oracle):
    """PoseGraph::setNeighborActive (src/PoseGraph.cpp:199-207) as PGOAgent::setRobotActive drives it (:1173-1184): the
    shared edges with an inactive neighbour are skipped by constructQ / constructG (:425-430, :527-532) -- unless
    useInactiveNeighbors is set and the pose is known --, the data matrices are dropped when the flag changes, the missing-
    pose check applies to ACTIVE neighbours only.  Product host code against the oracle's restatement."""
    import dpgo_amd
    path = os.path.join(DATA, "smallGrid3D.g2o")
    om, n = oracle.read_g2o(path)
    pm, _ = dpgo_amd.read_g2o_file(path)
    ranges, per = oracle.partition_contiguous(om, n, 5)
    _, per_p = dpgo_amd.partition_contiguous(pm, n, 5)
    r, d, a = 5, 3, 2
    X = oracle.polar_project(np.random.default_rng(1).standard_normal((n, d + 1, r)), d)
    s, e = ranges[a]
    pg = dpgo_amd.PoseGraph(a, r, d)
    pg.setMeasurements(per_p[a])
    nbrs = sorted({rob for rob, _ in pg.neighborPoseIDs()})
    assert nbrs == [1, 3] and pg.activeNeighborIDs() == nbrs and pg.hasNeighbor(1) and not pg.hasNeighbor(4)
    v0 = pg.q_version
    pg.setNeighborActive(4, False)  # not a neighbour: ignored (:200-202)
    pg.setNeighborActive(1, True)   # unchanged: nothing dropped
    assert pg.q_version == v0
    pg.setNeighborActive(3, False)
    assert pg.q_version > v0 and not pg.isNeighborActive(3) and pg.activeNeighborIDs() == [1]
    assert all(rob == 1 for rob, _ in pg.activeNeighborPublicPoseIDs())
    priv = oracle.Measurements.concat([per[a]["odometry"], per[a]["private"]])
    sh = oracle.active_shared_edges(per[a]["shared"], a, {3})
    assert 0 < sh.m < per[a]["shared"].m
    Qa = oracle.construct_Q(e - s, d, priv, sh, my_id=a)
    rp, ci, v = pg.quadraticMatrix()
    assert np.array_equal(rp, Qa.rowptr) and np.array_equal(ci, Qa.colidx)
    assert np.abs(v - Qa.vals).max() <= 1e-12 * np.abs(Qa.vals).max()
    Qfull = oracle.construct_Q(e - s, d, priv, per[a]["shared"], my_id=a)
    assert np.abs(v - Qfull.vals).max() > 1e-3 * np.abs(Qfull.vals).max()  # the edges really left
    # G: only the ACTIVE neighbour's poses are required
    act = {pid: X[ranges[pid[0]][0] + pid[1]] for pid in pg.neighborPoseIDs() if pid[0] == 1}
    pg.setNeighborPoses({k: t.T for k, t in act.items()})
    Gt = np.ascontiguousarray(pg.linearMatrix().T).reshape(e - s, d + 1, r)
    Ga = oracle.construct_G(e - s, d, r, sh, a, act)
    assert np.abs(Gt - Ga).max() <= 1e-12 * np.abs(Ga).max()
    # useInactiveNeighbors: an inactive neighbour's edge whose pose is known stays in the problem
    every = {pid: X[ranges[pid[0]][0] + pid[1]] for pid in pg.neighborPoseIDs()}
    pg.useInactiveNeighbors(True)
    pg.setNeighborPoses({k: t.T for k, t in every.items()})
    rp2, ci2, v2 = pg.quadraticMatrix()
    assert np.abs(v2 - Qfull.vals).max() <= 1e-12 * np.abs(Qfull.vals).max()
    Gfull = oracle.construct_G(e - s, d, r, per[a]["shared"], a, every)
    assert np.abs(np.ascontiguousarray(pg.linearMatrix().T).reshape(e - s, d + 1, r) - Gfull).max() <= 1e-12 * np.abs(Gfull).max()
    # back to active: the full matrices again, and the missing-pose check with them
    pg.useInactiveNeighbors(False)
    pg.setNeighborActive(3, True)
    pg.setNeighborPoses({k: t.T for k, t in act.items()})
    assert np.abs(pg.quadraticMatrix()[2] - Qfull.vals).max() <= 1e-12 * np.abs(Qfull.vals).max()
    with pytest.raises(LookupError):
        pg.linearMatrix()
    # the votes skip inactive robots (src/PGOAgent.cpp:861-862, 1016-1017), product rules = oracle rules
    from dpgo_amd.agent import PGOAgentParameters, PGOAgentStatus, should_terminate, should_update_measurement_weights
    prm, oprm = PGOAgentParameters(), oracle.AgentParameters()
    team = {q: PGOAgentStatus(q, "INITIALIZED", 0, 7, q != 3, 0.0) for q in range(5)}
    oteam = {q: oracle.AgentStatus(q, "INITIALIZED", 0, 7, q != 3, 0.0) for q in range(5)}
    for off in ((), (3,), (2,)):
        assert should_terminate(7, prm, 0, team, 5, off) == oracle.should_terminate(7, oprm, 0, oteam, 5, off) == (off == (3,))
    from dataclasses import replace
    rprm, orprm = replace(prm, robust=True), oracle.AgentParameters(**{**oprm.__dict__, "robust": True})
    for off in ((), (3,)):
        assert should_update_measurement_weights(rprm, 0, 1, 0, team, 5, off) == \
            oracle.should_update_weights(orprm, 0, 1, 0, oteam, 5, off) == (off == (3,))


def test_duplicate_and_irrelevant_edges(oracle):
    """PoseGraph::addMeasurement drops irrelevant edges (src/PoseGraph.cpp:68-71) and duplicates (:83-88)."""
    import dpgo_amd
    pm, n = dpgo_amd.read_g2o_file(os.path.join(DATA, "tinyGrid3D.g2o"))
    dup = dpgo_amd.RelativeSEMeasurements.concatenate([pm, pm.select(np.array([0, 1]))])
    other = pm.select(np.array([0]))
    other.r1[:] = 3
    other.r2[:] = 4
    pg = dpgo_amd.PoseGraph(0, 5, 3)
    pg.setMeasurements(dpgo_amd.RelativeSEMeasurements.concatenate([dup, other]))
    assert len(pg.measurements()) == len(pm)
    pg2 = dpgo_amd.PoseGraph(0, 5, 3)
    pg2.setMeasurements(pm)
    assert np.array_equal(pg.quadraticMatrix()[2], pg2.quadraticMatrix()[2])


def test_trajectory_csv_round_trip(tmp_path):
    """PGOLogger::logTrajectory / loadTrajectory (src/PGOLogger.cpp:56-155): header, quaternion convention
    (Eigen: x, y, z, w columns), 3-D only."""
    from dpgo_amd.trajectory import load_trajectory, log_measurements, log_trajectory
    import dpgo_amd
    rng = np.random.default_rng(5)
    n, d = 7, 3
    T = np.zeros((d, (d + 1) * n), order="F")
    for i in range(n):
        Q, _ = np.linalg.qr(rng.standard_normal((d, d)))
        if np.linalg.det(Q) < 0:
            Q[:, 0] *= -1
        T[:, 4 * i:4 * i + 3] = Q
        T[:, 4 * i + 3] = rng.standard_normal(3)
    f = str(tmp_path / "traj.csv")
    assert log_trajectory(d, n, T, f)
    assert open(f).readline().strip() == "pose_index,qx,qy,qz,qw,tx,ty,tz"
    assert np.abs(load_trajectory(f) - T).max() < 1e-14
    assert log_trajectory(2, n, np.zeros((2, 3 * n)), f) is False  # the reference returns silently for d == 2
    # identity rotation -> quaternion (0, 0, 0, 1)
    log_trajectory(3, 1, np.hstack([np.eye(3), np.zeros((3, 1))]), f)
    assert open(f).read().splitlines()[1].split(",")[1:5] == ["0", "0", "0", "1"]
    meas, _ = dpgo_amd.read_g2o_file(os.path.join(DATA, "tinyGrid3D.g2o"))
    g = str(tmp_path / "meas.csv")
    assert log_measurements(meas, g)
    lines = open(g).read().splitlines()
    assert lines[0].startswith("robot_src,pose_src,robot_dst,pose_dst,qx,qy,qz,qw,tx,ty,tz,kappa,tau,")
    assert len(lines) == len(meas) + 1


def test_aggregates_grow_and_merge_inside_index_ranges():
    """Large blocks (>= 65 536 poses) grow and merge their graph aggregates independently inside 8 contiguous index ranges
    (ml_growth_chunks; one host thread per range in the library): the rule is the oracle's (amg_growth_chunks and the range
    arguments of its growth / merge), node for node, and does not depend on the number of threads that execute it.  Checked
    on a 20 x 20 x 12 grid with the range count forced on both sides (DPGO_ML_GROWTH_CHUNKS = 5: ranges that do not fall on
    layer boundaries), on the default rule's threshold, and once at 65 536 nodes of a chain (8 ranges by size)."""
    import ctypes as C
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import dpgo_oracle as O
    import dpgo_amd.lib as L
    lib = L.load()
    assert O.amg_growth_chunks(65535) == 1 and O.amg_growth_chunks(65536) == 8 and O.amg_growth_chunks(1000000) == 8

    def device_side(Q, S, cap):
        n = Q.n
        rp, ci = L.i32(Q.rowptr), L.i32(Q.colidx)  # (kept alive across the calls)
        lab_c, par_c, na = np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32), C.c_int(0)
        L.check(lib.dpgo_multilevel_graph_aggregates(n, L.ptr(rp), L.ptr(ci), S, L.ptr(lab_c), L.ptr(par_c), C.byref(na)))
        lab_m, par_m, nm = np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32), C.c_int(0)
        L.check(lib.dpgo_multilevel_merged_aggregates(n, L.ptr(rp), L.ptr(ci), S, cap, L.ptr(lab_m), L.ptr(par_m),
                                                      C.byref(nm)))
        return (lab_c, par_c, na.value), (lab_m, par_m, nm.value)

    meas, n, _ = O.synthetic_grid(20, 20, 12, seed=0)
    Q = O.construct_Q(n, 3, meas)
    keep = {k: os.environ.get(k) for k in ("DPGO_ML_GROWTH_CHUNKS", "DPGO_SETUP_THREADS")}
    try:
        results = {}
        for chunks in (1, 5):
            for threads in (1, 3, 8):
                os.environ["DPGO_ML_GROWTH_CHUNKS"], os.environ["DPGO_SETUP_THREADS"] = str(chunks), str(threads)
                L.check(lib.dpgo_options_reload())
                assert O.amg_growth_chunks(n) == chunks
                S, cap = 30, 45
                (lab_c, par_c, na), (lab_m, par_m, nm) = device_side(Q, S, cap)
                lab_o, ptr_o, mem_o, par_o, _ = O.amg_graph_aggregates(Q, S)
                assert na == len(ptr_o) - 1 and np.array_equal(lab_c, lab_o) and np.array_equal(par_c, par_o), (chunks, threads)
                lab_q, ptr_q, mem_q, par_q, _ = O.amg_merge_small_aggregates(Q, S, lab_o, ptr_o, mem_o, cap)
                assert nm == len(ptr_q) - 1 and np.array_equal(lab_m, lab_q) and np.array_equal(par_m, par_q), (chunks, threads)
                results.setdefault(chunks, []).append((lab_m.copy(), par_m.copy()))
                if chunks > 1:  # no aggregate across a range boundary; ids ascend range after range; trees stay inside
                    rng_of = np.minimum(np.arange(n) * chunks // n, chunks - 1)
                    for c in range(chunks):  # (the boundaries are n c / chunks rounded down)
                        lo, hi = n * c // chunks, n * (c + 1) // chunks
                        rng_of[lo:hi] = c
                    first = np.array([rng_of[mem_q[ptr_q[a]]] for a in range(nm)])
                    assert np.all(np.diff(first) >= 0)
                    assert np.all(rng_of == first[lab_q])
                    assert np.all(rng_of[par_q[par_q >= 0]] == rng_of[par_q >= 0])
                    sizes = np.diff(ptr_q)
                    assert sizes.max() <= cap and sizes.min() >= 1 and sorted(mem_q.tolist()) == list(range(n))
        for chunks, runs in results.items():  # the thread count changes nothing
            assert all(np.array_equal(runs[0][0], r[0]) and np.array_equal(runs[0][1], r[1]) for r in runs[1:])
        assert not np.array_equal(results[1][0][0], results[5][0][0])  # (the rule itself does change the aggregates)
        # the default rule at its threshold: a 65 536-node chain, 8 ranges
        del os.environ["DPGO_ML_GROWTH_CHUNKS"]
        os.environ["DPGO_SETUP_THREADS"] = "0"
        L.check(lib.dpgo_options_reload())
        nn = 65536
        rowptr = np.zeros(nn + 1, dtype=np.int64)
        cols = []
        for i in range(nn):
            row = [j for j in (i - 1, i, i + 1) if 0 <= j < nn]
            cols.extend(row)
            rowptr[i + 1] = len(cols)

        class Pattern:
            n, rowptr, colidx = nn, None, None
        Pq = Pattern()
        Pq.rowptr, Pq.colidx = rowptr, np.array(cols, dtype=np.int64)
        (lab_c, par_c, na), (lab_m, par_m, nm) = device_side(Pq, 100, 150)
        lab_o, ptr_o, mem_o, par_o, _ = O.amg_graph_aggregates(Pq, 100)
        assert np.array_equal(lab_c, lab_o) and np.array_equal(par_c, par_o)
        assert na == 8 * 82 and np.all(lab_o[np.arange(1, 8) * 8192] != lab_o[np.arange(1, 8) * 8192 - 1])  # 8192 = 81 x 100 + 92
        lab_q, ptr_q, _, par_q, _ = O.amg_merge_small_aggregates(Pq, 100, lab_o, ptr_o, mem_o, 150)
        assert nm == len(ptr_q) - 1 and np.array_equal(lab_m, lab_q) and np.array_equal(par_m, par_q)
    finally:
        for k, v in keep.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        lib.dpgo_options_reload()


@pytest.mark.parametrize("name,ks", [("smallGrid3D", None), ("kitti_00", None), ("sphere2500", [4, 8])])
def test_multilevel_hierarchy_rules(name, ks):
    """Host-side rules of precond = "multilevel": the library's default aggregate sizes (dpgo_multilevel_default_ks, no
    GPU code) equal the oracle's mirror on the benchmark sizes, and the chain prolongations the device kernel
    (k_ml_build_P) restates reproduce the chain Laplacian's kernel on every level: for an odometry-only graph
    Q P_0 P_1 ... C has no residual inside an aggregate."""
    import ctypes as C
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import dpgo_oracle as O
    import dpgo_amd.lib as L
    lib = L.load()
    for n_, d_ in [(1, 3), (25, 3), (125, 3), (2500, 3), (5000, 3), (12500, 3), (25000, 3), (50000, 3), (100000, 3),
                   (1000000, 3), (4541, 2), (1136, 2), (39999, 2), (40000, 2)]:
        buf = np.zeros(16, dtype=np.int32)
        cnt = C.c_int(16)
        L.check(lib.dpgo_multilevel_default_ks(n_, d_, L.ptr(buf), C.byref(cnt)))
        assert [int(v) for v in buf[:cnt.value]] == O.amg_default_ks(n_, d_ + 1), (n_, d_)
    om, n = O.read_g2o(os.path.join(DATA, name + ".g2o"))
    d, b = om.d, om.d + 1
    ks = ks or O.amg_default_ks(n, b)
    odo = om.subset(np.nonzero(om.p1 + 1 == om.p2)[0])
    Qo = O.construct_Q(n, d, odo)
    # the greedy aggregation of the library (host code, no GPU) is the oracle's, node for node, on the full graph
    Qf = O.construct_Q(n, d, om)
    for S in (4, 16, abs(ks[0]) if len(ks) == 1 else 7):
        lab_o, ptr_o, _, par_o, _ = O.amg_graph_aggregates(Qf, S)
        lab_c, par_c, na = np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32), C.c_int(0)
        L.check(lib.dpgo_multilevel_graph_aggregates(n, L.ptr(L.i32(Qf.rowptr)), L.ptr(L.i32(Qf.colidx)), S, L.ptr(lab_c),
                                                     L.ptr(par_c), C.byref(na)))
        assert na.value == len(ptr_o) - 1 and np.array_equal(lab_c, lab_o) and np.array_equal(par_c, par_o)
    # ... and so is the merge of the growth's fragments (ks = [-S, -cap]: the additive preconditioner's aggregates beyond
    # ~3 500 poses, where an aggregate is a workgroup): same labels and trees, no aggregate beyond the bound, every
    # aggregate connected through its tree (one root), fewer aggregates than the plain growth leaves
    for S, cap in ((4, 6), (16, 24), (28, 42)):
        lab_g, ptr_g, mem_g, _, _ = O.amg_graph_aggregates(Qf, S)
        lab_o, ptr_o, mem_o, par_o, _ = O.amg_merge_small_aggregates(Qf, S, lab_g, ptr_g, mem_g, cap)
        lab_c, par_c, na = np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32), C.c_int(0)
        L.check(lib.dpgo_multilevel_merged_aggregates(n, L.ptr(L.i32(Qf.rowptr)), L.ptr(L.i32(Qf.colidx)), S, cap,
                                                      L.ptr(lab_c), L.ptr(par_c), C.byref(na)))
        assert na.value == len(ptr_o) - 1 and np.array_equal(lab_c, lab_o) and np.array_equal(par_c, par_o), (S, cap)
        sizes = np.diff(ptr_o)
        assert sizes.max() <= cap and sizes.min() >= 1 and sorted(mem_o.tolist()) == list(range(n))
        assert len(sizes) <= len(ptr_g) - 1
        roots = np.bincount(lab_o[par_o < 0], minlength=len(sizes))
        assert np.all(roots == 1), "a merged aggregate is connected: one tree"
        assert np.all(lab_o[par_o[par_o >= 0]] == lab_o[par_o >= 0])
    if len(ks) == 1 and ks[0] < 0:
        # graph aggregates (the default two-level hierarchy): the prolongation composed along each aggregate's
        # breadth-first tree (k_ml_build_P_tree) reproduces the rigid-body modes -- on the odometry chain, and on a spanning
        # tree of the whole graph whose edges are measured in either direction
        lab, ptr, mem, parent, pslot = O.amg_graph_aggregates(Qo, -ks[0])
        assert sorted(mem.tolist()) == list(range(n)) and np.all(np.diff(ptr) <= -ks[0]) and np.all(np.diff(ptr) >= 1)
        assert np.all(lab[parent[parent >= 0]] == lab[parent >= 0])
        V = O.amg_tree_prolongation(Qo, d, mem, parent, pslot)
        R = (Qo.to_scipy() @ V.reshape(n * b, b)).reshape(n, b, b)
        nb_same = np.array([all(lab[Qo.colidx[t]] == lab[i] for t in range(Qo.rowptr[i], Qo.rowptr[i + 1])) for i in range(n)])
        assert nb_same.any() and (~nb_same).any()
        assert np.abs(R[nb_same]).max() <= 1e-7 * np.abs(Qo.vals).max()
        assert np.abs(R[~nb_same]).max() > 1e-3 * np.abs(Qo.vals).max()  # rows at a cut see a residual
        # spanning tree of ALL edges, every other edge reversed (measured child -> parent): one aggregate, no residual
        Qa = O.construct_Q(n, d, om)
        _, _, mem_a, par_a, _ = O.amg_graph_aggregates(Qa, n)
        tree = {(int(par_a[i]), int(i)) for i in range(n) if par_a[i] >= 0}
        sel, flip = [], []
        for e in range(len(om.p1)):
            a, c = int(om.p1[e]), int(om.p2[e])
            if (a, c) in tree or (c, a) in tree:
                tree.discard((a, c)), tree.discard((c, a))
                sel.append(e), flip.append(len(sel) % 2 == 0)
        tm = om.subset(np.array(sel))
        for q, f in enumerate(flip):
            if f:  # the same constraint measured the other way round: T^-1
                Rm, tv = tm.R[q].copy(), tm.t[q].copy()
                tm.R[q], tm.t[q] = Rm.T, -Rm.T @ tv
                tm.p1[q], tm.p2[q] = tm.p2[q], tm.p1[q]
        if len(sel) == n - 1:  # (the data set's graph is connected)
            Qt = O.construct_Q(n, d, tm)
            _, ptr_t, mem_t, par_t, ps_t = O.amg_graph_aggregates(Qt, n)
            assert len(ptr_t) == 2
            Vt = O.amg_tree_prolongation(Qt, d, mem_t, par_t, ps_t)
            Rt = (Qt.to_scipy() @ Vt.reshape(n * b, b)).reshape(n, b, b)
            # (the reversed edges carry their information rotated: kappa, tau isotropic, so the kernel is the same)
            assert np.abs(Rt).max() <= 1e-7 * np.abs(Qt.vals).max()
        return
    Pbs = O.amg_chain_prolongations(Qo, d, ks)
    # V = P_0 P_1 ... restricted to blocks: V_i = Pb_0[i] Pb_1[i // k_0] ...
    V = np.zeros((n, b, b))
    for i in range(n):
        M, node = np.eye(b), i
        for Pb, k in zip(Pbs, ks):
            M = M @ Pb[node]
            node //= k
        V[i] = M
    R = (Qo.to_scipy() @ V.reshape(n * b, b)).reshape(n, b, b)
    span = int(np.prod(ks))
    inner = np.array([i for i in range(n) if i % span not in (0, span - 1) and i != n - 1])  # rows not touching a cut
    # (1e-9-level: the g2o rotations are not exactly orthonormal, so T is recovered to ~1e-9)
    assert np.abs(R[inner]).max() <= 1e-7 * np.abs(Qo.vals).max()
    assert np.abs(R[np.arange(span - 1, n - 1, span)]).max() > 1e-3 * np.abs(Qo.vals).max()  # cut rows see a residual



def _random_multi_robot_graph(rng, d, n, robots, extra):
    """Random connected pose graph (odometry chain + `extra` random loop closures, some reversed, random weights),
    cut into `robots` contiguous blocks; returns oracle-style global arrays."""
    def rot():
        Qm, _ = np.linalg.qr(rng.standard_normal((d, d)))
        if np.linalg.det(Qm) < 0:
            Qm[:, 0] *= -1
        return Qm
    p1 = list(range(n - 1))
    p2 = list(range(1, n))
    seen = set(zip(p1, p2))
    extra = min(extra, n * (n - 1) - (n - 1))  # distinct ordered pairs still available
    while len(p1) < n - 1 + extra:
        a, b = (int(v) for v in rng.integers(0, n, 2))
        if a != b and (a, b) not in seen:
            seen.add((a, b))
            p1.append(a)
            p2.append(b)
    m = len(p1)
    return dict(p1=np.array(p1), p2=np.array(p2), R=np.stack([rot() for _ in range(m)]),
                t=rng.standard_normal((m, d)), kappa=rng.uniform(0.5, 50, m), tau=rng.uniform(0.5, 50, m),
                weight=rng.uniform(0.1, 1.0, m))


def test_host_builders_on_random_multi_robot_graphs(oracle):
    """Property test (hypothesis): dpgo_build_Q_bsr / dpgo_build_G_coupling (the host C++ restatements of
    PoseGraph::constructQ / constructG, src/PoseGraph.cpp:381-580) against the oracle on random graphs -- 2-D and 3-D,
    reversed loop closures, non-unit weights, uneven blocks, priors, every robot's view."""
    from hypothesis import given, settings, strategies as st
    import dpgo_amd
    O = oracle

    @settings(max_examples=25, deadline=None)
    @given(seed=st.integers(0, 10 ** 6), d=st.sampled_from([2, 3]), n=st.integers(4, 40), robots=st.integers(1, 4),
           extra=st.integers(0, 30), r_extra=st.integers(0, 2))
    def check(seed, d, n, robots, extra, r_extra):
        robots = min(robots, n)
        rng = np.random.default_rng(seed)
        g = _random_multi_robot_graph(rng, d, n, robots, extra)
        r = d + r_extra
        z = np.zeros(len(g["p1"]), dtype=np.int64)
        om = O.Measurements(d, z, g["p1"], z.copy(), g["p2"], g["R"], g["t"], g["kappa"], g["tau"], g["weight"],
                            np.zeros(len(z), dtype=bool))
        pm = to_product_measurements(om)
        ranges, per = O.partition_contiguous(om, n, robots)
        ranges_p, per_p = dpgo_amd.partition_contiguous(pm, n, robots)
        assert ranges == ranges_p
        X = O.polar_project(rng.standard_normal((n, d + 1, r)), d)
        for a in range(robots):
            s, e = ranges[a]
            pg = dpgo_amd.PoseGraph(a, r, d)
            pg.setMeasurements(per_p[a])
            assert pg.n() == e - s
            prior = {0: X[s]} if (seed + a) % 2 == 0 else None
            if prior:
                pg.setPrior(0, X[s].T)
            priv = O.Measurements.concat([per[a]["odometry"], per[a]["private"]])
            Qa = O.construct_Q(e - s, d, priv, per[a]["shared"], my_id=a, priors=prior)
            rp, ci, v = pg.quadraticMatrix()
            assert np.array_equal(rp, Qa.rowptr) and np.array_equal(ci, Qa.colidx)
            assert np.abs(v - Qa.vals).max() <= 1e-12 * max(1.0, np.abs(Qa.vals).max())
            nbr = {pid: X[ranges[pid[0]][0] + pid[1]] for pid in pg.neighborPoseIDs()}
            pg.setNeighborPoses({k: t.T for k, t in nbr.items()})
            if per[a]["shared"].m or prior:
                Ga = O.construct_G(e - s, d, r, per[a]["shared"], a, nbr, priors=prior)
                Gt = np.ascontiguousarray(pg.linearMatrix().T).reshape(e - s, d + 1, r)
                assert np.abs(Gt - Ga).max() <= 1e-12 * max(1.0, np.abs(Ga).max())
                # the coupling-operator form used on the device gives the same G
                slots, crp, cci, cv, G0 = pg.couplingMatrix()
                Gop = np.ascontiguousarray(G0.T).reshape(e - s, d + 1, r).copy()
                for i in range(e - s):
                    for t in range(crp[i], crp[i + 1]):
                        Gop[i] += cv[t] @ nbr[slots[cci[t]]]
                assert np.abs(Gop - Ga).max() <= 1e-12 * max(1.0, np.abs(Ga).max())

    check()


@pytest.mark.parametrize("name", ["tinyGrid3D", "smallGrid3D", "kitti_00"])
def test_odometry_initialisation_matches_oracle(oracle, name):
    """dpgo_odometry_initialization (odometryInitialization, src/DPGO_solver.cpp:271-303; host code of the library)
    against the oracle's restatement; a chain with a missing link is refused (reference: CHECK(m.p1 == src)).  The
    chordal initialisation runs on the device: without one it fails loudly (its parity test is a gpu test)."""
    import dpgo_amd
    from dpgo_amd.initialization import chordal_initialization, odometry_initialization
    path = os.path.join(DATA, name + ".g2o")
    om, n = oracle.read_g2o(path)
    pm, _ = dpgo_amd.read_g2o_file(path)
    odo_p = pm.select(np.nonzero(pm.p1 + 1 == pm.p2)[0])
    odo_o = om.subset(np.nonzero(om.p1 + 1 == om.p2)[0])
    assert len(odo_p) == n - 1
    To = odometry_initialization(odo_p, n)
    assert To.shape == (n, om.d + 1, om.d)
    Too = oracle.odometry_initialization(odo_o, n)  # 4540 compositions: round-off grows along the chain
    assert np.abs(To - Too).max() <= 1e-13 * n * max(1.0, np.abs(Too).max())
    assert np.abs(odometry_initialization(pm, n) - To).max() == 0.0  # loop closures in the list are ignored
    with pytest.raises(dpgo_amd.DpgoError):
        odometry_initialization(odo_p.select(np.arange(1, len(odo_p))), n)
    if dpgo_amd.device_count() == 0:
        with pytest.raises(dpgo_amd.DpgoError):
            chordal_initialization(pm, n)


def test_python_mirror_rejects_bad_arguments_before_touching_the_device():
    """The reference aborts on these through glog CHECKs (src/PoseGraph.cpp:19, src/QuadraticProblem.cpp:30-31,
    src/PGOAgent.cpp:838-840); the mirror raises."""
    import dpgo_amd
    from dpgo_amd.trajectory import round_trajectory
    with pytest.raises(ValueError):
        dpgo_amd.PoseGraph(0, 2, 3)  # r < d
    with pytest.raises(ValueError):
        round_trajectory(np.zeros((5, 7)), 5, 3)  # columns not a multiple of d + 1
    with pytest.raises(ValueError):
        round_trajectory(np.zeros((5, 8)), 5, 3, anchor=np.zeros((5, 3)))  # anchor must be r x (d + 1)
    with pytest.raises(KeyError):
        dpgo_amd.ROptParameters(precond="cholesky").to_c()
    pm, n = dpgo_amd.read_g2o_file(os.path.join(DATA, "tinyGrid3D.g2o"))
    pg = dpgo_amd.PoseGraph(0, 5, 2)
    with pytest.raises(ValueError):
        pg.setMeasurements(pm)  # 3-D measurements into a 2-D graph
    with pytest.raises(ValueError):
        dpgo_amd.partition_contiguous(pm, n, n + 1)  # more robots than poses (examples/MultiRobotExample.cpp:74-77)


def test_locality_order_is_a_block_preserving_permutation_that_shortens_the_gathers():
    """dpgo_locality_order (host code of the library, what the agent layer renumbers blocks of >= 40 000 poses with): a
    permutation that maps every chunk (XCD share, boundaries at workgroup tiles) onto itself, deterministic, and on a
    lattice in odometry ("snake") order brings the neighbours of a pose from a plane's worth of rows to a fraction of it;
    relabelling a data set with it leaves the cost of a relabelled iterate unchanged (it is a renaming)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import dpgo_oracle as O
    import dpgo_amd.lib as L
    from dpgo_amd import measurements as M
    lib = L.load()
    om, n, Tt = O.synthetic_grid(20, 20, 10, seed=0)  # 4 000 poses; a plane is 400 rows
    Q = O.construct_Q(n, 3, om)
    rp, ci = L.i32(Q.rowptr), L.i32(Q.colidx)
    nparts, align = 8, 64
    ni = np.zeros(n, dtype=np.int32)
    L.check(lib.dpgo_locality_order(n, L.ptr(rp), L.ptr(ci), nparts, align, L.ptr(ni)))
    ni2 = np.zeros(n, dtype=np.int32)
    L.check(lib.dpgo_locality_order(n, L.ptr(rp), L.ptr(ci), nparts, align, L.ptr(ni2)))
    assert np.array_equal(ni, ni2) and sorted(ni.tolist()) == list(range(n))
    bounds = [(n * k // nparts) // align * align for k in range(nparts)] + [n]
    for c0, c1 in zip(bounds[:-1], bounds[1:]):
        assert sorted(ni[c0:c1].tolist()) == list(range(c0, c1))
    chunk = np.searchsorted(bounds, np.arange(n), side="right") - 1
    rows = np.repeat(np.arange(n), np.diff(Q.rowptr))
    cols = np.asarray(Q.colidx)
    same = chunk[rows] == chunk[cols]
    before = np.abs(rows - cols)[same].max()
    after = np.abs(ni[rows].astype(np.int64) - ni[cols])[same].max()
    assert before >= 399 and after <= before // 3, (before, after)
    assert lib.dpgo_locality_order(n, L.ptr(rp), L.ptr(ci), 0, 64, L.ptr(ni2)) != 0  # bad arguments are refused
    # the Python face: several robots, every robot's block keeps its poses; a renaming does not change the cost
    pm = M.RelativeSEMeasurements(3, om.r1, om.p1, om.r2, om.p2, om.R, om.t, om.kappa, om.tau, om.weight, om.fixed)
    order = M.locality_order(pm, n, 2)
    assert sorted(order[:2000].tolist()) == list(range(2000)) and sorted(order[2000:].tolist()) == list(range(2000, n))
    rm = M.relabel(pm, order)
    om2 = O.Measurements(3, om.r1, rm.p1.astype(np.int64), om.r2, rm.p2.astype(np.int64), om.R, om.t, om.kappa, om.tau,
                         om.weight, om.fixed)
    X = O.lift(O.perturbed_truth(Tt, seed=2), 5)
    X2 = np.empty_like(X)
    X2[order] = X
    f1 = O.QuadraticProblem(Q, None, 5, 3).f(X)
    f2 = O.QuadraticProblem(O.construct_Q(n, 3, om2), None, 5, 3).f(X2)
    assert abs(f1 - f2) <= 1e-12 * abs(f1)



def test_switches_are_one_table_read_once_and_the_auto_rule_is_restated(oracle):
    """Host-only pieces of round 5.  (1) Every DPGO_* switch lives in one table (csrc/host.h, DPGO_OPTIONS):
    dpgo_describe_options lists them with their values, the environment is read once and again on
    dpgo_options_reload.  (2) The constants of DPGO_PRECOND_AUTO's cost rule (dpgo_auto_rule_constants) are the ones
    the oracle's restatement (AutoCostRule) defaults to, and the restatement behaves as documented: block-Jacobi until
    one set-up is paid, one additive solve on trial, hand-back with a doubled wait when it is no cheaper, 15 % of
    hysteresis once accepted, no trial that cannot win."""
    import ctypes as C
    import dpgo_amd.lib as L
    lib = L.load()
    text = L.describe_options()
    rows = dict(ln.split("  # ")[0].split("=", 1) for ln in text.strip().splitlines())
    for name in ("DPGO_SPLIT", "DPGO_SPMM_SYMMETRIC", "DPGO_ITER_GRAPH", "DPGO_ML_OPERATOR_BITS", "DPGO_TILE_WALK",
                 "DPGO_AUTO_COST_RULE", "DPGO_ML_EARLY_STOP", "DPGO_OUTER_SYM", "DPGO_ML_GRAPH", "DPGO_PERSIST"):
        assert name in rows, name
    assert len(rows) >= 30 and all("[set]" not in v for k, v in rows.items() if k not in os.environ)
    old = os.environ.get("DPGO_TCG_AHEAD")
    try:
        os.environ["DPGO_TCG_AHEAD"] = "7"
        assert "DPGO_TCG_AHEAD=7 [set]" not in L.describe_options()  # read once ...
        L.check(lib.dpgo_options_reload())
        assert "DPGO_TCG_AHEAD=7 [set]" in L.describe_options()     # ... until told to look again
    finally:
        if old is None:
            os.environ.pop("DPGO_TCG_AHEAD", None)
        else:
            os.environ["DPGO_TCG_AHEAD"] = old
        lib.dpgo_options_reload()
    k = [C.c_int(0) for _ in range(4)]
    L.check(lib.dpgo_auto_rule_constants(*[C.byref(x) for x in k]))
    uj, ua, setup, minp = (x.value for x in k)
    r = oracle.AutoCostRule()
    assert (r.uj, r.ua, r.setup, r.minp, r.uj0) == (uj, ua, setup, minp, uj) == (10, 13, 2800, 6, 10)
    # 36 products per block-Jacobi solve: the set-up is paid after 8 solves (8 x 360 = 2 880 units)
    seq = []
    for _ in range(8):
        seq.append(r.next())
        r.record(36)
    assert seq == ["jacobi"] * 8 and r.next() == "additive" and r.state == 1 and r.ref == 36
    r.record(28)  # 28 x 13 = 364 >= 36 x 10: no cheaper -> handed back, the next trial waits for two set-ups
    assert r.next() == "jacobi" and r.backoff == 1 and r.units == 0
    for _ in range(16):
        r.record(36)
    assert r.next() == "additive"
    r.record(27)  # 351 < 360: accepted
    assert r.state == 2 and r.next() == "additive"
    r.record(31)  # 403 < 1.15 x 360 = 414: stays (hysteresis)
    assert r.next() == "additive"
    r.record(32)  # 416 >= 414: handed back
    assert r.next() == "jacobi" and r.backoff == 2
    # a handle that shares the device is charged for the part of the chip its launch blocks: a trial that cannot win is skipped
    s = oracle.AutoCostRule(units_jacobi=24, units_additive=65)
    for _ in range(8):
        s.record(36)
    assert s.next() == "additive" and s.switches == 1  # (65 x 6 = 390 < 24 x 36 = 864: the trial can win, so it runs)
    s.record(20)                                       # 65 x 20 = 1 300 >= 864: it did not
    assert s.next() == "jacobi" and s.backoff == 1
    t = oracle.AutoCostRule(units_jacobi=24, units_additive=65)
    for _ in range(19):
        t.record(15)                                   # 19 x 150 = 2 850 units: paid -- but 65 x 6 >= 24 x 15 = 360
    assert t.next() == "jacobi" and t.backoff == 1 and t.switches == 0 and t.units == 0
