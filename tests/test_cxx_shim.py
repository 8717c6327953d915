"""The C++17 mirror of the reference classes (include/dpgo_hip.hpp) compiles with the host compiler,
links against libdpgo_hip.so, and re-runs the reference's known-answer tests through it."""
import os
import subprocess

import pytest

from conftest import ROOT


def _build(tmp_path):
    exe = os.path.join(str(tmp_path), "test_shim")
    libdir = os.path.join(ROOT, "dpgo_amd")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cxx", "test_shim.cpp"), "-L", libdir, "-ldpgo_hip",
           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    subprocess.check_call(cmd)
    return exe


def test_cxx_shim_compiles_and_refuses_without_device(tmp_path):
    import dpgo_amd
    exe = _build(tmp_path)
    if dpgo_amd.device_count() > 0:
        pytest.skip("a GPU is present (covered by the gpu test)")
    p = subprocess.run([exe], capture_output=True, text=True)
    assert p.returncode == 77, p.stdout + p.stderr  # DPGO_ERR_HIP, no CPU fallback
    assert "inactive neighbours: ok" in p.stdout  # (host-only part: PoseGraph::setNeighborActive)


def _write_greedy_scenario(path):
    """smallGrid3D / 5 robots: measurements, partition, chordal initial iterate and what the Python driver
    (RBCDCluster.run_greedy: greedy selection + Nesterov acceleration, restart 30) does with them on this GPU."""
    import numpy as np
    import dpgo_amd
    from dpgo_amd.agent import DeviceAgent, ExchangePlan, RBCDCluster, build_pose_graphs
    from dpgo_amd.initialization import chordal_initialization
    from dpgo_amd import synthetic
    r, robots, restart = 5, 5, 30
    meas, n = dpgo_amd.read_g2o_file(os.path.join(ROOT, "data", "smallGrid3D.g2o"))
    d = meas.d
    X0 = synthetic.lift_tiles(chordal_initialization(meas, n), r)
    ranges, graphs = build_pose_graphs(meas, n, robots, r)
    plan = ExchangePlan(graphs)
    agents = {a: DeviceAgent(graphs, plan, a, X0[ranges[a][0]:ranges[a][1]], dpgo_amd.ROptParameters())
              for a in range(robots)}
    for ag in agents.values():
        ag.enable_acceleration(robots, restart)
    out = RBCDCluster(plan, agents).run_greedy()
    with open(path, "w") as fh:
        # the demo's relabelling (examples/MultiRobotExample.cpp:71-119), edges in the dataset's own order
        per = n // robots
        robot_of = np.minimum(np.arange(n) // per, robots - 1)
        local = np.arange(n) - robot_of * per
        fh.write("%d %d %d %d %d %d\n" % (d, r, robots, n, len(meas), restart))
        for e in range(len(meas)):
            fh.write("%d %d %d %d %.17g %.17g %d " % (robot_of[meas.p1[e]], local[meas.p1[e]], robot_of[meas.p2[e]],
                                                       local[meas.p2[e]], meas.kappa[e], meas.tau[e],
                                                       int(meas.fixedWeight[e])))
            fh.write(" ".join("%.17g" % v for v in np.asarray(meas.R[e]).reshape(-1)) + " ")
            fh.write(" ".join("%.17g" % v for v in np.asarray(meas.t[e]).reshape(-1)) + "\n")
        for a in range(robots):
            fh.write("%d %d\n" % ranges[a])
        fh.write(" ".join("%.17g" % v for v in np.ascontiguousarray(X0).reshape(-1)) + "\n")  # tiles = column-major matrix
        fh.write("%d\n" % out["iterations"] + " ".join(str(v) for v in out["selected"]) + "\n")
        fh.write("%.17g %.17g\n" % (out["cost"], out["gradnorm"]))
    return out


@pytest.mark.gpu
def test_cxx_shim_reference_known_answers(tmp_path):
    exe = _build(tmp_path)
    scenario = os.path.join(str(tmp_path), "greedy.txt")
    want = _write_greedy_scenario(scenario)
    p = subprocess.run([exe, scenario], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "lifted variable: ok" in p.stdout and "inactive neighbours: ok" in p.stdout
    assert "greedy: %d iterations" % want["iterations"] in p.stdout  # same selection sequence as the Python driver
    assert "triangle" in p.stdout and "prior" in p.stdout and "project: ok" in p.stdout
    assert "rounding" in p.stdout and "robust: inlier" in p.stdout and "robust: outlier" in p.stdout
    assert "precond 1:" in p.stdout and "precond 2:" in p.stdout  # block-Jacobi and multilevel
    assert "begin/end: ok" in p.stdout  # the solve in two halves + the additive plan through the mirror
