"""GPU probe: the kernels of one multilevel tCG iteration at 100k poses (dpgo_bench_iteration_kernels), for tuning builds.
usage: python tools/coarse_probe.py [label]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401
import dpgo_amd  # noqa: E402
from dpgo_amd import synthetic  # noqa: E402

meas, n, Ttrue = synthetic.synthetic_grid(50, 50, 40, seed=0)
pg = dpgo_amd.PoseGraph(0, 5, 3)
pg.setMeasurements(meas)
prob = dpgo_amd.QuadraticProblem(pg)
if os.environ.get("PROBE_BITS"):
    prob.multilevelCoarseBits(int(os.environ["PROBE_BITS"]))
X0 = synthetic.lift_tiles(synthetic.perturbed_truth(Ttrue, seed=2), 5)
Xd = torch.tensor(X0, device="cuda", dtype=torch.float64)
dpgo_amd.QuadraticOptimizer(prob, dpgo_amd.ROptParameters(precond="multilevel")).optimizeDevice(Xd)
lib = dpgo_amd.lib.load()
ms = (C.c_double * 5)()
for rep in range(2):
    dpgo_amd.lib.check(lib.dpgo_bench_iteration_kernels(prob.handle, 200, 20, ms))
print("%-28s bits=%s grid=%s nodes=%s update %.1f restrict %.1f coarse %.1f post %.1f tail %.1f us" % (
    sys.argv[1] if len(sys.argv) > 1 else "", os.environ.get("PROBE_BITS", "64"), os.environ.get("DPGO_COARSE_GRID", "-"),
    os.environ.get("DPGO_COARSE_NODES", "auto"), *(1e3 * v for v in ms)), flush=True)
