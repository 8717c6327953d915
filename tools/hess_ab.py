"""Rotating (HBM-only) and back-to-back rates of the tCG-step kernel, both storages, for one workload -- run once per
library build (DPGO_HIP_LIBRARY=...) to A/B a kernel change.  Usage: python tools/hess_ab.py [grid100k|grid1m|sphere2500]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
import dpgo_amd  # noqa: E402
from dpgo_amd.agent import build_pose_graphs  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "grid100k"
meas, n, X0, desc = bench.make_workload(name, 5)
ranges, graphs = build_pose_graphs(meas, n, 1, 5)
prob = dpgo_amd.QuadraticProblem(graphs[0])
lib = dpgo_amd.lib.load()
d, r = meas.d, 5
nnzb = len(graphs[0].quadraticMatrix()[1])
hb = bench.hess_bytes(n, nnzb, d, r)
set_b = nnzb * (8 * (d + 1) ** 2 + 4) + 2 * 8 * r * (d + 1) * n
nsets = int(min(512, max(3, -(-3 * 256 * 2 ** 20 // max(set_b, 1)) // 2 + 1)))
hsets = max(3, nsets // 2 + 1)
print("%s  library %s  n %d nnzb %d  hess bytes %.1f MB" % (name, dpgo_amd.lib.LIB_PATH, n, nnzb, hb / 1e6))
for variant in ("plain", "symmetric"):
    if prob.setSpmmVariant(variant) != variant:
        continue
    sb = bench.spmm_bytes(n, nnzb, d, r)
    ms_rot, setb, ms_w = C.c_double(0.0), C.c_double(0.0), C.c_double(0.0)
    dpgo_amd.lib.check(lib.dpgo_bench_spmm_rotating(prob.handle, nsets * (2 if variant == "symmetric" else 1), 200, 10,
                                                    C.byref(ms_rot), C.byref(setb)))
    dpgo_amd.lib.check(lib.dpgo_bench_spmm(prob.handle, 200, 10, C.byref(ms_w)))
    print("  %-9s k_spmm cold %.2f us (%.3f)   warm %.2f us (%.3f)   %s" % (
        variant, ms_rot.value * 1e3, sb / ms_rot.value / 1e6 / 8000.0, ms_w.value * 1e3, sb / ms_w.value / 1e6 / 8000.0,
        prob.tcgKernelInfo()), flush=True)
    for rep in range(2):
        cold, warm = C.c_double(0.0), C.c_double(0.0)
        dpgo_amd.lib.check(lib.dpgo_bench_hess_rotating(prob.handle, hsets + (variant == "symmetric"), 200, 10, C.byref(cold)))
        dpgo_amd.lib.check(lib.dpgo_bench_hess(prob.handle, 200, 10, C.byref(warm)))
        print("  %-9s cold %.2f us (%.3f)   warm %.2f us (%.3f)" % (
            variant, cold.value * 1e3, hb / cold.value / 1e6 / 8000.0, warm.value * 1e3, hb / warm.value / 1e6 / 8000.0), flush=True)
