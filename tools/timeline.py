"""In-kernel timeline of the two tCG kernels (diagnostic build libdpgo_tl.so, -DDPGO_TIMELINE): where does a
latency-bound launch spend its time?  usage: DPGO_LIB=dpgo_amd/libdpgo_tl.so python tools/timeline.py [workload]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dpgo_amd
from dpgo_amd.agent import DeviceAgent, ExchangePlan, build_pose_graphs
import bench
w = sys.argv[1] if len(sys.argv) > 1 else "sphere2500"
meas, n, X0, desc = bench.make_workload(w, 5)
ranges, graphs = build_pose_graphs(meas, n, 1, 5)
ag = DeviceAgent(graphs, ExchangePlan(graphs), 0, X0, dpgo_amd.ROptParameters())
ag.update(); ag.snapshot()
lib = dpgo_amd.lib.load()
lib.dpgo_debug_timeline.argtypes = [C.c_void_p]
names = {0: ["entry", "first tile issued+rowptr/colidx", "prologue done", "LDS staged", "gather done", "tile done", "partials stored"],
         1: ["entry", "-", "prologue done", "-", "-", "tile done", "partials stored"]}
acc = {}
for rep in range(20):
    ag.restore(); ag.update()
    tl = (C.c_longlong * 32)()
    lib.dpgo_debug_timeline(tl)
    t = np.array(list(tl)).reshape(2, 16)
    for k in (0, 1):
        acc.setdefault(k, []).append(t[k, :7] - t[k, 0])
    acc.setdefault("gap", []).append(t[1, 0] - t[0, 6])   # hess end -> update entry (if update ran after hess)
for k in (0, 1):
    m = np.median(np.array(acc[k]), axis=0) * 10.0  # 100 MHz ticks -> ns
    print("kernel", "hess" if k == 0 else "update", {nm: int(v) for nm, v in zip(names[k], m) if nm != "-"})
print("median (update entry - hess last stamp) ns:", int(np.median(acc["gap"]) * 10))
