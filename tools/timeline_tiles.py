"""Per-tile in-kernel timeline of k_tcg_hess_sym, k_ml_restrict and k_ml_post_ap (diagnostic build -DDPGO_TIMELINE): wave 0 of the first / middle / last
workgroup.  usage: DPGO_LIB=dpgo_amd/libdpgo_tl.so python tools/timeline_tiles.py [workload]   (times in ns from kernel entry
of the FIRST workgroup; 100 MHz clock, 10 ns steps)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dpgo_amd
from dpgo_amd.agent import DeviceAgent, ExchangePlan, build_pose_graphs
import bench
w = sys.argv[1] if len(sys.argv) > 1 else "grid100k"
meas, n, X0, desc = bench.make_workload(w, 5)
ranges, graphs = build_pose_graphs(meas, n, 1, 5)
ag = DeviceAgent(graphs, ExchangePlan(graphs), 0, X0, dpgo_amd.ROptParameters())
for _ in range(4):
    ag.update()
ag.snapshot()
lib = dpgo_amd.lib.load()
lib.dpgo_debug_timeline_tiles.argtypes = [C.c_void_p]
lib.dpgo_debug_timeline_cycle.argtypes = [C.c_void_p]
rows = {"hess": [], "restrict": [], "post": []}
for rep in range(15):
    ag.restore(); ag.update()
    tl = (C.c_longlong * 192)()
    lib.dpgo_debug_timeline_tiles(tl)
    rows["hess"].append(np.array(list(tl), dtype=np.int64).reshape(3, 64))
    tc = (C.c_longlong * 384)()
    lib.dpgo_debug_timeline_cycle(tc)
    c = np.array(list(tc), dtype=np.int64).reshape(2, 3, 64)
    rows["restrict"].append(c[0]); rows["post"].append(c[1])
NAMES = {"hess": ["top", "LDS staged", "gather done", "own requested", "projected", "stored+next requested"],
         "restrict": ["top", "gather done", "residual staged", "P^T res staged", "run sums written", "-"],
         "post": ["top", "gather done", "own rows staged", "smoothed+prolonged", "staged again", "projected+stored"]}
for kern in ("hess", "restrict", "post"):
    T = np.stack(rows[kern])                       # [rep][wg][slot]
    base = T[:, 0:1, 0:1]
    rel = np.where(T > 0, (T - base) * 10, -1)
    med = np.median(rel, axis=0).astype(int)
    print("==", {"hess": "k_tcg_hess_sym", "restrict": "k_ml_restrict", "post": "k_ml_post_ap"}[kern])
    for wg, nm in enumerate(["first", "middle", "last"]):
        print("workgroup %-6s entry %d  prologue done %d  end %d" % (nm, med[wg, 0], med[wg, 1], med[wg, 2]))
        for t in range(10):
            v = med[wg, 4 + 6 * t: 10 + 6 * t]
            if (v <= 0).all():
                break
            print("   tile %d: " % t + "  ".join("%s %d" % (a, b) for a, b in zip(NAMES[kern], v) if a != "-"))
