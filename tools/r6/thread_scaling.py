"""How the host threads of the hierarchy's symbolic set-up scale on the GPU box: the exported growth / merge (no GPU work)
on the 100k-pose grid's pattern for DPGO_SETUP_THREADS = 1, 2, 4, 8 and range counts 1 / 8; cores the process may use."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dpgo_amd, dpgo_amd.lib as L
from dpgo_amd import synthetic
lib = L.load()
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
    if os.path.exists(f):
        print(f, open(f).read().strip())
meas, n, _ = synthetic.synthetic_grid(50, 50, 40, seed=0)
pg = dpgo_amd.PoseGraph(0, 5, 3)
pg.setMeasurements(meas)
rowptr, colidx, _ = pg.quadraticMatrix()
rp, ci = L.i32(rowptr), L.i32(colidx)
lab, par, na = np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32), C.c_int(0)
for chunks in (1, 8):
    for threads in (1, 2, 4, 8):
        os.environ["DPGO_ML_GROWTH_CHUNKS"], os.environ["DPGO_SETUP_THREADS"] = str(chunks), str(threads)
        L.check(lib.dpgo_options_reload())
        best = [1e9, 1e9]
        for _ in range(5):
            t0 = time.perf_counter()
            L.check(lib.dpgo_multilevel_graph_aggregates(n, L.ptr(rp), L.ptr(ci), 182, L.ptr(lab), L.ptr(par), C.byref(na)))
            t1 = time.perf_counter()
            L.check(lib.dpgo_multilevel_merged_aggregates(n, L.ptr(rp), L.ptr(ci), 182, 273, L.ptr(lab), L.ptr(par), C.byref(na)))
            t2 = time.perf_counter()
            best = [min(best[0], t1 - t0), min(best[1], t2 - t1)]
        print("ranges %d threads %d: growth %.2f ms, growth + merge %.2f ms, %d aggregates" % (
            chunks, threads, 1e3 * best[0], 1e3 * best[1], na.value), flush=True)
