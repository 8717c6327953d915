"""Products to tolerance and fixed-work step of the 100k grid for several range counts of the aggregates' growth
(DPGO_ML_GROWTH_CHUNKS): what the parallel rule costs in preconditioner quality.  usage: python tools/r6/chunks_products.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import dpgo_amd
import bench
lib = dpgo_amd.lib.load()
for chunks in (sys.argv[1:] or ["1", "2", "4", "8", "16"]):
    os.environ["DPGO_ML_GROWTH_CHUNKS"] = chunks
    dpgo_amd.lib.check(lib.dpgo_options_reload())
    tt = bench.time_to_tolerance("grid100k", 5, "multilevel")
    step = bench.secondary_single_agent("grid100k", 5, "multilevel", 10, 2, 5)
    print("ranges %2s: to tolerance %d products %.2f ms, set-up %.2f ms (values %.2f) | step %.1f products %.3f ms %.1f it/s" % (
        chunks, tt["products"], tt["ms"], tt["hierarchy_setup_ms"], tt["hierarchy_values_only_ms"],
        step["tcg_iterations_per_step"], step["ms_per_step"], step["it_per_s"]), flush=True)
