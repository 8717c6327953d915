"""bench.py / __graft_entry__ contract checks."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"}


def test_algorithmic_byte_formulas_match_the_survey():
    """bytes = nnzb (8 b^2 + 4) + 4 (n + 1) + 16 r b n; 100k grid: 122.7 MB ("123.1" with MB = 1e6)."""
    sys.path.insert(0, ROOT)
    import bench
    assert bench.spmm_bytes(100000, 687000, 3, 5) == 687000 * 132 + 4 * 100001 + 320 * 100000 == 123084004
    assert bench.spmm_bytes(2500, 12398, 3, 5) == 2446540  # sphere2500: 2.44 MB
    assert bench.hess_bytes(100000, 687000, 3, 5) == 687000 * 132 + 4 * 100001 + 100000 * (6 * 160 + 72)
    assert bench.HBM_PEAK_GBS == 8000.0


def test_entry_points_exist():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    assert callable(g.build) and callable(g.smoke)


@pytest.mark.gpu
def test_bench_prints_one_valid_json_line():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "grid:12x10x6", "--steps", "2",
                        "--warmup", "1", "--settle", "2", "--cpu-budget-s", "3", "--spmm-reps", "20"],
                       capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert REQUIRED <= set(j)
    assert j["metric"] == "rbcd_iterations_per_sec" and j["unit"] == "it/s" and j["higher_is_better"] is True
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["warmup"] == 1 and j["dtype"] == "f64" and j["vs_baseline"] is None
    assert j["value"] > 0 and abs(j["value"] - 1e3 / j["ms_per_step"]) < 1e-6 * j["value"]
    assert "workload" in j["config"] and "model" not in j["config"]
    rf = j["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and rf["achieved"] > 0
    assert rf["spmm_only"]["frac"] > 0
    # the headline describes the kernel the timed loop launched: name, time and rate belong together.  A 720-pose block is
    # in the latency regime: its solve is ONE launch of the persistent kernel, and the tCG-step kernel of the multi-launch
    # scheme (frac = rotating / HBM-only, warm beside it) is reported next to it
    assert rf["spmm_storage_selected"] == "plain" and rf["kernel"].startswith("k_rtr_persist<3,5,")
    assert abs(rf["achieved"] - rf["bytes_per_launch"] / rf["avg_launch_us"] / 1e3) < 1e-9 * rf["achieved"]
    assert rf["products_per_launch"] > 0
    assert abs(rf["us_per_product"] - rf["avg_launch_us"] / rf["products_per_launch"]) < 1e-9 * rf["us_per_product"]
    ml = rf["multi_launch_kernel"]
    assert ml["kernel"].startswith("k_tcg_hess_span<3,5,4,0>") and ml["warm"]["frac"] > 0
    assert abs(ml["achieved"] - ml["bytes_per_launch"] / ml["avg_launch_us"] / 1e3) < 1e-9 * ml["achieved"]
    assert abs(ml["warm"]["achieved"] - ml["bytes_per_launch"] / ml["warm"]["avg_launch_us"] / 1e3) < 1e-9 * ml["achieved"]
    # (one-launch solve: frac = MODELLED HBM bytes of the launch over its time -- Q and the vectors live in registers / LDS --;
    # the figure comparable with the multi-launch kernel is effective_algorithmic_*: products x the step's bytes)
    assert abs(rf["effective_algorithmic_bytes"] - rf["products_per_launch"] * ml["bytes_per_launch"]) < 1e-6 * rf["bytes_per_launch"] * 1e3
    assert abs(rf["effective_algorithmic_GBs"] - rf["effective_algorithmic_bytes"] / rf["avg_launch_us"] / 1e3) < 1e-9 * rf["effective_algorithmic_GBs"]
    assert rf["bytes_per_launch"] < ml["bytes_per_launch"] and rf["traffic"] is None
    # the TIMED kernel's own bytes are always printed, whichever storage it reads
    assert rf["stored_bytes_per_launch"] > 0 and rf["frac_own_bytes"] > 0 and "storage" in rf
    assert abs(rf["frac_own_bytes"] - rf["stored_bytes_per_launch"] / rf["avg_launch_us"] / 1e3 / 8000.0) < 1e-9
    assert rf["symmetric_storage"] is None  # (this small block cannot run the symmetric kernels: 4 lane groups per pose)
    assert j["products_per_step"] > 0 and j["time_to_tolerance_ms"] > 0 and j["products_to_tolerance"] > 0
    # ... and repeated inside `config`, which the driver's parsed record keeps
    assert j["config"]["time_to_tolerance_ms"] == j["time_to_tolerance_ms"]
    assert "time_to_tolerance_ms = %.3f" % j["time_to_tolerance_ms"] in j["config"]["local_solver"]
    assert j["config"]["products_per_step"] == j["products_per_step"]
    assert [k["kernel"].split()[0].rstrip(",") for k in rf["kernels"]] == ["k_tcg_update", "k_ml_restrict",
                                                                           "k_ml_coarse_prolong", "k_ml_post"]
    assert all(k["avg_launch_us"] > 0 for k in rf["kernels"])
    # the once-per-Q cost of the preconditioner beside the time to tolerance (the reference factors inside its first solve)
    assert j["hierarchy_setup_ms"] > 0 and j["hierarchy_values_only_ms"] > 0
    assert abs(j["time_to_tolerance_incl_setup_ms"] - (j["time_to_tolerance_ms"] + j["hierarchy_setup_ms"])) < 1e-9
    assert j["config"]["hierarchy_setup_ms"] == j["hierarchy_setup_ms"]
    assert j["config"]["time_to_tolerance_incl_setup_ms"] == j["time_to_tolerance_incl_setup_ms"]
    assert rf["traffic_live"] is False
    cb = j["cpu_baseline"]  # reference configuration (exact factor, one core per agent) + the 1-core port beside it
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == "it/s"
    # cpu_baseline.value is the pair from the SETTLED iterate every timed step restores ...
    assert "exact sparse factor" in cb["sample"] and "SETTLED" in cb["sample"] and cb["factorisation_seconds"] > 0
    same = cb["gpu_same_work"]  # the like-for-like pair: same 8 blocks, same iterate, same one sweep, on the GPU
    assert same["agents"] == 8 and same["value"] > 0 and same["gradnorm_after"] > 0 and cb["gradnorm_after"] > 0
    assert abs(cb["gpu_over_cpu_same_work"] - same["value"] / cb["value"]) < 1e-9 * cb["gpu_over_cpu_same_work"]
    assert len(same["tcg_iterations_per_agent"]) == 8 and len(cb["tcg_iterations_per_agent"]) == 8
    hs = same["hierarchy_setup_ms"]  # the GPU side's once-per-Q cost, per block
    assert len(hs["first_ms"]) == 8 and min(hs["first_ms"]) > 0 and min(hs["values_only_ms"]) > 0
    assert len(same["auto_rule"]) == 8 and all(a["state"] in ("jacobi", "trial", "additive") for a in same["auto_rule"])
    ini = cb["initial_iterate"]  # ... the pair from the benchmark's initial iterate under its own key
    assert ini["value"] > 0 and "SETTLED" not in ini["sample"] and ini["gpu_same_work"]["agents"] == 8
    assert len(ini["tcg_iterations_per_agent"]) == 8 and len(ini["gpu_same_work"]["tcg_iterations_per_agent"]) == 8
    port = cb["single_agent_port"]
    assert port["cores"] == 1 and port["value"] > 0 and port["rel_diff_fOpt_vs_device"] < 1e-6
    tt = j["quality"]["to_tolerance"]
    assert set(tt) == {"grid:12x10x6/auto", "grid:12x10x6/multilevel", "grid:12x10x6/additive", "grid:12x10x6/jacobi",
                       "grid:12x10x6/multilevel+fp32_dense_level",   # opt-in storage mode, beside the headline
                       "grid:12x10x6/multilevel+fp64_cycle_operators"}  # (the default's fp32 operator copies switched off)
    assert rf["multilevel"]["coarse_inverse_bits"] == 64  # the dense level of the headline configuration is fp64
    # the cycle's operator copies: fp32 only where the symmetric storage runs (blocks beyond the Infinity Cache), named in config
    assert rf["multilevel"]["cycle_operator_copy_bits"] == 64 and "fp64 copies" in j["config"]["cycle_storage"]
    assert "dense level (inverse, restricted residual) in fp64" in j["config"]["cycle_storage"]
    assert all("products" in v for v in tt.values())


@pytest.mark.gpu
def test_bench_secondary_workload_reports_per_product_cost_and_in_kernel_split():
    """`also.sphere2500` of the headline line (BASELINE's metric is quoted on sphere2500 as well): the fixed-work step rate,
    microseconds per Hessian-vector product and, for a one-launch solve, the in-kernel phase split of participant 0."""
    sys.path.insert(0, ROOT)
    import bench
    e = bench.secondary_single_agent("sphere2500", 5, "auto", 3, 1, 2)
    assert e["it_per_s"] > 0 and e["tcg_iterations_per_step"] > 0 and e["precond_used"] in ("additive", "multilevel", "jacobi")
    assert abs(e["us_per_product"] - 1e3 * e["ms_per_step"] / e["tcg_iterations_per_step"]) < 1e-6 * e["us_per_product"]
    ph = e["in_kernel_us_per_iteration"]  # sphere2500 is a one-launch solve
    assert ph is not None and ph["hessian_phase"] > 0 and ph["total"] < e["us_per_product"] * 1.5
    assert abs(ph["total"] - (ph["hessian_phase"] + ph["all_reduce_after_hessian"] + ph["update_phase"]
                              + ph["reductions_after_update"])) < 1e-9


@pytest.mark.gpu
def test_bench_loopback_runs_the_multi_agent_path_through_rccl():
    """bench.py --loopback: 4 agents on one GPU, every exchange and reduction through the library-owned 1-rank RCCL
    communicator (the N > 1 data path), same-colour agents solved concurrently, exchange time from device events."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "grid:12x10x8", "--steps", "2",
                        "--warmup", "1", "--settle", "1", "--loopback", "--agents-per-gpu", "4", "--no-cpu-baseline",
                        "--no-secondary", "--spmm-reps", "10"],
                       capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    j = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert j["config"]["agents"] == 4 and "RCCL" in j["config"]["schedule"] and "loop-back" in j["config"]["schedule"]
    assert j["quality"]["exchange_ms_per_step_rank0"] > 0 and j["value"] > 0
    assert j["quality"]["cost_2f_after_step"] < j["quality"]["cost_2f_trajectory"][0]
    # coupled blocks under `auto`: the selection is brought to its steady state before the timed steps
    assert j["config"]["selection_sweeps_before_timing"] >= 0
    assert set(j["config"]["precond_used_in_timed_steps"]) <= {"jacobi", "additive", "multilevel"}


def test_bench_default_steps_time_more_than_a_second():
    """The default --steps keeps the timed region above one second at the headline workload's ~5 ms per step (the
    driver's utilisation sampling needs that long to see the run)."""
    sys.path.insert(0, ROOT)
    import bench
    old = sys.argv
    try:
        sys.argv = ["bench.py"]
        a = bench.parse_args()
    finally:
        sys.argv = old
    assert a.steps >= 200 and a.gpus == 1 and a.warmup >= 1


@pytest.mark.gpu
def test_bench_multi_gpu_branch_runs_at_world_size_one():
    """The N > 1 branch of bench.py executed on ONE device exactly as the driver launches it for N > 1 (python -m
    torch.distributed.run ... bench.py --gpus 1), with --dist: init_process_group("nccl") = RCCL, the library's communicator
    taken from the process group (DeviceComm.from_torch_distributed), dist.barrier around the timed region, the
    all-reduces of work and time, 2 agents per GPU, every exchange a grouped RCCL self send / recv timed by device events."""
    port = 29500 + (os.getpid() % 400)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--dist", "--workload",
           "grid:12x10x8", "--steps", "3", "--warmup", "1", "--settle", "1", "--no-cpu-baseline", "--no-secondary",
           "--spmm-reps", "10"]
    p = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    j = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert j["config"]["dist_backend"] == "nccl" and j["n_gpus"] == 1 and j["config"]["agents"] == 2
    assert j["config"]["agents_per_gpu"] == 2 and "RCCL" in j["config"]["schedule"]
    assert "torch.distributed process group" in j["config"]["schedule"]
    assert j["quality"]["exchange_ms_per_step_rank0"] > 0 and j["value"] > 0 and j["steps"] == 3
    assert j["quality"]["cost_2f_after_step"] < j["quality"]["cost_2f_trajectory"][0]


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu_through_the_peer_store():
    """bench.py with TWO ranks (torch.distributed.run --nproc-per-node 2, both on device 0) and --transport ipc: gloo process
    group, 2 agents per rank, the public poses written by the senders' pack kernels straight into the receivers'
    hipIpc-mapped neighbour buffers (dpgo_amd/ipc.py), barrier + max-over-ranks timing, rank 0 prints the line."""
    port = 29500 + ((os.getpid() + 211) % 400)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--transport", "ipc",
           "--workload", "grid:12x10x8", "--steps", "3", "--warmup", "1", "--settle", "1", "--no-cpu-baseline",
           "--no-secondary", "--spmm-reps", "10"]
    p = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, timeout=900,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1  # rank 0 only
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["agents"] == 4 and j["config"]["agents_per_gpu"] == 2
    assert j["config"]["dist_backend"] == "gloo" and j["config"]["transport"] == "ipc" and "peer store" in j["config"]["schedule"]
    assert j["quality"]["exchange_ms_per_step_rank0"] > 0 and j["value"] > 0 and j["scaling"] == "strong"
    assert j["quality"]["cost_2f_after_step"] < j["quality"]["cost_2f_trajectory"][0]
