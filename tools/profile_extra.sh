#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_extra.sh <tag>
# rocprofv3 kernel-stats summaries the bench profile cannot give by itself: (1) the rotating-operand (HBM-only) launches
# alone, (2) the small-block regime (sphere2500, 12.5k-pose slab, 625-pose block) with the multilevel, block-Jacobi and
# persistent paths.  Writes gpurun_out/<tag>_rotating_kernel_stats.csv and gpurun_out/<tag>_small_blocks_kernel_stats.csv.
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_rot_$TAG -o rot -- \
  python $R/tools/rotating_probe.py > $R/gpurun_out/${TAG}_rotating.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_small_$TAG -o small -- \
  python $R/tools/ml_probe.py grid625 sphere grid6250 slab > $R/gpurun_out/${TAG}_small_blocks.log 2>&1
cd $R
python tools/summarize_prof.py stats gpurun_out/prof_rot_$TAG gpurun_out/${TAG}_rotating_kernel_stats.csv
python tools/summarize_prof.py stats gpurun_out/prof_small_$TAG gpurun_out/${TAG}_small_blocks_kernel_stats.csv
rm -rf gpurun_out/prof_rot_$TAG gpurun_out/prof_small_$TAG
tail -2 gpurun_out/${TAG}_rotating.log
head -8 gpurun_out/${TAG}_rotating_kernel_stats.csv | cut -c1-140
grep -v "^dpgo_hip" gpurun_out/${TAG}_small_blocks.log | tail -12
# (3) the 16-agent loop-back sweep (what a rank of the 8-GPU configuration runs, eight times over): kernel stats of the
# one-launch solves and the batched exchange
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_loop_$TAG -o loop -- \
  python $R/bench.py --loopback --agents-per-gpu 16 --no-cpu-baseline --no-secondary --steps 10 > $R/gpurun_out/${TAG}_loopback16.log 2>&1
cd $R
python tools/summarize_prof.py stats gpurun_out/prof_loop_$TAG gpurun_out/${TAG}_loopback16_kernel_stats.csv
rm -rf gpurun_out/prof_loop_$TAG
grep '^{' gpurun_out/${TAG}_loopback16.log | tail -1 > gpurun_out/${TAG}_loopback16_bench.json
head -6 gpurun_out/${TAG}_loopback16_kernel_stats.csv | cut -c1-140
# (4) BASELINE configs[4]: kitti_00 GNC (4 agents, reference schedule) -- kernel stats of the weight updates (k_edge_weights,
# k_rebuild_Q, hierarchy values) and of the one-launch solves between them
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_kitti_$TAG -o kitti -- \
  python $R/tools/r6/kitti_gnc_probe.py > $R/gpurun_out/${TAG}_kitti_gnc.log 2>&1
cd $R
python tools/summarize_prof.py stats gpurun_out/prof_kitti_$TAG gpurun_out/${TAG}_kitti_gnc_kernel_stats.csv
rm -rf gpurun_out/prof_kitti_$TAG
grep '^{' gpurun_out/${TAG}_kitti_gnc.log | tail -1 > gpurun_out/${TAG}_kitti_gnc.json
head -8 gpurun_out/${TAG}_kitti_gnc_kernel_stats.csv | cut -c1-140
