#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_round.sh <tag>      e.g. r01_v6
# Produces gpurun_out/<tag>_bench_kernel_stats.csv, gpurun_out/<tag>_pmc_fetch_write.json, gpurun_out/<tag>_bench.json
# (copy them into profiles/).  Counters are collected in their own passes (no tracing flags next to --pmc).
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o bench -- \
  python $R/bench.py --no-cpu-baseline --no-secondary > $R/gpurun_out/prof_$TAG.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_${TAG}_$c -o b -- \
    python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-secondary --spmm-reps 5 > $R/gpurun_out/pmc_${TAG}_$c.log 2>&1
done
cd $R
python tools/summarize_prof.py stats gpurun_out/prof_$TAG gpurun_out/${TAG}_bench_kernel_stats.csv
python tools/summarize_prof.py pmc gpurun_out/${TAG}_pmc_fetch_write.json \
  FETCH_SIZE=gpurun_out/pmc_${TAG}_FETCH_SIZE WRITE_SIZE=gpurun_out/pmc_${TAG}_WRITE_SIZE
cp gpurun_out/${TAG}_pmc_fetch_write.json profiles/  # bench.py reads roofline.traffic from the newest committed summary
python bench.py 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench.json   # the driver's line (<= 6 KB)
cp bench_detail.json gpurun_out/${TAG}_bench_detail.json                 # everything measured (the DETAIL line)
rm -rf gpurun_out/prof_$TAG gpurun_out/pmc_${TAG}_FETCH_SIZE gpurun_out/pmc_${TAG}_WRITE_SIZE
head -12 gpurun_out/${TAG}_bench_kernel_stats.csv | cut -c1-150
python -c "
import json; j=json.load(open('gpurun_out/${TAG}_pmc_fetch_write.json'))
for k in j:
    for kk,v in j[k].items():
        if 'tcg' in kk or 'spmm' in kk: print(k, kk, v)
print(open('gpurun_out/${TAG}_bench.json').read()[:300])"
