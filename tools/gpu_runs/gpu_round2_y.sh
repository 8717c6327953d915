#!/bin/bash
mkdir -p gpurun_out/y
timeout 2400 python -m pytest tests -q -m gpu -x > gpurun_out/y/tests.log 2>&1; tail -4 gpurun_out/y/tests.log
timeout 600 python bench.py --no-cpu-baseline --no-secondary --precond multilevel > gpurun_out/y/bench_ml.json 2> gpurun_out/y/bench_ml.err
python -c "
import json; d=json.loads(open('gpurun_out/y/bench_ml.json').read().strip().splitlines()[-1]); print('multilevel', d['value'], d['ms_per_step'], d['quality']['tcg_iterations_per_step_rank0'])"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
