#!/bin/bash
# round-5: what the driver runs at round end -- the whole GPU suite, smoke(), the default bench line
export GPU_OUT=${GPU_OUT:-r5v}
OUT=gpurun_out/$GPU_OUT
mkdir -p $OUT
bash tools/gpu_run.sh tests
grep -E "^FAILED|^ERROR" $OUT/tests.log | head
bash tools/gpu_run.sh smoke
bash tools/gpu_run.sh bench default
