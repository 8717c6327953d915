#!/bin/bash
# end-of-round verification: full GPU suite, smoke, profile summaries (tag r02_v6)
mkdir -p gpurun_out/final
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/final/tests.log 2>&1; grep -E "passed|failed|error" gpurun_out/final/tests.log | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -1
bash tools/profile_round.sh r02_v6 > gpurun_out/final/profile_round.log 2>&1
bash tools/profile_extra.sh r02_v6 > gpurun_out/final/profile_extra.log 2>&1
tail -3 gpurun_out/r02_v6_rotating.log | grep -v rocprof
