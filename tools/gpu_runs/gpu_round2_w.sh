#!/bin/bash
bash tools/profile_round.sh r02_v5 > gpurun_out/r02_v5_profile_round.log 2>&1
tail -25 gpurun_out/r02_v5_profile_round.log | cut -c1-400
bash tools/profile_extra.sh r02_v5 > gpurun_out/r02_v5_profile_extra.log 2>&1
tail -22 gpurun_out/r02_v5_profile_extra.log | cut -c1-300
