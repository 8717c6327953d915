#!/bin/bash
# compile-time / run-time variants of the coarse kernel
mkdir -p gpurun_out/s
run() { for nd in 1 2 4; do DPGO_COARSE_NODES=$nd timeout 300 python tools/coarse_probe.py "$1" 2>&1 | grep -v amdgpu.ids; done; }
run "base u1 w2"
for v in "1 3" "2 2"; do
  set -- $v
  make -C dpgo_amd/csrc -B EXTRA="-DDPGO_COARSE_UNROLL=$1 -DDPGO_COARSE_WAVES=$2" > gpurun_out/s/build_$1_$2.log 2>&1 || { echo "build failed $v"; tail -3 gpurun_out/s/build_$1_$2.log; continue; }
  run "u$1 w$2"
done
