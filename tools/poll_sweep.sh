#!/bin/bash
# GPU box: first-wait sweep of the in-kernel all-reduce (DPGO_POLL_FIRST, units of 64 clocks) for one preconditioner / workload.
# usage: bash tools/poll_sweep.sh additive "slab grid6250" "16 24 32 44 60"
pc=$1; wl=$2; vals=$3
for v in $vals; do
  echo "== DPGO_POLL_FIRST=$v"
  DPGO_POLL_FIRST=$v PROBE_ONLY=$pc DPGO_PERSIST_VERBOSE=1 timeout 600 python tools/ml_probe.py $wl 2>&1 | grep -v amdgpu | \
    awk '/persistent tCG/ {last=$0} /products/ {print; print "   " last}' | cut -c1-330
done
