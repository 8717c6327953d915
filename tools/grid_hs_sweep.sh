#!/bin/bash
# GPU box: how does the tCG-step kernel on the symmetric storage depend on its (persistent) grid?  100k poses = 1 563 tiles;
# 768 resident workgroups walk 2.03 rounds, i.e. a third round of 27 tiles.  usage: bash tools/grid_hs_sweep.sh "520 768 784 1024 1563"
for g in $1; do
  echo "== DPGO_GRID_HESS_SYM=$g"
  DPGO_GRID_HESS_SYM=$g timeout 300 python tools/hess_ab.py grid100k 2>&1 | grep -v amdgpu | grep "symmetric" | cut -c1-200
done
