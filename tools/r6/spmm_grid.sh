for g in 0 782 1024 1563; do export DPGO_GRID_SPMM_SYM=$g; echo "GRID=$g"; bash tools/r6/spmm_ab.sh dpgo_amd/libdpgo_hip_s1b4.so dpgo_amd/libdpgo_hip_s1b3.so 2>&1 | grep -E "^rep 1|^rep 2|rror"; done
