#!/bin/bash
for bits in 64 32; do for g in 1024 768 400 270 200; do
PROBE_BITS=$bits DPGO_COARSE_GRID=$g timeout 300 python tools/coarse_probe.py "grid cap" 2>&1 | grep -v amdgpu.ids
done; done
