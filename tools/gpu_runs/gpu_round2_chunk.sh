#!/bin/bash
for c in 1 2 3; do DPGO_DENSE_CHUNK=$c timeout 300 python tools/coarse_probe.py "chunk=$c" 2>&1 | grep -v amdgpu.ids; done
