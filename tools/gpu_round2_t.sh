#!/bin/bash
timeout 300 python tools/coarse_probe.py "fp64 dense" 2>&1 | grep -v amdgpu.ids
timeout 1200 python -m pytest tests/test_parity_gpu.py -q -m gpu -x -k "multilevel or hierarchy or full_size or multi_agent" 2>&1 | tail -3
