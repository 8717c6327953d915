#!/bin/bash
timeout 900 python -m pytest tests/test_parity_gpu.py -q -m gpu -x -k "dense_spd or multilevel_preconditioner or hierarchy" 2>&1 | tail -4
PROBE_SETUP=1 timeout 600 python tools/ml_probe.py sphere slab grid100k 2>&1 | grep -E "setup|multilevel " | cut -c1-200
