#!/bin/bash
timeout 300 python tools/coarse_probe.py "default" 2>&1 | grep -v amdgpu.ids
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -3
