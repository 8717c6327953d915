#!/bin/bash
timeout 1200 python -m pytest tests/test_parity_gpu.py -q -m gpu -x -k "symmetric_storage" 2>&1 | tail -3
