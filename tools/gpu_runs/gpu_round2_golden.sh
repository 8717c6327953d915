#!/bin/bash
timeout 600 python -m pytest tests/test_parity_gpu.py -q -m gpu -x -k "committed_golden" 2>&1 | grep -E "passed|failed|^E " | head
