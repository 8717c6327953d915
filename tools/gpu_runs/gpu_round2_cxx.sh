#!/bin/bash
timeout 600 python -m pytest tests/test_cxx_shim.py -q -m gpu -x 2>&1 | grep -E "passed|failed|^E " | head
