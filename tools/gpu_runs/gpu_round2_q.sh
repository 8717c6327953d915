#!/bin/bash
mkdir -p gpurun_out/q
timeout 2400 python -m pytest tests -q -m gpu -x > gpurun_out/q/tests.log 2>&1; tail -6 gpurun_out/q/tests.log
