mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -k "chordal or cxx_shim or end_to_end or example_scripts or robust_pgo" > gpurun_out/t5_new.log 2>&1
tail -12 gpurun_out/t5_new.log
