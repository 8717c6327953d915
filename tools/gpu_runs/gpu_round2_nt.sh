#!/bin/bash
# A/B: streamed-once arrays (A P, res1, partial sums of the dense level) non-temporal (new .so) vs not (base .so)
mkdir -p gpurun_out/nt
cp dpgo_amd/libdpgo_hip.so /tmp/new.so
for v in base new base new; do
  if [ $v = base ]; then cp dpgo_amd/libdpgo_hip_base.so dpgo_amd/libdpgo_hip.so; else cp /tmp/new.so dpgo_amd/libdpgo_hip.so; fi
  timeout 300 python bench.py --no-cpu-baseline --no-secondary --precond multilevel 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$v', 'value %.2f ms/step %.3f' % (d['value'], d['ms_per_step']), [round(k['avg_launch_us'],1) for k in r['kernels']], round(r['cycle_tail_us'],1))"
done
cp /tmp/new.so dpgo_amd/libdpgo_hip.so
timeout 900 python -m pytest tests/test_parity_gpu.py -q -m gpu -x -k "multilevel_preconditioner or full_size or persistent" 2>&1 | grep -E "passed|failed"
