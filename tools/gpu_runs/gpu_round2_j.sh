mkdir -p gpurun_out
for pc in multilevel jacobi; do
  timeout 600 python bench.py --agents-per-gpu 16 --precond $pc --no-cpu-baseline --no-secondary --steps 5 --warmup 1 > gpurun_out/t10_a16_$pc.log 2>&1
  python - <<PY
import json
j=json.loads([l for l in open('gpurun_out/t10_a16_$pc.log') if l.startswith('{')][-1])
q=j['quality']
print('$pc', 'ms/sweep', round(j['ms_per_step'],2), 'tcg/step', q['tcg_iterations_per_step_rank0'], 'gradnorm traj', [round(g,2) for g in q['gradnorm_trajectory']], 'after', round(q['gradnorm_after_step'],3), 'cost', q['cost_2f_after_step'])
PY
done
timeout 600 python tools/ml_probe.py grid100k sphere > gpurun_out/t10_probe.log 2>&1
grep -v "^dpgo_hip" gpurun_out/t10_probe.log | tail -6
