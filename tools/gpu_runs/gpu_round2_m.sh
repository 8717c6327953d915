mkdir -p gpurun_out
( time timeout 2400 python -m pytest tests -m gpu -q --timeout 1200 -x ) > gpurun_out/t13_all.log 2>&1
tail -5 gpurun_out/t13_all.log
timeout 600 python bench.py --agents-per-gpu 16 --no-cpu-baseline --no-secondary --steps 5 --warmup 2 > gpurun_out/t13_a16.log 2>&1
python - <<PY
import json
j=json.loads([l for l in open('gpurun_out/t13_a16.log') if l.startswith('{')][-1])
print('16 agents ms/sweep', round(j['ms_per_step'],2), j['quality']['tcg_iterations_per_step_rank0'], j['config']['precond_used_in_timed_steps'])
PY
timeout 600 python tools/ml_probe.py grid100k > gpurun_out/t13_probe.log 2>&1; grep -v "^dpgo_hip" gpurun_out/t13_probe.log | tail -3
