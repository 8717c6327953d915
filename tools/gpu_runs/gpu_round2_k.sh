mkdir -p gpurun_out
( time timeout 2400 python -m pytest tests -m gpu -q --timeout 1200 ) > gpurun_out/t11_all.log 2>&1
tail -6 gpurun_out/t11_all.log
for a in 1 16; do
  timeout 600 python bench.py --agents-per-gpu $a --no-cpu-baseline --steps 5 --warmup 2 $( [ $a = 16 ] && echo --no-secondary ) > gpurun_out/t11_a$a.log 2>&1
  python - <<PY
import json
j=json.loads([l for l in open('gpurun_out/t11_a$a.log') if l.startswith('{')][-1])
q=j['quality']
print('agents', $a, 'it/s', round(j['value'],1), 'ms', round(j['ms_per_step'],2), 'tcg/step', q['tcg_iterations_per_step_rank0'], 'used', j['config']['precond_used_in_timed_steps'], 'after', round(q['gradnorm_after_step'],4))
if q.get('to_tolerance'):
    for k,v in q['to_tolerance'].items(): print('  ', k, v.get('products'), round(v.get('ms',0),2), v.get('preconditioners'))
if j.get('also'):
    for k,v in j['also'].items(): print('  also', k, round(v['it_per_s'],1), v['tcg_iterations_per_step'])
PY
done
