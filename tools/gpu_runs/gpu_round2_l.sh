mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -k "golden or default_preconditioner or bench_prints" > gpurun_out/t12_new.log 2>&1
tail -4 gpurun_out/t12_new.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/t12_smoke.log 2>&1; tail -2 gpurun_out/t12_smoke.log
bash tools/profile_round.sh r02_v4 > gpurun_out/t12_profile.log 2>&1
python - <<PY
import json
j=json.load(open('gpurun_out/r02_v4_bench.json'))
print(j['value'], j['ms_per_step'], j['config']['precond_used_in_timed_steps'], j['roofline']['frac'], j['roofline']['warm']['frac'], j['roofline']['traffic_source'])
print(j['cpu_baseline']['value'], j['cpu_baseline']['single_agent_port']['value'])
PY
