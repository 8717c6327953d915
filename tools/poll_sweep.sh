export GPU_OUT=r3f
for cfg in "0 1" "12 1" "24 3" "36 3" "48 6" "24 8"; do
  set -- $cfg
  echo "== first=$1 sleep=$2"
  DPGO_POLL_FIRST=$1 DPGO_POLL_SLEEP=$2 PROBE_ONLY=jacobi+persistent DPGO_PERSIST_VERBOSE=1 python tools/ml_probe.py grid625 sphere slab 2>&1 | grep -v "^dpgo_hip.* [12] iterations" | sed "s/, [0-9]* iterations//" | awk '/^dpgo_hip/{c[$0]++; next} {print substr($0,1,70)} END{for(k in c) if (c[k]>=2) print substr(k,30,200)}'
done
