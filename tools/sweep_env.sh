#!/bin/bash
# usage: tools/sweep_env.sh VAR v1 v2 ... ; prints the per-stage times of the default bench for each value
VAR=$1; shift
for v in "$@"; do
  env $VAR=$v timeout -k 10 200 python bench.py --steps 20 --warmup 3 --cpu-sample 0 --no-secondary 2>/dev/null > /tmp/sweep_$v.json
  python - "$v" <<'PY'
import json, sys
d = json.load(open(f"/tmp/sweep_{sys.argv[1]}.json"))
print(sys.argv[1], d["value"], d["kernels"]["stage_ms_one_batch_alone"])
PY
done
