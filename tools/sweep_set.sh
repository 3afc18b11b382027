#!/bin/bash
# usage: tools/sweep_set.sh "VAR1=a VAR2=b" "VAR1=c VAR2=d" ... ; one default-bench run per environment string
i=0
for envs in "$@"; do
  i=$((i+1))
  env $envs timeout -k 10 200 python bench.py --steps 20 --warmup 3 --cpu-sample 0 --no-secondary 2>/dev/null > /tmp/sweepset_$i.json
  python - "$envs" "$i" <<'PY'
import json, sys
d = json.load(open(f"/tmp/sweepset_{sys.argv[2]}.json"))
print(sys.argv[1], d["value"], d["kernels"]["stage_ms_one_batch_alone"])
PY
done
