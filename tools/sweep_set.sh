#!/bin/bash
# usage: tools/sweep_set.sh KEY v1 v2 ... ; default bench with --set KEY=v for each value
KEY=$1; shift
for v in "$@"; do
  timeout -k 10 200 python bench.py --steps 20 --warmup 3 --cpu-sample 0 --no-secondary --set $KEY=$v 2>/dev/null > /tmp/sweep_$v.json
  python - "$v" <<'PY'
import json, sys
d = json.load(open(f"/tmp/sweep_{sys.argv[1]}.json"))
print(sys.argv[1], d["value"], d.get("max_points_in_a_mask"), d["in_mask_points_per_step"], d["kernels"]["stage_ms"])
PY
done
