#!/bin/bash
# Runs ON THE GPU BOX: the default bench (one batch at a time, no CPU legs) once per value of an environment variable.
# usage: tools/sweep_env.sh VAR v1 v2 ... [-- bench flags]
VAR=$1; shift
VALS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do VALS+=("$1"); shift; done
[ "$1" == "--" ] && shift
for v in "${VALS[@]}"; do
  env $VAR=$v timeout -k 10 300 python3 bench.py --cpu-sample 0 --no-secondary --in-flight 1 --steps 200 "$@" > gpurun_out/sweep_$VAR_$v.json 2> gpurun_out/sweep_err.log || { echo "$VAR=$v failed"; tail -3 gpurun_out/sweep_err.log; continue; }
  python3 -c "
import json,sys
d=json.load(open('gpurun_out/sweep_$VAR_$v.json'))
print('$VAR=$v', d['value'], 'frames/s  project alone', d['roofline']['avg_launch_ms_alone'], 'ms  frac_alone', d['roofline']['frac_alone'], d['kernels']['stage_ms_one_batch_alone'])
"
done
