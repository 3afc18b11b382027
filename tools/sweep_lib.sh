#!/bin/bash
# Runs ON THE GPU BOX: the default bench (one batch at a time, no CPU legs) once per library build, interleaved `reps` times
# (run-to-run state of the box moves every kernel by several per cent: the masks stage, which no variant touches, is printed
# as the control).  usage: tools/sweep_lib.sh reps lib1.so lib2.so ... [-- bench flags]
REPS=$1; shift
LIBS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do LIBS+=("$1"); shift; done
[ "$1" == "--" ] && shift
for r in $(seq $REPS); do
  for l in "${LIBS[@]}"; do
    CM3D_LIB=$l timeout -k 10 300 python3 bench.py --cpu-sample 0 --no-secondary --in-flight 1 --steps 200 "$@" > gpurun_out/sweep_lib.json 2> gpurun_out/sweep_err.log || { echo "$l failed"; tail -3 gpurun_out/sweep_err.log; continue; }
    python3 -c "
import json
d=json.load(open('gpurun_out/sweep_lib.json'))
s=d['kernels']['stage_ms_one_batch_alone']
print('$l', round(d['value']), 'frames/s  project alone', d['roofline']['avg_launch_ms_alone'], ' in region', d['roofline']['avg_launch_ms'], ' control: masks', s['masks'], 'medoid', s['medoid'])
"
  done
done
