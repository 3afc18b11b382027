#!/bin/bash
# Runs ON THE GPU BOX: the default bench (three batches in flight, no CPU legs) once per environment setting.
# usage: tools/variants_bench.sh "" "VAR=value ..." ... [-- bench flags]
SETS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do SETS+=("$1"); shift; done
[ "$1" == "--" ] && shift
for s in "${SETS[@]}"; do
  env CM3D_BENCH_CACHE=/tmp/cm3d_bench_cache $s timeout -k 10 300 python3 bench.py --cpu-sample 0 --no-secondary --steps 300 --warmup 20 "$@" > gpurun_out/vb.json 2> gpurun_out/vb.err || { echo "failed: $s"; tail -3 gpurun_out/vb.err; continue; }
  python3 -c "
import json
d=json.load(open('gpurun_out/vb.json'))
print('${s:-default}:', d['value'], 'frames/s', d['ms_per_step'], 'ms/step; project in flight', d['roofline']['avg_launch_ms'], 'alone', d['roofline']['avg_launch_ms_alone'], d['kernels']['stage_ms_one_batch_alone'])
"
done
