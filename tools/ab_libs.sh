#!/bin/bash
# Runs ON THE GPU BOX: A/B of two builds of the library in ONE call (box-to-box differences are 5 %): alternating runs of the default bench.
# usage: tools/ab_libs.sh cm3d_amd/libcm3d_hip_prev.so cm3d_amd/libcm3d_hip.so [more .so ...] [bench args]
export CM3D_BENCH_CACHE=/tmp/c
LIBS=()
while [ $# -gt 0 ] && [[ "$1" == *.so ]]; do LIBS+=("$1"); shift; done
for rep in 1 2; do
  for lib in "${LIBS[@]}"; do
    CM3D_LIB=$lib python3 bench.py --cpu-sample 0 --no-secondary --steps 300 "$@" > /tmp/ab.json 2>/dev/null || exit 1
    python3 - "$lib" <<'PY'
import json, sys
d = json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1])
k = d['kernels']['stage_ms_one_batch_alone']
print(f"{sys.argv[1]:40s} {d['value']:10.0f} frames/s  {d['ms_per_step']:.4f} ms  project alone {k['project']:.4f}  medoid {k['medoid']:.4f}  long {d.get('value_long')}", flush=True)
PY
  done
done
