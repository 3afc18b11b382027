#!/bin/bash
# Runs ON THE GPU BOX: medoid stage time and pass rate of the C1 / C5 / C2 shapes for builds with different two-pass
# crossovers (libcm3d_hip_md<N>.so = -DMD_LONG_MIN=N; the product library is the baseline).
for cfg in "c1 64" "c5 64" "c2 256"; do
  set -- $cfg
  for lib in libcm3d_hip.so libcm3d_hip_md256.so libcm3d_hip_md128.so libcm3d_hip_md64.so; do
    [ -f cm3d_amd/$lib ] || continue
    for inf in 1 3; do
      CM3D_LIB=$PWD/cm3d_amd/$lib timeout -k 10 300 python3 bench.py --config $1 --frames $2 --cpu-sample 0 --no-secondary --in-flight $inf --steps 150 --warmup 10 > gpurun_out/mdx.json 2> gpurun_out/mdx.err || { echo "$1 $lib failed"; tail -2 gpurun_out/mdx.err; continue; }
      python3 -c "
import json
d=json.load(open('gpurun_out/mdx.json'))
s=d['kernels']['stage_ms_one_batch_alone']
print('$1 x$2 $lib in-flight $inf: %.0f frames/s  medoid %.1f us  project %.1f  compact %.1f  masks %.1f' % (d['value'], s['medoid']*1e3, s['project']*1e3, s['compact']*1e3, s['masks']*1e3))"
    done
  done
done
