export CM3D_BENCH_CACHE=/tmp/c
for v in "" _lm384 _lm256 _lm128; do   # builds: make -C cm3d_amd/csrc variant NAME=lm256 EXTRA="-DMD_LONG_MIN=256 -DMD_BATCH_LONG=256"
  for c in c2 c1 c4; do
    fr=256
    CM3D_LIB=cm3d_amd/libcm3d_hip$v.so python3 bench.py --config $c --frames $fr --reuse-batch --cpu-sample 0 --no-secondary --steps 100 > gpurun_out/r4_lm${v}_$c.json 2>/dev/null || exit 1
    python3 - <<PY
import json
d=json.loads(open('gpurun_out/r4_lm${v}_$c.json').read().strip().splitlines()[-1])
print('lm$v', '$c', round(d['value']), d['ms_per_step'], d['kernels']['stage_ms_one_batch_alone']['medoid'], flush=True)
PY
  done
done
