#!/bin/bash
# Runs ON THE GPU BOX (one gpurun call): the round's evidence under gpurun_out/r04/.  tools/summarize_profiles.py r04 afterwards.
#   part a: rocprof kernel stats (three batches in flight / one at a time) + PMC traffic + PMC vector instructions of the default bench,
#           the default bench line with the CPU legs, the one-batch line, the end-to-end line
#   part b: the other BASELINE shapes at 256 frames per batch with their kernel stats
set -o pipefail
export TMPDIR=/tmp
export CM3D_BENCH_CACHE=/tmp/cm3d_bench_cache
OUT=gpurun_out/r04
mkdir -p $OUT
part=${1:-a}
if [ "$part" == "a" ]; then
  tools/collect_profiles.sh r04 || exit 1
  echo "profiles collected"
  python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || exit 1
  echo "default bench done"
  python3 bench.py --steps 20 --warmup 5 --cpu-sample 0 --no-secondary > $OUT/bench_driver_flags.json 2>> $OUT/bench_default.err || exit 1
  python3 bench.py --cpu-sample 0 --no-secondary --in-flight 1 > $OUT/bench_one_batch.json 2>> $OUT/bench_default.err || exit 1
  python3 bench.py --end-to-end 4096 > $OUT/end_to_end_4096.json 2> $OUT/e2e.err || exit 1
  echo "part a done"
else
  for c in c1 c4; do
    timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $OUT/kt_$c -o k --output-format csv -- python3 bench.py --config $c --frames 256 --reuse-batch --cpu-sample 0 --no-secondary --steps 30 --warmup 3 > $OUT/bench_${c}_256frames_profiled.json 2> $OUT/kt_$c.err || exit 1
    python3 bench.py --config $c --frames 256 --reuse-batch --cpu-sample 0 --no-secondary --steps 100 > $OUT/bench_${c}_256frames.json 2>> $OUT/kt_$c.err || exit 1
    echo "$c done"
  done
  python3 bench.py --config c5 --frames 256 --reuse-batch --cpu-sample 0 --no-secondary --steps 20 --warmup 3 > $OUT/bench_c5_256frames.json 2> $OUT/c5.err || exit 1
  echo "part b done"
fi
