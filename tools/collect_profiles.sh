#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): kernel-trace stats + separate PMC passes for HBM traffic of the default
# bench command; everything lands in gpurun_out/$1.  tools/summarize_profiles.py turns it into profiles/.
set -o pipefail
TAG=${1:-r03}
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
export CM3D_BENCH_CACHE=/tmp/cm3d_bench_cache      # the same synthetic batches for all passes (generated once)
B="python3 bench.py --cpu-sample 0 --no-secondary"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/kt -o k --output-format csv -- $B --steps 20 --warmup 3 > $OUT/bench_profiled.json 2> $OUT/kt.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/kt1 -o k --output-format csv -- $B --steps 20 --warmup 3 --in-flight 1 > $OUT/bench_profiled_one_batch.json 2> $OUT/kt1.err || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o p --output-format csv -- $B --steps 3 --warmup 1 > $OUT/fetch.json 2> $OUT/fetch.err || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o p --output-format csv -- $B --steps 3 --warmup 1 > $OUT/write.json 2> $OUT/write.err || exit 1
# vector instructions per launch (the medoid's VALU roofline in bench.py): its own pass, kernel-trace only
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU --kernel-trace -d $OUT/insts -o p --output-format csv -- $B --steps 3 --warmup 1 > $OUT/insts.json 2> $OUT/insts.err || exit 1
echo collected
