#!/bin/bash
# Runs ON THE GPU BOX: instruction-fetch counters (SQC instruction cache, SQ_IFETCH) of the default bench's kernels, one batch at a time.
set -o pipefail
export TMPDIR=/tmp
export CM3D_BENCH_CACHE=/tmp/cm3d_bench_cache
OUT=gpurun_out/pmc_icache
mkdir -p $OUT
B="python3 bench.py --cpu-sample 0 --no-secondary --steps 3 --warmup 1 --in-flight 1"
i=0
for g in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" \
         "SQC_ICACHE_BUSY_CYCLES SQC_ICACHE_INPUT_VALID_READYB SQC_TC_INST_REQ SQC_TC_STALL" \
         "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $g --kernel-trace -d $OUT/g$i -o p --output-format csv -- $B > $OUT/g$i.json 2> $OUT/g$i.err || { echo "group $i failed"; tail -3 $OUT/g$i.err; }
done
python3 tools/pmc_mem_summary.py $OUT > gpurun_out/r4_pmc_icache_summary.txt
echo done
