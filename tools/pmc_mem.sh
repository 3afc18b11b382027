#!/bin/bash
# Runs ON THE GPU BOX: L1 / L2 request counters of the default bench's kernels (one batch at a time), one rocprofv3 pass per
# counter group; gpurun_out/pmc_mem/<group>/.  tools/pmc_mem_summary.py prints per-kernel averages.
set -o pipefail
export TMPDIR=/tmp
export CM3D_BENCH_CACHE=/tmp/cm3d_bench_cache
OUT=gpurun_out/pmc_mem
mkdir -p $OUT
B="python3 bench.py --cpu-sample 0 --no-secondary --steps 3 --warmup 1 --in-flight 1"
i=0
for g in "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_READ_sum TCP_TCC_WRITE_REQ_sum" \
         "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum" \
         "TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum" \
         "TA_BUSY_avr TCC_BUSY_avr TA_TOTAL_WAVEFRONTS_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
         "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum TCC_TAG_STALL_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $g --kernel-trace -d $OUT/g$i -o p --output-format csv -- $B > $OUT/g$i.json 2> $OUT/g$i.err || { echo "group $i failed"; tail -3 $OUT/g$i.err; }
done
echo done
