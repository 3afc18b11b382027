#!/bin/bash
# Runs ON THE GPU BOX: SQ (instruction issue / wait) counters of the default bench's kernels, one batch at a time.
set -o pipefail
export TMPDIR=/tmp
export CM3D_BENCH_CACHE=/tmp/cm3d_bench_cache
OUT=gpurun_out/pmc_sq
mkdir -p $OUT
B="python3 bench.py --cpu-sample 0 --no-secondary --steps 3 --warmup 1 --in-flight 1"
i=0
for g in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT" \
         "SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVES SQ_INSTS_FLAT"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $g --kernel-trace -d $OUT/g$i -o p --output-format csv -- $B > $OUT/g$i.json 2> $OUT/g$i.err || { echo "group $i failed"; tail -3 $OUT/g$i.err; }
done
echo done
