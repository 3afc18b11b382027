#!/bin/bash
# Runs ON THE GPU BOX: instruction-mix counters of the default bench (separate --pmc passes, kernel-trace only).
export TMPDIR=/tmp
OUT=gpurun_out/pmc_insts
mkdir -p $OUT
B="python3 bench.py --cpu-sample 0 --no-secondary --steps 2 --warmup 1 $*"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM --kernel-trace -d $OUT/a -o p --output-format csv -- $B > $OUT/a.json 2> $OUT/a.err || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace -d $OUT/b -o p --output-format csv -- $B > $OUT/b.json 2> $OUT/b.err || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA --kernel-trace -d $OUT/c -o p --output-format csv -- $B > $OUT/c.json 2> $OUT/c.err || exit 1
python3 - <<'PY'
import csv, collections, glob
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in "abc":
    for f in glob.glob(f"gpurun_out/pmc_insts/{d}/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    if k.startswith("k_"):
        print(k, {c: round(sum(x) / len(x) / 1e6, 3) for c, x in v.items()})
PY
