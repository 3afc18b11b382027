#!/bin/bash
# Runs ON THE GPU BOX: stall / level / instruction counters of the default bench's kernels (separate --pmc passes,
# kernel-trace only).  usage: tools/pmc_stall.sh [bench flags]      -> gpurun_out/pmc_stall/summary.txt
export TMPDIR=/tmp
OUT=gpurun_out/pmc_stall
rm -rf $OUT; mkdir -p $OUT
B="python3 bench.py --cpu-sample 0 --no-secondary --steps 2 --warmup 1 $*"
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_IFETCH_LEVEL" "SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_INSTS_BRANCH" "SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "SQ_WAVES SQ_INSTS_FLAT SQ_ACTIVE_INST_SCA SQ_IFETCH" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace -d $OUT/p$i -o p --output-format csv -- $B > $OUT/p$i.json 2> $OUT/p$i.err || { echo "pass $i failed"; tail -5 $OUT/p$i.err; }
done
python3 - <<'PY' | tee gpurun_out/pmc_stall/summary.txt
import csv, collections, glob
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_stall/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    if k in ("k_project_hits", "k_medoid_tiles", "k_rle_erode_pack", "k_compact_hits", "k_sweep_xform", "k_lane_nn_grid", "k_hit_offsets", "k_frame_tables"):
        print(k, {c: round(sum(x) / len(x) / 1e6, 3) for c, x in sorted(v.items())})
PY
