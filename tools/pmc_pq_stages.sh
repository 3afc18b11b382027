#!/bin/bash
# Runs ON THE GPU BOX with the diagnostic library: SQ counters of k_project_q with the chunk loop cut after each stage
# (CM3D_PQ_STAGE, project_q.h) -- the differences between consecutive stages say what a stage costs in issued instructions,
# in cycles parked at s_waitcnt and in issue stalls.  Counters only (--pmc with --kernel-trace), one batch at a time.
set -o pipefail
export TMPDIR=/tmp
export CM3D_BENCH_CACHE=/tmp/cm3d_bench_cache
export CM3D_LIB=cm3d_amd/libcm3d_hip_diag.so
OUT=gpurun_out/pmc_pq
mkdir -p $OUT
B="python3 bench.py --cpu-sample 0 --no-secondary --steps 3 --warmup 1 --in-flight 1"
for st in ${PQ_STAGES:-0 1 2 3 4 5 99}; do
  i=0
  for g in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAIT_INST_LDS"; do
    i=$((i+1))
    CM3D_PQ_STAGE=$st timeout -k 10 200 rocprofv3 --pmc $g --kernel-trace -d $OUT/s${st}_g$i -o p --output-format csv -- $B > $OUT/s${st}_g$i.json 2> $OUT/s${st}_g$i.err \
      || { echo "stage $st group $i failed"; tail -3 $OUT/s${st}_g$i.err; }
  done
done
python3 - <<'PY'
import collections, csv, glob, re
res = collections.defaultdict(dict)
for f in sorted(glob.glob("gpurun_out/pmc_pq/s*_g*/p_counter_collection.csv")):
    st = int(re.search(r"/s(\d+)_g", f).group(1))
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_project_q" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        res[st][k] = sum(v) / len(v)
keys = ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS",
        "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_BRANCH", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAIT_INST_LDS"]
with open("gpurun_out/r4_pmc_pq_stages.txt", "w") as out:
    out.write("k_project_q, chunk loop cut after stage (0 rows+results+draws, 1 +transform, 2 +wedges, 3 +pre-test, 4 +exact chain, 5/99 all); per launch, millions\n")
    out.write(f"{'counter':24s}" + "".join(f"{('s%d' % s):>10s}" for s in sorted(res)) + "\n")
    for k in keys:
        out.write(f"{k:24s}" + "".join(f"{res[s].get(k, float('nan')) / 1e6:10.2f}" for s in sorted(res)) + "\n")
print(open("gpurun_out/r4_pmc_pq_stages.txt").read())
PY
