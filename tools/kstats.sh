#!/bin/bash
# Runs ON THE GPU BOX: per-kernel average durations of the default bench (rocprofv3 --kernel-trace --stats).
export TMPDIR=/tmp
OUT=gpurun_out/kstats
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/kt -o k --output-format csv -- python3 bench.py --cpu-sample 0 --no-secondary --steps 20 --warmup 3 "$@" > $OUT/bench.json 2> $OUT/err.log || exit 1
python3 - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/kstats/kt/k_kernel_stats.csv")))
for r in rows[:16]:
    n = r["Name"].split("(")[0].replace("void ", "")
    print(f"{n[:40]:40s} {r['Calls']:>5s} {float(r['AverageNs'])/1e3:9.1f} us {float(r['Percentage']):6.2f} %")
PY
