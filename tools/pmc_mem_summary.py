#!/usr/bin/env python3
"""gpurun_out/pmc_mem/g*/p_counter_collection.csv (tools/pmc_mem.sh) -> per-kernel averages per launch."""
import collections, csv, glob, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob((sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc_mem") + "/g*/p_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if n.startswith("k_") and "selftest" not in n:
            agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, c in agg.items():
    print(n)
    for k, v in sorted(c.items()):
        print(f"    {k:44s} {sum(v) / len(v):16.1f}")
