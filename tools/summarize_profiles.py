#!/usr/bin/env python3
"""gpurun_out/<tag>/ (written by tools/collect_profiles.sh) -> profiles/<tag>_*  and profiles/traffic.json.

HBM traffic per launch follows MI355X_MICROARCH.md 'HBM': separate --pmc passes for FETCH_SIZE and WRITE_SIZE (KiB).
On gfx950 FETCH_SIZE reports half of the bytes of requests that pull whole 128-byte lines as a stream, so kernels whose reads
are such streams are doubled (STREAM_READERS).  For the projection kernel the factor is calibrated on a launch with known bytes
(tools/pmc_fetch_calibration.sh: camera loop off, 179.2 MB of raw rows read, FETCH_SIZE = 94.2 MB).
WRITE_SIZE is taken as is (it matches the projection's hit words + counts to the percent)."""
import collections
import csv
import json
import os
import shutil
import sys

STREAM_READERS = {"k_project_hits", "k_project_q", "k_erode_pack", "k_compact_hits"}        # 16-byte-per-lane loads


def kname(full):
    """'void k_project_hits<true, true, 5>(float const*, ...)' -> 'k_project_hits'"""
    n = full.split("(")[0].strip()
    if n.startswith("void "):
        n = n[5:]
    return n.split("<")[0]


def main(tag):
    src = os.path.join("gpurun_out", tag)
    os.makedirs("profiles", exist_ok=True)
    shutil.copy(os.path.join(src, "kt", "k_kernel_stats.csv"), f"profiles/{tag}_c2_rle_kernel_stats.csv")
    shutil.copy(os.path.join(src, "kt1", "k_kernel_stats.csv"), f"profiles/{tag}_c2_rle_one_batch_kernel_stats.csv")
    shutil.copy(os.path.join(src, "bench_profiled.json"), f"profiles/{tag}_c2_rle_bench_profiled.json")
    shutil.copy(os.path.join(src, "bench_profiled_one_batch.json"), f"profiles/{tag}_c2_rle_one_batch_bench_profiled.json")
    tr = collections.defaultdict(dict)
    for d, c in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(os.path.join(src, d, "p_counter_collection.csv"))):
            if r["Counter_Name"] == c:
                agg[kname(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            if k.startswith("k_"):
                tr[k][c + "_KiB_per_launch"] = sum(v) / len(v)
    out = {}
    for k, v in sorted(tr.items()):
        f, w = v.get("FETCH_SIZE_KiB_per_launch", 0.0), v.get("WRITE_SIZE_KiB_per_launch", 0.0)
        corr = 2.0 if k in STREAM_READERS else 1.0
        v["fetch_correction"] = corr
        v["hbm_bytes_per_launch"] = int((f * corr + w) * 1024)
        out[k] = v
    json.dump(out, open(f"profiles/{tag}_c2_rle_pmc_traffic.json", "w"), indent=1)
    traffic = {"c2_rle": {k: v["hbm_bytes_per_launch"] for k, v in out.items()}, "source": f"profiles/{tag}_c2_rle_pmc_traffic.json",
               "frames_per_gpu": 256}
    # vector instructions of the medoid kernels per pass (SQ_INSTS_VALU, its own PMC pass): bench.py's kernels.medoid
    insts_csv = os.path.join(src, "insts", "p_counter_collection.csv")
    if os.path.exists(insts_csv):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(insts_csv)):
            if r["Counter_Name"] == "SQ_INSTS_VALU":
                agg[kname(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        per = {k: sum(v) / len(v) for k, v in agg.items() if k.startswith("k_")}
        json.dump(per, open(f"profiles/{tag}_c2_rle_pmc_insts_valu.json", "w"), indent=1)
        # per PASS: every medoid dispatch of the run (both instantiations of the tile kernel, the reduction, the long lists'
        # second pass) summed, divided by the number of passes (= dispatches of the reduction kernel)
        passes = max(1, len(agg.get("k_medoid_reduce", [])))
        traffic["c2_rle"]["k_medoid_insts_valu"] = int(sum(sum(v) for k, v in agg.items() if k.startswith("k_medoid")) / passes)
        for k in ("k_project_q", "k_project_hits"):          # the projection launch's own count (bench.py: roofline.valu_pipe)
            if k in per:
                traffic["c2_rle"][k + "_insts_valu"] = int(per[k])
    json.dump(traffic, open("profiles/traffic.json", "w"), indent=1)
    for label, fn in (("batches in flight (default: four since the end of r04)", f"profiles/{tag}_c2_rle_kernel_stats.csv"),
                      ("one batch at a time", f"profiles/{tag}_c2_rle_one_batch_kernel_stats.csv")):
        rows = list(csv.DictReader(open(fn)))
        print(label)
        print(f"  {'kernel':42s} {'calls':>5s} {'avg_us':>9s} {'%':>6s} {'HBM MB/launch':>14s}")
        for r in rows[:16]:
            name = kname(r["Name"])
            mb = out.get(name, {}).get("hbm_bytes_per_launch")
            print(f"  {name[:42]:42s} {r['Calls']:>5s} {float(r['AverageNs']) / 1e3:9.1f} {float(r['Percentage']):6.2f} {'' if mb is None else f'{mb / 1e6:14.1f}'}")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r03")
