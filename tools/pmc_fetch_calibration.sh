#!/bin/bash
# Runs ON THE GPU BOX: FETCH_SIZE / WRITE_SIZE of k_project_hits with and without its camera loop (tools/fetch_calibration.py),
# separate --pmc passes, kernel-trace only  ->  gpurun_out/fetch_calibration.json
export TMPDIR=/tmp
OUT=gpurun_out/fetch_cal
rm -rf $OUT; mkdir -p $OUT
export CM3D_LIB=$PWD/cm3d_amd/libcm3d_hip_diag.so
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace -d $OUT/$c -o p --output-format csv -- python3 tools/fetch_calibration.py > $OUT/$c.log 2> $OUT/$c.err || { echo "$c pass failed"; tail -3 $OUT/$c.err; exit 1; }
done
python3 - <<'PY'
import csv, json
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = [r for r in csv.DictReader(open(f"gpurun_out/fetch_cal/{c}/p_counter_collection.csv")) if r["Counter_Name"] == c and "k_project_hits" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    v = [float(r["Counter_Value"]) * 1024 for r in rows]          # KiB -> bytes
    assert len(v) == 8, len(v)
    out[c] = {"no_camera_loop_bytes": sum(v[2:5]) / 3, "full_kernel_bytes": sum(v[5:8]) / 3}
out["known"] = open("gpurun_out/fetch_cal/FETCH_SIZE.log").read().strip().splitlines()[-1]
json.dump(out, open("gpurun_out/fetch_calibration.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
