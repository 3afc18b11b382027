#!/bin/bash
# Runs ON THE GPU BOX: builds and runs the micro-benchmarks of this directory -> gpurun_out/ubench.txt
set -e
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rates valu_rates.hip
/tmp/valu_rates | tee ../../gpurun_out/ubench.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/pk_clamp pk_clamp.hip 2>/dev/null
/tmp/pk_clamp | tee ../../gpurun_out/pk_clamp.txt
