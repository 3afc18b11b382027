#!/bin/bash
# Runs ON THE GPU BOX: tools/stage_time.py once per environment setting (one "VAR=value [VAR=value ...]" string per argument
# before "--"), configs after it.  usage: tools/variants.sh "" "CM3D_CP_SPAN=8" "CM3D_RLE_FORM=block" -- c2 256
SETS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do SETS+=("$1"); shift; done
[ "$1" == "--" ] && shift
for s in "${SETS[@]}"; do
  echo "== ${s:-default}"
  env $s timeout -k 10 300 python3 tools/stage_time.py "$@" 2>&1 | grep -v amdgpu.ids || echo "failed: $s"
done
