#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
for w in 5k euroc 50k; do
  for v in "" "FLAME_HIP_SCAN_CUB=1"; do
    echo "=== $w $v"
    env $v python tools/upload_time.py $w 2>&1 | grep upload | tail -n 3
  done
done
