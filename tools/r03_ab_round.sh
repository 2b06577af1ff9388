#!/bin/bash
# A/B of the phase-P read batch (FLAME_SLOT_ROUND) on small graphs: rebuilds the library on the GPU box
cd "$(dirname "$0")/.." || exit 1
for r in 6 12 8; do
  FLAME_EXTRA_HIPCC_FLAGS="-DFLAME_SLOT_ROUND=$r" python -c "from flame_ros_amd import build; build.build(force=True)" > /dev/null 2>&1
  for w in tum 5k euroc 50k; do
    python bench.py --no-cpu --no-facade --steps 200 --warmup 5 --workload $w 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('round $r $w %9.0f it/s  %.3f us/it  launch %.2f us' % (d['value'], d['us_per_iteration'], d['roofline']['launch_us']))"
  done
done
