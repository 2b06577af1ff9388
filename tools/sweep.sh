#!/bin/bash
# dev helper (gpurun): bench variants.  usage: scripts_sweep.sh <workload> "<own list>" "<depth list>"
run() { timeout 300 python bench.py --no-cpu --steps 5 --warmup 1 "$@" 2>&1 | tail -1 | python -c "
import sys,json
l=sys.stdin.read().strip()
try:
  d=json.loads(l); r=d['roofline']; c=d['config']
  print('%-44s it/s %8.0f  us/iter %6.3f  launch_us %6.2f  tiles %4s depth %s nt %s' % ('$*', d['value'], d['us_per_iteration'], r['launch_us'], c['num_tiles'], c['tile_depth'], c['tile_threads']))
except Exception as e:
  print('$*', 'FAIL', l[-200:])
"; }
w=$1
for own in $2; do for d in $3; do run --workload $w --tile-own $own --tile-depth $d; done; done
