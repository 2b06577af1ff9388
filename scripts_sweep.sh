#!/bin/bash
# quick sweep used during development (gpurun): bench variants
mkdir -p gpurun_out
run() { echo "== $*"; timeout 300 python bench.py --no-cpu --steps 10 --warmup 2 "$@" 2>&1 | tail -1 | python -c "
import sys,json
l=sys.stdin.read().strip()
try:
  d=json.loads(l); r=d['roofline']; c=d['config']
  print('  it/s %.0f  us/iter %.3f  launch_us %.2f  iters/launch %.1f  frac %.3f  tiles %s depth %s nt %s' % (d['value'], d['us_per_iteration'], r['launch_us'], r['iters_per_launch'], r['frac'], c['num_tiles'], c['tile_depth'], c['tile_threads']))
except Exception as e:
  print('  FAIL', l[-300:])
"; }
run --workload 50k --path 1
run --workload 50k --path 1 --no-graph
run --workload 50k
run --workload 50k --no-graph
for own in 128 192 256 384; do for d in 2 3 4 6; do run --workload 50k --tile-own $own --tile-depth $d; done; done
run --workload 5k
run --workload 5k --path 1
run --workload 200k
run --workload 200k --path 1
