#!/bin/bash
# map kernels of the next frame beside the resident solve (FLAME_HIP_MAPS_BESIDE=1) or ordered behind it (default)
for i in 1 2 3; do for m in 0 1; do
  if [ $m = 1 ]; then export FLAME_HIP_MAPS_BESIDE=1; else unset FLAME_HIP_MAPS_BESIDE; fi
  python tools/facade_bench.py --workloads 50k,euroc --repeats 15 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('beside' if '$m'=='1' else 'behind', ' '.join('%s %.3f (%.3f + %.3f)' % (w, r['update_ms']['p50'], r['sync_graph_ms_p50'], r['nltgv2_ms_p50']) for w,r in d.items()))"
done; done
