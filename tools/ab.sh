#!/bin/bash
# dev helper (gpurun): quick it/s of the three synthetic sizes.  usage: ab.sh [bench args]
for w in 50k 5k 200k; do
  python bench.py --no-cpu --steps 30 --warmup 5 --workload $w "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w %9.0f it/s  launch %.2f us' % (d['value'], d['roofline']['launch_us']))"
done
