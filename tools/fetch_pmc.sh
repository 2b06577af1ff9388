#!/bin/bash
# dev helper (gpurun): FETCH_SIZE / WRITE_SIZE per launch of the tile kernel.  usage: fetch_pmc.sh <bench args...>
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/fp_$c
  rocprofv3 --kernel-trace --pmc $c -d gpurun_out/fp_$c -o p --output-format csv -- python bench.py --no-cpu --steps 3 --warmup 1 "$@" > /dev/null 2>&1
  python - <<PY
import csv, glob
f = glob.glob('gpurun_out/fp_$c/**/*counter_collection.csv', recursive=True)[0]
tot = n = 0
for r in csv.DictReader(open(f)):
    if 'k_tile' in r['Kernel_Name'] and r['Counter_Name'] == '$c':
        tot += float(r['Counter_Value']); n += 1
print('$c', 'KiB/launch raw %.0f' % (tot / max(n, 1)), 'launches', n)
PY
done
