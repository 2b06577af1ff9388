#!/bin/bash
# r04 rocprofv3 evidence: kernel stats + FETCH/WRITE + LDS PMC passes per workload (profiles/collect.sh)
cd "$(dirname "$0")/.." || exit 1
for w in "$@"; do
  if [ "$w" = "batch256" ]; then bash profiles/collect.sh r06_batch256 --batch 256 --batch-win 16 > gpurun_out/collect_r06_$w.log 2>&1
  else bash profiles/collect.sh r06_$w --workload $w > gpurun_out/collect_r06_$w.log 2>&1; fi
  tail -n 3 gpurun_out/collect_r06_$w.log | cut -c1-300
done
