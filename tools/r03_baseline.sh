#!/bin/bash
# round-3 baseline: facade latency (reference-default params), small-graph timelines, small-graph it/s
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r03_base; mkdir -p $O
python tools/facade_bench.py --repeats 25 --getters 2 > $O/facade_getters2.json 2> $O/facade_getters2.err
python tools/facade_bench.py --repeats 25 --getters 0 > $O/facade_getters0.json 2> $O/facade_getters0.err
python tools/facade_bench.py --repeats 25 --getters 0 --no-debug > $O/facade_nodebug.json 2> $O/facade_nodebug.err
for w in tum 5k euroc 50k; do
  python tools/tile_timeline.py --workload $w > $O/timeline_$w.txt 2>&1
  python bench.py --workload $w --no-cpu --steps 50 > $O/bench_$w.json 2> $O/bench_$w.err
done
tail -n 3 $O/facade_*.json
