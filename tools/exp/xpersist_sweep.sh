#!/bin/bash
# depth / tile-size sweep of the resident tiles over all XCDs (tools/exp/xpersist_bench.py); output: one line per configuration
out=${1:-gpurun_out/r04_xp/sweep.txt}
mkdir -p $(dirname $out); : > $out
run() { timeout 120 python tools/exp/xpersist_bench.py "$@" 2>&1 | grep "persist" | sed "s/^/[$*] /" | cut -c1-200 >> $out; }
for d in 2 3 4 5 6; do run 50k --depth $d; done
for o in 128 160; do for d in 3 4 5 6; do run 50k --own $o --depth $d; done; done
for d in 2 3 4 5 6 8; do run euroc --depth $d; done
for o in 64 96; do for d in 4 6 8; do run euroc --own $o --depth $d; done; done
for d in 2 3 4 5 6 8; do run 5k --depth $d; done
for o in 24 48 64; do for d in 4 6 8; do run 5k --own $o --depth $d; done; done
for o in 16 24 32 50; do for d in 3 5 8; do run tum --own $o --depth $d; done; done
for o in 16 24 32 50; do for d in 3 5 8; do run v2000 --own $o --depth $d; done; done
