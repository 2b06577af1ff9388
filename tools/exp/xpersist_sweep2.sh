#!/bin/bash
# two resident tiles per CU: does the second tile's compute hide the first one's hand-off wait?
out=${1:-gpurun_out/r04_xp/sweep2.txt}
mkdir -p $(dirname $out); : > $out
run() { timeout 120 python tools/exp/xpersist_bench.py "$@" 2>&1 | grep "persist 1" | sed "s/^/[$*] /" | cut -c1-220 >> $out; }
for o in 98 110 128; do for d in 2 3 4; do run 50k --own $o --depth $d --percu 2 --threads 512; done; done
for o in 20 24 32; do for d in 3 4 5; do run euroc --own $o --depth $d --percu 2 --threads 512; done; done
for o in 12 16; do for d in 3 5; do run 5k --own $o --depth $d --percu 2 --threads 256; done; done
cat $out
