#!/bin/bash
# delay between a round's stores and its first poll pass (handle option poll_delay, x 256 clocks): us per iteration
out=gpurun_out/r04_xp/poll_delay.txt; mkdir -p $(dirname $out); : > $out
for d in 0 1 2 3 4 5 6 8 10; do
  for w in 50k euroc 5k tum; do
    timeout 120 python tools/exp/xpersist_bench.py $w --opt poll_delay=$d 2>&1 | grep "persist 1" | sed "s/^/[delay $d] /" | cut -c1-120 >> $out
  done
done
cat $out
