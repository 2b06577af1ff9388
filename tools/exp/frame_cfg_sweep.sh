# dev helper (gpurun): frame-stream latency of small graphs: one isolated tile vs halo tiles of a few depths
# columns: sync / solve+results / total ms of the last frames (tools/frame_trace.py)
for w in g20 g18 tum g14 g13 v2000; do
for cfg in "tile_single_max=2048" "tile_single_max=1 tile_depth=4" "tile_single_max=1 tile_depth=5"; do
  echo "== $w $cfg  $(python tools/frame_trace.py $w $cfg 2>&1 | tail -5 | awk '{printf "%s/%s/%s  ", $2,$4,$6}')"
done; done
