# dev helper (gpurun): facade frame latency with / without one frame-path feature (env toggle), alternating in one call
# usage: ab_env.sh FLAME_HIP_SCAN_CUB   (any env toggle of the library)
show() { python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('  '.join('%s %.3f (%.3f+%.3f)' % (k, v['update_ms']['p50'], v['sync_graph_ms_p50'], v['nltgv2_ms_p50']) for k,v in d.items()))
"; }
for i in 1 2 3; do
  echo -n "on  : "; python tools/facade_bench.py 2>&1 | tail -1 | show
  echo -n "off : "; env $1=1 python tools/facade_bench.py 2>&1 | tail -1 | show
done
