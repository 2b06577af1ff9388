# dev helper (gpurun): facade frame latency for several values of one env variable of the library, alternating
# usage: ab_env_val.sh FLAME_HIP_TRI_POS 0 1 2
show() { python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('  '.join('%s %.3f (%.3f+%.3f)' % (k, v['update_ms']['p50'], v['sync_graph_ms_p50'], v['nltgv2_ms_p50']) for k,v in d.items()))
"; }
var=$1; shift
for i in 1 2 3; do
  for v in "$@"; do echo -n "$var=$v : "; env $var=$v python tools/facade_bench.py --workloads tum,5k,euroc,50k 2>&1 | tail -1 | show; done
done
