#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r03_run7; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1
tail -n 12 $O/pytest.log
python tools/facade_bench.py --repeats 25 --getters 1 --workloads 5k,euroc,50k > $O/facade.json 2> $O/facade.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03_run7/facade.json'))
for k,v in d.items(): print(k, 'update', v['update_ms']['p50'], 'sync', v['sync_graph_ms_p50'], 'nltgv2', v['nltgv2_ms_p50'], 'dev', v['nltgv2_device_ms'])
PY
