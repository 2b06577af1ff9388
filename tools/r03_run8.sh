#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r03_run8; mkdir -p $O
timeout 900 python -m pytest tests/test_facade.py tests/test_gpu_offline_lite.py -x -q -m gpu > $O/pytest.log 2>&1
tail -n 5 $O/pytest.log
python tools/facade_bench.py --repeats 25 --getters 1 > $O/facade.json 2> $O/facade.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03_run8/facade.json'))
for k,v in d.items(): print(k, 'update', v['update_ms']['p50'], 'sync', v['sync_graph_ms_p50'], 'nltgv2', v['nltgv2_ms_p50'], 'dev', v['nltgv2_device_ms'])
PY
