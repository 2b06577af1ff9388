#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r03_run2; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1
tail -n 8 $O/pytest.log
python tools/facade_bench.py --repeats 25 --getters 2 > $O/facade_getters2.json 2> $O/facade.err
cat $O/facade_getters2.json
