#!/bin/bash
# debug images on the GPU: tests + facade latency
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r03_run1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_debug_images.py tests/test_facade.py tests/test_gpu_sync.py -x -q -m gpu > $O/pytest.log 2>&1
tail -n 15 $O/pytest.log
python tools/facade_bench.py --repeats 25 --getters 2 > $O/facade_getters2.json 2> $O/facade.err
python tools/facade_bench.py --repeats 25 --getters 0 > $O/facade_getters0.json 2>> $O/facade.err
cat $O/facade_getters2.json $O/facade_getters0.json
