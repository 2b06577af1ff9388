#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r03_run6; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_offline_lite.py tests/test_facade.py -x -q -m gpu > $O/pytest.log 2>&1
tail -n 25 $O/pytest.log
