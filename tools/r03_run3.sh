#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r03_run3; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1
tail -n 12 $O/pytest.log
