#!/bin/bash
# tools/gpu.sh <timeout-seconds> '<command>' -- dev aid around gpurun.  gpurun snapshots /root/repo when a GPU box has been
# acquired, i.e. AFTER queueing (minutes): edits made meanwhile travel half-done (r05: a bench.py newer than the .so it
# called).  This wrapper freezes the tree into .stage/ first and runs the command THERE; the working tree stays editable.
# .stage/gpurun_out is a link to the real gpurun_out/, so the command's relative output paths are merged back as usual.
set -e
cd "$(dirname "$0")/.."
t=$1; shift
mkdir -p gpurun_out
rm -rf .stage; mkdir .stage
tar --exclude=./.git --exclude=./gpurun_out --exclude=./.stage --exclude=./build --exclude=__pycache__ --exclude=./.pytest_cache -cf - . | (cd .stage && tar xf -)
ln -sfn ../gpurun_out .stage/gpurun_out
/usr/local/graft/bin/gpurun --timeout "$t" -- "mkdir -p gpurun_out; cd .stage && $*"
rc=$?
rm -rf .stage
exit $rc
