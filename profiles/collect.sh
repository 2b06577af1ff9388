#!/bin/bash
# profiles/collect.sh <round-tag> [bench args...] -- run on the GPU box (gpurun).  Produces under
# gpurun_out/<tag>/: the rocprofv3 kernel-trace stats of `python bench.py` and two SEPARATE PMC
# passes (FETCH_SIZE, WRITE_SIZE) as MI355X_MICROARCH.md "HBM" prescribes, then a summary
# (summary.md + kernel_stats.csv) that is copied by hand into profiles/.
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
tag=${1:-r01}; shift
out=gpurun_out/$tag; mkdir -p $out
BENCH="python bench.py --no-cpu --steps 10 --warmup 2 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o p -- $BENCH > $out/bench_trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o p -- $BENCH > $out/bench_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o p -- $BENCH > $out/bench_write.log 2>&1
python bench.py --steps 10 --warmup 2 $* > $out/bench.json 2> $out/bench.err
python profiles/summarize.py $out
