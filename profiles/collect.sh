#!/bin/bash
# profiles/collect.sh <round-tag> [bench args...] -- run on the GPU box (gpurun).  Produces under
# gpurun_out/<tag>/: the rocprofv3 kernel-trace stats of `python bench.py`, two SEPARATE PMC passes
# for the HBM traffic (FETCH_SIZE, WRITE_SIZE) as MI355X_MICROARCH.md "HBM" prescribes, one PMC pass
# for the LDS counters, one (r06) for the VALU / wait / issue-stall counters, then a summary (summary.md / summary.json / kernel_stats.csv) that is copied
# by hand into profiles/.  Every rocprofv3 run uses --kernel-trace only (no sys/hip/hsa trace).
here="$(cd "$(dirname "$0")/.." && pwd)"   # (the tree this script lies in: tools/gpu.sh runs a frozen copy)
cd /tmp && export TMPDIR=/tmp
cd "$here"
tag=${1:-r05}; shift
out=gpurun_out/$tag; mkdir -p $out gpurun_out
BENCH="python bench.py --no-cpu --no-facade --no-live-traffic --steps 10 --warmup 2 $*"  # (no profiler inside the profiler)
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o p -- $BENCH > $out/bench_trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o p -- $BENCH > $out/bench_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o p -- $BENCH > $out/bench_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $out/pmc_lds -o p -- $BENCH > $out/bench_lds.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $out/pmc_valu -o p -- $BENCH > $out/bench_valu.log 2>&1
python bench.py --no-live-traffic --steps 10 --warmup 2 $* > $out/bench.json 2> $out/bench.err
python profiles/summarize.py $out
# the raw traces are large (gpurun copies back <= 64 MiB): keep the summaries only
rm -rf $out/trace $out/pmc_fetch $out/pmc_write $out/pmc_lds $out/pmc_valu
