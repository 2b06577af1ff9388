# dev (gpurun): the tail of the 50 k frame stream by poll time-out (default: adaptive, 0.5 ms floor) -- resident / recovered frames, update() series
cd "${GRAFT_REPO_ROOT:-/root/repo}"; [ -d .stage ] && cd .stage
for t in default 4000; do
  echo "== FLAME_HIP_PERSIST_TIMEOUT_US=$t"
  if [ $t = default ]; then unset FLAME_HIP_PERSIST_TIMEOUT_US; else export FLAME_HIP_PERSIST_TIMEOUT_US=$t; fi
  FLAME_BENCH_SERIES=1 python tools/facade_bench.py --workloads 50k --repeats 10 2>&1 | tail -3 | cut -c1-1500
done
