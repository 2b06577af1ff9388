mkdir -p gpurun_out/r04_flaky2
T="tests/test_gpu_persist.py tests/test_facade.py tests/test_gpu_offline_lite.py tests/test_gpu_sync.py tests/test_gpu_part_native.py tests/test_gpu_nccl_self.py tests/test_gpu_partition_procs.py"
for i in 1 2 3 4 5 6; do python -m pytest $T -m gpu -q -rf 2>&1 | tail -30 > gpurun_out/r04_flaky2/plain$i.txt; tail -1 gpurun_out/r04_flaky2/plain$i.txt; grep -E "^FAILED" gpurun_out/r04_flaky2/plain$i.txt; done
nproc
PIDS=""
for k in $(seq 1 96); do python -c "
import time
t=time.time()
while time.time()-t<600: sum(range(10000))
" & PIDS="$PIDS $!"; done
for i in 1 2 3; do python -m pytest $T -m gpu -q -rf 2>&1 | tail -30 > gpurun_out/r04_flaky2/load$i.txt; tail -1 gpurun_out/r04_flaky2/load$i.txt; grep -E "^FAILED" gpurun_out/r04_flaky2/load$i.txt; done
kill $PIDS 2>/dev/null
wait 2>/dev/null
