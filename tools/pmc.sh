#!/bin/bash
# dev helper (gpurun): PMC counters of the tile kernel.  usage: scripts_pmc.sh <tag> <bench args...>
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
tag=$1; shift
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES"; do
  n=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set -d gpurun_out/pmc_${tag}_$n -o p --output-format csv -- python bench.py --no-cpu --steps 3 --warmup 1 "$@" > gpurun_out/pmc_${tag}_$n.log 2>&1
done
python - <<PY
import glob, csv, collections
for f in sorted(glob.glob('gpurun_out/pmc_${tag}_*/**/*counter_collection.csv', recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:40]
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    for k, d in acc.items():
        print(k)
        for c, v in d.items(): print('   %-26s %.4g' % (c, v))
PY
