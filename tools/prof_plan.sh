cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pp -o p -- python tools/upload_time.py $1 $2 > /dev/null 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/pp/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:14]:
    print('%-60s calls %5s avg %8.1f us total %7.2f ms' % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
rm -rf gpurun_out/pp
