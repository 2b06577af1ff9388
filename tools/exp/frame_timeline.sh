# dev helper (gpurun): kernels and copies of ONE whole from-features frame of the C++ facade bench, in launch order.
#   tools/exp/frame_timeline.sh <workload> [first-kernel-substring]
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
[ -d .stage ] && cd .stage
w=${1:-50k}; first=${2:-k_dt_}
rm -rf gpurun_out/ftl
FLAME_BENCH_FRONTEND=1 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/ftl -o p -- python tools/facade_bench.py --workloads $w --repeats 3 > gpurun_out/ftl_$w.log 2>&1
python - "$w" "$first" <<'PY'
import csv, glob, re, sys
w, first = sys.argv[1], sys.argv[2]
best = None
for f in glob.glob('gpurun_out/ftl/**/*kernel_trace.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    if best is None or len(rows) > len(best[1]): best = (f, rows)
f, krows = best
rows = [dict(kind='K', name=r['Kernel_Name'], s=int(r['Start_Timestamp']), e=int(r['End_Timestamp']), q=r.get('Queue_Id', '')) for r in krows]
m = f.replace('kernel_trace', 'memory_copy_trace')
try:
    for r in csv.DictReader(open(m)):
        rows.append(dict(kind='C', name='copy %s %s B' % (r.get('Direction', ''), r.get('Size', '?')), s=int(r['Start_Timestamp']), e=int(r['End_Timestamp']), q=''))
except Exception as e: print('no copies', e)
rows.sort(key=lambda r: r['s'])
idx = [i for i, r in enumerate(rows) if first in r['name']]
ks = [i for i, r in enumerate(rows) if r['kind'] == 'K']
starts = [i for k, i in enumerate(ks) if first in rows[i]['name'] and (k == 0 or first not in rows[ks[k - 1]]['name'])]
i0, i1 = starts[-2], starts[-1]
t0 = rows[i0]['s']
def short(n):
    n = re.sub(r'flamehip::\(anonymous namespace\)::', '', n); n = re.sub(r'^void ', '', n)
    mm = re.search(r'(radix_sort_\w+|merge\w*|onesweep\w*|scan\w*|lookback\w*|histogram\w*|partition\w*|block_sort\w*)', n)
    return ('rocprim:' + mm.group(1)) if 'rocprim' in n and mm else n.split('(')[0][:56]
prev = t0; busy = 0
with open('gpurun_out/frame_timeline_%s.txt' % w, 'w') as o:
    for r in rows[i0:i1]:
        line = '%8.1f us  dur %7.1f  gap %6.1f  %s q%s %s' % ((r['s'] - t0) / 1e3, (r['e'] - r['s']) / 1e3, (r['s'] - prev) / 1e3, r['kind'], r['q'], short(r['name']))
        o.write(line + '\n'); prev = max(prev, r['e']); busy += r['e'] - r['s']
    o.write('busy %.1f us, frame period %.1f us\n' % (busy / 1e3, (rows[i1]['s'] - t0) / 1e3))
print(open('gpurun_out/frame_timeline_%s.txt' % w).read())
PY
