# dev helper (gpurun): kernel trace of ONE plan build (the last upload of tools/upload_time.py), in launch order
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pt -o p -- python tools/upload_time.py $1 $2 > /dev/null 2>&1
python - <<'PY'
import csv, glob, re
f = glob.glob('gpurun_out/pt/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
# the last build = kernels between the last two k_tile bursts: find last k_rcb_init
idx = [i for i, r in enumerate(rows) if 'k_rcb_init' in r['Kernel_Name']]
i0 = idx[-1]
# back up to the rank sorts before it
while i0 > 0 and 'k_tile<' not in rows[i0 - 1]['Kernel_Name'] and 'k_download' not in rows[i0 - 1]['Kernel_Name']: i0 -= 1
t0 = int(rows[i0]['Start_Timestamp'])
def short(n):
    n = re.sub(r'flamehip::\(anonymous namespace\)::', '', n)
    m = re.search(r'(radix_sort_\w+|merge\w*|onesweep\w*|scan\w*|lookback\w*|histogram\w*|partition\w*|block_sort\w*)', n)
    return ('rocprim:' + m.group(1)) if 'rocprim' in n and m else n[:40]
prev_end = t0
tot = 0
for r in rows[i0:]:
    n = r['Kernel_Name']
    if 'k_tile<' in n: break
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('%8.1f us  dur %6.1f  gap %5.1f  %s' % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, short(n)))
    prev_end = e; tot += e - s
print('kernel time %.1f us, span %.1f us' % (tot / 1e3, (prev_end - t0) / 1e3))
PY
rm -rf gpurun_out/pt
