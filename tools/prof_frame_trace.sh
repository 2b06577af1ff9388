# dev helper (gpurun): kernels and copies of ONE whole frame (the last of tools/frame_trace.py), in launch order
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
rm -rf gpurun_out/ft
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/ft -o p -- python tools/frame_trace.py "$@" > /dev/null 2> gpurun_out/ft_stderr.txt
tail -n 3 gpurun_out/ft_stderr.txt
python - <<'PY'
import csv, glob, re
f = glob.glob('gpurun_out/ft/**/*kernel_trace.csv', recursive=True)[0]
rows = [dict(kind='K', name=r['Kernel_Name'], s=int(r['Start_Timestamp']), e=int(r['End_Timestamp'])) for r in csv.DictReader(open(f))]
m = glob.glob('gpurun_out/ft/**/*memory_copy_trace.csv', recursive=True)
if m:
    for r in csv.DictReader(open(m[0])):
        rows.append(dict(kind='C', name='copy %s %s B' % (r.get('Direction', ''), r.get('Size', '?')), s=int(r['Start_Timestamp']), e=int(r['End_Timestamp'])))
rows.sort(key=lambda r: r['s'])
# the last frame starts at the last k_he_count (device sync) or after the previous frame's last copy
idx = [i for i, r in enumerate(rows) if 'k_he_count' in r['name'] or 'k_mini_plan' in r['name']]
i0 = idx[-1] if idx else 0
if not idx:  # host sync path (single tile): start after the last k_tile burst before the final one
    kt = [i for i, r in enumerate(rows) if 'k_tile<' in r['name']]
    last = kt[-1]
    j = last
    while j > 0 and 'k_tile<' in rows[j - 1]['name']: j -= 1
    i0 = j
    while i0 > 0 and rows[i0 - 1]['kind'] == 'C' and 'HOST_TO_DEVICE' in rows[i0 - 1]['name'].upper(): i0 -= 1
while i0 > 0 and rows[i0 - 1]['kind'] == 'C' and 'DEVICE_TO_HOST' not in rows[i0 - 1]['name'].upper(): i0 -= 1
t0 = rows[i0]['s']
def short(n):
    n = re.sub(r'flamehip::\(anonymous namespace\)::', '', n)
    mm = re.search(r'(radix_sort_\w+|merge\w*|onesweep\w*|scan\w*|lookback\w*|histogram\w*|partition\w*|block_sort\w*)', n)
    return ('rocprim:' + mm.group(1)) if 'rocprim' in n and mm else n[:60]
prev = t0; tot = 0; ntile = 0; tile_t = 0
for r in rows[i0:]:
    if 'k_tile<' in r['name']:
        if not ntile: first = (r['s'] - t0) / 1e3, (r['s'] - prev) / 1e3
        ntile += 1; tile_t += r['e'] - r['s']; prev = r['e']; continue
    if ntile:
        print('%8.1f us  gap %5.1f  ... %d x k_tile, kernel time %.1f us' % (first[0], first[1], ntile, tile_t / 1e3)); ntile = 0; tile_t = 0
    print('%8.1f us  dur %6.1f  gap %5.1f  %s %s' % ((r['s'] - t0) / 1e3, (r['e'] - r['s']) / 1e3, (r['s'] - prev) / 1e3, r['kind'], short(r['name'])))
    prev = r['e']; tot += r['e'] - r['s']
print('non-tile busy %.1f us, span %.1f us' % (tot / 1e3, (prev - t0) / 1e3))
PY
rm -rf gpurun_out/ft
