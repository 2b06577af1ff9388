# (gpurun) rocprofv3 kernel stats of the FRAME path: the C++ facade bench of one workload under
# --kernel-trace --stats; writes gpurun_out/r04_frames_<w>_kernel_stats.csv and a short .md
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
for w in "$@"; do
  out=gpurun_out/pf_$w; rm -rf $out
  rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python tools/facade_bench.py --workloads $w --repeats 25 > gpurun_out/pf_$w.log 2>&1
  f=$(find $out -name '*kernel_stats.csv' | head -1)
  cp "$f" gpurun_out/r04_frames_${w}_kernel_stats.csv
  python - "$w" "$f" <<'PY'
import csv, sys, json
w, f = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(f)))
frames = 4 * (25 + 3)
upd = None
for line in open('gpurun_out/pf_%s.log' % w).read().splitlines():  # (rocprofv3 logs behind the bench's JSON line)
    if line.startswith('{"'):
        try: upd = json.loads(line)[w]['update_ms']['p50']
        except Exception: pass
with open('gpurun_out/r04_frames_%s_summary.md' % w, 'w') as o:
    o.write('## rocprofv3 --kernel-trace --stats (python tools/facade_bench.py --workloads %s --repeats 25): %d frames of flame::Flame::update, p50 %s ms under the profiler\n\n' % (w, frames, upd))
    o.write('| kernel | calls | calls per frame | avg us | total ms |\n|---|---|---|---|---|\n')
    for r in rows[:28]:
        n = int(r['Calls'])
        o.write('| `%s` | %d | %.2f | %.2f | %.2f |\n' % (r['Name'][:70], n, n / frames, float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6))
print(open('gpurun_out/r04_frames_%s_summary.md' % w).read()[:1500])
PY
done
