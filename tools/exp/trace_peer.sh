cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT/.stage 2>/dev/null || cd $GRAFT_REPO_ROOT
export PYTHONPATH=.
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/peer_trace -o p -- python tools/exp/peer_transport_ab.py 50k 2 48 > gpurun_out/peer_trace.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/peer_trace/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
# find the last 60 kernels around a push
idx = [i for i, r in enumerate(rows) if 'k_halo_push' in r['Kernel_Name']]
i0 = idx[len(idx)//2]
prev_end = None
with open('gpurun_out/r06_peer_trace_window.txt', 'w') as out:
    for r in rows[i0-6:i0+8]:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        name = r['Kernel_Name'].replace('flamehip::(anonymous namespace)::','')[:60]
        out.write("%-62s start +%8.1f us  dur %8.1f us  gap %7.1f us  grid %s wg %s\n" % (name, (s-int(rows[i0-6]['Start_Timestamp']))/1e3, (e-s)/1e3, (s-prev_end)/1e3 if prev_end else 0, r.get('Grid_Size_X', r.get('Grid_Size','?')), r.get('Workgroup_Size_X', r.get('Workgroup_Size','?'))))
        prev_end = e
print(open('gpurun_out/r06_peer_trace_window.txt').read())
PY
rm -rf gpurun_out/peer_trace
