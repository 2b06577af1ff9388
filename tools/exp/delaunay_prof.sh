# per-kernel times of flame_hip_delaunay (rocprofv3 kernel trace): per-dispatch durations of the star kernels in call order
mkdir -p gpurun_out/r04_dt
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/dtprof
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dtprof -o dt -- python $R/tools/exp/delaunay_gpu_time.py > /tmp/dtprof.log 2>&1
f=$(find /tmp/dtprof -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" $R/gpurun_out/r04_dt/kernel_stats.csv; fi
t=$(find /tmp/dtprof -name "*kernel_trace.csv" | head -1)
if [ -n "$t" ]; then python - "$t" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
star = [r for r in rows if "k_dt_star" in r["Kernel_Name"]]
# 23 calls per configuration (3 warm-up + 20 timed), 2 star launches per call
per = 46
for c in range(len(star) // per):
    grp = star[c * per:(c + 1) * per]
    a = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in grp if "false" in r["Kernel_Name"] or "Lb0" in r["Kernel_Name"]]
    b = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in grp if not ("false" in r["Kernel_Name"] or "Lb0" in r["Kernel_Name"])]
    print("config %d: star pass 1 %.1f us (min), pass 2 %.1f us; grid %s" % (c, min(a) if a else -1, min(b) if b else -1, grp[0].get("Grid_Size_X", "?")))
PY
fi
