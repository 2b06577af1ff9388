#!/bin/bash
# r06: kernel timeline of one keep-mode frame (delaunay + graph sync + 50 iterations) with T handed out after / before the stars
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT/.stage" 2>/dev/null || cd "$(dirname "$0")/../.."
export PYTHONPATH=.
n=${1:-1200}
for e in 0 1; do
  rm -rf gpurun_out/early_trace_$e
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/early_trace_$e -o p -- python tools/exp/early_T_laps.py $n $e noplan > /dev/null 2>&1
  python - <<PY
import csv, glob
f = glob.glob("gpurun_out/early_trace_$e/**/p_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last frame: from the last k_dt_prep* / k_dt_init to the end
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith(("k_dt_prep_small", "flamehip::(anonymous namespace)::k_dt_prep_small", "k_dt_init")) or "k_dt_prep_small" in r["Kernel_Name"] or "k_dt_init" in r["Kernel_Name"]]
i0 = idx[-1]
t0 = int(rows[i0]["Start_Timestamp"])
print("== early $e, V $n: kernel, start us, duration us (last frame)")
for r in rows[i0:i0 + 14]:
    print("  %-46s %8.1f %8.1f" % (r["Kernel_Name"].split("(")[0][-46:], (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
  rm -rf gpurun_out/early_trace_$e
done
