# instruction mix / stalls of the star kernel (rocprofv3 PMC, one pass per counter set)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/dt10k.py <<'PY'
import sys, os
import numpy as np
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from flame_ros_amd.regularizer import GraphRegularizer
h = GraphRegularizer.empty()
rng = np.random.default_rng(0)
pts = (rng.random((10000, 2)) * np.array([640.0, 480.0])).astype(np.float32)
for _ in range(3): h.delaunay(pts)
PY
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM"; do
  rm -rf /tmp/dtpmc
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/dtpmc -o p -- python /tmp/dt10k.py > /tmp/dtpmc.log 2>&1
  f=$(find /tmp/dtpmc -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "k_dt_star" not in k: continue
    key = "pass2" if ("Lb1" in k or "true" in k) else "pass1"
    acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
    n[(key, r["Counter_Name"])] += 1
for key in acc:
    print(key, {c: round(v / n[(key, c)]) for c, v in acc[key].items()})
PY
  else tail -3 /tmp/dtpmc.log; fi
done
