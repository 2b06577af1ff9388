"""Summarises a profiles/collect.sh output directory: per-kernel duration stats from the
rocprofv3 kernel trace, HBM traffic per launch from the FETCH_SIZE / WRITE_SIZE passes (gfx950
correction: FETCH_SIZE counts 64 B per 128-B request on wide coalesced reads -> x2; counters are in
KiB; WRITE_SIZE is uncalibrated -- MI355X_MICROARCH.md "HBM"), and the bench line beside them."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

d = sys.argv[1]


def find(sub, pat):
    f = glob.glob(os.path.join(d, sub, "**", pat), recursive=True)
    return f[0] if f else None


lines = []
kt = find("trace", "*kernel_trace.csv")
stats = defaultdict(list)
if kt:
    for r in csv.DictReader(open(kt)):
        stats[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
rows = []
for k, v in sorted(stats.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    rows.append((k, len(v), sum(v) / len(v), v[0], v[len(v) // 2], v[-1], sum(v) / 1e3))
with open(os.path.join(d, "kernel_stats.csv"), "w") as f:
    f.write("kernel,calls,avg_us,min_us,median_us,max_us,total_ms\n")
    for r in rows:
        f.write('"%s",%d,%.3f,%.3f,%.3f,%.3f,%.3f\n' % r)
lines.append("## rocprofv3 --kernel-trace --stats (python bench.py)\n")
lines.append("| kernel | calls | avg us | min | median | max | total ms |\n|---|---|---|---|---|---|---|")
for r in rows:
    lines.append("| `%s` | %d | %.2f | %.2f | %.2f | %.2f | %.2f |" % ((r[0][:70],) + r[1:]))

traffic = {}
for name, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    f = find(sub, "*counter_collection.csv")
    acc, cnt = defaultdict(float), defaultdict(int)
    if f:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                acc[r["Kernel_Name"]] += float(r["Counter_Value"])
                cnt[r["Kernel_Name"]] += 1
    for k in acc:
        traffic.setdefault(k, {})[name] = acc[k] / max(cnt[k], 1)
lines.append("\n## HBM traffic per launch (separate --pmc passes)\n")
lines.append("| kernel | FETCH_SIZE KiB (raw) | fetch bytes (x2 gfx950 correction) | WRITE_SIZE KiB (raw, uncalibrated) | total MB/launch |\n|---|---|---|---|---|")
summary = {}
for k, t in traffic.items():
    fe, wr = t.get("FETCH_SIZE", 0.0), t.get("WRITE_SIZE", 0.0)
    total = (2 * fe + wr) * 1024
    summary[k] = total
    lines.append("| `%s` | %.1f | %.0f | %.1f | %.3f |" % (k[:70], fe, 2 * fe * 1024, wr, total / 1e6))
# LDS counters of the dominant kernel (one separate PMC pass), per launch and per CU
lds = {}
f = find("pmc_lds", "*counter_collection.csv")
if f:
    acc, disp = defaultdict(lambda: defaultdict(float)), defaultdict(set)
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[r["Kernel_Name"]].add(r.get("Dispatch_Id", r.get("Correlation_Id", "")))
    lines.append("\n## LDS counters (separate --pmc pass), per launch, summed over the chip\n")
    lines.append("| kernel | launches | SQ_LDS_IDX_ACTIVE | SQ_LDS_BANK_CONFLICT | conflict share | SQ_INSTS_LDS | SQ_WAIT_INST_LDS | SQ_ACTIVE_INST_LDS | SQ_WAVE_CYCLES | SQ_BUSY_CYCLES |\n|---|---|---|---|---|---|---|---|---|---|")
    for k, c in acc.items():
        n = max(len(disp[k]), 1)
        if c.get("SQ_INSTS_LDS", 0) <= 0:
            continue
        per = {name: v / n for name, v in c.items()}
        lds[k] = dict(per, launches=n)
        act = per.get("SQ_LDS_IDX_ACTIVE", 0.0)
        lines.append("| `%s` | %d | %.4g | %.4g | %.2f | %.4g | %.4g | %.4g | %.4g | %.4g |" % (
            k[:60], n, act, per.get("SQ_LDS_BANK_CONFLICT", 0), per.get("SQ_LDS_BANK_CONFLICT", 0) / max(act, 1),
            per.get("SQ_INSTS_LDS", 0), per.get("SQ_WAIT_INST_LDS", 0), per.get("SQ_ACTIVE_INST_LDS", 0),
            per.get("SQ_WAVE_CYCLES", 0), per.get("SQ_BUSY_CYCLES", 0)))

# hash of the sources the counters belong to: bench.py only quotes them while it still matches
import hashlib
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
h = hashlib.sha256()
for rel in ("flame_ros_amd/csrc/kernels.hip", "flame_ros_amd/csrc/kernels.h", "flame_ros_amd/csrc/common.h",
            "flame_ros_amd/csrc/plan.cpp"):
    h.update(open(os.path.join(root, rel), "rb").read())
src_sha = h.hexdigest()[:16]
bj = os.path.join(d, "bench.json")
if os.path.exists(bj) and os.path.getsize(bj):
    try:
        b = json.loads(open(bj).read().strip().splitlines()[-1])
        lines.append("\n## bench.py line (un-profiled run)\n\n```json\n%s\n```" % json.dumps(b, indent=1))
    except Exception as e:  # noqa
        lines.append("\n(bench.json unreadable: %s)" % e)
json.dump({"kernel_src_sha": src_sha, "lds_per_launch": lds, "traffic_bytes_per_launch": summary, "kernels": [dict(zip(
    ("kernel", "calls", "avg_us", "min_us", "median_us", "max_us", "total_ms"), r)) for r in rows]},
    open(os.path.join(d, "summary.json"), "w"), indent=1)
open(os.path.join(d, "summary.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
