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
bj = os.path.join(d, "bench.json")
if os.path.exists(bj) and os.path.getsize(bj):
    try:
        b = json.loads(open(bj).read().strip().splitlines()[-1])
        lines.append("\n## bench.py line (un-profiled run)\n\n```json\n%s\n```" % json.dumps(b, indent=1))
    except Exception as e:  # noqa
        lines.append("\n(bench.json unreadable: %s)" % e)
json.dump({"traffic_bytes_per_launch": summary, "kernels": [dict(zip(
    ("kernel", "calls", "avg_us", "min_us", "median_us", "max_us", "total_ms"), r)) for r in rows]},
    open(os.path.join(d, "summary.json"), "w"), indent=1)
open(os.path.join(d, "summary.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
