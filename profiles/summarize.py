"""Summarises a profiles/collect.sh output directory: per-kernel duration stats from the
rocprofv3 kernel trace, HBM traffic per launch from the FETCH_SIZE / WRITE_SIZE passes (gfx950
correction: FETCH_SIZE counts 64 B per 128-B request on wide coalesced reads -> x2; counters are in
KiB; WRITE_SIZE is uncalibrated -- MI355X_MICROARCH.md "HBM"), and the bench line beside them.

Also home of the ON-CHIP roofline of the tile kernels: `lds_floor` counts, from the tile plan,
the LDS wave-instructions ONE PD iteration must issue for the vertices and edges a tile OWNS (no halo work), priced with
the MI355X guide's LDS table; `lds_roofline` turns that floor, a measured time per iteration and (optionally) the
SQ_* counters of a --pmc pass into the `roofline` block bench.py prints.

    python profiles/summarize.py gpurun_out/<tag>                    # summarise a collect.sh directory
    python profiles/summarize.py --roofline profiles/rNN_X_summary.json   # recompute the block from a committed summary
"""
import csv
import glob
import hashlib
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# MI355X_MICROARCH.md, LDS table: cycles per wave-instruction (conflict-free)
CYC_READ_B128 = 4    # ds_read_b128: 4 lane groups x 1 LDS cycle
CYC_WRITE_B96 = 10   # ds_write_b96: bound by the 4-dword operand transfer (LDS array: 8)
TILE_DESC_WORDS = 13 + 17 + 17  # csrc/common.h TileDesc
# r06, the second on-chip resource: VALU issue.  Wave-instructions of the tile kernel's iteration, counted in the gfx950 ISA of
# k_tile_persist<512,2,1> (hipcc -S): phase D per 64-edge block 10 plain + 6 packed fp32 instructions,
# phase P per 64-vertex block 14 plain + 2 packed (prox, clamp, extrapolation) and 1 plain + 1 packed per incidence slot of
# the longest row.  Priced 4 cycles per wave64 instruction per SIMD, plain or packed -- what the counters of this kernel say
# (profiles/r06_*_summary: SQ_ACTIVE_INST_VALU, in quad-cycles, equals SQ_INSTS_VALU to three digits) --; a CU has 4 SIMDs.
VALU_D_PLAIN, VALU_D_PACKED = 10, 6
VALU_P_PLAIN, VALU_P_PACKED = 14, 2
VALU_SLOT_PLAIN, VALU_SLOT_PACKED = 1, 1
CYC_VALU_PLAIN, CYC_VALU_PACKED, SIMDS_PER_CU = 4, 4, 4


def lds_floor(tiles, srow, num_cus=256):
    """LDS issue floor of ONE PD iteration, from the tile plan (flame_hip_debug_plan_array "tiles", "t_srow"): what the
    kernel's own instruction mix must issue for the vertices / edges every tile OWNS, nothing for halos.

      phase D, per 64-edge block of OWN edges : 2 x ds_read_b128 (x_bar gathers of both endpoints) + 2 x ds_write_b96
                                                 (the -K^T q terms into the endpoints' incidence slots)   = 28 cycles
      phase P, per 64-vertex block of OWN vertices: max degree of the block x ds_read_b128 (its incidence slots, the
                                                 wave reads the longest row's length) + 1 x ds_write_b96 (x_bar)

    tiles: int32 array (ntiles, 47) of TileDesc words; srow: uint32 {slot | degree << 16} per updated local vertex.
    Returns cycles and wave-instructions per iteration for the slowest CU (tiles are dealt to CUs round-robin when there
    are more tiles than CUs -- launches; resident tiles are one per CU) and the mean over tiles."""
    import numpy as np
    tiles = np.asarray(tiles, np.int64).reshape(-1, TILE_DESC_WORDS)
    srow = np.asarray(srow, np.uint32)
    cyc, ins = [], []
    for t in tiles:
        n_own, e_own, srow_off = int(t[1]), int(t[4]), int(t[11])
        eb = -(-e_own // 64)
        c = eb * (2 * CYC_READ_B128 + 2 * CYC_WRITE_B96)
        i = eb * 4
        deg = (srow[srow_off:srow_off + n_own] >> 16).astype(np.int64)
        for b in range(0, n_own, 64):
            m = int(deg[b:b + 64].max()) if n_own else 0
            c += m * CYC_READ_B128 + CYC_WRITE_B96
            i += m + 1
        cyc.append(c)
        ins.append(i)
    cyc, ins = np.asarray(cyc, np.float64), np.asarray(ins, np.float64)
    nt = len(cyc)
    if nt == 0:
        return None
    per_cu = np.zeros(min(nt, num_cus))
    for k in range(nt):  # (b % CUs: how a grid larger than the chip is dealt out; one tile per CU otherwise)
        per_cu[k % len(per_cu)] += cyc[k]
    # VALU issue floor of the same OWN work, spread perfectly over the CU's four SIMDs (r06)
    vcyc, vins = [], []
    for t in tiles:
        n_own, e_own, srow_off = int(t[1]), int(t[4]), int(t[11])
        eb = -(-e_own // 64)
        plain, packed = eb * VALU_D_PLAIN, eb * VALU_D_PACKED
        deg = (srow[srow_off:srow_off + n_own] >> 16).astype(np.int64)
        for b in range(0, n_own, 64):
            m = int(deg[b:b + 64].max())
            plain += VALU_P_PLAIN + m * VALU_SLOT_PLAIN
            packed += VALU_P_PACKED + m * VALU_SLOT_PACKED
        vcyc.append((plain * CYC_VALU_PLAIN + packed * CYC_VALU_PACKED) / float(SIMDS_PER_CU))
        vins.append(plain + packed)
    vper_cu = np.zeros(len(per_cu))
    for k in range(nt):
        vper_cu[k % len(vper_cu)] += vcyc[k]
    return {"floor_cycles_per_iteration_slowest_cu": float(per_cu.max()),
            "valu_floor_cycles_per_iteration_slowest_cu": float(vper_cu.max()),
            "useful_valu_insts_per_iteration_chip": float(np.sum(vins)),
            "valu_pricing": {"plain_wave64": CYC_VALU_PLAIN, "packed_wave64": CYC_VALU_PACKED, "simds_per_cu": SIMDS_PER_CU,
                             "phase_d_per_64_edges": [VALU_D_PLAIN, VALU_D_PACKED], "phase_p_per_64_vertices": [VALU_P_PLAIN, VALU_P_PACKED],
                             "per_slot": [VALU_SLOT_PLAIN, VALU_SLOT_PACKED],
                             "source": "ISA of k_tile_persist<512,2,1>; 4 cycles per wave64 VALU instruction (SQ_ACTIVE_INST_VALU = SQ_INSTS_VALU quad-cycles)"},
            "floor_cycles_per_iteration_mean_tile": float(cyc.mean()),
            "useful_lds_insts_per_iteration_chip": float(ins.sum()),
            "num_tiles": int(nt), "busy_cus": int(len(per_cu)),
            "pricing": {"ds_read_b128": CYC_READ_B128, "ds_write_b96": CYC_WRITE_B96,
                        "source": "MI355X_MICROARCH.md LDS table (cycles per wave-instruction)"}}


def lds_roofline(floor, us_per_iteration, clock_mhz, iterations_per_launch, lds_counters=None, iterate_frac=None,
                 measured_hbm_frac=None, contract_frac=None, valu_counters=None):
    """The `roofline` block of a tile-path bench line.  bound = "lds": frac = floor cycles of the slowest CU / measured
    shader cycles per iteration -- <= 1 by construction (the kernel issues at least the floor's instructions).  The
    counters (one launch, summed over the chip: SQ_INSTS_LDS, SQ_LDS_IDX_ACTIVE, SQ_LDS_BANK_CONFLICT) give, beside it,
    how much LDS work was executed per useful instruction (halo redundancy + padding lanes), how busy the LDS array was
    and what share of that was bank conflicts."""
    meas = us_per_iteration * clock_mhz
    lds_f = floor["floor_cycles_per_iteration_slowest_cu"]
    valu_f = floor.get("valu_floor_cycles_per_iteration_slowest_cu", 0.0)
    # r06: two on-chip resources, the larger floor is the bound
    top, bound = (lds_f, "lds") if lds_f >= valu_f else (valu_f, "valu")
    frac = top / max(meas, 1e-9)
    out = {"bound": bound,
           "achieved": top / max(us_per_iteration, 1e-12) / 1e3,
           "peak": clock_mhz / 1e3, "unit": "Gcycle/s of %s issue on the slowest CU (useful work only)" % bound.upper(),
           "frac": frac,
           "lds_frac": lds_f / max(meas, 1e-9), "valu_frac": valu_f / max(meas, 1e-9),
           "floor_cycles_per_iteration": top,
           "lds_floor_cycles_per_iteration": lds_f, "valu_floor_cycles_per_iteration": valu_f,
           "floor_cycles_per_iteration_mean_tile": floor["floor_cycles_per_iteration_mean_tile"],
           "measured_cycles_per_iteration": meas, "clock_mhz": clock_mhz,
           "floor_pricing": floor["pricing"]}
    if lds_counters:
        cus = float(floor["busy_cus"])
        its = float(max(iterations_per_launch, 1))
        insts = lds_counters.get("SQ_INSTS_LDS", 0.0)
        act = lds_counters.get("SQ_LDS_IDX_ACTIVE", 0.0)
        conf = lds_counters.get("SQ_LDS_BANK_CONFLICT", 0.0)
        out["work_redundancy"] = insts / its / max(floor["useful_lds_insts_per_iteration_chip"], 1.0)
        out["lds_busy"] = act / cus / its / max(meas, 1e-9)
        out["bank_conflict_share"] = conf / max(act, 1.0)
        out["executed_lds_insts_per_iteration_per_cu"] = insts / its / cus
        out["lds_array_cycles_per_iteration_per_cu"] = act / its / cus
    if valu_counters:
        # SQ_* "cycle" counters are in quad-cycles, summed over the waves (guide, rocprofv3 PMC slots): shares of the waves' time
        wc = max(valu_counters.get("SQ_WAVE_CYCLES", 0.0), 1.0)
        its = float(max(iterations_per_launch, 1))
        cus = float(floor["busy_cus"])
        out["valu_insts_per_iteration_per_cu"] = valu_counters.get("SQ_INSTS_VALU", 0.0) / its / cus
        out["valu_redundancy"] = valu_counters.get("SQ_INSTS_VALU", 0.0) / its / max(floor.get("useful_valu_insts_per_iteration_chip", 1.0), 1.0)
        out["wave_time_shares"] = {
            "waiting (s_waitcnt / barrier: SQ_WAIT_ANY)": valu_counters.get("SQ_WAIT_ANY", 0.0) / wc,
            "issue stall (SQ_WAIT_INST_ANY)": valu_counters.get("SQ_WAIT_INST_ANY", 0.0) / wc,
            "issuing (SQ_ACTIVE_INST_ANY)": valu_counters.get("SQ_ACTIVE_INST_ANY", 0.0) / wc,
            "of which VALU (SQ_ACTIVE_INST_VALU)": valu_counters.get("SQ_ACTIVE_INST_VALU", 0.0) / wc}
        # VALU busy of a SIMD: quad-cycles a VALU instruction was issuing x 4 / (SIMD-cycles of the launch)
        out["valu_busy"] = valu_counters.get("SQ_ACTIVE_INST_VALU", 0.0) * 4.0 / (cus * SIMDS_PER_CU * its * max(meas, 1e-9))
    if iterate_frac is not None:
        out["handoff_share"] = 1.0 - iterate_frac
    if measured_hbm_frac is not None:
        out["measured_hbm_frac"] = measured_hbm_frac
    if contract_frac is not None:
        out["contract_frac"] = contract_frac
    return out


def kernel_src_sha():
    """hash of the sources the counters belong to: bench.py only quotes them while it still matches"""
    h = hashlib.sha256()
    for rel in ("flame_ros_amd/csrc/kernels.hip", "flame_ros_amd/csrc/kernels.h", "flame_ros_amd/csrc/common.h",
                "flame_ros_amd/csrc/plan.cpp"):
        h.update(open(os.path.join(ROOT, rel), "rb").read())
    return h.hexdigest()[:16]


def short_name(k, n=70):
    """A printable kernel name: template arguments of library kernels (rocprim's run to 1 KB) are cut, never the whole
    name (r04's delaunay summary had blank rows: names that were nothing but a cut-off template head)."""
    k = (k or "").strip() or "(unnamed kernel)"
    k = k.replace("flamehip::(anonymous namespace)::", "").replace("void ", "")
    if len(k) > n:
        head = k.split("(")[0]
        k = (head if 0 < len(head) <= n else k[:n - 3]) + "..."
    return k


def roofline_from_summary(path):
    """Recompute the LDS roofline block from a committed summary (it carries the plan's floor, the bench line's time per
    iteration and the counters of the PMC pass)."""
    s = json.load(open(path))
    b, fl = s.get("bench") or {}, s.get("lds_floor")
    if not fl or not b:
        raise SystemExit("%s holds no lds_floor / bench line (a summary of r05 or later is needed)" % path)
    rl = b["roofline"]
    kern = rl.get("kernel", "k_tile")
    cnt, vcnt, hbm = None, None, rl.get("measured_hbm_frac")
    for k, v in s.get("lds_per_launch", {}).items():
        if kern + "<" in k:
            cnt = v
    for k, v in s.get("valu_per_launch", {}).items():
        if kern + "<" in k:
            vcnt = v
    for k, v in s.get("traffic_bytes_per_launch", {}).items():  # this summary's own FETCH x2 + WRITE passes
        if kern + "<" in k and v > 0:
            hbm = v / (rl["launch_us"] * 1e-6) / 1e9 / 8000.0
    out = lds_roofline(fl, rl["launch_us"] / rl["iters_per_launch"], rl["clock_mhz"], rl["iters_per_launch"], cnt,
                       rl.get("iterate_frac"), hbm, (rl.get("contract") or {}).get("frac", rl.get("contract_frac")), vcnt)
    out["kernel"] = kern
    return out


def summarize(d):
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
        lines.append("| `%s` | %d | %.2f | %.2f | %.2f | %.2f | %.2f |" % ((short_name(r[0]),) + r[1:]))

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
        lines.append("| `%s` | %.1f | %.0f | %.1f | %.3f |" % (short_name(k), fe, 2 * fe * 1024, wr, total / 1e6))
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
                short_name(k, 60), n, act, per.get("SQ_LDS_BANK_CONFLICT", 0), per.get("SQ_LDS_BANK_CONFLICT", 0) / max(act, 1),
                per.get("SQ_INSTS_LDS", 0), per.get("SQ_WAIT_INST_LDS", 0), per.get("SQ_ACTIVE_INST_LDS", 0),
                per.get("SQ_WAVE_CYCLES", 0), per.get("SQ_BUSY_CYCLES", 0)))

    # VALU / issue-stall counters (r06: their own PMC pass), per launch, summed over the chip
    valu = {}
    f = find("pmc_valu", "*counter_collection.csv")
    if f:
        acc, disp = defaultdict(lambda: defaultdict(float)), defaultdict(set)
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[r["Kernel_Name"]].add(r.get("Dispatch_Id", r.get("Correlation_Id", "")))
        names = ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_SALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES")
        lines.append("\n## VALU / wait counters (separate --pmc pass), per launch, summed over the chip (cycle counters in quad-cycles)\n")
        lines.append("| kernel | launches | " + " | ".join(names) + " | waiting | issue stall | issuing | VALU |\n|---|---|" + "---|" * (len(names) + 4))
        for k, c in acc.items():
            n = max(len(disp[k]), 1)
            if c.get("SQ_INSTS_VALU", 0) <= 0:
                continue
            per = {name: v / n for name, v in c.items()}
            valu[k] = dict(per, launches=n)
            wc = max(per.get("SQ_WAVE_CYCLES", 0.0), 1.0)
            lines.append("| `%s` | %d | %s | %.2f | %.2f | %.2f | %.2f |" % (
                short_name(k, 60), n, " | ".join("%.4g" % per.get(x, 0.0) for x in names), per.get("SQ_WAIT_ANY", 0) / wc,
                per.get("SQ_WAIT_INST_ANY", 0) / wc, per.get("SQ_ACTIVE_INST_ANY", 0) / wc, per.get("SQ_ACTIVE_INST_VALU", 0) / wc))

    out = {"kernel_src_sha": kernel_src_sha(), "lds_per_launch": lds, "valu_per_launch": valu, "traffic_bytes_per_launch": summary, "kernels": [dict(zip(
        ("kernel", "calls", "avg_us", "min_us", "median_us", "max_us", "total_ms"), r)) for r in rows]}
    bj = os.path.join(d, "bench.json")
    if os.path.exists(bj) and os.path.getsize(bj):
        try:
            b = json.loads(open(bj).read().strip().splitlines()[-1])
            out["bench"] = {k: b.get(k) for k in ("metric", "value", "unit", "ms_per_step", "us_per_iteration", "config", "roofline")}
            out["lds_floor"] = (b.get("roofline") or {}).get("lds_floor")
            lines.append("\n## bench.py line (un-profiled run)\n\n```json\n%s\n```" % json.dumps(b, indent=1))
        except Exception as e:  # noqa
            lines.append("\n(bench.json unreadable: %s)" % e)
    json.dump(out, open(os.path.join(d, "summary.json"), "w"), indent=1)
    if out.get("lds_floor") and out.get("bench"):
        try:  # the on-chip roofline from THIS directory's counters (the un-profiled line quotes an older committed pass)
            rl = roofline_from_summary(os.path.join(d, "summary.json"))
            out["roofline_from_these_counters"] = rl
            json.dump(out, open(os.path.join(d, "summary.json"), "w"), indent=1)
            lines.append("\n## on-chip (LDS) roofline of the dominant kernel, from the counters above\n\n```json\n%s\n```" % json.dumps(rl, indent=1))
        except (Exception, SystemExit) as e:  # noqa
            lines.append("\n(LDS roofline not computed: %s)" % e)
    open(os.path.join(d, "summary.md"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "--roofline":
        print(json.dumps(roofline_from_summary(sys.argv[2]), indent=1))
    else:
        summarize(sys.argv[1])
