#!/usr/bin/env python3
"""Summarises a tools/profile.sh output directory.

    python tools/pmc_summary.py gpurun_out/prof_<tag> [--update profiles/counters.json] [--round r3]

Prints, for the draw kernel of that run: the rocprofv3 --stats table, the average duration of the
TIMED launches only (the bench line in trace.log says how many warm-up and clock-ramp launches
precede them), and the per-launch average of every counter over the same launches.  With --update
the derived figures bench.py reports (HBM bytes per launch, VALU busy, LDS bank conflicts) are
merged into the JSON file under the key "<config>|<shading>|<variant>".
"""
import collections
import csv
import glob
import json
import os
import sys


def bench_line(log):
    line = None
    try:
        with open(log, errors="replace") as f:
            for ln in f:
                ln = ln.strip()
                if ln.startswith('{"metric"'):
                    line = json.loads(ln)
    except OSError:
        pass
    return line


def timed_slice(n_launches, line):
    """Indices of the timed launches among the draw-kernel launches of one bench run."""
    if not line:
        return slice(0, n_launches)
    first = line["warmup"] + line.get("warmup_clock_ramp_steps", 0)
    return slice(first, first + line["steps"])


def main():
    args = sys.argv[1:]
    d = args[0]
    update = args[args.index("--update") + 1] if "--update" in args else None
    rnd = args[args.index("--round") + 1] if "--round" in args else "r3"
    workload = open(os.path.join(d, "workload.txt")).read().strip() if os.path.exists(os.path.join(d, "workload.txt")) else None
    print(f"# workload: {workload}   directory: {d}")

    for f in sorted(glob.glob(d + "/trace/*/*_kernel_stats.csv")):
        print(open(f).read().strip())
    line = bench_line(os.path.join(d, "trace.log"))
    derived = {}
    for f in sorted(glob.glob(d + "/trace/*/*_kernel_trace.csv")):
        rows = [r for r in csv.DictReader(open(f)) if "rc_draw" in r["Kernel_Name"]]
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
        sl = timed_slice(len(dur), line)
        t = dur[sl]
        if t:
            name = rows[sl.start]["Kernel_Name"]
            print(f"# timed launches only: {name}: launches [{sl.start}, {sl.start + len(t)}) of {len(dur)}: "
                  f"avg {sum(t) / len(t):.1f} ns  min {min(t)} ns  max {max(t)} ns")
            derived["kernel"] = name
            derived["trace_timed_avg_ms"] = sum(t) / len(t) / 1e6
            derived["trace_all_avg_ms"] = sum(dur) / len(dur) / 1e6
        if line:
            print(f"# bench line of the traced run: ms_per_step {line['ms_per_step']:.4f}  frame_ms {line['frame_ms']:.4f}  "
                  f"(warmup {line['warmup']} + clock ramp {line.get('warmup_clock_ramp_steps', 0)} + timed {line['steps']} + 100 per-frame launches)")
            derived["bench_frame_ms_under_trace"] = line["frame_ms"]

    avg = {}
    for k, f in enumerate(sorted(glob.glob(d + "/pmc*/*/*_counter_collection.csv")), 1):
        pline = bench_line(os.path.join(d, os.path.basename(os.path.dirname(os.path.dirname(f))) + ".log"))
        per = collections.defaultdict(dict)        # dispatch -> counter -> value
        names, times = {}, {}
        for r in csv.DictReader(open(f)):
            if "rc_draw" in r["Kernel_Name"]:
                did = int(r["Dispatch_Id"])
                per[did][r["Counter_Name"]] = per[did].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                names[did] = r["Kernel_Name"]
                times[did] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        ids = sorted(per)
        sl = timed_slice(len(ids), pline)
        ids = ids[sl]
        if not ids:
            continue
        acc = collections.defaultdict(list)
        for did in ids:
            for c, v in per[did].items():
                acc[c].append(v)
        pass_ms = sum(times[i] for i in ids) / len(ids) / 1e6
        for c, v in sorted(acc.items()):
            avg[c] = sum(v) / len(v)
            print(f"{names[ids[0]]:44s} {c:26s} n={len(v):3d} avg={avg[c]:.1f}")
        if "GRBM_GUI_ACTIVE" in acc:
            avg["_grbm_pass_ms"] = pass_ms

    if "TCC_EA0_WRREQ_sum" in avg:
        derived["hbm_write_bytes"] = int(round(avg["TCC_EA0_WRREQ_sum"] * 64))
    if "TCC_EA0_RDREQ_sum" in avg:
        # gfx950 tallies 128-byte read requests at 64 B: doubled
        derived["hbm_read_bytes"] = int(round(avg["TCC_EA0_RDREQ_sum"] * 64 * 2))
    if "SQ_ACTIVE_INST_VALU" in avg and "GRBM_GUI_ACTIVE" in avg:
        # SQ_ACTIVE_INST_VALU counts quad-cycles of VALU issue summed over all SIMDs;
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs
        # (for launches shorter than ~0.1 ms GRBM_GUI_ACTIVE also covers the gaps around the kernel, so
        # valu_busy is then a lower bound and no clock can be derived from it)
        derived["valu_busy"] = avg["SQ_ACTIVE_INST_VALU"] * 4 / (avg["GRBM_GUI_ACTIVE"] / 8 * 1024)
        derived["effective_clock_ghz"] = (avg["GRBM_GUI_ACTIVE"] / 8 / (avg["_grbm_pass_ms"] * 1e6)
                                          if avg["_grbm_pass_ms"] >= 0.1 else None)
    for c, key in (("SQ_LDS_BANK_CONFLICT", "lds_bank_conflict"), ("SQ_INSTS_VALU", "insts_valu"),
                   ("SQ_INSTS_LDS", "insts_lds"), ("SQ_WAVES", "waves")):
        if c in avg:
            derived[key] = avg[c]
    derived["source"] = f"profiles/{rnd}_pmc_{(workload or 'unknown').replace('|', '_')}.txt"
    print("# derived:", json.dumps(derived))

    if update and workload:
        try:
            with open(update) as f:
                table = json.load(f)
        except (OSError, ValueError):
            table = {}
        table.setdefault("_comment", "Per-launch counter figures of the draw kernel, one entry per bench workload "
                         "'<config>|<shading>|<variant>', from tools/profile.sh (separate rocprofv3 --pmc passes) summarised "
                         "by tools/pmc_summary.py over the TIMED launches only. hbm_write_bytes = TCC_EA0_WRREQ_sum x 64 B; "
                         "hbm_read_bytes = TCC_EA0_RDREQ_sum x 64 B x 2 (gfx950 tallies 128-byte read requests at 64 B); "
                         "valu_busy = SQ_ACTIVE_INST_VALU x 4 / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs).")
        table[workload] = derived
        with open(update, "w") as f:
            json.dump(table, f, indent=1, sort_keys=True)
            f.write("\n")


if __name__ == "__main__":
    main()
