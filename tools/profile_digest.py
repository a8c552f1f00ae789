#!/usr/bin/env python3
"""Turn the raw rocprofv3 output of tools/profile_round.sh (gpurun_out/<round>prof/) into the files committed under
profiles/<round>/ and profiles/traffic.json.

    python tools/profile_digest.py r02

* <name>_kernel_stats.csv   the `rocprofv3 --kernel-trace --stats` summary of `python3 bench.py --no-cpu --no-configs <args>`
* <name>_bench_line.json    the JSON line the same process printed (its roofline.kernel_ms_per_launch comes from HIP events and
                            must agree with the summary's average for the tick kernel)
* pmc/<COUNTER>_<name>.csv  per-dispatch counter rows of the world-tick kernels (separate FETCH_SIZE / WRITE_SIZE passes)
* pmc/sq_<name>.csv         medians of the SQ counters per kernel
* ../traffic.json           HBM bytes per entity and launch: (2 x FETCH_SIZE + WRITE_SIZE) x 1024 / entities (gfx950 correction of
                            MI355X_MICROARCH.md), median of the LAST 5 dispatches of a run (steady state: for the first ~23 ticks after
                            the velocities are seeded the bodies are slower than the sleeping threshold and touch their
                            deactivation records)
"""
import csv
import glob
import json
import os
import shutil
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = ("k_tick", "k_bp_", "k_sort_", "k_scan_", "k_ground", "k_pose_only")


def newest(pattern):
    files = glob.glob(pattern, recursive=True)
    return max(files, key=os.path.getmtime) if files else None


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0].replace("bge::", "")


def counter_rows(path):
    with open(path) as f:
        for r in csv.DictReader(f):
            k = short(r["Kernel_Name"])
            if k.startswith(KERNELS):
                yield dict(Dispatch_Id=int(r["Dispatch_Id"]), Kernel=k, Grid_Size=int(r["Grid_Size"]), VGPR_Count=r["VGPR_Count"],
                           LDS_Block_Size=r["LDS_Block_Size"], Counter_Name=r["Counter_Name"], Counter_Value=float(r["Counter_Value"]),
                           DurationNs=int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else "r02"
    src = os.path.join(ROOT, "gpurun_out", rnd + "prof")
    dst = os.path.join(ROOT, "profiles", rnd)
    os.makedirs(os.path.join(dst, "pmc"), exist_ok=True)
    entities = {"flat1m": 1_000_000, "chains4": 1_000_000, "subtree64": 2_000_000, "cube4m": 4_000_000, "flat1m_basis": 1_000_000, "flat16m": 16_000_000}
    algorithmic = {"flat1m": 140.0, "chains4": 113.0, "subtree64": (140.0 + 63 * 104.0) / 64.0, "flat1m_basis": 140.0 + 0.12 * 44.0}  # basis: steady state; the PMC pass ends 25 ticks after the re-pose, when ~30 % of the bodies still step
    for name in entities:
        stats = newest(os.path.join(src, f"stats_{name}", "**", "*kernel_stats.csv"))
        if stats:
            shutil.copy(stats, os.path.join(dst, f"{name}_kernel_stats.csv"))
        line = os.path.join(src, f"stats_{name}.json")
        if os.path.exists(line):
            txt = open(line).read().strip().splitlines()
            if txt:
                d = json.loads(txt[-1])
                json.dump({k: d[k] for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "config", "roofline")},
                          open(os.path.join(dst, f"{name}_bench_line.json"), "w"), indent=1)
    traffic = {"_provenance": f"profiles/{rnd}/pmc/FETCH_SIZE_*.csv and WRITE_SIZE_*.csv (rocprofv3 --pmc, separate passes, tools/profile_round.sh); "
                              "(2 x FETCH_SIZE + WRITE_SIZE) x 1024 / entities, median of the last 5 dispatches; taken on the kernels of this commit, "
                              "not in the bench run that quotes it"}
    for name in ("flat1m", "chains4", "subtree64", "cube4m", "flat1m_basis"):
        per_kernel = {}
        for counter, tag in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
            path = newest(os.path.join(src, f"pmc_{tag}_{name}", "**", "*counter_collection.csv"))
            if not path:
                continue
            rows = list(counter_rows(path))
            with open(os.path.join(dst, "pmc", f"{counter}_{name}.csv"), "w", newline="") as f:
                wr = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
                wr.writeheader()
                wr.writerows(rows)
            by_kernel = {}
            for r in rows:
                by_kernel.setdefault(r["Kernel"], []).append(r["Counter_Value"])
            for k, v in by_kernel.items():
                per_kernel.setdefault(k, {})[tag] = statistics.median(v[-5:])
        n = entities[name]
        if name == "cube4m":
            tick = {k: v for k, v in per_kernel.items() if "fetch" in v and "write" in v and not k.startswith("k_tick<true, true, false")}
            total = sum((2 * v["fetch"] + v["write"]) * 1024 for v in tick.values())
            traffic[name] = {"hbm_bytes_per_launch": total,
                             "per_entity": {"total": total / n},
                             "per_kernel_MB": {k: {"fetch": 2 * v["fetch"] * 1024 / 1e6, "write": v["write"] * 1024 / 1e6} for k, v in sorted(tick.items())}}
        else:
            ks = [k for k in per_kernel if k.startswith("k_tick") and "fetch" in per_kernel[k] and "write" in per_kernel[k]]
            if not ks:
                continue
            v = per_kernel[ks[0]]
            fetch, write = 2 * v["fetch"] * 1024 / n, v["write"] * 1024 / n
            traffic[name] = {"hbm_bytes_per_launch": (fetch + write) * n,
                             "per_entity": {"fetch_size_x1024": v["fetch"] * 1024 / n, "fetch_corrected_x2": fetch, "write": write, "total": fetch + write,
                                            "algorithmic": algorithmic[name]}}
    json.dump(traffic, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
    # SQ counters: medians per kernel
    for name in ("flat1m", "chains4", "subtree64", "cube4m"):
        acc = {}
        for tag in ("sq1", "sq2"):
            path = newest(os.path.join(src, f"pmc_{tag}_{name}", "**", "*counter_collection.csv"))
            if not path:
                continue
            for r in counter_rows(path):
                acc.setdefault(r["Kernel"], {}).setdefault(r["Counter_Name"], []).append(r["Counter_Value"])
                acc[r["Kernel"]].setdefault("DurationNs_" + tag, []).append(r["DurationNs"])
        if not acc:
            continue
        cols = sorted({c for k in acc.values() for c in k})
        with open(os.path.join(dst, "pmc", f"sq_{name}.csv"), "w", newline="") as f:
            wr = csv.writer(f)
            wr.writerow(["Kernel"] + cols)
            for k, v in sorted(acc.items()):
                wr.writerow([k] + [f"{statistics.median(v[c][-5:]):.6g}" if c in v else "" for c in cols])
    print(json.dumps(traffic, indent=1)[:3000])


def ground(rnd):
    """profiles/<round>/pmc_ground.md: k_ground on 1 M resting bodies — duration from the kernel trace, SQ counters of the same
    launches (the 60 timed ticks = dispatches 150..210 of the kernel in tools/measure_ground.py's resting phase)."""
    src = os.path.join(ROOT, "gpurun_out", rnd + "prof")
    dst = os.path.join(ROOT, "profiles", rnd)
    trace = newest(os.path.join(src, "stats_ground", "**", "*kernel_trace.csv"))
    pmc = newest(os.path.join(src, "pmc_ground", "**", "*counter_collection.csv"))
    if not trace or not pmc:
        return
    dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(trace)) if "k_ground<" in r["Kernel_Name"]]
    rows = {}
    for r in csv.DictReader(open(pmc)):
        if "k_ground<" in r["Kernel_Name"]:
            rows.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(rows)[150:210]
    med = {c: statistics.median(rows[i][c] for i in ids if c in rows[i]) for c in rows[ids[0]]} if len(ids) == 60 else {}
    with open(os.path.join(dst, "pmc_ground.md"), "w") as f:
        f.write("# `k_ground` on 1 M bodies resting on the plane: SQ counters (round's profiling run, `tools/profile_round.sh`)\n\n")
        f.write(f"kernel trace: {len(dur)} launches, the 60 timed ones average {sum(dur[150:210]) / 60e3:.1f} us\n\n| counter | median per launch |\n|---|---|\n")
        for c, v in sorted(med.items()):
            f.write(f"| {c} | {v:.4g} |\n")
        if med.get("SQ_WAVES") and med.get("SQ_INSTS_VALU"):
            f.write(f"\nVALU instructions per wave: {med['SQ_INSTS_VALU'] / med['SQ_WAVES']:.0f}\n")
        if med.get("SQ_BUSY_CYCLES") and med.get("SQ_ACTIVE_INST_VALU"):
            f.write(f"SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES = {med['SQ_ACTIVE_INST_VALU'] / med['SQ_BUSY_CYCLES']:.3f}\n")


if __name__ == "__main__":
    main()
    ground(sys.argv[1] if len(sys.argv) > 1 else "r02")
