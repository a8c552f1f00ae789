#!/bin/bash
# SQ counters of the island kernels on the two-box stacks (or another scenario of tools/measure_islands.py):  bash tools/pmc_islands.sh [scenario]
# Counters in their own pass (never combined with --stats); raw output under gpurun_out/pmc_islands/, a per-kernel table on stdout.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_islands
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SC=${1:-stacks}
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/sq -- python3 $R/tools/measure_islands.py 200000 2000 $SC > $OUT/run.log 2> $OUT/run.err || echo "pmc run failed"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
files = glob.glob(out + "/sq/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for f in files:
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name", "")
        m = re.search(r"(k_island_\w+)(<[^>]*>)?", name)
        if not m:
            continue
        k = m.group(0)
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVES":
            calls[k] += 1
print("| kernel | launches | waves / launch | VALU instr / wave | LDS instr / wave | wave cycles / wave | waiting (SQ_WAIT_ANY / SQ_WAVE_CYCLES) | VALU active (SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES) |")
print("|---|---|---|---|---|---|---|---|")
for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    w = max(c.get("SQ_WAVES", 1.0), 1.0)
    cyc = max(c.get("SQ_WAVE_CYCLES", 1.0), 1.0)
    print(f"| `{k}` | {calls[k]} | {w / max(calls[k], 1):.0f} | {c.get('SQ_INSTS_VALU', 0) / w:.0f} | {c.get('SQ_INSTS_LDS', 0) / w:.0f} | {cyc / w:.0f} | {c.get('SQ_WAIT_ANY', 0) / cyc:.2f} | {c.get('SQ_ACTIVE_INST_VALU', 0) / cyc:.2f} |")
PY
