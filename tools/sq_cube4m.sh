#!/bin/bash
# SQ counters of the broadphase tick's kernels (run on the GPU box): bash tools/sq_cube4m.sh NAME
# Two --pmc passes (never combined with --stats), raw output under gpurun_out/sq_NAME/, a per-wave digest on stdout.
set -o pipefail
NAME=${1:-x}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/sq_$NAME
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SQ1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"
SQ2="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_BUSY_CYCLES"
rocprofv3 --pmc $SQ1 --kernel-trace --output-format csv -d $OUT/sq1 -- python3 $R/bench.py --no-cpu --no-configs --workload cube4m --steps 4 --warmup 2 > /dev/null 2> $OUT/sq1.err || { echo "sq1 failed"; tail -3 $OUT/sq1.err; exit 1; }
rocprofv3 --pmc $SQ2 --kernel-trace --output-format csv -d $OUT/sq2 -- python3 $R/bench.py --no-cpu --no-configs --workload cube4m --steps 4 --warmup 2 > /dev/null 2> $OUT/sq2.err || { echo "sq2 failed"; tail -3 $OUT/sq2.err; exit 1; }
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/sq*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    if "k_bp_pairs" not in k and "k_sort" not in k: continue
    waves = sorted(c.get("SQ_WAVES", [1]))[len(c.get("SQ_WAVES", [1])) // 2]
    print(k.replace("(anonymous namespace)::", "")[:60])
    for name in sorted(c):
        v = sorted(c[name])[len(c[name]) // 2]
        print(f"    {name:24s} {v:14.0f}  per wave {v / waves:10.1f}")
PY
