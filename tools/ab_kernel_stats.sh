#!/bin/bash
# A/B of one environment switch under rocprofv3 --kernel-trace --stats (run on the GPU box):
#   bash tools/ab_kernel_stats.sh NAME VAR VALUE_A VALUE_B -- <bench.py arguments>
# Prints the per-kernel average durations of both runs side by side; raw output under gpurun_out/ab_NAME/.
set -o pipefail
NAME=$1; VAR=$2; A=$3; B=$4; shift 5
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/ab_$NAME
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in $A $B; do
    export $VAR=$v
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$v -- python3 $R/bench.py --no-cpu --no-configs "$@" > $OUT/$v.json 2> $OUT/$v.err || { echo "$v failed"; tail -5 $OUT/$v.err; exit 1; }
done
python3 - $OUT $A $B <<'PY'
import csv, glob, sys
out, a, b = sys.argv[1:4]
def load(v):
    f = glob.glob(f"{out}/{v}/**/*kernel_stats.csv", recursive=True)[0]
    return {r["Name"]: (int(r["Calls"]), float(r["AverageNs"]) / 1e3) for r in csv.DictReader(open(f))}
da, db = load(a), load(b)
print(f"{'kernel':70s} {a:>10s} {b:>10s}")
for k in sorted(set(da) | set(db), key=lambda k: -(da.get(k, (0, 0))[1] + db.get(k, (0, 0))[1])):
    if max(da.get(k, (0, 0))[0], db.get(k, (0, 0))[0]) < 5: continue
    short = k.replace("(anonymous namespace)::", "").replace("void ", "")[:70]
    print(f"{short:70s} {da.get(k, (0, 0))[1]:10.2f} {db.get(k, (0, 0))[1]:10.2f}")
PY
