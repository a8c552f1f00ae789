# per-kernel times of configs[3] (4 M bodies + broadphase): rocprofv3 --kernel-trace --stats around bench.py
set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_cube4m_${1:-pred}
rm -rf $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --workload cube4m --steps 20 --warmup 5 --repeats 3 --no-cpu --no-configs > $OUT.json 2> $OUT.err
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/**/*kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:14]:
    print(r["Name"][:60].ljust(60), r["Calls"], r["AverageNs"], r["Percentage"])
PY
