cd $GRAFT_REPO_ROOT
export BGE_TRIGGER_PROFILE=1
python tools/measure_triggers.py 2>&1 | grep -E "bge\]|bodies"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trigprof -- python3 $GRAFT_REPO_ROOT/tools/measure_triggers.py > /dev/null 2>&1
python3 - <<'PY'
import csv,glob,os
f=glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/trigprof/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "trig" in r["Name"] or "query" in r["Name"] or "classify" in r["Name"]:
        print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.2f} us")
PY
