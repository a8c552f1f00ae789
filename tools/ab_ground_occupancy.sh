#!/bin/bash
# A/B of library builds on the resting ground-plane workload (1 M bodies in contact), kernel times from rocprofv3 (run on the GPU box):
#   bash tools/ab_ground_occupancy.sh libA.so libB.so ...      (files under banggameengine_amd/)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  echo "== $lib"
  export BGE_WORLD_LIB=$R/banggameengine_amd/$lib BGE_GROUND_PHASE=resting
  OUT=$R/gpurun_out/ab_ground/$lib
  mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/measure_ground.py > $OUT/out.txt 2> $OUT/err.txt || { echo failed; tail -3 $OUT/err.txt; exit 1; }
  grep resting $OUT/out.txt
  python3 - $OUT <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if int(r["Calls"]) >= 50:
        print(f"   {r['Name'][:60]:60s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:9.2f} us")
PY
done
