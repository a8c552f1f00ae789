#!/bin/bash
# Kernel statistics of the Dynamic-against-Dynamic contact path:  bash tools/prof_islands.sh [scenario n]   (on the GPU box)
# Writes gpurun_out/prof_islands/ (scratch) and prints the per-kernel summary; copy what should be judged into profiles/.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_islands
rm -rf $OUT/stats
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/tools/measure_islands.py "$@" > $OUT/run.log 2> $OUT/run.err || echo "profiled run failed"
grep -v amdgpu.ids $OUT/run.log
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $OUT/kernel_stats.csv && head -25 $OUT/kernel_stats.csv | cut -c1-160
