#!/bin/bash
# Profiles of one round, taken on the GPU box in ONE gpurun call:  bash tools/profile_round.sh r02
# Writes raw rocprofv3 output under gpurun_out/<round>prof/ (scratch); tools/profile_digest.py turns it into the files
# committed under profiles/<round>/.  Counters go in their own passes (never combined with --stats), FETCH_SIZE and
# WRITE_SIZE in separate passes (TCC slots), as MI355X_MICROARCH.md prescribes.
set -o pipefail
ROUND=${1:-r02}
PART=${2:-all}   # "a": kernel statistics + FETCH / WRITE passes, "b": SQ passes + the ground plane, "all": both (more than one 20-minute gpurun call)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${ROUND}prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { # name, then bench.py arguments
    local name=$1; shift
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$name -- python3 $R/bench.py --no-cpu --no-configs "$@" > $OUT/stats_$name.json 2> $OUT/stats_$name.err || echo "stats $name failed"
    echo "stats $name done"
}
pmc() { # name, counters (quoted), then bench.py arguments
    local name=$1; local counters=$2; shift; shift
    rocprofv3 --pmc $counters --kernel-trace --output-format csv -d $OUT/pmc_$name -- python3 $R/bench.py --no-cpu --no-configs "$@" > /dev/null 2> $OUT/pmc_$name.err || echo "pmc $name failed"
    echo "pmc $name done"
}
if [ $PART != b ]; then
run flat1m
run chains4 --workload chains4 --steps 600 --warmup 50
run subtree64 --workload subtree64 --steps 300 --warmup 30
run cube4m --workload cube4m --steps 60 --warmup 10
run flat1m_basis --bullet-basis --steps 600 --warmup 50
run flat16m --workload flat1m --entities 16000000 --steps 40 --warmup 5
for wl in flat1m chains4 subtree64 cube4m; do
    extra="--workload $wl --steps 20 --warmup 5"
    [ $wl = cube4m ] && extra="--workload cube4m --steps 5 --warmup 2"
    pmc fetch_$wl FETCH_SIZE $extra
    pmc write_$wl WRITE_SIZE $extra
done
pmc fetch_flat1m_basis FETCH_SIZE --bullet-basis --steps 20 --warmup 5
pmc write_flat1m_basis WRITE_SIZE --bullet-basis --steps 20 --warmup 5
fi
if [ $PART != a ]; then
SQ1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"
SQ2="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_BUSY_CYCLES"
for wl in flat1m chains4 subtree64; do
    pmc sq1_$wl "$SQ1" --workload $wl --steps 20 --warmup 5
    pmc sq2_$wl "$SQ2" --workload $wl --steps 20 --warmup 5
done
pmc sq1_cube4m "$SQ1" --workload cube4m --steps 4 --warmup 2
pmc sq2_cube4m "$SQ2" --workload cube4m --steps 4 --warmup 2
# the ground plane's solver kernel on 1 M resting bodies: kernel trace, then the SQ counters of the same script
export BGE_GROUND_PHASE=resting
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_ground -- python3 $R/tools/measure_ground.py > $OUT/stats_ground.txt 2> $OUT/stats_ground.err || echo "stats ground failed"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_ground -- python3 $R/tools/measure_ground.py > /dev/null 2> $OUT/pmc_ground.err || echo "pmc ground failed"
unset BGE_GROUND_PHASE
echo "ground done"
fi
echo "profiles done"
