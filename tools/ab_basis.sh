#!/bin/bash
# A/B of library builds on the BGE_TICK_BULLET_BASIS tick (run on the GPU box): bash tools/ab_basis.sh libA.so libB.so ...
cd ${GRAFT_REPO_ROOT:-/root/repo}
for round in 1 2; do
for lib in "$@"; do
  BGE_WORLD_LIB=$PWD/banggameengine_amd/$lib python bench.py --bullet-basis --steps 600 --warmup 50 --no-cpu --no-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', 'ms_per_step %.5f' % d['ms_per_step'], 'frac %.4f' % d['roofline']['frac'], 'kernel %.5f' % d['roofline'].get('kernel_ms_per_launch'))"
done
done
