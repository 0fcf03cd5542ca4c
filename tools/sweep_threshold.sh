#!/bin/bash
# tools/sweep_threshold.sh — on the GPU box: final_scene (C5) at several shade thresholds, then random_spheres (C2)
export TMPDIR=/tmp
run() { timeout -k 10 150 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --abi-multi off --no-cold-start --no-collective-at-1 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   %.1f Msamples/s  render %.2f ms' % (d['value'], d['render_kernel_ms_avg']))"; }
for t in 40 32 36 44 48 52 56 40; do echo "== C5 threshold $t"; run --shade-threshold $t; done
for t in 40 32 48 56; do echo "== C2 threshold $t"; run --shade-threshold $t --scene random_spheres --nx 1200 --ny 800 --spp 500; done
