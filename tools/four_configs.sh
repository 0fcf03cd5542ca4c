#!/bin/bash
# tools/four_configs.sh — on the GPU box: the bench value and render-kernel time of BASELINE configs C5, C2, C3, C4 (no CPU legs)
export TMPDIR=/tmp
run() { timeout -k 10 150 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --abi-multi off --no-cold-start --no-collective-at-1 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   %-34s %.1f Msamples/s  render %.2f ms' % (d['config']['workload'], d['value'], d['render_kernel_ms_avg']))"; }
for rep in 1 2; do
run; run --scene random_spheres --nx 1200 --ny 800 --spp 500; run --scene cornell_box --nx 800 --ny 800 --spp 1000; run --scene cornell_smoke --nx 800 --ny 800 --spp 1000
done
