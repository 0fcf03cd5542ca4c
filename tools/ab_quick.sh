#!/bin/bash
# tools/ab_quick.sh VARIANT... — on the GPU box: C5 and C2 bench lines only, each variant twice (interleaved), no parity run
export TMPDIR=/tmp
run() { timeout -k 10 150 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --abi-multi off --no-cold-start --no-collective-at-1 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   %.1f Msamples/s  render %.2f ms' % (d['value'], d['render_kernel_ms_avg']))"; }
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = base ]; then export RTMI_LIB_DIR=$PWD/raytracing_rust_amd/lib; else export RTMI_LIB_DIR=$PWD/raytracing_rust_amd/lib_$v; fi
  echo "== $v (rep $rep): C5, C2"; run; run --scene random_spheres --nx 1200 --ny 800 --spp 500
done
done
