#!/bin/bash
# tools/ab_env.sh VARIANT ENVNAME — on the GPU box: C5, C2, C3 with library VARIANT, without and with ENVNAME=1 (interleaved twice)
export TMPDIR=/tmp
export RTMI_LIB_DIR=$PWD/raytracing_rust_amd/lib_$1
run() { timeout -k 10 150 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --abi-multi off --no-cold-start --no-collective-at-1 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   %.1f Msamples/s  render %.2f ms' % (d['value'], d['render_kernel_ms_avg']))"; }
for rep in 1 2; do
  for on in 0 1; do
    if [ $on = 1 ]; then export $2=1; else unset $2; fi
    echo "== $2=$on (rep $rep): C5, C2, C3"; run; run --scene random_spheres --nx 1200 --ny 800 --spp 500; run --scene cornell_box --nx 800 --ny 800 --spp 1000
  done
done
