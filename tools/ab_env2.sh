#!/bin/bash
# tools/ab_env2.sh VARIANT "ENV1=1 ENV2=1" ... — on the GPU box: C5, C2, C3 with library VARIANT under each environment setting (twice)
export TMPDIR=/tmp
V=$1; shift
export RTMI_LIB_DIR=$PWD/raytracing_rust_amd/lib_$V
run() { timeout -k 10 150 env $E python bench.py --steps 3 --warmup 1 --no-cpu-baseline --abi-multi off --no-cold-start --no-collective-at-1 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   %.1f Msamples/s  render %.2f ms' % (d['value'], d['render_kernel_ms_avg']))"; }
for rep in 1 2; do
  for E in "$@"; do
    echo "== $E (rep $rep): C5, C2, C3"; run; run --scene random_spheres --nx 1200 --ny 800 --spp 500; run --scene cornell_box --nx 800 --ny 800 --spp 1000
  done
done
