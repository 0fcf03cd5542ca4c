#!/bin/bash
# tools/ab_run.sh VARIANT... — on the GPU box: parity smoke + C5 and C2 bench lines for each variant directory
# raytracing_rust_amd/lib_VARIANT ("base" = raytracing_rust_amd/lib), interleaved twice (A B A B) against clock drift.
export TMPDIR=/tmp
run() { timeout -k 10 150 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --abi-multi off --no-cold-start --no-collective-at-1 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   %.1f Msamples/s  render %.2f ms' % (d['value'], d['render_kernel_ms_avg']))"; }
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = base ]; then export RTMI_LIB_DIR=$PWD/raytracing_rust_amd/lib; else export RTMI_LIB_DIR=$PWD/raytracing_rust_amd/lib_$v; fi
  if [ $rep = 1 ]; then
    timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -x --timeout 120 -k "fast_cull_equals_exact or spill or (matches_fp32 and coop)" 2>&1 | tail -1
  fi
  echo "== $v (rep $rep): C5, C2, C3"; run; run --scene random_spheres --nx 1200 --ny 800 --spp 500; run --scene cornell_box --nx 800 --ny 800 --spp 1000
done
done
