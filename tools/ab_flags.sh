#!/bin/bash
# tools/ab_flags.sh VARIANT "FLAGS..." [bench args] — on the GPU box: C5 bench lines of ONE library build
# (raytracing_rust_amd/lib_VARIANT, "base" = lib) for several --flags values (run-time knobs), each twice, interleaved.
export TMPDIR=/tmp
V=$1; FL=$2; shift 2
if [ "$V" = base ]; then export RTMI_LIB_DIR=$PWD/raytracing_rust_amd/lib; else export RTMI_LIB_DIR=$PWD/raytracing_rust_amd/lib_$V; fi
run() { timeout -k 10 150 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --abi-multi off "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   %.1f Msamples/s  render %.2f ms' % (d['value'], d['render_kernel_ms_avg']))"; }
for rep in 1 2; do
for f in $FL; do
  echo "== $V flags=$f (rep $rep) $*"; run --flags $f "$@"
done
done
