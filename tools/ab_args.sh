#!/bin/bash
# tools/ab_args.sh "ARGS1" "ARGS2" ... — on the GPU box: C5 bench line of the default build for each argument string, twice, interleaved
export TMPDIR=/tmp
run() { timeout -k 10 150 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --abi-multi off $1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   %.1f Msamples/s  render %.2f ms' % (d['value'], d['render_kernel_ms_avg']))"; }
for rep in 1 2; do
for a in "$@"; do echo "== [$a] (rep $rep)"; run "$a"; done
done
