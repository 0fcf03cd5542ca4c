export TMPDIR=/tmp
run() { timeout -k 10 150 python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   %.1f Msamples/s  render %.2f ms' % (d['value'], d['render_kernel_ms_avg']))"; }
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -x --timeout 120 -k "fast_cull_equals_exact or spill or (matches_fp32 and coop)" 2>&1 | tail -1
for f in 1 1281 769; do echo "== C5 flags $f"; run --flags $f; echo "== C2 flags $f"; run --flags $f --scene random_spheres --nx 1200 --ny 800 --spp 500;  echo "== C3 flags $f"; run --flags $f --scene cornell_box --nx 800 --ny 800 --spp 1000; done
echo "== C5 flags 1 again"; run --flags 1; echo "== C5 flags 1281 again"; run --flags 1281
