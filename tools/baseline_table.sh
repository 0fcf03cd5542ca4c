#!/bin/bash
# tools/baseline_table.sh — run on the GPU box: one bench line per BASELINE.json config (GPU x1 + bounded CPU sample)
export TMPDIR=/tmp
OUT=gpurun_out/baseline; mkdir -p $OUT
run() { name=$1; shift; timeout -k 10 500 python3 bench.py --steps 2 --warmup 1 "$@" > $OUT/$name.json 2> $OUT/$name.err || echo "$name failed"; }
run C1_two_spheres    --scene two_spheres    --nx 400  --ny 225  --spp 100  --cpu-rows 15 --cpu-spp 100
run C2_random_spheres --scene random_spheres --nx 1200 --ny 800  --spp 500  --cpu-rows 16 --cpu-spp 64
run C3_cornell_box    --scene cornell_box    --nx 800  --ny 800  --spp 1000 --cpu-rows 16 --cpu-spp 256
run C4_cornell_smoke  --scene cornell_smoke  --nx 800  --ny 800  --spp 1000 --cpu-rows 16 --cpu-spp 512
run C5_final_scene    --scene final_scene    --nx 1920 --ny 1080 --spp 1000 --cpu-rows 16 --cpu-spp 96
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/baseline/*.json")):
    try: b = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unreadable", e); continue
    r, c = b["roofline"], b["cpu_baseline"]
    print("| %s | %s | %.0f | %.0f | %.4f | %.1f | %.3f | %.4f | x%.0f |" % (f.split("/")[-1][:-5], b["config"]["workload"], r["bytes_per_sample"], r["flops_per_sample"], c["value"], b["value"], r["frac"], r["valu"]["frac"], c["gpu_over_cpu"]))
PY
