#!/bin/bash
# tools/clock_watch.sh VARIANT... — on the GPU box: for each library variant, run the headline render 8 times and sample the
# GPU's shader clock and socket power ten times a second meanwhile (rocm-smi); prints Msamples/s, mean / max sclk and power.
export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = base ]; then export RTMI_LIB_DIR=$PWD/raytracing_rust_amd/lib; else export RTMI_LIB_DIR=$PWD/raytracing_rust_amd/lib_$v; fi
  ( for i in $(seq 1 90); do rocm-smi -d 0 --showclocks --showpower --json 2>/dev/null; echo; sleep 0.1; done ) > /tmp/smi_$v.log &
  SMI=$!
  timeout -k 10 150 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --abi-multi off --no-cold-start --no-collective-at-1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v  %.1f Msamples/s  render %.2f ms' % (d['value'], d['render_kernel_ms_avg']))"
  kill $SMI 2>/dev/null; wait $SMI 2>/dev/null
  python3 - <<PY
import json,re
sclk=[];pw=[]
for ln in open('/tmp/smi_$v.log'):
    ln=ln.strip()
    if not ln.startswith('{'): continue
    try: d=json.loads(ln)
    except Exception: continue
    c=d.get('card0',{})
    for k,val in c.items():
        if 'sclk' in k.lower() and 'level' not in k.lower():
            m=re.search(r'(\d+)',str(val)); 
            if m: sclk.append(int(m.group(1)))
        if 'power' in k.lower() and '(w)' in k.lower():
            try: pw.append(float(val))
            except Exception: pass
busy=[s for s in sclk if s>500]
print('   samples %d  sclk busy mean %.0f max %d min %d MHz   power mean(top half) %.0f max %.0f W' % (len(sclk), sum(busy)/max(len(busy),1), max(sclk or [0]), min(busy or [0]), sum(sorted(pw)[len(pw)//2:])/max(len(pw)-len(pw)//2,1), max(pw or [0])))
PY
done
