#!/bin/bash
# tools/pmc_ab.sh VARIANT... — on the GPU box: SQ counter passes of the headline launch for each library variant (A/B of
# what the hardware did, not only how long it took)
export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = base ]; then export RTMI_LIB_DIR=$PWD/raytracing_rust_amd/lib; else export RTMI_LIB_DIR=$PWD/raytracing_rust_amd/lib_$v; fi
  OUT=gpurun_out/pmc_ab_$v; mkdir -p $OUT
  pass() { name=$1; shift; timeout -k 10 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/pmc_$name -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --abi-multi off --no-cold-start --no-collective-at-1 > $OUT/pmc_$name.json 2> $OUT/pmc_$name.err || echo "pmc $name failed"; }
  pass A SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAIT_INST_ANY
  pass B SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_LDS
  pass C SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_TRANS_F32 SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_INT32 SQ_VALU_MFMA_BUSY_CYCLES
  find $OUT -name "*.db" -delete 2>/dev/null; find $OUT -size +8M -delete 2>/dev/null
  echo "#### $v"
  python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/pmc_*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(float); last = {}
    for r in csv.DictReader(open(f)):
        if "rtmi_render" not in r["Kernel_Name"]: continue
        last[r["Counter_Name"]] = max(last.get(r["Counter_Name"], 0), int(r["Dispatch_Id"]))
    for r in csv.DictReader(open(f)):
        if "rtmi_render" in r["Kernel_Name"] and int(r["Dispatch_Id"]) == last[r["Counter_Name"]]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
    print(f.split("/")[2], {k: "%.5g" % v for k, v in acc.items()})
PY
done
