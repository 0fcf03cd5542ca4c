#!/bin/bash
# tools/gpu_diag_pmc.sh TAG [bench args] — extra SQ / SQC counter passes (diagnostics, not the judged profile)
TAG=${1:-diag}; shift || true
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
pass() { name=$1; shift; echo "== pmc $name"; timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/pmc_$name -- python3 bench.py $BENCH_ARGS --steps 1 --warmup 1 --no-cpu-baseline --abi-multi off --no-cold-start --no-collective-at-1 > $OUT/pmc_$name.json 2> $OUT/pmc_$name.err || echo "pmc $name failed"; }
BENCH_ARGS="$*"
pass A SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAIT_INST_ANY
pass B SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_ANY SQ_WAIT_INST_LDS
pass C SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_VALU_TRANS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_INT32
pass D TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
find $OUT -name "*.db" -delete 2>/dev/null; find $OUT -size +8M -delete 2>/dev/null
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
    print(f.split("/")[2], {k: "%.4g" % v for k, v in acc.items()})
PY
