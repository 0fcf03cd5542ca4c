#!/bin/bash
# tools/gpu_profile.sh TAG [bench args...] — run on the GPU box (via gpurun):
#   1. the bench line (no CPU baseline)  -> gpurun_out/TAG/bench.json
#   2. rocprofv3 --kernel-trace --stats  -> gpurun_out/TAG/stats/*  (same command)
#   3. PMC passes (own runs, no tracing domains mixed in): FETCH_SIZE, WRITE_SIZE, SQ counters
# Afterwards, here: python tools/pmc_summary.py gpurun_out/TAG profiles/NAME "<scene> <nx>x<ny>x<spp>spp"
# and copy gpurun_out/TAG/{bench.json,kernel_stats.csv} into profiles/NAME.
set -e
TAG=${1:-run}; shift || true
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== bench $TAG $*" ; date
timeout -k 10 400 python3 bench.py --no-cpu-baseline --abi-multi off --no-cold-start --no-collective-at-1 "$@" > $OUT/bench.json 2> $OUT/bench.err
tail -c 1500 $OUT/bench.json
echo "== rocprofv3 stats" ; date
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py "$@" --no-cpu-baseline --abi-multi off --no-cold-start --no-collective-at-1 > $OUT/stats_bench.json 2> $OUT/stats.err || echo "rocprof stats failed"
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -r -I{} cp {} $OUT/kernel_stats.csv
head -4 $OUT/kernel_stats.csv || true
for C in FETCH_SIZE WRITE_SIZE; do
  echo "== pmc $C" ; date
  timeout -k 10 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -- python3 bench.py "$@" --steps 1 --warmup 1 --no-cpu-baseline --abi-multi off --no-cold-start --no-collective-at-1 > $OUT/pmc_$C.json 2> $OUT/pmc_$C.err || echo "pmc $C failed"
done
echo "== pmc SQ" ; date
timeout -k 10 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/pmc_SQ -- python3 bench.py "$@" --steps 1 --warmup 1 --no-cpu-baseline --abi-multi off --no-cold-start --no-collective-at-1 > $OUT/pmc_SQ.json 2> $OUT/pmc_SQ.err || echo "pmc SQ failed"
# keep only small files for the merge-back (<= 64 MiB)
find $OUT -name "*.db" -delete 2>/dev/null || true
find $OUT -size +8M -delete 2>/dev/null || true
du -sh $OUT
date
