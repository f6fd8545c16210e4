#!/bin/bash
# profiles/r03_profile.sh — round-3 rocprofv3 collection, run from the repo root on the GPU box:
#   bash profiles/r03_profile.sh            (outputs under gpurun_out/r03prof/, summaries copied into profiles/ afterwards)
# Kernel-trace/stats and each --pmc set are SEPARATE runs (counters perturb timing; FETCH_SIZE and WRITE_SIZE do not fit one pass).
set -o pipefail
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r03prof
mkdir -p $OUT
cd $ROOT
echo "== kernel stats: bench line" ; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_kstats -- python3 bench.py --no-cpu --no-sizes --no-prove --steps 20 > $OUT/bench_under_rocprof.json 2> $OUT/bench_kstats.err
echo "== pmc SQ: NTT workload" ; rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT/pmc_sq_ntt -- python3 profiles/pmc_workload.py > /dev/null 2> $OUT/pmc_sq_ntt.err
echo "== pmc FETCH_SIZE: NTT workload" ; rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 profiles/pmc_workload.py > /dev/null 2> $OUT/pmc_fetch.err
echo "== pmc WRITE_SIZE: NTT workload" ; rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 profiles/pmc_workload.py > /dev/null 2> $OUT/pmc_write.err
python3 profiles/summarize_pmc.py $OUT/pmc_sq_ntt $OUT/pmc_fetch $OUT/pmc_write > $OUT/pmc_summary_ntt.txt
python3 profiles/summarize_pmc.py --traffic $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_traffic.json
find $OUT -name "*kernel_stats.csv" | head
find $OUT -name "*.db" -delete ; find $OUT -name "*counter_collection.csv" -size +8M -delete ; find $OUT -name "*kernel_trace.csv" -size +8M -delete
du -sh $OUT
