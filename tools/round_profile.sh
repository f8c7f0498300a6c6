#!/bin/bash
# One GPU-box session that regenerates the evidence under profiles/: usage  tools/round_profile.sh <tag>   (run from the repo root)
set -eo pipefail
TAG=${1:-v8}
OUT=gpurun_out/$TAG
rm -rf $OUT/stats $OUT/stats_single $OUT/iso $OUT/pmc_sq $OUT/pmc_fetch $OUT/pmc_write
mkdir -p $OUT
export TMPDIR=/tmp
if [ "$2" != "profiles-only" ]; then
python bench.py --steps 5 --warmup 2 > $OUT/bench.json 2> $OUT/bench.log
head -c 700 $OUT/bench.json; echo
fi
if [ "$2" == "bench-only" ]; then exit 0; fi
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 2 --warmup 1 --no-extras > $OUT/stats.log 2>&1
RLPPO_TUNE=4=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_single -- python3 bench.py --steps 2 --warmup 1 --no-extras > $OUT/stats_single.log 2>&1
REPS=40 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/iso -- python3 tools/prof_kernels.py > $OUT/iso.log 2>&1   # 40 launches per shape: the clock ramps as in the bench
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq/a -- python3 tools/prof_kernels.py > $OUT/pmc_sq_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_sq/b -- python3 tools/prof_kernels.py > $OUT/pmc_sq_b.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 tools/prof_kernels.py > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 tools/prof_kernels.py > $OUT/pmc_write.log 2>&1
export PMC_CYCLE='rlppo::gemm_tn_dma_kernel<32>=dW hidden 256x256,dW L0 256x107,dW head 90x256;rlppo::tn_reduce_kernel=hidden,L0,head'
python tools/pmc_summary.py $OUT/pmc_sq > $OUT/pmc_sq.csv || true
python tools/pmc_summary.py $OUT/pmc_fetch > $OUT/pmc_fetch.csv
python tools/pmc_summary.py $OUT/pmc_write > $OUT/pmc_write.csv
python tools/pmc_traffic.py $OUT/pmc_fetch.csv $OUT/pmc_write.csv $OUT/traffic.json
find $OUT -name "*kernel_stats.csv" | head
find $OUT -name "*_kernel_trace.csv" -size +20M -delete || true
echo DONE
