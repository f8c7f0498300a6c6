#!/bin/bash
# HBM traffic per launch (rocprofv3 PMC: FETCH_SIZE and WRITE_SIZE in separate passes, tools/pmc_traffic.py's gfx950 correction) of the
# update's GEMM launches at several row counts per pass -- the grouped weight-gradient launch is a third of a 50,000 / 65,536-row step.
# usage: tools/traffic_rows.sh <outdir> [rows ...]      (run from the repo root on the GPU box)
set -eo pipefail
OUT=${1:-gpurun_out/traffic_rows}; shift || true
ROWS=${@:-"50000 65536 524288"}
mkdir -p $OUT
export TMPDIR=/tmp
export PMC_CYCLE='rlppo::tn_reduce_kernel=hidden,L0,head;rlppo::gemm_nt_dma_kernel<8, 1, 16, true, false>=fwd hidden 256->256,fwd L0 112->256;rlppo::gemm_nt_dma_kernel<8, 3, 16, true, false>=dX hidden 256->256,dX head 96->256'
for M in $ROWS; do
  export M
  rm -rf $OUT/fetch_$M $OUT/write_$M
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_$M -- python3 tools/prof_kernels.py > $OUT/fetch_$M.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write_$M -- python3 tools/prof_kernels.py > $OUT/write_$M.log 2>&1
  python tools/pmc_summary.py $OUT/fetch_$M > $OUT/fetch_$M.csv
  python tools/pmc_summary.py $OUT/write_$M > $OUT/write_$M.csv
  python tools/pmc_traffic.py $OUT/fetch_$M.csv $OUT/write_$M.csv $OUT/traffic_$M.json > /dev/null
  python - "$OUT/traffic_$M.json" $M <<'PY'
import json, sys
t, M = json.load(open(sys.argv[1])), int(sys.argv[2])
K0 = 112
alg = {"rlppo::gemm_tn_group_kernel": 4 * M * (2 * (256 + K0) + 4 * (256 + 256) + (96 + 256)),
       "rlppo::gemm_nt_dma_kernel<8, 1, 16, true, false> {fwd hidden 256->256}": 4 * M * (256 + 256) + M * 256 // 8,
       "rlppo::gemm_nt_dma_kernel<8, 3, 16, true, false> {dX hidden 256->256}": 4 * M * (256 + 256) + M * 256 // 8}
for k, a in alg.items():
    if k in t:
        print("rows %7d  %-72s HBM %8.1f MB per launch, operands + outputs once %8.1f MB: x %.3f" % (M, k, t[k]["hbm_bytes"] / 1e6, a / 1e6, t[k]["hbm_bytes"] / a))
PY
done
find $OUT -name "*.csv" -size +4M -delete || true
