#!/bin/bash
# Where the split-bf16 kernel's time goes: build timing-only variants of csrc/gemm_split.hip that leave parts of a K step out
# (-DSPLIT_ABL=bits, table at the top of that file) and time each at the update's launch shape.  Run on the GPU box:
#   bash tools/split_ablation.sh > gpurun_out/split_ablation.txt
set -e
cd "$(dirname "$0")/.."
LABELS=("0:product" "1:no stage traffic in the steps" "2:no split" "8:large product only" "16:stores dropped" "17:no stage traffic, stores dropped"
        "49:17 + no epilogue work" "113:49 + no stage-request instructions" "241:113 + no barrier per step" "57:49 + large product only")
for e in "${LABELS[@]}"; do
  v=${e%%:*}
  [ -f rlgym_ppo_amd/librlppo_abl$v.so ] || make -s -C rlgym_ppo_amd/csrc variant NAME=abl$v SRC=gemm_split DEFS=-DSPLIT_ABL=$v > /dev/null
done
for m in 1 0; do
  for e in "${LABELS[@]}"; do
    v=${e%%:*}
    printf "%-48s " "${e#*:}"
    MODE=$m RLPPO_LIB=$PWD/rlgym_ppo_amd/librlppo_abl$v.so python tools/split_ablation.py 2>&1 | grep " us"
  done
done
