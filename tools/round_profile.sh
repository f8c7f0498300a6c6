#!/bin/bash
# One GPU-box session that regenerates the evidence under profiles/: usage  tools/round_profile.sh <tag> [bench-only|profiles-only] [stage-a|stage-b]   (run from the repo root)
# Collects into gpurun_out/<tag>/; tools/collect_profiles.py then condenses it into profiles/<tag>_*.
set -eo pipefail
TAG=${1:-r04}
OUT=gpurun_out/$TAG
rm -rf $OUT/cfg5_stats $OUT/cfg5_fetch $OUT/cfg5_write $OUT/cfg5_sq $OUT/stats $OUT/stats_single $OUT/iso $OUT/pmc_sq $OUT/pmc_fetch $OUT/pmc_write $OUT/gae_stats $OUT/gae_fetch $OUT/gae_write $OUT/fa_depth $OUT/share8
mkdir -p $OUT
export TMPDIR=/tmp
if [ "$2" != "profiles-only" ]; then
python bench.py --steps 5 --warmup 2 > $OUT/bench.json 2> $OUT/bench.log
head -c 400 $OUT/bench.json; echo
python bench.py --config cfg5 --precision fp32 --steps 3 --warmup 1 > $OUT/bench_cfg5_fp32.json 2> $OUT/bench_cfg5_fp32.log
python bench.py --config cfg5 --precision bf16 --steps 3 --warmup 1 > $OUT/bench_cfg5_bf16.json 2> $OUT/bench_cfg5_bf16.log
fi
if [ "$2" == "bench-only" ]; then exit 0; fi
# a third argument splits the profile passes over two GPU-box calls of <= 20 minutes: stage-a = kernel traces + PMC passes, stage-b = the tools
if [ "$3" != "stage-b" ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 2 --warmup 1 --no-extras > $OUT/stats.log 2>&1
RLPPO_TUNE=4=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_single -- python3 bench.py --steps 2 --warmup 1 --no-extras > $OUT/stats_single.log 2>&1
REPS=40 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/iso -- python3 tools/prof_kernels.py > $OUT/iso.log 2>&1   # 40 launches per shape: the clock ramps as in the bench
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq/a -- python3 tools/prof_kernels.py > $OUT/pmc_sq_a.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 tools/prof_kernels.py > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 tools/prof_kernels.py > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/gae_stats -- python3 tools/prof_gae.py > $OUT/gae_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/gae_fetch -- python3 tools/prof_gae.py > $OUT/gae_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/gae_write -- python3 tools/prof_gae.py > $OUT/gae_write.log 2>&1
# BASELINE configs[4] in the bf16 update precision: per-kernel time inside the update (single stream) and HBM traffic per launch
RLPPO_TUNE=4=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cfg5_stats -- python3 bench.py --config cfg5 --precision bf16 --steps 2 --warmup 1 --no-extras > $OUT/cfg5_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/cfg5_fetch -- python3 bench.py --config cfg5 --precision bf16 --steps 1 --warmup 0 --no-extras > $OUT/cfg5_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/cfg5_write -- python3 bench.py --config cfg5 --precision bf16 --steps 1 --warmup 0 --no-extras > $OUT/cfg5_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/cfg5_sq -- python3 bench.py --config cfg5 --precision bf16 --steps 1 --warmup 0 --no-extras > $OUT/cfg5_sq.log 2>&1
export PMC_CYCLE='rlppo::tn_reduce_kernel=hidden,L0,head;rlppo::gemm_nt_dma_kernel<8, 1, 16, true, false>=fwd hidden 256->256,fwd L0 112->256;rlppo::gemm_nt_dma_kernel<8, 3, 16, true, false>=dX hidden 256->256,dX head 96->256'
python tools/pmc_summary.py $OUT/pmc_sq > $OUT/pmc_sq.csv || true
python tools/pmc_summary.py $OUT/pmc_fetch > $OUT/pmc_fetch.csv
python tools/pmc_summary.py $OUT/pmc_write > $OUT/pmc_write.csv
python tools/pmc_traffic.py $OUT/pmc_fetch.csv $OUT/pmc_write.csv $OUT/traffic.json
python tools/pmc_summary.py $OUT/gae_fetch > $OUT/gae_pmc_fetch.csv
python tools/pmc_summary.py $OUT/gae_write > $OUT/gae_pmc_write.csv
python tools/pmc_traffic.py $OUT/gae_pmc_fetch.csv $OUT/gae_pmc_write.csv $OUT/gae_traffic.json
PMC_CYCLE= python tools/pmc_summary.py $OUT/cfg5_fetch > $OUT/cfg5_pmc_fetch.csv
PMC_CYCLE= python tools/pmc_summary.py $OUT/cfg5_write > $OUT/cfg5_pmc_write.csv
PMC_CYCLE= python tools/pmc_summary.py $OUT/cfg5_sq > $OUT/cfg5_pmc_sq.csv
python tools/pmc_traffic.py $OUT/cfg5_pmc_fetch.csv $OUT/cfg5_pmc_write.csv $OUT/cfg5_traffic.json
if [ "$3" == "stage-a" ]; then echo DONE-A; exit 0; fi
fi  # (stage-b starts here)
python tools/rank_share.py > $OUT/rank_share.txt 2> $OUT/rank_share.log
python tools/ab_update.py 3 > $OUT/ab_update.txt 2> $OUT/ab_update.log
python tools/breakdown_rows.py > $OUT/breakdown_rows.txt 2> $OUT/breakdown_rows.log
python tools/act_kernel_time.py > $OUT/act_kernel_time.txt 2> $OUT/act_kernel_time.log
python tools/rollout_breakdown.py > $OUT/rollout_breakdown.txt 2> $OUT/rollout_breakdown.log
python tools/small_batch_latency.py > $OUT/small_batch_latency.txt 2> $OUT/small_batch_latency.log
python tools/get_action_profile.py > $OUT/get_action_profile.txt 2> $OUT/get_action_profile.log
{ python tools/heads_latency.py; RLPPO_TUNE="40=0" python tools/heads_latency.py; python tools/heads_latency.py; } > $OUT/heads_latency.txt 2> $OUT/heads_latency.log
{ python tools/host_window_probe.py 0; python tools/host_window_probe.py 1; python tools/host_window_probe.py 3; } 2>&1 | grep -v amdgpu.ids > $OUT/host_window_probe.txt
{ python tools/get_action_modes.py; RLPPO_ACT_PUSH=0 python tools/get_action_modes.py; python tools/get_action_modes.py; RLPPO_ACT_PUSH=0 python tools/get_action_modes.py; } > $OUT/get_action_modes.txt 2> $OUT/get_action_modes.log
python tools/host_noise_pipeline.py 0:1 1:1 2:2 3:2 3:3 4:2 > $OUT/host_noise_pipeline.txt 2> $OUT/host_noise_pipeline.log
rocprofv3 --kernel-trace --output-format csv -d $OUT/fa_depth -- python3 tools/fused_act_depth_time.py > $OUT/fa_depth.log 2>&1
python tools/fused_act_depth_report.py $(find $OUT/fa_depth -name "*kernel_trace.csv" | head -1) > $OUT/fused_act_depth.txt
rocprofv3 --kernel-trace --output-format csv -d $OUT/share8 -- python3 tools/rank_share.py 8 > $OUT/share8.log 2>&1
python tools/step_gaps.py $OUT/share8 8 > $OUT/rank_share_step_gaps.txt
python tools/gae_ab.py > $OUT/gae_floor.txt 2> $OUT/gae_floor.log
python tools/b16_k_sweep.py > $OUT/b16_k_sweep.txt 2> $OUT/b16_k_sweep.log
python tools/f32_k_sweep.py > $OUT/f32_k_sweep.txt 2> $OUT/f32_k_sweep.log
python tools/ceilings.py > $OUT/ceilings.txt 2> $OUT/ceilings.log
find $OUT -name "*_kernel_trace.csv" -size +8M -delete || true
find $OUT -name "*counter_collection.csv" -size +8M -delete || true
echo DONE
