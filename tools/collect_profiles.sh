#!/bin/bash
# Copies the summaries of tools/profile_round.sh (gpurun_out/<tag>_*) into profiles/<name>_*.
#   tools/collect_profiles.sh <tag> <name>
set -e
tag=$1; name=$2
cd "$(dirname "$0")/.."
cp gpurun_out/${tag}_trace/vm/*kernel_stats.csv profiles/${name}_kernel_stats.csv
cp gpurun_out/${tag}_mtrace/vm/*kernel_stats.csv profiles/${name}_matcher_kernel_stats.csv
python3 tools/pmc_summary.py gpurun_out/${tag}_fetch gpurun_out/${tag}_write profiles/${name}_pmc_traffic.json \
  "python3 bench.py --steps 1 --warmup 1 --no-extra --cpu-sample 0" "1024x640x480/1000" > /dev/null
python3 tools/pmc_sq_summary.py profiles/${name}_pmc_sq.json \
  "rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES, ORBHIP_STREAMS=1, python3 bench.py --steps 1 --warmup 1 --no-extra --cpu-sample 0 (1024 x 640x480 frames, 1000 features, per launch)" \
  gpurun_out/${tag}_sq > /dev/null
python3 tools/pmc_sq_summary.py profiles/${name}_matcher_pmc_sq.json \
  "rocprofv3 --pmc (same SQ counters), python3 tools/prof_matchers.py: k_knn2_mm on 4096 pairs of 2000 x 2000 and on 4000 x 1.25 M, windowed / projection matchers on 256 pairs of 2000 keypoints, BoW legs" \
  gpurun_out/${tag}_msq > /dev/null
ls -la profiles/${name}_*
