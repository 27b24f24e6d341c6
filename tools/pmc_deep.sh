#!/bin/bash
# Extra counter passes over the headline workload (memory pipeline and wave-state counters), one rocprofv3 run per
# pass; results under gpurun_out/$1_deep_*.   tools/pmc_deep.sh <tag>
set -u
tag=$1
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp ORBHIP_STREAMS=1
O=gpurun_out
B1="python3 bench.py --steps 1 --warmup 1 --no-extra --cpu-sample 0"
i=0
for pass in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
  "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LEVEL_WAVES" \
  "TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum" \
  "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum GRBM_GUI_ACTIVE" ; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $pass -d $O/${tag}_deep_$i --output-format csv -- $B1 > $O/${tag}_deep_$i.log 2>&1 || { echo "pass $i failed"; tail -5 $O/${tag}_deep_$i.log; exit 1$i; }
done
find $O/${tag}_deep_* -name '*.db' -delete 2>/dev/null
python3 tools/pmc_sq_summary.py $O/${tag}_deep.json "tools/pmc_deep.sh: memory pipeline and wave-state counters, ORBHIP_STREAMS=1, $B1 (1024 x 640x480 frames, 1000 features, per launch)" $O/${tag}_deep_1 $O/${tag}_deep_2 $O/${tag}_deep_3 $O/${tag}_deep_4 > /dev/null
ls -la $O/${tag}_deep.json
