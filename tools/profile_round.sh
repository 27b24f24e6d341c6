#!/bin/bash
# Profiling passes of one round on the GPU box; results under gpurun_out/$1_*.
#   tools/profile_round.sh <tag>
# Pass 1: kernel trace of the headline workload (single library stream so kernel durations do not overlap).
# Passes 2-4: PMC counters of the same command, each in a run of its own (FETCH_SIZE, WRITE_SIZE, SQ group).
# Passes 5-6: kernel trace and SQ counters of the matcher legs.
set -u
tag=$1
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp ORBHIP_STREAMS=1
O=gpurun_out
SQ="SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES"
B="python3 bench.py --steps 5 --warmup 2 --no-extra --cpu-sample 0"
B1="python3 bench.py --steps 1 --warmup 1 --no-extra --cpu-sample 0"
timeout -s KILL 240 rocprofv3 --kernel-trace --stats -d $O/${tag}_trace --output-format csv -- $B > $O/${tag}_trace.log 2>&1 || exit 11
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $O/${tag}_fetch --output-format csv -- $B1 > $O/${tag}_fetch.log 2>&1 || exit 12
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $O/${tag}_write --output-format csv -- $B1 > $O/${tag}_write.log 2>&1 || exit 13
timeout -s KILL 120 rocprofv3 --pmc $SQ -d $O/${tag}_sq --output-format csv -- $B1 > $O/${tag}_sq.log 2>&1 || exit 14
timeout -s KILL 240 rocprofv3 --kernel-trace --stats -d $O/${tag}_mtrace --output-format csv -- python3 tools/prof_matchers.py > $O/${tag}_mtrace.log 2>&1 || exit 15
timeout -s KILL 240 rocprofv3 --pmc $SQ -d $O/${tag}_msq --output-format csv -- python3 tools/prof_matchers.py > $O/${tag}_msq.log 2>&1 || exit 16
# keep only the small csv files (the merge back is capped)
find $O/${tag}_* -name '*.db' -delete 2>/dev/null; find $O/${tag}_* -name '*kernel_trace.csv' -size +20M -delete 2>/dev/null
du -sh $O/${tag}_* | tail -8
