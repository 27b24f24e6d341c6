#!/bin/bash
# Everything the round's last measurement needs, in one GPU call: the GPU test suite, the smoke call, the bench line
# with all legs, the two-rank rehearsal over gloo.   tools/final_round.sh <tag>
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-final}
O=gpurun_out/$tag
mkdir -p $O
timeout -k 10 900 python3 -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $O/gpu_tests.txt 2>&1
rc=$?; tail -3 $O/gpu_tests.txt; [ $rc -eq 0 ] || exit $rc
timeout -k 10 120 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
rc=$?; tail -1 $O/smoke.txt; [ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python3 bench.py > $O/bench.json 2> $O/bench.err
rc=$?; [ $rc -eq 0 ] || { tail -5 $O/bench.err; exit $rc; }
python3 -c "
import json
d = json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline'])"
timeout -k 10 600 python3 bench.py --gpus 2 --backend gloo --steps 5 --warmup 2 > $O/bench_two_ranks_gloo.json 2> $O/bench_two_ranks_gloo.err
rc=$?; [ $rc -eq 0 ] || { tail -5 $O/bench_two_ranks_gloo.err; exit $rc; }
python3 -c "
import json
d = json.loads(open('$O/bench_two_ranks_gloo.json').read().strip().splitlines()[-1])
print('two ranks:', d['value'], d['n_gpus'], d['config'])"
