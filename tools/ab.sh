#!/bin/bash
# A/B of two builds of the library on the same box: tools/ab.sh <libA.so> <libB.so> [rounds]; prints the stage times of
# the headline workload for each, alternating.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
for r in $(seq 1 ${3:-2}); do
  for l in "$1" "$2"; do
    ORBHIP_LIBRARY=$l timeout -k 10 200 python3 bench.py --no-extra --cpu-sample 0 --steps 10 --warmup 2 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$l', j['value'], j['ms_per_step'], {k: v['ms_per_step'] for k, v in j['roofline']['stages'].items() if v['launches_per_step']})" || exit 1
  done
done
