#!/bin/bash
# GPU tests (optional, "test" as first argument) and the stage times of the three extraction sizes.
cd "$GRAFT_REPO_ROOT" || exit 1
if [ "${1:-}" = "test" ]; then
  timeout -k 10 500 python3 -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -4 || exit 1
fi
for cfg in "640 480 1000" "1280 720 2000" "1920 1080 4000"; do
  set -- $cfg
  timeout -k 10 200 python3 -u bench.py --no-extra --cpu-sample 0 --width $1 --height $2 --features $3 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', j['value'], j['ms_per_step'], {k:v['ms_per_step'] for k,v in j['roofline']['stages'].items() if v['launches_per_step']})" || exit 2
done
