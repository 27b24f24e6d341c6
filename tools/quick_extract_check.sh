#!/bin/bash
# Extraction parity tests, then a short bench line with its stage times: tools/quick_extract_check.sh <tag>
set -o pipefail
tag=${1:-quick}
out=gpurun_out/$tag
mkdir -p $out
timeout -k 10 600 python -u -m pytest tests/test_gpu_extract.py tests/test_gpu_golden.py -x -q --timeout 120 --timeout-method thread -m gpu > $out/t_extract.txt 2>&1
rc=$?
tail -3 $out/t_extract.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-extra --cpu-sample 0 > $out/bench.json || exit $?
python - $out/bench.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"])
print({k: round(v["ms"], 3) if isinstance(v, dict) and "ms" in v else v for k, v in d["roofline"]["stages"].items()})
PY
