#!/bin/bash
# Headline rate by chunk size (frames that go through the stages together) and number of worker streams.
mkdir -p gpurun_out/$1
for mb in 1024 256 128 64 32; do
  for st in 1 2 4; do
    ORBHIP_STREAMS=$st timeout -k 10 120 python bench.py --steps 5 --warmup 2 --no-extra --cpu-sample 0 --max-batch $mb > gpurun_out/$1/b_${mb}_$st.json || exit 1
    python - gpurun_out/$1/b_${mb}_$st.json $mb $st <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("max_batch", sys.argv[2], "streams", sys.argv[3], d["value"], d["ms_per_step"], {k: v["ms_per_step"] for k, v in d["roofline"]["stages"].items()})
PY
  done
done
