# Development tool: headline step time for several numbers of worker streams.
for s in ${SWEEP:-1 2 3 4}; do
  echo "== ORBHIP_STREAMS=$s"
  ORBHIP_STREAMS=$s timeout -k 10 120 python bench.py --no-extra --cpu-sample 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" || exit 1
done
