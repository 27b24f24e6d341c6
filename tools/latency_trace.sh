#!/bin/bash
# Kernel trace of the one-frame-per-call path (tools/latency_probe.py) on the GPU box: per-kernel durations and the
# start-to-start timeline of one call, under gpurun_out/lat_trace.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
timeout -s KILL 200 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $O/lat_trace --output-format csv -- python3 tools/latency_probe.py > $O/lat_trace.log 2>&1 || exit 11
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/lat_trace/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
copies = []
for g in glob.glob("gpurun_out/lat_trace/*/*memory_copy_trace.csv"):
    copies = list(csv.DictReader(open(g)))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40]) for r in rows]
ev += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", r.get("Name", ""))[:30]) for r in copies]
ev.sort()
# the last complete call: find the last k_level0 and print from there
idx = [i for i, e in enumerate(ev) if "k_level0" in e[2]]
i0 = idx[-2]
t0 = ev[i0][0]
for s, e, n in ev[max(i0 - 2, 0):idx[-1]]:
    print("%9.1f us  +%7.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, n))
PY
find $O/lat_trace -name '*.db' -delete 2>/dev/null; find $O/lat_trace -name '*trace.csv' -size +5M -delete 2>/dev/null
tail -2 $O/lat_trace.log
