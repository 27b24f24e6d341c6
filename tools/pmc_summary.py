#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into a per-kernel traffic summary.

usage: pmc_summary.py <fetch_dir> <write_dir> <out.json> [<command note> [<workload BxWxH/N>]]

FETCH_SIZE / WRITE_SIZE are reported in KiB per dispatch.  On gfx950 FETCH_SIZE counts a 128-byte
request as 64 bytes for wide (16 B per lane) coalesced reads, so the fetch figure is doubled for
kernels whose global loads are 16 bytes per lane; other widths are left raw and flagged as
uncalibrated.  WRITE_SIZE is exact for coalesced stores.
"""
import collections
import csv
import glob
import json
import sys

WIDE_LOADS = {"k_fast_cell", "k_resize", "k_orient_describe"}  # kernels whose global loads are 16 B per lane


def per_kernel(d, counter):
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    tot = collections.defaultdict(float)
    n = collections.Counter()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter or "orbk::" not in r["Kernel_Name"]:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("orbk::", "").replace("void ", "").split("<")[0].strip()
        tot[k] += float(r["Counter_Value"]) * 1024.0
        n[k] += 1
    return tot, n


def main():
    fetch, nf = per_kernel(sys.argv[1], "FETCH_SIZE")
    write, nw = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {"unit": "bytes per launch", "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, "
           "ORBHIP_STREAMS=1, " + (sys.argv[4] if len(sys.argv) > 4 else "python3 bench.py --steps 2 --warmup 1 --no-extra --cpu-sample 0"),
           "workload": sys.argv[5] if len(sys.argv) > 5 else "1024x640x480/1000", "kernels": {}}
    for k in sorted(fetch):
        fr = fetch[k] / nf[k]
        wr = write.get(k, 0.0) / max(nw.get(k, 1), 1)
        wide = k in WIDE_LOADS
        out["kernels"][k] = {"launches_seen": nf[k], "fetch_raw": round(fr), "write": round(wr),
                             "fetch_corrected": round(fr * 2) if wide else None,
                             "traffic": round(fr * 2 + wr) if wide else None,
                             "note": "16 B/lane loads: fetch doubled (gfx950 correction)" if wide
                             else "load width not 16 B/lane: fetch counter uncalibrated, no traffic figure"}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
