#!/usr/bin/env python3
"""Per-kernel averages of the counters of one or more rocprofv3 --pmc passes.

usage: pmc_sq_summary.py <out.json> <source note> <pass_dir> [<pass_dir> ...]

Every pass directory holds one *counter_collection.csv; a counter's value is summed over the rows of a
dispatch (rocprofv3 writes one row per counter instance dimension) and averaged over the dispatches of
each kernel.  Kernel names are cut at the first '(' and stripped of namespaces."""
import collections
import csv
import glob
import json
import sys


def main():
    out_path, note, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
    kernels = collections.defaultdict(dict)
    for d in dirs:
        files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
        if not files:
            raise SystemExit("no counter_collection.csv under " + d)
        per = collections.defaultdict(lambda: collections.defaultdict(float))  # (kernel, counter) -> dispatch -> value
        for r in csv.DictReader(open(files[0])):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].split("::")[-1].strip()
            per[(k, r["Counter_Name"])][r["Dispatch_Id"]] += float(r["Counter_Value"])
        for (k, c), disp in per.items():
            kernels[k][c] = round(sum(disp.values()) / len(disp), 1)
            kernels[k]["launches_seen"] = len(disp)
    out = {"source": note, "unit": "per launch (average over the launches seen)", "kernels": {k: kernels[k] for k in sorted(kernels)}}
    json.dump(out, open(out_path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
