#!/usr/bin/env python3
"""Build-time check of the gfx950 assembly of one kernel (hipcc -S --cuda-device-only output).

  check_isa.py file.s --kernel NAME [--has REGEX]... [--lacks REGEX]... [--max-vgpr N]

NAME is a substring of the kernel's mangled name.  Exits non-zero, naming what failed, if an instruction the
kernel's design relies on is missing, an unwanted one is present, or the register count exceeds the bound.
"""
import argparse
import re
import sys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("asm")
    ap.add_argument("--kernel", required=True)
    ap.add_argument("--has", action="append", default=[])
    ap.add_argument("--lacks", action="append", default=[])
    ap.add_argument("--max-vgpr", type=int, default=None)
    a = ap.parse_args()
    text = open(a.asm).read()
    # body: from the kernel's label to its s_endpgm
    m = re.search(r"^(\S*%s\S*):[^\n]*\n(.*?)\n\s*s_endpgm" % re.escape(a.kernel), text, re.S | re.M)
    if not m:
        sys.exit("check_isa: no kernel matching %r in %s" % (a.kernel, a.asm))
    name, body = m.group(1), m.group(2)
    bad = []
    for r in a.has:
        if not re.search(r, body):
            bad.append("missing %r" % r)
    for r in a.lacks:
        if re.search(r, body):
            bad.append("contains %r" % r)
    if a.max_vgpr is not None:
        mm = re.search(r"^\s*\.set %s\.num_vgpr,\s*(\d+)" % re.escape(name), text, re.M)
        if not mm:
            bad.append("no num_vgpr record")
        elif int(mm.group(1)) > a.max_vgpr:
            bad.append("%s VGPRs, more than %d" % (mm.group(1), a.max_vgpr))
    if bad:
        sys.exit("check_isa: %s in %s: %s" % (name, a.asm, "; ".join(bad)))
    print("check_isa: %s ok (%d instructions)" % (name, len(body.splitlines())))


if __name__ == "__main__":
    main()
