#!/usr/bin/env python3
"""Register / LDS / spill table of every kernel in a `hipcc --cuda-device-only -S` listing (what decides waves per SIMD).
usage: kernel_resources.py file.s [substring ...]        (tools/build_asm.sh writes /tmp/isa/so3proj.s)"""
import re
import subprocess
import sys


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
        return dict(zip(names, out))
    except OSError:
        return {n: n for n in names}


def main():
    text = open(sys.argv[1]).read()
    keys = sys.argv[2:]
    rows = []
    for m in re.finditer(r"^\s*\.amdhsa_kernel (\S+)(.*?)^\s*\.end_amdhsa_kernel", text, re.S | re.M):
        name, body = m.group(1), m.group(2)
        get = lambda k: (re.search(r"\.amdhsa_%s (\S+)" % k, body) or [None, "?"])[1]
        rows.append([name, get("next_free_vgpr"), get("accum_offset"), get("next_free_sgpr"), get("group_segment_fixed_size"), get("private_segment_fixed_size")])
    # the comment block after each function has the spill counts
    spills = {}
    for m in re.finditer(r"; Kernel info:.*?; Function info:|^(\S+):\s*; @(\S+).*?; ScratchSize: (\d+).*?; Occupancy: (\d+)", text, re.S | re.M):
        if m.group(2):
            spills[m.group(2)] = (m.group(3), m.group(4))
    for m in re.finditer(r"; @(\S+)\n.*?; NumVgprs: (\d+)\n; NumAgprs: (\d+)\n.*?; ScratchSize: (\d+)\n; Occupancy: (\d+)", text, re.S):
        spills[m.group(1)] = (m.group(4), m.group(5), m.group(2), m.group(3))
    dm = demangle([r[0] for r in rows])
    print("%-6s %-6s %-6s %-7s %-8s %-4s  kernel" % ("vgpr", "agpr0", "sgpr", "lds", "scratch", "occ"))
    for r in rows:
        nice = dm.get(r[0], r[0])
        if keys and not any(k in nice for k in keys):
            continue
        sp = spills.get(r[0], ("?", "?"))
        print("%-6s %-6s %-6s %-7s %-8s %-4s  %s" % (r[1], r[2], r[3], r[4], r[5], sp[1], nice[:170]))


main()
