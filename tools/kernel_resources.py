#!/usr/bin/env python
"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` remarks: one line per kernel (development tool).

    hipcc ... -Rpass-analysis=kernel-resource-usage -c vfx_conv.hip 2> build.log ; python tools/kernel_resources.py build.log [filter]
"""
import re
import subprocess
import sys


def main():
    log = open(sys.argv[1]).read()
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    cur = None
    rows = {}
    for ln in log.splitlines():
        m = re.search(r"remark:\s+Function Name: (\S+)", ln)
        if m:
            cur = m.group(1)
            rows[cur] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", ln)
        if m and cur:
            rows[cur][m.group(1).strip()] = int(m.group(2))
    names = list(rows)
    try:
        dem = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"] + names, capture_output=True, text=True).stdout.splitlines()
    except OSError:
        dem = names
    for n, d in zip(names, dem):
        if flt and flt not in d:
            continue
        r = rows[n]
        print("%-95s VGPR %3d AGPR %3d SGPR %3d spillS %2d spillV %2d occ %d scratch %d" % (
            d[:95], r.get("VGPRs", -1), r.get("AGPRs", -1), r.get("TotalSGPRs", -1), r.get("SGPRs Spill", -1),
            r.get("VGPRs Spill", -1), r.get("Occupancy", -1), r.get("ScratchSize", -1)))


if __name__ == "__main__":
    main()
