#!/usr/bin/env python3
"""Reduce the counter passes of tools/pmc_diag.sh: per kernel, every counter averaged per launch and as a fraction of the
launch's SIMD cycles (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs).  SQ wave-cycle
counters that the hardware reports in quad-cycles are listed with the x4 column, so nothing hides behind a unit guess."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
# per[kernel][counter] = [sum, set(dispatch ids)] -- a counter that appears in several passes (GRBM_GUI_ACTIVE) is
# kept per pass and averaged over passes
per = defaultdict(lambda: defaultdict(list))
for pdir in sorted(glob.glob(os.path.join(out, "p*"))):
    if not os.path.isdir(pdir):
        continue
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, set()]))
    for path in glob.glob(os.path.join(pdir, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                k = row["Kernel_Name"]
                if "conv" not in k and "gru" not in k:
                    continue
                e = acc[k][row["Counter_Name"]]
                e[0] += float(row["Counter_Value"])
                e[1].add(row["Dispatch_Id"])
    for k, cs in acc.items():
        for name, (tot, ids) in cs.items():
            per[k][name].append(tot / max(1, len(ids)))
for k in sorted(per):
    c = {name: sum(v) / len(v) for name, v in per[k].items()}
    gui = c.get("GRBM_GUI_ACTIVE", 0.0)
    simd = 1024.0 * gui / 8.0
    print("==", k[:120])
    print("   launch: GRBM_GUI_ACTIVE/8 = %.0f cycles per XCD" % (gui / 8.0))
    for name in sorted(c):
        v = c[name]
        frac = v / simd if simd > 0 else float("nan")
        print("  %-34s %16.0f   per SIMD-cycle %8.4f   x4 %8.4f" % (name, v, frac, 4 * frac))
