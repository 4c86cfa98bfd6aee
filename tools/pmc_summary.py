#!/usr/bin/env python3
"""Reduce rocprofv3 --pmc counter_collection CSVs to the per-kernel JSON summaries kept in profiles/.

  python tools/pmc_summary.py hbm  fetch_counter_collection.csv write_counter_collection.csv out.json
  python tools/pmc_summary.py mfma m_counter_collection.csv out.json

hbm : FETCH_SIZE / WRITE_SIZE come from two separate passes of the same bench command
      (gpurun refuses mixed trace domains; the guide asks for separate --pmc passes).  Both are
      in KB; on gfx950 FETCH_SIZE reports half of the bytes read (MI355X_MICROARCH.md, HBM section, for
      16 B/lane loads; `calib` below measured the same factor for the dword buffer loads the Winograd
      kernels issue), so bytes = (2*FETCH + WRITE) * 1024, averaged per launch.
mfma: mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8 XCDs).
"""
import csv
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def build_id():
    """vfx_build_id() of the in-tree library the profiled command loaded (bench.py refuses a stale summary)."""
    try:
        from voicefixer_amd import _lib
        return _lib.lib().vfx_build_id().decode()
    except Exception as e:  # noqa: BLE001 - a summary without a stamp is still a summary
        return "unknown (%s)" % e


def read(path):
    per = defaultdict(lambda: defaultdict(float))
    disp = defaultdict(set)
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            k = row["Kernel_Name"]
            per[k][row["Counter_Name"]] += float(row["Counter_Value"])
            disp[k].add(row["Dispatch_Id"])
    return per, {k: len(v) for k, v in disp.items()}


def hbm(fetch_csv, write_csv, out):
    f, nf = read(fetch_csv)
    w, nw = read(write_csv)
    res = {}
    for k in f:
        if k not in w or not any(t in k for t in ("conv_taps", "convw", "convtw", "resblk", "gru", "stft", "conv_cout1")):
            continue
        fa = f[k]["FETCH_SIZE"] / nf[k]
        wa = w[k]["WRITE_SIZE"] / nw[k]
        res[k] = {"launches": nf[k], "FETCH_SIZE_KB_avg": fa, "WRITE_SIZE_KB_avg": wa,
                  "hbm_bytes_per_launch": int((2 * fa + wa) * 1024)}
    doc = {
        "lib_build_id": build_id(),
        "command": "rocprofv3 --pmc FETCH_SIZE (pass 1) / --pmc WRITE_SIZE (pass 2) --kernel-trace -- "
                   "python bench.py --steps 1 --warmup 1 --batch 32 --no-cpu-baseline",
        "correction": "gfx950: FETCH_SIZE reports exactly 1/2 of the bytes read -- calibrated on this library's own "
                      "access widths with tools/probe/fetch_calib.hip (1 GiB streams: dword buffer loads 0.5000, 16-byte "
                      "buffer loads 0.5000; WRITE_SIZE exact for dword, 16-byte and half-wave-row dword stores: "
                      "profiles/r03_fetch_size_calibration.json) -> bytes = (2*FETCH + WRITE) * 1024; counters are KB",
        "kernels": res,
    }
    json.dump(doc, open(out, "w"), indent=1)


def durations(trace_csv):
    """kernel name -> total duration (ns) and dispatch count, from the kernel_trace.csv of the SAME pass."""
    tot = defaultdict(float)
    cnt = defaultdict(int)
    with open(trace_csv, newline="") as f:
        for row in csv.DictReader(f):
            tot[row["Kernel_Name"]] += int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
            cnt[row["Kernel_Name"]] += 1
    return tot, cnt


def mfma(m_csv, out, trace_csv=None):
    m, n = read(m_csv)
    dur = durations(trace_csv)[0] if trace_csv else {}
    res = {}
    for k, c in m.items():
        if not any(t in k for t in ("conv_taps", "convw", "convtw", "resblk", "gru", "stft", "conv_cout1")):
            continue
        gui = c.get("GRBM_GUI_ACTIVE", 0.0)
        if gui <= 0:
            continue
        simd_cycles = 1024.0 * gui / 8.0
        ins_m = c.get("SQ_INSTS_MFMA", 0.0)
        res[k] = {
            "launches": n[k],
            "mfma_util": round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / simd_cycles, 4),
            "mfma_insts": ins_m,
            "valu_insts_per_mfma": round(c.get("SQ_INSTS_VALU", 0.0) / max(ins_m, 1.0), 2),
            "avg_waves_per_simd": round(4.0 * c.get("SQ_WAVE_CYCLES", 0.0) / simd_cycles, 2),
            "wait_inst_frac": round(c.get("SQ_WAIT_INST_ANY", 0.0) / max(c.get("SQ_WAVE_CYCLES", 1.0), 1.0), 3),
            "wait_any_frac": round(c.get("SQ_WAIT_ANY", 0.0) / max(c.get("SQ_WAVE_CYCLES", 1.0), 1.0), 3),
        }
        if dur.get(k):
            # shader clock while the kernel ran: cycles per XCD / duration, both from this pass
            res[k]["shader_clock_ghz"] = round(gui / 8.0 / dur[k], 3)
            res[k]["avg_launch_ms_in_this_pass"] = round(dur[k] / n[k] / 1e6, 4)
    doc = {
        "lib_build_id": build_id(),
        "command": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES "
                   "GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace -- python bench.py --steps 1 "
                   "--warmup 1 --batch 32 --no-cpu-baseline",
        "definition": "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); GRBM_GUI_ACTIVE "
                      "is summed over the 8 XCDs; SQ_WAVE_CYCLES counts quad-cycles; shader_clock_ghz = GRBM_GUI_ACTIVE / 8 / "
                      "kernel duration of the same pass (the 157.3 TFLOP/s fp32-MFMA peak assumes 2.4 GHz: a time-based "
                      "fraction of that peak is mfma_util x clock / 2.4)",
        "kernels": res,
    }
    json.dump(doc, open(out, "w"), indent=1)


def calib(fetch_csv, write_csv, out):
    """tools/probe/fetch_calib.hip under the two PMC passes: counter bytes / known bytes per access width."""
    known = float(1 << 30)
    f, nf = read(fetch_csv)
    w, nw = read(write_csv)
    res = {}
    for k in sorted(set(f) | set(w)):
        if "calib_" not in k:
            continue
        name = k.split("(")[0]
        res[name] = {"known_bytes": int(known),
                     "FETCH_SIZE_bytes": f[k]["FETCH_SIZE"] / nf[k] * 1024 if k in f else None,
                     "WRITE_SIZE_bytes": w[k]["WRITE_SIZE"] / nw[k] * 1024 if k in w else None}
        r = res[name]
        r["fetch_over_known"] = round(r["FETCH_SIZE_bytes"] / known, 4) if r["FETCH_SIZE_bytes"] is not None else None
        r["write_over_known"] = round(r["WRITE_SIZE_bytes"] / known, 4) if r["WRITE_SIZE_bytes"] is not None else None
    json.dump({"note": "counter KB x 1024 / bytes the kernel is known to move (1 GiB, > the 256 MB Infinity Cache); "
                       "read kernels should show fetch_over_known = 1 if the counter were exact",
               "kernels": res}, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "hbm":
        hbm(*sys.argv[2:5])
    elif sys.argv[1] == "calib":
        calib(*sys.argv[2:5])
    else:
        mfma(*sys.argv[2:5])
