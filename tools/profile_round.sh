#!/bin/bash
# Take the profile set of one round on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh r02 [extra bench args]
# Writes gpurun_out/prof_<tag>/: rocprofv3 kernel-trace stats of the bench command, three separate PMC passes
# (FETCH_SIZE / WRITE_SIZE / matrix-pipe counters; the guide asks for separate --pmc passes, gpurun refuses mixed trace
# domains) reduced by tools/pmc_summary.py, and the bench lines printed under the profiler.
TAG=${1:-rXX}; shift
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
# --streams 1: one batch at a time, so that a kernel's duration in the trace is ITS duration (with the default two streams the
# launches of two batches overlap and every duration contains whatever the other stream ran meanwhile); bench.py's roofline takes
# its launch durations from a single-stream leg for the same reason, and the two must agree
BENCH="python $PWD/bench.py --batch 32 --streams 1 --no-cpu-baseline --no-bf16x3 --no-host-leg $*"
cd /tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats -o r -- $BENCH --steps 3 --warmup 1 > $OUT/bench_under_rocprof.json 2> $OUT/stats.log
rocprofv3 --output-format csv --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc -o fetch -- $BENCH --steps 1 --warmup 1 > /dev/null 2> $OUT/fetch.log
rocprofv3 --output-format csv --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc -o write -- $BENCH --steps 1 --warmup 1 > /dev/null 2> $OUT/write.log
rocprofv3 --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace -d $OUT/pmc -o m -- $BENCH --steps 1 --warmup 1 > /dev/null 2> $OUT/m.log
cd - > /dev/null
F=$(find $OUT/pmc -name "fetch_counter_collection.csv" | head -1); W=$(find $OUT/pmc -name "write_counter_collection.csv" | head -1); M=$(find $OUT/pmc -name "m_counter_collection.csv" | head -1)
python tools/pmc_summary.py hbm "$F" "$W" $OUT/pmc_hbm_traffic.json
MT=$(find $OUT/pmc -name "m_kernel_trace.csv" | head -1)
python tools/pmc_summary.py mfma "$M" $OUT/pmc_mfma_util.json "$MT"
S=$(find $OUT/stats -name "r_kernel_stats.csv" | head -1); cp "$S" $OUT/kernel_stats.csv
# dispatch order and duration of every kernel of the LAST timed step (what the stats average over)
T=$(find $OUT/stats -name "r_kernel_trace.csv" | head -1)
python - "$T" $OUT/per_dispatch_last_step.txt <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("void stft_mel_kernel")]
with open(sys.argv[2], "w") as f:
    f.write("# last step of the bench command under rocprofv3 --kernel-trace: kernel, duration (ms), in dispatch order\n")
    for r in rows[starts[-1]:] if starts else rows:
        f.write("%s\t%.3f\n" % (r["Kernel_Name"][:70], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
PY
# keep the merge-back small: raw counter CSVs are large
rm -rf $OUT/pmc $OUT/stats
ls -la $OUT
