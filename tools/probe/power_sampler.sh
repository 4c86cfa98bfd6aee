#!/bin/bash
# Sample socket power and shader clock with rocm-smi every ~0.2 s while "$@" runs:  bash tools/probe/power_sampler.sh <out.txt> <command...>
OUT=$1; shift
( while true; do echo "$(date +%s.%N) $(rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'Power|sclk' | tr -s ' ' | tr '\n' ';')"; sleep 0.2; done ) > $OUT 2>&1 &
S=$!
"$@"
kill $S 2>/dev/null
