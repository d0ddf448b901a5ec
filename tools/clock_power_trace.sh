#!/bin/bash
# Samples the GPU's shader clock and socket power (rocm-smi) every 0.5 s while a command runs.
#   bash tools/clock_power_trace.sh <out-file> <command...>
OUT=$1; shift
( while true; do rocm-smi -d 0 --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|Socket Graphics|Average Graphics|Current Socket|junction|hotspot" | tr '\n' ' '; echo; sleep 0.5; done ) > $OUT &
SMI=$!
"$@"
kill $SMI
