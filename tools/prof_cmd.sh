#!/bin/bash
# usage: tools/prof_cmd.sh <tag> <program> [args...]   -- rocprofv3 kernel stats of one command on the GPU box;
# prints the top of the per-kernel table and leaves the CSVs under gpurun_out/<tag>/.
TAG=$1; shift
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o p -- "$@" > "$OUT/cmd.log" 2>&1 < /dev/null
tail -4 "$OUT/cmd.log"
F=$(find "$OUT/prof" -name "*kernel_stats.csv" 2>/dev/null | head -1)
if [ -n "$F" ]; then cp "$F" "$OUT/kernel_stats.csv"; cut -c1-200 "$F" | head -14; else echo "no kernel_stats.csv"; fi
find "$OUT" -name "*.db" -delete 2>/dev/null; true
