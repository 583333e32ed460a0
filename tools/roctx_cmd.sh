#!/bin/bash
# usage: tools/roctx_cmd.sh <tag> <program> [args...]  -- rocprofv3 marker + kernel trace of one command with the
# package's roctx ranges on (MMRAG_ROCTX=1); leaves the stats CSVs under gpurun_out/<tag>/ and prints the marker table.
TAG=$1; shift
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp MMRAG_ROCTX=1
timeout -k 10 300 rocprofv3 --marker-trace --kernel-trace --stats --output-format csv -d "$OUT/prof" -o m -- "$@" > "$OUT/cmd.log" 2>&1 < /dev/null
tail -3 "$OUT/cmd.log"
for f in $(find "$OUT/prof" -name "*marker*stats*.csv" -o -name "*domain_stats.csv" 2>/dev/null); do echo "== $f"; cut -c1-160 "$f" | head -12; done
find "$OUT" -name "*.db" -delete 2>/dev/null; true
