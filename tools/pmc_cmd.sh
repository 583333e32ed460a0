#!/bin/bash
# usage: tools/pmc_cmd.sh <tag> "<counters>" <program> [args...] -- one rocprofv3 --pmc pass (no other tracing),
# per-kernel sums of each counter printed and left under gpurun_out/<tag>/.
TAG=$1; shift
CTRS=$1; shift
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc $CTRS --output-format csv -d "$OUT/pmc" -o c -- "$@" > "$OUT/cmd.log" 2>&1 < /dev/null
tail -2 "$OUT/cmd.log"
F=$(find "$OUT/pmc" -name "*counter_collection.csv" 2>/dev/null | head -1)
if [ -n "$F" ]; then python3 - "$F" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:60]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    n[(k, r["Counter_Name"])] += 1
for k in acc:
    if "mmrag" not in k: continue
    print(k)
    for c, v in sorted(acc[k].items()):
        print(f"   {c:34s} {v / n[(k, c)]:16.0f} per dispatch ({n[(k, c)]} dispatches)")
PY
else echo "no counter csv"; fi
find "$OUT" -name "*.db" -delete 2>/dev/null; true
