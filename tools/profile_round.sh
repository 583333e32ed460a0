#!/bin/bash
# Evidence pass on the GPU box (run through gpurun): bench line, rocprofv3 kernel stats and PMC passes of the retrieve
# leg and of the embed leg.  Outputs under gpurun_out/<tag>/; tools/collect_profiles.py turns them into profiles/<tag>_*.
# usage: tools/profile_round.sh <tag>
TAG=${1:-r03}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python3 bench.py --no-cpu-baseline --no-embed --no-served --no-peaks"
prof() {   # prof <subdir> <rocprofv3 args...> -- <command...>
    local sub=$1; shift
    timeout -k 10 400 rocprofv3 "$@" > "$OUT/$sub.log" 2>&1 < /dev/null
    find "$OUT/$sub" -name "*.db" -delete 2>/dev/null
    echo "[profile_round] $sub: $(find "$OUT/$sub" -name '*.csv' | wc -l) csv files"
}
timeout -k 10 600 python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "[profile_round] bench rc=$? $(cut -c1-160 "$OUT/bench.json")"
prof stats --kernel-trace --stats --output-format csv -d "$OUT/stats" -o s -- $BENCH
prof pmc_fetch --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o c -- $BENCH --steps 10 --warmup 2
prof pmc_write --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o c -- $BENCH --steps 10 --warmup 2
prof pmc_mfma --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_mfma" -o c -- $BENCH --steps 10 --warmup 2
prof embed_stats --kernel-trace --stats --output-format csv -d "$OUT/embed_stats" -o s -- python3 tools/embed_once.py
prof embed_mfma --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/embed_mfma" -o c -- python3 tools/embed_once.py
prof embed_other_stats --kernel-trace --stats --output-format csv -d "$OUT/embed_other_stats" -o s -- python3 tools/embed_other_once.py
ls "$OUT"
