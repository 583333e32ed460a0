#!/bin/bash
# Full evidence pass on the GPU box: tests, bench, rocprofv3 kernel stats, PMC traffic.
# usage: tools/profile_round.sh <tag>     (outputs under gpurun_out/<tag>/)
set -e
TAG=${1:-r01}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1 && tail -1 $OUT/pytest_gpu.log
timeout -k 10 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err && cat $OUT/bench.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python bench.py --no-cpu-baseline --no-embed > $OUT/bench_under_rocprof.json 2> $OUT/rocprof_stats.err
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- python bench.py --no-cpu-baseline --no-embed --steps 5 --warmup 2 > /dev/null 2> $OUT/rocprof_fetch.err
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- python bench.py --no-cpu-baseline --no-embed --steps 5 --warmup 2 > /dev/null 2> $OUT/rocprof_write.err
find $OUT -name "*.db" -delete; find $OUT -name "*agent_info*" -delete
ls -R $OUT | head -40
