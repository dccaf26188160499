#!/bin/bash
# rocprofv3 kernel trace + stats of the default bench step (no extras, no host baseline): summary -> gpurun_out/prof_bench
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_bench && mkdir -p $R/gpurun_out/prof_bench
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench/out -- python3 $R/bench.py --steps 8 --warmup 2 --no-extras --cpu-queries 0 > $R/gpurun_out/prof_bench/line.log 2> $R/gpurun_out/prof_bench/err.log
f=$(find $R/gpurun_out/prof_bench/out -name '*kernel_stats.csv' | head -1)
cp "$f" $R/gpurun_out/prof_bench/kernel_stats.csv
find $R/gpurun_out/prof_bench/out -name '*kernel_trace.csv' -delete
head -30 $R/gpurun_out/prof_bench/kernel_stats.csv | cut -c1-200
