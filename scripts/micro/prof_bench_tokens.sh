#!/bin/bash
# rocprofv3 kernel stats of the bench step with a given cross-encoder mini-batch size (RAG_CE_TOKENS exported by the caller)
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_tok && mkdir -p $R/gpurun_out/prof_tok
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_tok/out -- python3 $R/bench.py --steps 4 --warmup 1 --no-extras --cpu-queries 0 > $R/gpurun_out/prof_tok/line.log 2> $R/gpurun_out/prof_tok/err.log
f=$(find $R/gpurun_out/prof_tok/out -name '*kernel_stats.csv' | head -1)
cp "$f" $R/gpurun_out/prof_tok/kernel_stats.csv
find $R/gpurun_out/prof_tok/out -name '*kernel_trace.csv' -delete
