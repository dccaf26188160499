#!/bin/bash
# per-kernel times of a dense search at small batch sizes (rocprofv3 --kernel-trace --stats)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for b in 1 32; do
  rm -rf $R/gpurun_out/prof_dense_$b
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_dense_$b -- python3 $R/scripts/kernel_bench.py --stage dense --batch $b --iters 20 > /dev/null 2>&1 || exit 1
  f=$(find $R/gpurun_out/prof_dense_$b -name '*kernel_stats.csv' | head -1)
  echo "== batch $b"; python3 - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print('%-70s calls %4s avg %8.1f us total %8.2f ms' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
  find $R/gpurun_out/prof_dense_$b -name '*kernel_trace.csv' -delete
done
