#!/bin/bash
# Which kernels sit either side of the device-to-device copies of a bench step?  (rocprofv3 kernel trace of a short run,
# reduced on the box to counts per (previous kernel, next kernel) pair -> gpurun_out/trace_copies.txt)
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trace_copies && mkdir -p /tmp/trace_copies $R/gpurun_out
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_copies/out -- python3 $R/bench.py --steps 3 --warmup 1 --no-extras --cpu-queries 0 > /tmp/trace_copies/line.log 2> /tmp/trace_copies/err.log
f=$(find /tmp/trace_copies/out -name '*kernel_trace.csv' | head -1)
python3 - "$f" > $R/gpurun_out/trace_copies.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
short = lambda n: n.replace('(anonymous namespace)::', '').replace('void ', '')[:60]
# the last third of the run: the timed steps
t0, t1 = int(rows[0]['Start_Timestamp']), int(rows[-1]['End_Timestamp'])
pairs, total, sizes = collections.Counter(), 0, collections.Counter()
last_steps = [r for r in rows if int(r['Start_Timestamp']) > t1 - (t1 - t0) * 0.2]
for i, r in enumerate(last_steps):
    if 'copyBuffer' in r['Kernel_Name'] and 0 < i < len(last_steps) - 1:
        d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
        key = (short(last_steps[i - 1]['Kernel_Name']), short(last_steps[i + 1]['Kernel_Name']))
        pairs[key] += 1
        sizes[key] += d
        total += d
print('window: last 20 %% of the trace, %d kernels, %d copies, %.3f ms in copies' % (len(last_steps), sum(pairs.values()), total * 1e-6))
for key, n in pairs.most_common(40):
    print('%5d x  avg %8.1f us   after %-60s before %s' % (n, sizes[key] / n * 1e-3, key[0], key[1]))
PY
cat $R/gpurun_out/trace_copies.txt | cut -c1-220
