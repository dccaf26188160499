#!/bin/bash
# How much of the timed steps the GPU sits idle: rocprofv3 kernel trace of the bench, busy time = union of the kernels'
# [start, end) intervals, over the span of the last eight steps' kernels; the largest gaps are listed with the kernels around them.
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/idle && mkdir -p /tmp/idle
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/idle/out -- python3 $R/bench.py --steps 8 --warmup 3 --no-extras --cpu-queries 0 > /tmp/idle/line.log 2> /tmp/idle/err.log
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/idle/out/**/*kernel_trace.csv', recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:60]))
rows.sort()
# the steps are delimited by fuse_kernel launches (one per step): take the last 8 + stage pass
fuse = [i for i, r in enumerate(rows) if 'fuse_kernel' in r[2]]
lo = fuse[-9]; hi = fuse[-1]
seg = rows[lo:hi]
span = seg[-1][1] - seg[0][0]
busy = 0; cur_s, cur_e = seg[0][0], seg[0][1]; gaps = []
for s, e, n in seg[1:]:
    if s > cur_e:
        gaps.append((s - cur_e, cur_e, n)); busy += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print('steps', len(fuse[-9:-1]), 'span ms', span / 1e6, 'busy ms', busy / 1e6, 'idle %', 100 * (1 - busy / span), 'per step idle ms', (span - busy) / 1e6 / 8)
gaps.sort(reverse=True)
for g, t, n in gaps[:16]:
    print('gap us %.1f before %s' % (g / 1e3, n))
PY
