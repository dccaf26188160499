#!/bin/bash
# Counters of mlp_fused_kernel (float16 instance): separate rocprofv3 --pmc passes over scripts/mlp_bench.py
# (memory-side bytes: FETCH_SIZE and WRITE_SIZE each in a pass of their own; then two passes of SQ counters).
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_mlp
rm -rf $O && mkdir -p $O
run() { timeout -s KILL 120 rocprofv3 --pmc $2 --output-format csv -d $O/$1 -- python3 $R/scripts/mlp_bench.py --dtype fp16 --only-mlp --iters 3 > $O/$1.log 2>&1; }
run fetch "FETCH_SIZE"
run write "WRITE_SIZE"
run sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES"
run sq2 "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA"
python3 - <<PY
import csv, glob, collections, json
out = {}
for pss in ('fetch', 'write', 'sq1', 'sq2'):
    tot = collections.defaultdict(float); n = collections.Counter()
    for f in glob.glob('$O/%s/**/*counter_collection.csv' % pss, recursive=True):
        for r in csv.DictReader(open(f)):
            if 'mlp_fused_kernel' in r['Kernel_Name']:
                tot[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
    for k in sorted(tot):
        out[k] = {'sum': tot[k], 'launches': n[k]}
        print(pss, k, tot[k], 'launches', n[k])
json.dump(out, open('$O/summary.json', 'w'), indent=1)
PY
grep "feed-forward block" $O/sq1.log
