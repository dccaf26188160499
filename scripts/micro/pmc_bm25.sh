#!/bin/bash
# SQ counters of bm25_tile_kernel: two rocprofv3 --pmc passes over scripts/kernel_bench.py --stage bm25
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_bm25
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/pmc_bm25/p1 -- python3 $R/scripts/kernel_bench.py --stage bm25 --iters 3 > $R/gpurun_out/pmc_bm25/p1.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA --output-format csv -d $R/gpurun_out/pmc_bm25/p2 -- python3 $R/scripts/kernel_bench.py --stage bm25 --iters 3 > $R/gpurun_out/pmc_bm25/p2.log 2>&1
python3 - <<PY
import csv, glob, collections
for pss in ('p1','p2'):
    tot = collections.defaultdict(float); n = collections.Counter()
    for f in glob.glob('$R/gpurun_out/pmc_bm25/%s/**/*counter_collection.csv' % pss, recursive=True):
        for r in csv.DictReader(open(f)):
            if 'bm25_tile_kernel' in r['Kernel_Name']:
                tot[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
    for k in sorted(tot): print(pss, k, tot[k], 'launches', n[k])
PY
