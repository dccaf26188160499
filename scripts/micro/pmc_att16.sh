#!/bin/bash
# Counters of one of the 16-bit kernels between the feed-forward blocks (scripts/micro/att16_bench.py --only $1),
# separate rocprofv3 --pmc passes; $2 = substring of the kernel's name.
#   scripts/micro/pmc_att16.sh fused qkv_attention_kernel --pack      scripts/micro/pmc_att16.sh qkv linear384_kernel        scripts/micro/pmc_att16.sh attention attention_kernel
set -e
ONLY=$1; KERNEL=$2; EXTRA=${3:-}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_$ONLY
rm -rf $O && mkdir -p $O
run() { timeout -s KILL 120 rocprofv3 --pmc $2 --output-format csv -d $O/$1 -- python3 $R/scripts/micro/att16_bench.py --only $ONLY --iters 3 $EXTRA > $O/$1.log 2>&1; }
run fetch "FETCH_SIZE"
run write "WRITE_SIZE"
run sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES"
run sq2 "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM"
run tcp "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TA_TCP_STATE_READ_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum"
python3 - <<PY
import csv, glob, collections, json
out = {}
for pss in ('fetch', 'write', 'sq1', 'sq2', 'tcp'):
    tot = collections.defaultdict(float); n = collections.Counter()
    for f in glob.glob('$O/%s/**/*counter_collection.csv' % pss, recursive=True):
        for r in csv.DictReader(open(f)):
            if '$KERNEL' in r['Kernel_Name']:
                tot[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
    for k in sorted(tot):
        out[k] = {'sum': tot[k], 'launches': n[k]}
        print(pss, k, tot[k], 'launches', n[k])
json.dump(out, open('$O/summary.json', 'w'), indent=1)
PY
grep "$ONLY (" $O/sq1.log
