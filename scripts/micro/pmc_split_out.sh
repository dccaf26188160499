#!/bin/bash
# HBM traffic and SQ counters of ffn_out_split_kernel<0> in the form a cross-encoder layer runs (fp32 operand + GELU,
# residual read from a split row, only the split row written): three rocprofv3 --pmc passes over
# scripts/split_out_bench.py --only fused-chain -> profiles-style JSON in gpurun_out/pmc_split_out.json
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_split_out
rm -rf $O && mkdir -p $O
CMD="python3 $R/scripts/split_out_bench.py --tokens 1048576 --k 1536 --iters 3 --only fused-chain"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/p1 -- $CMD > $O/p1.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/p2 -- $CMD > $O/p2.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/p3 -- $CMD > $O/p3.log 2>&1
python3 - <<PY
import csv, glob, collections, json
tot = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob('$O/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'ffn_out_split_kernel' in r['Kernel_Name']:
            tot[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
rows, k = 1048576, 1536
launches = n['FETCH_SIZE']
fetch = tot['FETCH_SIZE'] * 1024 / launches
write = tot['WRITE_SIZE'] * 1024 / max(n['WRITE_SIZE'], 1)
alg = rows * (k * 4 + 1536 + 2304)
out = {
 'kernel': 'ffn_out_split_kernel<0> (rag_enc_linear_ln_split_chain_dev with an fp32 input: second feed-forward product + bias + GELU + residual from a split row + LayerNorm + split, no fp32 row written)',
 'workload': {'rows': rows, 'k': k, 'n': 384, 'command': 'rocprofv3 --pmc <counters> --output-format csv -- ' + '$CMD'.replace('$R/', '') + '  (separate runs for FETCH_SIZE, WRITE_SIZE and the SQ counters)'},
 'launches': launches,
 'FETCH_SIZE_KB_sum': tot['FETCH_SIZE'], 'WRITE_SIZE_KB_sum': tot['WRITE_SIZE'],
 'fetch_bytes_per_launch': fetch, 'fetch_bytes_per_launch_doubled': 2 * fetch, 'write_bytes_per_launch': write,
 'correction': 'gfx950: FETCH_SIZE counts 128-byte read requests as 64 bytes, so reads are doubled; WRITE_SIZE is exact',
 'traffic_bytes_per_launch': 2 * fetch + write,
 'algorithmic_bytes_per_launch': alg,
 'traffic_over_algorithmic': round((2 * fetch + write) / alg, 3),
 'traffic_bytes_per_row': (2 * fetch + write) / rows,
 'SQ': {c: tot[c] for c in sorted(tot) if c.startswith('SQ_')},
}
if tot.get('SQ_BUSY_CYCLES'):
    # MFMA busy cycles summed over SIMDs against busy cycles of the shader engines x 4 SIMDs x CUs: as in the earlier file,
    # the fraction of SIMD-cycles with the matrix pipe busy
    out['mfma_busy_fraction'] = round(tot['SQ_VALU_MFMA_BUSY_CYCLES'] / (tot['SQ_BUSY_CYCLES'] * 32), 3)
json.dump(out, open('$R/gpurun_out/pmc_split_out.json', 'w'), indent=1)
print(json.dumps({kk: out[kk] for kk in ('launches', 'fetch_bytes_per_launch_doubled', 'write_bytes_per_launch', 'traffic_over_algorithmic', 'traffic_bytes_per_row')}))
print(out['SQ'], out.get('mfma_busy_fraction'))
PY
