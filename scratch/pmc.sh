#!/bin/bash
# usage: scratch/pmc.sh <outdir> ; runs PMC passes of bench.py (short) ; counters per pass below
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
mkdir -p $OUT
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" \
         "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_ANY" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d $OUT/pass$i -- python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline > $OUT/pass$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for p in sorted(glob.glob('$OUT/pass*/*/*counter_collection.csv')):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(p)):
        k = r['Kernel_Name'][:48]
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
    print('==', p.split('/')[-3])
    for k, d in agg.items():
        if 'rollout' in k or 'k_forward' in k or 'k_step' in k or 'wgrad' in k:
            print(k, {c: round(sum(v)/len(v), 1) for c, v in d.items()}, 'n=%d' % len(next(iter(d.values()))))
PY
