#!/bin/bash
export MPG_BENCH_NO_F32=1   # no child processes under the profiler (their kernels would be averaged into this profile; bench.py also detects the preload itself)
# PMC passes of the bench workload on the GPU box (run through gpurun):  bash tools/pmc.sh <outdir-under-gpurun_out>
# Counters are collected in their own rocprofv3 runs (no tracing options), FETCH_SIZE and WRITE_SIZE in separate passes
# (TCC slots), as MI355X_MICROARCH.md prescribes.  Writes <outdir>/pmc_summary.csv (per-kernel averages per dispatch) and
# <outdir>/pmc_traffic.json with HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) KB: on gfx950 FETCH_SIZE reports
# half of the bytes of a wide (16 B/lane) coalesced read - which is what the stash reads are - WRITE_SIZE is exact.
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
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for p in sorted(glob.glob('$OUT/pass*/*/*counter_collection.csv')):
    for r in csv.DictReader(open(p)):
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').replace('rollout::', '').replace('mlp::', '')
        k = k.split('(')[0]
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
names = sorted({c for d in agg.values() for c in d})
with open('$OUT/pmc_summary.csv', 'w') as fh:
    w = csv.writer(fh)
    w.writerow(['kernel', 'dispatches'] + names)
    for k, d in sorted(agg.items()):
        if k.startswith('at::') or 'rocclr' in k:
            continue
        w.writerow([k, len(next(iter(d.values())))] + [round(sum(d[c]) / len(d[c]), 1) if c in d else '' for c in names])
traffic = {}
for k, d in agg.items():
    if 'FETCH_SIZE' in d and 'WRITE_SIZE' in d:
        f, wr = sum(d['FETCH_SIZE']) / len(d['FETCH_SIZE']), sum(d['WRITE_SIZE']) / len(d['WRITE_SIZE'])
        traffic[k.split('<')[0]] = int((2 * f + wr) * 1024)
json.dump({'unit': 'bytes per launch', 'formula': '(2*FETCH_SIZE + WRITE_SIZE) KB, gfx950 correction for wide coalesced reads',
           'workload': 'bench.py N=1 B=4096 n=25', 'bytes_per_launch': traffic}, open('$OUT/pmc_traffic.json', 'w'), indent=1)
for k in ('k_rollout_fwd', 'k_rollout_bwd', 'k_wgrad_multi'):
    print(k, traffic.get(k))
PY
