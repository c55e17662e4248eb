#!/bin/bash
export MPG_BENCH_NO_F32=1   # no child processes under the profiler (their kernels would be averaged into this profile; bench.py also detects the preload itself)
# Attribution of SQ_LDS_BANK_CONFLICT in the network kernels (run through gpurun):  bash tools/lds_conflicts.sh
# Two builds - shipped, and -DMPG_AB_NO_IMGWRITE (the split-fp16 image stores of store_c_to_a dropped; results are garbage, only
# the counters matter) - each profiled with one rocprofv3 --pmc pass (no tracing) over a short bench run; prints per kernel
# LDS instructions, array cycles and conflict cycles of both and what the image stores account for.
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
. tools/ab_scaffold.sh      # the experiment branches live in archive/proto/ablation_macros.patch since round 6
OUT=gpurun_out/lds_conflicts
mkdir -p $OUT
for V in shipped noimg; do
  F=""; [ $V = noimg ] && F="-DMPG_AB_NO_IMGWRITE"
  MPG_EXTRA_CFLAGS="$F" python3 -m mpg_amd.build > /tmp/build.log 2>&1 || { tail -5 /tmp/build.log; exit 1; }
  rm -rf $OUT/$V
  rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_INSTS_VALU --output-format csv -d $OUT/$V -- python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline > $OUT/$V.log 2>&1
done
python3 -m mpg_amd.build > /tmp/build.log 2>&1
python3 - <<PY
import csv, glob, collections
def load(v):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for p in glob.glob('$OUT/%s/*/*counter_collection.csv' % v):
        for r in csv.DictReader(open(p)):
            k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').replace('rollout::', '').replace('mlp::', '').split('(')[0]
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
    return {k: {c: sum(x) / len(x) for c, x in d.items()} for k, d in agg.items()}
a, b = load('shipped'), load('noimg')
rows = [['kernel', 'LDS_insts', 'IDX_ACTIVE', 'BANK_CONFLICT', 'conflict_share', 'noimg_LDS_insts', 'noimg_IDX_ACTIVE', 'noimg_BANK_CONFLICT',
         'conflict_cycles_of_image_stores', 'share_of_all_conflicts', 'conflict_cycles_per_image_store_inst']]
for k in sorted(a):
    if k not in b or not any(s in k for s in ('k_rollout', 'k_critic', 'k_target', 'k_forward')):
        continue
    x, y = a[k], b[k]
    dc, di = x['SQ_LDS_BANK_CONFLICT'] - y['SQ_LDS_BANK_CONFLICT'], x['SQ_INSTS_LDS'] - y['SQ_INSTS_LDS']
    rows.append([k, int(x['SQ_INSTS_LDS']), int(x['SQ_LDS_IDX_ACTIVE']), int(x['SQ_LDS_BANK_CONFLICT']),
                 round(x['SQ_LDS_BANK_CONFLICT'] / x['SQ_LDS_IDX_ACTIVE'], 3), int(y['SQ_INSTS_LDS']), int(y['SQ_LDS_IDX_ACTIVE']),
                 int(y['SQ_LDS_BANK_CONFLICT']), int(dc), round(dc / max(x['SQ_LDS_BANK_CONFLICT'], 1), 3), round(dc / max(di, 1), 2)])
with open('$OUT/lds_conflict_attribution.csv', 'w') as fh:
    csv.writer(fh).writerows(rows)
for r in rows:
    print(','.join(str(v) for v in r))
PY
