#!/bin/bash
# Round profile of the bench workload on the GPU box (run through gpurun):  bash tools/profile.sh <tag>
#   1. rocprofv3 --kernel-trace --stats of `python3 bench.py --steps 200 --warmup 20` -> gpurun_out/<tag>/kernel_stats.csv + bench line
#   2. the PMC passes of tools/pmc.sh (their own runs, no tracing)                  -> gpurun_out/<tag>/pmc_summary.csv, pmc_traffic.json
#   3. un-profiled bench lines: the driver's form (--steps 20 --warmup 5) three times and the default form once
# Copy what is to be judged from gpurun_out/<tag>/ into profiles/ (named per round) afterwards.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-prof}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/trace.log
cp $(ls $OUT/trace/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
bash tools/pmc.sh $TAG/pmc > $OUT/pmc.log 2>&1
cp $OUT/pmc/pmc_summary.csv $OUT/pmc/pmc_traffic.json $OUT/ 2>/dev/null
for i in 1 2 3; do python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver_form_$i.json 2>/dev/null; done
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
# 4. the exact-fp32 engine (-DMPG_F32_MFMA) on the driver's command -> exact_fp32_bench.json (copy to profiles/: bench.py reports it)
MPG_EXTRA_CFLAGS="-DMPG_F32_MFMA" python3 -m mpg_amd.build > $OUT/build_exact.log 2>&1 && \
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print(json.dumps({'ms_per_step': d['ms_per_step'], 'step_ms_median': d['step_ms_median'], 'from': 'bash tools/profile.sh $TAG: -DMPG_F32_MFMA build, python3 bench.py --gpus 1 --steps 20 --warmup 5, ' + d['device']['name'], 'rollout_kernels_ms': [d['roofline']['avg_ms'], d['roofline_other_rollout_kernel']['avg_ms']]}))" > $OUT/exact_fp32_bench.json
python3 -m mpg_amd.build > $OUT/build_default.log 2>&1
rm -rf $OUT/trace/*/*.db $OUT/pmc/pass*/*/*.db 2>/dev/null
head -12 $OUT/kernel_stats.csv | cut -c1-150
cat $OUT/pmc_traffic.json | head -20
for f in $OUT/bench_*.json; do python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], 'ms/step %.4f median %.4f max %.4f' % (d['ms_per_step'], d['step_ms_median'], d['step_ms_max']), d['roofline']['kernel'], 'avg_ms %.4f frac %.3f' % (d['roofline']['avg_ms'], d['roofline']['frac']), d.get('cpu_baseline', {}).get('value'), d.get('cpu_baseline', {}).get('cores'))" $f; done
