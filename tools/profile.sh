#!/bin/bash
# Round profile of the bench workload on the GPU box (run through gpurun):  bash tools/profile.sh <tag>
#   1. rocprofv3 --kernel-trace --stats of `python3 bench.py --steps 200 --warmup 20` -> gpurun_out/<tag>/kernel_stats.csv + bench line
#   2. the PMC passes of tools/pmc.sh (their own runs, no tracing)                  -> gpurun_out/<tag>/pmc_summary.csv, pmc_traffic.json
#   3. un-profiled bench lines: the driver's form (--steps 20 --warmup 5) three times and the default form once (that one with the CPU
#      baseline leg and the exact-fp32 engine timed in a child process); 4. the c3 / c4 side lines, kernel_resources.txt
# Copy what is to be judged from gpurun_out/<tag>/ into profiles/ (named per round) afterwards.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-prof}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export MPG_BENCH_NO_F32=1        # (no child process under the profiler; the un-profiled default run below measures the exact-fp32 engine)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/trace.log
cp $(ls $OUT/trace/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
bash tools/pmc.sh $TAG/pmc > $OUT/pmc.log 2>&1
cp $OUT/pmc/pmc_summary.csv $OUT/pmc/pmc_traffic.json $OUT/ 2>/dev/null
unset MPG_BENCH_NO_F32
for i in 1 2 3; do python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver_form_$i.json 2>/dev/null; done
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
# 4. the side configurations (BASELINE.json configs[2], [3]) and the kernels' register / spill / LDS usage
for c in c3 c4; do MPG_BENCH_NO_F32=1 python3 bench.py --config $c --no-cpu-baseline > $OUT/bench_$c.json 2>/dev/null; done
export MPG_BENCH_NO_F32=1
for c in c3 c4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$c -- python3 bench.py --config $c --no-cpu-baseline > /dev/null 2> $OUT/trace_$c.log
  cp $(ls $OUT/trace_$c/*/*kernel_stats.csv | head -1) $OUT/kernel_stats_$c.csv
  rm -rf $OUT/trace_$c
done
unset MPG_BENCH_NO_F32
python3 tools/kernel_resources.py > $OUT/kernel_resources.txt 2>&1
rm -rf $OUT/trace/*/*.db $OUT/pmc/pass*/*/*.db 2>/dev/null
head -12 $OUT/kernel_stats.csv | cut -c1-150
cat $OUT/pmc_traffic.json | head -20
for f in $OUT/bench_*.json; do python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], 'ms/step %.4f' % d['ms_per_step'], 'regions', ['%.4f' % x for x in d.get('region_ms_per_step', [])], d['roofline']['kernel'][:60], 'frac %.3f' % d['roofline']['frac'], d.get('cpu_baseline', {}).get('value'), d.get('cpu_baseline', {}).get('cores'))" $f; done
