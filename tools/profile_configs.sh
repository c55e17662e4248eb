#!/bin/bash
# rocprofv3 kernel statistics of the side configurations (C1 reference defaults, C3 NADP B=8192, C4 TD3+PER B=65536; the parity-test
# configurations of BASELINE.json, tools/bench_configs.py):  bash tools/profile_configs.sh <tag>   (through gpurun)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-cfgprof}
mkdir -p $OUT
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/bench_configs.py > $OUT/bench_configs.jsonl 2> $OUT/trace.log
cp $(ls $OUT/trace/*/*kernel_stats.csv | head -1) $OUT/configs_kernel_stats.csv
rm -rf $OUT/trace/*/*.db 2>/dev/null
cut -d, -f1-4 $OUT/configs_kernel_stats.csv | sed 's/(anonymous namespace):://g' | cut -c1-120 | head -30
grep -a "C1\|C3\|C4" $OUT/bench_configs.jsonl | cut -c1-160
