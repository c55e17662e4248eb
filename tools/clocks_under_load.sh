#!/bin/bash
# Shader clock and socket power while the bench step (or a side configuration) runs back to back:  bash tools/clocks_under_load.sh [c3|c4]
# (read-only rocm-smi queries once per second beside a long bench run; through gpurun)
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
CFG=${1:-}
if [ -n "$CFG" ]; then ARGS="--config $CFG --steps 4000"; else ARGS="--steps 20000 --no-side-configs"; fi
MPG_BENCH_NO_F32=1 python3 bench.py $ARGS --warmup 20 --no-cpu-baseline > /tmp/clk_bench.json 2>/dev/null &
BP=$!
sleep 25
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power\|mclk" | tr -s ' ' | head -6
  echo --
  sleep 1
done
wait $BP
tail -1 /tmp/clk_bench.json | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms/step %.4f' % d['ms_per_step'])"
