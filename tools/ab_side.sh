#!/bin/bash
# A/B of build variants on the side lines (c3 / c4), alternating on one box:  bash tools/ab_side.sh "<cflags A>" "<cflags B>" ...   (through gpurun)
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
for V in "$@"; do
  MPG_EXTRA_CFLAGS="$V" python3 -m mpg_amd.build --split-only > /tmp/build.log 2>&1 || { echo "BUILD FAILED"; tail -5 /tmp/build.log; continue; }
  echo "== [$V]"
  for c in ${CONFIGS:-c3 c4}; do
    MPG_EXTRA_CFLAGS="$V" MPG_BENCH_NO_F32=1 python3 bench.py --config $c --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); g=d.get('kernel_groups_ms_per_step') or {}
print('$c ms/step %.4f' % d['ms_per_step'], {k: round(v['ms_per_step'], 4) for k, v in g.items()})"
  done
done
python3 -m mpg_amd.build --split-only > /tmp/build.log 2>&1
