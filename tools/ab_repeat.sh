#!/bin/bash
# Distribution of the per-kernel averages over N fresh processes per build variant (the target kernel's time differs
# between processes of one box: 26 / 33 / 39 us):  bash tools/ab_repeat.sh N "<cflags A>" "<cflags B>" ...
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
N=${1:-5}; shift
STEPS=${STEPS:-300}
for V in "$@"; do
  echo "== [$V]"
  MPG_EXTRA_CFLAGS="$V" python3 -m mpg_amd.build > /tmp/build.log 2>&1 || { echo "BUILD FAILED"; tail -5 /tmp/build.log; continue; }
  for i in $(seq $N); do
    python3 bench.py --steps $STEPS --warmup 30 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); o=d['other_kernels_avg_ms']
print('  ms/step %.4f target %.4f critic %.4f wgrad %.4f pol %.4f env %.4f' % (d['ms_per_step'], o['k_target_fused'], o['k_critic_fused'], o['k_wgrad_multi'], o['k_forward (worker policy)'], o['k_step_store_reset (env)']))"
  done
done
python3 -m mpg_amd.build > /tmp/build.log 2>&1
