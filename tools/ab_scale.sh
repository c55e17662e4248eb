#!/bin/bash
# per-step / fixed cost of the rollout sweeps for build variants:  bash tools/ab_scale.sh "<cflags A>" "<cflags B>" ...
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
for V in "$@"; do
  echo "== [$V]"
  MPG_EXTRA_CFLAGS="$V" python3 -m mpg_amd.build > /tmp/build.log 2>&1 || { echo "BUILD FAILED"; tail -5 /tmp/build.log; continue; }
  python3 tools/roll_scale.py 2>&1 | tail -2
done
python3 -m mpg_amd.build > /tmp/build.log 2>&1
