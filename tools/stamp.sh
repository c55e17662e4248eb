#!/bin/bash
# per-wave cycle accounting of the rollout sweeps' step phases (diagnostic build -DMPG_STAMP):  bash tools/stamp.sh
cd $GRAFT_REPO_ROOT
MPG_EXTRA_CFLAGS="-DMPG_STAMP" python3 -m mpg_amd.build > /tmp/build.log 2>&1 || { tail -20 /tmp/build.log; exit 1; }
python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline 2> /tmp/stamp.err > /dev/null
grep -a "stamp" /tmp/stamp.err | tail -16
python3 -m mpg_amd.build > /tmp/build.log 2>&1
