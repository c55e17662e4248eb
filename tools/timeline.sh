#!/bin/bash
# s_memtime timeline of the marked phases of k_target_fused (diagnostic build -DMPG_TIMELINE):  bash tools/timeline.sh
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
MPG_EXTRA_CFLAGS="-DMPG_TIMELINE $1" python3 -m mpg_amd.build > /tmp/build.log 2>&1 || { tail -20 /tmp/build.log; exit 1; }
python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline 2> /tmp/tl.err | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('target %.4f ms' % d['other_kernels_avg_ms']['k_target_fused'])"
grep -a "timeline" /tmp/tl.err | tail -${TL_LINES:-8}
python3 -m mpg_amd.build > /tmp/build.log 2>&1
