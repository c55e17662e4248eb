#!/bin/bash
# dW2-only weight gradients of the long jobs: 128-column slices (two workgroups per chunk, k_wgrad_w2_wide) against 64-column slices
# (four per chunk), alternating builds on one box; side lines c3 / c4.    bash tools/ab_wgrad_wide.sh     (through gpurun)
#   VARIANTS="..." overrides the list of MPG_EXTRA_CFLAGS; -DMPG_WG_WIDE_ADEPTH=1|2|4: pairs of A fragments requested ahead in the wide form
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
IFS='|' read -ra VS <<< "${VARIANTS:--DMPG_WGRAD_W2_WIDE_GROUPS=1000000000||-DMPG_WG_WIDE_ADEPTH=4|-DMPG_WG_WIDE_ADEPTH=1|-DMPG_WGRAD_W2_WIDE_GROUPS=1000000000||-DMPG_WG_WIDE_ADEPTH=4}"
for V in "${VS[@]}"; do
  MPG_EXTRA_CFLAGS="$V" python3 -m mpg_amd.build --split-only > /tmp/build.log 2>&1 || { echo "BUILD FAILED"; tail -5 /tmp/build.log; continue; }
  echo "== [$V]"
  for c in c3 c4; do
    MPG_EXTRA_CFLAGS="$V" MPG_BENCH_NO_F32=1 python3 bench.py --config $c --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); g=d.get('kernel_groups_ms_per_step') or {}
print('$c ms/step %.4f' % d['ms_per_step'], {k: round(v['ms_per_step'], 4) for k, v in g.items()})"
  done
done
python3 -m mpg_amd.build --split-only > /tmp/build.log 2>&1
