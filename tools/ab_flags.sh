#!/bin/bash
# A/B of extra compiler flags for the whole library: bash tools/ab_flags.sh "<flags>"
cd $GRAFT_REPO_ROOT
for V in "" "$1" "" "$1"; do
  MPG_EXTRA_CFLAGS="$V" python -m mpg_amd.build > /tmp/build.log 2>&1 || { tail -5 /tmp/build.log; exit 1; }
  echo "== variant [$V]"
  python bench.py --steps 400 --warmup 30 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms/step %.4f %s %.4f %s %.4f wgrad %.4f'%(d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_ms'], d['roofline_other_rollout_kernel']['kernel'], d['roofline_other_rollout_kernel']['avg_ms'], d['wgrad_kernel']['avg_ms']))"
done
MPG_EXTRA_CFLAGS="" python -m mpg_amd.build > /tmp/build.log 2>&1
