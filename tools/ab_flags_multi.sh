#!/bin/bash
# A/B of several flag sets against the default build: bash tools/ab_flags_multi.sh "<flags1>" "<flags2>" ...
cd $GRAFT_REPO_ROOT
run() {
  MPG_EXTRA_CFLAGS="$1" python -m mpg_amd.build > /tmp/build.log 2>&1 || { tail -5 /tmp/build.log; return; }
  python bench.py --steps 400 --warmup 30 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms/step %.4f bwd %.4f fwd %.4f wgrad %.4f'%(d['ms_per_step'], d['roofline']['avg_ms'], d['roofline_other_rollout_kernel']['avg_ms'], d['wgrad_kernel']['avg_ms']))"
}
echo "== default"; run ""
for F in "$@"; do echo "== [$F]"; run "$F"; done
echo "== default"; run ""
MPG_EXTRA_CFLAGS="" python -m mpg_amd.build > /tmp/build.log 2>&1
