#!/bin/bash
# A/B of per-translation-unit flags for the two rollout sweeps:  bash tools/ab_sweeps.sh "<fwd flags>|<bwd flags>" ...
cd $GRAFT_REPO_ROOT
run() {
  MPG_FWD_CFLAGS="$1" MPG_BWD_CFLAGS="$2" python -m mpg_amd.build > /tmp/build.log 2>&1 || { tail -5 /tmp/build.log; return; }
  MPG_FWD_CFLAGS="$1" MPG_BWD_CFLAGS="$2" python bench.py --steps 400 --warmup 30 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms/step %.4f bwd %.4f fwd %.4f'%(d['ms_per_step'], d['roofline']['avg_ms'], d['roofline_other_rollout_kernel']['avg_ms']))"
}
echo "== default"; run "" ""
for V in "$@"; do F="${V%%|*}"; B="${V##*|}"; echo "== fwd [$F] bwd [$B]"; run "$F" "$B"; done
echo "== default"; run "" ""
python -m mpg_amd.build > /tmp/build.log 2>&1
