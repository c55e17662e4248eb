#!/bin/bash
# A/B of build variants on the GPU box (through gpurun):  bash tools/ab.sh "<extra cflags A>" "<extra cflags B>" ...
# Each variant is built with MPG_EXTRA_CFLAGS (fwd/bwd per-file flags can be given as FWD=... / BWD=... prefixes:
# "FWD=-mllvm x BWD= -DFOO"), benched with the default workload, and its per-kernel averages are printed.  The baseline
# ("") is run first and last.  Leaves the tree built with the shipped flags.
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
STEPS=${STEPS:-400}
run() {
  local spec="$1" fwd bwd extra
  unset MPG_FWD_CFLAGS MPG_BWD_CFLAGS
  extra="$spec"
  if [[ "$spec" == FWD=* ]]; then fwd="${spec#FWD=}"; fwd="${fwd%% BWD=*}"; extra="${spec#* BWD=}"; bwd="${extra%% EXTRA=*}"; extra="${extra#* EXTRA=}"; [[ "$extra" == "$bwd" ]] && extra=""; export MPG_FWD_CFLAGS="$fwd" MPG_BWD_CFLAGS="$bwd"; fi
  MPG_EXTRA_CFLAGS="$extra" python3 -m mpg_amd.build > /tmp/build.log 2>&1 || { echo "BUILD FAILED"; tail -5 /tmp/build.log; return; }
  MPG_EXTRA_CFLAGS="$extra" python3 bench.py --steps $STEPS --warmup 30 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); o=d['other_kernels_avg_ms']
r={d['roofline']['kernel'][:13]:d['roofline']['avg_ms'], d['roofline_other_rollout_kernel']['kernel'][:13]:d['roofline_other_rollout_kernel']['avg_ms']}
print('ms/step %.4f (median %.4f) fwd %.4f bwd %.4f target %.4f critic %.4f wgrad %.4f pol %.4f env %.4f' % (d['ms_per_step'], d['step_ms_median'], r['k_rollout_fwd'], r['k_rollout_bwd'], o['k_target_fused'], o['k_critic_fused'], o['k_wgrad_multi'], o['k_forward (worker policy)'], o['k_step_store_reset (env)']))"
}
echo "== [baseline]"; run ""
for V in "$@"; do echo "== [$V]"; run "$V"; done
echo "== [baseline]"; run ""
unset MPG_FWD_CFLAGS MPG_BWD_CFLAGS
python3 -m mpg_amd.build > /tmp/build.log 2>&1
