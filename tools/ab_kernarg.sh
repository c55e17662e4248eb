#!/bin/bash
# Where the kernel arguments live: HIP_FORCE_DEV_KERNARG=0 (host memory) against =1 (device memory), alternating on one box.
#   bash tools/ab_kernarg.sh        (through gpurun)
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
STEPS=${STEPS:-400}
for V in ${VARIANTS:-unset 0 1 0 1 unset}; do
  if [ "$V" = unset ]; then unset HIP_FORCE_DEV_KERNARG; else export HIP_FORCE_DEV_KERNARG=$V; fi
  echo "== HIP_FORCE_DEV_KERNARG=$V"
  MPG_BENCH_NO_F32=1 python3 bench.py --steps $STEPS --warmup 30 --no-cpu-baseline --no-side-configs 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); o=d['other_kernels_avg_ms']
r={d['roofline']['kernel'][:13]:d['roofline']['avg_ms'], d['roofline_other_rollout_kernel']['kernel'][:13]:d['roofline_other_rollout_kernel']['avg_ms']}
print('ms/step %.4f (median %.4f) fwd %.4f bwd %.4f target %.4f critic %.4f wgrad %.4f worker %.4f adam %.4f' % (d['ms_per_step'], d['step_ms_median'], r['k_rollout_fwd'], r['k_rollout_bwd'], o['k_target_fused'], o['k_critic_fused'], o['k_wgrad_multi'], o['k_step_store_reset (env)'], o['k_clip_adam_polyak']))"
done
