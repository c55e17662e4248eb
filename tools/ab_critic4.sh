#!/bin/bash
# same-box A/B of the balanced critic launch (k_critic_fused4) against k_critic_fused:  bash tools/ab_critic4.sh   (through gpurun)
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
export MPG_BENCH_NO_F32=1
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); o=d["other_kernels_avg_ms"]; print("ms/step %.4f (median %.4f) fwd %.4f bwd %.4f target %.4f critic %.4f wgrad %.4f pol %.4f env %.4f adam %.4f" % (d["ms_per_step"], d["step_ms_median"], d["roofline_other_rollout_kernel"]["avg_ms"], d["roofline"]["avg_ms"], o["k_target_fused"], o["k_critic_fused"], o["k_wgrad_multi"], o["k_forward (worker policy)"], o["k_step_store_reset (env)"], o["k_clip_adam_polyak"]))'
run() { python3 bench.py --steps 400 --warmup 30 --no-cpu-baseline 2>/dev/null | python3 -c "$P"; }
for V in "" "-DMPG_CRITIC4_MIN_GROUPS=100000000" "" "-DMPG_CRITIC4_MIN_GROUPS=100000000"; do
  echo "== [${V:-balanced critic (shipped)}]"
  MPG_EXTRA_CFLAGS="$V" python3 -m mpg_amd.build --split-only > /tmp/b.log 2>&1 || { tail -3 /tmp/b.log; continue; }
  run
done
python3 -m mpg_amd.build --split-only > /tmp/b.log 2>&1
