#!/bin/bash
# A/B of per-file compiler flags for fused_kernels.hip / mlp_kernels.hip:  bash tools/ab_flags.sh FUSED "<flags A>" "<flags B>" ...
# (first argument: FUSED, MLP or ENV).  Prints the bench step and per-kernel averages per variant; leaves the shipped build.
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
which=$1; shift
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); o=d["other_kernels_avg_ms"]; print("ms/step %.4f  fwd %.4f bwd %.4f target %.4f critic %.4f wgrad %.4f env %.4f" % (d["ms_per_step"], d["roofline_other_rollout_kernel"]["avg_ms"], d["roofline"]["avg_ms"], o["k_target_fused"], o["k_critic_fused"], o["k_wgrad_multi"], o.get("k_step_store_reset (env)", 0)))'
for V in "" "$@" ""; do
  echo "== [$which: $V]"
  if [ "$which" = FUSED ]; then MPG_FUSED_CFLAGS="$V" python3 -m mpg_amd.build --split-only > /tmp/build.log 2>&1 || { echo BUILD FAILED; continue; }
  elif [ "$which" = ENV ]; then MPG_ENV_CFLAGS="$V" python3 -m mpg_amd.build --split-only > /tmp/build.log 2>&1 || { echo BUILD FAILED; continue; }
  else MPG_MLP_CFLAGS="$V" python3 -m mpg_amd.build --split-only > /tmp/build.log 2>&1 || { echo BUILD FAILED; continue; }; fi
  MPG_BENCH_NO_F32=1 python3 bench.py --steps 600 --warmup 20 --no-cpu-baseline 2>/dev/null | python3 -c "$P"
done
python3 -m mpg_amd.build --split-only > /tmp/build.log 2>&1
