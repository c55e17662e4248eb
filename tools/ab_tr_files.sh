#!/bin/bash
# Round 5: the transposed activation image (-DMPG_TR_IMAGE, bit-identical) per translation unit - the LDS image is kernel-internal,
# so the layout may differ between translation units.  A/B of the bench step (fused_kernels.hip: target / critic / weight-gradient
# launches; env_path_tracking.hip: the worker launch) and of the C4 / C3 side lines (mlp_kernels.hip: k_forward / k_backward):
#   bash tools/ab_tr_files.sh
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
. tools/ab_scaffold.sh      # the experiment branches live in archive/proto/ablation_macros.patch since round 6
export MPG_BENCH_NO_F32=1
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); o=d["other_kernels_avg_ms"]; print("ms/step %.4f  fwd %.4f bwd %.4f target %.4f critic %.4f wgrad %.4f worker %.4f" % (d["ms_per_step"], d["roofline_other_rollout_kernel"]["avg_ms"], d["roofline"]["avg_ms"], o["k_target_fused"], o["k_critic_fused"], o["k_wgrad_multi"], o.get("k_step_store_reset (env)", 0)))'
S='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("   side ms/step %.4f" % d["ms_per_step"], {k: round(v["ms_per_step"], 4) for k, v in d["kernel_groups_ms_per_step"].items()})'
run() {  # $1 fused flags, $2 mlp flags, $3 env flags
  MPG_FUSED_CFLAGS="$1" MPG_MLP_CFLAGS="$2" MPG_ENV_CFLAGS="$3" python3 -m mpg_amd.build --split-only > /tmp/build.log 2>&1 || { echo BUILD FAILED; tail -3 /tmp/build.log; return; }
  echo "== fused[$1] mlp[$2] env[$3]"
  python3 bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-side-configs 2>/dev/null | python3 -c "$P"
  for c in c4 c3; do python3 bench.py --config $c --no-cpu-baseline --steps 30 2>/dev/null | python3 -c "$S"; done
}
run "" "" ""
run "-DMPG_TR_IMAGE" "-DMPG_TR_IMAGE" "-DMPG_TR_IMAGE"
run "" "" ""
run "-DMPG_TR_IMAGE" "-DMPG_TR_IMAGE" "-DMPG_TR_IMAGE"
python3 -m mpg_amd.build --split-only > /tmp/build.log 2>&1
