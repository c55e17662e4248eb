#!/bin/bash
# A/B of build variants on the GPU box (through gpurun):  bash tools/ab.sh "<extra cflags A>" "<extra cflags B>" ...
# Each variant is built with MPG_EXTRA_CFLAGS (fwd/bwd per-file flags can be given as FWD=... / BWD=... prefixes:
# "FWD=-mllvm x BWD= -DFOO"), benched with the default workload, and its per-kernel averages are printed.  The baseline
# ("") is run first and last.  Leaves the tree built with the shipped flags.
# Ablation / variant macros the kernels understand (timing or diagnosis only - several give wrong numbers on purpose):
#   -DMPG_F32_MFMA            exact-fp32 engine (v_mfma_f32_16x16x4_f32) instead of split-fp16        [correct results]
#   -DMPG_AB_NOMFMA           hidden-layer matrix block reduced to one k-block          -DMPG_AB_NO_ELU8   element-wise ELU
#   -DMPG_AB_NO_DPPASM        compiler-lowered DPP reduction instead of v_add_f32_dpp   -DMPG_AB_NO_IMGWRITE  LDS image stores dropped
#   -DMPG_AB_NO_H1            forward sweep stashes h1 of step 0 only     -DMPG_AB_IMAGE_FIRST  image requested before the small pieces
#   -DMPG_AB_NO_PREDRAW       the env launch does not pre-gather the minibatch window   -DMPG_AB_NODYN        (build.py) no dynamic LDS promotion flag
#   -DMPG_AB_WG_NOMFMA / -DMPG_AB_WG_NOTHIN   weight gradients without the matrix loop / without the thin part
#   -DMPG_AB_PKFMA [-DMPG_AB_PKFMA_WAIT]      the packed-FMA form of the thin block that loses products (tools/pk_anomaly.sh)
#   -DMPG_AB_TARGET_SINGLE                    the target kernel as one launch over all three networks (tools/ab_split.sh)
#   -DMPG_AB_WGRAD_EARLY                      the critics' weight-gradient jobs ahead of the reverse sweep (tools/ab_early.sh)
#   -DMPG_AB_NO_PRIO                           the sweeps without the static issue priority of waves 4..7
#   -DMPG_AB_BWD_KERNARG                       reverse sweep with run-time indices into the kernel arguments (rho[t], sel[ks]) as before
#   -DMPG_STAMP / -DMPG_TIMELINE              per-phase cycle stamps (tools/stamp.sh, tools/timeline.sh)
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
. tools/ab_scaffold.sh      # the experiment branches live in archive/proto/ablation_macros.patch since round 6
STEPS=${STEPS:-400}
run() {
  local spec="$1" fwd bwd extra
  unset MPG_FWD_CFLAGS MPG_BWD_CFLAGS
  extra="$spec"
  if [[ "$spec" == FWD=* ]]; then fwd="${spec#FWD=}"; fwd="${fwd%% BWD=*}"; extra="${spec#* BWD=}"; bwd="${extra%% EXTRA=*}"; extra="${extra#* EXTRA=}"; [[ "$extra" == "$bwd" ]] && extra=""; export MPG_FWD_CFLAGS="$fwd" MPG_BWD_CFLAGS="$bwd"; fi
  MPG_EXTRA_CFLAGS="$extra" python3 -m mpg_amd.build --split-only > /tmp/build.log 2>&1 || { echo "BUILD FAILED"; tail -5 /tmp/build.log; return; }
  MPG_EXTRA_CFLAGS="$extra" MPG_BENCH_NO_F32=1 python3 bench.py --steps $STEPS --warmup 30 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); o=d['other_kernels_avg_ms']
r={d['roofline']['kernel'][:13]:d['roofline']['avg_ms'], d['roofline_other_rollout_kernel']['kernel'][:13]:d['roofline_other_rollout_kernel']['avg_ms']}
print('ms/step %.4f (median %.4f) fwd %.4f bwd %.4f target %.4f critic %.4f wgrad %.4f pol %.4f env %.4f' % (d['ms_per_step'], d['step_ms_median'], r['k_rollout_fwd'], r['k_rollout_bwd'], o['k_target_fused'], o['k_critic_fused'], o['k_wgrad_multi'], o['k_forward (worker policy)'], o['k_step_store_reset (env)']))"
}
echo "== [baseline]"; run ""
for V in "$@"; do echo "== [$V]"; run "$V"; done
echo "== [baseline]"; run ""
unset MPG_FWD_CFLAGS MPG_BWD_CFLAGS
python3 -m mpg_amd.build --split-only > /tmp/build.log 2>&1
