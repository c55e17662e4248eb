#!/bin/bash
# Reproducer of the lost packed-FMA products of the weight-gradient kernel's thin part, on the GPU box:
#   bash tools/pk_anomaly.sh [launches=5000]
# Builds three variants of k_wgrad_multi and prints the rate of launches whose gradient differs from the first launch's:
#   shipped          every multiply-add of the thin block one hand-written v_fmac_f32 / v_add_f32
#   MPG_AB_PKFMA     the plain C form the compiler may pack (v_pk_fma_f32 / v_pk_add_f32 with op_sel) - the failing one of round 2
#   + _WAIT          hypothesis: s_waitcnt lgkmcnt(0) + s_nop 4 between the staged LDS reads and the first packed consumer
# and leaves the thin block's ISA of the last two (and their diff) in gpurun_out/pk_anomaly/.
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
. tools/ab_scaffold.sh      # the experiment branches live in archive/proto/ablation_macros.patch since round 6
N=${1:-5000}
OUT=gpurun_out/pk_anomaly
mkdir -p $OUT
isa() {   # device assembly of fused_kernels.hip with the given extra flags -> $2, and the count of packed fp32 instructions
  MPG_EXTRA_CFLAGS="$1" python3 tools/asm.py fused_kernels.hip > /dev/null 2>&1
  awk '/^_ZN3mlp.*k_wgrad_multiILi8.*:/{f=1} f{print} f&&/^.Lfunc_end/{exit}' scratch_asm/fused_kernels.hip.s > $2
  echo "   k_wgrad_multi: $(grep -c 'v_pk_fma_f32' $2) v_pk_fma_f32, $(grep -c 'v_pk_add_f32' $2) v_pk_add_f32, $(grep -c 'v_pk_mul_f32' $2) v_pk_mul_f32, $(grep -c 'op_sel' $2) op_sel forms"
}
for V in "" "-DMPG_AB_PKFMA" "-DMPG_AB_PKFMA -DMPG_AB_PKFMA_WAIT"; do
  echo "== [${V:-shipped}]"
  MPG_EXTRA_CFLAGS="$V" python3 -m mpg_amd.build > /tmp/build.log 2>&1 || { echo "BUILD FAILED"; tail -5 /tmp/build.log; continue; }
  T=$(echo "${V:-shipped}" | tr -d ' -' | tr 'A-Z' 'a-z')
  isa "$V" $OUT/wgrad_multi_$T.s
  if [ -z "$V" ]; then PK_SAVE_REF=1 python3 tools/pk_repeat.py $N 2>&1 | grep -v amdgpu.ids | tail -8; else python3 tools/pk_repeat.py $N 2>&1 | grep -v amdgpu.ids | tail -8; fi
done
diff $OUT/wgrad_multi_dmpg_ab_pkfma.s $OUT/wgrad_multi_shipped.s > $OUT/pkfma_vs_shipped.diff
wc -l $OUT/pkfma_vs_shipped.diff
python3 -m mpg_amd.build > /tmp/build.log 2>&1
