# same-box A/B of the bench step: packed weight images with the eight waves' 1 KiB blocks interleaved (-DMPG_IMG_INTERLEAVE, mlp_core.h
# img_slot) against one contiguous 32 KB run per wave (default).  Round 4: interleaved is slower (0.2305 -> 0.2423 ms).
cd $GRAFT_REPO_ROOT
. tools/ab_scaffold.sh      # the experiment branches live in archive/proto/ablation_macros.patch since round 6
export MPG_BENCH_NO_F32=1
for V in ${VARIANTS:-"" "-DMPG_IMG_LAYOUT=5" "-DMPG_IMG_LAYOUT=4" "-DMPG_IMG_LAYOUT=3" "-DMPG_IMG_LAYOUT=2" "-DMPG_IMG_LAYOUT=1" ""}; do
  [ "$V" = "-" ] && V=""
  echo "== [$V]"; MPG_EXTRA_CFLAGS="$V" python3 -m mpg_amd.build --split-only > /tmp/b.log 2>&1 || tail -3 /tmp/b.log
  python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms/step %.4f median %.4f' % (d['ms_per_step'], d['step_ms_median']), 'bwd %.4f fwd %.4f' % (d['roofline']['avg_ms'], d['roofline_other_rollout_kernel']['avg_ms']), {k:round(v,4) for k,v in d.get('other_kernels_avg_ms',{}).items() if v})"
done
python3 -m mpg_amd.build --split-only > /tmp/b.log 2>&1
