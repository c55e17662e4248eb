# same-box A/B of the bench step: k_wgrad_multi with two kinds of workgroup - dW2 in 64-column slices + the thin pieces apart (default) -
# against every workgroup doing a 32-column slice of both with the thin pieces as a tail (-DMPG_AB_WG_ONE_ROLE: rounds 2 - 3)
cd $GRAFT_REPO_ROOT
. tools/ab_scaffold.sh      # the experiment branches live in archive/proto/ablation_macros.patch since round 6
export MPG_BENCH_NO_F32=1
for V in "-DMPG_AB_WG_ONE_ROLE" "" "-DMPG_AB_WG_ONE_ROLE" ""; do
  echo "== [$V]"; MPG_EXTRA_CFLAGS="$V" python3 -m mpg_amd.build --split-only > /tmp/b.log 2>&1 || tail -3 /tmp/b.log
  python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms/step %.4f median %.4f' % (d['ms_per_step'], d['step_ms_median']), {k:round(v,4) for k,v in d.get('other_kernels_avg_ms',{}).items() if v and ('wgrad' in k or 'critic' in k)})"
done
python3 -m mpg_amd.build --split-only > /tmp/b.log 2>&1
