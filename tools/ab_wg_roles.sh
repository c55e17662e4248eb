# same-box A/B of the bench step: k_wgrad_multi with the thin pieces in their own workgroups (-DMPG_WG_ROLES) against the tail form (default)
cd $GRAFT_REPO_ROOT
export MPG_BENCH_NO_F32=1
for V in "" "-DMPG_WG_ROLES" "" "-DMPG_WG_ROLES"; do
  echo "== [$V]"; MPG_EXTRA_CFLAGS="$V" python3 -m mpg_amd.build --split-only > /tmp/b.log 2>&1 || tail -3 /tmp/b.log
  python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms/step %.4f median %.4f' % (d['ms_per_step'], d['step_ms_median']), {k:round(v,4) for k,v in d.get('other_kernels_avg_ms',{}).items() if 'wgrad' in k or 'critic' in k})"
done
python3 -m mpg_amd.build --split-only > /tmp/b.log 2>&1
