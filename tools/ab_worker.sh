# same-box A/B of the bench step: worker.sample's policy pass and env step as ONE launch (mpg_worker_step, default) against two
# (-DMPG_AB_NO_WORKER_FUSION: rounds 1 - 3)
cd $GRAFT_REPO_ROOT
. tools/ab_scaffold.sh      # the experiment branches live in archive/proto/ablation_macros.patch since round 6
export MPG_BENCH_NO_F32=1
for V in "-DMPG_AB_NO_WORKER_FUSION" "" "-DMPG_AB_NO_WORKER_FUSION" ""; do
  echo "== [$V]"; MPG_EXTRA_CFLAGS="$V" python3 -m mpg_amd.build --split-only > /tmp/b.log 2>&1 || tail -3 /tmp/b.log
  python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms/step %.4f median %.4f' % (d['ms_per_step'], d['step_ms_median']), {k:(round(v,4) if v else v) for k,v in d.get('other_kernels_avg_ms',{}).items() if 'worker' in k or 'env' in k})"
done
python3 -m mpg_amd.build --split-only > /tmp/b.log 2>&1
