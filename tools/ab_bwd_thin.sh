# same-box A/B: thin parameter gradients inside k_backward (default) against the weight-gradient launch (-DMPG_AB_NO_BWD_THIN)
cd $GRAFT_REPO_ROOT
. tools/ab_scaffold.sh      # the experiment branches live in archive/proto/ablation_macros.patch since round 6
export MPG_BENCH_NO_F32=1
for V in "-DMPG_AB_NO_BWD_THIN" "" "-DMPG_AB_NO_BWD_THIN" ""; do
  echo "== [$V]"; MPG_EXTRA_CFLAGS="$V" python3 -m mpg_amd.build --split-only > /tmp/b.log 2>&1 || tail -3 /tmp/b.log
  for c in c4 c3; do python bench.py --config $c --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c ms/step %.4f' % d['ms_per_step'], {k:(round(v['avg_ms'],4), v['launches_per_step']) for k,v in d.get('kernel_groups_ms_per_step',{}).items()})"; done
done
python3 -m mpg_amd.build --split-only > /tmp/b.log 2>&1
