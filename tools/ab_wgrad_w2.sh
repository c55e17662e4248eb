# same-box A/B: the dW2-only weight-gradient launch with 64-column slices (default) against 32-column slices (-DMPG_AB_NO_WGRAD_W2)
cd $GRAFT_REPO_ROOT
. tools/ab_scaffold.sh      # the experiment branches live in archive/proto/ablation_macros.patch since round 6
export MPG_BENCH_NO_F32=1
for V in "-DMPG_AB_NO_WGRAD_W2" "" "-DMPG_WGRAD_MAX_CHUNKS_W2=64" "-DMPG_AB_NO_WGRAD_W2" ""; do
  echo "== [$V]"; MPG_EXTRA_CFLAGS="$V" python3 -m mpg_amd.build --split-only > /tmp/b.log 2>&1 || tail -3 /tmp/b.log
  for c in c4 c3; do python bench.py --config $c --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c ms/step %.4f' % d['ms_per_step'], {k:(round(v['avg_ms'],4), v['launches_per_step']) for k,v in d.get('kernel_groups_ms_per_step',{}).items() if 'wgrad' in k})"; done
done
python3 -m mpg_amd.build --split-only > /tmp/b.log 2>&1
