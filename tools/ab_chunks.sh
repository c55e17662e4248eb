cd $GRAFT_REPO_ROOT
export MPG_BENCH_NO_F32=1
for V in "" "-DMPG_WGRAD_MAX_CHUNKS_SINGLE=32" "-DMPG_WGRAD_MAX_CHUNKS_SINGLE=128"; do
  echo "== [$V]"; MPG_EXTRA_CFLAGS="$V" python3 -m mpg_amd.build --split-only > /tmp/b.log 2>&1 || tail -3 /tmp/b.log
  for c in c3 c4; do python bench.py --config $c --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c ms/step %.4f' % d['ms_per_step'], {k:(round(v['ms_per_step'],4), v['launches_per_step']) for k,v in d.get('kernel_groups_ms_per_step',{}).items() if 'wgrad' in k})"; done
done
python3 -m mpg_amd.build --split-only > /tmp/b.log 2>&1
