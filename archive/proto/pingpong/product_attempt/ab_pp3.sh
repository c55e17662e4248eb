cd $GRAFT_REPO_ROOT
for V in "-DMPG_PP_MIN_GROUPS_PER_WG=1000000" "-DMPG_PP_MIN_GROUPS_PER_WG=1000000 -DMPG_TR_IMAGE"; do
  echo "== [$V]"; MPG_FWD_CFLAGS="-mllvm -amdgpu-sched-strategy=max-memory-clause" MPG_EXTRA_CFLAGS="$V" python3 -m mpg_amd.build --split-only > /tmp/b.log 2>&1 || tail -3 /tmp/b.log; python3 tools/pp_bench.py 2>/dev/null | grep -E "65536|131072"
  MPG_BENCH_NO_F32=1 python bench.py --config c4 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c4 ms/step', d['ms_per_step'], {k:(round(v['ms_per_step'],4), v['launches_per_step']) for k,v in d.get('kernel_groups_ms_per_step',{}).items()})"
done
python3 -m mpg_amd.build --split-only > /tmp/b.log 2>&1
