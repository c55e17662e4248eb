#!/bin/bash
cd $GRAFT_REPO_ROOT
for V in "" "-DMPG_AB_WG_NOTHIN" "-DMPG_AB_WG_NOMFMA" "-DMPG_AB_WG_NOTHIN -DMPG_AB_WG_NOMFMA"; do
  MPG_EXTRA_CFLAGS="$V" python -m mpg_amd.build > /tmp/build.log 2>&1 || { tail -5 /tmp/build.log; exit 1; }
  echo "== variant [$V]"
  python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms/step %.4f wgrad %.4f'%(d['ms_per_step'], d['wgrad_kernel']['avg_ms']))"
done
