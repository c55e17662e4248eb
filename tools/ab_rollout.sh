#!/bin/bash
# A/B ablations of the rollout kernels: rebuild with a macro, run bench, print kernel averages
cd $GRAFT_REPO_ROOT
for V in "" "-DMPG_AB_NODYN" "-DMPG_AB_NOMFMA" "-DMPG_AB_NODYN -DMPG_AB_NOMFMA"; do
  MPG_EXTRA_CFLAGS="$V" python -m mpg_amd.build > /tmp/build.log 2>&1 || { tail -5 /tmp/build.log; exit 1; }
  echo "== variant [$V]"
  python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms/step %.4f fwd %.4f bwd %.4f'%(d['ms_per_step'], d['roofline']['avg_ms'], d['roofline_bwd']['avg_ms']))"
done
