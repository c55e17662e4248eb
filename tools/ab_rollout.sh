#!/bin/bash
# A/B ablation of the rollout kernels: rebuild with -DMPG_AB_NOMFMA (one k-block of the hidden layer instead of 16:
# what is left is everything that is NOT the MFMA block), run bench, print the kernel averages
cd $GRAFT_REPO_ROOT
for V in "" "-DMPG_AB_NOMFMA"; do
  MPG_EXTRA_CFLAGS="$V" python -m mpg_amd.build > /tmp/build.log 2>&1 || { tail -5 /tmp/build.log; exit 1; }
  echo "== variant [$V]"
  python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms/step %.4f %s %.4f %s %.4f'%(d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_ms'], d['roofline_other_rollout_kernel']['kernel'], d['roofline_other_rollout_kernel']['avg_ms']))"
done
MPG_EXTRA_CFLAGS="" python -m mpg_amd.build > /tmp/build.log 2>&1
