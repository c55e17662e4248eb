#!/bin/bash
cd $GRAFT_REPO_ROOT
for A in gfx950 "gfx950:xnack-" gfx950 "gfx950:xnack-"; do
  rm -rf mpg_amd/build mpg_amd/libmpg_hip.so
  MPG_ARCH="$A" python -m mpg_amd.build > /tmp/build.log 2>&1 || { tail -5 /tmp/build.log; exit 1; }
  echo "== arch [$A]"
  python bench.py --steps 400 --warmup 30 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms/step %.4f %s %.4f %s %.4f wgrad %.4f'%(d['ms_per_step'], d['roofline']['kernel'][:13], d['roofline']['avg_ms'], d['roofline_other_rollout_kernel']['kernel'][:13], d['roofline_other_rollout_kernel']['avg_ms'], d['wgrad_kernel']['avg_ms']))"
done
