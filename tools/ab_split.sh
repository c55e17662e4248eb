#!/bin/bash
# split target launch (shipped) against the single launch (-DMPG_AB_TARGET_SINGLE): parity tests with both, bench runs of both.
# Leaves the tree built with the shipped flags.   bash tools/ab_split.sh
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
. tools/ab_scaffold.sh      # the experiment branches live in archive/proto/ablation_macros.patch since round 6
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["ms_per_step"],4), "target %.1f critic %.1f us" % (d["other_kernels_avg_ms"]["k_target_fused"]*1e3, d["other_kernels_avg_ms"]["k_critic_fused"]*1e3))'
for v in "-DMPG_AB_TARGET_SINGLE" ""; do
  echo "== [$v]"
  MPG_EXTRA_CFLAGS="$v" python3 -m mpg_amd.build > /dev/null 2>&1
  timeout 400 python -m pytest tests/test_learner_gpu.py tests/test_dist_gpu.py -x -q -m gpu -k "golden or native_step or bench_size or trained or repeated or replicas" 2>&1 | tail -1
  for i in 1 2 3; do python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python3 -c "$P" "${v:-split}"; done
done
python3 -m mpg_amd.build > /dev/null 2>&1
