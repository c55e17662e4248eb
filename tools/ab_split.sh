#!/bin/bash
# split target launch (default) against the single launch (MPG_TARGET_SPLIT=0): parity tests with both, alternating bench runs
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
for v in 1 0; do echo "MPG_TARGET_SPLIT=$v"; MPG_TARGET_SPLIT=$v timeout 400 python -m pytest tests/test_learner_gpu.py tests/test_dist_gpu.py -x -q -m gpu -k "golden or native_step or bench_size or trained or repeated or replicas" 2>&1 | tail -2; done
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["ms_per_step"],4), "target %.1f critic %.1f us" % (d["other_kernels_avg_ms"]["k_target_fused"]*1e3, d["other_kernels_avg_ms"]["k_critic_fused"]*1e3))'
for i in 1 2 3; do
MPG_TARGET_SPLIT=1 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python3 -c "$P" split
MPG_TARGET_SPLIT=0 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python3 -c "$P" single
done
