#!/bin/bash
# the critics' weight-gradient jobs right behind the critic launch (-DMPG_AB_WGRAD_EARLY) against the shipped order: parity tests
# with the variant, then alternating bench runs.  Leaves the tree built with the shipped flags.   bash tools/ab_early.sh
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
. tools/ab_scaffold.sh      # the experiment branches live in archive/proto/ablation_macros.patch since round 6
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["ms_per_step"],4))'
MPG_EXTRA_CFLAGS=-DMPG_AB_WGRAD_EARLY python3 -m mpg_amd.build > /dev/null 2>&1
timeout 300 python -m pytest tests/test_learner_gpu.py -x -q -m gpu -k "golden or native_step or bench_size" 2>&1 | tail -2
for i in 1 2; do python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python3 -c "$P" early; done
python3 -m mpg_amd.build > /dev/null 2>&1
for i in 1 2; do python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python3 -c "$P" shipped; done
